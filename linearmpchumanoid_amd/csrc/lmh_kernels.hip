// lmh_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the batched NAO controller.
//
// ONE WORKGROUP OWNS ONE ROBOT INSTANCE.  All per-instance working data (kinematic tree, spatial inertias, mass
// matrix, Jacobians, QP blocks) lives in ~40 KB of LDS (4 instances per CU).  The fused rollout and the plain
// evaluation kernel give a robot TWO waves (128 threads) that split the independent pieces of an evaluation and
// join at workgroup barriers (see bsync / controller_eval); the debug, IK and model kernels run one wave.  Inside
// a wave the "barriers" between LDS phases are wave-scope fences.
// HBM is touched only for the 768-B state record in, the 640-B result out, the optional 288-B log per
// tick and the shared read-only tables (model, MPC gain row, ZMP window) which stay L2-resident.
//
// What is evaluated (reference file:line each block follows is cited inline):
//   Robot::updateState  -> FK, CoM, parent-relative Pluecker transforms    src/Robot.cpp
//   Dynamics::computeAll-> C, Cg, M (CRBA), AG, AGpqp, Jdot*qdot           src/Dynamics.cpp
//   Kinematics::computeAll -> feet Jacobian                                src/invKinematics.cpp
//   Mpc3dLip::compute   -> u0 = -K (Px x_k - zmp[k..k+N])                  src/mpcLinearPendulum.cpp
//   Controller::WBC     -> PD references, QP, torques                      src/controller.cpp
//   rk4Step(dynamics)   -> closed loop                                     rk4.hpp, apps/offline/main.cpp
//
// The QP (74 variables, 18 equalities, 32 bounds) is solved EXACTLY but not densely:
//   * H_aa = D + U' Om U (diagonal + rank 15..18)  -> Woodbury with a 15x15 (18x18) SPD solve,
//   * the 6 floating-base rows are eliminated through the 6x6 Schur complement S,
//   * w = G c is substituted, leaving a 32-variable bound-constrained strictly convex QP
//     min 1/2 c'(G'WG + eps I)c - (G'h)'c, c >= 0: a 12x12 push-through solve when every coefficient
//     is free, otherwise block principal pivoting from the previous active set with a Lawson-Hanson
//     pass as the finite fall-back.
//   * every SPD solve is a register-resident LDL' (row per lane; pivot broadcast by 64-bit DPP row_newbcast for
//     N <= 16, v_readlane for N = 32); the small dense products of the set-up run as v_mfma_f64_16x16x4_f64 tiles.
// The minimiser is unique (H is SPD), so this equals what qpOASES returns in the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "lmh_device.h"

// The parameter block is read through the CONSTANT address space: a wave-uniform load from it is a scalar-cache s_load into SGPRs
// (a generic pointer makes every field read a vector flat_load followed by a full s_waitcnt).  The block is written by the host
// before the launch and never by a kernel, which is what that address space promises.
typedef const __attribute__((address_space(4))) LmhDevParams LmhCParams;
// by-value kernel argument at offset 0 of the kernarg segment, seen through the constant address space
#define LMH_KERNARG_PARAMS() (*(LmhCParams *)__builtin_amdgcn_kernarg_segment_ptr())
#include "../../include/lmh.h"

// Wave-level fence: orders this wave's LDS traffic for the compiler; the LDS unit executes one wave's
// instructions in issue order, so no s_waitcnt / s_barrier is needed between a lane's store and another
// lane's load of the same wave.
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// The fused rollout kernel gives every robot TWO waves (NW = 2) that share its LDS: independent pieces of one
// evaluation (Newton-Euler | CRBA, the two reference chains, the MFMA tiles of the QP set-up) run side by side
// and join at workgroup barriers; the sequential factorisations stay on wave 0.  NW = 1 (the evaluation, IK and
// model kernels) degenerates to the single-wave schedule: the join is the wave fence.
#ifdef LMH_SUBSTAMPS
// diagnostic build: cycles each wave of the robot has spent inside workgroup barriers, in total (D_BWAIT) and split by the join's position
// inside the evaluation (D_JWAIT; D_JIDX is reset by the rollout loop).  The counters live in a hole of the robot's LDS image (see the map)
// so that the diagnostic build keeps the shipped kernel's four robots per CU; g_L points at the image.
__shared__ double *g_L;
#define D_BWAIT 2290
#define D_JIDX 2292
#define D_JWAIT 2294                 // [2][8]
#endif
template <int NW>
__device__ __forceinline__ void bsync()
{
    if constexpr (NW == 1) WSYNC();
    else {
#if defined(LMH_SUBSTAMPS) && !defined(LMH_DIAG_TL)               // (the production-timeline build keeps the plain barrier: its stamps bracket the joins)
        const long long t0 = clock64();
        __syncthreads();
        if ((threadIdx.x & 63u) == 0) {
            const int w_ = threadIdx.x >> 6;
            const long long d_ = clock64() - t0;
            double *L_ = g_L;
            L_[D_BWAIT + w_] += (double)d_;
            const int j_ = (int)L_[D_JIDX + w_];
            L_[D_JWAIT + 8 * w_ + ((j_ >= 0 && j_ < 7) ? j_ : 7)] += (double)d_;      // slot 7: everything outside the evaluations' seven joins
            L_[D_JIDX + w_] = (double)(j_ + 1);
        }
#else
        __syncthreads();
#endif
    }
}

// ------------------------------------------------------------------ constant tables
#define CPI2 6.123233995736766e-17   // libm's cos(-pi/2) for the reference's pi literal (theta[24], Robot.cpp:87)
// theta offsets, Robot.cpp:59-87, as multiples of the reference's pi literal
#define RPI 3.14159265358979323846
__constant__ double c_dh_off[24] = {0, (3.0 / 4) * RPI, 0, 0, 0, 0, -(1.0 / 2) * RPI, (1.0 / 4) * RPI, 0, 0, 0, 0,
                                    0, (1.0 / 2) * RPI, 0, 0, 0, 0, (1.0 / 2) * RPI, 0, 0, 0, 0, -(1.0 / 2) * RPI};
// Rf_q0_, Robot.cpp:28-31
__constant__ double c_rdes[9] = {0, 0, 1, 0, -1, 0, 1, 0, 0};


// ------------------------------------------------------------------ index maps (pure arithmetic:
// a per-lane __constant__ lookup is a global load, ~500 cycles of latency at one wave per SIMD)
// All branch-free (bit masks / compares folded into adds): select chains here turn into exec-mask code.
__device__ __forceinline__ int f_parent(int i) { return ((0x02108102u >> i) & 1u) ? 0 : i - 1; }   // Robot.cpp:165 (roots 1,8,15,20,25)
__device__ __forceinline__ int f_act(int i)                                                         // Robot.cpp:172
{
    const int a = i - (int)(i > 7) - (int)(i > 14);
    return ((0x08004081u >> i) & 1u) ? 0 : a;                       // frames 0, 7, 14, 27 carry no joint
}
__device__ __forceinline__ int f_jframe(int a) { return a + 1 + (int)(a >= 6) + (int)(a >= 12); }
__device__ __forceinline__ int f_jstart(int a) { return 6 * (int)(a >= 6) + 6 * (int)(a >= 12) + 5 * (int)(a >= 17) + 5 * (int)(a >= 22); }
__device__ __forceinline__ int f_jdepth(int a) { return a - f_jstart(a) + 1; }
__device__ __forceinline__ int f_body(int b) { return b + (int)(b >= 7) + (int)(b >= 13); }
__device__ __forceinline__ int f_chain_base(int c) { return 1 + 7 * (int)(c >= 1) + 7 * (int)(c >= 2) + 5 * (int)(c >= 3) + 5 * (int)(c >= 4); }
__device__ __forceinline__ int f_chain_len(int c) { return 7 - 2 * (int)(c >= 2) - 2 * (int)(c >= 4); }
__device__ __forceinline__ int f_chain(int c, int d) { return (d < f_chain_len(c)) ? f_chain_base(c) + d : -1; }
__device__ __forceinline__ int f_root(int r) { return 25 - 5 * (int)(r >= 1) - 5 * (int)(r >= 2) - 7 * (int)(r >= 3) - 7 * (int)(r >= 4); }  // head, LA, RA, LL, RL
// FK schedule (Robot.cpp:120-158): chain c, step s -> dst / src T slot, local-transform slot
__device__ __forceinline__ void fk_sched(int c, int s, int *dst, int *src, int *loc)
{
    const int base = f_chain_base(c);
    const bool leg = c < 2, arm = (c >= 2) && (c < 4);
    // legs: step 0 = T0*aux (slots 28/29), steps 1..6 joints, step 7 sole; arms: steps 0..4; head: steps 0..2
    const int d_leg = (s == 0) ? 28 + c : (s < 7) ? base + s - 1 : base + 6;
    const int s_leg = (s == 0) ? 0 : (s == 1) ? 28 + c : (s < 7) ? base + s - 2 : base + 5;
    const int l_leg = (s == 0) ? 25 + c : (s < 7) ? 6 * c + s - 1 : 27;
    const int d_oth = base + s, s_oth = (s == 0) ? 0 : base + s - 1;
    const int l_oth = arm ? 12 + 5 * (c - 2) + s : 22 + s;
    const bool on = leg || (arm && s < 5) || (c == 4 && s < 3);
    *dst = on ? (leg ? d_leg : d_oth) : -1;
    *src = leg ? s_leg : s_oth;
    *loc = leg ? l_leg : l_oth;
}
// Robot::desiredPosture (Robot.cpp:253-262)
__device__ __forceinline__ double qdes_of(int i)
{
    return (i == 0) ? -0.0185 : (i == 2) ? 0.282 : (i == 8 || i == 14) ? -0.5 : (i == 9 || i == 15) ? 0.8
         : (i == 10 || i == 16) ? -0.3 : (i == 18) ? 1.6 : (i == 23) ? -1.6 : 0.0;
}

// ------------------------------------------------------------------ LDS map (doubles)
enum {
    P_MODEL = 0,      // 28 x 14 (+ mass at 392)
    P_Q = 400, P_V = 430, P_VP = 460, P_TIME = 490,
    P_VHS = 496, P_VHN = 526, P_SC = 556,
    P_TB = 612,       // T0, T7, T14 (3 x 12)
    P_X0 = 648,       // E0(9) p0(3) B0(9)
    P_C = 672, P_CG = 702, P_AGPQP = 708, P_JPQP = 714, P_COM = 726, P_COMV = 729, P_ANGM = 732, P_MPC = 735,
    P_MTOP = 744, P_HL = 924, P_JC = 1068, P_AG = 1212,
    P_QREF = 1392, P_HREF = 1422, P_FREF = 1428, P_VFOOT = 1440,
    P_YT = 1452,      // 7 x 30 : (H^-1 [g | Mb'])' -- row n holds column n (the recovery reads YT[n][lane])
    P_SI = 1662, P_D6 = 1698, P_W = 1704, P_H12 = 1848, P_QV = 1860, P_CC = 1892, P_U12 = 1924,
    P_W12 = 1956, P_LAM6 = 1968, P_A = 1974,
    P_GCOL = 2004,    // friction-cone generators of ONE foot (16 x 6; both feet share vertices and rays)
    P_GI6 = 2100,     // (G_f G_f')^-1 (6x6)
    P_GPI = 2136,     // G_f' (G_f G_f')^-1 (16x6): min-norm coefficients of a foot wrench
    P_TAU = 2232, P_QDD = 2256,
    P_SCB = 2286,     // sin / cos of pitch and yaw (4), second buffer: the look-ahead kinematics of evaluation n + 1 write one buffer while the
                      // integrator of evaluation n still reads the other (P_SC + 52 is the first); [2290, 2310): counters of the diagnostic build
    P_POLY = 2310,    // foot polynomials: rF[3][8] | lF[3][8] | counts (6, stored as doubles)
    P_MPCK = 2366,    // MPC record K | Px0 | Px1 (3 (N+1) doubles) when N <= MPC_LDS_MAXN
    P_KI = 2504,      // K_f^-1 of the warm-start free set (2 x 6 x 6), prepared by the helper wave during the kinematics
    P_KF = 2576,      // [0] free set published by the last cone solve | [1] free set P_KI belongs to | [2] 1 = valid, 2 = singular, 0 = none
    P_PRE = 2580,     // per-evaluation references that depend on the clock only, prepared ahead of the kinematics:
                      // [0..1] sum K_i zmp_x/y[k+i] | [2..3] sum K_i Px0_i, sum K_i Px1_i (per launch) | [4..21] foot polynomial
                      // position / velocity / acceleration, 3 per (foot, axis)
    P_END = 2602,
    // ---- scratch, tree phases (kinematics, Newton-Euler, CRBA, Jacobian)
    S0 = P_END,
    A_LC = S0 + 0,    // 28 x 12 local transforms (dead once the chain products have loaded them: A_T takes their place)
    A_T = S0 + 0,     // 30 x 12 world transforms (dead after phase_com_x; the IK kernel reads them later).  S0 + [0, 384) is the one piece of
                      // scratch the cone solve of the shipped kernels never touches (C_WG is formed for the debug dump only), which is what lets
                      // the helper wave run the NEXT evaluation's kinematics there while wave 0 is still solving (lmh_rollout_kernel)
    A_T0S = S0 + 360, // 12 + 6: T0 and the sin / cos of roll, pitch, yaw while the chains start (look-ahead kinematics)
    A_VEL = S0 + 0, A_ACCG = S0 + 168, A_ACC0 = S0 + 336, A_FG = S0 + 504, A_F0 = S0 + 672,     // Newton-Euler: 28 x 6 each (wave 0)
    A_JL = S0 + 384,  // 2 x 6 x 12 feet Jacobian in sole axes (wave 0, after its Newton-Euler pass; clear of A_T, which the IK kernel reads after it)
    A_XF = S0 + 1200, // 28 x 36 : X_i = [A 0; B A] as full 6 x 6 images (row r contiguous, column r at stride 6)
    A_XP = S0 + 2208, // 28 x 3 : p_i
    A_XR = S0 + 2300, // 5 x 36 limb-root contributions to Ic_0 (CRBA, wave 1) | scratch of the helper wave's kinv_compute during the kinematics
    // ---- phase B (Woodbury + Schur)
    B_U = S0 + 0,     // 18 x 30
    B_K = S0 + 540,   // 25 x 19
    B_BP = S0 + 1016, // 30 x 7
    B_S = S0 + 1226,  // 12 x 7
    B_T1 = S0 + 1310, // 12 x 6
    B_OB = S0 + 1382, // Om*beta (18) | 1/Om (18) | beta (18)
    B_LS = S0 + 1436, // 18 x 19 rows of L
    B_CF = S0 + 1778, // 18 x 18 full symmetric copy of Cm
    // ---- phase B, NU = 15 (qp_setup15): every matrix-core operand is stored so that a lane's fragment is base + constant * k-step,
    // zero-padded to the tile shape (rows beyond the matrix point at Q_ZERO), so the tiles run without bounds selects
    // row strides are padded (34, 18, 17, 10 doubles) so that the 16 rows a fragment load touches fall into different LDS banks
    Q_U = S0 + 0,      // 16 x 34 : U rows in the order [J (12) ; AG_lin (3) ; zero row], columns 30, 31 zero (the Jacobian rows are written
                       //           early by wave 0, qp_prefill15: they and their U D^-1 image must stay inside S0 + [0, 408) and [544, 952))
    Q_UD = S0 + 544,   // 16 x 34 : U D^-1
    Q_BPT = S0 + 1088, // 8 x 34  : rows 0..6 = columns of bp' = [-qref | D^-1 Mb'], row 7 = -qref + D^-1 U' Om beta (V's g column)
    Q_TT = S0 + 1360,  // 8 x 18  : V' then t'' (row n = right-hand side n, k contiguous, [15] zero)
    Q_CM = S0 + 1504,  // 16 x 17 : Cm, full
    Q_Z = S0 + 1776,   // 6 x 18  : Mb D^-1 U'
    Q_MBP = S0 + 1884, // 6 x 8   : Mb bp'
    Q_S = S0 + 1932,   // 6 x 7
    Q_T1 = S0 + 1974,  // 12 x 10 (columns 6, 7 zero)
    Q_OB = S0 + 2094,  // Om*beta (16) | 1/Om (16) | beta (16)
    Q_LS = S0 + 2142,  // 6 x 7 scratch of the S^-1 step
    Q_ZERO = S0 + 2184, // 32 zeros
    Q_TRASH = S0 + 2216, // 64: where the lanes outside a tile's valid range store
    // ---- fp32 QP (LMH_PRECISION_FP32, phase_qp_f32): plain row-major arrays of float-valued slots
    F_U = S0 + 0,      // 18 x 30
    F_A = S0 + 540,    // 18 x 25 : [Cm | V] augmented system
    F_BP = S0 + 1016,  // 30 x 7  : bp' = [-qref | D^-1 Mb']
    F_BQ = S0 + 1226,  // 30 : -qref + D^-1 U' Om beta (V's g column)
    F_S = S0 + 1260,   // 6 x 12 : [S | I]
    F_OB = S0 + 1382,  // Om*beta (18) | 1/Om (18) | beta (18)
    F_T1 = S0 + 1436,  // 12 x 6
    F_K = S0 + 1520,   // 2 x 6 x 12 : [K_f | I] per foot
    F_KI = S0 + 1670,  // 2 x 36 : K_f^-1
    F_PT = S0 + 1750,  // 12 x 25 : [W + eps K^-1 | h | I]
    F_M = S0 + 2060,   // 12 x 12 : copy of the push-through matrix for the residual
    F_V = S0 + 2210,   // small vectors: w0 (12) | r (12) | w (12) | y (12)
    // ---- phase C (cone QP)
    C_WG = S0 + 0,    // 12 x 32
    C_P = S0 + 384,   // 32 x 33 (padded rows: conflict-free row-per-lane reads)
    C_LS = S0 + 1440, // 32 x 33 rows of L for the backward substitution
    C_IDX = S0 + 2496, // 32 ints: compact position -> coefficient index
#ifdef LMH_LDS_PROBE_DOUBLES
    LDS_DOUBLES = LMH_LDS_PROBE_DOUBLES   // occupancy probe builds only (scripts/occupancy_probe.py): such a library is queried, never launched
#else
    LDS_DOUBLES = S0 + 2512   // 40912 B + 4 B (ticket) per robot: four robots per CU fill 163.7 of the 163.84 KB
#endif
};

// -DLMH_POISON (experiment builds only): every robot starts from an LDS image full of NaNs, so that a read of a slot nobody wrote shows up
// as LMH_FLAG_NONFINITE deterministically instead of depending on what the previous kernel left behind
#ifdef LMH_POISON
#define LMH_POISON_LDS(L, n) do { for (int e_ = (int)threadIdx.x; e_ < (n); e_ += (int)blockDim.x) (L)[e_] = __longlong_as_double(0x7ff8000000000000ll); __syncthreads(); } while (0)
#else
#define LMH_POISON_LDS(L, n) do { } while (0)
#endif
// The lane id goes through an opaque (volatile) asm so that, in the rollout kernel, the large amount of
// lane-derived index arithmetic is NOT hoisted out of the tick/stage loops: hoisting it costs 256 VGPR +
// 256 AGPR + scratch spills (~100 MB of scratch traffic per launch) for no gain.
__device__ __forceinline__ int lane_opaque()
{
    int l = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(l));
    return l;
}
#define LANE lane_opaque()
#define MPC_LDS_MAXN 45            // 3 (N+1) <= 138 doubles of LDS
// diagnostic sub-phase stamps (s_memtime), compiled in only with -DLMH_SUBSTAMPS (LMH_DIAG=1 build): the
// shipped kernels execute no stamp.  Only the debug kernel points g_dbg at its dump buffer.
#ifdef LMH_SUBSTAMPS
__shared__ double *g_dbg;
__shared__ unsigned g_rt[6];          // cone-solve route counts of the robot's launch: all-free accepted | push-through | thin | general rounds | evaluations | all-free tried
#define RT_COUNT(i) do { if (LANE == 0) g_rt[i] += 1u; } while (0)
#ifdef LMH_DIAG_TL
__shared__ double *g_tl[2];
#endif
#define SUBSTAMP(i) do { if (g_dbg && LANE == 0) g_dbg[3900 + (i)] = (double)clock64(); } while (0)
#ifdef LMH_DIAG_TL
#define SET_GDBG(p) do { if (LANE == 0) { g_dbg = (p); g_tl[threadIdx.x >> 6] = nullptr; } } while (0)      // (every kernel starts with it)
#else
#define SET_GDBG(p) do { if (LANE == 0) g_dbg = (p); } while (0)
#endif
// per-wave timeline (two-wave debug kernel): wave w stamps slot 3700 + 100 w + i (scripts/gpu_wave_timeline.py)
#ifdef LMH_DIAG_TL
// ... of the PRODUCTION rollout kernel (scripts/diag.py ptimeline): the evaluation of Runge-Kutta stage LMH_DIAG_TL in the launch's last tick
// stamps into the log buffer, 256 doubles per robot (wave w: [100 w + i]); each wave switches its own pointer on and off
#define WSTAMP(i) do { double *p_ = g_tl[threadIdx.x >> 6]; if (p_ && LANE == 0) p_[100 * (int)(threadIdx.x >> 6) + (i)] = (double)clock64(); } while (0)
#else
#define WSTAMP(i) do { if (g_dbg && LANE == 0) g_dbg[3700 + 100 * (int)(threadIdx.x >> 6) + (i)] = (double)clock64(); } while (0)
#endif
#elif defined(LMH_PMARK)                 // static instruction census: the stamps become comments in the ISA listing (scripts/isa_census.py)
#define SUBSTAMP(i) do { } while (0)
#define SET_GDBG(p) do { } while (0)
#define WSTAMP(i) asm volatile("; PMARK " #i)
#define RT_COUNT(i) do { } while (0)
#else
#define RT_COUNT(i) do { } while (0)
#define SUBSTAMP(i) do { } while (0)
#define SET_GDBG(p) do { } while (0)
#define WSTAMP(i) do { } while (0)
#endif

// ---- LDS views with a working precision R.  The model-term phases (kinematics, Newton-Euler, CRBA, Jacobian) are
// written against LV<R>: element loads convert the stored fp64 value to R, stores convert back, so R = double is
// the plain fp64 path and R = float runs those phases in fp32 arithmetic ("mixed" precision: fp32 model terms,
// fp64 references and QP) without a second LDS layout.
template <typename R>
struct LRef {
    double *p;
    __device__ __forceinline__ operator R() const { return (R)*p; }
    __device__ __forceinline__ LRef &operator=(R v) { *p = (double)v; return *this; }
    __device__ __forceinline__ LRef &operator=(const LRef &o) { *p = (double)(R)*o.p; return *this; }
    __device__ __forceinline__ LRef &operator+=(R v) { *p = (double)((R)*p + v); return *this; }
};
template <typename R>
struct LV {
    double *p;
    __device__ __forceinline__ LV(double *q) : p(q) {}
    __device__ __forceinline__ LV(const double *q) : p(const_cast<double *>(q)) {}   // read-only use (tables in HBM / L2)
    __device__ __forceinline__ LV operator+(int o) const { return LV(p + o); }
    __device__ __forceinline__ LRef<R> operator[](int i) const { return LRef<R>{p + i}; }
};

// Lane exchange inside a 16-lane row through DPP (two 32-bit v_mov_dpp per double, ~10 cycles) instead of
// ds_bpermute (an LDS round trip per step).  CTRL: 0xB1 = quad_perm[1,0,3,2] (lane ^ 1), 0x4E = quad_perm[2,3,0,1]
// (lane ^ 2), 0x141 = row_half_mirror, 0x140 = row_mirror.  Needs a full exec mask (wave-uniform control flow).
template <int CTRL>
__device__ __forceinline__ double dpp_row(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_row(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double read_lane_f64(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// all-reduce over the 64 lanes: four DPP steps leave every lane with its row's total, the four row totals are
// combined through SGPRs in a fixed order (wave-uniform result)
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_row<0xB1>(v); v += dpp_row<0x4E>(v); v += dpp_row<0x141>(v); v += dpp_row<0x140>(v);
    return (read_lane_f64(v, 0) + read_lane_f64(v, 16)) + (read_lane_f64(v, 32) + read_lane_f64(v, 48));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_row<0xB1>(v); v += dpp_row<0x4E>(v); v += dpp_row<0x141>(v); v += dpp_row<0x140>(v);
    return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)))
         + (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}
__device__ __forceinline__ double wave_max(double v)
{
    v = fmax(v, dpp_row<0xB1>(v)); v = fmax(v, dpp_row<0x4E>(v)); v = fmax(v, dpp_row<0x141>(v)); v = fmax(v, dpp_row<0x140>(v));
    return fmax(fmax(read_lane_f64(v, 0), read_lane_f64(v, 16)), fmax(read_lane_f64(v, 32), read_lane_f64(v, 48)));
}

__device__ __forceinline__ double fast_rcp(double d)
{
    double y = __builtin_amdgcn_rcp(d);            // v_rcp_f64 (~2^-26) + two Newton steps -> full fp64
    y = fma(fma(-d, y, 1.0), y, y);
    y = fma(fma(-d, y, 1.0), y, y);
    return y;
}

// one Newton step: ~2 ulp.  For the multipliers of a Gauss-Jordan elimination that is as good as the exact quotient (the error is a 1e-16
// relative perturbation of the row operation; the eliminated column is never read again), and it takes two instructions off the dependent
// chain pivot -> reciprocal -> multiplier -> update of every pivot.
__device__ __forceinline__ double fast_rcp1(double d)
{
    const double y = __builtin_amdgcn_rcp(d);
    return fma(fma(-d, y, 1.0), y, y);
}

__device__ __forceinline__ double bcast_lane(double x, int l)     // l is a compile-time constant after unrolling
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

// ---- DPP row broadcast (gfx90a+: 64-bit DPP with row_newbcast): lane l of every 16-lane row reads lane C of
// its own row.  One instruction, no SGPR round trip (v_readlane needs two per double plus the VALU->SGPR hazard).
extern "C" __device__ double lmh_update_dpp_f64(double, double, int, int, int, bool) __asm("llvm.amdgcn.update.dpp.f64");
template <int C>
__device__ __forceinline__ double bcast16(double x)               // compiler-visible v_mov_b64_dpp (hazards handled by llc)
{
    return lmh_update_dpp_f64(x, x, 0x150 + C, 0xf, 0xf, true);
}
// acc_k += bcast16<C0 + k>(src) * m for k < K, as ONE asm block of v_fmac_f64_dpp: the leading s_nop covers the
// "VALU write -> DPP read" hazard on src for whatever the compiler placed before the block; inside the block
// nothing writes src.  Must run with a full exec mask (wave-uniform control flow only).
#define LMH_FD(k, s, m, c) "v_fmac_f64_dpp %" #k ", %" #s ", %" #m " row_newbcast:%" #c " row_mask:0xf bank_mask:0xf\n\t"
template <int A0, int B0, int K, int N>
__device__ __forceinline__ void dpp_fmac_cols(double (&a)[N], double src, double m)    // a[A0 + k] += bcast16<B0 + k>(src) * m
{
    static_assert(K == 1 || K == 2 || K == 4 || K == 8, "chunk size");
    constexpr int C0 = A0;
    if constexpr (K == 8)
        asm volatile("s_nop 1\n\t" LMH_FD(0, 8, 9, 10) LMH_FD(1, 8, 9, 11) LMH_FD(2, 8, 9, 12) LMH_FD(3, 8, 9, 13)
                     LMH_FD(4, 8, 9, 14) LMH_FD(5, 8, 9, 15) LMH_FD(6, 8, 9, 16) LMH_FD(7, 8, 9, 17)
                     : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]), "+v"(a[C0 + 3]), "+v"(a[C0 + 4]), "+v"(a[C0 + 5]), "+v"(a[C0 + 6]), "+v"(a[C0 + 7])
                     : "v"(src), "v"(m), "n"(B0), "n"(B0 + 1), "n"(B0 + 2), "n"(B0 + 3), "n"(B0 + 4), "n"(B0 + 5), "n"(B0 + 6), "n"(B0 + 7));
    else if constexpr (K == 4)
        asm volatile("s_nop 1\n\t" LMH_FD(0, 4, 5, 6) LMH_FD(1, 4, 5, 7) LMH_FD(2, 4, 5, 8) LMH_FD(3, 4, 5, 9)
                     : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]), "+v"(a[C0 + 3])
                     : "v"(src), "v"(m), "n"(B0), "n"(B0 + 1), "n"(B0 + 2), "n"(B0 + 3));
    else if constexpr (K == 2)
        asm volatile("s_nop 1\n\t" LMH_FD(0, 2, 3, 4) LMH_FD(1, 2, 3, 5)
                     : "+v"(a[C0]), "+v"(a[C0 + 1]) : "v"(src), "v"(m), "n"(B0), "n"(B0 + 1));
    else
        asm volatile("s_nop 1\n\t" LMH_FD(0, 1, 2, 3) : "+v"(a[C0]) : "v"(src), "v"(m), "n"(B0));
}
// a[c] += bcast16<c>(src) * m for c in [C0, N)
template <int C0, int N>
__device__ __forceinline__ void dpp_fmac_tail(double (&a)[N], double src, double m)
{
    constexpr int R = N - C0;
    if constexpr (R >= 8) { dpp_fmac_cols<C0, C0, 8>(a, src, m); dpp_fmac_tail<C0 + 8>(a, src, m); }
    else if constexpr (R >= 4) { dpp_fmac_cols<C0, C0, 4>(a, src, m); dpp_fmac_tail<C0 + 4>(a, src, m); }
    else if constexpr (R >= 2) { dpp_fmac_cols<C0, C0, 2>(a, src, m); dpp_fmac_tail<C0 + 2>(a, src, m); }
    else if constexpr (R == 1) { dpp_fmac_cols<C0, C0, 1>(a, src, m); }
}
// a[A0 + k] += bcast16<B0 + k>(src) * m for k < CNT
template <int A0, int B0, int CNT, int N>
__device__ __forceinline__ void dpp_fmac_range(double (&a)[N], double src, double m)
{
    if constexpr (CNT >= 8) { dpp_fmac_cols<A0, B0, 8>(a, src, m); dpp_fmac_range<A0 + 8, B0 + 8, CNT - 8>(a, src, m); }
    else if constexpr (CNT >= 4) { dpp_fmac_cols<A0, B0, 4>(a, src, m); dpp_fmac_range<A0 + 4, B0 + 4, CNT - 4>(a, src, m); }
    else if constexpr (CNT >= 2) { dpp_fmac_cols<A0, B0, 2>(a, src, m); dpp_fmac_range<A0 + 2, B0 + 2, CNT - 2>(a, src, m); }
    else if constexpr (CNT == 1) { dpp_fmac_cols<A0, B0, 1>(a, src, m); }
}
// acc += bcast16<J>(src) * m (acc and src may be the same register)
template <int J>
__device__ __forceinline__ void dpp_fmac_one(double &acc, double src, double m)
{
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(J));
}
// b[r] += bcast16<J>(b[r]) * m for r < M (each right-hand side broadcasts its own lane-J entry)
#define LMH_FS(k, m, c) "v_fmac_f64_dpp %" #k ", %" #k ", %" #m " row_newbcast:%" #c " row_mask:0xf bank_mask:0xf\n\t"
template <int J, int M>
__device__ __forceinline__ void dpp_fmac_rhs(double (&b)[M], double m)
{
    static_assert(M == 1 || M == 6 || M == 7, "right-hand-side counts in use");
    if constexpr (M == 1)
        asm volatile("s_nop 1\n\t" LMH_FS(0, 1, 2) : "+v"(b[0]) : "v"(m), "n"(J));
    else if constexpr (M == 6)
        asm volatile("s_nop 1\n\t" LMH_FS(0, 6, 7) LMH_FS(1, 6, 7) LMH_FS(2, 6, 7) LMH_FS(3, 6, 7) LMH_FS(4, 6, 7) LMH_FS(5, 6, 7)
                     : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]) : "v"(m), "n"(J));
    else
        asm volatile("s_nop 1\n\t" LMH_FS(0, 7, 8) LMH_FS(1, 7, 8) LMH_FS(2, 7, 8) LMH_FS(3, 7, 8) LMH_FS(4, 7, 8) LMH_FS(5, 7, 8) LMH_FS(6, 7, 8)
                     : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]) : "v"(m), "n"(J));
}

// One pivot of the row-per-lane LDL' for N <= 16 (all rows inside DPP row 0), then the next (compile-time recursion).
// `dadd`: a constant on the diagonal of the matrix, added where the pivot is read (the diagonal entry is touched nowhere else: lane J's own
// column entry is only ever used through this broadcast), so that the caller does not have to place it with a select per column.
template <int J, int N, int M>
__device__ __forceinline__ void ldl16_forward(double (&a)[N], double (&b)[M], unsigned live, int lane, int &bad, double &myinv, double dadd = 0.0)
{
    if constexpr (J < N) {
        if ((live >> J) & 1u) {                                   // wave-uniform
            // (DPP rows 1..3 hold no matrix rows: whatever they compute -- possibly non-finite -- stays in their lanes; `bad` is read from lane 0)
            const double d = bcast16<J>(a[J]) + dadd;
            if (!(d > 0.0)) bad = 1;
            const double invd = fast_rcp(d);
            const double f = a[J] * invd;                         // L_iJ in lanes i > J
            const double nfm = (lane > J) ? -f : 0.0;
            if (lane == J) myinv = invd;
            dpp_fmac_tail<J + 1>(a, a[J], -f);                    // a[c] -= f * (d_J L_cJ held by lane c)
            dpp_fmac_rhs<J>(b, nfm);                              // forward substitution
            a[J] = f;                                             // (rows <= J keep a don't-care there: only L_iJ, i > J, is read back)
        }
        ldl16_forward<J + 1>(a, b, live, lane, bad, myinv, dadd);
    }
}
template <int J, int N, int M>
__device__ __forceinline__ void ldl16_backward(double (&b)[M], unsigned live, int lane, const double *Ls)
{
    if constexpr (J > 0) {
        if ((live >> J) & 1u) {
            const double lv = Ls[J * (N + 1) + ((lane < N) ? lane : 0)];      // unconditional load (clamped), masked by value: no exec branch
            const double nl = (lane < J) ? -lv : 0.0;
            dpp_fmac_rhs<J>(b, nl);
        }
        ldl16_backward<J - 1, N>(b, live, lane, Ls);
    }
}

// ---- 16 < N <= 32, one right-hand side: TWO matrix rows per lane so that every row lives in DPP row 0 and the
// pivot broadcast is again a row_newbcast.  Lane i < 16 holds row i in a0[0..15] and row 16 + i in a1[0..N-1]
// (lower triangles), rhs entries b0 / b1.  Pivot J < 16: the scaled pivot column sits in a0[J] (rows < 16, lane c)
// and a1[J] (rows >= 16, lane c - 16); pivot J >= 16: in a1[J].
template <int J, int N2>
__device__ __forceinline__ void ldl2_forward(double (&a0)[16], double (&a1)[16 + N2], double &b0, double &b1, unsigned live, int lane,
                                             int &bad, double &inv0, double &inv1)
{
    constexpr int N = 16 + N2;
    if constexpr (J < N) {
        if ((live >> J) & 1u) {                                   // wave-uniform
            if constexpr (J < 16) {
                double d = bcast16<J>(a0[J]);
                d = (lane < 16) ? d : 1.0;
                if (!(d > 0.0)) bad = 1;
                const double invd = fast_rcp(d);
                const double f0 = a0[J] * invd, f1 = a1[J] * invd;     // L_iJ (rows < 16, valid for lane > J) | L_(16+i)J
                const double nfm0 = (lane > J) ? -f0 : 0.0;
                if (lane == J) inv0 = invd;
                dpp_fmac_range<J + 1, J + 1, 15 - J>(a0, a0[J], -f0);  // columns J+1..15: lane c holds d_J L_cJ in a0[J]
                dpp_fmac_range<J + 1, J + 1, 15 - J>(a1, a0[J], -f1);
                dpp_fmac_range<16, 0, N2>(a1, a1[J], -f1);             // columns 16..N-1: lane c - 16 holds d_J L_cJ in a1[J]
                dpp_fmac_one<J>(b1, b0, -f1);                          // forward substitution (lane J's b0 is z_J, untouched below)
                dpp_fmac_one<J>(b0, b0, nfm0);
                if (lane > J) a0[J] = f0;
                a1[J] = f1;
            } else {
                constexpr int Jp = J - 16;
                double d = bcast16<Jp>(a1[J]);
                d = (lane < 16) ? d : 1.0;
                if (!(d > 0.0)) bad = 1;
                const double invd = fast_rcp(d);
                const double f1 = a1[J] * invd;
                const double nfm1 = (lane > Jp) ? -f1 : 0.0;
                if (lane == Jp) inv1 = invd;
                dpp_fmac_range<J + 1, Jp + 1, N - 1 - J>(a1, a1[J], -f1);
                dpp_fmac_one<Jp>(b1, b1, nfm1);
                if (lane > Jp) a1[J] = f1;
            }
        }
        ldl2_forward<J + 1, N2>(a0, a1, b0, b1, live, lane, bad, inv0, inv1);
    }
}
template <int J, int N2>
__device__ __forceinline__ void ldl2_backward(double &b0, double &b1, unsigned live, int lane, const double *Ls)
{
    constexpr int N = 16 + N2;
    if constexpr (J > 0) {
        if ((live >> J) & 1u) {
            const int l16 = (lane < 16) ? lane : 0;                // lanes outside DPP row 0 read a valid address
            if constexpr (J >= 16) {
                constexpr int Jp = J - 16;
                const double l0 = (lane < 16) ? Ls[J * (N + 1) + l16] : 0.0;             // L[J][lane], rows < 16
                const double l1 = (lane < Jp) ? Ls[J * (N + 1) + 16 + l16] : 0.0;        // L[J][16 + lane], rows 16 .. J-1
                dpp_fmac_one<Jp>(b0, b1, -l0);
                dpp_fmac_one<Jp>(b1, b1, -l1);
            } else {
                const double l0 = (lane < J) ? Ls[J * (N + 1) + l16] : 0.0;
                dpp_fmac_one<J>(b0, b0, -l0);
            }
        }
        ldl2_backward<J - 1, N2>(b0, b1, live, lane, Ls);
    }
}
// On exit b0 of lane i < 16 holds x_i and b1 holds x_(16+i).  Returns non-zero (wave-uniform) if a pivot was not positive.
template <int N2>
__device__ __forceinline__ int ldl2_solve_regs(double (&a0)[16], double (&a1)[16 + N2], double &b0, double &b1, unsigned live, double *Ls)
{
    constexpr int N = 16 + N2;
    const int lane = LANE;
    int bad = 0;
    double inv0 = 0.0, inv1 = 0.0;
    ldl2_forward<0, N2>(a0, a1, b0, b1, live, lane, bad, inv0, inv1);
    bad = __builtin_amdgcn_readfirstlane(bad);
    b0 *= inv0; b1 *= inv1;                                       // w = D^-1 z
    WSYNC();
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 15; c++) Ls[lane * (N + 1) + c] = a0[c];                 // L[lane][c], c < lane
    }
    if (lane < N2) {
#pragma unroll
        for (int c = 0; c < N - 1; c++) Ls[(16 + lane) * (N + 1) + c] = a1[c];       // L[16 + lane][c], c < 16 + lane
    }
    WSYNC();
    ldl2_backward<N - 1, N2>(b0, b1, live, lane, Ls);
    return bad;
}

// Register-resident LDL' solve of an SPD system with M right-hand sides, N <= 32.
// Lane i < N holds row i of the matrix in a[] (entries a[c], c <= i, are used; rows / columns whose
// bit is clear in `live` must be zero and are skipped) and its rhs entries in b[].  The pivot column
// is broadcast lane to lane (DPP row broadcast for N <= 16, v_readlane above; no LDS round trip inside the
// factorisation); the rows of L are parked once in Ls (row stride N+1, conflict free) for the backward
// substitution.  On exit b[r] of lane i holds x_i.  Returns non-zero (wave-uniform) if a pivot was not positive.
template <int N, int M>
__device__ __forceinline__ int ldl_solve_regs(double (&a)[N], double (&b)[M], unsigned live, double *Ls, double dadd = 0.0)
{
    const int lane = LANE;
    int bad = 0;
    double myinv = 0.0;                                           // 1 / d_lane (0 on rows that are not live)
    if constexpr (N <= 16) {
        WSTAMP(40);
        ldl16_forward<0>(a, b, live, lane, bad, myinv, dadd);
        bad = __builtin_amdgcn_readfirstlane(bad);
#pragma unroll
        for (int r = 0; r < M; r++) b[r] *= myinv;                // w = D^-1 z
        WSTAMP(41);
        WSYNC();
        if (lane < N) {
#pragma unroll
            for (int c = 0; c < N - 1; c++) Ls[lane * (N + 1) + c] = a[c];          // L[lane][c], c < lane
        }
        WSYNC();
        WSTAMP(42);
        ldl16_backward<N - 1, N>(b, live, lane, Ls);
        WSTAMP(43);
        return bad;
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (!((live >> j) & 1u)) continue;                        // wave-uniform
        const double d = bcast_lane(a[j], j);
        if (!(d > 0.0)) bad = 1;
        const double invd = fast_rcp(d);
        const double f = a[j] * invd;                             // L_ij in lanes i > j
        const double fm = (lane > j) ? f : 0.0;
        if (lane == j) myinv = invd;
#pragma unroll
        for (int c = j + 1; c < N; c++) a[c] = fma(-f, bcast_lane(a[j], c), a[c]);   // lane c still holds d_j L_cj
#pragma unroll
        for (int r = 0; r < M; r++) b[r] = fma(-fm, bcast_lane(b[r], j), b[r]);      // forward substitution
        if (lane > j) a[j] = f;
    }
#pragma unroll
    for (int r = 0; r < M; r++) b[r] *= myinv;                    // w = D^-1 z
    WSYNC();
    if (lane < N) {
#pragma unroll
        for (int c = 0; c < N - 1; c++) Ls[lane * (N + 1) + c] = a[c];              // L[lane][c], c < lane
    }
    WSYNC();
#pragma unroll
    for (int j = N - 1; j > 0; j--) {
        if (!((live >> j) & 1u)) continue;
        const double lji = (lane < j) ? Ls[j * (N + 1) + lane] : 0.0;
#pragma unroll
        for (int r = 0; r < M; r++) b[r] = fma(-lji, bcast_lane(b[r], j), b[r]);
    }
    return bad;
}

// ---- Gauss-Jordan form of the register-resident SPD solve for N <= 16 (the well-conditioned systems of the QP set-up: Woodbury core,
// Schur complement, push-through system, K_f).  Lane i < N holds the FULL row i in a[]; pivot J eliminates column J from every other
// row, rows above the pivot included: a[c] += bcast16<J>(a[c]) * nf, b[r] += bcast16<J>(b[r]) * nf with nf = -a_iJ / d_J (0 on row J) --
// every register broadcasts its own lane-J entry, one v_fmac_f64_dpp each.  The instruction count per pivot equals the LDL' forward
// step's, and there is no backward substitution, no L parked in LDS and no fence: x_i = b_i / d_i at the end.  Without pivoting this is
// as accurate as LDL' on an SPD matrix (measured on the Woodbury core: 7e-14 both, condition 1e4).
#define LMH_GS8(J) LMH_FS(0, 8, 9) LMH_FS(1, 8, 9) LMH_FS(2, 8, 9) LMH_FS(3, 8, 9) LMH_FS(4, 8, 9) LMH_FS(5, 8, 9) LMH_FS(6, 8, 9) LMH_FS(7, 8, 9)
template <int C0, int CNT, int J, int N>
__device__ __forceinline__ void dpp_fmac_self(double (&a)[N], double m)         // a[c] += bcast16<J>(a[c]) * m for c in [C0, C0 + CNT)
{
    if constexpr (CNT >= 8) {
        asm volatile("s_nop 1\n\t" LMH_GS8(J)
                     : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]), "+v"(a[C0 + 3]), "+v"(a[C0 + 4]), "+v"(a[C0 + 5]), "+v"(a[C0 + 6]), "+v"(a[C0 + 7])
                     : "v"(m), "n"(J));
        dpp_fmac_self<C0 + 8, CNT - 8, J>(a, m);
    } else if constexpr (CNT == 7) {
        asm volatile("s_nop 1\n\t" LMH_FS(0, 7, 8) LMH_FS(1, 7, 8) LMH_FS(2, 7, 8) LMH_FS(3, 7, 8) LMH_FS(4, 7, 8) LMH_FS(5, 7, 8) LMH_FS(6, 7, 8)
                     : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]), "+v"(a[C0 + 3]), "+v"(a[C0 + 4]), "+v"(a[C0 + 5]), "+v"(a[C0 + 6]) : "v"(m), "n"(J));
    } else if constexpr (CNT == 6) {
        asm volatile("s_nop 1\n\t" LMH_FS(0, 6, 7) LMH_FS(1, 6, 7) LMH_FS(2, 6, 7) LMH_FS(3, 6, 7) LMH_FS(4, 6, 7) LMH_FS(5, 6, 7)
                     : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]), "+v"(a[C0 + 3]), "+v"(a[C0 + 4]), "+v"(a[C0 + 5]) : "v"(m), "n"(J));
    } else if constexpr (CNT == 5) {
        asm volatile("s_nop 1\n\t" LMH_FS(0, 5, 6) LMH_FS(1, 5, 6) LMH_FS(2, 5, 6) LMH_FS(3, 5, 6) LMH_FS(4, 5, 6)
                     : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]), "+v"(a[C0 + 3]), "+v"(a[C0 + 4]) : "v"(m), "n"(J));
    } else if constexpr (CNT == 4) {
        asm volatile("s_nop 1\n\t" LMH_FS(0, 4, 5) LMH_FS(1, 4, 5) LMH_FS(2, 4, 5) LMH_FS(3, 4, 5)
                     : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]), "+v"(a[C0 + 3]) : "v"(m), "n"(J));
    } else if constexpr (CNT == 3) {
        asm volatile("s_nop 1\n\t" LMH_FS(0, 3, 4) LMH_FS(1, 3, 4) LMH_FS(2, 3, 4) : "+v"(a[C0]), "+v"(a[C0 + 1]), "+v"(a[C0 + 2]) : "v"(m), "n"(J));
    } else if constexpr (CNT == 2) {
        asm volatile("s_nop 1\n\t" LMH_FS(0, 2, 3) LMH_FS(1, 2, 3) : "+v"(a[C0]), "+v"(a[C0 + 1]) : "v"(m), "n"(J));
    } else if constexpr (CNT == 1) {
        asm volatile("s_nop 1\n\t" LMH_FS(0, 1, 2) : "+v"(a[C0]) : "v"(m), "n"(J));
    }
}
#define LMH_DPP1(k) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf"
// NOP = false: `src` was not written by the instruction before (a DPP operand needs two wait states behind the VALU write of its register, and
// the compiler does not look inside the asm): the later members of a chain on the same `src`
template <int C, bool NOP = true>
__device__ __forceinline__ void dpp_fmac_lane(double &acc, double src, double m)     // acc += lane_C(src) * m   (C < 16, own 16-lane row)
{
    if constexpr (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(C));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(C));
}
// one pivot, then the next.  GUARD = true: `rowon` switches a whole 16-lane DPP row off (its pivots are replaced by 1) and a pivot that is
// not above `dmin` is replaced by 1 and reported in `bad` (kinv_compute: two feet on two DPP rows, a rank-deficient K_f is an expected
// outcome).  GUARD = false (gj_solve_regs): every DPP row carries a copy of the system, so the pivot a lane sees is always the true one --
// no guard selects, no test per pivot: lane J keeps 1 / d_J, and the caller looks at the signs once at the end.
template <int J, int N, int M, bool GUARD = true>
__device__ __forceinline__ void gj16_step(double (&a)[N], double (&b)[M], unsigned live, int l16, bool rowon, double dmin, int &bad, double &myinv)
{
    if constexpr (J < N) {
        if ((live >> J) & 1u) {                                   // wave-uniform
            double d = bcast16<J>(a[J]);
            if constexpr (GUARD) {
                if (rowon && !(d > dmin)) bad = 1;
                d = (rowon && d > dmin) ? d : 1.0;
            }
            const double invd = fast_rcp1(d);
            const bool piv = l16 == J;
            const double nf = piv ? 0.0 : -(a[J] * invd);
            myinv = piv ? invd : myinv;
            dpp_fmac_self<J + 1, N - 1 - J, J>(a, nf);
            dpp_fmac_self<0, M, J>(b, nf);
        }
        gj16_step<J + 1, N, M, GUARD>(a, b, live, l16, rowon, dmin, bad, myinv);
    }
}
// (-DLMH_GJ_PIPE only; measured and not shipped: wave 0's chain gets shorter, but the extra wait states it takes are issue slots the SIMD's
// other wave -- another robot's helper -- no longer gets, and the kernel's throughput follows the SIMD's total instruction count:
// 16.36 -> 16.00 M ticks/s on config 3.)
// The guard-free form, software-pipelined: the reciprocal of pivot J + 1 -- broadcast, v_rcp_f64, one Newton step: ~100 cycles of dependent
// latency, the longest link of a pivot's chain -- needs column J + 1 of pivot J's update only, so that column is updated by its own asm
// statement, the reciprocal chain starts, and the block of the other columns and the right-hand sides issues inside its latency
// (with all of a pivot's updates in one asm statement the chain could only start behind the block: a 15 x 15 solve with 7 right-hand
// sides spent 21..7 issue slots per pivot that way).  `invd`: 1 / d_J handed over by pivot J - 1 (`have`: it is; wave-uniform).
template <int J, int N, int M>
__device__ __forceinline__ void gj16_pipe(double (&a)[N], double (&b)[M], unsigned live, int l16, double &myinv, double invd, bool have)
{
    if constexpr (J < N) {
        bool have_next = false;
        double invn = 0.0;
        if ((live >> J) & 1u) {                                   // wave-uniform
            if (!have) invd = fast_rcp1(bcast16<J>(a[J]));         // (first live pivot, or the one before was not live)
            const bool piv = l16 == J;
            const double nf = piv ? 0.0 : -(a[J] * invd);
            myinv = piv ? invd : myinv;
            if constexpr (J + 1 < N) {
                // column J + 1, then straight away the broadcast of the next pivot and its v_rcp_f64 -- inside ONE asm statement: the compiler's
                // scheduler sees no latency in an asm and had moved a separately written reciprocal behind the block of the other columns
                double dn, rn;
                asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                             "v_mov_b64_dpp %1, %0 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\tv_rcp_f64_e32 %2, %1"
                             : "+v"(a[J + 1]), "=&v"(dn), "=&v"(rn) : "v"(nf), "n"(J), "n"(J + 1));
                dpp_fmac_self<J + 2, N - 2 - J, J>(a, nf);
                dpp_fmac_self<0, M, J>(b, nf);
                asm volatile("" : "+v"(dn), "+v"(rn));             // the Newton step stays behind the blocks
                invn = fma(fma(-dn, rn, 1.0), rn, rn);
                have_next = ((live >> (J + 1)) & 1u) != 0u;
            } else dpp_fmac_self<0, M, J>(b, nf);
        }
        gj16_pipe<J + 1, N, M>(a, b, live, l16, myinv, invn, have_next);
    }
}
// Lane l holds row l & 15 of the system (rows >= N: any finite copy, e.g. row 0 -- they are eliminated like every other row and never read):
// all four 16-lane DPP rows then run the same elimination.  On exit b[r] of lane i < N holds x_i.  Returns non-zero (wave-uniform) if a
// pivot was not positive (d_i > 0 <=> 0 < 1 / d_i < inf on the lane that kept it; the caller flags LMH_FLAG_NOT_SPD, and the non-finite
// values that follow a bad pivot are flagged LMH_FLAG_NONFINITE by the evaluation's own check).
template <int N, int M>
__device__ __forceinline__ int gj_solve_regs(double (&a)[N], double (&b)[M], unsigned live)
{
    static_assert(N <= 16, "one DPP row");
    const int l16 = LANE & 15;
    int bad = 0;
    double myinv = 0.0;
#ifdef LMH_GJ_PIPE                                                  // experiment switch: the software-pipelined pivots (gj16_pipe): measured -2.2 %
    gj16_pipe<0, N, M>(a, b, live, l16, myinv, 0.0, false);
#else
    gj16_step<0, N, M, false>(a, b, live, l16, true, 0.0, bad, myinv);
#endif
#pragma unroll
    for (int r = 0; r < M; r++) b[r] *= myinv;
    const bool pivot_lane = ((live >> l16) & 1u) != 0u;            // (bits >= N of `live` are clear)
    return (__ballot(pivot_lane && !(myinv > 0.0 && myinv <= 1.7976931348623157e308)) != 0ull) ? 1 : 0;
}


// ------------------------------------------------------------------ fp64 matrix-core tiles
// The small dense contractions of the QP set-up (K = 12..32) run on v_mfma_f64_16x16x4_f64: one
// 16 x 16 output tile per call, D = sum_k A[:,k] B[k,:].  Fragment maps (gfx950): lane l supplies
// A[l & 15][4 kk + (l >> 4)] and B[4 kk + (l >> 4)][l & 15]; it receives D[(l >> 4) + 4 reg][l & 15].
typedef double v4d __attribute__((ext_vector_type(4)));
template <int KSTEPS, class FA, class FB>
__device__ __forceinline__ v4d mfma_tile(FA a_of, FB b_of)
{
    const int r = LANE & 15, kq = LANE >> 4;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) {
        double bv;
        if constexpr (std::is_invocable_v<FB, int, int, int, int>) bv = b_of(4 * kk + kq, r, kk, kq);   // k split into its compile-time / lane parts
        else bv = b_of(4 * kk + kq, r);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_of(r, 4 * kk + kq), bv, acc, 0, 0, 0);
    }
    return acc;
}

// sin / cos of a joint angle (|x| of a few radians): Cody-Waite reduction by pi/2 in two pieces (exact for the |n| <= 2^20 that can occur)
// and the fdlibm kernel polynomials on |r| <= pi/4 -- about 35 dependent-free fp64 operations, against several hundred instructions of
// the library routine with its large-argument path.  Accurate to ~1 ulp (a host libm is ~0.5 ulp; the difference is 1e-16).
__device__ __forceinline__ void sincos_r(double x, double *s, double *c)
{
    const double n = rint(x * 6.36619772367581382433e-01);        // 2 / pi
    double r = fma(-n, 1.57079632673412561417e+00, x);             // pio2 to 33 bits: n * pio2_1 is exact
    r = fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                              -1.98412698298579493134e-04), 8.33333333332248946124e-03);
    const double sr = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                        2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double cr = 1.0 - fma(0.5, z, -(z * z) * pc);
    const int q = (int)n & 3;
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}
__device__ __forceinline__ void sincos_r(float x, float *s, float *c) { sincosf(x, s, c); }

// ============================================================================ kinematics
// Robot::forwardKinematics + matTrans + eulerAnglesToSO3 (Robot.cpp:45-160,176-223,
// generalizedFunctions.cpp:52-72).  Reads L[P_Q], writes A_T (30 x 3x4) and L[P_SC].
// AHEAD (helper wave of the rollout, inside the previous evaluation): the configuration comes from a register (lane i < 30 holds q_i)
// instead of L[P_Q], the roll / pitch / yaw terms go to scratch, and the four the integrator reads (sin / cos of pitch and yaw) to `xd4`.
template <typename R, bool AHEAD = false>
__device__ __forceinline__ void phase_fk(LV<R> L, const LV<R> lcoef, R qn = (R)0, int xd4 = P_SC + 52)
{
    const int lane = LANE;
    constexpr int RPY = AHEAD ? (int)A_T0S + 12 : (int)P_SC + 50;
    // lane = (slot fr < 5, entry el < 12) for the local transforms AND (chain c = fr, entry el) for the chain products; lanes 60..63 repeat
    // lanes 0..3 (the first row of chain 0: a complete quad, so its quad broadcasts see what lanes 0..3 see) -- no lane is switched off, and
    // everything derived from the lane is formed once.
    const unsigned l60 = (unsigned)((lane < 60) ? lane : lane - 60);
    const int lfr = (int)(__umul24(l60, 43u) >> 9), lel = (int)l60 - 12 * lfr;       // l60 / 12, l60 % 12 (exact below 60)
    // DH coefficient loads (L2-resident table) are issued first: their latency hides behind the sincos.  Five of the 28 local transforms
    // per round, six rounds.  An entry takes at most one trigonometric term, and which one depends on its position in the 3 x 4 only
    // (Khalil DH: columns 0, 1 of every row; cos in (0,0), (1,1), (2,1), sin in (0,1), (1,0), (2,0)): two coefficients per entry are loaded
    const bool cosT = (lel == 0) || (lel == 5) || (lel == 9);
    const int ksel = cosT ? 1 : 2;
    R c0[6], ck[6];
    {
        const LV<R> cf = lcoef + 3 * (12 * lfr + lel), cf5 = lcoef + 3 * (12 * ((lfr < 3) ? 25 + lfr : 27) + lel);     // round 5: slots 25..27 (28, 29 do not exist)
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const LV<R> cu = (u < 5) ? cf + 180 * u : cf5;
            c0[u] = cu[0]; ck[u] = cu[ksel];
        }
    }
    const double dh_off = c_dh_off[(lane < 24) ? lane : 0];        // theta offsets, Robot.cpp:59-87 (constant memory, L2-resident like lcoef)
    R qa = (R)0, qpos = (R)0;
    if constexpr (AHEAD) {                                         // lane permutes, outside the branches
        qa = __shfl(qn, (lane < 24) ? 6 + lane : (lane >= 25 && lane < 28) ? lane - 22 : 0, 64);
        qpos = __shfl(qn, (lane < 12) ? (lane >> 2) : 0, 64);
    }
    if (lane < 28) {                                               // one sincos for the 24 joint angles and roll / pitch / yaw
        R x;
        if constexpr (AHEAD) x = (lane < 24) ? qa + (R)dh_off : qa;
        else x = (lane < 24) ? (R)L[P_Q + 6 + lane] + (R)dh_off : (R)L[P_Q + 3 + ((lane >= 25) ? lane - 25 : 0)];
        R s, c;
        sincos_r(x, &s, &c);
        if (lane == 24) { s = -1.0; c = CPI2; }                    // theta[24] = -pi/2 (Robot.cpp:87)
        const int so = (lane < 25) ? P_SC + 2 * lane : RPY + 2 * (lane - 25);
        L[so] = s;
        L[so + 1] = c;
        if (AHEAD && lane >= 26) { L[xd4 + 2 * (lane - 26)] = s; L[xd4 + 2 * (lane - 26) + 1] = c; }
    }
    WSYNC();
    SUBSTAMP(0);
    WSTAMP(44);
    // local transforms (Khalil DH, Robot.cpp:200-214 + the fixed transforms / offsets of :92-154): every entry is
    // c0 + c1 cos(theta) + c2 sin(theta) with model constants (c0, c1, c2) tabulated once on the host
    // (lcoef, L2-resident); the zero / one coefficients make the fused form bit-identical to the products.
    {
        // slot 5 u + fr: its sin | cos sit at P_SC + 2 slot (slots 25..27 are constants, c1 = c2 = 0: what is read there only has to be
        // finite); round 5 also "writes" slots 28, 29 -- the chain products' own first step overwrites those two (T0 aux), nobody reads them
        const LV<R> sc = L + (P_SC + 2 * lfr + (cosT ? 1 : 0)), o = L + (A_LC + 12 * lfr + lel);
#pragma unroll
        for (int u = 0; u < 6; u++) o[60 * u] = fma(ck[u], (R)sc[10 * u], c0[u]);
    }
    if (lane < 12) {                                               // T0 = [R(rpy) p]
        const int r = lane >> 2, col = lane & 3;
        const R sr = L[RPY + 0], cr = L[RPY + 1], sp = L[RPY + 2], cp = L[RPY + 3], sy = L[RPY + 4], cy = L[RPY + 5];
        R val;
        if (col == 3) { if constexpr (AHEAD) val = qpos; else val = L[P_Q + r]; }
        else if (r == 0) val = (col == 0) ? cy * cp : (col == 1) ? cy * sp * sr - sy * cr : cy * sp * cr + sy * sr;
        else if (r == 1) val = (col == 0) ? sy * cp : (col == 1) ? sy * sp * sr + cy * cr : sy * sp * cr - cy * sr;
        else val = (col == 0) ? -sp : (col == 1) ? cp * sr : cp * cr;
        L[A_T0S + lane] = val;
    }
    SUBSTAMP(1);
    WSTAMP(45);
    // Chain products T_dst = T_src Lc: every chain is a path (step s reads what step s - 1 wrote), and lane 12 c + 4 r + col holds entry
    // (r, col) -- the four lanes of a quad are one row of the chain's current transform.  So the recurrence stays in a register and the
    // row of T_src comes from three quad broadcasts (DPP quad_perm) instead of an LDS store / fence / load per step; the local transforms
    // do not depend on the recurrence: all 24 of a lane are loaded before its first step, and from there on the world transforms are
    // written over them (A_T and A_LC are the same 360 doubles).
    // Schedule (Robot.cpp:120-158): legs (c < 2): step 0 = T0 * aux (local slot 25 + c -> T slot 28 + c), steps 1..6 the joints (local
    // 6 c + s - 1 -> T 7 c + s), step 7 the sole (local 27 -> T 7 c + 7); arms (c = 2, 3): steps 0..4 (local 5 c + 2 + s -> T 5 c + 5 + s);
    // head (c = 4): steps 0..2 (local 22 + s -> T 25 + s).  So local slot and T slot of a step are (per-lane base) + s except at the
    // legs' ends; a step a chain does not have works on whatever valid slot that gives and stores into T slot 28 / 29, which nobody reads.
    const int c = lfr, el = lel, col = el & 3;
    const bool leg = c < 2, head = c == 4;
    const int lb = leg ? 6 * c - 1 : head ? 22 : 5 * c + 2;        // local slot of step s = lb + s
    const int db = leg ? 7 * c : head ? 25 : 5 * c + 5;            // T slot of step s = db + s
    WSYNC();
    R tcur = L[A_T0S + el];                                        // T0 (every chain starts from the base)
    R l0[8], l1[8], l2[8];
    {
        const LV<R> Lr = L + (A_LC + 12 * lb + col), L0 = leg ? L + (A_LC + 12 * (25 + c) + col) : Lr, L7 = leg ? L + (A_LC + 12 * 27 + col) : Lr + 84;
#pragma unroll
        for (int s = 0; s < 8; s++) {                              // all 24 operand loads before the first dependent step
            const LV<R> Lo = (s == 0) ? L0 : (s == 7) ? L7 : Lr + 12 * s;
            l0[s] = Lo[0]; l1[s] = Lo[4]; l2[s] = Lo[8];
        }
    }
    WSYNC();                                                       // every lane has its local transforms: their place is free
    if (lane < 12) L[A_T + lane] = tcur;
    {
        const R m3 = (col == 3) ? (R)1 : (R)0;
        const LV<R> Dr = L + (A_T + 12 * db + el), Dx = L + (A_T + 12 * 28 + el);        // regular destination of step s: Dr + 12 s | the unread slot
        const LV<R> D0 = leg ? Dx + 12 * c : Dr;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const R t0 = dpp_row<0x00>(tcur), t1 = dpp_row<0x55>(tcur), t2 = dpp_row<0xAA>(tcur), t3 = dpp_row<0xFF>(tcur);
            R val = t0 * l0[s] + t1 * l1[s] + t2 * l2[s];
            val = fma(t3, m3, val);                                // + T[r][3] in the translation column
            const LV<R> D = (s == 0) ? D0 : (s < 3) ? Dr + 12 * s : (s < 5) ? (head ? Dx : Dr + 12 * s) : (leg ? Dr + 12 * s : Dx);
            D[0] = val;
            tcur = val;
        }
    }
    WSYNC();
}

// ------------------------------------------------------------------ row-per-lane tree layout
// The three tree recursions (Newton-Euler sweeps, composite inertias, feet Jacobian) keep their running quantity in REGISTERS:
// DPP row rho = lane >> 4 is a limb (0 right leg: frames 1..7, 1 left leg: 8..14, 2 right arm 15..19 followed by the head 25, 26,
// 3 left arm 20..24) and lane l16 = lane & 15 < 6 holds component r of the limb's current spatial vector / row r of its current 6 x 6.
// "Multiply by X" is then six v_fmac_f64_dpp with row_newbcast (lane k of the own row supplies component k) -- no LDS round trip per
// level, no index arithmetic: the coefficients come from the full 6 x 6 images A_XF of X_i = [A 0; B A] (row r contiguous, column r at
// stride 6, the zero block stored), addressed as (per-lane base) + (compile-time offset of the level).
// acc += sum_k lane_k(src) * m_k, k = 0..5; needs a full exec mask (wave-uniform control flow).
// NOP (default): `src` may have been written by the VALU instruction right in front of the block -- the compiler is free to place the
// instruction that produces an operand there, "VALU write -> DPP read" needs two wait states, and the hazard recognizer does not look
// into inline asm.  NOP = false only where the source is the accumulator of an earlier block with other blocks in between.
#define LMH_BD6(op) op " %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t" op " %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t" \
                    op " %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t" op " %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
                    op " %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t" op " %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf"
template <bool NOP = true>
__device__ __forceinline__ void bdot6(double &acc, double src, const double (&m)[6])
{
    if constexpr (NOP) asm volatile("s_nop 1\n\t" LMH_BD6("v_fmac_f64_dpp") : "+v"(acc) : "v"(src), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]));
    else asm volatile(LMH_BD6("v_fmac_f64_dpp") : "+v"(acc) : "v"(src), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]));
}
template <bool NOP = true>
__device__ __forceinline__ void bdot6(float &acc, float src, const float (&m)[6])
{
    if constexpr (NOP) asm volatile("s_nop 1\n\t" LMH_BD6("v_fmac_f32_dpp") : "+v"(acc) : "v"(src), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]));
    else asm volatile(LMH_BD6("v_fmac_f32_dpp") : "+v"(acc) : "v"(src), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]));
}
// Exec-masked LDS stores (s_and_saveexec / ds_write / s_or) cost ~28 cycles each; a store whose address is switched to a dump slot for
// the lanes that have nothing to write costs one v_cndmask more than a plain one.
#define P_DUMP (P_TIME + 5)        // never read
#define NE_DUMP (S0 + 840)         // 32 doubles nobody reads: between the Newton-Euler arrays and the X images (free during the kinematic / tree phases)

// limb rows: frame of (row, depth d) = fb + d with fb = base of the limb; the right-arm row switches to the head frames at depth 5
struct TreeRows {
    int rho, r, fbB, fbA, adj;
    bool on6;
};
__device__ __forceinline__ TreeRows tree_rows()
{
    TreeRows t;
    const int lane = LANE;
    t.rho = lane >> 4;
    t.on6 = (lane & 15) < 6;
    t.r = t.on6 ? (lane & 15) : 0;                                 // idle lanes shadow component 0 (their stores are switched off)
    t.fbB = (t.rho == 0) ? 1 : (t.rho == 1) ? 8 : (t.rho == 2) ? 15 : 20;   // depths 0..4
    t.fbA = (t.rho == 2) ? 20 : t.fbB;                             // depths 5, 6: the head (25, 26) rides behind the right arm
    t.adj = (t.rho == 0) ? 0 : (t.rho == 1) ? 1 : 2;               // act(frame) = frame - adj on every limb (Robot.cpp:172)
    return t;
}

// Robot::computeCoM (Robot.cpp:225-238) + parentTransMatrix/allVelocityMatrices/velocityMatrix
// (Robot.cpp:276-298, generalizedFunctions.cpp:11-27: R' used as inverse, kept).  X_i = [A 0; B A], A = E', is written as a full
// 6 x 6 image (A_XF, zero block included).
// NW = 2: wave 0 owns frames 0..13 and the persistent copies, wave 1 the CoM and frames 14..27 (B of a frame needs
// only that frame's E, p, so the two halves never wait for each other; the caller joins them).
template <int NW, typename R>
__device__ __forceinline__ void phase_com_x(LV<R> L, int wid)
{
    const int lane = LANE;
    if (wid == NW - 1) {                                           // NW = 2: the CoM reduction rides on wave 1 (wave 0 also makes the persistent copies)
        R cx = 0, cy = 0, cz = 0;
        if (lane < 28) {
            const LV<R> T = L + A_T + 12 * lane, mo = L + P_MODEL + LMH_BODY_STRIDE * lane;
            const R m = mo[12];
            if (m != 0.0) {
                // joint-frame com = (m c)/m is not stored; the model keeps m*c, so use it directly
                cx = T[0] * mo[9] + T[1] * mo[10] + T[2] * mo[11] + m * T[3];
                cy = T[4] * mo[9] + T[5] * mo[10] + T[6] * mo[11] + m * T[7];
                cz = T[8] * mo[9] + T[9] * mo[10] + T[10] * mo[11] + m * T[11];
            }
        }
        cx = wave_sum(cx); cy = wave_sum(cy); cz = wave_sum(cz);
        {   // one division sequence for the three components (lanes 0..2), not three on lane 0
            const R mass = L[P_MODEL + 392];
            const R num = (lane == 0) ? cx : (lane == 1) ? cy : cz;
            L[(lane < 3) ? P_COM + lane : (int)P_DUMP] = num / mass;
        }
    }
    SUBSTAMP(2);
    WSTAMP(46);
    // frames of this wave: NW = 2: wave 0 works on 1..14 (and frame 0 below), wave 1 on 14..27 -- frame 14 is done by both, from the same
    // inputs with the same instructions, so the two stores carry the same bits; NW = 1: 1..27.  Seven frames per round,
    // lane = (frame slot fr < 7, entry e9 = 3 a + col < 9); lane 63 repeats lane 62.  The same map serves E (this pass) and B (below), so
    // everything derived from the lane is computed once, and from round to round only compile-time offsets move.
    const int f_lo = (NW == 2 && wid == 1) ? 14 : 1;
    constexpr int ROUNDS = (NW == 1) ? 4 : 2;
    const unsigned ln = (unsigned)((lane < 62) ? lane : 62);
    const int fr = (int)(__umul24(ln, 57u) >> 9), e9 = (int)ln - 9 * fr;             // ln / 9, ln % 9 (exact for ln < 64)
    const int a = (int)(__umul24((unsigned)e9, 11u) >> 5), col = e9 - 3 * a;         // e9 / 3, e9 % 3 (exact for e9 < 9)
    const int i0 = f_lo + fr;                                                        // frame of round 0
    const unsigned rootm = 0x02108102u >> i0;                                        // bit 7 u: the frame of round u hangs off the base (Robot.cpp:165)
    {   // E = Rp' Ri: E[a][col] = A[col][a] goes to the top-left and bottom-right blocks of the image (A_XF + 36 i + 6 col + a, + 21), and
        // the lane also clears one entry of the image's zero block (+ 3): the QP phases reuse this scratch, so it is rewritten per evaluation
        const LV<R> Ti0 = L + (A_T + (int)__umul24((unsigned)i0, 12u) + col), Tr = L + (A_T + a), Tq0 = Ti0 + (a - col - 12);
        const LV<R> O0 = L + (A_XF + (int)__umul24((unsigned)i0, 36u) + 6 * col + a);
#pragma unroll
        for (int u = 0; u < ROUNDS; u++) {
            const bool guard = (NW == 1) && (1 + 7 * u + 6 > 27);  // the last round of the single-wave schedule runs past frame 27
            const bool on = !guard || (fr < 27 - 7 * u);
            const LV<R> Ti = Ti0 + 84 * u, Tp = ((rootm >> (7 * u)) & 1u) ? Tr : Tq0 + 84 * u;
            const R val = (R)Tp[0] * (R)Ti[0] + (R)Tp[4] * (R)Ti[4] + (R)Tp[8] * (R)Ti[8];
            if (!guard) { const LV<R> O = O0 + 252 * u; O[0] = val; O[21] = val; O[3] = 0.0; }
            else { const LV<R> O = on ? O0 + 252 * u : L + (int)NE_DUMP; O[0] = val; O[21 * (int)on] = val; O[3] = 0.0; }
        }
    }
    {   // p = Rp' (pi - pp) in the reference's association: lane = (frame slot < 21, component a3 < 3); lane 63 repeats lane 62
        const int fr3 = (int)(__umul24(ln, 43u) >> 7), a3 = (int)ln - 3 * fr3;        // ln / 3, ln % 3 (exact for ln < 64)
        const int j0 = f_lo + fr3;
        constexpr int PR = (NW == 1) ? 2 : 1;
#pragma unroll
        for (int u = 0; u < PR; u++) {
            const bool on = (NW == 2) ? (fr3 < 14) : (21 * u + fr3 < 27);
            const int i = on ? j0 + 21 * u : 1;
            const LV<R> Ti = L + (A_T + 12 * i + 3), Tp = L + (A_T + 12 * f_parent(i));
            const R t0 = Tp[a3], t1 = Tp[4 + a3], t2 = Tp[8 + a3];
            const R v1 = t0 * Ti[0] + t1 * Ti[4] + t2 * Ti[8];
            const R v2 = (-t0) * Tp[3] + (-t1) * Tp[7] + (-t2) * Tp[11];
            L[on ? A_XP + 3 * i + a3 : (int)NE_DUMP + 8] = v1 + v2;
        }
    }
    if (wid == 0 && lane < 12) {                                   // frame 0: E = R0, p = p0
        const LV<R> T0 = L + A_T;
        if (lane < 9) {
            const int a0 = lane / 3, c0 = lane % 3;
            const R v = T0[a0 * 4 + c0];
            L[A_XF + 6 * c0 + a0] = v; L[A_XF + 6 * (3 + c0) + 3 + a0] = v; L[A_XF + 6 * c0 + 3 + a0] = 0.0;
        } else L[A_XP + (lane - 9)] = T0[(lane - 9) * 4 + 3];
    }
    WSYNC();
    SUBSTAMP(3);
    WSTAMP(47);
    {   // B = (-E') [p]x, entry (a, b) = sg1 E[r1][a] p[j1] + sg2 E[r2][a] p[j2]; same lane map (a, bb = col); frames: wave 0 0..13,
        // wave 1 14..27 (NW = 1: 0..27)
        const int bb = col;
        const int r1 = (bb == 0) ? 1 : 0, j1 = (bb == 2) ? 1 : 2, r2 = (bb == 2) ? 1 : 2, j2 = (bb == 0) ? 1 : 0;
        const R sg1 = (bb == 1) ? (R)1 : (R)-1, sg2 = (bb == 1) ? (R)-1 : (R)1;
        const int ib = ((NW == 2 && wid == 1) ? 14 : 0) + fr;
        const LV<R> X0 = L + (A_XF + (int)__umul24((unsigned)ib, 36u) + 6 * a), q0 = L + (A_XP + 3 * ib);    // E[row][a] = A[a][row]: contiguous in the row index
        const LV<R> X1 = X0 + r1, X2 = X0 + r2, q1 = q0 + j1, q2 = q0 + j2, Ob = X0 + (18 + bb);
#pragma unroll
        for (int u = 0; u < ROUNDS; u++) {
            const R val = (sg1 * (R)X1[252 * u]) * (R)q1[21 * u] + (sg2 * (R)X2[252 * u]) * (R)q2[21 * u];
            Ob[252 * u] = val;
        }
    }
    WSYNC();
    SUBSTAMP(4);
    WSTAMP(48);
    if (wid == 0) {
        // persistent copies: T0, T7, T14, X0 = E0 (9) | p0 (3) | B0 (9)
        if (lane < 36) L[P_TB + lane] = L[A_T + 12 * ((lane < 12) ? 0 : (lane < 24) ? 7 : 14) + lane % 12];
        if (lane < 21) {
            const int e = (lane < 9) ? lane : (lane < 12) ? 0 : lane - 12, a0 = e / 3, b0 = e % 3;
            L[P_X0 + lane] = (lane < 9) ? L[A_XF + 6 * b0 + a0] : (lane < 12) ? L[A_XP + lane - 9] : L[A_XF + 6 * (3 + a0) + b0];
        }
        // base-frame reordered velocities, stale (Robot::v_: lanes 0..29) and fresh (lanes 32..61): swapBaseVelocityAndRefToWorldFrame
        {
            const int which = lane >> 5, i = lane & 31, ic = (i < 30) ? i : 29, i6 = (i < 6) ? i : 0;
            const LV<R> v = L + (which ? P_V : P_VP), X = L + A_XF + 6 * i6;              // row i of X_0
            const R dot = ((R)X[0] * (R)v[3] + (R)X[1] * (R)v[4] + (R)X[2] * (R)v[5]) + ((R)X[3] * (R)v[0] + (R)X[4] * (R)v[1] + (R)X[5] * (R)v[2]);
            const R cpy = v[ic];
            L[(which ? P_VHN : P_VHS) + ic] = (i < 6) ? dot : cpy;     // (lanes 30, 31 / 62, 63 repeat entry 29)
        }
    }
    WSYNC();
}

// ---- duplicate-lane form of the row layout (Newton-Euler, feet Jacobian).  Every lane of a 16-lane row works: lanes 6..11 repeat
// components 0..5, lanes 12..15 components 0, 1, 2, 5.  A DPP row_newbcast:k reads lanes 0..5 of the own row, so a repeating lane computes
// bit for bit what the lane it repeats computes (the +-1 lane shifts of the velocity cross term meet the right neighbours under this map:
// component 0 needs 1, 1 needs 0, 3 needs 4, 4 needs 3; 2 and 5 need none), and its stores hit the same address with the same bits.  What
// this buys: no "idle lane" store switches at all -- every store address is (per-lane base, computed once per sweep) + (compile-time offset
// of the level).  The left-arm row repeats the right-arm row's head frames at depths 5, 6 in the same way.
struct TreeDup { int rho, r, fb, fa, adj; bool arms; };
__device__ __forceinline__ TreeDup tree_dup()
{
    TreeDup t;
    const int lane = LANE, l16 = lane & 15;
    t.rho = lane >> 4;
    t.r = (int)((0x5210543210543210ull >> (4 * l16)) & 15ull);
    t.arms = t.rho >= 2;
    t.fb = 1 + 7 * t.rho - ((t.rho == 3) ? 2 : 0);               // 1, 8, 15, 20: depths 0..4
    t.fa = t.arms ? 20 : t.fb;                                     // depths 5, 6: the head (25, 26) behind either arm
    t.adj = t.arms ? 2 : t.rho;                                    // act(frame) = frame - adj on every limb (Robot.cpp:172)
    return t;
}

// Dynamics::computeC (gravity / no gravity) + computeJpqpFrame(7),(14): forward and backward
// Newton-Euler with qdd = 0 on the STALE velocity (Dynamics.cpp:29-60,124-200).
// PLANT = true is the build-defined plant's own pass (lmh_config.plant): the same recursion on the CURRENT velocity (P_VHN), result
// C(q, v) with gravity into P_VHS (the stale base-frame velocity is dead once the controller's pass has run); P_C, P_CG and P_JPQP are
// left alone -- the controller's terms keep the reference's stale-velocity semantics.
// ONE CHAIN in fp64 (LIN): the reference runs the recursion with and without gravity (Cg, of which only the base rows are read, and the
// soles' Jdot qdot come from the gravity-free one).  The recursion is affine in the base acceleration: the gravity part of every body's
// acceleration is gamma_i = X_(0->i) gamma_0 with gamma_0 = X_0 (0 0 0 0 0 9.81)', so
//     C[0:6] - Cg[0:6] = sum_i X_(i->0)' I_i X_(0->i) gamma_0 = Ic_0 gamma_0      (Ic_0: the composite inertia the CRBA forms with the same X, I)
//     a0_sole = ag_sole - (X_sole ... X_1) gamma_0                                 (the product is the base block of the sole's Jacobian)
// and the gravity-free chains (6 of 18 DPP FMAs per level on the way down, 6 of 12 on the way up, half of the body forces) are not run:
// refs_agpqp forms Cg[0:6] = C[0:6] - Ic_0 gamma_0 once the mass matrix is there, refs_jpqp the soles' accelerations behind the Jacobian.
// The subtractions cancel ~2 digits (52 N of weight against velocity products of O(0.1..1)): 1e-14 relative in fp64, which is why the fp32
// model-term modes (R = float) keep both chains.
template <typename R, bool PLANT = false>
__device__ __forceinline__ void phase_newton_euler(LV<R> L)
{
    constexpr bool LIN = std::is_same_v<R, double>;
    const int lane = LANE;
    constexpr int VSRC = PLANT ? (int)P_VHN : (int)P_VHS;
    constexpr int CDST = PLANT ? (int)P_VHS : (int)P_C;
    const TreeDup tr = tree_dup();
    const int r = tr.r;
    // per-lane bases: frame of depth d = fb + d (d < 5) | fa + d (d = 5, 6)
    const LV<R> Xrb = L + (A_XF + 36 * tr.fb + 6 * r), Xra = L + (A_XF + 36 * tr.fa + 6 * r);        // row r of X_f
    const LV<R> Qb = L + (VSRC + 5 + tr.fb - tr.adj), Qa = L + (VSRC + 5 + tr.fa - tr.adj);          // joint rate of frame f
    const LV<R> Vb = L + (A_VEL + 6 * tr.fb + r), Va = L + (A_VEL + 6 * tr.fa + r);                  // A_VEL | + 168 A_ACCG | + 336 A_ACC0
    // base: vel0 = vhat[0:6]; accg0 = X0 * [0 0 0 0 0 9.81]; acc00 = 0
    const R bv = L[VSRC + r], bg = (R)L[A_XF + 6 * r + 5] * (R)9.81;
    L[A_VEL + r] = bv; L[A_ACCG + r] = bg;
    if constexpr (!LIN) L[A_ACC0 + r] = 0.0;
    {   // velocity and acceleration sweeps down the limbs, one frame per depth, recurrences in registers:
        // v_i = X_i v_p + S qd_i; a_i = X_i a_p + crm(v_i) S qd_i (with / without gravity in the base acceleration)
        const R s2 = (r == 2) ? (R)1 : (R)0;                       // S = e_z (angular): component 2
        const R ca = (r == 0 || r == 3) ? (R)1 : (R)0, cb = (r == 1 || r == 4) ? (R)-1 : (R)0;   // crm(v) S = (wy, -wx, 0, vy, -vx, 0)
        R pv = bv, pg = bg, p0 = 0;
        // the coefficients do not depend on the recursion: all seven depths' loads are issued up front (49 doubles), so that the sweep
        // itself never waits for LDS
        R xs[7][6], qds[7];
#pragma unroll
        for (int d = 0; d < 7; d++) {
            const LV<R> X = (d < 5) ? Xrb + 36 * d : Xra + 36 * d;
#pragma unroll
            for (int k = 0; k < 6; k++) xs[d][k] = X[k];
            qds[d] = (d < 5) ? Qb[d] : Qa[d];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < 7; d++) {
            const R (&x)[6] = xs[d];
            R qd = qds[d];
            if (d == 6) qd = tr.arms ? qd : (R)0;                  // the soles carry no joint
            if (d == 5) { pv = tr.arms ? bv : pv; pg = tr.arms ? bg : pg; p0 = tr.arms ? (R)0 : p0; }   // the head starts from the base
            R v = s2 * qd;
            bdot6(v, pv, x);
            asm volatile("s_nop 1");                               // v feeds the lane shifts below
            const R cs = (ca * dpp_row<0x101>(v) + cb * dpp_row<0x111>(v)) * qd;      // row_shl:1 (lane + 1) | row_shr:1 (lane - 1)
            R ag = cs, a0 = cs;
            bdot6(ag, pg, x);
            if (!LIN && d > 0) bdot6(a0, p0, x);                   // (the base's gravity-free acceleration is zero)
            const LV<R> O = (d < 5) ? Vb + 6 * d : Va + 6 * d;
            O[0] = v; O[168] = ag;
            if constexpr (!LIN) O[336] = a0;
            if constexpr (LIN && !PLANT) { if (d == 6) L[tr.arms ? (int)NE_DUMP + 16 + r : P_JPQP + 6 * (tr.rho & 1) + r] = ag; }    // the soles' (refs_jpqp)
            pv = v; pg = ag; p0 = a0;
        }
    }
    WSYNC();
    SUBSTAMP(6);
    WSTAMP(49);
    // body forces f = I a + v x* (I v), one lane per (body, which)
    {
        const bool fon = lane < (LIN ? 25 : 50);
        const int which = (!LIN && fon) ? lane / 25 : 0, i = f_body(LIN ? (fon ? lane : 0) : lane % 25);
        const LV<R> mo = L + P_MODEL + LMH_BODY_STRIDE * i;
        const LV<R> v = L + A_VEL + 6 * i, a = L + (which ? A_ACC0 : A_ACCG) + 6 * i;
        const R m = mo[12], hx = mo[9], hy = mo[10], hz = mo[11];
        // I a
        R n0 = mo[0] * a[0] + mo[1] * a[1] + mo[2] * a[2] + (hy * a[5] - hz * a[4]);
        R n1 = mo[3] * a[0] + mo[4] * a[1] + mo[5] * a[2] + (hz * a[3] - hx * a[5]);
        R n2 = mo[6] * a[0] + mo[7] * a[1] + mo[8] * a[2] + (hx * a[4] - hy * a[3]);
        R f0 = m * a[3] - (hy * a[2] - hz * a[1]);
        R f1 = m * a[4] - (hz * a[0] - hx * a[2]);
        R f2 = m * a[5] - (hx * a[1] - hy * a[0]);
        // I v
        const R p0 = mo[0] * v[0] + mo[1] * v[1] + mo[2] * v[2] + (hy * v[5] - hz * v[4]);
        const R p1 = mo[3] * v[0] + mo[4] * v[1] + mo[5] * v[2] + (hz * v[3] - hx * v[5]);
        const R p2 = mo[6] * v[0] + mo[7] * v[1] + mo[8] * v[2] + (hx * v[4] - hy * v[3]);
        const R l0 = m * v[3] - (hy * v[2] - hz * v[1]);
        const R l1 = m * v[4] - (hz * v[0] - hx * v[2]);
        const R l2 = m * v[5] - (hx * v[1] - hy * v[0]);
        // v x* (p; l) = (w x p + vl x l ; w x l)
        n0 += (v[1] * p2 - v[2] * p1) + (v[4] * l2 - v[5] * l1);
        n1 += (v[2] * p0 - v[0] * p2) + (v[5] * l0 - v[3] * l2);
        n2 += (v[0] * p1 - v[1] * p0) + (v[3] * l1 - v[4] * l0);
        f0 += (v[1] * l2 - v[2] * l1);
        f1 += (v[2] * l0 - v[0] * l2);
        f2 += (v[0] * l1 - v[1] * l0);
        LV<R> fo = L + (which ? A_F0 : A_FG) + 6 * i;
        if (fon) { fo[0] = n0; fo[1] = n1; fo[2] = n2; fo[3] = f0; fo[4] = f1; fo[5] = f2; }
    }
    WSYNC();
    SUBSTAMP(7);
    WSTAMP(50);
    {   // backward sweep up the limbs: fs_i = f_i + X_c' fs_c (child c), C[joint of i] = fs_i[2]; the limb roots (and the head's) are parked
        // for the base sum.  Depth 6 is the massless sole on the leg rows (no force: the legs start at depth 5) and the head's leaf on the
        // arm rows.  The running sum of a level starts from the parent's own body force, so a level is the twelve DPP FMAs and nothing else.
        const LV<R> Xcb = L + (A_XF + 36 * tr.fb + r), Xca = L + (A_XF + 36 * tr.fa + r);              // column r of X_f: stride 6
        const LV<R> Fb = L + (A_FG + 6 * tr.fb + r), Fa = L + (A_FG + 6 * tr.fa + r);                  // A_FG | + 168 A_F0
        // C[5 + act(frame)] = fs[2] (with gravity): the lanes that hold component 2 store, the others go to the dump; depth 6: arms only
        const bool r2 = r == 2;
        const LV<R> Cb = L + (r2 ? CDST + 5 + tr.fb - tr.adj : (int)NE_DUMP), Ca5 = L + (r2 ? CDST + 5 + tr.fa - tr.adj : (int)NE_DUMP);
        const LV<R> Ca6 = L + ((r2 && tr.arms) ? CDST + 5 + tr.fa - tr.adj : (int)NE_DUMP);
        const LV<R> Pk5 = L + (tr.arms ? A_VEL + 48 + r : (int)NE_DUMP + 8);                          // the head's root -> base slot 4 (arm rows)
        R ws[7][6], fgs[7], f0s[7];                                // all depths' operands up front (see the forward sweep)
#pragma unroll
        for (int d = 6; d >= 0; d--) {
            const LV<R> X = (d < 5) ? Xcb + 36 * d : Xca + 36 * d;
#pragma unroll
            for (int k = 0; k < 6; k++) ws[d][k] = X[6 * k];
            const LV<R> F = (d < 5) ? Fb + 6 * d : Fa + 6 * d;
            fgs[d] = F[0]; f0s[d] = LIN ? (R)0 : (R)F[168];
        }
        __builtin_amdgcn_sched_barrier(0);
        R fg = tr.arms ? fgs[6] : (R)0, f0 = (!LIN && tr.arms) ? f0s[6] : (R)0;          // total force of the frame at depth 6
        Ca6[6] = fg;
#pragma unroll
        for (int d = 6; d >= 0; d--) {
            const R (&w)[6] = ws[d];
            if (d == 5) Ca5[5] = fg; else if (d < 5) Cb[d] = fg;
            if (d == 0) {                                          // limb roots: X' fs parked for the base sum, slot = chain (RL, LL, RA, LA)
                R ng = 0, n0 = 0;
                bdot6(ng, fg, w);
                if constexpr (!LIN) bdot6(n0, f0, w);
                L[A_VEL + 12 * tr.rho + r] = ng;                   // A_VEL is dead after the body forces
                if constexpr (!LIN) L[A_VEL + 12 * tr.rho + 6 + r] = n0;
            } else if (d == 5) {                                   // arm rows: the head's root (parked, slot 4), the arm's tip starts afresh
                R ng = tr.arms ? (R)0 : fgs[4], n0 = tr.arms ? (R)0 : f0s[4];
                bdot6(ng, fg, w);
                if constexpr (!LIN) bdot6(n0, f0, w);
                Pk5[0] = ng;
                if constexpr (!LIN) Pk5[6] = n0;
                fg = tr.arms ? fgs[4] : ng; f0 = tr.arms ? f0s[4] : n0;
            } else {
                R ng = fgs[d - 1], n0 = f0s[d - 1];
                bdot6(ng, fg, w);
                if constexpr (!LIN) bdot6(n0, f0, w);
                fg = ng; f0 = n0;
            }
        }
    }
    WSYNC();
    SUBSTAMP(8);
    WSTAMP(51);
    if (lane < (LIN ? 6 : 12)) {                                   // base: head, LA, RA, LL, RL (Dynamics.cpp:157-162 order)
        const int w2 = LIN ? 0 : lane / 6, k2 = LIN ? lane : lane % 6;
        R acc = L[(w2 ? A_F0 : A_FG) + k2];
#pragma unroll
        for (int cc = 4; cc >= 0; cc--) acc += L[A_VEL + 12 * cc + 6 * w2 + k2];
        if constexpr (PLANT) { if (w2 == 0) L[P_VHS + k2] = acc; }
        else { if (w2 == 0) L[P_C + k2] = acc; else L[P_CG + k2] = acc; }
    }
    if (!LIN && !PLANT && lane >= 40 && lane < 52) {               // Jpqp = blkdiag(R,R) acc0[sole]   (LIN: refs_jpqp, behind the Jacobian)
        const int foot = (lane - 40) / 6, k2 = (lane - 40) % 6, rr = k2 % 3, o = (k2 / 3) * 3;
        const LV<R> T = L + P_TB + 12 * (1 + foot), a = L + A_ACC0 + 6 * (foot ? 14 : 7);
        L[P_JPQP + 6 * foot + k2] = T[4 * rr] * a[o] + T[4 * rr + 1] * a[o + 1] + T[4 * rr + 2] * a[o + 2];
    }
    WSYNC();
}

// Dynamics::computeM (CRBA, Dynamics.cpp:62-101) -> Mtop = [Ic0 | F2], Hl (per-limb joint blocks).
// Row-per-lane: lane r of a limb's row holds row r of the composite inertia Ic of the limb's current frame; one iteration folds a frame
// into its parent: Y = Ic X (27 FMAs on the 18 entries of X, read at row-uniform addresses), Z = X' Y (36 v_fmac_f64_dpp: lane k supplies
// row k of Y), Ic_parent = I_parent + Z.  The joint columns f = Ic S ride along: every column still on its way to the base is carried
// one frame up per iteration (six DPP FMAs), leaving H(parent joint, column's joint) = f[2] on the way and F2 at the limb root.
// Seven iterations: the right-arm row folds the head's two frames first (it 0, 1), the legs start at it 1, the arms at it 2.
//
// row r of the 6x6 body inertia [Ibar, [h]x; -[h]x, m 1] (Dynamics.cpp:4-13) from the 13-entry model record: out[c] = sg[c] * mo[idx[c]]
template <typename R>
struct BodyRow6 { int i[6]; R s[6]; };
template <typename R>
__device__ __forceinline__ BodyRow6<R> body_row6(int r)
{
    const bool up = r < 3;
    const int a = up ? r : r - 3;
    BodyRow6<R> q;
    // [h]x row a = [0 -hz hy; hz 0 -hx; -hy hx 0] with h = mo[9..11]
    const int hi0 = (a == 1) ? 11 : 10, hi1 = (a == 0) ? 11 : 9, hi2 = (a == 0) ? 10 : 9;
    const R hs0 = (a == 0) ? (R)0 : (a == 1) ? (R)1 : (R)-1, hs1 = (a == 1) ? (R)0 : (a == 0) ? (R)-1 : (R)1, hs2 = (a == 2) ? (R)0 : (a == 0) ? (R)1 : (R)-1;
    if (up) {                                                       // [Ibar row a | [h]x row a]
        q.i[0] = 3 * a; q.i[1] = 3 * a + 1; q.i[2] = 3 * a + 2; q.s[0] = 1; q.s[1] = 1; q.s[2] = 1;
        q.i[3] = hi0; q.i[4] = hi1; q.i[5] = hi2; q.s[3] = hs0; q.s[4] = hs1; q.s[5] = hs2;
    } else {                                                        // [-[h]x row a | m e_a]
        q.i[0] = hi0; q.i[1] = hi1; q.i[2] = hi2; q.s[0] = -hs0; q.s[1] = -hs1; q.s[2] = -hs2;
        q.i[3] = 12; q.i[4] = 12; q.i[5] = 12; q.s[3] = (a == 0) ? 1 : 0; q.s[4] = (a == 1) ? 1 : 0; q.s[5] = (a == 2) ? 1 : 0;
    }
    return q;
}

template <typename R>
__device__ __forceinline__ void phase_crba_rows(LV<R> L)
{
    const int lane = LANE;
    SUBSTAMP(9);
    const TreeRows tr = tree_rows();
    const int r = tr.r;
    const BodyRow6<R> bs = body_row6<R>(r);
    for (int e = lane; e < 144; e += 64) L[P_HL + e] = 0.0;        // entries outside a limb's block stay zero
    // limb-local joint index of the frame at depth d: d (legs, arms), d - 5 (head); first joint of the limb
    const int jst = (tr.rho == 0) ? 0 : (tr.rho == 1) ? 6 : (tr.rho == 2) ? 12 : 17;
    R Z[6] = {0, 0, 0, 0, 0, 0};                                   // X' Ic X of the frame folded last (0 in front of a leaf)
    R Fc[7];                                                       // joint columns on their way up, slot = iteration that created them
    // operands of an iteration (none depends on the recursion): body row (6), A and B of X_f (18, row-uniform addresses), column r of X_f (6).
    // They are loaded one iteration ahead, in front of the DPP blocks of the running one, so that the recursion never waits for LDS.
    R mo6[6], A_[9], B_[9], w[6];
    auto load_ops = [&](int it, R (&m6)[6], R (&a9)[9], R (&b9)[9], R (&w6)[6]) {
        const int d = 6 - it;
        const int f = ((d < 5) ? tr.fbB : tr.fbA) + d;
        const LV<R> mo = L + P_MODEL + LMH_BODY_STRIDE * f, X = L + A_XF + 36 * f;
#pragma unroll
        for (int c = 0; c < 6; c++) { m6[c] = mo[bs.i[c]]; w6[c] = X[6 * c + r]; }
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) { a9[3 * k + c] = X[6 * k + c]; b9[3 * k + c] = X[6 * (3 + k) + c]; }
    };
    load_ops(0, mo6, A_, B_, w);
#pragma unroll
    for (int it = 0; it < 7; it++) {
        const int d = 6 - it;
        const bool active = (it >= 2) || (it == 1 && tr.rho != 3) || (it == 0 && tr.rho == 2);
        const bool root = (it == 6) || (it == 1 && tr.rho == 2);
        const bool head = (it < 2) && (tr.rho == 2);               // the right-arm row folds the head first (frames 26, 25), then 19..15
        const int jbase = head ? 22 : jst, dl = head ? d - 5 : d;  // this frame's joint = jbase + dl
        // ---- Ic = I_f + Z
        R Ic[6];
#pragma unroll
        for (int c = 0; c < 6; c++) Ic[c] = fma(bs.s[c], mo6[c], Z[c]);
        Fc[it] = Ic[2];                                            // f = Ic S
        const bool l2 = tr.on6 && active && (lane & 15) == 2;      // the lane that holds f[2]
        L[l2 ? P_HL + 6 * (jbase + dl) + dl : (int)P_DUMP] = Ic[2];                               // H(a, a) = f[2]
        // ---- Y = Ic X,  X = [A 0; B A]
        R Y[6];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            Y[c] = Ic[0] * A_[c] + Ic[1] * A_[3 + c] + Ic[2] * A_[6 + c] + Ic[3] * B_[c] + Ic[4] * B_[3 + c] + Ic[5] * B_[6 + c];
            Y[3 + c] = Ic[3] * A_[c] + Ic[4] * A_[3 + c] + Ic[5] * A_[6 + c];
        }
        const R wc[6] = {w[0], w[1], w[2], w[3], w[4], w[5]};
        __builtin_amdgcn_sched_barrier(0);
        if (it < 6) load_ops(it + 1, mo6, A_, B_, w);              // next iteration's operands: in flight behind the DPP blocks below
        __builtin_amdgcn_sched_barrier(0);
        // ---- Z = X' Y and the joint columns one frame up: lane r applies column r of X
        R Zn[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 6; c++) bdot6(Zn[c], Y[c], wc);
        // columns alive: the head's (slots 0, 1) until the head's root at it 1, the limb's own from its leaf on
#pragma unroll
        for (int ci = (it >= 2 ? 1 : 0); ci <= it; ci++) {
            R t = 0;
            bdot6(t, Fc[ci], wc);
            Fc[ci] = t;
            // the column's joint: created at iteration ci on this row
            const bool chead = ci < 2 && tr.rho == 2 && it < 2;
            const int cd = 6 - ci, cj_l = chead ? cd - 5 : cd, cj = (chead ? 22 : jst) + cj_l;
            const bool exists = (tr.rho == 2) ? (it < 2 ? true : ci >= 2) : (tr.rho == 3 ? ci >= 2 : ci >= 1);
            const int pj_l = dl - 1, pj = jbase + pj_l;            // parent joint of this frame (not at a root)
            if (it == 6) L[(tr.on6 && exists) ? P_MTOP + 30 * r + 6 + cj : (int)P_DUMP] = t;    // every row is at its root: F2 column
            else {                                                 // H(parent joint, column joint) = f[2], both triangles; the head's root (it 1, right-arm row): F2
                const bool f2 = (it == 1) && tr.rho == 2;
                L[(tr.on6 && active && exists && f2) ? P_MTOP + 30 * r + 6 + cj : (l2 && exists && !f2) ? P_HL + 6 * pj + cj_l : (int)P_DUMP] = t;
                L[(l2 && exists && !f2) ? P_HL + 6 * cj + pj_l : (int)P_DUMP] = t;
            }
        }
        if (it == 1 || it == 6) {                                  // park X' Ic X of the limb root for the base sum (reference order head, LA, RA, LL, RL)
            const int slot = (it == 1) ? 0 : 4 - tr.rho;
            const bool pk = tr.on6 && active && root;
            const int o = pk ? A_XR + 36 * slot + 6 * r : (int)P_DUMP;
#pragma unroll
            for (int c = 0; c < 6; c++) L[pk ? o + c : (int)P_DUMP] = Zn[c];
        }
        const bool keep = active && !root;
#pragma unroll
        for (int c = 0; c < 6; c++) Z[c] = (it < 2) ? (keep ? Zn[c] : (R)0) : Zn[c];
    }
    WSYNC();
    SUBSTAMP(10);
    WSTAMP(52);
    if (lane < 6) {                                                // Ic0 = I0 + head + LA + RA + LL + RL (Dynamics.cpp:80-82 order)
        const LV<R> mo = L + P_MODEL;
        R acc[6];
#pragma unroll
        for (int c = 0; c < 6; c++) acc[c] = bs.s[c] * (R)mo[bs.i[c]];
#pragma unroll
        for (int sl = 0; sl < 5; sl++) {
            const LV<R> o = L + A_XR + 36 * sl + 6 * r;
#pragma unroll
            for (int c = 0; c < 6; c++) acc[c] += o[c];
        }
        LV<R> mt = L + P_MTOP + 30 * r;
#pragma unroll
        for (int c = 0; c < 6; c++) mt[c] = acc[c];
    }
    SUBSTAMP(11);
    WSYNC();
}

// ---- CRBA on the fp64 matrix cores (the fp64 path; the fp32 model-term modes keep the row-per-lane form above).
// v_mfma_f64_4x4x4_4b_f64 multiplies FOUR independent 4 x 4 x 4 blocks per instruction: block b = (lane >> 2) & 3 is a limb (0 right
// leg, 1 left leg, 2 right arm preceded by the head, 3 left arm), and with rho = lane >> 4, q = lane & 3 the operand / result maps are
//     A[i][k] <- lane (rho = k, q = i),   B[k][j] <- lane (rho = k, q = j),   D[i][j] -> lane (rho = i, q = j)
// (probed on the device: scripts/microbench/mfma4_layout.hip).  A 6 x 6 matrix is four 4 x 4 tiles (zero-padded), one register each.
// A result (D map) is a B operand as it stands, and fed into the A slot it acts as its TRANSPOSE.  So one level,
//     Y = M' X  (A = M tiles, B = X tiles),   Z = X' Y  (A = the same X tiles with their indices swapped, B = Y)
// gives Z = X' M' X: with M = Ic it is (X' Ic X)', with M = Ic' it is X' Ic X -- the stored composite inertia alternates between Ic and its
// transpose from level to level, and the body inertia added on top is gathered transposed or not to match (the legs are one level out
// of step with the arms / head, so the parity is (it & 1) ^ leg).  The joint columns F (B operand throughout, no parity) go up one frame
// with the same X tiles: F <- X' F; the new column Ic S is dropped into its slot by a quad broadcast (M = Ic: column 2 sits at q = 2) or
// by one more product with a unit matrix (M = Ic': M' e_2 e_slot').  16 + 4..8 + 2 matrix instructions per level replace ~100 VALU ones and
// run beside the other wave's VALU work.
__constant__ int c_ib_idx[36] = {0, 1, 2, 12, 11, 10, 3, 4, 5, 11, 12, 9, 6, 7, 8, 10, 9, 12, 12, 11, 10, 12, 12, 12, 11, 12, 9, 12, 12, 12, 10, 9, 12, 12, 12, 12};
__constant__ double c_ib_sgn[36] = {1, 1, 1, 0, -1, 1, 1, 1, 1, 1, 0, -1, 1, 1, 1, -1, 1, 0, 0, 1, -1, 1, 0, 0, -1, 0, 1, 0, 1, 0, 1, -1, 0, 0, 0, 1};
// entry (i, j) of the 6 x 6 body inertia [Ibar, [h]x; -[h]x, m 1] (Dynamics.cpp:4-13) = sgn * record[idx]; per lane: tile t = 2 ta + tc holds
// element (4 ta + rho, 4 tc + q), set s = iteration parity (the gather of set s is transposed when s ^ leg)
struct IbSel { int i[2][4]; double s[2][4]; };
// The gather table of a lane is a launch constant but 24 registers wide; the rollout keeps it PACKED in two registers across its tick loop
// (six bits per (set, tile): record index 0..12 | sign as a two-bit two's-complement number) and unpacks it per evaluation with three
// instructions per entry -- against sixteen constant-memory loads behind ~100 instructions of index arithmetic per evaluation before.
struct IbPack { unsigned w[2]; };
__device__ __forceinline__ IbPack ib_pack()
{
    IbPack g;
    const int lane = LANE, rho = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
#pragma unroll
    for (int st = 0; st < 2; st++) {
        unsigned w = 0u;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int i = 4 * (t >> 1) + rho, j = 4 * (t & 1) + q;
            const bool ok = (i < 6) && (j < 6), tp = (st != 0) != (b < 2);
            const int e = ok ? (tp ? 6 * j + i : 6 * i + j) : 0;
            const int sg = ok ? (int)c_ib_sgn[e] : 0;              // -1, 0, 1
            w |= ((unsigned)c_ib_idx[e] | (((unsigned)sg & 3u) << 4)) << (6 * t);
        }
        g.w[st] = w;
    }
    return g;
}
__device__ __forceinline__ IbSel ib_unpack(const IbPack &p)
{
    IbSel g;
#pragma unroll
    for (int st = 0; st < 2; st++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            g.i[st][t] = (int)((p.w[st] >> (6 * t)) & 15u);
            g.s[st][t] = (double)(((int)(p.w[st] << (26 - 6 * t))) >> 30);      // sign-extended two-bit field
        }
    return g;
}
__device__ __forceinline__ IbSel ib_select() { return ib_unpack(ib_pack()); }
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)
#define CR_DUMP (S0 + 1000)        // 72 doubles nobody reads during the tree phases: S0 + [952, 1200) is the one stretch neither wave 0's Newton-Euler arrays
                                   // nor its early Jacobian rows of U, U D^-1 (qp_prefill15: up to S0 + 952) touch.  The store pointers of idle lanes point
                                   // at CR_DUMP + 24 and are used with level offsets in [-23, 42]
__device__ __forceinline__ void phase_crba_mfma(double *L, const IbSel &g)
{
    const int lane = LANE;
    SUBSTAMP(9);
    // Everything that depends on the lane is formed ONCE, as pointers: a level then addresses (per-lane base) + (compile-time offset of the
    // level), and a lane that has nothing to store at some place holds the dump region's address in that pointer -- no select per store.
    // Block 3 (left arm) repeats block 2's head frames at levels 0, 1 (same inputs, same instructions: the stores carry the same bits), and
    // the legs run their massless sole at level 0 on zeros (lmh_set_model insists on that), so no level switches a block off.
    const int rho = lane >> 4, b = (lane >> 2) & 3, q = lane & 3;
    const bool leg = b < 2, r2 = rho == 2;
    const int fb = leg ? 1 + 7 * b : 5 + 5 * b;                   // 1, 8, 15, 20: frames of levels 2..6 = fb + d (d = 6 - level)
    const int jst = leg ? 6 * b : 2 + 5 * b;                      // 0, 6, 12, 17: first joint of the limb
    const int dh = leg ? 0 : 20 - fb;                             // levels 0, 1: the legs go on (sole, ankle roll), both arm blocks work on the head (26, 25)
    double *const dump = L + (CR_DUMP + 24);
    // (the matrix-core tiles of the QP set-up read two entries past row 5 of P_MTOP, i.e. P_HL[0..1], against zero padding: H has to be
    // finite in EVERY evaluation, also in the ones whose torques nobody asks for -- found with the -DLMH_POISON build)
    L[P_HL + lane] = 0.0; L[P_HL + 64 + lane] = 0.0; L[(lane < 16) ? P_HL + 128 + lane : (int)CR_DUMP] = 0.0;       // entries outside a limb's block stay zero
    // X tiles: element (4 tk + rho, 4 tj + q) of the frame's image; outside the 6 x 6 -> a stored zero of the image (row 0, column 3)
    int xo[4], xt[4];
    bool tin[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int i = 4 * (t >> 1) + rho, j = 4 * (t & 1) + q;
        tin[t] = (i < 6) && (j < 6);
        xo[t] = tin[t] ? 6 * i + j : 3; xt[t] = 6 * j + i;
    }
    const double *xp[4], *mp[2][4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        xp[t] = L + (A_XF + 36 * fb + xo[t]);
        mp[0][t] = L + (P_MODEL + LMH_BODY_STRIDE * fb + g.i[0][t]); mp[1][t] = L + (P_MODEL + LMH_BODY_STRIDE * fb + g.i[1][t]);
    }
    // store pointers (see the level loop): joint-space inertia P_HL[6 a + local column], F2 columns P_MTOP[30 row + 6 + joint]
    const bool ex0 = q >= (leg ? 1 : 2), ex1 = q <= 2;            // column slot 4 tj + q belongs to the limb (legs: 1..6, arms: 2..6)
    double *const pD = (r2 && q == 2) ? L + (P_HL + 6 * jst) : dump;                                   // H(a, a): + 7 d
    double *const pDh = (r2 && q == 2 && !leg) ? L + (P_HL + 6 * 22 - 35) : dump;                      // the head's: + 7 d (d = 6, 5)
    double *const pH1[2] = {(r2 && ex0) ? L + (P_HL + 6 * jst - q) : dump, (r2 && ex1) ? L + (P_HL + 6 * jst - q) : dump};                  // H(parent, column): + 6 d - 4 tj
    double *const pH2[2] = {(r2 && ex0) ? L + (P_HL + 6 * jst - 6 * q + 35) : dump, (r2 && ex1) ? L + (P_HL + 6 * jst - 6 * q + 35) : dump}; // H(column, parent): + d - 24 tj
    double Z[4] = {0, 0, 0, 0}, F[4] = {0, 0, 0, 0};              // tiles t = 2 (row tile) + (column tile)
    double xb[4], ib[4];
    auto load_ops = [&](int it, double (&x4)[4], double (&i4)[4]) {
        const int d = 6 - it;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (it >= 2) { x4[t] = xp[t][36 * d]; i4[t] = mp[it & 1][t][LMH_BODY_STRIDE * d]; }
            else { x4[t] = (xp[t] + 36 * dh)[36 * d]; i4[t] = (mp[it & 1][t] + LMH_BODY_STRIDE * dh)[LMH_BODY_STRIDE * d]; }
        }
    };
    load_ops(0, xb, ib);
#pragma unroll
    for (int it = 0; it < 7; it++) {
        const int d = 6 - it;
        const bool par = ((it & 1) != 0) != leg;                   // true: the stored matrix is Ic'
        // ---- stored composite inertia of this frame: body inertia (gathered with the parity) + what the child handed up
        double M[4];
#pragma unroll
        for (int t = 0; t < 4; t++) M[t] = fma(g.s[it & 1][t], ib[t], Z[t]);
        const double x0 = xb[0], x1 = xb[1], x2 = xb[2], x3 = xb[3];
        __builtin_amdgcn_sched_barrier(0);
        if (it < 6) load_ops(it + 1, xb, ib);                      // next level's operands: in flight behind the products below
        __builtin_amdgcn_sched_barrier(0);
        // ---- new joint column f = Ic S = Ic[:, 2] into slot `it` of F (column tile it >> 2, q = it & 3)
        {
            const int tc = it >> 2, qc = it & 3;
            const bool sel0 = !par && (q == qc);                   // M = Ic: column 2 is (rho, q = 2) of tiles (0,0), (1,0): quad broadcast
            const double v0 = dpp_row<0xAA>(M[0]), v1 = dpp_row<0xAA>(M[2]);
            F[tc] = sel0 ? v0 : F[tc]; F[2 + tc] = sel0 ? v1 : F[2 + tc];
            const double E = (par && r2 && q == qc) ? 1.0 : 0.0;   // M = Ic': (M' e_2) e_slot' as a product (A = M tiles (0, ti): k tile 0 holds k = 2)
            F[tc] = MFMA4(M[0], E, F[tc]); F[2 + tc] = MFMA4(M[1], E, F[2 + tc]);
        }
        // H(a, a) = Ic[2][2] (either parity): joint a = jst + d (levels 2..6, the legs' level 1) | 22 + d - 5 (the head, arm blocks at levels 0, 1)
        if (it >= 2) pD[7 * d] = M[0];
        else if (it == 1) (leg ? pD : pDh)[7 * d] = M[0];
        else (leg ? dump : pDh)[7 * d] = M[0];                     // (level 0 of a leg is the massless sole)
        // ---- Y = M' X (A tile (ti, tk) = M tile (tk, ti)), Zn = X' Y (A tile (ti, tk) = X tile (tk, ti)), Fn = X' F
        const double y0 = MFMA4(M[2], x2, MFMA4(M[0], x0, 0.0)), y1 = MFMA4(M[2], x3, MFMA4(M[0], x1, 0.0));
        const double y2 = MFMA4(M[3], x2, MFMA4(M[1], x0, 0.0)), y3 = MFMA4(M[3], x3, MFMA4(M[1], x1, 0.0));
        double Zn[4], Fn[4];
        Fn[0] = MFMA4(x2, F[2], MFMA4(x0, F[0], 0.0)); Fn[2] = MFMA4(x3, F[2], MFMA4(x1, F[0], 0.0));
        if (it >= 4) { Fn[1] = MFMA4(x2, F[3], MFMA4(x0, F[1], 0.0)); Fn[3] = MFMA4(x3, F[3], MFMA4(x1, F[1], 0.0)); }
        else { Fn[1] = 0.0; Fn[3] = 0.0; }
        Zn[0] = MFMA4(x2, y2, MFMA4(x0, y0, 0.0)); Zn[1] = MFMA4(x2, y3, MFMA4(x0, y1, 0.0));
        Zn[2] = MFMA4(x3, y2, MFMA4(x1, y0, 0.0)); Zn[3] = MFMA4(x3, y3, MFMA4(x1, y1, 0.0));
        // ---- what the columns leave behind.  Column slot ci = 4 tj + q was created at level ci: its joint sits at depth 6 - ci of the limb.
        // On the way up: H(parent joint of this frame, column joint) = H(column joint, parent joint) = (X' f)[2] (row 2: lanes rho = 2):
        //   P_HL[6 (jst + d - 1) + 6 - ci] = pH1[6 d - 4 tj],   P_HL[6 (jst + 6 - ci) + d - 1] = pH2[d - 24 tj].
        // A slot whose column does not exist YET (ci > level) holds zeros and stores them into entries of joints further up, which get their
        // values at later levels; slots that never belong to the limb are switched off in the pointers (ex0, ex1).
        if (it == 6) {                                             // every limb is at its root: the columns are F2 (rows 0..3 here, 4, 5 from the second row tile)
#pragma unroll
            for (int tj = 0; tj < 2; tj++) {
                const bool ex = tj ? ex1 : ex0;
                double *const f2 = ex ? L + (P_MTOP + 30 * rho + 12 + jst - q) : dump, *const f2b = (ex && rho < 2) ? L + (P_MTOP + 30 * (4 + rho) + 12 + jst - q) : dump;
                f2[-4 * tj] = Fn[tj]; f2b[-4 * tj] = Fn[2 + tj];
            }
        } else if (it == 1) {                                      // legs: on the way up; arm blocks: the head's root (columns 0, 1 = joints 23, 22 -> F2)
            double *const f2 = (!leg && q < 2) ? L + (P_MTOP + 30 * rho + 6 + 23 - q) : dump, *const f2b = (!leg && q < 2 && rho < 2) ? L + (P_MTOP + 30 * (4 + rho) + 6 + 23 - q) : dump;
            f2[0] = Fn[0]; f2b[0] = Fn[2];
            double *const h1 = leg ? pH1[0] : dump, *const h2 = leg ? pH2[0] : dump;
            h1[6 * d] = Fn[0]; h2[d] = Fn[0];
        } else if (it == 0) {                                      // arm blocks: the head's leaf (joint 23) hands its column to joint 22; legs: nothing
            double *const hh = (!leg && r2 && q == 0) ? L + (P_HL + 6 * 22 + 1) : dump;
            hh[0] = Fn[0]; hh[5] = Fn[0];                          // H(22, 23) | H(23, 22)
        } else {
#pragma unroll
            for (int tj = 0; tj < ((it >= 4) ? 2 : 1); tj++) { pH1[tj][6 * d - 4 * tj] = Fn[tj]; pH2[tj][d - 24 * tj] = Fn[tj]; }
        }
        if (it == 1 || it == 6) {                                  // park X' Ic X of the limb root for the base sum (reference order head, LA, RA, LL, RL)
#pragma unroll
            for (int t = 0; t < 4; t++) {                          // M = Ic gave (X' Ic X)': level 6: legs hold Ic' (offsets as stored), arms Ic (transposed); level 1 (arms): Ic'
                double *const o = (it == 6) ? (tin[t] ? L + (A_XR + 36 * (4 - b) + (leg ? xo[t] : xt[t])) : dump) : ((tin[t] && !leg) ? L + (A_XR + xo[t]) : dump);
                o[0] = Zn[t];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) { Z[t] = (it == 1) ? (leg ? Zn[t] : 0.0) : Zn[t]; F[t] = Fn[t]; }      // (the arms start afresh above the head's root)
    }
    WSYNC();
    SUBSTAMP(10);
    WSTAMP(52);
    if (lane < 6) {                                                // Ic0 = I0 + head + LA + RA + LL + RL (Dynamics.cpp:80-82 order)
        const BodyRow6<double> bs = body_row6<double>(lane);
        const double *mo = L + P_MODEL;
        double acc[6];
#pragma unroll
        for (int c = 0; c < 6; c++) acc[c] = bs.s[c] * mo[bs.i[c]];
#pragma unroll
        for (int sl = 0; sl < 5; sl++) {
            const double *o = L + A_XR + 36 * sl + 6 * lane;
#pragma unroll
            for (int c = 0; c < 6; c++) acc[c] += o[c];
        }
        double *mt = L + P_MTOP + 30 * lane;
#pragma unroll
        for (int c = 0; c < 6; c++) mt[c] = acc[c];
    }
    SUBSTAMP(11);
    WSYNC();
}
// the fp64 path runs on the matrix cores; R = float (mixed / fp32 model terms) keeps the row-per-lane form
template <typename R>
__device__ __forceinline__ void phase_crba(LV<R> L, const IbSel &g)
{
    if constexpr (std::is_same_v<R, double>) {
        phase_crba_mfma(L.p, g);
        // the gravity part of the bias' base rows, Ic_0 gamma_0 with gamma_0 = 9.81 (0 0 0 | third row of E0): see phase_newton_euler (ONE CHAIN)
        const int lane = LANE, k = (lane < 6) ? lane : 0;
        const double *m = L.p + P_MTOP + 30 * k + 3, *e2 = L.p + P_X0 + 6;
        const double gv = 9.81 * (m[0] * e2[0] + m[1] * e2[1] + m[2] * e2[2]);
        if (lane < 6) L.p[P_CG + lane] = gv;
    }
    else phase_crba_rows<R>(L);
}

// Kinematics::feetJacobian / frameJacobian (invKinematics.cpp:72-149), chain products in the
// reference's association ((X7 X6) X5 ...).  Lane r of a leg's row holds row r of the running product Xn: Xn <- Xn X_f is row-wise
// (27 FMAs on the 18 entries of X_f, read at row-uniform addresses), no exchange between lanes at all; the z column Xn S of
// every intermediate is the joint's Jacobian column.
template <typename R>
__device__ __forceinline__ void phase_jacobian(LV<R> L)
{
    // duplicate-lane form (see tree_dup): no lane exchange happens here at all, so ANY lane that works on (foot, row r) is a valid
    // repeat -- DPP rows 2, 3 repeat the right / left foot, lanes 6..15 repeat rows -- and every store is unconditional.
    const TreeDup tr = tree_dup();
    const int r = tr.r, foot = tr.rho & 1;
    const LV<R> Xs = L + (A_XF + 36 * 7 + 36 * 7 * foot);          // X_sole; X of frame sole - 1 - s sits 36 (1 + s) below
    const LV<R> Jo = L + (A_JL + 72 * foot + 12 * r);
    R xn[6];
#pragma unroll
    for (int c = 0; c < 6; c++) xn[c] = Xs[6 * r + c];             // Xn = X_sole
    R A_[9], B_[9];
    auto load_x = [&](int below, R (&a9)[9], R (&b9)[9]) {
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) { a9[3 * k + c] = Xs[6 * k + c - 36 * below]; b9[3 * k + c] = Xs[6 * (3 + k) + c - 36 * below]; }
    };
    load_x(1, A_, B_);
#pragma unroll
    for (int s = 0; s < 6; s++) {                                  // frame 6..1 / 13..8; the next frame's X is loaded while this one is applied
        Jo[6 + (5 - s)] = xn[2];                                   // Xn S (z column)
        R nn[6];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            nn[c] = xn[0] * A_[c] + xn[1] * A_[3 + c] + xn[2] * A_[6 + c] + (xn[3] * B_[c] + xn[4] * B_[3 + c] + xn[5] * B_[6 + c]);
            nn[3 + c] = xn[3] * A_[c] + xn[4] * A_[3 + c] + xn[5] * A_[6 + c];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s < 5) load_x(2 + s, A_, B_);
#pragma unroll
        for (int c = 0; c < 6; c++) xn[c] = nn[c];
    }
#pragma unroll
    for (int c = 0; c < 6; c++) Jo[c] = xn[c];                     // base block: row r of the whole product
    WSYNC();
    {   // rotate to world axes: J[ft][3 b + r3][col] = sum_k R_sole[r3][k] JL[ft][3 b + k][col].  DPP row q = (ft, b), lane = column (lanes
        // 12..15 repeat columns 0..3); three outputs per lane
        const int lane = LANE, q = lane >> 4, l16 = lane & 15, col = (l16 < 12) ? l16 : l16 - 12, ft = q & 1, b3 = q >> 1;
        const LV<R> T = L + (P_TB + 12 + 12 * ft), J = L + (A_JL + 72 * ft + 36 * b3 + col), O = L + (P_JC + 72 * ft + 36 * b3 + col);
        const R j0 = J[0], j1 = J[12], j2 = J[24];
#pragma unroll
        for (int r3 = 0; r3 < 3; r3++) O[12 * r3] = T[4 * r3] * j0 + T[4 * r3 + 1] * j1 + T[4 * r3 + 2] * j2;
    }
    WSYNC();
}

// Jpqp = blkdiag(R, R) a0_sole (controller.cpp's Jdot qdot of the soles) from the ONE Newton-Euler chain, fp64 schedule:
// a0_sole = ag_sole - X_(0->sole) gamma_0, and blkdiag(R, R) X_(0->sole) is the base block of the rotated Jacobian (phase_jacobian).
// phase_newton_euler has parked the soles' ag raw in P_JPQP; runs behind phase_jacobian, in place.
__device__ __forceinline__ void refs_jpqp(double *L)
{
    const int lane = LANE, l12 = (lane < 12) ? lane : 0, foot = l12 / 6, k2 = l12 % 6, rr = k2 % 3, o = (k2 / 3) * 3;
    const double *T = L + P_TB + 12 * (1 + foot) + 4 * rr, *a = L + P_JPQP + 6 * foot + o, *J = L + P_JC + 72 * foot + 12 * k2 + 3, *e2 = L + P_X0 + 6;
    double val = T[0] * a[0] + T[1] * a[1] + T[2] * a[2];
    val -= 9.81 * (J[0] * e2[0] + J[1] * e2[1] + J[2] * e2[2]);
    __builtin_amdgcn_sched_barrier(0);                             // every lane has read the raw accelerations before any lane overwrites them
    if (lane < 12) L[P_JPQP + lane] = val;
    WSYNC();
}

// Masked LDS load without control flow: the load is always issued (from a clamped, valid address) and
// the value is zeroed by a select; `cond ? L[i] : 0.0` would become exec-mask branching.
__device__ __forceinline__ double ldz(const double *L, bool cond, int idx_if, int idx_safe = 0)
{
    const double v = L[cond ? idx_if : idx_safe];
    return cond ? v : 0.0;
}

// J[row][col] of the dense 12 x 30 feet Jacobian from the compact store
__device__ __forceinline__ double jdense(const double *L, int row, int col)
{
    const int ft = (int)(row >= 6), rr = row - 6 * ft;
    const int j = col - 6 - 6 * ft;                                // leg-joint column of this foot
    const bool base = col < 6, leg = (j >= 0) && (j < 6);
    const int cc = base ? col : (leg ? 6 + j : 0);
    const double v = L[P_JC + 72 * ft + 12 * rr + cc];
    return (base || leg) ? v : 0.0;
}

// Dynamics::centroidalMatrixAndBias (Dynamics.cpp:103-121), Robot::computeComMomentum
// (Robot.cpp:300-310), Mpc3dLip::compute (mpcLinearPendulum.cpp:78-109), PD references
// (controller.cpp:296-386).
// References that depend on the clock only: the preview window of the ZMP (through the gain row), the support phase and the foot
// polynomials.  Everything but the polynomial VALUES is a function of the preview index k = int(t / mpc_dt) alone, and k stays put for
// mpc_dt / dt ticks x 4 stages (40 evaluations at 1 kHz / 10 ms): the robot keeps {k, phase, sum K zmp, the swing segment's coefficients
// already scaled by its step length, the segment's t0} in LDS and goes back to HBM / L2 only when k moves.  One wave maintains the cache
// (NW = 2: the helper wave, beside the forward kinematics); the other reads k / phase after the first join.
#define P_RK (P_TIME + 1)          // cached preview index (as a double; set to -2^30 when a robot is loaded)
#define P_RPH (P_TIME + 2)         // support phase of sample k
#define P_RXS (P_TIME + 3)         // the robot's step-length scale (lmh_set_xscale), 1 without
#define P_ORI (P_TIME + 4)         // 1 = the feet's orientation term of the coming evaluation already sits in P_FREF (rollout: formed behind the look-ahead kinematics)
#define P_RT0 (P_POLY + 54)        // start time of the cached swing segment (0 without segments)
__device__ __forceinline__ void refs_prepare(double *L, LmhCParams &P, int inst, double t)
{
    const int lane = LANE;
    const int N = P.horizon;
    const int k = (int)(t / P.mpc_dt);                             // mpcLinearPendulum.cpp:92 (fp64, same op order; dt_ = the Mpc3dLip ctor's dt)
    if (k != __builtin_amdgcn_readfirstlane((int)L[P_RK])) {       // wave-uniform: the preview index moved
        const double *mp = P.mpc + (size_t)P.mpc_stride_inst * inst;
        const int ns = P.n_samples;
        double sx = 0.0, sy = 0.0;
        for (int i = lane; i <= N; i += 64) {                      // only N = 64 reaches a second round
            int kk = k + i;
            kk = (kk < 0) ? 0 : (kk >= ns ? ns - 1 : kk);
            const double K = (N <= MPC_LDS_MAXN) ? L[P_MPCK + i] : mp[i];
            sx += K * P.zmpx[kk]; sy += K * P.zmpy[kk];
        }
        const int k0 = (k < 0) ? 0 : (k >= ns ? ns - 1 : k);
        const int ph = P.phase ? (int)P.phase[k0] : 0;
        if (P.n_seg > 0) {                                         // walking extension: swing-polynomial segment of sample k, x axis in units of the step length
            const double *sg = P.segs + (size_t)LMH_SEG_STRIDE * (int)P.seg_of_sample[k0];
            const double xs = L[P_RXS];
            if (lane < 48) L[P_POLY + lane] = sg[1 + lane] * (((lane % 24) < 8) ? xs : 1.0);
            else if (lane < 54) L[P_POLY + lane] = 8.0;
            else if (lane == 54) L[P_RT0] = sg[0];
        }
        sx = wave_sum(sx); sy = wave_sum(sy);
        if (lane == 0) { L[P_PRE] = sx; L[P_PRE + 1] = sy; L[P_RK] = (double)k; L[P_RPH] = (double)ph; }
        WSYNC();
    }
    if (lane >= 16 && lane < 22) {                                 // polyval / polyder of the foot references (controller.cpp:355-386)
        const int ft = (lane - 16) / 3, ax = (lane - 16) % 3;
        double co[8];
        const double tl = t - L[P_RT0];
#pragma unroll
        for (int i = 0; i < 8; i++) co[i] = L[P_POLY + 24 * ft + 8 * ax + i];
        // polyval / polyder over all eight terms: the coefficients beyond a polynomial's count are stored as zeros (lmh_set_foot_coeffs, the
        // segment records), and a zero coefficient adds an exact zero -- no count, no selects; the powers of t are formed once
        double pw[8];
        pw[0] = 1.0;
#pragma unroll
        for (int i = 1; i < 8; i++) pw[i] = pw[i - 1] * tl;
        double pv = 0, vv = 0, av = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) pv += co[i] * pw[i];
#pragma unroll
        for (int i = 0; i < 7; i++) vv += ((i + 1) * co[i + 1]) * pw[i];
#pragma unroll
        for (int i = 0; i < 6; i++) av += ((i + 1) * ((i + 2) * co[i + 2])) * pw[i];
        double *pr = L + P_PRE + 4 + 3 * (3 * ft + ax);
        pr[0] = pv; pr[1] = vv; pr[2] = av;
    }
}

// The references form two independent chains: (A) AG, AGpqp -> momentum -> MPC -> PDMomentumAcc needs the mass
// matrix (refs_chain_a); (B) foot velocities, PDJointsAcc -> PDFeetAcc needs the Jacobian.  NW = 1 runs them one after
// the other in one wave; NW = 2 gives chain A to wave 1 and chain B to wave 0 (the caller joins them).
// `ang`: also the angular-momentum rows 0..2 (AG_ang, needed only when the angular-momentum weight is set or a debug record is dumped:
// with w_com_ang = 0 -- the reference's literal -- neither the QP nor any output reads them, and they are the expensive half: 6-term
// entries with three divisions by the mass each)
// AGpqp = X1G Cg[0:6] (Dynamics.cpp:103-121): needs the mass matrix (CRBA) AND the gravity-free bias (Newton-Euler); on the two-wave
// schedule those come from different waves, so this piece runs after their join (phase_qp), not inside the reference chains
template <bool CGLIN>
__device__ __forceinline__ void refs_agpqp(double *L, double mass, bool ang)
{
    const int lane = LANE;
    if (lane < 6) {
        // CGLIN: Cg[0:6] = C[0:6] - Ic_0 gamma_0 (see phase_newton_euler); P_CG holds Ic_0 gamma_0 (refs_cg_gravity, behind the CRBA)
        const double *T0 = L + P_TB, *c = L + P_C, *pg = L + P_CG;
        double cg[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cg[k] = CGLIN ? c[k] - pg[k] : pg[k];
        const int r = lane % 3;
        double val = 0.0;
        if (lane < 3) { if (ang) {
            const double p0 = L[P_MTOP + 30 * 2 + 4] / mass, p1 = L[P_MTOP + 30 * 0 + 5] / mass, p2 = L[P_MTOP + 30 * 1 + 3] / mass;
            const double R0 = T0[4 * r], R1 = T0[4 * r + 1], R2 = T0[4 * r + 2];
            const double u0 = -(R1 * p2 + R2 * (-p1)), u1 = -(R0 * (-p2) + R2 * p0), u2 = -(R0 * p1 + R1 * (-p0));
            val = R0 * cg[0] + R1 * cg[1] + R2 * cg[2] + u0 * cg[3] + u1 * cg[4] + u2 * cg[5];
        } } else val = T0[4 * r] * cg[3] + T0[4 * r + 1] * cg[4] + T0[4 * r + 2] * cg[5];
        L[P_AGPQP + lane] = val;                                   // the angular entries are 0 when nothing reads them (see refs_ag)
    }
}
__device__ __forceinline__ void refs_vfoot_pdjoints(double *L, LmhCParams &P)
{
    const int lane = LANE;
    if (lane >= 8 && lane < 20) {                                  // foot velocities J vhat
        const int row = lane - 8, ft = row / 6;
        const double *J = L + P_JC + 72 * ft + 12 * (row % 6);
        double s = 0.0;
        for (int c = 0; c < 6; c++) s += J[c] * L[P_VHN + c];
        for (int c = 0; c < 6; c++) s += J[6 + c] * L[P_VHN + 6 + 6 * ft + c];
        L[P_VFOOT + row] = s;
    }
    if (lane >= 32 && lane < 62) {                                 // PDJointsAcc, controller.cpp:296-308
        const int i = lane - 32;
        // Robot::desiredPosture (Robot.cpp:253-262): coordinates 0..27 ride in the spare slot of the model record (lmh_model_kernel); 28, 29 are 0
        const double qd0 = L[P_MODEL + LMH_BODY_STRIDE * ((i < 28) ? i : 27) + 13];
        const double val = P.kp_joints * (((i < 28) ? qd0 : 0.0) - L[P_Q + i]) + P.kd_joints * (0.0 - L[P_V + i]);
        L[P_QREF + ((i < 3) ? i + 3 : (i < 6) ? i - 3 : i)] = val;
    }
}
__device__ __forceinline__ double readlane_f64(double v, int src_lane)       // wave-uniform copy of lane src_lane's value (two v_readlane_b32)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffll), src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), src_lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Chain A in one piece: Dynamics::centroidalMatrixAndBias (AG, Dynamics.cpp:103-121), Robot::computeComMomentum (Robot.cpp:300-310),
// Mpc3dLip::compute (mpcLinearPendulum.cpp:78-109: u0 = -K (Px x_k - z[k : k+N+1]); K (Px x_k - z) = x sum K Px0 + xdot sum K Px1 - sum K z,
// the three sums do not depend on the robot state and are prepared early, refs_prepare / load_common) and PDMomentumAcc
// (controller.cpp:310-325), without LDS round trips between them; the angular rows (`ang`: weighted or dumped, see refs_ag)
// keep the plain form beside it.  DPP row a = row 3 + a of AG (row 3 of the wave repeats a = 2), lane l16 of it forms the entries of columns
// l16 and l16 + 16 with the expressions of refs_ag, multiplies them with vhat and the row sums go round through row broadcasts (the order
// of the 30-term sum differs from refs_momentum's: 1e-16 relative); the three quotients reach every lane as scalars (v_readlane), so the MPC
// step and the PD law follow in registers.
__device__ __forceinline__ int refs_chain_a(double *L, LmhCParams &P, int inst, int k, double mass, bool ang)
{
    const int lane = LANE, l16 = lane & 15, q4 = lane >> 4, a = (q4 < 3) ? q4 : 2;
    if (ang) {                                                     // wave-uniform: rows 0..2 of AG (refs_ag)
        const double *T0 = L + P_TB;
        for (int e = lane; e < 90; e += 64) {
            const int r = e / 30, c = e % 30;
            const double *Mt = L + P_MTOP + c;
            const double p0 = L[P_MTOP + 30 * 2 + 4] / mass, p1 = L[P_MTOP + 30 * 0 + 5] / mass, p2 = L[P_MTOP + 30 * 1 + 3] / mass;
            const double R0 = T0[4 * r], R1 = T0[4 * r + 1], R2 = T0[4 * r + 2];
            const double u0 = -(R1 * p2 + R2 * (-p1)), u1 = -(R0 * (-p2) + R2 * p0), u2 = -(R0 * p1 + R1 * (-p0));
            L[P_AG + e] = R0 * Mt[0] + R1 * Mt[30] + R2 * Mt[60] + u0 * Mt[90] + u1 * Mt[120] + u2 * Mt[150];
        }
    }
    const bool two = l16 < 14;
    const int c0 = l16, c1 = two ? l16 + 16 : l16;
    const double *T0 = L + P_TB + 4 * a, *Mt = L + P_MTOP + 90, *vh = L + P_VHN;
    const double r0 = T0[0], r1 = T0[1], r2 = T0[2];
    const double ag0 = r0 * Mt[c0] + r1 * Mt[30 + c0] + r2 * Mt[60 + c0];
    const double ag1 = r0 * Mt[c1] + r1 * Mt[30 + c1] + r2 * Mt[60 + c1];
    const double v0 = vh[c0], v1 = two ? vh[c1] : 0.0;
    double *o = L + P_AG + 30 * (3 + a);
    o[c0] = ag0; o[c1] = ag1;                                       // (lanes 14, 15 repeat their first entry)
    const double part = ag0 * v0 + ag1 * v1, one = 1.0;
    double s0 = 0.0, s1 = 0.0;
    dpp_fmac_lane<0>(s0, part, one); dpp_fmac_lane<1, false>(s1, part, one); dpp_fmac_lane<2, false>(s0, part, one); dpp_fmac_lane<3, false>(s1, part, one);
    dpp_fmac_lane<4, false>(s0, part, one); dpp_fmac_lane<5, false>(s1, part, one); dpp_fmac_lane<6, false>(s0, part, one); dpp_fmac_lane<7, false>(s1, part, one);
    dpp_fmac_lane<8, false>(s0, part, one); dpp_fmac_lane<9, false>(s1, part, one); dpp_fmac_lane<10, false>(s0, part, one); dpp_fmac_lane<11, false>(s1, part, one);
    dpp_fmac_lane<12, false>(s0, part, one); dpp_fmac_lane<13, false>(s1, part, one); dpp_fmac_lane<14, false>(s0, part, one); dpp_fmac_lane<15, false>(s1, part, one);
    const double cv = (s0 + s1) / mass;                             // CoM velocity, component a, on every lane of row a
    const double vxp = readlane_f64(cv, 0), vyp = readlane_f64(cv, 16), vzp = readlane_f64(cv, 32);
    int flags = 0;
    const int N = P.horizon;
    if (k < 0 || k + N >= P.n_samples) flags |= LMH_FLAG_ZMP_RANGE;
    const double *mp = P.mpc + (size_t)P.mpc_stride_inst * inst;
    const double zcom = mp[3 * (N + 1)];
    // MPC: u0 = -K (Px x_k - z[k : k+N+1]) (refs_mpc)
    const double cxp = L[P_COM], cyp = L[P_COM + 1], czp = L[P_COM + 2];
    const double kp0 = L[P_PRE + 2], kp1 = L[P_PRE + 3];
    const double ux = -((kp0 * cxp + kp1 * vxp) - L[P_RXS] * L[P_PRE]);
    const double uy = -((kp0 * cyp + kp1 * vyp) - L[P_PRE + 1]);
    const double xp = P.a00 * cxp + P.a01 * vxp + P.b0 * ux, xv = P.a10 * cxp + P.a11 * vxp + P.b1 * ux;
    const double yp = P.a00 * cyp + P.a01 * vyp + P.b0 * uy, yv = P.a10 * cyp + P.a11 * vyp + P.b1 * uy;
    if (lane == 0) {
        L[P_MPC + 0] = ux; L[P_MPC + 1] = uy;
        L[P_MPC + 2] = xp; L[P_MPC + 3] = xv; L[P_MPC + 4] = ux;
        L[P_MPC + 5] = yp; L[P_MPC + 6] = yv; L[P_MPC + 7] = uy;
    }
    if (lane < 3) {                                                // PDMomentumAcc (refs_pd_momentum); no angular momentum: h_ref[0:3] = kd (0 - 0)
        const double posRef = (lane == 0) ? xp : (lane == 1) ? yp : zcom, velRef = (lane == 0) ? xv : (lane == 1) ? yv : 0.0;
        const double accRef = (lane == 0) ? ux : (lane == 1) ? uy : 0.0;
        const double cm = (lane == 0) ? cxp : (lane == 1) ? cyp : czp, cvl = (lane == 0) ? vxp : (lane == 1) ? vyp : vzp;
        L[P_HREF + 3 + lane] = mass * (P.kp_mom * (posRef - cm) + P.kd_mom * (velRef - cvl) + accRef);
        L[P_COMV + lane] = cvl;
    }
    if (ang) WSYNC();                                              // the angular rows of AG
    if (lane < 3) {                                                // angular momentum AG_ang vhat and its PD term
        double hs = 0.0;
        if (ang) for (int c = 0; c < 30; c++) hs += L[P_AG + 30 * lane + c] * L[P_VHN + c];
        L[P_ANGM + lane] = hs;
        L[P_HREF + lane] = P.kd_mom * (0.0 - hs);
    }
    return flags;
}
// PDFeetAcc, orientation part (controller.cpp:344-353): kp_feet * (-R_des log(R_des' R_foot)) of both feet into P_FREF[6 ft + 0..2].  It needs
// the soles' world transforms only (T7 at Ts, T14 at Ts + stride), not the Jacobian: in single support / flight the rollout's helper wave
// forms it behind the look-ahead kinematics (acos + sin: ~200 instructions off wave 0's path where wave 0 is the long one); in double
// support wave 0 keeps it -- measured: the helper's instructions cost more than wave 0's (wave 0 has issue priority on the SIMD the two
// waves of DIFFERENT robots share), so work moves to the helper only where it has real slack.
__device__ __forceinline__ void refs_feet_orientation(double *L, LmhCParams &P, const double *Ts, int stride)
{
    const int lane = LANE;
    if (lane >= 8 && lane < 10) {
        const int ft = lane - 8;
        const double *T = Ts + stride * ft;
        // err = R_des' R_foot with R_des = Rf_q0_ = [0 0 1; 0 -1 0; 1 0 0] (Robot.cpp:28-31) written out: row 0 = row 2 of R_foot, row 1 = -row 1,
        // row 2 = row 0 (the products with the literal 0 / 1 / -1 entries and the sums with the resulting zeros are exact, so this is what
        // the general 3 x 3 product returns -- without nine constants held in scalar registers across the evaluation)
        double err[9];
        for (int b = 0; b < 3; b++) { err[b] = T[8 + b]; err[3 + b] = -T[4 + b]; err[6 + b] = T[b]; }
        const double tr = err[0] + err[4] + err[8];
        const double cc = fmax(-1.0, fmin(1.0, (tr - 1.0) / 2.0));
        const double phi = acos(cc);
        const double v0 = err[7] - err[5], v1 = err[2] - err[6], v2 = err[3] - err[1];
        double sphi, cphi;
        sincos_r(phi, &sphi, &cphi);                               // phi in [0, pi]: the reduced-range kernel (~1 ulp) instead of the library sin with its large-argument path
        (void)cphi;
        const double sc = (phi < 1e-6) ? 0.5 : (phi / (2.0 * sphi));
        const double r0 = sc * v0, r1 = sc * v1, r2 = sc * v2;
        const double e3[3] = {-r2, r1, -r0};                       // -R_des (r0, r1, r2)'
        for (int a = 0; a < 3; a++) L[P_FREF + 6 * ft + a] = P.kp_feet * e3[a];
    }
}
__device__ __forceinline__ void refs_feet_angular_velocity_term(double *L, LmhCParams &P)     // P_FREF[angular] += kd (0 - omega_foot)
{
    const int lane = LANE;
    if (lane >= 8 && lane < 14) {
        const int i6 = lane - 8, ft = (i6 >= 3) ? 1 : 0, a = i6 - 3 * ft;
        L[P_FREF + 6 * ft + a] = fma(P.kd_feet, 0.0 - L[P_VFOOT + 6 * ft + a], L[P_FREF + 6 * ft + a]);
    }
}
// ORI_MAYBE (rollout): the helper wave may have left the orientation term in P_FREF already (L[P_ORI], see lmh_rollout_kernel)
template <bool ORI_MAYBE = false>
__device__ __forceinline__ void refs_pd_feet(double *L, LmhCParams &P, int inst, double t, int k)
{
    const int lane = LANE;
    if (!ORI_MAYBE || __builtin_amdgcn_readfirstlane((int)L[P_ORI]) == 0) { refs_feet_orientation(L, P, L + P_TB + 12, 12); WSYNC(); }
    refs_feet_angular_velocity_term(L, P);
    if (lane >= 16 && lane < 22) {                                 // position part, polynomials (polyval/polyder)
        const int ft = (lane - 16) / 3, ax = (lane - 16) % 3;
        const double *pr = L + P_PRE + 4 + 3 * (3 * ft + ax);     // refs_prepare
        const double pv = pr[0], vv = pr[1], av = pr[2];
        const double pe = pv - L[P_TB + 12 * (1 + ft) + 4 * ax + 3];
        const double ve = vv - L[P_VFOOT + 6 * ft + 3 + ax];
        L[P_FREF + 6 * ft + 3 + ax] = P.kp_feet * pe + P.kd_feet * ve + av;
    }
}

template <int NW, bool ORI_MAYBE = false, bool CGLIN = true>
__device__ __forceinline__ int phase_refs(double *L, LmhCParams &P, int inst, double t, int wid, int *k_out, int *phase_out, bool ang)
{
    int flags = 0;
    const double mass = L[P_MODEL + 392];
    const int k = __builtin_amdgcn_readfirstlane((int)L[P_RK]);   // the clock-only references of this evaluation (refs_prepare, before the first join)
    *k_out = k; *phase_out = __builtin_amdgcn_readfirstlane((int)L[P_RPH]);
    if (k < 0 || k + P.horizon >= P.n_samples) flags |= LMH_FLAG_ZMP_RANGE;    // on every wave (wave 0 reports)
    if constexpr (NW == 1) {
        flags |= refs_chain_a(L, P, inst, k, mass, ang);           // the same code on either schedule: the results agree bit for bit
        refs_agpqp<CGLIN>(L, mass, ang);
        SUBSTAMP(12);
        refs_vfoot_pdjoints(L, P);
        WSYNC();
        SUBSTAMP(13);
        SUBSTAMP(14);
        refs_pd_feet<false>(L, P, inst, t, k);
        WSYNC();
    } else if (wid == 1) {                                         // chain A
        WSTAMP(66);
        flags |= refs_chain_a(L, P, inst, k, mass, ang);
        WSYNC();
    } else {                                                       // chain B
        WSTAMP(67);
        refs_vfoot_pdjoints(L, P);
        WSYNC();
        refs_pd_feet<ORI_MAYBE>(L, P, inst, t, k);
        WSYNC();
    }
    return flags;
}


// Cone Hessian for the general free-set solve: WG = W G (12 x 32), P = G' WG + eps I (32 x 32), six MFMA tiles.
// Formed lazily (cone_qp): the push-through route never reads it.
__device__ __forceinline__ void build_cone_matrix(double *L, LmhCParams &P)
{
    const int lane = LANE, tr = lane & 15, tq = lane >> 4;
    WSYNC();
    {
        auto a_w = [=](int m, int k) { return ldz(L, m < 12 && k < 12, P_W + 12 * m + k, P_W); };
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            auto b_g = [=](int k, int n) { const int kk = k - 6 * nt; return ldz(L, kk >= 0 && kk < 6, P_GCOL + 6 * n + kk, P_GCOL); };
            const v4d wg = mfma_tile<3>(a_w, b_g);
#pragma unroll
            for (int g = 0; g < 4; g++) { const int row = tq + 4 * g; if (row < 12) L[C_WG + 32 * row + 16 * nt + tr] = wg[g]; }
        }
    }
    WSYNC();
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        auto a_gt = [=](int m, int k) { const int kk = k - 6 * mt; return ldz(L, kk >= 0 && kk < 6, P_GCOL + 6 * m + kk, P_GCOL); };
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            auto b_wg = [=](int k, int n) { return ldz(L, k < 12, C_WG + 32 * k + 16 * nt + n, C_WG); };
            const v4d pp = mfma_tile<3>(a_gt, b_wg);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int i = 16 * mt + tq + 4 * g, j = 16 * nt + tr;
                L[C_P + 33 * i + j] = pp[g] + ((i == j) ? P.eps_coeff : 0.0);
            }
        }
    }
    WSYNC();
}

// Row `ia` of the cone Hessian P = G'WG + eps I without forming it: P_ij = g_i' W_(f(i) f(j)) g_j, g_i = generator of coefficient i
// (foot f(i) = i >> 4).  u_R = W_(f(i),R)' g_i and u_L = W_(f(i),L)' g_i (72 FMAs) serve every column of the row; column j then costs six
// FMAs on operands read at a wave-uniform LDS address.  Replaces the six MFMA tiles + 32 x 33 LDS image of build_cone_matrix on the
// general free-set route (|F| is usually <= 8 there: a foot pressing on an edge of its support polygon).
struct ConeRow { double uR[6], uL[6]; };
__device__ __forceinline__ ConeRow cone_row_prepare(const double *L, int ia)
{
    ConeRow q;
    const double *gi = L + P_GCOL + 6 * (ia & 15), *Wr = L + P_W + 72 * (ia >> 4);
    double g[6];
#pragma unroll
    for (int a = 0; a < 6; a++) g[a] = gi[a];
#pragma unroll
    for (int b = 0; b < 6; b++) {
        double sr = 0.0, sl = 0.0;
#pragma unroll
        for (int a = 0; a < 6; a++) { sr += g[a] * Wr[12 * a + b]; sl += g[a] * Wr[12 * a + 6 + b]; }
        q.uR[b] = sr; q.uL[b] = sl;
    }
    return q;
}
// The rows and columns of P that belong to the free set F (wave-uniform), written into the 32 x 33 image at C_P: lane r < |F| owns the r-th
// free coefficient; the column loop is a scalar bit-scan over F (right-foot columns take u_R, left-foot columns u_L).
__device__ __forceinline__ void build_cone_rows(double *L, unsigned F, double eps)
{
    const int lane = LANE;
    int ia = 0;
    {
        unsigned m = F;
#pragma unroll
        for (int c = 0; c < 32; c++) {
            const int j = m ? __builtin_ctz(m) : 0;
            m &= m - 1u;
            ia = (lane == c) ? j : ia;
        }
    }
    WSYNC();
    const ConeRow q = cone_row_prepare(L, ia);
    double *Pr = L + C_P + 33 * ia;                               // lanes beyond |F| recompute row 0 (same values)
    for (unsigned m = F & 0xFFFFu; m; m &= m - 1u) {
        const int j = __builtin_ctz(m);
        const double *gj = L + P_GCOL + 6 * j;
        double sacc = (ia == j) ? eps : 0.0;
#pragma unroll
        for (int b = 0; b < 6; b++) sacc += q.uR[b] * gj[b];
        Pr[j] = sacc;
    }
    for (unsigned m = F >> 16; m; m &= m - 1u) {
        const int j = __builtin_ctz(m);
        const double *gj = L + P_GCOL + 6 * j;
        double sacc = (ia == 16 + j) ? eps : 0.0;
#pragma unroll
        for (int b = 0; b < 6; b++) sacc += q.uL[b] * gj[b];
        Pr[16 + j] = sacc;
    }
    WSYNC();
}

// Compacted solve for |F| <= N: lane r < nF owns the r-th free coefficient; the N x N register LDL' then only
// visits live pivots.  F is wave-uniform, so the positions of its set bits come from a scalar bit-scan chain
// (no LDS index table); z of coefficient j is read back from lane pos(j) with one lane permute.
template <int N>
__device__ __forceinline__ int solve_compact(double *L, unsigned F, int nF, int pos, double eps, double *zj_out)
{
    const int lane = LANE;
    double a[N], b[1];
    {
        int idx[N], ia = 0;
        unsigned m = F;
#pragma unroll
        for (int c = 0; c < N; c++) {                              // idx[c] for c >= nF: 0, a valid (unused) index
            idx[c] = m ? __builtin_ctz(m) : 0;
            m &= m - 1u;
            ia = (lane == c) ? idx[c] : ia;
        }
        const bool on = lane < nF;
        const double *Pr = L + C_P + 33 * ia;                      // rows / columns of F only (build_cone_rows)
#pragma unroll
        for (int c = 0; c < N; c++) a[c] = (on && c <= lane) ? Pr[idx[c]] : 0.0;
        b[0] = on ? L[P_QV + ia] : 0.0;
    }
    const int bad = ldl_solve_regs<N, 1>(a, b, (nF >= 32) ? 0xFFFFFFFFu : ((1u << nF) - 1u), L + C_LS);
    *zj_out = __shfl(b[0], pos, 64);                               // coefficient j <- row pos(j)
    return bad;
}

// |F| in 17..32: two rows per lane (ldl2_solve_regs); coefficient of compact row r < 16 in lane r (b0), of row
// r >= 16 in lane r - 16 (b1).
__device__ __forceinline__ int solve_compact2(double *L, unsigned F, int nF, int pos, double eps, double *zj_out)
{
    const int lane = LANE;
    double a0[16], a1[32], b0, b1;
    {
        int idx[32], ia0 = 0, ia1 = 0;
        unsigned m = F;
#pragma unroll
        for (int c = 0; c < 32; c++) {
            idx[c] = m ? __builtin_ctz(m) : 0;
            m &= m - 1u;
            if (c < 16) ia0 = (lane == c) ? idx[c] : ia0; else ia1 = (lane == c - 16) ? idx[c] : ia1;
        }
        const bool on0 = (lane < 16) && (lane < nF), on1 = (lane < 16) && (16 + lane < nF);
        const double *P0 = L + C_P + 33 * ia0, *P1 = L + C_P + 33 * ia1;
#pragma unroll
        for (int c = 0; c < 16; c++) a0[c] = (on0 && c <= lane) ? P0[idx[c]] : 0.0;
#pragma unroll
        for (int c = 0; c < 32; c++) a1[c] = (on1 && c <= 16 + lane) ? P1[idx[c]] : 0.0;
        b0 = on0 ? L[P_QV + ia0] : 0.0;
        b1 = on1 ? L[P_QV + ia1] : 0.0;
    }
    const int bad = ldl2_solve_regs<16>(a0, a1, b0, b1, (nF >= 32) ? 0xFFFFFFFFu : ((1u << nF) - 1u), L + C_LS);
    const double z0 = __shfl(b0, pos & 15, 64), z1 = __shfl(b1, pos & 15, 64);
    *zj_out = (pos < 16) ? z0 : z1;                                // coefficient j <- compact row pos(j)
    return bad;
}

// Solve P_FF z_F = qv_F on the free set F (wave-uniform).  Returns z_j for lane j in F (0 otherwise) and, for
// lanes j < 32 not in F, the multiplier lam_j = (P z - qv)_j.
__device__ __forceinline__ int solve_free_set(double *L, unsigned F, double eps, double *z_out, double *lam_out)
{
    const int lane = LANE, r = lane & 31, half = lane >> 5;
    const bool inF = (lane < 32) && ((F >> lane) & 1u);
    const int nF = __popc(F);
    const int pos = __popc(F & ((1u << r) - 1u));
    int bad;
    double zr;
    if (nF <= 8) bad = solve_compact<8>(L, F, nF, pos, eps, &zr);
    else if (nF <= 16) bad = solve_compact<16>(L, F, nF, pos, eps, &zr);
    else bad = solve_compact2(L, F, nF, pos, eps, &zr);
    const double zj = inF ? zr : 0.0;
    WSYNC();
    if (lane < 32) L[P_CC + lane] = zj;
    WSYNC();
    double s = 0.0;
    {   // (P z)_r = g_r' W (G z) + eps z_r through the wrench G z: three short LDS steps instead of a 32 x 32 image of P
        (void)half;
        double *wz = L + C_LS, *Wwz = L + C_LS + 16;               // the rows of L parked by the solve above are dead
        if (lane < 12) {
            const int ft = lane / 6, k = lane % 6;
            double sacc = 0.0;
#pragma unroll
            for (int j = 0; j < 16; j++) sacc += L[P_GCOL + 6 * j + k] * L[P_CC + 16 * ft + j];
            wz[lane] = sacc;
        }
        WSYNC();
        if (lane < 12) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < 12; k++) sacc += L[P_W + 12 * lane + k] * wz[k];
            Wwz[lane] = sacc;
        }
        WSYNC();
        const double *g = L + P_GCOL + 6 * (r & 15), *y = Wwz + 6 * (r >> 4);
        s = eps * zj;
#pragma unroll
        for (int k = 0; k < 6; k++) s += g[k] * y[k];
    }
    *z_out = zj;
    *lam_out = (lane < 32 && !((F >> lane) & 1u)) ? s - L[P_QV + r] : 0.0;
    return bad;
}

// ---- thin free sets (|F| <= 8: a foot pressing on an edge or a corner of its support polygon -- the usual case in single support), the
// whole restricted solve in registers.  Lane r < |F| owns the r-th free coefficient i(r); with u_r = W' g_i(r) (the rows of W that belong to
// the coefficient's foot) the entries of its row are P_rc = u_r . g_i(c) + eps [r = c], formed in a scalar loop over the set bits of F
// (g_i(c) is read at a wave-uniform address) straight into the registers of the LDL' -- no image of P in LDS.  The multipliers of the
// coefficients outside F come through the wrench: w = G_F z and W w by DPP row broadcasts on lanes 0..11 (z sits in lanes 0..7 of the same
// 16-lane row), then lam_j = g_j . (W w - h) of the coefficient's foot: one LDS hand-over instead of four.
// Returns like solve_free_set: z_j for lane j in F (0 otherwise), lam_j for lanes j < 32 outside F.
// MODE 0: every free coefficient belongs to the right foot, 1: to the left foot (single support: the usual thin set), 2: mixed
// NT = 8 | 16 rows.  NT = 16 is the double-support set right after a touch-down: both feet press on one edge of their polygons (two vertices,
// 4 + 4 rays each: F = 0f0f0f0f for ~130 of the 200 ticks of every double-support phase of the walking workload, scripts/route_probe.py); each
// foot's K_f = G_F G_F' then has rank 5, the push-through route does not apply, and the |F| x |F| solve through the LDS image took 16.8k
// cycles of such an evaluation (profiles/r04_prod_timeline.txt).
// The set bits of F are walked with a scalar mask (no index array held in scalar registers across the solve).
template <int NT, int C, int MODE>
__device__ __forceinline__ void thin_cols(double *L, int nF, unsigned m, const double (&uR)[6], const double (&uL)[6], double (&a)[NT])
{
    if constexpr (C < NT) {
        if (C < nF) {                                              // wave-uniform
            const int j = __builtin_amdgcn_readfirstlane(__builtin_ctz(m));
            const double *g = L + P_GCOL + 6 * (j & 15);
            double sacc = 0.0;
            if (MODE == 1 || (MODE == 2 && (j >> 4))) {
#pragma unroll
                for (int b = 0; b < 6; b++) sacc += uL[b] * g[b];
            } else {
#pragma unroll
                for (int b = 0; b < 6; b++) sacc += uR[b] * g[b];
            }
            a[C] = sacc;
        } else a[C] = 0.0;
        thin_cols<NT, C + 1, MODE>(L, nF, m & (m - 1u), uR, uL, a);
    }
}
template <int NT, int C>
__device__ __forceinline__ void thin_wrench(double *L, int nF, unsigned m, double z, int kk, bool leftlane, double &wz)
{
    if constexpr (C < NT) {
        if (C < nF) {                                              // wave-uniform
            const int j = __builtin_amdgcn_readfirstlane(__builtin_ctz(m));
            const bool match = ((j >> 4) != 0) == leftlane;        // the coefficient pushes on this lane's foot
            const double gv = L[P_GCOL + 6 * (j & 15) + kk];       // (no zero slot of the set-up scratch survives the cone phase: select on the value)
            dpp_fmac_lane<C>(wz, z, match ? gv : 0.0);
        }
        thin_wrench<NT, C + 1>(L, nF, m & (m - 1u), z, kk, leftlane, wz);
    }
}
template <int NT>
__device__ __forceinline__ int solve_free_set_thin(double *L, unsigned F_in, double eps, double *z_out, double *lam_out)
{
    const int lane = LANE;
    const unsigned F = (unsigned)__builtin_amdgcn_readfirstlane((int)F_in);       // wave-uniform by construction (ballots); make it a scalar
    const int nF = __popc(F);
    int ia = 0;
    {
        unsigned m = F;
#pragma unroll
        for (int c = 0; c < NT; c++) {                             // scalar bit scan; lane c learns its coefficient through one v_writelane
            const int jc = __builtin_amdgcn_readfirstlane(m ? __builtin_ctz(m) : 0);  // (a scalar register whatever the compiler thinks of F)
            m &= m - 1u;
            asm("v_writelane_b32 %0, %1, %2" : "+v"(ia) : "s"(jc), "n"(c));
        }
    }
    double uR[6] = {0, 0, 0, 0, 0, 0}, uL[6] = {0, 0, 0, 0, 0, 0};
    {
        const double *gi = L + P_GCOL + 6 * (ia & 15), *Wr = L + P_W + 72 * (ia >> 4);
        double g[6];
#pragma unroll
        for (int a = 0; a < 6; a++) g[a] = gi[a];
        if (F & 0xFFFFu) {                                         // wave-uniform: some right-foot coefficient is free
#pragma unroll
            for (int b = 0; b < 6; b++) { double sr = 0.0; for (int a = 0; a < 6; a++) sr += g[a] * Wr[12 * a + b]; uR[b] = sr; }
        }
        if (F >> 16) {
#pragma unroll
            for (int b = 0; b < 6; b++) { double sl = 0.0; for (int a = 0; a < 6; a++) sl += g[a] * Wr[12 * a + 6 + b]; uL[b] = sl; }
        }
    }
    double a[NT], b[1];
    if constexpr (NT == 16) thin_cols<NT, 0, 2>(L, nF, F, uR, uL, a);       // (9..16 free coefficients with a singular K_f: both feet)
    else {
        if ((F >> 16) == 0u) thin_cols<NT, 0, 0>(L, nF, F, uR, uL, a);       // wave-uniform three-way: no per-column choice between u_R and u_L in the usual cases
        else if ((F & 0xFFFFu) == 0u) thin_cols<NT, 0, 1>(L, nF, F, uR, uL, a);
        else thin_cols<NT, 0, 2>(L, nF, F, uR, uL, a);
    }
    b[0] = L[P_QV + ia];
    const int bad = ldl_solve_regs<NT, 1>(a, b, (nF >= 32) ? 0xFFFFFFFFu : ((1u << nF) - 1u), L + C_LS, eps);     // + eps I: added where the pivots are read
    const double z = (lane < nF) ? b[0] : 0.0;                     // z_r in lane r; lanes 8..15 of the row must read as zeros below
    // w = G_F z (lanes 0..11: component kk of foot lane / 6), y = W w, r = y - h
    const int l12 = (lane < 12) ? lane : 0;
    const bool leftlane = l12 >= 6;
    const int kk = l12 - (leftlane ? 6 : 0);
    double wz = 0.0;
    thin_wrench<NT, 0>(L, nF, F, z, kk, leftlane, wz);
    L[(lane < 12) ? P_W12 + lane : (int)P_DUMP] = wz;              // the wrench G c itself: if this set is accepted the recovery starts from it (cone_qp: w_done)
    double y = -L[P_H12 + l12];
    {
        const double *Wk = L + P_W + 12 * l12;
        double wr[12];
#pragma unroll
        for (int m2 = 0; m2 < 12; m2++) wr[m2] = Wk[m2];
        dpp_fmac_lane<0>(y, wz, wr[0]); dpp_fmac_lane<1, false>(y, wz, wr[1]); dpp_fmac_lane<2, false>(y, wz, wr[2]); dpp_fmac_lane<3, false>(y, wz, wr[3]);
        dpp_fmac_lane<4, false>(y, wz, wr[4]); dpp_fmac_lane<5, false>(y, wz, wr[5]); dpp_fmac_lane<6, false>(y, wz, wr[6]); dpp_fmac_lane<7, false>(y, wz, wr[7]);
        dpp_fmac_lane<8, false>(y, wz, wr[8]); dpp_fmac_lane<9, false>(y, wz, wr[9]); dpp_fmac_lane<10, false>(y, wz, wr[10]); dpp_fmac_lane<11, false>(y, wz, wr[11]);
    }
    WSYNC();                                                       // (the L rows parked by the solve are dead)
    L[(lane < 12) ? C_LS + lane : C_LS + 16 + (lane & 15)] = y;    // r = W w - h
    const int pos = __popc(F & ((1u << (lane & 31)) - 1u));
    const double zr = __shfl(z, pos & (NT - 1), 64);               // coefficient j <- row pos(j)
    const bool inF = (lane < 32) && ((F >> lane) & 1u);
    const double zj = inF ? zr : 0.0;
    WSYNC();
    double sj = 0.0;
    {
        const double *g = L + P_GCOL + 6 * (lane & 15), *rr = L + C_LS + 6 * ((lane >> 4) & 1);
#pragma unroll
        for (int k = 0; k < 6; k++) sj += g[k] * rr[k];
    }
    *z_out = zj;
    *lam_out = (lane < 32 && !inF) ? sj : 0.0;
    return bad;
}

// ---- push-through solve on a free set F (generalises the all-free fast path of cone_qp).
// With G_F the generators of the free coefficients and K = G_F G_F' = blkdiag(K_R, K_L) (6 x 6 per foot), the
// minimiser of 1/2 c'(G'WG + eps I)c - (G'h)'c over c_F (other coefficients 0) is c_F = G_F' y, y = K^-1 w, where the
// wrench w = G_F c_F solves the 12 x 12 SPD system (W + eps K^-1) w = h -- PROVIDED each foot with a free
// coefficient has K_f nonsingular (its free generators span the foot's wrench space).  Then also
// (P c - q)_j = g_j'(W w - h) = -eps g_j' y for j outside F, so one vector s_j = g_j' y (j < 32) carries both the
// coefficients (j in F) and the multipliers (-eps s_j, j not in F; for a foot without free coefficients y is
// replaced by -(W w - h)/eps, the residual itself).  Two 6 x 6 inversions (both feet at once, one
// per DPP row) and a 12 x 12 solve replace the |F| x |F| factorisation.  A foot without any free coefficient
// carries no force (its rows are dropped).  Returns 0 (wave-uniform) when some K_f is numerically singular: the
// caller then takes the general P_FF solve.
// EDGE CONTACT.  A foot whose free coefficients all sit on two vertices of one side of the sole (coefficient j = 4 vertex + ray; vertices 0, 2
// have p_y = +0.025, 1, 3 p_y = -0.025, 0, 1 p_x = 0.1, 2, 3 p_x = -0.05, Robot.cpp:38-42) pushes along a line: its wrench obeys tau_x = p_y f_z
// (side edges) or tau_y = -p_x f_z (front / back edge), K_f has rank 5 and the identity above does not apply as it stands.  It does in the
// five free wrench coordinates: with w = E w~ (E = I but for the row of the bound torque, which reads kappa times f_z; w~ has a 0 there),
// K~ = K_f without that row and column, the minimiser is c_F = G_F' y~ with y~ = K~^-1 w~ and (E'WE + eps K~^-1) w~ = E'h -- the same 12 x 12
// solve with one row / column folded into the f_z one and pinned (unit pivot, zero right-hand side).  The multipliers of the foot's other
// coefficients are g_j'(W w - h) with the true residual (its bound component is not -eps y~).  This is the double-support set right after a
// touch-down of the walking workload: F = 0f0f0f0f for ~130 of the 200 ticks of every double-support phase (scripts/route_probe.py), which
// the |F| x |F| solve answered in 16.8k cycles of such an evaluation (profiles/r04_prod_timeline.txt).
// d: 0 = tau_x bound, 1 = tau_y bound, -1 = none (wave-uniform, from the foot's 16-bit free mask)
__device__ __forceinline__ int edge_bound_row(unsigned Ff)
{
#ifdef LMH_NO_EDGE                                                  // checker build `noedge`: such sets go the register / general route as before round 4
    return -1;
#endif
    if (Ff == 0u) return -1;
    if ((Ff & ~0x0F0Fu) == 0u || (Ff & ~0xF0F0u) == 0u) return 0;
    if ((Ff & ~0x00FFu) == 0u || (Ff & ~0xFF00u) == 0u) return 1;
    return -1;
}
// K_f^-1 of both feet for the free set F into Kdst (2 x 36); `scr` = 160 doubles of scratch.  Returns non-zero
// (wave-uniform) when a foot with free coefficients has a singular K_f (edge contact: a singular K~; the row / column of the bound torque
// of the stored inverse are zero).
__device__ __forceinline__ int kinv_compute(double *L, unsigned F, double *Kdst, double *scr)
{
    const int lane = LANE, l16 = lane & 15, row = lane >> 4;
    const unsigned FR = F & 0xFFFFu, FL = F >> 16;
    const bool useR = FR != 0u, useL = FL != 0u;
    const int dR = edge_bound_row(FR), dL = edge_bound_row(FL);
    double *K = scr, *Ki = Kdst;
    WSYNC();
    {   // K_f = G diag(free_f) G' for both feet as ONE 16 x 16 x 16 matrix-core product: row block f of A carries foot f's
        // mask, so the two diagonal 6 x 6 blocks of the tile are K_R and K_L (the off-diagonal blocks are not used)
        auto a_g = [=](int m, int k) { const unsigned msk = (m < 6) ? FR : FL; const bool ok = (m < 12) && ((msk >> k) & 1u);
                                       return ldz(L, ok, P_GCOL + 6 * k + ((m < 6) ? m : m - 6), P_GCOL); };
        auto b_g = [=](int k, int n) { return ldz(L, n < 12, P_GCOL + 6 * k + ((n < 6) ? n : n - 6), P_GCOL); };
        const v4d kk = mfma_tile<4>(a_g, b_g);
        const int tr = lane & 15, tq = lane >> 4;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int rw = tq + 4 * g;
            const int fr = (rw >= 6) ? 1 : 0, fc = (tr >= 6) ? 1 : 0;
            if (rw < 12 && tr < 12 && fr == fc) K[36 * fr + 6 * (rw - 6 * fr) + (tr - 6 * fc)] = kk[g];
        }
    }
    WSYNC();
    int bad = 0;
    {   // both 6 x 6 inverses at once: DPP row 0 = right foot, DPP row 1 = left foot (six unit right-hand sides each)
        const bool rowon = (row == 0 && useR) || (row == 1 && useL);
        const bool on = rowon && l16 < 6;
        const int rb = (row < 2) ? 36 * row : 0, lr = (l16 < 6) ? l16 : 0;
        double a[6], bb[6], myinv = 0.0;
        const int dd = (row == 0) ? dR : dL;                       // bound torque row of this DPP row's foot (-1: none)
#pragma unroll
        for (int c = 0; c < 6; c++) {                              // full rows (Gauss-Jordan), both feet at once; the bound row / column pinned
            const double kv = K[rb + 6 * lr + c];
            a[c] = (l16 == dd || c == dd) ? ((l16 == c) ? 1.0 : 0.0) : kv;
            bb[c] = (l16 == c) ? 1.0 : 0.0;
        }
        // K_f entries are O(1e-3 .. 10); a rank-deficient block pivots at ~1e-17
        gj16_step<0>(a, bb, 0x3Fu, l16, rowon, 1e-12, bad, myinv);
#pragma unroll
        for (int c = 0; c < 6; c++) bb[c] *= myinv;
        if (row < 2 && l16 < 6) {
#pragma unroll
            for (int c = 0; c < 6; c++) Ki[36 * row + 6 * l16 + c] = (on && l16 != dd && c != dd) ? bb[c] : 0.0;       // K_f^-1 (symmetric); 0 for a foot without force
        }
    }
    WSYNC();
    return (__ballot(bad != 0) != 0ull) ? 1 : 0;
}

// `have_ki`: 1 = K_f^-1 of this F already sits in L[P_KI] (helper wave), 2 = known singular, 0 = compute here.
__device__ __forceinline__ int cone_pushthrough(double *L, LmhCParams &P, unsigned F, int have_ki, double *s_out, int *flags)
{
    const int lane = LANE;
    const unsigned FR = F & 0xFFFFu, FL = F >> 16;
    const bool useR = FR != 0u, useL = FL != 0u;
    double *Ki = (have_ki == 1) ? L + P_KI : L + C_LS + 72, *Yv = L + C_LS + 144;
    if (have_ki == 2) return 0;
    if (have_ki == 0 && kinv_compute(L, F, L + C_LS + 72, L + C_LS + 240)) return 0;     // wave-uniform: some K_f is singular
    WSYNC();
    const int dR = edge_bound_row(FR), dL = edge_bound_row(FL);    // edge contact: the bound torque row of the foot (-1: none); wave-uniform
    const double kR = L[P_GCOL + 6 * (FR ? __builtin_ctz(FR) : 0) + ((dR > 0) ? 1 : 0)], kL = L[P_GCOL + 6 * (FL ? __builtin_ctz(FL) : 0) + ((dL > 0) ? 1 : 0)];
    const double kapR = (dR >= 0) ? kR : 0.0, kapL = (dL >= 0) ? kL : 0.0;      // tau_bound = kappa f_z: p_y | -p_x of the edge, the generator's own entry
    {   // (E'WE + eps K~^-1) w~ = E'h on the rows of the feet that carry force (E = I, K~ = K without edge contact)
        double a[12], b[1];
        const int l16 = lane & 15, lr = (l16 < 12) ? l16 : 0, fi = (lr >= 6) ? 1 : 0, ri = lr - 6 * fi;          // (a copy of the system per DPP row)
        const bool rowuse = (l16 < 12) && ((fi == 0) ? useR : useL);
        const int dd = fi ? dL : dR;
        const bool rowpin = ri == dd, isz = (ri == 5) && (dd >= 0);
        const double kz = isz ? (fi ? kapL : kapR) : 0.0;          // the f_z row takes kappa times the bound row
        const double eps = P.eps_coeff;
        const double *W1 = L + P_W + 12 * lr, *W2 = L + P_W + 12 * (6 * fi + ((dd > 0) ? dd : 0)), *Kr = Ki + 36 * fi + 6 * ri;
        double base[12];
#pragma unroll
        for (int c = 0; c < 12; c++) base[c] = W1[c] + kz * W2[c];
        base[5] += kapR * ((dR > 0) ? base[1] : base[0]);           // ... and the f_z column kappa times the bound column (kappa = 0: no edge)
        base[11] += kapL * ((dL > 0) ? base[7] : base[6]);
#pragma unroll
        for (int c = 0; c < 12; c++) {                             // full rows, unconditional loads: rows / columns of a foot without force are
            const double g = Kr[c % 6];                            // never pivots (live mask), so their entries are don't-cares
            const bool colpin = (c % 6) == ((c < 6) ? dR : dL);
            const double v = base[c] + ((c / 6 == fi) ? eps * g : 0.0);
            a[c] = (rowpin || colpin) ? ((lr == c) ? 1.0 : 0.0) : v;
        }
        const double hv = L[P_H12 + lr] + kz * L[P_H12 + 6 * fi + ((dd > 0) ? dd : 0)];
        b[0] = rowpin ? 0.0 : hv;
        const unsigned live = (useR ? 0x03Fu : 0u) | (useL ? 0xFC0u : 0u);      // (a pinned row stays a pivot: skipping it measured neutral)
        if (gj_solve_regs<12, 1>(a, b, live)) *flags |= LMH_FLAG_NOT_SPD;
        // w = E w~: the bound torque is kappa f_z (lanes 5 / 11 of the lane's own 16-lane row hold f_z)
        const double fzR = bcast16<5>(b[0]), fzL = bcast16<11>(b[0]);
        const double wv = rowpin ? (fi ? kapL * fzL : kapR * fzR) : b[0];
        WSYNC();
        if (lane < 12) { const double wo = rowuse ? wv : 0.0; Yv[lane] = wo; L[P_W12 + lane] = wo; }      // w: the wrench G c itself (cone_qp: w_done if the round is accepted)
    }
    WSYNC();
    if (lane < 12) {                                               // y = K~^-1 w (free coefficients), -(W w - h) / eps (the others), foot by foot
        const int fi = lane / 6, ri = lane % 6;
        const bool used = (fi == 0) ? useR : useL, edge = (fi ? dL : dR) >= 0;
        double yv = 0.0, yn = 0.0;
        if (used) {
#pragma unroll
            for (int k = 0; k < 6; k++) yv += Ki[36 * fi + 6 * ri + k] * Yv[6 * fi + k];     // (the bound column of the inverse is zero)
            yn = yv;                                               // K regular: g_j'(W w - h) = -eps g_j'y for every coefficient of the foot
        }
        if (!used || edge) {
            // a foot with no free coefficient, or on an edge: the multipliers are g_j'(W w - h) with the foot's rows of the residual;
            // stored as -(W w - h)/eps so that the caller's -eps s_j reproduces them
            double rv = -L[P_H12 + lane];
#pragma unroll
            for (int k = 0; k < 12; k++) rv += L[P_W + 12 * lane + k] * Yv[k];
            yn = -rv / P.eps_coeff;
            if (!used) yv = yn;
        }
        Yv[12 + lane] = yv; Yv[24 + lane] = yn;
    }
    WSYNC();
    double sj = 0.0;
    if (lane < 32) {
        const bool fr = (F >> lane) & 1u;
        const double *g = L + P_GCOL + 6 * (lane & 15), *y = Yv + (fr ? 12 : 24) + 6 * (lane >> 4);
#pragma unroll
        for (int k = 0; k < 6; k++) sj += g[k] * y[k];
    }
    *s_out = sj;
    return 1;
}

// ================================================================== fp32 QP (LMH_PRECISION_FP32, BASELINE config 5's tolerance sweep)
// The same algebra as qp_setup15 / qp_setup + the push-through cone solves, every product, elimination and accumulation in fp32
// (the LDS slots stay 8 bytes wide and hold float values: no second layout).  Written for the sweep, not for speed: plain loops, one wave,
// in-LDS Gauss-Jordan.  What stays fp64: the references (clock / PD laws), the residual of the one refinement step of the 12 x 12
// push-through system, the sign tests of the active-set iteration, and the general P_FF route for a rank-deficient contact set
// (cond ~1e13: not representable in fp32; the instance is marked LMH_FLAG_QP_FP64_ROUTE).
__device__ __forceinline__ float ldf(const double *L, int i) { return (float)L[i]; }

// Gauss-Jordan on an augmented row-major array A (n rows, row stride ld; columns [0, n) the SPD matrix, [n, n + m) right-hand sides; on exit
// those columns hold the solutions).  Rows / columns whose bit is clear in `live` are skipped (their solutions are 0).  A pivot that is not
// above rel * (its original diagonal) marks the system singular.  Returns non-zero (wave-uniform) in that case.
__device__ __forceinline__ int gj_lds_f32(double *A, int n, int m, int ld, unsigned live, float rel)
{
    const int lane = LANE;
    int bad = 0;
    WSYNC();
    const float d0 = ldf(A, ((lane < n) ? lane : 0) * (ld + 1));
    for (int j = 0; j < n; j++) {
        if (!((live >> j) & 1u)) continue;                         // wave-uniform
        WSYNC();
        float d = ldf(A, j * ld + j);
        const float dj0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d0), j));
        if (!(d > rel * dj0) || !(d > 0.0f)) { bad = 1; d = 1.0f; }
        const float inv = 1.0f / d;
        const int cols = n + m - 1 - j, total = n * cols;
        for (int e = lane; e < total; e += 64) {
            const int i = e / cols, c = j + 1 + (e - i * cols);
            if (i != j && ((live >> i) & 1u)) {
                const float f = ldf(A, i * ld + j) * inv;
                A[i * ld + c] = (double)fmaf(-f, ldf(A, j * ld + c), ldf(A, i * ld + c));
            }
        }
    }
    WSYNC();
    for (int e = lane; e < n * m; e += 64) {
        const int i = e / m, r = e - i * m;
        const bool on = (live >> i) & 1u;
        const float dd = ldf(A, i * ld + i);
        A[i * ld + n + r] = on ? (double)(ldf(A, i * ld + n + r) / ((dd > 0.0f) ? dd : 1.0f)) : 0.0;
    }
    WSYNC();
    return bad;
}

// controller.cpp:94-132 + the equality blocks of :388-436 down to W, h, qv, in fp32.  nU = 15 or 18 rows of U = [AG ; J] (run time).
__device__ __forceinline__ int qp_setup_f32(double *L, LmhCParams &P)
{
    const int lane = LANE;
    int flags = 0;
    const int nU = (P.w_com_ang == 0.0) ? 15 : 18, r0 = 18 - nU, nA = nU + 7;
    const float idp = 1.0f / (float)P.w_base_pos, ida = 1.0f / (float)P.w_base_ang, idj = 1.0f / (float)P.w_joints;
    auto iD = [=](int i) { return (i < 3) ? idp : (i < 6) ? ida : idj; };
    WSYNC();
    for (int e = lane; e < nU * 30; e += 64) {
        const int r = r0 + e / 30, c = e % 30;
        L[F_U + e] = (double)(float)((r < 6) ? L[P_AG + 30 * r + c] : jdense(L, r - 6, c));
    }
    if (lane < nU) {
        const int rr = r0 + lane;
        const float om = (float)((rr < 3) ? P.w_com_ang : (rr < 6) ? P.w_com_lin : P.w_foot);
        const float beta = (rr < 6) ? (ldf(L, P_AGPQP + rr) - ldf(L, P_HREF + rr)) : (ldf(L, P_JPQP + rr - 6) - ldf(L, P_FREF + rr - 6));
        L[F_OB + lane] = (double)(om * beta);
        L[F_OB + 18 + lane] = (double)(1.0f / om);
        L[F_OB + 36 + lane] = (double)beta;
    }
    for (int e = lane; e < 210; e += 64) {
        const int i = e / 7, c = e % 7;
        L[F_BP + e] = (double)((c == 0) ? -ldf(L, P_QREF + i) : ldf(L, P_MTOP + 30 * (c - 1) + i) * iD(i));
    }
    WSYNC();
    if (lane < 30) {                                               // q = D^-1 U' Om beta
        float q = 0.0f;
        for (int r = 0; r < nU; r++) q = fmaf(ldf(L, F_U + 30 * r + lane), ldf(L, F_OB + r), q);
        L[F_BQ + lane] = (double)fmaf(q, iD(lane), ldf(L, F_BP + 7 * lane));
    }
    WSYNC();
    for (int e = lane; e < nU * nA; e += 64) {                     // [Cm | V] = [Om^-1 + U D^-1 U' | U (bp'_g + q), U bp'_M]
        const int r = e / nA, c = e - r * nA;
        float sacc;
        if (c < nU) {
            sacc = (r == c) ? ldf(L, F_OB + 18 + r) : 0.0f;
            for (int i = 0; i < 30; i++) sacc = fmaf(ldf(L, F_U + 30 * r + i) * iD(i), ldf(L, F_U + 30 * c + i), sacc);
        } else {
            const int n = c - nU;
            sacc = 0.0f;
            for (int i = 0; i < 30; i++) sacc = fmaf(ldf(L, F_U + 30 * r + i), (n == 0) ? ldf(L, F_BQ + i) : ldf(L, F_BP + 7 * i + n), sacc);
        }
        L[F_A + 25 * r + c] = (double)sacc;
    }
    if (gj_lds_f32(L + F_A, nU, 7, 25, (1u << nU) - 1u, 0.0f)) flags |= LMH_FLAG_NOT_SPD;
    if (lane < nU) L[F_A + 25 * lane + nU] = (double)(ldf(L, F_A + 25 * lane + nU) - ldf(L, F_OB + lane));     // t'' = t_g - ob
    WSYNC();
    for (int e = lane; e < 210; e += 64) {                         // Y = bp' - D^-1 U' t''  (stored transposed)
        const int n = e / 30, i = e % 30;
        float sacc = 0.0f;
        for (int r = 0; r < nU; r++) sacc = fmaf(ldf(L, F_U + 30 * r + i), ldf(L, F_A + 25 * r + nU + n), sacc);
        L[P_YT + e] = (double)fmaf(-sacc, iD(i), ldf(L, F_BP + 7 * i + n));
    }
    WSYNC();
    if (lane < 42) {                                               // S | d = Mb Y
        const int m = lane / 7, n = lane % 7;
        float sacc = 0.0f;
        for (int i = 0; i < 30; i++) sacc = fmaf(ldf(L, P_MTOP + 30 * m + i), ldf(L, P_YT + 30 * n + i), sacc);
        if (n == 0) L[P_D6 + m] = (double)(ldf(L, P_C + m) - sacc);
        else L[F_S + 12 * m + n - 1] = (double)sacc;
    }
    if (lane >= 64 - 36) { const int e = lane - (64 - 36), i = e / 6, j = e % 6; L[F_S + 12 * i + 6 + j] = (i == j) ? 1.0 : 0.0; }
    if (gj_lds_f32(L + F_S, 6, 6, 12, 0x3Fu, 0.0f)) flags |= LMH_FLAG_NOT_SPD;
    if (lane < 36) { const int i = lane / 6, j = lane % 6; L[P_SI + lane] = (double)(0.5f * (ldf(L, F_S + 12 * i + 6 + j) + ldf(L, F_S + 12 * j + 6 + i))); }
    WSYNC();
    for (int e = lane; e < 72; e += 64) {                          // T1 = Jb S^-1
        const int row = e / 6, k = e % 6;
        float sacc = 0.0f;
        for (int j = 0; j < 6; j++) sacc = fmaf((float)jdense(L, row, j), ldf(L, P_SI + 6 * j + k), sacc);
        L[F_T1 + e] = (double)sacc;
    }
    WSYNC();
    for (int e = lane; e < 156; e += 64) {                         // [W | h] = [w_force I + T1 Jb' | T1 d]
        const int m = e / 13, n = e % 13;
        float sacc = (m == n) ? (float)P.w_force : 0.0f;
        for (int k = 0; k < 6; k++) sacc = fmaf(ldf(L, F_T1 + 6 * m + k), (n < 12) ? (float)jdense(L, (n < 12) ? n : 0, k) : ldf(L, P_D6 + k), sacc);
        L[(n < 12) ? F_M + 12 * m + n : P_H12 + m] = (double)sacc;
    }
    WSYNC();
    for (int e = lane; e < 144; e += 64) { const int m = e / 12, n = e % 12; L[P_W + e] = (double)(0.5f * (ldf(L, F_M + 12 * m + n) + ldf(L, F_M + 12 * n + m))); }
    if (lane < 32) {
        // qv = G' h, accumulated in fp64 from the fp32 h: only the general route (fp64) and the tolerance scale read it, and that route
        // needs qv_F in range(G_F') to ~eps_coeff relative (P_FF has eigenvalues eps_coeff on null(G_F): a 1e-7 rounding of qv would come
        // back multiplied by 1e8)
        const int o = 6 * (lane / 16);
        double sacc = 0.0;
        for (int k = 0; k < 6; k++) sacc += L[P_GCOL + 6 * (lane & 15) + k] * L[P_H12 + o + k];
        L[P_QV + lane] = sacc;
    }
    WSYNC();
    return flags;
}

// (W + eps Ki) w = h on the rows of `live` in fp32 with one refinement step (residual accumulated in fp64, correction through the fp32
// inverse that the same Gauss-Jordan pass produced).  Ki = 2 x 36 (one 6 x 6 block per foot).  The result goes to L[F_V + 24 ..].
__device__ __forceinline__ int pushthrough_solve_f32(double *L, const double *Ki, float eps, unsigned live)
{
    const int lane = LANE;
    WSYNC();
    for (int e = lane; e < 12 * 25; e += 64) {
        const int i = e / 25, c = e % 25;
        float v;
        if (c < 12) {
            v = ldf(L, P_W + 12 * i + c);
            if (c / 6 == i / 6) v = fmaf(eps, (float)Ki[36 * (i / 6) + 6 * (i % 6) + c % 6], v);
            L[F_M + 12 * i + c] = (double)v;
        } else v = (c == 12) ? ldf(L, P_H12 + i) : ((c - 13 == i) ? 1.0f : 0.0f);
        L[F_PT + e] = (double)v;
    }
    const int bad = gj_lds_f32(L + F_PT, 12, 13, 25, live, 0.0f);
    if (lane < 12) {
        double r = 0.0;
        if ((live >> lane) & 1u) {
            r = L[P_H12 + lane];
            for (int c = 0; c < 12; c++) if ((live >> c) & 1u) r -= L[F_M + 12 * lane + c] * L[F_PT + 25 * c + 12];
        }
        L[F_V + 12 + lane] = (double)(float)r;
    }
    WSYNC();
    if (lane < 12) {
        float dw = 0.0f;
        for (int c = 0; c < 12; c++) dw = fmaf(ldf(L, F_PT + 25 * lane + 13 + c), ldf(L, F_V + 12 + c), dw);
        L[F_V + 24 + lane] = ((live >> lane) & 1u) ? (double)(ldf(L, F_PT + 25 * lane + 12) + dw) : 0.0;
    }
    WSYNC();
    return bad;
}

// fp32 form of cone_pushthrough: K_f = G_F G_F' and its inverse by Gauss-Jordan in fp32 (a pivot below 1e-4 of its diagonal = rank
// deficient: return 0, the caller takes the fp64 general route), the 12 x 12 solve above, y = K^-1 w, s_j = g_j' y.
// `fmask_out`: bit f set = foot f carries free coefficients (its multipliers are -eps s_j, accurate to eps * 1e-6 |s|).
__device__ __forceinline__ int cone_pushthrough_f32(double *L, LmhCParams &P, unsigned F, double *s_out, int *flags)
{
    const int lane = LANE;
    const unsigned FR = F & 0xFFFFu, FL = F >> 16;
    const bool useR = FR != 0u, useL = FL != 0u;
    const float eps = (float)P.eps_coeff;
    WSYNC();
    for (int e = lane; e < 144; e += 64) {                         // [K_f | I]
        const int ft = e / 72, a = (e % 72) / 12, c = e % 12;
        float v;
        if (c < 6) {
            v = 0.0f;
            for (unsigned m = ft ? FL : FR; m; m &= m - 1u) { const int j = __builtin_ctz(m); v = fmaf(ldf(L, P_GCOL + 6 * j + a), ldf(L, P_GCOL + 6 * j + c), v); }
        } else v = (c - 6 == a) ? 1.0f : 0.0f;
        L[F_K + e] = (double)v;
    }
    int sing = 0;
    if (useR) sing |= gj_lds_f32(L + F_K, 6, 6, 12, 0x3Fu, 1e-4f);
    if (useL) sing |= gj_lds_f32(L + F_K + 72, 6, 6, 12, 0x3Fu, 1e-4f);
    if (sing) return 0;
    for (int e = lane; e < 72; e += 64) {
        const int ft = e / 36, a = (e % 36) / 6, c = e % 6;
        const bool used = ft ? useL : useR;
        L[F_KI + e] = used ? (double)(0.5f * (ldf(L, F_K + 72 * ft + 12 * a + 6 + c) + ldf(L, F_K + 72 * ft + 12 * c + 6 + a))) : 0.0;
    }
    const unsigned live = (useR ? 0x03Fu : 0u) | (useL ? 0xFC0u : 0u);
    if (pushthrough_solve_f32(L, L + F_KI, eps, live)) *flags |= LMH_FLAG_NOT_SPD;
    if (lane < 12) {
        const int fi = lane / 6, ri = lane % 6;
        const bool used = (fi == 0) ? useR : useL;
        float yv;
        if (used) {
            yv = 0.0f;
            for (int k = 0; k < 6; k++) yv = fmaf(ldf(L, F_KI + 36 * fi + 6 * ri + k), ldf(L, F_V + 24 + 6 * fi + k), yv);
        } else {                                                   // foot without force: -(W w - h) / eps (the multipliers come out as g_j'(W w - h))
            double rv = -L[P_H12 + lane];
            for (int k = 0; k < 12; k++) rv += L[P_W + 12 * lane + k] * L[F_V + 24 + k];
            yv = -(float)rv / eps;
        }
        L[F_V + 36 + lane] = (double)yv;
    }
    WSYNC();
    float sj = 0.0f;
    if (lane < 32) {
        for (int k = 0; k < 6; k++) sj = fmaf(ldf(L, P_GCOL + 6 * (lane & 15) + k), ldf(L, F_V + 36 + 6 * (lane >> 4) + k), sj);
    }
    *s_out = (double)sj;
    return 1;
}

// min 1/2 c'Pc - qv'c  s.t. c >= 0, c_j = 0 for j in `forced`.  P = G'WG + eps I is SPD, so the
// minimiser is unique.  Fast path: block principal pivoting from the incoming free set (one solve when
// the active set did not change, typically <= 8 from a cold start).  If that has not settled after
// BPP_MAX rounds, a Lawson-Hanson active-set pass from the empty set finishes (monotone, finite).
// One loop, one call site of the (large, fully unrolled) free-set solve.
// P.bpp_max (lmh_config.bpp_rounds): 10 by default; < 0 skips block pivoting altogether (diagnostic: Lawson-Hanson from the empty set)
// qv = G' h, the linear term of the cone problem (the Hessian G'WG + eps I itself is only formed if the general free-set solve is needed,
// build_cone_matrix); G[k][j] is the generator of coefficient j (foot j / 16) in wrench rows 6 (j / 16) .. + 5.  Formed lazily: the all-free
// solve -- the usual double-support case -- works on (W, h) and never reads it.
__device__ __forceinline__ void qv_form(double *L)
{
    const int lane = LANE;
    if (lane < 32) {
        const int o = 6 * (lane / 16);
        double sacc = 0.0;
        for (int k = 0; k < 6; k++) sacc += L[P_GCOL + 6 * (lane & 15) + k] * L[P_H12 + o + k];
        L[P_QV + lane] = sacc;
    }
    WSYNC();
}
template <bool F32 = false>
__device__ __forceinline__ int cone_qp(double *L, LmhCParams &P, unsigned forced, unsigned *F_io, int *iters, int *w_done, double *dbgp = nullptr)
{
    constexpr double TOLC = F32 ? 1e-5 : 1e-10;                   // primal sign test, relative to max |c| (fp32: ~100 ulp of the push-through solve)
    const int lane = LANE;
    int flags = 0, it = 0;
    forced = (unsigned)__builtin_amdgcn_readfirstlane((int)forced);
    unsigned F = (unsigned)__builtin_amdgcn_readfirstlane((int)(*F_io & ~forced));    // scalar from here on (ballots keep it so)
    WSTAMP(30);
    RT_COUNT(4);
    *w_done = 0;
    const bool mine = (lane < 32) && !((forced >> lane) & 1u);
    int ninf = 33, budget = 3;
    const int bpp_max = P.bpp_max;
    bool lh = bpp_max < 0;                                         // false: block pivoting, true: Lawson-Hanson
    if (lh) F = 0u;
    if (dbgp) { build_cone_matrix(L, P); if constexpr (!F32) qv_form(L); }      // the debug record dumps the 32 x 32 cone Hessian (the solves never form it) and qv
    double cj = 0.0, lj = 0.0;
    if (!lh && forced == 0u && F == 0xFFFFFFFFu) {
        // every coefficient free (the usual balance case): then w = G c solves the 12 x 12 SPD system
        // (W + eps (G G')^-1) w = h and c = G'(G G')^-1 w  (push-through identity; G G' is constant,
        // block diagonal and well conditioned) -- 12 pivots instead of 32.
        it++;
        RT_COUNT(5);
        double zj = 0.0;
        if constexpr (F32) {
            WSYNC();
            for (int e = lane; e < 72; e += 64) L[F_KI + e] = L[P_GI6 + e % 36];
            if (pushthrough_solve_f32(L, L + F_KI, (float)P.eps_coeff, 0xFFFu)) flags |= LMH_FLAG_NOT_SPD;
            if (lane < 32) {
                float z = 0.0f;
                for (int k = 0; k < 6; k++) z = fmaf(ldf(L, P_GPI + 6 * (lane & 15) + k), ldf(L, F_V + 24 + 6 * (lane >> 4) + k), z);
                zj = (double)z;
            }
        } else {
        double a[12], b[1];
        {
            const int l16 = lane & 15, lr = (l16 < 12) ? l16 : 0, fi = (lr >= 6) ? 1 : 0, ri = lr - 6 * fi;      // (a copy of the system per DPP row)
            const double eps = P.eps_coeff;
#pragma unroll
            for (int c = 0; c < 12; c++) {                         // full rows, unconditional loads (Gauss-Jordan)
                const double g = L[P_GI6 + 6 * ri + c % 6];
                a[c] = L[P_W + 12 * lr + c] + ((c / 6 == fi) ? eps * g : 0.0);
            }
            b[0] = L[P_H12 + lr];
        }
        WSTAMP(32);
        if (gj_solve_regs<12, 1>(a, b, 0xFFFu)) flags |= LMH_FLAG_NOT_SPD;
        WSTAMP(33);
        WSYNC();
        if (lane < 12) { L[P_U12 + lane] = b[0]; L[P_W12 + lane] = b[0]; }     // the wrench itself: G c = G G'(G G')^-1 u = u
        {   // c = G'(G G')^-1 u: lanes 0..11 of every DPP row hold u, so the six terms of a foot come through row broadcasts (no way through LDS)
            const double *gp = L + P_GPI + 6 * (lane & 15);
            double g6[6], zr = 0.0, zl = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) g6[k] = gp[k];
            dpp_fmac_lane<0>(zr, b[0], g6[0]); dpp_fmac_lane<6, false>(zl, b[0], g6[0]); dpp_fmac_lane<1, false>(zr, b[0], g6[1]); dpp_fmac_lane<7, false>(zl, b[0], g6[1]);
            dpp_fmac_lane<2, false>(zr, b[0], g6[2]); dpp_fmac_lane<8, false>(zl, b[0], g6[2]); dpp_fmac_lane<3, false>(zr, b[0], g6[3]); dpp_fmac_lane<9, false>(zl, b[0], g6[3]);
            dpp_fmac_lane<4, false>(zr, b[0], g6[4]); dpp_fmac_lane<10, false>(zl, b[0], g6[4]); dpp_fmac_lane<5, false>(zr, b[0], g6[5]); dpp_fmac_lane<11, false>(zl, b[0], g6[5]);
            zj = (lane < 32) ? ((lane & 16) ? zl : zr) : 0.0;
        }
        }
        WSTAMP(34);
        const double cmax = wave_max(fabs(zj));
        const unsigned bad = (unsigned)__ballot(lane < 32 && zj < -TOLC * (1.0 + cmax));
        WSTAMP(35);
        cj = zj;
        if (bad == 0u) {
            RT_COUNT(0);
            WSYNC();
            if (lane < 32) L[P_CC + lane] = cj;
            WSYNC();
            *F_io = F; *iters = it;
            if constexpr (!F32) *w_done = 1;                       // P_W12 holds the wrench (the fp32 form leaves it to the recovery)
            return flags;
        }
        ninf = __popc(bad);
        F ^= bad;
    }
    // the dual sign tests of the iteration below scale with max |q| (the all-free path above has no dual side)
    if constexpr (!F32) qv_form(L);
    double qmax = (lane < 32) ? fabs(L[P_QV + lane]) : 0.0;
    qmax = wave_max(qmax);
    WSTAMP(31);
    const double toll = 1e-14 * (1.0 + qmax);                    // ~10x the round-off of (P c - q): a looser bound lets a warm start keep a coefficient
                                                                   // out whose multiplier is slightly negative (1e-6 relative error in tau after a contact switch)
    for (;;) {
        if (it >= P.max_qp_iters) { flags |= LMH_FLAG_QP_MAXITER; break; }    // no further solve is started once the cap is reached
        it++;
        double zj;
        if (dbgp && lane == 0 && it <= 12) dbgp[4020 + 2 * it] = (double)clock64();
        double sj;
        bool w_thin = false;                                       // this round's solve left the wrench G z in P_W12 (thin sets; push-through: w itself)
        const int have_ki = F32 ? 0 : __builtin_amdgcn_readfirstlane((F == (unsigned)L[P_KF + 1]) ? (int)L[P_KF + 2] : 0);   // prepared by the helper wave for the warm-start set
        double tolm = toll;                                        // dual sign test of this round
        bool pt;
        if constexpr (F32) pt = !lh && cone_pushthrough_f32(L, P, F, &sj, &flags);
        else pt = !lh && cone_pushthrough(L, P, F, have_ki, &sj, &flags);
        if (pt) {                                                  // 12 x 12 route: coefficients and multipliers from one vector
            RT_COUNT(1);
            w_thin = !F32;
            const bool fr = (lane < 32) && ((F >> lane) & 1u);
            zj = fr ? sj : 0.0;
            lj = (lane < 32 && !fr) ? -P.eps_coeff * sj : 0.0;
            if constexpr (F32) {
                // fp32 multipliers: -eps s_j on a foot with free coefficients (good to eps * 1e-6 |s|); on a foot without, g_j'(W w - h), a
                // difference of O(|h|) terms (good to 1e-6 |q|)
                const double smax = wave_max((lane < 32) ? fabs(sj) : 0.0);
                const bool footfree = ((lane & 16) ? (F >> 16) : (F & 0xFFFFu)) != 0u;
                tolm = footfree ? 2e-5 * P.eps_coeff * (1.0 + smax) : 2e-5 * (1.0 + qmax);
            }
        } else {
            if constexpr (F32) flags |= LMH_FLAG_QP_FP64_ROUTE;    // rank-deficient contact set (or the Lawson-Hanson pass): fp64 general route
#ifdef LMH_NO_THIN                                                  // experiment switch: the general route for every set
            if (false) {
#else
            if (__builtin_amdgcn_readfirstlane(__popc(F)) <= 8) {  // thin set (wave-uniform): everything in registers
#endif
                WSTAMP(80);
                RT_COUNT(2);
                if (solve_free_set_thin<8>(L, F, P.eps_coeff, &zj, &lj)) flags |= LMH_FLAG_NOT_SPD;
                w_thin = true;
                WSTAMP(81);
#ifdef LMH_THIN16                                                   // experiment switch: measured 17.0k cycles against the general route's 16.8k for |F| = 16 (a
            // 16-pivot dependent chain does not get cheaper by staying in registers), so the shipped kernel does not carry it; what removed that
            // solve from the walking workload is the edge-contact push-through (cone_pushthrough)
            } else if (__builtin_amdgcn_readfirstlane(__popc(F)) <= 16) {      // up to 16 rows: the same solve, one DPP row full
                WSTAMP(84);
                RT_COUNT(2);
                if (solve_free_set_thin<16>(L, F, P.eps_coeff, &zj, &lj)) flags |= LMH_FLAG_NOT_SPD;
                w_thin = true;
                WSTAMP(85);
#endif
            } else {
                WSTAMP(82);
                RT_COUNT(3);
                build_cone_rows(L, F, P.eps_coeff);                // P_FF only (the multipliers go through the wrench G z)
                if (solve_free_set(L, F, P.eps_coeff, &zj, &lj)) flags |= LMH_FLAG_NOT_SPD;
                WSTAMP(83);
            }
        }
        if (dbgp && lane == 0 && it <= 12) { dbgp[4021 + 2 * it] = (double)clock64(); dbgp[4050 + it] = (double)__popc(F); }
        const bool inF = (lane < 32) && ((F >> lane) & 1u);
        if (!lh) {
            const double cmax = wave_max(fabs(zj));
            const double tolc = TOLC * (1.0 + cmax);
            const bool isbad = mine && ((inF && zj < -tolc) || (!inF && lj < -tolm));
            const unsigned bad = (unsigned)__ballot(isbad);
            if (dbgp && lane == 0 && it <= 12) { dbgp[3960 + it] = (double)F; dbgp[3975 + it] = (double)bad; }   // round trace (diagnostics)
            cj = zj;
            if (bad == 0u) { if (!F32 && w_thin) *w_done = 1; break; }
            if (it >= bpp_max) { lh = true; F = 0u; cj = 0.0; continue; }    // next solve: F empty, lam = -qv
            const int nb = __popc(bad);
            if (nb < ninf) { ninf = nb; budget = 3; F ^= bad; }
            else if (budget > 0) { budget--; F ^= bad; }
            else F ^= (1u << (31 - __clz((int)bad)));              // Murty: flip the highest-index violator only
        } else {
            const bool neg = inF && !(zj > 0.0);
            const unsigned negm = (unsigned)__ballot(neg);
            if (negm == 0u) {
                cj = inF ? zj : 0.0;                               // accept, then test dual feasibility
                double best = (mine && !inF) ? -lj : -1.0e300;
                int bi = lane;
                for (int o = 32; o > 0; o >>= 1) {
                    const double ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                if (!(best > toll)) break;                         // optimal
                F |= (1u << bi);
            } else {                                               // step towards z until a coefficient hits zero
                double al = neg ? ((cj - zj > 0.0) ? cj / (cj - zj) : 0.0) : 1.0e300;
                double amin = al; int ai = lane;
                for (int o = 32; o > 0; o >>= 1) {
                    const double ob = __shfl_xor(amin, o, 64); const int oi = __shfl_xor(ai, o, 64);
                    if (ob < amin || (ob == amin && oi < ai)) { amin = ob; ai = oi; }
                }
                if (inF) cj = cj + amin * (zj - cj);
                const bool drop = neg && (cj <= 0.0 || lane == ai);
                const unsigned dm = (unsigned)__ballot(drop);
                if (drop) cj = 0.0;
                F &= ~dm;
            }
        }
    }
    WSYNC();
    if (lane < 32) L[P_CC + lane] = ((F >> lane) & 1u) ? cj : 0.0;
    WSYNC();
    *F_io = F;
    *iters = it;
    return flags;
}

// One 16 x 16 matrix-core tile from operands that are already laid out for it: lane l loads A[l & 15][4 kk + (l >> 4)] from a0[ASTEP kk]
// and B[4 kk + (l >> 4)][l & 15] from b0[BSTEP kk]; all 2 KSTEPS loads are issued before the first v_mfma, no select in between.
template <int KSTEPS, int ASTEP, int BSTEP>
__device__ __forceinline__ v4d mfma_ptr(const double *a0, const double *b0)
{
    double av[KSTEPS], bv[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk++) { av[kk] = a0[ASTEP * kk]; bv[kk] = b0[BSTEP * kk]; }
    // two accumulation chains (even / odd k-steps): a dependent v_mfma_f64_16x16x4 pair is ~97 cycles apart, independent ones issue every ~33
    v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KSTEPS; kk += 2) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
        if (kk + 1 < KSTEPS) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk + 1], bv[kk + 1], acc2, 0, 0, 0);
    }
    return acc + acc2;
}

// The Jacobian rows of U and U D^-1 (rows 0..11 of the padded operands).  They depend on wave 0's own products only and land in LDS that is
// dead once its Newton-Euler pass is over (S0 + [0, 408) and S0 + [544, 952): the NE sweeps' scratch and the FK transforms), so on the
// two-wave schedule wave 0 writes them while wave 1 is still inside CRBA / its reference chain, ahead of the join.
__device__ __forceinline__ void qp_prefill15(double *L, LmhCParams &P)
{
    // 12 rows x 32 columns, six entries per lane: lane = (row parity rh, column c); the entry of pass `it` is row rh + 2 it, i.e. foot
    // it / 3, row rh + 2 (it % 3) of that foot's compact store (12 columns: base 6 | own leg 6).  Dense column c of the right foot is compact
    // column c (c < 12); of the left foot c (c < 6) or c - 6 (12 <= c < 18).  Everything that depends on the lane is formed once; a pass is a
    // load, two multiplications (by 0 | 1 and by 0 | 1 / D_c) and two stores.
    const int lane = LANE, c = lane & 31, rh = lane >> 5;
    const double idp = P.inv_w_base_pos, ida = P.inv_w_base_ang, idj = P.inv_w_joints;
    const double iD = (c < 3) ? idp : (c < 6) ? ida : idj;
    const bool ok0 = c < 12, ok1 = (c < 6) || (c >= 12 && c < 18);
    const double m0 = ok0 ? 1.0 : 0.0, m1 = ok1 ? 1.0 : 0.0, d0 = ok0 ? iD : 0.0, d1 = ok1 ? iD : 0.0;
    const double *p0 = L + (P_JC + 12 * rh + (ok0 ? c : 0)), *p1 = L + (P_JC + 72 + 12 * rh + (ok1 ? ((c < 6) ? c : c - 6) : 0));
    double *o = L + (Q_U + 34 * rh + c);
#pragma unroll
    for (int it = 0; it < 6; it++) {
        const double v = (it < 3) ? p0[24 * it] : p1[24 * (it - 3)];
        o[68 * it] = v * ((it < 3) ? m0 : m1);
        o[68 * it + (Q_UD - Q_U)] = v * ((it < 3) ? d0 : d1);
    }
}

// QP set-up for the reference's weights (angular-momentum weight 0: U = [J ; AG_lin] has 15 rows; the row order is a free choice, the
// solve permutes with it), controller.cpp:94-132 + the equality
// blocks of :388-436, down to the cone problem data W, h, qv.  Same algebra as qp_setup<NU> below with two changes of association that
// shorten the leading wave's path:
//   * Cm ob - beta = U D^-1 U' ob, so the right-hand side fix-up becomes a column of bp' (q = D^-1 U' Om beta, formed beside the fills);
//   * S | d = Mb Y = Mb bp' - (Mb D^-1 U') t'' : Z = Mb D^-1 U' and Mb bp' do not depend on the 15 x 15 solve and are formed by the helper
//     wave while wave 0 solves; the Y tiles (needed only by the recovery) are formed by the helper wave while wave 0 goes on to S^-1, W, h.
// NW = 2 joins: fills | Cm, q+V | solve, Z+Mbp | (S..qv), Y | -> the caller's join in front of the cone solve.
// `slack` (rollout): called by the helper wave where it is ahead of wave 0 -- after its Z / Mb bp' tiles, while wave 0 is in the 15 x 15 solve.
struct NoWindow { __device__ __forceinline__ void operator()(int) const {} };
template <int NW, class SF = NoWindow, bool CGLIN = true>
__device__ __forceinline__ int qp_setup15(double *L, LmhCParams &P, int wid, double *dbgp, SF slack = SF())
{
    const int lane = LANE;
    int flags = 0;
    constexpr int nU = 15;
    const double idp = P.inv_w_base_pos, ida = P.inv_w_base_ang, idj = P.inv_w_joints;   // D^-1 (wave-uniform)
    const int tr = lane & 15, tq = lane >> 4;                      // fragment row / k-quarter; result rows tq + 4 reg, column tr
    // ---- fills: U, U D^-1 (padded 16 x 32; row order: 12 Jacobian rows, 3 linear-momentum rows, one zero row), bp'' (8 x 32), weights
    if constexpr (NW == 1) qp_prefill15(L, P);                     // NW = 2: wave 0 has already written the Jacobian rows (controller_eval)
    {   // lane = (row parity hi, column c); everything derived from the lane once, a pass is a load, one or two multiplications and the stores
        const int c = lane & 31, hi = lane >> 5, cs = (c < 30) ? c : 0;
        const double iDc = (c < 3) ? idp : (c < 6) ? ida : idj, dm = (c < 30) ? iDc : 0.0;          // 1 / D_c, 0 on the padding columns
        // NW = 2: the helper forms q (below) and the momentum rows of U, wave 0 AGpqp, bp'' and the weights: with everything but q on wave 0 the
        // helper waited ~0.7k cycles at this join, with the bp'' rows split as well wave 0 did (profiles/r04_barrier_share*.txt)
        if (wid == NW - 1) {                                       // rows 12..15 of U, U D^-1
            const double um = (c < 30) ? 1.0 : 0.0;
            const double *ag = L + (P_AG + 90 + 30 * hi + cs);     // linear-momentum rows 3..5 of AG -> operand rows 12..14; row 15 is zero
            double *o = L + (Q_U + 34 * (12 + hi) + c);
            const double v0 = ag[0], v1 = ag[(hi == 0) ? 60 : 0];
            o[0] = v0 * um; o[Q_UD - Q_U] = v0 * dm;
            o[68] = hi ? 0.0 : v1 * um; o[68 + (Q_UD - Q_U)] = hi ? 0.0 : v1 * dm;
        }
        // rows 0..6 of bp'' = [-qref | D^-1 Mb'] (row 7: q, below): row parity hi -> rows hi, hi + 4, then hi + 2 and 6
        const int n0 = hi;
        if (wid == 0) {
            const double *pa = L + ((n0 == 0) ? P_QREF + cs : P_MTOP + 30 * (n0 - 1) + cs);
            const double sa = (n0 == 0) ? ((c < 30) ? -1.0 : 0.0) : dm;
            L[Q_BPT + 34 * n0 + c] = pa[0] * sa;
            L[Q_BPT + 34 * (n0 + 4) + c] = L[P_MTOP + 30 * (n0 + 3) + cs] * dm;
            const int n1 = hi + 2;
            L[Q_BPT + 34 * n1 + c] = L[P_MTOP + 30 * (n1 - 1) + cs] * dm;
            if (hi == 0) L[Q_BPT + 34 * 6 + c] = L[P_MTOP + 30 * 5 + cs] * dm;
        }
    }
    if (wid == 0) {                                                // wave 0: it has just formed AGpqp (phase_qp)
        if (lane < 16) {                                           // Om_r beta_r | 1 / Om_r | beta_r  (row 15: zeros)
            const bool in = lane < nU;
            const int rr = (lane < 12) ? 6 + lane : in ? 3 + (lane - 12) : 3;      // row of [AG ; J] behind operand row `lane`
            const double om = (rr < 6) ? P.w_com_lin : P.w_foot;
            const double beta = (rr < 6) ? (L[P_AGPQP + rr] - L[P_HREF + rr]) : (L[P_JPQP + rr - 6] - L[P_FREF + rr - 6]);
            L[Q_OB + lane] = in ? om * beta : 0.0;
            L[Q_OB + 16 + lane] = in ? ((rr < 6) ? P.inv_w_com_lin : P.inv_w_foot) : 1.0;
            L[Q_OB + 32 + lane] = in ? beta : 0.0;
        }
        if (lane >= 32) L[Q_ZERO + lane - 32] = 0.0;
        if (lane >= 16 && lane < 24) L[Q_TT + 18 * (lane - 16) + 15] = 0.0;     // k = 15 padding of the right-hand sides
    }
    if (NW == 2 && wid == 1) {
        // q = D^-1 U' Om beta and with it row 7 of bp'' (the g column of V), already HERE on the two-wave schedule: the helper used to wait ~0.8k
        // cycles at the join below and then made wave 0 wait ~0.7k at the next one, behind this step (profiles/r04_barrier_share_mid2.txt).
        // It forms Om beta itself -- the same expressions wave 0 evaluates for Q_OB beside it -- in lane r of every 16-lane row and feeds it to
        // the column sums through DPP row broadcasts, so nothing has to come back through LDS: q_i = sum_r lane_r(Om beta) U[r][i].
        const int l16 = lane & 15;
        const bool in = l16 < nU;
        const int rr = (l16 < 12) ? 6 + l16 : in ? 3 + (l16 - 12) : 3;          // row of [AG ; J] behind operand row l16
        const int r3 = (rr < 6) ? rr - 3 : 0, rj = (rr >= 6) ? rr - 6 : 0;
        const double *T0 = L + P_TB, *cc = L + P_C, *pg = L + P_CG;
        double cg[6];
#pragma unroll
        for (int k = 3; k < 6; k++) cg[k] = CGLIN ? cc[k] - pg[k] : pg[k];        // Cg[3:6] (refs_agpqp: the same expressions)
        const double agp = T0[4 * r3] * cg[3] + T0[4 * r3 + 1] * cg[4] + T0[4 * r3 + 2] * cg[5];     // AGpqp, linear rows (refs_agpqp)
        const double bj = L[P_JPQP + rj] - L[P_FREF + rj];
        const double beta = (rr < 6) ? agp - L[P_HREF + rr] : bj;
        const double ob = in ? ((rr < 6) ? P.w_com_lin : P.w_foot) * beta : 0.0;
        const int ci = lane & 31;
        const double *ucol = L + (Q_U + ci), *agc = L + (P_AG + 90 + ((ci < 30) ? ci : 29));        // rows 12..14 of U = AG's linear rows (wave 0 is storing them beside this)
        double ur[nU];
#pragma unroll
        for (int r = 0; r < 12; r++) ur[r] = ucol[34 * r];
#pragma unroll
        for (int r = 12; r < nU; r++) ur[r] = agc[30 * (r - 12)];
        double q = 0.0;
        dpp_fmac_lane<0>(q, ob, ur[0]); dpp_fmac_lane<1, false>(q, ob, ur[1]); dpp_fmac_lane<2, false>(q, ob, ur[2]); dpp_fmac_lane<3, false>(q, ob, ur[3]);
        dpp_fmac_lane<4, false>(q, ob, ur[4]); dpp_fmac_lane<5, false>(q, ob, ur[5]); dpp_fmac_lane<6, false>(q, ob, ur[6]); dpp_fmac_lane<7, false>(q, ob, ur[7]);
        dpp_fmac_lane<8, false>(q, ob, ur[8]); dpp_fmac_lane<9, false>(q, ob, ur[9]); dpp_fmac_lane<10, false>(q, ob, ur[10]); dpp_fmac_lane<11, false>(q, ob, ur[11]);
        dpp_fmac_lane<12, false>(q, ob, ur[12]); dpp_fmac_lane<13, false>(q, ob, ur[13]); dpp_fmac_lane<14, false>(q, ob, ur[14]);
        const double iDi = (ci < 3) ? idp : (ci < 6) ? ida : idj;
        const double qr = -L[P_QREF + ((ci < 30) ? ci : 0)];
        L[Q_BPT + 34 * 7 + ci] = (ci < 30) ? qr + q * iDi : 0.0;   // (lanes 32..63 repeat lanes 0..31)
    }
    WSTAMP(10);
    bsync<NW>();
    WSTAMP(11);
    if (dbgp && LANE == 0) dbgp[4070] = (double)clock64();
    const double *zero = L + Q_ZERO;
    const double *u_row = L + Q_U + 34 * tr + tq;                  // A fragment of U (row tr < 16: row 15 is zero)
    // ---- Cm = Om^-1 + U D^-1 U'  (wave 0)  |  q, then V = U [bp'_g + q | bp'_M]  (helper wave)
    if (NW == 1 || wid == 1) {
        if constexpr (NW == 1) {
        if (lane < 32) {                                           // q_i = D^-1_i sum_r U[r][i] (Om beta)_r ; row 7 of bp'' = -qref + q
            double q = 0.0;
#pragma unroll
            for (int r = 0; r < nU; r++) q += L[Q_U + 34 * r + lane] * L[Q_OB + r];
            const double iDi = (lane < 3) ? idp : (lane < 6) ? ida : idj;
            L[Q_BPT + 34 * 7 + lane] = L[Q_BPT + lane] + q * iDi;  // columns 30, 31: 0 + 0
        }
        WSYNC();
        }
        const double *b_row = (tr == 0) ? L + Q_BPT + 34 * 7 + tq : (tr < 7) ? L + Q_BPT + 34 * tr + tq : zero;
        const v4d vv = mfma_ptr<8, 4, 4>(u_row, b_row);
#pragma unroll
        for (int g = 0; g < 4; g++) {                              // V[r][n] -> TT[n][r], r = tq + 4 g < 16 (row 15 of U is zero: TT[n][15] = 0)
            const int r = tq + 4 * g;
            L[(tr < 8) ? Q_TT + 18 * tr + r : Q_TRASH + lane] = vv[g];
        }
    }
    if (NW == 1 || wid == 0) {
        const v4d cm = mfma_ptr<8, 4, 4>(u_row, L + Q_UD + 34 * tr + tq);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int r = tq + 4 * g;
            L[Q_CM + 17 * r + tr] = cm[g] + ((r == tr) ? L[Q_OB + 16 + r] : 0.0);
        }
    }
    WSTAMP(12);
    bsync<NW>();
    WSTAMP(13);
    // ---- Cm t = V (7 right-hand sides): wave 0  |  Z = Mb D^-1 U', Mb bp': helper wave
    if (NW == 1 || wid == 0) {
        double a[nU], bb[7];
        const int lr = lane & 15;                                  // every 16-lane DPP row carries a copy of the system (gj_solve_regs)
#pragma unroll
        for (int c = 0; c < nU; c++) a[c] = L[Q_CM + 17 * lr + c];
#pragma unroll
        for (int r = 0; r < 7; r++) bb[r] = L[Q_TT + 18 * r + lr];
        WSTAMP(14);
        if (gj_solve_regs<nU, 7>(a, bb, (1u << nU) - 1u)) flags |= LMH_FLAG_NOT_SPD;
        bb[0] -= L[Q_OB + lr];                                     // t'' = t_g - ob in column 0
        if (lane < 16) {
#pragma unroll
            for (int r = 0; r < 7; r++) L[Q_TT + 18 * r + lane] = (lane < nU) ? bb[r] : 0.0;
        }
        WSTAMP(15);
    }
    if (NW == 1 || wid == 1) {
        const double *m_row = (tr < 6) ? L + P_MTOP + 30 * tr + tq : zero;        // k = 30, 31 read the next row's first entries: finite, times the zero padding of B
        const v4d zz = mfma_ptr<8, 4, 4>(m_row, (tr < nU) ? L + Q_UD + 34 * tr + tq : zero);
        const v4d mb = mfma_ptr<8, 4, 4>(m_row, (tr < 7) ? L + Q_BPT + 34 * tr + tq : zero);
#pragma unroll
        for (int g = 0; g < 2; g++) {                              // rows m = tq + 4 g < 6
            const int m = tq + 4 * g;
            L[(m < 6) ? Q_Z + 18 * m + tr : Q_TRASH + lane] = zz[g];
            L[(m < 6 && tr < 8) ? Q_MBP + 8 * m + tr : Q_TRASH + lane] = mb[g];
        }
        if constexpr (NW == 2) slack(0);                           // (the helper waits ~1.7k cycles at this join otherwise: profiles/r04_barrier_share_mid2.txt)
    }
    bsync<NW>();
    WSTAMP(16);
    if (dbgp && LANE == 0) dbgp[4074] = (double)clock64();
    const double *t_row = (tr < 7) ? L + Q_TT + 18 * tr + tq : zero;              // B fragment of t'' (column tr)
    // ---- helper wave: Y = bp' - D^-1 U' t''  (30 x 7, stored transposed; only the recovery reads it)
    if (NW == 1 || wid == 1) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const v4d yy = mfma_ptr<4, 136, 4>(L + Q_U + 34 * tq + 16 * mt + tr, t_row);        // A[m = i][k] = U[k][i]
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int i = 16 * mt + tq + 4 * g;
                const double iDi = (i < 3) ? idp : (i < 6) ? ida : idj;
                const bool ok = (i < 30) && (tr < 7);
                const double bpv = L[ok ? Q_BPT + 34 * tr + i : Q_ZERO];
                L[ok ? P_YT + 30 * tr + i : Q_TRASH + lane] = bpv - yy[g] * iDi;
            }
        }
        WSTAMP(17);
        // K_f^-1 of the warm-start set (rollout; usually nothing to do: the set has not moved): behind the Y tiles, beside wave 0's chain S .. qv
        if constexpr (NW == 2) slack(1);
    }
    // ---- wave 0: S | d = Mb bp' - Z t''  ->  S^-1  ->  T1 = Jb S^-1  ->  [W | h]  ->  qv
    if (NW == 1 || wid == 0) {
        {
            const v4d sy = mfma_ptr<4, 4, 4>((tr < 6) ? L + Q_Z + 18 * tr + tq : zero, t_row);
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int m = tq + 4 * g;
                const bool ok = (m < 6) && (tr < 7);
                const double val = L[ok ? Q_MBP + 8 * m + tr : Q_ZERO] - sy[g];
                const double cv = L[P_C + ((m < 6) ? m : 0)];
                L[!ok ? Q_TRASH + lane : (tr == 0) ? P_D6 + m : Q_S + 7 * m + (tr - 1)] = (tr == 0) ? cv - val : val;
            }
        }
        WSYNC();
        WSTAMP(19);
        {   // Si = S^-1: six unit right-hand sides.  S is only moderately conditioned as far as LDL' is concerned, but Gauss-Jordan loses
            // ~cond(S) more digits and leaves S^-1 (hence W) unsymmetric at the 1e-10 level, which the active-set tests of the cone QP
            // (tolerances ~1e-14) do not survive: LDL' on the lower triangle here.
            double a[6], bb[6];
            const int l16 = lane & 15, lr = (l16 < 6) ? l16 : 0;   // (a copy of the system per DPP row)
            // Gauss-Jordan on the full S, then S^-1 <- (S^-1 + S^-T) / 2: plain Gauss-Jordan leaves S^-1 (hence W) unsymmetric at the 1e-10 level (it loses
            // ~cond(S) more digits than LDL'), which the active-set tests of the cone QP (tolerances ~1e-14) do not survive; symmetrised, W is symmetric to
            // round-off again and the solve is 1.5k cycles shorter than LDL' with its parked factor (LMH_LDL_SI keeps that form for comparison)
#pragma unroll
            for (int c = 0; c < 6; c++) { a[c] = L[Q_S + 7 * lr + c]; bb[c] = (l16 == c) ? 1.0 : 0.0; }
            if (gj_solve_regs<6, 6>(a, bb, 0x3Fu)) flags |= LMH_FLAG_NOT_SPD;
#pragma unroll
            for (int c = 0; c < 6; c++) L[(lane < 6) ? Q_LS + 6 * lane + c : Q_TRASH + lane] = bb[c];
            WSYNC();
            {
                const int e = (lane < 36) ? lane : 0, i = e / 6, j = e % 6;
                const double v = 0.5 * (L[Q_LS + 6 * i + j] + L[Q_LS + 6 * j + i]);
                L[(lane < 36) ? P_SI + e : Q_TRASH + lane] = v;    // (for the recovery; the tiles below symmetrise their own fragment)
            }
        }
        WSTAMP(20);
        {   // [W | h] = [w_force I + Jb Si Jb' | Jb Si d] as Jb (Si [Jb' | d]): the result fragment of the inner product, D[tq + 4 g][tr] in register g,
            // IS the B fragment B[4 kk + tq][tr] of the outer one for kk = g, so the 6 x 13 intermediate never leaves the registers (the form
            // (Jb Si) Jb' went through LDS twice: T1 out and back in the other fragment shape, behind a stored Si).
            const bool k2 = tq < 2, r6 = tr < 6;
            const int m6 = r6 ? tr : 0;
            const double *ls = L + Q_LS;
            const double s0 = 0.5 * (ls[6 * m6 + tq] + ls[6 * tq + m6]);                        // A[m = tr][k = tq] = Si[tr][tq], symmetrised
            const double s1 = 0.5 * (ls[6 * m6 + (k2 ? 4 + tq : 0)] + ls[6 * (k2 ? 4 + tq : 0) + m6]);   // k = 4 + tq < 6
            const double sa0 = r6 ? s0 : 0.0, sa1 = (r6 && k2) ? s1 : 0.0;
            const double *jb = L + ((tr < 12) ? P_JC + 12 * tr + tq : (tr == 12) ? P_D6 + tq : P_D6);   // B[k][n] = Jb[n][k] | d[k] (n = 12) | 0
            const double jv0 = jb[0], jv1 = jb[k2 ? 4 : 0];
            const double b0 = (tr < 13) ? jv0 : 0.0, b1 = (tr < 13 && k2) ? jv1 : 0.0;
            v4d xt = {0.0, 0.0, 0.0, 0.0};
            xt = __builtin_amdgcn_mfma_f64_16x16x4f64(sa0, b0, xt, 0, 0, 0);
            xt = __builtin_amdgcn_mfma_f64_16x16x4f64(sa1, b1, xt, 0, 0, 0);                    // xt[g] = (Si [Jb' | d])[tq + 4 g][tr]; rows 6, 7 are zero
            const double a0 = (tr < 12) ? jv0 : 0.0, a1 = (tr < 12 && k2) ? jv1 : 0.0;          // A[m = tr][k] = Jb[tr][k]: the same loads
            v4d ww = {0.0, 0.0, 0.0, 0.0};
            ww = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, xt[0], ww, 0, 0, 0);
            ww = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, xt[1], ww, 0, 0, 0);
            const double wf = P.w_force;
#pragma unroll
            for (int g = 0; g < 3; g++) {
                const int m = tq + 4 * g;
                L[(tr < 12) ? P_W + 12 * m + tr : (tr == 12) ? P_H12 + m : Q_TRASH + lane] = ww[g] + ((m == tr) ? wf : 0.0);
            }
        }
        WSTAMP(22);
        if (dbgp && LANE == 0) dbgp[4011] = (double)clock64();
        WSYNC();                                                   // (qv = G' h: cone_qp, only if the all-free solve does not settle it)
        WSTAMP(23);
    }
    return flags;
}

// QP set-up (controller.cpp:94-132 Hessian/gradient and the equality blocks of :388-436) down to the cone
// problem data P, qv; NU = rows of U = [AG ; J] that carry weight (15 when the angular-momentum weight is 0).
template <int NU, int NW>
__device__ __forceinline__ int qp_setup(double *L, LmhCParams &P, int wid, double *dbgp)
{
    const int lane = LANE;
    int flags = 0;
    // ---- rows of U = [AG ; J] with weights Om, skipping zero-weight rows
    constexpr int nU = NU, r0 = 18 - NU;
    const double idp = P.inv_w_base_pos, ida = P.inv_w_base_ang, idj = P.inv_w_joints;   // D^-1 (wave-uniform)
    const int tr = lane & 15, tq = lane >> 4;                      // MFMA result: rows tq + 4 reg, column tr
    for (int e = lane + 64 * wid; e < nU * 30; e += 64 * NW) {
        const int r = r0 + e / 30, c = e % 30;
        L[B_U + e] = (r < 6) ? L[P_AG + 30 * r + c] : jdense(L, r - 6, c);
    }
    if (wid == 0 && lane < nU) {                                   // Om_r * beta_r , 1 / Om_r , beta_r
        const int rr = r0 + lane;
        const double om = (rr < 3) ? P.w_com_ang : (rr < 6) ? P.w_com_lin : P.w_foot;
        const double beta = (rr < 6) ? (L[P_AGPQP + rr] - L[P_HREF + rr]) : (L[P_JPQP + rr - 6] - L[P_FREF + rr - 6]);
        L[B_OB + lane] = om * beta;
        L[B_OB + 18 + lane] = 1.0 / om;
        L[B_OB + 36 + lane] = beta;
    }
    // bp' = [-qref | D^-1 Mb']  (the U' Om beta part of g_a is folded into the right-hand side below:
    // U bp_g = Cm ob - beta - U qref  and  Y_g = D^-1 U'(ob - t_g) - qref; controller.cpp:127-132)
    for (int e = lane + 64 * wid; e < 210; e += 64 * NW) {
        const int i = e / 7, cidx = e % 7;
        const double iDi = (i < 3) ? idp : (i < 6) ? ida : idj;
        L[B_BP + e] = (cidx == 0) ? -L[P_QREF + i] : L[P_MTOP + 30 * (cidx - 1) + i] * iDi;
    }
    WSTAMP(10);
    bsync<NW>();
    WSTAMP(11);
    if (dbgp && LANE == 0) dbgp[4070] = (double)clock64();
    // ---- Cm = Om^-1 + U D^-1 U'  and  V = U bp'  on the matrix cores (K = 30 padded to 32)
    const int ld = 19;
    {
        auto a_u = [=](int m, int k) { return ldz(L, m < nU && k < 30, B_U + 30 * m + k, B_U); };
        // D^-1 entry of column k = 4 kk + kq: one two-way select per (compile-time) kk -- a three-way select of the
        // captured scalars is lowered to a scratch-resident table lookup
        auto b_ud = [=](int k, int n, int kk, int kq) {
            const double sc = (kk == 0) ? ((kq < 3) ? idp : ida) : (kk == 1) ? ((kq < 2) ? ida : idj) : idj;
            return ldz(L, n < nU && k < 30, B_U + 30 * n + k, B_U) * sc;
        };
        auto b_bp = [=](int k, int n) { return ldz(L, n < 7 && k < 30, B_BP + 7 * k + n, B_BP); };
        const bool do_cm = (NW == 1) || (wid == 0), do_vv = (NW == 1) || (wid == 1);    // NW = 2: one tile per wave
        v4d cm = {0.0, 0.0, 0.0, 0.0}, vv = {0.0, 0.0, 0.0, 0.0};
        if (do_cm) cm = mfma_tile<8>(a_u, b_ud);
        if (do_vv) vv = mfma_tile<8>(a_u, b_bp);
        if constexpr (NU > 16) if (wid == 0) {                                             // rows/cols 16, 17 (angular-momentum weight set): plain loops
            for (int e = lane; e < 2 * 18; e += 64) {
                const int r = 16 + e / 18, c = e % 18;
                if (c <= r) {
                    double sacc = (r == c) ? L[B_OB + 18 + r] : 0.0;
                    for (int i = 0; i < 30; i++) sacc += L[B_U + 30 * r + i] * L[B_U + 30 * c + i] * ((i < 3) ? idp : (i < 6) ? ida : idj);
                    L[B_K + ld * r + c] = sacc;
                }
            }
            for (int e = lane; e < 2 * 7; e += 64) {
                const int r = 16 + e / 7, cidx = e % 7;
                double sacc = 0.0;
                for (int i = 0; i < 30; i++) sacc += L[B_U + 30 * r + i] * L[B_BP + 7 * i + cidx];
                L[B_K + ld * (nU + cidx) + r] = sacc;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int row = tq + 4 * g;
            if (do_cm && row < nU && tr <= row) L[B_K + ld * row + tr] = cm[g] + ((row == tr) ? L[B_OB + 18 + row] : 0.0);
            if (do_vv && row < nU && tr < 7) L[B_K + ld * (nU + tr) + row] = vv[g];
            if (do_cm && row < nU && tr < nU) L[B_CF + 18 * row + tr] = cm[g] + ((row == tr) ? L[B_OB + 18 + row] : 0.0);   // full copy for Cm ob
        }
    }
    WSTAMP(12);
    bsync<NW>();
    WSTAMP(13);
    if (wid == 0) {                                                // wave 0: right-hand side fix-up and the nU x nU factorisation
    if (lane < nU) {                                               // V_g += Cm ob - beta
        double sacc = -L[B_OB + 36 + lane];
        for (int c = 0; c < nU; c++) {
            const int hi = (c > lane) ? c : lane, lo = (c > lane) ? lane : c;
            const double cmv = (hi < 16) ? L[B_CF + 18 * lane + c] : L[B_K + ld * hi + lo];
            sacc += cmv * L[B_OB + c];
        }
        L[B_K + ld * nU + lane] += sacc;
    }
    WSYNC();
    WSTAMP(14);
    if (dbgp && LANE == 0) dbgp[4073] = (double)clock64();
    {   // Cm t = V for the 7 right-hand sides, row per lane in registers
        double a[NU], bb[7];
        const bool on = lane < nU;
        if constexpr (NU <= 16) {                                  // Gauss-Jordan on full rows
            const int lr = ((lane & 15) < nU) ? (lane & 15) : 0;   // (a copy of the system per DPP row)
#pragma unroll
            for (int c = 0; c < NU; c++) a[c] = L[B_CF + 18 * lr + c];
#pragma unroll
            for (int r = 0; r < 7; r++) bb[r] = L[B_K + ld * (nU + r) + lr];
            if (gj_solve_regs<NU, 7>(a, bb, (1u << NU) - 1u)) flags |= LMH_FLAG_NOT_SPD;
        } else {
#pragma unroll
            for (int c = 0; c < NU; c++) a[c] = (on && c <= lane) ? L[B_K + ld * lane + c] : 0.0;
#pragma unroll
            for (int r = 0; r < 7; r++) bb[r] = on ? L[B_K + ld * (nU + r) + lane] : 0.0;
            if (ldl_solve_regs<NU, 7>(a, bb, (1u << NU) - 1u, L + B_LS)) flags |= LMH_FLAG_NOT_SPD;
        }
        if (on) {
            bb[0] -= L[B_OB + lane];                               // t' = t_g - ob in column 0
#pragma unroll
            for (int r = 0; r < 7; r++) L[B_K + ld * (nU + r) + lane] = bb[r];
        }
    }
    }
    WSTAMP(15);
    bsync<NW>();
    WSTAMP(16);
    if (dbgp && LANE == 0) dbgp[4074] = (double)clock64();
    // ---- Y = bp' - D^-1 U' t'   (30 x 7; two row tiles, K = nU padded to 20)
    {
        auto b_t = [=](int k, int n) { return ldz(L, k < nU && n < 7, B_K + ld * (nU + n) + k, B_K); };
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            if (NW == 2 && mt != wid) continue;                    // one row tile per wave
            auto a_ut = [=](int m, int k) { const int i = 16 * mt + m; return ldz(L, i < 30 && k < nU, B_U + 30 * k + i, B_U); };
            const v4d yy = mfma_tile<5>(a_ut, b_t);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int i = 16 * mt + tq + 4 * g;
                if (i < 30 && tr < 7) L[P_YT + 30 * tr + i] = L[B_BP + 7 * i + tr] - yy[g] * ((i < 3) ? idp : (i < 6) ? ida : idj);
            }
        }
    }
    WSTAMP(17);
    bsync<NW>();
    WSTAMP(18);
    if (dbgp && LANE == 0) dbgp[4010] = (double)clock64();
    // ---- S = Mb Y_M (6x6), d = C_b - Mb Y_g  (one tile, K = 30 padded to 32); Si = S^-1
    if (wid == 0) {                                                // a dependent chain of small products: wave 0
    {
        auto a_m = [=](int m, int k) { return ldz(L, m < 6 && k < 30, P_MTOP + 30 * m + k, P_MTOP); };
        auto b_y = [=](int k, int n) { return ldz(L, n < 7 && k < 30, P_YT + 30 * n + k, P_YT); };
        const v4d sy = mfma_tile<8>(a_m, b_y);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int row = tq + 4 * g;
            if (row < 6 && tr == 0) L[P_D6 + row] = L[P_C + row] - sy[g];
            if (row < 6 && tr >= 1 && tr < 7) L[B_S + 7 * row + (tr - 1)] = sy[g];
        }
    }
    WSYNC();
    WSTAMP(19);
    {   // Si = S^-1: six unit right-hand sides
        // S is only moderately conditioned as far as LDL' is concerned, but Gauss-Jordan loses ~cond(S) more digits and leaves S^-1 (hence W)
        // unsymmetric at the 1e-10 level, which the active-set tests of the cone QP (tolerances ~1e-14) do not survive: LDL' here.
        double a[6], bb[6];
        const int lr = (lane < 6) ? lane : 0;
#pragma unroll
        for (int c = 0; c < 6; c++) { const double sv = L[B_S + 7 * lr + c]; a[c] = (lane < 6 && c <= lane) ? sv : 0.0; bb[c] = (lane == c) ? 1.0 : 0.0; }
        if (ldl_solve_regs<6, 6>(a, bb, 0x3Fu, L + B_LS)) flags |= LMH_FLAG_NOT_SPD;
        if (lane < 6) {
#pragma unroll
            for (int c = 0; c < 6; c++) L[P_SI + 6 * lane + c] = bb[c];
        }
    }
    WSYNC();
    WSTAMP(20);
    // ---- T1 = Jb Si (12x6);  [W | h] = [w_force I + T1 Jb' | T1 d]   (K = 6 padded to 8)
    {
        auto a_jb = [=](int m, int k) { const bool ok = m < 12 && k < 6; const double v = jdense(L, ok ? m : 0, ok ? k : 0); return ok ? v : 0.0; };
        auto b_si = [=](int k, int n) { return ldz(L, k < 6 && n < 6, P_SI + 6 * k + n, P_SI); };
        const v4d t1 = mfma_tile<2>(a_jb, b_si);
#pragma unroll
        for (int g = 0; g < 4; g++) { const int row = tq + 4 * g; if (row < 12 && tr < 6) L[B_T1 + 6 * row + tr] = t1[g]; }
    }
    WSYNC();
    WSTAMP(21);
    {
        auto a_t1 = [=](int m, int k) { return ldz(L, m < 12 && k < 6, B_T1 + 6 * m + k, B_T1); };
        auto b_jd = [=](int k, int n) { const bool kin = k < 6; const double vj = jdense(L, (n < 12) ? n : 0, kin ? k : 0); const double vd = L[P_D6 + (kin ? k : 0)]; return !kin ? 0.0 : (n < 12) ? vj : (n == 12) ? vd : 0.0; };
        const v4d ww = mfma_tile<2>(a_t1, b_jd);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int row = tq + 4 * g;
            if (row < 12 && tr < 12) L[P_W + 12 * row + tr] = ww[g] + ((row == tr) ? P.w_force : 0.0);
            if (row < 12 && tr == 12) L[P_H12 + row] = ww[g];
        }
    }
    }
    WSTAMP(22);
    if (dbgp && LANE == 0) dbgp[4011] = (double)clock64();
    // ---- qv = G' h (the cone Hessian G'WG + eps I itself is only formed if the general free-set solve is needed,
    //      build_cone_matrix); G[k][j] is the generator of coefficient j (foot j/16) in wrench rows 6 (j/16) .. +5
    // (qv is formed by the cone solve when it needs it: cone_qp / qv_form)
    if (wid == 0) WSYNC();
    WSTAMP(23);
    return flags;
}

// K_f^-1 of the free set the cone solve will start from (same rule as phase_qp: previous active set minus the coefficients of feet out of
// support), prepared by the helper wave; scratch: the CRBA parking area.  Needs the support phase of THIS evaluation in L[P_RPH].
__device__ __forceinline__ void kinv_prework(double *L, LmhCParams &P)
{
    WSTAMP(63);
    const int ph0 = __builtin_amdgcn_readfirstlane((int)L[P_RPH]);
    unsigned forced = 0u;
    if (ph0 == LMH_PHASE_LEFT || ph0 == LMH_PHASE_FLIGHT) forced |= 0x0000FFFFu;
    if (ph0 == LMH_PHASE_RIGHT || ph0 == LMH_PHASE_FLIGHT) forced |= 0xFFFF0000u;
    const unsigned Fpub = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)L[P_KF]);
    const unsigned F0 = (P.warm_start ? Fpub : 0xFFFFFFFFu) & ~forced;
    // the set the stored inverse belongs to has not moved (the usual case from one RK4 stage to the next): nothing to do
    if (__builtin_amdgcn_readfirstlane((int)(F0 == (unsigned)L[P_KF + 1] && L[P_KF + 2] != 0.0)) != 0) return;
    int st = 0;
    if (F0 != 0xFFFFFFFFu) {                                       // all free: the constant table is used instead
        // a foot with 1..5 free generators has K_f = sum of fewer than six rank-one terms: singular without looking (the usual case in
        // single support, where the sole presses on an edge)
        const int nR = __popc(F0 & 0xFFFFu), nL = __popc(F0 >> 16);
        const bool thin = (nR > 0 && nR < 6) || (nL > 0 && nL < 6);
        st = thin ? 2 : (kinv_compute(L, F0, L + P_KI, L + A_XR) ? 2 : 1);
    }
    if (LANE == 0) { L[P_KF + 1] = (double)F0; L[P_KF + 2] = (double)st; }
}

// Controller::WBC Hessian/gradient + solveQP (controller.cpp:94-132,388-479), see file header.
// PIPE (rollout): the helper wave prepares K_f^-1 at the end of its set-up share instead of at the start of the evaluation, and spends the
// cone solve -- wave 0 alone -- inside `window` (the next evaluation's clock references and kinematics, lmh_rollout_kernel).
template <int NW, bool F32 = false, bool PIPE = false, class WF = NoWindow, bool CGLIN = true>
__device__ __forceinline__ int phase_qp(double *L, LmhCParams &P, int ph, int wid, unsigned *Fmask_io, int *iters_out, double *dbgp = nullptr, WF window = WF())
{
    const int lane = LANE;
    int flags;
    if (NW == 2 && wid == 0) { refs_agpqp<CGLIN>(L, L[P_MODEL + 392], (P.w_com_ang != 0.0) || (dbgp != nullptr)); WSYNC(); }     // first point where both M (wave 1) and Cg (wave 0) exist
    if constexpr (F32) {
        if (NW == 2 && wid != 0) { bsync<NW>(); return 0; }        // fp32 QP: one wave, the helper waits for the recovery
        flags = qp_setup_f32(L, P);
    } else {
    // helper wave, PIPE: K_f^-1 (scratch S0 + [2300, 2480): above every array of the set-up), then the part of the look-ahead that needs no
    // scratch; where the helper has slack (qp_setup15), else at the end of its share
    // slack(0): its own RK4 stage, beside wave 0's 15 x 15 solve; slack(1): K_f^-1 of the warm-start set if it moved, behind the Y tiles
    auto slack = [&](int part) { if constexpr (PIPE) { if (part == 0) window(0); else kinv_prework(L, P); } };
    if (P.w_com_ang == 0.0) flags = qp_setup15<NW, decltype(slack), CGLIN>(L, P, wid, dbgp, slack);
    else {
        flags = qp_setup<18, NW>(L, P, wid, dbgp);
        if (NW == 2 && wid == 1) { slack(0); slack(1); }
    }
    WSTAMP(18);
    bsync<NW>();                                                   // Y (helper wave) is complete; the cone solve may overwrite the set-up scratch
    if (NW == 2 && wid != 0) {                                     // the active-set iteration and the recovery are sequential: wave 0
        if constexpr (PIPE) window(1);
        WSTAMP(26); bsync<NW>(); WSTAMP(27);
        return 0;
    }
    }
    if (dbgp && LANE == 0) dbgp[4012] = (double)clock64();
    // ---- bound-constrained QP  min 1/2 c'Pc - qv'c, c >= 0  (forced zeros for feet out of support)
    unsigned forced = 0u;
    if (ph == LMH_PHASE_LEFT || ph == LMH_PHASE_FLIGHT) forced |= 0x0000FFFFu;    // right foot carries no force
    if (ph == LMH_PHASE_RIGHT || ph == LMH_PHASE_FLIGHT) forced |= 0xFFFF0000u;
    unsigned F = (P.warm_start ? *Fmask_io : 0xFFFFFFFFu) & ~forced;
    int it = 0, w_done = 0;
    WSTAMP(24);
    flags |= cone_qp<F32>(L, P, forced, &F, &it, &w_done, dbgp);
    WSYNC();
    WSTAMP(25);
    *Fmask_io = F;
    *iters_out = it;
    if (lane == 0) L[P_KF] = (double)F;                            // published for the helper wave (next evaluation's warm start)
    if (dbgp && LANE == 0) dbgp[4013] = (double)clock64();
    if constexpr (F32) {                                           // the recovery below in fp32
        if (lane < 12) {
            const int ft = lane / 6, k = lane % 6;
            float sacc = 0.0f;
            for (int j = 0; j < 16; j++) sacc = fmaf(ldf(L, P_GCOL + 6 * j + k), ldf(L, P_CC + 16 * ft + j), sacc);
            L[P_W12 + lane] = (double)sacc;
        }
        WSYNC();
        if (lane < 6) {
            float sacc = 0.0f;
            for (int row = 0; row < 12; row++) sacc = fmaf((float)jdense(L, row, lane), ldf(L, P_W12 + row), sacc);
            L[F_V + lane] = (double)(sacc - ldf(L, P_D6 + lane));
        }
        WSYNC();
        if (lane < 6) {
            float lam = 0.0f;
            for (int k = 0; k < 6; k++) lam = fmaf(ldf(L, P_SI + 6 * lane + k), ldf(L, F_V + k), lam);
            L[P_LAM6 + lane] = (double)(-lam);
        }
        WSYNC();
        if (lane < 30) {
            float sacc = ldf(L, P_YT + lane);
            for (int k = 0; k < 6; k++) sacc = fmaf(ldf(L, P_YT + 30 * (1 + k) + lane), ldf(L, P_LAM6 + k), sacc);
            L[P_A + lane] = (double)(-sacc);
        }
        WSYNC();
        bsync<NW>();
        return flags;
    }
    // ---- recover w = G c (unless the solve left the wrench behind), lam = -Si (Jb' w - d), a = -(Y_g + Y_M lam)
    if (!w_done) {                                                 // wave-uniform
        if (lane < 12) {
            const int ft = lane / 6, k = lane % 6;
            double s = 0.0;
            for (int j = 0; j < 16; j++) s += L[P_GCOL + 6 * j + k] * L[P_CC + 16 * ft + j];
            L[P_W12 + lane] = s;
        }
        WSYNC();
    }
    {   // the three products in registers: lane l16 of every DPP row holds w[l16], r and lam come out in lanes 0..5 of every row, and each
        // sum takes its terms through row broadcasts -- one read of the wrench instead of three LDS round trips on the path both waves wait for
        const int l16 = lane & 15, i6 = (l16 < 6) ? l16 : 0, j = lane & 31, js = (j < 30) ? j : 0;
        const double wv = L[P_W12 + ((l16 < 12) ? l16 : 0)];
        const double *jc = L + P_JC + i6, *si = L + P_SI + 6 * i6, *yt = L + P_YT + js;      // base columns of the feet Jacobian: P_JC[foot][row][0..5]
        double jm[12], sm[6], ym[6];
#pragma unroll
        for (int row = 0; row < 12; row++) jm[row] = jc[72 * (row / 6) + 12 * (row % 6)];
#pragma unroll
        for (int k = 0; k < 6; k++) { sm[k] = si[k]; ym[k] = yt[30 * (1 + k)]; }
        double r0 = -L[P_D6 + i6], r1 = 0.0, s = yt[0];
        dpp_fmac_lane<0>(r0, wv, jm[0]); dpp_fmac_lane<1, false>(r1, wv, jm[1]); dpp_fmac_lane<2, false>(r0, wv, jm[2]); dpp_fmac_lane<3, false>(r1, wv, jm[3]);
        dpp_fmac_lane<4, false>(r0, wv, jm[4]); dpp_fmac_lane<5, false>(r1, wv, jm[5]); dpp_fmac_lane<6, false>(r0, wv, jm[6]); dpp_fmac_lane<7, false>(r1, wv, jm[7]);
        dpp_fmac_lane<8, false>(r0, wv, jm[8]); dpp_fmac_lane<9, false>(r1, wv, jm[9]); dpp_fmac_lane<10, false>(r0, wv, jm[10]); dpp_fmac_lane<11, false>(r1, wv, jm[11]);
        const double r = r0 + r1;                                  // r = Jb' w - d
        double nl = 0.0;
        dpp_fmac_lane<0>(nl, r, sm[0]); dpp_fmac_lane<1, false>(nl, r, sm[1]); dpp_fmac_lane<2, false>(nl, r, sm[2]);
        dpp_fmac_lane<3, false>(nl, r, sm[3]); dpp_fmac_lane<4, false>(nl, r, sm[4]); dpp_fmac_lane<5, false>(nl, r, sm[5]);
        const double lam = -nl;                                    // lam = -Si r  (nothing else reads it)
        dpp_fmac_lane<0>(s, lam, ym[0]); dpp_fmac_lane<1, false>(s, lam, ym[1]); dpp_fmac_lane<2, false>(s, lam, ym[2]);
        dpp_fmac_lane<3, false>(s, lam, ym[3]); dpp_fmac_lane<4, false>(s, lam, ym[4]); dpp_fmac_lane<5, false>(s, lam, ym[5]);
        if (lane < 30) L[P_A + lane] = -s;                         // a = -(Y_g + Y_M lam)
    }
    WSYNC();
    WSTAMP(26);
    bsync<NW>();                                                   // the helper wave takes over from here: torques, then K^-1 and references of the next evaluation
    WSTAMP(27);
    return flags;
}

// ---- build-defined plant (lmh_config.plant): forward dynamics driven by the torques the WBC returns, with a spring-damper
// contact at the sole vertices.  The controller's bias C' = C(q, v_prev) belongs to the PREVIOUS call's velocity (controller.cpp:56 runs
// before :59); the plant's belongs to the state being integrated, C = C(q, v) (second Newton-Euler pass, phase_newton_euler<R, true>).
// With tau = (M a + C' - J'w)[6:30] and the floating-base rows of the QP, S'tau = M a + C' - J'w, so
//     M qdd_plant = S'tau + J'w_c - C   <=>   qdd_plant = a + M^-1 (J'(w_c - w) + C' - C):
// the plant's acceleration is the controller's plus the response to the contact-wrench error and to the lag of the controller's velocity
// products.  M = [Ic0 F2; F2' H] with H block diagonal
// per limb (legs 6x6, arms 5x5, head 2x2): the four limb blocks are eliminated at once on the four DPP rows (Gauss-Jordan, 7 right-hand
// sides [r_l | F2_l']), the 6x6 base Schur complement (not symmetric: the reference's inertia typos) by one more Gauss-Jordan.
enum { PL_0 = S0 + 384,     // above S0 + [0, 384): the next evaluation's world transforms may already be there (look-ahead kinematics)
       PL_VF = PL_0 + 0,    // 8 x 6 : r_v x f_v | f_v per vertex
       PL_DW = PL_0 + 48,   // 12 : w_c - w
       PL_R = PL_0 + 60,    // 30 : J'(w_c - w)
       PL_B = PL_0 + 90,    // 24 x 7 : H^-1 [r_J | F2']
       PL_SB = PL_0 + 258,  // 6 x 7 : Schur complement | right-hand side
       PL_DAB = PL_0 + 300, // 6
       PL_AP = PL_0 + 306,  // 30 : plant acceleration (WBC coordinates)
       PL_ZERO = PL_0 + 336 }; // 32
__device__ __forceinline__ void phase_plant(double *L, LmhCParams &P)
{
    const int lane = LANE;
    if (lane < 8) {                                                // one lane per (foot, vertex)
        const int ft = lane >> 2, vi = lane & 3;
        const double pwx = (vi < 2) ? 0.1 : -0.05, pwy = (vi & 1) ? -0.025 : 0.025, pwz = 0.0;        // Robot.cpp:38-42
        // offsets are world-aligned for the flat foot (controller.cpp:225-270) and turn with it: r_v = R_sole Rf_q0' p_v
        double p[3];
#pragma unroll
        for (int a = 0; a < 3; a++) p[a] = c_rdes[a] * pwx + c_rdes[3 + a] * pwy + c_rdes[6 + a] * pwz;
        const double *T = L + P_TB + 12 * (1 + ft), *om = L + P_VFOOT + 6 * ft, *vo = om + 3;
        double rp[3], x[3];
#pragma unroll
        for (int a = 0; a < 3; a++) { rp[a] = T[4 * a] * p[0] + T[4 * a + 1] * p[1] + T[4 * a + 2] * p[2]; x[a] = rp[a] + T[4 * a + 3]; }
        const double xd0 = vo[0] + (om[1] * rp[2] - om[2] * rp[1]);
        const double xd1 = vo[1] + (om[2] * rp[0] - om[0] * rp[2]);
        const double xd2 = vo[2] + (om[0] * rp[1] - om[1] * rp[0]);
        const double pen = -x[2];
        double fn = P.contact_k * pen - P.contact_d * xd2;
        fn = (fn < 0.0) ? 0.0 : fn;
        double ftx = -P.contact_dt * xd0, fty = -P.contact_dt * xd1;
        const double ft2 = sqrt(ftx * ftx + fty * fty), lim = P.contact_mu * fn;
        const double sc = (ft2 > lim) ? lim / ft2 : 1.0;
        const bool in = pen > 0.0;
        const double f0 = in ? ((ft2 > lim) ? ftx * sc : ftx) : 0.0, f1 = in ? ((ft2 > lim) ? fty * sc : fty) : 0.0, f2 = in ? fn : 0.0;
        double *o = L + PL_VF + 6 * lane;
        o[0] = rp[1] * f2 - rp[2] * f1; o[1] = rp[2] * f0 - rp[0] * f2; o[2] = rp[0] * f1 - rp[1] * f0;
        o[3] = f0; o[4] = f1; o[5] = f2;
    }
    if (lane >= 32) L[PL_ZERO + lane - 32] = 0.0;
    WSYNC();
    if (lane < 12) {                                               // wrench about the sole origin, vertices summed in order
        const int ft = lane / 6, k = lane % 6;
        const double *v = L + PL_VF + 24 * ft + k;
        const double wc = (((0.0 + v[0]) + v[6]) + v[12]) + v[18];
        L[PL_DW + lane] = wc - L[P_W12 + lane];
    }
    WSYNC();
    if (lane < 30) {                                               // r = J'(w_c - w) (base and leg columns only) + C(q, v_prev) - C(q, v)
        double r = 0.0;
        if (lane < 18) {
#pragma unroll
            for (int row = 0; row < 12; row++) r += jdense(L, row, lane) * L[PL_DW + row];
        }
        L[PL_R + lane] = r + (L[P_C + lane] - L[P_VHS + lane]);    // P_VHS: the plant's own Newton-Euler pass at the current velocity
    }
    WSYNC();
    {   // limb blocks: DPP row dr = limb (RL, LL, RA, LA), lane l16 < 6 = joint of the limb (arms: a unit row pads 5 -> 6)
        const int dr = lane >> 4, l16 = lane & 15;
        const int nl = (dr < 2) ? 6 : 5, js = (dr == 0) ? 0 : (dr == 1) ? 6 : (dr == 2) ? 12 : 17;
        const bool real = l16 < nl;
        const int ja = js + (real ? l16 : 0);
        double a[6], b[7];
#pragma unroll
        for (int c = 0; c < 6; c++) { const double hv = L[P_HL + 6 * ja + ((c < 5) ? c : (dr < 2 ? 5 : 0))]; a[c] = (real && c < nl) ? hv : ((l16 == c && l16 < 6) ? 1.0 : 0.0); }
        { const double rv = L[PL_R + 6 + ja]; b[0] = real ? rv : 0.0; }
#pragma unroll
        for (int m = 0; m < 6; m++) { const double fv = L[P_MTOP + 30 * m + 6 + ja]; b[1 + m] = real ? fv : 0.0; }
        int bad = 0;
        double myinv = 0.0;
        gj16_step<0>(a, b, 0x3Fu, l16, true, 0.0, bad, myinv);
#pragma unroll
        for (int c = 0; c < 7; c++) L[real ? PL_B + 7 * ja + c : Q_TRASH + lane] = b[c] * myinv;
    }
    if (lane < 14) {                                               // head: 2 x 2 in closed form
        const int i = lane / 7, c = lane % 7;
        const double h00 = L[P_HL + 6 * 22], h01 = L[P_HL + 6 * 22 + 1], h10 = L[P_HL + 6 * 23], h11 = L[P_HL + 6 * 23 + 1];
        const double r0 = (c == 0) ? L[PL_R + 6 + 22] : L[P_MTOP + 30 * (c - 1) + 6 + 22], r1 = (c == 0) ? L[PL_R + 6 + 23] : L[P_MTOP + 30 * (c - 1) + 6 + 23];
        const double det = h00 * h11 - h01 * h10;
        L[PL_B + 7 * (22 + i) + c] = (i == 0) ? (h11 * r0 - h01 * r1) / det : (h00 * r1 - h10 * r0) / det;
    }
    WSYNC();
    {   // base: S_b = Ic0 - F2 B_M, rhs = r_b - F2 B_r   (one matrix-core tile, K = 24)
        const int tr = lane & 15, tq = lane >> 4;
        const double *zero = L + PL_ZERO;
        const v4d fb = mfma_ptr<6, 4, 28>((tr < 6) ? L + P_MTOP + 30 * tr + 6 + tq : zero, (tr < 7) ? L + PL_B + 7 * tq + tr : zero);
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int m = tq + 4 * g;
            const bool ok = (m < 6) && (tr < 7);
            const double base = L[!ok ? PL_ZERO : (tr == 0) ? PL_R + m : P_MTOP + 30 * m + (tr - 1)];
            L[ok ? PL_SB + 7 * m + tr : Q_TRASH + lane] = base - fb[g];
        }
    }
    WSYNC();
    {
        double a[6], b[1];
        const int l16 = lane & 15, lr = (l16 < 6) ? l16 : 0;       // (a copy of the system per DPP row)
#pragma unroll
        for (int c = 0; c < 6; c++) a[c] = L[PL_SB + 7 * lr + 1 + c];
        b[0] = L[PL_SB + 7 * lr];
        (void)gj_solve_regs<6, 1>(a, b, 0x3Fu);
        if (lane < 6) L[PL_DAB + lane] = b[0];
    }
    WSYNC();
    if (lane < 30) {
        double da;
        if (lane < 6) da = L[PL_DAB + lane];
        else {
            const double *B = L + PL_B + 7 * (lane - 6);
            da = B[0];
#pragma unroll
            for (int n = 0; n < 6; n++) da -= B[1 + n] * L[PL_DAB + n];
        }
        L[PL_AP + lane] = L[P_A + lane] + da;
    }
    WSYNC();
}

// Controller::WBC tail (controller.cpp:134-153): tau, base acceleration back to the world frame.
// The torques are not needed by the integrator: on the two-wave schedule the helper wave computes them while wave 0
// already updates the state and runs the next forward kinematics.
__device__ __forceinline__ void phase_outputs_tau(double *L)
{
    const int lane = LANE;
    if (lane < 24) {
        const int a = lane, st = f_jstart(a), ft = (a < 6) ? 0 : (a < 12) ? 1 : -1;
        const int nl = (a < 12) ? 6 : (a < 22) ? 5 : 2;
        double s = 0.0;
        for (int c = 0; c < 6; c++) s += L[P_MTOP + 30 * c + 6 + a] * L[P_A + c];
        for (int b = 0; b < nl; b++) s += L[P_HL + 6 * a + b] * L[P_A + 6 + st + b];
        double jw = 0.0;
        if (ft >= 0) for (int rr = 0; rr < 6; rr++) jw += L[P_JC + 72 * ft + 12 * rr + 6 + (a - 6 * ft)] * L[P_W12 + 6 * ft + rr];
        L[P_TAU + a] = s + L[P_C + 6 + a] - jw;
    }
    WSYNC();
}
__device__ __forceinline__ void phase_outputs_qdd(double *L, int a_src = P_A)
{
    const int lane = LANE;
    if (lane >= 32 && lane < 38) {                                 // X0 acc = a[0:6]: w = R0 a_ang ; v = R0 (a_lin - B0 w)
        const int k = lane - 32, r = k % 3;
        const double *E0 = L + P_X0, *B0 = L + P_X0 + 12, *a = L + a_src;
        const double w0 = E0[0] * a[0] + E0[1] * a[1] + E0[2] * a[2];
        const double w1 = E0[3] * a[0] + E0[4] * a[1] + E0[5] * a[2];
        const double w2 = E0[6] * a[0] + E0[7] * a[1] + E0[8] * a[2];
        double val;
        if (k < 3) {                                               // linear part goes first in qdd
            const double u0 = a[3] - (B0[0] * w0 + B0[1] * w1 + B0[2] * w2);
            const double u1 = a[4] - (B0[3] * w0 + B0[4] * w1 + B0[5] * w2);
            const double u2 = a[5] - (B0[6] * w0 + B0[7] * w1 + B0[8] * w2);
            val = E0[3 * r] * u0 + E0[3 * r + 1] * u1 + E0[3 * r + 2] * u2;
        } else val = (r == 0) ? w0 : (r == 1) ? w1 : w2;
        L[P_QDD + k] = val;
    }
    if (lane >= 40 && lane < 64) L[P_QDD + 6 + (lane - 40)] = L[a_src + 6 + (lane - 40)];
    WSYNC();
}

// one controller evaluation on the state in L[P_Q], L[P_V], L[P_VP] at time t.  NW = 2: both waves of the robot
// call this with their wave index; every bsync below is reached by both (uniform control flow), all other fences
// are wave-local.  Wave 1 never touches P_Q / P_V / the QP scratch after its last bsync, so wave 0 may run ahead
// into the next evaluation's forward kinematics.
// PIPE (rollout kernel only): the kinematics of this evaluation were run ahead by the helper wave, inside the previous evaluation's
// `window` (phase_qp), and so were the clock-only references; wave 0 starts at the X images.
template <int NW, typename R, bool QF32 = false, bool PIPE = false, class WF = NoWindow>
__device__ __forceinline__ int controller_eval(double *L, LmhCParams &P, int inst, double t, int wid, unsigned *Fmask, int *k_out, int *iters_out, double *dbg, bool need_tau = true, WF window = WF(),
                                               const IbPack *ibp = nullptr)
{
    int flags = 0, ph = 0;
    // in-kernel stamps (debug build of the kernel only): s_memtime at the phase boundaries
#define STAMP(i) do { if (dbg && LANE == 0) dbg[(wid ? 3950 : 4000) + (i)] = (double)clock64(); } while (0)   // wave 1 (diagnostic two-wave debug kernel): 3950..
    STAMP(0);
    WSTAMP(0);
    IbSel ibsel;                                                   // gather tables of the matrix-core CRBA: constant-memory loads, issued long before their use
    if constexpr (std::is_same_v<R, double>) { if (NW == 1 || wid == 1) ibsel = ibp ? ib_unpack(*ibp) : ib_select(); }
    if (NW == 2 && wid == 1 && !PIPE) {
        // while wave 0 runs the forward kinematics: the clock-only references, then K_f^-1 of the free set the cone solve will start from
        refs_prepare(L, P, inst, t);
        WSYNC();
        kinv_prework(L, P);
    }
    if constexpr (NW == 1) {                                       // single-wave schedule: no K^-1 prepared; references first
        if (LANE == 0) L[P_KF + 2] = 0.0;
        refs_prepare(L, P, inst, t);
        WSYNC();
    }
    if (wid == 0 && !PIPE) phase_fk<R>(L, P.gcol + 228);
    WSTAMP(1);
    const bool plant = P.plant != 0;                               // wave-uniform
    // PIPE: the join that ended the previous evaluation already published the world transforms, and until the next join neither wave
    // writes what the other reads (wave 0: qdd, the state, X images of frames 0..13, persistent copies; wave 1: torques, CoM, frames
    // 14..27) -- except the plant, whose idle-lane stores (Q_TRASH) fall on wave 1's share of A_XP
    if (!PIPE || plant) bsync<NW>();
    WSTAMP(2);
    STAMP(1);
    phase_com_x<NW, R>(L, wid);
    WSTAMP(3);
    bsync<NW>();
    WSTAMP(4);
    STAMP(2);
    if (dbg) {
        for (int e = LANE; e < 336; e += 64) dbg[e] = L[A_T + e];
        for (int e = LANE; e < 252; e += 64) {                     // E (= A') and B of every frame, the record layout of the parity tests
            const int i = e / 9, a = (e % 9) / 3, b = e % 3;
            dbg[336 + e] = L[A_XF + 36 * i + 6 * b + a]; dbg[672 + e] = L[A_XF + 36 * i + 6 * (3 + a) + b];
        }
        for (int e = LANE; e < 84; e += 64) dbg[588 + e] = L[A_XP + e];
    }
    STAMP(3);
    if constexpr (NW == 1) {
        phase_newton_euler<R>(L);
        if (plant) phase_newton_euler<R, true>(L);                 // the plant's velocity products at the CURRENT velocity (before CRBA reuses the scratch)
        STAMP(4);
        phase_crba<R>(L, ibsel);
        STAMP(5);
        phase_jacobian<R>(L);
        if constexpr (std::is_same_v<R, double>) refs_jpqp(L);
    } else {
        if (wid == 0) {                                            // LDS regions of the three are disjoint
            phase_newton_euler<R>(L);
            if (plant) phase_newton_euler<R, true>(L);
            STAMP(4);
            WSTAMP(64);
            phase_jacobian<R>(L);
            if constexpr (std::is_same_v<R, double>) refs_jpqp(L);
        }
        else phase_crba<R>(L, ibsel);
        STAMP(5);                                                  // per wave: end of its share of the tree phases
    }
    WSTAMP(5);
    // no join here: each wave's reference chain reads only its own tree products (chain A: M; chain B: J, T) and what the com_x join
    // already published; AGpqp, which needs both, is formed after the next join (phase_qp)
    WSTAMP(6);
    STAMP(6);
    const bool ang = (P.w_com_ang != 0.0) || (dbg != nullptr);     // angular-momentum rows: only when weighted (or dumped)
    flags |= phase_refs<NW, PIPE, std::is_same_v<R, double>>(L, P, inst, t, wid, k_out, &ph, ang);        // PIPE: the feet's orientation term may have been formed behind the look-ahead kinematics
    if (NW == 2 && wid == 0 && P.w_com_ang == 0.0 && !QF32) { WSTAMP(68); qp_prefill15(L, P); }      // ahead of the join: wave 1's chain is the longer one
    WSTAMP(7);
    bsync<NW>();
    WSTAMP(8);
    STAMP(7);
    flags |= phase_qp<NW, QF32, PIPE, WF, std::is_same_v<R, double>>(L, P, ph, wid, Fmask, iters_out, dbg, window);
    STAMP(8);
    if (plant && (NW == 1 || wid == 0)) phase_plant(L, P);         // the torques drive a plant instead of being thrown away (main.cpp:118-121)
    // need_tau (wave-uniform): the integrator never reads the torques -- the rollout asks for them at the k4 stage only (log, final record)
    if constexpr (NW == 1) { if (need_tau) phase_outputs_tau(L); phase_outputs_qdd(L, plant ? (int)PL_AP : (int)P_A); }
    else { if (wid == 0) phase_outputs_qdd(L, plant ? (int)PL_AP : (int)P_A); else if (need_tau) phase_outputs_tau(L); }
    WSTAMP(28);
    STAMP(9);
    if (dbg) {
        const int lane = LANE;
        for (int e = lane; e < 30; e += 64) { dbg[924 + e] = L[P_C + e]; dbg[1643 + e] = L[P_QREF + e]; dbg[3181 + e] = L[P_A + e]; }
        for (int e = lane; e < 6; e += 64) { dbg[954 + e] = std::is_same_v<R, double> ? L[P_C + e] - L[P_CG + e] : L[P_CG + e]; dbg[1464 + e] = L[P_AGPQP + e]; dbg[1673 + e] = L[P_HREF + e]; }
        for (int e = lane; e < 180; e += 64) { dbg[960 + e] = L[P_MTOP + e]; dbg[1284 + e] = L[P_AG + e]; }
        for (int e = lane; e < 144; e += 64) { dbg[1140 + e] = L[P_HL + e]; dbg[1482 + e] = L[P_JC + e]; dbg[1937 + e] = L[P_W + e]; }
        for (int e = lane; e < 12; e += 64) { dbg[1470 + e] = L[P_JPQP + e]; dbg[1679 + e] = L[P_FREF + e]; dbg[2081 + e] = L[P_H12 + e]; }
        for (int e = lane; e < 9; e += 64) dbg[1626 + e] = L[P_COM + e];
        for (int e = lane; e < 8; e += 64) dbg[1635 + e] = L[P_MPC + e];
        for (int e = lane; e < 210; e += 64) dbg[1691 + e] = L[P_YT + 30 * (e % 7) + e / 7];
        for (int e = lane; e < 36; e += 64) dbg[1901 + e] = L[P_SI + e];
        for (int e = lane; e < 1024; e += 64) dbg[2093 + e] = L[C_P + 33 * (e / 32) + e % 32];
        for (int e = lane; e < 32; e += 64) { dbg[3117 + e] = L[P_QV + e]; dbg[3149 + e] = L[P_CC + e]; }
    }
    // non-finite guard (reference aborts on NaN/Inf, controller.cpp:448-466)
    // checked on the QP solution (accelerations, contact wrench): tau = M a + C - J'w is finite iff they are, and the
    // torques may still be in flight on the helper wave
    double chk = 0.0;
    if (LANE < 30) chk = L[P_A + LANE]; else if (LANE >= 32 && LANE < 44) chk = L[P_W12 + LANE - 32];
    const bool nf = !(fabs(chk) <= 1.0e300);
    if (__ballot(nf) != 0ull) flags |= LMH_FLAG_NONFINITE;
    return flags;
}

__device__ __forceinline__ void load_common(double *L, LmhCParams &P, int inst)
{
    const double *mo = P.model + (size_t)P.model_stride * inst;
    for (int e = LANE; e < 393; e += 64) L[P_MODEL + e] = mo[e];
    for (int e = LANE; e < 228; e += 64) L[P_GCOL + e] = P.gcol[e];            // gcol | gi6 | gpinv are contiguous
    if (LANE < 48) {
        const int ft = LANE / 24, ax = (LANE % 24) / 8, k = LANE % 8;
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                if (ax == a && k == kk) v = ft ? P.lF[a][kk] : P.rF[a][kk];
            }
        L[P_POLY + LANE] = v;
    } else if (LANE < 54) {
        const int q = LANE - 48;
        int n = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) { if (q == a) n = P.rFn[a]; if (q == 3 + a) n = P.lFn[a]; }
        L[P_POLY + LANE] = (double)n;
    }
    if (LANE == 0) {                                               // clock-only reference cache (refs_prepare): empty
        L[P_RK] = -1073741824.0; L[P_RPH] = 0.0; L[P_RT0] = 0.0;
        L[P_RXS] = P.xscale ? P.xscale[inst] : 1.0;                // walking extension: per-instance step length
    }
    {
        const double *mp = P.mpc + (size_t)P.mpc_stride_inst * inst;
        const int N = P.horizon;
        if (N <= MPC_LDS_MAXN) {                                   // gain row + Px columns stay on chip for the launch
            for (int e = LANE; e < 3 * (N + 1); e += 64) L[P_MPCK + e] = mp[e];
        }
        double s0 = 0.0, s1 = 0.0;                                 // sum K Px0, sum K Px1 (refs_mpc)
        for (int i = LANE; i <= N; i += 64) { s0 += mp[i] * mp[(N + 1) + i]; s1 += mp[i] * mp[2 * (N + 1) + i]; }
        s0 = wave_sum(s0); s1 = wave_sum(s1);
        if (LANE == 0) { L[P_PRE + 2] = s0; L[P_PRE + 3] = s1; }
    }
    WSYNC();
}

__device__ __forceinline__ void store_out(const double *L, double *out, bool keep_qdd_slots = false)
{
    const int lane = LANE;
    if (lane < 24) out[lane] = L[P_TAU + lane];
    else if (lane < 36) out[lane] = L[P_W12 + lane - 24];
    if (!keep_qdd_slots) for (int e = lane; e < 30; e += 64) out[36 + e] = L[P_QDD + e];   // (the diagnostic rollout keeps its counters there)
    if (lane >= 32 && lane < 38) out[66 + lane - 32] = L[P_COM + lane - 32];        // CoM | comVel (Robot::getCoM / getComVel)
    if (lane >= 40 && lane < 46) out[72 + lane - 40] = L[P_MPC + 2 + lane - 40];    // Mpc3dLip::getXRef | getYRef
}

// ============================================================================ kernels
// Controller::standStep + WBC for every instance (src/controller.cpp:48-154).
// The plain kernel runs two waves per robot like the rollout; the debug kernel (intermediate dumps, stamps) keeps
// the single-wave schedule.
template <bool DEBUG, typename R, int NW = (DEBUG ? 1 : 2), bool QF32 = false>
__global__ void __launch_bounds__(64 * NW) lmh_eval_kernel(LmhDevParams P_arg, double *state, double *out, int32_t *status, double *debug)
{
    LmhCParams &P = LMH_KERNARG_PARAMS();
    __shared__ double L[LDS_DOUBLES];
    const int inst = blockIdx.x;
    if (inst >= P.n_instances) return;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double *st = state + (size_t)LMH_STATE_STRIDE * inst;
    const double t = st[90];
    unsigned F = 0xFFFFFFFFu;
    SET_GDBG(DEBUG ? debug + (size_t)LMH_DEBUG_STRIDE * inst : nullptr);
#ifdef LMH_SUBSTAMPS
    if (threadIdx.x == 0) g_L = L;                                 // bsync's wait counters (unused here, but the pointer must be valid)
    __syncthreads();
#endif
    LMH_POISON_LDS(L, LDS_DOUBLES);
    if (wid == 0) {
        load_common(L, P, inst);
        for (int e = LANE; e < 91; e += 64) L[P_Q + e] = st[e];    // q | v | v_prev | t
        F = (unsigned)status[LMH_STATUS_STRIDE * inst + 3];
        F = P.warm_start ? ~F : 0xFFFFFFFFu;                       // status keeps the ACTIVE mask
        if (LANE == 0) { L[P_KF] = (double)F; L[P_KF + 2] = 0.0; L[P_ORI] = 0.0; }             // no K_f^-1 stored yet, no orientation term formed ahead
        WSYNC();
    }
    bsync<NW>();
    if constexpr (NW == 2) { if (wid == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }   // see lmh_rollout_kernel
    int k = 0, iters = 0;
    const int flags = controller_eval<NW, R, QF32>(L, P, inst, t, wid, &F, &k, &iters, DEBUG ? debug + (size_t)LMH_DEBUG_STRIDE * inst : nullptr);
    bsync<NW>();                                                   // torques of the helper wave
    if (wid == 0) {
        store_out(L, out + (size_t)LMH_OUT_STRIDE * inst);
        if (LANE < 30) st[60 + LANE] = L[P_V + LANE];              // Robot::v_ <- dq (controller.cpp:59)
        if (LANE == 0) {
            int32_t *s = status + LMH_STATUS_STRIDE * inst;
            s[0] = k; s[1] = iters; s[2] = flags; s[3] = (int32_t)(~F);
        }
    }
}

// One claim on the rollout's work queue (called by one thread of the workgroup): -1 when no unit is left, else the robot, with the ticks it has
// behind it in *tick0.  Claims below n_inst are the robots' first chunks (no memory traffic); the others wait for their ring entry, which
// is pushed by a workgroup that is RUNNING a chunk of that robot -- no workgroup owns a unit it has not claimed, so the queue drains with any
// number of resident workgroups >= 1 (two launches sharing the chip, a debugger, a partitioned device).  An entry is TAKEN by the exchange
// itself (the value the exchange returns, not the value a load saw before it): two claimers one lap apart may poll the same slot, and only
// one of them may leave with the robot.  The protocol is modelled under random schedules, the take split into its load and its exchange, in
// tests/test_host_logic.py::test_rollout_work_queue_protocol_drains_under_any_schedule.
// Both waits of the protocol are bounded (LMH_SPIN_LIMIT polls).  A wait that runs out sets the launch's error word ticket[3]; the robot
// concerned stays part-way with a non-zero progress entry, and the last workgroup to leave turns every such entry into LMH_FLAG_UNFINISHED
// in the robot's status record and puts the ring / progress words back to zero (lmh_rollout_kernel); the host reports the error word
// (lmh_synchronize, or the next lmh_rollout on the slot: LMH_ERR_UNFINISHED).  The reference never fails silently either
// (src/controller.cpp:448-476).
#ifndef LMH_SPIN_LIMIT
#define LMH_SPIN_LIMIT (1 << 26)
#endif
__device__ __forceinline__ int rollout_claim(int *ticket, int n_inst, long long n_units, int *tick0)
{
    int *const ring = ticket + 4, *const prog = ticket + 4 + n_inst;
    *tick0 = 0;
    const long long n = (long long)atomicAdd(&ticket[0], 1);
    if (n >= n_units) return -1;
    if (n < (long long)n_inst) return (int)n;
    int *slot = ring + (int)((n - n_inst) % n_inst);
    int v = 0;
    for (int spin = 0; spin < LMH_SPIN_LIMIT; spin++) {            // bounded: a lost push must not hang the chip
        if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            v = atomicExch(slot, 0);                               // the take: whoever gets the non-zero value back owns the robot
            if (v != 0) break;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    if (v == 0) { atomicOr(&ticket[3], 1); return -1; }            // loud: see above
    __threadfence();
    *tick0 = __hip_atomic_load(&prog[v - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v - 1;
}

// One stage of rk4Step (rk4.hpp:5-18) for state component `lane` < 60 (q | v): xdot of apps/offline/main.cpp:107-121 from the evaluation that
// has just run, then the stage bookkeeping -- ksum collects k1 + 2 k2 + 2 k3 + k4, xs is the state of the next evaluation, x the state at the
// start of the tick (advanced by the fourth stage).  `xd4`: LDS index of sin / cos of pitch and yaw of THIS evaluation's configuration.
// The position half (lane < 30) reads nothing the evaluation produces, which is what lets the helper wave call this ahead of wave 0
// (its lanes >= 30 then hold don't-cares).
// XDQ: 0 = plain; 1 = (helper wave, ahead) also leave the position half of xdot in L[P_XDQ]; 2 = (wave 0, behind the helper) take it from there
// instead of forming it again -- three fp64 divisions and the cross product leave wave 0's path between two evaluations.
#define P_XDQ P_AG                 // 30 doubles: the angular rows of AG are dead once the QP fills have read the references
template <int XDQ = 0>
__device__ __forceinline__ void rk4_stage(double *L, int stage, int lane, double dt, int xd4, double &x, double &ksum, double &xs)
{
    // branch-free: every lane loads from a valid address and forms the few products of the base rows, the lane's class picks the result
    // (the lane-class branches and the four-way stage switch were ~100 scalar instructions per call)
    double xd;
    const int l60 = (lane < 60) ? lane : 0;
    if constexpr (XDQ == 2) xd = L[(l60 >= 30) ? P_QDD + l60 - 30 : P_XDQ + l60];
    else {
        const int l30 = (l60 < 30) ? l60 : 0;
        const double vq = L[(l60 >= 30) ? P_QDD + l60 - 30 : P_V + l60];       // joint rates | accelerations: xdot as it stands
        const double w0 = L[P_V + 3], w1 = L[P_V + 4], w2 = L[P_V + 5];
        const double p0 = L[P_Q], p1 = L[P_Q + 1], p2 = L[P_Q + 2];
        const double sp = L[xd4], cp = L[xd4 + 1], sy = L[xd4 + 2], cy = L[xd4 + 3];
        // v_classic = v_spatial + w x p (apps/offline/main.cpp:107-112)
        const double cr = (l30 == 0) ? (-w2 * p1 + w1 * p2) : (l30 == 1) ? (w2 * p0 - w0 * p2) : (-w1 * p0 + w0 * p1);
        // matrixAngularVelToEulerDot (generalizedFunctions.cpp:43-50): one reciprocal of cos(pitch) (rcp + two Newton steps, full fp64)
        // serves the three quotients -- each was a ~13-instruction IEEE division on the helper's path beside wave 0's 15 x 15 solve
        const double icp = fast_rcp(cp), tp = sp * icp;
        const double eu = (l30 == 3) ? (cy * icp) * w0 + (sy * icp) * w1 + 0.0 * w2
                        : (l30 == 4) ? (-sy) * w0 + cy * w1 + 0.0 * w2
                                     : (cy * tp) * w0 + (sy * tp) * w1 + 1.0 * w2;
        xd = (l60 < 3) ? vq + cr : (l60 < 6) ? eu : vq;
        if constexpr (XDQ == 1) L[(lane < 30) ? P_XDQ + lane : (int)P_DUMP] = xd;
    }
    // rk4.hpp:12-17 with the stage's coefficients as scalars: ksum <- a ksum + b xd (k1 + 2 k2 + 2 k3 + k4), x <- x + e ksum (e = dt / 6 at the
    // fourth stage, else 0), xs <- x + c xd (c = dt / 2, dt / 2, dt, 0).  a in {0, 1} and the zero terms are exact: the same roundings as the
    // four written-out cases
    const double a = (stage == 0) ? 0.0 : 1.0, b = (stage == 1 || stage == 2) ? 2.0 : 1.0;
    const double c = (stage == 3) ? 0.0 : (stage == 2) ? dt : 0.5 * dt, e = (stage == 3) ? dt / 6.0 : 0.0;
    ksum = fma(b, xd, a * ksum);
    x = fma(e, ksum, x);
    xs = fma(c, xd, x);
}

// Closed loop of apps/offline/main.cpp:66-122: n_ticks x rk4Step(dynamics) with Clock::step.
// Workgroup = LMH_ROLLOUT_THREADS = 2 waves per robot (see bsync): 4 robots = 8 waves per CU, two per SIMD, so the
// kernel is held to 256 registers.  Wave 0 owns the RK4 state (lane i < 60 <-> component i) and everything
// sequential; wave 1 joins for the phases controller_eval<2> splits.
#ifndef LMH_CHUNK_TICKS
#define LMH_CHUNK_TICKS 250          // ticks of one robot a workgroup runs before the robot goes back to the queue (see lmh_rollout_kernel)
#endif
template <typename R, bool QF32 = false>
#ifndef LMH_ROLLOUT_ATTR
#ifndef LMH_WAVES_PER_EU
#define LMH_WAVES_PER_EU 2
#endif
#ifdef LMH_NUM_VGPR
#define LMH_ROLLOUT_ATTR __attribute__((amdgpu_num_vgpr(LMH_NUM_VGPR)))
#else
#define LMH_ROLLOUT_ATTR __attribute__((amdgpu_waves_per_eu(LMH_WAVES_PER_EU, LMH_WAVES_PER_EU)))
#endif
#endif
__global__ void __launch_bounds__(LMH_ROLLOUT_THREADS) LMH_ROLLOUT_ATTR
lmh_rollout_kernel(const LmhDevParams *__restrict__ Pg, int *__restrict__ ticket_a, double *state_a, double *out_a, int32_t *status_a, double *log_a, int n_ticks_a)
{
    // The parameter block is read through a pointer that is made opaque once per evaluation (params_of): hoisting its ~70 scalars out
    // of the tick loop pins them in SGPRs for the whole launch (round 1: 189 SGPR + 16 VGPR spills, 60 B of scratch per lane that reached
    // HBM); re-reading them costs a few scalar-cache loads per evaluation.
    __shared__ double L[LDS_DOUBLES];
    // The kernel's own arguments are read from the kernarg segment where they are used (a scalar load each: chunk start, chunk end, the log
    // once per tick) instead of being carried in scalar registers across the tick loop, where they were spilled to vector lanes and read back.
    (void)ticket_a; (void)state_a; (void)out_a; (void)status_a; (void)log_a; (void)n_ticks_a;
    struct Args { const LmhDevParams *Pg; int *ticket; double *state, *out; int32_t *status; double *log; int n_ticks; };
    typedef const __attribute__((address_space(4))) Args CArgs;
    auto KA = [&]() -> CArgs & { CArgs *p_ = (CArgs *)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p_)); return *p_; };
#define ticket (KA().ticket)
#define state (KA().state)
#define out (KA().out)
#define status (KA().status)
#define log (KA().log)
#define n_ticks (KA().n_ticks)
    LmhCParams *Pc = (LmhCParams *)(uintptr_t)Pg;
    LmhCParams &P = *Pc;
    // One workgroup runs several robots one after the other (grid = the number of workgroups the chip holds at once, lmh_launch_rollout):
    // when the hardware dispatcher refills the chip from a longer grid, throughput drops by ~15 % (measured: 1024 robots 4.1 ms per launch,
    // 2048 robots 11.4 ms, 4096 robots 19.2 ms for the same 40 ticks); looping inside the resident workgroups keeps the first round's placement.
    // The unit of work is (robot, chunk of LMH_CHUNK_TICKS ticks): a robot's state goes back to its HBM record at the end of a chunk and the
    // robot re-enters a ring queue, so the launch does not end with most workgroups idle behind the few that drew a slow robot last
    // (whole-robot units left the slots busy 97.6 % of a launch).  Every unit, a workgroup's first included, is claimed with one atomicAdd
    // on `head` (rollout_claim).  Which workgroup runs a chunk has no influence on its result: everything per-robot is re-read from the
    // record, the caches in LDS are rebuilt from it.  ticket: [0] head | [1] workgroups that left | [2] tail | [3] - | ring [n_inst]
    // (robot + 1, 0 = empty) | ticks done [n_inst].  The last workgroup to leave zeroes the counters for the next launch on this slot (ring
    // and progress entries zero themselves).
    __shared__ int s_next, s_tick0;
#ifdef LMH_SUBSTAMPS
    if (threadIdx.x == 0) g_L = L;                                 // bsync's wait counters live in the robot's LDS image
    __syncthreads();
#endif
    const int n_inst = P.n_instances;
    int *const ring = ticket + 4, *const prog = ticket + 4 + n_inst;
    const int n_chunks = (n_ticks + LMH_CHUNK_TICKS - 1) / LMH_CHUNK_TICKS;
    const long long n_units = (long long)n_inst * n_chunks;
    if (threadIdx.x == 0) { int t0_ = 0; s_next = rollout_claim(ticket, n_inst, n_units, &t0_); s_tick0 = t0_; }
    __syncthreads();
    int inst = __builtin_amdgcn_readfirstlane(s_next), tick0 = __builtin_amdgcn_readfirstlane(s_tick0);
    while (inst >= 0 && inst < n_inst) {                           // workgroup-uniform
    const int n_here = (n_ticks - tick0 < LMH_CHUNK_TICKS) ? n_ticks - tick0 : LMH_CHUNK_TICKS;
    const int lane = LANE;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double *st = state + (size_t)LMH_STATE_STRIDE * inst;
    SET_GDBG(nullptr);
    // lane i < 60 of wave 0 owns state component i (q | v); v_prev lives in LDS between evaluations.  Wave 1 integrates its own copy of the
    // position half (lane i < 30): the position part of xdot depends on the stage's state only, not on the evaluation's result, so the
    // helper wave knows the NEXT evaluation's configuration while wave 0 is still solving this one's QP -- and runs that evaluation's
    // forward kinematics there (window below).
    // Not with the fp32 QP (tolerance sweep): its set-up runs on wave 0 alone and uses all of the scratch; that build keeps the plain schedule.
    constexpr bool PIPE = !QF32;
    double x = 0.0, t = st[90];
    unsigned F = 0xFFFFFFFFu;
    LMH_POISON_LDS(L, LDS_DOUBLES);
    if (wid == 0) {
        load_common(L, P, inst);
        x = (lane < 60) ? st[lane] : 0.0;
        if (lane < 30) L[P_VP + lane] = st[60 + lane];
        F = (unsigned)status[LMH_STATUS_STRIDE * inst + 3];
        F = P.warm_start ? ~F : 0xFFFFFFFFu;
        if (lane == 0) { L[P_KF] = (double)F; L[P_KF + 2] = 0.0; L[P_ORI] = 0.0; }             // no K_f^-1 stored yet, no orientation term formed ahead
        if constexpr (PIPE) {
            if (lane < 60) L[P_Q + lane] = x;
            WSYNC();
            phase_fk<R>(L, P.gcol + 228);                          // kinematics of the first evaluation (every later one is run ahead by wave 1)
        }
    } else x = (lane < 30) ? st[lane] : 0.0;
    bsync<2>();
    if (PIPE && wid == 1) refs_prepare(L, P, inst, t);             // clock-only references of the first evaluation (needs load_common's cache reset)
    int k = 0, iters = 0, flags = 0, itmax = 0;
    if (tick0 > 0) { itmax = status[LMH_STATUS_STRIDE * inst + 1]; flags = status[LMH_STATUS_STRIDE * inst + 2]; }      // status[1], [2] cover the whole launch
    IbPack ibp = {{0u, 0u}};
    if constexpr (std::is_same_v<R, double>) ibp = ib_pack();      // the matrix-core CRBA's gather table, two registers for the whole chunk
    const double dt = P.dt;
#ifdef LMH_SUBSTAMPS
    if (lane == 0) { L[D_BWAIT + wid] = 0.0; for (int j_ = 0; j_ < 8; j_++) L[D_JWAIT + 8 * wid + j_] = 0.0; if (wid == 0) for (int j_ = 0; j_ < 6; j_++) g_rt[j_] = 0u; }
    __syncthreads();
    const long long t_launch = clock64(), t_real = wall_clock64();
#endif
    // the leading wave carries the critical path: it wins issue arbitration against the helper wave of the robot it
    // shares the SIMD with (+2.7 % measured; the reverse assignment gains nothing)
    if (wid == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    for (int tick = 0; tick < n_here; tick++) {
        double ksum = 0.0, xs = x;
        for (int stage = 0; stage < 4; stage++) {
            const double ts = (stage == 0) ? t : (stage == 3) ? t + dt : t + 0.5 * dt;      // rk4.hpp:12-15
            const double tn = (stage < 2) ? t + 0.5 * dt : t + dt;                          // time of the evaluation after this one (the next tick starts at t + dt)
            // sin / cos of pitch, yaw of this evaluation | of the next: the other buffer
            const int xd4 = (PIPE && (stage & 1)) ? (int)P_SCB : (int)P_SC + 52;
            const int xd4n = (stage & 1) ? (int)P_SC + 52 : (int)P_SCB;
            if (wid == 0) {
                WSYNC();
                if (lane < 60) L[P_Q + lane] = xs;
                WSYNC();
            }
            LmhCParams *Pe = Pc;
            asm volatile("" : "+s"(Pe));                           // opaque: the loads below belong to this evaluation
#ifdef LMH_DIAG_TL
            if (lane == 0) g_tl[wid] = (log && tick0 + tick == n_ticks - 1 && stage == LMH_DIAG_TL) ? log + (size_t)256 * inst : nullptr;
            WSYNC();
#endif
#ifdef LMH_SUBSTAMPS
#ifdef LMH_DIAG_STAGE                                              // the per-join split of ONE Runge-Kutta stage's evaluations (the others land in slot 7)
            if (lane == 0) L[D_JIDX + wid] = (stage == LMH_DIAG_STAGE) ? 0.0 : 100.0;
#else
            if (lane == 0) L[D_JIDX + wid] = 0.0;
#endif
#endif
            // wave 1, once its share of the QP set-up is done: the next stage's configuration (rk4_stage, position half) and the clock-only
            // references of its time (unless it is the same instant: stages 2 | 3, and 4 | 1 of the next tick); then, while wave 0 runs the
            // cone solve and the recovery, its forward kinematics.  The world transforms land in S0 + [0, 378), which nothing touches until the next evaluation's phase_com_x.
            auto window = [&](int part) {
                if (part == 0) { WSTAMP(60); rk4_stage<1>(L, stage, lane, dt, xd4, x, ksum, xs); WSTAMP(69); }    // in the helper's slack inside the QP set-up
                else {                                             // behind the join that frees the set-up scratch
                    WSTAMP(61);
                    if (tn != ts) refs_prepare(L, *Pe, inst, tn);
                    WSTAMP(62);
                    phase_fk<R, true>(L, Pe->gcol + 228, (R)xs, xd4n);
                    {   // the feet's orientation term of that evaluation (acos + sin, ~250 instructions; P_FREF is dead since the QP fills): here when
                        // the coming evaluation is in single support or flight -- wave 0's share behind this join (general-route cone solve) is
                        // then much the longer one (the helper waited ~9.6k cycles here, profiles/r04_barrier_share_mid.txt); in double support the
                        // two shares are level and wave 0 keeps the term (refs_pd_feet)
                        const bool ori = __builtin_amdgcn_readfirstlane((int)L[P_RPH]) != LMH_PHASE_DOUBLE;
                        if (ori) refs_feet_orientation(L, *Pe, L + A_T + 84, 84);
                        if (lane == 0) L[P_ORI] = ori ? 1.0 : 0.0;
                    }
                    WSTAMP(70);
                }
            };
            flags |= controller_eval<2, R, QF32, PIPE, decltype(window)>(L, *Pe, inst, ts, wid, &F, &k, &iters, nullptr, stage == 3, window, &ibp);
            WSTAMP(71);
            if (wid == 0) {
                itmax = (iters > itmax) ? iters : itmax;
                const double xprev = xs;
                if constexpr (PIPE) rk4_stage<2>(L, stage, lane, dt, xd4, x, ksum, xs); else rk4_stage<0>(L, stage, lane, dt, xd4, x, ksum, xs);
                // Robot::v_ <- dq for the next evaluation
                WSYNC();
                if (lane >= 30 && lane < 60) L[P_VP + lane - 30] = xprev;
            }
            WSTAMP(72);
#ifdef LMH_DIAG_TL
            WSYNC();
            if (lane == 0) g_tl[wid] = nullptr;
            WSYNC();
#endif
        }
#ifdef LMH_DIAG_TL
        if (false) {
#else
        if (log) {                                                  // each wave logs what it produced: wave 1 the torques, wave 0 the wrench
#endif
            double *lg = log + ((size_t)(tick0 + tick) * P.n_instances + inst) * 36;
            if (wid != 0) { if (lane < 24) lg[lane] = L[P_TAU + lane]; }
            else if (lane >= 24 && lane < 36) lg[lane] = L[P_W12 + lane - 24];
        }
        t += dt;                                                    // Clock::step, Clock.hpp:11
    }
    bsync<2>();                                                    // the last torques (helper wave) are in LDS
#ifdef LMH_SUBSTAMPS
    if (wid == 1 && lane == 0) st[94] = (double)__builtin_amdgcn_s_getreg(63492);      // HW_ID of wave 1
#endif
    if (wid == 0) {
        WSYNC();
#ifdef LMH_SUBSTAMPS
        store_out(L, out + (size_t)LMH_OUT_STRIDE * inst, true);
#else
        store_out(L, out + (size_t)LMH_OUT_STRIDE * inst);
#endif
#ifdef LMH_SUBSTAMPS
        if (lane == 0) {                                           // diagnostic build: this robot's cycles in the launch and inside barriers (pad slots), summed over its chunks
            double *o_ = out + (size_t)LMH_OUT_STRIDE * inst;
            const bool first_ = tick0 == 0;
            const long long now_ = wall_clock64();
            o_[78] = (first_ ? 0.0 : o_[78]) + (double)(clock64() - t_launch);
            o_[60] = first_ ? (double)t_real : o_[60]; o_[61] = (double)now_;      // 100 MHz, chip-wide: when the robot first came and last left
            o_[62] = (first_ ? 0.0 : o_[62]) + (double)(now_ - t_real);            // ... and how long workgroups were busy with it
            st[91] = (first_ ? 0.0 : st[91]) + L[D_BWAIT]; st[92] = (first_ ? 0.0 : st[92]) + L[D_BWAIT + 1];
            for (int j_ = 0; j_ < 8; j_++) { o_[36 + j_] = (first_ ? 0.0 : o_[36 + j_]) + L[D_JWAIT + j_]; o_[48 + j_] = (first_ ? 0.0 : o_[48 + j_]) + L[D_JWAIT + 8 + j_]; }
            for (int j_ = 0; j_ < 6; j_++) { double *r_ = o_ + ((j_ < 4) ? 44 + j_ : 52 + j_); *r_ = (first_ ? 0.0 : *r_) + (double)g_rt[j_]; }      // slots 44..47, 56, 57
            st[93] = (double)__builtin_amdgcn_s_getreg(63492);      // HW_ID of wave 0 (SIMD, CU, wave slot) | XCC_ID: where the robot ran
            st[95] = (double)__builtin_amdgcn_s_getreg(63508);
        }
#endif
        if (lane < 60) st[lane] = x;
        if (lane < 30) st[60 + lane] = L[P_VP + lane];
        if (lane == 0) {
            st[90] = t;
            int32_t *s = status + LMH_STATUS_STRIDE * inst;
            s[0] = k; s[1] = itmax; s[2] = flags; s[3] = (int32_t)(~F);
        }
    }
    __syncthreads();                                               // every store of the record has left both waves; every lane is done with the LDS image
    if (threadIdx.x == 0) {
        int next0 = 0;
        __threadfence();                                           // the record is visible before the robot is
        if (tick0 + n_here < n_ticks) {
            prog[inst] = tick0 + n_here;
            __threadfence();
            const int tpos = atomicAdd(&ticket[2], 1);             // the position is reserved; the entry goes in once the slot is empty: the taker of
            int *pslot = ring + tpos % n_inst;                     // the entry one lap earlier (a running workgroup, spinning on it) may not have been there yet
            bool in = false;
#ifdef LMH_TEST_LOSE_PUSH                                          // fault injection (checker build `qfault` only): the pusher of some robots never gets to its compare-and-swap
            if (inst % LMH_TEST_LOSE_PUSH != 3)
#endif
            for (int spin = 0; spin < LMH_SPIN_LIMIT && !in; spin++) {
                in = atomicCAS(pslot, 0, inst + 1) == 0;
                if (!in) __builtin_amdgcn_s_sleep(16);
            }
            if (!in) atomicOr(&ticket[3], 2);                      // the robot stays out of the queue: prog[inst] != 0 marks it (flagged by the last workgroup)
        } else prog[inst] = 0;
        const int next = rollout_claim(ticket, n_inst, n_units, &next0);
        s_next = next; s_tick0 = next0;
    }
    __syncthreads();
    inst = __builtin_amdgcn_readfirstlane(s_next);
    tick0 = __builtin_amdgcn_readfirstlane(s_tick0);
    __threadfence();                                               // acquire: this CU's vector cache may hold the record as it was chunks ago
    }                                                              // next work unit of this workgroup (the next write of s_next is many barriers away)
    __syncthreads();                                               // every wave has read the last claim's s_next
    if (threadIdx.x == 0) {                                        // every claim and push of this workgroup precedes this increment: the last one to leave resets the slot
        __threadfence();
        const bool last = atomicAdd(&ticket[1], 1) == (int)gridDim.x - 1;
        s_next = !last ? 0 : (__hip_atomic_load(&ticket[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 2 : 1;
    }
    __syncthreads();
    const int leave = s_next;                                      // 0: others are still at work | 1: last, clean launch | 2: last, a wait ran out
    if (leave == 2) {
        // nobody runs a robot any more: a non-zero progress entry is a robot that never got its remaining chunks.  It is flagged, and the
        // ring / progress words go back to zero so that the next launch on this slot starts from the state it expects; ticket[3] is left
        // for the host (lmh_capi.hip reads and clears it).
        __threadfence();
        for (int i = (int)threadIdx.x; i < n_inst; i += (int)blockDim.x) {
            if (__hip_atomic_load(&prog[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                atomicOr(&status[LMH_STATUS_STRIDE * i + 2], LMH_FLAG_UNFINISHED);
                prog[i] = 0;
            }
            ring[i] = 0;
        }
        __threadfence();
        __syncthreads();
    }
    if (leave != 0 && threadIdx.x == 0) { ticket[0] = 0; ticket[1] = 0; ticket[2] = 0; __threadfence(); }
}
#undef ticket
#undef state
#undef out
#undef status
#undef log
#undef n_ticks

// Robot::Robot model preparation (Robot.cpp:14-22) + Dynamics::spatialInertiaMatrix pieces
// (Dynamics.cpp:4-13): raw [28][13] -> device model record.
__global__ void __launch_bounds__(64) lmh_model_kernel(const double *raw, double *model, int n_models, const double *lcoef)
{
    __shared__ double L[LDS_DOUBLES];
    const int mi = blockIdx.x;
    if (mi >= n_models) return;
    const int lane = LANE;
    const double *rw = raw + (size_t)mi * 28 * LMH_LINK_STRIDE;
    SET_GDBG(nullptr);
    for (int e = lane; e < 30; e += 64) L[P_Q + e] = 0.0;          // FK at q = 0
    WSYNC();
    phase_fk<double>(L, lcoef);
    double mloc = 0.0;
    if (lane < 28) {
        const double *T = L + A_T + 12 * lane, *lk = rw + LMH_LINK_STRIDE * lane;
        const double m = lk[0];
        double c[3], I1[9], I2[9];
        for (int a = 0; a < 3; a++) c[a] = T[a] * lk[1] + T[4 + a] * lk[2] + T[8 + a] * lk[3];           // Rj' com
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) I1[3 * a + b] = T[a] * lk[4 + b] + T[4 + a] * lk[7 + b] + T[8 + a] * lk[10 + b];   // Rj' I
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) I2[3 * a + b] = I1[3 * a] * T[b] + I1[3 * a + 1] * T[4 + b] + I1[3 * a + 2] * T[8 + b]; // (Rj' I) Rj
        // cc = [c]x [c]x ; Ibar = I - m cc
        const double cm[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
        double *mo = model + (size_t)mi * LMH_MODEL_STRIDE + LMH_BODY_STRIDE * lane;
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                const double cc = cm[3 * a] * cm[b] + cm[3 * a + 1] * cm[3 + b] + cm[3 * a + 2] * cm[6 + b];
                mo[3 * a + b] = I2[3 * a + b] - (m * cc);
            }
        for (int a = 0; a < 3; a++) mo[9 + a] = m * c[a];
        mo[12] = m; mo[13] = qdes_of(lane);                       // the record's spare slot: Robot::desiredPosture of coordinate `lane` (PDJointsAcc)
        mloc = m;
    }
    // mass_ += links_[i].mass in frame order (Robot.cpp:21)
    if (lane == 0) {
        double s = 0.0;
        for (int i = 0; i < 28; i++) s += rw[LMH_LINK_STRIDE * i];
        model[(size_t)mi * LMH_MODEL_STRIDE + 392] = s;
    }
    (void)mloc;
}


// ============================================================================ inverse kinematics
// Kinematics::compute (src/invKinematics.cpp:27-52): Newton iteration on the operational state
// [feet pose(12) | arm+head joints(12) | base rpy(3) | CoM(3)], Jacobian as jacInvKinematics
// (:151-204) builds it, linear solve by Gaussian elimination with partial pivoting (the
// reference uses colPivHouseholderQr on the same square system).  One wave per instance.
enum { IK_J = LDS_DOUBLES, IK_E = LDS_DOUBLES + 930, IK_Q = LDS_DOUBLES + 960, IK_OM = LDS_DOUBLES + 1008, IK_LDS = LDS_DOUBLES + 1040 };

__device__ void rot_to_euler_dev(const double *T /*3x4*/, double *eta)   // invKinematics.cpp:256-267, newR = R * Rf_q0
{
    double nR[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) nR[3 * a + b] = T[4 * a] * c_rdes[b] + T[4 * a + 1] * c_rdes[3 + b] + T[4 * a + 2] * c_rdes[6 + b];
    eta[2] = atan2(nR[3], nR[0]);
    eta[1] = atan2(-nR[6], cos(eta[2]) * nR[0] + sin(eta[2]) * nR[3]);
    eta[0] = atan2(sin(eta[2]) * nR[2] - cos(eta[2]) * nR[5], -sin(eta[2]) * nR[1] + cos(eta[2]) * nR[4]);
}
__device__ void omega_mat(const double *eta, double *Om)                 // generalizedFunctions.cpp:43-50
{
    Om[0] = cos(eta[2]) / cos(eta[1]); Om[1] = sin(eta[2]) / cos(eta[1]); Om[2] = 0;
    Om[3] = -sin(eta[2]); Om[4] = cos(eta[2]); Om[5] = 0;
    Om[6] = cos(eta[2]) * tan(eta[1]); Om[7] = sin(eta[2]) * tan(eta[1]); Om[8] = 1;
}
__device__ void inv3_dev(const double *A, double *Ai)
{
    const double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    Ai[0] = (A[4] * A[8] - A[5] * A[7]) / det; Ai[1] = (A[2] * A[7] - A[1] * A[8]) / det; Ai[2] = (A[1] * A[5] - A[2] * A[4]) / det;
    Ai[3] = (A[5] * A[6] - A[3] * A[8]) / det; Ai[4] = (A[0] * A[8] - A[2] * A[6]) / det; Ai[5] = (A[2] * A[3] - A[0] * A[5]) / det;
    Ai[6] = (A[3] * A[7] - A[4] * A[6]) / det; Ai[7] = (A[1] * A[6] - A[0] * A[7]) / det; Ai[8] = (A[0] * A[4] - A[1] * A[3]) / det;
}

__global__ void __launch_bounds__(64) lmh_ik_kernel(LmhDevParams P_arg, double *qio, LmhIkTarget tgt, int32_t *iters_out)
{
    LmhCParams &P = LMH_KERNARG_PARAMS();
    __shared__ double L[IK_LDS];
    const int inst = blockIdx.x;
    if (inst >= P.n_instances) return;
    const int lane = LANE;
    SET_GDBG(nullptr);
    load_common(L, P, inst);
    if (lane < 30) L[P_Q + lane] = qio[30 * (size_t)inst + lane];
    WSYNC();
    // desiredOperationalState (:11-25): feet pose, CURRENT arm/head joints, zero base attitude, CoM target
    double des = 0.0;
    if (lane < 16) L[IK_E + lane] = tgt.v[lane];                  // kernel-argument record -> LDS (a per-lane index into it would be a scratch copy)
    WSYNC();
    if (lane < 12) des = L[IK_E + lane];
    else if (lane < 24) des = L[P_Q + 18 + (lane - 12)];
    else if (lane < 27) des = 0.0;
    else if (lane < 30) des = L[IK_E + 12 + (lane - 27)];
    WSYNC();
    const double mass = L[P_MODEL + 392];
    int iter = 0;
    for (;;) {
        phase_fk<double>(L, P.gcol + 228);
        phase_com_x<1, double>(L, 0);
        phase_jacobian<double>(L);
        // operationalState (:54-70)
        double Qv = 0.0;
        if (lane < 12) {
            const int ft = lane / 6, k = lane % 6;
            const double *T = L + P_TB + 12 * (1 + ft);
            if (k < 3) Qv = T[4 * k + 3];
            else { double eta[3]; rot_to_euler_dev(T, eta); Qv = eta[k - 3]; }
        } else if (lane < 24) Qv = L[P_Q + 18 + (lane - 12)];
        else if (lane < 27) Qv = L[P_Q + 3 + (lane - 24)];
        else if (lane < 30) Qv = L[P_COM + (lane - 27)];
        const double e = (lane < 30) ? des - Qv : 0.0;
        const double crit = wave_max(fabs(e));
        if (!(crit > 1e-10) || iter >= 200) break;
        if (lane < 30) L[IK_E + lane] = e;
        // ---- jacInvKinematics (:151-204)
        if (lane == 0) {
            double Om[9], Oi[9];
            omega_mat(L + P_Q + 3, Om); inv3_dev(Om, Oi);
            for (int k = 0; k < 9; k++) L[IK_OM + k] = Oi[k];
            for (int ft = 0; ft < 2; ft++) {
                double eta[3], Of[9];
                rot_to_euler_dev(L + P_TB + 12 * (1 + ft), eta); omega_mat(eta, Of);
                for (int k = 0; k < 9; k++) L[IK_OM + 9 + 9 * ft + k] = Of[k];
            }
        }
        WSYNC();
        for (int el = lane; el < 900; el += 64) {
            const int r = el / 30, c = el % 30;
            double val = 0.0;
            if (r < 12) {
                // rows: per foot [lin(3); ang(3)] <- source rows [ang; lin]; base columns [lin | ang] <- source [ang | lin]
                const int ft = r / 6, rk = r % 6, srow = 6 * ft + ((rk < 3) ? rk + 3 : rk - 3);
                if (c >= 6) val = jdense(L, srow, c);
                else if (c < 3) val = jdense(L, srow, c + 3);
                else {
                    // block(.,3,3,3) * Omega^-1 on source columns 0..2 (angular velocity of the base)
                    double b0 = 0.0;
                    for (int k = 0; k < 3; k++) b0 += jdense(L, srow, k) * L[IK_OM + 3 * k + (c - 3)];
                    val = b0;
                    if (rk >= 3) {                                 // OmegaFoot * block (angular rows only, :191-197)
                        double acc = 0.0;
                        for (int m2 = 0; m2 < 3; m2++) {
                            const int srow2 = 6 * ft + m2;         // source angular rows
                            double bm = 0.0;
                            for (int k = 0; k < 3; k++) bm += jdense(L, srow2, k) * L[IK_OM + 3 * k + (c - 3)];
                            acc += L[IK_OM + 9 + 9 * ft + 3 * (rk - 3) + m2] * bm;
                        }
                        val = acc;
                    }
                }
            } else if (r < 24) val = (c == 18 + (r - 12)) ? 1.0 : 0.0;
            else if (r < 27) val = (c == 3 + (r - 24)) ? 1.0 : 0.0;
            L[IK_J + 31 * r + c] = val;
        }
        WSYNC();
        // comJacobian (:206-244), rows 27..29: one lane per column
        if (lane < 30) {
            double j0 = 0, j1 = 0, j2 = 0;
            for (int i = 0; i < 27; i++) {
                const double *mo = L + P_MODEL + LMH_BODY_STRIDE * i;
                const double m = mo[12];
                if (m == 0.0) continue;
                const double *T = L + A_T + 12 * i;
                const double c0 = mo[9] / m, c1 = mo[10] / m, c2 = mo[11] / m;
                const double pc0 = T[0] * c0 + T[1] * c1 + T[2] * c2 + T[3];
                const double pc1 = T[4] * c0 + T[5] * c1 + T[6] * c2 + T[7];
                const double pc2 = T[8] * c0 + T[9] * c1 + T[10] * c2 + T[11];
                double x0 = 0, x1 = 0, x2 = 0;
                if (lane < 3) { x0 = (lane == 0); x1 = (lane == 1); x2 = (lane == 2); }
                else if (lane < 6) {                                // crossMatrix(pBase - pCom) column (lane-3)
                    const double d0 = L[A_T + 3] - pc0, d1 = L[A_T + 7] - pc1, d2 = L[A_T + 11] - pc2;
                    if (lane == 3) { x0 = 0; x1 = d2; x2 = -d1; }
                    else if (lane == 4) { x0 = -d2; x1 = 0; x2 = d0; }
                    else { x0 = d1; x1 = -d0; x2 = 0; }
                } else {
                    // joint (lane-6) contributes if its frame is on the path from frame i to the base
                    const int jf = f_jframe(lane - 6);
                    int j = i; bool on = false;
                    while (j != 0) { if (j == jf) { on = true; break; } j = f_parent(j); }
                    if (on) {
                        const double *Tj = L + A_T + 12 * jf;
                        const double z0 = Tj[2], z1 = Tj[6], z2 = Tj[10];
                        const double d0 = pc0 - Tj[3], d1 = pc1 - Tj[7], d2 = pc2 - Tj[11];
                        x0 = z1 * d2 - z2 * d1; x1 = z2 * d0 - z0 * d2; x2 = z0 * d1 - z1 * d0;
                    }
                }
                j0 = j0 + m * x0; j1 = j1 + m * x1; j2 = j2 + m * x2;
            }
            L[IK_J + 31 * 27 + lane] = j0 / mass; L[IK_J + 31 * 28 + lane] = j1 / mass; L[IK_J + 31 * 29 + lane] = j2 / mass;
        }
        WSYNC();
        if (lane < 9) {                                             // J.block(0,3,3,3) *= Omega^-1 on the CoM rows
            const int r = lane / 3, c = lane % 3;
            double s = 0.0;
            for (int k = 0; k < 3; k++) s += L[IK_J + 31 * (27 + r) + 3 + k] * L[IK_OM + 3 * k + c];
            L[IK_Q + lane] = s;
        }
        WSYNC();
        if (lane < 9) L[IK_J + 31 * (27 + lane / 3) + 3 + lane % 3] = L[IK_Q + lane];
        if (lane < 30) L[IK_J + 31 * lane + 30] = L[IK_E + lane];   // augmented rhs
        WSYNC();
        // ---- Gaussian elimination with partial pivoting on [J | e] (30 x 31)
        for (int c = 0; c < 30; c++) {
            double best = (lane >= c && lane < 30) ? fabs(L[IK_J + 31 * lane + c]) : -1.0;
            int bi = lane;
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (bi != c && lane < 31) { const double t = L[IK_J + 31 * c + lane]; L[IK_J + 31 * c + lane] = L[IK_J + 31 * bi + lane]; L[IK_J + 31 * bi + lane] = t; }
            WSYNC();
            const double piv = L[IK_J + 31 * c + c];
            const int nr = 29 - c, nc = 30 - c;                     // rows below, columns right (incl. rhs)
            for (int el = lane; el < nr * nc; el += 64) {
                const int r = c + 1 + el / nc, cc = c + 1 + el % nc;
                L[IK_J + 31 * r + cc] -= (L[IK_J + 31 * r + c] / piv) * L[IK_J + 31 * c + cc];
            }
            WSYNC();
        }
        for (int r = 29; r >= 0; r--) {                             // back substitution
            const double xr = L[IK_J + 31 * r + 30] / L[IK_J + 31 * r + r];
            WSYNC();
            if (lane < r) L[IK_J + 31 * lane + 30] -= L[IK_J + 31 * lane + r] * xr;
            if (lane == 0) L[IK_Q + 16 + r] = xr;
            WSYNC();
        }
        if (lane < 30) L[P_Q + lane] += L[IK_Q + 16 + lane];        // q += dq
        WSYNC();
        iter++;
    }
    if (lane < 30) qio[30 * (size_t)inst + lane] = L[P_Q + lane];
    if (lane == 0 && iters_out) iters_out[inst] = iter;
}

#ifndef LMH_ROLLOUT_ONLY              // (scripts/isa_census.py compiles the fp64 rollout kernel alone)
extern "C" void lmh_launch_ik(const LmhDevParams *P, double *q, const LmhIkTarget *target, int32_t *iters, hipStream_t s)
{
    hipLaunchKernelGGL(lmh_ik_kernel, dim3(P->n_instances), dim3(64), 0, s, *P, q, *target, iters);
}

#endif
// End-of-run summary: 16 doubles per instance, the record the one RCCL gather moves.  One lane per robot; HBM-bound
// (reads 66 + 36 doubles + 4 ints of each record once), 1.1 KB per robot.
__global__ void __launch_bounds__(256) lmh_summary_kernel(int n, const double *state, const double *out, const int32_t *status, double *summary)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *st = state + (size_t)LMH_STATE_STRIDE * i, *o = out + (size_t)LMH_OUT_STRIDE * i;
    const int32_t *s = status + LMH_STATUS_STRIDE * i;
    double *r = summary + (size_t)LMH_SUMMARY_WIDTH * i;
    double chk = 0.0, tmax = 0.0;
    for (int k = 0; k < 60; k++) chk += st[k];
    for (int k = 0; k < 24; k++) tmax = fmax(tmax, fabs(o[k]));
    for (int k = 0; k < 6; k++) r[k] = st[k];
    r[6] = st[90]; r[7] = tmax;
    r[8] = o[24 + 5] + o[24 + 11]; r[9] = o[24 + 5]; r[10] = o[24 + 11];
    r[11] = (double)s[0]; r[12] = (double)s[1]; r[13] = (double)s[2]; r[14] = (double)__popc((unsigned)s[3]);
    r[15] = chk;
}
extern "C" void lmh_launch_summary(int n, const double *state, const double *out, const int32_t *status, double *summary, hipStream_t s)
{
    hipLaunchKernelGGL(lmh_summary_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, state, out, status, summary);
}

// Robot::updateState + getCoM (Robot.cpp:264-269,225-238) for q only.
__global__ void __launch_bounds__(64) lmh_com_kernel(LmhDevParams P_arg, const double *q, double *com)
{
    LmhCParams &P = LMH_KERNARG_PARAMS();
    __shared__ double L[LDS_DOUBLES];
    const int inst = blockIdx.x;
    if (inst >= P.n_instances) return;
    SET_GDBG(nullptr);
    load_common(L, P, inst);
    if (LANE < 30) L[P_Q + LANE] = q[30 * (size_t)inst + LANE];
    if (LANE < 60) L[P_V + LANE] = 0.0;
    WSYNC();
    phase_fk<double>(L, P.gcol + 228);
    phase_com_x<1, double>(L, 0);
    if (LANE < 3) com[3 * (size_t)inst + LANE] = L[P_COM + LANE];
}
#ifndef LMH_ROLLOUT_ONLY
extern "C" void lmh_launch_com(const LmhDevParams *P, const double *q, double *com, hipStream_t s)
{
    hipLaunchKernelGGL(lmh_com_kernel, dim3(P->n_instances), dim3(64), 0, s, *P, q, com);
}

#endif
// ============================================================================ reference generators on the device
// The reference declares a walking generator (ZMP(Task, numSteps, timePerStep, simulationTime) / walkZMP, zmpGeneration.hpp:15-22) but never
// defines it, and produces one polynomial set per step with footCoeffTrajectory / findPolyCoeff (footRefTrajectory.cpp:4-47,
// generalizedFunctions.cpp:103-163).  The build's definition (the same one linearmpchumanoid_amd/trajectories.walk_plan states on the host):
//   settle in double support, then num_steps steps of [double support ds_time | single support time_per_step - ds_time]; the ZMP sits at the
//   mid-point of the feet in double support and under the support foot (y = -/+ foot_y) in single support; the swing foot follows the
//   footCoeffTrajectory polynomials (5th order x / y, 7th order z through step_height at half time); x is in units of the step length
//   (first and last step one unit, the others two), the per-instance scale (lmh_set_xscale) turns it into metres.
// The swing polynomials are written in closed form instead of solving findPolyCoeff's Vandermonde system: with s = t / T,
//   x, y : p0 + (p1 - p0)(10 s^3 - 15 s^4 + 6 s^5)
//   z    : z0 P0(s) + h 64 r(s) + z1 P0(1 - s),   r = s^3 (1 - s)^3,   P0 = 1 - (10 s^3 - 15 s^4 + 6 s^5) - 32 r + 120 (s - 1/2) r
// which is THE polynomial of degree <= 7 through the eight conditions of footRefTrajectory.cpp:20-44 (agrees with the solved system to 2e-13).
__device__ __forceinline__ int gen_idx(double t, double time_step, int n)      // min(n, max(0, int(round(t / time_step)))), round half to even
{
    const double r = rint(t / time_step);
    const int i = (r < 0.0) ? 0 : (r > (double)n ? n : (int)r);
    return i;
}
__global__ void __launch_bounds__(256) lmh_gen_walk_kernel(LmhWalkSpec W, double *zx, double *zy, uint8_t *phase, double *segs, uint16_t *sos)
{
    __shared__ int s_a[LMH_GEN_MAX_STEPS + 1], s_b[LMH_GEN_MAX_STEPS], s_c[LMH_GEN_MAX_STEPS], s_sup[LMH_GEN_MAX_STEPS];
    __shared__ double s_xr[LMH_GEN_MAX_STEPS + 1], s_xl[LMH_GEN_MAX_STEPS + 1], s_t[LMH_GEN_MAX_STEPS + 1];
    const int n = W.n_samples, ns = W.num_steps, tid = threadIdx.x;
    if (tid == 0) {                                                // the step boundaries accumulate t += time_per_step in order: one thread
        double t = W.settle_time, xr = 0.0, xl = 0.0;
        int sup = W.first_support;
        for (int s = 0; s < ns; s++) {
            const double stride = (s == 0 || s == ns - 1) ? 1.0 : 2.0;
            s_a[s] = gen_idx(t, W.time_step, n); s_b[s] = gen_idx(t + W.ds_time, W.time_step, n); s_c[s] = gen_idx(t + W.time_per_step, W.time_step, n);
            s_xr[s] = xr; s_xl[s] = xl; s_t[s] = t; s_sup[s] = sup;
            if (sup == LMH_PHASE_RIGHT) xl += stride; else xr += stride;
            sup = (sup == LMH_PHASE_RIGHT) ? LMH_PHASE_LEFT : LMH_PHASE_RIGHT;
            t += W.time_per_step;
        }
        s_a[ns] = gen_idx(t, W.time_step, n); s_xr[ns] = xr; s_xl[ns] = xl; s_t[ns] = t;
    }
    __syncthreads();
    // ---- segment records: 0 = initial stance, 1 + 2 s = double support of step s, 2 + 2 s = its single support, 2 ns + 1 = final stance
    const int n_seg = 2 * ns + 2;
    for (int e = tid; e < n_seg * LMH_SEG_STRIDE; e += blockDim.x) {
        const int g = e / LMH_SEG_STRIDE, f = e % LMH_SEG_STRIDE;
        const bool first = g == 0, last = g == n_seg - 1;
        const int s = first ? 0 : last ? ns : (g - 1) >> 1;
        const bool swing = !first && !last && ((g - 1) & 1);
        const double xr = s_xr[s], xl = s_xl[s];
        double v = 0.0;
        if (f == 0) v = first ? 0.0 : swing ? (double)s_b[s] * W.time_step : s_t[s];
        else if (f <= 48) {
            const int ft = (f - 1) / 24, ax = ((f - 1) % 24) / 8, k = (f - 1) % 8;
            const double x0 = ft ? xl : xr, y0 = ft ? W.foot_y : -W.foot_y;
            const bool moving = swing && ((s_sup[s] == LMH_PHASE_RIGHT) == (ft == 1));    // right foot supports <-> the left foot swings
            if (!moving) v = (k == 0) ? ((ax == 0) ? x0 : (ax == 1) ? y0 : 0.0) : 0.0;
            else {
                const double T = (double)(s_c[s] - s_b[s]) * W.time_step;
                const double stride = (s == 0 || s == ns - 1) ? 1.0 : 2.0;
                double chat;                                       // coefficient of s^k, s = t / T
                if (ax == 0) { const double d = (x0 + stride) - x0; chat = (k == 0) ? x0 : (k == 3) ? 10.0 * d : (k == 4) ? -15.0 * d : (k == 5) ? 6.0 * d : 0.0; }
                else if (ax == 1) chat = (k == 0) ? y0 : 0.0;      // the swing foot keeps its y
                else chat = W.step_height * ((k == 3) ? 64.0 : (k == 4) ? -192.0 : (k == 5) ? 192.0 : (k == 6) ? -64.0 : 0.0);   // z0 = z1 = 0
                double tp = 1.0;
                for (int i = 0; i < k; i++) tp *= T;
                v = chat / tp;
            }
        }
        segs[e] = v;
    }
    // ---- samples: the assignments of the host generator in their order (the last one that covers k wins)
    for (int k = tid; k < n; k += blockDim.x) {
        double x = 0.0, y = 0.0;
        int ph = LMH_PHASE_DOUBLE, sg = 0;
        for (int s = 0; s < ns; s++) {
            if (k >= s_a[s] && k < s_b[s]) { x = 0.5 * (s_xr[s] + s_xl[s]); y = 0.0; ph = LMH_PHASE_DOUBLE; sg = 1 + 2 * s; }
            if (k >= s_b[s] && k < s_c[s]) {
                const bool right = s_sup[s] == LMH_PHASE_RIGHT;
                x = right ? s_xr[s] : s_xl[s]; y = right ? -W.foot_y : W.foot_y; ph = s_sup[s]; sg = 2 + 2 * s;
            }
        }
        if (k >= s_a[ns]) { x = 0.5 * (s_xr[ns] + s_xl[ns]); y = 0.0; ph = LMH_PHASE_DOUBLE; sg = 2 * ns + 1; }
        zx[k] = x; zy[k] = y; phase[k] = (uint8_t)ph; sos[k] = (uint16_t)sg;
    }
}
// jumping schedule (BASELINE config 5): stance ZMP references (ZMP::stanceZMP, Double), PHASE_FLIGHT in [stance, stance + flight)
__global__ void __launch_bounds__(256) lmh_gen_jump_kernel(int n, double time_step, double stance_time, double flight_time, double *zx, double *zy, uint8_t *phase)
{
    const int a = gen_idx(stance_time, time_step, n), b = gen_idx(stance_time + flight_time, time_step, n);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        zx[k] = 0.0; zy[k] = 0.0;
        phase[k] = (uint8_t)((k >= a && k < b) ? LMH_PHASE_FLIGHT : LMH_PHASE_DOUBLE);
    }
}
extern "C" void lmh_launch_gen_walk(const LmhWalkSpec *W, double *zx, double *zy, uint8_t *phase, double *segs, uint16_t *sos, hipStream_t s)
{
    hipLaunchKernelGGL(lmh_gen_walk_kernel, dim3(1), dim3(256), 0, s, *W, zx, zy, phase, segs, sos);
}
extern "C" void lmh_launch_gen_jump(int n, double time_step, double stance_time, double flight_time, double *zx, double *zy, uint8_t *phase, hipStream_t s)
{
    hipLaunchKernelGGL(lmh_gen_jump_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, time_step, stance_time, flight_time, zx, zy, phase);
}

#ifndef LMH_ROLLOUT_ONLY
extern "C" void lmh_launch_eval(const LmhDevParams *P, double *state, double *out, int32_t *status, double *debug, hipStream_t s)
{
    // precision 1 (LMH_PRECISION_MIXED): model terms in fp32 arithmetic, references and QP in fp64; the debug kernel is fp64 only
    // LMH_DIAG_NW2=1: the debug kernel on the two-wave schedule (per-wave phase stamps; diagnostics only)
    static const bool diag_nw2 = getenv("LMH_DIAG_NW2") != nullptr;     // read once
    if (debug && diag_nw2) hipLaunchKernelGGL((lmh_eval_kernel<true, double, 2>), dim3(P->n_instances), dim3(LMH_ROLLOUT_THREADS), 0, s, *P, state, out, status, debug);
    else if (debug && P->precision == 2) hipLaunchKernelGGL((lmh_eval_kernel<true, float, 1, true>), dim3(P->n_instances), dim3(64), 0, s, *P, state, out, status, debug);
    else if (debug) hipLaunchKernelGGL((lmh_eval_kernel<true, double>), dim3(P->n_instances), dim3(64), 0, s, *P, state, out, status, debug);
    else if (P->precision == 2) hipLaunchKernelGGL((lmh_eval_kernel<false, float, 2, true>), dim3(P->n_instances), dim3(LMH_ROLLOUT_THREADS), 0, s, *P, state, out, status, debug);
    else if (P->precision == 1) hipLaunchKernelGGL((lmh_eval_kernel<false, float>), dim3(P->n_instances), dim3(LMH_ROLLOUT_THREADS), 0, s, *P, state, out, status, debug);
    else hipLaunchKernelGGL((lmh_eval_kernel<false, double>), dim3(P->n_instances), dim3(LMH_ROLLOUT_THREADS), 0, s, *P, state, out, status, debug);
}
#endif
// d_P: device copy of *P (the rollout kernel reads its parameters through a pointer, see lmh_rollout_kernel)
// Workgroups the device holds at once: LDS admits four robots per CU (40 KB each of 160 KB).
static int rollout_resident_groups()
{
    static int cached[64] = {0};                                   // per device index: a process may drive GPUs with different CU counts
    static int per_cu = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (per_cu == 0) {
            per_cu = 4;
            if (const char *e = getenv("LMH_ROLLOUT_GROUPS_PER_CU")) { const int v = atoi(e); if (v > 0) per_cu = v; }    // experiments only; read once
        }
        cached[dev] = cus * per_cu;
    }
    return cached[dev];
}
// Diagnostic (scripts/occupancy_probe.py): what the runtime's occupancy calculator and the code object say about lmh_rollout_kernel<double>:
// workgroups per CU, architected registers per lane, static LDS per workgroup.  Not part of the C ABI of include/lmh.h.
extern "C" int lmh_debug_rollout_occupancy(int *groups_per_cu, int *num_regs, int *static_lds_bytes)
{
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(&lmh_rollout_kernel<double, false>)) != hipSuccess) return -1;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lmh_rollout_kernel<double, false>, LMH_ROLLOUT_THREADS, 0) != hipSuccess) return -2;
    *groups_per_cu = nb; *num_regs = at.numRegs; *static_lds_bytes = (int)at.sharedSizeBytes;
    return 0;
}
// Diagnostic: which experiment switches THIS library was compiled with (bit 0 LMH_POISON, bit 1 LMH_SUBSTAMPS, bit 2 a non-default
// LMH_SPIN_LIMIT, bit 3 LMH_NUM_VGPR, bit 4 LMH_LDS_PROBE_DOUBLES, bit 5 LMH_TEST_LOSE_PUSH); the checker tests assert it, so that a variant library built without
// its flag cannot pass them vacuously.  Not part of the C ABI of include/lmh.h.
extern "C" int lmh_debug_build_flags(void)
{
    int f = 0;
#ifdef LMH_POISON
    f |= 1;
#endif
#ifdef LMH_SUBSTAMPS
    f |= 2;
#endif
    if (LMH_SPIN_LIMIT != (1 << 26)) f |= 4;
#ifdef LMH_NUM_VGPR
    f |= 8;
#endif
#ifdef LMH_LDS_PROBE_DOUBLES
    f |= 16;
#endif
#ifdef LMH_TEST_LOSE_PUSH
    f |= 32;
#endif
    return f;
}
// d_ticket: zero-initialised device memory owned by this launch until it completes (work-unit counters, ring, progress: see the kernel)
extern "C" void lmh_launch_rollout(const LmhDevParams *P, const LmhDevParams *d_P, int *d_ticket, double *state, double *out, int32_t *status, double *log, int n_ticks, hipStream_t s)
{
    const int slots = rollout_resident_groups();
    const dim3 grid((unsigned)((P->n_instances < slots) ? P->n_instances : slots));
#ifndef LMH_ROLLOUT_ONLY
    if (P->precision == 2) hipLaunchKernelGGL((lmh_rollout_kernel<float, true>), grid, dim3(LMH_ROLLOUT_THREADS), 0, s, d_P, d_ticket, state, out, status, log, n_ticks);
    else if (P->precision == 1) hipLaunchKernelGGL(lmh_rollout_kernel<float>, grid, dim3(LMH_ROLLOUT_THREADS), 0, s, d_P, d_ticket, state, out, status, log, n_ticks);
    else
#endif
    hipLaunchKernelGGL(lmh_rollout_kernel<double>, grid, dim3(LMH_ROLLOUT_THREADS), 0, s, d_P, d_ticket, state, out, status, log, n_ticks);
}
#ifndef LMH_ROLLOUT_ONLY
extern "C" void lmh_launch_model(const double *raw, double *model, int n_models, const double *lcoef, hipStream_t s)
{
    hipLaunchKernelGGL(lmh_model_kernel, dim3(n_models), dim3(64), 0, s, raw, model, n_models, lcoef);
}
#endif
