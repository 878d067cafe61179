// lmh_capi.hip -- host side of the C ABI in include/lmh.h: owns the device-resident tables
// (model, MPC gain rows, ZMP / phase references, friction generators), launches the gfx950
// kernels in lmh_kernels.hip.  No CPU compute path exists: without a HIP device every entry
// point fails with LMH_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <utility>
#include <string>
#include <vector>
#include "../../include/lmh.h"
#include "lmh_device.h"
#include "lmh_nao_model.h"

extern "C" void lmh_launch_eval(const LmhDevParams *P, double *state, double *out, int32_t *status, double *debug, hipStream_t s);
extern "C" void lmh_launch_rollout(const LmhDevParams *P, const LmhDevParams *d_P, int *d_ticket, double *state, double *out, int32_t *status, double *log, int n_ticks, hipStream_t s);
extern "C" void lmh_launch_model(const double *raw, double *model, int n_models, const double *lcoef, hipStream_t s);
extern "C" void lmh_launch_com(const LmhDevParams *P, const double *q, double *com, hipStream_t s);
extern "C" void lmh_launch_ik(const LmhDevParams *P, double *q, const LmhIkTarget *target, int32_t *iters, hipStream_t s);
extern "C" void lmh_launch_gen_walk(const LmhWalkSpec *W, double *zx, double *zy, uint8_t *phase, double *segs, uint16_t *sos, hipStream_t s);
extern "C" void lmh_launch_gen_jump(int n, double time_step, double stance_time, double flight_time, double *zx, double *zy, uint8_t *phase, hipStream_t s);
extern "C" void lmh_launch_summary(int n, const double *state, const double *out, const int32_t *status, double *summary, hipStream_t s);

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(LMH_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct lmh_handle {
    lmh_config cfg;
    double mpc_dt = 0.0;              // resolved MPC sample time (cfg.mpc_dt, or cfg.dt when that is 0)
    int B = 0, device = 0, N = 0, n_models = 0, n_gain = 0, n_samples = 0;
    double *d_model = nullptr, *d_mpc = nullptr, *d_zx = nullptr, *d_zy = nullptr, *d_gcol = nullptr, *d_raw = nullptr;
    double *d_segs = nullptr, *d_xscale = nullptr;
    uint16_t *d_sos = nullptr;
    int n_seg = 0;
    uint8_t *d_phase = nullptr;
    // staging for the host-buffer convenience calls
    double *d_state = nullptr, *d_out = nullptr;
    int32_t *d_status = nullptr;
    std::vector<double> h_state, h_out, h_gain;
    std::vector<int32_t> h_status;
    LmhDevParams P;
    // Launch slots of lmh_rollout: the kernel reads its parameter block through a pointer and draws robots from a ticket word, so every
    // launch in flight needs its own copy of both.  Slot i is reused by launch i + kSlots, after the event recorded behind launch i has
    // completed (normally long ago): launches on different streams of one handle never share a block that is being rewritten.
    static constexpr int kSlots = 8;
    struct Slot {
        LmhDevParams *d_P = nullptr;  // device copy read by the rollout kernel
        int *d_ticket = nullptr;      // work-unit counters, ring queue of robots, ticks done per robot (4 + 2 B ints, all zero between launches: the kernel leaves them so)
        hipEvent_t done = nullptr;    // recorded behind the last launch that used this slot
        LmhDevParams P_dev;           // what d_P currently holds
        bool valid = false, used = false;
        bool checked = true;          // the error word of the last launch on this slot (d_ticket[3]) has been read
    } slot[kSlots];
    unsigned next_slot = 0;
};

extern "C" const char *lmh_last_error(void) { return g_err.c_str(); }

extern "C" int lmh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void lmh_config_default(lmh_config *c)
{
    std::memset(c, 0, sizeof(*c));
    c->dt = 0.01; c->time_horizon = 0.5; c->z_com = 0.26;          // apps/offline/main.cpp:13-14,31,38
    c->gravity = 9.81; c->alpha = 1e-3; c->beta = 1.0;              // mpcLinearPendulum.hpp:46-48
    c->mu = 0.7;                                                    // controller.hpp:81
    c->kp_joints = 300; c->kd_joints = 34;                          // controller.hpp:102-103
    c->kp_mom = 10; c->kd_mom = 6.32;                               // :106-107
    c->kp_feet = 500; c->kd_feet = 44;                              // :110-111
    c->w_com_lin = 4000; c->w_com_ang = 0; c->w_base_pos = 10; c->w_base_ang = 10;   // :118-121
    c->w_joints = 1; c->w_force = 1; c->w_foot = 100000;            // :122-124
    c->eps_coeff = 1e-8;                                            // controller.cpp:117
    c->warm_start = 1; c->max_qp_iters = 64; c->precision = LMH_PRECISION_FP64; c->bpp_rounds = 0;
    c->plant = 0; c->contact_k = 2.0e4; c->contact_d = 3.0; c->contact_dt = 3.0; c->contact_mu = 0.7;
    c->mpc_dt = 0.0;                                                // = dt (apps/offline/main.cpp:18,21,39 pass one value to Clock, ZMP and Mpc3dLip)
}

extern "C" void lmh_nominal_links(double *raw) { std::memcpy(raw, kLmhNaoLinks, sizeof(kLmhNaoLinks)); }

// ---- Mpc3dLip::initialize (src/mpcLinearPendulum.cpp:41-68) + the algebraic gain row:
// u = -H^-1 g, g = beta Pu'(Px x - z), H = alpha I + beta Pu'Pu  =>  u0 = -K (Px x - z),
// K = beta e0' H^-1 Pu'.  Record layout: K | Px[:,0] | Px[:,1] | zcom | pad(3).
static int build_gain_row(const lmh_config &c, double dt /* MPC sample time */, double zcom, int N, double *rec)
{
    const int n = N + 1;
    std::vector<double> Pu((size_t)n * n, 0.0), H((size_t)n * n), h0(n, 0.0);
    double A[4] = {1, dt, 0, 1}, B[2] = {(dt * dt) / 2, dt}, Ap[4] = {1, 0, 0, 1};
    const double D = -zcom / c.gravity;
    double *K = rec, *px0 = rec + n, *px1 = rec + 2 * n;
    px0[0] = 1; px1[0] = 0;
    Pu[0] = D;
    for (int i = 1; i <= N; i++) {
        double t[4] = {Ap[0] * A[0] + Ap[1] * A[2], Ap[0] * A[1] + Ap[1] * A[3], Ap[2] * A[0] + Ap[3] * A[2], Ap[2] * A[1] + Ap[3] * A[3]};
        std::memcpy(Ap, t, sizeof(t));
        px0[i] = Ap[0]; px1[i] = Ap[1];                             // C = [1 0]
        Pu[(size_t)i * n + i - 1] = B[0];
        Pu[(size_t)i * n + i] = D;
        double Aj[4] = {1, 0, 0, 1};
        for (int j = 1; j <= N - i; j++) {
            double u[4] = {Aj[0] * A[0] + Aj[1] * A[2], Aj[0] * A[1] + Aj[1] * A[3], Aj[2] * A[0] + Aj[3] * A[2], Aj[2] * A[1] + Aj[3] * A[3]};
            std::memcpy(Aj, u, sizeof(u));
            Pu[(size_t)(i + j) * n + i - 1] = Aj[0] * B[0] + Aj[1] * B[1];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int l = 0; l < n; l++) s += Pu[(size_t)l * n + i] * Pu[(size_t)l * n + j];
            H[(size_t)i * n + j] = ((i == j) ? c.alpha : 0.0) + c.beta * s;
        }
    for (int j = 0; j < n; j++) {                                   // Cholesky (lower)
        double d = H[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= H[(size_t)j * n + k] * H[(size_t)j * n + k];
        if (!(d > 0)) return 1;
        d = std::sqrt(d);
        H[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = H[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= H[(size_t)i * n + k] * H[(size_t)j * n + k];
            H[(size_t)i * n + j] = s / d;
        }
    }
    std::vector<double> y(n, 0.0);
    for (int i = 0; i < n; i++) {                                   // H h0 = e0
        double s = (i == 0) ? 1.0 : 0.0;
        for (int k = 0; k < i; k++) s -= H[(size_t)i * n + k] * y[k];
        y[i] = s / H[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = y[i];
        for (int k = i + 1; k < n; k++) s -= H[(size_t)k * n + i] * h0[k];
        h0[i] = s / H[(size_t)i * n + i];
    }
    for (int i = 0; i < n; i++) {
        double s = 0;
        for (int j = 0; j < n; j++) s += Pu[(size_t)i * n + j] * h0[j];
        K[i] = c.beta * s;
    }
    rec[3 * n] = zcom; rec[3 * n + 1] = rec[3 * n + 2] = rec[3 * n + 3] = 0.0;
    return 0;
}

// friction-cone generators: column j = 16 foot + 4 vertex + edge  ->  [p_v x ray_e ; ray_e]
// (src/controller.cpp:33-36,185-270, vertices src/Robot.cpp:38-42; same for both feet)
static void build_gcol(double mu, double *g /*[16][6] (one foot; both feet are identical) | (G G')^-1 [6][6] | G'(G G')^-1 [16][6]*/)
{
    const double ray[4][3] = {{mu, 0, 1}, {0, mu, 1}, {-mu, 0, 1}, {0, -mu, 1}};
    const double vtx[4][3] = {{0.1, 0.025, 0}, {0.1, -0.025, 0}, {-0.05, 0.025, 0}, {-0.05, -0.025, 0}};
    {
        for (int v = 0; v < 4; v++)
            for (int e = 0; e < 4; e++) {
                double *o = g + 6 * (4 * v + e);
                const double *p = vtx[v], *r = ray[e];
                // crossMatrix(p) * ray, term by term as the dense product does
                o[0] = 0 * r[0] + (-p[2]) * r[1] + p[1] * r[2];
                o[1] = p[2] * r[0] + 0 * r[1] + (-p[0]) * r[2];
                o[2] = (-p[1]) * r[0] + p[0] * r[1] + 0 * r[2];
                o[3] = r[0]; o[4] = r[1]; o[5] = r[2];
            }
    }
    // Gamma = G_f G_f' (6x6, identical for both feet), its inverse and the min-norm map G_f' Gamma^-1
    double Gm[36] = {0}, A[6][12];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++)
            for (int j = 0; j < 16; j++) Gm[6 * a + b] += g[6 * j + a] * g[6 * j + b];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 12; b++) A[a][b] = (b < 6) ? Gm[6 * a + b] : ((b - 6 == a) ? 1.0 : 0.0);
    for (int c = 0; c < 6; c++) {                                   // Gauss-Jordan with partial pivoting
        int pv = c;
        for (int rr = c + 1; rr < 6; rr++) if (std::fabs(A[rr][c]) > std::fabs(A[pv][c])) pv = rr;
        for (int b = 0; b < 12; b++) std::swap(A[c][b], A[pv][b]);
        const double d = A[c][c];
        for (int b = 0; b < 12; b++) A[c][b] /= d;
        for (int rr = 0; rr < 6; rr++)
            if (rr != c) { const double f = A[rr][c]; for (int b = 0; b < 12; b++) A[rr][b] -= f * A[c][b]; }
    }
    double *gi = g + 96, *gp = g + 96 + 36;
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) gi[6 * a + b] = 0.5 * (A[a][6 + b] + A[b][6 + a]);
    for (int j = 0; j < 16; j++)
        for (int b = 0; b < 6; b++) {
            double sacc = 0;
            for (int a = 0; a < 6; a++) sacc += g[6 * j + a] * gi[6 * a + b];
            gp[6 * j + b] = sacc;
        }
}

// Local-transform coefficient table: entry e = 12 slot + 4 row + col of slot's 3x4 transform equals
// c0 + c1 cos(theta_slot) + c2 sin(theta_slot).  Slots 0..24 are the Khalil modified-DH transforms of
// matTrans (src/Robot.cpp:176-223; tables :180-196) with the shoulder / head offsets of :134,143,152 folded
// into the translations of slots 12, 17, 22; slots 25..27 are auxT01, auxT09 (0.7071 literals, :92-103) and
// the sole offset (:106-117).  cos/sin(alpha) are libm's values for the reference's pi literal.
static void build_lcoef(double *t /*[336][3]*/)
{
    const double pi = 3.14159265358979323846, h = pi / 2;
    const double r[25] = {-0.07071, 0, 0, 0, 0, 0, 0.07071, 0, 0, 0, 0, 0, 0, 0, 0.105, 0, 0.05595, 0, 0, 0.105, 0, 0.05595, 0, 0, 0};
    const double d[25] = {0, 0, 0, -0.1, -0.1029, 0, 0, 0, 0, -0.1, -0.1029, 0, 0, 0, -0.015, 0, 0, 0, 0, -0.015, 0, 0, 0, 0, 0.030};
    const double al[25] = {0, h, h, 0, 0, -h, -h, -h, h, 0, 0, -h, -h, h, h, -h, h, h, h, h, -h, h, 0, -h, 0};
    const double aux[3][12] = {{0, -1, 0, 0, 0.7071, 0, 0.7071, 0, -0.7071, 0, 0.7071, 0},
                               {1, 0, 0, 0, 0, 0.7071, 0.7071, 0, 0, -0.7071, 0.7071, 0},
                               {1, 0, 0, -0.0452, 0, 1, 0, 0, 0, 0, 1, 0}};
    std::memset(t, 0, sizeof(double) * 3 * 336);
    for (int s = 0; s < 25; s++) {
        const double ca = std::cos(al[s]), sa = std::sin(al[s]);
        double *e = t + 3 * 12 * s;
        // row 0: ct, -st, 0, d
        e[3 * 0 + 1] = 1.0; e[3 * 1 + 2] = -1.0; e[3 * 3 + 0] = d[s];
        // row 1: ca st, ca ct, -sa, -r sa
        e[3 * 4 + 2] = ca; e[3 * 5 + 1] = ca; e[3 * 6 + 0] = -sa; e[3 * 7 + 0] = -r[s] * sa;
        // row 2: sa st, sa ct, ca, r ca
        e[3 * 8 + 2] = sa; e[3 * 9 + 1] = sa; e[3 * 10 + 0] = ca; e[3 * 11 + 0] = r[s] * ca;
        if (s == 12 || s == 17 || s == 22) {
            e[3 * 3 + 0] = e[3 * 3 + 0] + 0.0;
            e[3 * 7 + 0] = e[3 * 7 + 0] + ((s == 12) ? -0.098 : (s == 17) ? 0.098 : 0.0);
            e[3 * 11 + 0] = e[3 * 11 + 0] + ((s == 22) ? 0.1615 : 0.13591);
        }
    }
    for (int s = 25; s < 28; s++)
        for (int k = 0; k < 12; k++) t[3 * (12 * s + k)] = aux[s - 25][k];
}

static void fill_params(lmh_handle *h)
{
    LmhDevParams &P = h->P;
    const lmh_config &c = h->cfg;
    P.model = h->d_model; P.mpc = h->d_mpc; P.zmpx = h->d_zx; P.zmpy = h->d_zy; P.phase = h->d_phase; P.gcol = h->d_gcol;
    P.segs = h->d_segs; P.seg_of_sample = h->d_sos; P.xscale = h->d_xscale; P.n_seg = h->n_seg;
    P.model_stride = (h->n_models > 1) ? LMH_MODEL_STRIDE : 0;
    P.mpc_stride = 3 * (h->N + 1) + 4;
    P.mpc_stride_inst = (h->n_gain > 1) ? P.mpc_stride : 0;
    P.n_samples = h->n_samples; P.horizon = h->N; P.n_instances = h->B;
    P.warm_start = c.warm_start; P.max_qp_iters = c.max_qp_iters; P.precision = c.precision;
    P.bpp_max = (c.bpp_rounds == 0) ? 10 : c.bpp_rounds;            // < 0: Lawson-Hanson from the empty set (diagnostic)
    P.plant = c.plant; P.contact_k = c.contact_k; P.contact_d = c.contact_d; P.contact_dt = c.contact_dt; P.contact_mu = c.contact_mu;
    P.dt = c.dt; P.mpc_dt = h->mpc_dt;
    P.kp_joints = c.kp_joints; P.kd_joints = c.kd_joints; P.kp_mom = c.kp_mom; P.kd_mom = c.kd_mom;
    P.kp_feet = c.kp_feet; P.kd_feet = c.kd_feet;
    P.w_com_lin = c.w_com_lin; P.w_com_ang = c.w_com_ang; P.w_base_pos = c.w_base_pos; P.w_base_ang = c.w_base_ang;
    P.w_joints = c.w_joints; P.w_force = c.w_force; P.w_foot = c.w_foot; P.eps_coeff = c.eps_coeff;
    P.inv_w_base_pos = 1.0 / c.w_base_pos; P.inv_w_base_ang = 1.0 / c.w_base_ang; P.inv_w_joints = 1.0 / c.w_joints;
    P.inv_w_com_lin = 1.0 / c.w_com_lin; P.inv_w_foot = 1.0 / c.w_foot;
    const double md = h->mpc_dt;                                     // mpcLinearPendulum.cpp:45-47 with the Mpc3dLip ctor's dt
    P.a00 = 1; P.a01 = md; P.a10 = 0; P.a11 = 1; P.b0 = (md * md) / 2; P.b1 = md;
}

static int upload_gain(lmh_handle *h, const double *zcom, int n)
{
    const int stride = 3 * (h->N + 1) + 4;
    h->h_gain.assign((size_t)n * stride, 0.0);
    for (int i = 0; i < n; i++)
        if (build_gain_row(h->cfg, h->mpc_dt, zcom[i], h->N, h->h_gain.data() + (size_t)i * stride))
            return fail(LMH_ERR_BAD_ARG, "MPC Hessian not positive definite");
    if (h->d_mpc) { HIPCHK(hipFree(h->d_mpc)); h->d_mpc = nullptr; }
    HIPCHK(hipMalloc(&h->d_mpc, sizeof(double) * h->h_gain.size()));
    HIPCHK(hipMemcpy(h->d_mpc, h->h_gain.data(), sizeof(double) * h->h_gain.size(), hipMemcpyHostToDevice));
    h->n_gain = n;
    fill_params(h);
    return LMH_OK;
}

// every literal the kernels divide by or take a Cholesky pivot from must be positive (a zero weight is 1/0 in the
// Woodbury set-up; the reference has no such check because its literals are compile-time constants)
static const char *validate_config(const lmh_config *c)
{
    if (!(c->dt > 0.0) || !(c->time_horizon > 0.0)) return "dt and time_horizon must be positive";
    if (!(c->mpc_dt >= 0.0) || !std::isfinite(c->mpc_dt)) return "mpc_dt must be >= 0 (0 = dt)";
    if (!(c->z_com > 0.0) || !(c->gravity > 0.0)) return "z_com and gravity must be positive";
    if (!(c->alpha > 0.0) || !(c->beta > 0.0)) return "alpha and beta must be positive";
    if (!(c->mu > 0.0)) return "mu must be positive";
    if (!(c->eps_coeff > 0.0)) return "eps_coeff must be positive";
    if (!(c->w_com_lin > 0.0) || !(c->w_base_pos > 0.0) || !(c->w_base_ang > 0.0) || !(c->w_joints > 0.0) || !(c->w_force > 0.0) || !(c->w_foot > 0.0))
        return "weights w_com_lin, w_base_pos, w_base_ang, w_joints, w_force, w_foot must be positive";
    if (!(c->w_com_ang >= 0.0)) return "w_com_ang must be >= 0";
    const double g[6] = {c->kp_joints, c->kd_joints, c->kp_mom, c->kd_mom, c->kp_feet, c->kd_feet};
    for (double v : g) if (!std::isfinite(v)) return "PD gains must be finite";
    if (c->max_qp_iters < 1) return "max_qp_iters must be >= 1";
    if (c->plant != 0 && c->plant != 1) return "plant must be 0 or 1";
    if (c->plant && (!(c->contact_k > 0.0) || !(c->contact_d >= 0.0) || !(c->contact_dt >= 0.0) || !(c->contact_mu >= 0.0)))
        return "contact_k must be positive, contact_d / contact_dt / contact_mu non-negative";
    if (c->precision != LMH_PRECISION_FP64 && c->precision != LMH_PRECISION_MIXED && c->precision != LMH_PRECISION_FP32)
        return "precision must be LMH_PRECISION_FP64, LMH_PRECISION_MIXED or LMH_PRECISION_FP32";
    return nullptr;
}

static int create_body(lmh_handle *h, const lmh_config *cfg, int n_instances)
{
    std::memset(&h->P, 0, sizeof(h->P));
    double g[16 * 6 + 36 + 96 + 3 * 336];
    build_gcol(cfg->mu, g);
    build_lcoef(g + 228);
    HIPCHK(hipMalloc(&h->d_gcol, sizeof(g)));
    HIPCHK(hipMemcpy(h->d_gcol, g, sizeof(g), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&h->d_state, sizeof(double) * LMH_STATE_STRIDE * (size_t)n_instances));
    HIPCHK(hipMalloc(&h->d_out, sizeof(double) * LMH_OUT_STRIDE * (size_t)n_instances));
    HIPCHK(hipMalloc(&h->d_status, sizeof(int32_t) * LMH_STATUS_STRIDE * (size_t)n_instances));
    HIPCHK(hipMemset(h->d_state, 0, sizeof(double) * LMH_STATE_STRIDE * (size_t)n_instances));
    HIPCHK(hipMemset(h->d_status, 0, sizeof(int32_t) * LMH_STATUS_STRIDE * (size_t)n_instances));
    h->h_state.assign((size_t)LMH_STATE_STRIDE * n_instances, 0.0);
    h->h_out.assign((size_t)LMH_OUT_STRIDE * n_instances, 0.0);
    h->h_status.assign((size_t)LMH_STATUS_STRIDE * n_instances, 0);
    int rc = lmh_set_model(h, nullptr, 1);
    if (rc == LMH_OK) rc = upload_gain(h, &cfg->z_com, 1);
    if (rc == LMH_OK) rc = lmh_set_refs_stance(h, 5.0, 2);
    if (rc == LMH_OK) {                                             // constant foot references at (0, -/+0.05, 0)
        double r[24] = {0}, l[24] = {0};
        int32_t n[3] = {6, 6, 8};
        r[8] = -0.05; l[8] = 0.05;
        rc = lmh_set_foot_coeffs(h, r, n, l, n);
    }
    return rc;
}

extern "C" int lmh_create(const lmh_config *cfg, int n_instances, int device, lmh_handle **out)
{
    if (!cfg || !out || n_instances < 1) return fail(LMH_ERR_BAD_ARG, "lmh_create: bad argument");
    *out = nullptr;
    if (const char *why = validate_config(cfg)) return fail(LMH_ERR_BAD_ARG, std::string("lmh_create: ") + why);
    const double mpc_dt = (cfg->mpc_dt > 0.0) ? cfg->mpc_dt : cfg->dt;
    const int N = (int)(cfg->time_horizon / mpc_dt);                // mpcLinearPendulum.cpp:43
    if (N < 1 || N > LMH_MAX_HORIZON) return fail(LMH_ERR_BAD_ARG, "horizon N = time_horizon/mpc_dt must be in [1, 64]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(LMH_ERR_NO_DEVICE, "no HIP device: the controller has no CPU path");
    if (device < 0 || device >= ndev) return fail(LMH_ERR_BAD_ARG, "lmh_create: device index out of range");
    HIPCHK(hipSetDevice(device));
    lmh_handle *h = new lmh_handle();
    h->cfg = *cfg; h->B = n_instances; h->device = device; h->N = N; h->mpc_dt = mpc_dt;
    const int rc = create_body(h, cfg, n_instances);                // every failure path releases what was allocated so far
    if (rc != LMH_OK) { std::string keep = g_err; lmh_destroy(h); g_err = keep; return rc; }
    *out = h;
    return LMH_OK;
}

extern "C" int lmh_destroy(lmh_handle *h)
{
    if (!h) return LMH_OK;
    (void)hipSetDevice(h->device);
    void *bufs[] = {h->d_model, h->d_mpc, h->d_zx, h->d_zy, h->d_gcol, h->d_raw, h->d_phase, h->d_state, h->d_out, h->d_status,
                    h->d_segs, h->d_xscale, h->d_sos};
    for (void *b : bufs) if (b) (void)hipFree(b);
    for (auto &sl : h->slot) {
        if (sl.done) { if (sl.used) (void)hipEventSynchronize(sl.done); (void)hipEventDestroy(sl.done); }
        if (sl.d_P) (void)hipFree(sl.d_P);
        if (sl.d_ticket) (void)hipFree(sl.d_ticket);
    }
    delete h;
    return LMH_OK;
}

extern "C" int lmh_num_instances(const lmh_handle *h) { return h ? h->B : 0; }
extern "C" int lmh_horizon(const lmh_handle *h) { return h ? h->N : 0; }

extern "C" int lmh_set_model(lmh_handle *h, const double *raw, int n_models)
{
    if (!h) return fail(LMH_ERR_BAD_ARG, "null handle");
    if (!raw) { raw = &kLmhNaoLinks[0][0]; n_models = 1; }
    if (n_models != 1 && n_models != h->B) return fail(LMH_ERR_BAD_ARG, "n_models must be 1 or n_instances");
    // frames 7, 14 (soles) and 27 (extra head frame) are massless virtual frames in createNaoParameters (src/robotParameters.cpp); the
    // kernels' tree recursions rely on that (they carry no body force and no inertia there)
    for (int m = 0; m < n_models; m++)
        for (int f : {7, 14, 27})
            for (int e = 0; e < LMH_LINK_STRIDE; e++)
                if (raw[((size_t)m * 28 + f) * LMH_LINK_STRIDE + e] != 0.0) return fail(LMH_ERR_BAD_ARG, "lmh_set_model: frames 7, 14, 27 are massless virtual frames: their records must be zero");
    HIPCHK(hipSetDevice(h->device));
    if (h->d_raw) { HIPCHK(hipFree(h->d_raw)); h->d_raw = nullptr; }
    if (h->d_model) { HIPCHK(hipFree(h->d_model)); h->d_model = nullptr; }
    const size_t rawb = sizeof(double) * 28 * LMH_LINK_STRIDE * (size_t)n_models;
    HIPCHK(hipMalloc(&h->d_raw, rawb));
    HIPCHK(hipMalloc(&h->d_model, sizeof(double) * LMH_MODEL_STRIDE * (size_t)n_models));
    HIPCHK(hipMemcpy(h->d_raw, raw, rawb, hipMemcpyHostToDevice));
    lmh_launch_model(h->d_raw, h->d_model, n_models, h->d_gcol + 228, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    h->n_models = n_models;
    fill_params(h);
    return LMH_OK;
}

extern "C" int lmh_get_mass(lmh_handle *h, double *mass)
{
    if (!h || !mass) return fail(LMH_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    for (int i = 0; i < h->n_models; i++)
        HIPCHK(hipMemcpy(mass + i, h->d_model + (size_t)i * LMH_MODEL_STRIDE + 392, sizeof(double), hipMemcpyDeviceToHost));
    return LMH_OK;
}

extern "C" int lmh_set_refs(lmh_handle *h, const double *zx, const double *zy, const uint8_t *phase, int n)
{
    if (!h || !zx || !zy || n < 1) return fail(LMH_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->d_zx) { HIPCHK(hipFree(h->d_zx)); h->d_zx = nullptr; }
    if (h->d_zy) { HIPCHK(hipFree(h->d_zy)); h->d_zy = nullptr; }
    if (h->d_phase) { HIPCHK(hipFree(h->d_phase)); h->d_phase = nullptr; }
    HIPCHK(hipMalloc(&h->d_zx, sizeof(double) * (size_t)n));
    HIPCHK(hipMalloc(&h->d_zy, sizeof(double) * (size_t)n));
    HIPCHK(hipMemcpy(h->d_zx, zx, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_zy, zy, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    if (phase) {
        HIPCHK(hipMalloc(&h->d_phase, (size_t)n));
        HIPCHK(hipMemcpy(h->d_phase, phase, (size_t)n, hipMemcpyHostToDevice));
    }
    h->n_samples = n;
    if (h->d_sos) { (void)hipFree(h->d_sos); h->d_sos = nullptr; }
    if (h->d_segs) { (void)hipFree(h->d_segs); h->d_segs = nullptr; }
    h->n_seg = 0;                                                    // segments are tied to the sample grid
    fill_params(h);
    return LMH_OK;
}

extern "C" int lmh_set_refs_stance(lmh_handle *h, double simulation_time, int support_foot)
{
    if (!h) return fail(LMH_ERR_BAD_ARG, "null handle");
    const int samples = (int)((simulation_time + 0.5) / h->mpc_dt);  // zmpGeneration.cpp:41 (timeStep_ = the MPC sample time)
    if (samples < 1) return fail(LMH_ERR_BAD_ARG, "no samples");
    std::vector<double> zx((size_t)samples, 0.0), zy((size_t)samples, (support_foot == 0) ? -0.05 : (support_foot == 1) ? 0.05 : 0.0);
    return lmh_set_refs(h, zx.data(), zy.data(), nullptr, samples);
}

extern "C" int lmh_set_foot_coeffs(lmh_handle *h, const double *r, const int32_t *rn, const double *l, const int32_t *ln)
{
    if (!h || !r || !rn || !l || !ln) return fail(LMH_ERR_BAD_ARG, "bad argument");
    for (int a = 0; a < 3; a++) {
        if (rn[a] < 1 || rn[a] > 8 || ln[a] < 1 || ln[a] > 8) return fail(LMH_ERR_BAD_ARG, "coefficient count must be 1..8");
        h->P.rFn[a] = rn[a]; h->P.lFn[a] = ln[a];
        // entries beyond the count are stored as zeros: the kernels evaluate all eight terms (a zero coefficient adds an exact zero)
        for (int k = 0; k < 8; k++) { h->P.rF[a][k] = (k < rn[a]) ? r[8 * a + k] : 0.0; h->P.lF[a][k] = (k < ln[a]) ? l[8 * a + k] : 0.0; }
    }
    return LMH_OK;
}

extern "C" int lmh_set_segments(lmh_handle *h, const double *segs, int n_seg, const uint16_t *sos, int n_samples)
{
    if (!h || n_seg < 0) return fail(LMH_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    if (h->d_segs) { HIPCHK(hipFree(h->d_segs)); h->d_segs = nullptr; }
    if (h->d_sos) { HIPCHK(hipFree(h->d_sos)); h->d_sos = nullptr; }
    h->n_seg = 0;
    if (n_seg > 0) {
        if (!segs || !sos) return fail(LMH_ERR_BAD_ARG, "null segment table");
        if (n_samples != h->n_samples) return fail(LMH_ERR_BAD_ARG, "seg_of_sample must cover the ZMP reference samples (call lmh_set_refs first)");
        for (int i = 0; i < n_samples; i++) if (sos[i] >= n_seg) return fail(LMH_ERR_BAD_ARG, "seg_of_sample entry out of range");
        HIPCHK(hipMalloc(&h->d_segs, sizeof(double) * LMH_SEG_STRIDE * (size_t)n_seg));
        HIPCHK(hipMalloc(&h->d_sos, sizeof(uint16_t) * (size_t)n_samples));
        HIPCHK(hipMemcpy(h->d_segs, segs, sizeof(double) * LMH_SEG_STRIDE * (size_t)n_seg, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_sos, sos, sizeof(uint16_t) * (size_t)n_samples, hipMemcpyHostToDevice));
        h->n_seg = n_seg;
    }
    fill_params(h);
    return LMH_OK;
}

extern "C" int lmh_set_xscale(lmh_handle *h, const double *xscale, int n)
{
    if (!h) return fail(LMH_ERR_BAD_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->d_xscale) { HIPCHK(hipFree(h->d_xscale)); h->d_xscale = nullptr; }
    if (xscale) {
        if (n != h->B) return fail(LMH_ERR_BAD_ARG, "n must be n_instances");
        HIPCHK(hipMalloc(&h->d_xscale, sizeof(double) * (size_t)n));
        HIPCHK(hipMemcpy(h->d_xscale, xscale, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    }
    fill_params(h);
    return LMH_OK;
}

extern "C" int lmh_set_zcom(lmh_handle *h, const double *z, int n)
{
    if (!h || !z || (n != 1 && n != h->B)) return fail(LMH_ERR_BAD_ARG, "n must be 1 or n_instances");
    HIPCHK(hipSetDevice(h->device));
    return upload_gain(h, z, n);
}

extern "C" int lmh_get_mpc_gain(lmh_handle *h, double *K)
{
    if (!h || !K) return fail(LMH_ERR_BAD_ARG, "bad argument");
    std::memcpy(K, h->h_gain.data(), sizeof(double) * (size_t)(h->N + 1));
    return LMH_OK;
}

static int ready(lmh_handle *h)
{
    if (!h) return fail(LMH_ERR_BAD_ARG, "null handle");
    if (!h->d_model || !h->d_mpc || !h->d_zx || h->n_samples < 1) return fail(LMH_ERR_NOT_READY, "model / references not set");
    return LMH_OK;
}

extern "C" int lmh_eval(lmh_handle *h, double *d_state, double *d_out, int32_t *d_status, void *stream)
{
    int rc = ready(h); if (rc) return rc;
    if (!d_state || !d_out || !d_status) return fail(LMH_ERR_BAD_ARG, "null device pointer");
    HIPCHK(hipSetDevice(h->device));
    lmh_launch_eval(&h->P, d_state, d_out, d_status, nullptr, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return LMH_OK;
}

extern "C" int lmh_eval_debug(lmh_handle *h, double *d_state, double *d_out, int32_t *d_status, double *d_debug, void *stream)
{
    int rc = ready(h); if (rc) return rc;
    if (!d_state || !d_out || !d_status || !d_debug) return fail(LMH_ERR_BAD_ARG, "null device pointer");
    HIPCHK(hipSetDevice(h->device));
    lmh_launch_eval(&h->P, d_state, d_out, d_status, d_debug, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return LMH_OK;
}

// The error word of a COMPLETED launch on this slot (d_ticket[3], lmh_rollout_kernel): read once, cleared, reported.  The kernel has already
// flagged the robots concerned and put the slot's ring / progress words back to zero.
static int slot_take_error(lmh_handle *h, lmh_handle::Slot &sl)
{
    if (sl.checked) return LMH_OK;
    int err = 0;
    HIPCHK(hipMemcpy(&err, sl.d_ticket + 3, sizeof(int), hipMemcpyDeviceToHost));
    sl.checked = true;
    if (err == 0) return LMH_OK;
    HIPCHK(hipMemset(sl.d_ticket + 3, 0, sizeof(int)));
    (void)h;
    return fail(LMH_ERR_UNFINISHED, "lmh_rollout: a wait of the kernel's work queue ran out (error word " + std::to_string(err) +
                "); the robots that did not get all their ticks carry LMH_FLAG_UNFINISHED in their status records");
}

extern "C" int lmh_rollout(lmh_handle *h, double *d_state, double *d_out, int32_t *d_status, double *d_log, int n_ticks, void *stream)
{
    int rc = ready(h); if (rc) return rc;
    if (!d_state || !d_out || !d_status || n_ticks < 0) return fail(LMH_ERR_BAD_ARG, "bad argument");
    if (n_ticks == 0) return LMH_OK;
    HIPCHK(hipSetDevice(h->device));
    lmh_handle::Slot &sl = h->slot[h->next_slot % lmh_handle::kSlots];
    if (!sl.d_P) {
        HIPCHK(hipMalloc(&sl.d_P, sizeof(LmhDevParams)));
        const size_t tbytes = (4 + 2 * (size_t)h->P.n_instances) * sizeof(int);   // counters | error word | ring of robots | ticks done per robot
        HIPCHK(hipMalloc(&sl.d_ticket, tbytes));
        HIPCHK(hipMemsetAsync(sl.d_ticket, 0, tbytes, (hipStream_t)stream));     // stream-ordered in front of the first launch on the slot
        HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    if (sl.used) {
        HIPCHK(hipEventSynchronize(sl.done));                       // the launch that last used this slot (kSlots launches ago) has left it
        rc = slot_take_error(h, sl);                                // ... and if it was incomplete, this call reports it instead of launching
        if (rc) return rc;
    }
    h->next_slot++;
    if (!sl.valid || std::memcmp(&sl.P_dev, &h->P, sizeof(LmhDevParams)) != 0) {   // set-up calls changed the block since this slot was filled
        std::memcpy(&sl.P_dev, &h->P, sizeof(LmhDevParams));
        HIPCHK(hipMemcpyAsync(sl.d_P, &sl.P_dev, sizeof(LmhDevParams), hipMemcpyHostToDevice, (hipStream_t)stream));   // stream-ordered in front of the launch; the slot is idle
        sl.valid = true;
    }
    lmh_launch_rollout(&h->P, sl.d_P, sl.d_ticket, d_state, d_out, d_status, d_log, n_ticks, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(sl.done, (hipStream_t)stream));
    sl.used = true;
    sl.checked = false;
    return LMH_OK;
}

extern "C" int lmh_ik(lmh_handle *h, double *d_q, const double *com_target, const double *rf6, const double *lf6, int32_t *d_iters, void *stream)
{
    int rc = ready(h); if (rc) return rc;
    if (!d_q || !com_target || !rf6 || !lf6) return fail(LMH_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    LmhIkTarget tgt;                                                 // travels by value in the kernel arguments: no shared staging buffer, no sync
    for (int k = 0; k < 6; k++) { tgt.v[k] = rf6[k]; tgt.v[6 + k] = lf6[k]; }
    for (int k = 0; k < 3; k++) tgt.v[12 + k] = com_target[k];
    tgt.v[15] = 0.0;
    lmh_launch_ik(&h->P, d_q, &tgt, d_iters, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return LMH_OK;
}

extern "C" int lmh_eval_host(lmh_handle *h, const double *q, const double *dq, double t, double *tau, double *f, double *qdd, int32_t *status)
{
    int rc = ready(h); if (rc) return rc;
    if (!q || !dq) return fail(LMH_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    // keep v_prev / active set from the previous call (Robot::v_ semantics): read-modify-write
    for (int i = 0; i < h->B; i++) {
        double *s = h->h_state.data() + (size_t)LMH_STATE_STRIDE * i;
        std::memcpy(s, q + 30 * (size_t)i, 30 * sizeof(double));
        std::memcpy(s + 30, dq + 30 * (size_t)i, 30 * sizeof(double));
        s[90] = t;
    }
    HIPCHK(hipMemcpy(h->d_state, h->h_state.data(), sizeof(double) * h->h_state.size(), hipMemcpyHostToDevice));
    lmh_launch_eval(&h->P, h->d_state, h->d_out, h->d_status, nullptr, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(h->h_state.data(), h->d_state, sizeof(double) * h->h_state.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h->h_out.data(), h->d_out, sizeof(double) * h->h_out.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h->h_status.data(), h->d_status, sizeof(int32_t) * h->h_status.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->B; i++) {
        const double *o = h->h_out.data() + (size_t)LMH_OUT_STRIDE * i;
        if (tau) std::memcpy(tau + 24 * (size_t)i, o, 24 * sizeof(double));
        if (f) std::memcpy(f + 12 * (size_t)i, o + 24, 12 * sizeof(double));
        if (qdd) std::memcpy(qdd + 30 * (size_t)i, o + 36, 30 * sizeof(double));
        if (status) std::memcpy(status + 4 * (size_t)i, h->h_status.data() + 4 * (size_t)i, 4 * sizeof(int32_t));
    }
    return LMH_OK;
}

extern "C" int lmh_robot_com(lmh_handle *h, const double *d_q, double *d_com, void *stream)
{
    int rc = ready(h); if (rc) return rc;
    if (!d_q || !d_com) return fail(LMH_ERR_BAD_ARG, "null device pointer");
    HIPCHK(hipSetDevice(h->device));
    lmh_launch_com(&h->P, d_q, d_com, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return LMH_OK;
}

extern "C" int lmh_robot_com_host(lmh_handle *h, const double *q, double *com)
{
    int rc = ready(h); if (rc) return rc;
    if (!q || !com) return fail(LMH_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(h->d_state, q, sizeof(double) * 30 * (size_t)h->B, hipMemcpyHostToDevice));
    rc = lmh_robot_com(h, h->d_state, h->d_out, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpy(com, h->d_out, sizeof(double) * 3 * (size_t)h->B, hipMemcpyDeviceToHost));
    return LMH_OK;
}

extern "C" int lmh_last_out_host(lmh_handle *h, double *out)
{
    if (!h || !out) return fail(LMH_ERR_BAD_ARG, "bad argument");
    std::memcpy(out, h->h_out.data(), sizeof(double) * h->h_out.size());
    return LMH_OK;
}

extern "C" int lmh_ik_host(lmh_handle *h, double *q, const double *com_target, const double *rf6, const double *lf6, double *com, int32_t *iters)
{
    int rc = ready(h); if (rc) return rc;
    if (!q || !com_target || !rf6 || !lf6) return fail(LMH_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    double *d_q = h->d_state;                                        // staging: reuse the state buffer
    double *d_com = h->d_out;
    HIPCHK(hipMemcpy(d_q, q, sizeof(double) * 30 * (size_t)h->B, hipMemcpyHostToDevice));
    rc = lmh_ik(h, d_q, com_target, rf6, lf6, h->d_status, nullptr);
    if (rc) return rc;
    rc = lmh_robot_com(h, d_q, d_com, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpy(q, d_q, sizeof(double) * 30 * (size_t)h->B, hipMemcpyDeviceToHost));
    if (com) HIPCHK(hipMemcpy(com, d_com, sizeof(double) * 3 * (size_t)h->B, hipMemcpyDeviceToHost));
    if (iters) HIPCHK(hipMemcpy(iters, h->d_status, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->d_status, 0, sizeof(int32_t) * LMH_STATUS_STRIDE * (size_t)h->B));
    return LMH_OK;
}

extern "C" int lmh_set_prev_velocity_host(lmh_handle *h, const double *v /*[B][30]*/)
{
    if (!h || !v) return fail(LMH_ERR_BAD_ARG, "bad argument");
    for (int i = 0; i < h->B; i++) std::memcpy(h->h_state.data() + (size_t)LMH_STATE_STRIDE * i + 60, v + 30 * (size_t)i, 30 * sizeof(double));
    return LMH_OK;
}

extern "C" int lmh_synchronize(lmh_handle *h, void *stream)
{
    if (!h) return fail(LMH_ERR_BAD_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int rc = LMH_OK;
    for (auto &sl : h->slot)                                         // launches that have completed (on this stream or any other) and were not looked at yet
        if (sl.used && !sl.checked && hipEventQuery(sl.done) == hipSuccess) { const int e = slot_take_error(h, sl); if (e) rc = e; }
    return rc;
}

// ---------------------------------------------------------------------------- end-of-run summary + on-disk records
extern "C" int lmh_make_summary(lmh_handle *h, const double *d_state, const double *d_out, const int32_t *d_status, double *d_summary, void *stream)
{
    if (!h || !d_state || !d_out || !d_status || !d_summary) return fail(LMH_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    lmh_launch_summary(h->B, d_state, d_out, d_status, d_summary, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return LMH_OK;
}

namespace {
#pragma pack(push, 1)
struct RecHeader {                                                   // 64 bytes, little-endian (include/lmh.h)
    char magic[8]; uint32_t version, dtype; uint64_t n_instances, n_ticks; uint32_t width, pad0; double dt, t0; uint64_t pad1;
};
#pragma pack(pop)
static_assert(sizeof(RecHeader) == 64, "record header is 64 bytes");
const char kMagicSum[8] = {'L', 'M', 'H', 'S', 'U', 'M', '1', 0}, kMagicLog[8] = {'L', 'M', 'H', 'L', 'O', 'G', '1', 0};

int write_rec(const char *path, const char *magic, const double *data, uint64_t n_inst, uint64_t n_ticks, uint32_t width, double dt, double t0)
{
    if (!path || !data) return fail(LMH_ERR_BAD_ARG, "bad argument");
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(LMH_ERR_BAD_ARG, std::string("cannot open ") + path);
    RecHeader hd;
    std::memset(&hd, 0, sizeof(hd));
    std::memcpy(hd.magic, magic, 8);
    hd.version = 1; hd.dtype = 1; hd.n_instances = n_inst; hd.n_ticks = n_ticks; hd.width = width; hd.dt = dt; hd.t0 = t0;
    const size_t count = (size_t)n_inst * width * (size_t)(n_ticks ? n_ticks : 1);
    const bool ok = std::fwrite(&hd, sizeof(hd), 1, f) == 1 && (count == 0 || std::fwrite(data, sizeof(double), count, f) == count);
    if (std::fclose(f) != 0 || !ok) return fail(LMH_ERR_BAD_ARG, std::string("short write to ") + path);
    return LMH_OK;
}

int read_rec(const char *path, const char *magic, uint32_t width, double *data, uint64_t capacity, RecHeader *hd)
{
    if (!path || !hd) return fail(LMH_ERR_BAD_ARG, "bad argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(LMH_ERR_BAD_ARG, std::string("cannot open ") + path);
    int rc = LMH_OK;
    if (std::fread(hd, sizeof(*hd), 1, f) != 1) rc = fail(LMH_ERR_BAD_ARG, "truncated header");
    else if (std::memcmp(hd->magic, magic, 8) != 0) rc = fail(LMH_ERR_BAD_ARG, "bad magic");
    else if (hd->version != 1 || hd->dtype != 1 || hd->width != width) rc = fail(LMH_ERR_BAD_ARG, "unsupported version / dtype / width");
    else {
        const bool is_log = std::memcmp(magic, kMagicLog, 8) == 0;
        const uint64_t count = hd->n_instances * width * (is_log ? hd->n_ticks : 1);
        long pos = std::ftell(f);
        std::fseek(f, 0, SEEK_END);
        const long end = std::ftell(f);
        std::fseek(f, pos, SEEK_SET);
        if ((uint64_t)(end - pos) != count * sizeof(double)) rc = fail(LMH_ERR_BAD_ARG, "payload size does not match the header");
        else if (data) {
            if (capacity < count) rc = fail(LMH_ERR_BAD_ARG, "buffer too small");
            else if (count && std::fread(data, sizeof(double), (size_t)count, f) != count) rc = fail(LMH_ERR_BAD_ARG, "short read");
        }
    }
    std::fclose(f);
    return rc;
}
}  // namespace

extern "C" int lmh_write_summary(const char *path, const double *summary, uint64_t n_instances, double dt)
{
    return write_rec(path, kMagicSum, summary, n_instances, 0, LMH_SUMMARY_WIDTH, dt, 0.0);
}
extern "C" int lmh_read_summary(const char *path, double *summary, uint64_t capacity, uint64_t *n_instances, double *dt)
{
    RecHeader hd;
    const int rc = read_rec(path, kMagicSum, LMH_SUMMARY_WIDTH, summary, capacity, &hd);
    if (rc != LMH_OK) return rc;
    if (n_instances) *n_instances = hd.n_instances;
    if (dt) *dt = hd.dt;
    return LMH_OK;
}
extern "C" int lmh_write_log(const char *path, const double *log, uint64_t n_ticks, uint64_t n_instances, double dt, double t0)
{
    if (n_ticks == 0) return fail(LMH_ERR_BAD_ARG, "a log holds at least one tick");
    return write_rec(path, kMagicLog, log, n_instances, n_ticks, 36, dt, t0);
}
extern "C" int lmh_read_log(const char *path, double *log, uint64_t capacity, uint64_t *n_ticks, uint64_t *n_instances, double *dt, double *t0)
{
    RecHeader hd;
    const int rc = read_rec(path, kMagicLog, 36, log, capacity, &hd);
    if (rc != LMH_OK) return rc;
    if (n_ticks) *n_ticks = hd.n_ticks;
    if (n_instances) *n_instances = hd.n_instances;
    if (dt) *dt = hd.dt;
    if (t0) *t0 = hd.t0;
    return LMH_OK;
}

// ---------------------------------------------------------------------------- reference generators on the device
static int alloc_refs(lmh_handle *h, int n, int n_seg)
{
    // allocate the new set first, swap it in, then free the old one: a failed hipMalloc leaves the handle on its previous references
    // (still consistent with P), never on freed pointers
    double *zx = nullptr, *zy = nullptr, *segs = nullptr;
    uint8_t *ph = nullptr;
    uint16_t *sos = nullptr;
    hipError_t e = hipMalloc(&zx, sizeof(double) * (size_t)n);
    if (e == hipSuccess) e = hipMalloc(&zy, sizeof(double) * (size_t)n);
    if (e == hipSuccess) e = hipMalloc(&ph, (size_t)n);
    if (e == hipSuccess && n_seg > 0) e = hipMalloc(&segs, sizeof(double) * LMH_SEG_STRIDE * (size_t)n_seg);
    if (e == hipSuccess && n_seg > 0) e = hipMalloc(&sos, sizeof(uint16_t) * (size_t)n);
    if (e != hipSuccess) {
        void *fresh[] = {zx, zy, ph, segs, sos};
        for (void *b : fresh) if (b) (void)hipFree(b);
        return fail(LMH_ERR_HIP, std::string("reference buffers: ") + hipGetErrorString(e));
    }
    void *old[] = {h->d_zx, h->d_zy, h->d_phase, h->d_segs, h->d_sos};
    h->d_zx = zx; h->d_zy = zy; h->d_phase = ph; h->d_segs = segs; h->d_sos = sos;
    h->n_seg = 0; h->n_samples = 0;                                 // the callers set both once the generator kernel has filled the buffers
    fill_params(h);                                                 // P never points at the freed set
    for (void *b : old) if (b) (void)hipFree(b);
    return LMH_OK;
}

extern "C" int lmh_gen_walk(lmh_handle *h, double simulation_time, int num_steps, double time_per_step, double ds_time, double step_height,
                            double settle_time, int first_support, double foot_y)
{
    if (!h) return fail(LMH_ERR_BAD_ARG, "null handle");
    if (num_steps < 1 || num_steps > LMH_GEN_MAX_STEPS) return fail(LMH_ERR_BAD_ARG, "num_steps must be in [1, 1022]");
    if (!(time_per_step > 0.0) || !(ds_time >= 0.0) || !(ds_time < time_per_step) || !(settle_time >= 0.0) || !(simulation_time > 0.0))
        return fail(LMH_ERR_BAD_ARG, "need 0 <= ds_time < time_per_step, settle_time >= 0, simulation_time > 0");
    if (first_support != LMH_PHASE_RIGHT && first_support != LMH_PHASE_LEFT) return fail(LMH_ERR_BAD_ARG, "first_support must be LMH_PHASE_RIGHT or LMH_PHASE_LEFT");
    const int n = (int)((simulation_time + 0.5) / h->mpc_dt);       // zmpGeneration.cpp:41
    if (n < 1) return fail(LMH_ERR_BAD_ARG, "no samples");
    HIPCHK(hipSetDevice(h->device));
    const int n_seg = 2 * num_steps + 2;
    int rc = alloc_refs(h, n, n_seg);
    if (rc != LMH_OK) return rc;
    LmhWalkSpec W;
    W.time_step = h->mpc_dt; W.time_per_step = time_per_step; W.ds_time = ds_time; W.step_height = step_height; W.settle_time = settle_time; W.foot_y = foot_y;
    W.n_samples = n; W.num_steps = num_steps; W.first_support = first_support; W.pad = 0;
    lmh_launch_gen_walk(&W, h->d_zx, h->d_zy, h->d_phase, h->d_segs, h->d_sos, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    h->n_samples = n; h->n_seg = n_seg;
    fill_params(h);
    return LMH_OK;
}

extern "C" int lmh_gen_jump(lmh_handle *h, double simulation_time, double stance_time, double flight_time)
{
    if (!h) return fail(LMH_ERR_BAD_ARG, "null handle");
    if (!(stance_time >= 0.0) || !(flight_time >= 0.0) || !(simulation_time > 0.0)) return fail(LMH_ERR_BAD_ARG, "times must be non-negative");
    const int n = (int)((simulation_time + 0.5) / h->mpc_dt);
    if (n < 1) return fail(LMH_ERR_BAD_ARG, "no samples");
    HIPCHK(hipSetDevice(h->device));
    int rc = alloc_refs(h, n, 0);
    if (rc != LMH_OK) return rc;
    lmh_launch_gen_jump(n, h->mpc_dt, stance_time, flight_time, h->d_zx, h->d_zy, h->d_phase, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    h->n_samples = n;
    fill_params(h);
    return LMH_OK;
}

extern "C" int lmh_num_ref_samples(const lmh_handle *h) { return h ? h->n_samples : 0; }
extern "C" int lmh_num_segments(const lmh_handle *h) { return h ? h->n_seg : 0; }

extern "C" int lmh_get_refs(lmh_handle *h, double *zx, double *zy, uint8_t *phase, double *segs, uint16_t *sos)
{
    if (!h) return fail(LMH_ERR_BAD_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    const size_t n = (size_t)h->n_samples;
    if (zx) HIPCHK(hipMemcpy(zx, h->d_zx, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (zy) HIPCHK(hipMemcpy(zy, h->d_zy, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (phase) {
        if (h->d_phase) HIPCHK(hipMemcpy(phase, h->d_phase, n, hipMemcpyDeviceToHost));
        else std::memset(phase, 0, n);
    }
    if (segs && h->n_seg > 0) HIPCHK(hipMemcpy(segs, h->d_segs, sizeof(double) * LMH_SEG_STRIDE * (size_t)h->n_seg, hipMemcpyDeviceToHost));
    if (sos && h->n_seg > 0) HIPCHK(hipMemcpy(sos, h->d_sos, sizeof(uint16_t) * n, hipMemcpyDeviceToHost));
    return LMH_OK;
}
