/*
 * lmh.h -- C ABI of the MI355X-native batched NAO whole-body controller
 *          (LIPM preview MPC -> whole-body QP -> rigid-body terms -> torques).
 *
 * This is the drop-in boundary for the per-tick hot path of Ema158/linearMpcHumanoid.
 * The reference has no FFI/plugin layer: its boundary is the C++ call surface used by
 * apps/offline/main.cpp.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root).  The C++ classes with the reference's
 * own names (Robot, Controller, Mpc3dLip, ZMP, ...) in
 * linearmpchumanoid_amd/csrc/shim/linearMpcHumanoid/ are thin wrappers over this ABI.
 *
 * All compute runs in hand-written HIP kernels (gfx950).  There is NO CPU fallback:
 * every call fails with LMH_ERR_NO_DEVICE if no HIP device is usable.
 *
 * Layouts (fp64, instance-major, contiguous):
 *   state  [B][LMH_STATE_STRIDE]  : q(30) | v(30) | v_prev(30) | t | pad(5)
 *       q = [p_base(3) world, rpy(3), qJ(24)], v = [v_lin(3), omega(3), qdJ(24)]
 *       (include/linearMpcHumanoid/controller/controller.hpp:20-31).
 *       v_prev is Robot::v_ as the previous Controller::standStep left it: the reference
 *       evaluates C, Cg and Jdot*qdot BEFORE it stores the new velocity
 *       (src/controller.cpp:56 vs :59), so those terms see the previous call's velocity.
 *   out    [B][LMH_OUT_STRIDE]    : tau(24) | f(12: n_R f_R n_L f_L) | qdd(30) | CoM(3) | comVel(3) |
 *                                   xRef(3) | yRef(3) | pad(2)
 *       (WBCOutput, controller.hpp:43-48; Robot::getCoM/getComVel; Mpc3dLip::getXRef/getYRef)
 *   status [B][LMH_STATUS_STRIDE] int32 : k | qp_iterations | flags | active_mask
 *       k = int(t/mpc_dt) of the last evaluation (src/mpcLinearPendulum.cpp:92), bit-exact.
 */
#ifndef LMH_H
#define LMH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMH_NQ 30
#define LMH_NJ 24
#define LMH_NFRAMES 28
#define LMH_STATE_STRIDE 96
#define LMH_OUT_STRIDE 80
#define LMH_STATUS_STRIDE 4
#define LMH_LINK_STRIDE 13        /* mass | com(3) | inertia(9 row-major), linkInertia.hpp:4-9 */
#define LMH_MAX_HORIZON 64
#define LMH_DEBUG_STRIDE 4096
#define LMH_SEG_STRIDE 52

/* status flags */
#define LMH_FLAG_QP_MAXITER 1     /* active-set iteration cap hit (reference: "QP failed", controller.cpp:472-476) */
#define LMH_FLAG_NONFINITE 2      /* NaN/Inf in the solution (reference aborts, controller.cpp:448-466) */
#define LMH_FLAG_ZMP_RANGE 4      /* preview window [k, k+N] left the reference arrays */
#define LMH_FLAG_NOT_SPD 8        /* a Cholesky pivot was not positive */
#define LMH_FLAG_UNFINISHED 32    /* lmh_rollout only: a wait of the kernel's work queue ran out and this robot did not get all its ticks (its state / out
                                   * records are those of the last chunk it completed); the call reports LMH_ERR_UNFINISHED, see lmh_rollout */
#define LMH_FLAG_QP_FP64_ROUTE 16 /* LMH_PRECISION_FP32 only, informational: a contact-force solve of this instance met a rank-deficient free set
                                   * (or the Lawson-Hanson pass) and went the fp64 general route -- that system (cond ~1e13) has no fp32 form */

/* support phase per preview sample (build-defined extension; reference: Task.hpp:9-13 SupportFoot) */
#define LMH_PHASE_DOUBLE 0
#define LMH_PHASE_RIGHT 1         /* right foot in support, left foot carries no force */
#define LMH_PHASE_LEFT 2
#define LMH_PHASE_FLIGHT 3

/* arithmetic of the model-term phases (BASELINE config 5 tolerance sweep; build-defined, the reference is fp64 only) */
#define LMH_PRECISION_FP64 0
#define LMH_PRECISION_MIXED 1
#define LMH_PRECISION_FP32 2      /* model terms AND the whole-body QP in fp32 (one fp64 residual refinement of the 12 x 12 contact solve) */
#define LMH_SUMMARY_WIDTH 16      /* end-of-run summary record (doubles per instance), see lmh_make_summary */

enum {
    LMH_OK = 0,
    LMH_ERR_NO_DEVICE = -1,
    LMH_ERR_BAD_ARG = -2,
    LMH_ERR_HIP = -3,
    LMH_ERR_NOT_READY = -4,
    LMH_ERR_UNFINISHED = -5       /* an earlier lmh_rollout on this handle left robots part-way (LMH_FLAG_UNFINISHED in their status records) */
};

/* Literals of the reference, gathered in one record (the reference has no config layer):
 * include/linearMpcHumanoid/controller/controller.hpp:80-124, mpcLinearPendulum.hpp:43-49,
 * src/controller.cpp:117, apps/offline/main.cpp:13-14,38-39. */
typedef struct lmh_config {
    double dt;             /* control step: Clock(timeStep, ...) of the caller (apps/offline/main.cpp:18), the RK4 step of lmh_rollout;
                              also the MPC sample time when mpc_dt == 0 (the reference's own app passes the same value to both)   */
    double time_horizon;   /* N = (int)(time_horizon / mpc_dt)  (mpcLinearPendulum.cpp:43)                                       */
    double z_com;          /* LIPM height: Mpc3dLip ctor argument (main.cpp:39)              */
    double gravity, alpha, beta;
    double mu;
    double kp_joints, kd_joints, kp_mom, kd_mom, kp_feet, kd_feet;
    double w_com_lin, w_com_ang, w_base_pos, w_base_ang, w_joints, w_force, w_foot;
    double eps_coeff;
    int32_t warm_start;    /* 1: start the active set from the previous evaluation's (same minimiser) */
    int32_t max_qp_iters;
    int32_t precision;     /* LMH_PRECISION_FP64 (reference arithmetic) | LMH_PRECISION_MIXED: model terms (kinematics, C, M, J) in
                              fp32 arithmetic, references + QP in fp64 | LMH_PRECISION_FP32: the QP too (Woodbury core, Schur
                              complement, push-through contact solves with one fp64 residual refinement; a rank-deficient contact set
                              falls back to the fp64 general route and raises LMH_FLAG_QP_FP64_ROUTE).  References, RK4 state and
                              k = int(t/dt) are fp64 in every mode */
    int32_t bpp_rounds;    /* block-principal-pivoting rounds of the contact-force QP before the Lawson-Hanson pass takes over:
                              0 = default (10); n > 0 = cap at n rounds; < 0 = skip block pivoting, solve by Lawson-Hanson from the
                              empty set (diagnostic: exercises the finite fall-back) */
    /* Build-defined plant.  plant = 0: the reference's closed loop, which integrates the controller's own acceleration
     * and throws the torques away (apps/offline/main.cpp:118-121).  plant = 1: the RK4 derivative is the forward dynamics
     * M qdd = S'tau + J'w_contact - C driven by the torques the WBC returns, with a spring-damper contact at the four vertices of each sole
     * (Robot.cpp:38-42) against the plane z = 0: normal force max(0, k d - c zdot) on penetration d, tangential force -c_t (xdot, ydot)
     * scaled back onto the friction disc mu f_n.  M, C, J are the terms the controller evaluated in the same call.  out.qdd then carries the
     * plant's acceleration.  (The reference's intended route, MuJoCo feedback, is commented out at apps/mujoco/main.cpp:115-122.) */
    int32_t plant;
    int32_t reserved;
    double contact_k;      /* normal stiffness per vertex [N/m]; the light distal links bound it for explicit RK4 (ankle inertia 1.4e-5 kg m^2) */
    double contact_d;      /* normal damping per vertex [N s/m]  */
    double contact_dt;     /* tangential damping per vertex [N s/m] */
    double contact_mu;     /* friction coefficient of the plant's ground */
    /* MPC sample time: the `dt` argument of Mpc3dLip(dt, timeHorizon, zCom) (apps/offline/main.cpp:39, mpcLinearPendulum.hpp:10-13) and the
     * `timeStep` of ZMP(task, simulationTime, timeStep, supportFoot) (main.cpp:21) -- the reference's caller picks them independently of the
     * Clock's step (main.cpp:18).  It sets k = int(t / mpc_dt) on the float-accumulated clock (mpcLinearPendulum.cpp:92), the LIPM A, B
     * (:45-47), the horizon N and the sample period of every reference array (ZMP x/y, support phase, segment index).
     * 0 (the default) = dt: one value for both, as apps/offline/main.cpp passes.  E.g. dt = 1e-3, mpc_dt = 1e-2, time_horizon = 0.32:
     * 1 kHz control with a 32 x 10 ms preview. */
    double mpc_dt;
} lmh_config;

typedef struct lmh_handle lmh_handle;

/* fills the reference literals; dt = 0.01, mpc_dt = 0 (= dt), time_horizon = 0.5, z_com = 0.26 */
void lmh_config_default(lmh_config *cfg);
const char *lmh_last_error(void);
int lmh_device_count(void);

/* replaces: Robot::Robot + Mpc3dLip::Mpc3dLip + Controller::Controller
 * (src/Robot.cpp:5-43, src/mpcLinearPendulum.cpp:10-76, src/controller.cpp:5-46) for
 * n_instances robots on HIP device `device`.  Loads the nominal NAO model and a constant
 * (stance) reference set covering `default_ref_samples` samples. */
int lmh_create(const lmh_config *cfg, int n_instances, int device, lmh_handle **out);
int lmh_destroy(lmh_handle *h);
int lmh_num_instances(const lmh_handle *h);
int lmh_horizon(const lmh_handle *h);

/* replaces: createNaoParameters (src/robotParameters.cpp:8-229) + the joint-frame
 * re-expression of Robot::Robot (src/Robot.cpp:14-22).  raw_links: HOST pointer,
 * [n_models][28][13] in the Aldebaran (world-aligned at q=0) convention; n_models is 1
 * (shared) or n_instances (domain randomisation).  NULL restores the nominal table. */
int lmh_set_model(lmh_handle *h, const double *raw_links, int n_models);
/* total mass per model, HOST out [n_models] (Robot::getMass) */
int lmh_get_mass(lmh_handle *h, double *mass);
/* nominal raw table (HOST out [28][13]) */
void lmh_nominal_links(double *raw_links);

/* replaces: ZMP::getZmpXRef/getZmpYRef arrays copied into Controller (src/zmpGeneration.cpp:39-60,
 * src/controller.cpp:14) + the support-phase extension.  HOST pointers, n_samples each;
 * phase may be NULL (all double support). */
int lmh_set_refs(lmh_handle *h, const double *zmp_x, const double *zmp_y, const uint8_t *phase, int n_samples);
/* replaces: ZMP::stanceZMP (src/zmpGeneration.cpp:39-60) with timeStep = mpc_dt; support_foot: 0 Right,1 Left,2 Double */
int lmh_set_refs_stance(lmh_handle *h, double simulation_time, int support_foot);
/* replaces: footCoeffTrajectory output copied into Controller (src/footRefTrajectory.cpp:4-47,
 * src/controller.cpp:15-16).  coeff: HOST [3][8] ascending powers, n: [3] counts. */
int lmh_set_foot_coeffs(lmh_handle *h, const double *r_coeff, const int32_t *r_n, const double *l_coeff, const int32_t *l_n);
/* Build-defined walking extension (the reference declares ZMP::walkZMP, zmpGeneration.hpp:22, but never
 * defines it; footCoeffTrajectory produces one polynomial set per step).  Piecewise foot references:
 * segment record = LMH_SEG_STRIDE doubles: t0 | rF[3][8] | lF[3][8] | pad(3), ascending powers, evaluated
 * at (t - t0); seg_of_sample[k] selects the segment from the preview index k.  HOST pointers;
 * n_seg = 0 restores the single polynomial set of lmh_set_foot_coeffs. */
int lmh_set_segments(lmh_handle *h, const double *segs, int n_seg, const uint16_t *seg_of_sample, int n_samples);
/* Reference generators ON THE DEVICE (no host arrays are uploaded): the same plans as the host statement in
 * linearmpchumanoid_amd/trajectories.py.  lmh_gen_walk: ZMP(Task, numSteps, timePerStep, simulationTime) as the reference declares it
 * (zmpGeneration.hpp:15-19; walkZMP is never defined there) + one footCoeffTrajectory polynomial set per step
 * (footRefTrajectory.cpp:4-47, in closed form): settle_time of stance, then num_steps x [ds_time double support | single support],
 * first_support = LMH_PHASE_RIGHT or LMH_PHASE_LEFT, feet at y = -/+ foot_y; x in units of the step length (lmh_set_xscale).
 * Fills the ZMP / phase samples ((int)((simulation_time + 0.5) / mpc_dt) of them, as ZMP::stanceZMP counts; sample k <-> t = k mpc_dt)
 * and 2 num_steps + 2 segments.
 * lmh_gen_jump: stance references with LMH_PHASE_FLIGHT in [stance_time, stance_time + flight_time) (BASELINE config 5). */
int lmh_gen_walk(lmh_handle *h, double simulation_time, int num_steps, double time_per_step, double ds_time, double step_height,
                 double settle_time, int first_support, double foot_y);
int lmh_gen_jump(lmh_handle *h, double simulation_time, double stance_time, double flight_time);
/* read the current reference set back (HOST out; any pointer may be NULL): n_samples doubles / bytes / uint16, n_seg x LMH_SEG_STRIDE doubles */
int lmh_num_ref_samples(const lmh_handle *h);
int lmh_num_segments(const lmh_handle *h);
int lmh_get_refs(lmh_handle *h, double *zmp_x, double *zmp_y, uint8_t *phase, double *segs, uint16_t *seg_of_sample);
/* per-instance scale of the ZMP x samples and of the x-axis foot polynomials (step length): HOST [n_instances]
 * or NULL for 1.0 */
int lmh_set_xscale(lmh_handle *h, const double *xscale, int n);
/* per-instance LIPM height (domain randomisation): HOST [n_instances]; rebuilds the gain rows */
int lmh_set_zcom(lmh_handle *h, const double *z_com, int n);
/* MPC gain row K (HOST out [N+1]) with u0 = -K (Px x_k - zmp[k:k+N+1]); instance 0 */
int lmh_get_mpc_gain(lmh_handle *h, double *K);

/* replaces: Controller::standStep + Controller::WBC (src/controller.cpp:48-154) for all
 * instances.  DEVICE pointers; d_state is read AND updated (v_prev <- v, as Robot::v_ is);
 * stream is a hipStream_t (NULL = default stream).  Asynchronous. */
int lmh_eval(lmh_handle *h, double *d_state, double *d_out, int32_t *d_status, void *stream);
/* same, additionally dumping intermediate terms for unit parity (DEVICE [B][LMH_DEBUG_STRIDE]) */
int lmh_eval_debug(lmh_handle *h, double *d_state, double *d_out, int32_t *d_status, double *d_debug, void *stream);

/* replaces: the closed loop of apps/offline/main.cpp:66-122 (rk4Step, rk4.hpp:5-18, of
 * dynamics(); Clock::step, Clock.hpp:11) for n_ticks ticks, state resident on chip.
 * d_out receives the k4-stage evaluation of the last tick; d_log (optional, DEVICE
 * [n_ticks][B][36]) receives tau|f of the k4 stage of every tick.  d_status: [0] k and [3] the active set of the last
 * tick, [1] the maximum of the QP rounds and [2] the OR of the flags over ALL ticks of the call.  Asynchronous.
 * Inside the call a robot is advanced in chunks of 250 ticks by whichever resident workgroup claims it next (its record in
 * d_state / d_out / d_status is the hand-over); the result does not depend on that: lmh_rollout(.., a + b, ..) equals
 * lmh_rollout(.., a, ..) followed by lmh_rollout(.., b, ..) bit for bit (with [1], [2] merged as max / OR).
 * Host side of "asynchronous": the call only enqueues (one small parameter copy + the kernel) on `stream`, except that a handle keeps
 * EIGHT launches in flight -- the ninth lmh_rollout waits on the host until the first has completed -- and that the first eight calls
 * allocate their launch slot (hipMalloc): not capturable into a hipGraph before every slot has been used once.
 * Incomplete launches are loud: the queue's waits are bounded, and if one runs out (a workgroup stalled for minutes: preemption, a
 * debugger) the robots that did not get all their ticks carry LMH_FLAG_UNFINISHED in d_status[.][2], and lmh_synchronize -- or the next
 * lmh_rollout that reuses the launch slot -- returns LMH_ERR_UNFINISHED once (the reference prints and aborts, src/controller.cpp:448-476).
 * The launch slot is clean again afterwards; the caller decides whether to re-run those robots. */
int lmh_rollout(lmh_handle *h, double *d_state, double *d_out, int32_t *d_status, double *d_log,
                int n_ticks, void *stream);

/* replaces: Kinematics::desiredOperationalState + Kinematics::compute
 * (src/invKinematics.cpp:11-52): Newton IK to feet (0,-/+0.05,0), com target, per instance.
 * DEVICE d_q [B][30] in/out (start posture in, solution out); com_target HOST [3]. */
int lmh_ik(lmh_handle *h, double *d_q, const double *com_target, const double *rf6, const double *lf6,
           int32_t *d_iters, void *stream);

/* replaces: Robot::updateState + Robot::getCoM (src/Robot.cpp:264-269,225-238).
 * DEVICE d_q [B][30] in, d_com [B][3] out. */
int lmh_robot_com(lmh_handle *h, const double *d_q, double *d_com, void *stream);

/* host-buffer convenience used by the C++ shim (B instances, staged through internal
 * device buffers, synchronous): q/dq [B][30], t, outputs tau[B][24], f[B][12], qdd[B][30] */
int lmh_eval_host(lmh_handle *h, const double *q, const double *dq, double t,
                  double *tau, double *f, double *qdd, int32_t *status);
/* Robot::updateState + getCoM through host buffers: q [B][30] in, com [B][3] out */
int lmh_robot_com_host(lmh_handle *h, const double *q, double *com);
/* full out records of the last lmh_eval_host call: HOST [B][LMH_OUT_STRIDE] */
int lmh_last_out_host(lmh_handle *h, double *out);
/* Kinematics::compute + Robot::getCoM through host buffers: q [B][30] in/out, com [B][3] out, iters [B] out */
int lmh_ik_host(lmh_handle *h, double *q, const double *com_target, const double *rf6, const double *lf6, double *com, int32_t *iters);
/* overwrite the staged Robot::v_ (v_prev) used by the next lmh_eval_host call: HOST [B][30] */
int lmh_set_prev_velocity_host(lmh_handle *h, const double *v);
/* hipStreamSynchronize(stream), then LMH_ERR_UNFINISHED if a completed lmh_rollout of this handle reported an incomplete launch (see there) */
int lmh_synchronize(lmh_handle *h, void *stream);

/* ---- end-of-run summary and on-disk records (the reference writes nothing but stdout,
 * apps/offline/main.cpp:86, so these formats are the build's own).
 * lmh_make_summary: DEVICE in (state/out/status as lmh_rollout leaves them), DEVICE out [B][LMH_SUMMARY_WIDTH]:
 *   base pose(6) | t | max|tau| | f_z R + f_z L | f_z R | f_z L | k | qp iterations | flags | active-bound count |
 *   checksum (sum of the 60 state doubles, in index order).  This record is what the one RCCL gather moves. */
int lmh_make_summary(lmh_handle *h, const double *d_state, const double *d_out, const int32_t *d_status, double *d_summary, void *stream);
/* Files: 64-byte little-endian header { char magic[8] "LMHSUM1\0" | "LMHLOG1\0"; uint32 version = 1; uint32 dtype = 1 (f64);
 * uint64 n_instances; uint64 n_ticks (0 for a summary); uint32 width (16 | 36); uint32 0; double dt; double t0; uint64 0 }
 * followed by the raw f64 payload: summary [n][16]; log [n_ticks][n][36] = lmh_rollout's d_log copied to the host.
 * HOST pointers.  Readers return LMH_ERR_BAD_ARG on a bad magic / version / size; `capacity` is in doubles. */
int lmh_write_summary(const char *path, const double *summary, uint64_t n_instances, double dt);
int lmh_read_summary(const char *path, double *summary, uint64_t capacity, uint64_t *n_instances, double *dt);
int lmh_write_log(const char *path, const double *log, uint64_t n_ticks, uint64_t n_instances, double dt, double t0);
int lmh_read_log(const char *path, double *log, uint64_t capacity, uint64_t *n_ticks, uint64_t *n_instances, double *dt, double *t0);

#ifdef __cplusplus
}
#endif
#endif
