/*
 * lmh_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the per-tick control hot path of Ema158/linearMpcHumanoid
 * (Robot -> Dynamics -> Kinematics -> Mpc3dLip -> Controller::WBC -> rk4Step), written
 * from the reference's behaviour with every function citing the reference file:line it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link or call anything in this directory; the shipped HIP path never does.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or published numbers
 * and cannot be compiled here (needs Eigen3 + qpOASES + MuJoCo + GLFW, none present,
 * no network).  The two QPs it hands to qpOASES (version unpinned, not vendored) are
 * strictly convex, so their minimisers are unique; this oracle reproduces them with
 *   - a closed-form Cholesky solve for the unconstrained LIPM preview QP, and
 *   - a dense Goldfarb-Idnani dual active-set solver for the 74-variable WBC QP,
 * both exact up to fp64 round-off.  Matrix-product association inside Eigen is not
 * reproduced bit-for-bit (it is Eigen-version dependent); all results are fp64.
 *
 * Conventions (reference include/linearMpcHumanoid/controller/controller.hpp:20-31):
 *   q = [p_base(3) world | rpy(3) | qJ(24)],  v = [v_lin(3) | omega(3) | qdJ(24)]
 *   all 6-D spatial quantities are [angular; linear]; matrices are ROW-major here.
 */
#ifndef LMH_ORACLE_H
#define LMH_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NQ 30       /* Robot.hpp:11  NUM_JOINTS        */
#define ORC_NJ 24       /* Robot.hpp:12  NUM_ACTUAL_JOINTS */
#define ORC_NF 28       /* Robot.hpp:13  NUM_FRAMES        */
#define ORC_NV 74       /* controller.hpp:94 numDesVariables_ */
#define ORC_NC 50       /* controller.hpp:89 numConstraints_  */
#define ORC_MAXH 64     /* max preview horizon N (N+1 <= 65)  */
#define ORC_INFTY 1.0e20 /* qpOASES::INFTY */
#define ORC_SEG_STRIDE 52

/* linkInertia.hpp:4-9 */
typedef struct {
    double mass;
    double com[3];
    double inertia[9];
} orc_link;

/* Robot.hpp:74-92 (state that survives between calls) */
typedef struct {
    orc_link links[ORC_NF];
    double mass;
    double q[ORC_NQ];
    double v[ORC_NQ];            /* Robot::v_ : STALE until updateVelocityState */
    double T[ORC_NF][16];        /* world transforms, 4x4 row-major */
    double X[ORC_NF][36];        /* parent-relative Pluecker transforms, 6x6 row-major */
    double CoM[3], comVel[3], comAngMom[3];
} orc_robot;

/* Dynamics.hpp:45-55 */
typedef struct {
    double I[ORC_NF][36];
    double C[ORC_NQ], Cg[ORC_NQ];
    double M[ORC_NQ * ORC_NQ];
    double AG[6 * ORC_NQ];
    double AGpqp[6];
    double Jpqp[12];
} orc_dynamics;

/* mpcLinearPendulum.hpp:34-53 */
typedef struct {
    double dt, timeHorizon, zCom, gravity, alpha, beta;
    int horizon;                                   /* N */
    double A[4], B[2], Cm[2], D;
    double Px[(ORC_MAXH + 1) * 2];
    double Pu[(ORC_MAXH + 1) * (ORC_MAXH + 1)];
    double Lh[(ORC_MAXH + 1) * (ORC_MAXH + 1)];    /* Cholesky factor of H (cached) */
    int have_factor;
    int faithful_rebuild;       /* 1: rebuild H = aI + b Pu'Pu every call like :89-90 */
    double xRef[3], yRef[3];
    int last_k;
    double zmp_xscale;          /* extension: per-instance scale of the ZMP x samples (1 = reference) */
    double last_gx[ORC_MAXH + 1], last_gy[ORC_MAXH + 1];
} orc_mpc;

/* controller.hpp:80-124 literals */
typedef struct {
    double mu;
    double KpJoints, KdJoints, KpMom, KdMom, KpFeet, KdFeet;
    double wCoML, wCoMK, wBasePos, wBaseAng, wJoints, wForce, wFoot;
    double epsCoeff;            /* controller.cpp:117 */
} orc_gains;

typedef struct {
    orc_gains gains;
    double friction[12];        /* 3x4 friction basis, controller.cpp:33-36 */
    double footVertices[4][3];  /* Robot.cpp:38-42 */
    double Rf_q0[9];            /* Robot.cpp:28-31 */
    /* references (copied like Controller copies ZMP and coefficient vectors) */
    int n_zmp;
    double *zmpX, *zmpY;        /* owned */
    unsigned char *phase;       /* owned, per-sample support phase (0=Double,1=Right,2=Left,3=Flight); NULL = all Double */
    double rF[3][8], lF[3][8];  /* foot polynomial coefficients, ascending powers */
    int rFn[3], lFn[3];
    /* build-defined walking extension: piecewise foot polynomials selected by the preview index k.
     * segment record (ORC_SEG_STRIDE doubles): t0 | rF[3][8] | lF[3][8] | pad(3); polynomials are
     * evaluated at (t - t0); x-axis coefficients and the ZMP x samples are multiplied by xscale. */
    int n_seg;
    double *segs;               /* owned, [n_seg][ORC_SEG_STRIDE] */
    unsigned short *seg_of_sample; /* owned, [n_zmp] */
    double xscale;
    int wbc_calls_per_eval;     /* 1 (result-neutral default) or 2 (apps/offline/main.cpp:103-105) */
    /* build-defined plant: 0 = the reference's loop, which integrates the controller's own acceleration
     * (apps/offline/main.cpp:118-121); 1 = forward dynamics M qdd = S'tau + J'w_contact - C(q, v) driven by the torques the WBC returns,
     * with a spring-damper contact at the four vertices of each sole (Robot.cpp:38-42) against the plane z = 0.  C(q, v) is evaluated at
     * the velocity of the state being integrated (a second Dynamics::computeC pass after controller.cpp:59), NOT the controller's own C,
     * which the reference evaluates with the previous call's velocity (controller.cpp:56 before :59) */
    int plant;
    double contact_k, contact_d, contact_dt, contact_mu;   /* normal stiffness [N/m], normal / tangential damping [N s/m], friction */
} orc_controller;

/* everything one evaluation produces (unit-parity taps) */
typedef struct {
    double tau[ORC_NJ];
    double f[12];
    double qpp[ORC_NQ];
    double x[ORC_NV];           /* raw QP primal */
    double H[ORC_NV * ORC_NV];
    double g[ORC_NV];
    double A[ORC_NC * ORC_NV];
    double lbA[ORC_NC], ubA[ORC_NC];
    double qppRef[ORC_NQ], hGpRef[6], footAccRef[12];
    double JFeet[12 * ORC_NQ];
    double u0x, u0y;
    int k;                      /* preview index int(t/dt), mpcLinearPendulum.cpp:92 */
    int phase;                  /* support phase used */
    int qp_iters;
    int qp_status;              /* 0 ok */
    unsigned int active_mask;   /* bit j set <=> coefficient c_j sits on its bound */
    double genForceBaseResidual[6]; /* rows 0..5 of M qdd + C - J'f (controller.cpp:138) */
} orc_eval;

typedef struct {
    orc_robot robot;
    orc_dynamics dyn;
    orc_mpc mpc;
    orc_controller ctl;
} orc_system;

/* ---- model / robot (Robot.cpp, robotParameters.cpp, generalizedFunctions.cpp) ---- */
void orc_nao_parameters(orc_link links[ORC_NF]);                      /* robotParameters.cpp:8-229 */
void orc_robot_init(orc_robot *r, const orc_link *raw_links /*NULL=nominal*/); /* Robot.cpp:5-43 */
void orc_initial_configuration(double q[ORC_NQ]);                     /* Robot.cpp:242-251 */
void orc_desired_posture(double q[ORC_NQ]);                           /* Robot.cpp:253-262 */
void orc_robot_update_state(orc_robot *r, const double *q);           /* Robot.cpp:264-269 */
void orc_robot_update_velocity(orc_robot *r, const double *v, const double *AG); /* Robot.cpp:271-274 */
extern const int orc_parent[ORC_NF];                                  /* Robot.cpp:165 */
extern const int orc_act[ORC_NF];                                     /* Robot.cpp:172 */

/* ---- dynamics / kinematics ---- */
void orc_dynamics_compute_all(orc_dynamics *d, const orc_robot *r);   /* Dynamics.cpp:202-216 */
void orc_dynamics_bias_now(const orc_dynamics *d, const orc_robot *r, double C[ORC_NQ]); /* Dynamics.cpp:29-60 at the CURRENT Robot::v_ (plant only) */
void orc_feet_jacobian(const orc_robot *r, double *JFeet /*12x30*/);  /* invKinematics.cpp:72-149 */
int  orc_ik_compute(orc_robot *r, const double *desOp);               /* invKinematics.cpp:27-52 */
void orc_ik_desired_op(const orc_robot *r, const double *Rf, const double *Lf,
                       const double *com, double *Qd);                /* invKinematics.cpp:11-25 */

/* ---- MPC ---- */
void orc_mpc_init(orc_mpc *m, double dt, double timeHorizon, double zCom); /* mpcLinearPendulum.cpp:10-76 */
void orc_mpc_compute(orc_mpc *m, const double pos[2], const double vel[2],
                     const double *zmpX, const double *zmpY, double t); /* :78-109 */
void orc_mpc_gain_row(const orc_mpc *m, double *K /*N+1*/);           /* K = beta e0' H^-1 Pu' */

/* ---- trajectories ---- */
int  orc_zmp_stance(double simulationTime, double timeStep, int supportFoot,
                    double **zx, double **zy);                        /* zmpGeneration.cpp:39-60 */
void orc_foot_coeff_trajectory(const double cur[3], const double des[3], double stepHeight,
                               double T, double coeff[3][8], int ncoef[3]); /* footRefTrajectory.cpp:4-47 */
int  orc_find_poly_coeff(int nPos, const double *pos, int nVel, const double *vel,
                         int nAcc, const double *acc, double *coeff); /* generalizedFunctions.cpp:103-163 */

/* ---- QP ---- */
int orc_qp_solve(int n, const double *H, const double *g, int m, const double *A,
                 const double *lbA, const double *ubA, double *x, int *iters,
                 unsigned char *active /*m, optional*/);

/* ---- controller ---- */
void orc_controller_init(orc_controller *c);                          /* controller.cpp:5-46 */
void orc_controller_set_refs(orc_controller *c, int n, const double *zx, const double *zy,
                             const unsigned char *phase);
void orc_controller_free(orc_controller *c);
void orc_controller_set_segments(orc_controller *c, int n_seg, const double *segs, const unsigned short *seg_of_sample, double xscale);
void orc_stand_step(orc_system *s, const double *q, const double *dq, double t, orc_eval *out); /* controller.cpp:48-79 */

/* ---- closed loop (apps/offline/main.cpp:66-122, rk4.hpp:5-18, Clock.hpp:11) ---- */
void orc_plant_derivative(orc_system *s, const double *state, double t, double *xdot, orc_eval *out);
void orc_contact_wrench(const orc_system *s, const double *JFeet, double w[12], double vertex_force[8][3]);  /* spring-damper plant contact */
void orc_rk4_tick(orc_system *s, double *state /*60*/, double t, double dt, orc_eval *last /*k4-stage eval*/);

/* convenience: the whole apps/offline set-up (main.cpp:12-58) */
void orc_system_init_offline(orc_system *s, double simulationTime, double timeStep,
                             double timeHorizon, int do_ik);
void orc_system_free(orc_system *s);
unsigned long orc_sizeof_system(void);
unsigned long orc_sizeof_eval(void);

#ifdef __cplusplus
}
#endif
#endif
