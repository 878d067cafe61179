/* orc_api.c -- CPU ORACLE (test infrastructure): flat, ctypes-friendly entry points over the
 * oracle structs, plus a pthread batch driver used ONLY as the timed CPU baseline leg of
 * bench.py and by tests.  Nothing here is linked into the shipped library. */
#define _POSIX_C_SOURCE 199309L
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include "lmh_oracle.h"

typedef struct {
    orc_system sys;
    orc_eval ev;
} orc_box;

void *orc_sys_create(double simulationTime, double timeStep, double timeHorizon, int do_ik)
{
    orc_box *b = (orc_box *)calloc(1, sizeof(orc_box));
    orc_system_init_offline(&b->sys, simulationTime, timeStep, timeHorizon, do_ik);
    return b;
}

/* same set-up but with caller-supplied RAW link parameters [28][13] (mass, com, inertia as in
 * createNaoParameters, i.e. before the joint-frame re-expression of Robot.cpp:14-22) */
void *orc_sys_create_model(double simulationTime, double timeStep, double timeHorizon, int do_ik,
                           const double *raw /*28*13 or NULL*/)
{
    orc_box *b = (orc_box *)calloc(1, sizeof(orc_box));
    if (!raw) { orc_system_init_offline(&b->sys, simulationTime, timeStep, timeHorizon, do_ik); return b; }
    orc_link links[ORC_NF];
    for (int i = 0; i < ORC_NF; i++) {
        links[i].mass = raw[i * 13];
        memcpy(links[i].com, raw + i * 13 + 1, 3 * sizeof(double));
        memcpy(links[i].inertia, raw + i * 13 + 4, 9 * sizeof(double));
    }
    orc_system_init_offline(&b->sys, simulationTime, timeStep, timeHorizon, 0);
    orc_robot_init(&b->sys.robot, links);
    if (do_ik) {
        double Rf[6] = {0, -0.05, 0, 0, 0, 0}, Lf[6] = {0, 0.05, 0, 0, 0, 0}, com[3] = {-0.02, 0.0, 0.26}, Qd[ORC_NQ];
        orc_ik_desired_op(&b->sys.robot, Rf, Lf, com, Qd);
        orc_ik_compute(&b->sys.robot, Qd);
    }
    orc_mpc_init(&b->sys.mpc, timeStep, timeHorizon, b->sys.robot.CoM[2]);
    return b;
}

void orc_sys_destroy(void *h) { orc_box *b = (orc_box *)h; orc_system_free(&b->sys); free(b); }

void orc_sys_nao_raw(double *raw /*28*13*/)
{
    orc_link links[ORC_NF];
    orc_nao_parameters(links);
    for (int i = 0; i < ORC_NF; i++) {
        raw[i * 13] = links[i].mass;
        memcpy(raw + i * 13 + 1, links[i].com, 3 * sizeof(double));
        memcpy(raw + i * 13 + 4, links[i].inertia, 9 * sizeof(double));
    }
}

/* joint-frame link data after Robot.cpp:14-22, [28][13] */
void orc_sys_get_links(void *h, double *out)
{
    orc_box *b = (orc_box *)h;
    for (int i = 0; i < ORC_NF; i++) {
        out[i * 13] = b->sys.robot.links[i].mass;
        memcpy(out + i * 13 + 1, b->sys.robot.links[i].com, 3 * sizeof(double));
        memcpy(out + i * 13 + 4, b->sys.robot.links[i].inertia, 9 * sizeof(double));
    }
}

double orc_sys_mass(void *h) { return ((orc_box *)h)->sys.robot.mass; }
int orc_sys_horizon(void *h) { return ((orc_box *)h)->sys.mpc.horizon; }
double orc_sys_zcom(void *h) { return ((orc_box *)h)->sys.mpc.zCom; }
int orc_sys_nzmp(void *h) { return ((orc_box *)h)->sys.ctl.n_zmp; }

void orc_sys_get_robot(void *h, double *q, double *v, double *CoM, double *comVel, double *angMom)
{
    orc_robot *r = &((orc_box *)h)->sys.robot;
    if (q) memcpy(q, r->q, sizeof(r->q));
    if (v) memcpy(v, r->v, sizeof(r->v));
    if (CoM) memcpy(CoM, r->CoM, sizeof(r->CoM));
    if (comVel) memcpy(comVel, r->comVel, sizeof(r->comVel));
    if (angMom) memcpy(angMom, r->comAngMom, sizeof(r->comAngMom));
}

/* overwrite Robot::v_ (the stale velocity the next evaluation's C/Cg/Jpqp will see) */
void orc_sys_set_prev_velocity(void *h, const double *v) { memcpy(((orc_box *)h)->sys.robot.v, v, ORC_NQ * sizeof(double)); }
void orc_sys_set_q(void *h, const double *q) { orc_robot_update_state(&((orc_box *)h)->sys.robot, q); }
void orc_sys_set_wbc_calls(void *h, int n, int faithful) { orc_box *b = (orc_box *)h; b->sys.ctl.wbc_calls_per_eval = n; b->sys.mpc.faithful_rebuild = faithful; }

void orc_sys_set_refs(void *h, int n, const double *zx, const double *zy, const unsigned char *phase)
{
    orc_controller_set_refs(&((orc_box *)h)->sys.ctl, n, zx, zy, phase);
}
void orc_sys_set_foot_coeffs(void *h, const double *rF /*3x8*/, const int *rFn, const double *lF, const int *lFn)
{
    orc_controller *c = &((orc_box *)h)->sys.ctl;
    memcpy(c->rF, rF, sizeof(c->rF)); memcpy(c->lF, lF, sizeof(c->lF));
    memcpy(c->rFn, rFn, sizeof(c->rFn)); memcpy(c->lFn, lFn, sizeof(c->lFn));
}
void orc_sys_get_foot_coeffs(void *h, double *rF, int *rFn, double *lF, int *lFn)
{
    orc_controller *c = &((orc_box *)h)->sys.ctl;
    memcpy(rF, c->rF, sizeof(c->rF)); memcpy(lF, c->lF, sizeof(c->lF));
    memcpy(rFn, c->rFn, sizeof(c->rFn)); memcpy(lFn, c->lFn, sizeof(c->lFn));
}
void orc_sys_get_zmp(void *h, double *zx, double *zy)
{
    orc_controller *c = &((orc_box *)h)->sys.ctl;
    memcpy(zx, c->zmpX, sizeof(double) * (size_t)c->n_zmp);
    memcpy(zy, c->zmpY, sizeof(double) * (size_t)c->n_zmp);
}
void orc_sys_gain_row(void *h, double *K) { orc_mpc_gain_row(&((orc_box *)h)->sys.mpc, K); }
void orc_sys_mpc_mats(void *h, double *Px /*(N+1)x2*/, double *Pu /*(N+1)^2 dense*/)
{
    orc_mpc *m = &((orc_box *)h)->sys.mpc;
    int n = m->horizon + 1;
    memcpy(Px, m->Px, sizeof(double) * 2 * (size_t)n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Pu[i * n + j] = m->Pu[i * (ORC_MAXH + 1) + j];
}

/* info[8] = {k, phase, qp_iters, qp_status, active_mask, 0,0,0} */
static void fill_info(const orc_eval *e, int *info)
{
    info[0] = e->k; info[1] = e->phase; info[2] = e->qp_iters; info[3] = e->qp_status;
    info[4] = (int)e->active_mask; info[5] = info[6] = info[7] = 0;
}

int orc_sys_eval(void *h, const double *q, const double *dq, double t,
                 double *tau, double *f, double *qpp, int *info)
{
    orc_box *b = (orc_box *)h;
    orc_stand_step(&b->sys, q, dq, t, &b->ev);
    memcpy(tau, b->ev.tau, sizeof(b->ev.tau));
    memcpy(f, b->ev.f, sizeof(b->ev.f));
    memcpy(qpp, b->ev.qpp, sizeof(b->ev.qpp));
    if (info) fill_info(&b->ev, info);
    return b->ev.qp_status;
}

/* model terms of the LAST evaluation */
void orc_sys_get_terms(void *h, double *T, double *X, double *C, double *Cg, double *M, double *AG,
                       double *AGpqp, double *Jpqp, double *J)
{
    orc_box *b = (orc_box *)h;
    if (T) memcpy(T, b->sys.robot.T, sizeof(b->sys.robot.T));
    if (X) memcpy(X, b->sys.robot.X, sizeof(b->sys.robot.X));
    if (C) memcpy(C, b->sys.dyn.C, sizeof(b->sys.dyn.C));
    if (Cg) memcpy(Cg, b->sys.dyn.Cg, sizeof(b->sys.dyn.Cg));
    if (M) memcpy(M, b->sys.dyn.M, sizeof(b->sys.dyn.M));
    if (AG) memcpy(AG, b->sys.dyn.AG, sizeof(b->sys.dyn.AG));
    if (AGpqp) memcpy(AGpqp, b->sys.dyn.AGpqp, sizeof(b->sys.dyn.AGpqp));
    if (Jpqp) memcpy(Jpqp, b->sys.dyn.Jpqp, sizeof(b->sys.dyn.Jpqp));
    if (J) memcpy(J, b->ev.JFeet, sizeof(b->ev.JFeet));
}

/* QP data of the LAST evaluation */
void orc_sys_get_qp(void *h, double *H, double *g, double *A, double *lbA, double *ubA, double *x,
                    double *qppRef, double *hGpRef, double *footAccRef, double *u0 /*2*/, double *mpcRef /*6*/)
{
    orc_box *b = (orc_box *)h;
    if (H) memcpy(H, b->ev.H, sizeof(b->ev.H));
    if (g) memcpy(g, b->ev.g, sizeof(b->ev.g));
    if (A) memcpy(A, b->ev.A, sizeof(b->ev.A));
    if (lbA) memcpy(lbA, b->ev.lbA, sizeof(b->ev.lbA));
    if (ubA) memcpy(ubA, b->ev.ubA, sizeof(b->ev.ubA));
    if (x) memcpy(x, b->ev.x, sizeof(b->ev.x));
    if (qppRef) memcpy(qppRef, b->ev.qppRef, sizeof(b->ev.qppRef));
    if (hGpRef) memcpy(hGpRef, b->ev.hGpRef, sizeof(b->ev.hGpRef));
    if (footAccRef) memcpy(footAccRef, b->ev.footAccRef, sizeof(b->ev.footAccRef));
    if (u0) { u0[0] = b->ev.u0x; u0[1] = b->ev.u0y; }
    if (mpcRef) { memcpy(mpcRef, b->sys.mpc.xRef, 3 * sizeof(double)); memcpy(mpcRef + 3, b->sys.mpc.yRef, 3 * sizeof(double)); }
}

/* closed loop: state[60] in/out, *t in/out (t += dt per tick as Clock::step does).
 * log (optional): per tick [tau(24) f(12)] of the k4-stage evaluation; klog: k of that stage;
 * comx (optional): Robot CoM x after each tick (what apps/offline prints, main.cpp:86). */
void orc_sys_rollout(void *h, double *state, double *t, double dt, int nticks,
                     double *log, int *klog, double *comx, int *info_last)
{
    orc_box *b = (orc_box *)h;
    double tt = *t;
    for (int i = 0; i < nticks; i++) {
        orc_rk4_tick(&b->sys, state, tt, dt, &b->ev);
        if (log) { memcpy(log + (size_t)i * 36, b->ev.tau, 24 * sizeof(double)); memcpy(log + (size_t)i * 36 + 24, b->ev.f, 12 * sizeof(double)); }
        if (klog) klog[i] = b->ev.k;
        if (comx) comx[i] = b->sys.robot.CoM[0];
        tt += dt;
    }
    *t = tt;
    if (info_last) fill_info(&b->ev, info_last);
}

/* ---------------- batch driver (CPU baseline): B independent instances over nthreads ---------------- */
typedef struct {
    int begin, end, nticks;
    double dt, t0, simT, horizonT, zCom;
    const double *q0;        /* shared initial posture [30] */
    double *states;          /* [B][60] in/out */
    double *prev_v;          /* [B][30] in/out (Robot::v_) or NULL */
    double *out;             /* [B][36] tau,f of last evaluation */
    int wbc_calls;
} batch_arg;

static void *batch_worker(void *p)
{
    batch_arg *a = (batch_arg *)p;
    orc_box *b = (orc_box *)orc_sys_create(a->simT, a->dt, a->horizonT, 0);
    orc_sys_set_wbc_calls(b, a->wbc_calls, a->wbc_calls > 1);
    if (a->zCom > 0) orc_mpc_init(&b->sys.mpc, a->dt, a->horizonT, a->zCom);   /* zCom of the IK posture (main.cpp:39) */
    b->sys.mpc.faithful_rebuild = a->wbc_calls > 1;
    for (int i = a->begin; i < a->end; i++) {
        double t = a->t0;
        if (a->prev_v) orc_sys_set_prev_velocity(b, a->prev_v + (size_t)i * 30);
        else memset(b->sys.robot.v, 0, sizeof(b->sys.robot.v));
        orc_sys_rollout(b, a->states + (size_t)i * 60, &t, a->dt, a->nticks, NULL, NULL, NULL, NULL);
        if (a->prev_v) memcpy(a->prev_v + (size_t)i * 30, b->sys.robot.v, 30 * sizeof(double));
        if (a->out) { memcpy(a->out + (size_t)i * 36, b->ev.tau, 24 * sizeof(double)); memcpy(a->out + (size_t)i * 36 + 24, b->ev.f, 12 * sizeof(double)); }
    }
    orc_sys_destroy(b);
    return NULL;
}

/* returns wall seconds of the rollout section */
double orc_batch_rollout(int B, double *states, double *prev_v, double *out, double t0, double dt, int nticks,
                         double simT, double horizonT, double zCom, int nthreads, int wbc_calls)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > B) nthreads = B;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    batch_arg *args = (batch_arg *)malloc(sizeof(batch_arg) * (size_t)nthreads);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int k = 0; k < nthreads; k++) {
        args[k].begin = (int)((long)B * k / nthreads);
        args[k].end = (int)((long)B * (k + 1) / nthreads);
        args[k].nticks = nticks; args[k].dt = dt; args[k].t0 = t0; args[k].simT = simT; args[k].horizonT = horizonT; args[k].zCom = zCom;
        args[k].states = states; args[k].prev_v = prev_v; args[k].out = out; args[k].q0 = NULL; args[k].wbc_calls = wbc_calls;
        pthread_create(&th[k], NULL, batch_worker, &args[k]);
    }
    for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    free(th); free(args);
    return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
}

/* re-initialise the LIPM with a caller-chosen height (Mpc3dLip ctor argument, main.cpp:39) */
void orc_sys_set_zcom(void *h, double zcom)
{
    orc_box *b = (orc_box *)h;
    int faithful = b->sys.mpc.faithful_rebuild;
    double zs = b->sys.mpc.zmp_xscale;
    orc_mpc_init(&b->sys.mpc, b->sys.mpc.dt, b->sys.mpc.timeHorizon, zcom);
    b->sys.mpc.faithful_rebuild = faithful;
    b->sys.mpc.zmp_xscale = zs;
}

/* walking extension: piecewise foot polynomials + per-instance x scale (see lmh_oracle.h) */
void orc_sys_set_segments(void *h, int n_seg, const double *segs, const unsigned short *seg_of_sample, double xscale)
{
    orc_box *b = (orc_box *)h;
    orc_controller_set_segments(&b->sys.ctl, n_seg, segs, seg_of_sample, xscale);
    b->sys.mpc.zmp_xscale = xscale;
}

/* controller.hpp:80-124 literals replaced by caller-chosen values (the reference has no setter: they are
 * in-class initialisers; the C ABI exposes them through lmh_config, so the checker must follow).
 * g[15] = mu | KpJoints KdJoints KpMom KdMom KpFeet KdFeet | wCoML wCoMK wBasePos wBaseAng wJoints wForce wFoot | epsCoeff.
 * The friction basis (controller.cpp:33-36) is rebuilt from mu. */
void orc_sys_set_gains(void *h, const double *g)
{
    orc_controller *c = &((orc_box *)h)->sys.ctl;
    orc_gains *G = &c->gains;
    G->mu = g[0];
    G->KpJoints = g[1]; G->KdJoints = g[2]; G->KpMom = g[3]; G->KdMom = g[4]; G->KpFeet = g[5]; G->KdFeet = g[6];
    G->wCoML = g[7]; G->wCoMK = g[8]; G->wBasePos = g[9]; G->wBaseAng = g[10]; G->wJoints = g[11]; G->wForce = g[12]; G->wFoot = g[13];
    G->epsCoeff = g[14];
    const double mu = G->mu;
    const double fm[12] = {mu, 0, -mu, 0, 0, mu, 0, -mu, 1, 1, 1, 1};
    memcpy(c->friction, fm, sizeof(fm));
}
void orc_sys_get_gains(void *h, double *g)
{
    const orc_gains *G = &((orc_box *)h)->sys.ctl.gains;
    g[0] = G->mu; g[1] = G->KpJoints; g[2] = G->KdJoints; g[3] = G->KpMom; g[4] = G->KdMom; g[5] = G->KpFeet; g[6] = G->KdFeet;
    g[7] = G->wCoML; g[8] = G->wCoMK; g[9] = G->wBasePos; g[10] = G->wBaseAng; g[11] = G->wJoints; g[12] = G->wForce; g[13] = G->wFoot;
    g[14] = G->epsCoeff;
}

/* ---------------- batch driver, general form (CPU baseline of the walking / randomised workloads) ----------------
 * Same static partition as orc_batch_rollout, but every robot may carry its own reference scale, LIPM height and
 * raw link table, and the reference set (ZMP samples, support phase, swing segments) is caller supplied. */
typedef struct {
    int B, nticks, wbc_calls, n_zmp, n_seg;
    double dt, t0, horizonT;
    double mpc_dt;                   /* Mpc3dLip / ZMP sample time (apps/offline/main.cpp:21,39); the Clock's step is dt (:18) */
    double *states;                  /* [B][60] in/out */
    double *out;                     /* [B][36] or NULL */
    const double *zx, *zy;           /* [n_zmp] */
    const unsigned char *phase;      /* [n_zmp] or NULL */
    const double *segs;              /* [n_seg][52] or NULL */
    const unsigned short *sos;       /* [n_zmp] or NULL */
    const double *xscale;            /* [B] or NULL */
    const double *zcom;              /* [B] or [1] (n_zcom) */
    int n_zcom;
    const double *raw;               /* [B][28][13] or NULL (nominal) */
} batch_ex_shared;
typedef struct { const batch_ex_shared *s; int begin, end; } batch_ex_arg;

static void *batch_ex_worker(void *p)
{
    const batch_ex_arg *a = (const batch_ex_arg *)p;
    const batch_ex_shared *s = a->s;
    orc_box *b = NULL;
    for (int i = a->begin; i < a->end; i++) {
        if (!b || s->raw) {                                       /* a randomised model needs its own Robot (Robot.cpp:14-22) */
            if (b) orc_sys_destroy(b);
            b = (orc_box *)orc_sys_create_model(1.0, s->mpc_dt, s->horizonT, 0, s->raw ? s->raw + (size_t)i * ORC_NF * 13 : NULL);
            orc_sys_set_wbc_calls(b, s->wbc_calls, s->wbc_calls > 1);
            orc_controller_set_refs(&b->sys.ctl, s->n_zmp, s->zx, s->zy, s->phase);
        }
        const double z = s->zcom[(s->n_zcom > 1) ? i : 0];
        orc_sys_set_zcom(b, z);
        const double xs = s->xscale ? s->xscale[i] : 1.0;
        if (s->n_seg > 0) orc_sys_set_segments(b, s->n_seg, s->segs, s->sos, xs);
        else b->sys.mpc.zmp_xscale = xs;
        memset(b->sys.robot.v, 0, sizeof(b->sys.robot.v));       /* Robot::v_ after construction */
        double t = s->t0;
        orc_sys_rollout(b, s->states + (size_t)i * 60, &t, s->dt, s->nticks, NULL, NULL, NULL, NULL);
        if (s->out) { memcpy(s->out + (size_t)i * 36, b->ev.tau, 24 * sizeof(double)); memcpy(s->out + (size_t)i * 36 + 24, b->ev.f, 12 * sizeof(double)); }
    }
    if (b) orc_sys_destroy(b);
    return NULL;
}

double orc_batch_rollout_ex(int B, double *states, double *out, double t0, double dt, int nticks, double horizonT,
                            int n_zmp, const double *zx, const double *zy, const unsigned char *phase,
                            int n_seg, const double *segs, const unsigned short *sos, const double *xscale,
                            int n_zcom, const double *zcom, const double *raw, int nthreads, int wbc_calls, double mpc_dt /* <= 0: dt */)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > B) nthreads = B;
    batch_ex_shared s = {B, nticks, wbc_calls, n_zmp, n_seg, dt, t0, horizonT, (mpc_dt > 0.0) ? mpc_dt : dt, states, out, zx, zy, phase, segs, sos, xscale, zcom, n_zcom, raw};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    batch_ex_arg *args = (batch_ex_arg *)malloc(sizeof(batch_ex_arg) * (size_t)nthreads);
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int k = 0; k < nthreads; k++) {
        args[k].s = &s;
        args[k].begin = (int)((long)B * k / nthreads);
        args[k].end = (int)((long)B * (k + 1) / nthreads);
        pthread_create(&th[k], NULL, batch_ex_worker, &args[k]);
    }
    for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    free(th); free(args);
    return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
}

/* plant switch + contact parameters (see lmh_oracle.h orc_controller.plant) */
void orc_sys_set_plant(void *h, int on, double k, double d, double dt, double mu)
{
    orc_controller *c = &((orc_box *)h)->sys.ctl;
    c->plant = on; c->contact_k = k; c->contact_d = d; c->contact_dt = dt; c->contact_mu = mu;
}
/* contact wrench [12] and vertex forces [8][3] of the compliant ground at the robot's CURRENT state (after the last evaluation) */
void orc_sys_contact(void *h, double *w, double *vf)
{
    orc_box *b = (orc_box *)h;
    orc_contact_wrench(&b->sys, b->ev.JFeet, w, (double (*)[3])vf);
}
