/* orc_controller.c -- CPU ORACLE (test infrastructure): Controller::standStep / WBC and the
 * apps/offline closed loop.  Follows reference src/controller.cpp, include/.../controller.hpp,
 * include/.../general/rk4.hpp and apps/offline/main.cpp.
 *
 * BUILD-DEFINED EXTENSION (no reference semantics): a per-sample support phase
 * (0 Double, 1 Right support, 2 Left support, 3 Flight).  The reference hard-wires
 * numReactionForces_ = 12 (controller.hpp:98, comment "12->DS 6->SS 0->noContact"); here a
 * foot that is not in support keeps its variables but its 16 coefficients are pinned to
 * zero (lbA = ubA = 0), which forces its wrench to zero through rows 6..17.  With every
 * sample in phase 0 the problem is exactly the reference's. */
#include <stdlib.h>
#include <math.h>
#include "lmh_oracle.h"
#include "orc_linalg.h"
#include "orc_internal.h"

void orc_controller_init(orc_controller *c)                      /* controller.cpp:5-46, controller.hpp:80-124 */
{
    memset(c, 0, sizeof(*c));
    orc_gains *g = &c->gains;
    g->mu = 0.7;
    g->KpJoints = 300; g->KdJoints = 34;
    g->KpMom = 10; g->KdMom = 6.32;
    g->KpFeet = 500; g->KdFeet = 44;
    g->wCoML = 4000; g->wCoMK = 0; g->wBasePos = 10; g->wBaseAng = 10;
    g->wJoints = 1; g->wForce = 1; g->wFoot = 100000;
    g->epsCoeff = 1e-8;
    const double mu = g->mu;
    const double fm[12] = {mu, 0, -mu, 0,
                           0, mu, 0, -mu,
                           1, 1, 1, 1};                          /* columns (mu,0,1),(0,mu,1),(-mu,0,1),(0,-mu,1) */
    memcpy(c->friction, fm, sizeof(fm));
    const double fv[4][3] = {{0.1, 0.025, 0}, {0.1, -0.025, 0}, {-0.05, 0.025, 0}, {-0.05, -0.025, 0}}; /* Robot.cpp:38-42 */
    memcpy(c->footVertices, fv, sizeof(fv));
    const double rf[9] = {0, 0, 1, 0, -1, 0, 1, 0, 0};           /* Robot.cpp:28-31 */
    memcpy(c->Rf_q0, rf, sizeof(rf));
    c->wbc_calls_per_eval = 1;
    c->xscale = 1.0;
    c->plant = 0;
    c->contact_k = 5.0e4; c->contact_d = 3.0e2; c->contact_dt = 3.0e2; c->contact_mu = 0.7;
}

void orc_controller_set_refs(orc_controller *c, int n, const double *zx, const double *zy, const unsigned char *phase)
{
    free(c->zmpX); free(c->zmpY); free(c->phase);
    c->n_zmp = n;
    c->zmpX = (double *)malloc(sizeof(double) * (size_t)n);
    c->zmpY = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(c->zmpX, zx, sizeof(double) * (size_t)n);
    memcpy(c->zmpY, zy, sizeof(double) * (size_t)n);
    c->phase = NULL;
    if (phase) {
        c->phase = (unsigned char *)malloc((size_t)n);
        memcpy(c->phase, phase, (size_t)n);
    }
}

void orc_controller_free(orc_controller *c)
{
    free(c->zmpX); free(c->zmpY); free(c->phase); free(c->segs); free(c->seg_of_sample);
    c->zmpX = c->zmpY = NULL; c->phase = NULL; c->segs = NULL; c->seg_of_sample = NULL; c->n_seg = 0;
}

void orc_controller_set_segments(orc_controller *c, int n_seg, const double *segs, const unsigned short *seg_of_sample, double xscale)
{
    free(c->segs); free(c->seg_of_sample);
    c->segs = NULL; c->seg_of_sample = NULL;
    c->n_seg = n_seg;
    c->xscale = xscale;
    if (n_seg > 0) {
        c->segs = (double *)malloc(sizeof(double) * ORC_SEG_STRIDE * (size_t)n_seg);
        memcpy(c->segs, segs, sizeof(double) * ORC_SEG_STRIDE * (size_t)n_seg);
        c->seg_of_sample = (unsigned short *)malloc(sizeof(unsigned short) * (size_t)c->n_zmp);
        memcpy(c->seg_of_sample, seg_of_sample, sizeof(unsigned short) * (size_t)c->n_zmp);
    }
}

static void pd_joints_acc(const orc_system *s, double qppRef[ORC_NQ])    /* :296-308 */
{
    double qDes[ORC_NQ];
    const orc_gains *g = &s->ctl.gains;
    orc_desired_posture(qDes);
    for (int i = 0; i < ORC_NQ; i++)
        qppRef[i] = g->KpJoints * (qDes[i] - s->robot.q[i]) + g->KdJoints * (0.0 - s->robot.v[i]);
    for (int i = 0; i < 3; i++) { double t = qppRef[i]; qppRef[i] = qppRef[3 + i]; qppRef[3 + i] = t; }
}

static void pd_momentum_acc(const orc_system *s, double hGpRef[6])       /* :310-325 */
{
    const orc_gains *g = &s->ctl.gains;
    const orc_mpc *m = &s->mpc;
    const orc_robot *r = &s->robot;
    double posRef[3] = {m->xRef[0], m->yRef[0], m->zCom};
    double velRef[3] = {m->xRef[1], m->yRef[1], 0};
    double accRef[3] = {m->xRef[2], m->yRef[2], 0};
    for (int k = 0; k < 3; k++) {
        hGpRef[3 + k] = r->mass * (g->KpMom * (posRef[k] - r->CoM[k]) + g->KdMom * (velRef[k] - r->comVel[k]) + accRef[k]);
        hGpRef[k] = g->KdMom * (0.0 - r->comAngMom[k]);
    }
}

static void pd_feet_acc(const orc_system *s, const double *JFeet, double t, int k, double footAccRef[12])  /* :327-386 */
{
    const orc_gains *g = &s->ctl.gains;
    const orc_robot *r = &s->robot;
    const orc_controller *c = &s->ctl;
    double v[ORC_NQ], vel[12];
    memcpy(v, r->v, sizeof(v));
    orc_swap_base_velocity(r->X[0], v);
    orc_mv(12, ORC_NQ, JFeet, v, vel);
    const int frames[2] = {7, 14};
    for (int sfoot = 0; sfoot < 2; sfoot++) {
        const double *T = r->T[frames[sfoot]];
        const double (*co)[8] = sfoot ? c->lF : c->rF;
        const int *nc = sfoot ? c->lFn : c->rFn;
        static const int nc8[3] = {8, 8, 8};
        double tl = t, cs[3][8];
        if (c->n_seg > 0) {                                        /* extension: segment of preview index k */
            int kk = (k < 0) ? 0 : (k >= c->n_zmp ? c->n_zmp - 1 : k);
            const double *sg = c->segs + (size_t)ORC_SEG_STRIDE * c->seg_of_sample[kk];
            tl = t - sg[0];
            for (int a = 0; a < 3; a++)
                for (int i = 0; i < 8; i++) cs[a][i] = sg[1 + 24 * sfoot + 8 * a + i] * ((a == 0) ? c->xscale : 1.0);
            co = (const double (*)[8])cs;
            nc = nc8;
        }
        double Rf[9], err[9], aa[3], e[3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Rf[a * 3 + b] = T[a * 4 + b];
        orc_mtm(3, 3, 3, c->Rf_q0, Rf, err);                     /* Rdes' * R_foot */
        orc_rot_to_axis_angle(err, aa);
        orc_mv(3, 3, c->Rf_q0, aa, e);
        for (int k = 0; k < 3; k++) e[k] = -e[k];                /* e = -Rdes * log(...) */
        for (int k = 0; k < 3; k++) {
            double d1[8], d2[8];
            int n1 = orc_polyder(co[k], nc[k], d1);
            int n2 = orc_polyder(d1, n1, d2);
            double pRef = orc_polyval(co[k], nc[k], tl);
            double vRef = orc_polyval(d1, n1, tl);
            double aRef = orc_polyval(d2, n2, tl);
            double posErrA = e[k];
            double posErrL = pRef - T[k * 4 + 3];
            double velErrA = 0.0 - vel[6 * sfoot + k];
            double velErrL = vRef - vel[6 * sfoot + 3 + k];
            footAccRef[6 * sfoot + k] = g->KpFeet * posErrA + g->KdFeet * velErrA + 0.0;
            footAccRef[6 * sfoot + 3 + k] = g->KpFeet * posErrL + g->KdFeet * velErrL + aRef;
        }
    }
}

/* frictionConstraints, :156-294 : rows of Aeq (12 x 74) and Aineq (32 x 74) */
static void friction_rows(const orc_controller *c, double *Aeq /*12x74*/)
{
    const int idx_nR = 30, idx_fR = 33, idx_nL = 36, idx_fL = 39, idx_muR = 42, idx_muL = 58;
    const int rfR = 0, rnR = 3, rfL = 6, rnL = 9;
    memset(Aeq, 0, 12 * ORC_NV * sizeof(double));
    for (int k = 0; k < 3; k++) {
        Aeq[(rfR + k) * ORC_NV + idx_fR + k] = -1;
        Aeq[(rfL + k) * ORC_NV + idx_fL + k] = -1;
        Aeq[(rnR + k) * ORC_NV + idx_nR + k] = -1;
        Aeq[(rnL + k) * ORC_NV + idx_nL + k] = -1;
        for (int v = 0; v < 4; v++)
            for (int e = 0; e < 4; e++) {
                Aeq[(rfR + k) * ORC_NV + idx_muR + 4 * v + e] = c->friction[k * 4 + e];
                Aeq[(rfL + k) * ORC_NV + idx_muL + 4 * v + e] = c->friction[k * 4 + e];
            }
    }
    for (int v = 0; v < 4; v++) {
        double cm[9], tmp[12];
        orc_cross_matrix(c->footVertices[v], cm);
        orc_mm(3, 3, 4, cm, c->friction, tmp);                   /* p_v x friction basis */
        for (int k = 0; k < 3; k++)
            for (int e = 0; e < 4; e++) {
                Aeq[(rnR + k) * ORC_NV + idx_muR + 4 * v + e] = tmp[k * 4 + e];
                Aeq[(rnL + k) * ORC_NV + idx_muL + 4 * v + e] = tmp[k * 4 + e];
            }
    }
}

static void wbc(orc_system *s, double t, const double *JFeet, int phase, orc_eval *out)   /* :81-154, 388-479 */
{
    const orc_gains *g = &s->ctl.gains;
    const orc_dynamics *d = &s->dyn;
    const int n = ORC_NQ;
    double *H = out->H, *gv = out->g, *A = out->A;
    pd_joints_acc(s, out->qppRef);
    pd_momentum_acc(s, out->hGpRef);
    pd_feet_acc(s, JFeet, t, s->mpc.last_k, out->footAccRef);

    double WJ[ORC_NQ], WC[6];
    for (int i = 0; i < 3; i++) { WJ[i] = g->wBasePos; WJ[3 + i] = g->wBaseAng; WC[i] = g->wCoMK; WC[3 + i] = g->wCoML; }
    for (int i = 6; i < n; i++) WJ[i] = g->wJoints;

    memset(H, 0, sizeof(out->H));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < 6; k++) s1 += d->AG[k * n + i] * WC[k] * d->AG[k * n + j];
            for (int k = 0; k < 12; k++) s2 += JFeet[k * n + i] * g->wFoot * JFeet[k * n + j];
            H[i * ORC_NV + j] = s1 + ((i == j) ? WJ[i] : 0.0) + s2;
        }
    for (int i = 0; i < 12; i++) H[(n + i) * ORC_NV + n + i] = g->wForce;
    for (int i = 42; i < ORC_NV; i++) H[i * ORC_NV + i] = g->epsCoeff;

    memset(gv, 0, sizeof(out->g));
    for (int i = 0; i < n; i++) {
        double a1 = 0, a2 = 0, a4 = 0, a5 = 0;
        for (int k = 0; k < 6; k++) { a1 += d->AG[k * n + i] * WC[k] * d->AGpqp[k]; a2 += d->AG[k * n + i] * WC[k] * out->hGpRef[k]; }
        for (int k = 0; k < 12; k++) { a4 += JFeet[k * n + i] * g->wFoot * d->Jpqp[k]; a5 += JFeet[k * n + i] * g->wFoot * out->footAccRef[k]; }
        gv[i] = a1 - a2 - WJ[i] * out->qppRef[i] + a4 - a5;
    }

    /* solveQP :388-479 */
    for (int i = 0; i < ORC_NV; i++)                              /* Hsym = 0.5 (H + H') */
        for (int j = i + 1; j < ORC_NV; j++) {
            double hs = 0.5 * (H[i * ORC_NV + j] + H[j * ORC_NV + i]);
            H[i * ORC_NV + j] = hs; H[j * ORC_NV + i] = hs;
        }
    memset(A, 0, sizeof(out->A));
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < n; j++) A[i * ORC_NV + j] = d->M[i * n + j];
        for (int j = 0; j < 12; j++) A[i * ORC_NV + n + j] = -JFeet[j * n + i];
        out->lbA[i] = -d->C[i]; out->ubA[i] = -d->C[i];
    }
    friction_rows(&s->ctl, A + 6 * ORC_NV);
    for (int i = 0; i < 12; i++) { out->lbA[6 + i] = 0; out->ubA[6 + i] = 0; }
    for (int i = 0; i < 32; i++) {
        A[(18 + i) * ORC_NV + 42 + i] = 1.0;
        out->lbA[18 + i] = 0;
        out->ubA[18 + i] = ORC_INFTY;
    }
    /* extension: swing / flight feet carry no force (see file header) */
    int rOff = (phase == 2 || phase == 3), lOff = (phase == 1 || phase == 3);
    for (int i = 0; i < 16; i++) {
        if (rOff) out->ubA[18 + i] = 0;
        if (lOff) out->ubA[18 + 16 + i] = 0;
    }

    unsigned char act[ORC_NC];
    out->qp_status = orc_qp_solve(ORC_NV, H, gv, ORC_NC, A, out->lbA, out->ubA, out->x, &out->qp_iters, act);
    out->active_mask = 0;
    for (int i = 0; i < 32; i++) if (act[18 + i]) out->active_mask |= (1u << i);

    const double *x = out->x;
    double gen[ORC_NQ];
    for (int i = 0; i < n; i++) {
        double a = 0, b = 0;
        for (int j = 0; j < n; j++) a += d->M[i * n + j] * x[j];
        for (int j = 0; j < 12; j++) b += JFeet[j * n + i] * x[n + j];
        gen[i] = a + d->C[i] - b;                                 /* :138-139 */
    }
    for (int i = 0; i < 6; i++) out->genForceBaseResidual[i] = gen[i];
    for (int i = 0; i < ORC_NJ; i++) out->tau[i] = gen[6 + i];
    for (int i = 0; i < 12; i++) out->f[i] = x[n + i];
    /* base acceleration back to the world frame, :143-147 */
    double X0[36], rhs[6], sol[6];
    memcpy(X0, s->robot.X[0], sizeof(X0));
    memcpy(rhs, x, sizeof(rhs));
    orc_solve_ge(6, X0, rhs, sol);
    for (int k = 0; k < 3; k++) { out->qpp[k] = sol[3 + k]; out->qpp[3 + k] = sol[k]; }
    for (int k = 0; k < ORC_NJ; k++) out->qpp[6 + k] = x[6 + k];
}

void orc_stand_step(orc_system *s, const double *q, const double *dq, double t, orc_eval *out)  /* :48-79 */
{
    orc_robot_update_state(&s->robot, q);
    orc_dynamics_compute_all(&s->dyn, &s->robot);               /* uses the STALE Robot::v_ */
    orc_feet_jacobian(&s->robot, out->JFeet);
    orc_robot_update_velocity(&s->robot, dq, s->dyn.AG);
    double pos[2] = {s->robot.CoM[0], s->robot.CoM[1]};
    double vel[2] = {s->robot.comVel[0], s->robot.comVel[1]};
    orc_mpc_compute(&s->mpc, pos, vel, s->ctl.zmpX, s->ctl.zmpY, t);
    out->k = s->mpc.last_k;
    out->u0x = s->mpc.xRef[2];
    out->u0y = s->mpc.yRef[2];
    out->phase = (s->ctl.phase && out->k >= 0 && out->k < s->ctl.n_zmp) ? s->ctl.phase[out->k] : 0;
    wbc(s, t, out->JFeet, out->phase, out);
}

/* BUILD-DEFINED (no reference semantics; the reference's intended feedback route, MuJoCo, is commented out at
 * apps/mujoco/main.cpp:115-122): contact wrench of the compliant ground on the two soles.  Vertex v of foot f sits at
 * x = o_sole + r_v, r_v = R_sole Rf_q0' p_v, and moves with xdot = v_o + w x r_v, (w, v_o) = J_f vhat (the sole's spatial velocity in
 * world axes, fresh velocity).  Penetration d = -x_z > 0: normal force max(0, k d - c xdot_z); tangential force -c_t xdot_xy, scaled back
 * onto the friction disc mu f_n.  Wrench about the sole origin in world axes, [n_R f_R n_L f_L] like the WBC's variables. */
void orc_contact_wrench(const orc_system *s, const double *JFeet, double w[12], double vf[8][3])
{
    const orc_controller *c = &s->ctl;
    const orc_robot *r = &s->robot;
    double v[ORC_NQ], vel[12];
    memcpy(v, r->v, sizeof(v));
    orc_swap_base_velocity(r->X[0], v);
    orc_mv(12, ORC_NQ, JFeet, v, vel);
    const int frames[2] = {7, 14};
    for (int f = 0; f < 2; f++) {
        const double *T = r->T[frames[f]];
        const double *om = vel + 6 * f, *vo = vel + 6 * f + 3;
        double n[3] = {0, 0, 0}, fs[3] = {0, 0, 0};
        for (int vi = 0; vi < 4; vi++) {
            /* the vertices (Robot.cpp:38-42) are offsets in WORLD-ALIGNED axes at the sole origin of a flat foot (that is how
             * frictionConstraints crosses them with world-axis forces, controller.cpp:225-270); they turn with the foot:
             * offset = R_sole Rf_q0' p_v, Rf_q0 = sole orientation of the flat foot (Robot.cpp:28-31) */
            const double *pw = c->footVertices[vi];
            double p[3], rp[3], x[3], xd[3], fv[3] = {0, 0, 0};
            for (int a = 0; a < 3; a++) p[a] = c->Rf_q0[0 * 3 + a] * pw[0] + c->Rf_q0[1 * 3 + a] * pw[1] + c->Rf_q0[2 * 3 + a] * pw[2];
            for (int a = 0; a < 3; a++) { rp[a] = T[a * 4] * p[0] + T[a * 4 + 1] * p[1] + T[a * 4 + 2] * p[2]; x[a] = rp[a] + T[a * 4 + 3]; }
            xd[0] = vo[0] + (om[1] * rp[2] - om[2] * rp[1]);
            xd[1] = vo[1] + (om[2] * rp[0] - om[0] * rp[2]);
            xd[2] = vo[2] + (om[0] * rp[1] - om[1] * rp[0]);
            const double pen = -x[2];
            if (pen > 0.0) {
                double fn = c->contact_k * pen - c->contact_d * xd[2];
                if (fn < 0.0) fn = 0.0;
                double ftx = -c->contact_dt * xd[0], fty = -c->contact_dt * xd[1];
                const double ft = sqrt(ftx * ftx + fty * fty), lim = c->contact_mu * fn;
                if (ft > lim) { const double sc = lim / ft; ftx *= sc; fty *= sc; }
                fv[0] = ftx; fv[1] = fty; fv[2] = fn;
            }
            if (vf) memcpy(vf[4 * f + vi], fv, sizeof(fv));
            n[0] += rp[1] * fv[2] - rp[2] * fv[1];
            n[1] += rp[2] * fv[0] - rp[0] * fv[2];
            n[2] += rp[0] * fv[1] - rp[1] * fv[0];
            for (int a = 0; a < 3; a++) fs[a] += fv[a];
        }
        for (int a = 0; a < 3; a++) { w[6 * f + a] = n[a]; w[6 * f + 3 + a] = fs[a]; }
    }
}

/* plant acceleration: M a = S'tau + J'w_contact - C(q, v) in the WBC's coordinates (base twist in the base frame, [ang; lin]), then back to
 * the world frame like controller.cpp:143-147.  M and J are the terms the controller evaluated in this call; the velocity products are
 * re-evaluated at the CURRENT velocity (Robot::v_ after controller.cpp:59): the controller's own C belongs to the previous call's
 * velocity (:56 runs before :59) and would let the momentum of the plant drift. */
static void plant_acceleration(orc_system *s, const orc_eval *out, double qpp[ORC_NQ])
{
    const int n = ORC_NQ;
    const orc_dynamics *d = &s->dyn;
    double w[12], rhs[ORC_NQ], a[ORC_NQ], M[ORC_NQ * ORC_NQ], Cnow[ORC_NQ];
    orc_contact_wrench(s, out->JFeet, w, NULL);
    orc_dynamics_bias_now(d, &s->robot, Cnow);
    for (int i = 0; i < n; i++) {
        double jw = 0.0;
        for (int k = 0; k < 12; k++) jw += out->JFeet[k * n + i] * w[k];
        rhs[i] = ((i >= 6) ? out->tau[i - 6] : 0.0) + jw - Cnow[i];
    }
    memcpy(M, d->M, sizeof(M));
    orc_solve_ge(n, M, rhs, a);
    double X0[36], rb[6], sol[6];
    memcpy(X0, s->robot.X[0], sizeof(X0));
    memcpy(rb, a, sizeof(rb));
    orc_solve_ge(6, X0, rb, sol);
    for (int k = 0; k < 3; k++) { qpp[k] = sol[3 + k]; qpp[3 + k] = sol[k]; }
    for (int k = 6; k < n; k++) qpp[k] = a[k];
}

/* apps/offline/main.cpp:91-122 */
void orc_plant_derivative(orc_system *s, const double *state, double t, double *xdot, orc_eval *out)
{
    const int n = ORC_NQ;
    const double *q = state, *qD = state + n;
    orc_stand_step(s, q, qD, t, out);
    if (s->ctl.wbc_calls_per_eval > 1)                            /* literal duplicate WBC(t), :105 (result-neutral) */
        wbc(s, t, out->JFeet, out->phase, out);
    double cm[9], Om[9], w[3], o[3];
    for (int i = 0; i < n; i++) xdot[i] = qD[i];
    w[0] = qD[3]; w[1] = qD[4]; w[2] = qD[5];
    orc_cross_matrix(w, cm);
    orc_mv(3, 3, cm, q, o);
    for (int k = 0; k < 3; k++) xdot[k] += o[k];                  /* v_classic = v_spatial + w x p, :111 */
    orc_omega_to_euler_rate(q + 3, Om);
    orc_mv(3, 3, Om, w, o);
    for (int k = 0; k < 3; k++) xdot[3 + k] = o[k];               /* :116 */
    if (s->ctl.plant) {
        double qpp[ORC_NQ];
        plant_acceleration(s, out, qpp);                          /* the torques drive a plant instead of being thrown away (main.cpp:118-121) */
        for (int i = 0; i < n; i++) xdot[n + i] = qpp[i];
    } else
        for (int i = 0; i < n; i++) xdot[n + i] = out->qpp[i];
}

/* rk4.hpp:5-18 */
void orc_rk4_tick(orc_system *s, double *x, double t, double dt, orc_eval *last)
{
    enum { NS = 2 * ORC_NQ };
    double k1[NS], k2[NS], k3[NS], k4[NS], xs[NS];
    orc_plant_derivative(s, x, t, k1, last);
    for (int i = 0; i < NS; i++) xs[i] = x[i] + 0.5 * dt * k1[i];
    orc_plant_derivative(s, xs, t + 0.5 * dt, k2, last);
    for (int i = 0; i < NS; i++) xs[i] = x[i] + 0.5 * dt * k2[i];
    orc_plant_derivative(s, xs, t + 0.5 * dt, k3, last);
    for (int i = 0; i < NS; i++) xs[i] = x[i] + dt * k3[i];
    orc_plant_derivative(s, xs, t + dt, k4, last);
    for (int i = 0; i < NS; i++) x[i] = x[i] + (dt / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}

/* apps/offline/main.cpp:12-58 */
void orc_system_init_offline(orc_system *s, double simulationTime, double timeStep, double timeHorizon, int do_ik)
{
    memset(s, 0, sizeof(*s));
    orc_robot_init(&s->robot, NULL);
    if (do_ik) {
        double Rf[6] = {0, -0.05, 0, 0, 0, 0}, Lf[6] = {0, 0.05, 0, 0, 0, 0}, com[3] = {-0.02, 0.0, 0.26}, Qd[ORC_NQ];
        orc_ik_desired_op(&s->robot, Rf, Lf, com, Qd);
        orc_ik_compute(&s->robot, Qd);
    }
    orc_mpc_init(&s->mpc, timeStep, timeHorizon, s->robot.CoM[2]);
    orc_controller_init(&s->ctl);
    double *zx, *zy;
    int n = orc_zmp_stance(simulationTime, timeStep, 2, &zx, &zy);
    orc_controller_set_refs(&s->ctl, n, zx, zy, NULL);
    free(zx); free(zy);
    double cur[3] = {0, -0.05, 0};
    orc_foot_coeff_trajectory(cur, cur, 0.0, simulationTime, s->ctl.rF, s->ctl.rFn);
    cur[1] = 0.05;
    orc_foot_coeff_trajectory(cur, cur, 0.0, simulationTime, s->ctl.lF, s->ctl.lFn);
}

void orc_system_free(orc_system *s) { orc_controller_free(&s->ctl); }
unsigned long orc_sizeof_system(void) { return (unsigned long)sizeof(orc_system); }
unsigned long orc_sizeof_eval(void) { return (unsigned long)sizeof(orc_eval); }
