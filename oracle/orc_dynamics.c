/* orc_dynamics.c -- CPU ORACLE (test infrastructure): rigid-body terms.
 * Follows reference src/Dynamics.cpp (C, Cg, M, AG, AGpqp, Jpqp) and the Jacobian / IK
 * halves of src/invKinematics.cpp.  Dense 6x6 arithmetic, as the reference does it. */
#include "lmh_oracle.h"
#include "orc_linalg.h"
#include "orc_internal.h"

static void spatial_inertia(const orc_link *l, double I[36])      /* Dynamics.cpp:4-13 */
{
    double c[9], cc[9];
    orc_cross_matrix(l->com, c);
    orc_mm(3, 3, 3, c, c, cc);
    memset(I, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            I[i * 6 + j] = l->inertia[i * 3 + j] - (l->mass * cc[i * 3 + j]);   /* inertia - m*cx*cx */
            I[i * 6 + 3 + j] = l->mass * c[i * 3 + j];
            I[(3 + i) * 6 + j] = -l->mass * c[i * 3 + j];
            I[(3 + i) * 6 + 3 + j] = (i == j) ? l->mass : 0.0;
        }
}

static void all_spatial_inertias(orc_dynamics *d, const orc_robot *r)  /* :15-27 */
{
    spatial_inertia(&r->links[0], d->I[0]);
    for (int i = 1; i < ORC_NF; i++)
        if (orc_act[i] != 0) spatial_inertia(&r->links[i], d->I[i]);
        else memset(d->I[i], 0, sizeof(d->I[i]));   /* never read by the reference */
}

/* f = I*a + crf(v)*I*v */
static void body_force(const double I[36], const double v[6], const double a[6], double f[6])
{
    double Ia[6], Iv[6], cf[36], t[6];
    orc_mv(6, 6, I, a, Ia);
    orc_mv(6, 6, I, v, Iv);
    orc_spatial_cross_force(v, cf);
    orc_mv(6, 6, cf, Iv, t);
    for (int k = 0; k < 6; k++) f[k] = Ia[k] + t[k];
}

/* Dynamics.cpp:124-146.  qD is the base-frame reordered velocity [ang; lin; joints]. */
static void forward_newton_euler(const orc_dynamics *d, const orc_robot *r, const double *qD,
                                 double vel[ORC_NF][6], double acc[ORC_NF][6], double f[ORC_NF][6])
{
    for (int i = 1; i < ORC_NF; i++) {
        int p = orc_parent[i];
        double xv[6], xa[6];
        orc_mv(6, 6, r->X[i], vel[p], xv);
        orc_mv(6, 6, r->X[i], acc[p], xa);
        if (orc_act[i] != 0) {
            double qd = qD[orc_act[i] + 6 - 1];
            for (int k = 0; k < 6; k++) vel[i][k] = xv[k] + ((k == 2) ? qd : 0.0);   /* + S*qd */
            double cm[36];
            orc_spatial_cross(vel[i], cm);
            for (int k = 0; k < 6; k++) acc[i][k] = xa[k] + cm[k * 6 + 2] * qd;       /* + crm(v)*S*qd */
            body_force(d->I[i], vel[i], acc[i], f[i]);
        } else {
            for (int k = 0; k < 6; k++) { vel[i][k] = xv[k]; acc[i][k] = xa[k]; }
            memset(f[i], 0, sizeof(f[i]));
        }
    }
}

/* :29-60 -- NOTE r->v is Robot::v_, i.e. the velocity of the PREVIOUS standStep call:
 * controller.cpp:56 runs dyn_.computeAll before robot_.updateVelocityState (:59). */
static void compute_C(const orc_dynamics *d, const orc_robot *r, int isGravity, double C[ORC_NQ])
{
    double g[6] = {0, 0, 0, 0, 0, isGravity * 9.81};
    double qD[ORC_NQ];
    double vel[ORC_NF][6], acc[ORC_NF][6], f[ORC_NF][6];
    memset(C, 0, ORC_NQ * sizeof(double));
    memcpy(qD, r->v, sizeof(qD));
    orc_swap_base_velocity(r->X[0], qD);
    memcpy(vel[0], qD, sizeof(vel[0]));
    orc_mv(6, 6, r->X[0], g, acc[0]);
    body_force(d->I[0], vel[0], acc[0], f[0]);
    forward_newton_euler(d, r, qD, vel, acc, f);
    /* backwardNewtonEuler :148-163 */
    for (int i = ORC_NF - 1; i > 0; i--)
        if (orc_act[i] != 0) {
            double t[6];
            C[orc_act[i] + 6 - 1] = f[i][2];
            orc_mtv(6, 6, r->X[i], f[i], t);
            for (int k = 0; k < 6; k++) f[orc_parent[i]][k] = f[orc_parent[i]][k] + t[k];
        }
    for (int k = 0; k < 6; k++) C[k] = f[0][k];
}

static void compute_M(orc_dynamics *d, const orc_robot *r)       /* :62-101 */
{
    double H[ORC_NJ * ORC_NJ], F2[6 * ORC_NJ];
    double IcL[ORC_NF][36];                           /* composite inertias, start as I_ (:72) */
    memcpy(IcL, d->I, sizeof(IcL));
    memset(H, 0, sizeof(H));
    memset(F2, 0, sizeof(F2));
    memset(d->M, 0, sizeof(d->M));
    for (int i = ORC_NF - 1; i >= 0; i--) {
        if (orc_act[i] == 0) continue;
        int p = orc_parent[i];
        double t1[36], t2[36], f[6], t[6];
        orc_mtm(6, 6, 6, r->X[i], IcL[i], t1);        /* X^T * Ic */
        orc_mm(6, 6, 6, t1, r->X[i], t2);             /* (X^T Ic) X */
        for (int k = 0; k < 36; k++) IcL[p][k] = IcL[p][k] + t2[k];
        for (int k = 0; k < 6; k++) f[k] = IcL[i][k * 6 + 2];   /* Ic*S */
        int ai = orc_act[i] - 1;
        H[ai * ORC_NJ + ai] = f[2];
        int j = i;
        while (orc_parent[j] != 0) {
            orc_mtv(6, 6, r->X[j], f, t);
            memcpy(f, t, sizeof(f));
            j = orc_parent[j];
            int aj = orc_act[j] - 1;
            H[aj * ORC_NJ + ai] = f[2];
            H[ai * ORC_NJ + aj] = H[aj * ORC_NJ + ai];
        }
        orc_mtv(6, 6, r->X[j], f, t);
        for (int k = 0; k < 6; k++) F2[k * ORC_NJ + ai] = t[k];
    }
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) d->M[a * ORC_NQ + b] = IcL[0][a * 6 + b];
    for (int a = 0; a < ORC_NJ; a++)
        for (int b = 0; b < ORC_NJ; b++) d->M[(6 + a) * ORC_NQ + 6 + b] = H[a * ORC_NJ + b];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < ORC_NJ; b++) {
            d->M[a * ORC_NQ + 6 + b] = F2[a * ORC_NJ + b];
            d->M[(6 + b) * ORC_NQ + a] = F2[a * ORC_NJ + b];
        }
}

static void centroidal(orc_dynamics *d, const orc_robot *r)      /* :103-121 */
{
    double X1G[36], R[9], p1G[3], cp[9], Rcp[9];
    const double m = r->mass;
    p1G[0] = d->M[2 * ORC_NQ + 4] / m;
    p1G[1] = d->M[0 * ORC_NQ + 5] / m;
    p1G[2] = d->M[1 * ORC_NQ + 3] / m;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = r->T[0][i * 4 + j];
    orc_cross_matrix(p1G, cp);
    orc_mm(3, 3, 3, R, cp, Rcp);
    memset(X1G, 0, sizeof(X1G));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            X1G[i * 6 + j] = R[i * 3 + j];
            X1G[(3 + i) * 6 + 3 + j] = R[i * 3 + j];
            X1G[i * 6 + 3 + j] = -Rcp[i * 3 + j];
        }
    /* AG = X1G * [Ic1 F] = X1G * M[0:6, :] */
    orc_mm(6, 6, ORC_NQ, X1G, d->M, d->AG);
    orc_mv(6, 6, X1G, d->Cg, d->AGpqp);
}

static void compute_jpqp_frame(const orc_dynamics *d, const orc_robot *r, int frame, double out[6]) /* :165-200 */
{
    double qD[ORC_NQ];
    double vel[ORC_NF][6], acc[ORC_NF][6], f[ORC_NF][6];
    memcpy(qD, r->v, sizeof(qD));
    orc_swap_base_velocity(r->X[0], qD);
    memcpy(vel[0], qD, sizeof(vel[0]));
    memset(acc[0], 0, sizeof(acc[0]));
    body_force(d->I[0], vel[0], acc[0], f[0]);
    forward_newton_euler(d, r, qD, vel, acc, f);
    const double *T = r->T[frame];
    for (int k = 0; k < 3; k++) {
        out[k] = T[k * 4] * acc[frame][0] + T[k * 4 + 1] * acc[frame][1] + T[k * 4 + 2] * acc[frame][2];
        out[3 + k] = T[k * 4] * acc[frame][3] + T[k * 4 + 1] * acc[frame][4] + T[k * 4 + 2] * acc[frame][5];
    }
}

/* Dynamics::computeC(robot, gravity = true) (:29-60) on whatever Robot::v_ holds NOW.  The controller never calls it at this point; the
 * build-defined plant does, after Controller::standStep has stored the new velocity (controller.cpp:59), so that its velocity products
 * belong to the state being integrated and not to the previous call's.  d->I must be current (computeAll of this call). */
void orc_dynamics_bias_now(const orc_dynamics *d, const orc_robot *r, double C[ORC_NQ]) { compute_C(d, r, 1, C); }

void orc_dynamics_compute_all(orc_dynamics *d, const orc_robot *r)   /* :202-216 */
{
    all_spatial_inertias(d, r);
    compute_C(d, r, 1, d->C);
    compute_C(d, r, 0, d->Cg);
    compute_M(d, r);
    centroidal(d, r);
    compute_jpqp_frame(d, r, 7, d->Jpqp);
    compute_jpqp_frame(d, r, 14, d->Jpqp + 6);
}

/* ------------------------------ invKinematics.cpp: Jacobian half ------------------------------ */

static void frame_jacobian(const orc_robot *r, int frame, double J[6 * ORC_NQ])  /* :105-149 */
{
    double Xn[8][36], Xnew[8][36];
    memset(J, 0, 6 * ORC_NQ * sizeof(double));
    int numFrame = 1, i = frame;
    while (orc_parent[i] > 0) { numFrame++; i--; }
    int j = 0;
    for (i = numFrame - 1; i >= 0; i--) { memcpy(Xnew[i], r->X[frame - j], sizeof(Xnew[i])); j++; }
    memcpy(Xn[numFrame - 1], Xnew[numFrame - 1], sizeof(Xn[0]));
    i = numFrame - 1;
    j = frame - 1;
    do {
        int col = orc_act[j] + 6 - 1;
        for (int k = 0; k < 6; k++) J[k * ORC_NQ + col] = Xn[i][k * 6 + 2];      /* Xn[i]*S */
        orc_mm(6, 6, 6, Xn[i], Xnew[i - 1], Xn[orc_parent[i]]);                  /* local index reuse, :141 */
        i--;
        j--;
    } while (orc_parent[i + 1] > 0);
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) J[a * ORC_NQ + b] = Xn[0][a * 6 + b];
}

void orc_feet_jacobian(const orc_robot *r, double *JFeet)        /* :72-103 */
{
    double Jf[6 * ORC_NQ];
    const int frames[2] = {7, 14};
    for (int s = 0; s < 2; s++) {
        frame_jacobian(r, frames[s], Jf);
        const double *T = r->T[frames[s]];
        for (int c = 0; c < ORC_NQ; c++)
            for (int k = 0; k < 3; k++) {
                JFeet[(6 * s + k) * ORC_NQ + c] =
                    T[k * 4] * Jf[0 * ORC_NQ + c] + T[k * 4 + 1] * Jf[1 * ORC_NQ + c] + T[k * 4 + 2] * Jf[2 * ORC_NQ + c];
                JFeet[(6 * s + 3 + k) * ORC_NQ + c] =
                    T[k * 4] * Jf[3 * ORC_NQ + c] + T[k * 4 + 1] * Jf[4 * ORC_NQ + c] + T[k * 4 + 2] * Jf[5 * ORC_NQ + c];
            }
    }
}

/* --------------------------------- invKinematics.cpp: IK half --------------------------------- */

static void rot_to_euler(const double T[16], const double ref[9], double eta[3])  /* :256-267 */
{
    double R[9], nR[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = T[i * 4 + j];
    orc_mm(3, 3, 3, R, ref, nR);
    eta[2] = atan2(nR[3], nR[0]);
    eta[1] = atan2(-nR[6], cos(eta[2]) * nR[0] + sin(eta[2]) * nR[3]);
    eta[0] = atan2(sin(eta[2]) * nR[2] - cos(eta[2]) * nR[5], -sin(eta[2]) * nR[1] + cos(eta[2]) * nR[4]);
}

static const double kRfq0[9] = {0, 0, 1, 0, -1, 0, 1, 0, 0};     /* Robot.cpp:28-31 */

static void operational_state(const orc_robot *r, double Q[ORC_NQ])  /* :54-70 */
{
    for (int k = 0; k < 3; k++) { Q[k] = r->T[7][k * 4 + 3]; Q[6 + k] = r->T[14][k * 4 + 3]; }
    rot_to_euler(r->T[7], kRfq0, Q + 3);
    rot_to_euler(r->T[14], kRfq0, Q + 9);
    for (int k = 0; k < 12; k++) Q[12 + k] = r->q[18 + k];
    for (int k = 0; k < 3; k++) { Q[24 + k] = r->q[3 + k]; Q[27 + k] = r->CoM[k]; }
}

static void inv3(const double *A, double *Ai)
{
    double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    Ai[0] = (A[4] * A[8] - A[5] * A[7]) / det; Ai[1] = (A[2] * A[7] - A[1] * A[8]) / det; Ai[2] = (A[1] * A[5] - A[2] * A[4]) / det;
    Ai[3] = (A[5] * A[6] - A[3] * A[8]) / det; Ai[4] = (A[0] * A[8] - A[2] * A[6]) / det; Ai[5] = (A[2] * A[3] - A[0] * A[5]) / det;
    Ai[6] = (A[3] * A[7] - A[4] * A[6]) / det; Ai[7] = (A[1] * A[6] - A[0] * A[7]) / det; Ai[8] = (A[0] * A[4] - A[1] * A[3]) / det;
}

static void com_jacobian(const orc_robot *r, double J[3 * ORC_NQ])   /* :206-244 */
{
    double JX[3 * ORC_NQ];
    memset(J, 0, 3 * ORC_NQ * sizeof(double));
    for (int i = 0; i < ORC_NF - 1; i++) {
        if (r->links[i].mass == 0) continue;
        const double *T = r->T[i];
        const double *c = r->links[i].com;
        double pCom[3], d[3], cm[9];
        for (int k = 0; k < 3; k++) pCom[k] = T[k * 4] * c[0] + T[k * 4 + 1] * c[1] + T[k * 4 + 2] * c[2] + T[k * 4 + 3];
        memset(JX, 0, sizeof(JX));
        for (int k = 0; k < 3; k++) { JX[k * ORC_NQ + k] = 1.0; d[k] = r->T[0][k * 4 + 3] - pCom[k]; }
        orc_cross_matrix(d, cm);                                              /* baseJacobian :246-254 */
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) JX[a * ORC_NQ + 3 + b] = cm[a * 3 + b];
        int j = i;
        while (j != 0) {
            if (orc_act[j] != 0) {
                double z[3] = {r->T[j][2], r->T[j][6], r->T[j][10]}, zc[9], dd[3], o[3];
                orc_cross_matrix(z, zc);
                for (int k = 0; k < 3; k++) dd[k] = pCom[k] - r->T[j][k * 4 + 3];
                orc_mv(3, 3, zc, dd, o);
                for (int k = 0; k < 3; k++) JX[k * ORC_NQ + orc_act[j] + 6 - 1] = o[k];
            }
            j = orc_parent[j];
        }
        for (int k = 0; k < 3 * ORC_NQ; k++) J[k] = J[k] + r->links[i].mass * JX[k];
    }
    for (int k = 0; k < 3 * ORC_NQ; k++) J[k] = J[k] / r->mass;
    double Om[9], Oi[9], blk[9], o[9];
    orc_omega_to_euler_rate(r->q + 3, Om);
    inv3(Om, Oi);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) blk[a * 3 + b] = J[a * ORC_NQ + 3 + b];
    orc_mm(3, 3, 3, blk, Oi, o);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) J[a * ORC_NQ + 3 + b] = o[a * 3 + b];
}

static void jac_inv_kinematics(const orc_robot *r, double J[ORC_NQ * ORC_NQ])  /* :151-204 */
{
    double Jf[12 * ORC_NQ], tmp;
    memset(J, 0, ORC_NQ * ORC_NQ * sizeof(double));
    orc_feet_jacobian(r, Jf);
    for (int row = 0; row < 12; row++)                       /* swap base column blocks */
        for (int c = 0; c < 3; c++) { tmp = Jf[row * ORC_NQ + c]; Jf[row * ORC_NQ + c] = Jf[row * ORC_NQ + 3 + c]; Jf[row * ORC_NQ + 3 + c] = tmp; }
    for (int s = 0; s < 2; s++)                              /* swap [ang;lin] rows of each foot */
        for (int row = 0; row < 3; row++)
            for (int c = 0; c < ORC_NQ; c++) {
                tmp = Jf[(6 * s + row) * ORC_NQ + c];
                Jf[(6 * s + row) * ORC_NQ + c] = Jf[(6 * s + 3 + row) * ORC_NQ + c];
                Jf[(6 * s + 3 + row) * ORC_NQ + c] = tmp;
            }
    double Om[9], Oi[9], blk[9], o[9];
    orc_omega_to_euler_rate(r->q + 3, Om);
    inv3(Om, Oi);
    for (int b4 = 0; b4 < 4; b4++) {                         /* Jf.block(3*b4,3,3,3) *= Omega^-1 */
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) blk[a * 3 + b] = Jf[(3 * b4 + a) * ORC_NQ + 3 + b];
        orc_mm(3, 3, 3, blk, Oi, o);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Jf[(3 * b4 + a) * ORC_NQ + 3 + b] = o[a * 3 + b];
    }
    const int feet[2] = {7, 14};
    for (int s = 0; s < 2; s++) {                            /* OmegaFoot * Jf.block(6s+3,3,3,3), :191-197 */
        double eta[3], Of[9];
        rot_to_euler(r->T[feet[s]], kRfq0, eta);
        orc_omega_to_euler_rate(eta, Of);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) blk[a * 3 + b] = Jf[(6 * s + 3 + a) * ORC_NQ + 3 + b];
        orc_mm(3, 3, 3, Of, blk, o);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Jf[(6 * s + 3 + a) * ORC_NQ + 3 + b] = o[a * 3 + b];
    }
    memcpy(J, Jf, sizeof(Jf));
    for (int k = 0; k < 12; k++) J[(12 + k) * ORC_NQ + 18 + k] = 1.0;
    for (int k = 0; k < 3; k++) J[(24 + k) * ORC_NQ + 3 + k] = 1.0;
    com_jacobian(r, J + 27 * ORC_NQ);
}

void orc_ik_desired_op(const orc_robot *r, const double *Rf, const double *Lf, const double *com, double *Qd) /* :11-25 */
{
    for (int k = 0; k < 6; k++) { Qd[k] = Rf[k]; Qd[6 + k] = Lf[k]; }
    for (int k = 0; k < 12; k++) Qd[12 + k] = r->q[18 + k];
    for (int k = 0; k < 3; k++) { Qd[24 + k] = 0.0; Qd[27 + k] = com[k]; }
}

/* :27-52.  The reference never increments `iter`; the cap below (A9 in SURVEY) only
 * prevents an endless loop, the shipped target converges in 4 Newton steps. */
int orc_ik_compute(orc_robot *r, const double *desOp)
{
    double q[ORC_NQ], Q[ORC_NQ], e[ORC_NQ], J[ORC_NQ * ORC_NQ], dq[ORC_NQ];
    memcpy(q, r->q, sizeof(q));
    operational_state(r, Q);
    double crit = 0;
    for (int k = 0; k < ORC_NQ; k++) { e[k] = desOp[k] - Q[k]; if (fabs(e[k]) > crit) crit = fabs(e[k]); }
    int iter = 0;
    while (crit > 1e-10 && iter < 200) {
        jac_inv_kinematics(r, J);
        if (orc_solve_ge(ORC_NQ, J, e, dq)) return -1;
        for (int k = 0; k < ORC_NQ; k++) q[k] += dq[k];
        orc_robot_update_state(r, q);
        operational_state(r, Q);
        crit = 0;
        for (int k = 0; k < ORC_NQ; k++) { e[k] = desOp[k] - Q[k]; if (fabs(e[k]) > crit) crit = fabs(e[k]); }
        iter++;
    }
    return iter;
}
