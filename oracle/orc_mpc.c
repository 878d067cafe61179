/* orc_mpc.c -- CPU ORACLE (test infrastructure): LIPM preview controller + reference generators.
 * Follows reference src/mpcLinearPendulum.cpp, src/zmpGeneration.cpp, src/footRefTrajectory.cpp.
 *
 * qpOASES (un-vendored, version unpinned) is called there on an UNCONSTRAINED strictly convex
 * QP (QProblem(nV, 0), no bounds: mpcLinearPendulum.cpp:17-18,120-127), whose minimiser is
 * u = -H^{-1} g.  This file computes exactly that with a Cholesky factorisation. */
#include <stdlib.h>
#include "lmh_oracle.h"
#include "orc_linalg.h"
#include "orc_internal.h"

#define LD (ORC_MAXH + 1)

static void build_hessian(const orc_mpc *m, double *H)            /* :89-90 */
{
    int n = m->horizon + 1;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int l = 0; l < n; l++) s += m->Pu[l * LD + i] * m->Pu[l * LD + j];
            H[i * n + j] = ((i == j) ? m->alpha : 0.0) + m->beta * s;
        }
}

void orc_mpc_init(orc_mpc *m, double dt, double timeHorizon, double zCom)   /* :10-76 */
{
    memset(m, 0, sizeof(*m));
    m->dt = dt; m->timeHorizon = timeHorizon; m->zCom = zCom;
    m->gravity = 9.81; m->alpha = 1e-3; m->beta = 1;              /* mpcLinearPendulum.hpp:46-48 */
    m->zmp_xscale = 1.0;
    m->horizon = (int)(timeHorizon / dt);                          /* :43 */
    if (m->horizon > ORC_MAXH) m->horizon = ORC_MAXH;
    const int N = m->horizon;
    m->A[0] = 1; m->A[1] = dt; m->A[2] = 0; m->A[3] = 1;
    m->B[0] = (dt * dt) / 2; m->B[1] = dt;
    m->Cm[0] = 1; m->Cm[1] = 0;
    m->D = -zCom / m->gravity;
    double Ap[4] = {1, 0, 0, 1}, t[4];
    m->Px[0] = m->Cm[0]; m->Px[1] = m->Cm[1];
    m->Pu[0] = m->D;
    for (int i = 1; i <= N; i++) {
        orc_mm(2, 2, 2, Ap, m->A, t); memcpy(Ap, t, sizeof(t));   /* A_power *= A */
        m->Px[i * 2 + 0] = m->Cm[0] * Ap[0] + m->Cm[1] * Ap[2];
        m->Px[i * 2 + 1] = m->Cm[0] * Ap[1] + m->Cm[1] * Ap[3];
        m->Pu[i * LD + (i - 1)] = m->Cm[0] * m->B[0] + m->Cm[1] * m->B[1];
        m->Pu[i * LD + i] = m->D;
        double Aj[4] = {1, 0, 0, 1};
        for (int j = 1; j <= N - i; j++) {
            orc_mm(2, 2, 2, Aj, m->A, t); memcpy(Aj, t, sizeof(t));
            double ca0 = m->Cm[0] * Aj[0] + m->Cm[1] * Aj[2];
            double ca1 = m->Cm[0] * Aj[1] + m->Cm[1] * Aj[3];
            m->Pu[(i + j) * LD + (i - 1)] = ca0 * m->B[0] + ca1 * m->B[1];
        }
    }
}

static void ensure_factor(orc_mpc *m)
{
    int n = m->horizon + 1;
    if (m->have_factor && !m->faithful_rebuild) return;
    double *H = (double *)malloc((size_t)n * n * sizeof(double));
    build_hessian(m, H);
    if (!m->have_factor) {                       /* qpOASES factors once (init), then hotstarts */
        orc_cholesky(n, H);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) m->Lh[i * LD + j] = H[i * n + j];
        m->have_factor = 1;
    }
    free(H);
}

static double solve_axis(const orc_mpc *m, const double xk[2], const double *zmp, double zscale, double *g)
{
    int n = m->horizon + 1;
    double r[LD], u[LD], Lc[LD * LD];
    for (int i = 0; i < n; i++) r[i] = (m->Px[i * 2] * xk[0] + m->Px[i * 2 + 1] * xk[1]) - zmp[i] * zscale;
    for (int j = 0; j < n; j++) {                /* g = beta * Pu' * r, :96-97 */
        double s = 0.0;
        for (int i = 0; i < n; i++) s += m->beta * m->Pu[i * LD + j] * r[i];
        g[j] = s;
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Lc[i * n + j] = m->Lh[i * LD + j];
    orc_chol_solve(n, Lc, g, u);
    return -u[0];                                /* first entry of the primal solution, :130-132 */
}

void orc_mpc_compute(orc_mpc *m, const double pos[2], const double vel[2],
                     const double *zmpX, const double *zmpY, double t)       /* :78-109 */
{
    double xk[2] = {pos[0], vel[0]}, yk[2] = {pos[1], vel[1]};
    ensure_factor(m);
    int k = (int)(t / m->dt);                    /* :92 -- the only data-dependent integer */
    m->last_k = k;
    double accx = solve_axis(m, xk, zmpX + k, m->zmp_xscale, m->last_gx);
    double accy = solve_axis(m, yk, zmpY + k, 1.0, m->last_gy);
    double nx0 = m->A[0] * xk[0] + m->A[1] * xk[1] + m->B[0] * accx;
    double nx1 = m->A[2] * xk[0] + m->A[3] * xk[1] + m->B[1] * accx;
    double ny0 = m->A[0] * yk[0] + m->A[1] * yk[1] + m->B[0] * accy;
    double ny1 = m->A[2] * yk[0] + m->A[3] * yk[1] + m->B[1] * accy;
    m->xRef[0] = nx0; m->xRef[1] = nx1; m->xRef[2] = accx;
    m->yRef[0] = ny0; m->yRef[1] = ny1; m->yRef[2] = accy;
}

/* K = beta * e0' H^{-1} Pu'   so that  u0 = -K (Px x_k - zmp[k:k+N+1])  (algebraic form of
 * :96-101; used to cross-check the device gain row, not by the oracle's own compute). */
void orc_mpc_gain_row(const orc_mpc *m, double *K)
{
    int n = m->horizon + 1;
    orc_mpc tmp = *m;
    tmp.have_factor = 0; tmp.faithful_rebuild = 0;
    ensure_factor(&tmp);
    double e0[LD], h0[LD], Lc[LD * LD];
    memset(e0, 0, sizeof(e0)); e0[0] = 1.0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Lc[i * n + j] = tmp.Lh[i * LD + j];
    orc_chol_solve(n, Lc, e0, h0);               /* H^{-1} e0 (H symmetric) */
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += m->Pu[i * LD + j] * h0[j];
        K[i] = m->beta * s;
    }
}

/* zmpGeneration.cpp:39-60; supportFoot: 0 Right, 1 Left, 2 Double (Task.hpp:9-13) */
int orc_zmp_stance(double simulationTime, double timeStep, int supportFoot, double **zx, double **zy)
{
    int samples = (int)((simulationTime + 0.5) / timeStep);
    double *x = (double *)calloc((size_t)samples, sizeof(double));
    double *y = (double *)calloc((size_t)samples, sizeof(double));
    double yv = (supportFoot == 0) ? -0.05 : ((supportFoot == 1) ? 0.05 : 0.0);
    for (int i = 0; i < samples; i++) y[i] = yv;
    *zx = x; *zy = y;
    return samples;
}

/* footRefTrajectory.cpp:4-47 : x,y 6 coefficients, z 8 coefficients (ascending powers) */
void orc_foot_coeff_trajectory(const double cur[3], const double des[3], double stepHeight,
                               double T, double coeff[3][8], int ncoef[3])
{
    double vel2[4] = {0, 0, T, 0}, acc2[4] = {0, 0, T, 0};
    memset(coeff, 0, 3 * 8 * sizeof(double));
    for (int ax = 0; ax < 2; ax++) {
        double pos[4] = {0, cur[ax], T, des[ax]};
        orc_find_poly_coeff(2, pos, 2, vel2, 2, acc2, coeff[ax]);
        ncoef[ax] = 6;
    }
    double posz[6] = {0, cur[2], T / 2, stepHeight, T, des[2]};
    double velz[6] = {0, 0, T / 2, 0, T, 0};
    orc_find_poly_coeff(3, posz, 3, velz, 2, acc2, coeff[2]);
    ncoef[2] = 8;
}
