/* orc_qp.c -- CPU ORACLE (test infrastructure): dense strictly-convex QP solver.
 *
 * Replaces qpOASES::SQProblem::init(H,g,A,nullptr,nullptr,lbA,ubA,nWSR) as called by the
 * reference at src/controller.cpp:467-469 (cold start every call).  qpOASES is NOT in
 * /root/reference (un-vendored, version unpinned, CMakeLists.txt:39-41,76), so its
 * published behaviour is restated instead: it returns the minimiser of
 *        min 1/2 x'Hx + g'x   s.t.  lbA <= A x <= ubA
 * which is UNIQUE here because H is symmetric positive definite (controller.cpp:110-123:
 * W_J >= 1 on the acceleration block, 1 on the wrench block, 1e-8 on the coefficients).
 * The method below is the Goldfarb-Idnani dual active-set algorithm (Math. Prog. 27, 1983):
 * finite termination, exact up to fp64 round-off.  Rows with lbA == ubA are equalities,
 * bounds beyond +-ORC_INFTY are ignored (qpOASES::INFTY = 1e20).
 *
 * PARITY UNPINNED for this function: the reference holds no test or golden vector of any
 * QP result; tests/ verify the KKT conditions of every solution independently. */
#include <stdlib.h>
#include <float.h>
#include "lmh_oracle.h"
#include "orc_linalg.h"

typedef struct {
    int n, q;            /* variables, active constraints */
    double *J;           /* n x n, J = L^-T Q  (J J' = H^-1) */
    double *R;           /* n x n upper triangular (leading q x q used) */
    double *d, *z, *r;   /* work */
} gi_t;

static void givens(double a, double b, double *c, double *s, double *rr)
{
    if (b == 0.0) { *c = 1.0; *s = 0.0; *rr = a; return; }
    double h = hypot(a, b);
    *c = a / h; *s = b / h; *rr = h;
}

/* d = J' np */
static void compute_d(gi_t *w, const double *np)
{
    int n = w->n;
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = 0; k < n; k++) s += w->J[k * n + i] * np[k];
        w->d[i] = s;
    }
}
/* z = J2 d2 ; r = R^-1 d1 */
static void update_z_r(gi_t *w)
{
    int n = w->n, q = w->q;
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = q; j < n; j++) s += w->J[i * n + j] * w->d[j];
        w->z[i] = s;
    }
    for (int i = q - 1; i >= 0; i--) {
        double s = w->d[i];
        for (int j = i + 1; j < q; j++) s -= w->R[i * n + j] * w->r[j];
        w->r[i] = s / w->R[i * n + i];
    }
}
/* add the constraint whose transformed normal is d: zero d[q+1..n) with Givens on J columns */
static int add_constraint(gi_t *w)
{
    int n = w->n, q = w->q;
    for (int j = n - 1; j >= q + 1; j--) {
        double c, s, rr;
        givens(w->d[j - 1], w->d[j], &c, &s, &rr);
        if (w->d[j] == 0.0) continue;
        w->d[j - 1] = rr; w->d[j] = 0.0;
        for (int k = 0; k < n; k++) {
            double t1 = w->J[k * n + j - 1], t2 = w->J[k * n + j];
            w->J[k * n + j - 1] = c * t1 + s * t2;
            w->J[k * n + j] = -s * t1 + c * t2;
        }
    }
    for (int i = 0; i <= q; i++) w->R[i * n + q] = w->d[i];
    if (fabs(w->d[q]) <= DBL_EPSILON * 1e-3) return 1;   /* linearly dependent (cannot happen: see header) */
    w->q = q + 1;
    return 0;
}
/* delete active column l (position in the active list) */
static void delete_constraint(gi_t *w, int l)
{
    int n = w->n, q = w->q;
    for (int j = l; j < q - 1; j++)
        for (int i = 0; i < q; i++) w->R[i * n + j] = w->R[i * n + j + 1];
    for (int j = l; j < q - 1; j++) {
        double c, s, rr;
        givens(w->R[j * n + j], w->R[(j + 1) * n + j], &c, &s, &rr);
        if (w->R[(j + 1) * n + j] == 0.0) continue;
        w->R[j * n + j] = rr; w->R[(j + 1) * n + j] = 0.0;
        for (int k = j + 1; k < q - 1; k++) {
            double t1 = w->R[j * n + k], t2 = w->R[(j + 1) * n + k];
            w->R[j * n + k] = c * t1 + s * t2;
            w->R[(j + 1) * n + k] = -s * t1 + c * t2;
        }
        for (int k = 0; k < n; k++) {
            double t1 = w->J[k * n + j], t2 = w->J[k * n + j + 1];
            w->J[k * n + j] = c * t1 + s * t2;
            w->J[k * n + j + 1] = -s * t1 + c * t2;
        }
    }
    for (int i = 0; i < q; i++) w->R[i * n + q - 1] = 0.0;
    w->q = q - 1;
}

int orc_qp_solve(int n, const double *H, const double *g, int m, const double *A,
                 const double *lbA, const double *ubA, double *x, int *iters,
                 unsigned char *active_out)
{
    /* constraint list: (row, sign, rhs):  sign * a_row' x >= rhs ; equalities first */
    int nc = 0, ne = 0;
    int *crow = (int *)malloc(sizeof(int) * 2 * (size_t)m);
    double *csgn = (double *)malloc(sizeof(double) * 2 * (size_t)m);
    double *crhs = (double *)malloc(sizeof(double) * 2 * (size_t)m);
    for (int i = 0; i < m; i++)
        if (lbA[i] == ubA[i]) { crow[nc] = i; csgn[nc] = 1.0; crhs[nc] = lbA[i]; nc++; }
    ne = nc;
    for (int i = 0; i < m; i++) {
        if (lbA[i] == ubA[i]) continue;
        if (lbA[i] > -ORC_INFTY) { crow[nc] = i; csgn[nc] = 1.0; crhs[nc] = lbA[i]; nc++; }
        if (ubA[i] < ORC_INFTY) { crow[nc] = i; csgn[nc] = -1.0; crhs[nc] = -ubA[i]; nc++; }
    }

    gi_t w;
    w.n = n; w.q = 0;
    w.J = (double *)calloc((size_t)n * n, sizeof(double));
    w.R = (double *)calloc((size_t)n * n, sizeof(double));
    w.d = (double *)calloc((size_t)n, sizeof(double));
    w.z = (double *)calloc((size_t)n, sizeof(double));
    w.r = (double *)calloc((size_t)n, sizeof(double));
    double *L = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *np = (double *)malloc(sizeof(double) * (size_t)n);
    double *u = (double *)calloc((size_t)nc + 1, sizeof(double));
    int *Aset = (int *)malloc(sizeof(int) * ((size_t)nc + 1));
    unsigned char *isact = (unsigned char *)calloc((size_t)nc + 1, 1);
    int status = 0, it = 0;

    memcpy(L, H, sizeof(double) * (size_t)n * n);
    if (orc_cholesky(n, L)) { status = 2; goto done; }
    /* J = L^-T : solve L' J = I column by column */
    for (int c = 0; c < n; c++) {
        for (int i = n - 1; i >= 0; i--) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = i + 1; k < n; k++) s -= L[k * n + i] * w.J[k * n + c];
            w.J[i * n + c] = s / L[i * n + i];
        }
    }
    orc_chol_solve(n, L, g, x);
    for (int i = 0; i < n; i++) x[i] = -x[i];

    /* scale for the feasibility test */
    double xs = 1.0;

    for (int pass = 0;; pass++) {
        int ip = -1;
        double sp = 0.0;
        if (w.q < ne) {                 /* equalities are added first, in order, full steps */
            ip = w.q;
        } else {
            xs = 1.0;
            for (int i = 0; i < n; i++) if (fabs(x[i]) > xs) xs = fabs(x[i]);
            double worst = -1e-11 * xs;
            for (int c = ne; c < nc; c++) {
                if (isact[c]) continue;
                const double *a = A + (size_t)crow[c] * n;
                double s = -crhs[c];
                for (int k = 0; k < n; k++) s += csgn[c] * a[k] * x[k];
                if (s < worst) { worst = s; ip = c; }
            }
            if (ip < 0) break;          /* primal feasible: optimal */
        }
        if (++it > 20 * (nc + n)) { status = 3; break; }
        {
            const double *a = A + (size_t)crow[ip] * n;
            for (int k = 0; k < n; k++) np[k] = csgn[ip] * a[k];
            sp = -crhs[ip];
            for (int k = 0; k < n; k++) sp += np[k] * x[k];
        }
        u[w.q] = 0.0;
        Aset[w.q] = ip;
        for (;;) {
            compute_d(&w, np);
            update_z_r(&w);
            /* step lengths */
            double t1 = ORC_INFTY, t2 = ORC_INFTY;
            int l = -1;
            for (int k = ne; k < w.q; k++)
                if (w.r[k] > 0.0 && u[k] / w.r[k] < t1) { t1 = u[k] / w.r[k]; l = k; }
            double zz = 0.0, zn = 0.0;
            for (int k = 0; k < n; k++) { zz += w.z[k] * w.z[k]; zn += w.z[k] * np[k]; }
            if (zz > DBL_EPSILON * DBL_EPSILON && zn > 0.0) t2 = -sp / zn;
            if (ip < ne) { t1 = ORC_INFTY; l = -1; if (zn > 0.0) t2 = -sp / zn; }   /* equality: signed full step */
            double t = (t1 < t2) ? t1 : t2;
            if (!(t == t)) { status = 5; goto done; }              /* NaN in the data */
            if (t >= ORC_INFTY) { status = 1; goto done; }         /* infeasible */
            if (t != t2 && l < 0) { status = 5; goto done; }
            if (t2 >= ORC_INFTY) {
                /* dual step only */
                for (int k = 0; k < w.q; k++) u[k] -= t * w.r[k];
                u[w.q] += t;
                isact[Aset[l]] = 0;
                for (int k = l; k < w.q; k++) { u[k] = u[k + 1]; Aset[k] = Aset[k + 1]; }
                delete_constraint(&w, l);
                continue;
            }
            for (int k = 0; k < n; k++) x[k] += t * w.z[k];
            for (int k = 0; k < w.q; k++) u[k] -= t * w.r[k];
            u[w.q] += t;
            if (t == t2) {              /* full step: constraint becomes active */
                if (add_constraint(&w)) { status = 4; goto done; }
                isact[ip] = 1;
                break;
            }
            /* partial step: drop l, re-evaluate the violated constraint */
            isact[Aset[l]] = 0;
            for (int k = l; k < w.q; k++) { u[k] = u[k + 1]; Aset[k] = Aset[k + 1]; }
            delete_constraint(&w, l);
            sp = -crhs[ip];
            for (int k = 0; k < n; k++) sp += np[k] * x[k];
        }
    }

done:
    if (iters) *iters = it;
    if (active_out) {
        memset(active_out, 0, (size_t)m);
        for (int k = 0; k < w.q; k++) active_out[crow[Aset[k]]] = 1;
    }
    free(crow); free(csgn); free(crhs); free(w.J); free(w.R); free(w.d); free(w.z); free(w.r);
    free(L); free(np); free(u); free(Aset); free(isact);
    return status;
}
