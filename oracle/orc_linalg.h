/* orc_linalg.h -- tiny dense row-major helpers for the CPU oracle (test infrastructure). */
#ifndef ORC_LINALG_H
#define ORC_LINALG_H
#include <math.h>
#include <string.h>

/* C(m x n) = A(m x k) * B(k x n) */
static inline void orc_mm(int m, int k, int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
/* C(k x n) = A(m x k)^T * B(m x n) */
static inline void orc_mtm(int m, int k, int n, const double *A, const double *B, double *C)
{
    for (int i = 0; i < k; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int l = 0; l < m; l++) s += A[l * k + i] * B[l * n + j];
            C[i * n + j] = s;
        }
}
/* y(m) = A(m x n) x(n) */
static inline void orc_mv(int m, int n, const double *A, const double *x, double *y)
{
    for (int i = 0; i < m; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += A[i * n + j] * x[j];
        y[i] = s;
    }
}
/* y(n) = A(m x n)^T x(m) */
static inline void orc_mtv(int m, int n, const double *A, const double *x, double *y)
{
    for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int i = 0; i < m; i++) s += A[i * n + j] * x[i];
        y[j] = s;
    }
}
static inline void orc_transpose(int m, int n, const double *A, double *At)
{
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) At[j * m + i] = A[i * n + j];
}
/* Solve A x = b (n x n) by Gaussian elimination with partial pivoting; A, b destroyed.
 * Stands in for Eigen's colPivHouseholderQr().solve() on square non-singular systems
 * (same solution up to round-off).  Returns 0 on success. */
static inline int orc_solve_ge(int n, double *A, double *b, double *x)
{
    for (int c = 0; c < n; c++) {
        int p = c;
        double best = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++)
            if (fabs(A[r * n + c]) > best) { best = fabs(A[r * n + c]); p = r; }
        if (best == 0.0) return 1;
        if (p != c) {
            for (int j = 0; j < n; j++) { double t = A[c * n + j]; A[c * n + j] = A[p * n + j]; A[p * n + j] = t; }
            double t = b[c]; b[c] = b[p]; b[p] = t;
        }
        for (int r = c + 1; r < n; r++) {
            double f = A[r * n + c] / A[c * n + c];
            if (f != 0.0) {
                for (int j = c; j < n; j++) A[r * n + j] -= f * A[c * n + j];
                b[r] -= f * b[c];
            }
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int j = r + 1; j < n; j++) s -= A[r * n + j] * x[j];
        x[r] = s / A[r * n + r];
    }
    return 0;
}
/* In-place lower Cholesky of the n x n SPD matrix A (row-major, ld = n). Returns 0 ok. */
static inline int orc_cholesky(int n, double *A)
{
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
        for (int i = 0; i < j; i++) A[i * n + j] = 0.0;
    }
    return 0;
}
/* solve L L^T x = b with the factor from orc_cholesky */
static inline void orc_chol_solve(int n, const double *L, const double *b, double *x)
{
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}
#endif
