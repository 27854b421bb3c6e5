/* CPU oracle (plain C, float64, OpenMP) for the NNGP/NTK hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path never links or calls it.
 *
 * PARITY UNPINNED: the arithmetic restated here lives in neural-tangents==0.6.1 / jax==0.3.23
 * (reference nngp.yaml:78-79,88), which is absent from /root/reference and not installable;
 * the reference holds no golden vectors for this path.  The restatement is
 * pinned by the known-answer tests in tests/test_oracle.py against oracle/nngp_oracle.py
 * and the Cho-Saul arc-cosine identity.
 *
 * Call sites restated:
 *   oracle_kernel_build : kernel_fn of stax.serial(Dense,Relu,Dense)    train.py:161-164
 *   oracle_fit          : gradient_descent_mse_ensemble (k_dd, relative
 *                         diag_reg, cho_factor, cho_solve)              train.py:171-172
 *   oracle_predict      : predict_fn(x_test, get, compute_cov=True)     train.py:157-158
 * This is the same algorithm class the reference runs on XLA-CPU (matmul Gram, fused
 * elementwise arc-cosine map, blocked potrf, trsm); it is the timed "port" CPU baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GET_NNGP 1
#define GET_NTK 2
#define PI 3.14159265358979323846

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static void row_sqnorm(const double* x, int64_t n, int d, double* q) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += x[i * d + k] * x[i * d + k];
        q[i] = s / d;
    }
}

/* One element through Dense,(Relu,Dense)* given k0 = x.x'/d, q1, q2.  train.py:161-164. */
static inline void layer_map(double k, double q1, double q2, int n_dense, const double* w,
                             const double* b, double* out_k, double* out_t) {
    double t = 0.0;
    for (int l = 0; l < n_dense; ++l) {
        const double w2 = w[l] * w[l], b2 = b[l] * b[l];
        k = w2 * k + b2;
        q1 = w2 * q1 + b2;
        q2 = w2 * q2 + b2;
        t = k + w2 * t;
        if (l < n_dense - 1) {
            double r = q1 * q2 - k * k;
            double s = r > 0.0 ? sqrt(r) : 0.0;
            double th = (s == 0.0 && k == 0.0) ? PI / 2 : atan2(s, k);
            double kd = (PI - th) / (2 * PI);
            k = s / (2 * PI) + kd * k;
            t = kd * t;
            q1 *= 0.5;
            q2 *= 0.5;
        }
    }
    *out_k = k;
    *out_t = t;
}

/* out[i*ld + j] for i in [0,n1), j in [0,n2).  x2 == NULL => symmetric (x2 = x1). */
int oracle_kernel_build(const double* x1, int64_t n1, const double* x2, int64_t n2, int d,
                        int n_dense, const double* w, const double* b, int get,
                        double* out_nngp, double* out_ntk, int64_t ld) {
    const int sym = (x2 == NULL);
    if (sym) { x2 = x1; n2 = n1; }
    double* q1 = (double*)malloc(sizeof(double) * n1);
    double* q2 = sym ? q1 : (double*)malloc(sizeof(double) * n2);
    if (!q1 || !q2) return -1;
    row_sqnorm(x1, n1, d, q1);
    if (!sym) row_sqnorm(x2, n2, d, q2);
    const int TB = 64;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int64_t ib = 0; ib < n1; ib += TB)
        for (int64_t jb = 0; jb < n2; jb += TB) {
            if (sym && jb > ib) continue;
            const int64_t ie = ib + TB < n1 ? ib + TB : n1, je = jb + TB < n2 ? jb + TB : n2;
            for (int64_t i = ib; i < ie; ++i) {
                double g[64];
                const double* xi = x1 + i * d;
                for (int64_t j = jb; j < je; ++j) {
                    const double* xj = x2 + j * d;
                    double s = 0.0;
#pragma omp simd reduction(+ : s)
                    for (int k = 0; k < d; ++k) s += xi[k] * xj[k];
                    g[j - jb] = s / d;
                }
                for (int64_t j = jb; j < je; ++j) {
                    double kk, tt;
                    layer_map(g[j - jb], q1[i], q2[j], n_dense, w, b, &kk, &tt);
                    if (get & GET_NNGP) out_nngp[i * ld + j] = kk;
                    if (get & GET_NTK) out_ntk[i * ld + j] = tt;
                    if (sym && j < i && j < n1 && i < n2) {
                        if (get & GET_NNGP) out_nngp[j * ld + i] = kk;
                        if (get & GET_NTK) out_ntk[j * ld + i] = tt;
                    }
                }
            }
        }
    free(q1);
    if (!sym) free(q2);
    return 0;
}

/* ---------------- blocked lower Cholesky (right-looking), float64, in place ---------------- */
#define NB 96

static int potf2(double* a, int64_t n, int64_t ld) {
    for (int64_t j = 0; j < n; ++j) {
        double d = a[j * ld + j];
        for (int64_t k = 0; k < j; ++k) d -= a[j * ld + k] * a[j * ld + k];
        if (!(d > 0.0)) return (int)(j + 1);
        d = sqrt(d);
        a[j * ld + j] = d;
        for (int64_t i = j + 1; i < n; ++i) {
            double s = a[i * ld + j];
            for (int64_t k = 0; k < j; ++k) s -= a[i * ld + k] * a[j * ld + k];
            a[i * ld + j] = s / d;
        }
    }
    return 0;
}

/* rows [r0,r1) of B (width nb) <- B * L^-T with L the nb x nb lower factor at l */
static void trsm_rows(double* bmat, int64_t r0, int64_t r1, int64_t ld, const double* l,
                      int64_t ldl, int64_t nb) {
    for (int64_t i = r0; i < r1; ++i) {
        double* row = bmat + i * ld;
        for (int64_t j = 0; j < nb; ++j) {
            double s = row[j];
            for (int64_t k = 0; k < j; ++k) s -= row[k] * l[j * ldl + k];
            row[j] = s / l[j * ldl + j];
        }
    }
}

int oracle_potrf_lower(double* a, int64_t n, int64_t ld) {
    int info = 0;
    double* bt = NULL;
    for (int64_t k = 0; k < n && !info; k += NB) {
        const int64_t nb = (n - k < NB) ? n - k : NB;
        info = potf2(a + k * ld + k, nb, ld);
        if (info) { info += (int)k; break; }
        const int64_t m = n - k - nb;  /* rows below */
        if (m <= 0) break;
        double* panel = a + (k + nb) * ld + k; /* m x nb */
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < m; r += 32)
            trsm_rows(panel, r, r + 32 < m ? r + 32 : m, ld, a + k * ld + k, ld, nb);
        /* packed transpose of the panel: bt[kk*m + j] = panel[j][kk] */
        bt = (double*)realloc(bt, sizeof(double) * (size_t)m * nb);
        if (!bt) return -1;
#pragma omp parallel for schedule(static)
        for (int64_t j = 0; j < m; ++j)
            for (int64_t kk = 0; kk < nb; ++kk) bt[kk * m + j] = panel[j * ld + kk];
        /* trailing update, lower tiles: C[i][j] -= sum_k P[i][k] P[j][k] */
        const int64_t TB = 64;
        const int64_t nt = (m + TB - 1) / TB;
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t tix = 0; tix < nt * nt; ++tix) {
            const int64_t ti = nt - 1 - tix / nt, tj = tix % nt; /* big rows first */
            if (tj > ti) continue;
            const int64_t i0 = ti * TB, i1 = i0 + TB < m ? i0 + TB : m;
            const int64_t j0 = tj * TB, j1 = j0 + TB < m ? j0 + TB : m;
            for (int64_t i = i0; i < i1; ++i) {
                double* c = a + (k + nb + i) * ld + (k + nb);
                const double* pi = panel + i * ld;
                const int64_t je = (j1 < i + 1) ? j1 : i + 1;
                for (int64_t kk = 0; kk < nb; ++kk) {
                    const double v = pi[kk];
                    const double* brow = bt + kk * m;
#pragma omp simd
                    for (int64_t j = j0; j < je; ++j) c[j] -= v * brow[j];
                }
            }
        }
    }
    free(bt);
    return info;
}

/* Row-oriented substitution, vectorised over the right-hand sides (contiguous in memory) and parallel over
 * column chunks; B is n x nrhs row-major (ldb), L lower n x n (ld).                                      */
#define RHS_CHUNK 32
static void forward_lower(const double* l, int64_t n, int64_t ld, double* bm, int64_t nrhs, int64_t ldb) {
#pragma omp parallel for schedule(static)
    for (int64_t c0 = 0; c0 < nrhs; c0 += RHS_CHUNK) {
        const int64_t w = c0 + RHS_CHUNK < nrhs ? RHS_CHUNK : nrhs - c0;
        for (int64_t i = 0; i < n; ++i) {
            double* bi = bm + i * ldb + c0;
            for (int64_t k = 0; k < i; ++k) {
                const double lik = l[i * ld + k];
                const double* bk = bm + k * ldb + c0;
#pragma omp simd
                for (int64_t c = 0; c < w; ++c) bi[c] -= lik * bk[c];
            }
            const double inv = 1.0 / l[i * ld + i];
            for (int64_t c = 0; c < w; ++c) bi[c] *= inv;
        }
    }
}

static void backward_lower_t(const double* l, int64_t n, int64_t ld, double* bm, int64_t nrhs, int64_t ldb) {
#pragma omp parallel for schedule(static)
    for (int64_t c0 = 0; c0 < nrhs; c0 += RHS_CHUNK) {
        const int64_t w = c0 + RHS_CHUNK < nrhs ? RHS_CHUNK : nrhs - c0;
        for (int64_t i = n - 1; i >= 0; --i) {
            double* bi = bm + i * ldb + c0;
            const double inv = 1.0 / l[i * ld + i];
            for (int64_t c = 0; c < w; ++c) bi[c] *= inv;
            for (int64_t k = 0; k < i; ++k) {  /* b[k] -= L[i][k] x[i] */
                const double lik = l[i * ld + k];
                double* bk = bm + k * ldb + c0;
#pragma omp simd
                for (int64_t c = 0; c < w; ++c) bk[c] -= lik * bi[c];
            }
        }
    }
}

/* solve L L^T X = B in place */
void oracle_potrs_lower(const double* l, int64_t n, int64_t ld, double* bm, int64_t nrhs, int64_t ldb) {
    forward_lower(l, n, ld, bm, nrhs, ldb);
    backward_lower_t(l, n, ld, bm, nrhs, ldb);
}

/* fit: builds K (get = 1 nngp | 2 ntk), A = K + reg I, L = chol(A) (in l_out, n x n), alpha.
 * y: n x ny row-major; alpha_out same shape.  stage_sec (may be NULL): [build, potrf, solve]. */
int oracle_fit(const double* x, const double* y, int64_t n, int d, int ny, int n_dense,
               const double* w, const double* b, int get, double diag_reg, int absolute,
               double* l_out, double* alpha_out, double* reg_out, double* stage_sec) {
#ifdef _OPENMP
    double t0 = omp_get_wtime();
#endif
    int rc = oracle_kernel_build(x, n, NULL, n, d, n_dense, w, b, get,
                                 get == GET_NNGP ? l_out : NULL, get == GET_NTK ? l_out : NULL, n);
    if (rc) return rc;
    double tr = 0.0;
    for (int64_t i = 0; i < n; ++i) tr += l_out[i * n + i];
    const double reg = absolute ? diag_reg : diag_reg * tr / (double)n;
    for (int64_t i = 0; i < n; ++i) l_out[i * n + i] += reg;
    if (reg_out) *reg_out = reg;
#ifdef _OPENMP
    double t1 = omp_get_wtime();
#endif
    int info = oracle_potrf_lower(l_out, n, n);
    if (info) return info;
#ifdef _OPENMP
    double t2 = omp_get_wtime();
#endif
    memcpy(alpha_out, y, sizeof(double) * (size_t)n * ny);
    oracle_potrs_lower(l_out, n, n, alpha_out, ny, ny);
#ifdef _OPENMP
    if (stage_sec) { stage_sec[0] = t1 - t0; stage_sec[1] = t2 - t1; stage_sec[2] = omp_get_wtime() - t2; }
#endif
    return 0;
}

/* predict (nngp): mean = K_td alpha (m x ny); var_i = K_tt,ii - |L^-1 k_i|^2 (cov_mode 1) or the
 * full m x m covariance (cov_mode 2).  train.py:157-158 (only diag(cov) is consumed, :180). */
int oracle_predict_nngp(const double* x, int64_t n, int d, int ny, int n_dense, const double* w,
                        const double* b, const double* l, const double* alpha,
                        const double* x_test, int64_t m, int cov_mode, double* mean,
                        double* var_or_cov) {
    double* ktd = (double*)malloc(sizeof(double) * (size_t)m * n);
    if (!ktd) return -1;
    oracle_kernel_build(x_test, m, x, n, d, n_dense, w, b, GET_NNGP, ktd, NULL, n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; ++i)
        for (int c = 0; c < ny; ++c) {
            double s = 0.0;
            for (int64_t k = 0; k < n; ++k) s += ktd[i * n + k] * alpha[k * ny + c];
            mean[i * ny + c] = s;
        }
    if (cov_mode == 0) { free(ktd); return 0; }
    /* V = L^-1 K_dt : n x m */
    double* v = (double*)malloc(sizeof(double) * (size_t)n * m);
    if (!v) { free(ktd); return -1; }
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n; ++k)
        for (int64_t i = 0; i < m; ++i) v[k * m + i] = ktd[i * n + k];
    forward_lower(l, n, n, v, m, m);
    if (cov_mode == 1) {
        double* qt = (double*)malloc(sizeof(double) * m);
        row_sqnorm(x_test, m, d, qt);
        for (int64_t i = 0; i < m; ++i) {
            double kk, tt, s = 0.0;
            layer_map(qt[i], qt[i], qt[i], n_dense, w, b, &kk, &tt);
            for (int64_t k = 0; k < n; ++k) s += v[k * m + i] * v[k * m + i];
            var_or_cov[i] = kk - s;
        }
        free(qt);
    } else {
        oracle_kernel_build(x_test, m, NULL, m, d, n_dense, w, b, GET_NNGP, var_or_cov, NULL, m);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < m; ++i)
            for (int64_t j = 0; j < m; ++j) {
                double s = 0.0;
                for (int64_t k = 0; k < n; ++k) s += v[k * m + i] * v[k * m + j];
                var_or_cov[i * m + j] -= s;
            }
    }
    free(v);
    free(ktd);
    return 0;
}
