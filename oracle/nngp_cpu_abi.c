/* The C ABI of include/nngp_hip.h compiled for the HOST: float64, OpenMP, plain pointers are host pointers.
 * TEST INFRASTRUCTURE ONLY (lives under oracle/): the checker and the timed CPU baseline ("port"), never the product --
 * nngp-src_amd/ loads libnngp_hip.so and nothing else, and fails loudly without it.
 *
 * Same entry points, same argument meaning, same error behaviour (0 / negative code + nngp_last_error) as the HIP
 * library for the reference's path a1-a4:
 *   nngp_kernel_build / nngp_kernel_diag            kernel_fn of stax.serial(...)                    train.py:161-164
 *   nngp_model_create .. fit / set_train / build_rows / factor / solve / alpha / info
 *                                                   gradient_descent_mse_ensemble(diag_reg)          train.py:171-172
 *   nngp_model_predict                              predict_fn(x_test, get, compute_cov)             train.py:157-158
 * The arithmetic is the oracle's (nngp_oracle.c: Gram + arc-cosine map, blocked float64 potrf, substitution) -- the
 * algorithm class the reference runs on XLA-CPU.  `stream` arguments are ignored.  What has no CPU counterpart
 * (the incremental fit, the block-column pieces of the multi-GPU factorisation, serving mode, the MFMA building blocks,
 * RCCL, the kernel timer) is exported and answers -2 "not in the CPU build", so a binding written against the header
 * links against either library.  nngp_encoder_* is host code already: encoder.cpp is compiled into this library as is.
 */
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/nngp_hip.h"

/* nngp_oracle.c */
int oracle_kernel_build(const double* x1, int64_t n1, const double* x2, int64_t n2, int d, int n_dense, const double* w,
                        const double* b, int get, double* out_nngp, double* out_ntk, int64_t ld);
int oracle_potrf_lower(double* a, int64_t n, int64_t ld);
void oracle_potrs_lower(const double* l, int64_t n, int64_t ld, double* bm, int64_t nrhs, int64_t ldb);
int oracle_predict_nngp(const double* x, int64_t n, int d, int ny, int n_dense, const double* w, const double* b, const double* l,
                        const double* alpha, const double* x_test, int64_t m, int cov_mode, double* mean, double* var_or_cov);

static __thread char g_err[512];

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
void nngp_cpu_set_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }  /* for the C++ glue (encoder.cpp) */
#define REQUIRE(cond, ...) do { if (!(cond)) return fail(-2, __VA_ARGS__); } while (0)
#define NOT_HERE(name) return fail(-2, name ": not in the CPU build of the ABI")

int nngp_version(void) { return NNGP_ABI_VERSION; }
const char* nngp_last_error(void) { return g_err; }

struct nngp_model {
    int64_t n_cap, m_cap, n;
    int d, ny, get, absolute;
    nngp_arch arch;
    double diag_reg, reg, trace_mean, relres;
    double *x, *y, *k, *l, *alpha;  /* k: kernel of `get` [n, n]; l: its Cholesky factor (with reg) */
    int have_train, built, factored, solved;
};

static int check_arch(const nngp_arch* a) {
    REQUIRE(a != NULL && a->n_dense >= 1 && a->n_dense <= NNGP_MAX_DENSE, "bad architecture");
    return 0;
}

int nngp_kernel_build(const double* x1, int64_t n1, const double* x2, int64_t n2, int32_t d, const nngp_arch* arch,
                      int32_t out_dtype, void* out_nngp, void* out_ntk, int64_t ld, int64_t row_begin, int64_t row_end,
                      void* stream) {
    (void)stream;
    if (check_arch(arch)) return -2;
    REQUIRE(x1 != NULL && n1 > 0 && d > 0, "kernel_build: bad arguments");
    const int symmetric = (x2 == NULL);
    if (symmetric) { x2 = x1; n2 = n1; }
    REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= n1 && ld >= n2, "kernel_build: bad row range or ld");
    REQUIRE(out_dtype == NNGP_DTYPE_F32 || out_dtype == NNGP_DTYPE_F64, "kernel_build: bad out_dtype");
    const int64_t rows = row_end - row_begin;
    if (rows == 0) return 0;
    /* the whole symmetric kernel: the oracle's own symmetric path (x2 = NULL) */
    const double* b2 = (symmetric && row_begin == 0 && row_end == n1) ? NULL : x2;
    int rc = 0;
    if (out_dtype == NNGP_DTYPE_F64) {  /* straight into the caller's matrix (the oracle computes one `get` per call) */
        if (out_nngp) rc = oracle_kernel_build(x1 + row_begin * d, rows, b2, n2, d, arch->n_dense, arch->w_std, arch->b_std, NNGP_GET_NNGP,
                                               (double*)out_nngp + row_begin * ld, NULL, ld);
        if (!rc && out_ntk) rc = oracle_kernel_build(x1 + row_begin * d, rows, b2, n2, d, arch->n_dense, arch->w_std, arch->b_std, NNGP_GET_NTK,
                                                     NULL, (double*)out_ntk + row_begin * ld, ld);
        return rc ? fail(-1, "kernel_build: oracle_kernel_build failed (%d)", rc) : 0;
    }
    double* tmp = (double*)malloc(sizeof(double) * (size_t)rows * n2);
    if (!tmp) return fail(-1, "kernel_build: out of memory");
    for (int which = 0; which < 2 && !rc; ++which) {
        float* dst = (float*)(which ? out_ntk : out_nngp);
        if (!dst) continue;
        rc = oracle_kernel_build(x1 + row_begin * d, rows, b2, n2, d, arch->n_dense, arch->w_std, arch->b_std, which ? NNGP_GET_NTK : NNGP_GET_NNGP,
                                 which ? NULL : tmp, which ? tmp : NULL, n2);
        for (int64_t i = 0; i < rows && !rc; ++i)
            for (int64_t j = 0; j < n2; ++j) dst[(row_begin + i) * ld + j] = (float)tmp[i * n2 + j];
    }
    free(tmp);
    return rc ? fail(-1, "kernel_build: oracle_kernel_build failed (%d)", rc) : 0;
}

int nngp_kernel_diag(const double* x, int64_t n, int32_t d, const nngp_arch* arch, double* diag_nngp, double* diag_ntk,
                     void* stream) {
    (void)stream;
    if (check_arch(arch)) return -2;
    REQUIRE(x != NULL && n > 0 && d > 0, "kernel_diag: bad arguments");
    for (int64_t i = 0; i < n; ++i) {  /* K(x, x): the 1 x 1 kernel of the row with itself */
        double kn = 0.0, kt = 0.0;
        if (diag_nngp && oracle_kernel_build(x + i * d, 1, NULL, 1, d, arch->n_dense, arch->w_std, arch->b_std, NNGP_GET_NNGP, &kn, NULL, 1)) return fail(-1, "kernel_diag failed");
        if (diag_ntk && oracle_kernel_build(x + i * d, 1, NULL, 1, d, arch->n_dense, arch->w_std, arch->b_std, NNGP_GET_NTK, NULL, &kt, 1)) return fail(-1, "kernel_diag failed");
        if (diag_nngp) diag_nngp[i] = kn;
        if (diag_ntk) diag_ntk[i] = kt;
    }
    return 0;
}

int nngp_model_create(nngp_model** out, int64_t n_cap, int64_t m_cap, int32_t d, int32_t ny, const nngp_arch* arch,
                      int32_t get, double diag_reg, int32_t diag_reg_absolute_scale) {
    REQUIRE(out != NULL, "model_create: NULL out");
    *out = NULL;
    if (check_arch(arch)) return -2;
    REQUIRE(n_cap > 0 && d > 0 && ny > 0 && m_cap >= 0, "model_create: bad sizes");
    REQUIRE(get == NNGP_GET_NNGP || get == NNGP_GET_NTK, "model_create: get must be NNGP_GET_NNGP or NNGP_GET_NTK");
    REQUIRE(diag_reg >= 0.0, "model_create: negative diag_reg");
    nngp_model* m = (nngp_model*)calloc(1, sizeof(nngp_model));
    if (!m) return fail(-1, "model_create: out of memory");
    m->n_cap = n_cap; m->m_cap = m_cap; m->d = d; m->ny = ny; m->get = get; m->arch = *arch;
    m->diag_reg = diag_reg; m->absolute = diag_reg_absolute_scale != 0;
    m->x = (double*)malloc(sizeof(double) * (size_t)n_cap * d);
    m->y = (double*)malloc(sizeof(double) * (size_t)n_cap * ny);
    m->alpha = (double*)malloc(sizeof(double) * (size_t)n_cap * ny);
    m->k = (double*)malloc(sizeof(double) * (size_t)n_cap * n_cap);
    m->l = (double*)malloc(sizeof(double) * (size_t)n_cap * n_cap);
    if (!m->x || !m->y || !m->alpha || !m->k || !m->l) { nngp_model_destroy(m); return fail(-1, "model_create: out of memory"); }
    *out = m;
    return 0;
}

int nngp_model_destroy(nngp_model* m) {
    if (!m) return 0;
    free(m->x); free(m->y); free(m->alpha); free(m->k); free(m->l);
    free(m);
    return 0;
}

int nngp_model_set_train(nngp_model* m, const double* x, const double* y, int64_t n, void* stream) {
    (void)stream;
    REQUIRE(m != NULL && x != NULL && y != NULL, "set_train: NULL argument");
    REQUIRE(n > 0 && n <= m->n_cap, "set_train: %lld rows exceed n_cap = %lld", (long long)n, (long long)m->n_cap);
    memcpy(m->x, x, sizeof(double) * (size_t)n * m->d);
    memcpy(m->y, y, sizeof(double) * (size_t)n * m->ny);
    m->n = n;
    m->have_train = 1;
    m->built = m->factored = m->solved = 0;
    return 0;
}

int nngp_model_build_rows(nngp_model* m, int64_t row_begin, int64_t row_end, void* stream) {
    REQUIRE(m != NULL && m->have_train, "build_rows: call set_train first");
    REQUIRE(row_begin >= 0 && row_begin <= row_end && row_end <= m->n, "build_rows: bad row range");
    const int rc = nngp_kernel_build(m->x, m->n, NULL, m->n, m->d, &m->arch, NNGP_DTYPE_F64, m->get == NNGP_GET_NNGP ? m->k : NULL,
                                     m->get == NNGP_GET_NTK ? m->k : NULL, m->n, row_begin, row_end, stream);
    if (rc) return rc;
    m->built = 1;  /* like the HIP library: the caller vouches for the rows it did not build here (all-gather) */
    m->factored = m->solved = 0;
    return 0;
}

int nngp_model_factor(nngp_model* m, void* stream) {
    (void)stream;
    REQUIRE(m != NULL && m->built, "factor: build the kernel rows first");
    const int64_t n = m->n;
    double tr = 0.0;
    for (int64_t i = 0; i < n; ++i) tr += m->k[i * n + i];
    m->trace_mean = tr / (double)n;
    m->reg = m->absolute ? m->diag_reg : m->diag_reg * m->trace_mean;
    memcpy(m->l, m->k, sizeof(double) * (size_t)n * n);
    for (int64_t i = 0; i < n; ++i) m->l[i * n + i] += m->reg;
    const int info = oracle_potrf_lower(m->l, n, n);
    if (info) return fail(-3, "factor: the float64 Cholesky broke down at pivot %d", info);
    m->factored = 1;
    m->solved = 0;
    return 0;
}

double nngp_model_factor_shift(nngp_model* m) { return m ? m->reg : 0.0; }

int nngp_model_solve(nngp_model* m, int32_t max_iters, double tol, void* stream) {
    (void)max_iters; (void)tol; (void)stream;
    REQUIRE(m != NULL && m->factored, "solve: factor first");
    const int64_t n = m->n;
    memcpy(m->alpha, m->y, sizeof(double) * (size_t)n * m->ny);
    oracle_potrs_lower(m->l, n, n, m->alpha, m->ny, m->ny);
    /* |y - (K + reg I) alpha| / |y| */
    double rr = 0.0, yy = 0.0;
    for (int c = 0; c < m->ny; ++c) {
        double r2 = 0.0, y2 = 0.0;
#pragma omp parallel for reduction(+ : r2, y2) schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            double s = m->reg * m->alpha[i * m->ny + c];
            for (int64_t j = 0; j < n; ++j) s += m->k[i * n + j] * m->alpha[j * m->ny + c];
            const double r = m->y[i * m->ny + c] - s;
            r2 += r * r;
            y2 += m->y[i * m->ny + c] * m->y[i * m->ny + c];
        }
        if (y2 > 0.0 && r2 / y2 > rr) rr = r2 / y2;
        yy += y2;
    }
    m->relres = sqrt(rr);
    m->solved = 1;
    return 0;
}

int nngp_model_fit(nngp_model* m, const double* x, const double* y, int64_t n, void* stream) {
    int rc = nngp_model_set_train(m, x, y, n, stream);
    if (!rc) rc = nngp_model_build_rows(m, 0, n, stream);
    if (!rc) rc = nngp_model_factor(m, stream);
    if (!rc) rc = nngp_model_solve(m, 0, 0.0, stream);
    return rc;
}

int nngp_model_kernel_buffer(nngp_model* m, double** k64, int64_t* ld) {
    REQUIRE(m != NULL && m->have_train, "kernel_buffer: call set_train first");
    if (k64) *k64 = m->k;
    if (ld) *ld = m->n;
    return 0;
}

int nngp_model_info(nngp_model* m, nngp_fit_info* info) {
    REQUIRE(m != NULL && info != NULL, "model_info: NULL argument");
    info->reg = m->reg; info->trace_mean = m->trace_mean; info->rel_residual = m->relres;
    info->refine_iters = 0; info->clamped_pivots = 0; info->n = m->n; info->n_padded = m->n;
    return 0;
}

int nngp_model_alpha(nngp_model* m, double* alpha_out, void* stream) {
    (void)stream;
    REQUIRE(m != NULL && m->solved && alpha_out != NULL, "alpha: solve first");
    memcpy(alpha_out, m->alpha, sizeof(double) * (size_t)m->n * m->ny);
    return 0;
}

/* mean = K^get_td alpha;  NNGP cov = K_tt - K_td A^-1 K_dt;  NTK cov (train.py --kernel_type ntk, neural-tangents
 * gradient_descent_mse_ensemble with get='ntk'): K_tt + Z K_dd Z^T - (K_td Z^T + h.c.), Z = Theta_td A^-1, K = NNGP kernels. */
int nngp_model_predict(nngp_model* m, const double* x_test, int64_t mt, int32_t cov_mode, double* mean, double* var_or_cov,
                       void* stream) {
    REQUIRE(m != NULL && m->solved, "predict: fit first");
    REQUIRE(cov_mode == NNGP_COV_NONE || cov_mode == NNGP_COV_DIAG || cov_mode == NNGP_COV_FULL, "predict: bad cov_mode");
    REQUIRE(mean != NULL && (cov_mode == NNGP_COV_NONE || var_or_cov != NULL), "predict: NULL output");
    const int64_t n = m->n;
    if (x_test == NULL) { x_test = m->x; mt = n; }  /* x_test=None: the training rows */
    REQUIRE(mt > 0, "predict: no test rows");
    const int ntk = m->get == NNGP_GET_NTK;
    if (!ntk) {  /* the oracle's own routine: var = K_tt - |L^-1 k|^2, one forward substitution */
        const int rc0 = oracle_predict_nngp(m->x, n, m->d, m->ny, m->arch.n_dense, m->arch.w_std, m->arch.b_std, m->l, m->alpha, x_test,
                                            mt, cov_mode, mean, var_or_cov);
        return rc0 ? fail(-1, "predict: out of memory") : 0;
    }
    int rc = 0;
    double* ktd = (double*)malloc(sizeof(double) * (size_t)mt * n);        /* kernel of `get` */
    double* z = cov_mode ? (double*)malloc(sizeof(double) * (size_t)mt * n) : NULL;
    double* ktdn = (ntk && cov_mode) ? (double*)malloc(sizeof(double) * (size_t)mt * n) : NULL;  /* NNGP cross kernel */
    double* kdd = (ntk && cov_mode) ? (double*)malloc(sizeof(double) * (size_t)n * n) : NULL;
    double* w = (ntk && cov_mode) ? (double*)malloc(sizeof(double) * (size_t)mt * n) : NULL;      /* Z K_dd */
    double* ktt = NULL;
    if (!ktd || (cov_mode && !z) || (ntk && cov_mode && (!ktdn || !kdd || !w))) { rc = fail(-1, "predict: out of memory"); goto done; }
    rc = nngp_kernel_build(x_test, mt, m->x, n, m->d, &m->arch, NNGP_DTYPE_F64, ntk ? NULL : ktd, ntk ? ktd : NULL, n, 0, mt, stream);
    if (rc) goto done;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < mt; ++i)
        for (int c = 0; c < m->ny; ++c) {
            double s = 0.0;
            for (int64_t k = 0; k < n; ++k) s += ktd[i * n + k] * m->alpha[k * m->ny + c];
            mean[i * m->ny + c] = s;
        }
    if (cov_mode == NNGP_COV_NONE) goto done;
    /* Z^T = A^-1 K_dt: solve with mt right-hand sides stored as columns [n, mt] */
    {
        double* zt = (double*)malloc(sizeof(double) * (size_t)n * mt);
        if (!zt) { rc = fail(-1, "predict: out of memory"); goto done; }
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < n; ++k)
            for (int64_t i = 0; i < mt; ++i) zt[k * mt + i] = ktd[i * n + k];
        oracle_potrs_lower(m->l, n, n, zt, mt, mt);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < mt; ++i)
            for (int64_t k = 0; k < n; ++k) z[i * n + k] = zt[k * mt + i];
        free(zt);
    }
    if (ntk) {
        rc = nngp_kernel_build(x_test, mt, m->x, n, m->d, &m->arch, NNGP_DTYPE_F64, ktdn, NULL, n, 0, mt, stream);
        if (!rc) rc = nngp_kernel_build(m->x, n, NULL, n, m->d, &m->arch, NNGP_DTYPE_F64, kdd, NULL, n, 0, n, stream);
        if (rc) goto done;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < mt; ++i)
            for (int64_t j = 0; j < n; ++j) {
                double s = 0.0;
                for (int64_t k = 0; k < n; ++k) s += z[i * n + k] * kdd[k * n + j];
                w[i * n + j] = s;
            }
    }
    if (cov_mode == NNGP_COV_DIAG) {
        double* dn = (double*)malloc(sizeof(double) * (size_t)mt);
        if (!dn) { rc = fail(-1, "predict: out of memory"); goto done; }
        rc = nngp_kernel_diag(x_test, mt, m->d, &m->arch, dn, NULL, stream);
        for (int64_t i = 0; i < mt && !rc; ++i) {
            double s = 0.0;
            if (!ntk) for (int64_t k = 0; k < n; ++k) s -= z[i * n + k] * ktd[i * n + k];
            else for (int64_t k = 0; k < n; ++k) s += z[i * n + k] * (w[i * n + k] - 2.0 * ktdn[i * n + k]);
            var_or_cov[i] = dn[i] + s;
        }
        free(dn);
        goto done;
    }
    ktt = (double*)malloc(sizeof(double) * (size_t)mt * mt);
    if (!ktt) { rc = fail(-1, "predict: out of memory"); goto done; }
    rc = nngp_kernel_build(x_test, mt, NULL, mt, m->d, &m->arch, NNGP_DTYPE_F64, ktt, NULL, mt, 0, mt, stream);
    if (rc) goto done;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < mt; ++i)
        for (int64_t j = 0; j < mt; ++j) {
            double s = 0.0;
            if (!ntk) for (int64_t k = 0; k < n; ++k) s -= z[i * n + k] * ktd[j * n + k];
            else for (int64_t k = 0; k < n; ++k) s += w[i * n + k] * z[j * n + k] - ktdn[i * n + k] * z[j * n + k] - z[i * n + k] * ktdn[j * n + k];
            var_or_cov[i * mt + j] = ktt[i * mt + j] + s;
        }
done:
    free(ktd); free(z); free(ktdn); free(kdd); free(w); free(ktt);
    return rc;
}

int nngp_model_set_refine(nngp_model* m, int32_t sweeps) { (void)sweeps; REQUIRE(m != NULL, "set_refine: NULL model"); return 0; }  /* float64 throughout */
int nngp_model_cov_iters(nngp_model* m) { (void)m; return 0; }
int nngp_model_sweep_estimate(nngp_model* m, double* row_rel, double* var_rel) {
    (void)m;
    if (row_rel) *row_rel = -1.0;
    if (var_rel) *var_rel = -1.0;
    return 0;
}

/* ---- no CPU counterpart: exported so that a binding of the header links, answering -2 ---- */
int nngp_model_append(nngp_model* m, const double* x_new, const double* y_new, int64_t b, void* stream) { (void)m; (void)x_new; (void)y_new; (void)b; (void)stream; NOT_HERE("nngp_model_append"); }
int nngp_model_factor_begin(nngp_model* m, void* stream) { (void)m; (void)stream; NOT_HERE("nngp_model_factor_begin"); }
/* the row-sharded layout's entry points (float32 exchange, host-driven CG) are multi-GPU plumbing: not in the CPU build */
int nngp_model_factor_input_rows(nngp_model* m, int64_t row_begin, int64_t row_end, double shift_scale, void* stream) { (void)m; (void)row_begin; (void)row_end; (void)shift_scale; (void)stream; NOT_HERE("nngp_model_factor_input_rows"); }
int nngp_model_factor_input_complete(nngp_model* m) { (void)m; NOT_HERE("nngp_model_factor_input_complete"); }
int nngp_model_precond(nngp_model* m, const double* r, double* z, void* stream) { (void)m; (void)r; (void)z; (void)stream; NOT_HERE("nngp_model_precond"); }
int nngp_model_matvec_rows(nngp_model* m, const double* p, double* q, int64_t row_begin, int64_t row_end, void* stream) { (void)m; (void)p; (void)q; (void)row_begin; (void)row_end; (void)stream; NOT_HERE("nngp_model_matvec_rows"); }
int nngp_model_set_alpha(nngp_model* m, const double* alpha, int32_t iters, double rel_residual, void* stream) { (void)m; (void)alpha; (void)iters; (void)rel_residual; (void)stream; NOT_HERE("nngp_model_set_alpha"); }
int nngp_model_factor_panel(nngp_model* m, int64_t col0, int64_t width, void* stream) { (void)m; (void)col0; (void)width; (void)stream; NOT_HERE("nngp_model_factor_panel"); }
int nngp_model_factor_update(nngp_model* m, int64_t pc, int64_t pw, int64_t c0, int64_t w, void* stream) { (void)m; (void)pc; (void)pw; (void)c0; (void)w; (void)stream; NOT_HERE("nngp_model_factor_update"); }
int nngp_model_factor_update_cols(nngp_model* m, int64_t panel_col0, int64_t panel_width, const int64_t* cols, int32_t ncols, int64_t width, void* stream) { (void)m; (void)panel_col0; (void)panel_width; (void)cols; (void)ncols; (void)width; (void)stream; NOT_HERE("nngp_model_factor_update_cols"); }
int nngp_model_factor_end(nngp_model* m, void* stream) { (void)m; (void)stream; NOT_HERE("nngp_model_factor_end"); }
int nngp_model_factor_buffers(nngp_model* m, float** a32, int64_t* ld, float** dinv) { (void)m; (void)a32; (void)ld; (void)dinv; NOT_HERE("nngp_model_factor_buffers"); }
int nngp_model_prepare_serving(nngp_model* m, void* stream) { (void)m; (void)stream; NOT_HERE("nngp_model_prepare_serving"); }
int nngp_model_update_timer(nngp_model* m, int32_t enable) { (void)m; (void)enable; NOT_HERE("nngp_model_update_timer"); }
int nngp_model_residual_floor(nngp_model* m, double* ratio, int32_t* distrusted) { (void)m; if (ratio) *ratio = -1.0; if (distrusted) *distrusted = 0; return 0; }  /* float64 throughout */
int nngp_model_residual_timer(nngp_model* m, int32_t enable) { (void)m; (void)enable; NOT_HERE("nngp_model_residual_timer"); }
int nngp_model_residual_timer_read(nngp_model* m, int64_t* launches, double* ms_total, double* flops_total, double* int8_ops_total) { (void)m; (void)launches; (void)ms_total; (void)flops_total; (void)int8_ops_total; NOT_HERE("nngp_model_residual_timer_read"); }
int nngp_trsm_ticket_order(int32_t row_tiles, int32_t block_cols, int32_t tail_tiles, int32_t backward, int32_t workers, int32_t merged, int32_t* items, int64_t cap, int64_t* count) { (void)row_tiles; (void)block_cols; (void)tail_tiles; (void)backward; (void)workers; (void)merged; (void)items; (void)cap; (void)count; NOT_HERE("nngp_trsm_ticket_order"); }  /* the host build solves by plain substitution */
int nngp_trsm_ticket_queues(int32_t row_tiles, int32_t block_cols, int32_t tail_tiles, int32_t backward, int32_t workers, int32_t merged, int32_t queues, int32_t* items, int32_t* queue_of, int64_t cap, int64_t* count) { (void)row_tiles; (void)block_cols; (void)tail_tiles; (void)backward; (void)workers; (void)merged; (void)queues; (void)items; (void)queue_of; (void)cap; (void)count; NOT_HERE("nngp_trsm_ticket_queues"); }
int nngp_model_reserve(nngp_model* m, int64_t rows, int32_t cov_mode) { (void)rows; (void)cov_mode; return m != NULL ? 0 : -2; }  /* the host build allocates per call */
int64_t nngp_alloc_count(void) { return 0; }
int nngp_model_trsm_timer(nngp_model* m, int32_t enable) { (void)m; (void)enable; NOT_HERE("nngp_model_trsm_timer"); }
int nngp_model_trsm_timer_read(nngp_model* m, int64_t* solves, double* ms_total, double* flops_total) { (void)m; (void)solves; (void)ms_total; (void)flops_total; NOT_HERE("nngp_model_trsm_timer_read"); }
int nngp_model_update_timer_bytes(nngp_model* m, double* bytes_total) { (void)m; (void)bytes_total; NOT_HERE("nngp_model_update_timer_bytes"); }
int nngp_model_update_timer_read(nngp_model* m, int64_t* launches, double* ms_total, double* flops_total) { (void)m; (void)launches; (void)ms_total; (void)flops_total; NOT_HERE("nngp_model_update_timer_read"); }
int nngp_potrf_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, void* stream) { (void)a; (void)n; (void)ld; (void)dinv; (void)clamped; (void)stream; NOT_HERE("nngp_potrf_f32"); }
int nngp_gemm_nt_f32(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int64_t n, int64_t k, float alpha, float beta, int32_t lower_only, void* stream) { (void)c; (void)ldc; (void)a; (void)lda; (void)b; (void)ldb; (void)m; (void)n; (void)k; (void)alpha; (void)beta; (void)lower_only; (void)stream; NOT_HERE("nngp_gemm_nt_f32"); }
int nngp_gemm_nt_h3(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int64_t n, int64_t k, float alpha, float beta, float scale, int32_t lower_only, void* stream) { (void)c; (void)ldc; (void)a; (void)lda; (void)b; (void)ldb; (void)m; (void)n; (void)k; (void)alpha; (void)beta; (void)scale; (void)lower_only; (void)stream; NOT_HERE("nngp_gemm_nt_h3"); }
int nngp_gemm_nt_f64(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda, const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta, void* stream) { (void)c; (void)ldc; (void)cin; (void)ldcin; (void)a; (void)lda; (void)b; (void)ldb; (void)m; (void)n; (void)k; (void)alpha; (void)beta; (void)stream; NOT_HERE("nngp_gemm_nt_f64"); }
/* The sliced int8 product of csrc/gemm_i8s.hip restated on the host: the same row scales, the same balanced base-256 digits, the
 * exact integer sums of the plane pairs with ia + ib <= cut, the same float64 Horner combination -- every step is either exact or
 * one IEEE operation in a fixed order, so the device result is reproduced BIT FOR BIT (tests/test_gpu_parity.py).  O(pairs m n k). */
static double i8s_row_scale(const double* p, int64_t k) {
    double mx = 0.0;
    for (int64_t c = 0; c < k; ++c) mx = fmax(mx, fabs(p[c]));
    if (!(mx > 0.0 && mx < 1.0e300)) return 1.0;
    int e = 0;
    const double f = frexp(mx, &e);
    return ldexp(1.0, f <= 0.984375 ? e + 1 : e + 2);  /* |x| / scale <= 126/256: top digit <= 126 + carry */
}
static void i8s_digits(const double* p, int64_t k, int ns, double scale, int8_t* dig /* [ns][k] */) {
    const double inv = ldexp(1.0, 8 * ns) / scale;
    for (int64_t c = 0; c < k; ++c) {
        long long x = llrint(p[c] * inv);
        for (int s = ns - 1; s >= 1; --s) {
            const long long d = ((x + 128) & 255) - 128;
            x = (x - d) >> 8;
            dig[(int64_t)s * k + c] = (int8_t)d;
        }
        dig[c] = (int8_t)x;
    }
}
__attribute__((optimize("fp-contract=off")))
int nngp_gemm_nt_i8s(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda, const double* b,
                     int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta, int32_t slices_a, int32_t slices_b,
                     int32_t cut, void* stream) {
    (void)stream;
    REQUIRE(a != NULL && b != NULL && c != NULL && m > 0 && n > 0 && k > 0, "gemm_nt_i8s: bad arguments");
    REQUIRE(slices_a >= 2 && slices_a <= 7 && slices_b >= 2 && slices_b <= 7 && cut >= 0, "gemm_nt_i8s: 2..7 planes per operand");
    if (cut > slices_a + slices_b - 2) cut = slices_a + slices_b - 2;
    int8_t* da = (int8_t*)malloc((size_t)(m * slices_a * k));
    int8_t* db = (int8_t*)malloc((size_t)(n * slices_b * k));
    double* sa = (double*)malloc(sizeof(double) * (size_t)(m + n));
    if (da == NULL || db == NULL || sa == NULL) { free(da); free(db); free(sa); return fail(-2, "gemm_nt_i8s: out of memory"); }
    double* sb = sa + m;
    for (int64_t r = 0; r < m; ++r) { sa[r] = i8s_row_scale(a + r * lda, k); i8s_digits(a + r * lda, k, slices_a, sa[r], da + r * slices_a * k); }
    for (int64_t r = 0; r < n; ++r) { sb[r] = i8s_row_scale(b + r * ldb, k); i8s_digits(b + r * ldb, k, slices_b, sb[r], db + r * slices_b * k); }
    if (cin == NULL) { cin = c; ldcin = ldc; }
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < m; ++r) {
        for (int64_t col = 0; col < n; ++col) {
            double t = 0.0;
            for (int e = cut; e >= 0; --e) {
                long long sum = 0;
                for (int ia = 0; ia < slices_a; ++ia) {
                    const int ib = e - ia;
                    if (ib < 0 || ib >= slices_b) continue;
                    const int8_t* pa = da + (r * slices_a + ia) * k;
                    const int8_t* pb = db + (col * slices_b + ib) * k;
                    long long acc = 0;
                    for (int64_t kk = 0; kk < k; ++kk) acc += (int)pa[kk] * (int)pb[kk];
                    sum += acc;
                }
                t = t * 0.00390625 + (double)sum;
            }
            const double w = alpha * sa[r] * (1.0 / 65536.0);
            double v = w * sb[col] * t;
            if (beta != 0.0) v += beta * cin[r * ldcin + col];
            c[r * ldc + col] = v;
        }
    }
    free(da); free(db); free(sa);
    return 0;
}
/* pool scoring: the same keys and the same order as the device kernels (csrc/posterior.hip).  biased: the reference's draw
 * jax.random.choice(PRNGKey(seed), m, (count,), replace=False, p = score / sum) (active/ActiveLearner.py:50-53), restated in
 * nngp-src_amd/jaxrand.py: Threefry-2x32-20 on the counter pair (i, m + i) -> 52 mantissa bits -> Gumbel top-k. */
static void threefry2x32_host(uint32_t k0, uint32_t k1, uint32_t* px0, uint32_t* px1) {
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    static const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    uint32_t x0 = *px0 + ks[0], x1 = *px1 + ks[1];
    for (int g = 0; g < 5; ++g) {
        for (int r = 0; r < 4; ++r) {
            x0 += x1;
            x1 = (x1 << rot[g & 1][r]) | (x1 >> (32 - rot[g & 1][r]));
            x1 ^= x0;
        }
        x0 += ks[(g + 1) % 3];
        x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
    }
    *px0 = x0; *px1 = x1;
}
int nngp_pool_select(const double* mean, int64_t m, int32_t ny, const double* var, int64_t count, int32_t biased, uint64_t seed,
                     int64_t* indices, void* stream) {
    (void)stream;
    if (mean == NULL || var == NULL || indices == NULL || m <= 0 || ny < 1 || count < 0 || count > m) return fail(-2, "pool_select: bad arguments");
    double* key = (double*)malloc(sizeof(double) * (size_t)m);
    if (key == NULL) return fail(-2, "pool_select: out of memory");
    double mx = -INFINITY;
    for (int64_t i = 0; i < m; ++i) mx = fmax(mx, mean[i * ny]);
    for (int64_t i = 0; i < m; ++i) {
        const double sc = sqrt(fmax(var[i], 0.0)) / mx;
        double k = sc;
        if (biased) {
            uint32_t x0 = (uint32_t)i, x1 = (uint32_t)(m + i);
            threefry2x32_host((uint32_t)(seed >> 32), (uint32_t)seed, &x0, &x1);
            const uint64_t bits = ((((uint64_t)x0 << 32) | (uint64_t)x1) >> 12) | 0x3FF0000000000000ULL;
            const double tiny = 2.2250738585072014e-308;
            double u;
            memcpy(&u, &bits, sizeof(u));
            u = fmax(tiny, (u - 1.0) * (1.0 - tiny) + tiny);
            k = (sc > 0.0) ? log(sc) - log(-log(u)) : -INFINITY;
        }
        key[i] = (k == k) ? k : -INFINITY;
    }
    for (int64_t i = 0; i < m; ++i) {
        int64_t r = 0;
        for (int64_t j = 0; j < m; ++j) r += (key[j] > key[i] || (key[j] == key[i] && (biased ? j < i : j > i))) ? 1 : 0;
        if (r < count) indices[biased ? r : count - 1 - r] = i;
    }
    free(key);
    return 0;
}

int nngp_symv_f64(const double* a, int64_t lda, int64_t n, const double* x, double* y, double diag_add, void* stream) { (void)a; (void)lda; (void)n; (void)x; (void)y; (void)diag_add; (void)stream; NOT_HERE("nngp_symv_f64"); }
int nngp_model_apply_factor(nngp_model* m, float* b, int64_t rows, int32_t mode, void* stream) { (void)m; (void)b; (void)rows; (void)mode; (void)stream; NOT_HERE("nngp_model_apply_factor"); }
int nngp_trsm_rlt_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ldl, const float* dinv, int64_t n, void* stream) { (void)b; (void)ldb; (void)m; (void)l; (void)ldl; (void)dinv; (void)n; (void)stream; NOT_HERE("nngp_trsm_rlt_f32"); }
int nngp_comm_unique_id(void* id128) { (void)id128; NOT_HERE("nngp_comm_unique_id"); }
int nngp_comm_create(nngp_comm** out, const void* id128, int32_t world, int32_t rank) { (void)out; (void)id128; (void)world; (void)rank; NOT_HERE("nngp_comm_create"); }
int nngp_comm_destroy(nngp_comm* c) { (void)c; return 0; }
const char* nngp_comm_library(void) { return ""; }
int nngp_allgather_rows(void* k, int64_t n, int64_t ld, int32_t dtype, nngp_comm* c, void* stream) { (void)k; (void)n; (void)ld; (void)dtype; (void)c; (void)stream; NOT_HERE("nngp_allgather_rows"); }
int nngp_bcast(void* buf, int64_t count, int32_t dtype, int32_t root, nngp_comm* c, void* stream) { (void)buf; (void)count; (void)dtype; (void)root; (void)c; (void)stream; NOT_HERE("nngp_bcast"); }
