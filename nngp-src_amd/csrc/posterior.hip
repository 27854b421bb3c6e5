// Posterior helpers (a4): conversions between the float64 kernel matrices and the float32 operands of
// the MFMA solves, the predictive-variance reduction and the covariance epilogue.  Replaces the tail of
// predict_fn(x_test, get, compute_cov=True) (reference train.py:157-158; estimator.py:66-67):
//   mean = K_td alpha,  cov = K_tt - K_td (K_dd + reg I)^-1 K_dt,  of which only diag(cov) is consumed
//   downstream (train.py:180; estimator.py:55).  The orchestration (float32 solves, float64 correction sweep,
//   second-order variance formula, NTK covariance) is nngp_model_predict in api.hip.
#include "common.h"

namespace nngp {

namespace {

// A32 = float32(K64) + reg I on the lower triangle (tile granularity), identity in the padding.
// grid = (ceil(np/1024), np); each workgroup converts up to 1024 columns of one row.
__global__ __launch_bounds__(256) void k_factor_input(const double* __restrict__ k64, int64_t ld64,
                                                      float* __restrict__ a32, int64_t ld32, int64_t n, int64_t np,
                                                      double reg, double pad_diag, int64_t row0) {
    const int64_t row = row0 + blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    const int64_t row_tile_end = (row / TB + 1) * TB;  // columns < this belong to tiles on/below the diagonal
    if (c0 >= np || c0 >= row_tile_end) return;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t c = c0 + e;
        double t;
        if (row < n && c < n) {
            t = k64[row * ld64 + c];
            if (c == row) t += reg;
        } else {
            t = (c == row) ? pad_diag : 0.0;  // decoupled padding block, scaled like a real diagonal entry
        }
        v[e] = (float)t;
    }
    *reinterpret_cast<float4*>(a32 + row * ld32 + c0) = make_float4(v[0], v[1], v[2], v[3]);
}

// dst[rows_p x cols_p] (float32, zero padded) = float32(src[rows x cols])
__global__ __launch_bounds__(256) void k_convert_pad(const double* __restrict__ src, int64_t lds,
                                                     float* __restrict__ dst, int64_t ldd, int64_t rows,
                                                     int64_t cols, int64_t cols_p, int64_t row0) {
    const int64_t row = row0 + blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    if (c0 >= cols_p) return;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t c = c0 + e;
        v[e] = (row < rows && c < cols) ? (float)src[row * lds + c] : 0.0f;
    }
    *reinterpret_cast<float4*>(dst + row * ldd + c0) = make_float4(v[0], v[1], v[2], v[3]);
}

// out[i] = base[i] - sum_j v[i][j]^2 (float64 accumulation), one workgroup per row
__global__ __launch_bounds__(256) void k_row_sqsum(const float* __restrict__ v, int64_t ld, int64_t cols,
                                                   const double* __restrict__ base, double* __restrict__ out) {
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    const float4* vr = reinterpret_cast<const float4*>(v + row * ld);
    double s = 0.0;
    for (int64_t j = threadIdx.x; j < cols / 4; j += 256) {
        const float4 t = vr[j];
        s += (double)t.x * t.x + (double)t.y * t.y + (double)t.z * t.z + (double)t.w * t.w;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = base[row] - (red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_cov_finish(const double* __restrict__ ktt, int64_t ldk,
                                                    const float* __restrict__ vvt, int64_t ldv, int64_t m,
                                                    double* __restrict__ cov) {
    const int64_t row = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < m) cov[row * m + c] = ktt[row * ldk + c] - (double)vvt[row * ldv + c];
}

__global__ void k_strided_copy(const double* src, int64_t incs, double* dst, int64_t incd, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i * incd] = src[i * incs];
}

// zero the padding of an [np, np] float64 matrix whose valid part is [n, n]
__global__ __launch_bounds__(256) void k_zero_pad_f64(double* __restrict__ a, int64_t ld, int64_t n, int64_t np,
                                                      int64_t row0) {
    const int64_t row = row0 + blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row < n) {
        if (c < np - n) a[row * ld + n + c] = 0.0;  // columns [n, np) of a valid row
    } else if (c < np) {
        a[row * ld + c] = 0.0;                       // a whole padding row
    }
}

// dst (f64) = [dst +] float64(src (f32)), [rows, cols]
__global__ __launch_bounds__(256) void k_f32_to_f64_mat(const float* __restrict__ src, int64_t lds, double* __restrict__ dst,
                                                        int64_t ldd, int64_t cols, int accumulate) {
    const int64_t row = blockIdx.y;
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (c >= cols) return;
    const float2 v = *reinterpret_cast<const float2*>(src + row * lds + c);
    double2* d = reinterpret_cast<double2*>(dst + row * ldd + c);
    double2 o = accumulate ? *d : make_double2(0.0, 0.0);
    o.x += (double)v.x;
    o.y += (double)v.y;
    *d = o;
}

// r = a * r + b * k  (elementwise, [rows, cols] with a common leading dimension)
__global__ __launch_bounds__(256) void k_axpby_mat(double* __restrict__ r, double a, const double* __restrict__ k,
                                                   double b, int64_t ld, int64_t cols) {
    const int64_t row = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    r[row * ld + c] = fma(b, k[row * ld + c], a * r[row * ld + c]);
}

// out[i] = base[i] + sign * sum_j z[i][j] * (kscale * k[i][j] + r[i][j])   (k, r or base may be NULL)
template <bool STAT>
__global__ __launch_bounds__(256) void k_rowdot_f64(const double* __restrict__ z, const double* __restrict__ k,
                                                    double kscale, const double* __restrict__ r, int64_t ld,
                                                    int64_t cols, const double* __restrict__ base, double sign,
                                                    double* __restrict__ out, double* __restrict__ zstat) {
    // STAT: also zstat[2 row] = |z_row|_2^2 and zstat[2 row + 1] = max |z_row| (the int8 residual's guard reads them: one pass over z less)
    __shared__ double red[12];
    const int64_t row = blockIdx.x;
    double s = 0.0, s2 = 0.0, mx = 0.0;
    for (int64_t j = threadIdx.x; j < cols; j += 256) {
        double g = 0.0;
        if (k) g = kscale * k[row * ld + j];
        if (r) g += r[row * ld + j];
        const double zv = z[row * ld + j];
        s = fma(zv, g, s);
        if (STAT) {
            s2 = fma(zv, zv, s2);
            mx = fmax(mx, fabs(zv));
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        if (STAT) {
            s2 += __shfl_down(s2, off);
            mx = fmax(mx, __shfl_down(mx, off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = s;
        if (STAT) { red[4 + (threadIdx.x >> 6)] = s2; red[8 + (threadIdx.x >> 6)] = mx; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[row] = (base ? base[row] : 0.0) + sign * (red[0] + red[1] + red[2] + red[3]);
        if (STAT) {
            zstat[2 * row] = (red[4] + red[5]) + (red[6] + red[7]);
            zstat[2 * row + 1] = fmax(fmax(red[8], red[9]), fmax(red[10], red[11]));
        }
    }
}

// ---- per-row preconditioned CG on [rows, cols] blocks: every row is an independent right-hand side of the same SPD ----
// system (rows_pcg_continue in api.hip drives these; RowsPcg in common.h holds the per-row scalars).
__device__ __forceinline__ double block_sum_256(double s, double* red) {
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Decides which rows need more than the fixed correction sweeps and sets their stopping tolerance.
//   mode 0 (NNGP): delta = z.r is the first-order term the second-order variance formula cancels; by Cauchy-Schwarz
//     the remaining error e^T A e is at least delta^2 / (k^T A^-1 k), so delta^2 > thr * q * var flags the row.
//     tol = the decrease of e^T A e per CG step below which the row stops: 1e-8 of the variance estimate.
//   mode 1 (NTK, no second-order formula): every row runs; tol relative to the energy q = z.k.
__global__ void k_rows_prepare(const double* __restrict__ delta, const double* __restrict__ ktt, const double* __restrict__ var,
                               const double* __restrict__ q, int mode, double thr, int64_t rows, double* __restrict__ tol,
                               int32_t* __restrict__ flagged) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    if (mode == 1) {
        tol[i] = 1e-13 * fabs(q[i]);
        return;
    }
    const double v = var[i], prior = ktt[i], e = prior - v;
    tol[i] = fmax(1e-8 * fmax(v, 0.0), 1e-15 * prior);
    const bool bad = !(v > 0.0) || !(delta[i] * delta[i] <= thr * fmax(e, 0.0) * v);
    if (bad) atomicAdd(flagged, 1);
}

// NTK rows before they go on by CG: the stopping tolerance on the per-step decrease of e^T A e.  The variance is first
// order in the row's error, |d var| ~ gamma * sqrt(e^T A e); the second sweep measured gamma = |dv| / sqrt(e1) (its
// correction had energy ~ e1 and moved the variance by dv).  tol = the energy at which gamma * sqrt(energy) = tau var,
// kept within [1e-30, 1e-13] of the row's energy z . k.  tau = 1e-10: gamma is measured along ONE direction and the
// later CG errors lie elsewhere -- against an 80-bit referee on the same kernel matrices (tests/
// test_gpu_extended_precision.py, N = 279 .. 907, d = 2 .. 3, cond ~ 1e8) tau = 1e-8 left 2.5e-5 .. 6.5e-5 in the variance,
// 1e-12 leaves 5e-6 .. 4e-5 for one or two more iterations; below that the float64 kernel entries themselves decide
// (last-bit differences in them move these variances by 1e-4).  zk_tol: z . k in, tol out.
__global__ void k_rows_prepare_ntk(const double* __restrict__ e1, const double* __restrict__ dv, double* zk_tol,
                                   const double* __restrict__ var, int64_t vstride, int64_t rows, double tau) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const double q = fabs(zk_tol[i]), v = tau * fabs(var[i * vstride]), d = dv[i];
    double t = (d != 0.0 && e1[i] > 0.0) ? v * v * (e1[i] / (d * d)) : 1e-13 * q;
    if (!(t <= 1e-13 * q)) t = 1e-13 * q;
    if (!(t >= 1e-30 * q)) t = 1e-30 * q;
    zk_tol[i] = t;
}

// rho_new = r . s (s = M^-1 r in float32); beta = rho_new / rho_old; a row whose rho is not positive is finished
__global__ __launch_bounds__(256) void k_rows_rho(const double* __restrict__ r, const float* __restrict__ s32, int64_t ld,
                                                  int64_t cols, int first, double* __restrict__ rho, double* __restrict__ coef,
                                                  int32_t* __restrict__ state) {
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < cols; j += 256) acc = fma(r[row * ld + j], (double)s32[row * ld + j], acc);
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        if (first) state[row] = 0;
        double beta = 0.0;
        if (state[row] >= 0) {
            if (!(acc > 0.0) || !isfinite(acc)) state[row] = -1;
            else if (!first && rho[row] > 0.0) beta = acc / rho[row];
        }
        rho[row] = acc;
        coef[row] = beta;
    }
}

// out[row] = r[row] . s32[row]  (s32 = M^-1 r: the energy e^T A e of the row's error, to the accuracy of M as a solver)
__global__ __launch_bounds__(256) void k_rows_energy(const double* __restrict__ r, const float* __restrict__ s32, int64_t ld,
                                                     int64_t cols, double* __restrict__ out) {
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < cols; j += 256) acc = fma(r[row * ld + j], (double)s32[row * ld + j], acc);
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[row] = acc;
}

// out[row] = 2 d32[row] . (w[row] + kscale * k[row]): the change of the NTK variance K_tt + z.(K z - 2 k) under the row's
// last correction d (first order; w = K z).  k may be NULL (w already holds K z - k).
__global__ __launch_bounds__(256) void k_rows_dvar(const float* __restrict__ d32, const double* __restrict__ w,
                                                   const double* __restrict__ k, double kscale, int64_t ld, int64_t cols,
                                                   double* __restrict__ out) {
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < cols; j += 256) {
        double g = w[row * ld + j];
        if (k) g = fma(kscale, k[row * ld + j], g);
        acc = fma((double)d32[row * ld + j], g, acc);
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[row] = 2.0 * acc;
}

// What two stationary sweeps leave in the NTK variance, predicted from what they removed.  e0, e1: the error energies
// r . M^-1 r before the first and the second correction, rho = sqrt(e1 / e0) the contraction per sweep in the energy
// norm; dv: the change of the variance under the second correction (k_rows_dvar).  The variance is first order in the
// error of the row z, so what is left is ~ rho / (1 - rho) * |dv|.
// out[0] = max over rows of sqrt(e1 * (e1 / e0) / |z . k|) (relative energy-norm error of z),
// out[1] = max over rows of rho / (1 - rho) * |dv| / |var| (predicted relative variance error).  One workgroup.
__global__ __launch_bounds__(256) void k_sweep_estimate(const double* __restrict__ e0, const double* __restrict__ e1,
                                                        const double* __restrict__ zk, const double* __restrict__ dv,
                                                        const double* __restrict__ var, int64_t vstride, int64_t rows,
                                                        double* __restrict__ out) {
    __shared__ double red[2][256];
    double w0 = 0.0, w1 = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += 256) {
        const double a = e0[i], b = e1[i], c = fabs(zk[i]);
        if (a > 0.0 && b > 0.0 && c > 0.0) {
            const double rho = sqrt(b / a);
            const double rel = sqrt(b * (b / a) / c);
            const double left = rho < 1.0 ? rho / (1.0 - rho) * fabs(dv[i]) / fabs(var[i * vstride]) : INFINITY;
            w0 = (rel > w0 || !isfinite(rel)) ? rel : w0;
            w1 = (left > w1 || !isfinite(left)) ? left : w1;
        }
    }
    red[0][threadIdx.x] = w0;
    red[1][threadIdx.x] = w1;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            for (int q = 0; q < 2; ++q) {
                const double o = red[q][threadIdx.x + w];
                if (o > red[q][threadIdx.x] || !isfinite(o)) red[q][threadIdx.x] = o;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 2) out[threadIdx.x] = red[threadIdx.x][0];
}

// p = s + beta p   (finished rows: p = 0)
__global__ __launch_bounds__(256) void k_rows_update_p(double* __restrict__ p, const float* __restrict__ s32, int64_t ld,
                                                       int64_t cols, const double* __restrict__ coef,
                                                       const int32_t* __restrict__ state) {
    const int64_t row = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const bool live = state[row] >= 0;
    p[row * ld + c] = live ? fma(coef[row], p[row * ld + c], (double)s32[row * ld + c]) : 0.0;
}

// alpha = rho / (p . q); gain = alpha rho = decrease of e^T A e in this step.  Two consecutive gains below tol finish
// the row (after this update).  `live` counts the rows that go on.
__global__ __launch_bounds__(256) void k_rows_alpha(const double* __restrict__ p, const double* __restrict__ q, int64_t ld,
                                                    int64_t cols, const double* __restrict__ rho, const double* __restrict__ tol,
                                                    double* __restrict__ coef, int32_t* __restrict__ state,
                                                    int32_t* __restrict__ live) {
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < cols; j += 256) acc = fma(p[row * ld + j], q[row * ld + j], acc);
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        double alpha = 0.0;
        int32_t st = state[row];
        if (st >= 0) {
            if (acc > 0.0 && isfinite(acc)) {
                alpha = rho[row] / acc;
                st = (alpha * rho[row] <= tol[row]) ? st + 1 : 0;
                if (st >= 2) st = -1;
            } else {
                st = -1;
            }
            state[row] = st;
            if (st >= 0) atomicAdd(live, 1);
        }
        coef[row] = alpha;
    }
}

// z += alpha p;  r -= alpha q
__global__ __launch_bounds__(256) void k_rows_axpy2(double* __restrict__ z, double* __restrict__ r, const double* __restrict__ p,
                                                    const double* __restrict__ q, int64_t ld, int64_t cols,
                                                    const double* __restrict__ coef) {
    const int64_t row = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const double a = coef[row];
    if (a == 0.0) return;
    z[row * ld + c] = fma(a, p[row * ld + c], z[row * ld + c]);
    r[row * ld + c] = fma(-a, q[row * ld + c], r[row * ld + c]);
}

// rows [r0, r0 + rows) of the identity, zero-padded to [rows_p, cols]
__global__ __launch_bounds__(256) void k_identity_rows(double* __restrict__ e, int64_t ld, int64_t cols, int64_t r0, int64_t rows) {
    const int64_t row = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < cols) e[row * ld + c] = (row < rows && c == r0 + row) ? 1.0 : 0.0;
}

// a <- (a + a^T) / 2 in place, [n, n]; one thread per pair (i > j)
__global__ __launch_bounds__(256) void k_symmetrize_f64(double* __restrict__ a, int64_t ld, int64_t n, int64_t row0) {
    const int64_t i = row0 + blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= i || i >= n) return;
    const double v = 0.5 * (a[i * ld + j] + a[j * ld + i]);
    a[i * ld + j] = v;
    a[j * ld + i] = v;
}

// C[r][j] = alpha * sum_i A[r][i] B[j][i] + beta * Cin[r][j] for a handful of rows r < M (M <= 8): serving-mode products
// Z = K_td X and Z K for one or a few queries.  The rows of B (N x N float64) are streamed once, four per workgroup --
// HBM-bound, where the 128-row MFMA GEMM would spend 128/M times the flops on padding.  A comes from L2.
template <int M>
__global__ __launch_bounds__(256) void k_skinny_nt_f64(double* __restrict__ C, int64_t ldc, const double* __restrict__ Cin,
                                                       int64_t ldcin, const double* __restrict__ A, int64_t lda,
                                                       const double* __restrict__ B, int64_t ldb, int64_t n, int64_t k,
                                                       double alpha, double beta) {
    constexpr int JB = 4;
    const int64_t j0 = (int64_t)blockIdx.x * JB;
    double acc[M][JB];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) acc[r][jj] = 0.0;
    const double* brow[JB];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) brow[jj] = B + (j0 + jj < n ? j0 + jj : n - 1) * ldb;
    for (int64_t i = (int64_t)threadIdx.x * 2; i < k; i += 512) {
        double2 b[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) b[jj] = *reinterpret_cast<const double2*>(brow[jj] + i);
#pragma unroll
        for (int r = 0; r < M; ++r) {
            const double2 a = *reinterpret_cast<const double2*>(A + r * lda + i);
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) acc[r][jj] = fma(a.x, b[jj].x, fma(a.y, b[jj].y, acc[r][jj]));
        }
    }
    __shared__ double red[4][M * JB];
#pragma unroll
    for (int r = 0; r < M; ++r)
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            double v = acc[r][jj];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][r * JB + jj] = v;
        }
    __syncthreads();
    if (threadIdx.x < M * JB) {
        const int r = threadIdx.x / JB, jj = threadIdx.x % JB;
        if (j0 + jj < n) {
            double v = alpha * (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
            if (Cin != nullptr) v = fma(beta, Cin[r * ldcin + j0 + jj], v);
            C[r * ldc + j0 + jj] = v;
        }
    }
}

// dst[m, m] (dense) = (src + src^T) / 2 of src[m, m] (leading dimension lds).  The symmetric part is where the
// first-order error of the refined solve cancels for off-diagonal covariance entries as well.
__global__ __launch_bounds__(256) void k_copy_mat_f64(const double* __restrict__ src, int64_t lds, double* __restrict__ dst,
                                                      int64_t m) {
    const int64_t row = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < m) dst[row * m + c] = 0.5 * (src[row * lds + c] + src[c * lds + row]);
}

// dst = src^T for a square float32 matrix (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void k_transpose_f32(const float* __restrict__ src, int64_t lds, float* __restrict__ dst,
                                                      int64_t ldd) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = src[(r0 + ty + 8 * k) * lds + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[(c0 + ty + 8 * k) * ldd + r0 + tx] = tile[tx][ty + 8 * k];
}

}  // namespace

// K[j, i] = K[i, j] for i in [n0, n1), j in [0, n0): the columns of the old rows for newly appended rows (32x32 tiles)
__global__ __launch_bounds__(256) void k_mirror_rows_f64(double* __restrict__ k, int64_t ld, int64_t n0, int64_t n1) {
    __shared__ double tile[32][33];
    const int64_t i0 = n0 + (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = i0 + ty + 8 * e, j = j0 + tx;
        tile[ty + 8 * e][tx] = (i < n1 && j < n0) ? k[i * ld + j] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t j = j0 + ty + 8 * e, i = i0 + tx;
        if (i < n1 && j < n0) k[j * ld + i] = tile[tx][ty + 8 * e];
    }
}

int launch_mirror_rows_f64(double* k, int64_t ld, int64_t n0, int64_t n1, hipStream_t s) {
    if (n1 <= n0 || n0 <= 0) return 0;
    hipLaunchKernelGGL(k_mirror_rows_f64, dim3((unsigned)((n0 + 31) / 32), (unsigned)((n1 - n0 + 31) / 32)), dim3(256), 0, s, k,
                       ld, n0, n1);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_factor_input(const double* k64, int64_t ld64, float* a32, int64_t ld32, int64_t n, int64_t np, double reg,
                        double pad_diag, hipStream_t s, int64_t row_begin, int64_t row_end) {
    NNGP_REQUIRE(np % TB == 0 && np <= 65535LL * 1024 && row_begin >= 0, "factor_input: bad padded size %lld", (long long)np);
    if (row_end < 0 || row_end > np) row_end = np;
    // grid.y is limited to 65535 rows per launch
    for (int64_t r0 = row_begin; r0 < row_end; r0 += 65535) {
        const int64_t rows = (row_end - r0 < 65535) ? row_end - r0 : 65535;
        hipLaunchKernelGGL(k_factor_input, dim3((unsigned)((np + 1023) / 1024), (unsigned)rows), dim3(256), 0, s,
                           k64, ld64, a32, ld32, n, np, reg, pad_diag, r0);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_convert_f64_f32(const double* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int64_t cols,
                           int64_t rows_p, int64_t cols_p, hipStream_t s) {
    NNGP_REQUIRE(cols_p % 4 == 0 && ldd % 4 == 0, "convert: padded width must be a multiple of 4");
    for (int64_t r0 = 0; r0 < rows_p; r0 += 65535) {
        const int64_t nr = (rows_p - r0 < 65535) ? rows_p - r0 : 65535;
        hipLaunchKernelGGL(k_convert_pad, dim3((unsigned)((cols_p + 1023) / 1024), (unsigned)nr), dim3(256), 0, s,
                           src, lds, dst, ldd, rows, cols, cols_p, r0);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_row_sqsum_f32(const float* v, int64_t ld, int64_t rows, int64_t cols, const double* base, double* out,
                         hipStream_t s) {
    if (rows <= 0) return 0;
    NNGP_REQUIRE(cols % 4 == 0 && ld % 4 == 0, "row_sqsum: width must be a multiple of 4");
    hipLaunchKernelGGL(k_row_sqsum, dim3((unsigned)rows), dim3(256), 0, s, v, ld, cols, base, out);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_cov_finish(const double* ktt, int64_t ldk, const float* vvt, int64_t ldv, int64_t m, double* cov,
                      hipStream_t s) {
    if (m <= 0) return 0;
    NNGP_REQUIRE(m <= 65535, "cov_finish: at most 65535 test rows per call");
    hipLaunchKernelGGL(k_cov_finish, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, s, ktt, ldk, vvt,
                       ldv, m, cov);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_strided_copy_f64(const double* src, int64_t incs, double* dst, int64_t incd, int64_t n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_strided_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, incs, dst, incd, n);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_zero_pad_f64(double* a, int64_t ld, int64_t n, int64_t np, hipStream_t s) {
    if (np == n) return 0;
    for (int64_t r0 = 0; r0 < np; r0 += 65535) {
        const int64_t nr = (np - r0 < 65535) ? np - r0 : 65535;
        hipLaunchKernelGGL(k_zero_pad_f64, dim3((unsigned)((np + 255) / 256), (unsigned)nr), dim3(256), 0, s, a, ld, n, np, r0);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_f32_to_f64_mat(const float* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols,
                          bool accumulate, hipStream_t s) {
    NNGP_REQUIRE(cols % 2 == 0 && lds % 2 == 0 && ldd % 2 == 0, "f32_to_f64_mat: even widths required");
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = (rows - r0 < 65535) ? rows - r0 : 65535;
        hipLaunchKernelGGL(k_f32_to_f64_mat, dim3((unsigned)((cols / 2 + 255) / 256), (unsigned)nr), dim3(256), 0, s,
                           src + r0 * lds, lds, dst + r0 * ldd, ldd, cols, accumulate ? 1 : 0);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_axpby_mat(double* r, double a, const double* k, double b, int64_t ld, int64_t rows, int64_t cols,
                     hipStream_t s) {
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = (rows - r0 < 65535) ? rows - r0 : 65535;
        hipLaunchKernelGGL(k_axpby_mat, dim3((unsigned)((cols + 255) / 256), (unsigned)nr), dim3(256), 0, s, r + r0 * ld, a,
                           k + r0 * ld, b, ld, cols);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rowdot_f64(const double* z, const double* k, double kscale, const double* r, int64_t ld, int64_t rows,
                      int64_t cols, const double* base, double sign, double* out, hipStream_t s, double* zstat) {
    if (rows <= 0) return 0;
    if (zstat != nullptr)
        hipLaunchKernelGGL((k_rowdot_f64<true>), dim3((unsigned)rows), dim3(256), 0, s, z, k, kscale, r, ld, cols, base, sign, out, zstat);
    else
        hipLaunchKernelGGL((k_rowdot_f64<false>), dim3((unsigned)rows), dim3(256), 0, s, z, k, kscale, r, ld, cols, base, sign, out, zstat);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rows_prepare(const double* delta, const double* ktt, const double* var, const double* q, int mode, double thr,
                        int64_t rows, double* tol, int32_t* flagged, hipStream_t s) {
    if (rows <= 0) return 0;
    NNGP_HIP_CHECK(hipMemsetAsync(flagged, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_rows_prepare, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, delta, ktt, var, q, mode, thr, rows,
                       tol, flagged);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rows_rho(const double* r, const float* s32, int64_t ld, int64_t rows, int64_t cols, bool first, RowsPcg& w,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_rows_rho, dim3((unsigned)rows), dim3(256), 0, s, r, s32, ld, cols, first ? 1 : 0, w.rho, w.coef, w.state);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rows_energy(const double* r, const float* s32, int64_t ld, int64_t rows, int64_t cols, double* out, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_rows_energy, dim3((unsigned)rows), dim3(256), 0, s, r, s32, ld, cols, out);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rows_prepare_ntk(const double* e1, const double* dv, double* zk_tol, const double* var, int64_t vstride,
                            int64_t rows, double tau, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_rows_prepare_ntk, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, e1, dv, zk_tol, var, vstride, rows, tau);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rows_dvar(const float* d32, const double* w, const double* k, double kscale, int64_t ld, int64_t rows,
                     int64_t cols, double* out, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_rows_dvar, dim3((unsigned)rows), dim3(256), 0, s, d32, w, k, kscale, ld, cols, out);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_sweep_estimate(const double* e0, const double* e1, const double* zk, const double* dv, const double* var,
                          int64_t vstride, int64_t rows, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sweep_estimate, dim3(1), dim3(256), 0, s, e0, e1, zk, dv, var, vstride, rows, out);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rows_update_p(double* p, const float* s32, int64_t ld, int64_t rows, int64_t cols, RowsPcg& w, hipStream_t s) {
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = (rows - r0 < 65535) ? rows - r0 : 65535;
        hipLaunchKernelGGL(k_rows_update_p, dim3((unsigned)((cols + 255) / 256), (unsigned)nr), dim3(256), 0, s, p + r0 * ld,
                           s32 + r0 * ld, ld, cols, w.coef + r0, w.state + r0);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rows_alpha(const double* p, const double* q, int64_t ld, int64_t rows, int64_t cols, RowsPcg& w, hipStream_t s) {
    NNGP_HIP_CHECK(hipMemsetAsync(w.live, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_rows_alpha, dim3((unsigned)rows), dim3(256), 0, s, p, q, ld, cols, w.rho, w.tol, w.coef, w.state, w.live);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_rows_axpy2(double* z, double* r, const double* p, const double* q, int64_t ld, int64_t rows, int64_t cols,
                      RowsPcg& w, hipStream_t s) {
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = (rows - r0 < 65535) ? rows - r0 : 65535;
        hipLaunchKernelGGL(k_rows_axpy2, dim3((unsigned)((cols + 255) / 256), (unsigned)nr), dim3(256), 0, s, z + r0 * ld,
                           r + r0 * ld, p + r0 * ld, q + r0 * ld, ld, cols, w.coef + r0);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_identity_rows(double* e, int64_t ld, int64_t cols, int64_t r0, int64_t rows, int64_t rows_p, hipStream_t s) {
    NNGP_REQUIRE(rows_p <= 65535, "identity_rows: at most 65535 rows");
    hipLaunchKernelGGL(k_identity_rows, dim3((unsigned)((cols + 255) / 256), (unsigned)rows_p), dim3(256), 0, s, e, ld, cols, r0, rows);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_symmetrize_f64(double* a, int64_t ld, int64_t n, hipStream_t s) {
    for (int64_t r0 = 0; r0 < n; r0 += 65535) {
        const int64_t nr = (n - r0 < 65535) ? n - r0 : 65535;
        hipLaunchKernelGGL(k_symmetrize_f64, dim3((unsigned)((r0 + nr + 255) / 256), (unsigned)nr), dim3(256), 0, s, a, ld, n, r0);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// rows of A / C / Cin up to the next power of two >= m must be addressable (they are: the buffers are padded to 128 rows)
int launch_skinny_nt_f64(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda,
                         const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta, hipStream_t s) {
    NNGP_REQUIRE(m >= 1 && m <= 8 && n > 0 && k > 0 && k % 2 == 0 && lda % 2 == 0 && ldb % 2 == 0, "skinny_nt_f64: bad shape");
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    if (m == 1) hipLaunchKernelGGL(k_skinny_nt_f64<1>, grid, block, 0, s, c, ldc, cin, ldcin, a, lda, b, ldb, n, k, alpha, beta);
    else if (m == 2) hipLaunchKernelGGL(k_skinny_nt_f64<2>, grid, block, 0, s, c, ldc, cin, ldcin, a, lda, b, ldb, n, k, alpha, beta);
    else if (m <= 4) hipLaunchKernelGGL(k_skinny_nt_f64<4>, grid, block, 0, s, c, ldc, cin, ldcin, a, lda, b, ldb, n, k, alpha, beta);
    else hipLaunchKernelGGL(k_skinny_nt_f64<8>, grid, block, 0, s, c, ldc, cin, ldcin, a, lda, b, ldb, n, k, alpha, beta);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_copy_mat_f64(const double* src, int64_t lds, double* dst, int64_t m, hipStream_t s) {
    if (m <= 0) return 0;
    NNGP_REQUIRE(m <= 65535, "copy_mat: at most 65535 rows");
    hipLaunchKernelGGL(k_copy_mat_f64, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, s, src, lds, dst, m);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_transpose_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t n, hipStream_t s) {
    NNGP_REQUIRE(n % 32 == 0 && n / 32 <= 65535, "transpose: n must be a multiple of 32");
    hipLaunchKernelGGL(k_transpose_f32, dim3((unsigned)(n / 32), (unsigned)(n / 32)), dim3(256), 0, s, src, lds, dst, ldd);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}


// ---- pool scoring of the active-learning loop (include/nngp_hip.h: nngp_pool_select; reference ActiveLearner.py:43-55) ----
namespace {
__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t seed, uint64_t idx) {  // nngp-src_amd/synth.py: splitmix64
    uint64_t z = (idx + 1ULL) * 0x9E3779B97F4A7C15ULL + seed * 0xD1B54A32D192ED03ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Threefry-2x32, 20 rounds (Salmon et al. 2011; the generator behind jax.random's PRNGKey): one block
__device__ __forceinline__ void threefry2x32_dev(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    x0 += ks[0];
    x1 += ks[1];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x0 += x1;
            x1 = (x1 << rot[g & 1][r]) | (x1 >> (32 - rot[g & 1][r]));
            x1 ^= x0;
        }
        x0 += ks[(g + 1) % 3];
        x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
    }
}

// key_i = score_i (top-k), or log(score_i) + Gumbel(u_i): the score-proportional draw without replacement as the reference makes
// it -- jax.random.choice(PRNGKey(seed), m, (count,), replace=False, p = score / sum(score)) = argsort(-gumbel - log p)[:count] with
// u_i from Threefry on the counter pair (i, m + i), restated in nngp-src_amd/jaxrand.py (log p and log score differ by a constant:
// the order is the same).  One workgroup.
__global__ __launch_bounds__(1024) void k_pool_keys(const double* __restrict__ mean, int64_t m, int ny, const double* __restrict__ var,
                                                    int biased, uint64_t seed, double* __restrict__ key) {
    __shared__ double red[16];
    double mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < m; i += 1024) mx = fmax(mx, mean[i * ny]);
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < 16; ++w) mx = fmax(mx, red[w]);
    for (int64_t i = threadIdx.x; i < m; i += 1024) {
        const double sc = sqrt(fmax(var[i], 0.0)) / mx;
        double k = sc;
        if (biased) {
            uint32_t x0 = (uint32_t)i, x1 = (uint32_t)(m + i);
            threefry2x32_dev((uint32_t)(seed >> 32), (uint32_t)seed, x0, x1);
            const uint64_t bits = ((((uint64_t)x0 << 32) | (uint64_t)x1) >> 12) | 0x3FF0000000000000ULL;
            const double tiny = 2.2250738585072014e-308;
            double u = __longlong_as_double((long long)bits) - 1.0;
            u = fmax(tiny, u * (1.0 - tiny) + tiny);
            k = (sc > 0.0) ? log(sc) - log(-log(u)) : -INFINITY;
        }
        key[i] = (k == k) ? k : -INFINITY;  // NaN never wins
    }
}

// rank by counting: r_i = #{j : key_j > key_i, or key_j == key_i and j beats i on the tie rule}; the `count` best go out
__global__ __launch_bounds__(256) void k_pool_rank(const double* __restrict__ key, int64_t m, int64_t count, int biased,
                                                   int64_t* __restrict__ out) {
    __shared__ double tile[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double ki = (i < m) ? key[i] : 0.0;
    int64_t r = 0;
    for (int64_t j0 = 0; j0 < m; j0 += 256) {
        __syncthreads();
        tile[threadIdx.x] = (j0 + threadIdx.x < m) ? key[j0 + threadIdx.x] : -INFINITY;
        __syncthreads();
        const int lim = (m - j0 < 256) ? (int)(m - j0) : 256;
        for (int t = 0; t < lim; ++t) {
            const double kj = tile[t];
            // equal keys: the stable ascending argsort of the reference puts the larger index last -- the better place of
            // np.argsort(score)[-count:], the worse of the biased draw's argsort(-gumbel - log p)[:count]
            r += (kj > ki || (kj == ki && (biased ? j0 + t < i : j0 + t > i))) ? 1 : 0;
        }
    }
    if (i < m && r < count) out[biased ? r : count - 1 - r] = i;
}
}  // namespace

int launch_pool_select(const double* mean, int64_t m, int ny, const double* var, int64_t count, int biased, uint64_t seed,
                       double* key_ws, int64_t* indices, hipStream_t s) {
    hipLaunchKernelGGL(k_pool_keys, dim3(1), dim3(1024), 0, s, mean, m, ny, var, biased, seed, key_ws);
    hipLaunchKernelGGL(k_pool_rank, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, key_ws, m, count, biased, indices);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nngp
