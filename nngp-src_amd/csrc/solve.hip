// GP solve alpha = (K + reg I)^-1 y -- replaces cho_solve(C, Y_train) inside
// nt.predict.gradient_descent_mse_ensemble (reference train.py:171-172).
//
// The reference solves in float64 with a float64 factor.  Here the factor is float32 (MFMA Cholesky,
// potrf.hip) and cond(K + reg I) ~ 1e7, so the float32 factor is used only as a preconditioner:
// conjugate gradients on the float64 kernel matrix (HBM-bound GEMV) converge to the float64 answer in
// a handful of iterations.  Triangular solves with the float32 factor walk the
// 128-blocks using the inverted diagonal blocks produced by the Cholesky leaf.
#include "common.h"

namespace nngp {

namespace {

// ---- forward: x_j = Linv_jj b_j ; b[i] -= L[i, blk j] x_j for rows below ----
// grid.x = 1 + number of 256-row chunks below block j; every workgroup recomputes x_j (128x128 matvec
// from L2-resident data), workgroup 0 stores it, workgroups >= 1 update their chunk of b.
__global__ __launch_bounds__(256) void k_trsv_fwd_step(const float* __restrict__ L, int64_t ld,
                                                       const float* __restrict__ dinv_j, int64_t jrow,
                                                       int64_t n, float* __restrict__ b, float* __restrict__ x) {
    __shared__ float bj[TB];
    __shared__ float xj[TB];
    const int tid = threadIdx.x;
    if (tid < TB) bj[tid] = b[jrow + tid];
    __syncthreads();
    {
        // two threads per output row: halves of the 128-long dot product
        const int r = tid >> 1, half = tid & 1;
        const float4* dr = reinterpret_cast<const float4*>(dinv_j + r * TB + half * 64);
        float s = 0.0f;
#pragma unroll 4
        for (int c = 0; c < 16; ++c) {
            const float4 v = dr[c];
            const float* bp = bj + half * 64 + c * 4;
            s = fmaf(v.x, bp[0], s);
            s = fmaf(v.y, bp[1], s);
            s = fmaf(v.z, bp[2], s);
            s = fmaf(v.w, bp[3], s);
        }
        s += __shfl_xor(s, 1);
        if (half == 0) xj[r] = s;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < TB) x[jrow + tid] = xj[tid];
        return;
    }
    // update rows: 32 lanes x float4 cover the 128 columns of one row; a wave does 2 rows per pass
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane >> 5, l32 = lane & 31;
    const float4 xv = make_float4(xj[l32 * 4], xj[l32 * 4 + 1], xj[l32 * 4 + 2], xj[l32 * 4 + 3]);
    const int64_t chunk0 = jrow + TB + (int64_t)(blockIdx.x - 1) * 256;
    for (int it = 0; it < 32; ++it) {
        const int64_t row = chunk0 + wave * 64 + it * 2 + sub;
        float s = 0.0f;
        if (row < n) {
            const float4 lv = *reinterpret_cast<const float4*>(L + row * ld + jrow + l32 * 4);
            s = lv.x * xv.x + lv.y * xv.y + lv.z * xv.z + lv.w * xv.w;
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (row < n && l32 == 0) b[row] -= s;
    }
}

// ---- backward: x_j = Linv_jj^T b_j ; b[c] -= sum_r L[jrow + r][c] x_j[r] for columns c < jrow ----
__global__ __launch_bounds__(256) void k_trsv_bwd_step(const float* __restrict__ L, int64_t ld,
                                                       const float* __restrict__ dinv_j, int64_t jrow,
                                                       float* __restrict__ b, float* __restrict__ x) {
    __shared__ float bj[TB];
    __shared__ float xj[TB];
    __shared__ float part[2][TB];
    const int tid = threadIdx.x;
    if (tid < TB) bj[tid] = b[jrow + tid];
    __syncthreads();
    {
        // x_j[c] = sum_r dinv[r][c] b_j[r]; thread (c, half) sums 64 rows, coalesced over c
        const int c = tid & 127, half = tid >> 7;
        float s = 0.0f;
#pragma unroll 8
        for (int r = 0; r < 64; ++r) s = fmaf(dinv_j[(half * 64 + r) * TB + c], bj[half * 64 + r], s);
        part[half][c] = s;
    }
    __syncthreads();
    if (tid < TB) xj[tid] = part[0][tid] + part[1][tid];
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < TB) x[jrow + tid] = xj[tid];
        return;
    }
    const int64_t c = (int64_t)(blockIdx.x - 1) * 256 + tid;
    if (c >= jrow) return;
    const float* lp = L + jrow * ld + c;
    float s = 0.0f;
#pragma unroll 8
    for (int r = 0; r < TB; ++r) s = fmaf(lp[(int64_t)r * ld], xj[r], s);
    b[c] -= s;
}

// ---- float64 GEMV: y[i*incy] = sum_j A[i][j] x[j*incx] + diag_add * x[i*incx] ----
__global__ __launch_bounds__(256) void k_gemv_f64(const double* __restrict__ A, int64_t lda, int64_t rows,
                                                  int64_t cols, const double* __restrict__ x, int64_t incx,
                                                  double* __restrict__ y, int64_t incy, double diag_add) {
    __shared__ double red[4];
    const int64_t row = blockIdx.x;
    const double* ar = A + row * lda;
    double s = 0.0;
    if (incx == 1 && (lda & 1) == 0 && ((uintptr_t)A & 15) == 0) {
        const int64_t c2 = cols >> 1;
        const double2* a2 = reinterpret_cast<const double2*>(ar);
        for (int64_t j = threadIdx.x; j < c2; j += 256) {
            const double2 v = a2[j];
            s = fma(v.x, x[2 * j], s);
            s = fma(v.y, x[2 * j + 1], s);
        }
        if ((cols & 1) && threadIdx.x == 0) s = fma(ar[cols - 1], x[cols - 1], s);
    } else {
        for (int64_t j = threadIdx.x; j < cols; j += 256) s = fma(ar[j], x[j * incx], s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0] + red[1] + red[2] + red[3];
        if (diag_add != 0.0) t = fma(diag_add, x[row * incx], t);
        y[row * incy] = t;
    }
}

// ---- small float64 vector kernels with device-resident scalars ----
// scal[0] = rz, scal[1] = pAp, scal[2] = rz_new, scal[3] = |r|^2, scal[4] = |b|^2
__global__ __launch_bounds__(1024) void k_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                              double* out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s = fma(a[i], b[i], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        *out = t;
    }
}

__global__ void k_pcg_update_xr(double* x, double* r, const double* p, const double* q, int64_t n, const double* scal) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double a = scal[1] != 0.0 ? scal[0] / scal[1] : 0.0;
    x[i] = fma(a, p[i], x[i]);
    r[i] = fma(-a, q[i], r[i]);
}

__global__ void k_pcg_update_p(double* p, const double* z, int64_t n, double* scal, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (first || scal[0] == 0.0) {
        p[i] = z[i];
    } else {
        p[i] = fma(scal[2] / scal[0], p[i], z[i]);
    }
}

__global__ void k_scal_shift(double* scal) { scal[0] = scal[2]; }

__global__ void k_f64_to_f32(const double* src, float* dst, int64_t n, int64_t np) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < np) dst[i] = i < n ? (float)src[i] : 0.0f;
}

__global__ void k_f32_to_f64(const float* src, double* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

inline unsigned blocks256(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// Solves L x = b.  b (length np, float32) is destroyed; x receives the solution.  n = np (padded size).
int trsv_forward_f32(const float* l, int64_t ld, const float* dinv, int64_t n, float* b, float* x, hipStream_t s) {
    const int64_t nb = n / TB;
    for (int64_t j = 0; j < nb; ++j) {
        const int64_t below = n - (j + 1) * TB;
        const unsigned grid = 1u + (unsigned)((below + 255) / 256);
        hipLaunchKernelGGL(k_trsv_fwd_step, dim3(grid), dim3(256), 0, s, l, ld, dinv + j * TB * TB, j * TB, n, b, x);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Solves L^T x = b.
int trsv_backward_f32(const float* l, int64_t ld, const float* dinv, int64_t n, float* b, float* x, hipStream_t s) {
    const int64_t nb = n / TB;
    for (int64_t j = nb - 1; j >= 0; --j) {
        const int64_t left = j * TB;
        const unsigned grid = 1u + (unsigned)((left + 255) / 256);
        hipLaunchKernelGGL(k_trsv_bwd_step, dim3(grid), dim3(256), 0, s, l, ld, dinv + j * TB * TB, j * TB, b, x);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemv_f64(const double* a, int64_t lda, int64_t rows, int64_t cols, const double* x, int64_t incx,
                    double* y, int64_t incy, double diag_add, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_gemv_f64, dim3((unsigned)rows), dim3(256), 0, s, a, lda, rows, cols, x, incx, y, incy,
                       diag_add);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Preconditioned CG for (K + reg I) x = b in float64; M^-1 = (L L^T)^-1 with the float32 factor.
int pcg_solve(const double* k64, int64_t ld, int64_t n, double reg, const float* l32, int64_t ld32,
              const float* dinv, int64_t np, const double* bcol, double* xcol, PcgWork& w, int max_iters,
              double tol, int* iters_out, double* relres_out, hipStream_t s) {
    auto precond = [&](const double* rin, double* zout) -> int {
        hipLaunchKernelGGL(k_f64_to_f32, dim3(blocks256(np)), dim3(256), 0, s, rin, w.f32a, n, np);
        NNGP_TRY(trsv_forward_f32(l32, ld32, dinv, np, w.f32a, w.f32b, s));
        NNGP_TRY(trsv_backward_f32(l32, ld32, dinv, np, w.f32b, w.f32c, s));
        hipLaunchKernelGGL(k_f32_to_f64, dim3(blocks256(n)), dim3(256), 0, s, w.f32c, zout, n);
        return 0;
    };
    // x = 0, r = b
    NNGP_HIP_CHECK(hipMemsetAsync(xcol, 0, sizeof(double) * n, s));
    NNGP_HIP_CHECK(hipMemcpyAsync(w.r, bcol, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, s, bcol, bcol, n, w.scal + 4);
    NNGP_HIP_CHECK(hipMemcpyAsync(w.host_scal + 4, w.scal + 4, sizeof(double), hipMemcpyDeviceToHost, s));
    NNGP_HIP_CHECK(hipStreamSynchronize(s));
    const double bnorm2 = w.host_scal[4];
    int iters = 0;
    double relres = 0.0;
    if (bnorm2 > 0.0) {
        relres = 1.0;
        for (int it = 0; it < max_iters; ++it) {
            NNGP_TRY(precond(w.r, w.z));
            hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, s, w.r, w.z, n, w.scal + 2);  // rz_new
            hipLaunchKernelGGL(k_pcg_update_p, dim3(blocks256(n)), dim3(256), 0, s, w.p, w.z, n, w.scal, it == 0);
            hipLaunchKernelGGL(k_scal_shift, dim3(1), dim3(1), 0, s, w.scal);  // rz = rz_new
            NNGP_TRY(launch_gemv_f64(k64, ld, n, n, w.p, 1, w.q, 1, reg, s));
            hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, s, w.p, w.q, n, w.scal + 1);  // pAp
            hipLaunchKernelGGL(k_pcg_update_xr, dim3(blocks256(n)), dim3(256), 0, s, xcol, w.r, w.p, w.q, n, w.scal);
            hipLaunchKernelGGL(k_dot, dim3(1), dim3(1024), 0, s, w.r, w.r, n, w.scal + 3);
            NNGP_HIP_CHECK(hipMemcpyAsync(w.host_scal + 3, w.scal + 3, sizeof(double), hipMemcpyDeviceToHost, s));
            NNGP_HIP_CHECK(hipStreamSynchronize(s));
            iters = it + 1;
            relres = sqrt(w.host_scal[3] / bnorm2);
            if (!(relres == relres)) {  // NaN: the preconditioner is unusable
                set_error("pcg_solve: residual became NaN at iteration %d", iters);
                return -3;
            }
            if (relres <= tol) break;
        }
    }
    NNGP_HIP_CHECK(hipGetLastError());
    if (iters_out) *iters_out = iters;
    if (relres_out) *relres_out = relres;
    return 0;
}

}  // namespace nngp
