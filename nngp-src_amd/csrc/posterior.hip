// Posterior helpers (a4): conversions between the float64 kernel matrices and the float32 operands of
// the MFMA solves, the predictive-variance reduction and the covariance epilogue.  Replaces the tail of
// predict_fn(x_test, get, compute_cov=True) (reference train.py:157-158; estimator.py:66-67):
//   mean = K_td alpha,  cov = K_tt - K_td (K_dd + reg I)^-1 K_dt,  of which only diag(cov) is consumed
//   downstream (train.py:180; estimator.py:55), so var_i = K_tt,ii - |L^-1 k_i|^2 is the default.
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

}  // namespace

int launch_factor_input(const double* k64, int64_t ld64, float* a32, int64_t ld32, int64_t n, int64_t np, double reg,
                        double pad_diag, hipStream_t s) {
    NNGP_REQUIRE(np % TB == 0 && np <= 65535LL * 1024, "factor_input: bad padded size %lld", (long long)np);
    // grid.y is limited to 65535 rows per launch
    for (int64_t r0 = 0; r0 < np; r0 += 65535) {
        const int64_t rows = (np - r0 < 65535) ? np - r0 : 65535;
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

}  // namespace nngp
