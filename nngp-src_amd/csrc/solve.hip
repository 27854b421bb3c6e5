// GP solve alpha = (K + reg I)^-1 y -- replaces cho_solve(C, Y_train) inside
// nt.predict.gradient_descent_mse_ensemble (reference train.py:171-172).
//
// The reference solves in float64 with a float64 factor.  Here the factor is float32 (MFMA Cholesky,
// potrf.hip) and cond(K + reg I) ~ 1e7, so the float32 factor is used only as a preconditioner:
// conjugate gradients on the float64 kernel matrix (HBM-bound GEMV) converge to the float64 answer in
// a handful of iterations (5-6 at N = 32768).  Triangular solves with the float32 factor walk
// block columns of 1024 with explicitly inverted diagonal blocks (built once per fit by a batched MFMA solve
// against the identity): vectors go through GEMV kernels (CG preconditioner), row blocks of right-hand sides
// through GEMMs (posterior covariance).
#include "common.h"

namespace nngp {

namespace {

// ---- float32 GEMV building blocks of the blocked triangular solves ----
// y[r] = dot(A[r,:], x)  (sub = 0)   or   y[r] -= dot(A[r,:], x)  (sub = 1); one wave per row, cols % 4 == 0
__global__ __launch_bounds__(256) void k_gemv_n_f32(const float* __restrict__ A, int64_t lda, int64_t rows,
                                                    int64_t cols, const float* __restrict__ x, float* y, int sub) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* ar = reinterpret_cast<const float4*>(A + row * lda);
    const float4* xv = reinterpret_cast<const float4*>(x);
    float s = 0.0f;
    for (int64_t c = lane; c < cols / 4; c += 64) {
        const float4 a = ar[c], v = xv[c];
        s = fmaf(a.x, v.x, s);
        s = fmaf(a.y, v.y, s);
        s = fmaf(a.z, v.z, s);
        s = fmaf(a.w, v.w, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) y[row] = sub ? y[row] - s : s;
}

// partial[chunk][c] = sum_{r in chunk} A[r][c] x[r];  grid = (ceil(cols/256), rows/128), rows % 128 == 0
__global__ __launch_bounds__(256) void k_gemv_t_partial_f32(const float* __restrict__ A, int64_t lda, int64_t cols,
                                                            const float* __restrict__ x, float* __restrict__ partial,
                                                            int64_t ldp) {
    __shared__ float xs[128];
    const int64_t r0 = (int64_t)blockIdx.y * 128;
    if (threadIdx.x < 128) xs[threadIdx.x] = x[r0 + threadIdx.x];
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const float* ap = A + r0 * lda + c;
    float s = 0.0f;
#pragma unroll 16
    for (int r = 0; r < 128; ++r) s = fmaf(ap[(int64_t)r * lda], xs[r], s);
    partial[(int64_t)blockIdx.y * ldp + c] = s;
}

// y[c] -= sum_chunk partial[chunk][c]  (fixed summation order: results are run-to-run reproducible)
__global__ __launch_bounds__(256) void k_sub_partials_f32(float* __restrict__ y, const float* __restrict__ partial,
                                                          int64_t ldp, int nchunk, int64_t cols) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.0f;
    for (int k = 0; k < nchunk; ++k) s += partial[(int64_t)k * ldp + c];
    y[c] -= s;
}

// blocks of `count` identity matrices (bs x bs, ld = bs)
__global__ __launch_bounds__(256) void k_set_identity_blocks(float* __restrict__ t, int64_t bs, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t e = i % (bs * bs);
    t[i] = (e / bs == e % bs) ? 1.0f : 0.0f;
}

// dst_J = src_J^T for `count` blocks of bs x bs (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void k_transpose_blocks(const float* __restrict__ src, float* __restrict__ dst,
                                                          int64_t bs) {
    __shared__ float tile[32][33];
    const int64_t base = (int64_t)blockIdx.z * bs * bs;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = src[base + (r0 + ty + 8 * k) * bs + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[base + (c0 + ty + 8 * k) * bs + r0 + tx] = tile[tx][ty + 8 * k];
}

// ---- float64 symmetric matrix-vector product from the lower triangle (the CG's A p: 8 N^2 / 2 bytes instead of 8 N^2) ----
// One workgroup per 128 x 128 tile (I >= J) of the lower triangle: u = A_IJ x_J goes to part[J][rows of I] and, off the
// diagonal, v = A_IJ^T x_I to part[I][rows of J] -- every (block, rows) slot of `part` is written by exactly one tile, and
// k_symv_reduce sums a row's slots in a fixed order.  Wave w owns rows 32 w .. 32 w + 31 of the tile, lane l columns l and 64 + l.
// At most 64 registers (__launch_bounds__(256, 8)): the kernel then fits into the wave slot that the posterior's float64 GEMM
// leaves free on every SIMD and runs beside it instead of waiting for one of its workgroups to retire.
__global__ __launch_bounds__(256, 8) void k_symv_tiles_f64(const double* __restrict__ A, int64_t lda, int64_t n,
                                                        const double* __restrict__ x, double* __restrict__ part, int64_t np) {
    __shared__ double xs[2][128];
    __shared__ double us[128];
    __shared__ double vs[4][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t p = blockIdx.x;
    int64_t I = (int64_t)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > p) --I;
    while ((I + 1) * (I + 2) / 2 <= p) ++I;
    const int64_t J = p - I * (I + 1) / 2;
    if (tid < 128) {
        const int64_t ri = I * 128 + tid, rj = J * 128 + tid;
        xs[0][tid] = ri < n ? x[ri] : 0.0;
        xs[1][tid] = rj < n ? x[rj] : 0.0;
    }
    __syncthreads();
    const int64_t c0 = J * 128 + lane, c1 = c0 + 64;
    const bool ok0 = c0 < n, ok1 = c1 < n, diag = (I == J);
    const double xj0 = xs[1][lane], xj1 = xs[1][64 + lane];
    double v0 = 0.0, v1 = 0.0;
#pragma unroll 1
    for (int r4 = 0; r4 < 32; r4 += 4) {
        double a0[4], a1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // four rows' loads in flight per wave (more would not fit 64 registers)
            const int64_t r = I * 128 + wave * 32 + r4 + e;
            const double* ar = A + r * lda;
            // a diagonal tile is used from its lower triangle only, like the matrix as a whole
            a0[e] = (r < n && ok0 && !(diag && c0 > r)) ? ar[c0] : 0.0;
            a1[e] = (r < n && ok1 && !(diag && c1 > r)) ? ar[c1] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t r = I * 128 + wave * 32 + r4 + e;
            const double xi = xs[0][wave * 32 + r4 + e];
            v0 = fma((diag && c0 >= r) ? 0.0 : a0[e], xi, v0);  // transposed part: strictly below the diagonal there
            v1 = fma((diag && c1 >= r) ? 0.0 : a1[e], xi, v1);
            double sdot = fma(a0[e], xj0, a1[e] * xj1);
            for (int off = 32; off > 0; off >>= 1) sdot += __shfl_xor(sdot, off);
            if (lane == 0) us[wave * 32 + r4 + e] = sdot;
        }
    }
    vs[wave][lane] = v0;
    vs[wave][64 + lane] = v1;
    __syncthreads();
    if (tid < 128) {
        const double vt = (vs[0][tid] + vs[1][tid]) + (vs[2][tid] + vs[3][tid]);
        part[J * np + I * 128 + tid] = diag ? us[tid] + vt : us[tid];
        if (!diag) part[I * np + J * 128 + tid] = vt;
    }
}

__global__ __launch_bounds__(256) void k_symv_reduce_f64(const double* __restrict__ part, int64_t nblk, int64_t np, int64_t n,
                                                         const double* __restrict__ x, double diag_add, double* __restrict__ y) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    double s0 = 0.0, s1 = 0.0;
    int64_t c = 0;
    for (; c + 1 < nblk; c += 2) {
        s0 += part[c * np + r];
        s1 += part[(c + 1) * np + r];
    }
    if (c < nblk) s0 += part[c * np + r];
    y[r] = fma(diag_add, x[r], s0 + s1);
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
// a . b in a fixed summation order.  kDotBlocks workgroups of 256 threads (NOT one of 1024: beside the posterior's float64
// GEMM, whose two workgroups per CU leave one wave slot of <= 64 registers per SIMD, a 16-wave workgroup waited for a
// GEMM workgroup to retire -- 0.7 ms on average, up to 4.8, per dot product, three per CG iteration); the last workgroup to
// finish (wrapping atomic counter) adds the partial sums in block order.  part / counter: per-model scratch (PcgWork).
constexpr int kDotBlocks = 32;
__global__ __launch_bounds__(256, 8) void k_dot(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                double* out, double* part, unsigned* counter) {
    __shared__ double red[4];
    __shared__ bool last;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kDotBlocks * 256) s = fma(a[i], b[i], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last = atomicInc(counter, kDotBlocks - 1) == kDotBlocks - 1;  // wraps to 0: ready for the next call
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        double t = 0.0;
        for (int w = 0; w < kDotBlocks; ++w) t += __hip_atomic_load(&part[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// B_J <- B_J L_JJ^-T for `batch` diagonal blocks at once (same recursion as trsm_rlt_f32, batched strides).
static int trsm_rlt_batched(float* b, int64_t ldb, int64_t sb, int64_t m, const float* l, int64_t ldl, int64_t sl,
                            const float* dinv, int64_t sd, int64_t n, int batch, hipStream_t s) {
    if (n == TB)
        return launch_gemm_nt_f32_batched(b, ldb, b, ldb, dinv, TB, m, TB, TB, 1.0f, 0.0f, false, batch, sb, sb, sd, s);
    const int64_t n1 = (n / TB / 2) * TB, n2 = n - n1;
    NNGP_TRY(trsm_rlt_batched(b, ldb, sb, m, l, ldl, sl, dinv, sd, n1, batch, s));
    NNGP_TRY(launch_gemm_nt_f32_batched(b + n1, ldb, b, ldb, l + n1 * ldl, ldl, m, n2, n1, -1.0f, 1.0f, false, batch, sb,
                                        sb, sl, s));
    return trsm_rlt_batched(b + n1, ldb, sb, m, l + n1 * ldl + n1, ldl, sl, dinv + (n1 / TB) * TB * TB, sd, n2, batch, s);
}

int launch_transpose_blocks_f32(const float* src, float* dst, int64_t bs, int64_t count, hipStream_t s) {
    NNGP_REQUIRE(bs % 32 == 0 && count > 0 && count <= 65535, "transpose_blocks: bad arguments");
    hipLaunchKernelGGL(k_transpose_blocks, dim3((unsigned)(bs / 32), (unsigned)(bs / 32), (unsigned)count), dim3(256), 0, s,
                       src, dst, bs);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Round 4 built 2048-wide inverted blocks (debug key 9 = 2, from np = 8192 on): the blocked solves then take half as many steps,
// each with two 1024-column panels (K = 2048) per pass over the right-hand sides, and the CG's blocked TRSVs half as many launches.
// Measured at N = 32768, M = 1024: the three solves 19.9 -> 19.2 ms (a step's diagonal GEMM is a full 2048^2 product: 95 us against
// 2 x 30), posterior -0.5 ms, but the inverses cost the factorisation stage +1.1 ms -- 1024 stays.
int64_t triinv_block(int64_t np) {
    int64_t bs = NNGP_KNOB(6) >= 128 ? (int64_t)NNGP_KNOB(6) : (np >= 8192 && NNGP_KNOB(9) == 2) ? 2048 : 1024;  // debug key 6: block size experiment
    return np < bs ? np : bs;
}

// Inverts the diagonal blocks (size bs, tail np % bs) of the float32 factor: T_J = L_JJ^-T by a batched
// triangular solve against the identity (MFMA GEMMs), X_J = T_J^T.  Used by the blocked TRSVs below.
// Blocks [j0, j1) only (round 4: the look-ahead Cholesky inverts the blocks of its finished block columns on a low-priority
// stream while its last, chain-bound block columns leave most of the chip idle; nngp_model_factor_end does the rest).
int triinv_build_range(const float* l, int64_t ld, const float* dinv, int64_t np, TriInv& ti, int64_t j0, int64_t j1, hipStream_t s) {
    const int64_t bs = ti.bs;
    NNGP_REQUIRE(bs % TB == 0 && bs > 0 && np % TB == 0, "triinv_build: bad block size");
    const int64_t nfull = np / bs, tail = np % bs, nall = nfull + (tail ? 1 : 0);
    if (j1 > nall) j1 = nall;
    if (j0 >= j1) return 0;
    const int64_t total = (j1 - j0) * bs * bs;
    float* t0 = ti.tinv + j0 * bs * bs;
    hipLaunchKernelGGL(k_set_identity_blocks, dim3(blocks256(total)), dim3(256), 0, s, t0, bs, total);
    const int64_t jf = j1 < nfull ? j1 : nfull;  // full blocks in the range: [j0, jf)
    if (jf > j0)
        NNGP_TRY(trsm_rlt_batched(t0, bs, bs * bs, bs, l + j0 * bs * (ld + 1), ld, bs * (ld + 1), dinv + j0 * (bs / TB) * TB * TB,
                                  (bs / TB) * TB * TB, bs, (int)(jf - j0), s));
    if (tail > 0 && j1 == nall) {
        const int64_t o = nfull * bs;
        NNGP_TRY(trsm_rlt_f32(ti.tinv + nfull * bs * bs, bs, tail, l + o * (ld + 1), ld, dinv + (o / TB) * TB * TB, tail, s));
    }
    hipLaunchKernelGGL(k_transpose_blocks, dim3((unsigned)(bs / 32), (unsigned)(bs / 32), (unsigned)(j1 - j0)), dim3(256), 0, s, t0,
                       ti.xinv + j0 * bs * bs, bs);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int triinv_build(const float* l, int64_t ld, const float* dinv, int64_t np, TriInv& ti, hipStream_t s) {
    const int64_t j0 = ti.done_blocks;  // already inverted for this factor (look-ahead Cholesky)
    ti.done_blocks = 0;
    return triinv_build_range(l, ld, dinv, np, ti, j0, (np + ti.bs - 1) / ti.bs, s);
}

// The inverted blocks are triangular (X_J = L_JJ^-1 lower, T_J = L_JJ^-T upper): the multiplication by them skips the k tiles where a
// column tile's rows of the block are zero -- 45 % of a 1024-block's product (round 4; debug key 9 = 4: the full product).
#define kTriLower (NNGP_KNOB(9) == 4 ? 0 : 1)
#define kTriUpper (NNGP_KNOB(9) == 4 ? 0 : 2)

// B[m, np] <- B L^-T with the inverted bs-blocks: per block column J one GEMM with X_J = L_JJ^-1 (out of place into
// `tmp`, [m, bs]) and one trailing GEMM -- 3 launches per block instead of the ~16 of the 128-wide recursion.
int trsm_rlt_blocks_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ld, const TriInv& ti, int64_t np,
                        float* tmp, hipStream_t s) {
    const int64_t bs = ti.bs;
    for (int64_t o = 0, j = 0; o < np; o += bs, ++j) {
        const int64_t sz = (np - o < bs) ? np - o : bs;
        NNGP_TRY(launch_gemm_nt_f32(tmp, sz, b + o, ldb, ti.xinv + j * bs * bs, bs, m, sz, sz, 1.0f, 0.0f, false, s, kTriLower));
        NNGP_HIP_CHECK(hipMemcpy2DAsync(b + o, sizeof(float) * ldb, tmp, sizeof(float) * sz, sizeof(float) * sz, m,
                                        hipMemcpyDeviceToDevice, s));
        const int64_t rest = np - o - sz;
        if (rest > 0)
            NNGP_TRY(launch_gemm_nt_f32(b + o + sz, ldb, b + o, ldb, l + (o + sz) * ld + o, ld, m, rest, sz, -1.0f, 1.0f,
                                        false, s));
    }
    return 0;
}

// B[m, np] <- B L^-1 = B U^-T with U = L^T stored in `lt`; block columns last to first, T_J = L_JJ^-T as the GEMM operand.
int trsm_rut_blocks_f32(float* b, int64_t ldb, int64_t m, const float* lt, int64_t ld, const TriInv& ti, int64_t np,
                        float* tmp, hipStream_t s) {
    const int64_t bs = ti.bs;
    const int64_t nblk = (np + bs - 1) / bs;
    for (int64_t j = nblk - 1; j >= 0; --j) {
        const int64_t o = j * bs;
        const int64_t sz = (np - o < bs) ? np - o : bs;
        NNGP_TRY(launch_gemm_nt_f32(tmp, sz, b + o, ldb, ti.tinv + j * bs * bs, bs, m, sz, sz, 1.0f, 0.0f, false, s, kTriUpper));
        NNGP_HIP_CHECK(hipMemcpy2DAsync(b + o, sizeof(float) * ldb, tmp, sizeof(float) * sz, sizeof(float) * sz, m,
                                        hipMemcpyDeviceToDevice, s));
        if (o > 0)
            NNGP_TRY(launch_gemm_nt_f32(b, ldb, b + o, ldb, lt + o, ld, m, o, sz, -1.0f, 1.0f, false, s));
    }
    return 0;
}

// Blocked solves with the large trailing products on the float16 pipe.  Per block column J (ti.bs columns: round 4 takes
// TWO 1024-column panels per step where the inverted diagonal blocks are 2048 wide -- half the steps of the latency chain
// {diagonal GEMM, split, update}, and every update tile accumulates K = 2048 per pass over C): the diagonal solve stays a
// float32 GEMM with the inverted block (into `tmp`); k_split_rows_rowscale copies it back and writes its split copy (one
// power-of-two scale per right-hand-side row, one 1024-column panel after the other, col_stride bytes apart like the
// factor's own panels); the trailing update is one multi-panel split-float16 launch against the split copy of L (forward)
// or L^T (backward).  Steps whose update has too few 256 x 256 tiles to fill the GPU stay on the float32 kernel
// (64 x 64 tiles).
static bool h3_worth(int64_t m, int64_t cols) { return ((m + 255) / 256) * ((cols + 255) / 256) >= 96; }

// C [m, n] -= X_J W^T for the step's block X_J (sz columns, split in sw.planes_b by panels of kc columns) and the matching
// panels of the factor's split copy: wrows = split rows of W's first row in the step's FIRST (lowest-k) panel.
static int h3_step_update(float* c, int64_t ldc, int64_t m, int64_t n, int64_t sz, const char* wrows, const SplitWork& sw, hipStream_t s) {
    const int64_t kc = sw.k_cap, ldp = 4 * kc;
    const int np = (int)((sz + kc - 1) / kc);
    const float alpha = -1.0f / sw.scale;
    if (sz % kc == 0)  // equal panels: one pass over C, the latest panel first
        return launch_gemm_nt_h3x(c, ldc, sw.planes_b + (int64_t)(np - 1) * sw.col_stride, wrows + (int64_t)(np - 1) * sw.col_stride, ldp,
                                  sw.col_stride, np, 0, m, n, kc, alpha, 1.0f, false, 0, sw.counters, sw.solve_reserve, s, sw.row_inv);
    for (int p = np - 1; p >= 0; --p) {  // a short last panel (tail of the matrix): one launch per panel
        const int64_t kp = (p == np - 1) ? sz - (int64_t)p * kc : kc;
        NNGP_TRY(launch_gemm_nt_h3(c, ldc, sw.planes_b + (int64_t)p * sw.col_stride, wrows + (int64_t)p * sw.col_stride, ldp, m, n, kp, alpha,
                                   1.0f, false, 0, sw.counters, sw.solve_reserve, s, sw.row_inv));
    }
    return 0;
}

static bool h3_step_ok(const SplitWork& sw, int64_t bs, int64_t m) {
    return bs % sw.k_cap == 0 && bs / sw.k_cap <= sw.b_panels && bs <= 2048 && m <= sw.mb_cap;
}

int trsm_rlt_blocks_h3(float* b, int64_t ldb, int64_t m, const float* l, int64_t ld, const TriInv& ti, int64_t np,
                       float* tmp, const SplitWork& sw, hipStream_t s) {
    const int64_t bs = ti.bs, kc = sw.k_cap, ldp = 4 * kc;
    NNGP_REQUIRE(sw.l_ready && h3_step_ok(sw, bs, m), "trsm_rlt_blocks_h3: split copy of L not available");
    for (int64_t o = 0, j = 0; o < np; o += bs, ++j) {
        const int64_t sz = (np - o < bs) ? np - o : bs;
        const int64_t rest = np - o - sz;
        NNGP_TRY(launch_gemm_nt_f32(tmp, sz, b + o, ldb, ti.xinv + j * bs * bs, bs, m, sz, sz, 1.0f, 0.0f, false, s, kTriLower));
        if (rest > 0 && sz == bs && h3_worth(m, rest)) {
            NNGP_TRY(launch_split_rows_rowscale(tmp, sz, m, sz, b + o, ldb, sw.planes_b, ldp, sw.row_inv, s, sw.col_stride));
            // rows o + sz .. of L, columns [o, o + sz): block columns o / kc .. of the factor's split copy, rows at their global index
            NNGP_TRY(h3_step_update(b + o + sz, ldb, m, rest, sz, sw.planes + (o / kc) * sw.col_stride + (o + sz) * ldp, sw, s));
        } else {
            NNGP_HIP_CHECK(hipMemcpy2DAsync(b + o, sizeof(float) * ldb, tmp, sizeof(float) * sz, sizeof(float) * sz, m,
                                            hipMemcpyDeviceToDevice, s));
            if (rest > 0)
                NNGP_TRY(launch_gemm_nt_f32(b + o + sz, ldb, b + o, ldb, l + (o + sz) * ld + o, ld, m, rest, sz, -1.0f, 1.0f,
                                            false, s));
        }
    }
    return 0;
}

int trsm_rut_blocks_h3(float* b, int64_t ldb, int64_t m, const float* lt, int64_t ld, const TriInv& ti, int64_t np,
                       float* tmp, const SplitWork& sw, hipStream_t s) {
    const int64_t bs = ti.bs, kc = sw.k_cap, ldp = 4 * kc;
    NNGP_REQUIRE(sw.lt_ready && h3_step_ok(sw, bs, m), "trsm_rut_blocks_h3: split copy of L^T not available");
    const int64_t nblk = (np + bs - 1) / bs;
    for (int64_t j = nblk - 1; j >= 0; --j) {
        const int64_t o = j * bs;
        const int64_t sz = (np - o < bs) ? np - o : bs;
        NNGP_TRY(launch_gemm_nt_f32(tmp, sz, b + o, ldb, ti.tinv + j * bs * bs, bs, m, sz, sz, 1.0f, 0.0f, false, s, kTriUpper));
        // lt == nullptr: no float32 copy of L^T exists (it is only built for the float32 path) -- every update, also the
        // few-tile ones and the tail block, goes through the split copy: +0.07 ms per small step, -1.8 ms of transposition
        if (o > 0 && (lt == nullptr || (sz == bs && h3_worth(m, o)))) {
            NNGP_TRY(launch_split_rows_rowscale(tmp, sz, m, sz, b + o, ldb, sw.planes_b, ldp, sw.row_inv, s, sw.col_stride));
            // block rows o / kc .. of the split copy of L^T: out_j[r][k] = L[j kc + k][r], rows r < o
            NNGP_TRY(h3_step_update(b, ldb, m, o, sz, sw.planes_t + (o / kc) * sw.col_stride, sw, s));
        } else {
            NNGP_HIP_CHECK(hipMemcpy2DAsync(b + o, sizeof(float) * ldb, tmp, sizeof(float) * sz, sizeof(float) * sz, m,
                                            hipMemcpyDeviceToDevice, s));
            if (o > 0) NNGP_TRY(launch_gemm_nt_f32(b, ldb, b + o, ldb, lt + o, ld, m, o, sz, -1.0f, 1.0f, false, s));
        }
    }
    return 0;
}

// Solves L x = b (float32, length np).  Block J: x_J = X_J b_J, then b[below] -= L[below, J] x_J.  b is consumed (its rows
// below the current block are updated in place); the solution goes to x, a different buffer -- writing x_J over b_J would
// need a staging copy per block, one more launch in a chain whose cost is its launch count.
// (Measured and dropped: fusing the backward sweep's partial-sum and reduction kernels with a last-arriver counter --
// the agent-scope release each workgroup then needs writes back its XCD's L2: CG alone 24.7 instead of 15.0 ms.)
int trsv_forward_f32(const float* l, int64_t ld, const TriInv& ti, int64_t np, float* b, float* x, hipStream_t s) {
    NNGP_REQUIRE(b != x, "trsv_forward: b and x must be different buffers");
    const int64_t bs = ti.bs;
    for (int64_t o = 0, j = 0; o < np; o += bs, ++j) {
        const int64_t sz = (np - o < bs) ? np - o : bs;
        hipLaunchKernelGGL(k_gemv_n_f32, dim3((unsigned)((sz + 3) / 4)), dim3(256), 0, s, ti.xinv + j * bs * bs, bs, sz, sz,
                           b + o, x + o, 0);
        const int64_t below = np - o - sz;
        if (below > 0)
            hipLaunchKernelGGL(k_gemv_n_f32, dim3((unsigned)((below + 3) / 4)), dim3(256), 0, s, l + (o + sz) * ld + o, ld,
                               below, sz, x + o, b + o + sz, 1);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// Solves L^T x = b.  Block J (last to first): x_J = X_J^T b_J = T_J b_J, then b[:o] -= L[J rows, :o]^T x_J.  b is consumed.
int trsv_backward_f32(const float* l, int64_t ld, const TriInv& ti, int64_t np, float* b, float* x, hipStream_t s) {
    NNGP_REQUIRE(b != x, "trsv_backward: b and x must be different buffers");
    const int64_t bs = ti.bs;
    const int64_t nblk = (np + bs - 1) / bs;
    for (int64_t j = nblk - 1; j >= 0; --j) {
        const int64_t o = j * bs;
        const int64_t sz = (np - o < bs) ? np - o : bs;
        hipLaunchKernelGGL(k_gemv_n_f32, dim3((unsigned)((sz + 3) / 4)), dim3(256), 0, s, ti.tinv + j * bs * bs, bs, sz, sz,
                           b + o, x + o, 0);
        if (o > 0) {
            const int nchunk = (int)(sz / 128);
            hipLaunchKernelGGL(k_gemv_t_partial_f32, dim3((unsigned)((o + 255) / 256), (unsigned)nchunk), dim3(256), 0, s,
                               l + o * ld, ld, o, x + o, ti.partial, np);
            hipLaunchKernelGGL(k_sub_partials_f32, dim3((unsigned)((o + 255) / 256)), dim3(256), 0, s, b, ti.partial, np,
                               nchunk, o);
        }
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_symv_f64(const double* a, int64_t lda, int64_t n, const double* x, double* y, double diag_add, double* part,
                    int64_t np, hipStream_t s) {
    if (n <= 0) return 0;
    NNGP_REQUIRE(part != nullptr && np % 128 == 0 && np >= n, "symv_f64: bad workspace");
    const int64_t nblk = (n + 127) / 128;
    hipLaunchKernelGGL(k_symv_tiles_f64, dim3((unsigned)(nblk * (nblk + 1) / 2)), dim3(256), 0, s, a, lda, n, x, part, np);
    hipLaunchKernelGGL(k_symv_reduce_f64, dim3(blocks256(n)), dim3(256), 0, s, part, nblk, np, n, x, diag_add, y);
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
// The scalars (rz, pAp, alpha, beta) live on the device; the host only reads |r|^2 to decide when to stop.  Two phases:
//   pcg_begin  enqueues x = 0, r = b and `ahead` iterations WITHOUT waiting (|b|^2 and the |r|^2 of every iteration go to
//              w.scal[8...]); iterating past convergence is harmless (the update kernels take alpha = beta = 0 when their
//              denominators vanish), so the solve can run on its own stream from the moment the factor is ready;
//   pcg_finish waits, reads the history, and continues one iteration at a time if `ahead` were not enough.
namespace {
constexpr int kPcgHist = 8;       // w.scal[8] = |b|^2, w.scal[9 + it] = |r|^2 after iteration it
constexpr int kPcgMaxAhead = 22;  // w.scal has 32 slots

int pcg_iteration(const double* k64, int64_t ld, int64_t n, double reg, const float* l32, int64_t ld32, const TriInv& ti,
                  int64_t np, double* xcol, PcgWork& w, int it, double* rr_out, hipStream_t s) {
    hipLaunchKernelGGL(k_f64_to_f32, dim3(blocks256(np)), dim3(256), 0, s, w.r, w.f32a, n, np);
    NNGP_TRY(trsv_forward_f32(l32, ld32, ti, np, w.f32a, w.f32b, s));
    NNGP_TRY(trsv_backward_f32(l32, ld32, ti, np, w.f32b, w.f32c, s));
    hipLaunchKernelGGL(k_f32_to_f64, dim3(blocks256(n)), dim3(256), 0, s, w.f32c, w.z, n);
    hipLaunchKernelGGL(k_dot, dim3(kDotBlocks), dim3(256), 0, s, w.r, w.z, n, w.scal + 2, w.dot_part, w.dot_ctr);  // rz_new
    hipLaunchKernelGGL(k_pcg_update_p, dim3(blocks256(n)), dim3(256), 0, s, w.p, w.z, n, w.scal, it == 0);
    hipLaunchKernelGGL(k_scal_shift, dim3(1), dim3(1), 0, s, w.scal);  // rz = rz_new
    // K is symmetric: half the bytes (debug key 5 = 7: plain GEMV).  Small fits keep the one-launch GEMV: the product is
    // launch-bound there (N = 1000: +0.1 ms per predict with the two-launch form).
    if (w.symv_part != nullptr && w.symv_np >= np && n >= 4096 && NNGP_KNOB(5) != 7)
        NNGP_TRY(launch_symv_f64(k64, ld, n, w.p, w.q, reg, w.symv_part, w.symv_np, s));
    else
        NNGP_TRY(launch_gemv_f64(k64, ld, n, n, w.p, 1, w.q, 1, reg, s));
    hipLaunchKernelGGL(k_dot, dim3(kDotBlocks), dim3(256), 0, s, w.p, w.q, n, w.scal + 1, w.dot_part, w.dot_ctr);  // pAp
    hipLaunchKernelGGL(k_pcg_update_xr, dim3(blocks256(n)), dim3(256), 0, s, xcol, w.r, w.p, w.q, n, w.scal);
    hipLaunchKernelGGL(k_dot, dim3(kDotBlocks), dim3(256), 0, s, w.r, w.r, n, rr_out, w.dot_part, w.dot_ctr);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}
}  // namespace

int precond_apply(const float* l32, int64_t ld32, const TriInv& ti, int64_t n, int64_t np, const double* r, double* z, PcgWork& w,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_f64_to_f32, dim3(blocks256(np)), dim3(256), 0, s, r, w.f32a, n, np);
    NNGP_TRY(trsv_forward_f32(l32, ld32, ti, np, w.f32a, w.f32b, s));
    NNGP_TRY(trsv_backward_f32(l32, ld32, ti, np, w.f32b, w.f32c, s));
    hipLaunchKernelGGL(k_f32_to_f64, dim3(blocks256(n)), dim3(256), 0, s, w.f32c, z, n);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int pcg_begin(const double* k64, int64_t ld, int64_t n, double reg, const float* l32, int64_t ld32, const TriInv& ti,
              int64_t np, const double* bcol, double* xcol, PcgWork& w, int ahead, hipStream_t s) {
    NNGP_REQUIRE(ahead >= 0 && ahead <= kPcgMaxAhead, "pcg_begin: at most %d iterations ahead", kPcgMaxAhead);
    NNGP_HIP_CHECK(hipMemsetAsync(xcol, 0, sizeof(double) * n, s));
    NNGP_HIP_CHECK(hipMemcpyAsync(w.r, bcol, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_dot, dim3(kDotBlocks), dim3(256), 0, s, bcol, bcol, n, w.scal + kPcgHist, w.dot_part, w.dot_ctr);
    for (int it = 0; it < ahead; ++it)
        NNGP_TRY(pcg_iteration(k64, ld, n, reg, l32, ld32, ti, np, xcol, w, it, w.scal + kPcgHist + 1 + it, s));
    NNGP_HIP_CHECK(hipMemcpyAsync(w.host_scal + kPcgHist, w.scal + kPcgHist, sizeof(double) * (1 + ahead), hipMemcpyDeviceToHost, s));
    return 0;
}

// resume: the solve was stopped after `ahead` iterations at a looser tolerance (the CG state in w is intact): go on from there.
int pcg_finish(const double* k64, int64_t ld, int64_t n, double reg, const float* l32, int64_t ld32, const TriInv& ti,
               int64_t np, double* xcol, PcgWork& w, int ahead, int max_iters, double tol, int* iters_out,
               double* relres_out, hipStream_t s, bool resume) {
    NNGP_HIP_CHECK(hipStreamSynchronize(s));
    const double bnorm2 = w.host_scal[kPcgHist];
    int iters = resume ? ahead : 0;
    double relres = 0.0;
    if (bnorm2 > 0.0) {
        relres = 1.0;
        bool done = false;
        for (int it = 0; it < ahead && !resume; ++it) {  // the iterations that ran ahead: where did the residual first meet tol?
            const double rel = sqrt(w.host_scal[kPcgHist + 1 + it] / bnorm2);
            if (!(rel == rel)) {  // NaN: the preconditioner is unusable
                set_error("pcg_solve: residual became NaN at iteration %d", it + 1);
                return -3;
            }
            if (!done) {
                iters = it + 1;
                relres = rel;
                if (rel <= tol) done = true;
            } else if (rel < relres) {
                relres = rel;  // the extra iterations only polish
            }
        }
        for (int it = ahead; it < max_iters && !done; ++it) {
            // Iterations >= 1 launch the same kernels with the same arguments: they can be captured once as a hipGraph and replayed
            // (debug key 14 = 2).  Measured (round 4, scripts/cov_alone.py, N = 32768): CG alone 11.8 ms either way, covariance + mean with
            // the CG beside it 35.7 ms either way -- the ~170 launches of an iteration are bound by the GPU's own kernel-to-kernel
            // latency and HBM, not by the host's enqueue cost; plain launches stay.
            const bool graphable = it >= 1 && NNGP_KNOB(14) == 2;
            const void* key[4] = {k64, l32, xcol, w.r};
            const int64_t dims[4] = {n, np, ti.bs, ld};
            if (graphable && w.iter_graph != nullptr && (memcmp(key, w.graph_key, sizeof(key)) != 0 || memcmp(dims, w.graph_dims, sizeof(dims)) != 0 || w.graph_reg != reg)) {
                (void)hipGraphExecDestroy(w.iter_graph);
                w.iter_graph = nullptr;
            }
            if (graphable && w.iter_graph == nullptr) {
                hipGraph_t g = nullptr;
                if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const int rc_it = pcg_iteration(k64, ld, n, reg, l32, ld32, ti, np, xcol, w, it, w.scal + 3, s);
                    const hipError_t ec = hipStreamEndCapture(s, &g);
                    if (rc_it == 0 && ec == hipSuccess && g != nullptr && hipGraphInstantiate(&w.iter_graph, g, nullptr, nullptr, 0) == hipSuccess) {
                        memcpy(w.graph_key, key, sizeof(key));
                        memcpy(w.graph_dims, dims, sizeof(dims));
                        w.graph_reg = reg;
                    } else {
                        w.iter_graph = nullptr;
                    }
                    if (g != nullptr) (void)hipGraphDestroy(g);
                    (void)hipGetLastError();
                }
            }
            if (graphable && w.iter_graph != nullptr) {
                NNGP_HIP_CHECK(hipGraphLaunch(w.iter_graph, s));
            } else {
                NNGP_TRY(pcg_iteration(k64, ld, n, reg, l32, ld32, ti, np, xcol, w, it, w.scal + 3, s));
            }
            NNGP_HIP_CHECK(hipMemcpyAsync(w.host_scal + 3, w.scal + 3, sizeof(double), hipMemcpyDeviceToHost, s));
            NNGP_HIP_CHECK(hipStreamSynchronize(s));
            iters = it + 1;
            relres = sqrt(w.host_scal[3] / bnorm2);
            if (!(relres == relres)) {
                set_error("pcg_solve: residual became NaN at iteration %d", iters);
                return -3;
            }
            if (relres <= tol) done = true;
        }
    }
    if (iters_out) *iters_out = iters;
    if (relres_out) *relres_out = relres;
    return 0;
}

int pcg_solve(const double* k64, int64_t ld, int64_t n, double reg, const float* l32, int64_t ld32,
              const TriInv& ti, int64_t np, const double* bcol, double* xcol, PcgWork& w, int max_iters,
              double tol, int* iters_out, double* relres_out, hipStream_t s) {
    NNGP_TRY(pcg_begin(k64, ld, n, reg, l32, ld32, ti, np, bcol, xcol, w, 0, s));
    return pcg_finish(k64, ld, n, reg, l32, ld32, ti, np, xcol, w, 0, max_iters, tol, iters_out, relres_out, s, false);
}

}  // namespace nngp
