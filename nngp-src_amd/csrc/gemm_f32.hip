// float32 "NT" GEMM on the gfx950 matrix cores:  C[M,N] = beta*C + alpha * A[M,K] * B[N,K]^T.
//
// This one kernel carries every heavy step of the blocked Cholesky that replaces the reference's
// cho_factor (train.py:171-172 via nt.predict): the SYRK trailing updates
// (lower_only), the panel GEMMs of the triangular solves, the multiplication by inverted diagonal
// blocks, and the posterior covariance products.
//
// Design (CDNA4): 256-thread workgroup = 4 waves in a 2x2 grid; the throughput shape is a 128x128 tile with a
// 64x64 sub-tile per wave = 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 VGPRs); 64x128 and 64x64 workgroup
// tiles serve problems with too few tiles to fill 256 CUs (a workgroup's K loop is one serial MFMA chain on one
// CU).  K tiles are consumed from the high end of K down (see load_tile: accuracy of the Cholesky updates).  Both operands are
// K-contiguous ("row-major x row-major^T"), so A and B use the same staging path: 16-byte global
// loads -> registers -> LDS, double-buffered over BK = 32, one barrier per K-step.  LDS rows are
// 128 bytes; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 so that the ds_read_b128
// fragment reads (32 rows x one chunk per half-wave) are bank-conflict free.  Each lane reads 4
// consecutive k per 16-byte load; lanes 0-31 and 32-63 hold different k, and the four MFMAs of a
// load use element kk of both operands, which keeps A and B on the same k (any k order is a valid
// dot product; fp32 MFMA is an exact k-ordered fma chain).
//
// All of M, N, K must be multiples of 128 (device matrices are padded by the caller); pointers and
// leading dimensions must be 16-byte aligned.  C may alias A when N == 128 (each workgroup then reads
// and writes only its own 128 rows, and all reads finish before the epilogue) -- the in-place
// multiplication by an inverted diagonal block relies on this.
#include "common.h"

namespace nngp {

namespace {

constexpr int BK = 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// byte offset of logical 16-byte chunk `ch` (0..7) of row `row` inside a [rows][32 floats] LDS image
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// Workgroup tile = (64*WM) x (64*WN): 4 waves in a 2x2 grid, each wave WM x WN accumulators of 32x32.
//   <2,2> 128x128  the throughput shape (big SYRK / GEMM)
//   <1,2>  64x128  in-place multiplication by an inverted diagonal block with few row tiles
//   <1,1>  64x64   small problems: 4x the workgroups, a quarter of the serial MFMA chain per workgroup
template <int WM, int WN, bool LOWER>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_f32(float* C, int64_t ldc, const float* A, int64_t lda,
                                                        const float* B, int64_t ldb, int tiles_n, int nk_all,
                                                        float alpha, float beta, int64_t sc, int64_t sa,
                                                        int64_t sb, int tri) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int STAGE_FLOATS = (BM + BN) * BK;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int bi, bj;
    if (LOWER) {
        const int p = blockIdx.x;
        bi = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
        while (bi * (bi + 1) / 2 > p) --bi;
        while ((bi + 1) * (bi + 2) / 2 <= p) ++bi;
        bj = p - bi * (bi + 1) / 2;
    } else {
        bi = blockIdx.x / tiles_n;
        bj = blockIdx.x % tiles_n;
    }
    // batched form: blockIdx.y selects one of several independent problems at constant strides
    C += (int64_t)blockIdx.y * sc;
    const float* Ab = A + (int64_t)blockIdx.y * sa + (int64_t)bi * BM * lda;
    const float* Bb = B + (int64_t)blockIdx.y * sb + (int64_t)bj * BN * ldb;

    // tri: B is triangular (the inverted diagonal blocks of the blocked solves) -- a column tile's K range shrinks to where its rows
    // of B are non-zero: 1 = lower (B[c][k] = 0 for k > c): k-tiles [0, hi]; 2 = upper (zero for k < c): k-tiles [lo, nk).  The
    // skipped terms are exact zeros, so the result is the full product's, bit for bit.
    int kt_hi = nk_all - 1, kt_lo = 0;
    if (tri == 1) {
        const int lim = ((bj + 1) * BN + BK - 1) / BK - 1;
        kt_hi = lim < kt_hi ? lim : kt_hi;
    } else if (tri == 2) {
        kt_lo = (bj * BN) / BK;
    }
    const int nk = kt_hi - kt_lo + 1;
    // global -> register staging: 16-byte chunks (4 k), 8 chunks per 128-byte row
    f32x4 ga[BM / 32], gb[BN / 32];
    const int ld_row = tid >> 3, ld_ch = tid & 7;  // + 32 rows per e
    // K tiles are consumed from the HIGH end of K down to 0.  In the Cholesky updates C -= L1 L2^T the products of
    // the early columns of L dominate (the factor's columns decay with k); summing the small late-column terms first
    // keeps the float32 partial sums small, which cuts the accumulated rounding error of a K ~ 16k chain by ~20x.
    auto load_tile = [&](int t) {
        const int64_t k0 = (int64_t)(kt_hi - t) * BK + ld_ch * 4;
#pragma unroll
        for (int e = 0; e < BM / 32; ++e)
            ga[e] = *reinterpret_cast<const f32x4*>(Ab + (int64_t)(ld_row + 32 * e) * lda + k0);
#pragma unroll
        for (int e = 0; e < BN / 32; ++e)
            gb[e] = *reinterpret_cast<const f32x4*>(Bb + (int64_t)(ld_row + 32 * e) * ldb + k0);
    };
    auto store_tile = [&](int buf) {
        char* sa_ = reinterpret_cast<char*>(smem + buf * STAGE_FLOATS);
        char* sb_ = sa_ + BM * BK * 4;
#pragma unroll
        for (int e = 0; e < BM / 32; ++e) *reinterpret_cast<f32x4*>(sa_ + lds_off(ld_row + 32 * e, ld_ch)) = ga[e];
#pragma unroll
        for (int e = 0; e < BN / 32; ++e) *reinterpret_cast<f32x4*>(sb_ + lds_off(ld_row + 32 * e, ld_ch)) = gb[e];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) load_tile(t + 1);
        const char* sa_ = reinterpret_cast<const char*>(smem + (t & 1) * STAGE_FLOATS);
        const char* sb_ = sa_ + BM * BK * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 fa[WM], fb[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i)
                fa[i] = *reinterpret_cast<const f32x4*>(sa_ + lds_off(wm * 32 * WM + i * 32 + frow, 2 * s + fh));
#pragma unroll
            for (int j = 0; j < WN; ++j)
                fb[j] = *reinterpret_cast<const f32x4*>(sb_ + lds_off(wn * 32 * WN + j * 32 + frow, 2 * s + fh));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nk) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // epilogue: acc[i][j][r] is element (row, col) with row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
    const int64_t row_base = (int64_t)bi * BM + wm * 32 * WM;
    const int64_t col_base = (int64_t)bj * BN + wn * 32 * WN;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            float* p0 = C + (row_base + i * 32 + 4 * fh) * ldc + col_base + j * 32 + frow;
            float cold[16];
            if (beta != 0.0f) {  // issue all 16 loads of the sub-tile before the first use
#pragma unroll
                for (int r = 0; r < 16; ++r) cold[r] = p0[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = alpha * acc[i][j][r];
                if (beta != 0.0f) v = fmaf(beta, cold[r], v);
                p0[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = v;
            }
        }
}

template <int WM, int WN, bool LOWER>
int launch_variant(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m,
                   int64_t n, int64_t k, float alpha, float beta, int batch, int64_t sc, int64_t sa, int64_t sb,
                   hipStream_t s, int tri = 0) {
    const int64_t tm = m / (64 * WM), tn = n / (64 * WN);
    const int64_t nb = LOWER ? tm * (tm + 1) / 2 : tm * tn;
    NNGP_REQUIRE(nb < (LOWER ? 16000000LL : 2147483647LL), "gemm_nt_f32: grid too large");
    hipLaunchKernelGGL((k_gemm_nt_f32<WM, WN, LOWER>), dim3((unsigned)nb, (unsigned)batch), dim3(256), 0, s, c, ldc, a,
                       lda, b, ldb, (int)tn, (int)(k / BK), alpha, beta, sc, sa, sb, tri);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int launch_gemm_nt_f32_batched(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb,
                               int64_t m, int64_t n, int64_t k, float alpha, float beta, bool lower_only, int batch,
                               int64_t stride_c, int64_t stride_a, int64_t stride_b, hipStream_t s, int tri) {
    NNGP_REQUIRE(tri == 0 || (tri >= 1 && tri <= 2 && n == k && !lower_only), "gemm_nt_f32: a triangular B must be square");
    if (m <= 0 || n <= 0 || batch <= 0) return 0;
    NNGP_REQUIRE(m % 128 == 0 && n % 128 == 0 && k % BK == 0 && k > 0,
                 "gemm_nt_f32: m, n must be multiples of 128 and k of 32 (m=%lld n=%lld k=%lld)", (long long)m, (long long)n,
                 (long long)k);
    NNGP_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 &&
                     ((uintptr_t)c & 3) == 0 && stride_a % 4 == 0 && stride_b % 4 == 0,
                 "gemm_nt_f32: operands must be 16-byte aligned");
    NNGP_REQUIRE(lda >= k && ldb >= k && ldc >= n, "gemm_nt_f32: leading dimension too small");
    NNGP_REQUIRE(batch <= 65535, "gemm_nt_f32: batch too large");
    // Tile choice: a 128x128 workgroup runs its K loop as one serial MFMA chain on one CU, so problems with few
    // tiles use smaller workgroup tiles to spread over more of the 256 CUs.
    const int64_t t128 = (lower_only ? (m / 128) * (m / 128 + 1) / 2 : (m / 128) * (n / 128)) * batch;
    const bool in_place = (c == a);  // C aliases A: legal only with a single column tile per row block (n == 128)
    if (in_place) NNGP_REQUIRE(n == 128 && beta == 0.0f, "gemm_nt_f32: in-place form needs n == 128 and beta == 0");
    if (lower_only) {
        NNGP_REQUIRE(m == n, "gemm_nt_f32: lower_only needs a square result");
        if (t128 >= 192)
            return launch_variant<2, 2, true>(c, ldc, a, lda, b, ldb, m, n, k, alpha, beta, batch, stride_c, stride_a,
                                              stride_b, s, tri);
        return launch_variant<1, 1, true>(c, ldc, a, lda, b, ldb, m, n, k, alpha, beta, batch, stride_c, stride_a,
                                          stride_b, s, tri);
    }
    if (t128 >= 192)
        return launch_variant<2, 2, false>(c, ldc, a, lda, b, ldb, m, n, k, alpha, beta, batch, stride_c, stride_a,
                                           stride_b, s, tri);
    if (in_place || t128 >= 96)
        return launch_variant<1, 2, false>(c, ldc, a, lda, b, ldb, m, n, k, alpha, beta, batch, stride_c, stride_a,
                                           stride_b, s, tri);
    return launch_variant<1, 1, false>(c, ldc, a, lda, b, ldb, m, n, k, alpha, beta, batch, stride_c, stride_a,
                                       stride_b, s, tri);
}

int launch_gemm_nt_f32(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m,
                       int64_t n, int64_t k, float alpha, float beta, bool lower_only, hipStream_t s, int tri) {
    return launch_gemm_nt_f32_batched(c, ldc, a, lda, b, ldb, m, n, k, alpha, beta, lower_only, 1, 0, 0, 0, s, tri);
}

}  // namespace nngp
