// float32 "NT" GEMM on the gfx950 matrix cores:  C[M,N] = beta*C + alpha * A[M,K] * B[N,K]^T.
//
// This one kernel carries every heavy step of the blocked Cholesky that replaces the reference's
// cho_factor (train.py:171-172 via nt.predict): the SYRK trailing updates
// (lower_only), the panel GEMMs of the triangular solves, the multiplication by inverted diagonal
// blocks, and the posterior covariance products.
//
// Design (CDNA4): 128x128 tile per 256-thread workgroup, 4 waves in a 2x2 grid, each wave owns a
// 64x64 sub-tile = 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 VGPRs).  Both operands are
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

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int STAGE_FLOATS = (BM + BN) * BK;  // 8192 floats = 32 KiB per stage

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// byte offset of logical 16-byte chunk `ch` (0..7) of row `row` inside a [rows][32 floats] LDS image
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

template <bool LOWER>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_f32(float* C, int64_t ldc, const float* A, int64_t lda,
                                                        const float* B, int64_t ldb, int tiles_n, int nk,
                                                        float alpha, float beta) {
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
    const float* Ab = A + (int64_t)bi * BM * lda;
    const float* Bb = B + (int64_t)bj * BN * ldb;

    // global -> register staging: 4 chunks of A and 4 of B per thread (chunk = 16 bytes = 4 k)
    f32x4 ga[4], gb[4];
    const int ld_row = tid >> 3, ld_ch = tid & 7;  // + 32 rows per e
    auto load_tile = [&](int t) {
        const int64_t k0 = (int64_t)t * BK + ld_ch * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = ld_row + 32 * e;
            ga[e] = *reinterpret_cast<const f32x4*>(Ab + (int64_t)row * lda + k0);
            gb[e] = *reinterpret_cast<const f32x4*>(Bb + (int64_t)row * ldb + k0);
        }
    };
    auto store_tile = [&](int buf) {
        char* sa = reinterpret_cast<char*>(smem + buf * STAGE_FLOATS);
        char* sb = sa + BM * BK * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = ld_row + 32 * e;
            *reinterpret_cast<f32x4*>(sa + lds_off(row, ld_ch)) = ga[e];
            *reinterpret_cast<f32x4*>(sb + lds_off(row, ld_ch)) = gb[e];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) load_tile(t + 1);
        const char* sa = reinterpret_cast<const char*>(smem + (t & 1) * STAGE_FLOATS);
        const char* sb = sa + BM * BK * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + frow;
                const int rb = wn * 64 + i * 32 + frow;
                fa[i] = *reinterpret_cast<const f32x4*>(sa + lds_off(ra, 2 * s + fh));
                fb[i] = *reinterpret_cast<const f32x4*>(sb + lds_off(rb, 2 * s + fh));
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nk) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // epilogue: acc[i][j][r] is element (row, col) with row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
    const int64_t row_base = (int64_t)bi * BM + wm * 64;
    const int64_t col_base = (int64_t)bj * BN + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
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

}  // namespace

int launch_gemm_nt_f32(float* c, int64_t ldc, const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m,
                       int64_t n, int64_t k, float alpha, float beta, bool lower_only, hipStream_t s) {
    if (m <= 0 || n <= 0) return 0;
    NNGP_REQUIRE(m % BM == 0 && n % BN == 0 && k % 128 == 0 && k > 0,
                 "gemm_nt_f32: dims must be multiples of 128 (m=%lld n=%lld k=%lld)", (long long)m, (long long)n,
                 (long long)k);
    NNGP_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 &&
                     ((uintptr_t)c & 3) == 0,
                 "gemm_nt_f32: operands must be 16-byte aligned");
    NNGP_REQUIRE(lda >= k && ldb >= k && ldc >= n, "gemm_nt_f32: leading dimension too small");
    const int64_t tm = m / BM, tn = n / BN;
    const int nk = (int)(k / BK);
    if (lower_only) {
        NNGP_REQUIRE(m == n, "gemm_nt_f32: lower_only needs a square result");
        const int64_t nb = tm * (tm + 1) / 2;
        NNGP_REQUIRE(nb < 16000000, "gemm_nt_f32: grid too large");  // float sqrt decode stays exact-correctable
        hipLaunchKernelGGL(k_gemm_nt_f32<true>, dim3((unsigned)nb), dim3(256), 0, s, c, ldc, a, lda, b, ldb, (int)tn,
                           nk, alpha, beta);
    } else {
        const int64_t nb = tm * tn;
        NNGP_REQUIRE(nb < 2147483647LL, "gemm_nt_f32: grid too large");
        hipLaunchKernelGGL(k_gemm_nt_f32<false>, dim3((unsigned)nb), dim3(256), 0, s, c, ldc, a, lda, b, ldb, (int)tn,
                           nk, alpha, beta);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nngp
