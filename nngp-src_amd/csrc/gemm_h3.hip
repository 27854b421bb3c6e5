// "Split-float16" NT GEMM on the gfx950 matrix cores:  C[M,N] = beta*C + alpha * A[M,K] * B[N,K]^T  with float32
// operands carried as two float16 planes (a*s = hi + lo, hi = fp16(a*s), lo = fp16(a*s - hi): 22 significant bits)
// and three v_mfma_f32_32x32x16_f16 products per term (hi*hi + hi*lo + lo*hi; lo*lo ~ 2^-22 relative is dropped),
// accumulated in float32.  The float16 matrix pipe runs 16x the float32 one, so a float32-grade product costs
// 3/16 of the float32-MFMA time.  Used for the large trailing updates of the blocked Cholesky (SURVEY.md 8a row a3,
// reference: cho_factor via nt.predict, train.py:171-172): that factor is a preconditioner, its
// accuracy requirement is "float32 grade", which the split meets (measured: same CG iteration count +-1).
//
// Operand format ("split rows", written by k_split_rows): row r is [K/32] blocks of 128 bytes, each block = 32 hi
// halfs followed by 32 lo halfs of the same 32 k -- so a row's k-block is one 128-byte line, and the panel has the same
// footprint as its float32 original (4 bytes per element).  s is a power of two chosen by the caller so that
// max|a*s| <= 2^14 (no float16 overflow; entries down to 2^-16 of the largest keep all 22 bits).
//
// Kernel: 512-thread workgroup = 8 waves (2 x 4), tile 256x256, wave sub-tile 128x64 = 4x2 accumulators of 32x32
// (128 VGPRs), BK = 32 per stage (two k16 MFMA sub-steps), LDS 2 stages x 64 KB, 128-byte rows with the 16-byte chunk
// index XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128, same image geometry as gemm_f32.hip).  K blocks are
// consumed from the high end down (accumulation order of the Cholesky updates, see gemm_f32.hip).
#include "common.h"

namespace nngp {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HT = 256;                    // workgroup tile (rows and columns)
constexpr int HROW = 128;                  // bytes per LDS row: 32 hi + 32 lo halfs
constexpr int HSTAGE = 2 * HT * HROW;      // A rows + B rows

__device__ __forceinline__ int lds_off(int row, int ch) { return row * HROW + ((ch ^ ((row >> 1) & 7)) << 4); }

// float32 rows -> split rows.  One thread per 8 consecutive k of one row.
__global__ __launch_bounds__(256) void k_split_rows(const float* __restrict__ p, int64_t ld, int64_t rows, int k8_per_row,
                                                    float scale, char* __restrict__ out, int64_t out_ld) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = idx / k8_per_row;
    const int c8 = (int)(idx % k8_per_row);
    if (r >= rows) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(p + r * ld + (int64_t)c8 * 8);
    const f32x4 v0 = src[0], v1 = src[1];
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = (e < 4 ? v0[e] : v1[e - 4]) * scale;
        const _Float16 h = (_Float16)a;
        hi[e] = h;
        lo[e] = (_Float16)(a - (float)h);
    }
    char* dst = out + r * out_ld + (int64_t)(c8 >> 2) * 128 + (c8 & 3) * 16;
    *reinterpret_cast<h8*>(dst) = hi;
    *reinterpret_cast<h8*>(dst + 64) = lo;
}

// LOWER: only tiles that touch the region col <= row + diag_shift are computed, and inside them only the 32x32
// sub-tiles whose 128-block column index <= 128-block row index (the same element set the float32 kernel writes).
template <bool LOWER>
__global__ __launch_bounds__(512) void k_gemm_nt_h3(float* C, int64_t ldc, const char* A, const char* B, int64_t ldp,
                                                    int m, int n, int tiles_m, int nk, float alpha, float beta,
                                                    int diag_shift) {
    __shared__ __attribute__((aligned(16))) char smem[2 * HSTAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int bi = blockIdx.x % tiles_m, bj = blockIdx.x / tiles_m;  // row tiles fastest: neighbours share the B rows
    if (LOWER && bj * HT > bi * HT + HT - 1 + diag_shift) return;

    const char* Ab = A + (int64_t)bi * HT * ldp;
    const char* Bb = B + (int64_t)bj * HT * ldp;
    const int lr = tid >> 3, ch = tid & 7;
    u32x4 ga[4], gb[4];
    auto load_tile = [&](int t) {
        const int64_t off = (int64_t)(nk - 1 - t) * 128 + ch * 16;
#pragma unroll
        for (int e = 0; e < 4; ++e) ga[e] = *reinterpret_cast<const u32x4*>(Ab + (int64_t)(lr + 64 * e) * ldp + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) gb[e] = *reinterpret_cast<const u32x4*>(Bb + (int64_t)(lr + 64 * e) * ldp + off);
    };
    auto store_tile = [&](int buf) {
        char* sa_ = smem + buf * HSTAGE;
        char* sb_ = sa_ + HT * HROW;
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x4*>(sa_ + lds_off(lr + 64 * e, ch)) = ga[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x4*>(sb_ + lds_off(lr + 64 * e, ch)) = gb[e];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
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
        const char* sa_ = smem + (t & 1) * HSTAGE;
        const char* sb_ = sa_ + HT * HROW;
#pragma unroll
        for (int kk = 1; kk >= 0; --kk) {
            h8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wr * 128 + i * 32 + frow;
                ah[i] = *reinterpret_cast<const h8*>(sa_ + lds_off(row, kk * 2 + fh));
                al[i] = *reinterpret_cast<const h8*>(sa_ + lds_off(row, 4 + kk * 2 + fh));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wc * 64 + j * 32 + frow;
                bh[j] = *reinterpret_cast<const h8*>(sb_ + lds_off(row, kk * 2 + fh));
                bl[j] = *reinterpret_cast<const h8*>(sb_ + lds_off(row, 4 + kk * 2 + fh));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (t + 1 < nk) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // epilogue: acc[i][j][r] is element (row, col) with row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
    const int row_base = bi * HT + wr * 128;
    const int col_base = bj * HT + wc * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r0 = row_base + i * 32, c0 = col_base + j * 32;
            if (r0 >= m || c0 >= n) continue;  // m, n are multiples of 128: a 32x32 sub-tile is inside or outside
            if (LOWER && (c0 >> 7) > ((r0 + diag_shift) >> 7)) continue;
            float* p0 = C + (int64_t)(r0 + 4 * fh) * ldc + c0 + frow;
            float cold[16];
            if (beta != 0.0f) {
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

int launch_split_rows(const float* p, int64_t ld, int64_t rows, int64_t k, float scale, char* out, int64_t out_ld,
                      hipStream_t s) {
    if (rows <= 0) return 0;
    NNGP_REQUIRE(k > 0 && k % 32 == 0 && ld % 4 == 0 && ((uintptr_t)p & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                     out_ld >= 4 * k && out_ld % 16 == 0,
                 "split_rows: k must be a multiple of 32 and the operands 16-byte aligned");
    const int64_t total = rows * (k / 8);
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, ld, rows, (int)(k / 8),
                       scale, out, out_ld);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// a, b: split rows (row stride ldp bytes); rows of a / b up to the next multiple of 256 must be readable (their
// products are never stored).  m, n multiples of 128; k a multiple of 32.
int launch_gemm_nt_h3(float* c, int64_t ldc, const char* a, const char* b, int64_t ldp, int64_t m, int64_t n, int64_t k,
                      float alpha, float beta, bool lower_only, int64_t diag_shift, hipStream_t s) {
    if (m <= 0 || n <= 0) return 0;
    NNGP_REQUIRE(m % 128 == 0 && n % 128 == 0 && k > 0 && k % 32 == 0 && diag_shift % 128 == 0 && diag_shift >= 0,
                 "gemm_nt_h3: m, n must be multiples of 128 and k of 32 (m=%lld n=%lld k=%lld)", (long long)m,
                 (long long)n, (long long)k);
    NNGP_REQUIRE(ldp >= 4 * k && ldp % 16 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ldc >= n,
                 "gemm_nt_h3: operands must be 16-byte aligned");
    const int64_t tm = (m + HT - 1) / HT, tn = (n + HT - 1) / HT;
    NNGP_REQUIRE(tm * tn < 2147483647LL && m < 2147483647LL && n < 2147483647LL, "gemm_nt_h3: grid too large");
    if (lower_only)
        hipLaunchKernelGGL(k_gemm_nt_h3<true>, dim3((unsigned)(tm * tn)), dim3(512), 0, s, c, ldc, a, b, ldp, (int)m,
                           (int)n, (int)tm, (int)(k / 32), alpha, beta, (int)diag_shift);
    else
        hipLaunchKernelGGL(k_gemm_nt_h3<false>, dim3((unsigned)(tm * tn)), dim3(512), 0, s, c, ldc, a, b, ldp, (int)m,
                           (int)n, (int)tm, (int)(k / 32), alpha, beta, 0);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nngp
