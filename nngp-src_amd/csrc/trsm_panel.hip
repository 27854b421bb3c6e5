// Fused panel triangular solve of the blocked Cholesky:  B[m, W] <- B * L^-T  in ONE launch.
//
// In the right-looking factorisation (potrf.hip; replaces cho_factor inside nt.predict, reference train.py:171-172) the
// rows below a factored W x W diagonal block (W <= 1024) are multiplied by its inverse transpose.  Round 1 did this as
// a recursion of float32 GEMMs down to the 128-wide inverted diagonal blocks: 15 dependent launches per 1024 columns,
// each too small to fill the GPU -- 0.35 .. 0.66 ms per block column at N = 32768, ~17 ms of a 57 ms factorisation, with
// the chip mostly idle (profiles/r2c timeline).  Here one workgroup owns 32 rows of B for the whole solve:
//
//   * the 32 x W row block lives in LDS (131 KB, row stride W + 4 floats: conflict-free 16-byte fragment reads);
//   * for each 128-column block j, left-looking:  T = B_j - X_{<j} L[j, <j]^T  (K = 128 j, float32 MFMA 32x32x2, one
//     32 x 32 accumulator per wave, K walked from the high end down like every Cholesky update here), then
//     X_j = T dinv_j^T with the inverted diagonal block the leaf kernel left behind -- the same arithmetic as the recursion;
//   * L and dinv stream from L2 through a 128 x 32 staging tile (XOR-swizzled 128-byte rows; the next tile waits in registers);
//   * 8 waves: two per SIMD, each pair splitting the k range of a tile (see the kernel);
//   * each finished X_j goes back to global memory in float32 AND, optionally, as the float16 hi/lo "split rows" the
//     trailing update on the float16 pipe reads (gemm_h3.hip) -- the separate k_split_rows passes disappear.
#include "common.h"

namespace nngp {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int PR = 32;            // rows of B per workgroup
constexpr int PWMAX = 1024;       // widest panel
constexpr int PXS = PWMAX + 4;    // LDS row stride of the row block (floats)
constexpr int PBK = 32;           // k per staging tile: waves 0-3 take its first 16 k, waves 4-7 the second (split-K)

__device__ __forceinline__ int stage_off(int row, int ch) { return row * PBK + ((ch ^ ((row >> 1) & 7)) << 2); }

// 512 threads = 8 waves = two waves per SIMD: wave (g, c) accumulates output columns [32 c, 32 c + 32) of the current
// 128-column block over the k16 half g of every staged 32-k tile, so each SIMD's matrix pipe is fed by two independent
// accumulation chains and one wave's LDS latency hides behind the other's MFMAs (one wave per SIMD measured 125-160 us per
// workgroup against 70 us of MFMA issue time); the two partial sums meet in LDS when the block is done.
__global__ __launch_bounds__(512) void k_trsm_panel_f32(float* __restrict__ b, int64_t ldb, const float* __restrict__ l,
                                                        int64_t ldl, const float* __restrict__ dinv, int w,
                                                        char* __restrict__ planes, int64_t ldp, float scale) {
    __shared__ __attribute__((aligned(16))) float Xs[PR * PXS];
    __shared__ __attribute__((aligned(16))) float Ls[128 * PBK];  // staged L / dinv tile; partial sums of group 1 between phases
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wc = wave & 3;
    const int frow = lane & 31, fh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * PR;
    float* bg = b + row0 * ldb;

    // the row block: 32 rows x w floats, 16-byte loads along the row
    const int w4 = w >> 2;
    for (int idx = tid; idx < PR * w4; idx += 512) {
        const int row = idx / w4, c4 = idx - row * w4;
        *reinterpret_cast<f32x4*>(&Xs[row * PXS + c4 * 4]) = *reinterpret_cast<const f32x4*>(bg + (int64_t)row * ldb + c4 * 4);
    }

    // acc += Xs[:, a_col0 + (this group's k16 halves of) 32 nk] * Bg[128 rows, 32 nk]^T, k-tiles from the high end down.
    // Single staging buffer, next tile prefetched into registers: two barriers per tile.  Ends with a barrier (Ls free).
    auto mac = [&](f32x16& acc, int a_col0, const float* Bg, int64_t ldbg, int nk) {
        if (nk <= 0) return;
        f32x4 g[2];
        auto load_tile = [&](int t) {
            const int k0 = (nk - 1 - t) * PBK;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int id = tid + 512 * e, row = id >> 3, ch = id & 7;
                g[e] = *reinterpret_cast<const f32x4*>(Bg + (int64_t)row * ldbg + k0 + ch * 4);
            }
        };
        auto store_tile = [&]() {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int id = tid + 512 * e, row = id >> 3, ch = id & 7;
                *reinterpret_cast<f32x4*>(&Ls[stage_off(row, ch)]) = g[e];
            }
        };
        load_tile(0);
        store_tile();
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            if (t + 1 < nk) load_tile(t + 1);
            const int k0 = (nk - 1 - t) * PBK + 16 * grp;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ch = 2 * s + fh;  // 16-byte chunk inside this group's k16 half
                const f32x4 fa = *reinterpret_cast<const f32x4*>(&Xs[frow * PXS + a_col0 + k0 + ch * 4]);
                const f32x4 fb = *reinterpret_cast<const f32x4*>(&Ls[stage_off(32 * wc + frow, 4 * grp + ch)]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc, 0, 0, 0);
            }
            __syncthreads();  // every wave has read the tile
            if (t + 1 < nk) {
                store_tile();
                __syncthreads();
            }
        }
    };
    // group 1 hands its partial sums to group 0 through Ls ([128 columns][32 rows], conflict-free both ways)
    auto reduce = [&](f32x16& acc) {
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Ls[(32 * wc + frow) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh] = acc[r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += Ls[(32 * wc + frow) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];
        }
    };

    const int nblk = w / 128;
    for (int j = 0; j < nblk; ++j) {
        __syncthreads();  // the row block (first pass) / the previous block's X_j are in place; Ls is free
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const int cj = 128 * j + 32 * wc + frow;  // this lane's column of the row block
        // T = B_j - X_{<j} L[j, <j]^T
        if (j > 0) {
            mac(acc, 0, l + (int64_t)(128 * j) * ldl, ldl, (128 * j) / PBK);
            reduce(acc);
            if (grp == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Xs[((r & 3) + 8 * (r >> 2) + 4 * fh) * PXS + cj] -= acc[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            __syncthreads();  // T complete in columns [128 j, 128 j + 128); Ls free again
        }
        // X_j = T dinv_j^T
        mac(acc, 128 * j, dinv + (int64_t)j * 128 * 128, 128, 128 / PBK);
        reduce(acc);  // (its barrier also says: every wave has finished reading T)
        if (grp == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Xs[((r & 3) + 8 * (r >> 2) + 4 * fh) * PXS + cj] = acc[r];
        }
        __syncthreads();
        // X_j to global: float32 rows and, if asked for, their float16 hi / lo split (row = [k/32][32 hi | 32 lo] halfs)
        {
            const int row = tid >> 4, c8 = tid & 15;
            const int kq = 128 * j + 8 * c8;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Xs[row * PXS + kq]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Xs[row * PXS + kq + 4]);
            float* dst = bg + (int64_t)row * ldb + kq;
            *reinterpret_cast<f32x4*>(dst) = v0;
            *reinterpret_cast<f32x4*>(dst + 4) = v1;
            if (planes != nullptr) {
                h8 hi, lo;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float a = (q < 4 ? v0[q] : v1[q - 4]) * scale;
                    const _Float16 h = (_Float16)a;
                    hi[q] = h;
                    lo[q] = (_Float16)(a - (float)h);
                }
                char* pd = planes + (row0 + row) * ldp + (int64_t)(kq >> 5) * 128 + ((kq & 31) >> 3) * 16;
                *reinterpret_cast<h8*>(pd) = hi;
                *reinterpret_cast<h8*>(pd + 64) = lo;
            }
        }
    }
}

}  // namespace

// b [m, w] <- b * L^-T for the w x w lower block at `l` with its inverted 128-blocks `dinv` (w/128 blocks of 128 x 128).
// m must be a multiple of 32, w a multiple of 128 up to 1024.  planes != NULL: also write the rows' float16 split copy,
// row i of b at planes + i * ldp (scale as in launch_split_rows).
int launch_trsm_panel_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ldl, const float* dinv, int64_t w,
                          char* planes, int64_t ldp, float scale, hipStream_t s) {
    if (m <= 0 || w <= 0) return 0;
    NNGP_REQUIRE(m % PR == 0 && w % 128 == 0 && w <= PWMAX, "trsm_panel: m must be a multiple of 32, w of 128 and <= 1024");
    NNGP_REQUIRE(ldb % 4 == 0 && ldl % 4 == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)l & 15) == 0 && ((uintptr_t)dinv & 15) == 0,
                 "trsm_panel: operands must be 16-byte aligned");
    NNGP_REQUIRE(planes == nullptr || (((uintptr_t)planes & 15) == 0 && ldp % 16 == 0 && ldp >= 4 * w),
                 "trsm_panel: split rows must be 16-byte aligned with ldp >= 4 w");
    NNGP_REQUIRE(m / PR < 2147483647LL, "trsm_panel: too many rows");
    hipLaunchKernelGGL(k_trsm_panel_f32, dim3((unsigned)(m / PR)), dim3(512), 0, s, b, ldb, l, ldl, dinv, (int)w, planes, ldp, scale);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nngp
