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
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int PR = 32;            // rows of B per workgroup
constexpr int PWMAX = 1024;       // widest panel
constexpr int PXS = PWMAX + 4;    // LDS row stride of the row block (floats)
constexpr int PBK = 32;           // k per staging tile: waves 0-3 take its first 16 k, waves 4-7 the second (split-K)

__device__ __forceinline__ int stage_off(int row, int ch) { return row * PBK + ((ch ^ ((row >> 1) & 7)) << 2); }

// 512 threads = 8 waves = two waves per SIMD: wave (g, c) accumulates output columns [32 c, 32 c + 32) of the current
// 128-column block over the k16 half g of every staged 32-k tile, so each SIMD's matrix pipe is fed by two independent
// accumulation chains and one wave's LDS latency hides behind the other's MFMAs (one wave per SIMD measured 125-160 us per
// workgroup against 70 us of MFMA issue time); the two partial sums meet in LDS when the block is done.
// planes_t (optional): the rows' TRANSPOSED split copy as well -- element (global row i, global column c) of the solved panel goes to
// planes_t + (i / 1024) * tstride + c * ldp, k = i % 1024: the operand of the posterior's "B L^-1" solves (SplitWork::planes_t), which
// a separate pass over the finished factor used to write (k_split_lower_t: 2 x 2.1 GB at N = 32768, beside the posterior's first
// solve).  A workgroup's 32 rows are one 32-k block of 128 columns' rows: one 128-byte line per column and 128-column block.
struct TPlanes {
    char* base;        // SplitWork::planes_t (NULL: not written)
    int64_t tstride;   // bytes between block rows (SplitWork::col_stride)
    int64_t row0, col0;  // global row of the launch's first row, global column of the panel's first column
};

__device__ __forceinline__ void write_planes_t(const TPlanes& tp, int64_t ldp, float scale, const float* blk, int blk_stride, int64_t grow0,
                                               int64_t gcol0, int tid) {
    // blk: the 32 x 128 block of solved rows in LDS (row stride blk_stride floats); thread -> (column c, group of 8 rows)
    const int c = tid & 127, part = tid >> 7;
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = blk[(8 * part + e) * blk_stride + c] * scale;
        const _Float16 h = (_Float16)a;
        hi[e] = h;
        lo[e] = (_Float16)(a - (float)h);
    }
    const int64_t jrow = grow0 >> 10;
    const int kb = (int)((grow0 & 1023) >> 5);
    char* dst = tp.base + jrow * tp.tstride + (gcol0 + c) * ldp + (int64_t)kb * 128 + part * 16;
    *reinterpret_cast<h8*>(dst) = hi;
    *reinterpret_cast<h8*>(dst + 64) = lo;
}

__global__ __launch_bounds__(512) void k_trsm_panel_f32(float* __restrict__ b, int64_t ldb, const float* __restrict__ l,
                                                        int64_t ldl, const float* __restrict__ dinv, int w,
                                                        char* __restrict__ planes, int64_t ldp, float scale, TPlanes tp) {
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
    // group 1 hands its partial sums to group 0 through Ls ([32 rows][128 columns]: lanes along a row)
    auto reduce = [&](f32x16& acc) {
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Ls[((r & 3) + 8 * (r >> 2) + 4 * fh) * 128 + 32 * wc + frow] = acc[r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += Ls[((r & 3) + 8 * (r >> 2) + 4 * fh) * 128 + 32 * wc + frow];
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
        if (tp.base != nullptr) write_planes_t(tp, ldp, scale, Xs + 128 * j, PXS, tp.row0 + row0, tp.col0 + 128 * j, tid);
    }
}


// ---- the same solve with the left-looking products on the float16 matrix pipe --------------------------------------------
// T = B_j - X_{<j} L[j, <j]^T is 7/9 of the solve's flops at w = 1024.  Here X_{<j} is kept in LDS as float16 hi/lo split rows
// (the format the solved rows are written in anyway, gemm_h3.hip) and L streams in from the split copy of the diagonal block,
// so a 32-k step costs a wave 3 v_mfma_f32_32x32x16_f16 (96 cycles) instead of 16 float32 MFMAs (1024 cycles).  The
// multiplication by the inverted diagonal block, X_j = T dinv_j^T, which sets the accuracy of the result directly, stays
// float32.  Operand rounding of the products: 22 significant bits, as in the trailing updates.
#ifdef NNGP_TIMING_KNOBS
__device__ unsigned long long g_trsm_stamps[8];
__device__ unsigned long long g_trsm_wave[16];   // per wave of workgroup 0: cycles from block start to the end of its phase A; [8+w]: to T complete  // cycles of workgroup 0 / thread 0 per phase (timing study, knob 7 = 9)
#define TRSM_STAMP(i) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                           atomicAdd(&g_trsm_stamps[i], now_ - last_); last_ = now_; } } while (0)
#else
#define TRSM_STAMP(i) do { } while (0)
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter (vmcnt(0)), i.e. it
// would wait for the NEXT block's operand loads that this kernel keeps in flight across its barriers on purpose
// (measured: 6.3k cycles per block lost there).  Global stores need no ordering inside the kernel.
#define LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
constexpr int HXS = (PWMAX - 128) * 4 + 16;  // bytes per row of the split row block (blocks 0..6; the last block is never an operand)
constexpr int TFS = 132;                     // floats per row of the float32 block buffer

__device__ __forceinline__ int lds_off16(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// Split copy of a diagonal block in the order k_trsm_panel_h3's waves consume it: for (128-row block jb, 32-k block kb, 32-row
// group wc, k16 half g) one contiguous 2 KB piece = [64 lanes x 16 bytes of hi halfs | the same of lo halfs], lane = row % 32 +
// 32 * (k8 group & 1) -- so that a wave's operand load is ONE fully coalesced kilobyte.  (Loading the same fragments from
// row-major split rows costs 32 cache lines per instruction; measured 760-1480 cycles per 32-k step, TCP-bound.)
// The same launch re-orders the inverted 128-blocks the same way -- round 5: as split halves too, every ROW of an inverted block (an
// output column of the panel solve) with its own power-of-two scale (largest entry -> [2^13, 2^14); dscale[row] = 1 / scale): piece
// (jb, t, wc, g) = [64 lanes x 16 bytes of hi halfs | the same of lo halfs], lane (row % 32, fh) holding
// dinv_jb[32 wc + row % 32][(3 - t) 32 + 16 g + 8 fh .. + 7] -- the operand of phase B on the float16 pipe.
__global__ __launch_bounds__(256) void k_split_diag_frag(const float* __restrict__ a, int64_t ld, int w, float scale,
                                                         char* __restrict__ out, const float* __restrict__ dinv,
                                                         char* __restrict__ dfrag, float* __restrict__ dscale) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int k8_per_row = w >> 3;
    const int r = idx / k8_per_row, c8 = idx - r * k8_per_row;
    if (r >= w) return;
    if (c8 < 16) {  // the 128 x 128 inverse of this row's diagonal block: 16 groups of 8 k per row = 16 consecutive lanes of one wave
        const int jb = r >> 7, cjr = r & 127, wcr = cjr >> 5, fr = cjr & 31;
        const float* srcd = dinv + (int64_t)jb * 128 * 128 + cjr * 128 + c8 * 8;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(srcd), d1 = *reinterpret_cast<const f32x4*>(srcd + 4);
        float mx = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(d0[e]), fabsf(d1[e])));
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        int e2 = 0;
        (void)frexpf(mx, &e2);
        const float sc = (mx > 0.0f && mx < 3.0e38f) ? ldexpf(1.0f, 14 - e2) : 1.0f;
        if (c8 == 0) dscale[r] = 1.0f / sc;
        h8 dhi, dlo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e < 4 ? d0[e] : d1[e - 4]) * sc;
            const _Float16 h = (_Float16)x;
            dhi[e] = h;
            dlo[e] = (_Float16)(x - (float)h);
        }
        // k = 8 c8 + e  ->  t = 3 - (k >> 5), g = (k >> 4) & 1, fh = (k >> 3) & 1
        const int k0 = 8 * c8, t = 3 - (k0 >> 5), g = (k0 >> 4) & 1, fhd = (k0 >> 3) & 1;
        char* piece = dfrag + ((((int64_t)jb * 4 + t) * 4 + wcr) * 2 + g) * 2048 + (fr + 32 * fhd) * 16;
        *reinterpret_cast<h8*>(piece) = dhi;
        *reinterpret_cast<h8*>(piece + 1024) = dlo;
    }
    const f32x4* src = reinterpret_cast<const f32x4*>(a + (int64_t)r * ld + c8 * 8);
    const f32x4 v0 = src[0], v1 = src[1];
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (e < 4 ? v0[e] : v1[e - 4]) * scale;
        const _Float16 h = (_Float16)x;
        hi[e] = h;
        lo[e] = (_Float16)(x - (float)h);
    }
    const int jb = r >> 7, wc = (r >> 5) & 3, frow = r & 31;
    const int kb = c8 >> 2, within = c8 & 3, g = within >> 1, fh = within & 1;
    const int64_t piece = ((((int64_t)jb * (w >> 5) + kb) * 4 + wc) * 2 + g) * 2048;
    char* dst = out + piece + (frow + 32 * fh) * 16;
    *reinterpret_cast<h8*>(dst) = hi;
    *reinterpret_cast<h8*>(dst + 1024) = lo;
}

__global__ __launch_bounds__(512) void k_trsm_panel_h3(float* __restrict__ b, int64_t ldb, const char* __restrict__ lsplit,
                                                       const char* __restrict__ dfrag, const float* __restrict__ dscale, int w,
                                                       char* __restrict__ planes, int64_t ldp, float scale, TPlanes tp) {
    __shared__ __attribute__((aligned(16))) char Xp[PR * HXS];     // solved blocks, split rows: [row][k / 32][32 hi | 32 lo]
    __shared__ __attribute__((aligned(16))) float Tf[PR * TFS];    // current block in float32: B_j, then T, then X_j
    __shared__ __attribute__((aligned(16))) float Lf[PR * TFS];    // partial sums of wave group 1, laid out like Tf (lanes along a row:
                                                                   // [column][row] put all 32 lanes of a store on two banks)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wc = wave & 3;
    const int frow = lane & 31, fh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * PR;
    float* bg = b + row0 * ldb;
    const float inv_s2 = 1.0f / (scale * scale);
    // Phase B on the float16 pipe (round 5): T = X_j L_jj^T is bounded by max A <= (max |L|)^2, and max |L| scale < 2^15: with
    // scale_t = scale^2 / 2^16 every entry of T scale_t stays below 2^14 -- one constant power of two, no pass over T for a scale
    const float scale_t = scale * scale * (1.0f / 65536.0f);
    const int crow = tid >> 4, cc8 = tid & 15;  // this thread's (row, group of 8 columns) in the cooperative passes
    const int cj = 32 * wc + frow;              // this lane's output column inside a 128-column block = its row of L / dinv
    // A wave's B operand is private to it (rows 32 wc .. 32 wc + 31 of the L / dinv block, the k16 half `grp`): it is loaded
    // straight from L2 into registers, four k-blocks ahead, with no LDS staging and NO workgroup barrier inside the products --
    // the staged form (a barrier pair and an exposed L2 round trip per 32-k step) ran 115 us per workgroup for 10 us of MFMAs.
    const int chi = (grp * 2 + fh) * 16, clo = (4 + grp * 2 + fh) * 16;  // this lane's hi / lo chunk inside a 128-byte k-block of Xp
    const int nkb = w >> 5;                                              // k-blocks per row of the diagonal block

    const int nblk = w / 128;
    // Operands of block j + 1 that do not depend on block j -- its rows of B, this wave's part of dinv_{j+1}, the first group of
    // L[j+1, :] fragments -- are requested while block j is still being finished, so that no L2 round trip sits between blocks.
    f32x4 b0, b1;
    h8 dvh[4], dvl[4];                // this wave's fragments of the inverted 128-block (split halves), four k16 steps
    float dsc = 1.0f;                 // 1 / scale of its row of that block (= this lane's output column)
    h8 nh[4], nl[4], nh2[4], nl2[4];  // L fragments one and two groups (of four 32-k blocks) ahead of the MFMAs
    auto load_block_inputs = [&](int jj) {
        b0 = *reinterpret_cast<const f32x4*>(bg + (int64_t)crow * ldb + 128 * jj + 8 * cc8);
        b1 = *reinterpret_cast<const f32x4*>(bg + (int64_t)crow * ldb + 128 * jj + 8 * cc8 + 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const char* piece = dfrag + ((((int64_t)jj * 4 + t) * 4 + wc) * 2 + grp) * 2048 + lane * 16;
            dvh[t] = *reinterpret_cast<const h8*>(piece);
            dvl[t] = *reinterpret_cast<const h8*>(piece + 1024);
        }
        dsc = dscale[128 * jj + cj];
    };
    // fragments of L[jj, :] for k-blocks kb_hi .. kb_hi - 3 (k_split_diag_frag order) into (dh, dl)
    auto load_group = [&](h8 (&dh)[4], h8 (&dl)[4], int jj, int kb_hi) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const char* piece = lsplit + ((((int64_t)jj * nkb + (kb_hi - u)) * 4 + wc) * 2 + grp) * 2048 + lane * 16;
            dh[u] = *reinterpret_cast<const h8*>(piece);
            dl[u] = *reinterpret_cast<const h8*>(piece + 1024);
        }
    };
    load_block_inputs(0);
#ifdef NNGP_TIMING_KNOBS
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    for (int j = 0; j < nblk; ++j) {
#ifdef NNGP_TIMING_KNOBS
        const unsigned long long wstart_ = __builtin_amdgcn_s_memtime();
#endif
        const f32x4 cb0 = b0, cb1 = b1;
        h8 cdh[4], cdl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { cdh[t] = dvh[t]; cdl[t] = dvl[t]; }
        const float cds = dsc * (1.0f / scale_t);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // ---- phase A: acc = X_{<j} L[j, <j]^T (times scale^2), split operands, k-blocks from the high end down ----
        if (j > 0) {
            const int nk = 4 * j;
            for (int g = 0; g < j; ++g) {
                h8 ch[4], cl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { ch[u] = nh[u]; cl[u] = nl[u]; nh[u] = nh2[u]; nl[u] = nl2[u]; }
                // two groups ahead: an L2 miss on the freshly written split copy costs ~1500 cycles, a group of MFMAs ~600
                if (g + 2 < j) load_group(nh2, nl2, j, nk - 1 - 4 * (g + 2));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const char* xa = Xp + frow * HXS + (nk - 1 - 4 * g - u) * 128;
                    const h8 ah = *reinterpret_cast<const h8*>(xa + chi);
                    const h8 al = *reinterpret_cast<const h8*>(xa + clo);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cl[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ch[u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ch[u], acc, 0, 0, 0);
                }
            }
        }
        TRSM_STAMP(0);  // phase A
#ifdef NNGP_TIMING_KNOBS
        if (blockIdx.x == 0 && lane == 0) atomicAdd(&g_trsm_wave[wave], __builtin_amdgcn_s_memtime() - wstart_);
#endif
        if (j + 1 < nblk) {  // the next block's inputs, in flight through the rest of this block
            load_block_inputs(j + 1);
            load_group(nh, nl, j + 1, 4 * (j + 1) - 1);
            if (j + 1 >= 2) load_group(nh2, nl2, j + 1, 4 * (j + 1) - 5);
        }
        // B_j into Tf (Tf was drained by the previous block's write-out; no one reads it before the barrier below)
        *reinterpret_cast<f32x4*>(&Tf[crow * TFS + 8 * cc8]) = cb0;
        *reinterpret_cast<f32x4*>(&Tf[crow * TFS + 8 * cc8 + 4]) = cb1;
        if (j > 0) {
            if (grp == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) Lf[((r & 3) + 8 * (r >> 2) + 4 * fh) * TFS + cj] = acc[r];
            }
            LDS_BARRIER();
            if (grp == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                    Tf[row * TFS + cj] -= (acc[r] + Lf[row * TFS + cj]) * inv_s2;
                }
            }
        }
        LDS_BARRIER();  // T complete
        TRSM_STAMP(1);  // B_j wait + reduce + T
#ifdef NNGP_TIMING_KNOBS
        if (blockIdx.x == 0 && lane == 0) atomicAdd(&g_trsm_wave[8 + wave], __builtin_amdgcn_s_memtime() - wstart_);
#endif
        // ---- phase B: X_j = T dinv_j^T on the float16 pipe too (round 5; until then float32 MFMAs: 60 % of the kernel's matrix time):
        // split-K over the two wave groups, the block's fragments already in registers; a lane's 8 consecutive k of T are split as
        // they are read (hi + lo of T scale_t), three products per k16 step, the row scales of the inverted block come back per lane ----
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k0 = (3 - t) * 32 + 16 * grp + 8 * fh;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(&Tf[frow * TFS + k0]);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(&Tf[frow * TFS + k0 + 4]);
            h8 th, tl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (e < 4 ? t0[e] : t1[e - 4]) * scale_t;
                const _Float16 h = (_Float16)x;
                th[e] = h;
                tl[e] = (_Float16)(x - (float)h);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, cdl[t], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tl, cdh[t], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, cdh[t], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= cds;
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Lf[((r & 3) + 8 * (r >> 2) + 4 * fh) * TFS + cj] = acc[r];
        }
        TRSM_STAMP(2);  // phase B MFMAs
        LDS_BARRIER();  // group 1's sums are in place AND every wave has finished reading T
        if (grp == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                Tf[row * TFS + cj] = acc[r] + Lf[row * TFS + cj];
            }
        }
        LDS_BARRIER();
        TRSM_STAMP(3);  // reduce + X
        // ---- X_j out: float32 rows, their split copy in global memory, and the split copy in LDS for the later blocks ----
        {
            const int kq = 128 * j + 8 * cc8;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&Tf[crow * TFS + 8 * cc8]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&Tf[crow * TFS + 8 * cc8 + 4]);
            float* dst = bg + (int64_t)crow * ldb + kq;
            *reinterpret_cast<f32x4*>(dst) = v0;
            *reinterpret_cast<f32x4*>(dst + 4) = v1;
            h8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float a = (q < 4 ? v0[q] : v1[q - 4]) * scale;
                const _Float16 h = (_Float16)a;
                hi[q] = h;
                lo[q] = (_Float16)(a - (float)h);
            }
            const int off = (kq >> 5) * 128 + ((kq & 31) >> 3) * 16;
            if (planes != nullptr) {
                char* pd = planes + (row0 + crow) * ldp + off;
                *reinterpret_cast<h8*>(pd) = hi;
                *reinterpret_cast<h8*>(pd + 64) = lo;
            }
            if (j + 1 < nblk) {
                *reinterpret_cast<h8*>(Xp + crow * HXS + off) = hi;
                *reinterpret_cast<h8*>(Xp + crow * HXS + off + 64) = lo;
            }
        }
        if (tp.base != nullptr) write_planes_t(tp, ldp, scale, Tf, TFS, tp.row0 + row0, tp.col0 + 128 * j, tid);  // Tf holds X_j (read-only here)
        LDS_BARRIER();  // Tf, Lf and Xp settled before the next block reads / overwrites them
        TRSM_STAMP(4);  // write-out
    }
}

}  // namespace

// b [m, w] <- b * L^-T for the w x w lower block at `l` with its inverted 128-blocks `dinv` (w/128 blocks of 128 x 128).
// m must be a multiple of 32, w a multiple of 128 up to 1024.  planes != NULL: also write the rows' float16 split copy,
// row i of b at planes + i * ldp (scale as in launch_split_rows).
static TPlanes make_tplanes(char* planes_t, int64_t tstride, int64_t row0, int64_t col0) {
    TPlanes tp;
    tp.base = planes_t; tp.tstride = tstride; tp.row0 = row0; tp.col0 = col0;
    return tp;
}

int launch_trsm_panel_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ldl, const float* dinv, int64_t w,
                          char* planes, int64_t ldp, float scale, hipStream_t s, char* planes_t, int64_t tstride, int64_t row0,
                          int64_t col0) {
    if (m <= 0 || w <= 0) return 0;
    NNGP_REQUIRE(m % PR == 0 && w % 128 == 0 && w <= PWMAX, "trsm_panel: m must be a multiple of 32, w of 128 and <= 1024");
    NNGP_REQUIRE(ldb % 4 == 0 && ldl % 4 == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)l & 15) == 0 && ((uintptr_t)dinv & 15) == 0,
                 "trsm_panel: operands must be 16-byte aligned");
    NNGP_REQUIRE(planes == nullptr || (((uintptr_t)planes & 15) == 0 && ldp % 16 == 0 && ldp >= 4 * w),
                 "trsm_panel: split rows must be 16-byte aligned with ldp >= 4 w");
    NNGP_REQUIRE(m / PR < 2147483647LL, "trsm_panel: too many rows");
    NNGP_REQUIRE(planes_t == nullptr || (planes != nullptr && row0 % 32 == 0 && col0 % 128 == 0 && tstride % 16 == 0 && ((uintptr_t)planes_t & 15) == 0),
                 "trsm_panel: the transposed split copy needs the plain one, rows at multiples of 32 and columns at multiples of 128");
    hipLaunchKernelGGL(k_trsm_panel_f32, dim3((unsigned)(m / PR)), dim3(512), 0, s, b, ldb, l, ldl, dinv, (int)w, planes, ldp, scale,
                       make_tplanes(planes_t, tstride, row0, col0));
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nngp

namespace nngp {
// Split copy of the w x w diagonal block at `a` in the fragment order of k_trsm_panel_h3 (4 w^2 bytes at `out`).
int launch_split_diag_frag(const float* a, int64_t ld, int64_t w, float scale, char* out, const float* dinv, float* dfrag,
                           float* dscale, hipStream_t s) {
    NNGP_REQUIRE(w > 0 && w % 128 == 0 && w <= PWMAX && ld % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                     dinv != nullptr && dfrag != nullptr && dscale != nullptr && ((uintptr_t)dinv & 15) == 0 && ((uintptr_t)dfrag & 15) == 0,
                 "split_diag_frag: w must be a multiple of 128 up to 1024, operands 16-byte aligned");
    const int64_t total = w * (w / 8);
    hipLaunchKernelGGL(k_split_diag_frag, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, ld, (int)w, scale, out, dinv,
                       reinterpret_cast<char*>(dfrag), dscale);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// b [m, w] <- b * L^-T with the left-looking products on the float16 pipe.  lfrag: launch_split_diag_frag of the diagonal block
// (same scale) -- only its strictly lower 128-blocks are read; dfrag: the inverted 128-blocks in the order the same launch wrote them.
int launch_trsm_panel_h3(float* b, int64_t ldb, int64_t m, const char* lfrag, const float* dinv, const float* dscale, int64_t w,
                         char* planes, int64_t ldp, float scale, hipStream_t s, char* planes_t, int64_t tstride, int64_t row0, int64_t col0) {
    if (m <= 0 || w <= 0) return 0;
    NNGP_REQUIRE(m % PR == 0 && w % 128 == 0 && w <= PWMAX, "trsm_panel_h3: m must be a multiple of 32, w of 128 and <= 1024");
    NNGP_REQUIRE(ldb % 4 == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)dinv & 15) == 0 && dscale != nullptr && lfrag != nullptr &&
                     ((uintptr_t)lfrag & 15) == 0 && ldp % 16 == 0 && ldp >= 4 * w && scale > 0.0f,
                 "trsm_panel_h3: operands must be 16-byte aligned with ldp >= 4 w");
    NNGP_REQUIRE(planes == nullptr || ((uintptr_t)planes & 15) == 0, "trsm_panel_h3: split rows must be 16-byte aligned");
    NNGP_REQUIRE(planes_t == nullptr || (row0 % 32 == 0 && col0 % 128 == 0 && tstride % 16 == 0 && ((uintptr_t)planes_t & 15) == 0),
                 "trsm_panel_h3: the transposed split copy needs rows at multiples of 32 and columns at multiples of 128");
    hipLaunchKernelGGL(k_trsm_panel_h3, dim3((unsigned)(m / PR)), dim3(512), 0, s, b, ldb, lfrag, reinterpret_cast<const char*>(dinv), dscale,
                       (int)w, planes, ldp, scale, make_tplanes(planes_t, tstride, row0, col0));
    NNGP_HIP_CHECK(hipGetLastError());
#ifdef NNGP_TIMING_KNOBS
    if (NNGP_KNOB(7) == 9) {  // timing study: cycles of workgroup 0 per phase, summed over the launches so far
        unsigned long long h[8];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trsm_stamps), sizeof(h)) == hipSuccess)
            fprintf(stderr, "trsm_panel_h3 stamps (m=%lld): phaseA %llu  wait+reduce+T %llu  phaseB %llu  reduce+X %llu  writeout %llu\n",
                    (long long)m, h[0], h[1], h[2], h[3], h[4]);
        unsigned long long hw[16];
        if (hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_trsm_wave), sizeof(hw)) == hipSuccess) {
            fprintf(stderr, "   per wave, block start -> end of phase A:");
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %llu", hw[i]);
            fprintf(stderr, "\n   per wave, block start -> T complete:");
            for (int i = 8; i < 16; ++i) fprintf(stderr, " %llu", hw[i]);
            fprintf(stderr, "\n");
        }
    }
#endif
    return 0;
}
}  // namespace nngp
