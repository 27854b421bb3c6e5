// One persistent kernel per blocked triangular solve of the posterior (the cho_solve inside predict_fn,
// reference call site train.py:157-158):  B[m, np] <- B L^-T (forward) or B L^-1 (backward), float32-grade, on the float16
// matrix pipe.  Rounds 1-4 ran each solve as np / 1024 dependent steps of {float32 diagonal product, row split, update launch}:
// ~100 launches whose update rounds were 6-100 % full (DESIGN_NOTES R4-Z).  Here the whole solve is ONE launch whose workgroups
// draw work items from a ticket counter, in an order fixed on the host:
//
//   SB(r, J, q)  split copy (hi + lo float16, one power-of-two scale per row) of 16 rows of the fully updated block B_J[r]
//   D (r, c)     X[r, c] = sum_k B_J[r, k] Linv_J[c, k]      128 x 128 tile of the diagonal product, K <= 1024, triangular
//   SX(r, J, q)  split copy of 16 rows of X_J[r]
//   U (r, c, P)  B[r, c] -= sum_{J in P} X_J[r] L[c, J]^T     128 x 128 tile, K = 1024 |P|, up to four finished block columns per pass
//   UB            the same for a 2 x 2 group of tiles as one 256 x 256 tile (the bulk of the flops)
//   UW(r, c, J)  B[r, c] -= B_J[r] W_J[c]^T, W_J = L[next block, J] Linv_J: the LAST update of the next block column straight from the
//                split rows of B_J -- X_J = B_J Linv_J^T and its split leave the dependency chain (they still feed the other updates)
//
// r = tile of 128 right-hand-side rows, c = tile of 128 columns, J = block column of 1024.  Different r never interact, so the
// dependencies are four monotone counters per r (device memory, zeroed before the launch): updates applied to tile (r, c),
// split rows of B_J[r] written, diagonal tiles of X_J[r] done, split rows of X_J[r] written.
//
// Hang safety by construction: an item waits only for items with a LOWER ticket (the host's list scheduler starts an item only
// after the items it depends on have finished in its simulation, so they precede it in the table).  Every lower ticket is held by a
// workgroup that is already running or has finished, so the lowest unfinished ticket can always proceed: no co-residency
// assumption, any grid size.  Every spin loop is bounded in wall-clock time (s_memrealtime); on expiry the workgroup sets the
// error word and leaves, every other workgroup sees the word at its next poll or ticket and leaves too, and the host reports
// NNGP_ERR through nngp_last_error at its next call -- a bug costs a red test, not the box.
//
// Visibility between workgroups (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): producer =
// write-through (sc1) stores, every wave s_waitcnt vmcnt(0), workgroup barrier, one lane: agent-scope release fence, vmcnt(0),
// agent-scope atomic add on the counter; consumer = one wave polls with agent-scope relaxed loads, agent-scope acquire fence,
// vmcnt(0), workgroup barrier, then plain loads and LDS-DMA.
//
// Tile arithmetic: the 128 x 128 form of the split-float16 product (256 threads = 4 waves 2 x 2, wave sub-tile 64 x 64 as 4 x 4
// accumulators of v_mfma_f32_16x16x32_f16 with swapped operands, 2 x 32 KB LDS stages by global_load_lds_dwordx4, one barrier per
// stage), two workgroups per compute unit so that one's waits, epilogue and split items run under the other's MFMAs.
#include <stdlib.h>
#include <unistd.h>

#include <algorithm>
#include <functional>
#include <queue>
#include <vector>

#include "common.h"

namespace nngp {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int TT = 128;                  // tile edge
constexpr int TROW = 128;                // bytes per LDS row: 32 hi + 32 lo halfs
constexpr int TSTAGE = 2 * TT * TROW;    // A rows + B rows: 32 KB
constexpr int64_t kLdp = 4096;           // bytes per split row (1024 k x 4 bytes), every plane buffer
constexpr int kBs = 1024;                // block-column width
constexpr int kQ = 4;                    // split items per (r, J): 32 rows each (8 waves x 4 rows)

enum { IT_SB = 0, IT_D = 1, IT_SX = 2, IT_U = 3, IT_UB = 4, IT_UW = 5 };
enum { SY_TICKET = 0, SY_ERROR = 1, SY_TICKETQ = 2, SY_COUNTERS = 16 };  // SY_TICKETQ .. + 7: one ticket counter per queue
constexpr int kMaxQ = 8;

struct TkParams {
    float* b;                 // right-hand sides [mt * 128, np], row stride ldb
    int64_t ldb;
    const char* lplanes;      // split copy of L by block column (forward) / of L^T by block row (backward)
    int64_t l_stride;         // bytes between block columns
    const char* dinv_s;       // split copies of the inverted diagonal blocks, [nb][1024 rows][4096 bytes]
    const float* dinv_iscale; // [nb] 1 / scale of each
    const char* wplanes;      // split copies of the merged chain operands W_J (IT_UW), [nb][1024 rows][4096 bytes]
    const float* w_iscale;    // [nb] 1 / scale of each
    char* planes_x;           // [nb][m_cap][4096]
    char* planes_d;
    int64_t p_stride;         // bytes between block columns of planes_x / planes_d
    float* rinv_x;            // [nb][m_cap]
    float* rinv_d;
    int64_t r_stride;
    const int4* items;        // the queues' tables one behind the other
    int n_items;
    int nq;                   // queues: 1, or 8 = one per XCD (a workgroup draws from the queue of the XCD it runs on)
    int q_off[kMaxQ + 1];     // queue q = items[q_off[q] .. q_off[q + 1])
    int* sync;
    int mt, nb, ctiles, tail_ct;
    float l_iscale;           // 1 / scale of the factor's split copy
    int backward;
    unsigned long long spin_ticks;  // bound of every wait, in s_memrealtime ticks (100 MHz)
    unsigned spin_max;              // ... and in polls (a second bound that does not depend on a clock)
    int fault_ticket;               // test hook: the item with this ticket is processed but never published (-1: none)
};

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte / 4-byte write-through stores (sc1): the bytes leave the XCD's L2 with the store, so the release fence behind them finds
// nothing to write back ("publish-large" in the guide's price list).  Inline assembly, because HIP has no 128-bit store with a
// cache policy -- and therefore with the wait states hipcc would insert itself: a VALU write to the data registers of a VMEM
// store of more than 64 bits needs them (CDNA3 ISA, manually inserted wait states), and hipcc, which does not know that the
// statement is a store, reuses those registers at once (first runs of this kernel: split rows with values of the NEXT row in them).
__device__ __forceinline__ void st16_wt(void* p, f32x4v v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st16h_wt(void* p, h8 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st4_wt(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

#ifdef NNGP_TIMING_KNOBS
// timing study (knobs build): s_memrealtime ticks (10 ns) summed over all workgroups, per item type t = 0..3:
// [t*4 + 0] wait + acquire, [t*4 + 1] body, [t*4 + 2] publish + next ticket, [t*4 + 3] items (t = 0..5); [24] kernel ticks summed, [25] workgroups
__device__ unsigned long long g_tk_stamps[28];
#define TK_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define TK_NOW() 0ULL
#endif

constexpr int kRing = 4;                 // 128-tile items: operand stages in LDS, three in flight while one is multiplied
constexpr int HT2 = 256;                 // bulk update tile edge
constexpr int HSTAGE2 = 2 * HT2 * TROW;  // its stage: A rows + B rows, 64 KB

// 512 threads = 8 waves, one workgroup per compute unit (all 160 KB of LDS).  Bulk updates are 256 x 256 tiles on all eight waves --
// the main loop of k_gemm_nt_h3v2 (gemm_h3.hip: 2 x 4 waves, wave sub-tile 128 x 64, two 64 KB stages, two staggered wave groups):
// with 128 x 128 tiles everywhere the solve moved 33 GB of operands per solve through the fabric and ran at ITS rate (5.6 ms at
// N = 32768, M = 1024, the matrix pipe 40 % busy; DESIGN_NOTES R5-T) -- a 256 x 256 tile needs half the bytes per flop.  The items on
// the dependency chain (diagonal products, the last update of a block column before it is solved) stay 128 x 128 tiles on waves 0-3
// with a four-stage ring: they are latency, not bandwidth.
__global__ __launch_bounds__(512, 1) void k_trsm_tickets(TkParams P) {
    __shared__ __attribute__((aligned(1024))) char smem[128 * 1024];  // ONE object (gemm_h3.hip: a second one drains the LDS-DMA queue)
    int& s_word = *reinterpret_cast<int*>(smem);                      // ticket / wait status: only touched between items
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = __builtin_amdgcn_readfirstlane((wave >> 1) & 1);  // 128-tile items (waves 0-3 as 2 x 2)
    const int wn = __builtin_amdgcn_readfirstlane(wave & 1);
    const int group = __builtin_amdgcn_readfirstlane(wave >> 2);     // 256-tile items (2 x 4)
    const int w4 = __builtin_amdgcn_readfirstlane(wave & 3);
    const int l3 = lane >> 3;
    const int r16 = lane & 15, q4 = lane >> 4;
    // an 8-row piece of split rows lands as 1 KB of LDS; the 16-byte chunk index is XOR-swizzled by (row >> 1) & 7, a lane constant
    // because every wave only moves pieces of one parity (128-tile items: piece = 4 e + wave; 256-tile items: 16 group + 4 e + w4)
    const unsigned lane_off = (unsigned)l3 * (unsigned)kLdp + (((lane & 7) ^ ((4 * (w4 & 1) + (l3 >> 1)) & 7)) << 4);
    const unsigned frag_hi = (unsigned)r16 * TROW + (((unsigned)q4 ^ ((unsigned)r16 >> 1)) << 4);
    const unsigned frag_lo = frag_hi ^ 64u;
    int* const sync = P.sync;
    const int mt = P.mt, nb = P.nb;
    int* const cnt_xd = sync + SY_COUNTERS;           // [mt][nb] diagonal tiles of X_J[r] finished
    int* const cnt_xs = cnt_xd + mt * nb;             // [mt][nb] split items of X_J[r] finished
    int* const cnt_bs = cnt_xs + mt * nb;             // [mt][nb] split items of B_J[r] finished
    int* const cnt_up = cnt_bs + mt * nb;             // [mt][ctiles] block columns applied to tile (r, c)

    // The loop holds exactly ONE block that only thread 0 executes (publish + next ticket, at its bottom, between two workgroup
    // barriers).  With the ticket fetch in a block of its own at the top, hipcc threaded thread 0's path from the publish block
    // across the back edge into the fetch block, which made the rest of the loop an INNER loop that the other threads never leave:
    // lane 0 of wave 0 waited for its wave's reconvergence, nothing was ever published (first GPU run of this kernel: every workgroup
    // holding its first ticket, no counter moving).
    // Queues (round 5): with nq = 8 the table is eight tables, one per XCD, and a workgroup draws from the table of the XCD it runs on
    // (HW_REG_XCC_ID): the items of one pair of row tiles meet in one L2.  Every queue is consumed in its own order whoever draws
    // from it -- a workgroup whose queue has run out goes on with the others' (the end of a solve, or an XCD without workgroups).
    int my_q = 0;
    if (P.nq > 1) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        my_q = (int)(xcc & 15u) % P.nq;
    }
    auto next_ticket = [&]() -> int {  // thread 0 only: index of the next item in P.items, or n_items when every queue is through
        if (ld_agent(sync + SY_ERROR) != 0) return 0x7fffffff;
#pragma unroll
        for (int k = 0; k < kMaxQ; ++k) {  // (unrolled: no loop of thread 0's own inside the item loop, see above)
            if (k >= P.nq) break;
            int q = my_q + k;
            if (q >= P.nq) q -= P.nq;
            const int len = P.q_off[q + 1] - P.q_off[q];
            if (ld_agent(sync + SY_TICKETQ + q) >= len) continue;  // (an exhausted queue's counter is left alone)
            const int t = atomicAdd(sync + SY_TICKETQ + q, 1);
            if (t < len) return P.q_off[q] + t;
        }
        return 0x7fffffff;
    };
    if (tid == 0) s_word = next_ticket();
    __syncthreads();
    int ticket = __builtin_amdgcn_readfirstlane(s_word);
    __syncthreads();
#ifdef NNGP_TIMING_KNOBS
    unsigned long long st[24] = {};
    const unsigned long long st_k0 = TK_NOW();
#endif
    while (ticket < P.n_items) {
        const unsigned long long st_a = TK_NOW();
        const int4 it = P.items[ticket];
        const int type = __builtin_amdgcn_readfirstlane(it.x & 15), npan = __builtin_amdgcn_readfirstlane((it.x >> 4) & 15);
        const int r = __builtin_amdgcn_readfirstlane(it.y), cq = __builtin_amdgcn_readfirstlane(it.z),
                  J = __builtin_amdgcn_readfirstlane(it.w);
        // r, cq: tile of 128 rows / columns (IT_UB: tile of 256 = tiles 2 r, 2 r + 1 / 2 cq, 2 cq + 1; split items: cq = row group)
        const int Jc = type == IT_UB ? cq / 4 : (type == IT_U || type == IT_D || type == IT_UW) ? cq / 8 : J;  // block column of the target
        const int ct_j = (Jc == nb - 1) ? P.tail_ct : 8;

        // ---- dependencies: wave 0 polls, one counter per lane ----
        if (wv == 0) {
            const int* addr = sync + SY_ERROR;
            int need = 0;
            bool active = false;
            // an update: the latest block column of the item is split (for its rows); its tiles have received everything before the
            // item's earliest block column
            const int first_in_time = P.backward ? J + npan - 1 : J - npan + 1;
            const int need_up = P.backward ? nb - 1 - first_in_time : first_in_time;
            if (type == IT_U) {
                if (lane == 0) { addr = cnt_xs + r * nb + J; need = kQ; active = true; }
                if (lane == 1) { addr = cnt_up + r * P.ctiles + cq; need = need_up; active = true; }
            } else if (type == IT_UW) {  // the split rows of B_J instead of those of X_J
                if (lane == 0) { addr = cnt_bs + r * nb + J; need = kQ; active = true; }
                if (lane == 1) { addr = cnt_up + r * P.ctiles + cq; need = need_up; active = true; }
            } else if (type == IT_UB) {
                if (lane < 2) { addr = cnt_xs + (2 * r + lane) * nb + J; need = kQ; active = true; }
                else if (lane < 6) { addr = cnt_up + (2 * r + ((lane - 2) >> 1)) * P.ctiles + 2 * cq + ((lane - 2) & 1); need = need_up; active = true; }
            } else if (type == IT_D) {
                if (lane == 0) { addr = cnt_bs + r * nb + Jc; need = kQ; active = true; }
            } else if (type == IT_SX) {
                if (lane == 0) { addr = cnt_xd + r * nb + J; need = ct_j; active = true; }
            } else {  // IT_SB: every tile of the block has received all its updates
                if (lane < ct_j) { addr = cnt_up + r * P.ctiles + J * 8 + lane; need = P.backward ? nb - 1 - J : J; active = true; }
            }
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int status = 0;
            for (unsigned spins = 0;; ++spins) {
                const int v = active ? ld_agent(addr) : need;
                const int e = lane == 63 ? ld_agent(sync + SY_ERROR) : 0;
                if (__any(e != 0)) { status = 1; break; }
                if (__all(v >= need)) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > P.spin_ticks || spins > P.spin_max) {
                    if (lane == 0) atomicOr(sync + SY_ERROR, 0x100 | type);
                    status = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) s_word = status;
        }
        __syncthreads();
        const int status = __builtin_amdgcn_readfirstlane(s_word);
        __syncthreads();
        if (status != 0) break;  // (uniform)
        const unsigned long long st_b = TK_NOW();

        int* done_counter;
        int done_add = 1, done_n = 1, done_stride2 = 0;  // counters done_counter[0 .. done_n), UB: also the next row tile's (stride2)
        if (type == IT_SB || type == IT_SX) {
            // ---- split item: rows r*128 + cq*32 + wave*4 + i, columns of block J ----
            const bool is_x = type == IT_SX;
            char* planes = (is_x ? P.planes_x : P.planes_d) + (int64_t)J * P.p_stride;
            float* rinv = (is_x ? P.rinv_x : P.rinv_d) + (int64_t)J * P.r_stride;
            const int width = ct_j * TT;
            const int row0 = r * TT + cq * 32 + wv * 4;
            f32x4v v[4][2][2];
            float mx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* src = P.b + (int64_t)(row0 + i) * P.ldb + (int64_t)J * kBs;
                mx[i] = 0.0f;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k0 = h * 512 + lane * 8;
                    if (k0 < width) {
                        v[i][h][0] = *reinterpret_cast<const f32x4v*>(src + k0);
                        v[i][h][1] = *reinterpret_cast<const f32x4v*>(src + k0 + 4);
                    } else {
                        v[i][h][0] = v[i][h][1] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) mx[i] = fmaxf(mx[i], fabsf(v[i][h][g][e]));
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], off));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int e2 = 0;
                (void)frexpf(mx[i], &e2);  // mx = f * 2^e2, f in [0.5, 1): largest entry -> [2^13, 2^14)
                const float sc = (mx[i] > 0.0f && mx[i] < 3.0e38f) ? ldexpf(1.0f, 14 - e2) : 1.0f;
                if (lane == 0) st4_wt(rinv + row0 + i, 1.0f / sc);
                char* drow = planes + (int64_t)(row0 + i) * kLdp;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k0 = h * 512 + lane * 8;
                    if (k0 >= width) continue;
                    h8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a = v[i][h][e >> 2][e & 3] * sc;
                        const _Float16 hh = (_Float16)a;
                        hi[e] = hh;
                        lo[e] = (_Float16)(a - (float)hh);
                    }
                    char* dst = drow + (k0 >> 5) * 128 + (k0 & 31) * 2;
                    st16h_wt(dst, hi);
                    st16h_wt(dst + 64, lo);
                }
            }
            done_counter = (is_x ? cnt_xs : cnt_bs) + r * nb + J;
        } else if (type == IT_UB) {
            // ---- bulk update: B[256 r .., 256 cq ..] -= sum over npan block columns X_J[rows] L[cols, J]^T ----
            const int ct_p = (J == nb - 1) ? P.tail_ct : 8;  // (a tail-width block column is never grouped with others)
            const int kb_hi = ct_p * 4 - 1;
            const char* pa = P.planes_x + (int64_t)J * P.p_stride + (int64_t)(r * HT2) * kLdp;
            const char* pb = P.lplanes + (int64_t)J * P.l_stride + (int64_t)(cq * HT2) * kLdp;
            const int64_t a_pst = P.backward ? P.p_stride : -P.p_stride;
            const int64_t b_pst = P.backward ? P.l_stride : -P.l_stride;
            const int nkp = kb_hi + 1;
            const int nstage = nkp * npan;
            int cur_p = 0, cur_kb = kb_hi;
            int64_t offa = 0, offb = 0;
            auto cursor_take = [&]() {  // offsets of the cursor's k-block; advances the cursor
                offa = (int64_t)cur_kb * 128 + (int64_t)cur_p * a_pst;
                offb = (int64_t)cur_kb * 128 + (int64_t)cur_p * b_pst;
                if (--cur_kb < 0) { cur_kb = kb_hi; ++cur_p; }
            };
            auto glds_b = [&](int t) {  // the wave's 4 B pieces of stage t
                char* dst = smem + (t & 1) * HSTAGE2 + HT2 * TROW;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int piece = group * 16 + e * 4 + w4;
                    __builtin_amdgcn_global_load_lds((glb_void*)(pb + offb + (int64_t)piece * 8 * kLdp + lane_off), (lds_void*)(dst + piece * 1024), 16, 0, 0);
                }
            };
            auto glds_a = [&](int t) {  // the wave's 4 A pieces of stage t: rows 0-63 of its group first (e = 0, 1), then rows 64-127
                char* dst = smem + (t & 1) * HSTAGE2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int piece = group * 16 + e * 4 + w4;
                    __builtin_amdgcn_global_load_lds((glb_void*)(pa + offa + (int64_t)piece * 8 * kLdp + lane_off), (lds_void*)(dst + piece * 1024), 16, 0, 0);
                }
            };
            cursor_take();
            glds_b(0);
            glds_a(0);
            // acc[i][j][e] = tile(group*128 + 16 i + r16, w4*64 + 16 j + 4 q4 + e): the lane's rows 16 i + r16 of its group's 128
            const int row_base = r * HT2 + group * 128;
            // (the rows' scales are re-read where they are needed -- at block-column boundaries and in the epilogue -- rather than held:
            // eight registers that decide whether a light kernel of another stream fits beside this one on the compute unit)
            f32x4v acc[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (group == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier behind group 0
            int next_bound = nkp, pcur = 0;
            for (int t = 0; t < nstage; ++t) {
                const char* sa_ = smem + (t & 1) * HSTAGE2;
                const char* sb_ = sa_ + HT2 * TROW;
                const bool more = t + 1 < nstage;
                if (t == next_bound) {
                    // block-column boundary: the sums so far carry the row scales of block column pcur, the terms to come those of the
                    // next one (ratios of powers of two: exact).  The scale loads wait for everything in flight: once per 32 stages.
                    next_bound += nkp;
                    const float* rc = P.rinv_x + (int64_t)(P.backward ? J + pcur : J - pcur) * P.r_stride + row_base + r16;
                    ++pcur;
                    const float* rv = P.rinv_x + (int64_t)(P.backward ? J + pcur : J - pcur) * P.r_stride + row_base + r16;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float f = rc[16 * i] / rv[16 * i];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][e] *= f;
                    }
                }
                if (more) cursor_take();
                h8 bh[4], bl[4], ah[4], al[4];
                // ---- phase A: wave rows 0-63 ----
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rb = (w4 * 64 + j * 16) * TROW;
                    bh[j] = *reinterpret_cast<const h8*>(sb_ + rb + frag_hi);
                    bl[j] = *reinterpret_cast<const h8*>(sb_ + rb + frag_lo);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rb = (group * 128 + i * 16) * TROW;
                    ah[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_hi);
                    al[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_lo);
                }
                if (more) {
                    glds_b(t + 1);
                    if (t > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // retires the wave's last two A pieces of stage t (read in phase B)
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- phase B: wave rows 64-127 ----
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rb = (group * 128 + 64 + i * 16) * TROW;
                    ah[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_hi);
                    al[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_lo);
                }
                if (more) {
                    glds_a(t + 1);
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // retires the B pieces issued in phase A
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[4 + i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[4 + i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[4 + i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // retires the A pieces of rows 0-63 of stage t+1
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (group == 0) __builtin_amdgcn_s_barrier();  // every wave executes the same number of barriers
            // ---- epilogue: two 16-row bands at a time ----
            const unsigned voff = ((unsigned)r16 * (unsigned)P.ldb + 4u * (unsigned)q4) * 4u;
            char* cw = reinterpret_cast<char*>(P.b + (int64_t)row_base * P.ldb + (int64_t)cq * HT2 + w4 * 64);
            const int64_t band = (int64_t)16 * P.ldb * 4;
            const float* rs_last = P.rinv_x + (int64_t)(P.backward ? J + pcur : J - pcur) * P.r_stride + row_base + r16;
#pragma unroll
            for (int ib = 0; ib < 8; ib += 2) {
                f32x4v cold[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) cold[i][j] = *reinterpret_cast<const f32x4v*>(cw + (ib + i) * band + voff + 64 * j);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float fs = -rs_last[16 * (ib + i)] * P.l_iscale;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[ib + i][j][e] = fmaf(fs, acc[ib + i][j][e], cold[i][j][e]);
                        st16_wt(cw + (ib + i) * band + voff + 64 * j, acc[ib + i][j]);
                    }
                }
            }
            done_counter = cnt_up + (2 * r) * P.ctiles + 2 * cq;
            done_add = npan;
            done_n = 2;
            done_stride2 = P.ctiles;
        } else {
            // ---- 128 x 128 tile item (diagonal product, or an update on the chain) on waves 0-3; waves 4-7 keep the barriers ----
            const bool is_w = type == IT_UW;
            const bool is_u = type == IT_U || is_w;
            const int cl = cq - Jc * 8;  // tile inside its block column
            // operands: k-blocks kb_hi down to kb_lo of every panel; panel p lies a_pst / b_pst bytes from panel p - 1
            int kb_lo, kb_hi;
            const char* pa;
            const char* pb;
            int64_t a_pst = 0, b_pst = 0;
            if (is_u) {
                const int ct_p = (J == nb - 1) ? P.tail_ct : 8;  // (a tail panel is never grouped with others)
                kb_lo = 0;
                kb_hi = ct_p * 4 - 1;
                pa = (is_w ? P.planes_d : P.planes_x) + (int64_t)J * P.p_stride + (int64_t)(r * TT) * kLdp;
                pb = is_w ? P.wplanes + (int64_t)J * ((int64_t)kBs * kLdp) + (int64_t)(cl * TT) * kLdp
                          : P.lplanes + (int64_t)J * P.l_stride + (int64_t)(cq * TT) * kLdp;
                a_pst = P.backward ? P.p_stride : -P.p_stride;
                b_pst = P.backward ? P.l_stride : -P.l_stride;
            } else {
                // forward: Linv lower (k <= column) -> k tiles 0 .. cl; backward: L^-T upper -> k tiles cl .. ct_j - 1
                kb_lo = P.backward ? cl * 4 : 0;
                kb_hi = P.backward ? ct_j * 4 - 1 : cl * 4 + 3;
                pa = P.planes_d + (int64_t)Jc * P.p_stride + (int64_t)(r * TT) * kLdp;
                pb = P.dinv_s + (int64_t)Jc * ((int64_t)kBs * kLdp) + (int64_t)(cl * TT) * kLdp;
            }
            const int nkp = kb_hi - kb_lo + 1;
            const int nstage = nkp * (is_u ? npan : 1);
            if (wv >= 4) {
                for (int t = 0; t < nstage; ++t) __builtin_amdgcn_s_barrier();
                done_counter = is_u ? cnt_up + r * P.ctiles + cq : cnt_xd + r * nb + Jc;
                done_add = is_u ? npan : 1;
            } else {
                int cur_p = 0, cur_kb = kb_hi;
                auto glds_stage = [&](int t) {  // the wave's 4 B pieces and 4 A pieces of stage t at the cursor; advances the cursor
                    const int64_t offa = (int64_t)cur_kb * 128 + (int64_t)cur_p * a_pst;
                    const int64_t offb = (int64_t)cur_kb * 128 + (int64_t)cur_p * b_pst;
                    if (--cur_kb < kb_lo) { cur_kb = kb_hi; ++cur_p; }
                    char* dst = smem + (t & (kRing - 1)) * TSTAGE;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int piece = e * 4 + wv;
                        __builtin_amdgcn_global_load_lds((glb_void*)(pb + offb + (int64_t)piece * 8 * kLdp + lane_off),
                                                         (lds_void*)(dst + TT * TROW + piece * 1024), 16, 0, 0);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int piece = e * 4 + wv;
                        __builtin_amdgcn_global_load_lds((glb_void*)(pa + offa + (int64_t)piece * 8 * kLdp + lane_off),
                                                         (lds_void*)(dst + piece * 1024), 16, 0, 0);
                    }
                };
                // The operands arrive latency bound (~1.7 us a stage under load, against 0.4 us of MFMAs): stages t + 1 .. t + 3 are in
                // flight while stage t is multiplied
                glds_stage(0);
                if (nstage > 1) glds_stage(1);
                if (nstage > 2) glds_stage(2);
                // per-row scales of the A operand's panels (powers of two): rs[p][i] for the lane's rows 16 i + r16 of the wave's 64
                const int row_base = r * TT + wm * 64;
                float rs[4][4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) rs[p][i] = 1.0f;
                if (is_u) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (p < npan) {
                            const float* rv = (is_w ? P.rinv_d : P.rinv_x) + (int64_t)(P.backward ? J + p : J - p) * P.r_stride + row_base + r16;
#pragma unroll
                            for (int i = 0; i < 4; ++i) rs[p][i] = rv[16 * i];
                        }
                } else {
                    const float* rv = P.rinv_d + (int64_t)Jc * P.r_stride + row_base + r16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) rs[0][i] = rv[16 * i];
                }
                f32x4v acc[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
                int next_bound = nkp, pcur = 0;
                for (int t = 0; t < nstage; ++t) {
                    // the wave's own 8 pieces of stage t have landed (in-order return): the 16 of stages t + 1, t + 2 may still be on their way
                    if (t + 2 < nstage)
                        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else if (t + 1 < nstage)
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // buffer (t + 3) % 4 held stage t - 1: every wave retired its reads of it before it reached this barrier
                    if (t + 3 < nstage) glds_stage(t + 3);
                    const char* sa_ = smem + (t & (kRing - 1)) * TSTAGE;
                    const char* sb_ = sa_ + TT * TROW;
                    h8 bh[4], bl[4], ah[4], al[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rb = (wn * 64 + j * 16) * TROW;
                        bh[j] = *reinterpret_cast<const h8*>(sb_ + rb + frag_hi);
                        bl[j] = *reinterpret_cast<const h8*>(sb_ + rb + frag_lo);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rb = (wm * 64 + i * 16) * TROW;
                        ah[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_hi);
                        al[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_lo);
                    }
                    if (t == next_bound) {
                        // panel boundary: the sums so far carry panel pcur's row scales, the terms to come panel pcur + 1's
                        // (ratios of powers of two: exact)
                        next_bound += nkp;
                        float f[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float a = pcur == 0 ? rs[0][i] : pcur == 1 ? rs[1][i] : rs[2][i];
                            const float b = pcur == 0 ? rs[1][i] : pcur == 1 ? rs[2][i] : rs[3][i];
                            f[i] = a / b;
                        }
                        ++pcur;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i][j][e] *= f[i];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- epilogue: acc[i][j][e] = tile(wm*64 + 16 i + r16, wn*64 + 16 j + 4 q4 + e) ----
                float fs[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float last = pcur == 0 ? rs[0][i] : pcur == 1 ? rs[1][i] : pcur == 2 ? rs[2][i] : rs[3][i];
                    fs[i] = last * (is_w ? -P.w_iscale[J] : is_u ? -P.l_iscale : P.dinv_iscale[Jc]);
                }
                const unsigned voff = ((unsigned)r16 * (unsigned)P.ldb + 4u * (unsigned)q4) * 4u;
                char* cw = reinterpret_cast<char*>(P.b + (int64_t)(row_base)*P.ldb + (int64_t)cq * TT + wn * 64);
                const int64_t band = (int64_t)16 * P.ldb * 4;
                if (is_u) {
                    f32x4v cold[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) cold[i][j] = *reinterpret_cast<const f32x4v*>(cw + i * band + voff + 64 * j);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][e] = fmaf(fs[i], acc[i][j][e], cold[i][j][e]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][e] *= fs[i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) st16_wt(cw + i * band + voff + 64 * j, acc[i][j]);
                if (is_u) {
                    done_counter = cnt_up + r * P.ctiles + cq;
                    done_add = npan;
                } else {
                    done_counter = cnt_xd + r * nb + Jc;
                }
            }
        }
        // ---- publish: every wave's stores have left, then one lane releases and counts -- and takes the next ticket ----
        const unsigned long long st_c = TK_NOW();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ticket != P.fault_ticket)
                for (int a = 0; a < done_n; ++a) {
                    atomicAdd(done_counter + a, done_add);
                    if (done_stride2 != 0) atomicAdd(done_counter + done_stride2 + a, done_add);
                }
            s_word = next_ticket();
        }
        __syncthreads();
        ticket = __builtin_amdgcn_readfirstlane(s_word);
        __syncthreads();
#ifdef NNGP_TIMING_KNOBS
        {
            const unsigned long long st_d = TK_NOW();
            st[type * 4 + 0] += st_b - st_a; st[type * 4 + 1] += st_c - st_b; st[type * 4 + 2] += st_d - st_c; st[type * 4 + 3] += 1;
        }
#else
        (void)st_a; (void)st_b; (void)st_c;
#endif
    }
#ifdef NNGP_TIMING_KNOBS
    if (tid == 0) {
        for (int i = 0; i < 24; ++i) atomicAdd(&g_tk_stamps[i], st[i]);
        atomicAdd(&g_tk_stamps[24], TK_NOW() - st_k0);
        atomicAdd(&g_tk_stamps[25], 1ULL);
    }
#endif
}

// absolute maximum of each bs x bs block (bit pattern of a non-negative float orders like an unsigned)
__global__ __launch_bounds__(256) void k_block_absmax(const float* __restrict__ a, int64_t per_block, unsigned* __restrict__ out) {
    const float* p = a + (int64_t)blockIdx.y * per_block;
    float mx = 0.0f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < per_block; i += (int64_t)gridDim.x * 1024) {
        const f32x4v v = *reinterpret_cast<const f32x4v*>(p + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[e]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0 && mx > 0.0f && mx < 3.0e38f) atomicMax(out + blockIdx.y, __float_as_uint(mx));
}

// split rows of every block with the block's own power-of-two scale (largest entry -> [2^13, 2^14)); iscale[b] = 1 / scale
__global__ __launch_bounds__(256) void k_split_blocks(const float* __restrict__ a, int bs, const unsigned* __restrict__ amax,
                                                      char* __restrict__ out, float* __restrict__ iscale) {
    const int blk = blockIdx.y;
    const float mx = __uint_as_float(amax[blk]);
    int e2 = 0;
    (void)frexpf(mx, &e2);
    const float sc = mx > 0.0f ? ldexpf(1.0f, 14 - e2) : 1.0f;
    const int k8 = bs / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx == 0) iscale[blk] = 1.0f / sc;
    const int64_t rr = idx / k8;
    const int c8 = (int)(idx % k8);
    if (rr >= bs) return;
    const f32x4v* src = reinterpret_cast<const f32x4v*>(a + (int64_t)blk * bs * bs + rr * bs + (int64_t)c8 * 8);
    const f32x4v v0 = src[0], v1 = src[1];
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (e < 4 ? v0[e] : v1[e - 4]) * sc;
        const _Float16 h = (_Float16)x;
        hi[e] = h;
        lo[e] = (_Float16)(x - (float)h);
    }
    char* dst = out + (int64_t)blk * ((int64_t)kBs * kLdp) + rr * kLdp + (int64_t)(c8 >> 2) * 128 + (c8 & 3) * 16;
    *reinterpret_cast<h8*>(dst) = hi;
    *reinterpret_cast<h8*>(dst + 64) = lo;
}

// dst_b [c][n] = src_b [n][c] for 1024 x 1024 blocks b: src_b = src + b * sstride (row stride lds), dst_b = dst + b * 1024 * 1024
__global__ __launch_bounds__(256) void k_transpose_1024(const float* __restrict__ src, int64_t lds, int64_t sstride, float* __restrict__ dst, int rows_valid_last, int nblk) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float* sp = src + (int64_t)b * sstride;
    float* dp = dst + (int64_t)b * kBs * kBs;
    const int n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;   // source rows n, source columns c
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int nvalid = (b == nblk - 1) ? rows_valid_last : kBs;  // the last block may have fewer source rows (tail of the factor)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i;
        tile[ty + 8 * i][tx] = n < nvalid ? sp[(int64_t)n * lds + c0 + tx] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) dp[(int64_t)(c0 + ty + 8 * i) * kBs + n0 + tx] = tile[tx][ty + 8 * i];
}

// ---- host: the ticket order ----
// List scheduling on `workers` simulated workgroups with rough item durations: an item enters the table when the simulation starts
// it, i.e. after everything it depends on has FINISHED there -- so every dependency has a lower ticket whatever the real timing is.
// Priority: earliest deadline first, the deadline of an item being the block column it feeds (diagonal chain items before the
// updates of the next block column).  Updates come in two shapes: a 2 x 2 group of tiles takes every finished block column that is
// waiting for it EXCEPT the two right before its own block, up to four at a time, as one 256 x 256 item (deep K where the chip is
// throughput-bound); the last two block columns before a tile's own -- the updates on or next to the dependency chain -- are applied
// tile by tile (128 x 128 items: four times the parallelism where only latency counts).  Tiles without a partner (odd row-tile count, odd tail)
// take all their updates as 128 x 128 items.
// merged: the chain update of a paired tile is a UW item (it waits for the split rows of the previous block B_J, not for X_J)
// nq > 1: the workers are nq pools (the XCDs) with a table each.  A pair of row tiles belongs to 8 / (pairs) pools -- its bulk items are
// dealt to them by column pair, its two chains to the first two -- so that everything that reads the split rows of X of one row pair
// (1 MB per block column, read by every column pair's item) runs behind ONE L2, in the order in which it becomes ready: the items of one
// block-column range over all column pairs are queued together and start together.  `queue_of` receives each item's pool; `out` stays
// the global start order (an item's dependencies all precede it there).
void tk_build_order(int mt, int nb, int tail_ct, bool backward, int workers, bool merged, std::vector<int4>& out, int nq = 1,
                    std::vector<int>* queue_of = nullptr) {
    const int ctiles = (nb - 1) * 8 + tail_ct;
    auto ct_of = [&](int J) { return J == nb - 1 ? tail_ct : 8; };
    auto pos_of = [&](int J) { return backward ? nb - 1 - J : J; };  // position of block column J in solve order (0 = solved first)
    auto blk_at = [&](int pos) { return backward ? nb - 1 - pos : pos; };
    const double t_split = 10.0, t_tile0 = 14.0, t_stage = 0.6, t_big0 = 24.0, t_bigstage = 1.6;  // us, from the stamps of GPU runs
    int max_pan = 4;
    if (const char* e = getenv("NNGP_TK_MAXPAN")) max_pan = std::max(1, std::min(4, atoi(e)));  // development aid
    const bool no_big = getenv("NNGP_TK_NOBIG") != nullptr;                                      // development aid: 128 x 128 items only
    enum { K_TILE = 0, K_PAIR = 1 };
    struct Ready {
        int key, c, r, type, J, kind;  // type IT_*; kind: what an update entry stands for
        bool operator<(const Ready& o) const {  // priority_queue: largest first -> invert
            if (key != o.key) return key > o.key;
            if (c != o.c) return c > o.c;
            if (r != o.r) return r > o.r;
            return type > o.type;
        }
    };
    struct Done {
        double t;
        int type, r, cq, J, npan;
        int q = 0;
        bool operator<(const Done& o) const { return t > o.t; }
    };
    if (nq < 1 || nq > kMaxQ) nq = 1;
    std::vector<std::priority_queue<Ready>> ready_q((size_t)nq);
    std::priority_queue<Done> running;
    const int mt2p = mt / 2;
    const int per_pair = (mt2p >= 1 && mt2p <= nq && nq % mt2p == 0) ? nq / mt2p : 1;  // pools per pair of row tiles
    auto pool_of = [&](int type, int r, int c) -> int {  // r: row tile (IT_UB: first of the pair), c: column tile (split items: anything)
        if (nq == 1) return 0;
        const int rp = r / 2;
        if (rp >= mt2p) return r % nq;  // the unpaired last row tile
        const int base = (rp * per_pair) % nq;
        if (type == IT_UB) return base + (c / 2) % per_pair;
        return base + (r % 2) % per_pair;  // a row tile's chain (splits, diagonal products, 128 x 128 updates)
    };
    struct ReadyProxy {  // the single-queue code below pushes here
        std::vector<std::priority_queue<Ready>>* q;
        std::function<int(int, int, int)> pool;
        void push(const Ready& it) { (*q)[(size_t)pool(it.type, it.r, it.c)].push(it); }
    } ready{&ready_q, pool_of};
    std::vector<int> up((size_t)mt * ctiles, 0);      // block columns applied to a tile (positions in solve order)
    std::vector<char> busy((size_t)mt * ctiles, 0);   // an update of the tile is queued or running
    std::vector<int> xs((size_t)mt * nb, 0), xd((size_t)mt * nb, 0), bs((size_t)mt * nb, 0), tiles_final((size_t)mt * nb, 0);
    std::vector<int> xready((size_t)mt, 0);           // positions [0, xready[r]) have their split X
    std::vector<int> bready((size_t)mt, 0);           // positions [0, bready[r]) have their split B (the merged chain update's operand)
    const int mt2 = mt / 2, ct2 = ctiles / 2;         // 2 x 2 groups: row tiles (2 r2, 2 r2 + 1), column tiles (2 c2, 2 c2 + 1) -- same block (8 | 2 c2)
    auto paired = [&](int r, int c) { return !no_big && r < 2 * mt2 && c < 2 * ct2 && (c / 8 != nb - 1 || (c % 8) + (c % 2 == 0 ? 1 : 0) < ct_of(nb - 1)); };
    auto tile_id = [&](int r, int c) { return (size_t)r * ctiles + c; };
    // the chain update of a tile: only its last block column is missing and that one is split
    auto queue_final = [&](int r, int c) {
        const size_t id = tile_id(r, c);
        const int pc = pos_of(c / 8);
        if (busy[id] || pc == 0 || up[id] != pc - 1 || (merged ? bready[r] : xready[r]) < pc) return;
        busy[id] = 1;
        ready.push(Ready{2 * pc - 1, c, r, merged ? IT_UW : IT_U, 0, K_TILE});
    };
    // the update before that (two block columns back): a 128 x 128 item as well -- as part of a 256 x 256 item (~100 us) it would sit
    // between X_J and the chain update of the block after next, i.e. on the chain
    auto queue_near = [&](int r, int c) {
        const size_t id = tile_id(r, c);
        const int pc = pos_of(c / 8);
        if (busy[id] || pc < 2 || up[id] != pc - 2 || xready[r] < pc - 1) return;
        busy[id] = 1;
        ready.push(Ready{2 * pc - 1, c, r, IT_U, 0, K_TILE});
    };
    // an unpaired tile takes everything as 128 x 128 items
    auto queue_single = [&](int r, int c) {
        const size_t id = tile_id(r, c);
        const int pc = pos_of(c / 8);
        if (busy[id] || up[id] >= pc || up[id] >= xready[r]) return;
        busy[id] = 1;
        ready.push(Ready{2 * pc - 1, c, r, IT_U, 0, K_TILE});
    };
    auto queue_pair = [&](int r2, int c2) {
        const int r = 2 * r2, c = 2 * c2;
        const int pc = pos_of(c / 8);
        const size_t id = tile_id(r, c);
        const int p0 = up[id];
        for (int dr = 0; dr < 2; ++dr)
            for (int dc = 0; dc < 2; ++dc)
                if (busy[tile_id(r + dr, c + dc)] || up[tile_id(r + dr, c + dc)] != p0) return;
        const int xr = std::min(xready[r], xready[r + 1]);
        if (std::min(pc - 2, xr) - p0 < 1) return;
        for (int dr = 0; dr < 2; ++dr)
            for (int dc = 0; dc < 2; ++dc) busy[tile_id(r + dr, c + dc)] = 1;
        ready.push(Ready{2 * pc - 1, c, r, IT_UB, 0, K_PAIR});
    };
    auto requeue_tile = [&](int r, int c) {
        if (paired(r, c)) {
            queue_pair(r / 2, c / 2);
            queue_near(r, c);
            queue_final(r, c);
        } else {
            queue_single(r, c);
        }
    };
    auto push_block_items = [&](int type, int r, int J) {
        const int n = type == IT_D ? ct_of(J) : kQ;
        for (int i = 0; i < n; ++i) ready.push(Ready{2 * pos_of(J), type == IT_D ? J * 8 + i : i, r, type, J, K_TILE});
    };
    for (int r = 0; r < mt; ++r) push_block_items(IT_SB, r, blk_at(0));
    std::vector<int> free_q((size_t)nq, workers / nq > 0 ? workers / nq : 1);
    double now = 0.0;
    out.clear();
    if (queue_of != nullptr) queue_of->clear();
    for (;;) {
        for (int q = 0; q < nq; ++q)
        while (free_q[(size_t)q] > 0 && !ready_q[(size_t)q].empty()) {
            int& free_workers = free_q[(size_t)q];
            const Ready it = ready_q[(size_t)q].top();
            ready_q[(size_t)q].pop();
            int4 rec;
            Done d{};
            if (it.type == IT_UB) {
                const int r = it.r, c = it.c, pc = pos_of(c / 8);
                const int p0 = up[tile_id(r, c)];
                int avail = std::min(pc - 2, std::min(xready[r], xready[r + 1])) - p0;
                if (avail > max_pan) avail = max_pan;
                if (backward && tail_ct != 8 && p0 == 0) avail = 1;  // a tail-width block column goes alone
                const int plast = p0 + avail - 1;                    // latest block column of the item in solve order = first processed
                rec = int4{IT_UB | (avail << 4), r / 2, c / 2, blk_at(plast)};
                const int nk = (backward && tail_ct != 8 && p0 == 0) ? tail_ct * 4 : 32;
                d = Done{now + t_big0 + t_bigstage * nk * avail, IT_UB, r, c, 0, avail};
            } else if (it.type == IT_UW) {
                const int pc = pos_of(it.c / 8);
                const int Jsrc = blk_at(pc - 1);
                rec = int4{IT_UW | (1 << 4), it.r, it.c, Jsrc};
                d = Done{now + t_tile0 + t_stage * 4 * ct_of(Jsrc), IT_UW, it.r, it.c, 0, 1};
            } else if (it.type == IT_U) {
                const size_t id = tile_id(it.r, it.c);
                const int pc = pos_of(it.c / 8);
                const int p0 = up[id];
                int avail = std::min(pc, xready[it.r]) - p0;
                if (avail > max_pan) avail = max_pan;
                if (backward && tail_ct != 8 && p0 == 0) avail = 1;
                if (paired(it.r, it.c)) avail = 1;  // the near or the chain update of a paired tile: one block column
                const int plast = p0 + avail - 1;
                rec = int4{IT_U | (avail << 4), it.r, it.c, blk_at(plast)};
                const int nk = (backward && tail_ct != 8 && p0 == 0) ? tail_ct * 4 : 32;
                d = Done{now + t_tile0 + t_stage * nk * avail, IT_U, it.r, it.c, 0, avail};
            } else {
                rec = int4{it.type | (1 << 4), it.r, it.c, it.J};
                double dur = t_split;
                if (it.type == IT_D) {
                    const int cl = it.c - it.J * 8;
                    const int nk = backward ? (ct_of(it.J) - cl) * 4 : (cl + 1) * 4;
                    dur = t_tile0 + t_stage * nk;
                }
                d = Done{now + dur, it.type, it.r, it.c, it.J, 1};
            }
            out.push_back(rec);
            if (queue_of != nullptr) queue_of->push_back(q);
            d.q = q;
            running.push(d);
            --free_workers;
        }
        if (running.empty()) break;
        const Done d = running.top();
        running.pop();
        now = d.t;
        ++free_q[(size_t)d.q];
        const int r = d.r;
        if (d.type == IT_SB) {
            if (++bs[(size_t)r * nb + d.J] == kQ) {
                push_block_items(IT_D, r, d.J);
                bready[r] = pos_of(d.J) + 1;
                if (merged && pos_of(d.J) + 1 < nb) {
                    const int Jn = blk_at(pos_of(d.J) + 1);
                    for (int c = Jn * 8; c < Jn * 8 + ct_of(Jn); ++c)
                        if (paired(r, c)) queue_final(r, c);
                }
            }
        } else if (d.type == IT_D) {
            if (++xd[(size_t)r * nb + d.J] == ct_of(d.J)) push_block_items(IT_SX, r, d.J);
        } else if (d.type == IT_SX) {
            if (++xs[(size_t)r * nb + d.J] == kQ) {
                xready[r] = pos_of(d.J) + 1;
                for (int c = 0; c < ctiles; ++c)
                    if (pos_of(c / 8) > pos_of(d.J)) requeue_tile(r, c);
            }
        } else {
            const int nr = d.type == IT_UB ? 2 : 1;
            for (int dr = 0; dr < nr; ++dr)
                for (int dc = 0; dc < nr; ++dc) {
                    const size_t id = tile_id(r + dr, d.cq + dc);
                    up[id] += d.npan;
                    busy[id] = 0;
                }
            for (int dr = 0; dr < nr; ++dr)
                for (int dc = 0; dc < nr; ++dc) {
                    const int rr = r + dr, cc = d.cq + dc, Jt = cc / 8;
                    if (up[tile_id(rr, cc)] == pos_of(Jt)) {
                        if (++tiles_final[(size_t)rr * nb + Jt] == ct_of(Jt)) push_block_items(IT_SB, rr, Jt);
                    } else {
                        requeue_tile(rr, cc);
                    }
                }
        }
    }
}

}  // namespace

struct TrsmTickets {
    int64_t m_cap = 0, np_cap = 0, nb_cap = 0;
    char* planes_x = nullptr;
    char* planes_d = nullptr;
    float* rinv_x = nullptr;
    float* rinv_d = nullptr;
    char* xinv_s = nullptr;
    char* tinv_s = nullptr;
    float* inv_iscale = nullptr;  // [2][nb_cap]: forward blocks, backward blocks
    // merged chain operands (IT_UW): W_J = L[next block, J] Linv_J (forward), L[J, previous block]^T L_JJ^-T (backward), float32 scratch + split copies
    float* wtmp = nullptr;        // [nb_cap][1024 x 1024]
    float* ltb = nullptr;         // [nb_cap][1024 x 1024] transposed sub-diagonal blocks of L (backward operands)
    char* wplanes[2] = {nullptr, nullptr};
    float* w_iscale = nullptr;    // [2][nb_cap]
    unsigned* wmax = nullptr;     // [2][nb_cap]
    bool w_ready[2] = {false, false};
    int key_merged[2] = {-1, -1};
    unsigned* amax = nullptr;     // [2][nb_cap]
    int* sync = nullptr;
    int64_t sync_ints = 0;
    int4* items[2] = {nullptr, nullptr};
    int64_t items_cap = 0;
    int n_items[2] = {0, 0};
    int nq = 1;                   // queues of the item table: 8 (one per XCD) on a device with 8 XCDs of 32 compute units, else 1
    int q_off[2][kMaxQ + 1] = {};
    int nq_dir[2] = {1, 1};       // queues of the current tables
    int key_mt = 0, key_nb = 0, key_tail = 0;
    std::vector<int4> host_items[2];
    int* host_err = nullptr;      // pinned: error word of the last launches
    hipEvent_t ev_err = nullptr;
    bool err_pending = false;
    bool inv_ready = false;
    int workers = 256;
};

static int64_t tk_items_bound(int64_t mt, int64_t nb) { return mt * (8 * nb * (nb - 1) / 2 + 24 * nb) + 64; }

void tk_destroy(TrsmTickets* tk) {
    if (tk == nullptr) return;
    (void)hipFree(tk->planes_x); (void)hipFree(tk->planes_d); (void)hipFree(tk->rinv_x); (void)hipFree(tk->rinv_d);
    (void)hipFree(tk->xinv_s); (void)hipFree(tk->tinv_s); (void)hipFree(tk->inv_iscale); (void)hipFree(tk->amax);
    (void)hipFree(tk->sync); (void)hipFree(tk->items[0]); (void)hipFree(tk->items[1]);
    (void)hipFree(tk->wtmp); (void)hipFree(tk->ltb); (void)hipFree(tk->wplanes[0]); (void)hipFree(tk->wplanes[1]); (void)hipFree(tk->w_iscale);
    (void)hipFree(tk->wmax);
    if (tk->host_err) (void)hipHostFree(tk->host_err);
    if (tk->ev_err) (void)hipEventDestroy(tk->ev_err);
    delete tk;
}

// Buffers for right-hand-side blocks of up to m_cap rows against a factor of up to np_cap columns.  Returns 1 (and leaves *out NULL)
// when the device has no room: the caller keeps the step-by-step solves.
int tk_create(TrsmTickets** out, int64_t np_cap, int64_t m_cap) {
    *out = nullptr;
    if (np_cap < 2 * kBs || m_cap < TT || m_cap > 64 * TT) return 1;
    TrsmTickets* tk = new (std::nothrow) TrsmTickets();
    NNGP_REQUIRE(tk != nullptr, "tk_create: out of host memory");
    tk->m_cap = round_up(m_cap, TT);
    tk->np_cap = np_cap;
    tk->nb_cap = (np_cap + kBs - 1) / kBs;
    const int64_t nb = tk->nb_cap, mt = tk->m_cap / TT;
    tk->sync_ints = SY_COUNTERS + 3 * mt * nb + mt * nb * 8;
    tk->items_cap = tk_items_bound(mt, nb);
    bool ok = true;
    auto A = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); ok = false; } note_alloc(); };
    A((void**)&tk->planes_x, (size_t)nb * tk->m_cap * kLdp);
    A((void**)&tk->planes_d, (size_t)nb * tk->m_cap * kLdp);
    A((void**)&tk->rinv_x, sizeof(float) * nb * tk->m_cap);
    A((void**)&tk->rinv_d, sizeof(float) * nb * tk->m_cap);
    A((void**)&tk->xinv_s, (size_t)nb * kBs * kLdp);
    A((void**)&tk->tinv_s, (size_t)nb * kBs * kLdp);
    A((void**)&tk->inv_iscale, sizeof(float) * 2 * nb);
    A((void**)&tk->amax, sizeof(unsigned) * 2 * nb);
    A((void**)&tk->sync, sizeof(int) * tk->sync_ints);
    A((void**)&tk->items[0], sizeof(int4) * tk->items_cap);
    A((void**)&tk->items[1], sizeof(int4) * tk->items_cap);
    A((void**)&tk->wtmp, sizeof(float) * nb * kBs * kBs);
    A((void**)&tk->ltb, sizeof(float) * nb * kBs * kBs);
    A((void**)&tk->wplanes[0], (size_t)nb * kBs * kLdp);
    A((void**)&tk->wplanes[1], (size_t)nb * kBs * kLdp);
    A((void**)&tk->w_iscale, sizeof(float) * 2 * nb);
    A((void**)&tk->wmax, sizeof(unsigned) * 2 * nb);
    if (ok && hipHostMalloc(reinterpret_cast<void**>(&tk->host_err), 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) ok = false;
    if (ok && hipEventCreateWithFlags(&tk->ev_err, hipEventDisableTiming) != hipSuccess) ok = false;
    if (!ok) {
        (void)hipGetLastError();
        tk_destroy(tk);
        return 1;
    }
    tk->host_err[0] = tk->host_err[1] = 0;
    if (hipMemset(tk->sync, 0, sizeof(int) * tk->sync_ints) != hipSuccess) {  // (the error word is only ever cleared here and after a report)
        (void)hipGetLastError();
        tk_destroy(tk);
        return 1;
    }
    int dev = 0, count = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && count > 0)
        tk->workers = count;
    // One queue.  NNGP_TK_QUEUES=8 (read when the model is created; a chip of 8 XCDs of 32 compute units only: workgroup i of a launch goes
    // to XCD i % 8, HW_REG_XCC_ID names it): one table per XCD.  Measured at N = 32768, M = 1024 (profiles/r5_tk_queues_ab.txt): the
    // launches fetch 16 % fewer bytes through the fabric (47.9 -> 40.1 GB for three solves: the split rows of X of a pair of row tiles are
    // shared in one L2), bit-identical results, and take the same time (4.34 / 8.45 ms forward / both halves either way; in the step
    // 13.8-13.9 ms against 13.65) -- a stage of a bulk item waits for its slowest piece, and the L panels still come from HBM.  Off.
    tk->nq = 1;
    if (const char* e = getenv("NNGP_TK_QUEUES")) tk->nq = (atoi(e) == 8 && tk->workers == 256) ? kMaxQ : 1;
    *out = tk;
    return 0;
}

// split copies of the inverted diagonal blocks (after triinv_build, on the same stream) and the merged chain operands
//   forward   W_J [c, k] = sum_n L[(J+1) 1024 + c, J 1024 + n] Linv_J[n, k]      (J = 0 .. nb - 2; rows c of the next block)
//   backward  W_J [c, k] = sum_n L[J 1024 + n, (J-1) 1024 + c] Linv_J[k, n]      (J = nb - 1 .. 1; rows c = columns of the previous block)
// as batched float32 products with the triangular inverted blocks (l: the float32 factor, row stride ld), then split like them.
int tk_prepare_inverses(TrsmTickets* tk, const TriInv& ti, int64_t np, const float* l, int64_t ld, hipStream_t s) {
    NNGP_REQUIRE(tk != nullptr && ti.bs == kBs && np <= tk->np_cap, "tk_prepare_inverses: bad arguments");
    const int nb = (int)((np + kBs - 1) / kBs);
    const int tail = (int)(np - (int64_t)(nb - 1) * kBs);  // width of the last block column
    NNGP_HIP_CHECK(hipMemsetAsync(tk->amax, 0, sizeof(unsigned) * 2 * tk->nb_cap, s));
    const dim3 g1(64, (unsigned)nb), g2((unsigned)((int64_t)kBs * (kBs / 8) / 256), (unsigned)nb);
    hipLaunchKernelGGL(k_block_absmax, g1, dim3(256), 0, s, ti.xinv, (int64_t)kBs * kBs, tk->amax);
    hipLaunchKernelGGL(k_block_absmax, g1, dim3(256), 0, s, ti.tinv, (int64_t)kBs * kBs, tk->amax + tk->nb_cap);
    hipLaunchKernelGGL(k_split_blocks, g2, dim3(256), 0, s, ti.xinv, kBs, tk->amax, tk->xinv_s, tk->inv_iscale);
    hipLaunchKernelGGL(k_split_blocks, g2, dim3(256), 0, s, ti.tinv, kBs, tk->amax + tk->nb_cap, tk->tinv_s, tk->inv_iscale + tk->nb_cap);
    NNGP_HIP_CHECK(hipGetLastError());
    tk->inv_ready = true;
    tk->w_ready[0] = tk->w_ready[1] = false;
    // The merged chain updates (and the 128 x 128 near updates that go with them) pay where a solve is bound by its dependency chain,
    // not by the operand traffic of its bulk: measured per predict of three solves after a fit, merged against plain, solves / step in ms --
    // N = 8192, M = 1024: 1.96 / 2.37 (step equal: the operands cost the fit 0.3 ms); N = 16384 NTK: 9.1 / 10.8, step -0.8; N = 32768: 13.6 /
    // 14.0 but step +0.7 (1 ms of operands per fit); N = 65536: 49.1 / 46.0; N = 10800, M = 3600: step +0.6.  Hence: row tiles x block columns.
    static const char* merge_env = getenv("NNGP_TK_MERGE");  // (development aid: 0 never, 1 always)
    const bool want = merge_env != nullptr ? atoi(merge_env) != 0 : (tk->m_cap / TT) * (int64_t)nb <= 192;
    if (l == nullptr || nb < 2 || !want) return 0;
    const int64_t blk = (int64_t)kBs * kBs;
    NNGP_HIP_CHECK(hipMemsetAsync(tk->wmax, 0, sizeof(unsigned) * 2 * tk->nb_cap, s));
    // ---- forward: W_J = L[J+1 block rows, J block columns] T_J^T with T_J = L_JJ^-T (upper): batch over J, the last target block may be a tail
    {
        const int full = (tail == kBs) ? nb - 1 : nb - 2;  // products whose target block has 1024 rows
        if (full > 0)
            NNGP_TRY(launch_gemm_nt_f32_batched(tk->wtmp, kBs, l + (int64_t)kBs * ld, ld, ti.tinv, kBs, kBs, kBs, kBs, 1.0f, 0.0f, false, full, blk,
                                                (int64_t)kBs * ld + kBs, blk, s, 2));
        if (full < nb - 1) {  // J = nb - 2 onto the tail block: `tail` rows; the rest of its W block is cleared (never read, but split)
            NNGP_HIP_CHECK(hipMemsetAsync(tk->wtmp + (int64_t)(nb - 2) * blk, 0, sizeof(float) * blk, s));
            NNGP_TRY(launch_gemm_nt_f32(tk->wtmp + (int64_t)(nb - 2) * blk, kBs, l + (int64_t)(nb - 1) * kBs * ld + (int64_t)(nb - 2) * kBs, ld,
                                        ti.tinv + (int64_t)(nb - 2) * blk, kBs, tail, kBs, kBs, 1.0f, 0.0f, false, s, 2));
        }
        const dim3 gw1(64, (unsigned)(nb - 1)), gw2((unsigned)((int64_t)kBs * (kBs / 8) / 256), (unsigned)(nb - 1));
        hipLaunchKernelGGL(k_block_absmax, gw1, dim3(256), 0, s, tk->wtmp, blk, tk->wmax);
        hipLaunchKernelGGL(k_split_blocks, gw2, dim3(256), 0, s, tk->wtmp, kBs, tk->wmax, tk->wplanes[0], tk->w_iscale);
        NNGP_HIP_CHECK(hipGetLastError());
        tk->w_ready[0] = true;
    }
    // ---- backward: W_J = (L[J block rows, J-1 block columns])^T X_J^T with X_J = L_JJ^-1 (lower): transposed blocks first (index J - 1)
    {
        hipLaunchKernelGGL(k_transpose_1024, dim3(32, 32, (unsigned)(nb - 1)), dim3(256), 0, s, l + (int64_t)kBs * ld, ld, (int64_t)kBs * ld + kBs, tk->ltb, tail,
                           nb - 1);
        const int full = (tail == kBs) ? nb - 1 : nb - 2;  // J = 1 .. full with a 1024-wide source block
        if (full > 0)
            NNGP_TRY(launch_gemm_nt_f32_batched(tk->wtmp + blk, kBs, tk->ltb, kBs, ti.xinv + blk, kBs, kBs, kBs, kBs, 1.0f, 0.0f, false, full, blk, blk, blk, s, 1));
        if (full < nb - 1) {  // J = nb - 1: a source block of `tail` rows -- K = tail, `tail` columns of W
            NNGP_HIP_CHECK(hipMemsetAsync(tk->wtmp + (int64_t)(nb - 1) * blk, 0, sizeof(float) * blk, s));
            NNGP_TRY(launch_gemm_nt_f32(tk->wtmp + (int64_t)(nb - 1) * blk, kBs, tk->ltb + (int64_t)(nb - 2) * blk, kBs, ti.xinv + (int64_t)(nb - 1) * blk, kBs,
                                        kBs, tail, tail, 1.0f, 0.0f, false, s, 1));
        }
        const dim3 gw1(64, (unsigned)(nb - 1)), gw2((unsigned)((int64_t)kBs * (kBs / 8) / 256), (unsigned)(nb - 1));
        hipLaunchKernelGGL(k_block_absmax, gw1, dim3(256), 0, s, tk->wtmp + blk, blk, tk->wmax + tk->nb_cap + 1);
        hipLaunchKernelGGL(k_split_blocks, gw2, dim3(256), 0, s, tk->wtmp + blk, kBs, tk->wmax + tk->nb_cap + 1, tk->wplanes[1] + (int64_t)kBs * kLdp,
                           tk->w_iscale + tk->nb_cap + 1);
        NNGP_HIP_CHECK(hipGetLastError());
        tk->w_ready[1] = true;
    }
    return 0;
}

// the ticket table of a shape, four ints per item {type | panels << 4, r, c or q, J}, for the host-side proof that every item's
// dependencies hold lower tickets (tests/test_host.py)
int tk_order_export(int mt, int nb, int tail_ct, int backward, int workers, int merged, int queues, int32_t* out, int32_t* queue_of, int64_t cap,
                    int64_t* count) {
    NNGP_REQUIRE(mt >= 1 && nb >= 1 && tail_ct >= 1 && tail_ct <= 8 && workers >= 1 && count != nullptr && (queues == 1 || queues == kMaxQ),
                 "trsm_ticket_order: bad shape");
    std::vector<int4> items;
    std::vector<int> qof;
    tk_build_order(mt, nb, tail_ct, backward != 0, workers, merged != 0, items, queues, &qof);
    *count = (int64_t)items.size();
    if (out != nullptr) {
        NNGP_REQUIRE((int64_t)items.size() <= cap, "trsm_ticket_order: %lld items, room for %lld", (long long)items.size(), (long long)cap);
        for (size_t i = 0; i < items.size(); ++i) {
            out[4 * i] = items[i].x; out[4 * i + 1] = items[i].y; out[4 * i + 2] = items[i].z; out[4 * i + 3] = items[i].w;
            if (queue_of != nullptr) queue_of[i] = qof[i];
        }
    }
    return 0;
}

bool tk_inverses_ready(const TrsmTickets* tk) { return tk != nullptr && tk->inv_ready; }
void tk_invalidate_inverses(TrsmTickets* tk) { if (tk != nullptr) tk->inv_ready = false; }

bool tk_usable(const TrsmTickets* tk, int64_t m, int64_t np) {
    return tk != nullptr && tk->inv_ready && m % TT == 0 && m >= TT && m <= tk->m_cap && np % TT == 0 && np <= tk->np_cap && np >= 2 * kBs;
}

// B[m, np] <- B L^-T (backward = false; sw.planes) or B L^-1 (backward = true; sw.planes_t) in one launch
int tk_solve(TrsmTickets* tk, float* b, int64_t ldb, int64_t m, int64_t np, const SplitWork& sw, bool backward, hipStream_t s, int reserve_cus) {
    NNGP_REQUIRE(tk_usable(tk, m, np), "tk_solve: not usable for m=%lld np=%lld", (long long)m, (long long)np);
    NNGP_REQUIRE(sw.k_cap == kBs && (backward ? (sw.planes_t != nullptr && sw.lt_ready) : (sw.planes != nullptr && sw.l_ready)),
                 "tk_solve: split copy of the factor not available");
    NNGP_REQUIRE(ldb % 32 == 0 && ldb < (1LL << 26) && ((uintptr_t)b & 127) == 0, "tk_solve: right-hand sides must be 128-byte aligned rows");
    const int mt = (int)(m / TT), nb = (int)((np + kBs - 1) / kBs);
    const int tail_ct = (int)((np - (int64_t)(nb - 1) * kBs) / TT);
    if (tk->key_mt != mt || tk->key_nb != nb || tk->key_tail != tail_ct || tk->key_merged[0] != (int)tk->w_ready[0] || tk->key_merged[1] != (int)tk->w_ready[1]) {
        // a new shape (the first solve, or another batch size): the device may still read the old tables
        NNGP_HIP_CHECK(hipDeviceSynchronize());
        for (int dir = 0; dir < 2; ++dir) {
            std::vector<int4> order;
            std::vector<int> qof;
            // (eight tables only for shapes whose launches fill the chip: every XCD then has workgroups drawing from its table)
            const int nq = (tk->nq == kMaxQ && (int64_t)mt * nb >= 32) ? kMaxQ : 1;
            tk->nq_dir[dir] = nq;
            tk_build_order(mt, nb, tail_ct, dir == 1, tk->workers, tk->w_ready[dir], order, nq, &qof);
            tk->key_merged[dir] = (int)tk->w_ready[dir];
            NNGP_REQUIRE((int64_t)order.size() <= tk->items_cap, "tk_solve: item table overflow");
            // the queues' tables one behind the other, each in the global start order
            std::vector<int4>& tab = tk->host_items[dir];
            tab.clear();
            tab.reserve(order.size());
            for (int q = 0; q < nq; ++q) {
                tk->q_off[dir][q] = (int)tab.size();
                for (size_t i = 0; i < order.size(); ++i)
                    if (qof[i] == q) tab.push_back(order[i]);
            }
            for (int q = nq; q <= kMaxQ; ++q) tk->q_off[dir][q] = (int)tab.size();
            tk->n_items[dir] = (int)tab.size();
            NNGP_HIP_CHECK(hipMemcpy(tk->items[dir], tab.data(), sizeof(int4) * tab.size(), hipMemcpyHostToDevice));
        }
        tk->key_mt = mt; tk->key_nb = nb; tk->key_tail = tail_ct;
    }
    const int dir = backward ? 1 : 0;
    TkParams P{};
    P.b = b; P.ldb = ldb;
    P.lplanes = backward ? sw.planes_t : sw.planes;
    P.l_stride = sw.col_stride;
    P.dinv_s = backward ? tk->tinv_s : tk->xinv_s;
    P.dinv_iscale = tk->inv_iscale + (backward ? tk->nb_cap : 0);
    P.wplanes = tk->wplanes[backward ? 1 : 0];
    P.w_iscale = tk->w_iscale + (backward ? tk->nb_cap : 0);
    P.planes_x = tk->planes_x; P.planes_d = tk->planes_d;
    P.p_stride = tk->m_cap * kLdp;
    P.rinv_x = tk->rinv_x; P.rinv_d = tk->rinv_d;
    P.r_stride = tk->m_cap;
    P.items = tk->items[dir];
    P.n_items = tk->n_items[dir];
    P.nq = tk->nq_dir[dir];
    for (int q = 0; q <= kMaxQ; ++q) P.q_off[q] = tk->q_off[dir][q];
    P.sync = tk->sync;
    P.mt = mt; P.nb = nb; P.ctiles = (nb - 1) * 8 + tail_ct; P.tail_ct = tail_ct;
    P.l_iscale = 1.0f / sw.scale;
    P.backward = backward ? 1 : 0;
    P.spin_ticks = 200000000ULL;  // 2 s
    P.spin_max = 1u << 22;        // > 4 s of polls at ~1 us each
    P.fault_ticket = -1;
    if (const char* f = getenv("NNGP_TK_FAULT")) {  // test hook (tests/test_gpu_parity.py): lose one item's completion, give up after 50 ms
        P.fault_ticket = atoi(f);
        P.spin_ticks = 5000000ULL;
    }
    const int64_t used = SY_COUNTERS + 3LL * mt * nb + (int64_t)mt * P.ctiles;
    NNGP_REQUIRE(used <= tk->sync_ints, "tk_solve: counter block overflow");
    // the error word survives (sticky until the host has read it); ticket and counters restart
    NNGP_HIP_CHECK(hipMemsetAsync(tk->sync, 0, sizeof(int) * SY_ERROR, s));
    NNGP_HIP_CHECK(hipMemsetAsync(tk->sync + SY_ERROR + 1, 0, sizeof(int) * (used - SY_ERROR - 1), s));
    // reserve_cus: compute units left to a kernel of another stream that needs whole units (the cut of K's digit planes: 132 KB of LDS
    // per workgroup); the table was scheduled for all of them, which only costs fidelity of the simulated order
    int grid = tk->workers - (reserve_cus > 0 && reserve_cus < tk->workers / 2 ? reserve_cus : 0);
    if (grid > P.n_items) grid = P.n_items;
    hipLaunchKernelGGL(k_trsm_tickets, dim3((unsigned)grid), dim3(512), 0, s, P);
    NNGP_HIP_CHECK(hipGetLastError());
    NNGP_HIP_CHECK(hipMemcpyAsync(tk->host_err, tk->sync + SY_ERROR, sizeof(int), hipMemcpyDeviceToHost, s));
    NNGP_HIP_CHECK(hipEventRecord(tk->ev_err, s));
    tk->err_pending = true;
#ifdef NNGP_TIMING_KNOBS
    if (getenv("NNGP_TK_STAMPS") != nullptr) {
        unsigned long long h[28];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tk_stamps), sizeof(h)) == hipSuccess) {
            static const char* names[6] = {"SB", "D", "SX", "U", "UB", "UW"};
            fprintf(stderr, "tk stamps (%s, mt %d nb %d): %llu workgroups, mean kernel time per workgroup %.1f us\n", backward ? "backward" : "forward", mt, nb, h[25],
                    h[25] ? 0.01 * (double)h[24] / (double)h[25] : 0.0);
            for (int t = 0; t < 6; ++t)
                fprintf(stderr, "   %-2s items %6llu: wait %.2f us  body %.2f us  publish %.2f us per item; share of workgroup time: wait %.1f %% body %.1f %% publish %.1f %%\n", names[t],
                        h[t * 4 + 3], h[t * 4 + 3] ? 0.01 * h[t * 4] / h[t * 4 + 3] : 0.0, h[t * 4 + 3] ? 0.01 * h[t * 4 + 1] / h[t * 4 + 3] : 0.0,
                        h[t * 4 + 3] ? 0.01 * h[t * 4 + 2] / h[t * 4 + 3] : 0.0, 100.0 * h[t * 4] / (double)h[24], 100.0 * h[t * 4 + 1] / (double)h[24], 100.0 * h[t * 4 + 2] / (double)h[24]);
            unsigned long long z[28] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tk_stamps), z, sizeof(z));
        }
    }
#endif
    if (getenv("NNGP_TK_DUMP") != nullptr && !backward) {  // development aid: what the split rows of block column 0 hold after the solve
        (void)hipDeviceSynchronize();
        std::vector<_Float16> h(8 * 2048);
        for (int row : {6, 7, 14, 15, 135}) {
            if (hipMemcpy(h.data(), tk->planes_d + (int64_t)row * kLdp, kLdp, hipMemcpyDeviceToHost) != hipSuccess) break;
            int bad = 0, firstbad = -1;
            for (int kb = 0; kb < 32; ++kb)
                for (int e = 0; e < 64; ++e) {
                    const float v = (float)h[kb * 64 + e];
                    if (!(fabsf(v) <= 16384.0f)) { ++bad; if (firstbad < 0) firstbad = kb * 64 + e; }
                }
            fprintf(stderr, "tk dump: planes_d[0] row %d: %d halfs out of range (first at %d); kb 0: %g %g, kb 24: %g %g %g %g, kb 31: %g\n", row, bad, firstbad,
                    (double)(float)h[0], (double)(float)h[1], (double)(float)h[24 * 64], (double)(float)h[24 * 64 + 1], (double)(float)h[24 * 64 + 8], (double)(float)h[24 * 64 + 32], (double)(float)h[31 * 64]);
        }
    }
    if (getenv("NNGP_TK_WATCH") != nullptr) {  // development aid: watch the launch from a second stream
        hipStream_t w = nullptr;
        std::vector<int> h((size_t)used);
        if (hipStreamCreateWithFlags(&w, hipStreamNonBlocking) == hipSuccess) {
            for (int it = 0; it < 40; ++it) {
                if (hipEventQuery(tk->ev_err) == hipSuccess) { fprintf(stderr, "tk watch: launch finished (poll %d), error word 0x%x\n", it, tk->host_err[0]); break; }
                (void)hipGetLastError();
                usleep(250000);
                if (hipMemcpyAsync(h.data(), tk->sync, sizeof(int) * (size_t)used, hipMemcpyDeviceToHost, w) != hipSuccess || hipStreamSynchronize(w) != hipSuccess) break;
                long long sx = 0, sd = 0, sb = 0, su = 0;
                for (int i = 0; i < mt * nb; ++i) { sd += h[SY_COUNTERS + i]; sx += h[SY_COUNTERS + mt * nb + i]; sb += h[SY_COUNTERS + 2 * mt * nb + i]; }
                for (int i = 0; i < mt * P.ctiles; ++i) su += h[SY_COUNTERS + 3 * mt * nb + i];
                fprintf(stderr, "tk watch %d: ticket %d / %d, error 0x%x, done: D %lld SX %lld SB %lld U-panels %lld\n", it, h[0], P.n_items, h[1], sd, sx, sb, su);
            }
            (void)hipStreamDestroy(w);
        }
    }
    return 0;
}

// 0: no error seen (wait = false: only looks if the last launch has completed); otherwise the error word of a timed-out launch.
// The word is cleared on the device once reported.
int tk_poll_error(TrsmTickets* tk, bool wait) {
    if (tk == nullptr || !tk->err_pending) return 0;
    if (wait) {
        if (hipEventSynchronize(tk->ev_err) != hipSuccess) return -1;
    } else if (hipEventQuery(tk->ev_err) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    tk->err_pending = false;
    const int e = tk->host_err[0];
    if (e != 0) {
        tk->host_err[0] = 0;
        (void)hipMemset(tk->sync + SY_ERROR, 0, sizeof(int));
    }
    return e;
}

}  // namespace nngp
