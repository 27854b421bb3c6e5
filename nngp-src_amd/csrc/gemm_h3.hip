// "Split-float16" NT GEMM on the gfx950 matrix cores:  C[M,N] = beta*C + alpha * A[M,K] * B[N,K]^T  with float32
// operands carried as two float16 planes (a*s = hi + lo, hi = fp16(a*s), lo = fp16(a*s - hi), both rounded to nearest:
// |a*s - hi - lo| <= 2^-24 |a*s|, and float16 subnormals are exact on this pipe) and three v_mfma_f32_32x32x16_f16
// products per term (hi*hi + hi*lo + lo*hi; lo*lo ~ 2^-24 relative is dropped), accumulated in float32.  The float16
// matrix pipe runs 16x the float32 one, so a float32-grade product costs 3/16 of the float32-MFMA time.
// Used for the large trailing updates of the blocked Cholesky (reference: cho_factor via
// nt.predict, train.py:171-172) and for the large updates of the posterior's blocked triangular solves (row a4).
// Accuracy: on random and on real factor panels the result is closer to the float64 product than the float32-MFMA
// kernel's (2.6e-7 vs 4.0e-7 relative Frobenius).  One caveat: the float16 pipe's accumulator truncates toward zero,
// a multiplicative bias of -2.6e-8 per 1024 k on same-sign sums and none on mixed signs (scripts/h3_bias.py) -- see
// potrf_lookahead_f32 for the one place where that matters.
//
// Operand format ("split rows", written by k_split_rows): row r is [K/32] blocks of 128 bytes, each block = 32 hi
// halfs followed by 32 lo halfs of the same 32 k -- so a row's k-block is one 128-byte line, and the panel has the same
// footprint as its float32 original (4 bytes per element).  s is a power of two chosen by the caller so that
// max|a*s| < 2^15 (no float16 overflow).
//
// Kernel: 512-thread workgroup = 8 waves (2 x 4), tile 256x256, wave sub-tile 128x64 = 4x2 accumulators of 32x32
// (128 VGPRs), BK = 32 per stage (two k16 MFMA sub-steps), LDS 2 stages x 64 KB filled by global_load_lds_dwordx4,
// 128-byte rows with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 (conflict-free for the ds_read_b128 lane
// groups of gfx950).  K blocks are consumed from the high end down (accumulation order of the Cholesky updates, see
// gemm_f32.hip).  The grid is persistent: one workgroup per compute unit (128 KB of LDS each) pulls tiles from per-XCD
// work counters, which lets the caller keep a few compute units free for a concurrent latency-critical stream.
#include <atomic>

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

// One row per wave: split a float32 row with its OWN power-of-two scale (largest entry -> 2^14), write 1/scale to
// row_inv (the GEMM epilogue multiplies it back) and, optionally, copy the float32 row to copy_dst.  Used for the
// right-hand-side blocks of the blocked triangular solves, whose magnitudes change from sweep to sweep.
template <int NV>
__global__ __launch_bounds__(256) void k_split_rows_rowscale(const float* __restrict__ src, int64_t ld_src, int64_t rows,
                                                             int k, float* __restrict__ copy_dst, int64_t ld_copy,
                                                             char* __restrict__ out, int64_t out_ld, int64_t pan_stride,
                                                             float* __restrict__ row_inv) {
    // NV float4 per lane: k <= 256 NV.  Columns [1024 p, 1024 p + 1024) of a row go to panel p, pan_stride bytes after panel p - 1
    // (round 4: a step of the blocked solves covers several 1024-column panels; ONE scale per row over all of them)
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* p = src + r * ld_src;
    f32x4 v[NV];
    float mx = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k0 = 4 * (lane + 64 * i);
        v[i] = (k0 < k) ? *reinterpret_cast<const f32x4*>(p + k0) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[i][e]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    int e2 = 0;
    (void)frexpf(mx, &e2);  // mx = f * 2^e2, f in [0.5, 1)
    const float sc = (mx > 0.0f && mx < 3.0e38f) ? ldexpf(1.0f, 14 - e2) : 1.0f;
    if (lane == 0) row_inv[r] = 1.0f / sc;
    char* o = out + r * out_ld;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k0 = 4 * (lane + 64 * i);
        if (k0 >= k) continue;
        if (copy_dst != nullptr) *reinterpret_cast<f32x4*>(copy_dst + r * ld_copy + k0) = v[i];
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = v[i][e] * sc;
            const _Float16 h = (_Float16)a;
            hi[e] = h;
            lo[e] = (_Float16)(a - (float)h);
        }
        const int kp = k0 & 1023;
        char* dst = o + (int64_t)(k0 >> 10) * pan_stride + (int64_t)(kp >> 5) * 128 + (kp & 31) * 2;
        *reinterpret_cast<h4*>(dst) = hi;
        *reinterpret_cast<h4*>(dst + 64) = lo;
    }
}

// Split copy of L^T by block row: for block row j (rows [o, o+sz) of L, o = j*bs) and every column r < o,
// out_j[r][k] = L[o + k][r] -- the operand of the "B L^-1" half of the blocked solves.  One 32 x 32 tile per workgroup
// over the (row of L, column of L) plane, transposed through LDS; tiles on or right of the block diagonal exit.
__global__ __launch_bounds__(256) void k_split_lower_t(const float* __restrict__ l, int64_t ld, int64_t n, int64_t bs,
                                                       float scale, char* __restrict__ out, int64_t col_stride) {
    __shared__ float tile[32][33];
    const int64_t gk0 = (int64_t)blockIdx.y * 32;  // rows of L (k of the output)
    const int64_t r0 = (int64_t)blockIdx.x * 32;   // columns of L (rows of the output)
    const int64_t j = gk0 / bs, o = j * bs;
    if (r0 >= o) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = l[(gk0 + ty + 8 * i) * ld + r0 + tx];
    __syncthreads();
    if (threadIdx.x < 128) {
        const int rr = threadIdx.x >> 2, g = threadIdx.x & 3;  // output row, group of 8 k
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = tile[g * 8 + e][rr] * scale;
            const _Float16 h = (_Float16)a;
            hi[e] = h;
            lo[e] = (_Float16)(a - (float)h);
        }
        char* dst = out + j * col_stride + (r0 + rr) * (bs * 4) + ((gk0 - o) >> 5) * 128 + g * 16;
        *reinterpret_cast<h8*>(dst) = hi;
        *reinterpret_cast<h8*>(dst + 64) = lo;
    }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// number of valid block columns of block row gr (blocks of br x bc tiles); lower: tiles with bj <= bi + sh_t
__host__ __device__ __forceinline__ int h3_block_cols(int gr, int br, int bc, int tiles_m, int tiles_n, bool lower, int sh_t) {
    int last = tiles_n - 1;
    if (lower) {
        int rmax = gr * br + br - 1;
        if (rmax > tiles_m - 1) rmax = tiles_m - 1;
        if (rmax + sh_t < last) last = rmax + sh_t;
    }
    return last / bc + 1;
}

// Tile order: the workgroups with blockIdx % 8 == x run on XCD x and share the work counter x; slot s of XCD x is tile
// (s % (br*bc)) of tile block (s / (br*bc)) * 8 + x, blocks of br x bc tiles enumerated row by row over the blocks that
// contain work -- so the workgroups resident on an XCD share br A panels and bc B panels through its L2.
//
// LOWER: only tiles that touch the region col <= row + diag_shift are computed, and inside them only the 32x32
// sub-tiles whose 128-block column index <= 128-block row index (the same element set the float32 kernel writes).
//
// Schedule inside a tile: waves 0-3 (upper 128 rows, group 0) and waves 4-7 (group 1) run the identical phase sequence
// one barrier apart, so on every SIMD one wave is inside its MFMA cluster while the other issues its LDS fragment reads
// and its direct-to-LDS loads.  Phase = { ds_read fragments of one k16 sub-step; lgkmcnt(0); barrier; 24 MFMA; barrier };
// two phases per BK = 32 stage, raw s_barrier only (a __syncthreads would drain the loads in flight).
// Hazards (b = buffer of stage t; G1's barriers pair with G0's next one):
//   WAR  stage t+1 is loaded into buffer 1-b after the issuing group's last barrier of stage t-1; the other group's
//        last reads of that buffer were retired by the lgkmcnt(0) in front of the barrier paired with it.
//   RAW  a group reads in stage t+1 only pieces whose loading waves executed vmcnt(0) before a barrier the reader has
//        passed: G0 loads all B rows and A rows 0-127 (everything G0 reads; G1 reads B one barrier later still), G1 loads
//        A rows 128-255 (read by G1 only).
// ablate (timing diagnostics only): 1 no operand loads after the first stage, 2 no MFMA, 8 no C traffic.
template <bool LOWER>
__global__ __launch_bounds__(512) void k_gemm_nt_h3(float* C, int64_t ldc, const char* A, const char* B, int64_t ldp,
                                                    int m, int n, int tiles_m, int nk, float alpha, float beta,
                                                    int diag_shift, int order_br, int order_bc, int* counters,
                                                    int slots_per_xcd, int ablate, const float* row_alpha) {
    // ONE __shared__ object: with a second one beside the LDS-DMA staging array hipcc attaches alias scopes to the LDS accesses
    // and then drains the DMA queue (s_waitcnt vmcnt(0)) in front of EVERY phase's fragment reads -- the prefetch of stage t+1
    // never overlapped the second phase of stage t (rounds 1-2 shipped that: .s lines "s_waitcnt vmcnt(0); ds_read_b128").  The
    // work-slot word therefore lives in the tail of the same array.
    __shared__ __attribute__((aligned(1024))) char smem[2 * HSTAGE + 16];
    int& s_slot = *reinterpret_cast<int*>(smem + 2 * HSTAGE);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (n + HT - 1) / HT;
    const int xcd = blockIdx.x & 7;
    const int group = __builtin_amdgcn_readfirstlane(wr);
    const int w4 = __builtin_amdgcn_readfirstlane(wc);
    const int l3 = lane >> 3;
    const int frow = lane & 31, fh = lane >> 5;
    // direct-to-LDS pieces of 8 rows: wave (group g, index w4) moves A pieces g*16 + e*4 + w4 (e < 4) and, in group 0,
    // B pieces e*4 + w4 (e < 8).  piece parity = w4 & 1, so the source swizzle is one per-lane constant.
    const int64_t lane_off = (int64_t)l3 * ldp + (((lane & 7) ^ ((4 * (w4 & 1) + (l3 >> 1)) & 7)) << 4);

    for (;;) {
        if (tid == 0) s_slot = atomicAdd(&counters[xcd], 1);
        __syncthreads();
        const int slot = s_slot;
        __syncthreads();  // s_slot may be rewritten only after every wave has read it
        if (slot >= slots_per_xcd) {
            // The work counters reset themselves: the last workgroup to run out of work (every other one has taken its final
            // slot by then) zeroes them for the next launch -- no memset kernel in front of every launch (5 us each, ~130 per fit
            // and predict at N = 32768).  counters[8] counts the workgroups that are done.
            // (No fence: this thread's own slot fetches have returned their values before it counts itself done, and the end
            // of the kernel publishes the zeros; an agent-scope release here would write back the XCD's L2.)
            if (tid == 0 && atomicAdd(&counters[8], 1) == (int)gridDim.x - 1)
                for (int i = 0; i < 9; ++i) __hip_atomic_store(&counters[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        int bi, bj;
        {
            const int per = order_br * order_bc;
            int G = (slot / per) * 8 + xcd;
            const int i = slot % per;
            const int brows = (tiles_m + order_br - 1) / order_br;
            const int sh_t = (diag_shift + HT - 1) / HT;
            int gr = 0;
            for (; gr < brows; ++gr) {
                const int cnt = h3_block_cols(gr, order_br, order_bc, tiles_m, tiles_n, LOWER, sh_t);
                if (G < cnt) break;
                G -= cnt;
            }
            if (gr >= brows) continue;
            bi = gr * order_br + (i % order_br);
            bj = G * order_bc + (i / order_br);
            if (bi >= tiles_m || bj >= tiles_n) continue;
        }
        if (LOWER && bj * HT > bi * HT + HT - 1 + diag_shift) continue;

        const char* Ab = A + (int64_t)bi * HT * ldp;
        const char* Bb = B + (int64_t)bj * HT * ldp;
        auto glds_stage = [&](int t, int buf) {
            const int64_t koff = (int64_t)(nk - 1 - t) * 128 + lane_off;
            char* dst = smem + buf * HSTAGE;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int piece = group * 16 + e * 4 + w4;
                __builtin_amdgcn_global_load_lds((glb_void*)(Ab + (int64_t)piece * 8 * ldp + koff),
                                                 (lds_void*)(dst + piece * 1024), 16, 0, 0);
            }
            if (group == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int piece = e * 4 + w4;
                    __builtin_amdgcn_global_load_lds((glb_void*)(Bb + (int64_t)piece * 8 * ldp + koff),
                                                     (lds_void*)(dst + HT * HROW + piece * 1024), 16, 0, 0);
                }
            }
        };

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        glds_stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (group == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs one barrier behind group 0
        for (int t = 0; t < nk; ++t) {
            const char* sa_ = smem + (t & 1) * HSTAGE;
            const char* sb_ = sa_ + HT * HROW;
#pragma unroll
            for (int kk = 1; kk >= 0; --kk) {
                h8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wc * 64 + j * 32 + frow;
                    bh[j] = *reinterpret_cast<const h8*>(sb_ + lds_off(row, kk * 2 + fh));
                    bl[j] = *reinterpret_cast<const h8*>(sb_ + lds_off(row, 4 + kk * 2 + fh));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = wr * 128 + i * 32 + frow;
                    ah[i] = *reinterpret_cast<const h8*>(sa_ + lds_off(row, kk * 2 + fh));
                    al[i] = *reinterpret_cast<const h8*>(sa_ + lds_off(row, 4 + kk * 2 + fh));
                }
                if (kk == 1 && t + 1 < nk && !(ablate & 1)) glds_stage(t + 1, (t + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                if (!(ablate & 2)) {
                    // product-major order: consecutive MFMAs write different accumulators (small terms first)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                } else {  // keep the fragment reads alive without the matrix work
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]));
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(bh[j]), "v"(bl[j]));
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of stage t+1 have landed
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (group == 0) __builtin_amdgcn_s_barrier();  // every wave executes the same number of barriers

        // epilogue: acc[i][j][r] is element (row, col) with row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31.
        // (Measured and dropped: beta == 1 as 128 no-return global_atomic_add_f32 per lane instead of load, add, store --
        // no C round trips in the wave, one add per element so the same sum; the Cholesky at N = 32768 went from 50.5 to
        // 51.4 ms and the solves of the posterior from 55.2 to 56.7: the L2's read-modify-write costs more than the loads it saves.)
        // The C loads are HBM-latency bound (~2.5 us a round trip): the values of two sub-tiles are requested together, and
        // the in/out tests use wave-uniform indices so that they are scalar branches (per-lane tests made hipcc wait for
        // every sub-tile's loads separately).
        const int row_base = bi * HT + group * 128;  // (group, w4) = (wr, wc) in scalar registers
        const int col_base = bj * HT + w4 * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float cold[2][16], ra[16];
            bool live[2];
            if (row_alpha != nullptr) {  // per-row operand scales of this 32-row band (both column sub-tiles share them)
                const int rr = row_base + i * 32 + 4 * fh;
#pragma unroll
                for (int r = 0; r < 16; ++r) ra[r] = (rr < m) ? row_alpha[rr + (r & 3) + 8 * (r >> 2)] : 1.0f;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r0 = row_base + i * 32, c0 = col_base + j * 32;
                // m, n are multiples of 128: a 32x32 sub-tile is inside or outside
                live[j] = r0 < m && c0 < n && !(LOWER && (c0 >> 7) > ((r0 + diag_shift) >> 7)) && !(ablate & 8);
                if (live[j] && beta != 0.0f) {
                    const float* p0 = C + (int64_t)(r0 + 4 * fh) * ldc + c0 + frow;
#pragma unroll
                    for (int r = 0; r < 16; ++r) cold[j][r] = p0[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!live[j]) {
                    asm volatile("" ::"v"(acc[i][j]));
                    continue;
                }
                const int r0 = row_base + i * 32, c0 = col_base + j * 32;
                float* p0 = C + (int64_t)(r0 + 4 * fh) * ldc + c0 + frow;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = alpha * acc[i][j][r];
                    if (row_alpha != nullptr) v *= ra[r];
                    if (beta != 0.0f) v = fmaf(beta, cold[j][r], v);
                    p0[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = v;
                }
            }
        }
    }
}


// ---- second form of the same product (round 3) ----
// Same tile (256 x 256, 8 waves as 2 x 4, wave sub-tile 128 x 64), same LDS image and swizzle, same persistent grid.  Differences:
//   * v_mfma_f32_16x16x32_f16 (one k32 step per instruction) instead of 32x32x16: the chip holds a higher clock on this shape
//     under a full MFMA load (MI355X_MICROARCH.md, DVFS give-back item 7), and with the operands SWAPPED -- the B fragment
//     as the instruction's first operand -- a lane's four results are four CONSECUTIVE COLUMNS of one row of C, so the
//     epilogue reads and writes C with 16-byte accesses (32 + 32 instructions per lane and tile instead of 128 + 128).
//   * the direct-to-LDS loads of stage t+1 are spread evenly: every wave issues 4 pieces in each of the stage's two phases
//     (the first form: 12 / 0 in wave group 0 and 4 / 0 in group 1), and no wait drains the queue inside the loop: counted
//     vmcnt(4) / vmcnt(2), so every piece has at least a full phase pair (~1500 cycles) to land.
//   * two panels per launch (A2, B2, nk2): C is read and written once for a 2048-deep update (depth-2 look-ahead Cholesky).
// Phases of stage t (buffer t & 1; group 1 runs one barrier behind group 0):
//   A: read the 8 B fragments and the A fragments of the wave's rows 0-63; issue the wave's 4 B pieces of stage t+1; vmcnt(4)
//      (retires the wave's last two A pieces of stage t: rows 64-127 of its group, read in phase B); barrier; 48 MFMA; barrier
//   B: read the A fragments of rows 64-127; issue the wave's 4 A pieces of stage t+1 (rows 0-63 first); vmcnt(4) (retires the B
//      pieces issued in phase A); barrier; 48 MFMA; vmcnt(2) (retires the A pieces of rows 0-63); barrier
// RAW: B pieces are read by both groups in phase A of stage t+1; a group-1 wave retires its B pieces before ITS barrier B1 of
// phase B, which is group 0's B2 of phase B -- the last barrier group 0 passes before those reads; group 0's own are retired one
// barrier earlier still.  A pieces are read only by the group that loads them, after the barrier that follows their wait.
// WAR: buffer (t+1) & 1 was last read in stage t-1, whose reads every wave retired (lgkmcnt(0)) before a barrier that precedes
// any issue of stage t.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Tile boundaries (a workgroup's compute unit idles from its last MFMA of one tile to the first of the next): the work slot of the
// NEXT tile is fetched during the last stage of the current one, the next tile's first operand stage is requested BEFORE the
// current tile's epilogue, and the epilogue itself requests the old C values of three 32-row quarters at once, combines them in
// place as they arrive and issues all 32 stores at the end -- no store sits between two loads in the in-order vmcnt queue, so no
// wait for a load ever includes a store's acknowledgement, and the next main loop starts without waiting for the stores.
struct H3Region {
    int64_t c_off;  // elements from C to the region's origin
    int64_t a_off;  // bytes from A to the split rows of the region's first row
    int64_t b_off;  // bytes from B to the split rows of the region's first column
    int m, n, shift, blk0;  // extent, diagonal shift (LOWER: col <= row + shift), first tile block in the launch's enumeration
};
struct H3Regions {
    int count, nblk;
    H3Region r[4];
};

#ifdef NNGP_TIMING_KNOBS
// ABL = 256: cycles (s_memtime) of workgroup 0, waves 0 and 4, per tile segment, summed over its tiles:
// [w][0] main loop, [1] slot + decode, [2] old C requested -> combined, [3] stores issued, [4] barrier + next stage 0 landed,
// [5] tiles, [6] s_memtime over the kernel, [7] s_memrealtime (100 MHz) over the kernel
__device__ unsigned long long g_h3_stamps[2 * 8];
#endif
// ABL (timing diagnostics, instantiated in libnngp_hip_knobs.so only; a run-time switch changed the code hipcc generates for the
// epilogue): 1 no operand loads after the first stage, 2 no MFMA, 8 no C traffic, 16 no C loads, 32 no C stores.
template <bool LOWER, int ABL>
__global__ __launch_bounds__(512) void k_gemm_nt_h3v2(float* C, int64_t ldc, const char* A, const char* B, int64_t ldp,
                                                      int64_t pstride, int npanels, int nk_first, int first_off, H3Regions regs, int nk,
                                                      float alpha, float beta, int order_br, int order_bc,
                                                      int* counters, int slots_per_xcd, const float* row_alpha, int total_wgs) {
    constexpr int ablate = ABL;
    // ONE object (see k_gemm_nt_h3), all 160 KB: two operand stages of 64 KB; in the epilogue 20 KB of C staging per wave.  The
    // work-slot word is only touched between a tile's last fragment read and its epilogue, when nothing else lives in LDS.
    __shared__ __attribute__((aligned(1024))) char smem[8 * 20480];
    int& s_slot = *reinterpret_cast<int*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7;
    const int group = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int w4 = __builtin_amdgcn_readfirstlane(wave & 3);
    const int l3 = lane >> 3;
    const int r16 = lane & 15, q4 = lane >> 4;
    // per-lane byte offset inside an 8-row piece (32-bit: the piece bases below are wave-uniform)
    const unsigned lane_off = (unsigned)l3 * (unsigned)ldp + (((lane & 7) ^ ((4 * (w4 & 1) + (l3 >> 1)) & 7)) << 4);
    // fragment reads: row = 16-row block base + r16, so the swizzle term ((row >> 1) & 7) = (r16 >> 1) is a lane constant and
    // every fragment address is one lane offset + a compile-time constant (hi chunk q4, lo chunk 4 + q4 = hi ^ 64 bytes)
    const unsigned frag_hi = (unsigned)r16 * HROW + (((unsigned)q4 ^ ((unsigned)r16 >> 1)) << 4);
    const unsigned frag_lo = frag_hi ^ 64u;
    // K runs over `npanels` panels of one row stride: panel p (p = 0 is the LATEST block column, processed first) lies pstride bytes
    // below panel p - 1; every panel has nk k-blocks except the earliest (processed last): nk_first blocks starting first_off bytes
    // into its rows.  Inside a panel the k-blocks are walked from the high end down (small terms first, see gemm_f32.hip).
    const int nstage = (npanels - 1) * nk + nk_first;

    // slot -> tile (wave-uniform arithmetic); false: the slot holds no tile of this launch
    // Work stealing: a workgroup pulls slots from its own XCD's counter (tiles whose panels its neighbours share through the L2)
    // until that runs dry, then from the next XCD's, and so on round the ring -- a launch of a few tile blocks (27 blocks of a
    // 4-tile-wide update over 8 XCDs: three get 64 tiles, five 48) otherwise ends when its fullest XCD does.
    int vx = xcd, tries = 0;
    // A launch covers up to four regions of C with the same panels (the pieces of one far update of the grouped Cholesky): the
    // tile blocks of region r follow those of region r - 1 in the enumeration.
    struct Tile {
        const char* pa;  // split rows of the tile's 256 C rows, k block 0 of the latest panel
        const char* pb;  // ... of its 256 C columns
        float* pc;       // C(row0, col0) of the tile
        int row0, col0;  // inside its region
        int m, n, shift; // the region's extent and diagonal shift
    };
    auto decode = [&](int slot, Tile& tl) -> bool {
        const int per = order_br * order_bc;
        int G = (slot / per) * 8 + vx;
        const int i = slot % per;
        if (G >= regs.nblk) return false;
        int ri = 0;
#pragma unroll
        for (int r = 1; r < 4; ++r)
            if (r < regs.count && G >= regs.r[r].blk0) ri = r;
        const int m = regs.r[ri].m, n = regs.r[ri].n, diag_shift = regs.r[ri].shift;
        G -= regs.r[ri].blk0;
        const int tiles_m = (m + HT - 1) / HT, tiles_n = (n + HT - 1) / HT;
        const int brows = (tiles_m + order_br - 1) / order_br;
        const int sh_t = (diag_shift + HT - 1) / HT;
        int gr = 0;
        for (; gr < brows; ++gr) {
            const int cnt = h3_block_cols(gr, order_br, order_bc, tiles_m, tiles_n, LOWER, sh_t);
            if (G < cnt) break;
            G -= cnt;
        }
        if (gr >= brows) return false;
        const int bi = gr * order_br + (i % order_br);
        const int bj = G * order_bc + (i / order_br);
        if (bi >= tiles_m || bj >= tiles_n) return false;
        if (LOWER && bj * HT > bi * HT + HT - 1 + diag_shift) return false;
        tl.pa = A + regs.r[ri].a_off + (int64_t)bi * HT * ldp;
        tl.pb = B + regs.r[ri].b_off + (int64_t)bj * HT * ldp;
        tl.pc = C + regs.r[ri].c_off + (int64_t)bi * HT * ldc + (int64_t)bj * HT;
        tl.row0 = bi * HT;
        tl.col0 = bj * HT;
        tl.m = m;
        tl.n = n;
        tl.shift = diag_shift;
        return true;
    };
    // load cursor: panel and k-block of the next stage to request (wave-uniform)
    int cur_p = 0, cur_kb = 0;
    auto cursor_reset = [&]() { cur_p = 0; cur_kb = (npanels == 1 ? nk_first : nk) - 1; };
    auto cursor_take = [&]() -> int64_t {  // byte offset of the cursor's k-block from the tile's row base; advances the cursor
        const int64_t off = (int64_t)cur_kb * 128 - (int64_t)cur_p * pstride + (cur_p == npanels - 1 ? first_off : 0);
        if (--cur_kb < 0) {
            ++cur_p;
            cur_kb = (cur_p == npanels - 1 ? nk_first : nk) - 1;
        }
        return off;
    };
    auto glds_b = [&](int t, int64_t koff, const char* tile_pb) {  // the wave's 4 B pieces of stage t
        const char* pb = tile_pb + koff;
        char* dst = smem + (t & 1) * HSTAGE + HT * HROW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int piece = group * 16 + e * 4 + w4;
            __builtin_amdgcn_global_load_lds((glb_void*)(pb + (int64_t)piece * 8 * ldp + lane_off), (lds_void*)(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto glds_a = [&](int t, int64_t koff, const char* tile_pa) {  // the wave's 4 A pieces of stage t: rows 0-63 of its group first (e = 0, 1), then rows 64-127
        const char* pa = tile_pa + koff;
        char* dst = smem + (t & 1) * HSTAGE;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int piece = group * 16 + e * 4 + w4;
            __builtin_amdgcn_global_load_lds((glb_void*)(pa + (int64_t)piece * 8 * ldp + lane_off), (lds_void*)(dst + piece * 1024), 16, 0, 0);
        }
    };
    // next valid tile of this workgroup, fetched synchronously (first tile, and after a slot without a tile)
    auto fetch_tile = [&](Tile& tl) -> bool {
        for (;;) {
            if (tid == 0) s_slot = atomicAdd(&counters[vx], 1);
            __syncthreads();
            const int slot = __builtin_amdgcn_readfirstlane(s_slot);
            __syncthreads();  // s_slot may be rewritten only after every wave has read it
            if (slot >= slots_per_xcd) {
                if (++tries >= 8) return false;
                vx = (vx + 1) & 7;
                continue;
            }
            if (decode(slot, tl)) return true;
        }
    };

#ifdef NNGP_TIMING_KNOBS
    const bool stamping = (ablate & 256) && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
    unsigned long long st_last = 0, st_k0 = 0, st_r0 = 0;
    if (stamping) { st_k0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#define H3_STAMP(i) do { if (stamping) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                         if ((i) >= 0) g_h3_stamps[(wave >> 2) * 8 + ((i) < 0 ? 0 : (i))] += now_ - st_last; st_last = now_; } } while (0)
#else
#define H3_STAMP(i) do { } while (0)
#endif
    Tile tl = {};
    bool have = fetch_tile(tl);
    if (have) {
        cursor_reset();
        const int64_t k0 = cursor_take();
        glds_b(0, k0, tl.pb);
        glds_a(0, k0, tl.pa);
    }
    while (have) {
        f32x4v acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};

        // stage 0 was requested before the previous tile's epilogue (or just above); every wave waits for its own pieces
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (group == 1) __builtin_amdgcn_s_barrier();  // stagger
        H3_STAMP(st_last == 0 ? -1 : 4);
        int next_slot = 0;
        for (int t = 0; t < nstage; ++t) {
            const char* sa_ = smem + (t & 1) * HSTAGE;
            const char* sb_ = sa_ + HT * HROW;
            const bool more = t + 1 < nstage && !(ablate & 1);
            const int64_t knext = more ? cursor_take() : 0;
            h8 bh[4], bl[4], ah[4], al[4];
            // ---- phase A: wave rows 0-63 ----
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = (w4 * 64 + j * 16) * HROW;
                bh[j] = *reinterpret_cast<const h8*>(sb_ + rb + frag_hi);
                bl[j] = *reinterpret_cast<const h8*>(sb_ + rb + frag_lo);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rb = (group * 128 + i * 16) * HROW;
                ah[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_hi);
                al[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_lo);
            }
            if (more) {
                glds_b(t + 1, knext, tl.pb);
                // retires the wave's last two A pieces of stage t (read in phase B).  Stage 0 has none outstanding, and a wait
                // here would also cover the previous tile's stores, which are still on their way
                if (t > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // last stage: take the next work slot now; its round trip runs under this stage's MFMAs (no counted wait follows)
                if (tid == 0) next_slot = atomicAdd(&counters[vx], 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (!(ablate & 2)) {
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
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[i]), "v"(bl[i]));
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase B: wave rows 64-127 ----
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rb = (group * 128 + 64 + i * 16) * HROW;
                ah[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_hi);
                al[i] = *reinterpret_cast<const h8*>(sa_ + rb + frag_lo);
            }
            if (more) {
                glds_a(t + 1, knext, tl.pa);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // retires the B pieces issued in phase A
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (!(ablate & 2)) {
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
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ah[i]), "v"(al[i]), "v"(bh[i]), "v"(bl[i]));
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // retires the A pieces of rows 0-63 of stage t+1
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (group == 0) __builtin_amdgcn_s_barrier();  // every wave executes the same number of barriers

        H3_STAMP(0);
        // ---- the next tile: slot (requested in the last stage), decode, first operand stage ----
        const Tile ct = tl;  // the tile whose accumulators are in registers
        if (tid == 0) s_slot = next_slot;
        __syncthreads();
        int slot = __builtin_amdgcn_readfirstlane(s_slot);
        __syncthreads();
        have = slot < slots_per_xcd && decode(slot, tl);
        if (!have) {  // this counter ran dry (move on to the next XCD's), or a slot of the padded enumeration held no tile
            if (slot >= slots_per_xcd) {
                ++tries;
                vx = (vx + 1) & 7;
            }
            have = tries < 8 && fetch_tile(tl);
        }

        H3_STAMP(1);
        // ---- epilogue of tile (ci, cj): acc[i][j][r] = C[row_base + 16 i + (lane & 15)][col_base + 16 j + 4 (lane >> 4) + r], one
        // 16-byte access per accumulator, lane address = wave-uniform band base + one 32-bit lane offset.  m, n are multiples of
        // 128, so a 16 x 64 band of the wave's sub-tile is inside or outside as a whole; tiles entirely inside (all but the
        // diagonal and edge tiles) take the branch-free path.
        const int m = ct.m, n = ct.n, diag_shift = ct.shift;
        const int row_base = ct.row0 + group * 128;  // inside the tile's region
        const int col_base = ct.col0 + w4 * 64;
        const unsigned voff = ((unsigned)r16 * (unsigned)ldc + 4u * (unsigned)q4) * 4u;  // bytes; 16 ldc < 2^30 (checked by the launcher)
        char* cw = reinterpret_cast<char*>(ct.pc + (int64_t)(group * 128) * ldc + w4 * 64);  // wave-uniform
        const int64_t band = (int64_t)16 * ldc * 4;                                       // bytes between 16-row bands
        const bool full = !(ablate & 8) && beta != 0.0f && col_base + 64 <= n && row_base + 128 <= m &&
                          !(LOWER && (col_base >> 7) > ((row_base + diag_shift) >> 7));  // the first band's 128-block decides: later bands lie lower
        if (full) {  // (beta == 0 -- the C-ABI tests only -- takes the band-by-band path: a second store-only branch here shared its
                     // alpha * acc products with this one, hipcc hoisted all 128 of them above the branch and spilled)
            // ALL old values are requested at once: a workgroup reads its C tile at the rate its requests in flight allow
            // (measured: 14 us per tile with 96 KB in flight, two to three dependent round trips), and the registers hold only
            // part of a sub-tile beside the accumulators -- so bands 3-7 come in by LDS-DMA into the wave's own 20 KB of the (now
            // idle) LDS, bands 0-2 into registers; the LDS part is then read back band by band.
            constexpr int RB = 3;  // bands through registers
            char* stage_c = smem + (wave * 20480);
            if (!(ablate & 16)) {
#pragma unroll
                for (int i = RB; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        __builtin_amdgcn_global_load_lds((glb_void*)(cw + i * band + voff + 64 * j),
                                                         (lds_void*)(stage_c + ((i - RB) * 4 + j) * 1024), 16, 0, (ablate & 64) ? 2 : 0);
            }
            f32x4v cold[RB][4];
            float ra[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = alpha * (row_alpha != nullptr ? row_alpha[row_base + 16 * i + r16] : 1.0f);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    cold[i][j] = (ablate & 16) ? f32x4v{0.0f, 0.0f, 0.0f, 0.0f}
                                                : (ablate & 64) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(cw + i * band + voff + 64 * j))
                                                                : *reinterpret_cast<const f32x4v*>(cw + i * band + voff + 64 * j);
            auto fma_b = [&](int i, int slot_) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(beta, cold[slot_][j][r], ra[i] * acc[i][j][r]);
                    // pin: the combined values exist HERE (hipcc otherwise sinks the arithmetic down to the stores)
                    asm volatile("" : "+v"(acc[i][j]) : : "memory");
                }
            };
#pragma unroll
            for (int i = 0; i < RB; ++i) fma_b(i, i);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the wave's own LDS-DMA pieces (older than the register loads) have landed
#pragma unroll
            for (int i = RB; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    cold[(i - RB) % RB][j] = (ablate & 16) ? f32x4v{0.0f, 0.0f, 0.0f, 0.0f}
                                                           : *reinterpret_cast<const f32x4v*>(stage_c + ((i - RB) * 4 + j) * 1024 + lane * 16);
                fma_b(i, (i - RB) % RB);
            }
            H3_STAMP(2);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ablate & 32)
                        asm volatile("" ::"v"(acc[i][j]));
                    else if (ablate & 64)
                        __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4v*>(cw + i * band + voff + 64 * j));
                    else
                        *reinterpret_cast<f32x4v*>(cw + i * band + voff + 64 * j) = acc[i][j];
                }
        } else {
#pragma unroll 1
            for (int i = 0; i < 8; ++i) {  // diagonal / edge tiles: band by band
                const int r0 = row_base + 16 * i;
                const bool live = col_base < n && r0 < m && !(LOWER && (col_base >> 7) > ((r0 + diag_shift) >> 7)) && !(ablate & 8);
                // acc[i] with a runtime i would go to scratch: select the band with wave-uniform compares instead
                f32x4v v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[0][j];
#pragma unroll
                    for (int ii = 1; ii < 8; ++ii)
                        if (i == ii) v[j] = acc[ii][j];
                }
                if (!live) continue;
                const float sa = alpha * (row_alpha != nullptr ? row_alpha[r0 + r16] : 1.0f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4v* pc = reinterpret_cast<f32x4v*>(cw + i * band + voff + 64 * j);
                    f32x4v o = v[j];
                    if (beta != 0.0f) {
                        const f32x4v c0 = *pc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = fmaf(beta, c0[r], sa * o[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] *= sa;
                    }
                    *pc = o;
                }
            }
        }
        H3_STAMP(3);
#ifdef NNGP_TIMING_KNOBS
        if (stamping) g_h3_stamps[(wave >> 2) * 8 + 5] += 1;
#endif
        if (ablate & 8) {  // keep the products alive in the no-C-traffic diagnostic build
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        }
        if (have) {  // the C staging areas overlap the operand stages: every wave's reads of them are retired first
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cursor_reset();
            const int64_t k0 = cursor_take();
            glds_b(0, k0, tl.pb);
            glds_a(0, k0, tl.pa);
        }
    }
#ifdef NNGP_TIMING_KNOBS
    if (stamping) {
        g_h3_stamps[(wave >> 2) * 8 + 6] += __builtin_amdgcn_s_memtime() - st_k0;
        g_h3_stamps[(wave >> 2) * 8 + 7] += __builtin_amdgcn_s_memrealtime() - st_r0;
    }
#endif
#undef H3_STAMP
    // The work counters reset themselves: the last workgroup to run out of work zeroes them for the next launch (see k_gemm_nt_h3)
    // (total_wgs: the workgroups of ALL launches that share this pass -- the update-stream grid and, in the Cholesky, a helper grid
    // that joins it on the panel stream once the diagonal-block chain has released the reserved compute units)
    if (tid == 0 && atomicAdd(&counters[8], 1) == total_wgs - 1)
        for (int i = 0; i < 9; ++i) __hip_atomic_store(&counters[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

int launch_split_rows_rowscale(const float* src, int64_t ld_src, int64_t rows, int64_t k, float* copy_dst, int64_t ld_copy,
                               char* out, int64_t out_ld, float* row_inv, hipStream_t s, int64_t pan_stride) {
    if (rows <= 0) return 0;
    const int64_t kp = k < 1024 ? k : 1024;  // columns per panel
    NNGP_REQUIRE(k > 0 && k <= 2048 && k % 32 == 0 && ld_src % 4 == 0 && ((uintptr_t)src & 15) == 0 && out_ld >= 4 * kp &&
                     out_ld % 16 == 0 && ((uintptr_t)out & 15) == 0 && row_inv != nullptr && (k <= 1024 || (pan_stride > 0 && pan_stride % 16 == 0)) &&
                     (copy_dst == nullptr || (ld_copy % 4 == 0 && ((uintptr_t)copy_dst & 15) == 0)),
                 "split_rows_rowscale: k must be a multiple of 32, at most 2048, operands 16-byte aligned");
    if (k <= 1024)
        hipLaunchKernelGGL((k_split_rows_rowscale<4>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, src, ld_src, rows, (int)k,
                           copy_dst, ld_copy, out, out_ld, pan_stride, row_inv);
    else
        hipLaunchKernelGGL((k_split_rows_rowscale<8>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, src, ld_src, rows, (int)k,
                           copy_dst, ld_copy, out, out_ld, pan_stride, row_inv);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_split_lower_t(const float* l, int64_t ld, int64_t n, int64_t bs, float scale, char* out, int64_t col_stride,
                         hipStream_t s) {
    if (n <= bs) return 0;
    NNGP_REQUIRE(n % 32 == 0 && bs % 32 == 0 && bs > 0 && ((uintptr_t)out & 15) == 0 && col_stride % 16 == 0,
                 "split_lower_t: sizes must be multiples of 32");
    hipLaunchKernelGGL(k_split_lower_t, dim3((unsigned)(n / 32), (unsigned)(n / 32)), dim3(256), 0, s, l, ld, n, bs, scale,
                       out, col_stride);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// a, b: split rows (row stride ldp bytes); rows of a / b up to the next multiple of 256 must be readable (their
// products are never stored).  m, n multiples of 128; k a multiple of 32.  counters: 16 device ints owned by the caller,
// zero before the first launch (the kernel leaves them zero again).  reserve_cus: compute units left free for other streams (the grid is one workgroup
// per remaining unit).  Several panels per launch (launch_gemm_nt_h3x): C = beta C + alpha sum_p A_p B_p^T over `npanels` panels of
// k columns each; a, b point at the LATEST panel (processed first) and panel p lies pstride bytes below panel p - 1 (the block
// columns of SplitWork::planes); the earliest panel contributes its columns [lead, k) only.  One pass over C for the whole sum.
constexpr int kH3DefaultForm = 2;  // 1: k_gemm_nt_h3 (32x32x16), 2: k_gemm_nt_h3v2 (16x16x32, balanced loads); debug key 5 = 40 + form

// helper: 0 = one launch; 1 = this launch will be joined by a helper grid of `reserve_cus` workgroups (launched with helper = 2 and
// the SAME other arguments on another stream, ordered after everything this pass depends on): both pull tiles from the same
// counters, and the counters reset when the workgroups of both have finished.
int launch_gemm_nt_h3r(float* c, int64_t ldc, const char* a, const char* b, int64_t ldp, int64_t pstride, int npanels, int64_t lead,
                       const H3RegionSpec* spec, int nreg, int64_t k, float alpha, float beta, bool lower_only, int* counters,
                       int reserve_cus, hipStream_t s, const float* row_alpha, int helper) {
    NNGP_REQUIRE(nreg >= 1 && nreg <= 4 && spec != nullptr, "gemm_nt_h3: 1 to 4 regions per launch");
    NNGP_REQUIRE(k > 0 && k % 32 == 0, "gemm_nt_h3: k must be a multiple of 32 (k=%lld)", (long long)k);
    NNGP_REQUIRE(ldp >= 4 * k && ldp % 16 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && counters != nullptr,
                 "gemm_nt_h3: operands must be 16-byte aligned");
    NNGP_REQUIRE(npanels >= 1 && lead >= 0 && lead < k && lead % 32 == 0 && (npanels == 1 || (pstride > 0 && pstride % 16 == 0)) &&
                     ldc < (1LL << 26),
                 "gemm_nt_h3: bad panel layout");
    int form = (NNGP_KNOB(5) >= 41 && NNGP_KNOB(5) <= 42) ? NNGP_KNOB(5) - 40 : kH3DefaultForm;
    const bool c16 = ldc % 4 == 0 && ((uintptr_t)c & 15) == 0;  // form 2 reads and writes C with 16-byte accesses
    if (npanels > 1 || lead > 0 || nreg > 1) {
        NNGP_REQUIRE(c16, "gemm_nt_h3: a multi-panel or multi-region product needs a 16-byte aligned C with ldc a multiple of 4");
        form = 2;
    } else if (!c16) {
        form = 1;
    }
    // tile-block shape (debug key 5 = 10 + variant for A/B timing); 4 x 4 measured best at N = 8k .. 32k
    static const int kOrders[][2] = {{4, 4}, {8, 4}, {4, 8}, {2, 8}, {8, 2}, {8, 8}, {2, 16}, {2, 4}};
    const int variant = (NNGP_KNOB(5) >= 10 && NNGP_KNOB(5) < 18) ? NNGP_KNOB(5) - 10 : 0;
    const int br = kOrders[variant][0], bc = kOrders[variant][1];
    H3Regions regs = {};
    int64_t nblk = 0;
    for (int r = 0; r < nreg; ++r) {
        const H3RegionSpec& sp = spec[r];
        if (sp.m <= 0 || sp.n <= 0) continue;
        NNGP_REQUIRE(sp.m % 128 == 0 && sp.n % 128 == 0 && sp.shift % 128 == 0 && sp.shift >= 0 && sp.row0 >= 0 && sp.col0 >= 0 &&
                         sp.m < 2147483647LL && sp.n < 2147483647LL && sp.col0 + sp.n <= ldc && sp.col0 % 4 == 0,
                     "gemm_nt_h3: m, n must be multiples of 128 (m=%lld n=%lld)", (long long)sp.m, (long long)sp.n);
        H3Region& rg = regs.r[regs.count++];
        rg.c_off = sp.row0 * ldc + sp.col0;
        rg.a_off = sp.row0 * ldp;
        rg.b_off = sp.col0 * ldp;
        rg.m = (int)sp.m;
        rg.n = (int)sp.n;
        rg.shift = lower_only ? (int)sp.shift : 0;
        rg.blk0 = (int)nblk;
        const int64_t tm = (sp.m + HT - 1) / HT, tn = (sp.n + HT - 1) / HT;
        for (int64_t gr = 0; gr < (tm + br - 1) / br; ++gr)
            nblk += h3_block_cols((int)gr, br, bc, (int)tm, (int)tn, lower_only, (int)((rg.shift + HT - 1) / HT));
    }
    if (regs.count == 0) return 0;
    regs.nblk = (int)nblk;
    const int64_t slots_per_xcd = ((nblk + 7) / 8) * br * bc;
    NNGP_REQUIRE(slots_per_xcd < 2147483647LL / 8, "gemm_nt_h3: too many tiles");
    static std::atomic<int> ncu_cached{0};  // compute units of the device (every MI355X of a node has the same count)
    int ncu = ncu_cached.load(std::memory_order_relaxed);
    if (ncu == 0) {
        int dev = 0, count = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess &&
               hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && count > 0) ? count : 256;
        ncu_cached.store(ncu, std::memory_order_relaxed);
    }
    int64_t grid = ncu - reserve_cus;
    if (grid > slots_per_xcd * 8) grid = slots_per_xcd * 8;
    grid = (grid / 8) * 8;  // the same number of workgroups on every XCD
    if (grid < 8) grid = 8;
    int total_wgs = (int)grid;
    if (helper != 0) {
        NNGP_REQUIRE(form == 2 && reserve_cus >= 8 && reserve_cus % 8 == 0, "gemm_nt_h3: a helper grid needs form 2 and reserved compute units");
        total_wgs = (int)grid + reserve_cus;
        if (helper == 2) grid = reserve_cus;
    }
    const int ablate = NNGP_KNOB(0) & 507;  // 1 no loads, 2 no MFMA, 8 no C traffic; form 2 also: 16 no C loads, 32 no C stores
    if (form == 2) {
#define NNGP_H3V2_LAUNCH(LOW, ABL)                                                                                                  \
    hipLaunchKernelGGL((k_gemm_nt_h3v2<LOW, ABL>), dim3((unsigned)grid), dim3(512), 0, s, c, ldc, a, b, ldp, pstride, npanels,          \
                       (int)((k - lead) / 32), (int)(lead * 4), regs, (int)(k / 32), alpha, beta, br, bc, counters,                  \
                       (int)slots_per_xcd, row_alpha, total_wgs)
#ifdef NNGP_TIMING_KNOBS
        if (lower_only) {
            switch (ablate) {
                case 1: NNGP_H3V2_LAUNCH(true, 1); break;
                case 2: NNGP_H3V2_LAUNCH(true, 2); break;
                case 8: NNGP_H3V2_LAUNCH(true, 8); break;
                case 9: NNGP_H3V2_LAUNCH(true, 9); break;
                case 16: NNGP_H3V2_LAUNCH(true, 16); break;
                case 32: NNGP_H3V2_LAUNCH(true, 32); break;
                case 64: NNGP_H3V2_LAUNCH(true, 64); break;
                case 256: NNGP_H3V2_LAUNCH(true, 256); break;
                default: NNGP_H3V2_LAUNCH(true, 0); break;
            }
        } else
#else
        (void)ablate;
        if (lower_only)
            NNGP_H3V2_LAUNCH(true, 0);
        else
#endif
            NNGP_H3V2_LAUNCH(false, 0);
#undef NNGP_H3V2_LAUNCH
    } else {
        // the first form: one region, one panel (kept for A/B timing and for a C that is not 16-byte aligned)
        const H3Region& rg = regs.r[0];
        const int64_t tm = (rg.m + HT - 1) / HT;
        if (lower_only)
            hipLaunchKernelGGL((k_gemm_nt_h3<true>), dim3((unsigned)grid), dim3(512), 0, s, c + rg.c_off, ldc, a + rg.a_off, b + rg.b_off, ldp,
                               rg.m, rg.n, (int)tm, (int)(k / 32), alpha, beta, rg.shift, br, bc, counters, (int)slots_per_xcd, ablate & 11,
                               row_alpha);
        else
            hipLaunchKernelGGL((k_gemm_nt_h3<false>), dim3((unsigned)grid), dim3(512), 0, s, c + rg.c_off, ldc, a + rg.a_off, b + rg.b_off, ldp,
                               rg.m, rg.n, (int)tm, (int)(k / 32), alpha, beta, 0, br, bc, counters, (int)slots_per_xcd, ablate & 11, row_alpha);
    }
    NNGP_HIP_CHECK(hipGetLastError());
#ifdef NNGP_TIMING_KNOBS
    if (ablate == 256 && form == 2 && lower_only) {  // timing study: segments of workgroup 0, cumulative over the launches so far
        unsigned long long h[16];
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_h3_stamps), sizeof(h)) == hipSuccess)
            for (int w = 0; w < 2; ++w)
                fprintf(stderr, "h3v2 stamps m=%d wave %d: tiles %llu  main %llu  slot %llu  c_in %llu  stores %llu  next0 %llu  | kernel cycles %llu  "
                                "realtime ticks %llu (clock %.3f GHz)\n", regs.r[0].m, 4 * w, h[w * 8 + 5], h[w * 8 + 0], h[w * 8 + 1], h[w * 8 + 2],
                        h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 6], h[w * 8 + 7], h[w * 8 + 7] ? 0.1 * (double)h[w * 8 + 6] / (double)h[w * 8 + 7] : 0.0);
    }
#endif
    return 0;
}

int launch_gemm_nt_h3x(float* c, int64_t ldc, const char* a, const char* b, int64_t ldp, int64_t pstride, int npanels, int64_t lead,
                       int64_t m, int64_t n, int64_t k, float alpha, float beta, bool lower_only, int64_t diag_shift, int* counters,
                       int reserve_cus, hipStream_t s, const float* row_alpha) {
    if (m <= 0 || n <= 0) return 0;
    NNGP_REQUIRE(ldc >= n && diag_shift % 128 == 0 && diag_shift >= 0, "gemm_nt_h3: bad ldc / diag_shift");
    const H3RegionSpec one = {0, 0, m, n, diag_shift};
    return launch_gemm_nt_h3r(c, ldc, a, b, ldp, pstride, npanels, lead, &one, 1, k, alpha, beta, lower_only, counters, reserve_cus, s,
                              row_alpha, 0);
}

int launch_gemm_nt_h3(float* c, int64_t ldc, const char* a, const char* b, int64_t ldp, int64_t m, int64_t n, int64_t k,
                      float alpha, float beta, bool lower_only, int64_t diag_shift, int* counters, int reserve_cus,
                      hipStream_t s, const float* row_alpha) {
    return launch_gemm_nt_h3x(c, ldc, a, b, ldp, 0, 1, 0, m, n, k, alpha, beta, lower_only, diag_shift, counters, reserve_cus, s,
                              row_alpha);
}

}  // namespace nngp
