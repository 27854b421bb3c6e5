// Float64-grade NT product on the 8-bit-integer matrix pipe of gfx950 ("sliced" operands, Ozaki-style):
//     C[M,N] = beta*Cin + alpha * A[M,K] * B[N,K]^T + gamma*G        (all float64 in HBM)
// Every row of A and of B is scaled by its own power of two sc (|x| / sc <= 126/256) and written as `ns` balanced base-256 digits
//     x / sc = sum_{i < ns} dig_i(x) 256^-(i+1),   dig_i in [-128, 127]   (fixed point of 8 ns bits, ns <= 7, rounded to nearest once),
// one int8 plane per digit.  The product of two planes is EXACT on v_mfma_i32_16x16x64_i8 (int32 accumulation, no rounding), so
//     A B^T = sc_a sc_b sum_{i,j} 256^-(i+j+2) (A_i B_j^T),
// and only the pairs with i + j <= cut are formed: the dropped ones are below 256^-(cut+3) 2^14 ~ 2^-42 (cut = 4) of the row scales
// per k.  Two grades are used (api.hip): 5 x 5 planes with cut 4 (15 products) and 7 x 7 with cut 6 (28: float64 grade proper).
// All pairs of one "diagonal" i + j = e share a weight and ONE int32 accumulator; K is cut into chunks of <= 16384 so that no
// accumulator can overflow (7 pairs x 16384 k x 2^14 < 2^31; <= 40960 when no diagonal has more than 3 pairs: round 4).  A second kernel adds the diagonals in float64 (Horner in 2^-8,
// fixed order: bitwise reproducible), applies the row scales and fuses beta*Cin and gamma*G.
//
// Why: the posterior's residual product R = K_td - Z (K + reg I) (reference: predict_fn(..., compute_cov=True),
// train.py:157-158) cancels to ~1e-4 of terms 1e3..1e5 larger, so nothing below float64 GRADE works -- but the float64 matrix
// pipe runs at 78.6 TF/s and the int8 pipe at ~5 POPS: 15 exact slice products (5 x 5 digits, cut 4: 40 bits below each row's
// maximum; scripts/ozaki8_check.py) cost a third of one float64 product.  M = 1024, N = K = 32768: 32-35 ms -> 12.5.
//
// Kernel k_gemm_nt_i8s: the tile machinery of k_gemm_nt_h3v2 (gemm_h3.hip) on the int8 pipe.  512 threads = 8 waves (2 x 4), tile
// 256 x 256, wave sub-tile 128 x 64 = 8 x 4 accumulators of v_mfma_i32_16x16x64_i8 with the operands swapped (a lane's 4 results =
// 4 consecutive columns of one C row -> 16-byte stores); one stage = 128 k = one 128-byte line per row (two MFMA k-steps), 64 KB
// per stage, two stages, filled by global_load_lds_dwordx4 with the source-side XOR swizzle; counted vmcnt waits; two wave groups
// one barrier apart.  A work item = (diagonal, K chunk, tile): its K loop runs over the diagonal's pairs x the chunk's k-blocks.
// Items are dealt largest diagonal first from per-XCD counters with cross-XCD stealing; a tile block of 4 x 8 tiles (one per
// compute unit of an XCD) shares its operand lines through that XCD's L2.  The int32 result goes straight to HBM (no C read).
#include <atomic>

#include "common.h"

namespace nngp {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int IT = 256;               // workgroup tile (rows and columns)
constexpr int IROW = 128;             // bytes per LDS row: 128 k of one plane
constexpr int ISTAGE = 2 * IT * IROW; // A rows + B rows: 64 KB
constexpr int kI8ChunkBlocks = 128;   // k-blocks (of 128) per K chunk: 16384 k (up to 7 pairs on a diagonal)
constexpr int kI8ChunkBlocks3 = 320;  // ... when no diagonal has more than 3 pairs: 3 x 40960 k x 2^14 = 2.01e9 < 2^31

// ---- row scales ----
// A power of two (integers stay exact, and the scaling itself never rounds): with max|row| = f 2^e, f in [0.5, 1), scale = 2^(e+1)
// if f <= 126/128 and 2^(e+2) otherwise, so that |x| / scale <= 126/256 and the top digit is at most 126 + 1 (carry): int8 holds it.
__device__ __forceinline__ double i8s_scale_of(double mx) {
    if (!(mx > 0.0 && mx < 1.0e300)) return 1.0;
    int e = 0;
    const double f = frexp(mx, &e);
    return ldexp(1.0, f <= 0.984375 ? e + 1 : e + 2);
}

// Symmetric positive semi-definite src (a kernel matrix): |K_ij| <= sqrt(K_ii K_jj), so row i is bounded by sqrt(K_ii max_j K_jj)
// without a pass over the matrix.  (The 1e-9 covers the rounding of a computed entry.)
__global__ __launch_bounds__(1024) void k_i8s_diag_bound_scale(const double* __restrict__ src, int64_t ld, int64_t n,
                                                               double* __restrict__ scale) {
    __shared__ double red[16];
    double mx = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) mx = fmax(mx, fabs(src[i * ld + i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmax(mx, red[i]);
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        scale[i] = i8s_scale_of(sqrt(fabs(src[i * ld + i]) * mx) * (1.0 + 1e-9));
    }
}

// ---- float64 rows -> NS int8 digit planes ----
// One workgroup per row: its maximum first (unless the caller supplies the scales), then 4096 columns at a time -- coalesced 16-byte
// loads, digits through LDS, 16-byte stores per plane (the second pass over the row is served by the L2 / MALL).
// planes[p][r][c] at planes + p * pstride + r * ldp + c; columns [cols, kp) are written as zero.
template <int NS>
__global__ __launch_bounds__(256) void k_i8s_slice_rows(const double* src, int64_t ld, int64_t cols, int64_t kp,
                                                        const double* __restrict__ scale_in, double* __restrict__ scale_out,
                                                        int8_t* __restrict__ planes, int64_t ldp, int64_t pstride, double* wb) {
    __shared__ __attribute__((aligned(16))) unsigned char dig[NS][4096];
    __shared__ double red[4];
    const int t = threadIdx.x;
    const int64_t r = blockIdx.x;
    const double* p = src + r * ld;
    double scale;
    if (scale_in != nullptr) {
        scale = scale_in[r];
    } else {
        double mx = 0.0;
        for (int64_t c = 2 * (int64_t)t; c < cols; c += 512) {
            if (c + 1 < cols) {
                const f64x2 v = *reinterpret_cast<const f64x2*>(p + c);
                mx = fmax(mx, fmax(fabs(v[0]), fabs(v[1])));
            } else {
                mx = fmax(mx, fabs(p[c]));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
        if ((t & 63) == 0) red[t >> 6] = mx;
        __syncthreads();
        scale = i8s_scale_of(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])));
        if (t == 0) scale_out[r] = scale;
    }
    const double inv = ldexp(1.0, 8 * NS) / scale;
    const double back = scale * ldexp(1.0, -8 * NS);  // scale is a power of two: x * back is the value the planes hold, exactly
    double* q = wb != nullptr ? wb + r * ld : nullptr;
    for (int64_t seg0 = 0; seg0 < kp; seg0 += 4096) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int lc = 2 * t + 512 * i;
            const int64_t c = seg0 + lc;
            f64x2 v = {0.0, 0.0};
            if (c + 1 < cols) {
                v = *reinterpret_cast<const f64x2*>(p + c);
            } else if (c < cols) {
                v[0] = p[c];
            }
            long long x0 = __double2ll_rn(v[0] * inv), x1 = __double2ll_rn(v[1] * inv);
            if (q != nullptr) {  // the caller goes on with the ROUNDED row: its planes are then the row itself, not an approximation
                if (c + 1 < cols) {
                    *reinterpret_cast<f64x2*>(q + c) = f64x2{(double)x0 * back, (double)x1 * back};
                } else if (c < cols) {
                    q[c] = (double)x0 * back;
                }
            }
#pragma unroll
            for (int s = NS - 1; s >= 1; --s) {  // least significant digit first; plane 0 = most significant
                const long long d0 = ((x0 + 128) & 255) - 128, d1 = ((x1 + 128) & 255) - 128;
                x0 = (x0 - d0) >> 8;
                x1 = (x1 - d1) >> 8;
                *reinterpret_cast<unsigned short*>(&dig[s][lc]) = (unsigned short)((d0 & 255) | ((d1 & 255) << 8));
            }
            // what is left is the top digit: at most 127 in magnitude (see the scale)
            *reinterpret_cast<unsigned short*>(&dig[0][lc]) = (unsigned short)((x0 & 255) | ((x1 & 255) << 8));
        }
        __syncthreads();
        const int64_t c16 = seg0 + 16 * t;
        if (c16 < kp) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
                *reinterpret_cast<i32x4*>(planes + s * pstride + r * ldp + c16) = *reinterpret_cast<const i32x4*>(&dig[s][16 * t]);
        }
        __syncthreads();
    }
}

// ---- the same planes of a bitwise SYMMETRIC matrix, every entry read once (round 4) ----
// One workgroup per 128 x 128 tile (I, J) of the lower triangle: the tile goes to LDS, is cut with its rows' scales into the planes'
// rows 128 I.., columns 128 J.. and -- off the diagonal -- transposed with its columns' scales into rows 128 J.., columns 128 I..:
// 8 N^2 / 2 bytes read instead of 8 N^2 for the same NS N^2 written, in 128-byte plane lines.  Same digits as k_i8s_slice_rows (the
// same rounding of the same product), so the planes are bit-identical when src is.  n: multiple of 128 (padding rows / columns hold zeros).
constexpr int kSymT = 128, kSymLd = kSymT + 1;
template <int NS>
__device__ __forceinline__ void i8s_digits16(const double* v, int stride, double inv, i32x4* dig) {
    unsigned char b[NS][16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        long long x = __double2ll_rn(v[c * stride] * inv);
#pragma unroll
        for (int s = NS - 1; s >= 1; --s) {
            const long long d = ((x + 128) & 255) - 128;
            x = (x - d) >> 8;
            b[s][c] = (unsigned char)(d & 255);
        }
        b[0][c] = (unsigned char)(x & 255);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        i32x4 w;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            w[q] = (int)((unsigned)b[s][4 * q] | ((unsigned)b[s][4 * q + 1] << 8) | ((unsigned)b[s][4 * q + 2] << 16) | ((unsigned)b[s][4 * q + 3] << 24));
        dig[s] = w;
    }
}

template <int NS>
__global__ __launch_bounds__(256) void k_i8s_slice_sym(const double* __restrict__ src, int64_t ld, const double* __restrict__ scale,
                                                       int8_t* __restrict__ planes, int64_t ldp, int64_t pstride) {
    __shared__ double tile[kSymT * kSymLd];  // 132 KB: one workgroup per compute unit
    const int t = threadIdx.x;
    // lower-triangle tile (I, J), I >= J, from the linear index
    const long long b = blockIdx.x;
    long long I = (long long)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= b) ++I;
    while (I * (I + 1) / 2 > b) --I;
    const long long J = b - I * (I + 1) / 2;
    const double* p = src + (I * kSymT) * ld + J * kSymT;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        const int row = 4 * i + (t >> 6), col = 2 * (t & 63);
        const f64x2 v = *reinterpret_cast<const f64x2*>(p + (int64_t)row * ld + col);
        tile[row * kSymLd + col] = v[0];
        tile[row * kSymLd + col + 1] = v[1];
    }
    __syncthreads();
    const double top = ldexp(1.0, 8 * NS);
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {  // rows of the tile with their own scales
        const int item = it * 256 + t, row = item >> 3, g = item & 7;
        const int64_t r = I * kSymT + row;
        i32x4 dig[NS];
        i8s_digits16<NS>(tile + row * kSymLd + 16 * g, 1, top / scale[r], dig);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<i32x4*>(planes + s * pstride + r * ldp + J * kSymT + 16 * g) = dig[s];
    }
    if (I == J) return;
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {  // its columns = rows 128 J.. of the matrix, with their scales
        const int item = it * 256 + t, col = item >> 3, g = item & 7;
        const int64_t r = J * kSymT + col;
        i32x4 dig[NS];
        i8s_digits16<NS>(tile + (16 * g) * kSymLd + col, kSymLd, top / scale[r], dig);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<i32x4*>(planes + s * pstride + r * ldp + I * kSymT + 16 * g) = dig[s];
    }
}

// ---- the exact int8 plane products ----
struct I8Tile {
    const char* pa;  // plane 0 of the tile's 256 A rows, k = 0
    const char* pb;  // ... of its 256 B rows
    int* pc;         // the item's int32 tile
    int row0, col0;
    int p0, p1;      // its diagonal's pairs [p0, p1)
    int kb0, nk;     // its K chunk: k-blocks [kb0, kb0 + nk)
};

// pa_lo/hi, pb_lo/hi: plane index of A / B of pair p in nibble p (32 pairs); dstart: first pair of diagonal dd in byte dd (8 diagonals
// + end).  P: [nchunk][ndiag][slab] int32, slab = rows * ldc.
__global__ __launch_bounds__(512) void k_gemm_nt_i8s(int* P, int64_t ldc, int64_t slab, const char* A, int64_t lda, int64_t sa,
                                                     const char* B, int64_t ldb, int64_t sb, int m, int n, int nkb, int kcb, int nchunk,
                                                     int ndiag, unsigned long long pa_lo, unsigned long long pa_hi,
                                                     unsigned long long pb_lo, unsigned long long pb_hi, unsigned long long dstart_lo,
                                                     unsigned dstart_hi, int order_br, int order_bc, int* counters, int slots_per_xcd,
                                                     int total_wgs) {
    // ONE LDS object (a second one makes hipcc drain the LDS-DMA queue before every fragment read: gemm_h3.hip)
    __shared__ __attribute__((aligned(1024))) char smem[2 * ISTAGE];
    int& s_slot = *reinterpret_cast<int*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7;
    const int group = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int w4 = __builtin_amdgcn_readfirstlane(wave & 3);
    const int l3 = lane >> 3;
    const int r16 = lane & 15, q4 = lane >> 4;
    const unsigned swz = (((lane & 7) ^ ((4 * (w4 & 1) + (l3 >> 1)) & 7)) << 4);
    const unsigned lane_off_a = (unsigned)l3 * (unsigned)lda + swz;
    const unsigned lane_off_b = (unsigned)l3 * (unsigned)ldb + swz;
    const unsigned frag_hi = (unsigned)r16 * IROW + (((unsigned)q4 ^ ((unsigned)r16 >> 1)) << 4);  // k bytes 16 q4 .. of the first 64
    const unsigned frag_lo = frag_hi ^ 64u;                                                        // the same of the second 64

    auto nib = [&](unsigned long long lo, unsigned long long hi, int p) -> int {
        return (int)(((p < 16 ? lo : hi) >> (4 * (p & 15))) & 15ull);
    };
    auto dstart = [&](int dd) -> int {
        return dd < 8 ? (int)((dstart_lo >> (8 * dd)) & 255ull) : (int)(dstart_hi & 255u);
    };
    const int tiles_m = (m + IT - 1) / IT, tiles_n = (n + IT - 1) / IT;
    const int bm = (tiles_m + order_br - 1) / order_br, bn = (tiles_n + order_bc - 1) / order_bc;
    const int nb = bm * bn;

    int vx = xcd, tries = 0;
    auto decode = [&](int slot, I8Tile& tl) -> bool {
        const int per = order_br * order_bc;
        const int G = (slot / per) * 8 + vx;
        const int i = slot % per;
        if (G >= ndiag * nchunk * nb) return false;
        const int dd = G / (nchunk * nb);
        const int rem = G - dd * (nchunk * nb);
        const int ch = rem / nb, blk = rem - ch * nb;
        const int gr = blk / bn, gc = blk - gr * bn;
        const int bi = gr * order_br + (i % order_br);
        const int bj = gc * order_bc + (i / order_br);
        if (bi >= tiles_m || bj >= tiles_n) return false;
        tl.pa = A + (int64_t)bi * IT * lda;
        tl.pb = B + (int64_t)bj * IT * ldb;
        tl.pc = P + ((int64_t)ch * ndiag + dd) * slab + (int64_t)bi * IT * ldc + (int64_t)bj * IT;
        tl.row0 = bi * IT;
        tl.col0 = bj * IT;
        tl.p0 = dstart(dd);
        tl.p1 = dstart(dd + 1);
        tl.kb0 = ch * kcb;
        tl.nk = (nkb - tl.kb0 < kcb) ? nkb - tl.kb0 : kcb;
        return true;
    };
    // load cursor (wave-uniform): pair and k-block of the next stage to request
    int cur_p = 0, cur_kb = 0;
    auto cursor_reset = [&](const I8Tile& tl) { cur_p = tl.p0; cur_kb = tl.nk - 1; };
    auto cursor_take = [&](const I8Tile& tl, int64_t& offa, int64_t& offb) {
        const int64_t kbyte = (int64_t)(tl.kb0 + cur_kb) * 128;
        offa = (int64_t)nib(pa_lo, pa_hi, cur_p) * sa + kbyte;
        offb = (int64_t)nib(pb_lo, pb_hi, cur_p) * sb + kbyte;
        if (--cur_kb < 0) {
            ++cur_p;
            cur_kb = tl.nk - 1;
        }
    };
    auto glds_b = [&](int t, int64_t koff, const char* tile_pb) {  // the wave's 4 B pieces (8 rows x 128 bytes each) of stage t
        const char* pb = tile_pb + koff;
        char* dst = smem + (t & 1) * ISTAGE + IT * IROW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int piece = group * 16 + e * 4 + w4;
            __builtin_amdgcn_global_load_lds((glb_void*)(pb + (int64_t)piece * 8 * ldb + lane_off_b), (lds_void*)(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto glds_a = [&](int t, int64_t koff, const char* tile_pa) {  // the wave's 4 A pieces: rows 0-63 of its group first (e = 0, 1)
        const char* pa = tile_pa + koff;
        char* dst = smem + (t & 1) * ISTAGE;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int piece = group * 16 + e * 4 + w4;
            __builtin_amdgcn_global_load_lds((glb_void*)(pa + (int64_t)piece * 8 * lda + lane_off_a), (lds_void*)(dst + piece * 1024), 16, 0, 0);
        }
    };
    auto fetch_tile = [&](I8Tile& tl) -> bool {
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

    I8Tile tl = {};
    bool have = fetch_tile(tl);
    bool stored = false;  // the wave issued 32 stores after the stage-0 loads now in flight
    if (have) {
        cursor_reset(tl);
        int64_t oa, ob;
        cursor_take(tl, oa, ob);
        glds_b(0, ob, tl.pb);
        glds_a(0, oa, tl.pa);
    }
    while (have) {
        i32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
        const int nstage = (tl.p1 - tl.p0) * tl.nk;

        // stage 0 was requested before the previous tile's stores: wait for the loads only (the queue retires in order)
        if (stored)
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (group == 1) __builtin_amdgcn_s_barrier();  // stagger
        int next_slot = 0;
        for (int t = 0; t < nstage; ++t) {
            const char* sa_ = smem + (t & 1) * ISTAGE;
            const char* sb_ = sa_ + IT * IROW;
            const bool more = t + 1 < nstage;
            int64_t na = 0, nbo = 0;
            if (more) cursor_take(tl, na, nbo);
            i32x4 bh[4], bl[4], ah[4], al[4];
            // ---- phase A: wave rows 0-63 ----
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rb = (w4 * 64 + j * 16) * IROW;
                bh[j] = *reinterpret_cast<const i32x4*>(sb_ + rb + frag_hi);
                bl[j] = *reinterpret_cast<const i32x4*>(sb_ + rb + frag_lo);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rb = (group * 128 + i * 16) * IROW;
                ah[i] = *reinterpret_cast<const i32x4*>(sa_ + rb + frag_hi);
                al[i] = *reinterpret_cast<const i32x4*>(sa_ + rb + frag_lo);
            }
            if (more) {
                glds_b(t + 1, nbo, tl.pb);
                // retires the wave's last two A pieces of stage t (read in phase B); at stage 0 nothing of this tile is outstanding
                // and a wait would cover the previous tile's stores
                if (t > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) next_slot = atomicAdd(&counters[vx], 1);  // its round trip runs under this stage's MFMAs
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bl[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bh[j], ah[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase B: wave rows 64-127 ----
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rb = (group * 128 + 64 + i * 16) * IROW;
                ah[i] = *reinterpret_cast<const i32x4*>(sa_ + rb + frag_hi);
                al[i] = *reinterpret_cast<const i32x4*>(sa_ + rb + frag_lo);
            }
            if (more) {
                glds_a(t + 1, na, tl.pa);
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
                for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bl[j], al[i], acc[4 + i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bh[j], ah[i], acc[4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // retires the A pieces of rows 0-63 of stage t+1
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (group == 0) __builtin_amdgcn_s_barrier();  // every wave executes the same number of barriers

        // ---- the next item: slot (requested in the last stage), decode, first operand stage; then this tile's stores ----
        const I8Tile ct = tl;
        if (tid == 0) s_slot = next_slot;
        __syncthreads();
        const int slot = __builtin_amdgcn_readfirstlane(s_slot);
        __syncthreads();
        have = slot < slots_per_xcd && decode(slot, tl);
        if (!have) {
            if (slot >= slots_per_xcd) {
                ++tries;
                vx = (vx + 1) & 7;
            }
            have = tries < 8 && fetch_tile(tl);
        }
        if (have) {  // every wave's fragment reads were retired before the barriers above; the slot word is dead from here on
            cursor_reset(tl);
            int64_t oa, ob;
            cursor_take(tl, oa, ob);
            glds_b(0, ob, tl.pb);
            glds_a(0, oa, tl.pa);
        }
        // acc[i][j][r] = C[row_base + 16 i + (lane & 15)][col_base + 16 j + 4 (lane >> 4) + r]; m, n are multiples of 128, the
        // wave's sub-tile (128 x 64 at multiples of 128 / 64) is inside or outside as a whole
        const int row_base = ct.row0 + group * 128, col_base = ct.col0 + w4 * 64;
        stored = row_base < m && col_base < n;
        if (stored) {
            const unsigned voff = ((unsigned)r16 * (unsigned)ldc + 4u * (unsigned)q4) * 4u;
            char* cw = reinterpret_cast<char*>(ct.pc + (int64_t)(group * 128) * ldc + w4 * 64);
            const int64_t band = (int64_t)16 * ldc * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<i32x4*>(cw + i * band + voff + 64 * j) = acc[i][j];
        }
    }
    // the last workgroup to run out of work zeroes the counters for the next launch
    if (tid == 0 && atomicAdd(&counters[8], 1) == total_wgs - 1)
        for (int i = 0; i < 9; ++i) __hip_atomic_store(&counters[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- float64 combination of the diagonals ----
// out = beta cin + alpha sa[r] sb[c] 2^-16 sum_dd 2^(-8 e_dd) sum_ch P[ch][dd] + gamma g,   e_dd = ndiag - 1 - dd (largest first).
// One thread per 4 columns; Horner from the least significant diagonal.
// FUSE (the level-1 variance, api.hip): the pass also writes the float32 copy of the result (right-hand side of the remainder solve) and,
// per row and block of 1024 columns, the partial sums of g.(cin + out), g.out, g.g and max |g| -- g = z, cin = k, out = the residual r:
// what two row-dot passes and a conversion pass over [rows, cols] delivered before.  k_i8s_rowstat_finish adds the blocks in order.
template <bool FUSE>
__global__ __launch_bounds__(256) void k_i8s_combine(double* __restrict__ out, int64_t ldo, const double* __restrict__ cin, int64_t ldcin,
                                                     double beta, double alpha, const double* __restrict__ g, int64_t ldg, double gamma,
                                                     const int* __restrict__ P, int64_t ldc, int64_t slab, int nchunk, int ndiag,
                                                     const double* __restrict__ sa, const double* __restrict__ sb, int64_t rows,
                                                     int64_t cols, float* __restrict__ out32, int64_t ld32, double* __restrict__ part) {
    const int64_t c4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t r = blockIdx.y;
    const bool live = c4 < cols && r < rows;
    if (!FUSE && !live) return;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, mx = 0.0;
    if (live) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int dd = 0; dd < ndiag; ++dd) {
            long long sum[4] = {0, 0, 0, 0};
            for (int ch = 0; ch < nchunk; ++ch) {
                const i32x4 v = *reinterpret_cast<const i32x4*>(P + ((int64_t)ch * ndiag + dd) * slab + r * ldc + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] += v[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = t[e] * 0.00390625 + (double)sum[e];
        }
        const double w = alpha * sa[r] * (1.0 / 65536.0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c4 + e >= cols) break;
            double v = w * sb[c4 + e] * t[e];
            const double cv = (beta != 0.0 || FUSE) ? cin[r * ldcin + c4 + e] : 0.0;
            const double gv = (gamma != 0.0 || FUSE) ? g[r * ldg + c4 + e] : 0.0;
            if (beta != 0.0) v += beta * cv;
            if (gamma != 0.0) v += gamma * gv;
            out[r * ldo + c4 + e] = v;
            if (FUSE) {
                out32[r * ld32 + c4 + e] = (float)v;
                s1 = fma(gv, cv + v, s1);
                s2 = fma(gv, v, s2);
                s3 = fma(gv, gv, s3);
                mx = fmax(mx, fabs(gv));
            }
        }
    }
    if (FUSE) {
        __shared__ double red[16];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_down(s1, off);
            s2 += __shfl_down(s2, off);
            s3 += __shfl_down(s3, off);
            mx = fmax(mx, __shfl_down(mx, off));
        }
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            red[w] = s1; red[4 + w] = s2; red[8 + w] = s3; red[12 + w] = mx;
        }
        __syncthreads();
        if (threadIdx.x < 4 && r < rows) {
            const int q = threadIdx.x;
            const double v = q < 3 ? (red[4 * q] + red[4 * q + 1]) + (red[4 * q + 2] + red[4 * q + 3])
                                   : fmax(fmax(red[12], red[13]), fmax(red[14], red[15]));
            part[(r * gridDim.x + blockIdx.x) * 4 + q] = v;
        }
    }
}

// var[r] = base[r] - sum_b part[r][b][0], delta[r] = sum_b part[r][b][1], zstat[2 r] = sum_b part[r][b][2], zstat[2 r + 1] = max_b part[r][b][3]
__global__ void k_i8s_rowstat_finish(const double* __restrict__ part, int nblk, int64_t rows, const double* __restrict__ base,
                                     double* __restrict__ var, double* __restrict__ delta, double* __restrict__ zstat) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, mx = 0.0;
    for (int b = 0; b < nblk; ++b) {
        const double* p = part + (r * nblk + b) * 4;
        s1 += p[0]; s2 += p[1]; s3 += p[2]; mx = fmax(mx, p[3]);
    }
    var[r] = base[r] - s1;
    delta[r] = s2;
    zstat[2 * r] = s3;
    zstat[2 * r + 1] = mx;
}


// out[0] = sum of the squares of n row scales (the planes' scales enter the error estimate below as their root mean square)
__global__ __launch_bounds__(1024) void k_i8s_scale_sqsum(const double* __restrict__ scale, int64_t n, double* __restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += scale[i] * scale[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = t;
    }
}

// What the dropped digit pairs may have cost a variance that was formed with z . r, r from the int8 product: the dropped pairs of one
// entry sum N products of two digits (rms 2^12.4) with random signs, so   |z . dr| ~ |z|_2 2^12.4 sqrt(pairs) 256^-(cut+3) sc_z rms_n(sc_k) sqrt(N)
// = |z|_2 coef sc_z sqrt(sum_n sc_k^2).  One workgroup per row; out[0] = max over the rows of that estimate / var[row] (as the bits of
// a non-negative double, by atomicMax: order-preserving, deterministic).
__global__ __launch_bounds__(256) void k_i8s_floor_ratio(const double* __restrict__ z, int64_t ld, int64_t cols, const double* __restrict__ var,
                                                         double coef, const double* __restrict__ sk2, unsigned long long* out) {
    __shared__ double red[8];
    const int64_t r = blockIdx.x;
    const double* p = z + r * ld;
    double mx = 0.0, s2 = 0.0;
    for (int64_t c = 2 * (int64_t)threadIdx.x; c + 1 < cols; c += 512) {
        const f64x2 v = *reinterpret_cast<const f64x2*>(p + c);
        mx = fmax(mx, fmax(fabs(v[0]), fabs(v[1])));
        s2 += v[0] * v[0] + v[1] * v[1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmax(mx, __shfl_xor(mx, off));
        s2 += __shfl_xor(s2, off);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mx; red[4 + (threadIdx.x >> 6)] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        s2 = (red[4] + red[5]) + (red[6] + red[7]);
        const double est = sqrt(s2) * coef * i8s_scale_of(mx) * sqrt(sk2[0]);
        const double v = var[r];
        const double ratio = v > 0.0 ? est / v : (est > 0.0 ? 1.0e300 : 0.0);
        atomicMax(out, (unsigned long long)__double_as_longlong(ratio));
    }
}

// the same from the rows' statistics (zstat[2 r] = |z_r|_2^2, zstat[2 r + 1] = max |z_r|): one thread per row
__global__ __launch_bounds__(256) void k_i8s_floor_ratio_rows(const double* __restrict__ zstat, int64_t rows, const double* __restrict__ var,
                                                              double coef, const double* __restrict__ sk2, unsigned long long* out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const double est = sqrt(zstat[2 * r]) * coef * i8s_scale_of(zstat[2 * r + 1]) * sqrt(sk2[0]);
    const double v = var[r];
    const double ratio = v > 0.0 ? est / v : (est > 0.0 ? 1.0e300 : 0.0);
    atomicMax(out, (unsigned long long)__double_as_longlong(ratio));
}

}  // namespace

int launch_i8s_diag_bound_scale(const double* src, int64_t ld, int64_t n, double* scale, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_i8s_diag_bound_scale, dim3(1), dim3(1024), 0, s, src, ld, n, scale);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// scale_in == nullptr: every row is scaled by its own maximum, written to scale_out; otherwise scale_in[r] >= max|row r| * 256/126, a power of two.
// writeback (may be src itself): receives every row ROUNDED to its ns digits -- the planes then hold that row exactly
int launch_i8s_slice_rows(const double* src, int64_t ld, int64_t rows, int64_t cols, int ns, const double* scale_in, double* scale_out,
                          int8_t* planes, int64_t ldp, int64_t pstride, hipStream_t s, double* writeback) {
    if (rows <= 0) return 0;
    const int64_t kp = round_up(cols, 128);
    NNGP_REQUIRE(ns >= 2 && ns <= 7 && ld % 2 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)planes & 15) == 0 && ldp >= kp &&
                     ldp % 16 == 0 && pstride % 16 == 0 && rows < 2147483647LL && (scale_in != nullptr || scale_out != nullptr),
                 "i8s_slice_rows: 2..7 planes, 16-byte aligned operands");
    const dim3 g((unsigned)rows);
    switch (ns) {
        case 2: hipLaunchKernelGGL(k_i8s_slice_rows<2>, g, dim3(256), 0, s, src, ld, cols, kp, scale_in, scale_out, planes, ldp, pstride, writeback); break;
        case 3: hipLaunchKernelGGL(k_i8s_slice_rows<3>, g, dim3(256), 0, s, src, ld, cols, kp, scale_in, scale_out, planes, ldp, pstride, writeback); break;
        case 4: hipLaunchKernelGGL(k_i8s_slice_rows<4>, g, dim3(256), 0, s, src, ld, cols, kp, scale_in, scale_out, planes, ldp, pstride, writeback); break;
        case 5: hipLaunchKernelGGL(k_i8s_slice_rows<5>, g, dim3(256), 0, s, src, ld, cols, kp, scale_in, scale_out, planes, ldp, pstride, writeback); break;
        case 6: hipLaunchKernelGGL(k_i8s_slice_rows<6>, g, dim3(256), 0, s, src, ld, cols, kp, scale_in, scale_out, planes, ldp, pstride, writeback); break;
        default: hipLaunchKernelGGL(k_i8s_slice_rows<7>, g, dim3(256), 0, s, src, ld, cols, kp, scale_in, scale_out, planes, ldp, pstride, writeback); break;
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// planes of a bitwise symmetric [n, n] matrix (n a multiple of 128; rows and columns beyond the data hold zeros) with the given row
// scales: every entry is read once (k_i8s_slice_sym).  5 or 7 planes.
int launch_i8s_slice_sym(const double* src, int64_t ld, int64_t n, int ns, const double* scale, int8_t* planes, int64_t ldp,
                         int64_t pstride, hipStream_t s) {
    if (n <= 0) return 0;
    NNGP_REQUIRE((ns == 5 || ns == 7) && n % kSymT == 0 && ld % 2 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)planes & 15) == 0 &&
                     ldp >= n && ldp % 16 == 0 && pstride % 16 == 0 && scale != nullptr,
                 "i8s_slice_sym: 5 or 7 planes of a matrix padded to 128");
    const int64_t tn = n / kSymT, tiles = tn * (tn + 1) / 2;
    NNGP_REQUIRE(tiles < 2147483647LL, "i8s_slice_sym: too many tiles");
    if (ns == 5) hipLaunchKernelGGL(k_i8s_slice_sym<5>, dim3((unsigned)tiles), dim3(256), 0, s, src, ld, scale, planes, ldp, pstride);
    else hipLaunchKernelGGL(k_i8s_slice_sym<7>, dim3((unsigned)tiles), dim3(256), 0, s, src, ld, scale, planes, ldp, pstride);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// the pairs (ia, ib) with ia + ib <= cut, grouped by diagonal e = ia + ib, LARGEST e first (most pairs first: the long items lead)
int launch_i8s_scale_sqsum(const double* scale, int64_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_i8s_scale_sqsum, dim3(1), dim3(1024), 0, s, scale, n, out);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// out: one device word, zeroed by the caller; receives the bits of max_r (estimate of |z_r . dr_r|) / var[r] (see k_i8s_floor_ratio)
int launch_i8s_floor_ratio(const double* z, int64_t ld, int64_t rows, int64_t cols, const double* var, const I8Plan& pl, int nsa, int nsb,
                           const double* sk2, unsigned long long* out, hipStream_t s) {
    if (rows <= 0) return 0;
    // pairs on the first dropped diagonal (+ 2 for the two operands' own truncation at their last plane)
    int cut = -1;
    for (int p = 0; p < pl.npairs; ++p) cut = pl.pa[p] + pl.pb[p] > cut ? pl.pa[p] + pl.pb[p] : cut;
    int dropped = 2;
    for (int ia = 0; ia < nsa; ++ia) dropped += (cut + 1 - ia >= 0 && cut + 1 - ia < nsb) ? 1 : 0;
    const double coef = 5404.7 * sqrt((double)dropped) * pow(256.0, -(double)(cut + 3));  // 2^12.4 = 5404.7
    hipLaunchKernelGGL(k_i8s_floor_ratio, dim3((unsigned)rows), dim3(256), 0, s, z, ld, cols, var, coef, sk2, out);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

static double i8s_floor_coef(const I8Plan& pl, int nsa, int nsb) {
    int cut = -1;
    for (int p = 0; p < pl.npairs; ++p) cut = pl.pa[p] + pl.pb[p] > cut ? pl.pa[p] + pl.pb[p] : cut;
    int dropped = 2;  // pairs on the first dropped diagonal (+ 2 for the two operands' own truncation at their last plane)
    for (int ia = 0; ia < nsa; ++ia) dropped += (cut + 1 - ia >= 0 && cut + 1 - ia < nsb) ? 1 : 0;
    return 5404.7 * sqrt((double)dropped) * pow(256.0, -(double)(cut + 3));  // 2^12.4 = 5404.7
}

int launch_i8s_floor_ratio_rows(const double* zstat, int64_t rows, const double* var, const I8Plan& pl, int nsa, int nsb, const double* sk2,
                                unsigned long long* out, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_i8s_floor_ratio_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, zstat, rows, var,
                       i8s_floor_coef(pl, nsa, nsb), sk2, out);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int i8s_plan(int nsa, int nsb, int cut, I8Plan* pl) {
    NNGP_REQUIRE(nsa >= 1 && nsa <= 7 && nsb >= 1 && nsb <= 7 && cut >= 0, "i8s: 1..7 planes per operand");
    if (cut > nsa + nsb - 2) cut = nsa + nsb - 2;
    *pl = I8Plan{};
    int np = 0;
    for (int e = cut; e >= 0; --e) {
        NNGP_REQUIRE(pl->ndiag < 8, "i8s: too many diagonals");
        pl->dstart[pl->ndiag++] = np;
        int cnt = 0;
        for (int ia = 0; ia < nsa; ++ia) {
            const int ib = e - ia;
            if (ib < 0 || ib >= nsb) continue;
            NNGP_REQUIRE(np < 32, "i8s: too many slice pairs");
            pl->pa[np] = ia;
            pl->pb[np] = ib;
            ++np;
            ++cnt;
        }
        NNGP_REQUIRE(cnt >= 1 && cnt <= 7, "i8s: a diagonal needs 1..7 pairs (int32 accumulation over 16384 k)");
    }
    pl->dstart[pl->ndiag] = np;
    pl->npairs = np;
    return 0;
}

static int chunk_blocks(const I8Plan* pl) {
    if (pl == nullptr || NNGP_KNOB(5) == 60) return kI8ChunkBlocks;  // key 5 = 60: timing experiment, short chunks always
    int most = 0;
    for (int d = 0; d < pl->ndiag; ++d) most = pl->dstart[d + 1] - pl->dstart[d] > most ? pl->dstart[d + 1] - pl->dstart[d] : most;
    return most <= 3 ? kI8ChunkBlocks3 : kI8ChunkBlocks;
}

int64_t i8s_chunks(int64_t k, const I8Plan* pl) {
    const int64_t nkb = (k + 127) / 128, cb = chunk_blocks(pl);
    return (nkb + cb - 1) / cb;
}

// a [m rows, ns planes], b [n rows]: int8 planes with row strides lda / ldb bytes (multiples of 128 bytes holding >= round_up(k, 128)
// columns, zero beyond k) and plane strides sa / sb; rows up to the next multiple of 256 must be readable.  partial: nchunk * ndiag
// slabs of `slab` int32 (slab >= m * ldc).  counters: 16 zeroed device ints (left zero again).
int launch_gemm_nt_i8s(int32_t* partial, int64_t ldc, int64_t slab, const int8_t* a, int64_t lda, int64_t sa, const int8_t* b,
                       int64_t ldb, int64_t sb, const I8Plan& pl, int64_t m, int64_t n, int64_t k, int* counters, int reserve_cus,
                       hipStream_t s) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    NNGP_REQUIRE(m % 128 == 0 && n % 128 == 0 && m < (1 << 30) && n < (1 << 30) && ldc >= n && ldc % 4 == 0 && slab >= m * ldc &&
                     ((uintptr_t)partial & 15) == 0,
                 "gemm_nt_i8s: m, n must be multiples of 128 (m=%lld n=%lld)", (long long)m, (long long)n);
    const int64_t nkb = (k + 127) / 128;
    NNGP_REQUIRE(lda >= nkb * 128 && ldb >= nkb * 128 && lda % 128 == 0 && ldb % 128 == 0 && sa % 16 == 0 && sb % 16 == 0 &&
                     ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && counters != nullptr && 8 * lda < (1LL << 31) &&
                     8 * ldb < (1LL << 31) && 16 * ldc * 4 < (1LL << 31),
                 "gemm_nt_i8s: operand planes must be 16-byte aligned with row strides that are multiples of 128");
    const int64_t nchunk = i8s_chunks(k, &pl);
    const int64_t kcb = (nkb + nchunk - 1) / nchunk;
    unsigned long long pa_lo = 0, pa_hi = 0, pb_lo = 0, pb_hi = 0, ds_lo = 0;
    unsigned ds_hi = 0;
    for (int p = 0; p < pl.npairs; ++p) {
        (p < 16 ? pa_lo : pa_hi) |= (unsigned long long)pl.pa[p] << (4 * (p & 15));
        (p < 16 ? pb_lo : pb_hi) |= (unsigned long long)pl.pb[p] << (4 * (p & 15));
    }
    for (int d = 0; d <= pl.ndiag; ++d) {
        if (d < 8) ds_lo |= (unsigned long long)pl.dstart[d] << (8 * d);
        else ds_hi = (unsigned)pl.dstart[d];
    }
    const int br = 4, bc = 8;  // one tile block = the 32 compute units of an XCD
    const int64_t tm = (m + IT - 1) / IT, tn = (n + IT - 1) / IT;
    const int64_t nb = ((tm + br - 1) / br) * ((tn + bc - 1) / bc);
    const int64_t nblk = nb * nchunk * pl.ndiag;
    const int64_t slots_per_xcd = ((nblk + 7) / 8) * br * bc;
    NNGP_REQUIRE(slots_per_xcd < 2147483647LL / 8 && nblk < 2147483647LL, "gemm_nt_i8s: too many tiles");
    static std::atomic<int> ncu_cached{0};
    int ncu = ncu_cached.load(std::memory_order_relaxed);
    if (ncu == 0) {
        int dev = 0, count = 0;
        ncu = (hipGetDevice(&dev) == hipSuccess &&
               hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && count > 0) ? count : 256;
        ncu_cached.store(ncu, std::memory_order_relaxed);
    }
    int64_t grid = ncu - reserve_cus;
    if (grid > slots_per_xcd * 8) grid = slots_per_xcd * 8;
    grid = (grid / 8) * 8;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(k_gemm_nt_i8s, dim3((unsigned)grid), dim3(512), 0, s, partial, ldc, slab, reinterpret_cast<const char*>(a), lda, sa,
                       reinterpret_cast<const char*>(b), ldb, sb, (int)m, (int)n, (int)nkb, (int)kcb, (int)nchunk, pl.ndiag, pa_lo, pa_hi,
                       pb_lo, pb_hi, ds_lo, ds_hi, br, bc, counters, (int)slots_per_xcd, (int)grid);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_i8s_combine(double* out, int64_t ldo, const double* cin, int64_t ldcin, double beta, double alpha, const double* g,
                       int64_t ldg, double gamma, const int32_t* partial, int64_t ldc, int64_t slab, int nchunk, int ndiag,
                       const double* sa, const double* sb, int64_t rows, int64_t cols, hipStream_t s, const I8Fuse* fuse) {
    if (rows <= 0 || cols <= 0) return 0;
    NNGP_REQUIRE((beta == 0.0 || cin != nullptr) && (gamma == 0.0 || g != nullptr) && ldc % 4 == 0, "i8s_combine: NULL input");
    NNGP_REQUIRE(fuse == nullptr || (cin != nullptr && g != nullptr && fuse->out32 != nullptr && fuse->part != nullptr && fuse->ld32 >= cols),
                 "i8s_combine: the fused row statistics need both addends");
    const unsigned nbx = (unsigned)i8s_col_blocks(cols);
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        if (fuse != nullptr)
            hipLaunchKernelGGL(k_i8s_combine<true>, dim3(nbx, (unsigned)nr), dim3(256), 0, s, out + r0 * ldo, ldo, cin + r0 * ldcin, ldcin,
                               beta, alpha, g + r0 * ldg, ldg, gamma, partial + r0 * ldc, ldc, slab, nchunk, ndiag, sa + r0, sb, nr, cols,
                               fuse->out32 + r0 * fuse->ld32, fuse->ld32, fuse->part + r0 * nbx * 4);
        else
            hipLaunchKernelGGL(k_i8s_combine<false>, dim3(nbx, (unsigned)nr), dim3(256), 0, s, out + r0 * ldo, ldo,
                               cin ? cin + r0 * ldcin : nullptr, ldcin, beta, alpha, g ? g + r0 * ldg : nullptr, ldg, gamma,
                               partial + r0 * ldc, ldc, slab, nchunk, ndiag, sa + r0, sb, nr, cols, (float*)nullptr, (int64_t)0,
                               (double*)nullptr);
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int64_t i8s_col_blocks(int64_t cols) { return (cols + 1023) / 1024; }

int launch_i8s_rowstat_finish(const double* part, int64_t cols, int64_t rows, const double* base, double* var, double* delta,
                              double* zstat, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(k_i8s_rowstat_finish, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, part, (int)i8s_col_blocks(cols), rows,
                       base, var, delta, zstat);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nngp
