// Blocked lower Cholesky in float32 for gfx950 -- replaces scipy/LAPACK cho_factor inside
// nt.predict.gradient_descent_mse_ensemble (reference train.py:171-172).
//
// Structure: a host-side recursion (potrf -> trsm -> syrk -> potrf) bottoms out at 128x128 leaves, so
// every flop above the leaves runs in the MFMA GEMM of gemm_f32.hip with a deep K.  The leaf kernel
// factors one 128x128 diagonal block in LDS *and* inverts the factor; triangular solves against a
// diagonal block then become an in-place GEMM with the inverse (no serial substitution anywhere above
// the leaf).
//
// Leaf (one 256-thread workgroup): the block is treated as 4x4 sub-blocks of 32x32.  Per sub-block
// column: wave 0 factors the 32x32 diagonal sub-block and inverts it entirely in registers (lane = row,
// v_readlane broadcasts, no barriers), then the sub-blocks below are multiplied by the inverse and the
// trailing sub-blocks updated with v_mfma_f32_32x32x2_f32.  The 128x128 inverse is assembled from the
// 32x32 inverses by block forward substitution on the matrix cores.  LDS row stride is 129 floats, which
// makes row-wise and column-wise 4-byte fragment reads conflict free.
#include <atomic>
#include <cstdio>
#include <new>

#include "common.h"

namespace nngp {

#ifdef NNGP_TIMING_KNOBS
std::atomic<int> g_knobs[16];  // timing experiments only (libnngp_hip_knobs.so: nngp_debug_set); zero-initialised
#endif

namespace {

constexpr int LS = 33;            // LDS row stride of one 32x32 sub-block (odd: row- and column-wise reads conflict free)
constexpr int BLK = 32 * LS;      // floats per packed sub-block
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Only the 10 sub-blocks on or below the diagonal are kept in LDS (L and X = L^-1 are lower triangular):
// 2 x 10 x 32 x 33 floats = 84.5 KB, so the leaf co-resides with a 64 KB GEMM workgroup on the same CU.
__device__ __forceinline__ constexpr int blk(int ib, int jb) { return ib * (ib + 1) / 2 + jb; }

// acc += sign * A_blk(32x32) * op(B_blk);  NN: op(B) = B,  NT: op(B) = B^T.  Blocks live in LDS, stride LS.
template <bool NN>
__device__ __forceinline__ f32x16 blk_mma(const float* ab, const float* bb, f32x16 acc, float sign, int lane) {
    const int r = lane & 31, h = lane >> 5;
    // all 32 operand reads first, then the 16 MFMAs back to back (interleaved, each MFMA waited for its own LDS round trip)
    float av[16], bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = 2 * s + h;
        av[s] = ab[r * LS + k];
        bv[s] = NN ? bb[k * LS + r] : bb[r * LS + k];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sign * av[s], bv[s], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ f32x16 blk_load(const float* p, int lane) {
    f32x16 v;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = p[((r & 3) + 8 * (r >> 2) + 4 * h) * LS + c];
    return v;
}

__device__ __forceinline__ void blk_store(float* p, f32x16 v, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2) + 4 * h) * LS + c] = v[r];
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.0f;
    return v;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_potrf_leaf(float* A, int64_t ld, float* dinv, int32_t* clamped,
                                                       float pivot_floor, int dbg) {
    __shared__ float Lb[10 * BLK];
    __shared__ float Xb[10 * BLK];
    __shared__ float Tb[3 * BLK];   // scratch of the inverse assembly (one block per active wave)
    __shared__ float colbuf[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = tid >> 3, lc = (tid & 7) * 4;  // this thread's (row, first column) inside a 32x32 sub-block

    {
        // one 16-byte load per thread per lower sub-block, all 10 issued back to back
        float4 v[10];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb <= ib; ++jb)
                v[blk(ib, jb)] = *reinterpret_cast<const float4*>(A + (int64_t)(ib * 32 + lr) * ld + jb * 32 + lc);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb <= ib; ++jb) {
                const float4 t = v[blk(ib, jb)];
                float* lp = Lb + blk(ib, jb) * BLK + lr * LS + lc;
                float* xp = Xb + blk(ib, jb) * BLK + lr * LS + lc;
                const bool diag = (ib == jb);  // strictly-upper entries of a diagonal sub-block are not part of A
                lp[0] = (!diag || lc + 0 <= lr) ? t.x : 0.0f;
                lp[1] = (!diag || lc + 1 <= lr) ? t.y : 0.0f;
                lp[2] = (!diag || lc + 2 <= lr) ? t.z : 0.0f;
                lp[3] = (!diag || lc + 3 <= lr) ? t.w : 0.0f;
                xp[0] = xp[1] = xp[2] = xp[3] = 0.0f;
            }
    }
    __syncthreads();

    int nclamp = 0;
#pragma unroll 1
    for (int jb = 0; jb < 4; ++jb) {
        float* Djj = Lb + blk(jb, jb) * BLK;
        float* Xjj = Xb + blk(jb, jb) * BLK;
        if (wave == 0 && !(dbg & 1)) {
            // One wave factors and inverts the 32x32 diagonal sub-block.  Lane i owns row i in registers; the
            // values every lane needs (pivot, column j) travel through a 32-float LDS line read back as
            // broadcasts -- an LDS round trip per column instead of ~500 v_readlane + hazard nops.
            const int i = lane & 31;
            float a[32], x[32], pinv[32];  // pinv[j] = 1 / L[j][j] (wave-uniform), reused by the inversion
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k] = Djj[i * LS + k];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float* col = colbuf + (j & 1) * 32;
                if (VARIANT == 0 || lane < 32) col[i] = a[j];  // current column j (rows >= j are valid)
                float d = col[j];
                asm volatile("" : "+v"(d));  // keep wave-uniform values in VGPRs (no SGPR spills / readlane traffic)
                if (!(d > pivot_floor)) {
                    d = pivot_floor > 0.0f ? pivot_floor : 1.0e-30f;
                    ++nclamp;
                }
                float inv = __builtin_amdgcn_rsqf(d);  // v_rsq_f32, ~1 ulp: ample for a preconditioner
                asm volatile("" : "+v"(inv));
                pinv[j] = inv;
                const float lij = a[j] * inv;                // L[i][j] for i > j
                a[j] = (i == j) ? d * inv : lij;
                const float t = -lij * inv;                  // a[i][k] -= L[i][j] L[k][j] = a[i][k] + t * col[k]
#pragma unroll
                for (int k = j + 1; k < 32; ++k) {
                    float ck = col[k];
                    if (VARIANT == 0) asm volatile("" : "+v"(ck));  // stay in VGPRs: scalarising 496 broadcasts spills SGPRs
                    a[k] = fmaf(t, ck, a[k]);
                }
            }
            if (lane < 32) {
#pragma unroll
                for (int k = 0; k < 32; ++k) Djj[i * LS + k] = (k <= i) ? a[k] : 0.0f;
            }
            // inverse: lane c owns column c of X = L^-1;  X[ii][c] = (delta - sum_{k<ii} L[ii][k] X[k][c]) / L[ii][ii]
            // (measured and dropped in round 2: computing row j of X inside step j of the factorisation loop above, so that the two
            // 32-step dependent passes become one -- 60 us per leaf instead of 50: the merged loop body schedules worse)
#pragma unroll
            for (int ii = 0; ii < 32; ++ii) {
                float s0 = 0.0f, s1 = 0.0f;  // two chains: the sum is latency-, not throughput-bound
#pragma unroll
                for (int k = 0; k + 1 < ii; k += 2) {
                    s0 = fmaf(Djj[ii * LS + k], x[k], s0);  // broadcast LDS reads
                    s1 = fmaf(Djj[ii * LS + k + 1], x[k + 1], s1);
                }
                if (ii & 1) s0 = fmaf(Djj[ii * LS + ii - 1], x[ii - 1], s0);
                x[ii] = (((i == ii) ? 1.0f : 0.0f) - (s0 + s1)) * pinv[ii];
            }
            if (lane < 32) {
#pragma unroll
                for (int k = 0; k < 32; ++k) Xjj[k * LS + i] = x[k];  // X[k][c = i]
            }
        }
        __syncthreads();
        // ---- sub-blocks below the diagonal: A[ib][jb] <- A[ib][jb] * Dinv^T ----
        if (!(dbg & 4)) {
            const int ib = jb + 1 + wave;
            if (ib < 4) {
                float* Aij = Lb + blk(ib, jb) * BLK;
                f32x16 acc = blk_mma<false>(Aij, Xjj, zero16(), 1.0f, lane);
                blk_store(Aij, acc, lane);
            }
        }
        __syncthreads();
        // ---- trailing sub-blocks: A[ib][kb] -= A[ib][jb] * A[kb][jb]^T, jb < kb <= ib ----
        if (!(dbg & 4)) {
            int cnt = 0;
            for (int ib = jb + 1; ib < 4; ++ib)
                for (int kb = jb + 1; kb <= ib; ++kb, ++cnt) {
                    if ((cnt & 3) != wave) continue;
                    float* Cik = Lb + blk(ib, kb) * BLK;
                    f32x16 acc = blk_load(Cik, lane);
                    acc = blk_mma<false>(Lb + blk(ib, jb) * BLK, Lb + blk(kb, jb) * BLK, acc, -1.0f, lane);
                    blk_store(Cik, acc, lane);
                }
        }
        __syncthreads();
    }

    // ---- assemble the 128x128 inverse: X[ib][jb] = -Dinv_ii * sum_{k=jb}^{ib-1} L[ib][k] X[k][jb] ----
#pragma unroll 1
    for (int dist = 1; dist < ((dbg & 2) ? 1 : 4); ++dist) {
        const int ib = dist + wave, jb = wave;  // wave w owns block (dist + w, w)
        const bool active = ib < 4;
        float* scratch = Tb + (wave < 3 ? wave : 0) * BLK;
        if (active) {
            f32x16 t = zero16();
            for (int k = jb; k < ib; ++k)
                t = blk_mma<true>(Lb + blk(ib, k) * BLK, Xb + blk(k, jb) * BLK, t, 1.0f, lane);
            blk_store(scratch, t, lane);
        }
        __syncthreads();
        if (active) {
            f32x16 xr = blk_mma<true>(Xb + blk(ib, ib) * BLK, scratch, zero16(), -1.0f, lane);
            blk_store(Xb + blk(ib, jb) * BLK, xr, lane);
        }
        __syncthreads();
    }

    // ---- write back: L into the lower triangle of A, X (with its zero upper blocks) into dinv ----
#pragma unroll 1
    for (int b = 0; b < 16; ++b) {
        const int ib = b >> 2, jb = b & 3;
        float4 xo = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jb <= ib) {
            const float* lp = Lb + blk(ib, jb) * BLK + lr * LS + lc;
            const float* xp = Xb + blk(ib, jb) * BLK + lr * LS + lc;
            xo = make_float4(xp[0], xp[1], xp[2], xp[3]);
            float* ap = A + (int64_t)(ib * 32 + lr) * ld + jb * 32 + lc;
            if (ib != jb || lc + 3 <= lr) {
                *reinterpret_cast<float4*>(ap) = make_float4(lp[0], lp[1], lp[2], lp[3]);
            } else {  // the 4-wide group straddles or lies above the diagonal: keep the caller's upper entries
                for (int k = 0; k < 4; ++k)
                    if (lc + k <= lr) ap[k] = lp[k];
            }
        }
        *reinterpret_cast<float4*>(dinv + (ib * 32 + lr) * 128 + jb * 32 + lc) = xo;
    }
    if (wave == 0 && lane == 0 && nclamp > 0 && clamped != nullptr) atomicAdd(clamped, nclamp);
}

// Panel form of the leaf (the product path; k_potrf_leaf above is kept for A/B timing, debug key 3 = 2).  In k_potrf_leaf
// three of the four waves idle while wave 0 factors AND inverts each 32x32 diagonal sub-block (two dependent 32-step passes,
// ~8 of the ~11 us a sub-block column takes), and the sub-blocks below wait for that inverse.  Here the wave that owns block
// row w eliminates its 32 rows against the diagonal sub-block's columns AS the leader (wave jb) produces them: the leader
// publishes column j of the diagonal sub-block in an LDS ring and bumps a step counter, the followers poll the counter and
// apply the same rank-1 step to their own rows (LDS operations of one wave execute in order, so the counter is visible after
// the column; no barrier inside the 32 steps).  The sub-blocks below are therefore solved by substitution when the leader
// finishes -- no inverse on the critical path -- and the four diagonal inverses are computed at the end, one per wave, in
// parallel.  Critical path per sub-block column: one 32-step pass + the MFMA updates.
#ifdef NNGP_TIMING_KNOBS
__device__ unsigned long long g_leaf_stamps[4 * 16];  // per wave: cycles from kernel start to the end of each phase (debug key 7 = 8)
#define LEAF_STAMP(idx) do { if (lane == 0) g_leaf_stamps[wave * 16 + (idx)] = __builtin_amdgcn_s_memtime() - t_start_; } while (0)
#else
#define LEAF_STAMP(idx) do { } while (0)
#endif
template <int FORM>
__global__ __launch_bounds__(256) void k_potrf_leaf_panel(float* A, int64_t ld, float* dinv, int32_t* clamped,
                                                             float pivot_floor) {
    __shared__ float Lb[10 * BLK];
    __shared__ float Xb[10 * BLK];
    __shared__ float Tb[6 * BLK];   // scratch of the inverse assembly
    // FORM 1: word (j, lane) = the A operand of step j for the followers (low half; see below) tagged with the step number (high half)
    // FORM 0: as floats, line j/2: columns j and j+1 of the current diagonal sub-block before elimination step j (layout: see rd)
    __shared__ unsigned long long ring64[32 * 64];
    __shared__ float ring_inv[32];   // FORM 1: 1 / L[j][j] of the sub-block column being eliminated
    __shared__ int step_flag;        // FORM 0: column pairs published so far: 16 jb + j/2 + 1
    float* ring = reinterpret_cast<float*>(ring64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = tid >> 3, lc = (tid & 7) * 4;  // this thread's (row, first column) inside a 32x32 sub-block
    if (tid == 0) step_flag = 0;
    if (FORM == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) ring64[k * 256 + tid] = 0ull;  // no stale tag may look like a published step
    }
#ifdef NNGP_TIMING_KNOBS
    const unsigned long long t_start_ = __builtin_amdgcn_s_memtime();
#endif
    {
        float4 v[10];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb <= ib; ++jb)
                v[blk(ib, jb)] = *reinterpret_cast<const float4*>(A + (int64_t)(ib * 32 + lr) * ld + jb * 32 + lc);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb <= ib; ++jb) {
                const float4 t = v[blk(ib, jb)];
                float* lp = Lb + blk(ib, jb) * BLK + lr * LS + lc;
                float* xp = Xb + blk(ib, jb) * BLK + lr * LS + lc;
                const bool diag = (ib == jb);  // strictly-upper entries of a diagonal sub-block are not part of A
                lp[0] = (!diag || lc + 0 <= lr) ? t.x : 0.0f;
                lp[1] = (!diag || lc + 1 <= lr) ? t.y : 0.0f;
                lp[2] = (!diag || lc + 2 <= lr) ? t.z : 0.0f;
                lp[3] = (!diag || lc + 3 <= lr) ? t.w : 0.0f;
                xp[0] = xp[1] = xp[2] = xp[3] = 0.0f;
            }
    }
    __syncthreads();
    LEAF_STAMP(0);

    int nclamp = 0;
    const int i = lane & 31, h = lane >> 5;
    // Lane (i, h) owns the entries of row i in the columns of parity h: b[m] = entry (i, 2m + h) of block (wave, jb) -- the two
    // half-waves split the rank-1 updates between them instead of duplicating them.
    float b[16], pinv[32];  // pinv[j] = 1 / L[j][j] of the sub-block this wave led (FORM 0)
    float invl = 0.0f;      // FORM 1: lane c (both halves): 1 / L[c][c] of the sub-block this wave led
    const float* rd = ring + h * 32;  // a column pair is stored by row parity: entry (k, A|B) at ((k & 1) * 16 + (k >> 1)) * 2 + (A: 0, B: 1)
#pragma unroll 1
    for (int jb = 0; jb < 4; ++jb) {
        f32x16 acc;
        if (FORM == 1 && wave >= jb) {
            // Round 5: the sub-blocks stay in the MFMA accumulator layout for the 32 steps and every step is ONE rank-1
            // v_mfma_f32_32x32x2_f32 per wave -- no LDS round trip and no per-element FMAs in the leader's chain (readlane of the
            // pivot -> v_rsq_f32 -> one multiply -> MFMA: 54 ns a step against 75 with a separate flag behind a release,
            // scripts/micro/mfma_chain.hip).  The leader holds the diagonal sub-block S as a full symmetric matrix: row j of S IS
            // column j, already one entry per lane (lanes of half hj), which is the layout of both MFMA operands; the finished
            // columns are frozen by zeroing the operand there, so at the end entry (m, c), c < m, holds L[m][c] / inv_c.  A follower
            // holds its sub-block TRANSPOSED and UNSCALED (accumulator row j = column j of the block before its division by
            // L[j][j]): that row is the B operand as it stands, the A operand is the leader's -L[.][j] / L[j][j], published as
            // ONE 8-byte word per lane {value, step number} -- value and tag arrive together, so neither side orders anything: no
            // flag, no release wait in the leader, one LDS round trip per step in the follower.
            const bool leader = (wave == jb);
            float* Bw = Lb + blk(wave, jb) * BLK;
            if (leader) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // the full symmetric block from its lower triangle
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    acc[r] = Bw[(i <= row) ? row * LS + i : i * LS + row];
                }
                const float floor_eff = pivot_floor > 0.0f ? pivot_floor : 1.0e-30f;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const int rj = (j & 3) + 4 * (j >> 3), hj = (j >> 2) & 1;  // accumulator register / lane half that hold row j
                    const float xr = acc[rj];
                    const float xm = (h == hj && i > j) ? xr : 0.0f;  // rows below the pivot (beside the rsq, not behind it)
                    const float d_raw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xr), 32 * hj + j));
                    const float inv = __builtin_amdgcn_rsqf(fmaxf(d_raw, floor_eff));  // v_rsq_f32, ~1 ulp: ample for a preconditioner
                    const float x = xm * inv;  // L[i][j]
                    const float xn = xm * -inv;
                    // (the publish goes in FRONT of the MFMA: behind it the LDS store waits for its data while the MFMA streams the
                    // accumulator through the register file -- 80 cycles a step against ~12)
                    if (jb < 3) {  // (the last sub-block column has no followers)
                        const float xq = xn * inv;
                        const unsigned long long word = ((unsigned long long)(unsigned)(jb * 32 + j + 1) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, xq);
                        __hip_atomic_store(&ring64[j * 64 + lane], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xn, x, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // Nothing but the chain was kept per step: entry (c, c) of the accumulator was frozen at step c and IS the pivot of
                // column c as the chain read it, so every lane redoes its own column's clamp and v_rsq_f32 (same instructions on the
                // same value: the same bits) -- 1 / L[c][c] for the scaling below, for the followers and for this wave's inverse.
                float dsel = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dsel = ((r & 3) + 8 * (r >> 2) + 4 * h == i) ? acc[r] : dsel;
                const float dother = __shfl_xor(dsel, 32);
                const float d_raw = (((i >> 2) & 1) == h) ? dsel : dother;  // the half whose rows include row i holds it
                nclamp += __builtin_popcount((unsigned)__ballot(!(d_raw > pivot_floor)));  // lanes 0 .. 31: one per column
                const float dcl = fmaxf(d_raw, floor_eff);  // (a NaN pivot is clamped as well)
                invl = __builtin_amdgcn_rsqf(dcl);
                const float diagl = dcl * invl;
                if (jb < 3 && lane < 32) ring_inv[lane] = invl;  // (the followers read it behind the barrier below)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    Bw[row * LS + i] = (i < row) ? acc[r] * invl : (i == row ? diagl : 0.0f);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = Bw[i * LS + (r & 3) + 8 * (r >> 2) + 4 * h];  // transposed
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const int rj = (j & 3) + 4 * (j >> 3), hj = (j >> 2) & 1;
                    const int target = jb * 32 + j + 1;
                    const float bq = (h == hj) ? acc[rj] : 0.0f;  // this wave's column j, not yet divided by L[j][j]
                    unsigned long long word;
                    for (;;) {  // every lane waits for its own word
                        word = __hip_atomic_load(&ring64[j * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if ((int)(word >> 32) == target) break;
                    }
                    const float a = __builtin_bit_cast(float, (unsigned)word);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc, 0, 0, 0);
                }
            }
        }
        if (FORM == 1) {
            __syncthreads();  // the leader's 1 / L[j][j] are in place
            if (wave > jb) {
                float* Bw = Lb + blk(wave, jb) * BLK;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;  // column of the block
                    Bw[i * LS + row] = acc[r] * ring_inv[row];
                }
            }
        }
        if (FORM == 0 && wave >= jb) {
            const bool leader = (wave == jb);
            float* Bw = Lb + blk(wave, jb) * BLK;
#pragma unroll
            for (int m = 0; m < 16; ++m) b[m] = Bw[i * LS + 2 * m + h];
            // Two columns per LDS round trip (the round trip, ~300 cycles with its issue, is what a step costs): the leader
            // publishes columns j and j+1 as they are BEFORE step j; every lane applies step j to its copy of column j+1 itself
            // (one more FMA per element) and then both rank-1 updates to its own entries.
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int j = 2 * p;
                float* line = ring + p * 64;
                const int target = jb * 16 + p + 1;
                float d0, c10, b11;
                if (leader) {
                    // plain LDS stores kept in program order by the compiler barriers: the LDS executes one wave's operations
                    // in order, and `volatile` would make hipcc wait for each of them (770 instead of ~300 cycles a step)
                    line[((i & 1) * 16 + (i >> 1)) * 2 + h] = b[p];  // half 0: column j, half 1: column j+1 (rows >= j valid)
                    asm volatile("" ::: "memory");
                    if (lane == 0) step_flag = target;
                    asm volatile("" ::: "memory");
                    // the leader's own pivots: no LDS round trip in front of the rsq chain
                    d0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b[p]), j));
                    c10 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b[p]), j + 1));
                    b11 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b[p]), 32 + j + 1));
                } else {
                    while (__hip_atomic_load(&step_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {}
                    asm volatile("" ::: "memory");
                    d0 = line[p * 2];              // (k = j,   A)
                    c10 = line[(16 + p) * 2];      // (k = j+1, A)
                    b11 = line[(16 + p) * 2 + 1];  // (k = j+1, B)
                }
                asm volatile("" : "+v"(d0), "+v"(c10), "+v"(b11));  // keep wave-uniform values in VGPRs (no SGPR spills / readlane traffic)
                // the columns are requested before the rsq chain below, not behind it
                float ck0[16], ck1[16];
#pragma unroll
                for (int m = p + 1; m < 16; ++m) {
                    ck0[m] = rd[p * 64 + 2 * m];
                    ck1[m] = rd[p * 64 + 2 * m + 1];
                }
                const float other = __shfl_xor(b[p], 32);  // half 0 gets its row's entry of column j+1, half 1 that of column j
                __builtin_amdgcn_sched_barrier(0);
                if (!(d0 > pivot_floor)) {
                    d0 = pivot_floor > 0.0f ? pivot_floor : 1.0e-30f;
                    if (leader) ++nclamp;
                }
                float inv0 = __builtin_amdgcn_rsqf(d0);  // v_rsq_f32, ~1 ulp: ample for a preconditioner
                asm volatile("" : "+v"(inv0));
                const float l10 = c10 * inv0;            // L[j+1][j]
                const float u10 = -l10 * inv0;           // column j+1 after step j: B[k] + u10 * A[k]
                float d1 = fmaf(u10, c10, b11);          // = A[j+1][j+1] - L[j+1][j]^2
                if (!(d1 > pivot_floor)) {
                    d1 = pivot_floor > 0.0f ? pivot_floor : 1.0e-30f;
                    if (leader) ++nclamp;
                }
                float inv1 = __builtin_amdgcn_rsqf(d1);
                asm volatile("" : "+v"(inv1));
                pinv[j] = inv0;
                pinv[j + 1] = inv1;
                const float aj0 = h ? other : b[p];          // this row's entries of columns j and j+1 (before step j)
                const float aj1p = h ? b[p] : other;
                const float lij0 = aj0 * inv0;               // L[row][j]
                const float t0 = -lij0 * inv0;
                const float aj1 = fmaf(t0, c10, aj1p);       // entry of column j+1 after step j
                const float lij1 = aj1 * inv1;               // L[row][j+1]
                const float t1 = -lij1 * inv1;
                b[p] = h ? ((leader && i == j + 1) ? d1 * inv1 : lij1) : ((leader && i == j) ? d0 * inv0 : lij0);
#pragma unroll
                for (int m = p + 1; m < 16; ++m) {
                    const float c1 = fmaf(u10, ck0[m], ck1[m]);
                    b[m] = fmaf(t1, c1, fmaf(t0, ck0[m], b[m]));
                }
            }
            {
#pragma unroll
                for (int m = 0; m < 16; ++m) Bw[i * LS + 2 * m + h] = (!leader || 2 * m + h <= i) ? b[m] : 0.0f;
            }
        }
        LEAF_STAMP(1 + 2 * jb);
        __syncthreads();
        // ---- trailing sub-blocks: A[ib][kb] -= A[ib][jb] * A[kb][jb]^T, jb < kb <= ib ----
        {
            int cnt = 0;
            for (int ib = jb + 1; ib < 4; ++ib)
                for (int kb = jb + 1; kb <= ib; ++kb, ++cnt) {
                    if ((cnt & 3) != wave) continue;
                    float* Cik = Lb + blk(ib, kb) * BLK;
                    f32x16 acc = blk_load(Cik, lane);
                    acc = blk_mma<false>(Lb + blk(ib, jb) * BLK, Lb + blk(kb, jb) * BLK, acc, -1.0f, lane);
                    blk_store(Cik, acc, lane);
                }
        }
        __syncthreads();
        LEAF_STAMP(2 + 2 * jb);
    }

    // ---- the four diagonal inverses, one per wave: lane c owns column c of X = L^-1 ----
    // Column-oriented forward substitution: once x[k] is known every later row's sum takes its term, s[r] += L[r][k] x[k] --
    // independent FMAs (the row-oriented form is a dependent chain per row and ran 7.7k-12k cycles depending on how hipcc
    // scheduled its LDS reads); the dependent chain is one FMA + one multiply per k.
    // (Measured and dropped, round 5: a second accumulator Y = I in the LEADER that takes every step like a follower's block ends as
    // L^-1 and this phase disappears -- but a second MFMA per step costs the chain-carrying wave 85 cycles of matrix pipe
    // (scripts/micro/mfma_chain.hip: 128 -> 211 cycles a step), the column phases grew by more than the 8.5k cycles saved.)
    {
        const float* Djj = Lb + blk(wave, wave) * BLK;
        float* Xjj = Xb + blk(wave, wave) * BLK;
        float x[32], sacc[32], lcol[3][32];  // lcol: columns k, k+1, k+2 of L (requested two steps ahead of their use:
                                             // left alone, hipcc reads each column right before its FMAs and waits)
#pragma unroll
        for (int r = 0; r < 32; ++r) sacc[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = c + 1; r < 32; ++r) lcol[c][r] = Djj[r * LS + c];  // broadcast LDS reads
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k + 2 < 32) {
#pragma unroll
                for (int r = k + 3; r < 32; ++r) lcol[(k + 2) % 3][r] = Djj[r * LS + k + 2];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float pk = FORM == 1 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, invl), k)) : pinv[k];
            x[k] = (((i == k) ? 1.0f : 0.0f) - sacc[k]) * pk;
#pragma unroll
            for (int r = k + 1; r < 32; ++r) sacc[r] = fmaf(lcol[k % 3][r], x[k], sacc[r]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k) Xjj[k * LS + i] = x[k];  // X[k][c = i]
        }
    }
    LEAF_STAMP(9);
    __syncthreads();

    // ---- assemble the 128x128 inverse.  With 64x64 halves, X = [[X_lo, 0], [-X_hi L_hl X_lo, X_hi]]: the off-diagonal 32x32
    // blocks of X_lo and X_hi first (X10, X32), together with the terms of T = L_hl X_lo that need neither; then the rest
    // of T; then X[2..3][0..1] = -X_hi T.  Six block products on the critical path and three barriers (the block-row
    // substitution this replaces: nine and six).  A wave reads back only scratch it wrote itself inside a phase (the LDS
    // executes one wave's operations in order). ----
    {
        auto L_ = [&](int ib, int jb) { return Lb + blk(ib, jb) * BLK; };
        auto X_ = [&](int ib, int jb) { return Xb + blk(ib, jb) * BLK; };
        float* T00 = Tb + 0 * BLK; float* T01 = Tb + 1 * BLK; float* T10 = Tb + 2 * BLK; float* T11 = Tb + 3 * BLK;
        float* S0 = Tb + 4 * BLK;  float* S1 = Tb + 5 * BLK;
        // phase 1
        if (wave == 0) {
            blk_store(S0, blk_mma<true>(L_(1, 0), X_(0, 0), zero16(), 1.0f, lane), lane);
            blk_store(X_(1, 0), blk_mma<true>(X_(1, 1), S0, zero16(), -1.0f, lane), lane);
        } else if (wave == 1) {
            blk_store(S1, blk_mma<true>(L_(3, 2), X_(2, 2), zero16(), 1.0f, lane), lane);
            blk_store(X_(3, 2), blk_mma<true>(X_(3, 3), S1, zero16(), -1.0f, lane), lane);
        } else if (wave == 2) {
            blk_store(T01, blk_mma<true>(L_(2, 1), X_(1, 1), zero16(), 1.0f, lane), lane);
            blk_store(T00, blk_mma<true>(L_(2, 0), X_(0, 0), zero16(), 1.0f, lane), lane);  // + L21 X10 in phase 2
        } else {
            blk_store(T11, blk_mma<true>(L_(3, 1), X_(1, 1), zero16(), 1.0f, lane), lane);
            blk_store(T10, blk_mma<true>(L_(3, 0), X_(0, 0), zero16(), 1.0f, lane), lane);  // + L31 X10 in phase 2
        }
        __syncthreads();
        // phase 2
        if (wave == 0) {
            blk_store(T00, blk_mma<true>(L_(2, 1), X_(1, 0), blk_load(T00, lane), 1.0f, lane), lane);
        } else if (wave == 1) {
            blk_store(T10, blk_mma<true>(L_(3, 1), X_(1, 0), blk_load(T10, lane), 1.0f, lane), lane);
        } else if (wave == 2) {
            blk_store(X_(2, 1), blk_mma<true>(X_(2, 2), T01, zero16(), -1.0f, lane), lane);
        } else {
            f32x16 q = blk_mma<true>(X_(3, 2), T01, zero16(), -1.0f, lane);
            blk_store(X_(3, 1), blk_mma<true>(X_(3, 3), T11, q, -1.0f, lane), lane);
        }
        __syncthreads();
        // phase 3
        if (wave == 0) {
            blk_store(X_(2, 0), blk_mma<true>(X_(2, 2), T00, zero16(), -1.0f, lane), lane);
        } else if (wave == 1) {
            f32x16 q = blk_mma<true>(X_(3, 2), T00, zero16(), -1.0f, lane);
            blk_store(X_(3, 0), blk_mma<true>(X_(3, 3), T10, q, -1.0f, lane), lane);
        }
        __syncthreads();
    }

    LEAF_STAMP(10);
    // ---- write back: L into the lower triangle of A, X (with its zero upper blocks) into dinv ----
#pragma unroll
    for (int b = 0; b < 16; ++b) {  // (unrolled: rolled up, every block waited for its own LDS and store round trips)
        const int ib = b >> 2, jb = b & 3;
        float4 xo = make_float4(0.f, 0.f, 0.f, 0.f);
        if (jb <= ib) {
            const float* lp = Lb + blk(ib, jb) * BLK + lr * LS + lc;
            const float* xp = Xb + blk(ib, jb) * BLK + lr * LS + lc;
            xo = make_float4(xp[0], xp[1], xp[2], xp[3]);
            float* ap = A + (int64_t)(ib * 32 + lr) * ld + jb * 32 + lc;
            if (ib != jb || lc + 3 <= lr) {
                *reinterpret_cast<float4*>(ap) = make_float4(lp[0], lp[1], lp[2], lp[3]);
            } else {  // the 4-wide group straddles or lies above the diagonal: keep the caller's upper entries
                for (int k = 0; k < 4; ++k)
                    if (lc + k <= lr) ap[k] = lp[k];
            }
        }
        *reinterpret_cast<float4*>(dinv + (ib * 32 + lr) * 128 + jb * 32 + lc) = xo;
    }
    if (lane == 0 && nclamp > 0 && clamped != nullptr) atomicAdd(clamped, nclamp);
    LEAF_STAMP(11);
}

}  // namespace

int launch_potrf_leaf(float* a, int64_t ld, float* dinv_block, int32_t* clamped, float pivot_floor, hipStream_t s) {
    if (NNGP_KNOB(3) != 1 && NNGP_KNOB(3) != 2)
    {
        #ifdef NNGP_TIMING_KNOBS
        if (NNGP_KNOB(3) == 3)  // A/B: round 4's column phases (column pairs through an LDS ring, per-lane FMAs)
            hipLaunchKernelGGL(k_potrf_leaf_panel<0>, dim3(1), dim3(256), 0, s, a, ld, dinv_block, clamped, pivot_floor);
        else
#endif
            hipLaunchKernelGGL(k_potrf_leaf_panel<1>, dim3(1), dim3(256), 0, s, a, ld, dinv_block, clamped, pivot_floor);
#ifdef NNGP_TIMING_KNOBS
        if (NNGP_KNOB(7) == 8) {  // timing study: per wave, cycles from kernel start to the end of each phase
            unsigned long long h[64];
            (void)hipDeviceSynchronize();
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_leaf_stamps), sizeof(h)) == hipSuccess)
                for (int w = 0; w < 4; ++w) {
                    fprintf(stderr, "leaf wave %d: load %llu |", w, h[w * 16]);
                    for (int jb = 0; jb < 4; ++jb) fprintf(stderr, " F%d %llu U%d %llu |", jb, h[w * 16 + 1 + 2 * jb], jb, h[w * 16 + 2 + 2 * jb]);
                    fprintf(stderr, " inv %llu asm %llu wb %llu\n", h[w * 16 + 9], h[w * 16 + 10], h[w * 16 + 11]);
                }
        }
#endif
    }
    else if (NNGP_KNOB(3) != 1)  // variant 1 (scalarised broadcasts) measured 57 us vs 84 us for variant 0
        hipLaunchKernelGGL(k_potrf_leaf<1>, dim3(1), dim3(256), 0, s, a, ld, dinv_block, clamped, pivot_floor, NNGP_KNOB(0));
    else
        hipLaunchKernelGGL(k_potrf_leaf<0>, dim3(1), dim3(256), 0, s, a, ld, dinv_block, clamped, pivot_floor, NNGP_KNOB(0));
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

// B[m, n] <- B * L^-T.  L is the n x n lower factor at `l`; dinv holds its inverted diagonal blocks.
int trsm_rlt_f32(float* b, int64_t ldb, int64_t m, const float* l, int64_t ldl, const float* dinv, int64_t n,
                 hipStream_t s) {
    if (m <= 0 || n <= 0) return 0;
    if (n == TB) {
        // in place: C aliases A, one column tile (see gemm_f32.hip header)
        return launch_gemm_nt_f32(b, ldb, b, ldb, dinv, TB, m, TB, TB, 1.0f, 0.0f, false, s);
    }
    // up to 1024 columns: one fused launch, 32 rows per workgroup resident in LDS (trsm_panel.hip) instead of a recursion of
    // 2 n / 128 - 1 small GEMMs (debug key 2 = 4: the recursion, for A/B timing)
    if (n <= 1024 && m % 32 == 0 && NNGP_KNOB(2) != 4)
        return launch_trsm_panel_f32(b, ldb, m, l, ldl, dinv, n, nullptr, 0, 1.0f, s);
    const int64_t n1 = (n / TB / 2) * TB, n2 = n - n1;
    NNGP_TRY(trsm_rlt_f32(b, ldb, m, l, ldl, dinv, n1, s));
    // B2 -= B1 * L21^T,  L21 = L[n1:, :n1]
    NNGP_TRY(launch_gemm_nt_f32(b + n1, ldb, b, ldb, l + n1 * ldl, ldl, m, n2, n1, -1.0f, 1.0f, false, s));
    return trsm_rlt_f32(b + n1, ldb, m, l + n1 * ldl + n1, ldl, dinv + (n1 / TB) * TB * TB, n2, s);
}

// B[m, n] <- B * U^-T with U = L^T stored explicitly (`lt`, upper triangular, row-major); dinvt holds the
// TRANSPOSED inverted diagonal blocks.  Together with trsm_rlt_f32 this applies (L L^T)^-1 to the rows of B:
// Z = (B L^-T) L^-1.  Columns are resolved last-to-first; the update B1 -= X2 U12^T is again an NT GEMM.
int trsm_rut_f32(float* b, int64_t ldb, int64_t m, const float* lt, int64_t ldl, const float* dinvt, int64_t n,
                 hipStream_t s) {
    if (m <= 0 || n <= 0) return 0;
    if (n == TB) return launch_gemm_nt_f32(b, ldb, b, ldb, dinvt, TB, m, TB, TB, 1.0f, 0.0f, false, s);
    const int64_t n1 = (n / TB / 2) * TB, n2 = n - n1;
    NNGP_TRY(trsm_rut_f32(b + n1, ldb, m, lt + n1 * ldl + n1, ldl, dinvt + (n1 / TB) * TB * TB, n2, s));
    // B1 -= X2 * U12^T,  U12 = lt[0:n1, n1:n]
    NNGP_TRY(launch_gemm_nt_f32(b, ldb, b + n1, ldb, lt + n1, ldl, m, n1, n2, -1.0f, 1.0f, false, s));
    return trsm_rut_f32(b, ldb, m, lt, ldl, dinvt, n1, s);
}

static int potrf_rec(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor,
                     hipStream_t s) {
    if (n == TB) return launch_potrf_leaf(a, ld, dinv, clamped, pivot_floor, s);
    const int64_t n1 = (n / TB / 2) * TB, n2 = n - n1;
    NNGP_TRY(potrf_rec(a, n1, ld, dinv, clamped, pivot_floor, s));
    float* a21 = a + n1 * ld;
    float* a22 = a21 + n1;
    NNGP_TRY(trsm_rlt_f32(a21, ld, n2, a, ld, dinv, n1, s));
    NNGP_TRY(launch_gemm_nt_f32(a22, ld, a21, ld, a21, ld, n2, n2, n1, -1.0f, 1.0f, true, s));
    return potrf_rec(a22, n2, ld, dinv + (n1 / TB) * TB * TB, clamped, pivot_floor, s);
}

// ---- look-ahead driver -----------------------------------------------------------------------------------
// The recursion above runs ~2300 dependent launches; about a third of its wall time is spent in kernels too small
// to fill 256 CUs (leaves, 128-wide triangular solves).  The look-ahead form cuts the matrix into block columns of
// width nb and uses two HIP streams: the high-priority "panel" stream factors block column k+1 (small kernels) while
// the low-priority "update" stream applies block column k to the rest of the trailing matrix (large SYRKs), so the
// small kernels run in the shadow of the large ones instead of in sequence with them.
//   panel  : wait col[k-1]; potrf(A_kk) (recursive chain of small kernels); record panel[k]
//   update : wait panel[k]; trsm(rows below); update diagonal block k+1 first; record col[k]; update the rest
int lookahead_create(LookAhead** out) {
    LookAhead* la = new (std::nothrow) LookAhead();
    NNGP_REQUIRE(la != nullptr, "lookahead_create: out of memory");
    // Optional CU partition (timing experiment, NNGP debug key 5 = 2): the panel stream owns `panel_cus` compute units,
    // the update stream the rest.  Measured at N = 32768: 138.6 ms (32 CUs) / 202 ms (16) / 140 ms (64) against 124.2 ms
    // for plain priority streams and 126.1 ms for the single-stream recursion -- so the default is priority streams.
    int ncu = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    const int panel_cus = NNGP_KNOB(4) > 0 ? NNGP_KNOB(4) : 32;
    bool masked = false;
    if (NNGP_KNOB(5) == 2 && ncu >= 64 && panel_cus < ncu) {
        const int words = (ncu + 31) / 32;
        uint32_t mp[16] = {0}, mu[16] = {0};
        // mask bit i addresses CU (i / 8) of XCD (i % 8) (measured: masks that thin out one XCD make it the straggler of
        // every GEMM), so the first 8*r bits take r CUs from every XCD
        for (int c = 0; c < ncu; ++c) (c < panel_cus ? mp : mu)[c / 32] |= (1u << (c % 32));
        if (words <= 16 && hipExtStreamCreateWithCUMask(&la->panel, words, mp) == hipSuccess) {
            if (hipExtStreamCreateWithCUMask(&la->update, words, mu) == hipSuccess) {
                masked = true;
            } else {
                (void)hipStreamDestroy(la->panel);
                la->panel = nullptr;
            }
        }
    }
    la->masked = masked;
    if (!masked) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
        // Priorities (numerically lower = higher).  With three levels or more: the diagonal-block chain and the bulk panel solves on
        // top, the trailing updates in the middle, and a lowest level for work that may only use what everything else leaves idle
        // (the `side` stream: inverted blocks, digit planes, split copies cut under the chain-bound last block columns).  The bulk
        // solves must outrank the update stream: their workgroups take the compute units a trailing-update launch gives back before
        // the next launch's persistent grid settles there.  With two levels there is no side stream.
        const bool three = least - greatest >= 2;
        const int prio_update = (three && (NNGP_KNOB(8) & 1)) ? least - 1 : least;
        if (hipStreamCreateWithPriority(&la->panel, hipStreamNonBlocking, greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&la->update, hipStreamNonBlocking, prio_update) != hipSuccess) {
            set_error("lookahead_create: hipStreamCreateWithPriority failed");
            delete la;
            return -1;
        }
        // Every further stream costs: HIP multiplexes streams onto a few hardware queues, and a fifth look-ahead stream with work pending
        // on it stretched the whole factorisation by a third (measured: 40.7 -> 55 ms at N = 32768; not when profiled).  The streams of
        // the round-4 schedule only exist when that experiment is switched on.
        if (!(NNGP_KNOB(8) & 1) || hipStreamCreateWithPriority(&la->bulk, hipStreamNonBlocking, greatest) != hipSuccess) la->bulk = nullptr;
        if (!(NNGP_KNOB(8) & 1) || !three || hipStreamCreateWithPriority(&la->side, hipStreamNonBlocking, NNGP_KNOB(11) == 1 ? prio_update : NNGP_KNOB(11) == 2 ? greatest : least) != hipSuccess) la->side = nullptr;
        // (the stream of the early diagonal-block product, debug key 8 = 4, only exists when that experiment is on: every further stream
        // costs -- see the note on the side stream in DESIGN.md)
        if (!(NNGP_KNOB(8) & 4) || hipStreamCreateWithPriority(&la->aux, hipStreamNonBlocking, greatest) != hipSuccess) la->aux = nullptr;
        la->prio_levels = least - greatest + 1;
    }
    hipEvent_t* all[5] = {&la->ev_in, &la->ev_panel_done, &la->ev_update_done, &la->ev_bulk_done, &la->ev_side_done};
    for (auto e : all)
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return -1; }
    for (int i = 0; i < LookAhead::kMaxSteps; ++i) {
        // (the seven events per step of round 4's schedule with the panel solves off the update stream exist in the knobs build only:
        // that schedule never runs in the product library)
#ifdef NNGP_TIMING_KNOBS
        hipEvent_t* per[11] = {&la->ev_panel[i], &la->ev_col[i], &la->ev_chunk[i], &la->ev_helper[i], &la->ev_far[i],
                               &la->ev_near[i],  &la->ev_tc[i],  &la->ev_tb[i],    &la->ev_split[i],  &la->ev_gp[i], &la->ev_c1[i]};
#else
        hipEvent_t* per[4] = {&la->ev_panel[i], &la->ev_col[i], &la->ev_chunk[i], &la->ev_helper[i]};
#endif
        for (auto e : per)
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
                set_error("hipEventCreate failed");
                return -1;
            }
    }
    *out = la;
    return 0;
}

void lookahead_destroy(LookAhead* la) {
    if (!la) return;
    (void)hipStreamSynchronize(la->panel);
    (void)hipStreamSynchronize(la->update);
    if (la->bulk) (void)hipStreamSynchronize(la->bulk);
    (void)hipStreamDestroy(la->panel);
    (void)hipStreamDestroy(la->update);
    if (la->bulk) (void)hipStreamDestroy(la->bulk);
    if (la->aux) { (void)hipStreamSynchronize(la->aux); (void)hipStreamDestroy(la->aux); }
    if (la->side) { (void)hipStreamSynchronize(la->side); (void)hipStreamDestroy(la->side); }
    if (la->ev_side_done) (void)hipEventDestroy(la->ev_side_done);
    (void)hipEventDestroy(la->ev_in); (void)hipEventDestroy(la->ev_panel_done); (void)hipEventDestroy(la->ev_update_done);
    if (la->ev_bulk_done) (void)hipEventDestroy(la->ev_bulk_done);
    for (int i = 0; i < LookAhead::kMaxSteps; ++i) {
        hipEvent_t per[11] = {la->ev_panel[i], la->ev_col[i], la->ev_chunk[i], la->ev_helper[i], la->ev_far[i],
                              la->ev_near[i],  la->ev_tc[i],  la->ev_tb[i],    la->ev_split[i],  la->ev_gp[i], la->ev_c1[i]};
        for (auto e : per)
            if (e) (void)hipEventDestroy(e);
    }
    for (int i = 0; i < LookAhead::kMaxTimed; ++i) {
        if (la->tu0[i]) (void)hipEventDestroy(la->tu0[i]);
        if (la->tu1[i]) (void)hipEventDestroy(la->tu1[i]);
    }
    delete la;
}

// ---- grouped form (round 3): deep-K trailing updates -----------------------------------------------------------------------
// A 256 x 256 tile of the trailing update costs its workgroup ~60k cycles of C traffic and tile hand-over beside ~112k cycles
// of matrix work per 1024 columns of K (in-kernel stamps, profiles/r3_h3_stamps.txt) -- a compute unit reads and writes its
// 512 KB of C at ~25 GB/s while its matrix pipe idles, and with one workgroup per unit nothing else runs there meanwhile.
// So K is deepened instead: the block columns are taken in groups of D.  A finished block column k is applied at once only
// to the remaining columns of ITS group (a narrow update, K = 1024); the columns beyond the group receive the whole group in
// ONE pass over C (K = 1024 D: measured 459 TF/s at K = 4096 against 360 at K = 1024).  The far update of a group is issued in
// pieces, in stream order between the next group's panel steps, so that every diagonal-block factorisation on the panel
// stream still runs under a large update:
//   F0        next diagonal block (float32 GEMM, K = 1024 D)                    -> releases the panel stream
//   chunk i   (while diagonal block gend + i is factored)  column gend + i + 1 of the next group (i = 0: also the rows of column
//             gend below its diagonal block) and the i-th share of the columns beyond the next group
// Every C tile is read and written 1 + (columns of its group before it) times per group pass instead of once per block column.
struct FarWork {
    bool active = false;
    int g0 = 0, np = 0;        // first block column of the group, number of panels
    int64_t r0 = 0;            // global row / column where the far region starts (= first row below the group)
    int next = 0;              // next chunk to issue
    int nchunks = 0;
    int64_t share_lo[LookAhead::kMaxSteps + 1] = {};  // far-far shares: global column ranges [share_lo[i], share_lo[i + 1])
};

static int h3_update_timed(LookAhead* la, float* c, int64_t ldc, const char* a, const char* b, int64_t ldp, int64_t pstride, int np,
                           int64_t lead, int64_t m, int64_t n, int64_t k, float alpha, bool lower, int64_t diag_shift, SplitWork* sw,
                           int reserve, double entries) {
    if (m <= 0 || n <= 0) return 0;
    const bool timed = la->time_updates && la->tu_count < LookAhead::kMaxTimed;
    if (timed) {
        const int t = la->tu_count;
        if (la->tu0[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&la->tu0[t]));
        if (la->tu1[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&la->tu1[t]));
        NNGP_HIP_CHECK(hipEventRecord(la->tu0[t], la->update));
    }
    NNGP_TRY(launch_gemm_nt_h3x(c, ldc, a, b, ldp, pstride, np, lead, m, n, k, alpha, 1.0f, lower, diag_shift, sw->counters, reserve,
                                la->update));
    if (timed) {
        const int t = la->tu_count++;
        NNGP_HIP_CHECK(hipEventRecord(la->tu1[t], la->update));
        la->tu_flops[t] = 2.0 * entries * ((double)np * (double)k - (double)lead);
        la->tu_bytes[t] = 8.0 * entries + 4.0 * ((double)m + (double)n) * ((double)np * (double)k - (double)lead);
    }
    return 0;
}

// entries of the region rows [0, m) x cols [0, n) with col <= row + shift
static double trap_entries(int64_t m, int64_t n, int64_t shift) {
    double e = 0.0;
    // rows r < n - shift see r + shift + 1 columns, the others n
    const int64_t full_from = (n - shift - 1 > 0) ? n - shift - 1 : 0;  // first row that sees all n columns
    const int64_t rt = full_from < m ? full_from : m;
    e += (double)rt * (double)(shift + 1) + 0.5 * (double)rt * (double)(rt - 1);
    if (m > rt) e += (double)(m - rt) * (double)n;
    return e;
}

// float32-MFMA update of the lower trapezoid rows [0, m) x cols [0, n), n <= m (lower triangle inside the top n x n square):
// c -= pa pb^T over K columns; pa, pb: rows of the factor (row stride ld), pb = the rows of the trapezoid's columns
static int f32_update_trap(float* c, int64_t ld, const float* pa, const float* pb, int64_t m, int64_t n, int64_t kk, hipStream_t s) {
    if (m <= 0 || n <= 0 || kk <= 0) return 0;
    NNGP_TRY(launch_gemm_nt_f32(c, ld, pa, ld, pb, ld, n, n, kk, -1.0f, 1.0f, true, s));
    if (m > n) NNGP_TRY(launch_gemm_nt_f32(c + n * ld, ld, pa + n * ld, ld, pb, ld, m - n, n, kk, -1.0f, 1.0f, false, s));
    return 0;
}

static int potrf_lookahead_grouped(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor, LookAhead* la,
                                   SplitWork* sw, hipStream_t user, int64_t nb, int D) {
    la->tu_count = 0;
    NNGP_HIP_CHECK(hipEventRecord(la->ev_in, user));
    NNGP_HIP_CHECK(hipStreamWaitEvent(la->panel, la->ev_in, 0));
    NNGP_HIP_CHECK(hipStreamWaitEvent(la->update, la->ev_in, 0));
    const int nblk = (int)((n + nb - 1) / nb);
    const int64_t ldp = 4 * sw->k_cap;
    const float ascale = -1.0f / (sw->scale * sw->scale);
    const int reserve = NNGP_KNOB(4) > 0 ? NNGP_KNOB(4) : 32;
    const int64_t lead0 = 64;  // columns of block column 0 that stay on the float32 MFMA (see potrf_lookahead_f32)
    const bool use_helper = !(NNGP_KNOB(2) >= 31 && NNGP_KNOB(2) <= 46) && reserve >= 8 && reserve % 8 == 0;  // debug key 2 = 30 + D: no helper grids
    auto plane_rows = [&](int col, int64_t row) { return sw->planes + (int64_t)col * sw->col_stride + row * ldp; };
    auto width = [&](int64_t col0) { return (n - col0 < nb) ? n - col0 : nb; };
    FarWork far;
    bool wrote_t = true;  // every panel solve also left the transposed split copy

    // One launch per chunk: up to three regions of the pending group's far update -- rows [row0, n) x cols [col0, col0 + w), lower
    // trapezoid (col <= row + shift relative to the region's origin) -- with all panels of the group in one pass over C; the
    // group's lead columns go through the float32 GEMM region by region.
    auto far_chunk = [&](int step) -> int {
        if (!far.active || far.next >= far.nchunks) return 0;
        const int i = far.next++;
        const int64_t r0 = far.r0;
        H3RegionSpec reg[4];
        int nreg = 0;
        auto add = [&](int64_t row0, int64_t col0, int64_t w, int64_t shift) {
            if (n - row0 > 0 && w > 0) reg[nreg++] = H3RegionSpec{row0, col0, n - row0, w, shift};
        };
        if (i == 0) {  // rows of the first column below its diagonal block (the block itself was F0)
            const int64_t w0 = width(r0);
            add(r0 + w0, r0, w0, w0);
        }
        const int64_t c1 = r0 + (int64_t)(i + 1) * nb;  // column gend + i + 1, if it belongs to the next group
        if (c1 < n && i + 1 < D) add(c1, c1, width(c1), 0);
        // shares are dealt from the LAST chunk backwards: the early chunks already carry the next group's own columns
        const int si = far.nchunks - 1 - i;
        if (far.share_lo[si + 1] > far.share_lo[si]) add(far.share_lo[si], far.share_lo[si], far.share_lo[si + 1] - far.share_lo[si], 0);
        if (far.next >= far.nchunks) far.active = false;
        if (nreg == 0) return 0;
        const int kl = far.g0 + far.np - 1;  // latest panel
        const int64_t lead = far.g0 == 0 ? lead0 : 0;
        double entries = 0.0;
        for (int r = 0; r < nreg; ++r) entries += trap_entries(reg[r].m, reg[r].n, reg[r].shift);
        const bool timed = la->time_updates && la->tu_count < LookAhead::kMaxTimed;
        if (timed) {
            const int t = la->tu_count;
            if (la->tu0[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&la->tu0[t]));
            if (la->tu1[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&la->tu1[t]));
            NNGP_HIP_CHECK(hipEventRecord(la->tu0[t], la->update));
        }
        // The `reserve` compute units this launch leaves to the panel stream idle once the diagonal-block chain (0.5 ms) is done:
        // a helper grid of that many workgroups, enqueued on the PANEL stream behind the chain, then joins the pass through the
        // shared work counters.  It may not start before everything the pass depends on (the update stream up to here), and the
        // update stream may not go on before it has finished.  Only for passes long enough to outlive the chain.
        double tiles = 0.0;
        for (int r = 0; r < nreg; ++r) tiles += trap_entries(reg[r].m, reg[r].n, reg[r].shift) / 65536.0;
        const bool helper = use_helper && step >= 0 && tiles * (double)far.np >= 4.0 * 3.0 * 224.0;  // >= ~3 rounds of K = 4096 tiles
        if (helper) NNGP_HIP_CHECK(hipEventRecord(la->ev_chunk[step], la->update));
        NNGP_TRY(launch_gemm_nt_h3r(a, ld, plane_rows(kl, 0), plane_rows(kl, 0), ldp, sw->col_stride, far.np, lead, reg, nreg, nb, ascale, 1.0f,
                                    true, sw->counters, reserve, la->update, nullptr, helper ? 1 : 0));
        if (helper) {
            NNGP_HIP_CHECK(hipStreamWaitEvent(la->panel, la->ev_chunk[step], 0));
            NNGP_TRY(launch_gemm_nt_h3r(a, ld, plane_rows(kl, 0), plane_rows(kl, 0), ldp, sw->col_stride, far.np, lead, reg, nreg, nb, ascale,
                                        1.0f, true, sw->counters, reserve, la->panel, nullptr, 2));
            NNGP_HIP_CHECK(hipEventRecord(la->ev_helper[step], la->panel));
            NNGP_HIP_CHECK(hipStreamWaitEvent(la->update, la->ev_helper[step], 0));
        }
        if (timed) {
            const int t = la->tu_count++;
            NNGP_HIP_CHECK(hipEventRecord(la->tu1[t], la->update));
            la->tu_flops[t] = 2.0 * entries * ((double)far.np * (double)nb - (double)lead);
            double rows_cols = 0.0;
            for (int r = 0; r < nreg; ++r) rows_cols += (double)reg[r].m + (double)reg[r].n;
            la->tu_bytes[t] = 8.0 * entries + 4.0 * rows_cols * ((double)far.np * (double)nb - (double)lead);
        }
        if (lead > 0) {  // columns [0, lead) of block column 0
            for (int r = 0; r < nreg; ++r) {
                float* cr = a + reg[r].row0 * ld + reg[r].col0;
                const float* pa = a + reg[r].row0 * ld;
                const float* pb = a + reg[r].col0 * ld;
                if (reg[r].shift == 0) {
                    NNGP_TRY(f32_update_trap(cr, ld, pa, pb, reg[r].m, reg[r].n, lead, la->update));
                } else {  // rows start `shift` below the columns: every entry of the columns is in
                    NNGP_TRY(launch_gemm_nt_f32(cr, ld, pa, ld, pb, ld, reg[r].m, reg[r].n, lead, -1.0f, 1.0f, false, la->update));
                }
            }
        }
        return 0;
    };

    int rc = 0;
    for (int k = 0; k < nblk && rc == 0; ++k) {
        const int64_t o = (int64_t)k * nb;
        const int64_t nbk = width(o);
        const int64_t m = n - o - nbk;  // rows below this block column
        float* akk = a + o * ld + o;
        float* dk = dinv + (o / TB) * TB * TB;
        const int g0 = (k / D) * D;
        const int gend = (g0 + D < nblk) ? g0 + D : nblk;
        // panel stream: factor the diagonal block (chain of small kernels)
        if (k > 0) NNGP_HIP_CHECK(hipStreamWaitEvent(la->panel, la->ev_col[k - 1], 0));
        rc = potrf_rec(akk, nbk, ld, dk, clamped, pivot_floor, la->panel);
        NNGP_HIP_CHECK(hipEventRecord(la->ev_panel[k], la->panel));
        if (rc != 0) break;
        // update stream, while that factorisation runs: the next piece of the previous group's far update
        rc = far_chunk(k);
        if (rc != 0 || m == 0) break;
        NNGP_HIP_CHECK(hipStreamWaitEvent(la->update, la->ev_panel[k], 0));
        // solve all rows below in one fused launch; it leaves their float16 split copy in this block column's planes
        const int64_t nb2 = width(o + nbk);
        float* below = akk + nbk * ld;           // panel rows below the diagonal block: [m, nbk]
        float* c = below + nbk;                  // trailing matrix: [m, m]
        char* pk_rows = plane_rows(k, o + nbk);
        // (round 4: the same launch also leaves the rows' TRANSPOSED split copy -- the operand of the posterior's "B L^-1" solves -- so
        // that no pass over the finished factor has to write it beside the first solve of a predict; debug key 9 = 8: not here)
        char* pt = (sw->planes_t != nullptr && nb == sw->k_cap && NNGP_KNOB(9) != 8) ? sw->planes_t : nullptr;
        wrote_t = wrote_t && pt != nullptr;
        if (k > 0 && nbk == 1024 && sw->ldiag != nullptr && sw->dfrag != nullptr) {
            rc = launch_split_diag_frag(akk, ld, nbk, sw->scale, sw->ldiag, dk, sw->dfrag, sw->dscale, la->update);
            if (rc == 0)
                rc = launch_trsm_panel_h3(below, ld, m, sw->ldiag, sw->dfrag, sw->dscale, nbk, pk_rows, ldp, sw->scale, la->update, pt, sw->col_stride, o + nbk, o);
        } else {
            rc = launch_trsm_panel_f32(below, ld, m, akk, ld, dk, nbk, pk_rows, ldp, sw->scale, la->update, pt, sw->col_stride, o + nbk, o);
        }
        if (rc != 0) break;
        if (k + 1 < gend) {
            // ---- inside the group: block column k goes to the group's remaining columns only (K = nbk) ----
            rc = launch_gemm_nt_f32(c, ld, below, ld, below, ld, nb2, nb2, nbk, -1.0f, 1.0f, true, la->update);  // next diagonal block
            NNGP_HIP_CHECK(hipEventRecord(la->ev_col[k], la->update));
            int64_t wn = (int64_t)(gend - 1 - k) * nb;  // columns of the group after block column k
            if (wn > m) wn = m;
            const int64_t lead = (k == 0) ? lead0 : 0;
            if (rc == 0 && m > nb2)
                rc = h3_update_timed(la, c + nb2 * ld, ld, pk_rows + nb2 * ldp + lead * 4, pk_rows + lead * 4, ldp, 0, 1, 0, m - nb2, wn, nbk - lead,
                                     ascale, true, nb2, sw, reserve, trap_entries(m - nb2, wn, nb2));
            if (rc == 0 && lead > 0 && m > nb2) {
                rc = launch_gemm_nt_f32(c + nb2 * ld, ld, below + nb2 * ld, ld, below, ld, m - nb2, nb2, lead, -1.0f, 1.0f, false, la->update);
                if (rc == 0 && wn > nb2)
                    rc = f32_update_trap(c + nb2 * ld + nb2, ld, below + nb2 * ld, below + nb2 * ld, m - nb2, wn - nb2, lead, la->update);
            }
        } else {
            // ---- the group is complete: its far update starts with the next diagonal block (all panels, float32 GEMM) ----
            const int np = k + 1 - g0;
            const float* rows = a + (o + nbk) * ld + (int64_t)g0 * nb;  // rows below the group, columns of the group
            rc = launch_gemm_nt_f32(c, ld, rows, ld, rows, ld, nb2, nb2, (int64_t)np * nb, -1.0f, 1.0f, true, la->update);
            NNGP_HIP_CHECK(hipEventRecord(la->ev_col[k], la->update));
            far = FarWork();
            far.active = true;
            far.g0 = g0;
            far.np = np;
            far.r0 = o + nbk;
            const int dn = (nblk - (k + 1) < D) ? nblk - (k + 1) : D;  // block columns of the next group
            far.nchunks = dn;
            // shares of the columns beyond the next group: equal trapezoid areas, whole block columns; few, large launches
            const int64_t f0 = far.r0 + (int64_t)dn * nb;
            for (int i = 0; i <= dn; ++i) far.share_lo[i] = f0 < n ? f0 : n;
            if (f0 < n) {
                const int64_t mf = n - f0;
                const double total = 0.5 * (double)mf * (double)mf;
                const double tiles = total / (256.0 * 256.0);
                int ns = (int)(tiles / 700.0);  // >= ~3 tiles per compute unit and launch
                if (ns < 1) ns = 1;
                if (ns > dn) ns = dn;
                double acc_area = 0.0;
                int sidx = 1;
                for (int64_t col = f0; col < n; col += nb) {
                    const int64_t w = width(col);
                    acc_area += (double)(n - col) * (double)w - 0.5 * (double)w * (double)w;
                    if (sidx < ns && acc_area >= total * sidx / ns) far.share_lo[sidx++] = col + w;
                }
                for (int i = sidx; i <= dn; ++i) far.share_lo[i] = n;
            }
        }
    }
    while (rc == 0 && far.active) rc = far_chunk(-1);  // (nothing is left when the loop ran to the last block column)
    if (rc == 0) sw->l_ready = true;
    if (rc == 0 && wrote_t && sw->planes_t != nullptr) sw->lt_ready = true;
    NNGP_HIP_CHECK(hipEventRecord(la->ev_panel_done, la->panel));
    NNGP_HIP_CHECK(hipEventRecord(la->ev_update_done, la->update));
    NNGP_HIP_CHECK(hipStreamWaitEvent(user, la->ev_panel_done, 0));
    NNGP_HIP_CHECK(hipStreamWaitEvent(user, la->ev_update_done, 0));
    return rc;
}

// ---- grouped form, round 4: the panel solves leave the update stream ---------------------------------------------------------------
// Round 3's timeline (profiles/r3_timeline_cfg3.csv) has 7.8 ms of panel solves, float32 diagonal-block updates and splits IN LINE
// with the 28.4 ms of split-float16 trailing updates on the update stream, most of the chip idle meanwhile.  Here the update stream
// carries the trailing updates only.  Per block column k:
//   panel stream   P_k  factor the diagonal block               (needs the update stream up to the far chunk issued at step k - 1)
//                  Tc_k solve the nb rows right below it        (needs the near update of block column k - 1)
//                  G_k  their product onto diagonal block k + 1 (float32 MFMA, K = nb)           -> P_{k+1}
//   bulk stream    Tb_k solve all other rows below              (same inputs as Tc_k; a priority above the update stream's: its
//                       workgroups run on the compute units the trailing updates leave free and take over whatever a finishing
//                       update launch gives back, before the next launch's persistent grid settles there)
//   update stream  the far chunk of step k (previous group's panels, K = nb D), then N_k = block column k onto the rest of its
//                       group (K = nb; needs Tc_k and Tb_k)
// The next group's first diagonal block used to receive the whole finished group in one float32 GEMM (K = nb D) at the head of the
// chain; now the D - 1 earlier panels go there as soon as THEY are solved (bulk stream, behind Tb of the group's last-but-one
// column) and only the last panel's K = nb product is left in the chain.  The split copy of a diagonal block and its inverted
// 128-blocks (operands of the fused solves) alternate between two buffers: Tb_k may still be reading one while the panel stream
// prepares block column k + 1.
static int potrf_lookahead_grouped_v4(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor, LookAhead* la,
                                      SplitWork* sw, hipStream_t user, int64_t nb, int D, TriInv* ti) {
    la->tu_count = 0;
    hipStream_t SP = la->panel, SU = la->update, SB = (NNGP_KNOB(8) & 2) ? la->panel : la->bulk;
#ifdef NNGP_TIMING_KNOBS
    static hipEvent_t dbg_ev[8] = {};
    const bool dbg = NNGP_KNOB(12) != 0;
    if (dbg && dbg_ev[0] == nullptr)
        for (auto& e : dbg_ev) (void)hipEventCreate(&e);
    if (dbg) (void)hipEventRecord(dbg_ev[0], user);
#endif
    NNGP_HIP_CHECK(hipEventRecord(la->ev_in, user));
    NNGP_HIP_CHECK(hipStreamWaitEvent(SP, la->ev_in, 0));
    NNGP_HIP_CHECK(hipStreamWaitEvent(SU, la->ev_in, 0));
    NNGP_HIP_CHECK(hipStreamWaitEvent(SB, la->ev_in, 0));
    const int nblk = (int)((n + nb - 1) / nb);
    const int64_t ldp = 4 * sw->k_cap;
    const float ascale = -1.0f / (sw->scale * sw->scale);
    const int reserve = NNGP_KNOB(4) > 0 ? NNGP_KNOB(4) : 32;
    const int64_t lead0 = 64;  // columns of block column 0 that stay on the float32 MFMA (see potrf_lookahead_f32)
    // Helper grids (round 3: the reserved compute units join a far chunk once the diagonal-block chain is done) are OFF here: the chain
    // now runs ahead of the update stream and the bulk solves live on those units -- a persistent helper grid starves both (measured:
    // chains of 1.3 - 2.8 ms instead of 0.6, Cholesky 54.8 ms).  Debug key 8 = 8: on, behind Tb_k on the bulk stream.
    const bool use_helper = (NNGP_KNOB(8) & 8) && reserve >= 8 && reserve % 8 == 0;
    const bool early_gp = (NNGP_KNOB(8) & 4) != 0;  // measured: one more CG iteration (6 instead of 5 at N = 32768) -- off
    auto plane_rows = [&](int col, int64_t row) { return sw->planes + (int64_t)col * sw->col_stride + row * ldp; };
    auto width = [&](int64_t col0) { return (n - col0 < nb) ? n - col0 : nb; };
    FarWork far;
    // a far chunk's helper grid (see potrf_lookahead_grouped) is enqueued on the panel stream BEHIND the step's chain work
    struct PendingHelper {
        bool on = false;
        H3RegionSpec reg[4];
        int nreg = 0, kl = 0, np = 0, step = 0;
        int64_t lead = 0;
    } ph;

    auto timed_begin = [&]() -> int {
        const bool timed = la->time_updates && la->tu_count < LookAhead::kMaxTimed;
        if (timed) {
            const int t = la->tu_count;
            if (la->tu0[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&la->tu0[t]));
            if (la->tu1[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&la->tu1[t]));
            NNGP_HIP_CHECK(hipEventRecord(la->tu0[t], SU));
        }
        return timed ? 1 : 0;
    };
    auto timed_end = [&](const H3RegionSpec* reg, int nreg, double kk) -> int {
        const int t = la->tu_count++;
        NNGP_HIP_CHECK(hipEventRecord(la->tu1[t], SU));
        double entries = 0.0, rows_cols = 0.0;
        for (int r = 0; r < nreg; ++r) {
            entries += trap_entries(reg[r].m, reg[r].n, reg[r].shift);
            rows_cols += (double)reg[r].m + (double)reg[r].n;
        }
        la->tu_flops[t] = 2.0 * entries * kk;
        la->tu_bytes[t] = 8.0 * entries + 4.0 * rows_cols * kk;
        return 0;
    };
    // one split-float16 launch of the pending group's far update over `reg` (+ the float32 pass of the group's lead columns)
    auto far_launch = [&](const H3RegionSpec* reg, int nreg, int step, bool allow_helper) -> int {
        if (nreg == 0) return 0;
        const int kl = far.g0 + far.np - 1;  // latest panel
        const int64_t lead = far.g0 == 0 ? lead0 : 0;
        double tiles = 0.0;
        for (int r = 0; r < nreg; ++r) tiles += trap_entries(reg[r].m, reg[r].n, reg[r].shift) / 65536.0;
        // (a pass with lead columns is followed by float32 launches over the same regions: no helper there)
        const bool helper = allow_helper && use_helper && lead == 0 && step >= 0 && tiles * (double)far.np >= 4.0 * 3.0 * 224.0;
        const int timed = timed_begin();
        if (timed < 0) return timed;
        if (helper) NNGP_HIP_CHECK(hipEventRecord(la->ev_chunk[step], SU));
        NNGP_TRY(launch_gemm_nt_h3r(a, ld, plane_rows(kl, 0), plane_rows(kl, 0), ldp, sw->col_stride, far.np, lead, reg, nreg, nb, ascale, 1.0f,
                                    true, sw->counters, reserve, SU, nullptr, helper ? 1 : 0));
        if (timed) NNGP_TRY(timed_end(reg, nreg, (double)far.np * (double)nb - (double)lead));
        if (helper) {
            ph.on = true;
            ph.nreg = nreg;
            for (int r = 0; r < nreg; ++r) ph.reg[r] = reg[r];
            ph.kl = kl; ph.np = far.np; ph.step = step; ph.lead = lead;
        }
        if (lead > 0) {  // columns [0, lead) of block column 0
            for (int r = 0; r < nreg; ++r) {
                float* cr = a + reg[r].row0 * ld + reg[r].col0;
                const float* pa = a + reg[r].row0 * ld;
                const float* pb = a + reg[r].col0 * ld;
                if (reg[r].shift == 0) {
                    NNGP_TRY(f32_update_trap(cr, ld, pa, pb, reg[r].m, reg[r].n, lead, SU));
                } else {  // rows start `shift` below the columns: every entry of the columns is in
                    NNGP_TRY(launch_gemm_nt_f32(cr, ld, pa, ld, pb, ld, reg[r].m, reg[r].n, lead, -1.0f, 1.0f, false, SU));
                }
            }
        }
        return 0;
    };
    // the pieces of the pending far update that belong to step `step` (block column gend + i of the next group is being factored)
    auto far_step = [&](int step, bool* colrows_event, bool* c1_event) -> int {
        if (!far.active || far.next >= far.nchunks) return 0;
        const int i = far.next++;
        const int64_t r0 = far.r0;
        H3RegionSpec reg[4];
        int nreg = 0;
        auto add = [&](int64_t row0, int64_t col0, int64_t w, int64_t shift) {
            if (n - row0 > 0 && w > 0) reg[nreg++] = H3RegionSpec{row0, col0, n - row0, w, shift};
        };
        if (i == 0) {  // rows of the next group's first column below its diagonal block: its panel solve waits for exactly these
            const int64_t w0 = width(r0);
            add(r0 + w0, r0, w0, w0);
            if (nreg > 0 && step >= 0) {
                NNGP_TRY(far_launch(reg, nreg, step, false));
                NNGP_HIP_CHECK(hipEventRecord(la->ev_col[step], SU));
                *colrows_event = true;
                nreg = 0;
            }
        }
        // column gend + i + 1 of the next group in a launch of its own: the chain (G, P of that column) waits for this launch only,
        // not for the chunk's share of the columns beyond -- the diagonal-block chain then runs a whole step ahead of the update
        // stream and the bulk solve of a block column has the following far chunk to hide under (debug key 8 = 16 only)
        const int64_t c1 = r0 + (int64_t)(i + 1) * nb;
        const bool split_c1 = (NNGP_KNOB(8) & 16) && step >= 0;  // measured: +0.6 ms (one more launch per step) -- off
        if (c1 < n && i + 1 < D) add(c1, c1, width(c1), 0);
        if (split_c1 && nreg > 0) {
            NNGP_TRY(far_launch(reg, nreg, step, false));
            nreg = 0;
        }
        if (split_c1) {
            NNGP_HIP_CHECK(hipEventRecord(la->ev_c1[step], SU));
            *c1_event = true;
        }
        // shares are dealt from the LAST chunk backwards: the early chunks already carry the next group's own columns
        const int si = far.nchunks - 1 - i;
        if (far.share_lo[si + 1] > far.share_lo[si]) add(far.share_lo[si], far.share_lo[si], far.share_lo[si + 1] - far.share_lo[si], 0);
        if (far.next >= far.nchunks) far.active = false;
        return far_launch(reg, nreg, step, true);
    };

    int rc = 0;
    bool side_used = false;
    const int kTriInvTail = NNGP_KNOB(10) > 0 ? NNGP_KNOB(10) : 7;  // block columns from the end where the finished blocks' inverses are issued
    for (int k = 0; k < nblk && rc == 0; ++k) {
        const int64_t o = (int64_t)k * nb;
        const int64_t nbk = width(o);
        const int64_t m = n - o - nbk;  // rows below this block column
        float* akk = a + o * ld + o;
        float* dk = dinv + (o / TB) * TB * TB;
        const int g0 = (k / D) * D;
        const int gend = (g0 + D < nblk) ? g0 + D : nblk;
        // ---- panel stream: P_k ----
        if (k > 0) NNGP_HIP_CHECK(hipStreamWaitEvent(SP, la->ev_chain[k - 1], 0));
        rc = potrf_rec(akk, nbk, ld, dk, clamped, pivot_floor, SP);
        NNGP_HIP_CHECK(hipEventRecord(la->ev_panel[k], SP));
        if (rc != 0) break;
        // The inverted diagonal blocks of the blocked solves (solve.hip, triinv_build): from here on the chain of diagonal-block
        // factorisations bounds the factorisation and most of the chip idles -- the blocks of the block columns behind us are
        // inverted now, on the lowest-priority stream, instead of after the factorisation.  Debug key 8 = 32 only: measured, the side work
        // delays the chain by what it saves after the factorisation (42.1 against 42.3 ms at N = 32768).
        if (ti != nullptr && la->side != nullptr && (NNGP_KNOB(8) & 32) && k == nblk - kTriInvTail && ti->bs % nb == 0 && k >= 2) {
            const int64_t jdone = ((int64_t)(k - 1) * nb) / ti->bs;  // block columns 0 .. k - 2 are final (P_{k-1} has been waited for by Tc_{k-1})
            if (jdone > 0) {
                hipStream_t SS = NNGP_KNOB(11) == 3 ? la->aux : NNGP_KNOB(11) == 4 ? la->bulk : la->side;
                NNGP_HIP_CHECK(hipStreamWaitEvent(SS, la->ev_tc[k - 1], 0));
#ifdef NNGP_TIMING_KNOBS
                if (dbg) (void)hipEventRecord(dbg_ev[1], SS);
#endif
                rc = triinv_build_range(a, ld, dinv, n, *ti, 0, jdone, SS);
                if (rc != 0) break;
#ifdef NNGP_TIMING_KNOBS
                if (dbg) (void)hipEventRecord(dbg_ev[2], SS);
#endif
                NNGP_HIP_CHECK(hipEventRecord(la->ev_side_done, SS));
                ti->done_blocks = jdone;
                side_used = true;
            }
        }
        // ---- update stream: this step's share of the previous group's far update ----
        bool colrows = false, c1ev = false;
        ph.on = false;
        rc = far_step(k, &colrows, &c1ev);
        NNGP_HIP_CHECK(hipEventRecord(la->ev_far[k], SU));
        // what the chain's next links (G_k, P_{k+1}) wait for: everything on the update stream that touches diagonal block k + 1 --
        // the near updates up to N_{k-1} and this step's launch over column k + 1 (or, at a group's end, the whole chunk: the
        // previous group's shares cover the next group's first diagonal block)
        la->ev_chain[k] = (c1ev && k + 1 < gend) ? la->ev_c1[k] : la->ev_far[k];
        if (rc != 0 || m == 0) break;
        // ---- the panel solves: Tc_k on the panel stream, Tb_k on the bulk stream ----
        const int64_t nb2 = width(o + nbk);
        float* below = akk + nbk * ld;           // panel rows below the diagonal block: [m, nbk]
        float* c = below + nbk;                  // trailing matrix: [m, m]
        char* pk_rows = plane_rows(k, o + nbk);
        hipEvent_t ev_rows = nullptr;            // block column k below its diagonal block has received everything
        if (k > g0) ev_rows = la->ev_near[k - 1];
        else if (colrows) ev_rows = la->ev_col[k];
        if (ev_rows != nullptr) NNGP_HIP_CHECK(hipStreamWaitEvent(SP, ev_rows, 0));
        const bool solve_h3 = k > 0 && nbk == 1024 && sw->ldiag != nullptr && sw->dfrag != nullptr;
        char* ldiag = sw->ldiag + (int64_t)(k & 1) * 4 * sw->k_cap * sw->k_cap;
        float* dfrag = sw->dfrag + (int64_t)(k & 1) * sw->k_cap * 128;
        float* dscale = sw->dscale + (int64_t)(k & 1) * sw->k_cap;
        if (solve_h3) {
            rc = launch_split_diag_frag(akk, ld, nbk, sw->scale, ldiag, dk, dfrag, dscale, SP);
            if (rc == 0) rc = launch_trsm_panel_h3(below, ld, nb2, ldiag, dfrag, dscale, nbk, pk_rows, ldp, sw->scale, SP);
        } else {
            rc = launch_trsm_panel_f32(below, ld, nb2, akk, ld, dk, nbk, pk_rows, ldp, sw->scale, SP);
        }
        if (rc != 0) break;
        NNGP_HIP_CHECK(hipEventRecord(la->ev_tc[k], SP));  // (also: the diagonal block is factored and split)
        if (m > nb2) {
            NNGP_HIP_CHECK(hipStreamWaitEvent(SB, la->ev_tc[k], 0));
            if (ev_rows != nullptr) NNGP_HIP_CHECK(hipStreamWaitEvent(SB, ev_rows, 0));
            if (solve_h3)
                rc = launch_trsm_panel_h3(below + nb2 * ld, ld, m - nb2, ldiag, dfrag, dscale, nbk, pk_rows + nb2 * ldp, ldp, sw->scale, SB);
            else
                rc = launch_trsm_panel_f32(below + nb2 * ld, ld, m - nb2, akk, ld, dk, nbk, pk_rows + nb2 * ldp, ldp, sw->scale, SB);
            if (rc != 0) break;
        }
        NNGP_HIP_CHECK(hipEventRecord(la->ev_tb[k], SB));
        // The far chunk's helper grid: the `reserve` compute units the chunk's own grid leaves free carry the diagonal-block chain (a
        // workgroup at a time) and the bulk solve; that many workgroups join the chunk from a stream of their own (debug key 8 = 8:
        // from the bulk stream, behind Tb_k).  Not from the panel stream as in round 3: the chain no longer waits for the chunk.
        if (ph.on) {
            hipStream_t SH = SB;
            NNGP_HIP_CHECK(hipStreamWaitEvent(SH, la->ev_chunk[ph.step], 0));
            rc = launch_gemm_nt_h3r(a, ld, plane_rows(ph.kl, 0), plane_rows(ph.kl, 0), ldp, sw->col_stride, ph.np, ph.lead, ph.reg, ph.nreg, nb,
                                    ascale, 1.0f, true, sw->counters, reserve, SH, nullptr, 2);
            if (rc != 0) break;
            NNGP_HIP_CHECK(hipEventRecord(la->ev_helper[ph.step], SH));
            NNGP_HIP_CHECK(hipStreamWaitEvent(SU, la->ev_helper[ph.step], 0));
            ph.on = false;
        }
        // ---- G_k, the product of the rows just solved onto the next diagonal block.  This step's far chunk carries that block's
        // share of the previous group (same C tiles): the product waits for it, as P_{k+1} has to anyway. ----
        NNGP_HIP_CHECK(hipStreamWaitEvent(SP, la->ev_chain[k], 0));
        if (k + 1 < gend) {
            rc = launch_gemm_nt_f32(c, ld, below, ld, below, ld, nb2, nb2, nbk, -1.0f, 1.0f, true, SP);
        } else {
            // The next group's first diagonal block receives the whole finished group.  Its D - 1 earlier panels are solved once the
            // bulk stream has passed Tb_{k-1}, and the previous group's last far chunk (this step's: it covers that block too) has to
            // be through: behind both, on a stream of its own, the K = nb (D - 1) product runs beside P_k when the chain is what
            // bounds the factorisation (the late block columns); only the last panel's K = nb product is left in the chain.
            const int np = k + 1 - g0;
            const float* rows = a + (o + nbk) * ld + (int64_t)g0 * nb;  // rows of the next diagonal block, columns of the group
            if (early_gp && np >= 2 && la->aux != nullptr) {
                NNGP_HIP_CHECK(hipStreamWaitEvent(la->aux, la->ev_far[k], 0));
                NNGP_HIP_CHECK(hipStreamWaitEvent(la->aux, la->ev_tb[k - 1], 0));
                rc = launch_gemm_nt_f32(c, ld, rows, ld, rows, ld, nb2, nb2, (int64_t)(np - 1) * nb, -1.0f, 1.0f, true, la->aux);
                if (rc != 0) break;
                NNGP_HIP_CHECK(hipEventRecord(la->ev_gp[k], la->aux));
                NNGP_HIP_CHECK(hipStreamWaitEvent(SP, la->ev_gp[k], 0));
                rc = launch_gemm_nt_f32(c, ld, below, ld, below, ld, nb2, nb2, nbk, -1.0f, 1.0f, true, SP);
            } else {  // all panels of the group at once (the earlier panels' rows were solved on the bulk stream)
                if (np >= 2) NNGP_HIP_CHECK(hipStreamWaitEvent(SP, la->ev_tb[k - 1], 0));
                rc = launch_gemm_nt_f32(c, ld, rows, ld, rows, ld, nb2, nb2, (int64_t)np * nb, -1.0f, 1.0f, true, SP);
            }
        }
        if (rc != 0) break;
        // ---- update stream: block column k is solved ----
        NNGP_HIP_CHECK(hipStreamWaitEvent(SU, la->ev_tc[k], 0));
        NNGP_HIP_CHECK(hipStreamWaitEvent(SU, la->ev_tb[k], 0));
        if (k + 1 < gend) {
            // inside the group: block column k goes to the group's remaining columns only (K = nbk)
            int64_t wn = (int64_t)(gend - 1 - k) * nb;  // columns of the group after block column k
            if (wn > m) wn = m;
            const int64_t lead = (k == 0) ? lead0 : 0;
            if (m > nb2) {
                rc = h3_update_timed(la, c + nb2 * ld, ld, pk_rows + nb2 * ldp + lead * 4, pk_rows + lead * 4, ldp, 0, 1, 0, m - nb2, wn, nbk - lead,
                                     ascale, true, nb2, sw, reserve, trap_entries(m - nb2, wn, nb2));
                if (rc == 0 && lead > 0) {
                    rc = launch_gemm_nt_f32(c + nb2 * ld, ld, below + nb2 * ld, ld, below, ld, m - nb2, nb2, lead, -1.0f, 1.0f, false, SU);
                    if (rc == 0 && wn > nb2)
                        rc = f32_update_trap(c + nb2 * ld + nb2, ld, below + nb2 * ld, below + nb2 * ld, m - nb2, wn - nb2, lead, SU);
                }
            }
            NNGP_HIP_CHECK(hipEventRecord(la->ev_near[k], SU));
        } else {
            // the group is complete: set up its far update (issued in pieces at the next steps)
            const int np = k + 1 - g0;
            far = FarWork();
            far.active = true;
            far.g0 = g0;
            far.np = np;
            far.r0 = o + nbk;
            const int dn = (nblk - (k + 1) < D) ? nblk - (k + 1) : D;  // block columns of the next group
            far.nchunks = dn;
            // shares of the columns beyond the next group: equal trapezoid areas, whole block columns; few, large launches
            const int64_t f0 = far.r0 + (int64_t)dn * nb;
            for (int i = 0; i <= dn; ++i) far.share_lo[i] = f0 < n ? f0 : n;
            if (f0 < n) {
                const int64_t mf = n - f0;
                const double total = 0.5 * (double)mf * (double)mf;
                const double tiles = total / (256.0 * 256.0);
                int ns = (int)(tiles / 700.0);  // >= ~3 tiles per compute unit and launch
                if (ns < 1) ns = 1;
                if (ns > dn) ns = dn;
                double acc_area = 0.0;
                int sidx = 1;
                for (int64_t col = f0; col < n; col += nb) {
                    const int64_t w = width(col);
                    acc_area += (double)(n - col) * (double)w - 0.5 * (double)w * (double)w;
                    if (sidx < ns && acc_area >= total * sidx / ns) far.share_lo[sidx++] = col + w;
                }
                for (int i = sidx; i <= dn; ++i) far.share_lo[i] = n;
            }
        }
    }
    bool dummy = false, dummy2 = false;
    while (rc == 0 && far.active) rc = far_step(-1, &dummy, &dummy2);  // (nothing is left when the loop ran to the last block column)
    if (rc == 0) sw->l_ready = true;
#ifdef NNGP_TIMING_KNOBS
    if (dbg) { (void)hipEventRecord(dbg_ev[3], SP); (void)hipEventRecord(dbg_ev[4], SU); (void)hipEventRecord(dbg_ev[5], SB); }
#endif
    NNGP_HIP_CHECK(hipEventRecord(la->ev_panel_done, SP));
    NNGP_HIP_CHECK(hipEventRecord(la->ev_update_done, SU));
    NNGP_HIP_CHECK(hipStreamWaitEvent(user, la->ev_panel_done, 0));
    NNGP_HIP_CHECK(hipStreamWaitEvent(user, la->ev_update_done, 0));
    NNGP_HIP_CHECK(hipEventRecord(la->ev_bulk_done, SB));
    NNGP_HIP_CHECK(hipStreamWaitEvent(user, la->ev_bulk_done, 0));
    if (side_used) NNGP_HIP_CHECK(hipStreamWaitEvent(user, la->ev_side_done, 0));
#ifdef NNGP_TIMING_KNOBS
    if (dbg) {
        (void)hipEventRecord(dbg_ev[6], user);
        (void)hipEventSynchronize(dbg_ev[6]);
        float t[7] = {};
        for (int i = 1; i <= 6; ++i)
            if (i > 2 || side_used) (void)hipEventElapsedTime(&t[i], dbg_ev[0], dbg_ev[i]);
        fprintf(stderr, "v4 timing (ms from entry): side %.2f -> %.2f | panel end %.2f  update end %.2f  bulk end %.2f | user resumes %.2f\n", t[1], t[2], t[3],
                t[4], t[5], t[6]);
    }
#endif
    return rc;  // (the aux stream's last product was waited for by the panel stream)
}

int potrf_lookahead_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor,
                        LookAhead* la, SplitWork* sw, hipStream_t user, TriInv* ti) {
    if (ti != nullptr) ti->done_blocks = 0;
    NNGP_REQUIRE(n > 0 && n % TB == 0, "potrf_f32: n must be a positive multiple of %d (got %lld)", TB, (long long)n);
    // block-column width: 1024 measured best at N = 32768 (119.4 ms; 2048: 121.3, 4096: 123.2, recursion only: 125)
    int64_t nb = NNGP_KNOB(1) > 0 ? (int64_t)NNGP_KNOB(1) : kLookAheadNb;
    nb = (nb / TB) * TB;
    // large trailing updates on the float16 matrix pipe (gemm_h3.hip) unless the workspace is missing / too small or
    // debug key 2 == 2 asks for the float32-MFMA updates (A/B timing)
    const bool h3 = sw != nullptr && sw->planes != nullptr && sw->counters != nullptr && sw->k_cap == nb && sw->rows_cap >= n + 256 &&
                    sw->col_stride >= sw->rows_cap * 4 * sw->k_cap && NNGP_KNOB(2) != 2;
    if (sw != nullptr) sw->l_ready = sw->lt_ready = false;
    // (measured and dropped: solving the panel rows in four row chunks on a third stream, each chunk's trailing update
    // starting as soon as it is solved -- 64.8 vs 59.2 ms: four smaller split-float16 launches lose more in their tails
    // than the overlap gains)
    // (measured twice and dropped: inverting each diagonal block on the panel stream as it is factored -- neutral, 59.1 vs
    // 58.9 ms, and 58.1 vs 57.9 ms with CUs reserved for the panel stream -- and solving the panel rows with that inverse as one GEMM: Cholesky -3.7 ms but CG iterations 6 -> 8)
    if (la == nullptr || NNGP_KNOB(2) == 1 || n < 4 * nb || (n + nb - 1) / nb > LookAhead::kMaxSteps)
        return potrf_f32(a, n, ld, dinv, clamped, pivot_floor, user);
    // grouped form: deep-K far updates (debug key 2 = 10 + D overrides the group size; D = 1: the round-2 form below)
    const int group = (NNGP_KNOB(2) >= 11 && NNGP_KNOB(2) <= 26) ? NNGP_KNOB(2) - 10 : (NNGP_KNOB(2) >= 31 && NNGP_KNOB(2) <= 46) ? NNGP_KNOB(2) - 30 : kLookAheadGroup;
    if (h3 && group > 1 && nb == 1024 && (NNGP_KNOB(2) == 0 || NNGP_KNOB(2) >= 11) && NNGP_KNOB(3) == 0 && ld % 4 == 0) {
        // (round 4's schedule with the panel solves off the update stream -- measured equal at N = 32768, slower at the other sizes,
        // see the note above it -- runs on request only: debug key 8 bit 1, set before the model is created)
        if (la->bulk != nullptr && (NNGP_KNOB(8) & 1))
            return potrf_lookahead_grouped_v4(a, n, ld, dinv, clamped, pivot_floor, la, sw, user, nb, group, ti);
        return potrf_lookahead_grouped(a, n, ld, dinv, clamped, pivot_floor, la, sw, user, nb, group);
    }
    la->tu_count = 0;
    NNGP_HIP_CHECK(hipEventRecord(la->ev_in, user));
    NNGP_HIP_CHECK(hipStreamWaitEvent(la->panel, la->ev_in, 0));
    NNGP_HIP_CHECK(hipStreamWaitEvent(la->update, la->ev_in, 0));
    int rc = 0;
    int k = 0;
    bool panel_split_done = false;  // the panel solve already left the split copy of this block column's rows in place
    for (int64_t o = 0; o < n && rc == 0; o += nb, ++k) {
        const int64_t nbk = (n - o < nb) ? n - o : nb;
        const int64_t m = n - o - nbk;  // rows below this block column
        float* akk = a + o * ld + o;
        float* dk = dinv + (o / TB) * TB * TB;
        // panel stream: factor the diagonal block (chain of small kernels)
        if (k > 0) NNGP_HIP_CHECK(hipStreamWaitEvent(la->panel, la->ev_col[k - 1], 0));
        rc = potrf_rec(akk, nbk, ld, dk, clamped, pivot_floor, la->panel);
        NNGP_HIP_CHECK(hipEventRecord(la->ev_panel[k], la->panel));
        if (rc != 0 || m == 0) break;
        // update stream: triangular solve of the rows below, then the trailing update
        NNGP_HIP_CHECK(hipStreamWaitEvent(la->update, la->ev_panel[k], 0));
        const int64_t nb2 = (m < nb) ? m : nb;
        const float* p = akk + nbk * ld;        // panel rows below the diagonal block: [m, nbk]
        float* c = akk + nbk * ld + nbk;        // trailing matrix: [m, m]
        // critical path first: solve only the nb2 panel rows the next diagonal block needs, update that block, and
        // release the panel stream; the remaining panel rows and the rest of the trailing update follow.  Both solves are
        // single fused launches (trsm_panel.hip) that also leave the rows' float16 split copy in this block column's planes
        // (rows at their global index: the trailing update and, later, the posterior's blocked solves read them there).
        const bool fused = NNGP_KNOB(2) != 4 && nbk <= 1024;
        const bool planes_here = h3 && nbk == nb && fused;
        const int64_t ldp = h3 ? 4 * sw->k_cap : 0;
        char* pk_rows = h3 ? sw->planes + (int64_t)k * sw->col_stride + (o + nbk) * ldp : nullptr;  // split copy of row o + nbk
        // (One launch for ALL m rows: a separate launch for the nb2 rows the next diagonal block needs kept 32 workgroups -- an
        // eighth of the GPU -- busy for a whole 0.14 ms workgroup round per block column, 4.4 ms per factorisation at N = 32768;
        // the first workgroups of the merged launch are those rows anyway.)
        // From the second block column on the solve's left-looking products run on the float16 pipe (k_trsm_panel_h3): the
        // diagonal block is split into its own buffer first, in the order the kernel's waves read it (k_split_diag_frag).  (Block column 0 stays float32: same-sign
        // data, see below.  Debug key 2 = 5: float32 everywhere.)
        const bool solve_h3 = fused && planes_here && k > 0 && nbk == 1024 && sw->ldiag != nullptr && sw->dfrag != nullptr && NNGP_KNOB(2) != 5;
        if (solve_h3) {
            rc = launch_split_diag_frag(akk, ld, nbk, sw->scale, sw->ldiag, dk, sw->dfrag, sw->dscale, la->update);
            if (rc == 0) rc = launch_trsm_panel_h3(akk + nbk * ld, ld, m, sw->ldiag, sw->dfrag, sw->dscale, nbk, pk_rows, ldp, sw->scale, la->update);
        } else if (fused)
            rc = launch_trsm_panel_f32(akk + nbk * ld, ld, m, akk, ld, dk, nbk, planes_here ? pk_rows : nullptr, ldp,
                                       h3 ? sw->scale : 1.0f, la->update);
        else
            rc = trsm_rlt_f32(akk + nbk * ld, ld, nb2, akk, ld, dk, nbk, la->update);
        if (rc == 0) rc = launch_gemm_nt_f32(c, ld, p, ld, p, ld, nb2, nb2, nbk, -1.0f, 1.0f, true, la->update);
        NNGP_HIP_CHECK(hipEventRecord(la->ev_col[k], la->update));
        // ... then the other panel rows and the rest of the trailing matrix, overlapped with the next diagonal block
        if (rc == 0 && m > nb2) {
            panel_split_done = false;
            if (fused) {
                panel_split_done = planes_here;  // every row was solved (and split) by the launch above
            } else {
                // Round-1 form (debug key 2 = 4).  The solve of the remaining panel rows, X = B L_kk^-T, splits as
                // X1 = B1 L11^-T, B2 -= X1 L21^T, X2 = B2 L22^-T over the two 512-column halves; the product in the middle runs
                // on the float16 pipe from the second block column on (X1 split straight into the planes, L21 into the planes'
                // unused rows of the diagonal block).
                const bool h3_panel = h3 && k > 0 && nbk == nb && nb == 1024 && m - nb2 >= 2048 && NNGP_KNOB(2) != 3 &&
                                      !(NNGP_KNOB(3) >= 10 && NNGP_KNOB(3) < 20 && k < NNGP_KNOB(3) - 10);
                if (h3_panel) {
                    const int64_t h = nbk / 2, mr = m - nb2;
                    char* pk = sw->planes + (int64_t)k * sw->col_stride;        // planes of block column k, global row 0
                    char* xrows = pk + (o + nbk + nb2) * ldp;                    // rows of the panel being solved
                    float* b = akk + (nbk + nb2) * ld;
                    rc = trsm_rlt_f32(b, ld, mr, akk, ld, dk, h, la->update);                                  // X1
                    if (rc == 0) rc = launch_split_rows(b, ld, mr, h, sw->scale, xrows, ldp, la->update);
                    if (rc == 0) rc = launch_split_rows(akk + h * ld, ld, h, h, sw->scale, pk + (o + h) * ldp, ldp, la->update);  // L21
                    if (rc == 0)
                        rc = launch_gemm_nt_h3(b + h, ld, xrows, pk + (o + h) * ldp, ldp, mr, h, h, -1.0f / (sw->scale * sw->scale), 1.0f,
                                               false, 0, sw->counters, NNGP_KNOB(4) > 0 ? NNGP_KNOB(4) : 32, la->update);
                    if (rc == 0) rc = trsm_rlt_f32(b + h, ld, mr, akk + h * ld + h, ld, dk + (h / TB) * TB * TB, h, la->update);  // X2
                    if (rc == 0) rc = launch_split_rows(b + h, ld, mr, h, sw->scale, xrows + h * 4, ldp, la->update);
                    if (rc == 0)  // the rows solved first (critical path) go into the planes as well
                        rc = launch_split_rows(p, ld, nb2, nbk, sw->scale, pk + (o + nbk) * ldp, ldp, la->update);
                    panel_split_done = true;
                } else {
                    rc = trsm_rlt_f32(akk + (nbk + nb2) * ld, ld, m - nb2, akk, ld, dk, nbk, la->update);
                }
            }
            // The leading columns of the factor are large and of one sign (K is a positive kernel); the float16 MFMA
            // accumulator truncates toward zero, which biases long same-sign sums (-2.6e-8 relative at K = 1024 on
            // positive data, nothing on mixed signs; float32 MFMA: 1e-10).  A coherent error of that size in the first
            // trailing update costs two CG iterations and 40x in the refined variances, so the first `lead` columns
            // of block column 0 go through the float32-MFMA kernels and the rest through the float16 pipe.  Measured at
            // N = 32768 (Cholesky ms / CG iterations / level-2 variance error): lead 0: 59.8 / 7 / 1.6e-5, 128: 61.0 / 6 /
            // 4.1e-7, 256: 61.9 / 5 / 4.0e-7, 512: 63.6 / 6, whole column: 63.6 / 5 / 4.0e-7.  Round 2 (scripts/lead_study.py;
            // level-1 variance against level 3): 0: 47.1 / 7 / 7.5e-5, 32: 47.6 / 6 / 1.16e-6, 64: 47.8 / 5 / 1.13e-6, 128: 48.0 /
            // 5 / 1.13e-6, 256: 48.8 / 6 / 1.17e-6 -- the bias sits in the first few (dominant, one-signed) columns: 64 it is.
            // (debug key 3 = 10 + n: n whole block columns on float32; 20 + c: lead = 128 c; 31 / 32: lead = 32 / 64)
            const bool f32_first = !h3 || (NNGP_KNOB(3) >= 10 && NNGP_KNOB(3) < 20 && k < NNGP_KNOB(3) - 10);
            int64_t lead = 0;
            if (h3 && !f32_first && k == 0) lead = (NNGP_KNOB(3) >= 20 && NNGP_KNOB(3) < 28) ? 128 * (int64_t)(NNGP_KNOB(3) - 20) :
                                                  (NNGP_KNOB(3) == 31 || NNGP_KNOB(3) == 32) ? 32 * (int64_t)(NNGP_KNOB(3) - 30) : 64;
            if (lead > nbk - 128) lead = 0;
            if (!f32_first) {
                // the persistent GEMM grid leaves `reserve` compute units to the panel stream, whose small kernels
                // otherwise queue behind 128-KB-LDS workgroups (measured at N = 32768: 0/8/16 -> 62.7 ms, 32 -> 58.7,
                // 24 -> 65.0, 48 -> 58.8, 64 -> 60.4; debug key 4 overrides)
                const int reserve = NNGP_KNOB(4) > 0 ? NNGP_KNOB(4) : 32;
                // one launch: rows [nb2, m) x columns [0, m) of the trailing matrix, on or below its diagonal
                // (the split copy of block column k stays in place, rows at their global index: the blocked triangular
                // solves of the posterior read it again)
                char* planes = pk_rows;
                if (rc == 0 && !panel_split_done) rc = launch_split_rows(p, ld, m, nbk, sw->scale, planes, ldp, la->update);
                const bool timed = la->time_updates && rc == 0 && la->tu_count < LookAhead::kMaxTimed;
                if (timed) {  // events are created on first use (timing stream: the one the kernel is launched on)
                    const int t = la->tu_count;
                    if (la->tu0[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&la->tu0[t]));
                    if (la->tu1[t] == nullptr) NNGP_HIP_CHECK(hipEventCreate(&la->tu1[t]));
                    NNGP_HIP_CHECK(hipEventRecord(la->tu0[t], la->update));
                }
                if (rc == 0)  // columns [lead, nbk) of the panel (K blocks are walked from the high end down)
                    rc = launch_gemm_nt_h3(c + nb2 * ld, ld, planes + nb2 * ldp + lead * 4, planes + lead * 4, ldp, m - nb2, m,
                                           nbk - lead, -1.0f / (sw->scale * sw->scale), 1.0f, true, nb2, sw->counters, reserve,
                                           la->update);
                if (timed) {
                    const int t = la->tu_count++;
                    NNGP_HIP_CHECK(hipEventRecord(la->tu1[t], la->update));
                    // entries on or below the diagonal in rows [nb2, m) of the m x m trailing matrix
                    la->tu_flops[t] = 2.0 * (0.5 * ((double)m * (m + 1) - (double)nb2 * (nb2 + 1))) * (double)(nbk - lead);
                    la->tu_bytes[t] = 8.0 * (0.5 * ((double)m * (m + 1) - (double)nb2 * (nb2 + 1))) + 4.0 * (2.0 * (double)m - (double)nb2) * (double)(nbk - lead);
                }
                if (rc == 0 && lead > 0)
                    rc = launch_gemm_nt_f32(c + nb2 * ld, ld, p + nb2 * ld, ld, p, ld, m - nb2, nb2, lead, -1.0f, 1.0f, false, la->update);
                if (rc == 0 && lead > 0)
                    rc = launch_gemm_nt_f32(c + nb2 * ld + nb2, ld, p + nb2 * ld, ld, p + nb2 * ld, ld, m - nb2, m - nb2, lead,
                                            -1.0f, 1.0f, true, la->update);
            } else {
                if (h3 && rc == 0 && !panel_split_done)  // the split copy is still needed by the posterior solves
                    rc = launch_split_rows(p, ld, m, nbk, sw->scale, pk_rows, ldp, la->update);
                if (rc == 0)
                    rc = launch_gemm_nt_f32(c + nb2 * ld, ld, p + nb2 * ld, ld, p, ld, m - nb2, nb2, nbk, -1.0f, 1.0f, false, la->update);
                if (rc == 0)
                    rc = launch_gemm_nt_f32(c + nb2 * ld + nb2, ld, p + nb2 * ld, ld, p + nb2 * ld, ld, m - nb2, m - nb2, nbk,
                                            -1.0f, 1.0f, true, la->update);
            }
        } else if (rc == 0 && h3 && nbk == nb && !planes_here) {  // last panel: no trailing update left, but keep its split copy complete
            rc = launch_split_rows(p, ld, m, nbk, sw->scale, pk_rows, ldp, la->update);
        }
    }
    if (rc == 0 && h3) sw->l_ready = true;
    NNGP_HIP_CHECK(hipEventRecord(la->ev_panel_done, la->panel));
    NNGP_HIP_CHECK(hipEventRecord(la->ev_update_done, la->update));
    NNGP_HIP_CHECK(hipStreamWaitEvent(user, la->ev_panel_done, 0));
    NNGP_HIP_CHECK(hipStreamWaitEvent(user, la->ev_update_done, 0));
    return rc;
}

// ---- block-column pieces of the right-looking factorisation (multi-GPU: block columns are dealt cyclically) ----
// Factor block column [o, o+w): Cholesky of the diagonal block, then the rows below times its inverse transpose.
int potrf_panel_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor, int64_t o,
                    int64_t w, hipStream_t s, SplitWork* sw) {
    NNGP_REQUIRE(o >= 0 && w > 0 && o + w <= n && o % TB == 0 && w % TB == 0, "potrf_panel: bad block column");
    float* akk = a + o * ld + o;
    float* dk = dinv + (o / TB) * TB * TB;
    NNGP_TRY(potrf_rec(akk, w, ld, dk, clamped, pivot_floor, s));
    const int64_t m = n - o - w;
    if (m <= 0) return 0;
    // rows below: ONE fused launch (trsm_panel.hip) as in the single-GPU look-ahead, on the float16 pipe from the second block
    // column on; with the model's split workspace it also leaves the rows' split copy in place (the owner's own updates of this
    // block column then need no separate split pass).  Round 2 ran the 128-wide GEMM recursion here (15 dependent launches).
    const bool planes_ok = sw != nullptr && sw->planes != nullptr && w == sw->k_cap && o % w == 0 && sw->rows_cap >= n + 256 &&
                           sw->col_stride >= sw->rows_cap * 4 * sw->k_cap && NNGP_KNOB(2) != 2;
    if (w <= 1024 && NNGP_KNOB(2) != 4) {
        const int64_t ldp = planes_ok ? 4 * sw->k_cap : 0;
        char* rows = planes_ok ? sw->planes + (o / w) * sw->col_stride + (o + w) * ldp : nullptr;
        if (planes_ok && o > 0 && w == 1024 && sw->ldiag != nullptr && sw->dfrag != nullptr && NNGP_KNOB(2) != 5) {
            NNGP_TRY(launch_split_diag_frag(akk, ld, w, sw->scale, sw->ldiag, dk, sw->dfrag, sw->dscale, s));
            NNGP_TRY(launch_trsm_panel_h3(akk + w * ld, ld, m, sw->ldiag, sw->dfrag, sw->dscale, w, rows, ldp, sw->scale, s));
        } else {
            NNGP_TRY(launch_trsm_panel_f32(akk + w * ld, ld, m, akk, ld, dk, w, rows, ldp, planes_ok ? sw->scale : 1.0f, s));
        }
        if (planes_ok) sw->split_panel = o;
        return 0;
    }
    return trsm_rlt_f32(akk + w * ld, ld, m, akk, ld, dk, w, s);
}

// Apply the finished block column [po, po+pw) to block column [o, o+w), o >= po + pw:
//   A[o:, o:o+w] -= L[o:, po:po+pw] L[o:o+w, po:po+pw]^T   (lower part of the diagonal block only)
int potrf_update_f32(float* a, int64_t n, int64_t ld, int64_t po, int64_t pw, int64_t o, int64_t w, hipStream_t s,
                     SplitWork* sw) {
    NNGP_REQUIRE(po >= 0 && pw > 0 && o >= po + pw && w > 0 && o + w <= n && o % TB == 0 && w % TB == 0 && po % TB == 0 &&
                     pw % TB == 0, "potrf_update: bad block columns");
    const float* p = a + o * ld + po;   // panel rows [o, n), columns [po, po+pw)
    float* c = a + o * ld + o;
    // float16 pipe (same split copies as the single-GPU look-ahead, so the posterior solves find them afterwards): the
    // panel is split once, when its first update arrives; the leading columns of the first panel stay on the float32
    // MFMA (accumulator truncation on same-sign sums, see potrf_lookahead_f32)
    if (sw != nullptr && sw->planes != nullptr && sw->counters != nullptr && pw == sw->k_cap && po % pw == 0 &&
        sw->rows_cap >= n + 256 && NNGP_KNOB(2) != 2 && (n - o) * w >= 96 * 256 * 256) {
        const int64_t ldp = 4 * sw->k_cap;
        char* col = sw->planes + (po / pw) * sw->col_stride;  // rows at their global index
        if (sw->split_panel != po) {
            NNGP_TRY(launch_split_rows(a + (po + pw) * ld + po, ld, n - po - pw, pw, sw->scale, col + (po + pw) * ldp, ldp, s));
            sw->split_panel = po;
        }
        const int64_t lead = (po == 0 && pw > 256) ? 64 : 0;  // as in potrf_lookahead_f32
        NNGP_TRY(launch_gemm_nt_h3(c, ld, col + o * ldp + lead * 4, col + o * ldp + lead * 4, ldp, n - o, w, pw - lead,
                                   -1.0f / (sw->scale * sw->scale), 1.0f, true, 0, sw->counters, 0, s));
        if (lead > 0) {
            NNGP_TRY(launch_gemm_nt_f32(c, ld, p, ld, p, ld, w, w, lead, -1.0f, 1.0f, true, s));
            if (n - o - w > 0)
                NNGP_TRY(launch_gemm_nt_f32(c + w * ld, ld, p + w * ld, ld, p, ld, n - o - w, w, lead, -1.0f, 1.0f, false, s));
        }
        return 0;
    }
    NNGP_TRY(launch_gemm_nt_f32(c, ld, p, ld, p, ld, w, w, pw, -1.0f, 1.0f, true, s));
    const int64_t below = n - o - w;
    if (below > 0)
        NNGP_TRY(launch_gemm_nt_f32(c + w * ld, ld, p + w * ld, ld, p, ld, below, w, pw, -1.0f, 1.0f, false, s));
    return 0;
}

// The same for SEVERAL target block columns of one rank (multi-GPU: the block columns a rank owns, dealt cyclically): the columns
// that qualify for the float16 pipe go out four to a launch (regions of one split-float16 pass: fewer launches, fewer tails), the
// others one by one as above.  cols: first rows / columns of the targets, ascending, all of width w except possibly the last.
int potrf_update_cols_f32(float* a, int64_t n, int64_t ld, int64_t po, int64_t pw, const int64_t* cols, int ncols, int64_t w,
                          hipStream_t s, SplitWork* sw) {
    NNGP_REQUIRE(cols != nullptr && ncols >= 0 && w > 0 && w % TB == 0, "potrf_update_cols: bad arguments");
    const bool h3 = sw != nullptr && sw->planes != nullptr && sw->counters != nullptr && pw == sw->k_cap && po % pw == 0 &&
                    sw->rows_cap >= n + 256 && NNGP_KNOB(2) != 2 && ld % 4 == 0;
    H3RegionSpec reg[4];
    int nreg = 0;
    const int64_t lead = (po == 0 && pw > 256) ? 64 : 0;
    auto flush = [&]() -> int {
        if (nreg == 0) return 0;
        const int64_t ldp = 4 * sw->k_cap;
        char* col = sw->planes + (po / pw) * sw->col_stride;  // rows at their global index
        if (sw->split_panel != po) {
            NNGP_TRY(launch_split_rows(a + (po + pw) * ld + po, ld, n - po - pw, pw, sw->scale, col + (po + pw) * ldp, ldp, s));
            sw->split_panel = po;
        }
        NNGP_TRY(launch_gemm_nt_h3r(a, ld, col + lead * 4, col + lead * 4, ldp, 0, 1, 0, reg, nreg, pw - lead, -1.0f / (sw->scale * sw->scale),
                                    1.0f, true, sw->counters, 0, s));
        for (int r = 0; r < nreg && lead > 0; ++r) {
            const float* p = a + reg[r].row0 * ld + po;
            float* c = a + reg[r].row0 * ld + reg[r].col0;
            NNGP_TRY(launch_gemm_nt_f32(c, ld, p, ld, p, ld, reg[r].n, reg[r].n, lead, -1.0f, 1.0f, true, s));
            if (reg[r].m > reg[r].n)
                NNGP_TRY(launch_gemm_nt_f32(c + reg[r].n * ld, ld, p + reg[r].n * ld, ld, p, ld, reg[r].m - reg[r].n, reg[r].n, lead, -1.0f, 1.0f,
                                            false, s));
        }
        nreg = 0;
        return 0;
    };
    for (int i = 0; i < ncols; ++i) {
        const int64_t o = cols[i];
        const int64_t wi = (n - o < w) ? n - o : w;
        NNGP_REQUIRE(o >= po + pw && o % TB == 0 && wi > 0, "potrf_update_cols: bad target column");
        if (h3 && (n - o) * wi >= 96 * 256 * 256) {
            reg[nreg++] = H3RegionSpec{o, o, n - o, wi, 0};
            if (nreg == 4) NNGP_TRY(flush());
        } else {
            NNGP_TRY(flush());
            NNGP_TRY(potrf_update_f32(a, n, ld, po, pw, o, wi, s, sw));
        }
    }
    return flush();
}

int potrf_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor, hipStream_t s) {
    NNGP_REQUIRE(n > 0 && n % TB == 0, "potrf_f32: n must be a positive multiple of %d (got %lld)", TB, (long long)n);
    NNGP_REQUIRE(ld >= n && ld % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)dinv & 15) == 0,
                 "potrf_f32: matrix must be 16-byte aligned with ld >= n");
    return potrf_rec(a, n, ld, dinv, clamped, pivot_floor, s);
}

}  // namespace nngp
