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
#include "common.h"

namespace nngp {

int g_debug[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // timing experiments only (nngp_debug_set); 0 = product behaviour

namespace {

constexpr int LS = 129;  // LDS row stride (floats)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// acc += sign * A_blk(32x32) * op(B_blk);  NN: op(B) = B,  NT: op(B) = B^T.  Blocks live in LDS, stride LS.
template <bool NN>
__device__ __forceinline__ f32x16 blk_mma(const float* ab, const float* bb, f32x16 acc, float sign, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = 2 * s + h;
        const float a = sign * ab[r * LS + k];
        const float b = NN ? bb[k * LS + r] : bb[r * LS + k];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
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

__global__ __launch_bounds__(256) void k_potrf_leaf(float* A, int64_t ld, float* dinv, int32_t* clamped,
                                                    float pivot_floor, int dbg) {
    __shared__ float Ls[128 * LS];
    __shared__ float Xs[128 * LS];
    __shared__ float colbuf[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    {
        // 16 independent 16-byte loads per thread, issued back to back (one HBM round trip for the whole block)
        float4 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int idx = (tid + 256 * e) * 4;
            const int r = idx >> 7, c = idx & 127;
            v[e] = *reinterpret_cast<const float4*>(A + (int64_t)r * ld + c);  // upper entries are masked below
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int idx = (tid + 256 * e) * 4;
            const int r = idx >> 7, c = idx & 127;
            float* lp = Ls + r * LS + c;
            float* xp = Xs + r * LS + c;
            lp[0] = (c <= r) ? v[e].x : 0.0f;
            lp[1] = (c + 1 <= r) ? v[e].y : 0.0f;
            lp[2] = (c + 2 <= r) ? v[e].z : 0.0f;
            lp[3] = (c + 3 <= r) ? v[e].w : 0.0f;
            xp[0] = xp[1] = xp[2] = xp[3] = 0.0f;
        }
    }
    __syncthreads();

    int nclamp = 0;
    for (int jb = 0; jb < 4; ++jb) {
        float* Djj = Ls + (jb * 32) * LS + jb * 32;
        float* Xjj = Xs + (jb * 32) * LS + jb * 32;
        if (wave == 0 && !(dbg & 1)) {
            // One wave factors and inverts the 32x32 diagonal sub-block.  Lane i owns row i in registers; the
            // values every lane needs (pivot, column j) travel through a 32-float LDS line read back as
            // broadcasts -- an LDS round trip per column instead of ~500 v_readlane + hazard nops.
            const int i = lane & 31;
            float a[32], x[32], invd[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k] = Djj[i * LS + k];
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                float* col = colbuf + (j & 1) * 32;
                if (lane < 32) col[i] = a[j];  // current column j (rows >= j are valid)
                float d = col[j];
                asm volatile("" : "+v"(d));  // keep wave-uniform values in VGPRs (no SGPR spills / readlane traffic)
                if (!(d > pivot_floor)) {
                    d = pivot_floor > 0.0f ? pivot_floor : 1.0e-30f;
                    ++nclamp;
                }
                float inv = __builtin_amdgcn_rsqf(d);  // v_rsq_f32, ~1 ulp: ample for a preconditioner
                asm volatile("" : "+v"(inv));
                invd[j] = inv;
                const float lij = a[j] * inv;                // L[i][j] for i > j
                a[j] = (i == j) ? d * inv : lij;
#pragma unroll
                for (int k = j + 1; k < 32; ++k) a[k] = fmaf(-lij, col[k] * inv, a[k]);  // a[i][k] -= L[i][j] L[k][j]
            }
            if (lane < 32) {
#pragma unroll
                for (int k = 0; k < 32; ++k) Djj[i * LS + k] = (k <= i) ? a[k] : 0.0f;
            }
            // inverse: lane c owns column c of X = L^-1;  X[ii][c] = (delta - sum_{k<ii} L[ii][k] X[k][c]) / L[ii][ii]
#pragma unroll
            for (int ii = 0; ii < 32; ++ii) {
                float sacc = 0.0f;
#pragma unroll
                for (int k = 0; k < ii; ++k) sacc = fmaf(Djj[ii * LS + k], x[k], sacc);  // broadcast LDS reads
                x[ii] = (((i == ii) ? 1.0f : 0.0f) - sacc) * invd[ii];
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
                float* Aij = Ls + (ib * 32) * LS + jb * 32;
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
                    float* Cik = Ls + (ib * 32) * LS + kb * 32;
                    f32x16 acc = blk_load(Cik, lane);
                    acc = blk_mma<false>(Ls + (ib * 32) * LS + jb * 32, Ls + (kb * 32) * LS + jb * 32, acc, -1.0f, lane);
                    blk_store(Cik, acc, lane);
                }
        }
        __syncthreads();
    }

    // ---- assemble the 128x128 inverse: X[ib][jb] = -Dinv_ii * sum_{k=jb}^{ib-1} L[ib][k] X[k][jb] ----
    for (int dist = 1; dist < ((dbg & 2) ? 1 : 4); ++dist) {
        const int ib = dist + wave, jb = wave;  // wave w owns block (dist + w, w)
        const bool active = ib < 4;
        float* scratch = Xs + (jb * 32) * LS + ib * 32;  // the (zero) upper block (jb, ib), valid when active
        if (active) {
            f32x16 t = zero16();
            for (int k = jb; k < ib; ++k)
                t = blk_mma<true>(Ls + (ib * 32) * LS + k * 32, Xs + (k * 32) * LS + jb * 32, t, 1.0f, lane);
            blk_store(scratch, t, lane);
        }
        __syncthreads();
        f32x16 xr = zero16();
        if (active) xr = blk_mma<true>(Xs + (ib * 32) * LS + ib * 32, scratch, zero16(), -1.0f, lane);
        __syncthreads();
        if (active) {
            blk_store(Xs + (ib * 32) * LS + jb * 32, xr, lane);
            blk_store(scratch, zero16(), lane);
        }
        __syncthreads();
    }

#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int idx = (tid + 256 * e) * 4;
        const int r = idx >> 7, c = idx & 127;
        const float* lp = Ls + r * LS + c;
        const float* xp = Xs + r * LS + c;
        if (c + 3 <= r) {
            *reinterpret_cast<float4*>(A + (int64_t)r * ld + c) = make_float4(lp[0], lp[1], lp[2], lp[3]);
        } else if (c <= r) {  // the 4-wide group straddles the diagonal: keep the caller's upper entries
            for (int k = 0; k < 4; ++k)
                if (c + k <= r) A[(int64_t)r * ld + c + k] = lp[k];
        }
        *reinterpret_cast<float4*>(dinv + idx) = make_float4(xp[0], xp[1], xp[2], xp[3]);
    }
    if (wave == 0 && lane == 0 && nclamp > 0 && clamped != nullptr) atomicAdd(clamped, nclamp);
}

}  // namespace

int launch_potrf_leaf(float* a, int64_t ld, float* dinv_block, int32_t* clamped, float pivot_floor, hipStream_t s) {
    hipLaunchKernelGGL(k_potrf_leaf, dim3(1), dim3(256), 0, s, a, ld, dinv_block, clamped, pivot_floor, g_debug[0]);
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

int potrf_f32(float* a, int64_t n, int64_t ld, float* dinv, int32_t* clamped, float pivot_floor, hipStream_t s) {
    NNGP_REQUIRE(n > 0 && n % TB == 0, "potrf_f32: n must be a positive multiple of %d (got %lld)", TB, (long long)n);
    NNGP_REQUIRE(ld >= n && ld % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)dinv & 15) == 0,
                 "potrf_f32: matrix must be 16-byte aligned with ld >= n");
    return potrf_rec(a, n, ld, dinv, clamped, pivot_floor, s);
}

}  // namespace nngp
