// float64 "NT" GEMM on the gfx950 matrix cores:  C[M,N] = beta*Cin + alpha * A[M,K] * B[N,K]^T.
//
// Used by the float64 refinement of the posterior covariance (a4, reference train.py:157-158): the
// residual R = K_td - Z (K_dd + reg I) is an [M, N] x [N, N] product that must be float64 because the
// float32 Cholesky factor only approximates (K + reg I)^-1 to ~cond * eps32.  K_dd is
// symmetric, so its rows serve as the K-contiguous "B" operand and no transposed copy is needed.
//
// Same skeleton as gemm_f32.hip: 128x128 tile, 4 waves (2x2), 64x64 per wave = 4x4 accumulators of
// v_mfma_f64_16x16x4_f64 (128 VGPRs), BK = 16 doubles (128-byte LDS rows, same XOR swizzle), double-buffered
// register staging.  Lane l holds A[i = l & 15][k = l >> 4]; a 16-byte LDS read gives it two consecutive k,
// used by two MFMAs.  C/D layout of the f64 MFMA: col = l & 15, row = (l >> 4) + 4 * reg  (it differs from
// the f32 shapes -- cdna_hip_programming.md section 3).
#include "common.h"

namespace nngp {

namespace {

constexpr int DBM = 128, DBN = 128, DBK = 16;
constexpr int DSTAGE = (DBM + DBN) * DBK;  // doubles per stage (32 KiB)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lds_off64(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256, 2) void k_gemm_nt_f64(double* C, int64_t ldc, const double* Cin, int64_t ldcin,
                                                        const double* A, int64_t lda, const double* B, int64_t ldb,
                                                        int tiles_m, int nk, double alpha, double beta, int kmode,
                                                        int tiles_n, int ksplit, double* part, int64_t part_stride, int xcd_order) {
    __shared__ __attribute__((aligned(16))) double smem[2 * DSTAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Row tile is the fast index: the workgroups that share one 128-row panel of B (the big symmetric kernel matrix) are
    // neighbours in tile order.  Workgroup b runs on XCD b % 8, so XCD x takes a CONTIGUOUS range of that order: the
    // panel's sharers sit in one XCD's L2 and the panel leaves HBM once (with plain blockIdx order its tiles_m sharers
    // were dealt to tiles_m different XCDs and each fetched it: 118 GB per posterior instead of the matrix's 8.6).
    int g_ = blockIdx.x;
    if (xcd_order) {
        const int total = tiles_m * tiles_n * (ksplit > 1 ? ksplit : 1);
        const int q = total >> 3, r = total & 7, x = g_ & 7, slot = g_ >> 3;
        g_ = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
    }
    const int tile = g_ % (tiles_m * tiles_n), kp = g_ / (tiles_m * tiles_n);
    const int bi = tile % tiles_m;
    int bj = tile / tiles_m;
    // kmode (B square and symmetric, quadratic forms z^T B z from its lower triangle only): 1 = only the k blocks
    // strictly left of the column tile's own 128-block, 2 = only that diagonal block.  The long tiles go first.
    int t0 = 0, t1 = nk;
    if (kmode == 1) {
        bj = tiles_n - 1 - bj;
        t1 = bj * (DBN / DBK);
    } else if (kmode == 2) {
        t0 = bj * (DBN / DBK);
        t1 = t0 + DBN / DBK;
    }
    if (ksplit > 1) {  // split K: this workgroup's share of [t0, t1); its tile goes to part[kp] (summed by k_splitk_reduce)
        const int len = (t1 - t0 + ksplit - 1) / ksplit;
        t0 += kp * len;
        if (t0 + len < t1) t1 = t0 + len;
    }
    const double* Ab = A + (int64_t)bi * DBM * lda;
    const double* Bb = B + (int64_t)((xcd_order & 2) ? (bj & 7) : bj) * DBN * ldb;  // (bit 1: timing diagnostic, B aliased to 8 panels: cache-hot operands, wrong result)

    f64x2 ga[4], gb[4];
    const int ld_row = tid >> 3, ld_ch = tid & 7;
    auto load_tile = [&](int t) {
        const int64_t k0 = (int64_t)t * DBK + ld_ch * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ga[e] = *reinterpret_cast<const f64x2*>(Ab + (int64_t)(ld_row + 32 * e) * lda + k0);
            gb[e] = *reinterpret_cast<const f64x2*>(Bb + (int64_t)(ld_row + 32 * e) * ldb + k0);
        }
    };
    auto store_tile = [&](int buf) {
        char* sa_ = reinterpret_cast<char*>(smem + buf * DSTAGE);
        char* sb_ = sa_ + DBM * DBK * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<f64x2*>(sa_ + lds_off64(ld_row + 32 * e, ld_ch)) = ga[e];
            *reinterpret_cast<f64x2*>(sb_ + lds_off64(ld_row + 32 * e, ld_ch)) = gb[e];
        }
    };

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

    const int r16 = lane & 15, g = lane >> 4;

    if (t0 < t1) {
        load_tile(t0);
        store_tile(t0 & 1);
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) load_tile(t + 1);
        const char* sa_ = reinterpret_cast<const char*>(smem + (t & 1) * DSTAGE);
        const char* sb_ = sa_ + DBM * DBK * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f64x2 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *reinterpret_cast<const f64x2*>(sa_ + lds_off64(wm * 64 + i * 16 + r16, 4 * s + g));
                fb[i] = *reinterpret_cast<const f64x2*>(sb_ + lds_off64(wn * 64 + i * 16 + r16, 4 * s + g));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < t1) store_tile((t + 1) & 1);
        __syncthreads();
    }

    const int64_t row_base = (int64_t)bi * DBM + wm * 64;
    const int64_t col_base = (int64_t)bj * DBN + wn * 64;
    if (ksplit > 1) {
        double* P = part + (int64_t)kp * part_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    P[(row_base + i * 16 + g + 4 * r) * ldc + col_base + j * 16 + r16] = acc[i][j][r];
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t row0 = row_base + i * 16 + g, col = col_base + j * 16 + r16;
            double cin[4];
            if (beta != 0.0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) cin[r] = Cin[(row0 + 4 * r) * ldcin + col];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = alpha * acc[i][j][r];
                if (beta != 0.0) v = fma(beta, cin[r], v);
                C[(row0 + 4 * r) * ldc + col] = v;
            }
        }
}

// C = beta * Cin + alpha * sum_p part[p]   (fixed summation order: the split-K product is deterministic)
__global__ __launch_bounds__(256) void k_splitk_reduce(double* __restrict__ C, int64_t ldc, const double* __restrict__ Cin,
                                                       int64_t ldcin, const double* __restrict__ part, int64_t part_stride,
                                                       int ksplit, int64_t n, double alpha, double beta) {
    const int64_t row = blockIdx.y;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    double v = 0.0;
    for (int p = 0; p < ksplit; ++p) v += part[(int64_t)p * part_stride + row * ldc + c];
    v *= alpha;
    if (beta != 0.0) v = fma(beta, Cin[row * ldcin + c], v);
    C[row * ldc + c] = v;
}

}  // namespace

int launch_gemm_nt_f64(double* c, int64_t ldc, const double* cin, int64_t ldcin, const double* a, int64_t lda,
                       const double* b, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta,
                       hipStream_t s, int kmode) {
    if (m <= 0 || n <= 0) return 0;
    NNGP_REQUIRE(kmode == 0 || (kmode >= 1 && kmode <= 2 && n == k), "gemm_nt_f64: the triangular modes need a square B");
    NNGP_REQUIRE(m % DBM == 0 && n % DBN == 0 && k % DBK == 0 && k > 0,
                 "gemm_nt_f64: m, n must be multiples of 128 and k of 16 (m=%lld n=%lld k=%lld)", (long long)m,
                 (long long)n, (long long)k);
    NNGP_REQUIRE(lda % 2 == 0 && ldb % 2 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0,
                 "gemm_nt_f64: operands must be 16-byte aligned");
    NNGP_REQUIRE(lda >= k && ldb >= k && ldc >= n && (beta == 0.0 || (cin != nullptr && ldcin >= n)),
                 "gemm_nt_f64: leading dimension too small");
    const int64_t tm = m / DBM, tn = n / DBN;
    NNGP_REQUIRE(tm * tn < 2147483647LL / 8, "gemm_nt_f64: grid too large");
    // Few tiles and a long K (a small block of queries against the N x N kernel): split K over workgroups so that the
    // 256 compute units (two workgroups each) have work; partial tiles go to a stream-ordered scratch buffer and are
    // summed in a fixed order.  Measured (serving mode, 128 queries, N = 10800): see DESIGN.md section 7.
    int ksplit = 1;
    if (kmode != 2 && tm * tn <= 256 && k / DBK >= 128 && NNGP_KNOB(5) != 9) {
        ksplit = (int)(512 / (tm * tn));
        if (ksplit > 8) ksplit = 8;
        while (ksplit > 1 && (k / DBK) / ksplit < 32) --ksplit;
    }
    if (ksplit <= 1) {
        hipLaunchKernelGGL(k_gemm_nt_f64, dim3((unsigned)(tm * tn)), dim3(256), 0, s, c, ldc, cin ? cin : c, ldcin ? ldcin : ldc,
                           a, lda, b, ldb, (int)tm, (int)(k / DBK), alpha, beta, kmode, (int)tn, 1, nullptr, 0, (NNGP_KNOB(5) != 8 ? 1 : 0) | (NNGP_KNOB(5) == 7 ? 2 : 0));
        NNGP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int64_t part_stride = m * ldc;
    double* part = nullptr;
    NNGP_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&part), sizeof(double) * part_stride * ksplit, s));
    if (kmode == 1)  // tiles whose k range is empty or short leave (parts of) their slots unwritten
        NNGP_HIP_CHECK(hipMemsetAsync(part, 0, sizeof(double) * part_stride * ksplit, s));
    hipLaunchKernelGGL(k_gemm_nt_f64, dim3((unsigned)(tm * tn * ksplit)), dim3(256), 0, s, c, ldc, cin ? cin : c,
                       ldcin ? ldcin : ldc, a, lda, b, ldb, (int)tm, (int)(k / DBK), alpha, beta, kmode, (int)tn, ksplit, part,
                       part_stride, NNGP_KNOB(5) != 8);
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, s, c, ldc, cin ? cin : c,
                       ldcin ? ldcin : ldc, part, part_stride, ksplit, n, alpha, beta);
    NNGP_HIP_CHECK(hipGetLastError());
    NNGP_HIP_CHECK(hipFreeAsync(part, s));
    return 0;
}

}  // namespace nngp
