// K1: fused NNGP/NTK kernel build for gfx950.
//
// Replaces kernel_fn of stax.serial(Dense, Relu, Dense) (reference train.py:161-164,
// estimator.py:27-30; closed forms in SURVEY.md 8a row a1).  One 64x64 output tile per
// 256-thread workgroup: the x1 / x2 row panels are staged through LDS, the Gram entries
// x.x'/d are accumulated in float64 registers (4x4 per thread) and the whole layer recursion
// (Dense affine, ReLU arc-cosine map with sqrt + atan2, NTK chain) runs in the epilogue on the
// accumulators, so K (and Theta) are written to HBM exactly once.  In symmetric mode only tiles on
// or below the diagonal are computed; the mirror image is transposed through LDS and written with
// the same coalesced 16-byte stores.
//
// Everything is float64: the GP solve behind this kernel has cond ~ 1e7, and a float32 Gram
// perturbs posterior means by ~2.5e-4, over the 1e-4 parity gate.
#include "common.h"
#include "trig_tab.h"
#include <math.h>
#include <mutex>
#include <vector>

namespace nngp {

namespace {

constexpr int KT = 64;       // output tile edge
constexpr int KC = 16;       // k-chunk staged per iteration
constexpr int LDP = KT + 2;  // LDS row stride in doubles (keeps 16-byte alignment of every row)
constexpr int LDM = KT + 16; // staging stride of the MFMA variant: 2*LDM = 32 (mod 64) dwords, so the four k-rows a
                             // v_mfma_f64_16x16x4 fragment read touches fall on disjoint bank halves (conflict free)
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr double kPi = 3.14159265358979323846;
constexpr int kCompNI = 16, kCompDeg = 10;                       // intervals of [0, pi] and polynomial degree of the composite map
constexpr int kCompSize = kCompNI * (kCompDeg + 1) + 1;          // + the amplitude A

__global__ __launch_bounds__(256) void k_row_sqnorm(const double* __restrict__ x, int64_t n, int d,
                                                    double* __restrict__ q) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= n) return;
    const double* xr = x + row * (int64_t)d;
    double s = 0.0;
    for (int k = lane; k < d; k += 64) {
        const double v = xr[k];
        s = fma(v, v, s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) q[row] = s / (double)d;
}

__global__ __launch_bounds__(256) void k_diag_from_q(const double* __restrict__ q, int64_t n, ArchDev arch,
                                                     double* __restrict__ dn, double* __restrict__ dt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double k = q[i], t = 0.0;
    for (int l = 0; l < arch.n_dense; ++l) {
        k = fma(arch.w2[l], k, arch.b2[l]);
        t = fma(arch.w2[l], t, k);
        if (l < arch.n_dense - 1) {
            k *= 0.5;  // s = 0, theta = 0 on the diagonal
            t *= 0.5;
        }
    }
    if (dn) dn[i] = k;
    if (dt) dt[i] = t;
}

// One matrix element through Dense,(Relu,Dense)*.  exact_diag: element (i, i) of a symmetric build,
// where q q' - k^2 == 0 exactly (k is replaced by q so rounding of the dot product cannot leak in).
__device__ __forceinline__ void layer_map(double k, double q1, double q2, const ArchDev& arch, bool exact_diag,
                                          double& out_k, double& out_t) {
    double t = 0.0;
    if (exact_diag) k = q1;
    for (int l = 0; l < arch.n_dense; ++l) {
        const double w2 = arch.w2[l], b2 = arch.b2[l];
        k = fma(w2, k, b2);
        q1 = fma(w2, q1, b2);
        q2 = fma(w2, q2, b2);
        t = fma(w2, t, k);
        if (l < arch.n_dense - 1) {
            if (exact_diag) {
                k *= 0.5;
                t *= 0.5;
            } else {
                const double r = fma(q1, q2, -k * k);
                const double s = r > 0.0 ? sqrt(r) : 0.0;
                const double th = (s == 0.0 && k == 0.0) ? 0.5 * kPi : atan2(s, k);
                const double kd = (kPi - th) * (0.5 / kPi);
                k = fma(kd, k, s * (0.5 / kPi));
                t *= kd;
            }
            q1 *= 0.5;
            q2 *= 0.5;
        }
    }
    out_k = k;
    out_t = t;
}

template <typename T>
__device__ __forceinline__ void store4(T* base, int64_t ld, int64_t i, int64_t j, int64_t i_end, int64_t j_end,
                                       const double v[4], bool vec_ok) {
    if (base == nullptr || i >= i_end) return;
    T* p = base + i * ld + j;
    if (vec_ok && j + 3 < j_end) {
        if constexpr (sizeof(T) == 8) {
            reinterpret_cast<double2*>(p)[0] = make_double2(v[0], v[1]);
            reinterpret_cast<double2*>(p)[1] = make_double2(v[2], v[3]);
        } else {
            reinterpret_cast<float4*>(p)[0] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (j + c < j_end) p[c] = (T)v[c];
    }
}

// MFMA = true: the Gram entries x.x' are accumulated on the float64 matrix cores (v_mfma_f64_16x16x4_f64, 2x2 tiles of
// 16x16 per wave, 4 waves = the 64x64 tile) and handed to the epilogue's 4x4-per-thread layout through LDS;
// MFMA = false: the same sums on the float64 VALU (the default; the MFMA form is selected by nngp_debug_set(3, 3)).
template <bool MFMA>
__global__ __launch_bounds__(256) void k_build(BuildArgs a, ArchDev arch, int64_t tiles_c, int vec_ok) {
    __shared__ __attribute__((aligned(16))) double smem[KT * LDP];  // As | Bs during the k-loop, T for the mirror
    constexpr int LDS_ = MFMA ? LDM : LDP;
    double* As = smem;              // [KC][LDS_]
    double* Bs = smem + KC * LDS_;  // [KC][LDS_]

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;

    int64_t bi, bj;
    if (a.sym) {
        const int64_t p = blockIdx.x;
        bi = (int64_t)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
        while (bi * (bi + 1) / 2 > p) --bi;
        while ((bi + 1) * (bi + 2) / 2 <= p) ++bi;
        bj = p - bi * (bi + 1) / 2;
    } else {
        bi = blockIdx.x / tiles_c;
        bj = blockIdx.x % tiles_c;
    }
    const int64_t i0 = a.row_begin + bi * KT, j0 = bj * KT;
    const int64_t i_end = a.row_end, j_end = a.n2;

    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;

    f64x4 macc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) macc[i][j][r] = 0.0;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, lg = lane >> 4;

    // Software pipeline: the global loads of chunk c+1 are issued into registers before chunk c is consumed from LDS,
    // so their L2 / Infinity Cache latency overlaps the FMAs instead of preceding them.
    double ra[4], rb[4];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e;
            const int kk = idx & (KC - 1), row = idx >> 4;
            const int kg = k0 + kk;
            const int64_t gi = i0 + row, gj = j0 + row;
            ra[e] = (kg < a.d && gi < i_end) ? a.x1[gi * a.d + kg] : 0.0;
            rb[e] = (kg < a.d && gj < j_end) ? a.x2[gj * a.d + kg] : 0.0;
        }
    };
    load_chunk(0);
    for (int k0 = 0; k0 < a.d; k0 += KC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e;
            const int kk = idx & (KC - 1), row = idx >> 4;
            As[kk * LDS_ + row] = ra[e];
            Bs[kk * LDS_ + row] = rb[e];
        }
        __syncthreads();
        if (k0 + KC < a.d) load_chunk(k0 + KC);
        if (MFMA) {
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                double fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i] = As[(ks * 4 + lg) * LDS_ + wm * 32 + i * 16 + l16];  // A[row = l & 15][k = l >> 4]
                    fb[i] = Bs[(ks * 4 + lg) * LDS_ + wn * 32 + i * 16 + l16];  // B[k = l >> 4][col = l & 15]
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        macc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], macc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                double av[4], bv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) av[r] = As[kk * LDS_ + ty + 16 * r];
                const double2 b01 = *reinterpret_cast<const double2*>(&Bs[kk * LDS_ + tx * 4]);
                const double2 b23 = *reinterpret_cast<const double2*>(&Bs[kk * LDS_ + tx * 4 + 2]);
                bv[0] = b01.x; bv[1] = b01.y; bv[2] = b23.x; bv[3] = b23.y;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
            }
        }
        __syncthreads();
    }
    if (MFMA) {
        // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg.  Re-tile through LDS into the
        // 4x4-per-thread layout of the epilogue (rows ty + 16 r, columns 4 tx + c).
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    smem[(wm * 32 + i * 16 + lg + 4 * r) * LDP + wn * 32 + j * 16 + l16] = macc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 v01 = *reinterpret_cast<const double2*>(&smem[(ty + 16 * r) * LDP + tx * 4]);
            const double2 v23 = *reinterpret_cast<const double2*>(&smem[(ty + 16 * r) * LDP + tx * 4 + 2]);
            acc[r][0] = v01.x; acc[r][1] = v01.y; acc[r][2] = v23.x; acc[r][3] = v23.y;
        }
        // (the mirror below re-synchronises before it reuses smem)
    }

    // ---- epilogue: layer recursion on the accumulators ----
    const double inv_d = 1.0 / (double)a.d;
    double q2v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t gj = j0 + tx * 4 + c;
        q2v[c] = gj < j_end ? a.q2[gj] : 0.0;
    }
    double kn[4][4], kt[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t gi = i0 + ty + 16 * r;
        const double q1v = gi < i_end ? a.q1[gi] : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t gj = j0 + tx * 4 + c;
            layer_map(acc[r][c] * inv_d, q1v, q2v[c], arch, a.sym && gi == gj, kn[r][c], kt[r][c]);
        }
    }
    const bool vec = vec_ok != 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t gi = i0 + ty + 16 * r, gj = j0 + tx * 4;
        store4(a.nngp64, a.ld64, gi, gj, i_end, j_end, kn[r], vec);
        store4(a.ntk64, a.ld64, gi, gj, i_end, j_end, kt[r], vec);
        if (a.nngp32 != nullptr || a.ntk32 != nullptr) {  // float32 copies (factorisation input): regulariser on the diagonal
            double vn[4], vt[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool diag = a.sym && gi == gj + c;
                vn[c] = kn[r][c] + (diag ? a.diag_add_nngp32 : 0.0);
                vt[c] = kt[r][c] + (diag ? a.diag_add_ntk32 : 0.0);
            }
            store4(a.nngp32, a.ld32, gi, gj, i_end, j_end, vn, vec);
            store4(a.ntk32, a.ld32, gi, gj, i_end, j_end, vt, vec);
        }
    }

    // ---- mirror image of an off-diagonal tile (symmetric build): transpose through LDS ----
    if (a.sym && bi != bj) {
        for (int which = 0; which < 2; ++which) {
            if (which == 0 && a.nngp64 == nullptr && a.nngp32 == nullptr) continue;
            if (which == 1 && a.ntk64 == nullptr && a.ntk32 == nullptr) continue;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    smem[(tx * 4 + c) * LDP + ty + 16 * r] = which == 0 ? kn[r][c] : kt[r][c];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v[4];
                const double2 v01 = *reinterpret_cast<const double2*>(&smem[(ty + 16 * r) * LDP + tx * 4]);
                const double2 v23 = *reinterpret_cast<const double2*>(&smem[(ty + 16 * r) * LDP + tx * 4 + 2]);
                v[0] = v01.x; v[1] = v01.y; v[2] = v23.x; v[3] = v23.y;
                // element (j0 + ty + 16 r, i0 + tx*4 + c); rows bounded by n2 (= n1), columns by row_end
                const int64_t mi = j0 + ty + 16 * r, mj = i0 + tx * 4;
                if (which == 0) {
                    store4(a.nngp64, a.ld64, mi, mj, j_end, i_end, v, vec);
                    if (!a.lower32) store4(a.nngp32, a.ld32, mi, mj, j_end, i_end, v, vec);
                } else {
                    store4(a.ntk64, a.ld64, mi, mj, j_end, i_end, v, vec);
                    if (!a.lower32) store4(a.ntk32, a.ld32, mi, mj, j_end, i_end, v, vec);
                }
            }
        }
    }
}

// ---- float64 arithmetic of the ReLU map at ~1/2 of the libm cost (the epilogue is bound by the float64 ALUs) ---------------
// Measured on gfx950 (scripts/micro/f64_seed_accuracy.hip, 2^24 inputs over 2^-20 .. 2^21): v_rcp_f64 is good to 2^-24.4 and one
// Newton step brings it to 2.2e-15; v_rsq_f64 is good to 2^-24.2 and one coupled (Goldschmidt) iteration plus the residual
// correction reproduces the correctly rounded sqrt on every input tried.
__device__ __forceinline__ double fast_rcp(double x) {  // 1 / x to 2.2e-15, x > 0
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}

__device__ __forceinline__ double fast_sqrt_pos(double r) {  // sqrt(r), r > 0
    const double y = __builtin_amdgcn_rsq(r);
    double g = r * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    return fma(fma(-g, g, r), h, g);
}

// pi - atan2(s, k) for s >= 0 (what the ReLU map needs; s = k = 0 gives pi / 2, the reference's fill value).
// A float32 estimate of the angle picks the nearest of 65 table angles a_i = i pi / 64; the pair (k, s) is rotated by -a_i in
// float64, which leaves a residual angle below 0.03 rad whose arctangent is u - u^3/3 + ... + u^9/9 (next term < 2e-18).
// tab[i] = {cos a_i, sin a_i, a_i, pi - a_i} in LDS.  Absolute error <= 5e-16 (checked against libm over 2e6 angles, radii
// 1e-3 .. 1e6, and within 1e-12 of 0 and pi, in a NumPy emulation of these exact steps).
__device__ __forceinline__ double pi_minus_atan2(double s, double k, const double* __restrict__ tab) {
    const double ak = fabs(k);
    const double mx = fmax(s, ak), mn = fmin(s, ak);
    float t = mx > 0.0 ? (float)(mn * __builtin_amdgcn_rcp(mx)) : 0.0f;
    float at = t * fmaf(-0.1919f, t * t, 0.9724f);  // atan on [0, 1] to 5e-3: only the table index depends on it
    at = s > ak ? 1.57079637f - at : at;
    at = k < 0.0 ? 3.14159274f - at : at;
    int i = (int)rintf(at * 20.3718327f);  // 64 / pi
    i = i < 0 ? 0 : (i > 64 ? 64 : i);
    const double2 cs = *reinterpret_cast<const double2*>(tab + 4 * i);
    const double xp = fma(k, cs.x, s * cs.y);      // rho cos(theta - a_i) > 0
    const double yp = fma(s, cs.x, -(k * cs.y));   // rho sin(theta - a_i)
    const double u = yp * fast_rcp(xp);
    const double w = u * u;
    double p = fma(w, 1.0 / 9.0, -1.0 / 7.0);
    p = fma(p, w, 1.0 / 5.0);
    p = fma(p, w, -1.0 / 3.0);
    const double atu = fma(u, p * w, u);
    const double pmt = tab[4 * i + 3] - atu;       // (pi - a_i) - (theta - a_i)
    return mx > 0.0 ? pmt : 0.5 * kPi;
}

// K1, second form.  One 64 x 64 output tile per 256-thread workgroup (4 waves, 32 x 32 outputs each), MANY workgroups per CU:
//   * Gram entries on the float64 matrix cores (v_mfma_f64_16x16x4_f64, 2 x 2 blocks per wave).  The x1 / x2 row panels are
//     staged through LDS in k-chunks of 16, ROW-major with a 2-double pad (stores run along k without bank conflicts, and a
//     lane reads the two k values it feeds to two consecutive MFMAs with one 16-byte read);
//   * the layer recursion runs in the MFMA accumulator layout -- no re-tiling pass -- one 16 x 16 block (4 entries per lane)
//     at a time, with the fast float64 sqrt / atan2 above;
//   * results leave through a per-WAVE 16 x 32 buffer in LDS: read back by rows for the tile itself and by columns for its
//     mirror image (symmetric build), so every global store is 16 bytes per lane in runs of 128 / 256 contiguous bytes -- and
//     no workgroup barrier is needed after the k-loop;
//   * <= 128 VGPRs and 21 KB of LDS: 4 workgroups per CU.  The phases of a tile are serial (panel loads -> MFMAs -> layer map
//     -> stores), so the float64 ALUs are only kept busy by OTHER workgroups' phases.  (Measured: a persistent-workgroup form
//     with 2 workgroups per CU was slower than one tile per workgroup, 8.1 vs 7.4 ms at N = 32768.)
//   * tiles are dealt to the XCDs in 8 x 8 super-tiles (512 x 512 entries): workgroup b runs on XCD b % 8, and the 64 tiles of a
//     super-tile share 16 row panels of X through that XCD's L2 instead of fetching 128 KB per tile from the Infinity Cache.
// Bound: the float64 ALUs.  On gfx950 v_mfma_f64 and the float64 VALU instructions do NOT overlap -- a kernel whose even waves
// issue only float64 MFMAs and whose odd waves issue only float64 FMAs takes the SUM of the two times, not the maximum
// (scripts/micro/f64_pipes.hip: 17.2 ms MFMA, 21.1 ms FMA, 18.7 ms half / half) -- so Gram and layer map add up.
constexpr int TLD = 34;       // row stride of the per-wave 16 x 32 output buffer

// MKC: k-chunk; PREFETCH: the next chunk's global loads are issued into registers before the MFMAs of the current one;
// MINWG: workgroups per CU the register allocation must allow; LDSOUT: tile + mirror leave through the LDS buffer with 16-byte
// stores (false: the tile itself is stored straight from the accumulator layout, 8 bytes per lane).
// COMP (round 4): NNGP outputs only, no biases, >= 2 ReLU layers -- the whole layer recursion is then sqrt(q q') A G(pi - theta0) with ONE
// univariate function G of the first layer's angle (comp_table below): one sqrt + one arctangent + a degree-10 polynomial per entry
// instead of a sqrt and an arctangent per layer.  a.comp: [kCompNI][kCompDeg + 1] coefficients + A, staged in LDS.
template <int MKC, bool PREFETCH, int MINWG, bool LDSOUT, bool COMP>
__global__ __launch_bounds__(256, MINWG) void k_build_mfma(BuildArgs a, ArchDev arch, int64_t tiles_r, int64_t tiles_c,
                                                           int64_t sup_r, int64_t sup_c, int vec_ok, int ablate) {
    constexpr int MLD = MKC + 2;  // LDS row stride (doubles): rows 16-byte aligned, quarter-waves on distinct banks
    constexpr int NLD = MKC * 64 / 256;  // doubles per thread and operand per chunk
    __shared__ __attribute__((aligned(16))) double smem[2 * KT * MLD];  // As | Bs in the k-loop, 4 output buffers afterwards
    __shared__ __attribute__((aligned(16))) double tab[65 * 4];
    __shared__ double ctab[COMP ? kCompSize : 1];
    static_assert(4 * 16 * TLD <= 2 * KT * MLD, "the output buffers alias the panel buffers");
    double* As = smem;             // [KT][MLD]
    double* Bs = smem + KT * MLD;  // [KT][MLD]
    const int tid = threadIdx.x;

    // workgroup -> tile: XCD = blockIdx % 8; each XCD walks whole super-tiles of 8 x 8 tiles
    const int64_t b = blockIdx.x;
    const int64_t local = b >> 3, sid = (local >> 6) * 8 + (b & 7);
    const int slot = (int)(local & 63);
    int64_t sbi, sbj;
    if (a.sym) {
        if (sid >= sup_r * (sup_r + 1) / 2) return;
        sbi = (int64_t)((sqrt(8.0 * (double)sid + 1.0) - 1.0) * 0.5);
        while (sbi * (sbi + 1) / 2 > sid) --sbi;
        while ((sbi + 1) * (sbi + 2) / 2 <= sid) ++sbi;
        sbj = sid - sbi * (sbi + 1) / 2;
    } else {
        if (sid >= sup_r * sup_c) return;
        sbi = sid / sup_c;
        sbj = sid % sup_c;
    }
    const int64_t bi = sbi * 8 + (slot >> 3), bj = sbj * 8 + (slot & 7);
    if (bi >= tiles_r || bj >= tiles_c || (a.sym && bj > bi)) return;
    const int64_t i0 = a.row_begin + bi * KT, j0 = bj * KT;
    const int64_t i_end = a.row_end, j_end = a.n2;
    for (int i = tid; i < 65 * 4; i += 256) tab[i] = kTrigTab[i >> 2][i & 3];
    if (COMP)
        for (int i = tid; i < kCompSize; i += 256) ctab[i] = a.comp[i];

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, lg = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

    double ra[NLD], rb[NLD];
    auto load_chunk = [&](int k0) {  // 64 rows x MKC k per operand, lanes along k
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int idx = tid + 256 * e;
            const int kk = idx & (MKC - 1), row = idx / MKC;
            const int kg = k0 + kk;
            const int64_t gi = i0 + row, gj = j0 + row;
            ra[e] = (kg < a.d && gi < i_end) ? a.x1[gi * a.d + kg] : 0.0;
            rb[e] = (kg < a.d && gj < j_end) ? a.x2[gj * a.d + kg] : 0.0;
        }
    };
    if (PREFETCH) load_chunk(0);
    for (int k0 = 0; k0 < ((ablate & 2) ? MKC : a.d); k0 += MKC) {
        if (!PREFETCH) load_chunk(k0);
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int idx = tid + 256 * e;
            const int kk = idx & (MKC - 1), row = idx / MKC;
            As[row * MLD + kk] = ra[e];
            Bs[row * MLD + kk] = rb[e];
        }
        __syncthreads();
        if (PREFETCH && k0 + MKC < a.d) load_chunk(k0 + MKC);
#pragma unroll
        for (int ks = 0; ks < MKC / 8; ++ks) {
            double2 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const double2*>(&As[(wm * 32 + i * 16 + l16) * MLD + ks * 8 + 2 * lg]);
                fb[i] = *reinterpret_cast<const double2*>(&Bs[(wn * 32 + i * 16 + l16) * MLD + ks * 8 + 2 * lg]);
            }
            // lane group g supplies k = 8 ks + 2 g (first MFMA) and 8 ks + 2 g + 1 (second): the sum over k does not care
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // ---- layer recursion in the accumulator layout: entry (row = 16 i + lg + 4 r, col = 16 j + l16) of the wave's block ----
    const double inv_d = 1.0 / (double)a.d;
    const int64_t wi0 = i0 + wm * 32, wj0 = j0 + wn * 32;
    const bool diag_tile = a.sym && bi == bj && wm == wn;  // entries (i, i) exist only in these blocks
    const bool mirror = a.sym && bi != bj;
    const bool vec = vec_ok != 0;
    const bool want_t = a.ntk64 != nullptr || a.ntk32 != nullptr;
    double* tbuf = smem + wave * (16 * TLD);  // this wave's output buffer (As / Bs are dead: the k-loop ended with a barrier)
    double q2in[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t gj = wj0 + j * 16 + l16;
        q2in[j] = gj < j_end ? a.q2[gj] : 0.0;
    }
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {  // 16 rows x 32 columns of the wave's block per pass
        double q1in[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t gi = wi0 + i * 16 + lg + 4 * r;
            q1in[r] = gi < i_end ? a.q1[gi] : 0.0;
        }
        double kv[2][4], tv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f64x4 av = i == 0 ? acc[0][j] : acc[1][j];
            double q1v[4], q2v = q2in[j];
            bool dg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                q1v[r] = q1in[r];
                dg[r] = diag_tile && i == j && lg + 4 * r == l16;
                kv[j][r] = dg[r] ? q1v[r] : av[r] * inv_d;  // exact diagonal: q q' - k^2 == 0 must hold exactly
                tv[j][r] = 0.0;
            }
            if (COMP) {
                const double amp = ctab[kCompSize - 1];  // A = prod w2 / 2^n_relu
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double k = kv[j][r];
                    if (dg[r]) {  // theta = 0 exactly: the closed-form diagonal, in the recursion's own order of operations
                        for (int l = 0; l < arch.n_dense; ++l) {
                            k = arch.w2[l] * k;
                            if (l < arch.n_dense - 1) k *= 0.5;
                        }
                    } else {
                        const double qq = q1v[r] * q2v;
                        const double rr = fma(q1v[r], q2v, -k * k);
                        const double sn = rr > 0.0 ? fast_sqrt_pos(rr > 0.0 ? rr : 1.0) : 0.0;
                        const double pmt = pi_minus_atan2(sn, k, tab);  // pi - theta0 in [0, pi]
                        int iv = (int)(pmt * (kCompNI / kPi));
                        iv = iv < 0 ? 0 : (iv > kCompNI - 1 ? kCompNI - 1 : iv);
                        const double u = fma(pmt, 2.0 * kCompNI / kPi, -(double)(2 * iv + 1));  // (pmt - mid) / half-width
                        const double* cf = ctab + iv * (kCompDeg + 1);
                        double p = cf[kCompDeg];
#pragma unroll
                        for (int e = kCompDeg - 1; e >= 0; --e) p = fma(p, u, cf[e]);
                        const double rho = qq > 0.0 ? fast_sqrt_pos(qq > 0.0 ? qq : 1.0) : 0.0;
                        k = rho * (amp * p);
                    }
                    kv[j][r] = k;
                }
            } else
            for (int l = 0; l < arch.n_dense; ++l) {
                const double w2 = arch.w2[l], b2 = arch.b2[l];
                const bool relu = l < arch.n_dense - 1 && !(ablate & 4);
                q2v = fma(w2, q2v, b2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q1v[r] = fma(w2, q1v[r], b2);
                    double k = fma(w2, kv[j][r], b2);
                    double t = fma(w2, tv[j][r], k);
                    if (relu) {
                        const double rr = dg[r] ? 0.0 : fma(q1v[r], q2v, -k * k);
                        const double s = rr > 0.0 ? fast_sqrt_pos(rr > 0.0 ? rr : 1.0) : 0.0;
                        double kd = pi_minus_atan2(s, k, tab) * (0.5 / kPi);
                        kd = dg[r] ? 0.5 : kd;  // theta = 0 on the diagonal, exactly
                        k = fma(kd, k, s * (0.5 / kPi));
                        t *= kd;
                    }
                    kv[j][r] = k;
                    tv[j][r] = t;
                    q1v[r] *= 0.5;
                }
                q2v *= 0.5;
            }
        }
        if ((ablate & 1) && kv[0][0] != 12345.678 && tv[1][3] != 12345.678) continue;
        if (!LDSOUT) {  // the tile itself straight from the accumulator layout: 16 lanes x 8 bytes = one 128-byte line per row piece
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gi = wi0 + i * 16 + lg + 4 * r, gj = wj0 + j * 16 + l16;
                    if (gi < i_end && gj < j_end) {
                        const bool dgl = a.sym && gi == gj;
                        if (a.nngp64) a.nngp64[gi * a.ld64 + gj] = kv[j][r];
                        if (a.ntk64) a.ntk64[gi * a.ld64 + gj] = tv[j][r];
                        if (a.nngp32) a.nngp32[gi * a.ld32 + gj] = (float)(kv[j][r] + (dgl ? a.diag_add_nngp32 : 0.0));
                        if (a.ntk32) a.ntk32[gi * a.ld32 + gj] = (float)(tv[j][r] + (dgl ? a.diag_add_ntk32 : 0.0));
                    }
                }
            if (!mirror) continue;
        }

        // ---- out through the wave's LDS buffer: rows for the tile, columns for its mirror image ----
        for (int which = 0; which < 2; ++which) {
            double* o64 = which == 0 ? a.nngp64 : a.ntk64;
            float* o32 = which == 0 ? a.nngp32 : a.ntk32;
            if (o64 == nullptr && o32 == nullptr) continue;
            const double dadd = which == 0 ? a.diag_add_nngp32 : a.diag_add_ntk32;
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous pass's reads of tbuf
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) tbuf[(lg + 4 * r) * TLD + j * 16 + l16] = which == 0 ? kv[j][r] : tv[j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int64_t r0 = wi0 + i * 16;  // first global row of this pass
            if (LDSOUT && o64 != nullptr) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {  // 4 rows x 32 columns per store instruction, 16 bytes per lane
                    const int trow = it * 4 + lg, tc = l16 * 2;
                    const double2 v = *reinterpret_cast<const double2*>(&tbuf[trow * TLD + tc]);
                    const int64_t gi = r0 + trow, gj = wj0 + tc;
                    if (gi >= i_end) continue;
                    if (vec && gj + 1 < j_end) *reinterpret_cast<double2*>(o64 + gi * a.ld64 + gj) = v;
                    else {
                        if (gj < j_end) o64[gi * a.ld64 + gj] = v.x;
                        if (gj + 1 < j_end) o64[gi * a.ld64 + gj + 1] = v.y;
                    }
                }
            }
            if (LDSOUT && o32 != nullptr) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {  // 8 rows x 32 columns per store instruction, 4 floats per lane
                    const int trow = it * 8 + (lane >> 3), tc = (lane & 7) * 4;
                    const double2 v0 = *reinterpret_cast<const double2*>(&tbuf[trow * TLD + tc]);
                    const double2 v1 = *reinterpret_cast<const double2*>(&tbuf[trow * TLD + tc + 2]);
                    const int64_t gi = r0 + trow, gj = wj0 + tc;
                    if (gi >= i_end) continue;
                    double v[4] = {v0.x, v0.y, v1.x, v1.y};
                    if (a.sym)
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] += (gi == gj + c) ? dadd : 0.0;
                    float* dst = o32 + gi * a.ld32 + gj;
                    if (vec && gj + 3 < j_end) *reinterpret_cast<float4*>(dst) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                    else
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (gj + c < j_end) dst[c] = (float)v[c];
                }
            }
            if (!mirror) continue;
            float* m32 = a.lower32 ? nullptr : o32;
#pragma unroll
            for (int it = 0; it < 4; ++it) {  // mirror image: 32 rows (the block's columns) of 16 entries, 8 rows per instruction
                const int mrow = it * 8 + (lane >> 3), mc = (lane & 7) * 2;
                const double vx = tbuf[mc * TLD + mrow], vy = tbuf[(mc + 1) * TLD + mrow];
                const int64_t mi = wj0 + mrow, mj = r0 + mc;
                if (mi >= j_end) continue;
                if (o64 != nullptr) {
                    if (vec && mj + 1 < i_end) *reinterpret_cast<double2*>(o64 + mi * a.ld64 + mj) = make_double2(vx, vy);
                    else {
                        if (mj < i_end) o64[mi * a.ld64 + mj] = vx;
                        if (mj + 1 < i_end) o64[mi * a.ld64 + mj + 1] = vy;
                    }
                }
                if (m32 != nullptr) {
                    if (mj < i_end) m32[mi * a.ld32 + mj] = (float)vx;
                    if (mj + 1 < i_end) m32[mi * a.ld32 + mj + 1] = (float)vy;
                }
            }
        }
    }
    (void)want_t;
}

}  // namespace

int launch_row_sqnorm(const double* x, int64_t n, int d, double* q, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, x, n, d, q);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_diag_from_q(const double* q, int64_t n, const ArchDev& arch, double* dn, double* dt, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_diag_from_q, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q, n, arch, dn, dt);
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

static int device_cus() {
    static std::atomic<int> cached{0};
    int n = cached.load(std::memory_order_relaxed);
    if (n == 0) {
        int dev = 0, count = 0;
        n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
             count >= 8) ? count : 256;
        cached.store(n, std::memory_order_relaxed);
    }
    return n;
}

// ---- the composite ReLU map (round 4) --------------------------------------------------------------------------------------
// Without biases the layer recursion is homogeneous: a Dense layer scales k, q, q' alike and a ReLU layer maps the cosine
// c = k / sqrt(q q') to J(theta) / pi, J(theta) = sin theta + (pi - theta) cos theta, theta = acos c, while q halves.  So after all
// layers   K = sqrt(q q') A F(theta0),   A = prod w2 / 2^n_relu,   F = (J / pi) o acos o ... o (J / pi)  (n_relu maps)
// -- ONE univariate function of the FIRST layer's angle, analytic on [0, pi] (at theta0 = 0 the inner acos meets
// 1 - theta0^2 / 2 + theta0^3 / (3 pi) ...: theta1 = theta0 sqrt(1 - 2 theta0 / (3 pi) ...)).  The kernel evaluates
// G(t) = F(pi - t) on kCompNI intervals of t = pi - theta0 (the quantity pi_minus_atan2 returns) by a degree-kCompDeg polynomial
// in the interval's own variable u in [-1, 1]: Chebyshev interpolation of the recursion evaluated in long double here, turned into
// monomial coefficients, CHECKED in double Horner arithmetic against the long-double recursion (<= 4e-16 absolute where G >= 1 / pi,
// else the table is refused and the kernel keeps the per-layer recursion).  Measured accuracy for 2..4 ReLU layers: 1.1e-16.
struct CompEntry {
    int device;
    int n_dense;
    double w2[NNGP_MAX_DENSE];
    const double* dev;  // NULL: refused
};
static std::mutex g_comp_mu;
static std::vector<CompEntry> g_comp;

static long double comp_exact(long double t, int n_relu) {
    const long double pi = 3.14159265358979323846264338327950288L;
    long double th = pi - t, c = 0.0L;
    for (int l = 0; l < n_relu; ++l) {
        if (l > 0) th = acosl(c > 1.0L ? 1.0L : (c < -1.0L ? -1.0L : c));
        c = (sinl(th) + (pi - th) * cosl(th)) / pi;
    }
    return c;
}

static bool comp_build_host(const ArchDev& arch, double* out) {
    const int n_relu = arch.n_dense - 1, D = kCompDeg;
    const long double pi = 3.14159265358979323846264338327950288L;
    // Chebyshev polynomials T_0 .. T_D as monomial coefficients
    long double T[kCompDeg + 1][kCompDeg + 1] = {};
    T[0][0] = 1.0L;
    T[1][1] = 1.0L;
    for (int n = 2; n <= D; ++n)
        for (int k = 0; k <= n; ++k) T[n][k] = (k > 0 ? 2.0L * T[n - 1][k - 1] : 0.0L) - T[n - 2][k];
    double worst = 0.0;
    for (int i = 0; i < kCompNI; ++i) {
        const long double a = pi / (2.0L * kCompNI), m = (2 * i + 1) * a;
        long double f[kCompDeg + 1], cheb[kCompDeg + 1], mono[kCompDeg + 1] = {};
        for (int k = 0; k <= D; ++k) f[k] = comp_exact(m + a * cosl(pi * (k + 0.5L) / (D + 1)), n_relu);
        for (int j = 0; j <= D; ++j) {
            long double sum = 0.0L;
            for (int k = 0; k <= D; ++k) sum += f[k] * cosl(j * pi * (k + 0.5L) / (D + 1));
            cheb[j] = sum * 2.0L / (D + 1);
        }
        cheb[0] *= 0.5L;
        for (int j = 0; j <= D; ++j)
            for (int k = 0; k <= j; ++k) mono[k] += cheb[j] * T[j][k];
        double* cf = out + i * (D + 1);
        for (int k = 0; k <= D; ++k) cf[k] = (double)mono[k];
        for (int e = 0; e <= 64; ++e) {  // the check, in the kernel's own arithmetic
            const double u = -1.0 + e / 32.0;
            double p = cf[D];
            for (int k = D - 1; k >= 0; --k) p = fma(p, u, cf[k]);
            const double err = fabs((double)((long double)p - comp_exact(m + a * (long double)u, n_relu)));
            worst = err > worst ? err : worst;
        }
    }
    double amp = 1.0;
    for (int l = 0; l < arch.n_dense; ++l) amp *= arch.w2[l];
    out[kCompSize - 1] = ldexp(amp, -n_relu);
    return worst <= 4e-16;
}

// the table of this architecture on the device (built once per process and architecture), or NULL when the composite form does
// not apply: biases, fewer than two ReLU layers, a table that failed its check
static const double* comp_table(const ArchDev& arch) {
    if (arch.n_dense < 3) return nullptr;
    for (int l = 0; l < arch.n_dense; ++l)
        if (arch.b2[l] != 0.0 || !(arch.w2[l] > 0.0)) return nullptr;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_comp_mu);
    for (const CompEntry& e : g_comp)
        if (e.device == device && e.n_dense == arch.n_dense && memcmp(e.w2, arch.w2, sizeof(double) * arch.n_dense) == 0) return e.dev;
    CompEntry e{};
    e.device = device;
    e.n_dense = arch.n_dense;
    memcpy(e.w2, arch.w2, sizeof(double) * arch.n_dense);
    std::vector<double> host(kCompSize);
    double* dev = nullptr;
    if (comp_build_host(arch, host.data()) && hipMalloc(reinterpret_cast<void**>(&dev), sizeof(double) * kCompSize) == hipSuccess) {
        if (hipMemcpy(dev, host.data(), sizeof(double) * kCompSize, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(dev);
            dev = nullptr;
        }
    }
    (void)hipGetLastError();
    e.dev = dev;
    g_comp.push_back(e);
    return dev;
}

int launch_kernel_build(const BuildArgs& a_in, const ArchDev& arch, hipStream_t s) {
    BuildArgs a = a_in;
    // NNGP outputs only, no biases, >= 2 ReLU layers: the composite map (debug key 5 = 63: the per-layer recursion)
    a.comp = (a.ntk64 == nullptr && a.ntk32 == nullptr && !a.no_comp && NNGP_KNOB(5) != 63) ? comp_table(arch) : nullptr;
    const int64_t rows = a.row_end - a.row_begin;
    if (rows <= 0 || a.n2 <= 0) return 0;
    NNGP_REQUIRE(a.d > 0, "kernel_build: d must be positive");
    const int64_t tiles_r = (rows + KT - 1) / KT, tiles_c = (a.n2 + KT - 1) / KT;
    int64_t nblocks;
    if (a.sym) {
        NNGP_REQUIRE(a.row_begin == 0 && a.row_end == a.n1 && a.n1 == a.n2,
                     "kernel_build: symmetric mode needs the full row range");
        nblocks = tiles_r * (tiles_r + 1) / 2;
    } else {
        nblocks = tiles_r * tiles_c;
    }
    NNGP_REQUIRE(nblocks < (int64_t)2147483647, "kernel_build: grid too large (%lld tiles)", (long long)nblocks);
    // 16-byte stores are legal when every row start is 16-byte aligned for every output in use
    auto aligned = [](const void* p, int64_t ld, int esz) {
        return p == nullptr || ((((uintptr_t)p) & 15) == 0 && ((ld * esz) & 15) == 0);
    };
    int vec_ok = aligned(a.nngp64, a.ld64, 8) && aligned(a.ntk64, a.ld64, 8) && aligned(a.nngp32, a.ld32, 4) &&
                 aligned(a.ntk32, a.ld32, 4) && (a.row_begin % 4 == 0);
    // Round 1 (scripts/k1_study.py, N=32768, d=128, n_relu=3): the all-VALU kernel k_build<false> took 10.7 ms -- 3.3 ms Gram
    // (41 TF/s float64 FMA), 1.7 ms per ReLU layer (libm sqrt + atan2), 3.0 ms stores; its MFMA variant was slower because it
    // re-tiled the accumulators through LDS and staged the panels k-major (8-way bank conflicts on the stores).  k_build_mfma
    // keeps the epilogue in the accumulator layout, overlaps MFMA and VALU across co-resident workgroups and evaluates the
    // map with the fast float64 helpers.  Debug key 3 = 4: the round-1 kernel, for A/B timing.
    if (NNGP_KNOB(3) == 4 || NNGP_KNOB(3) == 3) {
        if (NNGP_KNOB(3) == 3)
            hipLaunchKernelGGL(k_build<true>, dim3((unsigned)nblocks), dim3(256), 0, s, a, arch, tiles_c, vec_ok);
        else
            hipLaunchKernelGGL(k_build<false>, dim3((unsigned)nblocks), dim3(256), 0, s, a, arch, tiles_c, vec_ok);
    } else {
        const int64_t sup_r = (tiles_r + 7) / 8, sup_c = (tiles_c + 7) / 8;
        const int64_t nsup = a.sym ? sup_r * (sup_r + 1) / 2 : sup_r * sup_c;
        const int64_t grid = ((nsup + 7) / 8) * 8 * 64;  // 64 tile slots per super-tile, super-tiles dealt round-robin to 8 XCDs
        // (Measured and dropped: a persistent grid of 3 workgroups per compute unit walking the slots -- the 133 k one-tile
        // workgroups spend ~1.3 ms in dispatch and set-up when everything else is masked out -- took 7.95 ms against 6.97: the
        // loop state costs 17 spilled registers at the 168 the occupancy allows, and the same body with one slot per
        // workgroup already loses 0.45 ms to them.)
        NNGP_REQUIRE(grid < (int64_t)2147483647, "kernel_build: grid too large (%lld workgroups)", (long long)grid);
        const int ablate = NNGP_KNOB(3) >= 32 && NNGP_KNOB(3) < 40 ? NNGP_KNOB(3) - 32 : 0;  // timing ablations (wrong results)
        const int variant = NNGP_KNOB(5) >= 20 && NNGP_KNOB(5) < 30 ? NNGP_KNOB(5) - 20 : 0;  // A/B timing of the kernel forms
#define NNGP_K1_LAUNCH(KC_, PF_, WG_, LO_) \
        hipLaunchKernelGGL((k_build_mfma<KC_, PF_, WG_, LO_, false>), dim3((unsigned)grid), dim3(256), 0, s, a, arch, tiles_r, tiles_c, sup_r, sup_c, vec_ok, ablate)
        switch (variant) {
#ifdef NNGP_TIMING_KNOBS  // measured (scripts/k1_variants.py, ms at N = 32768, d = 128, n_relu = 3; all stores / no stores):
            case 1: NNGP_K1_LAUNCH(16, false, 4, true); break;   // 7.66 / 5.82  (29 registers spilled)
            case 2: NNGP_K1_LAUNCH(32, true, 3, false); break;   // 6.76 / 6.17
            case 3: NNGP_K1_LAUNCH(32, true, 3, true); break;    // 6.74 / 6.11
            case 4: NNGP_K1_LAUNCH(32, true, 4, false); break;   // 7.11 / 5.90  (25 spilled)
            case 5: NNGP_K1_LAUNCH(16, true, 4, false); break;   // 7.00 / 5.81  (24 spilled)
            case 7: NNGP_K1_LAUNCH(32, false, 4, false); break;  // 7.23 / 5.81  (24 spilled)
#endif
            default:
                if (a.comp != nullptr && ablate == 0)
                    hipLaunchKernelGGL((k_build_mfma<16, true, 3, false, true>), dim3((unsigned)grid), dim3(256), 0, s, a, arch, tiles_r, tiles_c,
                                       sup_r, sup_c, vec_ok, ablate);
                else
                    NNGP_K1_LAUNCH(16, true, 3, false);  // 6.7 ms at N = 32768, d = 128, n_relu = 3 (round 1: 10.1)
                break;
        }
#undef NNGP_K1_LAUNCH
    }
    NNGP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace nngp
