// Do float64 MFMA and float64 VALU FMA run on separate pipes on gfx950?  Three launches of the same grid (1 WG of 512 threads per
// CU x 4 WGs: 2+ waves per SIMD): (a) every wave issues N dependent-free f64 MFMAs, (b) every wave issues the f64 FMAs of equal
// flop count, (c) even waves MFMA / odd waves FMA at the per-wave counts of (a) and (b).  Separate pipes: t(c) ~ max / 2..., shared: ~ (a+b)/2.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int mode, int iters, double* out) {
    const int wave = threadIdx.x >> 6;
    f64x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3 + i;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1) == 0);
    const bool do_fma = mode == 1 || (mode == 2 && (wave & 1) == 1);
    if (do_mfma)
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    if (do_fma)
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)  // 4 MFMAs = 4 * 2048 flops = 64 lanes * 2 * 64 FMAs
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = fma(v[j], a, b);
    double s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int j = 0; j < 16; ++j) s += v[j];
    if (s == 12345.678) out[0] = s;
}
int main() {
    double* out; hipMalloc(&out, 8);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * 4), dim3(512), 0, 0, mode, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 256.0 * 4 * 8 * iters * 4 * 2048.0 * (mode == 2 ? 1.0 : 1.0);
            printf("mode %d (%s): %.3f ms  -> %.1f TF/s if all waves did that work\n", mode, mode == 0 ? "mfma" : mode == 1 ? "valu fma" : "half/half", ms, flops / ms / 1e9);
        }
    return 0;
}
