// Accuracy of the float64 hardware seeds on gfx950: max relative error of v_rcp_f64 and v_rsq_f64 over 2^24 random inputs,
// and after one / two Newton (Goldschmidt) steps.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(double* out) {
    const unsigned long long i = blockIdx.x * 256ull + threadIdx.x;
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    const double x = exp2(40.0 * u - 20.0) * (1.0 + u);
    double r0 = __builtin_amdgcn_rcp(x), e = fma(-x, r0, 1.0), r1 = fma(r0, e, r0);
    e = fma(-x, r1, 1.0); double r2 = fma(r1, e, r1);
    const double exact = 1.0 / x;
    double y = __builtin_amdgcn_rsq(x), g = x * y, h = 0.5 * y;
    const double sq = sqrt(x);
    double ee = fma(-h, g, 0.5), g1 = fma(g, ee, g), h1 = fma(h, ee, h);
    double d1 = fma(-g1, g1, x), s1 = fma(d1, h1, g1);   // one iteration + residual correction
    double e2 = fma(-h1, g1, 0.5), g2 = fma(g1, e2, g1), h2 = fma(h1, e2, h1);
    double d2 = fma(-g2, g2, x), s2 = fma(d2, h2, g2);   // two iterations + correction
    double errs[6] = {fabs(r0 - exact) / exact, fabs(r1 - exact) / exact, fabs(r2 - exact) / exact,
                      fabs(g - sq) / sq, fabs(s1 - sq) / sq, fabs(s2 - sq) / sq};
    for (int j = 0; j < 6; ++j) {
        double v = errs[j];
        for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
        if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)&out[j], (unsigned long long)__double_as_longlong(v));
    }
}
int main() {
    double* out; (void)hipMalloc(&out, 48); (void)hipMemset(out, 0, 48);
    hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, out);
    double h[6]; (void)hipMemcpy(h, out, 48, hipMemcpyDeviceToHost);
    printf("v_rcp_f64 seed %.3e (2^%.1f)  +1 Newton %.3e  +2 Newton %.3e\n", h[0], log2(h[0]), h[1], h[2]);
    printf("v_rsq_f64: x*rsq %.3e (2^%.1f)  1 iteration + correction %.3e  2 iterations + correction %.3e\n", h[3], log2(h[3]), h[4], h[5]);
    return 0;
}
