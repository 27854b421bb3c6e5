// Which XCD does workgroup i of a launch run on?  HW_REG_XCC_ID per workgroup of a 64-workgroup launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    unsigned* d; hipMalloc(&d, 4 * 64);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
    unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%s%08x", i % 8 ? " " : "\n", h[i]);
    printf("\n");
    return 0;
}
