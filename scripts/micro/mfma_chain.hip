// Latency of the leaf's column step on gfx950: MFMA 32x32x2 f32 -> read one accumulator row -> (readlane, rsq) -> multiply -> MFMA.
// hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form scripts/micro/mfma_chain.hip -o gpurun_out/mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_out[16];

template <int KIND>
__global__ __launch_bounds__(256) void k_chain(float* sink, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float ring[32 * 64];
    __shared__ int flag;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    f32x16 acc, accy;
    for (int r = 0; r < 16; ++r) acc[r] = 1.0f + 0.001f * (float)((lane * 17 + r * 5) % 31);
    for (int r = 0; r < 16; ++r) accy[r] = 0.5f + 0.001f * (float)((lane * 13 + r * 7) % 29);
    unsigned long long t0 = 0, t1 = 0;
    if (wave == 0 || KIND == 4) {
    for (int rep = 0; rep < reps; ++rep) {
        if (rep == reps - 1 || rep == 0) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int rj = (j & 3) + 4 * (j >> 3), hj = (j >> 2) & 1;
            const float xr = acc[rj];
            float x, xn;
            asm volatile("" : "+v"(acc));
            if (KIND == 0) {  // MFMA -> mul -> MFMA
                x = xr * 0.001f; xn = -x;
            } else if (KIND == 1) {  // + readlane
                const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xr), 32 * hj + j));
                x = xr * d * 1e-3f; xn = -x;
            } else {  // + rsq (the leaf's chain)
                const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xr), 32 * hj + j));
                const float inv = __builtin_amdgcn_rsqf(d > 1e-3f ? d : 1e-3f);
                const float xm = ((lane >> 5) == hj && (lane & 31) > j) ? xr : 0.0f;
                x = xm * inv * 1e-2f; xn = -x;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xn, x, acc, 0, 0, 0);
            if (KIND == 5) {  // a second, independent accumulator chain: its own row j is its B operand
                const float yb = ((lane >> 5) == hj) ? accy[rj] : 0.0f;
                accy = __builtin_amdgcn_mfma_f32_32x32x2f32(xn * 1e-3f, yb, accy, 0, 0, 0);
            }
            if (KIND == 6) {  // a second MFMA that depends on nothing but its own accumulator (no row read)
                accy = __builtin_amdgcn_mfma_f32_32x32x2f32(xn * 1e-3f, x, accy, 0, 0, 0);
            }
            if (KIND == 3 || KIND == 4) {  // + the publish
                ring[j * 64 + lane] = xn;
                if (lane == 0) __hip_atomic_store(&flag, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (rep == reps - 1 || rep == 0) t1 = __builtin_amdgcn_s_memtime();
        if (rep == 0 && threadIdx.x == 0) g_out[8 + KIND] = t1 - t0;
    }
    }
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += acc[r] + accy[r];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) g_out[KIND] = t1 - t0;
}

// Leader wave 0 publishes one tagged 8-byte word per lane and step; NF follower waves poll it and run their own MFMA.
// VAR 0: tight poll; 1: s_sleep 1 in the poll loop; 2: s_sleep 4; 3: followers poll but run no MFMA
template <int NF, int VAR>
__global__ __launch_bounds__(256) void k_lead(float* sink, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ unsigned long long ring64[32 * 64];
    for (int k = 0; k < 8; ++k) ring64[k * 256 + threadIdx.x] = 0ull;
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 1.0f + 0.001f * (float)((lane * 17 + r * 5) % 31);
    unsigned long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < reps; ++rep) {
        if (wave == 0) {
            if (rep == reps - 1) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int rj = (j & 3) + 4 * (j >> 3), hj = (j >> 2) & 1;
                const float xr = acc[rj];
                const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xr), 32 * hj + j));
                const float inv = __builtin_amdgcn_rsqf(fmaxf(d, 1e-3f));
                const float xm = ((lane >> 5) == hj && (lane & 31) > j) ? xr : 0.0f;
                const float x = xm * inv * 1e-2f, xn = -x;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xn, x, acc, 0, 0, 0);
                const unsigned long long word = ((unsigned long long)(unsigned)(rep * 32 + j + 1) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, xn * inv);
                __hip_atomic_store(&ring64[j * 64 + lane], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (rep == reps - 1) t1 = __builtin_amdgcn_s_memtime();
        } else if (wave <= NF) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int rj = (j & 3) + 4 * (j >> 3), hj = (j >> 2) & 1;
                const int target = rep * 32 + j + 1;
                const float bq = ((lane >> 5) == hj) ? acc[rj] : 0.0f;
                unsigned long long word;
                do {
                    word = __hip_atomic_load(&ring64[j * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (VAR == 1) __builtin_amdgcn_s_sleep(1);
                    if (VAR == 2) __builtin_amdgcn_s_sleep(4);
                } while ((int)(word >> 32) != target);
                if (VAR != 3) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, (unsigned)word) * 1e-3f, bq, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    sink[threadIdx.x] = s;
    if (threadIdx.x == 0) g_out[0] = t1 - t0;
}
template <int NF, int VAR>
static void run_lead(const char* what, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 2000;
    hipLaunchKernelGGL((k_lead<NF, VAR>), dim3(1), dim3(256), 0, 0, sink, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_lead<NF, VAR>), dim3(1), dim3(256), 0, 0, sink, reps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_out), sizeof(h));
    printf("%-58s %7.1f ns per step (wall, incl. barrier), leader's last pass %.1f ticks per step\n", what, ms * 1e6 / (reps * 32.0), h[0] / 32.0);
}

template <int KIND>
static void run(const char* what, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 2000;
    hipLaunchKernelGGL(k_chain<KIND>, dim3(1), dim3(256), 0, 0, sink, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_chain<KIND>, dim3(1), dim3(256), 0, 0, sink, reps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_out), sizeof(h));
    printf("%-40s %7.1f ns per step (wall), memtime ticks per step: first pass %.1f, last pass %.1f\n", what, ms * 1e6 / (reps * 32.0), h[8 + KIND] / 32.0, h[KIND] / 32.0);
}
int main() {
    float* sink; hipMalloc(&sink, 4096);
    run<0>("mfma -> mul -> mfma", sink);
    run<1>("mfma -> readlane -> mul -> mfma", sink);
    run<2>("mfma -> readlane -> rsq -> mul -> mfma", sink);
    run<3>("... + ring store + release flag", sink);
    run<4>("... the same on all four waves", sink);
    run<5>("chain + a second accumulator chain (row read)", sink);
    run<6>("chain + a second accumulator (no row read)", sink);
    run_lead<0, 0>("leader publishing tagged words, no follower", sink);
    run_lead<1, 0>("... 1 follower (poll + mfma)", sink);
    run_lead<2, 0>("... 2 followers", sink);
    run_lead<3, 0>("... 3 followers", sink);
    run_lead<3, 1>("... 3 followers, s_sleep 1 in the poll loop", sink);
    run_lead<3, 2>("... 3 followers, s_sleep 4 in the poll loop", sink);
    run_lead<3, 3>("... 3 followers polling only (no mfma)", sink);
    return 0;
}
