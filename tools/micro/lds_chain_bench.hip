// Micro-benchmark: cost of a dependent LDS chain (pointer chase) and of a DPP wave reduction for ONE wave on an idle chip,
// in s_memtime ticks and in wall time -- the unit costs the LU solve kernels (relp_amd/csrc/lu.hip) are built from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) int lds_i32;
__global__ void chase(int* out, unsigned long long* ticks, int steps, int waves_active) {
    extern __shared__ int smem[];
    lds_i32* s = (lds_i32*)smem;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (i * 97 + 13) & 4095;
    __syncthreads();
    if ((int)(threadIdx.x / 64) >= waves_active) return;
    int p = threadIdx.x & 63;
    const unsigned long long t0 = clock64();
    for (int k = 0; k < steps; ++k) p = s[p];
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = p;
}
__global__ void chase_store(int* out, unsigned long long* ticks, int steps) {
    extern __shared__ int smem[];
    lds_i32* s = (lds_i32*)smem;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (i * 97 + 13) & 4095;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    int p = threadIdx.x & 63;
    const unsigned long long t0 = clock64();
    for (int k = 0; k < steps; ++k) {  // read -> dependent store -> dependent read (what a level of the solve does)
        const int q = s[p];
        s[(q + 1) & 4095] = q;
        p = s[(q + 1) & 4095] ;
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = p;
}
int main() {
    int* out;
    unsigned long long* ticks;
    hipMalloc(&out, 1024 * sizeof(int));
    hipMalloc(&ticks, 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&chase), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&chase_store), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int lds_kb : {16, 150})
        for (int waves : {1, 16}) {
            const int steps = 20000;
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            chase<<<1, 1024, lds_kb * 1024>>>(out, ticks, 100, waves);
            hipDeviceSynchronize();
            hipEventRecord(a);
            chase<<<1, 1024, lds_kb * 1024>>>(out, ticks, steps, waves);
            hipEventRecord(b);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            unsigned long long t = 0;
            hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            printf("LDS %3d KB, %2d waves chasing: %.1f ticks / dependent ds_read, %.1f ns wall / read (kernel %.3f ms)\n", lds_kb, waves,
                   (double)t / steps, ms * 1e6 / steps, ms);
        }
    {
        const int steps = 20000;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        chase_store<<<1, 1024, 150 * 1024>>>(out, ticks, steps);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long t = 0;
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("read -> store -> read trip: %.1f ticks, %.1f ns wall\n", (double)t / steps, ms * 1e6 / steps);
    }
    return 0;
}
