// Micro-benchmark: what one level of a level-synchronous LDS solve costs on one CU of gfx950, by piece:
//   (a) s_barrier alone, 4 / 8 / 16 waves;
//   (b) barrier + one wave doing { 4 dependent-address ds_reads, 4 f64 FMAs, ds_write } (the chain of lu_solve_tasks);
//   (c) the same with a DPP tree in the chain.
// Ticks are s_memtime (clock64) -- compare with the wall time printed beside them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int MODE>
__global__ void __launch_bounds__(1024) levels(double* out, unsigned long long* ticks, int n_levels) {
    extern __shared__ double smem[];
    volatile lds_f64* x = (volatile lds_f64*)smem;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) x[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int tid = threadIdx.x;
    int c0 = (tid * 7 + 1) & 4095, c1 = (tid * 13 + 5) & 4095, c2 = (tid * 29 + 3) & 4095, c3 = (tid * 31 + 11) & 4095;
    const int my_level = tid / 8;  // eight lanes work per level, one wave holds eight consecutive levels
    double acc = 0.0;
    const unsigned long long t0 = clock64();
    for (int l = 0; l < n_levels; ++l) {
        if (MODE >= 1 && (my_level & 127) == (l & 127)) {
            const double a = x[c0], b = x[c1], c = x[c2], d = x[c3];
            double s = a * 0.5;
            s += b * 0.25;
            s += c * 0.125;
            s += d * 0.0625;
            if (MODE >= 2) {
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
            }
            x[(c0 + l) & 4095] = s * 0.999;
            acc += s;
        }
        lds_barrier();
    }
    const unsigned long long t1 = clock64();
    if (tid == 0) ticks[0] = t1 - t0;
    out[tid] = acc;
}
int main() {
    double* out;
    unsigned long long* ticks;
    hipMalloc(&out, 1024 * 8);
    hipMalloc(&ticks, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n = 4096;
    for (int threads : {256, 512, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(levels<0>, dim3(1), dim3(threads), 4096 * 8, 0, out, ticks, n);
                if (mode == 1) hipLaunchKernelGGL(levels<1>, dim3(1), dim3(threads), 4096 * 8, 0, out, ticks, n);
                if (mode == 2) hipLaunchKernelGGL(levels<2>, dim3(1), dim3(threads), 4096 * 8, 0, out, ticks, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t = 0;
            hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            printf("threads %4d mode %d (%s): %.1f ticks per level, %.1f ns per level\n", threads, mode,
                   mode == 0 ? "barrier only" : mode == 1 ? "barrier + 4 reads, 4 FMAs, write" : "+ 3 shuffles", (double)t / n, ms * 1e6 / n);
        }
    }
    return 0;
}
