// Round 6: what does a Toeplitz fragment cost the LDS pipe?  The update tiles of the exact simplex (exact.hip, mfma_update_tile) read one
// 16-byte fragment per lane and MFMA at a 4-byte-aligned address (four copies of the byte string, shifted by 0..3 bytes); the alternative is
// sixteen copies (shifted by 0..15 bytes) and 16-byte-aligned reads.  Same lanes -> same slots as the tile (g = lane >> 4, gq = (lane & 15) >> 2,
// rq = lane & 3, tq = 0..3), eight waves per CU as in the kernel, no MFMAs: nanoseconds per ds_read_b128 per wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lfb tools/micro/lds_fragment_bench.hip && /tmp/lfb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned lds_u32;
constexpr int WB = 1024, STRIDE = WB + 64;  // 128 limbs

template <int COPIES>
__global__ void __launch_bounds__(256) bench(int iterations, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned image[COPIES * STRIDE / 4];
    for (int k = threadIdx.x; k < COPIES * STRIDE / 4; k += blockDim.x) image[k] = k * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, gq = (lane & 15) >> 2, rq = lane & 3;
    const lds_u32* base_of = (const lds_u32*)image;
    v4i acc = {0, 0, 0, 0};
    for (int it = 0; it < iterations; ++it) {
        const int base = WB - 16 - 64 * (it & 7) - 16 * (gq - g);  // (the block distance moves through the string)
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const lds_u32* at;
            if (COPIES == 4) at = base_of + (3 - rq) * (STRIDE / 4) + (base + 4 * (3 - tq)) / 4;   // 4-byte aligned
            else at = base_of + (4 * (3 - tq) + (3 - rq)) * (STRIDE / 4) + base / 4;               // 16-byte aligned
            const v4i f = {(int)at[0], (int)at[1], (int)at[2], (int)at[3]};
            acc ^= f;
        }
    }
    if (acc[0] == 0x12345 && acc[1] == 7) out[0] = acc[2] ^ acc[3];
}

int main() {
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, 64));
    const int iterations = 20000, grid = 512;
    for (int rep = 0; rep < 2; ++rep)
        for (int copies : {4, 16}) {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a));
            CHECK(hipEventCreate(&b));
            CHECK(hipEventRecord(a));
            if (copies == 4) hipLaunchKernelGGL(bench<4>, dim3(grid), dim3(256), 0, 0, iterations, d_out);
            else hipLaunchKernelGGL(bench<16>, dim3(grid), dim3(256), 0, 0, iterations, d_out);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            // eight waves per CU (two workgroups of four): per CU 8 x 4 x iterations reads of 1 KB
            const double reads_per_cu = 8.0 * 4.0 * iterations;
            printf("%2d copies (%s): %.3f ms, %.2f ns per ds_read_b128 per CU = %.1f bytes per clock at 2.4 GHz (peak 128)\n", copies,
                   copies == 4 ? "4-byte aligned, as the tiles read" : "16-byte aligned", ms, ms * 1e6 / reads_per_cu, 1024.0 / (ms * 1e6 / reads_per_cu) / 2.4);
        }
    return 0;
}
