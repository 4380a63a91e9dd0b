// Issue rates of the VALU instructions the dense pricing kernel is made of, on one SIMD: cycles per wave instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/micro/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
template <int WHICH>
__global__ void __launch_bounds__(64) rate_kernel(double* out, long long* cycles, int iters) {
    double a0 = threadIdx.x, a1 = 1.0, a2 = 2.0, a3 = 3.0, a4 = 4.0, a5 = 5.0, a6 = 6.0, a7 = 7.0;
    double b = 1.0000001, c = 0.5;
    int i0 = threadIdx.x, i1 = 7, i2 = 9, i3 = 11;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (WHICH == 0) {  // v_fmac_f64, 8 independent chains
            REP8(asm volatile("v_fmac_f64 %0, %8, %9\n v_fmac_f64 %1, %8, %9\n v_fmac_f64 %2, %8, %9\n v_fmac_f64 %3, %8, %9\n"
                              "v_fmac_f64 %4, %8, %9\n v_fmac_f64 %5, %8, %9\n v_fmac_f64 %6, %8, %9\n v_fmac_f64 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (WHICH == 1) {  // v_fmac_f64_dpp row_newbcast
            REP8(asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (WHICH == 2) {  // v_cvt_f64_i32
            REP8(asm volatile("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %10\n v_cvt_f64_i32 %3, %11\n"
                              "v_cvt_f64_i32 %4, %8\n v_cvt_f64_i32 %5, %9\n v_cvt_f64_i32 %6, %10\n v_cvt_f64_i32 %7, %11\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));)
        } else if (WHICH == 3) {  // v_bfe_i32
            REP8(asm volatile("v_bfe_i32 %0, %4, 8, 8\n v_bfe_i32 %1, %4, 16, 8\n v_bfe_i32 %2, %4, 0, 8\n v_bfe_i32 %3, %4, 24, 8\n"
                              "v_bfe_i32 %0, %4, 8, 8\n v_bfe_i32 %1, %4, 16, 8\n v_bfe_i32 %2, %4, 0, 8\n v_bfe_i32 %3, %4, 24, 8\n"
                              : "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i0) : "v"(it));)
        } else if (WHICH == 4) {  // v_cvt_f32_ubyte1 (byte -> f32 in one instruction) then v_cvt_f64_f32
            float f0, f1, f2, f3;
            REP8(asm volatile("v_cvt_f32_ubyte0 %8, %12\n v_cvt_f32_ubyte1 %9, %12\n v_cvt_f32_ubyte2 %10, %12\n v_cvt_f32_ubyte3 %11, %12\n"
                              "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(i0));)
        } else if (WHICH == 5) {  // v_add_f64 (the 2^52 trick's subtraction)
            REP8(asm volatile("v_add_f64 %0, %8, %9\n v_add_f64 %1, %8, %9\n v_add_f64 %2, %8, %9\n v_add_f64 %3, %8, %9\n"
                              "v_add_f64 %4, %8, %9\n v_add_f64 %5, %8, %9\n v_add_f64 %6, %8, %9\n v_add_f64 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (WHICH == 6) {  // v_fma_f32 for reference
            float g0 = a0, g1 = a1, g2 = a2, g3 = a3;
            REP8(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n"
                              "v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n"
                              : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"((float)b), "v"((float)c));)
            a0 += g0 + g1 + g2 + g3;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
template <int WHICH>
void run(const char* name, int waves_per_simd) {
    double* out;
    long long* cyc;
    const int blocks = 256 * 4 * waves_per_simd, iters = 2000;
    hipMalloc(&out, blocks * 64 * sizeof(double));
    hipMalloc(&cyc, blocks * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<WHICH><<<blocks, 64>>>(out, cyc, 10);
    hipEventRecord(e0);
    rate_kernel<WHICH><<<blocks, 64>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // every SIMD runs waves_per_simd waves of iters * 64 instructions
    const double per_simd = (double)waves_per_simd * iters * 64;
    std::printf("%-28s waves/SIMD %d: %.2f ns per wave instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(out);
    hipFree(cyc);
}
int main() {
    for (int w : {1, 4}) {
        if (w == 1) { run<0>("v_fmac_f64", 1); run<1>("v_fmac_f64_dpp newbcast", 1); run<2>("v_cvt_f64_i32", 1); run<3>("v_bfe_i32", 1); run<4>("cvt_f32_ubyte + cvt_f64_f32 (per pair)", 1); run<5>("v_add_f64", 1); run<6>("v_fmac_f32", 1); }
        else { run<0>("v_fmac_f64", 4); run<1>("v_fmac_f64_dpp newbcast", 4); run<2>("v_cvt_f64_i32", 4); run<3>("v_bfe_i32", 4); run<4>("cvt_f32_ubyte + cvt_f64_f32 (per pair)", 4); run<5>("v_add_f64", 4); run<6>("v_fmac_f32", 4); }
    }
    return 0;
}
