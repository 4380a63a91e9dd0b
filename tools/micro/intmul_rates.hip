// What bounds the exact simplex's big-integer products on gfx950: issue rates of the three ways to form 64 x 64 -> 128-bit word
// products, whole chip (every SIMD busy), and the lane map of the i8 MFMA checked with exact integer data.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/intmul_rates tools/micro/intmul_rates.hip && /tmp/intmul_rates
//  (0) v_mad_u64_u32, eight independent chains                      -> cycles per wave instruction per SIMD
//  (1) the 4 x 4-word block product of exact.hip (u128 arithmetic as hipcc compiles it), operands in registers
//                                                                    -> 64 x 64 word products per second, the VALU ceiling of the kernel as written
//  (2) v_fma_f64 (exact on 24-bit digits: 48-bit products, 32 of them per 53-bit sum)
//  (3) v_mfma_i32_16x16x64_i8, four independent accumulators        -> 8-bit MACs per second
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
typedef unsigned __int128 u128;
typedef int v4i __attribute__((ext_vector_type(4)));
#define REP8(X) X X X X X X X X

template <int WHICH>
__global__ void __launch_bounds__(64) rate_kernel(u64* out, int iters, u64 seed) {
    const int lane = threadIdx.x;
    u64 r = 0;
    if (WHICH == 0) {
        u64 a0 = lane, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
        unsigned x = (unsigned)seed | 1u, y = (unsigned)(seed >> 32) | 3u;
        for (int it = 0; it < iters; ++it) {
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");)
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else if (WHICH == 1) {
        u64 a4[4], b4[4], acc[9];
        for (int t = 0; t < 4; ++t) { a4[t] = seed * (lane + t + 1); b4[t] = (seed >> 3) * (lane + 5 + t); }
        for (int k = 0; k < 9; ++k) acc[k] = k;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                u64 carry = 0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const u128 t = (u128)a4[ii] * b4[jj] + acc[ii + jj] + carry;
                    acc[ii + jj] = (u64)t;
                    carry = (u64)(t >> 64);
                }
#pragma unroll
                for (int k = ii + 4; k < 9; ++k) {
                    const u128 t = (u128)acc[k] + carry;
                    acc[k] = (u64)t;
                    carry = (u64)(t >> 64);
                }
            }
            a4[it & 3] ^= acc[8];  // (keeps the block from being hoisted)
        }
        for (int k = 0; k < 9; ++k) r += acc[k];
    } else if (WHICH == 2) {
        double a0 = lane, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0000001, c = 0.5;
        for (int it = 0; it < iters; ++it) {
            REP8(asm volatile("v_fmac_f64 %0, %8, %9\n v_fmac_f64 %1, %8, %9\n v_fmac_f64 %2, %8, %9\n v_fmac_f64 %3, %8, %9\n"
                              "v_fmac_f64 %4, %8, %9\n v_fmac_f64 %5, %8, %9\n v_fmac_f64 %6, %8, %9\n v_fmac_f64 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
        r = (u64)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    } else if (WHICH == 3) {
        v4i a = {(int)seed + lane, 2, 3, 4}, b = {(int)(seed >> 7) ^ lane, 6, 7, 8};
        v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
            }
        }
        r = (u64)(c0[0] + c1[1] + c2[2] + c3[3]);
    }
    out[blockIdx.x * 64 + lane] = r;
}

template <int WHICH>
double run(int waves_per_simd, int iters, double per_iteration) {  // returns "units" per second, whole chip
    u64* out;
    const int blocks = 256 * 4 * waves_per_simd;
    hipMalloc(&out, blocks * 64 * sizeof(u64));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<WHICH><<<blocks, 64>>>(out, 10, 0x9E3779B97F4A7C15ull);
    hipEventRecord(e0);
    rate_kernel<WHICH><<<blocks, 64>>>(out, iters, 0x9E3779B97F4A7C15ull);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return (double)blocks * iters * per_iteration / (ms * 1e-3);
}

// the lane map of v_mfma_i32_16x16x64_i8 with exact data: C = A B for random A (16 x 64), B (64 x 16) under the assumed map
//   A: lane l holds A[row l & 15][k = 16 (l >> 4) + j], j = 0 .. 15 (16 bytes);  B: lane l holds B[k = 16 (l >> 4) + j][col l & 15]
//   C: lane l register r holds C[row 4 (l >> 4) + r][col l & 15]
__global__ void __launch_bounds__(64) map_kernel(const signed char* A, const signed char* B, int* C) {
    const int l = threadIdx.x;
    v4i a, b;
    signed char* pa = (signed char*)&a;
    signed char* pb = (signed char*)&b;
    for (int j = 0; j < 16; ++j) {
        pa[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
        pb[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
    }
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

int main() {
    std::vector<signed char> A(16 * 64), B(64 * 16);
    u64 s = 12345;
    auto next = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (signed char)(s >> 56); };
    for (auto& v : A) v = next();
    for (auto& v : B) v = next();
    signed char *dA, *dB;
    int* dC;
    hipMalloc(&dA, A.size());
    hipMalloc(&dB, B.size());
    hipMalloc(&dC, 256 * sizeof(int));
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    map_kernel<<<1, 64>>>(dA, dB, dC);
    std::vector<int> C(256);
    hipMemcpy(C.data(), dC, 256 * sizeof(int), hipMemcpyDeviceToHost);
    int wrong = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            int ref = 0;
            for (int k = 0; k < 64; ++k) ref += (int)A[i * 64 + k] * (int)B[k * 16 + j];
            wrong += ref != C[i * 16 + j];
        }
    std::printf("v_mfma_i32_16x16x64_i8 lane map (A row l&15, k 16(l>>4)+j; C row 4(l>>4)+r, col l&15): %d of 256 results wrong\n", wrong);
    for (int w : {1, 2, 4}) {
        const double mad = run<0>(w, 2000, 64.0 * 64);       // 64 instructions x 64 lanes
        const double blk = run<1>(w, 4000, 16.0 * 64);       // 16 word products x 64 lanes
        const double fma = run<2>(w, 2000, 64.0 * 64);
        const double mfma = run<3>(w, 500, 64.0 * 16384);    // 64 MFMAs x 16384 MACs
        std::printf("waves/SIMD %d: v_mad_u64_u32 %.2f T lane-ops/s (%.1f cycles per wave instruction per SIMD at 2.4 GHz) | 4x4 block of u128 products %.3f T word products/s | "
                    "v_fma_f64 %.2f T lane-ops/s | mfma_i32_16x16x64_i8 %.1f T MAC/s = %.2f T word-product equivalents/s\n",
                    w, mad / 1e12, 65536.0 * 2.4e9 / mad, blk / 1e12, fma / 1e12, mfma / 1e12, mfma / 64 / 1e12);
    }
    return 0;
}
