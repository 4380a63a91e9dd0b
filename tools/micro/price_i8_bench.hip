// Micro-benchmark (not part of the product): what bounds the byte-block pricing pass (price_dense_kernel<false, true>)?
// Same loop as the kernel (4096 rows x 8192 columns of signed bytes, three f64 dot products per column against vectors in
// LDS), with parts removed one at a time:
//   0 full            the kernel's inner loop
//   1 no LDS          the three vectors replaced by register constants (LDS traffic gone, FMAs and widening kept)
//   2 no widening     the bytes reinterpreted instead of converted (one cheap op per entry), LDS reads and FMAs kept
//   3 one vector      only -pi (a pass without a pending weight update)
//   4 stream only     the 16-byte loads and one add per load
// hipcc --offload-arch=gfx950 -O3 -o price_i8_bench price_i8_bench.hip && ./price_i8_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 1024, WAVE = 64, CHUNK = 1024;

template <int MODE>
__global__ void __launch_bounds__(THREADS) pass(const signed char* A, const double* pi, const double* rho, const double* w, int m, int n,
                                                double* out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_pi = smem;
    double* s_rho = smem + m;
    double* s_w = smem + 2 * m;
    for (int i = threadIdx.x; i < m; i += THREADS) {
        s_pi[i] = pi[i];
        s_rho[i] = rho[i];
        s_w[i] = w[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x / 64;
    const int waves_total = gridDim.x * (THREADS / WAVE);
    const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
    const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
    const double2* w2 = reinterpret_cast<const double2*>(s_w);
    const int chunks = m / CHUNK;
    for (int jd = blockIdx.x * (THREADS / WAVE) + wave; jd < n; jd += waves_total) {
        const i32x4* col = reinterpret_cast<const i32x4*>(A + (size_t)jd * m);
        double p0 = 0.0, p1 = 0.0, r0 = 0.0, r1 = 0.0, w0 = 0.0, w1 = 0.0;
        for (int c0 = 0; c0 < chunks; c0 += 4) {
            i32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(col + (size_t)(c0 + u) * WAVE + lane);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 4) {
                    p0 += (double)(v[u].x + v[u].y + v[u].z + v[u].w);
                    continue;
                }
                const int base2 = (c0 + u) * (CHUNK / 2) + lane;
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    double x0, x1;
                    if (MODE == 2) {
                        x0 = __hiloint2double(0x3ff00000, v[u][t / 4] >> (8 * (t % 4)));
                        x1 = __hiloint2double(0x3ff00000, v[u][(t + 1) / 4] >> (8 * ((t + 1) % 4)));
                    } else {
                        x0 = (double)((int)((unsigned)v[u][t / 4] << (24 - 8 * (t % 4))) >> 24);
                        x1 = (double)((int)((unsigned)v[u][(t + 1) / 4] << (24 - 8 * ((t + 1) % 4))) >> 24);
                    }
                    const int at = base2 + (t / 2) * WAVE;
                    if (MODE == 1) {
                        p0 += x0 * 1.25; p1 += x1 * 0.75; r0 += x0 * 1.5; r1 += x1 * 2.5; w0 += x0 * 3.5; w1 += x1 * 4.5;
                    } else {
                        const double2 vp = pi2[at];
                        p0 += x0 * vp.x;
                        p1 += x1 * vp.y;
                        if (MODE != 3) {
                            const double2 vr = rho2[at], vw = w2[at];
                            r0 += x0 * vr.x; r1 += x1 * vr.y; w0 += x0 * vw.x; w1 += x1 * vw.y;
                        }
                    }
                }
            }
        }
        double a = p0 + p1, b = r0 + r1, c = w0 + w1;
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); c += __shfl_down(c, off); }
        if (lane == 0) out[jd] = a + b * c;
    }
}

template <int MODE>
static void run(const char* name, const signed char* A, const double* pi, const double* rho, const double* w, int m, int n, double* out) {
    const size_t lds = (size_t)3 * m * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pass<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(pass<MODE>, dim3(256), dim3(THREADS), lds, 0, A, pi, rho, w, m, n, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 3 && ms < best) best = ms;
    }
    printf("%-14s %7.2f us  (%5.2f TB/s of bytes, %5.1f TB/s of LDS reads at 24 B per entry)\n", name, best * 1e3,
           (double)m * n / (best * 1e-3) / 1e12, (double)m * n * 24 / (best * 1e-3) / 1e12);
}

int main() {
    const int m = 4096, n = 8192;
    std::vector<signed char> hA((size_t)m * n);
    for (size_t k = 0; k < hA.size(); ++k) hA[k] = (signed char)(1 + (k * 2654435761u >> 7) % 100);
    std::vector<double> hv(m);
    for (int i = 0; i < m; ++i) hv[i] = 1.0 / (1 + i % 17);
    signed char* A; double *pi, *rho, *w, *out;
    CHECK(hipMalloc(&A, hA.size())); CHECK(hipMalloc(&pi, m * 8)); CHECK(hipMalloc(&rho, m * 8)); CHECK(hipMalloc(&w, m * 8)); CHECK(hipMalloc(&out, n * 8));
    CHECK(hipMemcpy(A, hA.data(), hA.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(pi, hv.data(), m * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(rho, hv.data(), m * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(w, hv.data(), m * 8, hipMemcpyHostToDevice));
    run<0>("full", A, pi, rho, w, m, n, out);
    run<1>("no LDS", A, pi, rho, w, m, n, out);
    run<2>("no widening", A, pi, rho, w, m, n, out);
    run<3>("one vector", A, pi, rho, w, m, n, out);
    run<4>("stream only", A, pi, rho, w, m, n, out);
    return 0;
}
