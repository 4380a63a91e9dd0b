// Why do one-workgroup, latency-bound kernels (the LU carries' pivot kernel: ~70 dependent global / LDS round trips per pivot) slow each
// other -- and everything else -- down when several LPs are in flight?  A dependent chain of global loads (one wave, the chain's lines
// resident in L2) timed alone, beside a streaming kernel of G workgroups on another stream (what the pricing passes of the other LPs
// are), and beside K copies of itself (what the other LPs' pivot kernels are).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/latency_under_load tools/micro/latency_under_load.hip && /tmp/latency_under_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void __launch_bounds__(64) chase_kernel(const int* next, int steps, int* out, unsigned long long* ticks) {
    int at = threadIdx.x == 0 ? 0 : 0;
    const unsigned long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) at = next[at];  // every lane the same chain: one dependent round trip per step
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = at;
        ticks[blockIdx.x] = t1 - t0;  // 100 MHz
    }
}
__global__ void __launch_bounds__(256) stream_kernel(const double4* data, size_t count, int passes, double* out) {
    double acc = 0.0;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
            const double4 v = data[i];
            acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 12345.678) out[0] = acc;
}

int main() {
    const int nodes = 1 << 14;  // 16 K lines of 64 B = 1 MB: resident in one XCD's L2
    std::vector<int> order(nodes), next(nodes * 16);
    std::iota(order.begin(), order.end(), 0);
    std::mt19937 rng(7);
    std::shuffle(order.begin() + 1, order.end(), rng);
    for (int k = 0; k < nodes; ++k) next[(size_t)order[k] * 16] = order[(k + 1) % nodes] * 16;  // one int per 64-byte line
    int *d_next, *d_out;
    unsigned long long* d_ticks;
    hipMalloc(&d_next, next.size() * sizeof(int));
    hipMalloc(&d_out, 4096 * sizeof(int));
    hipMalloc(&d_ticks, 4096 * sizeof(unsigned long long));
    hipMemcpy(d_next, next.data(), next.size() * sizeof(int), hipMemcpyHostToDevice);
    const size_t big = (size_t)1 << 30;  // 1 GB streamed
    double4* d_big;
    double* d_sink;
    hipMalloc(&d_big, big);
    hipMalloc(&d_sink, 8);
    hipMemset(d_big, 0, big);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    const int steps = 20000;
    auto run = [&](int copies, int stream_groups, const char* what) {
        chase_kernel<<<1, 64, 0, a>>>(d_next, 2000, d_out, d_ticks);  // warm the chain into L2
        hipStreamSynchronize(a);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        if (stream_groups > 0) stream_kernel<<<stream_groups, 256, 0, b>>>(d_big, big / sizeof(double4), 64, d_sink);  // (long enough to cover the chase)
        hipEventRecord(e0, a);
        chase_kernel<<<copies, 64, 0, a>>>(d_next, steps, d_out, d_ticks);
        hipEventRecord(e1, a);
        hipStreamSynchronize(a);
        std::vector<unsigned long long> ticks(copies);
        hipMemcpy(ticks.data(), d_ticks, copies * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        hipDeviceSynchronize();
        std::sort(ticks.begin(), ticks.end());
        std::printf("%-58s dependent L2 round trip: median %6.0f ns, slowest copy %6.0f ns\n", what, 10.0 * ticks[copies / 2] / steps, 10.0 * ticks[copies - 1] / steps);
    };
    for (int warm = 0; warm < 200; ++warm) chase_kernel<<<256, 64, 0, a>>>(d_next, 20000, d_out, d_ticks);  // (half a second of work: the clock is up)
    hipStreamSynchronize(a);
    run(1, 0, "one chain alone");
    run(1, 8, "beside a streaming kernel of 8 workgroups");
    run(1, 64, "beside a streaming kernel of 64 workgroups");
    run(1, 256, "beside a streaming kernel of 256 workgroups");
    run(1, 1024, "beside a streaming kernel of 1024 workgroups");
    run(1, 4096, "beside a streaming kernel of 4096 workgroups");
    run(8, 0, "8 chains at once (one workgroup each)");
    run(32, 0, "32 chains at once");
    run(64, 0, "64 chains at once");
    run(256, 0, "256 chains at once");
    run(32, 1024, "32 chains beside a streaming kernel of 1024 workgroups");
    run(1, 0, "one chain alone (again, at the end)");
    return 0;
}
