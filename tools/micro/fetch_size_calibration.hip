// FETCH_SIZE calibration for the access widths of price_unit_kernel (MI355X_MICROARCH.md, "HBM": the counter reports half the bytes of a
// 16-B-per-lane streaming read on gfx950; "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Kernels that stream a KNOWN number of bytes at 16, 8, 4 and 1 byte per lane (a lane per element, 256-thread workgroups, four
// elements per lane 256 apart -- the layout of price_unit_kernel's load_arcs), a gather of 8-byte entries of a 0.5 MB table by a million
// lanes (its -pi), and the three streams of the pricing pass together.  Run under `rocprofv3 --pmc FETCH_SIZE`; tools/fetch_size_calibration.py
// divides the counter by the bytes each kernel asked for.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_size_calibration fetch_size_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

template <typename T> __device__ unsigned fold(T v);
template <> __device__ unsigned fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <> __device__ unsigned fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <> __device__ unsigned fold<unsigned>(unsigned v) { return v; }
template <> __device__ unsigned fold<unsigned char>(unsigned char v) { return v; }

template <typename T>
__global__ void __launch_bounds__(256) stream_kernel(const T* __restrict__ in, long long n, unsigned* out) {
    unsigned acc = 0;
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long j = base + u * 256;
        if (j < n) acc ^= fold<T>(in[j]);
    }
    if (acc == 0x12345678u) out[0] = acc;  // (never: keeps the loads)
}
__global__ void __launch_bounds__(256) stream_16_kernel(const uint4* in, long long n, unsigned* out) { }  // (name holder for the csv)

__global__ void __launch_bounds__(256) gather_kernel(const double* __restrict__ table, const unsigned* __restrict__ index, long long n, double* out) {
    double acc = 0.0;
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
    unsigned at[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) at[u] = base + u * 256 < n ? index[base + u * 256] : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += table[at[u]];
    if (acc == 1.2345e300) out[0] = acc;
}
// the three streams of the pricing pass: 8 bytes, 4 bytes and 1 byte per arc
__global__ void __launch_bounds__(256) three_streams_kernel(const uint2* __restrict__ arcs, const int* __restrict__ pos, const signed char* __restrict__ cost, long long n, unsigned* out) {
    unsigned acc = 0;
    const long long base = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long j = base + u * 256;
        if (j < n) {
            const uint2 c = arcs[j];
            acc ^= c.x ^ c.y ^ (unsigned)pos[j] ^ (unsigned)cost[j];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const long long n = 1 << 20;       // a million elements: the arcs of config 5
    const long long table = 65536;     // its rows
    void* buffer;
    CHECK(hipMalloc(&buffer, (size_t)n * 16));
    CHECK(hipMemset(buffer, 1, (size_t)n * 16));
    unsigned* out;
    CHECK(hipMalloc(&out, 64));
    double* d_table;
    CHECK(hipMalloc(&d_table, table * sizeof(double)));
    CHECK(hipMemset(d_table, 0, table * sizeof(double)));
    std::vector<unsigned> index(2 * n);
    unsigned long long state = 0x5EED0005ull;
    for (auto& v : index) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        v = (unsigned)((state >> 33) % table);
    }
    unsigned* d_index;
    CHECK(hipMalloc(&d_index, 2 * n * sizeof(unsigned)));
    CHECK(hipMemcpy(d_index, index.data(), 2 * n * sizeof(unsigned), hipMemcpyHostToDevice));
    int* d_pos;
    signed char* d_cost;
    CHECK(hipMalloc(&d_pos, n * 4));
    CHECK(hipMalloc(&d_cost, n));
    CHECK(hipMemset(d_pos, 0, n * 4));
    CHECK(hipMemset(d_cost, 0, n));
    // a 300 MB sweep between the measured kernels, so that none of them finds its input in a cache
    void* flush;
    const size_t flush_bytes = (size_t)320 << 20;
    CHECK(hipMalloc(&flush, flush_bytes));
    const dim3 grid((unsigned)((n + 1023) / 1024)), block(256);
    for (int round = 0; round < 5; ++round) {
        CHECK(hipMemsetAsync(flush, round, flush_bytes, 0));
        hipLaunchKernelGGL(stream_kernel<uint4>, grid, block, 0, 0, (const uint4*)buffer, n, out);
        CHECK(hipMemsetAsync(flush, round, flush_bytes, 0));
        hipLaunchKernelGGL(stream_kernel<uint2>, grid, block, 0, 0, (const uint2*)buffer, n, out);
        CHECK(hipMemsetAsync(flush, round, flush_bytes, 0));
        hipLaunchKernelGGL(stream_kernel<unsigned>, grid, block, 0, 0, (const unsigned*)buffer, n, out);
        CHECK(hipMemsetAsync(flush, round, flush_bytes, 0));
        hipLaunchKernelGGL(stream_kernel<unsigned char>, grid, block, 0, 0, (const unsigned char*)buffer, n, out);
        CHECK(hipMemsetAsync(flush, round, flush_bytes, 0));
        hipLaunchKernelGGL(gather_kernel, grid, block, 0, 0, d_table, d_index, n, (double*)out);
        CHECK(hipMemsetAsync(flush, round, flush_bytes, 0));
        hipLaunchKernelGGL(three_streams_kernel, grid, block, 0, 0, (const uint2*)buffer, d_pos, d_cost, n, out);
        // ... and the same kernels with their inputs warm in the Infinity Cache (the pricing pass finds its arcs there: 13.6 MB, read every pivot)
        hipLaunchKernelGGL(three_streams_kernel, grid, block, 0, 0, (const uint2*)buffer, d_pos, d_cost, n, out);
        hipLaunchKernelGGL(gather_kernel, grid, block, 0, 0, d_table, d_index, n, (double*)out);
    }
    CHECK(hipDeviceSynchronize());
    printf("bytes asked for per launch: 16 B/lane %lld, 8 B/lane %lld, 4 B/lane %lld, 1 B/lane %lld, gather %lld index + %lld table (x 8 L2s = %lld), three streams %lld\n",
           n * 16, n * 8, n * 4, n, n * 4, table * 8, table * 64, n * 13);
    return 0;
}
