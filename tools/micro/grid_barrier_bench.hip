// cooperative_groups' grid.sync() costs 0.1 us PER WORKGROUP on gfx950 (25 us at 256 workgroups, 50 at 512: profiles/r1_micro_grid_sync.txt)
// -- every workgroup's arrival is an atomic on one word.  The exact simplex makes ~19 grid barriers per pivot on 512 workgroups: a
// millisecond.  Here: the same exchange test (every workgroup publishes a value, everybody reads all of them, twice per iteration)
// with (0) grid.sync(), (1) a hand-rolled flat barrier (monotonic counter), (2) a two-level one (groups of GROUP workgroups count on a
// word of their own, the last arrival of a group counts on the top word, the last arrival there bumps the generation everybody polls).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbb tools/micro/grid_barrier_bench.hip && /tmp/gbb
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Barrier {           // 64-byte spaced words (device memory, zeroed before a launch)
    unsigned* generation;  // [0]
    unsigned* top;         // [16]
    unsigned* group;       // [32 + 16 g]
};
constexpr int GROUP = 32;

__device__ __forceinline__ void barrier_flat(const Barrier b, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned target = (epoch + 1) * gridDim.x;
        __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(b.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}
__device__ __forceinline__ void barrier_tree(const Barrier b, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned g = blockIdx.x / GROUP, groups = (gridDim.x + GROUP - 1) / GROUP;
        const unsigned members = min((unsigned)GROUP, gridDim.x - g * GROUP);
        const unsigned arrived = __hip_atomic_fetch_add(b.group + 16 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (epoch + 1) * members - 1) {  // the last of the group
            const unsigned at_top = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * groups - 1) __hip_atomic_store(b.generation, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(b.generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}

template <int KIND>
__global__ void __launch_bounds__(256) sync_kernel(double* slots, double* out, int iters, Barrier b) {
    cg::grid_group grid = cg::this_grid();
    unsigned epoch = 0;
    auto barrier = [&]() {
        if (KIND == 0) grid.sync();
        else if (KIND == 1) barrier_flat(b, epoch);
        else barrier_tree(b, epoch);
    };
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) slots[blockIdx.x] = (double)(it + blockIdx.x);
        barrier();
        double v = 0.0;
        for (int k = threadIdx.x; k < (int)gridDim.x; k += blockDim.x) v += slots[k];
        acc += v;
        barrier();
    }
    // every workgroup must have seen every value of every iteration: sum over threads of acc = sum_it sum_b (it + b)
    __shared__ double s_acc[256];
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int t = 0; t < (int)blockDim.x; ++t) total += s_acc[t];
        out[blockIdx.x] = total;
    }
}

int main() {
    double *slots, *out;
    unsigned* words;
    CHECK(hipMalloc(&slots, 4096 * 8));
    CHECK(hipMalloc(&out, 4096 * 8));
    CHECK(hipMalloc(&words, 4096 * sizeof(unsigned)));
    Barrier b{words, words + 16, words + 32};
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char* names[] = {"cooperative_groups grid.sync()", "flat counter", "two-level (groups of 32)"};
    for (int kind = 0; kind < 3; ++kind)
        for (int blocks : {64, 128, 256, 512}) {
            int iters = 2000;
            void* args[] = {&slots, &out, &iters, &b};
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemsetAsync(words, 0, 4096 * sizeof(unsigned), s));
                CHECK(hipEventRecord(e0, s));
                void* fn = kind == 0 ? (void*)sync_kernel<0> : kind == 1 ? (void*)sync_kernel<1> : (void*)sync_kernel<2>;
                CHECK(hipLaunchCooperativeKernel(fn, dim3(blocks), dim3(256), args, 0, s));
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            double host[4096];
            CHECK(hipMemcpy(host, out, blocks * 8, hipMemcpyDeviceToHost));
            double expect = 0.0;
            for (int it = 0; it < iters; ++it)
                for (int k = 0; k < blocks; ++k) expect += it + k;
            int wrong = 0;
            for (int k = 0; k < blocks; ++k) wrong += host[k] != expect;
            printf("%-32s %3d workgroups: %6.2f us per barrier, %d of %d workgroups saw a stale value\n", names[kind], blocks, ms * 1e3 / iters / 2, wrong, blocks);
        }
    return 0;
}
