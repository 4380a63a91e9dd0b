// cooperative_groups' grid.sync() costs 0.1 us PER WORKGROUP on gfx950 (25 us at 256 workgroups, 50 at 512: profiles/r1_micro_grid_sync.txt)
// -- every workgroup's arrival is an atomic on one word.  The exact simplex makes ~16 grid barriers per pivot on 512 workgroups.
// Here: the same exchange test (every workgroup publishes a value, everybody reads all of them, twice per iteration) with
// (0) grid.sync(), (1) a hand-rolled flat barrier (monotonic counter), (2) a two-level one (groups of GROUP workgroups count on a word
// of their own, the last arrival of a group counts on the top word, the last arrival there bumps the generation everybody polls),
// (3)-(11) variations of it (release words per group, group sizes, spacing of the words, polling), (12) the same WITHOUT fences (not a
// barrier: what the fences cost -- 2 of 11.5 us are the counting), (13) the barrier the kernel uses since: groups = the workgroups of
// one XCD, one release fence per die.  profiles/r5_micro_grid_barrier.txt
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

// (3) two levels as (2), but the release goes down the tree as well: the last arrival at the top bumps one word PER GROUP and a workgroup
//     polls its group's word -- 32 pollers per line instead of 512 on one.  (4), (5): (2) with groups of GROUP_B workgroups.
template <int GROUP_B>
__device__ __forceinline__ void barrier_tree_sized(const Barrier b, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned g = blockIdx.x / GROUP_B, groups = (gridDim.x + GROUP_B - 1) / GROUP_B;
        const unsigned members = min((unsigned)GROUP_B, gridDim.x - g * GROUP_B);
        const unsigned arrived = __hip_atomic_fetch_add(b.group + 16 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (epoch + 1) * members - 1) {
            const unsigned at_top = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * groups - 1) __hip_atomic_store(b.generation, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(b.generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}
__device__ __forceinline__ void barrier_tree_release(const Barrier b, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned g = blockIdx.x / GROUP, groups = (gridDim.x + GROUP - 1) / GROUP;
        const unsigned members = min((unsigned)GROUP, gridDim.x - g * GROUP);
        unsigned* release = b.group + 16 * 64;  // a word per group, 64 bytes apart
        const unsigned arrived = __hip_atomic_fetch_add(b.group + 16 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (epoch + 1) * members - 1) {
            const unsigned at_top = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * groups - 1)
                for (unsigned k = 0; k < groups; ++k) __hip_atomic_store(release + 16 * k, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(release + 16 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}
// (6) as (2) with the groups made of the workgroups of one XCD's share: workgroup w is in group w % groups (consecutive workgroups go to
//     different XCDs, so a group's counter is touched from one die)
__device__ __forceinline__ void barrier_tree_strided(const Barrier b, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned groups = (gridDim.x + GROUP - 1) / GROUP, g = blockIdx.x % groups;
        const unsigned members = gridDim.x / groups + (g < gridDim.x % groups ? 1u : 0u);
        const unsigned arrived = __hip_atomic_fetch_add(b.group + 16 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (epoch + 1) * members - 1) {
            const unsigned at_top = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * groups - 1) __hip_atomic_store(b.generation, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(b.generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}
// (7) as (2) with a longer sleep between the polls of the generation (fewer reads in the way of the arrivals)
__device__ __forceinline__ void barrier_tree_patient(const Barrier b, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned g = blockIdx.x / GROUP, groups = (gridDim.x + GROUP - 1) / GROUP;
        const unsigned members = min((unsigned)GROUP, gridDim.x - g * GROUP);
        const unsigned arrived = __hip_atomic_fetch_add(b.group + 16 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (epoch + 1) * members - 1) {
            const unsigned at_top = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * groups - 1) __hip_atomic_store(b.generation, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(b.generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}

// (8), (9): (3) with the groups' words SPACING words apart (256 bytes / 4 KB: other memory channels than their neighbours')
template <int SPACING, int GROUP_B, bool FENCES = true>
__device__ __forceinline__ void barrier_tree_spaced(const Barrier b, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned g = blockIdx.x / GROUP_B, groups = (gridDim.x + GROUP_B - 1) / GROUP_B;
        const unsigned members = min((unsigned)GROUP_B, gridDim.x - g * GROUP_B);
        unsigned* counters = b.group + 32 * 64;                  // (past the words of the other variants)
        unsigned* release = counters + (size_t)SPACING * 64 + 16;  // a word per group
        const unsigned arrived = __hip_atomic_fetch_add(counters + (size_t)SPACING * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (epoch + 1) * members - 1) {
            const unsigned at_top = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * groups - 1)
                for (unsigned k = 0; k < groups; ++k) __hip_atomic_store(release + (size_t)SPACING * k, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(release + (size_t)SPACING * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1) __builtin_amdgcn_s_sleep(1);
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}

// (13) groups = the workgroups of one XCD (HW_REG_XCC_ID: no assumption on the placement).  Their stores are in THEIR die's L2 when they
//      arrive (every wave's vmcnt(0) in front of the workgroup barrier), so ONE release fence -- the write-back of that L2 -- by the
//      last arrival of the die serves them all: 8 write-backs a barrier instead of one per workgroup.  Every workgroup still invalidates
//      its own CU's L1 (the acquire) behind its die's generation word.
struct XcdBarrier {
    unsigned* words;  // [0] generation, [16] top, [32 + 16 x] arrivals of die x, [32 + 16 (8 + x)] generation of die x, [32 + 16 (16 + x)] workgroups on die x
    unsigned die, members, dies;
};
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }  // HW_REG_XCC_ID, 4 bits
__device__ __forceinline__ void barrier_xcd(const XcdBarrier b, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* arrivals = b.words + 32 + 16 * b.die;
        unsigned* generation_of_die = b.words + 32 + 16 * (8 + b.die);
        const unsigned arrived = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (epoch + 1) * b.members - 1) {  // the last of its die
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned at_top = __hip_atomic_fetch_add(b.words + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * b.dies - 1) __hip_atomic_store(b.words, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(b.words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(generation_of_die, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(generation_of_die, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch + 1) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++epoch;
    __syncthreads();
}

template <int KIND>
__global__ void __launch_bounds__(256) sync_kernel(double* slots, double* out, int iters, Barrier b) {
    cg::grid_group grid = cg::this_grid();
    unsigned epoch = 0;
    XcdBarrier xb{b.generation + 8192, 0, 0, 0};  // (words of its own)
    if (KIND == 13) {  // who shares a die: counted once, behind an ordinary barrier
        xb.die = xcc_id();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(xb.words + 32 + 16 * (16 + xb.die), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned setup_epoch = 0;
        Barrier setup{b.generation + 4096, b.generation + 4096 + 16, b.generation + 4096 + 32};
        barrier_tree(setup, setup_epoch);
        xb.members = __hip_atomic_load(xb.words + 32 + 16 * (16 + xb.die), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned x = 0; x < 8; ++x) xb.dies += __hip_atomic_load(xb.words + 32 + 16 * (16 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1u : 0u;
    }
    auto barrier = [&]() {
        if (KIND == 0) grid.sync();
        else if (KIND == 1) barrier_flat(b, epoch);
        else if (KIND == 2) barrier_tree(b, epoch);
        else if (KIND == 3) barrier_tree_release(b, epoch);
        else if (KIND == 4) barrier_tree_sized<16>(b, epoch);
        else if (KIND == 5) barrier_tree_sized<64>(b, epoch);
        else if (KIND == 6) barrier_tree_strided(b, epoch);
        else if (KIND == 7) barrier_tree_patient(b, epoch);
        else if (KIND == 8) barrier_tree_spaced<64, 32>(b, epoch);
        else if (KIND == 9) barrier_tree_spaced<1024, 32>(b, epoch);
        else if (KIND == 10) barrier_tree_spaced<1024, 64>(b, epoch);
        else if (KIND == 11) barrier_tree_spaced<1024, 16>(b, epoch);
        else if (KIND == 12) barrier_tree_spaced<64, 32, false>(b, epoch);  // (NOT a barrier for data: what the fences cost)
        else barrier_xcd(xb, epoch);
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
    CHECK(hipMalloc(&words, (size_t)(1 << 20) * sizeof(unsigned)));
    Barrier b{words, words + 16, words + 32};
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char* names[] = {"cooperative_groups grid.sync()", "flat counter", "two-level (groups of 32)", "two-level, release per group", "two-level (groups of 16)",
                           "two-level (groups of 64)", "two-level, groups strided", "two-level, sleep 8 between polls", "release per group, 256 B apart", "release per group, 4 KB apart",
                           "... groups of 64, 4 KB apart", "... groups of 16, 4 KB apart", "(no fences: not a barrier)", "per XCD, one release per die"};
    for (int kind = 0; kind < 14; ++kind)
        for (int blocks : {64, 128, 256, 512}) {
            int iters = 2000;
            void* args[] = {&slots, &out, &iters, &b};
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemsetAsync(words, 0, (size_t)(1 << 20) * sizeof(unsigned), s));
                CHECK(hipEventRecord(e0, s));
                void* fns[] = {(void*)sync_kernel<0>, (void*)sync_kernel<1>, (void*)sync_kernel<2>, (void*)sync_kernel<3>, (void*)sync_kernel<4>, (void*)sync_kernel<5>,
                               (void*)sync_kernel<6>, (void*)sync_kernel<7>, (void*)sync_kernel<8>, (void*)sync_kernel<9>, (void*)sync_kernel<10>, (void*)sync_kernel<11>, (void*)sync_kernel<12>, (void*)sync_kernel<13>};
                void* fn = fns[kind];
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
