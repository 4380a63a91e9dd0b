// relp_debug_grid_barrier: the grid barrier of the exact simplex (grid_barrier.hpp) under an exchange test of its own, inside the
// suite (tests/test_gpu_grid_barrier.py).  No reference counterpart (relp is single-threaded).
//
// Exchange test (mode 0): in round r EVERY thread of every workgroup stores a fresh value (a hash of round, workgroup and thread) into
// the slot of its workgroup -- two sets of slots, taken in turn --, the grid meets at the barrier, and every thread reads the slots of
// `reads` other workgroups (the neighbours b + 1 + r * reads .. of its own: consecutive workgroups sit on different dies, so every
// round reads from every die, and the offset rotates through all pairs) and counts what is not this round's value.  A barrier that lets
// a workgroup through before another one's stores are visible to it shows up as a stale value (the previous use of the slot, two
// rounds ago) -- the argument "a die's last arrival releases for all of them" is checked here, not taken on trust.
// Mode 1: workgroup grid - 1 leaves one barrier out half-way: the watchdog has to end the launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "grid_barrier.hpp"
#include "solver.hpp"

namespace relp {

namespace {

constexpr int GB_THREADS = 256;

__device__ __forceinline__ unsigned gb_value(unsigned round, unsigned block, unsigned thread) {
    unsigned x = round * 0x9E3779B1u + block * 0x85EBCA77u + thread * 0xC2B2AE3Du + 0x27D4EB2Fu;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 12;
    return x | 1u;  // (never the zero the slots start with)
}

__global__ void __launch_bounds__(GB_THREADS) grid_barrier_test_kernel(unsigned* barrier, unsigned* slots, int rounds, int reads, int mode, unsigned long long* out) {
    unsigned epoch = 0;
    const BarrierPlace place = grid_barrier_place(barrier);
    const unsigned G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    unsigned long long bad = 0, first_bad = ~0ull;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        unsigned* set = slots + (size_t)(r & 1) * G * GB_THREADS;
        set[(size_t)b * GB_THREADS + tid] = gb_value((unsigned)r, b, tid);
        if (mode == 1 && b == G - 1 && r == rounds / 2) continue;  // (one barrier fewer than the others: they wait for it in vain)
        grid_barrier(barrier, epoch, place);
        for (int k = 0; k < reads; ++k) {
            const unsigned other = (b + 1u + (unsigned)(((unsigned long long)r * reads + k) % (G > 1 ? G - 1 : 1))) % G;
            // (a plain load: what the kernels behind the barrier do)
            const unsigned seen = set[(size_t)other * GB_THREADS + tid];
            if (seen != gb_value((unsigned)r, other, tid)) {
                ++bad;
                if (first_bad == ~0ull) first_bad = ((unsigned long long)r << 32) | ((unsigned long long)b << 16) | other;
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (bad != 0) {
        atomicAdd(&out[0], bad);
        atomicMin(&out[1], first_bad);
    }
    if (b == 0 && tid == 0) {
        out[2] = place.dies;
        out[3] = t1 - t0;  // ticks of 10 ns, the whole loop
        out[4] = (unsigned long long)rounds;
    }
    if (tid == 0) atomicAdd(&out[5], 1ull);  // workgroups that ran to the end
}

}  // namespace

// out[0] stale values, [1] the first one as round << 32 | reader << 16 | writer (or ~0), [2] dies in use, [3] ticks of 10 ns for the loop of
// workgroup 0, [4] rounds, [5] workgroups that ran to the end, [6] the barrier's abort word (0: nobody gave up), [7] workgroups found waiting
// when the launch was given up
void grid_barrier_test(int device, int grid, int rounds, int reads, int mode, long long limit_ticks, long long* out8) {
    RELP_HIP(hipSetDevice(device));
    std::vector<void*> owned;
    struct Free {
        std::vector<void*>& p;
        ~Free() { for (void* q : p) (void)hipFree(q); }
    } free_all{owned};
    auto dalloc_bytes = [&](size_t bytes) {
        void* p = nullptr;
        RELP_HIP(hipMalloc(&p, std::max<size_t>(bytes, 8)));
        owned.push_back(p);
        return p;
    };
    int per_cu = 0, cus = 0;
    RELP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)grid_barrier_test_kernel, GB_THREADS, 0));
    RELP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    if (grid < 1 || grid > EX_BARRIER_MAX_GRID || grid > per_cu * cus) throw std::invalid_argument("grid_barrier_test: the grid does not fit the device");
    unsigned* d_barrier = (unsigned*)dalloc_bytes(EX_BARRIER_WORDS * sizeof(unsigned));
    unsigned* d_slots = (unsigned*)dalloc_bytes((size_t)2 * grid * GB_THREADS * sizeof(unsigned));
    unsigned long long* d_out = (unsigned long long*)dalloc_bytes(8 * sizeof(unsigned long long));
    std::vector<unsigned> words(EX_BARRIER_WORDS, 0u);
    words[EX_BARRIER_LIMIT] = (unsigned)((unsigned long long)limit_ticks & 0xffffffffull);
    words[EX_BARRIER_LIMIT + 1] = (unsigned)((unsigned long long)limit_ticks >> 32);
    RELP_HIP(hipMemcpy(d_barrier, words.data(), words.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    RELP_HIP(hipMemset(d_slots, 0, (size_t)2 * grid * GB_THREADS * sizeof(unsigned)));
    unsigned long long host_out[8] = {0, ~0ull, 0, 0, 0, 0, 0, 0};
    RELP_HIP(hipMemcpy(d_out, host_out, sizeof(host_out), hipMemcpyHostToDevice));
    void* args[] = {(void*)&d_barrier, (void*)&d_slots, (void*)&rounds, (void*)&reads, (void*)&mode, (void*)&d_out};
    RELP_HIP(hipLaunchCooperativeKernel((const void*)grid_barrier_test_kernel, dim3(grid), dim3(GB_THREADS), args, 0, nullptr));
    RELP_HIP(hipGetLastError());
    RELP_HIP(hipDeviceSynchronize());
    RELP_HIP(hipMemcpy(host_out, d_out, sizeof(host_out), hipMemcpyDeviceToHost));
    RELP_HIP(hipMemcpy(words.data(), d_barrier, words.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    host_out[6] = words[EX_BARRIER_ABORT];
    unsigned long long waiting = 0;
    for (int g = 0; g < grid; ++g) waiting += words[EX_BARRIER_STUCK + g] != 0 ? 1 : 0;
    host_out[7] = waiting;
    for (int k = 0; k < 8; ++k) out8[k] = (long long)host_out[k];
}

}  // namespace relp
