// The grid barrier of the cooperative launches (exact.hip's pivot loop; tests/test_gpu_grid_barrier.py drives it by itself through
// relp_debug_grid_barrier).  No reference counterpart: relp is single-threaded; this is what lets ONE launch run a whole solve.
//
// cooperative_groups' grid.sync() costs 0.1 us per workgroup on gfx950 -- every arrival is an atomic on one word: 26 us at 256
// workgroups, 53 at 512 (tools/micro/grid_barrier_bench.hip, profiles/r5_micro_grid_barrier.txt) -- and a pivot makes about sixteen of
// them.  First two levels (grid_barrier_two_level: groups of 32 workgroups on a word of their own, the last arrival of a group on the
// top word, the last arrival there publishes the generation everybody polls: 6.6 us at 256 workgroups, 11.5 at 512), now the levels
// of the chip (grid_barrier): what makes a barrier expensive is not the counting -- 2 us without fences -- but 512 release fences,
// each the write-back of a die's L2.  The workgroups of one XCD (HW_REG_XCC_ID: nothing is assumed about the placement) count on a
// word of their die; their stores are in THAT die's L2 when they arrive (every wave waits for its stores in front of the workgroup
// barrier: stores count in vmcnt on gfx9, and the CU's L1 is write-through), so ONE release by the die's last arrival serves them all:
// 8 write-backs a barrier.  That workgroup counts on the top word, waits for the generation and passes it on to its die's generation
// word, which the others poll; every workgroup invalidates its own CU's L1 (the acquire).  6.5 us at 512 workgroups, 4.6 at 256.
// Counters only grow (no reset to race with); every workgroup must call it the same number of times (`epoch`).
// (The explicit waits: the compiler may drop the wait behind a release fence that follows a returned atomic, and the invalidate of an
//  acquire completes asynchronously -- MI355X guide, inter-workgroup visibility.)
//
// Round 6: a WATCHDOG.  A workgroup that waits longer than the limit (words[EX_BARRIER_LIMIT] ticks of the 100 MHz clock, 0: ten
// seconds) -- because another one made a different number of barriers -- raises words[EX_BARRIER_ABORT], every waiting workgroup sees
// that within a thousand polls, leaves the barrier it stands in at words[EX_BARRIER_STUCK + block] and ENDS (s_endpgm by every wave
// behind the workgroup barrier): the launch returns instead of hanging the device, the host finds the abort word set and reports
// which workgroups stood where.  (A workgroup that spins somewhere else is not helped by this; one that waits at a barrier is.)
#pragma once
#include <hip/hip_runtime.h>

namespace relp {

constexpr int EX_BARRIER_GROUP = 32;
constexpr int EX_BARRIER_DIE_WORDS = 2048;  // the words of the per-die barrier: [+0] generation, [+16] top, [+32 + 16 x] arrivals of die x,
                                            // [+32 + 16 (8 + x)] its generation, [+32 + 16 (16 + x)] the workgroups on it
constexpr int EX_BARRIER_COUNTER_WORDS = EX_BARRIER_DIE_WORDS + 32 + 16 * 24;  // [0] generation, [16] top, [32 + 16 g] group g of the two-level barrier (the launch's first)
constexpr int EX_BARRIER_MAX_GRID = 1024;
constexpr int EX_BARRIER_ABORT = EX_BARRIER_COUNTER_WORDS;       // != 0: a workgroup gave up waiting (the barrier's number, from 1)
constexpr int EX_BARRIER_LIMIT = EX_BARRIER_COUNTER_WORDS + 1;   // the watchdog's limit in ticks of 10 ns, low and high word (0: the default); set by the host
constexpr int EX_BARRIER_STUCK = EX_BARRIER_COUNTER_WORDS + 16;  // [+ block] the barrier (from 1) the workgroup stood in when the launch was given up
constexpr int EX_BARRIER_WORDS = EX_BARRIER_STUCK + EX_BARRIER_MAX_GRID;
constexpr unsigned long long EX_BARRIER_DEFAULT_LIMIT = 1000000000ull;  // ten seconds

struct BarrierPlace {  // where this workgroup stands: its die, the workgroups on it, the dies in use
    unsigned die = 0, members = 0, dies = 0;
};

// thread 0 of a workgroup waits for *word >= target; false: the launch is given up (by this workgroup's watchdog or another's)
__device__ __forceinline__ bool barrier_wait(unsigned* word, unsigned target, unsigned* words, unsigned epoch) {
    unsigned polls = 0;
    unsigned long long since = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++polls & 1023u) != 0) continue;
        bool give_up = __hip_atomic_load(words + EX_BARRIER_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (!give_up) {
            const unsigned long long now = wall_clock64();
            if (since == 0) since = now;
            else {
                unsigned long long limit = ((unsigned long long)__hip_atomic_load(words + EX_BARRIER_LIMIT + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) |
                                           __hip_atomic_load(words + EX_BARRIER_LIMIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (limit == 0) limit = EX_BARRIER_DEFAULT_LIMIT;
                if (now - since > limit) {
                    __hip_atomic_store(words + EX_BARRIER_ABORT, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    give_up = true;
                }
            }
        }
        if (give_up) {
            if (blockIdx.x < (unsigned)EX_BARRIER_MAX_GRID)
                __hip_atomic_store(words + EX_BARRIER_STUCK + blockIdx.x, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}
// every thread, behind the workgroup barrier that follows thread 0's wait: the whole workgroup ends when the launch was given up
__device__ __forceinline__ void barrier_leave(const int* given_up) {
    if (*(volatile const int*)given_up != 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_endpgm" ::: "memory");
    }
}

__device__ __forceinline__ void grid_barrier_two_level(unsigned* words, unsigned& epoch) {
    __shared__ int s_given_up;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned g = blockIdx.x / EX_BARRIER_GROUP, groups = (gridDim.x + EX_BARRIER_GROUP - 1) / EX_BARRIER_GROUP;
        const unsigned members = min((unsigned)EX_BARRIER_GROUP, gridDim.x - g * EX_BARRIER_GROUP);
        const unsigned arrived = __hip_atomic_fetch_add(words + 32 + 16 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (epoch + 1) * members - 1) {  // the last of its group
            const unsigned at_top = __hip_atomic_fetch_add(words + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * groups - 1) __hip_atomic_store(words, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool through = barrier_wait(words, epoch + 1, words, epoch);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_given_up = through ? 0 : 1;
    }
    ++epoch;
    __syncthreads();
    barrier_leave(&s_given_up);
}
// (once per launch: who shares a die, counted behind a barrier of the other kind)
__device__ __forceinline__ BarrierPlace grid_barrier_place(unsigned* words) {
    BarrierPlace place;
    place.die = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;  // HW_REG_XCC_ID
    unsigned* die_words = words + EX_BARRIER_DIE_WORDS;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(die_words + 32 + 16 * (16 + place.die), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned first = 0;
    grid_barrier_two_level(words, first);
    place.members = __hip_atomic_load(die_words + 32 + 16 * (16 + place.die), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned x = 0; x < 8; ++x)
        place.dies += __hip_atomic_load(die_words + 32 + 16 * (16 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1u : 0u;
    return place;
}
__device__ __forceinline__ void grid_barrier(unsigned* words, unsigned& epoch, const BarrierPlace place) {
    __shared__ int s_given_up;
    unsigned* die_words = words + EX_BARRIER_DIE_WORDS;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every wave: its stores are in the die's L2)
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* generation_of_die = die_words + 32 + 16 * (8 + place.die);
        const unsigned arrived = __hip_atomic_fetch_add(die_words + 32 + 16 * place.die, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool through;
        if (arrived == (epoch + 1) * place.members - 1) {  // the last of its die
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned at_top = __hip_atomic_fetch_add(die_words + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at_top == (epoch + 1) * place.dies - 1) __hip_atomic_store(die_words, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            through = barrier_wait(die_words, epoch + 1, words, epoch);
            if (through) __hip_atomic_store(generation_of_die, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            through = barrier_wait(generation_of_die, epoch + 1, words, epoch);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_given_up = through ? 0 : 1;
    }
    ++epoch;
    __syncthreads();
    barrier_leave(&s_given_up);
}

}  // namespace relp
