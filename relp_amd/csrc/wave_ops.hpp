// Wave64 / workgroup reduction helpers shared by the kernel files (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include "solver.hpp"

namespace relp {

constexpr int WAVE = 64;

#ifdef RELP_STAMPS
#define STAMP(k) do { if (threadIdx.x == 0) { unsigned long long t__ = clock64(); lp.dbg[(k)] += t__ - t_prev__; t_prev__ = t__; } } while (0)
#define STAMP_INIT unsigned long long t_prev__ = clock64(); if (threadIdx.x == 0) lp.dbg[63] += 1
#else
#define STAMP(k) do {} while (0)
#define STAMP_INIT do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------
// reductions: wave64 DPP (data-parallel primitives) moves instead of ds_bpermute shuffles.  A __shfl_down chain is
// six DEPENDENT LDS-crossbar round trips per value (measured: 3-4 k cycles per struct arg-max); the DPP sequence
// quad_perm -> quad_perm -> row_ror:4 -> row_ror:8 -> row_bcast:15 -> row_bcast:31 runs at VALU speed and leaves the
// result in lane 63.  Fixed combination order => deterministic results.
// ---------------------------------------------------------------------------------------------------
struct Cand {
    double key;
    int idx;  // -1: empty
    int aux;
};

enum : int { TIE_LARGER_IDX = 0, TIE_SMALLER_IDX = 1, TIE_SMALLER_AUX = 2 };

// Branch-free selection on scalar fields: keeps candidates in registers (a by-reference version put them in scratch
// memory and cost 6-7 k cycles per block-wide arg-max).
template <int TIE>
__device__ __forceinline__ Cand better(Cand a, Cand b) {
    bool take_b;
    if (TIE == TIE_LARGER_IDX) take_b = (b.key > a.key) | ((b.key == a.key) & (b.idx > a.idx));
    else if (TIE == TIE_SMALLER_IDX) take_b = (b.key > a.key) | ((b.key == a.key) & (b.idx < a.idx));
    else take_b = (b.key > a.key) | ((b.key == a.key) & ((b.aux < a.aux) | ((b.aux == a.aux) & (b.idx < a.idx))));
    take_b = (a.idx < 0) | ((b.idx >= 0) & take_b);
    Cand r;
    r.key = take_b ? b.key : a.key;
    r.idx = take_b ? b.idx : a.idx;
    r.aux = take_b ? b.aux : a.aux;
    return r;
}

constexpr int DPP_QUAD_1032 = 0xB1;    // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_2301 = 0x4E;    // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i <-> 7 - i inside each group of 8
constexpr int DPP_ROW_ROR4 = 0x124;    // row_ror:4
constexpr int DPP_ROW_ROR8 = 0x128;    // row_ror:8
constexpr int DPP_ROW_BCAST15 = 0x142; // lane 15 of each row -> every lane of the next row
constexpr int DPP_ROW_BCAST31 = 0x143; // lane 31 -> every lane of rows 2 and 3

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(old), __double2loint(v));
    const int hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(old), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Cand dpp_cand(Cand v) {  // masked-off rows see their own value (idempotent op)
    Cand o;
    o.key = dpp_f64<CTRL, ROW_MASK>(v.key, v.key);
    o.idx = dpp_i32<CTRL, ROW_MASK>(v.idx, v.idx);
    o.aux = dpp_i32<CTRL, ROW_MASK>(v.aux, v.aux);
    return o;
}

// result valid in lane 63
template <int TIE>
__device__ __forceinline__ Cand wave_best(Cand v) {
    v = better<TIE>(v, dpp_cand<DPP_QUAD_1032, 0xF>(v));
    v = better<TIE>(v, dpp_cand<DPP_QUAD_2301, 0xF>(v));
    v = better<TIE>(v, dpp_cand<DPP_ROW_ROR4, 0xF>(v));
    v = better<TIE>(v, dpp_cand<DPP_ROW_ROR8, 0xF>(v));
    v = better<TIE>(v, dpp_cand<DPP_ROW_BCAST15, 0xA>(v));
    v = better<TIE>(v, dpp_cand<DPP_ROW_BCAST31, 0xC>(v));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {  // lane 63
    v += dpp_f64<DPP_QUAD_1032, 0xF>(0.0, v);
    v += dpp_f64<DPP_QUAD_2301, 0xF>(0.0, v);
    v += dpp_f64<DPP_ROW_ROR4, 0xF>(0.0, v);
    v += dpp_f64<DPP_ROW_ROR8, 0xF>(0.0, v);
    v += dpp_f64<DPP_ROW_BCAST15, 0xA>(0.0, v);
    v += dpp_f64<DPP_ROW_BCAST31, 0xC>(0.0, v);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {  // lane 63
    v = fmin(v, dpp_f64<DPP_QUAD_1032, 0xF>(v, v));
    v = fmin(v, dpp_f64<DPP_QUAD_2301, 0xF>(v, v));
    v = fmin(v, dpp_f64<DPP_ROW_ROR4, 0xF>(v, v));
    v = fmin(v, dpp_f64<DPP_ROW_ROR8, 0xF>(v, v));
    v = fmin(v, dpp_f64<DPP_ROW_BCAST15, 0xA>(v, v));
    v = fmin(v, dpp_f64<DPP_ROW_BCAST31, 0xC>(v, v));
    return v;
}
constexpr int LAST = WAVE - 1;

// ---- cheap block-wide arg-max: max of the f64 key, then min of a 64-bit rank among the ties ----------------------
// A candidate is (key, rank); an empty one has key = -inf.  Two scalar DPP reductions (18 + 30 VALU instructions)
// replace the struct reduction (about 150), the wave results go through LDS once and EVERY thread scans them, so there is
// one barrier instead of three.  rank encodes the tie rule (smaller wins) and may carry a payload in its low bits.
__device__ __forceinline__ double wave_max(double v) {  // lane 63
    v = fmax(v, dpp_f64<DPP_QUAD_1032, 0xF>(v, v));
    v = fmax(v, dpp_f64<DPP_QUAD_2301, 0xF>(v, v));
    v = fmax(v, dpp_f64<DPP_ROW_ROR4, 0xF>(v, v));
    v = fmax(v, dpp_f64<DPP_ROW_ROR8, 0xF>(v, v));
    v = fmax(v, dpp_f64<DPP_ROW_BCAST15, 0xA>(v, v));
    v = fmax(v, dpp_f64<DPP_ROW_BCAST31, 0xC>(v, v));
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const int lo = dpp_i32<CTRL, ROW_MASK>((int)(unsigned)v, (int)(unsigned)v);
    const int hi = dpp_i32<CTRL, ROW_MASK>((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32));
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long umin64(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {  // lane 63
    v = umin64(v, dpp_u64<DPP_QUAD_1032, 0xF>(v));
    v = umin64(v, dpp_u64<DPP_QUAD_2301, 0xF>(v));
    v = umin64(v, dpp_u64<DPP_ROW_ROR4, 0xF>(v));
    v = umin64(v, dpp_u64<DPP_ROW_ROR8, 0xF>(v));
    v = umin64(v, dpp_u64<DPP_ROW_BCAST15, 0xA>(v));
    v = umin64(v, dpp_u64<DPP_ROW_BCAST31, 0xC>(v));
    return v;
}
__device__ __forceinline__ double lane63_f64(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned long long lane63_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
constexpr unsigned long long RANK_NONE = ~0ull;
// s_key / s_rank: one slot per wave.  Returns the winner in (key, rank) for every thread; rank == RANK_NONE: none.
__device__ __forceinline__ void block_argbest(double& key, unsigned long long& rank, double* s_key, unsigned long long* s_rank) {
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int nwaves = (blockDim.x + WAVE - 1) / WAVE;
    const double k0 = rank == RANK_NONE ? -INFINITY : key;
    const double wmax = lane63_f64(wave_max(k0));
    const unsigned long long tie = (rank != RANK_NONE && k0 == wmax) ? rank : RANK_NONE;
    const unsigned long long wmin = lane63_u64(wave_min_u64(tie));
    __syncthreads();  // previous users of the slots are done
    if (lane == 0) {
        s_key[wave] = wmax;
        s_rank[wave] = wmin;
    }
    __syncthreads();
    double bk = -INFINITY;
    unsigned long long br = RANK_NONE;
    for (int wv = 0; wv < nwaves; ++wv) {
        const double k = s_key[wv];
        const unsigned long long r = s_rank[wv];
        const bool take = r != RANK_NONE && (br == RANK_NONE || k > bk || (k == bk && r < br));
        bk = take ? k : bk;
        br = take ? r : br;
    }
    key = bk;
    rank = br;
}

// Block-wide argmax; result valid in every thread.  `s` holds at least blockDim.x/64 + 1 entries.
template <int TIE>
__device__ __forceinline__ Cand block_best(Cand v, Cand* s) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int nwaves = (blockDim.x + WAVE - 1) / WAVE;
    v = wave_best<TIE>(v);
    __syncthreads();
    if (lane == LAST) {
        s[wave].key = v.key;
        s[wave].idx = v.idx;
        s[wave].aux = v.aux;
    }
    __syncthreads();
    if (wave == 0) {
        Cand t;
        t.key = 0.0;
        t.idx = -1;
        t.aux = 0;
        if (lane < nwaves) {
            t.key = s[lane].key;
            t.idx = s[lane].idx;
            t.aux = s[lane].aux;
        }
        t = wave_best<TIE>(t);
        if (lane == LAST) {
            s[nwaves].key = t.key;
            s[nwaves].idx = t.idx;
            s[nwaves].aux = t.aux;
        }
    }
    __syncthreads();
    Cand r;
    r.key = s[nwaves].key;
    r.idx = s[nwaves].idx;
    r.aux = s[nwaves].aux;
    return r;
}

// op: 0 sum, 1 min.  Deterministic order (fixed tree).  `s` holds blockDim.x/64 + 1 doubles.
template <int OP>
__device__ __forceinline__ double block_reduce(double v, double* s) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int nwaves = (blockDim.x + WAVE - 1) / WAVE;
    v = OP == 0 ? wave_sum(v) : wave_min(v);
    __syncthreads();
    if (lane == LAST) s[wave] = v;
    __syncthreads();
    if (wave == 0) {
        double t = OP == 0 ? 0.0 : INFINITY;
        if (lane < nwaves) t = s[lane];
        t = OP == 0 ? wave_sum(t) : wave_min(t);
        if (lane == LAST) s[nwaves] = t;
    }
    __syncthreads();
    return s[nwaves];
}

}  // namespace relp
