// `BasisInverse::invert` as a kernel: parallel-pivot sparse LU of the basis on one workgroup (see lu_factor.hpp for the design and
// the reference lines it replaces: lower_upper/mod.rs:78-92, decomposition/mod.rs:27-143,146-210, decomposition/pivoting.rs:45-81).
#include "lu_factor.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "solver.hpp"
#include "wave_ops.hpp"

namespace relp {

// =====================================================================================================
// host: work memory
// =====================================================================================================
LuFactorScratch::~LuFactorScratch() {
    if (dev_) (void)hipFree(dev_);
}

void LuFactorScratch::reserve(int m, size_t nnz_basis, size_t cap_l, size_t cap_u, size_t cap_inverse) {
    if (dev_ && m == m_ && nnz_basis <= nnz_ && cap_l <= cap_l_ && cap_u <= cap_u_ && cap_inverse <= cap_inv_) return;
    if (m != m_) nnz_ = cap_l_ = cap_u_ = cap_inv_ = 0;
    m_ = m;
    nnz_ = std::max(nnz_, nnz_basis);
    cap_l_ = std::max(cap_l_, cap_l);
    cap_u_ = std::max(cap_u_, cap_u);
    cap_inv_ = std::max(cap_inv_, cap_inverse);
    size_t cap_w = std::min<size_t>((size_t)1 << 27, 6 * nnz_ + 16 * (size_t)m + 4096);
    if (thread_tuning().luf_arena_cap > 0) cap_w = std::max<size_t>(64, (size_t)thread_tuning().luf_arena_cap);  // test hook: an arena the basis outgrows
    size_t offset = 0;
    auto take = [&](size_t bytes) {
        offset = (offset + 63) & ~size_t(63);
        const size_t at = offset;
        offset += bytes;
        return at;
    };
    const size_t mi = (size_t)m * sizeof(int);
    size_t o_acr[2], o_aval[2], o_active[2];
    for (int b = 0; b < 2; ++b) {
        o_acr[b] = take(cap_w * sizeof(unsigned));
        o_aval[b] = take(cap_w * sizeof(double));
        o_active[b] = take(mi);
    }
    const size_t o_rstart = take(mi), o_rlen = take(mi), o_rnew = take(mi), o_growth = take(mi), o_targets = take(mi), o_ccount = take(mi);
    const size_t o_rmax = take((size_t)m * sizeof(unsigned long long)), o_rmaxd = take((size_t)m * sizeof(double));
    const size_t o_rowbest = take(mi), o_beste = take(mi), o_colmark = take(mi), o_kill = take(mi), o_tflag = take(mi);
    const size_t o_pkr = take(mi), o_pkc = take(mi);
    const size_t o_utstart = take(mi + sizeof(int));
    const size_t o_utcol = take(cap_u_ * sizeof(int)), o_utrow = take(cap_u_ * sizeof(int)), o_utval = take(cap_u_ * sizeof(double));
    const size_t o_ltrow = take(cap_l_ * sizeof(int)), o_ltstep = take(cap_l_ * sizeof(int)), o_ltval = take(cap_l_ * sizeof(double));
    const size_t cap_t = std::max(cap_l_, cap_u_);
    const size_t o_tstart = take(mi + sizeof(int)), o_tcursor = take(mi + sizeof(int));
    const size_t o_tidx = take(cap_t * sizeof(int)), o_tval = take(cap_t * sizeof(double)), o_trow = take(cap_t * sizeof(int));
    const size_t o_rpos = take(mi), o_cpos = take(mi), o_rowat = take(mi), o_colat = take(mi);
    const size_t o_info = take(LUF_INFO_WORDS * sizeof(int));
    // the inversion of the two triangles and the record packing (lu_device_tasks.hip)
    const size_t ci = cap_inv_;
    const size_t o_rawcol = take(2 * ci * sizeof(int)), o_rawval = take(2 * ci * sizeof(double));
    size_t o_rawstart[2], o_rawlen[2], o_rawdesc[2], o_cstart[4], o_cidx[4], o_cval[4];
    for (int f = 0; f < 2; ++f) {
        o_rawstart[f] = take(mi);
        o_rawlen[f] = take(mi);
        o_rawdesc[f] = take((size_t)m * sizeof(unsigned long long));
    }
    const size_t o_acc = take(ci ? (size_t)2 * 16 * m * sizeof(double) : 0);
    for (int k = 0; k < 4; ++k) {
        o_cstart[k] = take(mi + sizeof(int));
        o_cidx[k] = take(ci * sizeof(int));
        o_cval[k] = take(ci * sizeof(double));
    }
    // (per workgroup of the kernels that use them: two transposing, four packing)
    const size_t o_icursor = take(2 * 16 * mi + sizeof(int));
    const size_t o_itidx = take(2 * ci * sizeof(int)), o_itcol = take(2 * ci * sizeof(int)), o_itval = take(2 * ci * sizeof(double));
    const size_t o_rrank = take(4 * mi), o_rxoff = take(4 * mi), o_rfirst = take(4 * mi);
    if (dev_) (void)hipFree(dev_);
    dev_ = nullptr;
    RELP_HIP(hipMalloc(reinterpret_cast<void**>(&dev_), offset));
    bytes_ = offset;
    auto I = [&](size_t o) { return reinterpret_cast<int*>(dev_ + o); };
    auto D = [&](size_t o) { return reinterpret_cast<double*>(dev_ + o); };
    LuFactorWork w;
    w.m = m;
    w.cap_w = (int)cap_w;
    for (int b = 0; b < 2; ++b) {
        w.a_cr[b] = reinterpret_cast<unsigned*>(dev_ + o_acr[b]);
        w.a_val[b] = D(o_aval[b]);
        w.active[b] = I(o_active[b]);
    }
    w.r_start = I(o_rstart); w.r_len = I(o_rlen); w.r_newstart = I(o_rnew); w.growth = I(o_growth); w.targets = I(o_targets);
    w.ccount = I(o_ccount);
    w.rmax = reinterpret_cast<unsigned long long*>(dev_ + o_rmax);
    w.rmaxd = D(o_rmaxd);
    w.rowbest = reinterpret_cast<unsigned*>(dev_ + o_rowbest);
    w.best_e = I(o_beste);
    w.colmark = reinterpret_cast<unsigned*>(dev_ + o_colmark);
    w.kill = I(o_kill); w.tflag = I(o_tflag); w.pivk_row = I(o_pkr); w.pivk_col = I(o_pkc);
    w.ut_start = I(o_utstart); w.ut_col = I(o_utcol); w.ut_row = I(o_utrow); w.ut_val = D(o_utval);
    w.cap_u = (int)cap_u_;
    w.lt_row = I(o_ltrow); w.lt_step = I(o_ltstep); w.lt_val = D(o_ltval);
    w.cap_l = (int)cap_l_;
    w.tmp_start = I(o_tstart); w.tmp_cursor = I(o_tcursor); w.tmp_idx = I(o_tidx); w.tmp_val = D(o_tval); w.tmp_row = I(o_trow);
    w.rpos = I(o_rpos); w.cpos = I(o_cpos); w.row_at = I(o_rowat); w.col_at = I(o_colat);
    w.info = I(o_info);
    w_ = w;
    LuInverseWork iw;
    iw.m = m;
    iw.cap = (int)std::min<size_t>(ci, (size_t)1 << 30);
    iw.raw_col = I(o_rawcol);
    iw.raw_val = D(o_rawval);
    iw.raw_cap = (int)std::min<size_t>(2 * ci, (size_t)1 << 30);
    for (int f = 0; f < 2; ++f) {
        iw.raw_start[f] = I(o_rawstart[f]);
        iw.raw_len[f] = I(o_rawlen[f]);
        iw.raw_desc[f] = reinterpret_cast<unsigned long long*>(dev_ + o_rawdesc[f]);
    }
    iw.acc = D(o_acc);
    for (int k = 0; k < 4; ++k) {
        iw.csr_start[k] = I(o_cstart[k]);
        iw.csr_idx[k] = I(o_cidx[k]);
        iw.csr_val[k] = D(o_cval[k]);
    }
    iw.cursor = I(o_icursor);
    iw.tmp_idx = I(o_itidx); iw.tmp_col = I(o_itcol); iw.tmp_val = D(o_itval);
    iw.row_rank = I(o_rrank); iw.row_xoff = I(o_rxoff); iw.row_first = I(o_rfirst);
    iw.info = w.info;
    iw_ = iw;
}

// =====================================================================================================
// device
// =====================================================================================================
namespace {

constexpr unsigned NONE32 = 0xffffffffu;
constexpr unsigned HOLE = 0xffffffffu;   // arena entry without content; a live one is row << 16 | column
constexpr unsigned long long NONE64 = ~0ull;
constexpr int LUF_WAVES = LUF_THREADS / WAVE;

struct FactorShared {
    unsigned scan[LUF_WAVES + 2];
    unsigned long long best64;
    int smin;
    int n_active, kbase, ubase, top, n_targets, cur, error, n_acc, rounds, l_top, top_new, n_active_new, u_round, ref_row, ref_col, peak;
    int dense_rows, lds_rounds, spilled, next_target;
    unsigned long long dbg[6];  // diagnostic cycle sums of the eliminating waves (lane 0): preamble | pivot set-up | entry loop | write-out | pairs | targets
};

// The active sub-matrix of a round: entries (row << 16 | column, value), rows contiguous (r_start / r_len), in global memory or --
// once it is small enough -- in LDS (the same code: the pointers' address space is known at each call site after inlining).
struct Arena {
    unsigned* cr;
    double* val;
};

// Inclusive scan over the wave by DPP row shifts and row broadcasts: ten VALU instructions, no LDS crossbar (a __shfl_up scan is six
// dependent ds_bpermute round trips).
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1 (lanes without a source read 0)
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8  -> inclusive inside each row of 16
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, true);  // rows 1, 3 += last lane of rows 0, 2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, true);  // rows 2, 3 += lane 31
    return v;
}
// exclusive prefix of `v` over the workgroup's threads in thread order; *total = the sum.  Three barriers.
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, FactorShared& sh, unsigned* total) {
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const unsigned incl = wave_inclusive_scan(v);
    __syncthreads();  // earlier readers of sh.scan are done
    if (lane == WAVE - 1) sh.scan[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const unsigned w = lane < LUF_WAVES ? sh.scan[lane] : 0u;
        const unsigned wi = wave_inclusive_scan(w);
        if (lane < LUF_WAVES) sh.scan[lane] = wi - w;
        if (lane == LUF_WAVES - 1) sh.scan[LUF_WAVES] = wi;
    }
    __syncthreads();
    *total = sh.scan[LUF_WAVES];
    return sh.scan[wave] + incl - v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, DPP_QUAD_1032, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, DPP_QUAD_2301, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, DPP_ROW_ROR4, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, DPP_ROW_ROR8, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, DPP_ROW_BCAST15, 0xA, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, DPP_ROW_BCAST31, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(v, WAVE - 1);
}
__device__ __forceinline__ double wave_max_f64(double v) { return lane63_f64(wave_max(v)); }
// broadcast of lane `y` (wave-uniform): v_readlane, an SGPR move -- __shfl is a ds_bpermute, an LDS-crossbar round trip
__device__ __forceinline__ int lane_value(int v, int y) { return __builtin_amdgcn_readlane(v, y); }
__device__ __forceinline__ double lane_value(double v, int y) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), y), __builtin_amdgcn_readlane(__double2loint(v), y));
}
__device__ __forceinline__ int lanes_below(unsigned long long mask) {  // set bits of `mask` below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ unsigned candidate_key(const LuFactorWork& w, int r, int c, double v, double threshold) {
    const int cnt = w.ccount[c];
    const double rmax = w.rmaxd[r];
    const double mag = fabs(v);
    if (!(cnt == 1 || mag >= threshold * rmax)) return NONE32;  // (a column singleton needs no elimination: any non-zero is stable)
    long long score = (long long)(w.r_len[r] - 1) * (long long)(cnt - 1);
    if (score > 4095) score = 4095;
    int q = rmax > 0.0 ? (int)(mag / rmax * 15.0) : 15;
    q = 15 - min(15, max(0, q));  // larger magnitude: smaller rank
    return ((unsigned)score << 20) | ((unsigned)q << 16) | (unsigned)c;
}
__device__ __forceinline__ unsigned priority_of(unsigned key, int row) { return ((key >> 20) << 16) | (unsigned)row; }

// Row `r` of the active sub-matrix minus its multiples of the pivot rows of this round, by one wave: the row in registers
// (LUF_ROW_SLOTS entries per lane), a pivot row's entries broadcast one at a time.  Writes the new row compactly to the other arena.
// SLOTS: register entries per lane -- 1 for the rows that cannot outgrow 64 entries in this round (nearly all of them: every
// instruction of the inner loop is issued per slot, and sixteen waves on four SIMDs make the loop issue-bound), LUF_ROW_SLOTS otherwise.
template <int SLOTS>
__device__ __forceinline__ void eliminate_row(const LuFactorWork& w, FactorShared& sh, const LuFactorOut& out, const int r, const Arena old, const Arena nw,
                                              const bool exact_mode) {
    const int lane = threadIdx.x & (WAVE - 1);
#ifdef RELP_STAMPS  // (diagnostic build: cycle sums of the eliminating waves, tests/test_gpu_lu_factor_device.py prints them)
    long long t_mark = clock64();
    auto mark = [&](int k) {
        const long long now = clock64();
        if (lane == 0) atomicAdd(&sh.dbg[k], (unsigned long long)(now - t_mark));
        t_mark = now;
    };
    auto count = [&](int k) { if (lane == 0) atomicAdd(&sh.dbg[k], 1ull); };
#else
    auto mark = [](int) {};
    auto count = [](int) {};
#endif
    const int s0 = w.r_start[r];
    int len = w.r_len[r];
    const int d0 = w.r_newstart[r];
    const int capacity = max(0, len + w.growth[r]);
    int col[SLOTS], pk[SLOTS];
    double val[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int x = s * WAVE + lane;
        col[s] = x < len ? (int)(old.cr[s0 + x] & 0xffffu) : -1;
        val[s] = x < len ? old.val[s0 + x] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) pk[s] = col[s] >= 0 ? w.pivk_col[col[s]] : -1;
    mark(0);
    count(5);
    for (;;) {
        int mine = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (pk[s] >= 0) mine = min(mine, pk[s]);
        const int k = wave_min_i32(mine);
        if (k == 0x7fffffff) break;
        count(4);
        double a = 0.0;
        int pivot_column = -1;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (pk[s] == k) {
                a = val[s];
                pivot_column = col[s];
                col[s] = -1;
                pk[s] = -1;
            }
        const unsigned long long who = __ballot(pivot_column >= 0);
        const int src = __ffsll((long long)who) - 1;
        a = lane_value(a, src);
        pivot_column = lane_value(pivot_column, src);
        // the pivot row is read where it still lies in the OLD arena (its entries but the pivot are row k of U): out of LDS once the
        // active sub-matrix lives there, so that an elimination makes no global round trip at all
        const int pivot_row = (int)w.colmark[pivot_column];  // (written when the pivot was accepted; the markers are reset after the round)
        const int us = w.r_start[pivot_row], un = w.r_len[pivot_row], pe = w.best_e[pivot_row] - us;
        const double diagonal = old.val[us + pe];
        int pc = lane < un && lane != pe ? (int)(old.cr[us + lane] & 0xffffu) : -1;
        double pv = lane < un ? old.val[us + lane] : 0.0;
        const double ratio = a / diagonal;
        if (lane == 0) {
            const int at = atomicAdd(&sh.l_top, 1);
            if (at < w.cap_l) {
                w.lt_row[at] = r;
                w.lt_step[at] = k;
                w.lt_val[at] = ratio;
            } else {
                sh.error = LUF_ERR_L_CAPACITY;
            }
        }
        mark(1);
        for (int y0 = 0; y0 < un; y0 += WAVE) {
            if (y0 > 0) {
                pc = y0 + lane < un && y0 + lane != pe ? (int)(old.cr[us + y0 + lane] & 0xffffu) : -1;
                pv = y0 + lane < un ? old.val[us + y0 + lane] : 0.0;
            }
            const int cnt = min(WAVE, un - y0);
            for (int y = 0; y < cnt; ++y) {
                const int cc = lane_value(pc, y);
                if (cc < 0) continue;  // the pivot entry itself
                const double product = ratio * lane_value(pv, y);
                bool found = false;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (col[s] == cc) {
                        const double before = val[s];
                        double updated = before - product;
                        // (what floating point adds to the reference's exact cancellation, as lu_host.hpp: decomposition/mod.rs:176-186)
                        if (!exact_mode && updated != 0.0 && fabs(updated) <= 1e-15 * (fabs(before) + fabs(product))) updated = 0.0;
                        if (updated == 0.0) {
                            col[s] = -1;
                            atomicSub(&w.ccount[cc], 1);
                        } else {
                            val[s] = updated;
                        }
                        found = true;
                    }
                if (__ballot(found) == 0ull) {  // fill-in: the next free slot
                    if (len >= WAVE * SLOTS) {
                        if (lane == 0) sh.error = LUF_ERR_LONG_ROW;
                    } else {
                        const int slot = len >> 6;
                        if (lane == (len & (WAVE - 1))) {
#pragma unroll
                            for (int s = 0; s < SLOTS; ++s)
                                if (s == slot) {
                                    col[s] = cc;
                                    val[s] = -product;
                                    pk[s] = -1;
                                }
                            atomicAdd(&w.ccount[cc], 1);
                        }
                        ++len;
                    }
                }
            }
        }
    }
    mark(2);
    // the new row, holes squeezed out, into the other arena; its largest magnitude for the next rounds' threshold test
    int written = 0;
    double biggest = 0.0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const bool live = col[s] >= 0;
        const unsigned long long mask = __ballot(live);
        if (live) {
            const int at = written + lanes_below(mask);
            if (at < capacity) {
                nw.cr[d0 + at] = ((unsigned)r << 16) | (unsigned)col[s];
                nw.val[d0 + at] = val[s];
            }
            biggest = fmax(biggest, fabs(val[s]));
        }
        written += __popcll(mask);
    }
    for (int x = written + lane; x < capacity; x += WAVE) nw.cr[d0 + x] = HOLE;
    biggest = wave_max_f64(biggest);
    if (lane == 0) {
        if (written == 0) sh.error = LUF_ERR_SINGULAR;
        if (written > capacity) sh.error = LUF_ERR_ARENA;  // (cannot happen: the capacity is the bound old - pivots + sum of the pivot rows)
        w.r_len[r] = written;
        w.rmaxd[r] = biggest;
    }
    mark(3);
}

// cycle sums per phase (thread 0, shader clock) into info[LUF_STAMPS + k]: diagnostic, a few s_memtime per round
#define LUF_STAMP(k)                                                   \
    do {                                                               \
        if (tid == 0) {                                                \
            const long long now__ = clock64();                         \
            stamp_sum[(k)] += now__ - stamp_prev;                      \
            stamp_prev = now__;                                        \
        }                                                              \
    } while (0)

// ONE round: candidates -> competition -> conflicts -> accepted pivots -> U rows and targets -> layout -> copy + elimination -> reset.
// `old` holds the active sub-matrix, `nw` receives the next one; `cap` is the capacity of `nw`.  Every thread of the workgroup calls
// it; returns with the shared round state (sh.top, n_active, kbase, ubase, cur) advanced, or sh.error set.
// `spill` (capacity `spill_cap`): where the next sub-matrix goes when `nw` is too small for it (an LDS arena that the round's fill-in
// bound outgrows: the sub-matrix moves back to global memory; sh.spilled tells the caller).
__device__ __forceinline__ void factor_round(LuFactorWork& w, FactorShared& sh, const LuFactorOut& out, const Arena old, const Arena nw, const int cap,
                                             const Arena spill, const int spill_cap, const double threshold, const bool ref, const int m,
                                             long long* stamp_sum, long long& stamp_prev) {
    const int tid = threadIdx.x, T = LUF_THREADS;
    const int lane = tid & (WAVE - 1);
    const int n_active = sh.n_active, cur = sh.cur, top = sh.top, kbase = sh.kbase, ubase = sh.ubase;
    const int* __restrict__ act = (cur ? w.active[1] : w.active[0]);
    int* __restrict__ act_new = (cur ? w.active[0] : w.active[1]);
    if (tid == 0) {
        sh.smin = 0x7fffffff;
        sh.best64 = NONE64;
    }
    __syncthreads();
    // (1) every active row's best admissible entry
    if (!ref) {
        for (int e = tid; e < top; e += T) {
            const unsigned cr = old.cr[e];
            if (cr == HOLE) continue;
            const int c = (int)(cr & 0xffffu), r = (int)(cr >> 16);
            const unsigned key = candidate_key(w, r, c, old.val[e], threshold);
            if (key != NONE32) atomicMin(&w.rowbest[r], key);
        }
    } else {  // the reference's rule: ONE pivot, minimum score, ties by current column position, then row position
        unsigned long long mine = NONE64;
        for (int e = tid; e < top; e += T) {
            const unsigned cr = old.cr[e];
            if (cr == HOLE) continue;
            const int c = (int)(cr & 0xffffu), r = (int)(cr >> 16);
            unsigned long long score = (unsigned long long)(w.r_len[r] - 1) * (unsigned long long)(w.ccount[c] - 1);
            if (score > 0x7fffffffull) score = 0x7fffffffull;
            const unsigned long long key = (score << 32) | ((unsigned long long)w.cpos[c] << 16) | (unsigned long long)w.rpos[r];
            mine = key < mine ? key : mine;
        }
        mine = lane63_u64(wave_min_u64(mine));
        if (lane == 0 && mine != NONE64) atomicMin(&sh.best64, mine);
    }
    __syncthreads();
    LUF_STAMP(1);
    int limit = 0;
    if (!ref) {
        // (2) the round's minimum score; candidates within a slack of it compete (16 x, at least + 16: lu_factor.hpp)
        int mine = 0x7fffffff;
        for (int t = tid; t < n_active; t += T) {
            const unsigned key = w.rowbest[act[t]];
            if (key != NONE32) mine = min(mine, (int)(key >> 20));
        }
        mine = wave_min_i32(mine);
        if (lane == 0 && mine != 0x7fffffff) atomicMin(&sh.smin, mine);
        __syncthreads();
        const int smin = sh.smin;
        if (smin == 0x7fffffff) {
            if (tid == 0) sh.error = LUF_ERR_SINGULAR;
            __syncthreads();
            return;
        }
        limit = max(w.score_slack * smin, smin + w.score_slack);
        // (3) the entry of each competing candidate; the best candidate per column
        for (int e = tid; e < top; e += T) {
            const unsigned cr = old.cr[e];
            if (cr == HOLE) continue;
            const int c = (int)(cr & 0xffffu), r = (int)(cr >> 16);
            const unsigned best = w.rowbest[r];
            if ((int)(best >> 20) > limit || (int)(best & 0xffffu) != c) continue;
            if (candidate_key(w, r, c, old.val[e], threshold) != best) continue;
            w.best_e[r] = e;
            atomicMin(&w.colmark[c], priority_of(best, r));
        }
        __syncthreads();
        LUF_STAMP(2);
        // (4) conflicts: an entry (r, c) with c the pivot column of another row's candidate and r a candidate row itself -- the
        //     two pivots are not compatible, the worse one waits for a later round
        for (int e = tid; e < top; e += T) {
            const unsigned cr = old.cr[e];
            if (cr == HOLE) continue;
            const int c = (int)(cr & 0xffffu), r = (int)(cr >> 16);
            const unsigned pc = w.colmark[c];
            if (pc == NONE32 || (int)(pc & 0xffffu) == r) continue;
            const unsigned kr = w.rowbest[r];
            if (kr == NONE32 || (int)(kr >> 20) > limit) continue;
            const unsigned pr = priority_of(kr, r);
            if (w.colmark[kr & 0xffffu] != pr) continue;  // row r lost its own column: no candidate
            const unsigned loser = pc > pr ? pc : pr;
            w.kill[loser & 0xffffu] = 1;
        }
        __syncthreads();
    } else {
        const unsigned long long best = sh.best64;
        if (best == NONE64) {
            if (tid == 0) sh.error = LUF_ERR_SINGULAR;
            __syncthreads();
            return;
        }
        const int rw = w.row_at[best & 0xffffull], cw = w.col_at[(best >> 16) & 0xffffull];
        if (tid == 0) {
            sh.ref_row = rw;
            sh.ref_col = cw;
        }
        const unsigned wanted = ((unsigned)rw << 16) | (unsigned)cw;
        for (int e = tid; e < top; e += T)
            if (old.cr[e] == wanted) w.best_e[rw] = e;
        __syncthreads();
    }
    LUF_STAMP(3);
    // (5) the accepted pivots take consecutive positions in row order; their rows become rows of U
    {
        unsigned carry_n = 0, carry_u = 0;
        for (int base = 0; base < n_active; base += T) {
            const int t = base + tid;
            int i = -1;
            bool accepted = false;
            if (t < n_active) {
                i = act[t];
                if (ref) {
                    accepted = i == sh.ref_row;
                } else {
                    const unsigned key = w.rowbest[i];
                    accepted = key != NONE32 && (int)(key >> 20) <= limit && w.colmark[key & 0xffffu] == priority_of(key, i) && !w.kill[i];
                }
            }
            // (two fields in one word: pivots of a chunk <= 1024 in the top bits, their U entries below -- a row holds <= 256)
            const unsigned v = accepted ? ((1u << 20) | (unsigned)(w.r_len[i] - 1)) : 0u;
            unsigned total;
            const unsigned ex = block_exclusive_scan(v, sh, &total);
            if (accepted) {
                const int k = kbase + (int)carry_n + (int)(ex >> 20);
                const int e = w.best_e[i];
                const int c = (int)(old.cr[e] & 0xffffu);
                w.pivk_row[i] = k;
                w.pivk_col[c] = k;
                w.colmark[c] = (unsigned)i;  // from here to the end of the round: the pivot row of a pivot column (eliminate_row)
                out.rowpos[i] = k;
                out.colpos[c] = k;
                out.diag[k] = old.val[e];
                w.ut_start[k] = ubase + (int)carry_u + (int)(ex & 0xfffffu);
            }
            carry_n += total >> 20;
            carry_u += total & 0xfffffu;
        }
        if (tid == 0) {
            sh.n_acc = (int)carry_n;
            sh.u_round = (int)carry_u;
            w.ut_start[kbase + sh.n_acc] = ubase + sh.u_round;
            if (sh.n_acc == 0) sh.error = LUF_ERR_SINGULAR;
            if (ubase + sh.u_round > w.cap_u) sh.error = LUF_ERR_U_CAPACITY;
            if (ref && sh.n_acc == 1) {  // swap the pivot to (k, k): positions only (decomposition/mod.rs:224-273)
                const int k = kbase, pi = sh.ref_row, pj = sh.ref_col;
                const int other_row = w.row_at[k], pr = w.rpos[pi];
                w.row_at[pr] = other_row;
                w.rpos[other_row] = pr;
                w.row_at[k] = pi;
                w.rpos[pi] = k;
                const int other_col = w.col_at[k], pc = w.cpos[pj];
                w.col_at[pc] = other_col;
                w.cpos[other_col] = pc;
                w.col_at[k] = pj;
                w.cpos[pj] = k;
            }
        }
    }
    __syncthreads();
    if (sh.error != LUF_OK) return;
    LUF_STAMP(4);
    // (6) pivot rows -> U (decomposition/mod.rs:60-70); the rows with an entry in a pivot column are this round's targets
    for (int e = tid; e < top; e += T) {
        const unsigned cr = old.cr[e];
        if (cr == HOLE) continue;
        const int c = (int)(cr & 0xffffu), r = (int)(cr >> 16);
        const int k = w.pivk_row[r];
        if (k >= 0) {
            const int be = w.best_e[r];
            if (e == be) continue;
            const int at = w.ut_start[k] + (e - w.r_start[r]) - (e > be ? 1 : 0);
            w.ut_col[at] = c;
            w.ut_val[at] = old.val[e];
            w.ut_row[at] = k;
            atomicSub(&w.ccount[c], 1);
        } else {
            const int kc = w.pivk_col[c];
            if (kc >= kbase) {  // (a column pivoted in an earlier round has no active entry left)
                w.tflag[r] = 1;
                atomicAdd(&w.growth[r], w.ut_start[kc + 1] - w.ut_start[kc] - 1);
            }
        }
    }
    __syncthreads();
    LUF_STAMP(5);
    // (7) layout of the next arena: the remaining rows in order, a target row with room for its fill-in
    {
        unsigned carry_cap = 0, carry_cnt = 0;
        for (int base = 0; base < n_active; base += T) {
            const int t = base + tid;
            int i = -1;
            bool stays = false, target = false;
            int capacity = 0;
            if (t < n_active) {
                i = act[t];
                if (w.pivk_row[i] < 0) {
                    stays = true;
                    target = w.tflag[i] != 0;
                    capacity = max(0, w.r_len[i] + w.growth[i]);
                }
            }
            unsigned total_cap, total_cnt;
            const unsigned ex_cap = block_exclusive_scan((unsigned)capacity, sh, &total_cap);
            const unsigned ex_cnt = block_exclusive_scan(stays ? ((1u << 16) | (target ? 1u : 0u)) : 0u, sh, &total_cnt);
            if (stays) {
                w.r_newstart[i] = (int)(carry_cap + ex_cap);
                act_new[(carry_cnt >> 16) + (ex_cnt >> 16)] = i;
                if (target) w.targets[(carry_cnt & 0xffffu) + (ex_cnt & 0xffffu)] = i;
            }
            carry_cap += total_cap;
            carry_cnt += total_cnt;  // (rows <= 65535: neither 16-bit field overflows)
        }
        if (tid == 0) {
            sh.top_new = (int)carry_cap;
            sh.n_active_new = (int)(carry_cnt >> 16);
            sh.n_targets = (int)(carry_cnt & 0xffffu);
            sh.next_target = 0;
            sh.spilled = sh.top_new > cap ? 1 : 0;
            if (sh.top_new > (sh.spilled ? spill_cap : cap)) sh.error = LUF_ERR_ARENA;
            sh.peak = max(sh.peak, sh.top_new);
        }
    }
    __syncthreads();
    if (sh.error != LUF_OK) return;
    LUF_STAMP(6);
    // (8) the untouched rows are copied, the targets eliminated (decomposition/mod.rs:71-100,146-210) -- into the other arena
    {
        auto rewrite = [&](const Arena to) {
            for (int e = tid; e < top; e += T) {
                const unsigned cr = old.cr[e];
                if (cr == HOLE) continue;
                const int r = (int)(cr >> 16);
                if (w.pivk_row[r] >= 0 || w.tflag[r]) continue;
                const int at = w.r_newstart[r] + (e - w.r_start[r]);
                to.cr[at] = cr;
                to.val[at] = old.val[e];
            }
            // (nothing a target produces depends on which wave eliminated it)
            const int n_targets = sh.n_targets;
            const bool fixed_map = w.fixed_target_map != 0;
            for (int turn = 0;; ++turn) {
                int t = 0;
                if (fixed_map) {
                    t = (int)(threadIdx.x / WAVE) + turn * LUF_WAVES;
                } else {
                    // (every lane takes part, lane 0 adds the one: `if (lane == 0) t = atomicAdd(..); t = readlane(t, 0)` at the head of a
                    //  for (;;) loop compiled to a loop that never ends with this toolchain -- twice, here and in the inversion)
                    t = atomicAdd(&sh.next_target, lane == 0 ? 1 : 0);
                    t = __builtin_amdgcn_readfirstlane(t);
                }
                if (t >= n_targets) break;
                const int r = w.targets[t];
                const int len = w.r_len[r];
                // (slots appended <= the pivot rows' entries = growth + pivots of the row <= growth + len)
                if (2 * len + max(0, w.growth[r]) <= WAVE) eliminate_row<1>(w, sh, out, r, old, to, ref);
                else eliminate_row<LUF_ROW_SLOTS>(w, sh, out, r, old, to, ref);
            }
        };
        if (sh.spilled) rewrite(spill);
        else rewrite(nw);
    }
    __syncthreads();
    LUF_STAMP(7);
    // (9) the next round's state
    {
        const int n_new = sh.n_active_new;
        for (int t = tid; t < n_new; t += T) {
            const int i = act_new[t];
            w.r_start[i] = w.r_newstart[i];
            w.rowbest[i] = NONE32;
            w.kill[i] = 0;
            w.tflag[i] = 0;
            w.growth[i] = 0;
        }
        for (int c = tid; c < m; c += T) w.colmark[c] = NONE32;
    }
    __syncthreads();
    if (tid == 0) {
        sh.top = sh.top_new;
        sh.n_active = sh.n_active_new;
        sh.kbase = kbase + sh.n_acc;
        sh.ubase = ubase + sh.u_round;
        sh.cur = cur ^ 1;
        sh.rounds += 1;
    }
    __syncthreads();
    LUF_STAMP(8);
}

// `lds_level`: which of the per-row work arrays live in LDS instead of global memory (2: all of them, m <= ~1900; 1: the ones the
// entry passes gather from, m <= ~3500; 0: none); at level 2 whatever LDS is left holds the two arenas of the active sub-matrix once
// it fits (`arena_lds` entries each).  A round is a chain of dependent gathers and atomics on these arrays: ~150 cycles each out of
// LDS against ~800 through L2.  The pointers are derived from the dynamic LDS block unconditionally per instantiation, so that the
// compiler emits ds_ instructions (a pointer that MAY be global or LDS compiles to flat_ accesses, which are slower than either).
template <int lds_level>
__global__ void __launch_bounds__(LUF_THREADS) lu_factor_kernel(LuFactorSource src, LuFactorWork w_in, LuFactorOut out, double threshold,
                                                                int reference_ties, int dense_tail, int arena_lds) {
    extern __shared__ unsigned char luf_dynamic_lds[];
    __shared__ FactorShared sh;
    LuFactorWork w = w_in;
    Arena lds_arena[2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    if constexpr (lds_level >= 1) {
        unsigned char* at = luf_dynamic_lds;
        auto carve = [&](size_t bytes) {
            unsigned char* p = at;
            at += (bytes + 7) & ~size_t(7);
            return p;
        };
        const size_t mi = (size_t)w.m * sizeof(int);
        w.rmaxd = (double*)carve((size_t)w.m * sizeof(double));
        w.ccount = (int*)carve(mi);
        w.rowbest = (unsigned*)carve(mi);
        w.colmark = (unsigned*)carve(mi);
        w.kill = (int*)carve(mi);
        w.tflag = (int*)carve(mi);
        w.growth = (int*)carve(mi);
        w.r_len = (int*)carve(mi);
        w.pivk_col = (int*)carve(mi);
        if constexpr (lds_level >= 2) {
            w.r_start = (int*)carve(mi);
            w.r_newstart = (int*)carve(mi);
            w.targets = (int*)carve(mi);
            w.best_e = (int*)carve(mi);
            w.pivk_row = (int*)carve(mi);
            w.active[0] = (int*)carve(mi);
            w.active[1] = (int*)carve(mi);
            w.ut_start = (int*)carve(mi + sizeof(int));
            lds_arena[0].val = (double*)carve((size_t)arena_lds * sizeof(double));
            lds_arena[1].val = (double*)carve((size_t)arena_lds * sizeof(double));
            lds_arena[0].cr = (unsigned*)carve((size_t)arena_lds * sizeof(unsigned));
            lds_arena[1].cr = (unsigned*)carve((size_t)arena_lds * sizeof(unsigned));
        }
    }
    __shared__ double dense[LUF_DENSE_MAX][LUF_DENSE_MAX + 1];
    __shared__ int dense_cols[LUF_DENSE_MAX];
    const int tid = threadIdx.x, T = LUF_THREADS;
    const int lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int m = w.m;
    const bool ref = reference_ties != 0;
    if (ref) threshold = 0.0;
    long long stamp_prev = clock64();
    long long stamp_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (tid == 0) {
        sh.n_active = m;
        sh.kbase = 0;
        sh.ubase = 0;
        sh.top = 0;
        sh.cur = 0;
        sh.error = LUF_OK;
        sh.rounds = 0;
        sh.l_top = 0;
        sh.peak = 0;
        sh.dense_rows = 0;
        sh.lds_rounds = 0;
        sh.spilled = 0;
        for (int k = 0; k < 6; ++k) sh.dbg[k] = 0ull;
    }
    // ---- load: the basis columns by rows into arena 0 ------------------------------------------------------------------------
    for (int i = tid; i < m; i += T) {
        w.r_len[i] = 0;
        w.rmax[i] = 0ull;
        w.pivk_row[i] = -1;
        w.pivk_col[i] = -1;
        out.rowpos[i] = -1;
        out.colpos[i] = -1;
        w.rpos[i] = w.cpos[i] = w.row_at[i] = w.col_at[i] = i;
        w.active[0][i] = i;
        w.rowbest[i] = NONE32;
        w.colmark[i] = NONE32;
        w.kill[i] = 0;
        w.tflag[i] = 0;
        w.growth[i] = 0;
        w.tmp_cursor[i] = 0;
    }
    __syncthreads();
    for (int j = tid; j < m; j += T) {
        const int cj = src.basis ? src.basis[j] : j;
        int count = 0;
        for (int e = src.col_start[cj]; e < src.col_start[cj + 1]; ++e)
            if (src.value[e] != 0.0) {
                atomicAdd(&w.r_len[src.row_index[e]], 1);
                ++count;
            }
        w.ccount[j] = count;
        if (count == 0) sh.error = LUF_ERR_SINGULAR;
    }
    __syncthreads();
    {
        unsigned carry = 0;
        for (int base = 0; base < m; base += T) {
            const int i = base + tid;
            const unsigned v = i < m ? (unsigned)w.r_len[i] : 0u;
            unsigned total;
            const unsigned ex = block_exclusive_scan(v, sh, &total) + carry;
            if (i < m) {
                w.r_start[i] = (int)ex;
                if (v == 0) sh.error = LUF_ERR_SINGULAR;
                if (v > (unsigned)LUF_MAX_ROW) sh.error = LUF_ERR_LONG_ROW;
            }
            carry += total;
        }
        if (tid == 0) {
            sh.top = (int)carry;
            sh.peak = (int)carry;
            w.info[LUF_NNZ_B] = (int)carry;
            if (carry > (unsigned)w.cap_w) sh.error = LUF_ERR_ARENA;
        }
    }
    __syncthreads();
    const Arena g0 = {w.a_cr[0], w.a_val[0]}, g1 = {w.a_cr[1], w.a_val[1]};
    if (sh.error == LUF_OK) {
        for (int j = tid; j < m; j += T) {
            const int cj = src.basis ? src.basis[j] : j;
            const double sign = (src.flipped && src.flipped[cj]) ? -1.0 : 1.0;
            for (int e = src.col_start[cj]; e < src.col_start[cj + 1]; ++e) {
                const double v = src.value[e];
                if (v == 0.0) continue;
                const int row = src.row_index[e];
                const int at = w.r_start[row] + atomicAdd(&w.tmp_cursor[row], 1);
                g0.cr[at] = ((unsigned)row << 16) | (unsigned)j;
                g0.val[at] = sign * v;
                atomicMax(&w.rmax[row], (unsigned long long)__double_as_longlong(fabs(v)));
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < m; i += T) w.rmaxd[i] = __longlong_as_double((long long)w.rmax[i]);
    __syncthreads();
    LUF_STAMP(0);

    // ---- rounds ----------------------------------------------------------------------------------------------------------------
    bool in_lds = false;  // where the active sub-matrix lives (uniform over the workgroup)
    while (sh.error == LUF_OK && sh.n_active > 0) {
        if (!ref && sh.n_active <= dense_tail) break;  // the rest goes through the dense tail
        const int cur = sh.cur;
        if constexpr (lds_level >= 2) {
            if (!in_lds && arena_lds >= 256 && 4 * sh.top <= 3 * arena_lds) {  // it fits with room for a round's fill-in: move it into LDS
                const Arena from = cur ? g1 : g0, to = cur ? lds_arena[1] : lds_arena[0];
                const int top = sh.top;
                for (int e = tid; e < top; e += T) {
                    to.cr[e] = from.cr[e];
                    to.val[e] = from.val[e];
                }
                in_lds = true;
                __syncthreads();
            }
            if (in_lds) {
                if (tid == 0) sh.lds_rounds += 1;
                factor_round(w, sh, out, cur ? lds_arena[1] : lds_arena[0], cur ? lds_arena[0] : lds_arena[1], arena_lds, cur ? g0 : g1, w.cap_w, threshold, ref, m,
                             stamp_sum, stamp_prev);
                if (sh.spilled) in_lds = false;  // (uniform: written before the round's last barriers) the next sub-matrix is in global memory again
                continue;
            }
        }
        factor_round(w, sh, out, cur ? g1 : g0, cur ? g0 : g1, w.cap_w, cur ? g0 : g1, w.cap_w, threshold, ref, m, stamp_sum, stamp_prev);
    }
    __syncthreads();
    LUF_STAMP(8);

    // ---- dense tail: the last rows by partial pivoting out of LDS, one wave ------------------------------------------------------
    if (sh.error == LUF_OK && sh.n_active > 0) {
        const int n = sh.n_active, cur = sh.cur, top = sh.top, kbase = sh.kbase;
        const int* __restrict__ act = (cur ? w.active[1] : w.active[0]);
        // local column numbers: the unpivoted columns in ascending order (ordered compaction over all columns)
        {
            unsigned carry = 0;
            for (int base = 0; base < m; base += T) {
                const int c = base + tid;
                const bool open = c < m && out.colpos[c] < 0;
                unsigned total;
                const unsigned ex = block_exclusive_scan(open ? 1u : 0u, sh, &total) + carry;
                if (open) {
                    w.growth[c] = (int)ex;  // (growth: free between rounds) local number of column c
                    if (ex < (unsigned)LUF_DENSE_MAX) dense_cols[ex] = c;
                }
                carry += total;
            }
            if (tid == 0 && (int)carry != n) sh.error = LUF_ERR_SINGULAR;
        }
        for (int t = tid; t < n; t += T) w.tflag[act[t]] = t;  // local number of a row
        for (int x = tid; x < LUF_DENSE_MAX * (LUF_DENSE_MAX + 1); x += T) (&dense[0][0])[x] = 0.0;
        __syncthreads();
        if (sh.error == LUF_OK) {
            auto gather = [&](const Arena from) {
                for (int e = tid; e < top; e += T) {
                    const unsigned cr = from.cr[e];
                    if (cr == HOLE) continue;
                    dense[w.tflag[cr >> 16]][w.growth[cr & 0xffffu]] = from.val[e];
                }
            };
            if constexpr (lds_level >= 2) {
                if (in_lds) gather(cur ? lds_arena[1] : lds_arena[0]);
                else gather(cur ? g1 : g0);
            } else {
                gather(cur ? g1 : g0);
            }
        }
        __syncthreads();
        if (sh.error == LUF_OK && wave == 0) {
            // lane = local row.  Step s eliminates local column s: the unpivoted row with the largest entry pivots.
            bool done = lane >= n;  // this lane's row has pivoted (or does not exist)
            int u_at = sh.ubase;
            for (int s = 0; s < n; ++s) {
                const double mine = done ? -1.0 : fabs(dense[lane][s]);
                const double best = wave_max_f64(mine);
                if (!(best > 0.0)) {
                    if (lane == 0) sh.error = LUF_ERR_SINGULAR;
                    break;
                }
                const unsigned long long who = __ballot(mine == best);
                const int p = __ffsll((long long)who) - 1;  // lowest local row among equals: deterministic
                const int k = kbase + s;
                const double pivot = dense[p][s];
                if (lane == p) {
                    done = true;
                    out.rowpos[act[p]] = k;
                    out.colpos[dense_cols[s]] = k;
                    out.diag[k] = pivot;
                    w.ut_start[k] = u_at;
                }
                // row k of U: the pivot row's entries in the columns still open (lane = local column here)
                const double uv = (lane > s && lane < n) ? dense[p][lane] : 0.0;
                const unsigned long long umask = __ballot(uv != 0.0);
                if (uv != 0.0) {
                    const int at = u_at + lanes_below(umask);
                    if (at < w.cap_u) {
                        w.ut_col[at] = dense_cols[lane];
                        w.ut_val[at] = uv;
                        w.ut_row[at] = k;
                    }
                }
                u_at += __popcll(umask);
                if (u_at > w.cap_u) {
                    if (lane == 0) sh.error = LUF_ERR_U_CAPACITY;
                    break;
                }
                // the rows below
                if (!done) {
                    const double a = dense[lane][s];
                    if (a != 0.0) {
                        const double ratio = a / pivot;
                        const int at = atomicAdd(&sh.l_top, 1);
                        if (at < w.cap_l) {
                            w.lt_row[at] = act[lane];
                            w.lt_step[at] = k;
                            w.lt_val[at] = ratio;
                        } else {
                            sh.error = LUF_ERR_L_CAPACITY;
                        }
                        for (int j = s + 1; j < n; ++j) {
                            const double pj = dense[p][j];
                            if (pj == 0.0) continue;
                            const double before = dense[lane][j], product = ratio * pj;
                            double updated = before - product;
                            if (updated != 0.0 && fabs(updated) <= 1e-15 * (fabs(before) + fabs(product))) updated = 0.0;
                            dense[lane][j] = updated;
                        }
                    }
                }
            }
            if (lane == 0) {
                w.ut_start[kbase + n] = u_at;
                sh.ubase = u_at;
                sh.kbase = kbase + n;
                sh.dense_rows = n;
                sh.n_active = 0;
            }
        }
        __syncthreads();
    }

    LUF_STAMP(9);
    // ---- finalisation: L by rows of the position space, U with positions as columns, every row sorted -----------------------------
    if (sh.error == LUF_OK && sh.l_top > w.cap_l) sh.error = LUF_ERR_L_CAPACITY;
    __syncthreads();
    if (sh.error == LUF_OK) {
        const int nl = sh.l_top, nu = sh.ubase;
        for (int p = tid; p <= m; p += T) w.tmp_cursor[p] = 0;
        __syncthreads();
        for (int t = tid; t < nl; t += T) atomicAdd(&w.tmp_cursor[out.rowpos[w.lt_row[t]]], 1);
        __syncthreads();
        {
            unsigned carry = 0;
            for (int base = 0; base < m; base += T) {
                const int p = base + tid;
                const unsigned v = p < m ? (unsigned)w.tmp_cursor[p] : 0u;
                unsigned total;
                const unsigned ex = block_exclusive_scan(v, sh, &total) + carry;
                if (p < m) out.l_start[p] = (int)ex;
                carry += total;
            }
            if (tid == 0) out.l_start[m] = nl;
        }
        __syncthreads();
        for (int p = tid; p < m; p += T) w.tmp_cursor[p] = 0;
        __syncthreads();
        for (int t = tid; t < nl; t += T) {
            const int p = out.rowpos[w.lt_row[t]];
            const int at = out.l_start[p] + atomicAdd(&w.tmp_cursor[p], 1);
            w.tmp_idx[at] = w.lt_step[t];
            w.tmp_val[at] = w.lt_val[t];
            w.tmp_row[at] = p;
        }
        __syncthreads();
        for (int e = tid; e < nl; e += T) {  // rank sort inside the row (the steps of a row are distinct)
            const int p = w.tmp_row[e], c = w.tmp_idx[e];
            int rank = 0;
            for (int x = out.l_start[p]; x < out.l_start[p + 1]; ++x) rank += w.tmp_idx[x] < c ? 1 : 0;
            out.l_col[out.l_start[p] + rank] = c;
            out.l_val[out.l_start[p] + rank] = w.tmp_val[e];
        }
        for (int k = tid; k <= m; k += T) out.u_start[k] = w.ut_start[k];
        for (int e = tid; e < nu; e += T) {
            const int k = w.ut_row[e];
            const int c = out.colpos[w.ut_col[e]];
            int rank = 0;
            for (int x = w.ut_start[k]; x < w.ut_start[k + 1]; ++x) rank += out.colpos[w.ut_col[x]] < c ? 1 : 0;
            out.u_col[w.ut_start[k] + rank] = c;
            out.u_val[w.ut_start[k] + rank] = w.ut_val[e];
        }
    }
    __syncthreads();
    LUF_STAMP(10);
    if (tid == 0) {
        for (int k = 0; k < 11; ++k) w.info[LUF_STAMPS + k] = (int)(stamp_sum[k] >> 4);  // units of 16 cycles
        w.info[LUF_STATUS] = sh.error;
        w.info[LUF_NNZ_L] = sh.l_top;
        w.info[LUF_NNZ_U] = sh.ubase;
        w.info[LUF_ROUNDS] = sh.rounds;
        w.info[LUF_DENSE_ROWS] = sh.dense_rows;
        w.info[LUF_ARENA_PEAK] = sh.peak;
        w.info[LUF_LDS_ROUNDS] = sh.lds_rounds;
        for (int k = 0; k < 4; ++k) w.info[23 + k] = (int)(sh.dbg[k] >> 10);  // kilo-cycles summed over the eliminating waves
        w.info[27] = (int)sh.dbg[4];
        w.info[28] = (int)sh.dbg[5];
    }
}

}  // namespace

void launch_lu_factor(const LuFactorSource& src, const LuFactorWork& w, const LuFactorOut& out, double threshold, int reference_ties,
                      int dense_tail, hipStream_t stream) {
    if (w.m > 65535) throw std::invalid_argument("device LU factorisation: more than 65535 rows");
    if (reference_ties) dense_tail = 0;
    dense_tail = std::max(0, std::min(dense_tail, LUF_DENSE_MAX));
    LuFactorOut o = out;
    o.cap_l = std::min(out.cap_l, w.cap_l);
    o.cap_u = std::min(out.cap_u, w.cap_u);
    LuFactorWork ww = w;
    // (targets go to the waves by a fixed map; RELP_LUF_CLAIM_TARGETS=1 lets the waves claim them from a counter instead -- balanced, but
    //  the claim, an LDS atomic every lane of the wave takes part in, costs more than the imbalance: 2.0 M against 0.96 M cycles on 25FV47)
    ww.fixed_target_map = thread_tuning().has(RELP_SW_LUF_CLAIM_TARGETS) ? 0 : 1;
    ww.score_slack = thread_tuning().luf_slack > 0 ? thread_tuning().luf_slack : 16;  // (A/B hook; see lu_factor.hpp)
    ww.cap_l = o.cap_l;
    ww.cap_u = o.cap_u;
    static PerDeviceOnce once;
    once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_factor_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_factor_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);
    });
    const size_t m8 = ((size_t)w.m * sizeof(int) + 7) & ~size_t(7);
    const size_t level1 = (size_t)w.m * sizeof(double) + 8 * m8, level2 = level1 + 7 * m8 + m8 + 8;
    const size_t budget = (size_t)140 * 1024;
    const int forced = thread_tuning().luf_lds > 0 ? thread_tuning().luf_lds - 1 : -1;  // diagnostic: 0 keeps every work array in global memory
    int lds_level = level2 <= budget ? 2 : level1 <= budget ? 1 : 0;
    if (forced >= 0) lds_level = std::min(lds_level, forced);
    // what LDS the per-row arrays leave holds the two arenas of the active sub-matrix (12 bytes per entry and arena) once it fits
    int arena_lds = 0;
    if (lds_level == 2 && !thread_tuning().has(RELP_SW_LUF_NO_LDS_ARENA)) arena_lds = (int)(((budget - level2) / 24) & ~size_t(63));
    if (thread_tuning().luf_lds_arena > 0) arena_lds = std::min(arena_lds, thread_tuning().luf_lds_arena & ~63);  // test hook: a small arena (spills)
    if (arena_lds < 256) arena_lds = 0;
    const size_t lds = (lds_level == 2 ? level2 : lds_level == 1 ? level1 : 0) + (size_t)arena_lds * 24 + 64;
    if (lds_level == 2) hipLaunchKernelGGL(lu_factor_kernel<2>, dim3(1), dim3(LUF_THREADS), lds, stream, src, ww, o, threshold, reference_ties, dense_tail, arena_lds);
    else if (lds_level == 1) hipLaunchKernelGGL(lu_factor_kernel<1>, dim3(1), dim3(LUF_THREADS), lds, stream, src, ww, o, threshold, reference_ties, dense_tail, 0);
    else hipLaunchKernelGGL(lu_factor_kernel<0>, dim3(1), dim3(LUF_THREADS), lds, stream, src, ww, o, threshold, reference_ties, dense_tail, 0);
}

}  // namespace relp
