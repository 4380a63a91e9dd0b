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
    if (const char* limit = getenv("RELP_LUF_ARENA_CAP")) cap_w = std::max<size_t>(64, (size_t)atoll(limit));  // test hook: an arena the basis outgrows
    size_t offset = 0;
    auto take = [&](size_t bytes) {
        offset = (offset + 63) & ~size_t(63);
        const size_t at = offset;
        offset += bytes;
        return at;
    };
    const size_t mi = (size_t)m * sizeof(int);
    size_t o_acol[2], o_arow[2], o_aval[2], o_active[2];
    for (int b = 0; b < 2; ++b) {
        o_acol[b] = take(cap_w * sizeof(int));
        o_arow[b] = take(cap_w * sizeof(int));
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
    const size_t o_icursor = take(mi + sizeof(int));
    const size_t o_itidx = take(ci * sizeof(int)), o_itcol = take(ci * sizeof(int)), o_itval = take(ci * sizeof(double));
    const size_t o_rrank = take(mi), o_rxoff = take(mi), o_rfirst = take(mi);
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
        w.a_col[b] = I(o_acol[b]);
        w.a_row[b] = I(o_arow[b]);
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
constexpr unsigned long long NONE64 = ~0ull;
constexpr int LUF_WAVES = LUF_THREADS / WAVE;

struct FactorShared {
    unsigned long long scan[LUF_WAVES + 2];
    unsigned long long best64;
    int smin;
    int n_active, kbase, ubase, top, n_targets, cur, error, n_acc, rounds, l_top, top_new, n_active_new, u_round, ref_row, ref_col, peak;
    int dense_rows;
};

__device__ __forceinline__ unsigned long long wave_inclusive_scan(unsigned long long v) {
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned lo = (unsigned)__shfl_up((int)(unsigned)v, d, WAVE);
        const unsigned hi = (unsigned)__shfl_up((int)(unsigned)(v >> 32), d, WAVE);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        if (lane >= d) v += other;
    }
    return v;
}
// exclusive prefix of `v` over the workgroup's threads in thread order; *total = the sum.  Three barriers.
__device__ __forceinline__ unsigned long long block_exclusive_scan(unsigned long long v, FactorShared& sh, unsigned long long* total) {
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const unsigned long long incl = wave_inclusive_scan(v);
    __syncthreads();  // earlier readers of sh.scan are done
    if (lane == WAVE - 1) sh.scan[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const unsigned long long w = lane < LUF_WAVES ? sh.scan[lane] : 0ull;
        const unsigned long long wi = wave_inclusive_scan(w);
        if (lane < LUF_WAVES) sh.scan[lane] = wi - w;
        if (lane == LUF_WAVES - 1) sh.scan[LUF_WAVES] = wi;
    }
    __syncthreads();
    *total = sh.scan[LUF_WAVES];
    return sh.scan[wave] + incl - v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) v = min(v, __shfl_xor(v, d, WAVE));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) v = fmax(v, __shfl_xor(v, d, WAVE));
    return v;
}
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
__device__ __forceinline__ void eliminate_row(const LuFactorWork& w, FactorShared& sh, const LuFactorOut& out, const int r, const int cur, const bool exact_mode) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int s0 = w.r_start[r];
    int len = w.r_len[r];
    const int* __restrict__ acol = (cur ? w.a_col[1] : w.a_col[0]);
    const double* __restrict__ aval = (cur ? w.a_val[1] : w.a_val[0]);
    int col[LUF_ROW_SLOTS], pk[LUF_ROW_SLOTS];
    double val[LUF_ROW_SLOTS];
#pragma unroll
    for (int s = 0; s < LUF_ROW_SLOTS; ++s) {
        const int x = s * WAVE + lane;
        col[s] = x < len ? acol[s0 + x] : -1;
        val[s] = x < len ? aval[s0 + x] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < LUF_ROW_SLOTS; ++s) pk[s] = col[s] >= 0 ? w.pivk_col[col[s]] : -1;
    for (;;) {
        int mine = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < LUF_ROW_SLOTS; ++s)
            if (pk[s] >= 0) mine = min(mine, pk[s]);
        const int k = wave_min_i32(mine);
        if (k == 0x7fffffff) break;
        double a = 0.0;
        bool holder = false;
#pragma unroll
        for (int s = 0; s < LUF_ROW_SLOTS; ++s)
            if (pk[s] == k) {
                a = val[s];
                col[s] = -1;
                pk[s] = -1;
                holder = true;
            }
        const unsigned long long who = __ballot(holder);
        const int src = __ffsll((long long)who) - 1;
        a = lane_value(a, src);
        const double ratio = a / out.diag[k];
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
        const int us = w.ut_start[k], un = w.ut_start[k + 1] - us;
        for (int y0 = 0; y0 < un; y0 += WAVE) {
            const int pc = y0 + lane < un ? w.ut_col[us + y0 + lane] : -1;
            const double pv = y0 + lane < un ? w.ut_val[us + y0 + lane] : 0.0;
            const int cnt = min(WAVE, un - y0);
            for (int y = 0; y < cnt; ++y) {
                const int cc = lane_value(pc, y);
                const double product = ratio * lane_value(pv, y);
                bool found = false;
#pragma unroll
                for (int s = 0; s < LUF_ROW_SLOTS; ++s)
                    if (col[s] == cc) {
                        const double old = val[s];
                        double updated = old - product;
                        // (what floating point adds to the reference's exact cancellation, as lu_host.hpp: decomposition/mod.rs:176-186)
                        if (!exact_mode && updated != 0.0 && fabs(updated) <= 1e-15 * (fabs(old) + fabs(product))) updated = 0.0;
                        if (updated == 0.0) {
                            col[s] = -1;
                            atomicSub(&w.ccount[cc], 1);
                        } else {
                            val[s] = updated;
                        }
                        found = true;
                    }
                if (__ballot(found) == 0ull) {  // fill-in: the next free slot
                    if (len >= LUF_MAX_ROW) {
                        if (lane == 0) sh.error = LUF_ERR_LONG_ROW;
                    } else {
                        const int slot = len >> 6;
                        if (lane == (len & (WAVE - 1))) {
#pragma unroll
                            for (int s = 0; s < LUF_ROW_SLOTS; ++s)
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
    // the new row, holes squeezed out, into the other arena; its largest magnitude for the next rounds' threshold test
    const int d0 = w.r_newstart[r];
    const int capacity = max(0, w.r_len[r] + w.growth[r]);
    int* __restrict__ ncol = (cur ? w.a_col[0] : w.a_col[1]);
    int* __restrict__ nrow = (cur ? w.a_row[0] : w.a_row[1]);
    double* __restrict__ nval = (cur ? w.a_val[0] : w.a_val[1]);
    int written = 0;
    double biggest = 0.0;
#pragma unroll
    for (int s = 0; s < LUF_ROW_SLOTS; ++s) {
        const bool live = col[s] >= 0;
        const unsigned long long mask = __ballot(live);
        if (live) {
            const int at = written + lanes_below(mask);
            if (at < capacity) {
                ncol[d0 + at] = col[s];
                nrow[d0 + at] = r;
                nval[d0 + at] = val[s];
            }
            biggest = fmax(biggest, fabs(val[s]));
        }
        written += __popcll(mask);
    }
    for (int x = written + lane; x < capacity; x += WAVE) ncol[d0 + x] = -1;
    biggest = wave_max_f64(biggest);
    if (lane == 0) {
        if (written == 0) sh.error = LUF_ERR_SINGULAR;
        if (written > capacity) sh.error = LUF_ERR_ARENA;  // (cannot happen: the capacity is the bound old - pivots + sum of the pivot rows)
        w.r_len[r] = written;
        w.rmaxd[r] = biggest;
    }
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

// `lds_level`: which of the per-row work arrays live in LDS instead of global memory (2: all of them, m <= ~1900; 1: the ones the
// entry passes gather from, m <= ~3500; 0: none).  The kernel reaches them through generic pointers either way: a round is a chain of
// dependent gathers and atomics on these arrays, ~150 cycles each out of LDS against ~800 through L2.
template <int lds_level>
__global__ void __launch_bounds__(LUF_THREADS) lu_factor_kernel(LuFactorSource src, LuFactorWork w_in, LuFactorOut out, double threshold,
                                                                int reference_ties, int dense_tail) {
    extern __shared__ unsigned char luf_dynamic_lds[];
    __shared__ FactorShared sh;
    LuFactorWork w = w_in;
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
        unsigned long long carry = 0;
        for (int base = 0; base < m; base += T) {
            const int i = base + tid;
            const unsigned long long v = i < m ? (unsigned long long)w.r_len[i] : 0ull;
            unsigned long long total;
            const unsigned long long ex = block_exclusive_scan(v, sh, &total) + carry;
            if (i < m) {
                w.r_start[i] = (int)ex;
                if (v == 0) sh.error = LUF_ERR_SINGULAR;
            }
            carry += total;
        }
        if (tid == 0) {
            sh.top = (int)carry;
            sh.peak = (int)carry;
            w.info[LUF_NNZ_B] = (int)carry;
            if (carry > (unsigned long long)w.cap_w) sh.error = LUF_ERR_ARENA;
        }
    }
    __syncthreads();
    if (sh.error == LUF_OK) {
        for (int j = tid; j < m; j += T) {
            const int cj = src.basis ? src.basis[j] : j;
            const double sign = (src.flipped && src.flipped[cj]) ? -1.0 : 1.0;
            for (int e = src.col_start[cj]; e < src.col_start[cj + 1]; ++e) {
                const double v = src.value[e];
                if (v == 0.0) continue;
                const int row = src.row_index[e];
                const int at = w.r_start[row] + atomicAdd(&w.tmp_cursor[row], 1);
                w.a_col[0][at] = j;
                w.a_row[0][at] = row;
                w.a_val[0][at] = sign * v;
                atomicMax(&w.rmax[row], (unsigned long long)__double_as_longlong(fabs(v)));
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < m; i += T) w.rmaxd[i] = __longlong_as_double((long long)w.rmax[i]);
    __syncthreads();
    LUF_STAMP(0);

    // ---- rounds ----------------------------------------------------------------------------------------------------------------
    while (sh.error == LUF_OK && sh.n_active > 0) {
        const int n_active = sh.n_active, cur = sh.cur, top = sh.top, kbase = sh.kbase, ubase = sh.ubase;
        if (!ref && n_active <= dense_tail) break;  // the rest goes through the dense tail
        const int* __restrict__ act = (cur ? w.active[1] : w.active[0]);
        const int* __restrict__ acol = (cur ? w.a_col[1] : w.a_col[0]);
        const int* __restrict__ arow = (cur ? w.a_row[1] : w.a_row[0]);
        const double* __restrict__ aval = (cur ? w.a_val[1] : w.a_val[0]);
        if (tid == 0) {
            sh.smin = 0x7fffffff;
            sh.best64 = NONE64;
        }
        __syncthreads();
        // (1) every active row's best admissible entry
        if (!ref) {
            for (int e = tid; e < top; e += T) {
                const int c = acol[e];
                if (c < 0) continue;
                const int r = arow[e];
                const unsigned key = candidate_key(w, r, c, aval[e], threshold);
                if (key != NONE32) atomicMin(&w.rowbest[r], key);
            }
        } else {  // the reference's rule: ONE pivot, minimum score, ties by current column position, then row position
            unsigned long long mine = NONE64;
            for (int e = tid; e < top; e += T) {
                const int c = acol[e];
                if (c < 0) continue;
                const int r = arow[e];
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
            // (2) the round's minimum score; candidates within a slack of it compete (4 x, at least + 4: lu_factor.hpp)
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
                break;
            }
            limit = max(4 * smin, smin + 4);
            // (3) the entry of each competing candidate; the best candidate per column
            for (int e = tid; e < top; e += T) {
                const int c = acol[e];
                if (c < 0) continue;
                const int r = arow[e];
                const unsigned best = w.rowbest[r];
                if ((int)(best >> 20) > limit || (int)(best & 0xffffu) != c) continue;
                if (candidate_key(w, r, c, aval[e], threshold) != best) continue;
                w.best_e[r] = e;
                atomicMin(&w.colmark[c], priority_of(best, r));
            }
            __syncthreads();
            LUF_STAMP(2);
            // (4) conflicts: an entry (r, c) with c the pivot column of another row's candidate and r a candidate row itself -- the
            //     two pivots are not compatible, the worse one waits for a later round
            for (int e = tid; e < top; e += T) {
                const int c = acol[e];
                if (c < 0) continue;
                const int r = arow[e];
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
                break;
            }
            const int rw = w.row_at[best & 0xffffull], cw = w.col_at[(best >> 16) & 0xffffull];
            if (tid == 0) {
                sh.ref_row = rw;
                sh.ref_col = cw;
            }
            for (int e = tid; e < top; e += T)
                if (acol[e] == cw && arow[e] == rw) w.best_e[rw] = e;
            __syncthreads();
        }
        LUF_STAMP(3);
        // (5) the accepted pivots take consecutive positions in row order; their rows become rows of U
        {
            unsigned long long carry = 0;
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
                const unsigned long long v = accepted ? ((1ull << 32) | (unsigned long long)(w.r_len[i] - 1)) : 0ull;
                unsigned long long total;
                const unsigned long long ex = block_exclusive_scan(v, sh, &total) + carry;
                if (accepted) {
                    const int k = kbase + (int)(ex >> 32);
                    const int e = w.best_e[i];
                    const int c = acol[e];
                    w.pivk_row[i] = k;
                    w.pivk_col[c] = k;
                    out.rowpos[i] = k;
                    out.colpos[c] = k;
                    out.diag[k] = aval[e];
                    w.ut_start[k] = ubase + (int)(ex & 0xffffffffull);
                }
                carry += total;
            }
            if (tid == 0) {
                sh.n_acc = (int)(carry >> 32);
                sh.u_round = (int)(carry & 0xffffffffull);
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
        if (sh.error != LUF_OK) break;
        LUF_STAMP(4);
        // (6) pivot rows -> U (decomposition/mod.rs:60-70); the rows with an entry in a pivot column are this round's targets
        for (int e = tid; e < top; e += T) {
            const int c = acol[e];
            if (c < 0) continue;
            const int r = arow[e];
            const int k = w.pivk_row[r];
            if (k >= 0) {
                const int be = w.best_e[r];
                if (e == be) continue;
                const int at = w.ut_start[k] + (e - w.r_start[r]) - (e > be ? 1 : 0);
                w.ut_col[at] = c;
                w.ut_val[at] = aval[e];
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
            unsigned long long carry = 0;
            int* __restrict__ act_new = (cur ? w.active[0] : w.active[1]);
            for (int base = 0; base < n_active; base += T) {
                const int t = base + tid;
                int i = -1;
                unsigned long long v = 0;
                bool target = false;
                if (t < n_active) {
                    i = act[t];
                    if (w.pivk_row[i] < 0) {
                        target = w.tflag[i] != 0;
                        const int capacity = max(0, w.r_len[i] + w.growth[i]);
                        v = (unsigned long long)capacity | (1ull << 28) | (target ? 1ull << 46 : 0ull);
                    }
                }
                unsigned long long total;
                const unsigned long long ex = block_exclusive_scan(v, sh, &total) + carry;
                if (v != 0) {
                    w.r_newstart[i] = (int)(ex & 0xfffffffull);
                    act_new[(ex >> 28) & 0x3ffffull] = i;
                    if (target) w.targets[ex >> 46] = i;
                }
                carry += total;
            }
            if (tid == 0) {
                sh.top_new = (int)(carry & 0xfffffffull);
                sh.n_active_new = (int)((carry >> 28) & 0x3ffffull);
                sh.n_targets = (int)(carry >> 46);
                if (sh.top_new > w.cap_w) sh.error = LUF_ERR_ARENA;
                sh.peak = max(sh.peak, sh.top_new);
            }
        }
        __syncthreads();
        if (sh.error != LUF_OK) break;
        LUF_STAMP(6);
        // (8) the untouched rows are copied, the targets eliminated (decomposition/mod.rs:71-100,146-210) -- into the other arena
        {
            int* __restrict__ ncol = (cur ? w.a_col[0] : w.a_col[1]);
            int* __restrict__ nrow = (cur ? w.a_row[0] : w.a_row[1]);
            double* __restrict__ nval = (cur ? w.a_val[0] : w.a_val[1]);
            for (int e = tid; e < top; e += T) {
                const int c = acol[e];
                if (c < 0) continue;
                const int r = arow[e];
                if (w.pivk_row[r] >= 0 || w.tflag[r]) continue;
                const int at = w.r_newstart[r] + (e - w.r_start[r]);
                ncol[at] = c;
                nrow[at] = r;
                nval[at] = aval[e];
            }
            const int n_targets = sh.n_targets;
            for (int t = wave; t < n_targets; t += LUF_WAVES) eliminate_row(w, sh, out, w.targets[t], cur, ref);
        }
        __syncthreads();
        LUF_STAMP(7);
        // (9) the next round's state
        {
            const int n_new = sh.n_active_new;
            const int* __restrict__ act_new = (cur ? w.active[0] : w.active[1]);
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
    __syncthreads();
    LUF_STAMP(8);

    // ---- dense tail: the last rows by partial pivoting out of LDS, one wave ------------------------------------------------------
    if (sh.error == LUF_OK && sh.n_active > 0) {
        const int n = sh.n_active, cur = sh.cur, top = sh.top, kbase = sh.kbase;
        const int* __restrict__ act = (cur ? w.active[1] : w.active[0]);
        // local column numbers: the unpivoted columns in ascending order (ordered compaction over all columns)
        {
            unsigned long long carry = 0;
            for (int base = 0; base < m; base += T) {
                const int c = base + tid;
                const bool open = c < m && out.colpos[c] < 0;
                unsigned long long total;
                const unsigned long long ex = block_exclusive_scan(open ? 1ull : 0ull, sh, &total) + carry;
                if (open) {
                    w.growth[c] = (int)ex;  // (growth: free between rounds) local number of column c
                    if (ex < (unsigned long long)LUF_DENSE_MAX) dense_cols[ex] = c;
                }
                carry += total;
            }
            if (tid == 0 && (int)carry != n) sh.error = LUF_ERR_SINGULAR;
        }
        for (int t = tid; t < n; t += T) w.tflag[act[t]] = t;  // local number of a row
        for (int x = tid; x < LUF_DENSE_MAX * (LUF_DENSE_MAX + 1); x += T) (&dense[0][0])[x] = 0.0;
        __syncthreads();
        if (sh.error == LUF_OK) {
            for (int e = tid; e < top; e += T) {
                const int c = (cur ? w.a_col[1] : w.a_col[0])[e];
                if (c < 0) continue;
                dense[w.tflag[(cur ? w.a_row[1] : w.a_row[0])[e]]][w.growth[c]] = (cur ? w.a_val[1] : w.a_val[0])[e];
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
                            const double old = dense[lane][j], product = ratio * pj;
                            double updated = old - product;
                            if (updated != 0.0 && fabs(updated) <= 1e-15 * (fabs(old) + fabs(product))) updated = 0.0;
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
            unsigned long long carry = 0;
            for (int base = 0; base < m; base += T) {
                const int p = base + tid;
                const unsigned long long v = p < m ? (unsigned long long)w.tmp_cursor[p] : 0ull;
                unsigned long long total;
                const unsigned long long ex = block_exclusive_scan(v, sh, &total) + carry;
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
    ww.cap_l = o.cap_l;
    ww.cap_u = o.cap_u;
    static PerDeviceOnce once;
    once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_factor_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_factor_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    });
    const size_t m8 = ((size_t)w.m * sizeof(int) + 7) & ~size_t(7);
    const size_t level1 = (size_t)w.m * sizeof(double) + 8 * m8, level2 = level1 + 7 * m8 + m8 + 8;
    static const int forced = getenv("RELP_LUF_LDS") ? atoi(getenv("RELP_LUF_LDS")) : -1;  // diagnostic: 0 keeps every work array in global memory
    int lds_level = level2 <= (size_t)140 * 1024 ? 2 : level1 <= (size_t)140 * 1024 ? 1 : 0;
    if (forced >= 0) lds_level = std::min(lds_level, forced);
    const size_t lds = lds_level == 2 ? level2 : lds_level == 1 ? level1 : 0;
    if (lds_level == 2) hipLaunchKernelGGL(lu_factor_kernel<2>, dim3(1), dim3(LUF_THREADS), lds, stream, src, ww, o, threshold, reference_ties, dense_tail);
    else if (lds_level == 1) hipLaunchKernelGGL(lu_factor_kernel<1>, dim3(1), dim3(LUF_THREADS), lds, stream, src, ww, o, threshold, reference_ties, dense_tail);
    else hipLaunchKernelGGL(lu_factor_kernel<0>, dim3(1), dim3(LUF_THREADS), lds, stream, src, ww, o, threshold, reference_ties, dense_tail);
}

}  // namespace relp
