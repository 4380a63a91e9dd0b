// What `BasisInverse::invert` produces for the inverse-factor carry, built ON THE DEVICE from the factors lu_factor.hip left there:
//   * L^-1 (strict part) and U^-1 (with its diagonal) as sparse matrices of the position space -- the role of `lu_invert_factors`
//     (lu_host.hpp):  row_i(L^-1) = e_i - sum_{j<i} l_ij row_j(L^-1),   row_i(U^-1) = (e_i - sum_{j>i} u_ij row_j(U^-1)) / u_ii;
//   * both inverses in both orientations as the COMPACT SLOT RECORDS the product kernels stream (`LuTasks::c_hdr / c_col / c_val`,
//     lu.hpp; the role of count_inverse_slots / fill_inverse_records in lu.hip): rows packed widest group first, a row = 1, 2, 4 ...
//     64 aligned slots of four entries, per-wave summaries in the headers, rows of more than 256 entries with their tail in the
//     extras arena, rows without entries in the z list.
// Reference lines this stands for: lower_upper/mod.rs:78-92 (`invert`), :180-237 (what the solves read).
//
// The inversion is a DATAFLOW over rows, one workgroup: wave w takes the rows w, w + 16, ... of L^-1 in ascending order (then the
// rows of U^-1 in descending order); a row waits for the rows it reads by polling their length word (workgroup-scope acquire: the
// waves of one workgroup share the CU's L1, so a release is a wait for the stores, not a cache write-back), accumulates
// `- l_ij row_j` into a dense accumulator of its wave (LDS when sixteen of them fit, else global) with a bitmap of the touched
// columns, and emits its entries in COLUMN ORDER by walking the bitmap -- no sort, and nothing depends on which wave finished first.
// The smallest unfinished row never waits (everything it reads is finished), so the schedule cannot deadlock.
#include "lu_factor.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

#include "lu.hpp"
#include "solver.hpp"
#include "wave_ops.hpp"

namespace relp {

namespace {

constexpr int LUT_THREADS = LUF_THREADS;
constexpr int LUT_WAVES = LUT_THREADS / WAVE;

typedef __attribute__((address_space(3))) double lds_f64_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;

struct TaskShared {
    unsigned long long scan[LUT_WAVES + 2];
    int cursor;
    int next_row;
    int stuck_row, stuck_on;
    int error;
    int totals[16];
    unsigned long long dbg[4];  // cycle sums over the waves (lane 0): waiting for rows | streaming + accumulating | emitting | rows
};

__device__ __forceinline__ unsigned long long t_wave_inclusive_scan(unsigned long long v) {
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
__device__ __forceinline__ unsigned long long t_block_exclusive_scan(unsigned long long v, TaskShared& sh, unsigned long long* total) {
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const unsigned long long incl = t_wave_inclusive_scan(v);
    __syncthreads();
    if (lane == WAVE - 1) sh.scan[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const unsigned long long w = lane < LUT_WAVES ? sh.scan[lane] : 0ull;
        const unsigned long long wi = t_wave_inclusive_scan(w);
        if (lane < LUT_WAVES) sh.scan[lane] = wi - w;
        if (lane == LUT_WAVES - 1) sh.scan[LUT_WAVES] = wi;
    }
    __syncthreads();
    *total = sh.scan[LUT_WAVES];
    return sh.scan[wave] + incl - v;
}
__device__ __forceinline__ int t_lane_value(int v, int y) { return __builtin_amdgcn_readlane(v, y); }  // (y wave-uniform: an SGPR move, not a ds_bpermute)
__device__ __forceinline__ double t_lane_value(double v, int y) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), y), __builtin_amdgcn_readlane(__double2loint(v), y));
}
__device__ __forceinline__ int t_lanes_below(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ int t_group_log2(int n) {  // smallest g with 4 << g >= n, at most 6 (group_log2 of lu.hip)
    int g = 0;
    while (g < 6 && (LU_TE << g) < n) ++g;
    return g;
}
// a finished row's descriptor: (length + 1) << 32 | first entry in the raw arena; 0: not finished yet
// (the descriptors live in LDS: a poll is a ds_read, ~100 cycles, where a word in global memory cost an L2 round trip per look.
//  Plain volatile LDS accesses with explicit fences: the row's entries are global stores, the release waits for them.)
__device__ __forceinline__ unsigned long long poll_descriptor(volatile lds_u64_t* p) { return *p; }
__device__ __forceinline__ void publish_descriptor(volatile lds_u64_t* p, unsigned long long v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *p = v;
}
__device__ __forceinline__ unsigned long long row_descriptor(int start, int len) { return ((unsigned long long)(unsigned)(len + 1) << 32) | (unsigned)start; }

// One row of an inverse by one wave.  `src_*`: the entries (j, f_ij) of row i of the factor; `UPPER`: row of U^-1 (starts from e_i,
// scaled by 1 / u_ii at the end, the rows it reads carry their diagonal) or of L^-1 (strict part: the rows it reads have an implied 1).
template <bool ACC_LDS, bool UPPER>
__device__ void invert_row(const LuInverseWork& iw, TaskShared& sh, const int i, const int* src_col, const double* src_val, const int s, const int e,
                           const double scale, volatile lds_f64_t* acc_lds, double* acc_glb, volatile lds_u64_t* bits, const int words,
                           const int arena_first, volatile lds_u64_t* desc) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int f = UPPER ? 1 : 0;
    long long t_mark = clock64();
    auto mark = [&](int k) {
        const long long now = clock64();
        if (lane == 0) atomicAdd(&sh.dbg[k], (unsigned long long)(now - t_mark));
        t_mark = now;
    };
    auto add = [&](int c, double delta) {  // (the lanes of one step hold distinct columns)
        if (ACC_LDS) acc_lds[c] = acc_lds[c] + delta;
        else acc_glb[c] = acc_glb[c] + delta;
        atomicOr((unsigned long long*)(bits + (c >> 6)), 1ull << (c & 63));
    };
    if (UPPER && lane == 0) add(i, 1.0);
    for (int x0 = s; x0 < e; x0 += WAVE) {
        // the descriptors of up to 64 rows this one reads: wait until each is finished (they are: lower rows of L^-1, higher of U^-1)
        int j = 0, sj = 0, nj = 0;
        double fij = 0.0;
        if (x0 + lane < e) {
            j = src_col[x0 + lane];
            fij = src_val[x0 + lane];
            unsigned long long d;
            int spins = 0;
            while ((d = poll_descriptor(desc + j)) == 0ull) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 21)) {  // (a row that never arrives: reported, not waited for -- about 0.1 s)
                    sh.error = LUF_ERR_DATAFLOW;
                    sh.stuck_row = i;
                    sh.stuck_on = j;
                    d = row_descriptor(0, 0);
                    break;
                }
            }
            nj = (int)(d >> 32) - 1;
            sj = (int)(unsigned)d;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // the rows read below were written before their descriptors
        mark(0);
        const int cnt = min(WAVE, e - x0);
        // The rows read are streamed FOUR at a time: their first 64 entries are requested together (one global round trip for four
        // source rows instead of four), then accumulated one source row after the other -- in source order, so that a column's sum has
        // one order whatever the timing.  The rest of a long source row follows in batches of four pieces.
        constexpr int PF = 4;
        for (int y0 = 0; y0 < cnt; y0 += PF) {
            int jj[PF], sjj[PF], njj[PF], c0[PF];
            double factor[PF], v0[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const bool live = y0 + u < cnt;
                const int y = live ? y0 + u : y0;
                jj[u] = t_lane_value(j, y);
                sjj[u] = t_lane_value(sj, y);
                njj[u] = live ? t_lane_value(nj, y) : -1;  // -1: no such source
                factor[u] = t_lane_value(fij, y);
                c0[u] = lane < njj[u] ? iw.raw_col[sjj[u] + lane] : -1;
                v0[u] = lane < njj[u] ? iw.raw_val[sjj[u] + lane] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (njj[u] < 0) continue;  // (wave-uniform)
                if (!UPPER && lane == 0) add(jj[u], -factor[u]);  // the unit diagonal of row jj of L^-1
                if (c0[u] >= 0) add(c0[u], -factor[u] * v0[u]);
                for (int z0 = WAVE; z0 < njj[u]; z0 += PF * WAVE) {
                    int cz[PF];
                    double vz[PF];
#pragma unroll
                    for (int t = 0; t < PF; ++t) {
                        const int z = z0 + t * WAVE + lane;
                        cz[t] = z < njj[u] ? iw.raw_col[sjj[u] + z] : -1;
                        vz[t] = z < njj[u] ? iw.raw_val[sjj[u] + z] : 0.0;
                    }
#pragma unroll
                    for (int t = 0; t < PF; ++t)
                        if (cz[t] >= 0) add(cz[t], -factor[u] * vz[t]);
                }
                // (global accumulators: the next row's read-modify-writes must see these stores -- another lane may hold the column then)
                if (!ACC_LDS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
        }
    }
    mark(1);
    // emit in column order: the set bits of the bitmap, word by word
    int total = 0;
    for (int w0 = 0; w0 < words; w0 += WAVE) {
        const int w = w0 + lane;
        const unsigned long long word = w < words ? bits[w] : 0ull;
        total += __popcll(word);
    }
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) total += __shfl_xor(total, d, WAVE);
    int at = 0;
    if (lane == 0) at = arena_first + atomicAdd(&sh.cursor, total);
    at = t_lane_value(at, 0);
    if (at + total > arena_first + iw.raw_cap / 2) {
        if (lane == 0) sh.error = LUF_ERR_INVERSE_CAPACITY;
        total = 0;
    }
    int written = 0;
    for (int w0 = 0; w0 < words; w0 += WAVE) {
        const int w = w0 + lane;
        unsigned long long word = w < words ? bits[w] : 0ull;
        if (w < words) bits[w] = 0ull;
        const int mine = __popcll(word);
        // exclusive prefix of `mine` over the lanes
        int incl = mine;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const int other = __shfl_up(incl, d, WAVE);
            if (lane >= d) incl += other;
        }
        int dst = at + written + incl - mine;
        while (word) {
            const int b = __ffsll((long long)word) - 1;
            word &= word - 1;
            const int c = (w << 6) | b;
            double v;
            if (ACC_LDS) { v = acc_lds[c]; acc_lds[c] = 0.0; }
            else { v = acc_glb[c]; acc_glb[c] = 0.0; }
            if (total) {  // (exact zeros from cancellation stay as explicit entries: they are rare and harmless to a product)
                iw.raw_col[dst] = c;
                iw.raw_val[dst] = v * scale;
            }
            ++dst;
        }
        written += __shfl(incl, WAVE - 1, WAVE);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // every lane's stores have left before the length is published
    if (lane == 0) {
        iw.raw_len[f][i] = total;
        iw.raw_start[f][i] = at;
        publish_descriptor(desc + i, row_descriptor(at, total));
        atomicAdd(&sh.dbg[3], 1ull);
    }
    mark(2);
}

// Two workgroups: block 0 inverts L, block 1 inverts U (independent; each has its own half of the raw arena and its own waves).
template <bool ACC_LDS>
__global__ void __launch_bounds__(LUT_THREADS) lu_invert_kernel(LuFactorOut fac, LuInverseWork iw, const int* status_in) {
    extern __shared__ unsigned char dyn_lds[];
    __shared__ TaskShared sh;
    const int tid = threadIdx.x, T = LUT_THREADS;
    const int lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int m = iw.m;
    const int words = (m + 63) / 64;
    const int f = blockIdx.x;  // 0: L^-1, 1: U^-1
    if (status_in && status_in[LUF_STATUS] != LUF_OK) return;  // the factorisation failed: nothing to invert (the host falls back)
    volatile lds_u64_t* desc = (volatile lds_u64_t*)dyn_lds;      // [m] descriptors of the finished rows
    volatile lds_u64_t* bits_all = (volatile lds_u64_t*)(dyn_lds + (size_t)m * sizeof(unsigned long long));
    volatile lds_f64_t* acc_all = (volatile lds_f64_t*)(dyn_lds + ((size_t)m + (size_t)LUT_WAVES * words) * sizeof(unsigned long long));
    double* acc_block = iw.acc + (size_t)f * LUT_WAVES * m;
    if (tid == 0) {
        sh.cursor = 0;
        sh.next_row = 0;
        sh.error = LUF_OK;
        for (int k = 0; k < 4; ++k) sh.dbg[k] = 0ull;
    }
    const long long t_kernel = clock64();
    for (int x = tid; x < LUT_WAVES * words; x += T) bits_all[x] = 0ull;
    if (ACC_LDS)
        for (int x = tid; x < LUT_WAVES * m; x += T) acc_all[x] = 0.0;
    else
        for (int x = tid; x < LUT_WAVES * m; x += T) acc_block[x] = 0.0;
    for (int i = tid; i < m; i += T) desc[i] = 0ull;
    __syncthreads();
    volatile lds_u64_t* bits = bits_all + (size_t)wave * words;
    volatile lds_f64_t* acc_lds = acc_all + (size_t)wave * m;
    double* acc_glb = acc_block + (size_t)wave * m;
    const int arena_first = f * (iw.raw_cap / 2);
    // Rows are taken in dependency order, row k by wave k mod 16 (or claimed from a counter: see launch_lu_invert).  The smallest
    // unfinished row is always in progress, and everything it reads is finished.
    const bool static_rows = iw.static_rows != 0;
    for (int turn = 0;; ++turn) {
        int k = 0;
        if (static_rows) {
            k = wave + turn * LUT_WAVES;
        } else {
            k = atomicAdd(&sh.next_row, lane == 0 ? 1 : 0);  // (every lane takes part, lane 0 adds the one: see lu_factor.hip)
            k = __builtin_amdgcn_readfirstlane(k);
        }
        if (k >= m) break;
        if (f == 0) {  // L^-1, ascending
            const int i = k;
            const int s = fac.l_start[i], e = fac.l_start[i + 1];
            if (s == e) {
                if (lane == 0) {
                    iw.raw_len[0][i] = 0;
                    iw.raw_start[0][i] = arena_first;
                    publish_descriptor(desc + i, row_descriptor(arena_first, 0));
                }
                continue;
            }
            invert_row<ACC_LDS, false>(iw, sh, i, fac.l_col, fac.l_val, s, e, 1.0, acc_lds, acc_glb, bits, words, arena_first, desc);
        } else {       // U^-1, descending
            const int i = m - 1 - k;
            invert_row<ACC_LDS, true>(iw, sh, i, fac.u_col, fac.u_val, fac.u_start[i], fac.u_start[i + 1], 1.0 / fac.diag[i], acc_lds, acc_glb, bits, words,
                                      arena_first, desc);
        }
    }
    __syncthreads();
    // ---- this inverse by rows, compact, in row order (rows are already sorted by column) ----------------------------------------
    {
        unsigned long long carry = 0;
        for (int base = 0; base < m; base += T) {
            const int i = base + tid;
            const unsigned long long v = i < m ? (unsigned long long)iw.raw_len[f][i] : 0ull;
            unsigned long long total;
            const unsigned long long ex = t_block_exclusive_scan(v, sh, &total) + carry;
            if (i < m) iw.csr_start[f][i] = (int)ex;
            carry += total;
        }
        if (tid == 0) {
            iw.csr_start[f][m] = (int)carry;
            iw.info[f == 0 ? LUF_NNZ_LI : LUF_NNZ_UI] = (int)carry;
            if (carry > (unsigned long long)iw.cap) sh.error = LUF_ERR_INVERSE_CAPACITY;
        }
        __syncthreads();
        if (sh.error == LUF_OK) {
            for (int i = wave; i < m; i += LUT_WAVES) {
                const int n = iw.raw_len[f][i], s = iw.raw_start[f][i], d = iw.csr_start[f][i];
                for (int t = lane; t < n; t += WAVE) {
                    iw.csr_idx[f][d + t] = iw.raw_col[s + t];
                    iw.csr_val[f][d + t] = iw.raw_val[s + t];
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0 && sh.error != LUF_OK) iw.info[LUF_STATUS] = sh.error;
    if (tid == 0) {  // diagnostic: kilo-cycles of this block -- whole kernel | waiting | streaming + accumulating | emitting (summed over the waves)
        int* out = iw.info + 23 - 4 * f;  // words 23.. for L^-1 (block 0), 19.. for U^-1 (block 1)
        out[0] = (int)((clock64() - t_kernel) >> 10);
        if (sh.error == LUF_ERR_DATAFLOW) {
            iw.info[29] = sh.stuck_row;
            iw.info[30] = sh.stuck_on;
        }
        for (int k = 0; k < 3; ++k) out[1 + k] = (int)(sh.dbg[k] >> 10);
    }
}

// Transposes of the two inverses (lists 2 and 3): two workgroups, block f turns list f (by rows) into its column orientation.
__global__ void __launch_bounds__(LUT_THREADS) lu_transpose_inverse_kernel(LuInverseWork iw_in, Ctl* ctl, int failed_status) {
    __shared__ TaskShared sh;
    const int tid = threadIdx.x, T = LUT_THREADS;
    const int lane = tid & (WAVE - 1);
    LuInverseWork iw = iw_in;
    const int m = iw.m;
    if (iw.info[LUF_STATUS] != LUF_OK) {
        if (tid == 0 && ctl) ctl->status = failed_status;  // the pivots enqueued behind this become no-ops; the host refactorises
        return;
    }
    iw.cursor += (size_t)blockIdx.x * LUT_WAVES * m;  // (each block its own counters)
    // ---- column orientations: list 2 = U^-1 by columns (from list 1), list 3 = L^-1 by columns (from list 0) -------------------
    {
        const int f = blockIdx.x;
        const int* __restrict__ rs = f == 0 ? iw.csr_start[0] : iw.csr_start[1];
        const int* __restrict__ ri = f == 0 ? iw.csr_idx[0] : iw.csr_idx[1];
        const double* __restrict__ rv = f == 0 ? iw.csr_val[0] : iw.csr_val[1];
        int* __restrict__ ds = f == 0 ? iw.csr_start[3] : iw.csr_start[2];
        int* __restrict__ di = f == 0 ? iw.csr_idx[3] : iw.csr_idx[2];
        double* __restrict__ dv = f == 0 ? iw.csr_val[3] : iw.csr_val[2];
        const int nnz = rs[m];
        // A STABLE transpose without a sort: the rows are cut into sixteen consecutive chunks, one per wave; a wave counts its chunk's
        // entries per column (its own row of counters), a thread per column turns the sixteen counts into the chunks' first positions
        // inside the column, and every wave then walks its rows IN ORDER handing out positions from its own counters -- the rows of a
        // column come out ascending whatever the timing.  (Round 4's first form scattered with one atomic cursor per column and
        // rank-sorted every column afterwards: 193 us per refactorisation of 25FV47.)
        const int wave = tid / WAVE;
        const int chunk_rows = (m + LUT_WAVES - 1) / LUT_WAVES;
        const int row0 = min(m, wave * chunk_rows), row1 = min(m, row0 + chunk_rows);
        int* __restrict__ counts = iw.cursor;   // [16][m]
        int* __restrict__ mine = counts + (size_t)wave * m;
        for (int x = tid; x < LUT_WAVES * m; x += T) counts[x] = 0;
        __syncthreads();
        // (the counting pass needs no order: the chunk's entries are one contiguous range, walked flat)
        for (int x = rs[row0] + lane; x < rs[row1]; x += WAVE) atomicAdd(&mine[ri[x]], 1);  // (no value returned: nothing to wait for)
        __syncthreads();
        {
            unsigned long long carry = 0;
            for (int base = 0; base < m; base += T) {
                const int c = base + tid;
                unsigned long long v = 0;
                if (c < m) {
                    int running = 0;
                    for (int w = 0; w < LUT_WAVES; ++w) {
                        const int t = counts[(size_t)w * m + c];
                        counts[(size_t)w * m + c] = running;
                        running += t;
                    }
                    v = (unsigned long long)running;
                }
                unsigned long long total;
                const unsigned long long ex = t_block_exclusive_scan(v, sh, &total) + carry;
                if (c < m) ds[c] = (int)ex;
                carry += total;
            }
            if (tid == 0) {
                ds[m] = nnz;
                if ((int)carry != nnz) iw.info[LUF_STATUS] = LUF_ERR_DATAFLOW;
            }
        }
        __syncthreads();
        for (int i0 = row0; i0 < row1; i0 += WAVE) {  // (the row starts of 64 rows in one load, then row after row)
            const int my_start = i0 + lane <= row1 ? rs[min(i0 + lane, m)] : 0;
            const int my_end = i0 + lane < row1 ? rs[i0 + lane + 1] : 0;
            const int rows_here = min(WAVE, row1 - i0);
            for (int y = 0; y < rows_here; ++y) {
                const int a = t_lane_value(my_start, y), b = t_lane_value(my_end, y);
                for (int x = a + lane; x < b; x += WAVE) {
                    const int c = ri[x];
                    const int at = ds[c] + atomicAdd(&mine[c], 1);  // (this wave's counter only: positions in row order)
                    di[at] = i0 + y;
                    dv[at] = rv[x];
                }
            }
        }
        __syncthreads();
    }
    (void)lane;
    (void)sh;
}

// The compact records: four workgroups, block k packs list k (0 L^-1 by rows, 1 U^-1 by rows, 2 U^-1 by columns, 3 L^-1 by columns).
__global__ void __launch_bounds__(LUT_THREADS) lu_pack_inverse_kernel(DeviceLU lu, LuInverseWork iw_in, Ctl* ctl, int failed_status) {
    __shared__ TaskShared sh;
    const int tid = threadIdx.x, T = LUT_THREADS;
    const int lane = tid & (WAVE - 1);
    LuInverseWork iw = iw_in;
    const int m = iw.m;
    // one read of the status word per workgroup, then a block-uniform branch: another workgroup of this launch may write its own pack
    // error into the same word when it finishes (the lists are independent; a failed list is repaired by the host fallback the caller
    // runs on failed_status before anything solves with these records)
    if (tid == 0) sh.error = iw.info[LUF_STATUS];
    __syncthreads();
    if (sh.error != LUF_OK) {
        if (tid == 0 && ctl) ctl->status = failed_status;
        return;
    }
    iw.row_rank += (size_t)blockIdx.x * m;
    iw.row_xoff += (size_t)blockIdx.x * m;
    iw.row_first += (size_t)blockIdx.x * m;
    // ---- the compact records, list by list ------------------------------------------------------------------------------------------
    const int stride = lu.task_stride;
    for (int k = blockIdx.x; k == (int)blockIdx.x; ++k) {  // (one list per workgroup; a loop so that an error leaves through `break`)
        const LuTasks tk = k == 0 ? lu.tasks[0] : k == 1 ? lu.tasks[1] : k == 2 ? lu.tasks[2] : lu.tasks[3];
        unsigned* hdr = (unsigned*)tk.c_hdr;
        unsigned long long* colw = (unsigned long long*)tk.c_col;
        double* vals = (double*)tk.c_val;
        int* zpos = (int*)tk.c_zpos;
        int* xstart = (int*)tk.s_xstart;
        int* xn = (int*)tk.s_xn;
        int* xidx = (int*)tk.x_idx;
        double* xval = (double*)tk.x_val;
        int* counts = (int*)tk.counts;
        const int xcap = (k == 0 || k == 3) ? iw.cap_extra_l : iw.cap_extra_u;
        const int* __restrict__ rs = k == 0 ? iw.csr_start[0] : k == 1 ? iw.csr_start[1] : k == 2 ? iw.csr_start[2] : iw.csr_start[3];
        const int* __restrict__ ri = k == 0 ? iw.csr_idx[0] : k == 1 ? iw.csr_idx[1] : k == 2 ? iw.csr_idx[2] : iw.csr_idx[3];
        const double* __restrict__ rv = k == 0 ? iw.csr_val[0] : k == 1 ? iw.csr_val[1] : k == 2 ? iw.csr_val[2] : iw.csr_val[3];
        // (a) rank of every row inside its group class (ordered), rows without entries, extras: three packed scans
        unsigned long long ca = 0, cb = 0, cx = 0;
        for (int base = 0; base < m; base += T) {
            const int i = base + tid;
            int n = 0, g = -1;
            if (i < m) {
                n = rs[i + 1] - rs[i];
                g = n > 0 ? t_group_log2(n) : 7;  // 7: the z list
            }
            const unsigned long long va = (i < m && g < 4) ? 1ull << (16 * g) : 0ull;
            const unsigned long long vb = (i < m && g >= 4) ? 1ull << (16 * (g - 4)) : 0ull;
            const unsigned long long vx = n > LU_TE * 64 ? (unsigned long long)(n - LU_TE * 64) : 0ull;
            unsigned long long ta, tb, tx;
            const unsigned long long ea = t_block_exclusive_scan(va, sh, &ta) + ca;
            const unsigned long long eb = t_block_exclusive_scan(vb, sh, &tb) + cb;
            const unsigned long long ex = t_block_exclusive_scan(vx, sh, &tx) + cx;
            if (i < m) {
                const unsigned long long field = g < 4 ? ea >> (16 * g) : eb >> (16 * (g - 4));
                iw.row_rank[i] = (int)(field & 0xffffull);
                iw.row_xoff[i] = (int)ex;
            }
            ca += ta;
            cb += tb;
            cx += tx;
        }
        if (tid == 0) {
            int count[8];
            for (int g = 0; g < 4; ++g) count[g] = (int)((ca >> (16 * g)) & 0xffffull);
            for (int g = 4; g < 8; ++g) count[g] = (int)((cb >> (16 * (g - 4))) & 0xffffull);
            int slots = 0;
            for (int g = 6; g >= 0; --g) {
                sh.totals[g] = slots;  // first slot of the class
                slots += count[g] << g;
            }
            sh.totals[7] = count[7];   // rows without entries
            sh.totals[8] = slots;
            sh.totals[9] = (int)cx;
            if (slots + 1024 > stride) sh.error = LUF_ERR_TASK_CAPACITY;
            if ((int)cx > xcap) sh.error = LUF_ERR_TASK_CAPACITY;
        }
        __syncthreads();
        if (sh.error != LUF_OK) break;
        const int n_slots = sh.totals[8];
        // (b) headers of every row's slots, the z list
        for (int i = tid; i < m; i += T) {
            const int n = rs[i + 1] - rs[i];
            if (n == 0) {
                zpos[iw.row_rank[i]] = i;
                continue;
            }
            const int g = t_group_log2(n), G = 1 << g;
            const int first = sh.totals[g] + iw.row_rank[i] * G;
            const int extra = n - min(n, LU_TE * G);
            iw.row_first[i] = first;
            for (int j = 0; j < G; ++j) {
                hdr[first + j] = (unsigned)i | ((unsigned)g << 16) | ((j == G - 1) ? 1u << 19 : 0u) | (extra > 0 ? 1u << 20 : 0u);
                if (extra > 0) {
                    xstart[first + j] = iw.row_xoff[i];
                    xn[first + j] = extra;
                }
            }
        }
        __syncthreads();
        // (c) the entries of every slot: slot j of a row takes its entries j, j + G, j + 2 G, j + 3 G; the wave summaries
        for (int base = 0; base < n_slots; base += T) {
            const int sl = base + tid;
            unsigned h = 0;
            int g = 0;
            if (sl < n_slots) {
                h = hdr[sl];
                g = (int)(h >> 16) & 7;
                const int i = (int)(h & 0xffffu), G = 1 << g, j = sl & (G - 1);
                const int s = rs[i], n = rs[i + 1] - s;
                const int inline_n = min(n, LU_TE * G);
                unsigned long long cw = 0;
                double v[LU_TE] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < LU_TE; ++t) {
                    const int e = j + t * G;
                    if (e < inline_n) {
                        cw |= (unsigned long long)(unsigned)ri[s + e] << (16 * t);
                        v[t] = rv[s + e];
                    }
                }
                colw[sl] = cw;
#pragma unroll
                for (int t = 0; t < LU_TE; ++t) vals[(size_t)LU_TE * sl + t] = v[t];
                if ((h >> 20) & 1u) {  // the tail of a row of more than 256 entries, by the 64 slots of the row
                    const int xo = iw.row_xoff[i];
                    for (int e = inline_n + j; e < n; e += G) {
                        xidx[xo + e - inline_n] = ri[s + e];
                        xval[xo + e - inline_n] = rv[s + e];
                    }
                }
            }
            // bits 21-26: some row of this wave's 64 slots spans more than 2^j lanes; bit 27: some row of it has extras
            int gmax = sl < n_slots ? g : 0;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) gmax = max(gmax, __shfl_xor(gmax, d, WAVE));
            const unsigned long long any_extra = __ballot(sl < n_slots && ((h >> 20) & 1u));
            unsigned summary = 0;
            for (int j = 0; j < 6; ++j)
                if (gmax > j) summary |= 1u << (21 + j);
            if (any_extra) summary |= 1u << 27;
            if (sl < n_slots) hdr[sl] = h | summary;
        }
        // (d) padding the product reads: up to the next multiple of the workgroup, and the last slot of the stride
        {
            const int pad_end = min(stride, ((n_slots + T - 1) / T) * T + T);
            for (int sl = n_slots + tid; sl < pad_end; sl += T) {
                hdr[sl] = 0u;
                colw[sl] = 0ull;
#pragma unroll
                for (int t = 0; t < LU_TE; ++t) vals[(size_t)LU_TE * sl + t] = 0.0;
            }
            if (tid == 0) {
                hdr[stride - 1] = 0u;
                colw[stride - 1] = 0ull;
                for (int t = 0; t < LU_TE; ++t) vals[(size_t)LU_TE * (stride - 1) + t] = 0.0;
                counts[LU_CNT_Z] = sh.totals[7];
                counts[LU_CNT_SLOTS] = n_slots;
                counts[LU_CNT_LEVELS] = 1;
                counts[LU_CNT_CHUNKS] = 1;
                counts[LU_CNT_C0_END] = n_slots;
                counts[LU_CNT_C0_L0] = 0;
                counts[LU_CNT_C0_L1] = 1;
                counts[LU_CNT_C0_TAIL] = 1;
            }
        }
        __syncthreads();
    }
    // the inverse-factor form keeps U's diagonal inside U^-1: the diagonal array the kernels see is ones (as lu_invert_factors' out.diag)
    if (blockIdx.x == 0)
        for (int i = tid; i < m; i += T) lu.diag[i] = 1.0;
    __syncthreads();
    if (tid == 0 && sh.error != LUF_OK) {
        iw.info[LUF_STATUS] = sh.error;
        if (ctl) ctl->status = failed_status;
    }
    (void)lane;
}

}  // namespace

size_t lu_invert_lds_bytes(int m, bool* acc_in_lds) {
    const size_t words = (size_t)(m + 63) / 64;
    const size_t descriptors = (size_t)m * sizeof(unsigned long long);
    const size_t bitmap = (size_t)LUT_WAVES * words * sizeof(unsigned long long);
    const size_t acc = (size_t)LUT_WAVES * m * sizeof(double);
    const bool fits = descriptors + bitmap + acc <= (size_t)140 * 1024;
    if (acc_in_lds) *acc_in_lds = fits;
    return descriptors + bitmap + (fits ? acc : 0);
}

void launch_lu_invert(const LuFactorOut& factors, const LuInverseWork& iw_in, const int* status_in, hipStream_t stream) {
    LuInverseWork iw = iw_in;
    // (rows go to the waves by a fixed map, row k to wave k mod 16; RELP_LUI_CLAIM_ROWS=1: claimed from a counter instead -- measured slower,
    //  the claim is an LDS atomic every lane of the wave takes part in: 1.42 M against 1.22 M cycles for the L^-1 block of 25FV47)
    iw.static_rows = thread_tuning().has(RELP_SW_LUI_CLAIM_ROWS) ? 0 : 1;
    bool in_lds = false;
    const size_t lds = lu_invert_lds_bytes(iw.m, &in_lds);
    static PerDeviceOnce once;
    once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_invert_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_invert_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    if (in_lds) hipLaunchKernelGGL(lu_invert_kernel<true>, dim3(2), dim3(LUT_THREADS), lds, stream, factors, iw, status_in);
    else hipLaunchKernelGGL(lu_invert_kernel<false>, dim3(2), dim3(LUT_THREADS), lds, stream, factors, iw, status_in);
}

void launch_lu_pack_inverse(const DeviceLU& lu, const LuInverseWork& iw, Ctl* ctl, int failed_status, hipStream_t stream) {
    hipLaunchKernelGGL(lu_transpose_inverse_kernel, dim3(2), dim3(LUT_THREADS), 0, stream, iw, ctl, failed_status);
    hipLaunchKernelGGL(lu_pack_inverse_kernel, dim3(4), dim3(LUT_THREADS), 0, stream, lu, iw, ctl, failed_status);
}

}  // namespace relp
