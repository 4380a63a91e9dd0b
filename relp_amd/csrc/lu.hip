// LU basis factorisation on the device: FTRAN, BTRAN and the Forrest-Tomlin update as single-workgroup kernels for gfx950.
//
// Replaces (paths relative to /root/reference/src/algorithm/two_phase/tableau/inverse_maintenance/carry/lower_upper/):
//   left_multiply_by_basis_inverse  (FTRAN)     mod.rs:180-210  (+ left_multiply_by_{lower,upper}_inverse :286-321)
//   right_multiply_by_basis_inverse (BTRAN)     mod.rs:212-237  (+ right_multiply_by_{upper,lower}_inverse :347-397)
//   basis_inverse_row                           mod.rs:254-272
//   change_basis (Forrest-Tomlin)               mod.rs:94-178
//   EtaFile::{apply_right, apply_left, update_spike_pivot_value}   eta_file.rs:49-134
//
// Design (MI355X): one LP's factor is a few 10^4 non-zeros; a triangular solve with it is a dependency DAG, not a stream.
// ONE workgroup (16 waves on one CU) owns the solve; the vector lives in LDS; every row is a thread that gathers its
// entries and waits -- without blocking its wave -- for the entries' own rows to be published ("sync-free" solve: a
// dependency costs one LDS write + one LDS read, no barrier, no level sets to rebuild after an update).  The reference walks
// ordered maps (`BTreeMap`) with a column scan per popped entry; here both orientations of L and U are resident so every
// solve is a gather.  Everything is deterministic: a row adds its entries in storage order.
#include "lu.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>

#include "solver.hpp"
#include "wave_ops.hpp"

namespace relp {

// =====================================================================================================
// host: memory + upload
// =====================================================================================================
LuFactors::~LuFactors() {
    if (dev_) (void)hipFree(dev_);
    if (staging_) (void)hipHostFree(staging_);
}

void LuFactors::reserve(size_t device_bytes, size_t staging_bytes) {
    if (device_bytes > dev_capacity_) {
        if (dev_) (void)hipFree(dev_);
        dev_ = nullptr;
        dev_capacity_ = device_bytes + device_bytes / 2;
        RELP_HIP(hipMalloc(reinterpret_cast<void**>(&dev_), dev_capacity_));
    }
    if (staging_bytes > staging_capacity_) {
        if (staging_) (void)hipHostFree(staging_);
        staging_ = nullptr;
        staging_capacity_ = staging_bytes + staging_bytes / 2;
        RELP_HIP(hipHostMalloc(reinterpret_cast<void**>(&staging_), staging_capacity_, hipHostMallocDefault));
    }
}

namespace {
struct Carver {
    size_t offset = 0;
    template <class T>
    size_t take(size_t count) {
        offset = (offset + 15) & ~size_t(15);
        const size_t at = offset;
        offset += count * sizeof(T);
        return at;
    }
};

__global__ void __launch_bounds__(256) lu_init_kernel(DeviceLU lu) {
    const int m = lu.m;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        lu.u_rlen[i] = lu.u_rstart[i + 1] - lu.u_rstart[i];
        lu.u_app_len[i] = 0;
        lu.rank[i] = i;
        lu.seq[i] = i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        lu.state[LU_N_UPDATES] = 0;
        lu.state[LU_UC_TOP] = lu.u_app_first;  // the arena of replaced columns starts behind the base capacity
        lu.state[LU_ETA_TOP] = 0;
        lu.state[LU_FLAGS] = 0;
        lu.eta_start[0] = 0;
    }
}
}  // namespace

bool LuFactors::upload(const HostLU& f, int max_updates, hipStream_t stream) {
    const int m = f.m;
    const size_t nl = (size_t)f.nnz_l(), nu = (size_t)f.nnz_u();
    if (max_updates < 1) max_updates = 1;
    // The layout depends on capacities only, so that the device addresses (and a captured hipGraph that holds them) survive
    // a refactorisation; it changes when a factor outgrows its capacity (or m / the update capacity change).
    bool layout_changed = m != d_.m || max_updates != d_.max_updates;
    if (layout_changed) cap_l_ = cap_u_ = 0;
    if (nl > cap_l_ || cap_l_ == 0) { cap_l_ = nl + nl / 2 + 256; layout_changed = true; }
    if (nu > cap_u_ || cap_u_ == 0) { cap_u_ = nu + nu / 2 + 256; layout_changed = true; }
    const size_t cl = cap_l_, cu = cap_u_;
    // ---- uploaded prefix ----------------------------------------------------------------------------------------------
    Carver c;
    const size_t o_rowpos = c.take<int>(m), o_colpos = c.take<int>(m);
    const size_t o_lrs = c.take<int>(m + 1), o_lcs = c.take<int>(m + 1), o_urs = c.take<int>(m + 1), o_ucs = c.take<int>(m);
    const size_t o_uclen = c.take<int>(m);
    const size_t o_diag = c.take<double>(m);
    const size_t o_lrcol = c.take<int>(cl), o_lcrow = c.take<int>(cl);
    const size_t o_lrval = c.take<double>(cl), o_lcval = c.take<double>(cl);
    const size_t upload_bytes = c.offset;
    // ---- device only (the base parts of the U arrays are uploaded one by one) ----------------------------------------
    // U rows: base entries [0, cap_u) then the append area; U columns: base entries then the arena of replaced columns.
    const size_t app = (size_t)m * max_updates;
    const size_t o_urcol = c.take<int>(cu + app), o_ucrow = c.take<int>(cu + app);
    const size_t o_urval = c.take<double>(cu + app), o_ucval = c.take<double>(cu + app);
    const size_t o_urlen = c.take<int>(m), o_applen = c.take<int>(m), o_eta_pivot = c.take<int>(max_updates + 1);
    const size_t o_rank = c.take<int>(m), o_seq = c.take<int>(m);
    const size_t o_eta_start = c.take<int>(max_updates + 2);
    const size_t o_eta_idx = c.take<int>(app), o_eta_val = c.take<double>(app);
    const size_t o_spike = c.take<double>(m);
    const size_t o_state = c.take<int>(LU_STATE_WORDS);
    const size_t device_bytes = c.offset;
    // staging: prefix + the four compact U arrays
    Carver s;
    s.offset = upload_bytes;
    const size_t s_urcol = s.take<int>(nu), s_ucrow = s.take<int>(nu), s_urval = s.take<double>(nu), s_ucval = s.take<double>(nu);
    {
        char* before = dev_;
        reserve(device_bytes, s.offset);
        if (dev_ != before) layout_changed = true;
    }

    char* h = staging_;
    std::memcpy(h + o_rowpos, f.rowpos.data(), m * sizeof(int));
    std::memcpy(h + o_colpos, f.colpos.data(), m * sizeof(int));
    std::memcpy(h + o_lrs, f.l_start.data(), (m + 1) * sizeof(int));
    std::memcpy(h + o_urs, f.u_start.data(), (m + 1) * sizeof(int));
    if (nl) {
        std::memcpy(h + o_lrcol, f.l_col.data(), nl * sizeof(int));
        std::memcpy(h + o_lrval, f.l_val.data(), nl * sizeof(double));
    }
    std::memcpy(h + o_diag, f.diag.data(), m * sizeof(double));
    // column orientation of L and U (counting transposes)
    {
        int* lcs = reinterpret_cast<int*>(h + o_lcs);
        int* lcrow = reinterpret_cast<int*>(h + o_lcrow);
        double* lcval = reinterpret_cast<double*>(h + o_lcval);
        std::fill(lcs, lcs + m + 1, 0);
        for (size_t e = 0; e < nl; ++e) lcs[f.l_col[e] + 1]++;
        for (int j = 0; j < m; ++j) lcs[j + 1] += lcs[j];
        std::vector<int> fill(lcs, lcs + m);
        for (int i = 0; i < m; ++i)
            for (int e = f.l_start[i]; e < f.l_start[i + 1]; ++e) {
                const int dst = fill[f.l_col[e]]++;
                lcrow[dst] = i;
                lcval[dst] = f.l_val[e];
            }
        int* ucs = reinterpret_cast<int*>(h + o_ucs);
        int* uclen = reinterpret_cast<int*>(h + o_uclen);
        int* ucrow = reinterpret_cast<int*>(h + s_ucrow);
        double* ucval = reinterpret_cast<double*>(h + s_ucval);
        std::vector<int> count(m + 1, 0);
        for (size_t e = 0; e < nu; ++e) count[f.u_col[e] + 1]++;
        for (int j = 0; j < m; ++j) count[j + 1] += count[j];
        for (int j = 0; j < m; ++j) {
            ucs[j] = count[j];
            uclen[j] = count[j + 1] - count[j];
        }
        std::vector<int> fillu(count.begin(), count.end() - 1);
        for (int i = 0; i < m; ++i)
            for (int e = f.u_start[i]; e < f.u_start[i + 1]; ++e) {
                const int dst = fillu[f.u_col[e]]++;
                ucrow[dst] = i;
                ucval[dst] = f.u_val[e];
            }
        if (nu) {
            std::memcpy(h + s_urcol, f.u_col.data(), nu * sizeof(int));
            std::memcpy(h + s_urval, f.u_val.data(), nu * sizeof(double));
        }
    }
    RELP_HIP(hipMemcpyAsync(dev_, h, upload_bytes, hipMemcpyHostToDevice, stream));
    if (nu) {
        RELP_HIP(hipMemcpyAsync(dev_ + o_urcol, h + s_urcol, nu * sizeof(int), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(dev_ + o_ucrow, h + s_ucrow, nu * sizeof(int), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(dev_ + o_urval, h + s_urval, nu * sizeof(double), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(dev_ + o_ucval, h + s_ucval, nu * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    DeviceLU d;
    d.m = m;
    d.max_updates = max_updates;
    auto I = [&](size_t o) { return reinterpret_cast<int*>(dev_ + o); };
    auto D = [&](size_t o) { return reinterpret_cast<double*>(dev_ + o); };
    d.rowpos = I(o_rowpos);
    d.colpos = I(o_colpos);
    d.l_rstart = I(o_lrs); d.l_rcol = I(o_lrcol); d.l_rval = D(o_lrval);
    d.l_cstart = I(o_lcs); d.l_crow = I(o_lcrow); d.l_cval = D(o_lcval);
    d.u_rstart = I(o_urs); d.u_rlen = I(o_urlen); d.u_rcol = I(o_urcol); d.u_rval = D(o_urval);
    d.u_app_len = I(o_applen); d.u_app_first = (int)cu; d.u_app_stride = max_updates;
    d.u_cstart = I(o_ucs); d.u_clen = I(o_uclen); d.u_crow = I(o_ucrow); d.u_cval = D(o_ucval);
    d.u_c_capacity = (int)(cu + app);
    d.diag = D(o_diag);
    d.rank = I(o_rank); d.seq = I(o_seq);
    d.eta_start = I(o_eta_start); d.eta_pivot = I(o_eta_pivot); d.eta_idx = I(o_eta_idx); d.eta_val = D(o_eta_val);
    d.eta_capacity = (int)app;
    d.spike = D(o_spike);
    d.state = I(o_state);
    d_ = d;
    hipLaunchKernelGGL(lu_init_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, d_);
    nnz_l = (long long)nl;
    nnz_u = (long long)nu;
    lu_depths(f, &depth_l, &depth_u);
    return layout_changed;
}

// LDS of the solve kernels: x0, x1 (doubles), flags (ints), one count per 64 rows for the ordered compactions, reductions
static size_t lu_lds_bytes_for(int m) {
    const size_t mm = (size_t)((m + 1) & ~1);
    return 2 * mm * sizeof(double) + mm * sizeof(int) + ((size_t)(m + 63) / 64 + 2) * sizeof(int) + 64 * sizeof(double);
}
size_t LuFactors::lds_bytes(int) const { return lu_lds_bytes_for(d_.m); }
bool lu_fits_lds(int m) { return lu_lds_bytes_for(m) <= 160 * 1024 - 4096; }

// =====================================================================================================
// device: sync-free triangular solves in LDS
// =====================================================================================================
struct TriView {
    const int* start;      // first entry of row (column) i
    const int* len;        // nullptr: start[i + 1] - start[i]
    const int* idx;
    const double* val;
    const int* app_len;    // second segment (U rows): entries at app_first + i * app_stride; nullptr: none
    int app_first, app_stride;
    const double* diag;    // nullptr: unit diagonal
    const int* seq;        // visiting order of the rows (nullptr: identity)
};

// In place:  x[i] <- (x[i] - sum_e val[e] x[idx[e]]) / diag[i].  Thread t owns the rows order(t), order(t + T), ...; a row
// depends only on rows earlier in the visiting order (triangularity), so the thread that owns the first unfinished row can
// always proceed: no deadlock.  A published row has flag[i] == epoch (the value is written before the flag; LDS keeps a
// wave's accesses in order).  Lanes never spin inside a divergent branch: every trip of the loop each lane either consumes
// its next entry or not, so a lane waiting on another lane of its own wave cannot block it.
template <int NRHS, bool REVERSE>
__device__ __forceinline__ void solve_gather(const TriView tv, const int m, volatile double* x0, volatile double* x1,
                                             volatile int* flag, const int epoch) {
    int k = threadIdx.x;
    bool have = k < m;
    int i = 0, e = 0, end = 0, e2 = 0, end2 = 0, c = 0;
    double v = 0.0, a0 = 0.0, a1 = 0.0, dinv = 1.0;
    auto begin_row = [&]() {
        const int r = REVERSE ? m - 1 - k : k;
        i = tv.seq ? tv.seq[r] : r;
        e = tv.start[i];
        end = e + (tv.len ? tv.len[i] : tv.start[i + 1] - e);
        if (tv.app_len) {
            e2 = tv.app_first + i * tv.app_stride;
            end2 = e2 + tv.app_len[i];
        } else {
            e2 = end2 = 0;
        }
        if (e == end) {
            e = e2;
            end = end2;
            e2 = end2;
        }
        dinv = tv.diag ? 1.0 / tv.diag[i] : 1.0;
        a0 = x0[i];
        if (NRHS == 2) a1 = x1[i];
        if (e < end) {
            c = tv.idx[e];
            v = tv.val[e];
        }
    };
    if (have) begin_row();
    while (__any(have)) {
        bool progressed = false;
        if (have) {
            if (e < end) {
                const int f = flag[c];
                const double xc0 = x0[c];
                const double xc1 = NRHS == 2 ? x1[c] : 0.0;
                if (f == epoch) {
                    a0 -= v * xc0;
                    if (NRHS == 2) a1 -= v * xc1;
                    ++e;
                    if (e == end) {
                        e = e2;
                        end = end2;
                        e2 = end2;
                    }
                    if (e < end) {
                        c = tv.idx[e];
                        v = tv.val[e];
                    }
                    progressed = true;
                }
            }
            if (e >= end) {
                x0[i] = a0 * dinv;
                if (NRHS == 2) x1[i] = a1 * dinv;
                flag[i] = epoch;
                k += blockDim.x;
                have = k < m;
                if (have) begin_row();
                progressed = true;
            }
        }
        if (!__any(progressed)) __builtin_amdgcn_s_sleep(1);
    }
}

// ---- eta files --------------------------------------------------------------------------------------------------------
// FTRAN direction (eta_file.rs:72-105): for each update in order  v[t] -= sum_k r_k v[k].  One wave; a dot product per eta.
__device__ __forceinline__ void apply_etas_forward(const DeviceLU& lu, const int n_updates, volatile double* x0) {
    if (threadIdx.x >= WAVE) return;
    const int lane = threadIdx.x;
    int s = 0;
    int nidx = 0;
    double nval = 0.0;
    int s_end = n_updates > 0 ? lu.eta_start[1] : 0;
    if (n_updates > 0 && s + lane < s_end) {
        nidx = lu.eta_idx[s + lane];
        nval = lu.eta_val[s + lane];
    }
    for (int k = 0; k < n_updates; ++k) {
        const int e_end = s_end;
        const int t = lu.eta_pivot[k];
        const int cidx = nidx;
        const double cval = nval;
        const bool chave = s + lane < e_end;
        // prefetch the first chunk of the next eta while this one is reduced
        const int s_next = e_end;
        s_end = k + 1 < n_updates ? lu.eta_start[k + 2] : e_end;
        if (k + 1 < n_updates && s_next + lane < s_end) {
            nidx = lu.eta_idx[s_next + lane];
            nval = lu.eta_val[s_next + lane];
        }
        double partial = chave ? cval * x0[cidx] : 0.0;
        for (int e = s + lane + WAVE; e < e_end; e += WAVE) partial += lu.eta_val[e] * x0[lu.eta_idx[e]];
        const double total = wave_sum(partial);
        if (lane == LAST && total != 0.0) x0[t] = x0[t] - total;
        s = s_next;
    }
}
// BTRAN direction (eta_file.rs:49-65): for each update in reverse  v[j] -= r_j v[t].
template <int NRHS>
__device__ __forceinline__ void apply_etas_backward(const DeviceLU& lu, const int n_updates, volatile double* x0, volatile double* x1) {
    if (threadIdx.x >= WAVE) return;
    const int lane = threadIdx.x;
    for (int k = n_updates - 1; k >= 0; --k) {
        const int s = lu.eta_start[k], e_end = lu.eta_start[k + 1];
        const int t = lu.eta_pivot[k];
        const double v0 = x0[t];
        const double v1 = NRHS == 2 ? x1[t] : 0.0;
        if (v0 == 0.0 && v1 == 0.0) continue;
        for (int e = s + lane; e < e_end; e += WAVE) {
            const int j = lu.eta_idx[e];
            const double r = lu.eta_val[e];
            x0[j] = x0[j] - r * v0;
            if (NRHS == 2) x1[j] = x1[j] - r * v1;
        }
    }
}

// LDS carve-up shared by every kernel of this file
struct LuShared {
    volatile double* x0;
    volatile double* x1;
    volatile int* flag;
    int* group_count;  // one slot per 64 rows (+2)
    double* red;       // 64 doubles
    unsigned long long* dbg;  // diagnostic builds (-DRELP_STAMPS): per-segment cycle sums; nullptr otherwise
    unsigned long long* t_prev;
};
__device__ __forceinline__ void lu_stamp(const LuShared& sh, int k) {
#ifdef RELP_STAMPS
    if (sh.dbg && threadIdx.x == 0) {
        const unsigned long long t = clock64();
        sh.dbg[k] += t - *sh.t_prev;
        *sh.t_prev = t;
    }
#endif
}
__device__ __forceinline__ LuShared lu_shared(char* smem, int m) {
    const int mm = (m + 1) & ~1;
    LuShared s;
    s.x0 = reinterpret_cast<volatile double*>(smem);
    s.x1 = s.x0 + mm;
    s.red = const_cast<double*>(s.x1 + mm);
    s.flag = reinterpret_cast<volatile int*>(s.red + 64);
    s.group_count = const_cast<int*>(s.flag + mm);
    s.dbg = nullptr;
    s.t_prev = nullptr;
    return s;
}

// FTRAN on the vector in sh.x0 (position space, P already applied): L solve, etas, [spike], U solve.  Ends with a barrier.
__device__ __forceinline__ void lu_ftran_block(const DeviceLU& lu, const LuShared& sh, const int n_updates, int& epoch,
                                               double* spike_out) {
    const int m = lu.m;
    TriView L{lu.l_rstart, nullptr, lu.l_rcol, lu.l_rval, nullptr, 0, 0, nullptr, nullptr};
    solve_gather<1, false>(L, m, sh.x0, sh.x1, sh.flag, ++epoch);
    __syncthreads();
    lu_stamp(sh, 2);
    if (n_updates > 0) {
        apply_etas_forward(lu, n_updates, sh.x0);
        __syncthreads();
    }
    lu_stamp(sh, 3);
    if (spike_out)
        for (int i = threadIdx.x; i < m; i += blockDim.x) spike_out[i] = sh.x0[i];
    TriView U{lu.u_rstart, lu.u_rlen, lu.u_rcol, lu.u_rval, lu.u_app_len, lu.u_app_first, lu.u_app_stride, lu.diag, lu.seq};
    solve_gather<1, true>(U, m, sh.x0, sh.x1, sh.flag, ++epoch);
    __syncthreads();
    lu_stamp(sh, 4);
}

// BTRAN on the vectors in sh.x0 (and sh.x1), position space with Q applied.  `after_upper` runs between the U solve and the
// etas, when x0 = (e_t' U^-1) for a unit input (the Forrest-Tomlin row comes from there).  Ends with a barrier.
template <int NRHS, class AfterUpper>
__device__ __forceinline__ void lu_btran_block(const DeviceLU& lu, const LuShared& sh, const int n_updates, int& epoch,
                                               AfterUpper after_upper) {
    const int m = lu.m;
    TriView U{lu.u_cstart, lu.u_clen, lu.u_crow, lu.u_cval, nullptr, 0, 0, lu.diag, lu.seq};
    solve_gather<NRHS, false>(U, m, sh.x0, sh.x1, sh.flag, ++epoch);
    __syncthreads();
    lu_stamp(sh, 7);
    after_upper();
    lu_stamp(sh, 8);
    if (n_updates > 0) {
        apply_etas_backward<NRHS>(lu, n_updates, sh.x0, sh.x1);
        __syncthreads();
    }
    lu_stamp(sh, 9);
    TriView L{lu.l_cstart, nullptr, lu.l_crow, lu.l_cval, nullptr, 0, 0, nullptr, nullptr};
    solve_gather<NRHS, true>(L, m, sh.x0, sh.x1, sh.flag, ++epoch);
    __syncthreads();
    lu_stamp(sh, 10);
}

// Ordered compaction: every row i with keep(i) gets the number of kept rows before it.  Returns the total.  Two barriers.
template <class Keep, class Emit>
__device__ __forceinline__ int ordered_compact(const int m, int* group_count, Keep keep, Emit emit) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int groups = (m + WAVE - 1) / WAVE;
    __syncthreads();
    for (int base = threadIdx.x - lane; base < m; base += blockDim.x) {
        const int i = base + lane;
        const unsigned long long mask = __ballot(i < m && keep(i));
        if (lane == 0) group_count[base / WAVE] = __popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int running = 0;
        for (int g = 0; g < groups; ++g) {
            const int cnt = group_count[g];
            group_count[g] = running;
            running += cnt;
        }
        group_count[groups] = running;
    }
    __syncthreads();
    for (int base = threadIdx.x - lane; base < m; base += blockDim.x) {
        const int i = base + lane;
        const bool kept = i < m && keep(i);
        const unsigned long long mask = __ballot(kept);
        if (kept) emit(i, group_count[base / WAVE] + __popcll(mask & ((1ull << lane) - 1ull)));
    }
    return group_count[groups];
}

// The row eta of a Forrest-Tomlin update from y = e_t' U^-1 (in sh.x0, position space): r_j = -y_j / y_t for the positions
// logically behind t (every other non-zero of y), i.e. r = u_bar U^-1 of mod.rs:112-125 without a second solve (the classic
// identity: row t of U^-1 is (1/u_tt)(e_t' - r) behind the diagonal).  Also the new diagonal element
// spike_t - sum_k r_k spike_k (eta_file.rs:112-134).  Writes the eta behind the existing ones; returns its length.
__device__ __forceinline__ int lu_build_eta(const DeviceLU& lu, const LuShared& sh, const int t, const double* spike,
                                            double* new_diag) {
    const int m = lu.m;
    const int eta_top = lu.state[LU_ETA_TOP];
    const double yt = sh.x0[t];
    double dot = 0.0;
    const int count = ordered_compact(
        m, sh.group_count, [&](int i) { return i != t && sh.x0[i] != 0.0; },
        [&](int i, int slot) {
            const double r = -sh.x0[i] / yt;
            lu.eta_idx[eta_top + slot] = i;
            lu.eta_val[eta_top + slot] = r;
            dot += r * spike[i];
        });
    const double total = block_reduce<0>(dot, sh.red);
    *new_diag = spike[t] - total;
    return count;
}

// Structural part of the Forrest-Tomlin update (mod.rs:127-176): row t leaves U, column t becomes the spike, position t
// moves to the end of the logical order.  `eta_count` entries were already written by lu_build_eta.
__device__ __forceinline__ void lu_ft_update_block(const DeviceLU& lu, const LuShared& sh, const int t, const int eta_count,
                                                   const double new_diag, const double* spike) {
    const int m = lu.m;
    const int tid = threadIdx.x, T = blockDim.x;
    const int n_updates = lu.state[LU_N_UPDATES];
    const int top = lu.state[LU_UC_TOP];
    const int eta_top = lu.state[LU_ETA_TOP];
    const int rank_t = lu.rank[t];
    __syncthreads();  // everyone has read the state words
    // 1. row t leaves the column orientation
    {
        const int rs = lu.u_rstart[t], rl = lu.u_rlen[t];
        const int as = lu.u_app_first + t * lu.u_app_stride, al = lu.u_app_len[t];
        for (int e = tid; e < rl + al; e += T) {
            const int j = lu.u_rcol[e < rl ? rs + e : as + (e - rl)];
            const int cs = lu.u_cstart[j], cl = lu.u_clen[j];
            for (int s = 0; s < cl; ++s)
                if (lu.u_crow[cs + s] == t) {
                    lu.u_crow[cs + s] = lu.u_crow[cs + cl - 1];
                    lu.u_cval[cs + s] = lu.u_cval[cs + cl - 1];
                    lu.u_clen[j] = cl - 1;
                    break;
                }
        }
    }
    // 2. the old column t leaves the row orientation
    {
        const int cs = lu.u_cstart[t], cl = lu.u_clen[t];
        for (int e = tid; e < cl; e += T) {
            const int i = lu.u_crow[cs + e];
            const int rs = lu.u_rstart[i], rl = lu.u_rlen[i];
            bool found = false;
            for (int s = 0; s < rl; ++s)
                if (lu.u_rcol[rs + s] == t) {
                    lu.u_rcol[rs + s] = lu.u_rcol[rs + rl - 1];
                    lu.u_rval[rs + s] = lu.u_rval[rs + rl - 1];
                    lu.u_rlen[i] = rl - 1;
                    found = true;
                    break;
                }
            if (!found) {
                const int as = lu.u_app_first + i * lu.u_app_stride, al = lu.u_app_len[i];
                for (int s = 0; s < al; ++s)
                    if (lu.u_rcol[as + s] == t) {
                        lu.u_rcol[as + s] = lu.u_rcol[as + al - 1];
                        lu.u_rval[as + s] = lu.u_rval[as + al - 1];
                        lu.u_app_len[i] = al - 1;
                        break;
                    }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        lu.u_rlen[t] = 0;
        lu.u_app_len[t] = 0;
    }
    // 3. the spike becomes column t: arena of the column orientation, one appended entry per row
    const int count = ordered_compact(
        m, sh.group_count, [&](int i) { return i != t && spike[i] != 0.0; },
        [&](int i, int slot) {
            const double v = spike[i];
            lu.u_crow[top + slot] = i;
            lu.u_cval[top + slot] = v;
            const int a = lu.u_app_first + i * lu.u_app_stride + lu.u_app_len[i];
            lu.u_rcol[a] = t;
            lu.u_rval[a] = v;
            lu.u_app_len[i] += 1;
        });
    // 4. logical order: t goes to the back (RotateToBack, permutation/rotate_to_back.rs:15-122)
    for (int x = tid; x < m; x += T) {
        const int r = lu.rank[x];
        const int nr = x == t ? m - 1 : (r > rank_t ? r - 1 : r);
        lu.rank[x] = nr;
        lu.seq[nr] = x;
    }
    if (tid == 0) {
        lu.u_cstart[t] = top;
        lu.u_clen[t] = count;
        lu.diag[t] = new_diag;
        lu.eta_pivot[n_updates] = t;
        lu.eta_start[n_updates + 1] = eta_top + eta_count;
        lu.state[LU_N_UPDATES] = n_updates + 1;
        lu.state[LU_UC_TOP] = top + count;
        lu.state[LU_ETA_TOP] = eta_top + eta_count;
        if (!(fabs(new_diag) > 0.0) || new_diag != new_diag) lu.state[LU_FLAGS] |= LU_FLAG_UNSTABLE;
    }
    __syncthreads();
}

// =====================================================================================================
// stand-alone kernels (fine-grained `BasisInverse` operations)
// =====================================================================================================
__device__ __forceinline__ void lu_clear(const LuShared& sh, int m, bool two) {
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        sh.x0[i] = 0.0;
        if (two) sh.x1[i] = 0.0;
        sh.flag[i] = 0;
    }
    __syncthreads();
}

// dense != nullptr: dense right-hand side in original row order; else the sparse (rows, vals)
__global__ void __launch_bounds__(LU_THREADS) lu_ftran_kernel(DeviceLU lu, const int* rows, const double* vals, int nnz,
                                                               const double* dense, double* out, int keep_spike) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuShared sh = lu_shared(smem, m);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear(sh, m, false);
    if (dense) {
        for (int i = threadIdx.x; i < m; i += blockDim.x) sh.x0[lu.rowpos[i]] = dense[i];
    } else {
        for (int e = threadIdx.x; e < nnz; e += blockDim.x) sh.x0[lu.rowpos[rows[e]]] = vals[e];
    }
    __syncthreads();
    int epoch = 0;
    lu_ftran_block(lu, sh, n_updates, epoch, keep_spike ? lu.spike : nullptr);
    for (int s = threadIdx.x; s < m; s += blockDim.x) out[s] = sh.x0[lu.colpos[s]];
}

__global__ void __launch_bounds__(LU_THREADS) lu_btran_kernel(DeviceLU lu, const int* slots, const double* vals, int nnz,
                                                               const double* dense, double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuShared sh = lu_shared(smem, m);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear(sh, m, false);
    if (dense) {
        for (int s = threadIdx.x; s < m; s += blockDim.x) sh.x0[lu.colpos[s]] = dense[s];
    } else {
        for (int e = threadIdx.x; e < nnz; e += blockDim.x) sh.x0[lu.colpos[slots[e]]] = vals[e];
    }
    __syncthreads();
    int epoch = 0;
    lu_btran_block<1>(lu, sh, n_updates, epoch, [] {});
    for (int i = threadIdx.x; i < m; i += blockDim.x) out[i] = sh.x0[lu.rowpos[i]];
}

__global__ void __launch_bounds__(LU_THREADS) lu_update_kernel(DeviceLU lu, int p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuShared sh = lu_shared(smem, m);
    const int t = lu.colpos[p];
    if (lu.state[LU_N_UPDATES] >= lu.max_updates) {  // no room for another eta: the caller has to refactor
        if (threadIdx.x == 0) lu.state[LU_FLAGS] |= LU_FLAG_OVERFLOW;
        return;
    }
    lu_clear(sh, m, false);
    if (threadIdx.x == 0) sh.x0[t] = 1.0;
    __syncthreads();
    // y = e_t' U^-1: the U stage of a BTRAN (mod.rs:373-397), then the eta from it
    TriView U{lu.u_cstart, lu.u_clen, lu.u_crow, lu.u_cval, nullptr, 0, 0, lu.diag, lu.seq};
    solve_gather<1, false>(U, m, sh.x0, sh.x1, sh.flag, 1);
    __syncthreads();
    double new_diag = 0.0;
    const int eta_count = lu_build_eta(lu, sh, t, lu.spike, &new_diag);
    lu_ft_update_block(lu, sh, t, eta_count, new_diag, lu.spike);
}

static bool g_lu_lds_configured = false;
static void allow_full_lds(const void* kernel);
static void configure_lu_lds() {
    if (g_lu_lds_configured) return;
    allow_full_lds(reinterpret_cast<const void*>(&lu_ftran_kernel));
    allow_full_lds(reinterpret_cast<const void*>(&lu_btran_kernel));
    allow_full_lds(reinterpret_cast<const void*>(&lu_update_kernel));
    g_lu_lds_configured = true;
}
static void check_launch(const char* what) {
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) throw DeviceError(std::string(what) + ": " + hipGetErrorString(err));
}

void launch_lu_ftran(const DeviceLU& lu, const int* rows, const double* vals, int nnz, double* out, int keep_spike, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_ftran_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, lu, rows, vals, nnz, (const double*)nullptr, out, keep_spike);
    check_launch("lu_ftran_kernel");
}
void launch_lu_ftran_dense(const DeviceLU& lu, const double* rhs, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_ftran_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, lu, (const int*)nullptr, (const double*)nullptr, 0, rhs, out, 0);
    check_launch("lu_ftran_kernel");
}
void launch_lu_btran(const DeviceLU& lu, const int* slots, const double* vals, int nnz, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_btran_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, lu, slots, vals, nnz, (const double*)nullptr, out);
    check_launch("lu_btran_kernel");
}
void launch_lu_btran_dense(const DeviceLU& lu, const double* in_slots, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_btran_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, lu, (const int*)nullptr, (const double*)nullptr, 0, in_slots, out);
    check_launch("lu_btran_kernel");
}
void launch_lu_update(const DeviceLU& lu, int p, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_update_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, lu, p);
    check_launch("lu_update_kernel");
}

// =====================================================================================================
// The LU carry inside the device-resident simplex loop (relp_options.carry = RELP_CARRY_LU)
// =====================================================================================================
// ONE single-workgroup kernel per pivot does everything that is not the pricing pass:
//   entering column (reduction of the pricing workgroups' candidates)      pivot_rule.rs:221-241
//   FTRAN alpha_q = B^-1 a_q, spike kept                                    tableau/mod.rs:126-130 -> lower_upper/mod.rs:180-210
//   ratio test (Harris two-pass; ties: Bland, lowest leaving column)       tableau/mod.rs:287-313
//   update_b                                                               carry/mod.rs:295-325
//   ONE two-right-hand-side BTRAN: w = alpha_q' B^-1 (carry/mod.rs:575) and e_p' B^-1, from which
//       rho_p of the NEW basis = (e_p' B_old^-1) / alpha_pq (lower_upper/mod.rs:254-272), the Forrest-Tomlin row eta
//       (its U stage is e_t' U^-1; lower_upper/mod.rs:112-125 computes it with a third solve) and
//       update_minus_pi_and_obj                                            carry/mod.rs:338-349
//   Forrest-Tomlin update of U                                             lower_upper/mod.rs:94-178
//   or, after `refactor_period` updates, status = ST_REFACTOR: the host factorises the new basis (carry/mod.rs:584-591:
//   polled before the update; the new basis is inverted from scratch and no update is made).
// The explicit-inverse pipeline needs two kernels for this (K2, K3) and rewrites an m x m matrix per pivot.
template <int RULE>
__global__ void __launch_bounds__(LU_THREADS) lu_pivot_kernel(DeviceLP lp, DeviceLU lu, int n_price_blocks, double tol_pivot,
                                                               double harris_delta, int skip_artificial_rows, int mode,
                                                               int refactor_period) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_akey[LU_THREADS / WAVE];
    __shared__ unsigned long long s_arank[LU_THREADS / WAVE];
    __shared__ double s_bcast[4];
    Ctl* ctl = lp.ctl;
    const int tid = threadIdx.x, T = blockDim.x;
    const int m = lp.m;
    LuShared sh = lu_shared(smem, m);
#ifdef RELP_STAMPS
    __shared__ unsigned long long s_tprev;
    if (tid == 0) {
        s_tprev = clock64();
        lp.dbg[63] += 1;
    }
    sh.dbg = lp.dbg;
    sh.t_prev = &s_tprev;
#endif
    // ---- round trip 1: control word, update count, the candidates ---------------------------------------------------
    const int status = ctl->status;
    const long long iters = ctl->iters;
    const long long budget = ctl->budget;
    const int forced_q = ctl->forced_q;
    const int forced_p = ctl->forced_p;
    const double minus_obj = ctl->minus_obj;
    const int n_updates = lu.state[LU_N_UPDATES];
    double ckey = 0.0;
    unsigned long long crank = RANK_NONE;
    for (int b = tid; b < n_price_blocks; b += T) {
        const int j = lp.cand_j[b];
        const double k = lp.cand_key[b];
        if (j >= 0) {
            const unsigned long long order = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? (unsigned long long)(0x7fffffff - j) : (unsigned long long)j;
            const unsigned long long r = (order << 16) | (unsigned long long)b;
            if (crank == RANK_NONE || k > ckey || (k == ckey && r < crank)) {
                ckey = k;
                crank = r;
            }
        }
    }
    if (status != ST_RUNNING) return;
    if (mode == 0 && iters >= budget) {
        if (tid == 0) {
            ctl->status = ST_BUDGET;
            ctl->pending = 0;
        }
        return;
    }
    // ---- entering column ----------------------------------------------------------------------------------------------
    int q;
    double cbar_q;
    if (forced_q < 0) {
        block_argbest(ckey, crank, s_akey, s_arank);
        if (crank == RANK_NONE) {
            q = -1;
            cbar_q = 0.0;
        } else {
            const int order = (int)(crank >> 16);
            q = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? 0x7fffffff - order : order;
            cbar_q = lp.cand_cbar[(int)(crank & 0xffff)];
        }
    } else {
        q = forced_q;
        if (tid == 0) {
            double cb = lp.cost[forced_q];
            for (int e = lp.col_start[forced_q]; e < lp.col_start[forced_q + 1]; ++e) cb += lp.value[e] * lp.minus_pi[lp.row_index[e]];
            s_bcast[0] = cb;
        }
        __syncthreads();
        cbar_q = s_bcast[0];
        __syncthreads();
    }
    if (q < 0) {
        if (tid == 0) {
            if (mode == 0) ctl->status = ST_NO_ENTERING;
            ctl->q = -1;
            ctl->pending = 0;
            if (mode == 0) ctl->last_selected = -1;
        }
        return;
    }
    if (mode == 1) {
        if (tid == 0) {
            ctl->q = q;
            ctl->cbar_q = cbar_q;
            ctl->pending = 0;
        }
        return;
    }
    // ---- FTRAN ------------------------------------------------------------------------------------------------------------
    lu_stamp(sh, 0);
    lu_clear(sh, m, false);
    for (int e = lp.col_start[q] + tid; e < lp.col_start[q + 1]; e += T) sh.x0[lu.rowpos[lp.row_index[e]]] = lp.value[e];
    __syncthreads();
    lu_stamp(sh, 1);
    int epoch = 0;
    lu_ftran_block(lu, sh, n_updates, epoch, lu.spike);
    // ---- alpha per basis slot (kept in x1), gamma_q, Harris pass 1 ----------------------------------------------------------
    double sumsq = 0.0, theta = INFINITY;
    for (int s = tid; s < m; s += T) {
        const double a = sh.x0[lu.colpos[s]];
        sh.x1[s] = a;
        lp.alpha[s] = a;
        sumsq += a * a;
        if (a > tol_pivot && !(skip_artificial_rows && lp.basis[s] < lp.n_art)) theta = fmin(theta, (fmax(lp.xB[s], 0.0) + harris_delta) / a);
    }
    const double gamma_q = 1.0 + block_reduce<0>(sumsq, sh.red);  // pivot_rule.rs:258
    const double theta_max = block_reduce<1>(theta, sh.red + 32);
    // ---- Harris pass 2: the largest eligible pivot, ties by the lowest leaving column (Bland, tableau/mod.rs:295) -------------
    int p = forced_p;
    if (forced_p < 0) {
        double hkey = 0.0;
        unsigned long long hrank = RANK_NONE;
        for (int s = tid; s < m; s += T) {
            const double a = sh.x1[s];
            if (!(a > tol_pivot)) continue;
            const int bs = lp.basis[s];
            if (skip_artificial_rows && bs < lp.n_art) continue;
            if (fmax(lp.xB[s], 0.0) / a <= theta_max) {
                const unsigned long long rk = ((unsigned long long)(unsigned)bs << 32) | (unsigned)s;
                if (hrank == RANK_NONE || a > hkey || (a == hkey && rk < hrank)) {
                    hkey = a;
                    hrank = rk;
                }
            }
        }
        block_argbest(hkey, hrank, s_akey, s_arank);
        p = hrank == RANK_NONE ? -1 : (int)(hrank & 0xffffffffu);
    }
    if (p < 0) {
        if (tid == 0) {
            if (mode == 0) ctl->status = ST_UNBOUNDED;
            ctl->q = q;
            ctl->p = -1;
            ctl->pending = 0;
            ctl->forced_q = -1;
            ctl->forced_p = -1;
        }
        return;
    }
    lu_stamp(sh, 5);
    const double alpha_pq = sh.x1[p];
    if (mode == 2) {
        if (tid == 0) {
            ctl->q = q;
            ctl->p = p;
            ctl->cbar_q = cbar_q;
            ctl->gamma_q = gamma_q;
            ctl->pending = 0;
            ctl->forced_q = -1;
            ctl->forced_p = -1;
        }
        return;
    }
    if (alpha_pq == 0.0) return;  // (a forced pivot on a zero element: the host sees that nothing happened)
    const int leaving = lp.basis[p];
    const double xp = fmax(lp.xB[p], 0.0) / alpha_pq;
    __syncthreads();  // every thread has read basis[p] / xB[p] before they change
    // ---- x_B update (carry/mod.rs:295-325) ------------------------------------------------------------------------------
    for (int s = tid; s < m; s += T) lp.xB[s] = (s == p) ? xp : lp.xB[s] - sh.x1[s] * xp;
    // ---- BTRAN with two right-hand sides: x0 <- e_p, x1 <- alpha (both per basis slot -> position space) -----------------------
    const int t = lu.colpos[p];
    const bool do_update = n_updates < refactor_period && n_updates < lu.max_updates;
    __syncthreads();
    for (int s = tid; s < m; s += T) {
        sh.x0[s] = 0.0;
        sh.x1[lu.colpos[s]] = lp.alpha[s];  // (this thread wrote lp.alpha[s] itself)
    }
    __syncthreads();
    if (tid == 0) sh.x0[t] = 1.0;
    __syncthreads();
    lu_stamp(sh, 6);
    const double diag_t = lu.diag[t];
    int eta_count = 0;
    double new_diag = 0.0;
    lu_btran_block<2>(lu, sh, n_updates, epoch, [&] {
        if (do_update) eta_count = lu_build_eta(lu, sh, t, lu.spike, &new_diag);
    });
    // ---- rho_p of the new basis, w, -pi (carry/mod.rs:338-349) ----------------------------------------------------------------
    for (int i = tid; i < m; i += T) {
        const int k = lu.rowpos[i];
        const double r = sh.x0[k] / alpha_pq;
        const double w = sh.x1[k];
        const double pi_new = lp.minus_pi[i] - cbar_q * r;
        lp.rho[i] = r;
        lp.w[i] = w;
        lp.minus_pi[i] = pi_new;
        if (lp.prw) {
            lp.prw[(size_t)4 * i] = pi_new;
            lp.prw[(size_t)4 * i + 1] = r;
            lp.prw[(size_t)4 * i + 2] = w;
        }
    }
    lu_stamp(sh, 11);
    // ---- Forrest-Tomlin update, or hand the new basis to the host ---------------------------------------------------------------
    bool refactor = !do_update;
    if (do_update) {
        lu_ft_update_block(lu, sh, t, eta_count, new_diag, lu.spike);
        // det(B_new) = alpha_pq det(B_old)  =>  the new diagonal element must equal alpha_pq * u_tt: a free accuracy check
        const double expect = alpha_pq * diag_t;
        if (!(fabs(new_diag - expect) <= 1e-7 * (fabs(new_diag) + fabs(expect)))) refactor = true;
    }
    if (tid == 0) {
        lp.basis[p] = q;
        lp.pos[q] = p;
        lp.pos[leaving] = -1;
        ctl->q = q;
        ctl->p = p;
        ctl->leaving = leaving;
        ctl->cbar_q = cbar_q;
        ctl->alpha_pq = alpha_pq;
        ctl->gamma_q = gamma_q;
        ctl->xp = xp;
        ctl->nz_count = 0;
        ctl->minus_obj = minus_obj - cbar_q * xp;
        ctl->iters = iters + 1;
        ctl->pending = 1;
        ctl->forced_q = -1;
        ctl->forced_p = -1;
        ctl->last_selected = q;
        if (refactor) ctl->status = ST_REFACTOR;
    }
    lu_stamp(sh, 12);
}

// x_B = B^-1 b  (InverseMaintainer::from_basis, carry/mod.rs:452-463) through the resident factors
__global__ void __launch_bounds__(LU_THREADS) lu_xb_kernel(DeviceLP lp, DeviceLU lu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuShared sh = lu_shared(smem, m);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear(sh, m, false);
    for (int i = threadIdx.x; i < m; i += blockDim.x) sh.x0[lu.rowpos[i]] = lp.rhs[i];
    __syncthreads();
    int epoch = 0;
    lu_ftran_block(lu, sh, n_updates, epoch, nullptr);
    for (int s = threadIdx.x; s < m; s += blockDim.x) lp.xB[s] = sh.x0[lu.colpos[s]];
}
// -pi = -c_B' B^-1 and -obj = -c_B' x_B  (carry/mod.rs:226-283: the reference forms all of B^-1 with m FTRANs)
__global__ void __launch_bounds__(LU_THREADS) lu_pi_kernel(DeviceLP lp, DeviceLU lu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuShared sh = lu_shared(smem, m);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear(sh, m, false);
    double obj = 0.0;
    for (int s = threadIdx.x; s < m; s += blockDim.x) {
        const double c = lp.cost[lp.basis[s]];
        sh.x0[lu.colpos[s]] = c;
        obj += c * lp.xB[s];
    }
    __syncthreads();
    int epoch = 0;
    lu_btran_block<1>(lu, sh, n_updates, epoch, [] {});
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double v = -sh.x0[lu.rowpos[i]];
        lp.minus_pi[i] = v;
        if (lp.prw) lp.prw[(size_t)4 * i] = v;
    }
    const double total = block_reduce<0>(obj, sh.red);
    if (threadIdx.x == 0) lp.ctl->minus_obj = -total;
}
// gamma_j = 1 + |B^-1 a_j|^2 for every non-basic provider column (pivot_rule.rs:202-219, 299-305): one FTRAN per column,
// a workgroup takes every gridDim.x-th column.  (Warm starts only: between the phases the weights are carried over.)
__global__ void __launch_bounds__(LU_THREADS) lu_gamma_kernel(DeviceLP lp, DeviceLU lu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuShared sh = lu_shared(smem, m);
    const int n_updates = lu.state[LU_N_UPDATES];
    for (int j = lp.n_art + blockIdx.x; j < lp.n; j += gridDim.x) {
        if (lp.pos[j] >= 0) continue;
        __syncthreads();
        lu_clear(sh, m, false);
        for (int e = lp.col_start[j] + threadIdx.x; e < lp.col_start[j + 1]; e += blockDim.x) sh.x0[lu.rowpos[lp.row_index[e]]] = lp.value[e];
        __syncthreads();
        int epoch = 0;
        lu_ftran_block(lu, sh, n_updates, epoch, nullptr);
        double sumsq = 0.0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const double a = sh.x0[i];
            sumsq += a * a;
        }
        const double total = block_reduce<0>(sumsq, sh.red);
        if (threadIdx.x == 0) lp.gamma[j] = 1.0 + total;
    }
}
// Zero-level pivots (phase_one.rs:232-278): first non-basic provider column with a non-zero in tableau row r, given
// row r of the inverse (`rowvec` = e_r' B^-1, one BTRAN) instead of one FTRAN per candidate (lower_upper/mod.rs:239-247).
__global__ void __launch_bounds__(256) lu_row_scan_kernel(DeviceLP lp, const double* rowvec, double tol) {
    for (int j = lp.n_art + blockIdx.x * blockDim.x + threadIdx.x; j < lp.n; j += gridDim.x * blockDim.x) {
        if (lp.pos[j] >= 0) continue;
        double acc = 0.0;
        for (int e = lp.col_start[j]; e < lp.col_start[j + 1]; ++e) acc += lp.value[e] * rowvec[lp.row_index[e]];
        if (fabs(acc) > tol) atomicMin(&lp.ctl->scan_column, j);
    }
}

static bool g_lu_pivot_configured = false;
static void allow_full_lds(const void* kernel) {
    // dynamic + static LDS may reach the 160 KB of a CU; the attribute is process-wide, so it is set once to the maximum
    hipFuncAttributes attr{};
    size_t fixed = 0;
    if (hipFuncGetAttributes(&attr, kernel) == hipSuccess) fixed = attr.sharedSizeBytes;
    const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - fixed));
    if (err != hipSuccess) {
        (void)hipGetLastError();  // not sticky: the launch itself reports a request that is too large
    }
}
static void configure_lu_pivot_lds() {
    if (g_lu_pivot_configured) return;
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_STEEPEST_EDGE>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_DANTZIG>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_FIRST_PROFITABLE>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_FIRST_PROFITABLE_MEMORY>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_xb_kernel));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pi_kernel));
    allow_full_lds(reinterpret_cast<const void*>(&lu_gamma_kernel));
    g_lu_pivot_configured = true;
}
template <int RULE>
static void launch_lu_pivot_rule(const DeviceLP& d, const DeviceLU& lu, int n_price_blocks, double tol_pivot, double harris_delta,
                                 int skip_art, int mode, int refactor_period, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    if (start)
        hipExtLaunchKernelGGL((lu_pivot_kernel<RULE>), dim3(1), dim3(LU_THREADS), (std::uint32_t)lu_lds_bytes_for(lu.m), s, start, stop, 0,
                              d, lu, n_price_blocks, tol_pivot, harris_delta, skip_art, mode, refactor_period);
    else
        hipLaunchKernelGGL((lu_pivot_kernel<RULE>), dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, d, lu, n_price_blocks, tol_pivot,
                           harris_delta, skip_art, mode, refactor_period);
}
// `capturing`: inside a stream capture hipGetLastError must not be polled per launch (the capture's end reports failures)
void launch_lu_pivot(const DeviceLP& d, const DeviceLU& lu, int rule, int n_price_blocks, double tol_pivot, double harris_delta,
                     int skip_art, int mode, int refactor_period, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    configure_lu_pivot_lds();
    switch (rule) {
        case RELP_PIVOT_DANTZIG: launch_lu_pivot_rule<RELP_PIVOT_DANTZIG>(d, lu, n_price_blocks, tol_pivot, harris_delta, skip_art, mode, refactor_period, s, start, stop); break;
        case RELP_PIVOT_FIRST_PROFITABLE: launch_lu_pivot_rule<RELP_PIVOT_FIRST_PROFITABLE>(d, lu, n_price_blocks, tol_pivot, harris_delta, skip_art, mode, refactor_period, s, start, stop); break;
        case RELP_PIVOT_FIRST_PROFITABLE_MEMORY: launch_lu_pivot_rule<RELP_PIVOT_FIRST_PROFITABLE_MEMORY>(d, lu, n_price_blocks, tol_pivot, harris_delta, skip_art, mode, refactor_period, s, start, stop); break;
        default: launch_lu_pivot_rule<RELP_PIVOT_STEEPEST_EDGE>(d, lu, n_price_blocks, tol_pivot, harris_delta, skip_art, mode, refactor_period, s, start, stop); break;
    }
}
void launch_lu_xb(const DeviceLP& d, const DeviceLU& lu, hipStream_t s) {
    configure_lu_pivot_lds();
    hipLaunchKernelGGL(lu_xb_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, d, lu);
    check_launch("lu_xb_kernel");
}
void launch_lu_pi(const DeviceLP& d, const DeviceLU& lu, hipStream_t s) {
    configure_lu_pivot_lds();
    hipLaunchKernelGGL(lu_pi_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, d, lu);
    check_launch("lu_pi_kernel");
}
void launch_lu_gamma(const DeviceLP& d, const DeviceLU& lu, hipStream_t s) {
    configure_lu_pivot_lds();
    const int blocks = std::max(1, std::min(512, d.n - d.n_art));
    hipLaunchKernelGGL(lu_gamma_kernel, dim3(blocks), dim3(LU_THREADS), lu_lds_bytes_for(lu.m), s, d, lu);
    check_launch("lu_gamma_kernel");
}
void launch_lu_row_scan(const DeviceLP& d, const double* rowvec, double tol, hipStream_t s) {
    int blocks = (d.n - d.n_art + 255) / 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(lu_row_scan_kernel, dim3(blocks), dim3(256), 0, s, d, rowvec, tol);
    check_launch("lu_row_scan_kernel");
}

// =====================================================================================================
// LuBasis: the stand-alone `BasisInverse` object
// =====================================================================================================
LuBasis::LuBasis(int device, int m, const LuOptions& options, int refactor_period)
    : device_(device), m_(m), period_(refactor_period > 0 ? refactor_period : 31), options_(options) {
    if (m < 1) throw std::invalid_argument("m < 1");
    if (!lu_fits_lds(m)) throw std::invalid_argument("m too large for the LDS-resident LU solve");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw DeviceError("no HIP device available (relp_amd has no CPU fallback)");
    if (device < 0 || device >= count) throw DeviceError("device ordinal out of range");
    RELP_HIP(hipSetDevice(device_));
    RELP_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    RELP_HIP(hipMalloc(reinterpret_cast<void**>(&d_idx_), (size_t)m * sizeof(int)));
    RELP_HIP(hipMalloc(reinterpret_cast<void**>(&d_val_), (size_t)m * sizeof(double)));
    RELP_HIP(hipMalloc(reinterpret_cast<void**>(&d_out_), (size_t)m * sizeof(double)));
    columns_.assign(m, {});
}
LuBasis::~LuBasis() {
    if (d_idx_) (void)hipFree(d_idx_);
    if (d_val_) (void)hipFree(d_val_);
    if (d_out_) (void)hipFree(d_out_);
    if (stream_) (void)hipStreamDestroy(stream_);
}
void LuBasis::factor_and_upload() {
    RELP_HIP(hipSetDevice(device_));
    std::vector<int> cs(m_ + 1, 0), rows;
    std::vector<double> vals;
    for (int j = 0; j < m_; ++j) {
        for (auto& [r, v] : columns_[j]) {
            rows.push_back(r);
            vals.push_back(v);
        }
        cs[j + 1] = (int)rows.size();
    }
    HostLU f = lu_factor(m_, cs.data(), rows.data(), vals.data(), options_);
    if (f.singular) throw std::runtime_error("singular basis");
    lu_.upload(f, period_ + 1, stream_);
    RELP_HIP(hipStreamSynchronize(stream_));
    have_spike_ = false;
}
void LuBasis::identity() {  // lower_upper/mod.rs:67-76: identity permutations, empty L and U, unit diagonal
    for (int j = 0; j < m_; ++j) columns_[j] = {{j, 1.0}};
    HostLU f;
    f.m = m_;
    f.rowpos.resize(m_);
    f.colpos.resize(m_);
    for (int i = 0; i < m_; ++i) f.rowpos[i] = f.colpos[i] = i;
    f.l_start.assign(m_ + 1, 0);
    f.u_start.assign(m_ + 1, 0);
    f.diag.assign(m_, 1.0);
    RELP_HIP(hipSetDevice(device_));
    lu_.upload(f, period_ + 1, stream_);
    RELP_HIP(hipStreamSynchronize(stream_));
    have_spike_ = false;
}
void LuBasis::invert(const long long* col_start, const int* rows, const double* vals) {  // lower_upper/mod.rs:78-92
    for (int j = 0; j < m_; ++j) {
        columns_[j].clear();
        for (long long e = col_start[j]; e < col_start[j + 1]; ++e) {
            if (rows[e] < 0 || rows[e] >= m_) throw std::invalid_argument("row index out of range");
            if (vals[e] != 0.0) columns_[j].push_back({rows[e], vals[e]});
        }
    }
    factor_and_upload();
}
void LuBasis::left_multiply(int nnz, const int* rows, const double* vals, double* out) {
    if (nnz < 0 || nnz > m_ || (nnz > 0 && (!rows || !vals))) throw std::invalid_argument("bad sparse vector");
    for (int e = 0; e < nnz; ++e)
        if (rows[e] < 0 || rows[e] >= m_) throw std::invalid_argument("row index out of range");
    RELP_HIP(hipSetDevice(device_));
    if (nnz) {
        RELP_HIP(hipMemcpyAsync(d_idx_, rows, nnz * sizeof(int), hipMemcpyHostToDevice, stream_));
        RELP_HIP(hipMemcpyAsync(d_val_, vals, nnz * sizeof(double), hipMemcpyHostToDevice, stream_));
    }
    launch_lu_ftran(lu_.device(), d_idx_, d_val_, nnz, d_out_, 1, stream_);
    RELP_HIP(hipMemcpyAsync(out, d_out_, m_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
    last_column_.clear();
    for (int e = 0; e < nnz; ++e)
        if (vals[e] != 0.0) last_column_.push_back({rows[e], vals[e]});
    have_spike_ = true;
}
void LuBasis::right_multiply(int nnz, const int* slots, const double* vals, double* out) {
    if (nnz < 0 || nnz > m_ || (nnz > 0 && (!slots || !vals))) throw std::invalid_argument("bad sparse vector");
    for (int e = 0; e < nnz; ++e)
        if (slots[e] < 0 || slots[e] >= m_) throw std::invalid_argument("index out of range");
    RELP_HIP(hipSetDevice(device_));
    if (nnz) {
        RELP_HIP(hipMemcpyAsync(d_idx_, slots, nnz * sizeof(int), hipMemcpyHostToDevice, stream_));
        RELP_HIP(hipMemcpyAsync(d_val_, vals, nnz * sizeof(double), hipMemcpyHostToDevice, stream_));
    }
    launch_lu_btran(lu_.device(), d_idx_, d_val_, nnz, d_out_, stream_);
    RELP_HIP(hipMemcpyAsync(out, d_out_, m_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    RELP_HIP(hipStreamSynchronize(stream_));
}
void LuBasis::basis_inverse_row(int slot, double* out) {  // lower_upper/mod.rs:254-272
    const double one = 1.0;
    right_multiply(1, &slot, &one, out);
}
bool LuBasis::generate_element(int i, int nnz, const int* rows, const double* vals, double* out) {  // lower_upper/mod.rs:239-247
    if (i < 0 || i >= m_) throw std::invalid_argument("row index out of range");
    std::vector<double> column(m_);
    const bool had = have_spike_;
    auto saved = last_column_;
    // (a full FTRAN followed by a lookup, as the reference; the spike of a pending change_basis must survive it)
    std::vector<double> spike;
    if (had) {
        spike.resize(m_);
        RELP_HIP(hipMemcpyAsync(spike.data(), lu_.device().spike, m_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    left_multiply(nnz, rows, vals, column.data());
    if (had) {
        RELP_HIP(hipMemcpyAsync(lu_.device().spike, spike.data(), m_ * sizeof(double), hipMemcpyHostToDevice, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
    }
    last_column_ = saved;
    have_spike_ = had;
    *out = column[i];
    return column[i] != 0.0;
}
void LuBasis::change_basis(int pivot_row) {  // lower_upper/mod.rs:94-178
    if (pivot_row < 0 || pivot_row >= m_) throw std::invalid_argument("pivot row out of range");
    if (!have_spike_) throw std::logic_error("change_basis needs the column computed by the last left_multiply");
    RELP_HIP(hipSetDevice(device_));
    launch_lu_update(lu_.device(), pivot_row, stream_);
    RELP_HIP(hipStreamSynchronize(stream_));
    columns_[pivot_row] = last_column_;
    have_spike_ = false;
    const int fl = flags();
    if (fl & LU_FLAG_OVERFLOW) throw std::runtime_error("no room for another update: refactor first (should_refactor)");
    if (fl & LU_FLAG_UNSTABLE) throw std::runtime_error("singular basis after the update");
}
int LuBasis::updates() {
    int v = 0;
    RELP_HIP(hipMemcpy(&v, lu_.device().state + LU_N_UPDATES, sizeof(int), hipMemcpyDeviceToHost));
    return v;
}
int LuBasis::flags() {
    int v = 0;
    RELP_HIP(hipMemcpy(&v, lu_.device().state + LU_FLAGS, sizeof(int), hipMemcpyDeviceToHost));
    return v;
}
bool LuBasis::should_refactor() { return updates() > period_ - 1; }  // lower_upper/mod.rs:249-252 (`> 30` for period 31)
void LuBasis::remove_basis_part(int count, const int* indices) {  // carry/mod.rs:176-180; basis_inverse_rows.rs:212-229
    std::vector<char> gone(m_, 0);
    for (int k = 0; k < count; ++k) {
        if (indices[k] < 0 || indices[k] >= m_ || gone[indices[k]]) throw std::invalid_argument("bad index list");
        gone[indices[k]] = 1;
    }
    std::vector<int> new_index(m_, -1);
    int next = 0;
    for (int i = 0; i < m_; ++i)
        if (!gone[i]) new_index[i] = next++;
    if (next < 1) throw std::invalid_argument("nothing would be left");
    std::vector<std::vector<std::pair<int, double>>> kept;
    for (int j = 0; j < m_; ++j) {
        if (gone[j]) continue;  // the same index removes row i and the basis column of row i (an artificial of a redundant row)
        std::vector<std::pair<int, double>> c;
        for (auto& [r, v] : columns_[j])
            if (!gone[r]) c.push_back({new_index[r], v});
        kept.push_back(std::move(c));
    }
    m_ = next;
    columns_ = std::move(kept);
    factor_and_upload();
}

LuBasis::Factors LuBasis::factors() {
    RELP_HIP(hipSetDevice(device_));
    const DeviceLU& d = lu_.device();
    const int m = m_;
    auto geti = [&](const int* p, size_t n) {
        std::vector<int> v(n);
        if (n) RELP_HIP(hipMemcpy(v.data(), p, n * sizeof(int), hipMemcpyDeviceToHost));
        return v;
    };
    auto getd = [&](const double* p, size_t n) {
        std::vector<double> v(n);
        if (n) RELP_HIP(hipMemcpy(v.data(), p, n * sizeof(double), hipMemcpyDeviceToHost));
        return v;
    };
    RELP_HIP(hipStreamSynchronize(stream_));
    std::vector<int> state = geti(d.state, LU_STATE_WORDS);
    const int n_updates = state[LU_N_UPDATES];
    Factors f;
    f.row_permutation = geti(d.rowpos, m);
    f.column_permutation = geti(d.colpos, m);
    std::vector<int> rank = geti(d.rank, m), seq = geti(d.seq, m);
    // L by columns (never rotated: the rotations only act on U, mod.rs:141-161)
    std::vector<int> lcs = geti(d.l_cstart, m + 1), lcrow = geti(d.l_crow, lcs[m]);
    std::vector<double> lcval = getd(d.l_cval, lcs[m]);
    f.l_start.assign(lcs.begin(), lcs.end());
    f.l_row = lcrow;
    f.l_val = lcval;
    for (int j = 0; j < m; ++j) {  // ascending rows inside a column (sparse vectors are sorted, sparse.rs:90-95)
        std::vector<std::pair<int, double>> col;
        for (int e = lcs[j]; e < lcs[j + 1]; ++e) col.push_back({lcrow[e], lcval[e]});
        std::sort(col.begin(), col.end());
        for (int e = lcs[j], k = 0; e < lcs[j + 1]; ++e, ++k) {
            f.l_row[e] = col[k].first;
            f.l_val[e] = col[k].second;
        }
    }
    // U by logical column: entries (logical row, value), ascending
    std::vector<int> ucs = geti(d.u_cstart, m), ucl = geti(d.u_clen, m), ucrow = geti(d.u_crow, state[LU_UC_TOP]);
    std::vector<double> ucval = getd(d.u_cval, state[LU_UC_TOP]), diag = getd(d.diag, m);
    f.u_start.assign(m + 1, 0);
    f.upper_diagonal.resize(m);
    for (int c = 0; c < m; ++c) {
        const int j = seq[c];
        f.upper_diagonal[c] = diag[j];
        std::vector<std::pair<int, double>> col;
        for (int e = ucs[j]; e < ucs[j] + ucl[j]; ++e) col.push_back({rank[ucrow[e]], ucval[e]});
        std::sort(col.begin(), col.end());
        for (auto& [r, v] : col) {
            f.u_row.push_back(r);
            f.u_val.push_back(v);
        }
        f.u_start[c + 1] = (long long)f.u_row.size();
    }
    // etas: indices as the reference saw them when each one was made (before its own rotation)
    std::vector<int> es = geti(d.eta_start, n_updates + 1), ep = geti(d.eta_pivot, n_updates), ei = geti(d.eta_idx, state[LU_ETA_TOP]);
    std::vector<double> ev = getd(d.eta_val, state[LU_ETA_TOP]);
    std::vector<int> r(m);
    for (int i = 0; i < m; ++i) r[i] = i;
    f.eta_start.assign(1, 0);
    for (int k = 0; k < n_updates; ++k) {
        std::vector<std::pair<int, double>> entries;
        for (int e = es[k]; e < es[k + 1]; ++e) entries.push_back({r[ei[e]], ev[e]});
        std::sort(entries.begin(), entries.end());
        for (auto& [i, v] : entries) {
            f.eta_index.push_back(i);
            f.eta_value.push_back(v);
        }
        f.eta_pivot.push_back(r[ep[k]]);
        f.eta_start.push_back((long long)f.eta_index.size());
        const int rt = r[ep[k]];
        for (int i = 0; i < m; ++i) r[i] = i == ep[k] ? m - 1 : (r[i] > rt ? r[i] - 1 : r[i]);
    }
    return f;
}

}  // namespace relp
