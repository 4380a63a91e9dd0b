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
// ONE workgroup (16 waves on one CU) owns the solve; the vector(s) and the factor being solved with are staged in LDS
// (`lu_stage_and_solve`).  The host's refactorisation also produces the level sets of L, U, U' and L' (`lu_schedules`); a
// WIDE level is solved by all threads followed by one barrier, a run of NARROW levels (the long tail every sparse factor has)
// by wave 0 alone without barriers, several lanes per row, as a software pipeline over the levels.  Forrest-Tomlin updates do
// not touch those schedules: the spiked columns are bordered into a dense trailing block T (<= 64 x 64) that one wave solves
// out of registers (lu.hpp).  The reference walks ordered maps (`BTreeMap`) with a column scan per popped entry; here both
// orientations of L and U are resident so every solve is a gather.  Everything is deterministic: a row adds its entries in
// storage order, reductions have a fixed tree.
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

__global__ void __launch_bounds__(256) lu_init_kernel(DeviceLU lu, const int* n_levels) {
    const int m = lu.m;
    const int stride = blockDim.x * gridDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = first; i < m; i += stride) {
        lu.u_rlen[i] = lu.u_rstart[i + 1] - lu.u_rstart[i];
        lu.app_len[i] = 0;
        lu.slot_of[i] = -1;
    }
    for (int i = first; i < lu.max_updates * lu.ldt; i += stride) lu.T[i] = 0.0;
    for (int i = first; i < lu.max_updates; i += stride) {
        lu.trail_pos[i] = -1;
        lu.s_clen[i] = 0;
        lu.s_cstart[i] = 0;
    }
    if (first == 0) {
        lu.state[LU_N_UPDATES] = 0;
        lu.state[LU_S_TOP] = 0;
        lu.state[LU_ETA_TOP] = 0;
        lu.state[LU_FLAGS] = 0;
        for (int k = 0; k < 4; ++k) lu.state[LU_N_LEVELS + k] = n_levels[k];
        lu.eta_start[0] = 0;
    }
}
}  // namespace

bool LuFactors::upload(const HostLU& f, int max_updates, hipStream_t stream) {
    const int m = f.m;
    const size_t nl = (size_t)f.nnz_l(), nu = (size_t)f.nnz_u();
    if (max_updates < 1) max_updates = 1;
    if (max_updates > LU_MAX_SLOTS) max_updates = LU_MAX_SLOTS;
    // The layout depends on capacities only, so that the device addresses (and a captured hipGraph that holds them) survive
    // a refactorisation; it changes when a factor outgrows its capacity (or m / the update capacity change).
    bool layout_changed = m != d_.m || max_updates != d_.max_updates;
    if (layout_changed) cap_l_ = cap_u_ = 0;
    if (nl > cap_l_ || cap_l_ == 0) { cap_l_ = nl + nl / 2 + 256; layout_changed = true; }
    if (nu > cap_u_ || cap_u_ == 0) { cap_u_ = nu + nu / 2 + 256; layout_changed = true; }
    const size_t cl = cap_l_, cu = cap_u_;
    const int ldt = max_updates + 1;
    // ---- uploaded prefix ----------------------------------------------------------------------------------------------
    Carver c;
    const size_t o_rowpos = c.take<int>(m), o_colpos = c.take<int>(m);
    const size_t o_lrs = c.take<int>(m + 1), o_lcs = c.take<int>(m + 1), o_urs = c.take<int>(m + 1), o_ucs = c.take<int>(m);
    const size_t o_uclen = c.take<int>(m);
    const size_t o_diag = c.take<double>(m);
    const size_t o_lrcol = c.take<int>(cl), o_lcrow = c.take<int>(cl);
    const size_t o_lrval = c.take<double>(cl), o_lcval = c.take<double>(cl);
    size_t o_sched_start[4], o_sched_row[4];
    for (int k = 0; k < 4; ++k) {
        o_sched_start[k] = c.take<int>(m + 2);
        o_sched_row[k] = c.take<int>(m);
    }
    const size_t o_nlev = c.take<int>(4);
    const size_t upload_bytes = c.offset;
    // ---- device only (the four U_bb arrays are uploaded one by one) ----------------------------------------------------
    const size_t app = (size_t)m * max_updates;
    const size_t o_urcol = c.take<int>(cu), o_ucrow = c.take<int>(cu);
    const size_t o_urval = c.take<double>(cu), o_ucval = c.take<double>(cu);
    const size_t o_urlen = c.take<int>(m);
    const size_t o_applen = c.take<int>(m), o_appslot = c.take<int>(app), o_appval = c.take<double>(app);
    const size_t o_scs = c.take<int>(max_updates), o_scl = c.take<int>(max_updates), o_scrow = c.take<int>(app), o_scval = c.take<double>(app);
    const size_t o_T = c.take<double>((size_t)max_updates * ldt);
    const size_t o_trail = c.take<int>(max_updates), o_slotof = c.take<int>(m);
    const size_t o_eta_start = c.take<int>(max_updates + 2), o_eta_pivot = c.take<int>(max_updates + 1);
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
    {
        HostLU& fs = const_cast<HostLU&>(f);
        if (fs.lev_row[0].empty()) lu_schedules(fs);
        int* nlev = reinterpret_cast<int*>(h + o_nlev);
        for (int k = 0; k < 4; ++k) {
            std::memcpy(h + o_sched_start[k], f.lev_start[k].data(), f.lev_start[k].size() * sizeof(int));
            std::memcpy(h + o_sched_row[k], f.lev_row[k].data(), m * sizeof(int));
            nlev[k] = (int)f.lev_start[k].size() - 1;
        }
    }
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
    d.ldt = ldt;
    auto I = [&](size_t o) { return reinterpret_cast<int*>(dev_ + o); };
    auto D = [&](size_t o) { return reinterpret_cast<double*>(dev_ + o); };
    d.rowpos = I(o_rowpos);
    d.colpos = I(o_colpos);
    d.l_rstart = I(o_lrs); d.l_rcol = I(o_lrcol); d.l_rval = D(o_lrval);
    d.l_cstart = I(o_lcs); d.l_crow = I(o_lcrow); d.l_cval = D(o_lcval);
    d.u_rstart = I(o_urs); d.u_rlen = I(o_urlen); d.u_rcol = I(o_urcol); d.u_rval = D(o_urval);
    d.u_cstart = I(o_ucs); d.u_clen = I(o_uclen); d.u_crow = I(o_ucrow); d.u_cval = D(o_ucval);
    d.app_len = I(o_applen); d.app_slot = I(o_appslot); d.app_val = D(o_appval);
    d.s_cstart = I(o_scs); d.s_clen = I(o_scl); d.s_crow = I(o_scrow); d.s_cval = D(o_scval);
    d.s_capacity = (int)app;
    d.T = D(o_T);
    d.trail_pos = I(o_trail);
    d.slot_of = I(o_slotof);
    d.diag = D(o_diag);
    d.eta_start = I(o_eta_start); d.eta_pivot = I(o_eta_pivot); d.eta_idx = I(o_eta_idx); d.eta_val = D(o_eta_val);
    d.eta_capacity = (int)app;
    d.spike = D(o_spike);
    d.state = I(o_state);
    for (int k = 0; k < 4; ++k) {
        d.sched_start[k] = I(o_sched_start[k]);
        d.sched_row[k] = I(o_sched_row[k]);
    }
    d_ = d;
    hipLaunchKernelGGL(lu_init_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, d_, I(o_nlev));
    nnz_l = (long long)nl;
    nnz_u = (long long)nu;
    lu_depths(f, &depth_l, &depth_u);
    return layout_changed;
}

// LDS of the solve kernels: x0, x1 (doubles); one count per 64 rows for the ordered compactions; reductions; T and the
// trailing parts of the two vectors; and the factor area -- the headers of the triangular factor being solved with
// (start, length, level order, 1/diagonal: 24 bytes per row) always, its entries (12 bytes each) when they fit.
static size_t lu_lds_fixed_bytes(int m, int max_updates) {
    const size_t mm = (size_t)((m + 1) & ~1);
    return 2 * mm * sizeof(double) + ((size_t)(m + 63) / 64 + 2) * sizeof(int) + 64 * sizeof(double) +
           ((size_t)max_updates * (max_updates + 1) + 4 * LU_MAX_SLOTS) * sizeof(double) + 64 +
           mm * sizeof(double) + 3 * mm * sizeof(int) + (size_t)(m + 2) * sizeof(int) + 64;
}
constexpr size_t LU_LDS_TOTAL = 160 * 1024 - 1024;  // what a kernel may ask for (static LDS of the fused kernel comes on top)
static size_t lu_lds_bytes_for(int m, int max_updates) {
    (void)m;
    (void)max_updates;
    return LU_LDS_TOTAL - 2048;  // always the whole CU: one workgroup per solve, and the factor area takes what is left
}
size_t LuFactors::lds_bytes(int) const { return lu_lds_bytes_for(d_.m, d_.max_updates); }
bool lu_fits_lds(int m, int max_updates) {
    if (max_updates < 1) max_updates = 1;
    if (max_updates > LU_MAX_SLOTS) max_updates = LU_MAX_SLOTS;
    return lu_lds_fixed_bytes(m, max_updates) + 16 * 1024 <= LU_LDS_TOTAL - 2048;
}

// =====================================================================================================
// device: level-scheduled triangular solves out of LDS
// =====================================================================================================
// Explicit LDS pointer types.  A generic `volatile double*` that happens to point into LDS compiles to flat_load ... sc0 sc1
// -- measured: about 2000 cycles per dependent access instead of the ~64 of a ds_read -- so every LDS array of this file is
// typed by address space.
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) char lds_i8;
// One triangular factor in one orientation, staged for a solve.  The row records are LDS arrays in SCHEDULE order (record r
// = the r-th row of the level order): position, first entry, length (-1: not part of this triangle), 1 / diagonal.  `idx` /
// `val` are LDS copies of the entries when the factor fits, else the L2-resident arrays.
template <class IdxPtr, class ValPtr>
struct Factor {
    const lds_i32* rec_i;
    const lds_i32* rec_s;
    const lds_i32* rec_n;
    const lds_f64* rec_dinv;
    IdxPtr idx;
    ValPtr val;
    const lds_i32* lev_start;
    int n_levels;
};

// In place:  x[i] <- (x[i] - sum_e val[e] x[idx[e]]) / diag[i], level by level (the rows of a level are independent; the
// host computed the levels at refactorisation time and Forrest-Tomlin updates only remove entries, so they stay valid).
// A level wider than a wave is shared by all threads and followed by a barrier; a run of narrow levels -- the long tail of a
// basis factor: chains of one or two rows -- is walked by wave 0 alone WITHOUT barriers (LDS keeps one wave's accesses in
// order) as a three-stage software pipeline: while level l waits for its operands x[idx], the entries of level l + 1 and the
// row records of level l + 2 are already in flight, so a level costs ONE LDS round trip.  Everything is branch-free (clamped
// addresses, selected contributions): a branch per entry would put an s_waitcnt behind every load (measured: 3000 cycles per
// level).  Deterministic: a row adds its entries in storage order.
// sum over aligned groups of G = 2^k lanes by DPP moves; valid in the LAST lane of every group (G <= 16: in all its lanes)
__device__ __forceinline__ double group_sum(double v, const int G) {
    if (G >= 2) v += dpp_f64<DPP_QUAD_1032, 0xF>(0.0, v);
    if (G >= 4) v += dpp_f64<DPP_QUAD_2301, 0xF>(0.0, v);
    if (G == 8) v += dpp_f64<DPP_ROW_HALF_MIRROR, 0xF>(0.0, v);
    if (G >= 16) {
        v += dpp_f64<DPP_ROW_ROR4, 0xF>(0.0, v);
        v += dpp_f64<DPP_ROW_ROR8, 0xF>(0.0, v);
    }
    if (G >= 32) v += dpp_f64<DPP_ROW_BCAST15, 0xA>(0.0, v);
    if (G == 64) v += dpp_f64<DPP_ROW_BCAST31, 0xC>(0.0, v);
    return v;
}
__device__ __forceinline__ int lanes_per_row(int width) {
    return width <= 1 ? 64 : width <= 2 ? 32 : width <= 4 ? 16 : width <= 8 ? 8 : width <= 16 ? 4 : width <= 32 ? 2 : 1;
}

template <int NRHS, bool HAS_DIAG, class F>
__device__ __forceinline__ void solve_levels(const F fac, volatile lds_f64* x0, volatile lds_f64* x1, unsigned long long* dbg = nullptr) {
    (void)dbg;
    const int tid = threadIdx.x, T = blockDim.x;
    // ---- wide levels: a thread per row ---------------------------------------------------------------------------------
    auto whole_row = [&](int r) {
        const int i = fac.rec_i[r], st = fac.rec_s[r], n = fac.rec_n[r];
        if (n < 0) return;
        double a0 = x0[i], a1 = NRHS == 2 ? x1[i] : 0.0;
        for (int k = 0; k < n; k += 4) {  // four entries per trip: two LDS round trips instead of eight
            const int last = n - 1;
            const int k0 = st + k, k1 = st + min(k + 1, last), k2 = st + min(k + 2, last), k3 = st + min(k + 3, last);
            const int c0 = fac.idx[k0], c1 = fac.idx[k1], c2 = fac.idx[k2], c3 = fac.idx[k3];
            const double v0 = fac.val[k0], v1 = fac.val[k1], v2 = fac.val[k2], v3 = fac.val[k3];
            const double p0 = x0[c0], p1 = x0[c1], p2 = x0[c2], p3 = x0[c3];
            a0 -= v0 * p0;
            a0 -= k + 1 < n ? v1 * p1 : 0.0;
            a0 -= k + 2 < n ? v2 * p2 : 0.0;
            a0 -= k + 3 < n ? v3 * p3 : 0.0;
            if (NRHS == 2) {
                const double q0 = x1[c0], q1 = x1[c1], q2 = x1[c2], q3 = x1[c3];
                a1 -= v0 * q0;
                a1 -= k + 1 < n ? v1 * q1 : 0.0;
                a1 -= k + 2 < n ? v2 * q2 : 0.0;
                a1 -= k + 3 < n ? v3 * q3 : 0.0;
            }
        }
        const double dinv = HAS_DIAG ? fac.rec_dinv[r] : 1.0;
        x0[i] = a0 * dinv;
        if (NRHS == 2) x1[i] = a1 * dinv;
    };
    // ---- levels of at most a wave's worth of rows: blockDim / 64 threads per row (entries split among them, DPP group sum) ------
    // One LDS round trip per level whatever the row lengths, then the barrier.  (A barrier-free software pipeline of wave 0 over
    // runs of such levels was measured at 1.8-2.3 k cycles per level -- every level is a dependent header -> entry -> operand ->
    // sum -> publish chain for ONE wave; the chains of tiny levels at the end of the schedule are solved as a dense block
    // instead, solve_dense_tail.)
    auto group_rows = [&](int ls, int le) {
        const int G = T / WAVE;
        const int row = tid / G, sub = tid % G;
        const bool in = row < le - ls;
        const int r = ls + (in ? row : 0);
        const int i = fac.rec_i[r], st = fac.rec_s[r];
        const int n = in ? fac.rec_n[r] : -1;
        double s0 = 0.0, s1 = 0.0;
        for (int k = sub; k < n; k += G) {
            const int c = fac.idx[st + k];
            const double v = fac.val[st + k];
            s0 += v * x0[c];
            if (NRHS == 2) s1 += v * x1[c];
        }
        s0 = group_sum(s0, G);
        if (NRHS == 2) s1 = group_sum(s1, G);
        if (n >= 0 && sub == G - 1) {
            const double dinv = HAS_DIAG ? fac.rec_dinv[r] : 1.0;
            x0[i] = (x0[i] - s0) * dinv;
            if (NRHS == 2) x1[i] = (x1[i] - s1) * dinv;
        }
    };
    // (fetching the next level's row records and first entries while a level waits for its operands was measured: no gain --
    //  LDS returns in order, so the operand read waits for the prefetch anyway)
    const int n_levels = fac.n_levels;
    for (int l = 0; l < n_levels; ++l) {
        const int ls = fac.lev_start[l], le = fac.lev_start[l + 1];
        if (le - ls > WAVE || T < 2 * WAVE) {
            for (int r = ls + tid; r < le; r += T) whole_row(r);
        } else {
            group_rows(ls, le);
        }
        __syncthreads();
    }
}

// ---- eta files --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_value(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double lane_value(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// FTRAN direction (eta_file.rs:72-105): for each update in order  v[t] -= sum_k r_k v[k].  One wave; a dot product per eta.
// Lane l keeps the bounds and the pivot of eta l (at most 64 of them), so the loop's only memory traffic are the entries,
// and those are prefetched one eta ahead.
__device__ __forceinline__ void apply_etas_forward(const DeviceLU& lu, const int n_updates, volatile lds_f64* x0) {
    if (threadIdx.x >= WAVE || n_updates <= 0) return;
    const int lane = threadIdx.x;
    const int my_start = lane <= n_updates ? lu.eta_start[lane] : 0;
    const int my_end = lane < n_updates ? lu.eta_start[lane + 1] : 0;
    const int my_pivot = lane < n_updates ? lu.eta_pivot[lane] : 0;
    int s = lane_value(my_start, 0), s_end = lane_value(my_end, 0);
    int nidx = 0;
    double nval = 0.0;
    if (s + lane < s_end) {
        nidx = lu.eta_idx[s + lane];
        nval = lu.eta_val[s + lane];
    }
    for (int k = 0; k < n_updates; ++k) {
        const int e_end = s_end;
        const int t = lane_value(my_pivot, k);
        const int cidx = nidx;
        const double cval = nval;
        const bool chave = s + lane < e_end;
        const int s_next = e_end;
        s_end = k + 1 < n_updates ? lane_value(my_end, k + 1) : e_end;
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
__device__ __forceinline__ void apply_etas_backward(const DeviceLU& lu, const int n_updates, volatile lds_f64* x0, volatile lds_f64* x1) {
    if (threadIdx.x >= WAVE || n_updates <= 0) return;
    const int lane = threadIdx.x;
    const int my_start = lane < n_updates ? lu.eta_start[lane] : 0;
    const int my_end = lane < n_updates ? lu.eta_start[lane + 1] : 0;
    const int my_pivot = lane < n_updates ? lu.eta_pivot[lane] : 0;
    int s = lane_value(my_start, n_updates - 1), e_end = lane_value(my_end, n_updates - 1);
    int nidx = 0;
    double nval = 0.0;
    if (s + lane < e_end) {
        nidx = lu.eta_idx[s + lane];
        nval = lu.eta_val[s + lane];
    }
    for (int k = n_updates - 1; k >= 0; --k) {
        const int t = lane_value(my_pivot, k);
        const int cidx = nidx;
        const double cval = nval;
        const bool chave = s + lane < e_end;
        const int cs = s, cend = e_end;
        if (k > 0) {
            s = lane_value(my_start, k - 1);
            e_end = lane_value(my_end, k - 1);
            if (s + lane < e_end) {
                nidx = lu.eta_idx[s + lane];
                nval = lu.eta_val[s + lane];
            }
        }
        const double p0 = x0[t];
        const double p1 = NRHS == 2 ? x1[t] : 0.0;
        if (p0 == 0.0 && p1 == 0.0) continue;
        if (chave) {
            x0[cidx] = x0[cidx] - cval * p0;
            if (NRHS == 2) x1[cidx] = x1[cidx] - cval * p1;
        }
        for (int e = cs + lane + WAVE; e < cend; e += WAVE) {
            const int j = lu.eta_idx[e];
            const double r = lu.eta_val[e];
            x0[j] = x0[j] - r * p0;
            if (NRHS == 2) x1[j] = x1[j] - r * p1;
        }
    }
}

constexpr int TAIL_MAX = 64;  // rows of the dense tail: one lane each
// LDS carve-up shared by every kernel of this file
struct LuShared {
    volatile lds_f64* x0;
    volatile lds_f64* x1;
    int* group_count;  // one slot per 64 rows (+2)   (generic pointers: used with barriers around, a handful of accesses)
    double* red;       // 64 doubles
    volatile lds_f64* T;     // the trailing block, max_updates x ldt
    volatile lds_f64* xt0;   // trailing values by slot (LU_MAX_SLOTS each)
    volatile lds_f64* xt1;
    volatile lds_f64* st0;   // BTRAN: right-hand sides of the trailing solve
    volatile lds_f64* st1;
    // factor area
    lds_f64* f_dinv;    // [m]
    lds_i32* f_start;   // [m]
    lds_i32* f_len;     // [m]
    lds_i32* f_levrow;  // [m]
    lds_i32* f_levstart;  // [m + 2]
    lds_i8* f_entries;  // what is left of the LDS
    int f_entry_capacity;  // entries (12 bytes each) that fit there
    // dense tail of a triangular solve (solve_dense_tail): nullptr when the LDS has no room for it
    lds_f64* tail_M;     // [64 x 64], M[k * 64 + i] = entry (tail row i, tail column k)
    lds_i32* tail_map;   // [m] position -> index in the tail, or -1
    lds_f64* tail_r0;    // [64] right-hand sides
    lds_f64* tail_r1;
    lds_i32* tail_info;  // [2] first level of the tail, its first record
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
__device__ __forceinline__ LuShared lu_shared(char* smem_generic, int m, int max_updates, int lds_bytes) {
    const int mm = (m + 1) & ~1;
    lds_i8* smem = (lds_i8*)smem_generic;
    LuShared s;
    lds_f64* x0 = (lds_f64*)smem;
    lds_f64* x1 = x0 + mm;
    lds_f64* red = x1 + mm;
    lds_f64* T = red + 64;
    lds_f64* xt0 = T + max_updates * (max_updates + 1);
    lds_f64* xt1 = xt0 + LU_MAX_SLOTS;
    lds_f64* st0 = xt1 + LU_MAX_SLOTS;
    lds_f64* st1 = st0 + LU_MAX_SLOTS;
    s.x0 = x0;
    s.x1 = x1;
    s.red = (double*)red;
    s.T = T;
    s.xt0 = xt0;
    s.xt1 = xt1;
    s.st0 = st0;
    s.st1 = st1;
    s.f_dinv = st1 + LU_MAX_SLOTS;
    s.f_start = (lds_i32*)(s.f_dinv + mm);
    s.f_len = s.f_start + mm;
    s.f_levrow = s.f_len + mm;
    s.f_levstart = s.f_levrow + mm;
    lds_i32* group_count = s.f_levstart + ((m + 2 + 1) & ~1);
    s.group_count = (int*)group_count;
    lds_i8* end = (lds_i8*)(group_count + (((m + 63) / 64 + 2 + 1) & ~1));
    s.tail_M = nullptr;
    s.tail_map = nullptr;
    s.tail_r0 = s.tail_r1 = nullptr;
    s.tail_info = nullptr;
    {   // the dense tail takes 33 KB + 4 m bytes when at least 24 KB stay for the factor's entries
        const int tail_bytes = (TAIL_MAX * TAIL_MAX + 2 * TAIL_MAX) * 8 + ((mm + 2) * 4);
        if (lds_bytes - (int)(end - smem) - tail_bytes >= 24 * 1024) {
            s.tail_M = (lds_f64*)end;
            s.tail_r0 = s.tail_M + TAIL_MAX * TAIL_MAX;
            s.tail_r1 = s.tail_r0 + TAIL_MAX;
            s.tail_map = (lds_i32*)(s.tail_r1 + TAIL_MAX);
            s.tail_info = s.tail_map + mm;
            end = (lds_i8*)(s.tail_info + 2);
        }
    }
    s.f_entries = end;
    s.f_entry_capacity = (int)((lds_bytes - (int)(end - smem)) / 12);
    if (s.f_entry_capacity < 0) s.f_entry_capacity = 0;
    s.dbg = nullptr;
    s.t_prev = nullptr;
    return s;
}
// x0 (x1) <- 0, T staged from global.  Ends with a barrier.
__device__ __forceinline__ void lu_clear(const DeviceLU& lu, const LuShared& sh, int n_updates, bool two) {
    const int m = lu.m;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        sh.x0[i] = 0.0;
        if (two) sh.x1[i] = 0.0;
    }
    for (int i = threadIdx.x; i < n_updates * lu.ldt; i += blockDim.x) sh.T[i] = lu.T[i];
    __syncthreads();
}

// The last `n_tail` <= 64 records of the schedule as one dense triangular block.  All threads: every tail row, blockDim / 64
// threads each, walks its entries -- an entry whose column is a tail row goes into the dense block (LDS, column-major by
// dependency), the others are multiplied with their operands, which the head levels have finished -- and leaves the row's
// right-hand side.  Then ONE wave substitutes through the block: lane i owns row i, step k broadcasts x_k by readlane -- n_tail
// steps of a few instructions instead of one LDS round trip per level.  Rows that are not part of the triangle (len < 0) stay
// as they are.  x0 / x1 of the head must be complete (barrier); ends with a barrier.
template <int NRHS, bool HAS_DIAG, class F>
__device__ __forceinline__ void solve_dense_tail(const F fac, const LuShared& sh, const int tail_first, const int n_tail) {
    const int tid = threadIdx.x, T = blockDim.x;
    const int G = T / TAIL_MAX;  // threads per tail row (a power of two: 16 at 1024 threads)
    const int row = tid / G, sub = tid % G;
    {
        const bool active = row < n_tail;
        const int rr = tail_first + (active ? row : 0);
        const int st = fac.rec_s[rr];
        const int n = active ? fac.rec_n[rr] : -1;
        const int pos = fac.rec_i[rr];
        double p0 = 0.0, p1 = 0.0;
        for (int k = sub; k < n; k += G) {
            const int c = fac.idx[st + k];
            const double v = fac.val[st + k];
            const int ti = sh.tail_map[c];
            if (ti >= 0) {
                sh.tail_M[ti * TAIL_MAX + row] = v;
            } else {
                p0 += v * sh.x0[c];
                if (NRHS == 2) p1 += v * sh.x1[c];
            }
        }
        p0 = group_sum(p0, G);
        if (NRHS == 2) p1 = group_sum(p1, G);
        if (active && sub == G - 1) {
            sh.tail_r0[row] = sh.x0[pos] - p0;
            if (NRHS == 2) sh.tail_r1[row] = sh.x1[pos] - p1;
        }
    }
    __syncthreads();
    if (tid < WAVE) {
        const int lane = tid;
        const bool in = lane < n_tail;
        const int rr = tail_first + (in ? lane : 0);
        const bool active = in && fac.rec_n[rr] >= 0;
        const int pos = fac.rec_i[rr];
        const double dinv = (HAS_DIAG && active) ? fac.rec_dinv[rr] : 1.0;
        double r0 = active ? sh.tail_r0[lane] : 0.0;
        double r1 = (NRHS == 2 && active) ? sh.tail_r1[lane] : 0.0;
        for (int k = 0; k < n_tail; ++k) {
            const double mk = sh.tail_M[k * TAIL_MAX + lane];
            const double xk0 = lane_value(r0 * dinv, k);
            if (lane > k) r0 -= mk * xk0;
            if (NRHS == 2) {
                const double xk1 = lane_value(r1 * dinv, k);
                if (lane > k) r1 -= mk * xk1;
            }
        }
        if (active) {
            sh.x0[pos] = r0 * dinv;
            if (NRHS == 2) sh.x1[pos] = r1 * dinv;
        }
    }
    __syncthreads();
}

// Copy one orientation of one factor into the factor area (row records in schedule order always, entries when they fit)
// and solve with it.
//   g_start / g_len: first entry and length per row (g_len == nullptr: g_start has m + 1 entries); nnz: entries to copy;
//   skip: rows with skip[i] >= 0 are not part of the triangle; diag: nullptr = unit; sched: which level schedule.
// The caller's x0 / x1 must be complete (barrier) before; ends with a barrier (solve_levels does).
template <int NRHS, bool HAS_DIAG>
__device__ __forceinline__ void lu_stage_and_solve(const DeviceLU& lu, const LuShared& sh, const int* g_start, const int* g_len,
                                                   const int* g_idx, const double* g_val, const int nnz, const int* skip,
                                                   const double* diag, const int sched) {
    const int m = lu.m;
    const int tid = threadIdx.x, T = blockDim.x;
    const int n_levels = lu.state[LU_N_LEVELS + sched];
    for (int r = tid; r < m; r += T) {
        const int i = lu.sched_row[sched][r];
        const int st = g_start[i];
        int n = g_len ? g_len[i] : g_start[i + 1] - st;
        if (skip && skip[i] >= 0) n = -1;
        sh.f_levrow[r] = i;
        sh.f_start[r] = st;
        sh.f_len[r] = n;
        if (HAS_DIAG) sh.f_dinv[r] = 1.0 / diag[i];
        if (sh.tail_map) sh.tail_map[i] = -1;
    }
    for (int l = tid; l <= n_levels; l += T) {
        const int first = lu.sched_start[sched][l];
        sh.f_levstart[l] = first;
        if (sh.tail_info) {  // the tail: the last levels with at most TAIL_MAX rows altogether
            const int before = l > 0 ? lu.sched_start[sched][l - 1] : -1;
            if (m - first <= TAIL_MAX && (l == 0 || m - before > TAIL_MAX)) {
                sh.tail_info[0] = l;
                sh.tail_info[1] = first;
            }
        }
    }
    const bool fits = nnz + 4 <= sh.f_entry_capacity;
    lds_i32* e_idx = (lds_i32*)(sh.f_entries + sh.f_entry_capacity * 8);
    lds_f64* e_val = (lds_f64*)sh.f_entries;
    if (fits) {
        for (int e = tid; e < nnz; e += T) {
            e_idx[e] = g_idx[e];
            e_val[e] = g_val[e];
        }
        if (tid < 4) {  // (the clamped reads of an empty last row land here)
            e_idx[nnz + tid] = 0;
            e_val[nnz + tid] = 0.0;
        }
    }
    __syncthreads();
    lu_stamp(sh, 13 + sched);
#ifdef RELP_STAMPS
    if (sh.dbg && threadIdx.x == 0) {
        sh.dbg[20 + sched] += n_levels;
        int wide = 0;
        for (int l = 0; l < n_levels; ++l) wide += (sh.f_levstart[l + 1] - sh.f_levstart[l] > WAVE) ? 1 : 0;
        sh.dbg[24 + sched] += wide;
        sh.dbg[28 + sched] += fits ? 1 : 0;
    }
#endif
    // The narrow levels at the end of the schedule -- chains of a few rows each, 10-30 levels of them on a basis of 25FV47, every
    // level one dependent LDS round trip for one wave -- are solved as ONE DENSE triangular block instead (solve_dense_tail).
    int head_levels = n_levels, tail_first = m;
    if (sh.tail_info) {
        const int l_tail = sh.tail_info[0];
        tail_first = sh.tail_info[1];
        if (n_levels - l_tail >= 3 && m - tail_first >= 4) head_levels = l_tail;  // (a short tail is not worth the set-up)
        else tail_first = m;
    }
    if (tail_first < m) {
        for (int e = tid; e < TAIL_MAX * TAIL_MAX; e += T) sh.tail_M[e] = 0.0;
        if (tid < m - tail_first) sh.tail_map[sh.f_levrow[tail_first + tid]] = tid;
        __syncthreads();
    }
    if (fits) {
        Factor<const lds_i32*, const lds_f64*> f{sh.f_levrow, sh.f_start, sh.f_len, sh.f_dinv, e_idx, e_val, sh.f_levstart, head_levels};
        solve_levels<NRHS, HAS_DIAG>(f, sh.x0, sh.x1, sh.dbg);
        if (tail_first < m) solve_dense_tail<NRHS, HAS_DIAG>(f, sh, tail_first, m - tail_first);
    } else {
        Factor<const int*, const double*> f{sh.f_levrow, sh.f_start, sh.f_len, sh.f_dinv, g_idx, g_val, sh.f_levstart, head_levels};
        solve_levels<NRHS, HAS_DIAG>(f, sh.x0, sh.x1, sh.dbg);
        if (tail_first < m) solve_dense_tail<NRHS, HAS_DIAG>(f, sh, tail_first, m - tail_first);
    }
}

// FTRAN on the vector in sh.x0 (position space, P already applied): L solve, etas, [spike], U solve (trailing block by one
// wave, spike contributions, then the base rows).  Ends with a barrier.
__device__ __forceinline__ void lu_ftran_block(const DeviceLU& lu, const LuShared& sh, const int n_updates, int& epoch,
                                               double* spike_out) {
    (void)epoch;
    const int m = lu.m;
    lu_stage_and_solve<1, false>(lu, sh, lu.l_rstart, nullptr, lu.l_rcol, lu.l_rval, lu.l_rstart[m], nullptr, nullptr, 0);
    lu_stamp(sh, 2);
    if (n_updates > 0) {
        apply_etas_forward(lu, n_updates, sh.x0);
        __syncthreads();
    }
    lu_stamp(sh, 3);
    if (spike_out)
        for (int i = threadIdx.x; i < m; i += blockDim.x) spike_out[i] = sh.x0[i];
    if (n_updates > 0) {
        if (threadIdx.x < WAVE) {  // T x_T = y_T, back substitution by slot (lower_upper/mod.rs:307-321 on the trailing block)
            const int lane = threadIdx.x;
            const int pos = lane < n_updates ? lu.trail_pos[lane] : -1;
            double xk = pos >= 0 ? sh.x0[pos] : 0.0;
            const double dk = pos >= 0 ? lu.diag[pos] : 1.0;
            for (int b = n_updates - 1; b >= 0; --b) {
                if (lane == b) xk = xk / dk;
                const double xb = lane_value(xk, b);
                if (lane < b && xb != 0.0) xk -= sh.T[lane * lu.ldt + b] * xb;
            }
            if (lane < n_updates) sh.xt0[lane] = xk;
            if (pos >= 0) sh.x0[pos] = xk;
        }
        __syncthreads();
        // y_b -= S x_T: a base row's spike entries (one per update at most), operands final
        const int stride = lu.max_updates;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const int n_app = lu.app_len[i];
            if (n_app <= 0) continue;
            double acc = 0.0;
            for (int e = 0; e < n_app; ++e) acc += lu.app_val[i * stride + e] * sh.xt0[lu.app_slot[i * stride + e]];
            sh.x0[i] = sh.x0[i] - acc;
        }
    }
    lu_stage_and_solve<1, true>(lu, sh, lu.u_rstart, lu.u_rlen, lu.u_rcol, lu.u_rval, lu.u_rstart[m], n_updates > 0 ? lu.slot_of : nullptr, lu.diag, 1);
    lu_stamp(sh, 4);
}

// BTRAN on the vectors in sh.x0 (and sh.x1), position space with Q applied.  `after_upper` runs between the U solve and the
// etas, when x0 = (e_t' U^-1) for a unit input (the Forrest-Tomlin row comes from there).  Ends with a barrier.
template <int NRHS, class AfterUpper>
__device__ __forceinline__ void lu_btran_block(const DeviceLU& lu, const LuShared& sh, const int n_updates, int& epoch,
                                               AfterUpper after_upper) {
    (void)epoch;
    const int m = lu.m;
    lu_stage_and_solve<NRHS, true>(lu, sh, lu.u_cstart, lu.u_clen, lu.u_crow, lu.u_cval, lu.u_rstart[m], n_updates > 0 ? lu.slot_of : nullptr, lu.diag, 2);
    if (n_updates > 0) {
        // right-hand side of the trailing solve: v_T - z_b S, one wave per spike column (lower_upper/mod.rs:373-397)
        const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE, nwaves = blockDim.x / WAVE;
        for (int k = wave; k < n_updates; k += nwaves) {
            const int pos = lu.trail_pos[k];
            double p0 = 0.0, p1 = 0.0;
            if (pos >= 0) {
                const int cs = lu.s_cstart[k], cl = lu.s_clen[k];
                for (int e = lane; e < cl; e += WAVE) {
                    const int i = lu.s_crow[cs + e];
                    const double v = lu.s_cval[cs + e];
                    p0 += v * sh.x0[i];
                    if (NRHS == 2) p1 += v * sh.x1[i];
                }
            }
            p0 = wave_sum(p0);
            if (NRHS == 2) p1 = wave_sum(p1);
            if (lane == LAST) {
                sh.st0[k] = pos >= 0 ? sh.x0[pos] - p0 : 0.0;
                if (NRHS == 2) sh.st1[k] = pos >= 0 ? sh.x1[pos] - p1 : 0.0;
            }
        }
        __syncthreads();
        if (threadIdx.x < WAVE) {  // z_T T = s_T, forward substitution by slot
            const int pos = lane < n_updates ? lu.trail_pos[lane] : -1;
            double s0 = lane < n_updates ? sh.st0[lane] : 0.0;
            double s1 = (NRHS == 2 && lane < n_updates) ? sh.st1[lane] : 0.0;
            const double dk = pos >= 0 ? lu.diag[pos] : 1.0;
            for (int a = 0; a < n_updates; ++a) {
                if (lane == a) {
                    s0 = s0 / dk;
                    if (NRHS == 2) s1 = s1 / dk;
                }
                const double z0 = lane_value(s0, a);
                const double z1 = NRHS == 2 ? lane_value(s1, a) : 0.0;
                if (lane > a && lane < n_updates && (z0 != 0.0 || z1 != 0.0)) {
                    const double tv = sh.T[a * lu.ldt + lane];
                    s0 -= z0 * tv;
                    if (NRHS == 2) s1 -= z1 * tv;
                }
            }
            if (pos >= 0) {
                sh.x0[pos] = s0;
                if (NRHS == 2) sh.x1[pos] = s1;
            }
        }
        __syncthreads();
    }
    lu_stamp(sh, 7);
    after_upper();
    lu_stamp(sh, 8);
    if (n_updates > 0) {
        apply_etas_backward<NRHS>(lu, n_updates, sh.x0, sh.x1);
        __syncthreads();
    }
    lu_stamp(sh, 9);
    lu_stage_and_solve<NRHS, false>(lu, sh, lu.l_cstart, nullptr, lu.l_crow, lu.l_cval, lu.l_rstart[m], nullptr, nullptr, 3);
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
// moves to the end of the logical order -- i.e. it takes the next slot of the trailing block.  `eta_count` entries were
// already written by lu_build_eta.
__device__ __forceinline__ void lu_ft_update_block(const DeviceLU& lu, const LuShared& sh, const int t, const int eta_count,
                                                   const double new_diag, const double* spike) {
    const int m = lu.m;
    const int tid = threadIdx.x, T = blockDim.x;
    const int n_updates = lu.state[LU_N_UPDATES];
    const int top = lu.state[LU_S_TOP];
    const int eta_top = lu.state[LU_ETA_TOP];
    const int old_slot = lu.slot_of[t];
    const int new_slot = n_updates;
    const int stride = lu.max_updates;
    __syncthreads();  // everyone has read the state words
    if (old_slot < 0) {
        // 1. row t leaves: its base entries leave the base columns, its spike entries leave the spike columns
        {
            const int rs = lu.u_rstart[t], rl = lu.u_rlen[t];
            for (int e = tid; e < rl; e += T) {
                const int j = lu.u_rcol[rs + e];
                const int cs = lu.u_cstart[j], cl = lu.u_clen[j];
                for (int s = 0; s < cl; ++s)
                    if (lu.u_crow[cs + s] == t) {
                        lu.u_crow[cs + s] = lu.u_crow[cs + cl - 1];
                        lu.u_cval[cs + s] = lu.u_cval[cs + cl - 1];
                        lu.u_clen[j] = cl - 1;
                        break;
                    }
            }
            const int al = lu.app_len[t];
            for (int e = tid; e < al; e += T) {
                const int k = lu.app_slot[t * stride + e];
                const int cs = lu.s_cstart[k], cl = lu.s_clen[k];
                for (int s = 0; s < cl; ++s)
                    if (lu.s_crow[cs + s] == t) {
                        lu.s_crow[cs + s] = lu.s_crow[cs + cl - 1];
                        lu.s_cval[cs + s] = lu.s_cval[cs + cl - 1];
                        lu.s_clen[k] = cl - 1;
                        break;
                    }
            }
        }
        // 2. the old column t leaves the base rows
        {
            const int cs = lu.u_cstart[t], cl = lu.u_clen[t];
            for (int e = tid; e < cl; e += T) {
                const int i = lu.u_crow[cs + e];
                const int rs = lu.u_rstart[i], rl = lu.u_rlen[i];
                for (int s = 0; s < rl; ++s)
                    if (lu.u_rcol[rs + s] == t) {
                        lu.u_rcol[rs + s] = lu.u_rcol[rs + rl - 1];
                        lu.u_rval[rs + s] = lu.u_rval[rs + rl - 1];
                        lu.u_rlen[i] = rl - 1;
                        break;
                    }
            }
        }
    } else {
        // t was replaced before: its row and column live in T and S under `old_slot`, which dies
        for (int b = tid; b < lu.max_updates; b += T) {
            lu.T[old_slot * lu.ldt + b] = 0.0;   // row of T (the u_bar part)
            lu.T[b * lu.ldt + old_slot] = 0.0;   // column of T
        }
        const int cs = lu.s_cstart[old_slot], cl = lu.s_clen[old_slot];
        for (int e = tid; e < cl; e += T) {
            const int i = lu.s_crow[cs + e];
            const int al = lu.app_len[i];
            for (int s = 0; s < al; ++s)
                if (lu.app_slot[i * stride + s] == old_slot) {
                    lu.app_slot[i * stride + s] = lu.app_slot[i * stride + al - 1];
                    lu.app_val[i * stride + s] = lu.app_val[i * stride + al - 1];
                    lu.app_len[i] = al - 1;
                    break;
                }
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (old_slot < 0) {
            lu.u_rlen[t] = 0;
            lu.app_len[t] = 0;
            lu.u_clen[t] = 0;
        } else {
            lu.s_clen[old_slot] = 0;
            lu.trail_pos[old_slot] = -1;
        }
    }
    // 3. the spike becomes the column of the new slot: base rows -> S (column arena + one appended entry per row),
    //    live trailing positions -> T
    const int count = ordered_compact(
        m, sh.group_count, [&](int i) { return i != t && spike[i] != 0.0 && lu.slot_of[i] < 0; },
        [&](int i, int offset) {
            const double v = spike[i];
            lu.s_crow[top + offset] = i;
            lu.s_cval[top + offset] = v;
            const int a = i * stride + lu.app_len[i];
            lu.app_slot[a] = new_slot;
            lu.app_val[a] = v;
            lu.app_len[i] += 1;
        });
    for (int i = tid; i < m; i += T) {
        const int b = lu.slot_of[i];
        if (i != t && b >= 0 && b != old_slot) lu.T[b * lu.ldt + new_slot] = spike[i];
    }
    __syncthreads();
    if (tid == 0) {
        lu.s_cstart[new_slot] = top;
        lu.s_clen[new_slot] = count;
        lu.trail_pos[new_slot] = t;
        lu.slot_of[t] = new_slot;
        lu.diag[t] = new_diag;
        lu.eta_pivot[n_updates] = t;
        lu.eta_start[n_updates + 1] = eta_top + eta_count;
        lu.state[LU_N_UPDATES] = n_updates + 1;
        lu.state[LU_S_TOP] = top + count;
        lu.state[LU_ETA_TOP] = eta_top + eta_count;
        if (!(fabs(new_diag) > 0.0) || new_diag != new_diag) lu.state[LU_FLAGS] |= LU_FLAG_UNSTABLE;
    }
    __syncthreads();
}

// =====================================================================================================
// stand-alone kernels (fine-grained `BasisInverse` operations)
// =====================================================================================================
// dense != nullptr: dense right-hand side in original row order; else the sparse (rows, vals)
__global__ void __launch_bounds__(LU_THREADS) lu_ftran_kernel(DeviceLU lu, const int* rows, const double* vals, int nnz,
                                                               const double* dense, double* out, int keep_spike) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuShared sh = lu_shared(smem, m, lu.max_updates, (int)LU_LDS_TOTAL - 2048);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear(lu, sh, n_updates, false);
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
    const LuShared sh = lu_shared(smem, m, lu.max_updates, (int)LU_LDS_TOTAL - 2048);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear(lu, sh, n_updates, false);
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
    const LuShared sh = lu_shared(smem, m, lu.max_updates, (int)LU_LDS_TOTAL - 2048);
    const int t = lu.colpos[p];
    const int n_updates = lu.state[LU_N_UPDATES];
    if (n_updates >= lu.max_updates) {  // no room for another eta: the caller has to refactor
        if (threadIdx.x == 0) lu.state[LU_FLAGS] |= LU_FLAG_OVERFLOW;
        return;
    }
    lu_clear(lu, sh, n_updates, false);
    if (threadIdx.x == 0) sh.x0[t] = 1.0;
    __syncthreads();
    // y = e_t' U^-1: the U stage of a BTRAN (mod.rs:373-397), then the eta from it.  (The etas and the L stage run too --
    // they cost nothing next to a second code path -- but the eta is taken right behind the U stage.)
    int epoch = 0;
    int eta_count = 0;
    double new_diag = 0.0;
    lu_btran_block<1>(lu, sh, n_updates, epoch, [&] { eta_count = lu_build_eta(lu, sh, t, lu.spike, &new_diag); });
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
    hipLaunchKernelGGL(lu_ftran_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, lu, rows, vals, nnz, (const double*)nullptr, out, keep_spike);
    check_launch("lu_ftran_kernel");
}
void launch_lu_ftran_dense(const DeviceLU& lu, const double* rhs, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_ftran_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, lu, (const int*)nullptr, (const double*)nullptr, 0, rhs, out, 0);
    check_launch("lu_ftran_kernel");
}
void launch_lu_btran(const DeviceLU& lu, const int* slots, const double* vals, int nnz, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_btran_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, lu, slots, vals, nnz, (const double*)nullptr, out);
    check_launch("lu_btran_kernel");
}
void launch_lu_btran_dense(const DeviceLU& lu, const double* in_slots, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_btran_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, lu, (const int*)nullptr, (const double*)nullptr, 0, in_slots, out);
    check_launch("lu_btran_kernel");
}
void launch_lu_update(const DeviceLU& lu, int p, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_update_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, lu, p);
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
    LuShared sh = lu_shared(smem, m, lu.max_updates, (int)LU_LDS_TOTAL - 2048);
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
    lu_clear(lu, sh, n_updates, false);
    for (int e = lp.col_start[q] + tid; e < lp.col_start[q + 1]; e += T) sh.x0[lu.rowpos[lp.row_index[e]]] = lp.value[e];
    __syncthreads();
    lu_stamp(sh, 1);
    int epoch = 0;
    lu_ftran_block(lu, sh, n_updates, epoch, lu.spike);
    // ---- alpha per basis slot (kept in x1), gamma_q, Harris pass 1 ----------------------------------------------------------
    // harris_delta < 0: the reference's ratio test (exact minimum, ties to the lowest leaving column; tableau/mod.rs:287-313)
    const bool textbook = harris_delta < 0.0;
    const double harris_slack = textbook ? 0.0 : harris_delta;
    double sumsq = 0.0, theta = INFINITY;
    for (int s = tid; s < m; s += T) {
        const double a = sh.x0[lu.colpos[s]];
        sh.x1[s] = a;
        lp.alpha[s] = a;
        sumsq += a * a;
        if (a > tol_pivot && !(skip_artificial_rows && lp.basis[s] < lp.n_art)) theta = fmin(theta, (fmax(lp.xB[s], 0.0) + harris_slack) / a);
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
                const double key = textbook ? 1.0 : a;
                if (hrank == RANK_NONE || key > hkey || (key == hkey && rk < hrank)) {
                    hkey = key;
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
    const LuShared sh = lu_shared(smem, m, lu.max_updates, (int)LU_LDS_TOTAL - 2048);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear(lu, sh, n_updates, false);
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
    const LuShared sh = lu_shared(smem, m, lu.max_updates, (int)LU_LDS_TOTAL - 2048);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear(lu, sh, n_updates, false);
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
    const LuShared sh = lu_shared(smem, m, lu.max_updates, (int)LU_LDS_TOTAL - 2048);
    const int n_updates = lu.state[LU_N_UPDATES];
    for (int j = lp.n_art + blockIdx.x; j < lp.n; j += gridDim.x) {
        if (lp.pos[j] >= 0) continue;
        __syncthreads();
        lu_clear(lu, sh, n_updates, false);
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
        hipExtLaunchKernelGGL((lu_pivot_kernel<RULE>), dim3(1), dim3(LU_THREADS), (std::uint32_t)lu_lds_bytes_for(lu.m, lu.max_updates), s, start, stop, 0,
                              d, lu, n_price_blocks, tol_pivot, harris_delta, skip_art, mode, refactor_period);
    else
        hipLaunchKernelGGL((lu_pivot_kernel<RULE>), dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, d, lu, n_price_blocks, tol_pivot,
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
    hipLaunchKernelGGL(lu_xb_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, d, lu);
    check_launch("lu_xb_kernel");
}
void launch_lu_pi(const DeviceLP& d, const DeviceLU& lu, hipStream_t s) {
    configure_lu_pivot_lds();
    hipLaunchKernelGGL(lu_pi_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, d, lu);
    check_launch("lu_pi_kernel");
}
void launch_lu_gamma(const DeviceLP& d, const DeviceLU& lu, hipStream_t s) {
    configure_lu_pivot_lds();
    const int blocks = std::max(1, std::min(512, d.n - d.n_art));
    hipLaunchKernelGGL(lu_gamma_kernel, dim3(blocks), dim3(LU_THREADS), lu_lds_bytes_for(lu.m, lu.max_updates), s, d, lu);
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
    : device_(device), m_(m), period_(std::min(refactor_period > 0 ? refactor_period : 31, LU_MAX_SLOTS - 1)), options_(options) {
    if (m < 1) throw std::invalid_argument("m < 1");
    if (!lu_fits_lds(m, period_ + 1)) throw std::invalid_argument("m too large for the LDS-resident LU solve with this refactor period");
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
    const int fl = flags();
    if (fl & LU_FLAG_OVERFLOW) {  // the kernel made no update: the host copy of the basis must not move either
        const int zero = 0;
        RELP_HIP(hipMemcpy(lu_.device().state + LU_FLAGS, &zero, sizeof(int), hipMemcpyHostToDevice));
        throw std::runtime_error("no room for another update: refactor first (should_refactor)");
    }
    columns_[pivot_row] = last_column_;
    have_spike_ = false;
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
    // logical order of the positions = the reference's rotated index: base positions in position order, then the live slots
    std::vector<int> slot_of = geti(d.slot_of, m), trail_pos = geti(d.trail_pos, d.max_updates);
    std::vector<int> rank(m, -1), seq;
    for (int i = 0; i < m; ++i)
        if (slot_of[i] < 0) { rank[i] = (int)seq.size(); seq.push_back(i); }
    for (int k = 0; k < n_updates; ++k)
        if (trail_pos[k] >= 0) { rank[trail_pos[k]] = (int)seq.size(); seq.push_back(trail_pos[k]); }
    if ((int)seq.size() != m) throw std::runtime_error("inconsistent slot bookkeeping");
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
    std::vector<int> ucs = geti(d.u_cstart, m), ucl = geti(d.u_clen, m);
    size_t used = 0;
    for (int j = 0; j < m; ++j) used = std::max(used, (size_t)ucs[j] + (size_t)ucl[j]);
    std::vector<int> ucrow = geti(d.u_crow, used);
    std::vector<double> ucval = getd(d.u_cval, used), diag = getd(d.diag, m);
    std::vector<int> scs = geti(d.s_cstart, d.max_updates), scl = geti(d.s_clen, d.max_updates), scrow = geti(d.s_crow, state[LU_S_TOP]);
    std::vector<double> scval = getd(d.s_cval, state[LU_S_TOP]), T = getd(d.T, (size_t)d.max_updates * d.ldt);
    f.u_start.assign(m + 1, 0);
    f.upper_diagonal.resize(m);
    for (int c = 0; c < m; ++c) {
        const int j = seq[c];
        f.upper_diagonal[c] = diag[j];
        std::vector<std::pair<int, double>> col;
        if (slot_of[j] < 0) {
            for (int e = ucs[j]; e < ucs[j] + ucl[j]; ++e) col.push_back({rank[ucrow[e]], ucval[e]});
        } else {
            const int k = slot_of[j];
            for (int e = scs[k]; e < scs[k] + scl[k]; ++e) col.push_back({rank[scrow[e]], scval[e]});
            for (int a2 = 0; a2 < k; ++a2)
                if (trail_pos[a2] >= 0 && T[(size_t)a2 * d.ldt + k] != 0.0) col.push_back({rank[trail_pos[a2]], T[(size_t)a2 * d.ldt + k]});
        }
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
