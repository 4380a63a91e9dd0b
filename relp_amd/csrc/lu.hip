// LU basis factorisation on the device: FTRAN, BTRAN and the Forrest-Tomlin update as single-workgroup kernels for gfx950.
//
// Replaces (paths relative to /root/reference/src/algorithm/two_phase/tableau/inverse_maintenance/carry/lower_upper/):
//   left_multiply_by_basis_inverse  (FTRAN)     mod.rs:180-210  (+ left_multiply_by_{lower,upper}_inverse :286-321)
//   right_multiply_by_basis_inverse (BTRAN)     mod.rs:212-237  (+ right_multiply_by_{upper,lower}_inverse :347-397)
//   basis_inverse_row                           mod.rs:254-272
//   change_basis (Forrest-Tomlin)               mod.rs:94-178
//   EtaFile::{apply_right, apply_left, update_spike_pivot_value}   eta_file.rs:49-134
//
// Design (MI355X): one LP's factor is a few 10^4 non-zeros; a triangular solve with it is a dependency DAG, not a stream, and
// its cost is the LENGTH of its longest chain times the latency of one hop.  ONE workgroup (16 waves on one CU) owns a solve
// with the vector(s) in LDS, where a hop is a ds_read (~100 cycles) -- across workgroups a hop is an L2 / fabric round trip
// (0.8-1 us, MI355X_MICROARCH.md "handoff"), ten times as much, so more CUs would only make a Netlib-sized solve slower.  The
// solve is SYNCHRONISATION-FREE (`lu_solve_tasks`): every row of the factor is a task owned by a lane (or, a long row, by a
// wave); a task polls its operands in LDS -- an unsolved component holds a sentinel NaN -- and publishes its own component
// the moment the last operand arrives.  No barrier and no level loop inside a solve: a level costs one LDS round trip
// instead of the ~1 k cycles of round 2's barrier-per-level schedule.  The host's refactorisation lays the tasks out in
// dependency order (`LuTasks`, lu.hpp).  Forrest-Tomlin updates do not touch the task lists: a replaced position is
// bordered into a dense trailing block T (<= 64 x 64) that one wave solves out of registers, and its old row and column are
// MASKED, not removed (lu.hpp).  The reference walks ordered maps (`BTreeMap`) with a column scan per popped entry; here both
// orientations of L and U are resident so every solve is a gather.  Everything is deterministic: a row adds its entries in
// storage order, reductions have a fixed tree.
#include "lu.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>
#include <chrono>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>

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

struct LuLayout {
    size_t o_rowpos, o_colpos, o_lrs, o_lcs, o_urs, o_ucs, o_diag, o_counts, small_bytes;
    struct TaskOffsets {
        size_t z_pos, z_dinv, s_pos, s_lev, s_flags, s_dinv, s_xstart, s_xn, chunk, s_col, s_val, x_idx, x_val;
    } to[4];
    size_t o_lrcol, o_lcrow, o_lrval, o_lcval, o_urcol, o_ucrow, o_urval, o_ucval;
    struct CompactOffsets {
        size_t hdr, col, val, zpos;
    } co[4];
    size_t compact_begin, compact_end, upload_bytes;
    size_t app, o_applen, o_appslot, o_appval, o_scs, o_scl, o_scrow, o_scval, o_T, o_trail, o_slotof, o_eta_start, o_eta_pivot, o_eta_idx, o_eta_val;
    size_t o_eta_mf, o_eta_of, o_eta_first, o_eta_prev, o_eapp_len, o_eapp_eta, o_eapp_val, o_spike, pf_ld, o_pf_m, o_pf_slot, o_pf_col_of, o_state, o_log_factor, o_log_p, o_log_w, o_log_plan;
    size_t device_bytes;
};

__global__ void __launch_bounds__(256) lu_init_kernel(DeviceLU lu) {
    const int m = lu.m;
    const int stride = blockDim.x * gridDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = first; i < m; i += stride) {
        lu.app_len[i] = 0;
        lu.slot_of[i] = -1;
        lu.eta_of_pos[i] = -1;
        lu.eta_first[i] = -1;
        lu.eapp_len[i] = 0;
        if (lu.pf_col_of) lu.pf_col_of[i] = -1;
    }
    for (int i = first; i < lu.max_updates * lu.ldt; i += stride) {
        lu.T[i] = 0.0;
        lu.eta_mf[i] = 0.0;
    }
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
        lu.state[LU_PF_COUNT] = 0;
        lu.state[LU_LOG_ON] = 0;
        lu.state[LU_LOG_COUNT] = 0;
        lu.eta_start[0] = 0;
    }
}

// ---- the refactorisation beside the pivots (round 5) ---------------------------------------------------------------------------
__global__ void lu_start_log_kernel(DeviceLU lu) {
    lu.state[LU_LOG_COUNT] = 0;
    lu.state[LU_LOG_ON] = 1;
}
// The etas logged on `old` folded into the product form of `fresh` (factors of the basis at the start of the log; its M is the
// identity): for every logged pivot (row factors f, slot p)  M[s][c] -= f_s M[p][c]  over the kept columns, and a new column
// e_p - f for p when it has none -- the same arithmetic, in the same order, as the pivot kernel's fold.
// Round 5 did this as the pivot kernel does, one workgroup walking eta after eta over the kept columns in global memory: 16 us per eta,
// 0.5 ms for 32 of them -- what the refactorisation beside the pivots was meant to hide.  Round 6: row s of M needs of the other rows
// only  w_e = row p_e of M as it stands BEFORE eta e  (M[s][:] = M0[s][:] - f_1[s] w_1 - f_2[s] w_2 - ...), and the w_e need only the
// rows p_1, p_2, ... themselves.  So: (plan) one workgroup walks the at most LU_LOG_CAPACITY rows p_e through the etas in LDS (a thread
// per entry, one barrier per eta), leaves the w_e and the new columns' slots behind; (rows) then every row of M is independent of the
// others -- a thread per row, the w_e in LDS, the row's factors in flight eight at a time.  (A kept column that eta e finds not yet
// made has w_e = 0 in it: its subtraction changes nothing, so all the columns are treated as there from the start.)
constexpr int LU_REPLAY_THREADS = 1024;
constexpr int LU_REPLAY_PER_THREAD = (LU_LOG_CAPACITY * LU_MAX_SLOTS + LU_REPLAY_THREADS - 1) / LU_REPLAY_THREADS;
__global__ void __launch_bounds__(LU_REPLAY_THREADS) lu_replay_plan_kernel(DeviceLU fresh, DeviceLU old) {
    __shared__ double s_w[2][LU_MAX_SLOTS];
    __shared__ double s_f[LU_LOG_CAPACITY][LU_LOG_CAPACITY];  // s_f[e][u] = f_e at the u-th distinct logged slot
    __shared__ int s_p[LU_LOG_CAPACITY], s_have[LU_LOG_CAPACITY], s_row_of[LU_LOG_CAPACITY], s_row_slot[LU_LOG_CAPACITY], s_slot_of_col[LU_MAX_SLOTS];
    __shared__ int s_k0, s_k, s_rows, s_count;
    const int tid = threadIdx.x;
    const int logged = min(old.state[LU_LOG_COUNT], LU_LOG_CAPACITY);
    if (tid < logged) {
        s_p[tid] = old.log_p[tid];
        s_have[tid] = fresh.pf_col_of[s_p[tid]];
    }
    if (tid < LU_MAX_SLOTS) s_slot_of_col[tid] = tid < fresh.state[LU_PF_COUNT] ? fresh.pf_slot[tid] : -1;
    __syncthreads();
    if (tid == 0) {  // the kept columns the etas make, and the distinct slots among the p_e
        const int k0 = fresh.state[LU_PF_COUNT];
        int k = k0, rows = 0, count = 0;
        for (int e = 0; e < logged; ++e) {
            const int p = s_p[e];
            int u = -1;
            for (int b = 0; b < rows; ++b)
                if (s_row_slot[b] == p) u = b;
            if (u < 0) {
                if (s_have[e] < 0) {
                    if (k >= min(fresh.max_updates, LU_MAX_SLOTS)) break;  // (no room: the etas from here on are left out, and the caller's guard sees it)
                    s_slot_of_col[k++] = p;
                }
                u = rows++;
                s_row_slot[u] = p;
            }
            s_row_of[e] = u;
            count = e + 1;
        }
        s_k0 = k0;
        s_k = k;
        s_rows = rows;
        s_count = count;
    }
    __syncthreads();
    const int k0 = s_k0, k = s_k, rows = s_rows, count = s_count;
    for (int i = tid; i < count * rows; i += LU_REPLAY_THREADS) {
        const int e = i / rows, u = i % rows;
        s_f[e][u] = old.log_factor[(size_t)e * old.pf_ld + s_row_slot[u]];
    }
    // entry (u, c) of the walked rows: thread tid holds the entries tid, tid + LU_REPLAY_THREADS, ...
    double held[LU_REPLAY_PER_THREAD];
#pragma unroll
    for (int j = 0; j < LU_REPLAY_PER_THREAD; ++j) {
        const int i = tid + j * LU_REPLAY_THREADS, u = i / LU_MAX_SLOTS, c = i % LU_MAX_SLOTS;
        held[j] = 0.0;
        if (u < rows && c < k) held[j] = c < k0 ? fresh.pf_M[(size_t)c * fresh.pf_ld + s_row_slot[u]] : (s_slot_of_col[c] == s_row_slot[u] ? 1.0 : 0.0);
    }
    for (int e = 0; e < count; ++e) {
        const int at = s_row_of[e];
#pragma unroll
        for (int j = 0; j < LU_REPLAY_PER_THREAD; ++j) {
            const int i = tid + j * LU_REPLAY_THREADS, u = i / LU_MAX_SLOTS, c = i % LU_MAX_SLOTS;
            if (u == at) {
                s_w[e & 1][c] = held[j];
                fresh.log_w[(size_t)e * LU_MAX_SLOTS + c] = held[j];
            }
        }
        __syncthreads();  // (the other half of s_w is written by the next eta: one barrier per eta)
#pragma unroll
        for (int j = 0; j < LU_REPLAY_PER_THREAD; ++j) {
            const int i = tid + j * LU_REPLAY_THREADS, u = i / LU_MAX_SLOTS, c = i % LU_MAX_SLOTS;
            if (u < rows) held[j] = held[j] - s_f[e][u] * s_w[e & 1][c];
        }
    }
    if (tid >= k0 && tid < k) {
        fresh.pf_slot[tid] = s_slot_of_col[tid];
        fresh.pf_col_of[s_slot_of_col[tid]] = tid;
    }
    if (tid == 0) {
        fresh.log_plan[0] = k0;
        fresh.log_plan[1] = k;
        fresh.log_plan[2] = count;
        fresh.state[LU_PF_COUNT] = k;
        fresh.state[LU_N_UPDATES] = count;
    }
}
constexpr int LU_REPLAY_ROW_THREADS = 256;
__global__ void __launch_bounds__(LU_REPLAY_ROW_THREADS) lu_replay_rows_kernel(DeviceLU fresh, DeviceLU old) {
    __shared__ double s_w[LU_LOG_CAPACITY][LU_MAX_SLOTS];
    __shared__ int s_slot_of_col[LU_MAX_SLOTS];
    const int tid = threadIdx.x, m = fresh.m;
    const int k0 = fresh.log_plan[0], k = fresh.log_plan[1], count = fresh.log_plan[2];
    for (int i = tid; i < LU_LOG_CAPACITY * LU_MAX_SLOTS; i += LU_REPLAY_ROW_THREADS) {
        const int e = i / LU_MAX_SLOTS, c = i % LU_MAX_SLOTS;
        s_w[e][c] = (e < count && c < k) ? fresh.log_w[i] : 0.0;
    }
    if (tid < LU_MAX_SLOTS) s_slot_of_col[tid] = tid < k ? fresh.pf_slot[tid] : -1;
    __syncthreads();
    const int s = blockIdx.x * LU_REPLAY_ROW_THREADS + tid;
    if (s >= m) return;
    const double* f = old.log_factor + s;
    const size_t ld = old.pf_ld;
    for (int c0 = 0; c0 < k; c0 += 8) {
        double acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            acc[u] = c < k0 ? fresh.pf_M[(size_t)c * fresh.pf_ld + s] : (c < k && s_slot_of_col[c] == s ? 1.0 : 0.0);
        }
        for (int e0 = 0; e0 < count; e0 += 8) {
            double fe[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) fe[v] = e0 + v < count ? f[(size_t)(e0 + v) * ld] : 0.0;
#pragma unroll
            for (int v = 0; v < 8; ++v)
                if (e0 + v < count) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] = acc[u] - fe[v] * s_w[e0 + v][(c0 + u) & (LU_MAX_SLOTS - 1)];
                }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + u < k) fresh.pf_M[(size_t)(c0 + u) * fresh.pf_ld + s] = acc[u];
    }
}

// One orientation of one triangular factor -> task list (lu.hpp `LuTasks`).  `start` / `idx` / `val`: its rows (CSR over the
// position space); `diag`: nullptr for L; `lev_start` / `order`: the rows in dependency-level order (lu_schedules: a row of
// level l reads rows of lower levels only).  Inside a level the rows are packed widest group first, so only the first group of
// a level can need padding slots to reach its alignment.
struct HostTasks {
    std::vector<int> z_pos, s_pos, s_lev, s_flags, s_xstart, s_xn, x_idx, chunk;
    std::vector<double> z_dinv, s_dinv, x_val;
    std::vector<std::vector<int>> col;      // [LU_TE][ns]
    std::vector<std::vector<double>> val;
    int levels = 0;
};
int group_log2(int n) {  // smallest g with LU_TE << g >= n, at most 6
    int g = 0;
    while (g < 6 && (LU_TE << g) < n) ++g;
    return g;
}
void build_tasks(const int* start, const int* idx, const double* val, const double* diag, const std::vector<int>& lev_start,
                 const std::vector<int>& order, HostTasks& out) {
    // (the vectors keep their capacity from one refactorisation to the next: thousands of push_backs, no allocation)
    out.z_pos.clear(); out.s_pos.clear(); out.s_lev.clear(); out.s_flags.clear(); out.s_xstart.clear(); out.s_xn.clear();
    out.x_idx.clear(); out.chunk.clear(); out.z_dinv.clear(); out.s_dinv.clear(); out.x_val.clear();
    out.col.resize(LU_TE);
    out.val.resize(LU_TE);
    for (int e = 0; e < LU_TE; ++e) {
        out.col[e].clear();
        out.val[e].clear();
    }
    out.levels = (int)lev_start.size() - 1;
    auto push_slot = [&](int pos, int lev, int flags, double dinv, int xstart, int xn) {
        out.s_pos.push_back(pos);
        out.s_lev.push_back(lev);
        out.s_flags.push_back(flags);
        out.s_dinv.push_back(dinv);
        out.s_xstart.push_back(xstart);
        out.s_xn.push_back(xn);
        for (int e = 0; e < LU_TE; ++e) {
            out.col[e].push_back(0);
            out.val[e].push_back(0.0);
        }
    };
    // chunk under construction: [first slot, first level); closed at a level boundary -- or inside a level wider than a chunk,
    // whose rows are independent of each other
    int chunk_first = 0, chunk_level = 1;
    auto close_chunk = [&](int end_level, int next_level) {
        if ((int)out.s_pos.size() > chunk_first) {
            // tail: the levels whose slots all lie in the chunk's last wave (LDS keeps one wave's accesses in order: no barrier needed)
            const int end = (int)out.s_pos.size();
            const int last_wave_first = chunk_first + ((end - 1 - chunk_first) / WAVE) * WAVE;
            int tail = end_level;
            if (last_wave_first > chunk_first || end - chunk_first <= WAVE) {
                tail = out.s_lev[last_wave_first];
                if (last_wave_first > chunk_first && out.s_lev[last_wave_first - 1] == tail) ++tail;  // that level starts in an earlier wave
            }
            tail = std::max(chunk_level, std::min(tail, end_level));
            out.chunk.insert(out.chunk.end(), {chunk_first, end, chunk_level, end_level, tail, 0, 0, 0});
            while ((int)out.s_pos.size() % WAVE) push_slot(0, 0x7fffffff, 0, 1.0, 0, 0);  // the next chunk starts on a wave boundary
        }
        chunk_first = (int)out.s_pos.size();
        chunk_level = next_level;
    };
    std::vector<int> rows;
    for (int l = 0; l < out.levels; ++l) {
        rows.clear();
        for (int r = lev_start[l]; r < lev_start[l + 1]; ++r) {
            const int i = order[r];
            if (start[i + 1] == start[i]) {
                out.z_pos.push_back(i);
                out.z_dinv.push_back(diag ? 1.0 / diag[i] : 1.0);
            } else {
                rows.push_back(i);
            }
        }
        if (rows.empty()) continue;
        std::stable_sort(rows.begin(), rows.end(), [&](int a, int b) { return group_log2(start[a + 1] - start[a]) > group_log2(start[b + 1] - start[b]); });
        int level_slots = 0;  // (an upper bound with the alignment padding of the first group)
        for (int i : rows) level_slots += 1 << group_log2(start[i + 1] - start[i]);
        if ((int)out.s_pos.size() - chunk_first + level_slots + WAVE > LU_CHUNK_SLOTS) close_chunk(l, l);
        for (int i : rows) {
            const int n = start[i + 1] - start[i];
            const int g = group_log2(n), G = 1 << g;
            if ((int)out.s_pos.size() - chunk_first + 2 * G > LU_CHUNK_SLOTS) {  // a level wider than a chunk: continue it in the next one
                close_chunk(l + 1, l);
            }
            while ((int)out.s_pos.size() % G) push_slot(0, l, 0, 1.0, 0, 0);  // padding: a slot of this level that does nothing
            const int first = (int)out.s_pos.size();
            const int inline_n = std::min(n, LU_TE * G);
            const int xn = n - inline_n;
            const int xstart = (int)out.x_idx.size();
            for (int j = 0; j < G; ++j) {
                // slot j of the group takes the entries j, j + G, j + 2 G, ... of the first LU_TE * G
                push_slot(i, l, g | ((j == G - 1) ? 1 << 8 : 0) | (xn > 0 ? 1 << 9 : 0), diag ? 1.0 / diag[i] : 1.0, xstart, xn);
                int mine = 0;
                for (int e = j; e < inline_n; e += G) {
                    out.col[mine][first + j] = idx[start[i] + e];
                    out.val[mine][first + j] = val[start[i] + e];
                    ++mine;
                }
            }
            for (int e = inline_n; e < n; ++e) {
                out.x_idx.push_back(idx[start[i] + e]);
                out.x_val.push_back(val[start[i] + e]);
            }
        }
    }
    close_chunk(out.levels, out.levels);
    if ((int)out.chunk.size() / 8 > LU_MAX_CHUNKS) throw std::runtime_error("LU task list: too many chunks");
}
// The slots of one orientation of an INVERTED factor (lu.hpp): no levels and no chunks -- a product reads its input only, so the
// rows are simply packed widest group first (which keeps every group aligned to its size without padding slots) and thread t of
// the product takes the slots t, t + 1024, ...  Sizes are known from the row lengths alone (`count_inverse_slots`), so the compact
// records are written straight into the staging buffer (`fill_inverse_records`): the general builder above appends slot by slot
// into vectors that are copied afterwards -- 0.18 ms per refactorisation of 25FV47 for the four inverse lists, 0.38 of GREENBEA.
struct InverseListInfo {
    int slots = 0, empty_rows = 0;
    size_t extras = 0;
    int first_of[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // first slot of the rows with group size 2^g
};
InverseListInfo count_inverse_slots(const int m, const int* start) {
    InverseListInfo info;
    int count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < m; ++i) {
        const int n = start[i + 1] - start[i];
        if (n == 0) {
            ++info.empty_rows;
            continue;
        }
        count[group_log2(n)]++;
        if (n > LU_TE * 64) info.extras += (size_t)(n - LU_TE * 64);
    }
    for (int g = 6; g >= 0; --g) {
        info.first_of[g] = info.slots;
        info.slots += count[g] << g;
    }
    return info;
}
void fill_inverse_records(const int m, const int* start, const int* idx, const double* val, InverseListInfo info, const int stride,
                          unsigned int* hdr, unsigned long long* col, double* values, int* zpos, int* xstart, int* xn, int* x_idx,
                          double* x_val) {
    static_assert(LU_TE == 4, "the compact records of the inverse-factor form hold four entries per slot");
    const int slots = info.slots;
    std::fill(hdr, hdr + stride, 0u);   // padding: no group, no write ...
    std::fill(col, col + stride, 0ull);  // ... operands at position 0 (values: whatever -- a padding slot writes nowhere)
    std::memset(values, 0, (size_t)4 * stride * sizeof(double));  // the padding records too: their values reach the FMAs of the product (header 0 only suppresses the write)
    int nz = 0;
    size_t nx = 0;
    for (int i = 0; i < m; ++i) {
        const int n = start[i + 1] - start[i];
        if (n == 0) {
            zpos[nz++] = i;
            continue;
        }
        const int g = group_log2(n), G = 1 << g;
        const int first = info.first_of[g];
        info.first_of[g] += G;
        const int inline_n = std::min(n, LU_TE * G);
        const int extra = n - inline_n;
        for (int j = 0; j < G; ++j) {
            hdr[first + j] = (unsigned)i | ((unsigned)g << 16) | ((j == G - 1) ? 1u << 19 : 0u) | (extra > 0 ? 1u << 20 : 0u);
            if (extra > 0) {
                xstart[first + j] = (int)nx;
                xn[first + j] = extra;
            }
        }
        const int* ri = idx + start[i];
        const double* rv = val + start[i];
        for (int e = 0; e < inline_n; ++e) {  // slot j of the group takes the entries j, j + G, j + 2 G, ...
            const int slot = first + e % G, lane_entry = e / G;
            col[slot] |= (unsigned long long)(unsigned)ri[e] << (16 * lane_entry);
            values[4 * (size_t)slot + lane_entry] = rv[e];
        }
        for (int e = inline_n; e < n; ++e) {
            x_idx[nx] = ri[e];
            x_val[nx++] = rv[e];
        }
    }
    std::fill(zpos + nz, zpos + stride, 0);
    for (int w0 = 0; w0 < slots; w0 += WAVE) {  // the wave summaries: bits 21-26 group sizes present, bit 27 extras present
        const int w1 = std::min(slots, w0 + WAVE);
        unsigned summary = 0;
        for (int k = w0; k < w1; ++k) {
            const int g = (int)(hdr[k] >> 16) & 7;
            for (int j = 0; j < 6; ++j)
                if (g > j) summary |= 1u << (21 + j);
            if ((hdr[k] >> 20) & 1u) summary |= 1u << 27;
        }
        for (int k = w0; k < w1; ++k) hdr[k] |= summary;
    }
}

// ---- device layout of one factorisation: offsets into ONE allocation, by capacities only (so that the addresses -- and a captured
// hipGraph that holds them -- survive a refactorisation).  The uploaded prefix first, then the arrays only kernels touch.
LuLayout compute_layout(int m, int max_updates, bool inverse_factors, size_t cl, size_t cu, int stride) {
    LuLayout L;
    Carver c;
    L.o_rowpos = c.take<int>(m); L.o_colpos = c.take<int>(m);
    L.o_lrs = c.take<int>(m + 1); L.o_lcs = c.take<int>(m + 1); L.o_urs = c.take<int>(m + 1); L.o_ucs = c.take<int>(m + 1);
    L.o_diag = c.take<double>(m);
    L.o_counts = c.take<int>(4 * LU_CNT_WORDS);
    for (int k = 0; k < 4; ++k) {
        L.to[k].z_pos = c.take<int>(stride);
        L.to[k].z_dinv = c.take<double>(stride);
        L.to[k].s_pos = c.take<int>(stride);
        L.to[k].s_lev = c.take<int>(stride);
        L.to[k].s_flags = c.take<int>(stride);
        L.to[k].s_dinv = c.take<double>(stride);
        L.to[k].s_xstart = c.take<int>(stride);
        L.to[k].s_xn = c.take<int>(stride);
        L.to[k].chunk = c.take<int>(8 * LU_MAX_CHUNKS);
    }
    L.small_bytes = c.offset;  // everything up to here goes in one copy
    for (int k = 0; k < 4; ++k) {
        const size_t cap = (k == 0 || k == 3) ? cl : cu;
        L.to[k].s_col = c.take<int>((size_t)LU_TE * stride);
        L.to[k].s_val = c.take<double>((size_t)LU_TE * stride);
        L.to[k].x_idx = c.take<int>(cap);
        L.to[k].x_val = c.take<double>(cap);
    }
    L.o_lrcol = c.take<int>(cl); L.o_lcrow = c.take<int>(cl); L.o_lrval = c.take<double>(cl); L.o_lcval = c.take<double>(cl);
    L.o_urcol = c.take<int>(cu); L.o_ucrow = c.take<int>(cu); L.o_urval = c.take<double>(cu); L.o_ucval = c.take<double>(cu);
    // the compact records of the inverse-factor form (lu.hpp), the four lists in ONE region (one copy per refactorisation)
    L.compact_begin = (c.offset + 63) & ~size_t(63);
    c.offset = L.compact_begin;
    for (int k = 0; k < 4; ++k) {
        L.co[k].hdr = c.take<unsigned int>(inverse_factors ? stride : 0);
        L.co[k].zpos = c.take<int>(inverse_factors ? stride : 0);
        L.co[k].col = c.take<unsigned long long>(inverse_factors ? stride : 0);
        c.offset = (c.offset + 31) & ~size_t(31);
        L.co[k].val = c.take<double>(inverse_factors ? (size_t)4 * stride : 0);
    }
    L.compact_end = c.offset;
    L.upload_bytes = c.offset;
    // ---- device only ----------------------------------------------------------------------------------------------------------
    const int ldt = max_updates + 1;
    const size_t app = (size_t)m * max_updates;
    L.app = app;
    L.o_applen = c.take<int>(m); L.o_appslot = c.take<int>(app); L.o_appval = c.take<double>(app);
    L.o_scs = c.take<int>(max_updates); L.o_scl = c.take<int>(max_updates); L.o_scrow = c.take<int>(app); L.o_scval = c.take<double>(app);
    L.o_T = c.take<double>((size_t)max_updates * ldt);
    L.o_trail = c.take<int>(max_updates); L.o_slotof = c.take<int>(m);
    L.o_eta_start = c.take<int>(max_updates + 2); L.o_eta_pivot = c.take<int>(max_updates + 1);
    L.o_eta_idx = c.take<int>(app); L.o_eta_val = c.take<double>(app);
    L.o_eta_mf = c.take<double>((size_t)max_updates * ldt); L.o_eta_of = c.take<int>(m); L.o_eta_first = c.take<int>(m); L.o_eta_prev = c.take<int>(max_updates + 1);
    L.o_eapp_len = c.take<int>(m); L.o_eapp_eta = c.take<int>(app); L.o_eapp_val = c.take<double>(app);
    L.o_spike = c.take<double>(m);
    L.pf_ld = ((size_t)m + 15) & ~(size_t)15;
    L.o_pf_m = c.take<double>(inverse_factors ? L.pf_ld * max_updates : 0);
    L.o_pf_slot = c.take<int>(max_updates); L.o_pf_col_of = c.take<int>(m);
    L.o_state = c.take<int>(LU_STATE_WORDS);
    L.o_log_factor = c.take<double>(inverse_factors ? L.pf_ld * LU_LOG_CAPACITY : 0);
    L.o_log_p = c.take<int>(LU_LOG_CAPACITY);
    L.o_log_w = c.take<double>(inverse_factors ? (size_t)LU_LOG_CAPACITY * LU_MAX_SLOTS : 0);
    L.o_log_plan = c.take<int>(4);
    L.device_bytes = c.offset;
    return L;
}
DeviceLU bind_layout(const LuLayout& L, char* dev_, int m, int max_updates, int inverse_vectors, int stride) {
    const bool inverse_factors = inverse_vectors != 0;
    const int ldt = max_updates + 1;
    DeviceLU d;
    d.m = m;
    d.max_updates = max_updates;
    d.ldt = ldt;
    auto I = [&](size_t o) { return reinterpret_cast<int*>(dev_ + o); };
    auto D = [&](size_t o) { return reinterpret_cast<double*>(dev_ + o); };
    d.rowpos = I(L.o_rowpos);
    d.colpos = I(L.o_colpos);
    d.l_rstart = I(L.o_lrs); d.l_rcol = I(L.o_lrcol); d.l_rval = D(L.o_lrval);
    d.l_cstart = I(L.o_lcs); d.l_crow = I(L.o_lcrow); d.l_cval = D(L.o_lcval);
    d.u_rstart = I(L.o_urs); d.u_rcol = I(L.o_urcol); d.u_rval = D(L.o_urval);
    d.u_cstart = I(L.o_ucs); d.u_crow = I(L.o_ucrow); d.u_cval = D(L.o_ucval);
    d.app_len = I(L.o_applen); d.app_slot = I(L.o_appslot); d.app_val = D(L.o_appval);
    d.s_cstart = I(L.o_scs); d.s_clen = I(L.o_scl); d.s_crow = I(L.o_scrow); d.s_cval = D(L.o_scval);
    d.s_capacity = (int)L.app;
    d.T = D(L.o_T);
    d.trail_pos = I(L.o_trail);
    d.slot_of = I(L.o_slotof);
    d.diag = D(L.o_diag);
    d.eta_start = I(L.o_eta_start); d.eta_pivot = I(L.o_eta_pivot); d.eta_idx = I(L.o_eta_idx); d.eta_val = D(L.o_eta_val);
    d.eta_capacity = (int)L.app;
    d.eta_mf = D(L.o_eta_mf); d.eta_of_pos = I(L.o_eta_of); d.eta_first = I(L.o_eta_first); d.eta_prev = I(L.o_eta_prev);
    d.eapp_len = I(L.o_eapp_len); d.eapp_eta = I(L.o_eapp_eta); d.eapp_val = D(L.o_eapp_val);
    d.spike = D(L.o_spike);
    d.state = I(L.o_state);
    d.task_stride = stride;
    d.inverse_factors = inverse_vectors;
    d.pf_M = inverse_factors ? D(L.o_pf_m) : nullptr;
    d.pf_ld = (int)L.pf_ld;
    d.pf_slot = I(L.o_pf_slot);
    d.pf_col_of = I(L.o_pf_col_of);
    d.log_factor = inverse_factors ? D(L.o_log_factor) : nullptr;
    d.log_p = I(L.o_log_p);
    d.log_w = inverse_factors ? D(L.o_log_w) : nullptr;
    d.log_plan = I(L.o_log_plan);
    for (int k = 0; k < 4; ++k) {
        LuTasks& t = d.tasks[k];
        auto GI = [&](size_t o) { return (lu_gptr_i32) reinterpret_cast<const int*>(dev_ + o); };
        auto GD = [&](size_t o) { return (lu_gptr_f64) reinterpret_cast<const double*>(dev_ + o); };
        t.z_pos = GI(L.to[k].z_pos); t.z_dinv = GD(L.to[k].z_dinv);
        t.s_pos = GI(L.to[k].s_pos); t.s_lev = GI(L.to[k].s_lev); t.s_flags = GI(L.to[k].s_flags); t.s_dinv = GD(L.to[k].s_dinv);
        t.s_xstart = GI(L.to[k].s_xstart); t.s_xn = GI(L.to[k].s_xn); t.chunk = GI(L.to[k].chunk);
        t.s_col = GI(L.to[k].s_col); t.s_val = GD(L.to[k].s_val);
        t.x_idx = GI(L.to[k].x_idx); t.x_val = GD(L.to[k].x_val);
        t.counts = GI(L.o_counts + (size_t)k * LU_CNT_WORDS * sizeof(int));
        if (inverse_factors) {
            t.c_hdr = (const __attribute__((address_space(1))) unsigned int*)reinterpret_cast<const unsigned int*>(dev_ + L.co[k].hdr);
            t.c_col = (const __attribute__((address_space(1))) unsigned long long*)reinterpret_cast<const unsigned long long*>(dev_ + L.co[k].col);
            t.c_val = GD(L.co[k].val);
            t.c_zpos = GI(L.co[k].zpos);
        }
    }
    return d;
}
}  // namespace

static int lu_inverse_vectors(int m, int max_updates);  // (below, with the LDS sizes)

bool LuFactors::upload(const HostLU& factors, int max_updates, hipStream_t stream, bool inverse_factors) {
    // The inverse-factor form uploads L^-1 and U^-1 through the same task lists (every row in one level; lu.hpp).
    thread_local HostLU inverted;
    if (inverse_factors) {
        const size_t limit = std::max<size_t>(1u << 16, 64 * (size_t)(factors.nnz_l() + factors.nnz_u() + factors.m));
        if (!lu_invert_factors(factors, std::min<size_t>(limit, (size_t)LU_MAX_CHUNKS * LU_CHUNK_SLOTS * LU_TE / 4), inverted))
            throw std::runtime_error("the inverse-factor carry: L^-1 and U^-1 of this basis are too dense (use the LU or the explicit carry)");
        nnz_l_inverse = inverted.nnz_l();
        nnz_u_inverse = inverted.nnz_u();
    }
    const HostLU& f = inverse_factors ? inverted : factors;
    const int m = f.m;
    const size_t nl = (size_t)f.nnz_l(), nu = (size_t)f.nnz_u();
    if (max_updates < 1) max_updates = 1;
    if (max_updates > LU_MAX_SLOTS) max_updates = LU_MAX_SLOTS;
    // The layout depends on capacities only, so that the device addresses (and a captured hipGraph that holds them) survive
    // a refactorisation; it changes when a factor outgrows its capacity (or m / the update capacity change).
    const int inverse_vectors = inverse_factors ? lu_inverse_vectors(m, max_updates) : 0;
    if (inverse_factors && inverse_vectors == 0) throw std::invalid_argument("the inverse-factor carry: too many rows for its vectors in LDS");
    bool layout_changed = m != d_.m || max_updates != d_.max_updates || inverse_vectors != d_.inverse_factors;
    if (layout_changed) cap_l_ = cap_u_ = 0;
    if (nl > cap_l_ || cap_l_ == 0) { cap_l_ = nl + nl / 2 + 256; layout_changed = true; }
    if (nu > cap_u_ || cap_u_ == 0) { cap_u_ = nu + nu / 2 + 256; layout_changed = true; }
    const size_t cl = cap_l_, cu = cap_u_;
    // ---- host: the four orientations and their task lists -----------------------------------------------------------------
    static const bool time_parts = diagnostic("RELP_TIME_REFACTOR");
    thread_local double part_seconds[5] = {0, 0, 0, 0, 0};  // (diagnostic sums per calling thread: handles of a batch refactorise concurrently)
    thread_local long long uploads = 0;
    auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = time_parts ? wall() : 0.0;
    auto mark = [&](int which) {
        if (!time_parts) return;
        const double now = wall();
        part_seconds[which] += now - t_mark;
        t_mark = now;
    };
    HostLU& fs = const_cast<HostLU&>(f);
    if (fs.lev_row[0].empty()) lu_schedules(fs);
    mark(0);
    std::vector<int> lcs(m + 1, 0), lcrow(nl), ucs(m + 1, 0), ucrow(nu);
    std::vector<double> lcval(nl), ucval(nu);
    {   // column orientation of L and U (counting transposes)
        for (size_t e = 0; e < nl; ++e) lcs[f.l_col[e] + 1]++;
        for (size_t e = 0; e < nu; ++e) ucs[f.u_col[e] + 1]++;
        for (int j = 0; j < m; ++j) {
            lcs[j + 1] += lcs[j];
            ucs[j + 1] += ucs[j];
        }
        std::vector<int> fill_l(lcs.begin(), lcs.end() - 1), fill_u(ucs.begin(), ucs.end() - 1);
        for (int i = 0; i < m; ++i) {
            for (int e = f.l_start[i]; e < f.l_start[i + 1]; ++e) {
                const int dst = fill_l[f.l_col[e]]++;
                lcrow[dst] = i;
                lcval[dst] = f.l_val[e];
            }
            for (int e = f.u_start[i]; e < f.u_start[i + 1]; ++e) {
                const int dst = fill_u[f.u_col[e]]++;
                ucrow[dst] = i;
                ucval[dst] = f.u_val[e];
            }
        }
    }
    mark(1);
    // (Built one after the other on the calling thread: helper threads were measured -- the thread start alone costs more than a
    // list does at Netlib sizes, 8.5 -> 11.9 ms over the 75 refactorisations of 25FV47.)
    thread_local HostTasks task_storage[4];  // (per calling thread: handles of a batch refactorise concurrently; capacity is kept)
    HostTasks* tasks = task_storage;
    InverseListInfo inverse_list[4];
    const int* list_start[4] = {f.l_start.data(), f.u_start.data(), ucs.data(), lcs.data()};
    const int* list_idx[4] = {f.l_col.data(), f.u_col.data(), ucrow.data(), lcrow.data()};
    const double* list_val[4] = {f.l_val.data(), f.u_val.data(), ucval.data(), lcval.data()};
    if (inverse_factors) {  // (U^-1 carries its diagonal as entries; the records are written into the staging buffer below)
        if (m > 65535) throw std::runtime_error("the inverse-factor carry: more than 65535 rows");
        for (int k = 0; k < 4; ++k) inverse_list[k] = count_inverse_slots(m, list_start[k]);
    } else {
        build_tasks(f.l_start.data(), f.l_col.data(), f.l_val.data(), nullptr, f.lev_start[0], f.lev_row[0], tasks[0]);
        build_tasks(f.u_start.data(), f.u_col.data(), f.u_val.data(), f.diag.data(), f.lev_start[1], f.lev_row[1], tasks[1]);
        build_tasks(ucs.data(), ucrow.data(), ucval.data(), f.diag.data(), f.lev_start[2], f.lev_row[2], tasks[2]);
        build_tasks(lcs.data(), lcrow.data(), lcval.data(), nullptr, f.lev_start[3], f.lev_row[3], tasks[3]);
    }
    mark(2);
    size_t max_slots = 0;
    for (int k = 0; k < 4; ++k) max_slots = std::max(max_slots, inverse_factors ? (size_t)inverse_list[k].slots : tasks[k].s_pos.size());
    if (max_slots + 1024 > cap_slots_ || (size_t)m + 1024 > cap_slots_ || cap_slots_ == 0 || layout_changed) {
        if (max_slots + 1024 > cap_slots_ || (size_t)m + 1024 > cap_slots_ || cap_slots_ == 0) layout_changed = true;
        max_slots = std::max(max_slots, (size_t)m);
        // (the inverse-factor lists are four times longer and every per-slot array is uploaded up to the stride: a tighter margin)
        const size_t margin = inverse_factors ? max_slots / 4 + 1024 : max_slots / 2 + 2048;
        cap_slots_ = std::max(cap_slots_, ((max_slots + margin) + 1023) & ~size_t(1023));
    }
    const int stride = (int)cap_slots_;
    // ---- uploaded prefix + device-only part: the layout (compute_layout above) ------------------------------------------------
    const LuLayout lay = compute_layout(m, max_updates, inverse_factors, cl, cu, stride);
    const size_t o_rowpos = lay.o_rowpos, o_colpos = lay.o_colpos, o_lrs = lay.o_lrs, o_lcs = lay.o_lcs, o_urs = lay.o_urs, o_ucs = lay.o_ucs;
    const size_t o_diag = lay.o_diag, o_counts = lay.o_counts, small_bytes = lay.small_bytes;
    const LuLayout::TaskOffsets* to = lay.to;
    const LuLayout::CompactOffsets* co = lay.co;
    const size_t o_lrcol = lay.o_lrcol, o_lcrow = lay.o_lcrow, o_lrval = lay.o_lrval, o_lcval = lay.o_lcval;
    const size_t o_urcol = lay.o_urcol, o_ucrow = lay.o_ucrow, o_urval = lay.o_urval, o_ucval = lay.o_ucval;
    const size_t compact_begin = lay.compact_begin, compact_end = lay.compact_end, upload_bytes = lay.upload_bytes, device_bytes = lay.device_bytes;
    {
        char* before = dev_;
        reserve(device_bytes, upload_bytes);
        if (dev_ != before) layout_changed = true;
    }

    char* h = staging_;
    auto put_i = [&](size_t at, const std::vector<int>& v) { if (!v.empty()) std::memcpy(h + at, v.data(), v.size() * sizeof(int)); };
    auto put_d = [&](size_t at, const std::vector<double>& v) { if (!v.empty()) std::memcpy(h + at, v.data(), v.size() * sizeof(double)); };
    put_i(o_rowpos, f.rowpos);
    put_i(o_colpos, f.colpos);
    put_i(o_lrs, f.l_start);
    put_i(o_lcs, lcs);
    put_i(o_urs, f.u_start);
    put_i(o_ucs, ucs);
    put_d(o_diag, f.diag);
    int* counts = reinterpret_cast<int*>(h + o_counts);
    for (int k = 0; k < 4; ++k) {
        const HostTasks& t = tasks[k];
        counts[k * LU_CNT_WORDS + LU_CNT_Z] = (int)t.z_pos.size();
        counts[k * LU_CNT_WORDS + LU_CNT_SLOTS] = (int)t.s_pos.size();
        counts[k * LU_CNT_WORDS + LU_CNT_LEVELS] = t.levels;
        counts[k * LU_CNT_WORDS + LU_CNT_CHUNKS] = (int)t.chunk.size() / 8;
        counts[k * LU_CNT_WORDS + LU_CNT_C0_END] = t.chunk.empty() ? 0 : t.chunk[1];
        counts[k * LU_CNT_WORDS + LU_CNT_C0_L0] = t.chunk.empty() ? 0 : t.chunk[2];
        counts[k * LU_CNT_WORDS + LU_CNT_C0_L1] = t.chunk.empty() ? 0 : t.chunk[3];
        counts[k * LU_CNT_WORDS + LU_CNT_C0_TAIL] = t.chunk.empty() ? 0 : t.chunk[4];
        // every per-slot array is written up to `stride`: a thread reads slot first + tid without knowing where the slots end
        auto put_i_padded = [&](size_t at, const std::vector<int>& v, int pad) {
            int* dst = reinterpret_cast<int*>(h + at);
            if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(int));
            std::fill(dst + v.size(), dst + stride, pad);
        };
        auto put_d_padded = [&](size_t at, const std::vector<double>& v, double pad) {
            double* dst = reinterpret_cast<double*>(h + at);
            if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(double));
            std::fill(dst + v.size(), dst + stride, pad);
        };
        if (inverse_factors) {
            const InverseListInfo& info = inverse_list[k];
            counts[k * LU_CNT_WORDS + LU_CNT_Z] = info.empty_rows;
            counts[k * LU_CNT_WORDS + LU_CNT_SLOTS] = info.slots;
            counts[k * LU_CNT_WORDS + LU_CNT_LEVELS] = 1;
            counts[k * LU_CNT_WORDS + LU_CNT_CHUNKS] = 1;
            counts[k * LU_CNT_WORDS + LU_CNT_C0_END] = info.slots;  // (one chunk, one level: nothing stale from another handle's task_storage)
            counts[k * LU_CNT_WORDS + LU_CNT_C0_L0] = 0;
            counts[k * LU_CNT_WORDS + LU_CNT_C0_L1] = 1;
            counts[k * LU_CNT_WORDS + LU_CNT_C0_TAIL] = 1;
            if (info.extras > ((k == 0 || k == 3) ? cl : cu)) throw std::runtime_error("LU inverse lists: extra entries exceed their capacity");
            fill_inverse_records(m, list_start[k], list_idx[k], list_val[k], info, stride, reinterpret_cast<unsigned int*>(h + co[k].hdr),
                                 reinterpret_cast<unsigned long long*>(h + co[k].col), reinterpret_cast<double*>(h + co[k].val),
                                 reinterpret_cast<int*>(h + co[k].zpos), reinterpret_cast<int*>(h + to[k].s_xstart),
                                 reinterpret_cast<int*>(h + to[k].s_xn), reinterpret_cast<int*>(h + to[k].x_idx),
                                 reinterpret_cast<double*>(h + to[k].x_val));
            continue;
        }
        put_i_padded(to[k].z_pos, t.z_pos, 0);
        put_d_padded(to[k].z_dinv, t.z_dinv, 1.0);
        put_i_padded(to[k].s_pos, t.s_pos, 0);
        put_i_padded(to[k].s_lev, t.s_lev, 0x7fffffff);
        put_i_padded(to[k].s_flags, t.s_flags, 0);
        put_d_padded(to[k].s_dinv, t.s_dinv, 1.0);
        put_i_padded(to[k].s_xstart, t.s_xstart, 0);
        put_i_padded(to[k].s_xn, t.s_xn, 0);
        put_i(to[k].chunk, t.chunk);
        for (int e = 0; e < LU_TE; ++e) {
            put_i_padded(to[k].s_col + (size_t)e * stride * sizeof(int), t.col[e], 0);
            put_d_padded(to[k].s_val + (size_t)e * stride * sizeof(double), t.val[e], 0.0);
        }
        put_i(to[k].x_idx, t.x_idx);
        put_d(to[k].x_val, t.x_val);
    }
    if (!inverse_factors) {  // (the plain orientations serve the Forrest-Tomlin update only)
        put_i(o_lrcol, f.l_col);
        put_d(o_lrval, f.l_val);
        put_i(o_lcrow, lcrow);
        put_d(o_lcval, lcval);
        put_i(o_urcol, f.u_col);
        put_d(o_urval, f.u_val);
        put_i(o_ucrow, ucrow);
        put_d(o_ucval, ucval);
    }
    // A small factor goes over in ONE copy, gaps and unused capacity included: every hipMemcpyAsync costs 5-10 us of host time,
    // which at Netlib sizes is more than the bytes do.  A large one copies the headers at once and each entry array up to what
    // is used (the capacities are 1.5 x larger, the ELL rows m wide).
    mark(3);
    if (inverse_factors) {  // the header (permutations, counts), the compact records, and the extras of the few very long rows
        RELP_HIP(hipMemcpyAsync(dev_, h, o_counts + 4 * LU_CNT_WORDS * sizeof(int), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(dev_ + compact_begin, h + compact_begin, compact_end - compact_begin, hipMemcpyHostToDevice, stream));
        for (int k = 0; k < 4; ++k) {
            if (inverse_list[k].extras == 0) continue;
            RELP_HIP(hipMemcpyAsync(dev_ + to[k].s_xstart, h + to[k].s_xstart, (size_t)stride * sizeof(int), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipMemcpyAsync(dev_ + to[k].s_xn, h + to[k].s_xn, (size_t)stride * sizeof(int), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipMemcpyAsync(dev_ + to[k].x_idx, h + to[k].x_idx, inverse_list[k].extras * sizeof(int), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipMemcpyAsync(dev_ + to[k].x_val, h + to[k].x_val, inverse_list[k].extras * sizeof(double), hipMemcpyHostToDevice, stream));
        }
    } else if (upload_bytes <= (size_t)(2u << 20)) {
        RELP_HIP(hipMemcpyAsync(dev_, h, upload_bytes, hipMemcpyHostToDevice, stream));
    } else {
        RELP_HIP(hipMemcpyAsync(dev_, h, small_bytes, hipMemcpyHostToDevice, stream));
        auto copy = [&](size_t at, size_t bytes) {
            if (bytes) RELP_HIP(hipMemcpyAsync(dev_ + at, h + at, bytes, hipMemcpyHostToDevice, stream));
        };
        for (int k = 0; k < 4; ++k) {
            copy(to[k].s_col, (size_t)LU_TE * stride * sizeof(int));  // (whole: the padding is part of the contract)
            copy(to[k].s_val, (size_t)LU_TE * stride * sizeof(double));
            copy(to[k].x_idx, tasks[k].x_idx.size() * sizeof(int));
            copy(to[k].x_val, tasks[k].x_val.size() * sizeof(double));
        }
        if (!inverse_factors) {
            copy(o_lrcol, nl * sizeof(int));
            copy(o_lrval, nl * sizeof(double));
            copy(o_lcrow, nl * sizeof(int));
            copy(o_lcval, nl * sizeof(double));
            copy(o_urcol, nu * sizeof(int));
            copy(o_urval, nu * sizeof(double));
            copy(o_ucrow, nu * sizeof(int));
            copy(o_ucval, nu * sizeof(double));
        }
    }
    mark(4);
    if (time_parts && ++uploads % 25 == 0)
        fprintf(stderr, "[refactor]   upload parts, %lld so far: level schedules %.2f, column orientations %.2f, slot lists %.2f, staging %.2f (%zu bytes), copies %.2f ms\n",
                uploads, part_seconds[0] * 1e3, part_seconds[1] * 1e3, part_seconds[2] * 1e3, part_seconds[3] * 1e3, upload_bytes, part_seconds[4] * 1e3);
    const DeviceLU d = bind_layout(lay, dev_, m, max_updates, inverse_vectors, stride);
    d_ = d;
    hipLaunchKernelGGL(lu_init_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, d_);
    nnz_l = factors.nnz_l();
    nnz_u = factors.nnz_u();
    lu_depths(factors, &depth_l, &depth_u);
    return layout_changed;
}

// ---- the refactorisation as kernels --------------------------------------------------------------------------------------------
bool LuFactors::prepare_device(int m, int max_updates, bool inverse_factors, size_t nnz_bound) {
    if (max_updates < 1) max_updates = 1;
    if (max_updates > LU_MAX_SLOTS) max_updates = LU_MAX_SLOTS;
    const int inverse_vectors = inverse_factors ? lu_inverse_vectors(m, max_updates) : 0;
    if (inverse_factors && inverse_vectors == 0) throw std::invalid_argument("the inverse-factor carry: too many rows for its vectors in LDS");
    if (!inverse_factors) throw std::invalid_argument("the device refactorisation builds the inverse-factor form only");
    if (m > 65535) throw std::invalid_argument("the device refactorisation: more than 65535 rows");
    bool layout_changed = m != d_.m || max_updates != d_.max_updates || inverse_vectors != d_.inverse_factors;
    if (layout_changed) cap_l_ = cap_u_ = cap_slots_ = 0;
    // bounds instead of sizes: L + U of a simplex basis hold 1.0-1.5 x its entries (4 x here), the two inverted triangles 3-10 x
    // those of L + U (lu_host.hpp) -- 6 x the bound here; what does not fit is reported by the kernels and redone by the host path
    const size_t want_factor = 4 * nnz_bound + 8 * (size_t)m + 4096;
    const size_t want_inverse = 6 * nnz_bound + 8 * (size_t)m + 4096;
    const size_t want_slots = ((want_inverse / 2 + 2 * (size_t)m + 2048) + 1023) & ~size_t(1023);
    if (want_factor > cap_l_) { cap_l_ = want_factor; layout_changed = true; }
    if (want_factor > cap_u_) { cap_u_ = want_factor; layout_changed = true; }
    if (want_slots > cap_slots_) { cap_slots_ = want_slots; layout_changed = true; }
    const int stride = (int)cap_slots_;
    const LuLayout lay = compute_layout(m, max_updates, inverse_factors, cap_l_, cap_u_, stride);
    {
        char* before = dev_;
        reserve(lay.device_bytes, 0);
        if (dev_ != before) layout_changed = true;
    }
    d_ = bind_layout(lay, dev_, m, max_updates, inverse_vectors, stride);
    scratch_.reserve(m, nnz_bound, cap_l_, cap_u_, want_inverse);
    device_prepared_ = true;
    return layout_changed;
}

void LuFactors::refactor_device(const LuFactorSource& src, double threshold, int reference_ties, int dense_tail, Ctl* ctl, int failed_status,
                                hipStream_t stream) {
    if (!device_prepared_) throw std::logic_error("LuFactors::refactor_device before prepare_device");
    LuFactorOut out;
    out.rowpos = d_.rowpos; out.colpos = d_.colpos; out.diag = d_.diag;
    out.l_start = d_.l_rstart; out.l_col = d_.l_rcol; out.l_val = d_.l_rval;
    out.u_start = d_.u_rstart; out.u_col = d_.u_rcol; out.u_val = d_.u_rval;
    out.cap_l = (int)std::min<size_t>(cap_l_, (size_t)1 << 30);
    out.cap_u = (int)std::min<size_t>(cap_u_, (size_t)1 << 30);
    launch_lu_factor(src, scratch_.work(), out, threshold, reference_ties, dense_tail, stream);
    LuInverseWork iw = scratch_.inverse_work();
    iw.cap_extra_l = out.cap_l;
    iw.cap_extra_u = out.cap_u;
    launch_lu_invert(out, iw, scratch_.work().info, stream);
    launch_lu_pack_inverse(d_, iw, ctl, failed_status, stream);
    hipLaunchKernelGGL(lu_init_kernel, dim3((d_.m + 255) / 256), dim3(256), 0, stream, d_);
}

// |B x - v| for the basis as it stands: one workgroup, the products accumulated in LDS (m doubles), x = what the new factors and
// the replayed etas make of the probe v.  The replayed etas were computed with the OLD factors, so their rounding errors reach the
// new inverse and from there the next cycle's etas: left alone the residual grows from cycle to cycle (25FV47: 1e-10 -> 1e-4 in
// thirty swaps, then a singular basis); the host swaps only while this stays small and factorises synchronously when it does not.
__global__ void __launch_bounds__(1024) lu_probe_fill_kernel(double* v, int m) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) v[i] = 1.0 + 0.01 * (i % 17);
}
__global__ void __launch_bounds__(1024) lu_basis_residual_kernel(const int* col_start, const int* row_index, const double* value, const int* basis, const int* flipped,
                                                                const double* x, const double* v, int m, double* out) {
    extern __shared__ double s_back[];
    __shared__ double s_max[1024 / WAVE];
    const int tid = threadIdx.x, T = blockDim.x;
    for (int i = tid; i < m; i += T) s_back[i] = -v[i];
    __syncthreads();
    for (int k = tid; k < m; k += T) {
        const int j = basis[k];
        const double xk = (flipped && flipped[j]) ? -x[k] : x[k];
        for (int e = col_start[j]; e < col_start[j + 1]; ++e) atomicAdd(&s_back[row_index[e]], value[e] * xk);
    }
    __syncthreads();
    double worst = 0.0;
    for (int i = tid; i < m; i += T) worst = fmax(worst, fabs(s_back[i]));
    for (int d = WAVE / 2; d > 0; d /= 2) worst = fmax(worst, __shfl_xor(worst, d));
    if ((tid & (WAVE - 1)) == 0) s_max[tid / WAVE] = worst;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < T / WAVE; ++w) worst = fmax(worst, s_max[w]);
        out[0] = worst;
    }
}
void launch_lu_probe_fill(double* v, int m, hipStream_t s) { hipLaunchKernelGGL(lu_probe_fill_kernel, dim3(4), dim3(1024), 0, s, v, m); }
void launch_lu_basis_residual(const int* col_start, const int* row_index, const double* value, const int* basis, const int* flipped, const double* x, const double* v,
                              int m, double* out, hipStream_t s) {
    hipLaunchKernelGGL(lu_basis_residual_kernel, dim3(1), dim3(1024), (size_t)m * sizeof(double), s, col_start, row_index, value, basis, flipped, x, v, m, out);
}

void LuFactors::start_log(hipStream_t stream) { hipLaunchKernelGGL(lu_start_log_kernel, dim3(1), dim3(1), 0, stream, d_); }
void LuFactors::replay_log_of(const LuFactors& old, hipStream_t stream) {
    hipLaunchKernelGGL(lu_replay_plan_kernel, dim3(1), dim3(LU_REPLAY_THREADS), 0, stream, d_, old.d_);
    hipLaunchKernelGGL(lu_replay_rows_kernel, dim3((d_.m + LU_REPLAY_ROW_THREADS - 1) / LU_REPLAY_ROW_THREADS), dim3(LU_REPLAY_ROW_THREADS), 0, stream, d_, old.d_);
}

// LDS of the solve kernels: the two vectors (16 bytes per row), the mask of the replaced positions, one count per 64 rows for
// the ordered compactions, reductions, and the trailing block T with its four slot vectors.
// `inverse_vectors`: 0 = the Forrest-Tomlin form; 4 / 3 = the inverse-factor form with four vectors in LDS (a product is out of
// place and the BTRAN has two right-hand sides) or, for the rows that leaves no room for, with three (the two right-hand sides go
// through the factors one after the other: twice the passes over them).
static size_t lu_lds_fixed_bytes(int m, int max_updates, int inverse_vectors = 0) {
    const size_t mm = (size_t)((m + 1) & ~1);
    if (inverse_vectors)  // no T / MF; the per-wave partials of M' r
        return (size_t)inverse_vectors * mm * sizeof(double) + ((size_t)(m + 31) / 32 + 2) * sizeof(int) + ((size_t)(m + 63) / 64 + 4) * sizeof(int) + 64 * sizeof(double) +
               ((size_t)4 * LU_MAX_SLOTS + (size_t)2 * (LU_THREADS / 64) * LU_MAX_SLOTS) * sizeof(double) + 256;
    return 2 * mm * sizeof(double) + ((size_t)(m + 31) / 32 + 2) * sizeof(int) + ((size_t)(m + 63) / 64 + 4) * sizeof(int) + 64 * sizeof(double) +
           ((size_t)2 * max_updates * (max_updates + 1) + 4 * LU_MAX_SLOTS) * sizeof(double) + 256;
}
constexpr size_t LU_LDS_TOTAL = 160 * 1024 - 1024;  // what a kernel may ask for (static LDS of the fused kernel comes on top)
static size_t lu_lds_bytes_for(const DeviceLU& lu) {
    return std::min(LU_LDS_TOTAL - 2048, lu_lds_fixed_bytes(lu.m, lu.max_updates, lu.inverse_factors));
}
size_t LuFactors::lds_bytes(int) const { return lu_lds_bytes_for(d_); }
static int lu_inverse_vectors(int m, int max_updates) {  // 4 when they fit, else 3, else 0 (does not fit at all)
    for (int vectors = 4; vectors >= 3; --vectors)
        if (lu_lds_fixed_bytes(m, max_updates, vectors) <= LU_LDS_TOTAL - 2048) return vectors;
    return 0;
}
bool lu_fits_lds(int m, int max_updates, bool inverse_factors) {
    if (max_updates < 1) max_updates = 1;
    if (max_updates > LU_MAX_SLOTS) max_updates = LU_MAX_SLOTS;
    if (inverse_factors) return lu_inverse_vectors(m, max_updates) != 0;
    return lu_lds_fixed_bytes(m, max_updates, 0) <= LU_LDS_TOTAL - 2048;
}

// =====================================================================================================
// device: synchronisation-free triangular solves out of LDS
// =====================================================================================================
// Explicit LDS pointer types.  A generic `volatile double*` that happens to point into LDS compiles to flat_load ... sc0 sc1
// -- measured: about 2000 cycles per dependent access instead of the ~64 of a ds_read -- so every LDS array of this file is
// typed by address space.
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(3))) char lds_i8;
typedef const __attribute__((address_space(1))) int* gptr_i32;
typedef const __attribute__((address_space(1))) double* gptr_f64;

// ---- eta files --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_value(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ double lane_value(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// LDS carve-up shared by every kernel of this file
struct LuShared {
    volatile lds_f64* x0;
    volatile lds_f64* x1;
    lds_u32* mask;     // one bit per position: replaced by a Forrest-Tomlin update (no row of U_bb any more; operand held at zero)
    int* group_count;  // one slot per 64 rows (+2)   (generic pointers: used with barriers around, a handful of accesses)
    double* red;       // 64 doubles
    volatile lds_f64* T;     // the trailing block, max_updates x ldt
    volatile lds_f64* MF;    // the etas' chaining matrix (lu.hpp), max_updates x ldt
    volatile lds_f64* xt0;   // trailing values by slot (LU_MAX_SLOTS each)
    volatile lds_f64* xt1;
    volatile lds_f64* st0;   // BTRAN: right-hand sides of the trailing solve
    volatile lds_f64* st1;
    volatile lds_f64* x2;    // inverse-factor form only: the other two vectors of the out-of-place products
    volatile lds_f64* x3;
    volatile lds_f64* pfpart;  // ... and [2][waves][LU_MAX_SLOTS] per-wave partial sums of M' r
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
__device__ __forceinline__ LuShared lu_shared(char* smem_generic, int m, int max_updates, int inverse_factors = 0) {
    const int mm = (m + 1) & ~1;
    if (inverse_factors) max_updates = 0;  // no T, no MF
    lds_i8* smem = (lds_i8*)smem_generic;
    LuShared s;
    lds_f64* x0 = (lds_f64*)smem;
    lds_f64* x1 = x0 + mm;
    lds_f64* x2 = inverse_factors ? x1 + mm : x1;
    lds_f64* x3 = inverse_factors == 4 ? x2 + mm : x2;  // (three vectors: x3 is x2 -- the callers know)
    lds_f64* red = x3 + mm;
    lds_f64* T = red + 64;
    lds_f64* MF = T + max_updates * (max_updates + 1);
    lds_f64* xt0 = MF + max_updates * (max_updates + 1);
    lds_f64* xt1 = xt0 + LU_MAX_SLOTS;
    lds_f64* st0 = xt1 + LU_MAX_SLOTS;
    lds_f64* st1 = st0 + LU_MAX_SLOTS;
    lds_f64* pfpart = st1 + LU_MAX_SLOTS;
    lds_f64* after = inverse_factors ? pfpart + 2 * (LU_THREADS / 64) * LU_MAX_SLOTS : pfpart;
    s.x0 = x0;
    s.x1 = x1;
    s.x2 = x2;
    s.x3 = x3;
    s.pfpart = pfpart;
    s.red = (double*)red;
    s.T = T;
    s.MF = MF;
    s.xt0 = xt0;
    s.xt1 = xt1;
    s.st0 = st0;
    s.st1 = st1;
    lds_u32* mask = (lds_u32*)after;
    s.mask = mask;
    s.group_count = (int*)(mask + (((m + 31) / 32 + 2 + 1) & ~1));
    s.dbg = nullptr;
    s.t_prev = nullptr;
    return s;
}
// Barrier for data exchanged through LDS only: this wave's LDS operations have completed, global loads may stay in flight.
// (__syncthreads() also drains vmcnt: with the next task's record on its way from L2 that put a global round trip -- 2-3 k
// cycles -- into every level of a solve.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ bool lu_masked(const LuShared& sh, int pos) { return (sh.mask[pos >> 5] >> (pos & 31)) & 1u; }
// x0 (x1) <- 0, T staged from global, the mask of the replaced positions built.  Ends with a barrier.
template <bool INV>
__device__ __forceinline__ void lu_clear(const DeviceLU& lu, const LuShared& sh, int n_updates, bool two) {
    const int m = lu.m;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        sh.x0[i] = 0.0;
        if (two) sh.x1[i] = 0.0;
    }
    if constexpr (INV) {  // no trailing block, no etas, no masked positions
        __syncthreads();
        return;
    }
    for (int i = threadIdx.x; i < (m + 31) / 32; i += blockDim.x) sh.mask[i] = 0u;
    for (int i = threadIdx.x; i < n_updates * lu.ldt; i += blockDim.x) {
        sh.T[i] = lu.T[i];
        sh.MF[i] = lu.eta_mf[i];
    }
    __syncthreads();
    if ((int)threadIdx.x < n_updates) {
        const int pos = lu.trail_pos[threadIdx.x];
        if (pos >= 0) __hip_atomic_fetch_or((unsigned int*)(sh.mask + (pos >> 5)), 1u << (pos & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
}

// Sum over aligned groups of 2^g lanes (g per lane: the groups of one wave differ), valid in the LAST lane of each group.
// `gbits`: bit j set when some lane of the wave has g > j (wave-uniform), so that a wave of single-lane rows skips all of it.
__device__ __forceinline__ double group_sum_by(double v, const int g, const unsigned gbits) {
    if (gbits == 0) return v;
    double s1 = v + dpp_f64<DPP_QUAD_1032, 0xF>(0.0, v);
    double out = g >= 1 ? s1 : v;
    if (gbits & 2u) {
        const double s2 = s1 + dpp_f64<DPP_QUAD_2301, 0xF>(0.0, s1);
        out = g >= 2 ? s2 : out;
        if (gbits & 4u) {
            const double s3 = s2 + dpp_f64<DPP_ROW_HALF_MIRROR, 0xF>(0.0, s2);
            out = g == 3 ? s3 : out;
            if (gbits & 8u) {
                double s4 = s2 + dpp_f64<DPP_ROW_ROR4, 0xF>(0.0, s2);
                s4 += dpp_f64<DPP_ROW_ROR8, 0xF>(0.0, s4);
                out = g >= 4 ? s4 : out;
                if (gbits & 16u) {
                    const double s5 = s4 + dpp_f64<DPP_ROW_BCAST15, 0xA>(0.0, s4);
                    out = g >= 5 ? s5 : out;
                    if (gbits & 32u) {
                        const double s6 = s5 + dpp_f64<DPP_ROW_BCAST31, 0xC>(0.0, s5);
                        out = g >= 6 ? s6 : out;
                    }
                }
            }
        }
    }
    return out;
}

// What a thread needs of one triangular solve before its first level: its slot of chunk 0, its first row without entries, and
// the solve's counts.  Nothing here depends on anything but the thread index (the arrays are padded up to their stride), so a
// kernel asks for the NEXT solve's record while the current phase runs: by the time the solve starts its loads have landed.
struct LuSlot {
    int pos, lev, flags;
    double dinv;
    int col[LU_TE];
    double val[LU_TE];
    int z_pos;
    double z_dinv;
    int nz, n_chunks, c_end, l0, l1, l_tail;  // (the same in every lane; read through readfirstlane)
    int n_slots;
};
__device__ __forceinline__ LuSlot lu_load_slot(const DeviceLU& lu, const int sched, const int k) {
    const LuTasks& tk = lu.tasks[sched];
    const int stride = lu.task_stride;
    LuSlot r;
    r.pos = tk.s_pos[k];
    r.lev = tk.s_lev[k];
    r.flags = tk.s_flags[k];
    r.dinv = tk.s_dinv[k];
#pragma unroll
    for (int e = 0; e < LU_TE; ++e) {
        r.col[e] = tk.s_col[(size_t)e * stride + k];
        r.val[e] = tk.s_val[(size_t)e * stride + k];
    }
    r.z_pos = tk.z_pos[threadIdx.x];
    r.z_dinv = tk.z_dinv[threadIdx.x];
    r.nz = tk.counts[LU_CNT_Z];
    r.n_chunks = tk.counts[LU_CNT_CHUNKS];
    r.c_end = tk.counts[LU_CNT_C0_END];
    r.l0 = tk.counts[LU_CNT_C0_L0];
    r.l1 = tk.counts[LU_CNT_C0_L1];
    r.l_tail = tk.counts[LU_CNT_C0_TAIL];
    r.n_slots = tk.counts[LU_CNT_SLOTS];
    return r;
}
// The inverse-factor form: a slot as its compact record (lu.hpp), and what a product needs besides its slots.
typedef double lui_f64x2 __attribute__((ext_vector_type(2)));
struct LuiSlot {
    unsigned hdr;
    unsigned long long cols;
    lui_f64x2 v01, v23;
};
struct LuiHead {
    int nz, n_slots, z_pos;  // rows without entries (this thread's first), slots of the list
};
__device__ __forceinline__ LuiSlot lui_load_slot(const DeviceLU& lu, const int sched, const int k) {
    const LuTasks& tk = lu.tasks[sched];
    typedef const __attribute__((address_space(1))) lui_f64x2* gptr_f64x2;
    const gptr_f64x2 vals = (gptr_f64x2)tk.c_val;
    LuiSlot r;
    r.hdr = tk.c_hdr[k];
    r.cols = tk.c_col[k];
    r.v01 = vals[2 * (size_t)k];
    r.v23 = vals[2 * (size_t)k + 1];
    return r;
}
__device__ __forceinline__ LuiHead lui_load_head(const DeviceLU& lu, const int sched) {
    const LuTasks& tk = lu.tasks[sched];
    LuiHead h;
    h.nz = tk.counts[LU_CNT_Z];
    h.n_slots = tk.counts[LU_CNT_SLOTS];
    h.z_pos = tk.c_zpos[threadIdx.x];
    return h;
}

// In place:  x[i] <- (x[i] - sum_e val[e] x[idx[e]]) / diag[i]  for every row i of one triangular factor in one orientation
// (`sched`: 0 L by rows, 1 U by rows, 2 U by columns, 3 L by columns), level by level, one barrier per level.
//   rows without entries   all threads, one pass (only the division by the diagonal, if any);
//   every other row        1, 2, 4 ... 64 consecutive SLOTS of at most LU_TE entries each (lu.hpp); thread t holds slot
//                          first + t of the current chunk in registers (`first_record`: chunk 0, requested by the caller a phase
//                          earlier).  In a level the lanes of its rows read their operands from LDS in one batch, multiply-add
//                          (padding entries are zeros: no predicates), combine by a fixed DPP tree, and the last lane of a row
//                          writes the component.  The slots of a wave are consecutive in level order, so "which level is this
//                          wave's next" is ONE scalar: a wave with nothing to do in a level executes a scalar compare and the
//                          barrier, no vector instruction (a wave64 vector instruction costs 4 cycles of its SIMD: 16 waves
//                          evaluating even a short vector condition per level cost more than the level's arithmetic).
// Measured (tools/micro/barrier_bench.hip): s_barrier of 16 waves 72 cycles; barrier + 4 reads + 4 FMAs + write 270.
// Round 2 kept records and entries in LDS too (record -> entries -> operands: three dependent round trips per level, ~1 k
// cycles with the barrier).  Forms built this round, measured and dropped: (i) barrier-free, every row polling its operands
// (a sentinel NaN for "not solved yet"): 25-50 k cycles per triangle on 25FV47 against round 2's 18-35 k -- four spinning
// waves per SIMD starve the wave the chain waits for, whereas a wave parked at s_barrier costs nothing; (ii) long rows as one
// wave task each, entries read from L2 in the level: every such row paid a global round trip; (iii) several slots per thread
// in registers: the compiler merges the per-round code into one body with ~150 register selects per level.
// In a U solve (HAS_DIAG) a replaced position (mask) is no task and keeps its value (the callers hold it at zero).
// x0 / x1 must be complete (barrier) on entry; ends with a barrier.
template <int NRHS, bool HAS_DIAG>
__device__ __forceinline__ void lu_solve_tasks(const DeviceLU& lu, const LuShared& sh, const int sched, const LuSlot& first_record) {
    const LuTasks& tk = lu.tasks[sched];
    const int nz = __builtin_amdgcn_readfirstlane(first_record.nz);
    const int n_chunks = __builtin_amdgcn_readfirstlane(first_record.n_chunks);
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & (WAVE - 1);
    constexpr int NONE = 0x7fffffff;
    volatile lds_f64* x0 = sh.x0;
    volatile lds_f64* x1 = sh.x1;
    if (HAS_DIAG) {  // the rows without entries
        if (tid < nz && !lu_masked(sh, first_record.z_pos)) {
            x0[first_record.z_pos] = x0[first_record.z_pos] * first_record.z_dinv;
            if (NRHS == 2) x1[first_record.z_pos] = x1[first_record.z_pos] * first_record.z_dinv;
        }
        for (int z = tid + T; z < nz; z += T) {
            const int p = tk.z_pos[z];
            if (lu_masked(sh, p)) continue;
            const double d = tk.z_dinv[z];
            x0[p] = x0[p] * d;
            if (NRHS == 2) x1[p] = x1[p] * d;
        }
    }
    LuSlot rec = first_record;
    int first_slot = 0, end_slot = __builtin_amdgcn_readfirstlane(first_record.c_end);
    int first_level = __builtin_amdgcn_readfirstlane(first_record.l0), end_level = __builtin_amdgcn_readfirstlane(first_record.l1);
    int tail_level = __builtin_amdgcn_readfirstlane(first_record.l_tail);
    for (int ch = 0; ch < n_chunks; ++ch) {
        if (ch > 0) {  // (a factor of more than 1024 slots: the later chunks are fetched here)
            first_slot = tk.chunk[8 * ch];
            end_slot = tk.chunk[8 * ch + 1];
            first_level = tk.chunk[8 * ch + 2];
            end_level = tk.chunk[8 * ch + 3];
            tail_level = tk.chunk[8 * ch + 4];
            rec = lu_load_slot(lu, sched, first_slot + tid);
        }
        const int k = first_slot + tid;
        const int pos = rec.pos;
        const int lev = k < end_slot ? rec.lev : NONE;  // (the slots behind this chunk belong to the next one)
        int flags = rec.flags;
        const double dinv = rec.dinv;
        if (HAS_DIAG && lev != NONE && lu_masked(sh, pos)) flags &= ~(1 << 8);  // a replaced position: computed, never written
        const int g = flags & 0xff;
        const unsigned gbits = (__any(g > 0) ? 1u : 0u) | (__any(g > 1) ? 2u : 0u) | (__any(g > 2) ? 4u : 0u) | (__any(g > 3) ? 8u : 0u) |
                               (__any(g > 4) ? 16u : 0u) | (__any(g > 5) ? 32u : 0u);
        const bool any_extra = __any((flags >> 9) & 1);
        __syncthreads();
#ifdef RELP_STAMPS
        if (sh.dbg && tid == 0) {  // diagnostic: the solve's preamble (loads, rows without entries) apart from its level loop
            const unsigned long long t = clock64();
            sh.dbg[32 + sched] += t - *sh.t_prev;
            sh.dbg[36 + sched] += end_level - first_level;
            sh.dbg[40 + sched] += end_slot - first_slot;
            *sh.t_prev = t;
        }
#endif
        // ---- the levels of the chunk ---------------------------------------------------------------------------------------------
        int first_lane = 0;                                       // wave-uniform: this wave's next pending lane
        int wave_next = __builtin_amdgcn_readfirstlane(lev);      // ... and its level
        auto level = [&](const int l) {  // this wave's slots of level l
            while (wave_next == l) {
                const bool active = lane >= first_lane && lev == l;
                double xv[LU_TE];
#pragma unroll
                for (int e = 0; e < LU_TE; ++e) xv[e] = x0[rec.col[e]];
                const double own0 = x0[pos];
                double s0 = 0.0, s1 = 0.0, own1 = 0.0;
#pragma unroll
                for (int e = 0; e < LU_TE; ++e) s0 += rec.val[e] * xv[e];
                if (NRHS == 2) {
#pragma unroll
                    for (int e = 0; e < LU_TE; ++e) xv[e] = x1[rec.col[e]];
                    own1 = x1[pos];
#pragma unroll
                    for (int e = 0; e < LU_TE; ++e) s1 += rec.val[e] * xv[e];
                }
                if (any_extra) {  // rows of more than 64 LU_TE entries: the rest from the arena (rare)
                    if (active && ((flags >> 9) & 1)) {
                        const int xs = tk.s_xstart[k], xn = tk.s_xn[k];
                        for (int e = lane; e < xn; e += WAVE) {
                            const int c = tk.x_idx[xs + e];
                            const double v = tk.x_val[xs + e];
                            s0 += v * x0[c];
                            if (NRHS == 2) s1 += v * x1[c];
                        }
                    }
                }
                // (only when a row of THIS level in this wave spans several lanes: most levels are single-lane rows, and a stage of the
                //  tree is ~8 wave instructions = 32 cycles of the one wave the level waits for)
                const unsigned level_bits = __ballot(active && g > 0) ? gbits : 0u;
                s0 = group_sum_by(s0, g, level_bits);
                if (NRHS == 2) s1 = group_sum_by(s1, g, level_bits);
                if (active && ((flags >> 8) & 1)) {
                    x0[pos] = (own0 - s0) * dinv;
                    if (NRHS == 2) x1[pos] = (own1 - s1) * dinv;
                }
                first_lane += __popcll(__ballot(active));
                wave_next = first_lane < WAVE ? __builtin_amdgcn_readlane(lev, first_lane < WAVE ? first_lane : 0) : NONE;
            }
        };
        for (int l = first_level; l < tail_level; ++l) {
            level(l);
            lds_barrier();
        }
        // the tail: every remaining slot sits in ONE wave, and LDS keeps one wave's accesses in order -- no barrier between its levels
        // (a level here costs its LDS round trip; with the barrier and the bookkeeping of sixteen waves it costs three times that)
        for (int l = tail_level; l < end_level; ++l) level(l);
        __syncthreads();
#ifdef RELP_STAMPS
        if (sh.dbg && tid == 0) {
            const unsigned long long t = clock64();
            sh.dbg[44 + sched] += t - *sh.t_prev;
            *sh.t_prev = t;
        }
#endif
    }
    if (n_chunks == 0) __syncthreads();
}

// ---- the inverse-factor form (lu.hpp): products with L^-1 / U^-1 and the product-form updates on top ------------------------------
// out <- A in for one of the four task lists (`sched`: 0 L^-1 by rows, 1 U^-1 by rows, 2 U^-1 by columns, 3 L^-1 by columns), OUT OF
// PLACE: every row reads `in` only, so there is no level loop and no barrier between the chunks -- one pass over the slots, the
// same packed slots and DPP group sums as the solves.  UNIT: the factor has an implied unit diagonal (the L lists); the U lists
// carry their diagonal as entries.  in0 / in1 complete (barrier) on entry; ends with a barrier.
template <int NRHS, bool UNIT>
__device__ __forceinline__ void lui_apply(const DeviceLU& lu, const int sched, const LuiHead& head, const LuiSlot& first_record,
                                          volatile lds_f64* in0, volatile lds_f64* in1, volatile lds_f64* out0, volatile lds_f64* out1) {
    const LuTasks& tk = lu.tasks[sched];
    const int nz = __builtin_amdgcn_readfirstlane(head.nz);
    const int n_slots = __builtin_amdgcn_readfirstlane(head.n_slots);
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & (WAVE - 1);
    if (UNIT) {  // rows without entries: a copy
        if (tid < nz) {
            out0[head.z_pos] = in0[head.z_pos];
            if (NRHS == 2) out1[head.z_pos] = in1[head.z_pos];
        }
        for (int z = tid + T; z < nz; z += T) {
            const int p = tk.c_zpos[z];
            out0[p] = in0[p];
            if (NRHS == 2) out1[p] = in1[p];
        }
    }
    // Thread t takes the slots t, t + T, t + 2 T, ... (a row's slots never straddle a wave, and T is a multiple of the wave).  FOUR
    // records are requested before the first is used -- the passes over a factor are a handful, and each was a global round trip of
    // its own when they were fetched one by one.  Slots past the end are padding up to the stride: header 0, operands at position 0.
    constexpr int BATCH = 4;
    const int rounds = (n_slots + T - 1) / T;
    const int padding_slot = lu.task_stride - 1;
    auto work = [&](const LuiSlot& rec, const int k) {
        const unsigned hdr = rec.hdr;
        const int g = (int)(hdr >> 16) & 7;
        const unsigned summary = (unsigned)__builtin_amdgcn_readfirstlane((int)hdr);  // the host's wave summary
        const unsigned gbits = (summary >> 21) & 63u;
        const int c0 = (int)(rec.cols & 0xffffu), c1 = (int)((rec.cols >> 16) & 0xffffu), c2 = (int)((rec.cols >> 32) & 0xffffu), c3 = (int)(rec.cols >> 48);
        double s0 = rec.v01.x * in0[c0] + rec.v01.y * in0[c1] + rec.v23.x * in0[c2] + rec.v23.y * in0[c3];
        double s1 = 0.0;
        if (NRHS == 2) s1 = rec.v01.x * in1[c0] + rec.v01.y * in1[c1] + rec.v23.x * in1[c2] + rec.v23.y * in1[c3];
        if ((summary >> 27) & 1u) {  // rows of more than 256 entries: the rest from the arena, by the lanes of the row's wave
            if ((hdr >> 20) & 1u) {
                const int xs = tk.s_xstart[k], xn = tk.s_xn[k];
                for (int e = lane; e < xn; e += WAVE) {
                    const int c = tk.x_idx[xs + e];
                    const double v = tk.x_val[xs + e];
                    s0 += v * in0[c];
                    if (NRHS == 2) s1 += v * in1[c];
                }
            }
        }
        s0 = group_sum_by(s0, g, gbits);
        if (NRHS == 2) s1 = group_sum_by(s1, g, gbits);
        if ((hdr >> 19) & 1u) {
            const int pos = (int)(hdr & 0xffffu);
            out0[pos] = UNIT ? in0[pos] + s0 : s0;
            if (NRHS == 2) out1[pos] = UNIT ? in1[pos] + s1 : s1;
        }
    };
    for (int r0 = 0; r0 < rounds; r0 += BATCH) {
        LuiSlot rec[BATCH];
        int slot[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            slot[u] = (r0 + u < rounds) ? (r0 + u) * T + tid : padding_slot;
            if (r0 + u == 0) rec[u] = first_record;
            else rec[u] = lui_load_slot(lu, sched, slot[u]);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) work(rec[u], slot[u]);
    }
    __syncthreads();
}

// alpha <- M y in place on the position-space vector x0 (component of basis slot s at x0[colpos[s]]):
//   alpha_s = y_s + sum_c (M[s][c] - [s == slot_c]) y[slot_c].
// The k operands y[slot_c] are read first (xt0), then every row adds its k terms -- rows are independent.  Ends with a barrier.
__device__ __forceinline__ void lui_apply_updates_forward(const DeviceLU& lu, const LuShared& sh, const int k) {
    if (k <= 0) return;
    const int m = lu.m;
    const gptr_f64 M = (gptr_f64)lu.pf_M;
    const gptr_i32 slot_of = (gptr_i32)lu.pf_slot;
    if ((int)threadIdx.x < k) sh.xt0[threadIdx.x] = sh.x0[lu.colpos[slot_of[threadIdx.x]]];
    __syncthreads();
    for (int s = threadIdx.x; s < m; s += blockDim.x) {
        double acc = 0.0;
        for (int c0 = 0; c0 < k; c0 += 8) {  // (eight loads in flight: one after the other they were a round trip each)
            double mc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) mc[u] = (c0 + u < k) ? M[(size_t)(c0 + u) * lu.pf_ld + s] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < k) acc += (mc[u] - (slot_of[c0 + u] == s ? 1.0 : 0.0)) * sh.xt0[c0 + u];
        }
        const int pos = lu.colpos[s];
        sh.x0[pos] = sh.x0[pos] + acc;
    }
    __syncthreads();
}

// r <- r M in place on the position-space vectors x0 (x1), basis slot s at x[colpos[s]]:  only the k components at the kept slots
// change,  r'[slot_c] = sum_s r_s M[s][c].  Every thread forms its rows' share of the k sums, a wave reduces them by DPP, the
// sixteen partials per sum are added in wave order by one thread (fixed order: deterministic).  Ends with a barrier.
template <int NRHS>
__device__ __forceinline__ void lui_apply_updates_backward(const DeviceLU& lu, const LuShared& sh, const int k) {
    if (k <= 0) return;
    const int m = lu.m;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE, nwaves = blockDim.x / WAVE;
    const gptr_f64 M = (gptr_f64)lu.pf_M;
    for (int c0 = 0; c0 < k; c0 += 8) {  // eight sums at a time: their loads in flight together, their wave reductions back to back
        double p0[8], p1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p0[u] = p1[u] = 0.0;
        for (int s = threadIdx.x; s < m; s += blockDim.x) {
            const int pos = lu.colpos[s];
            const double r0 = sh.x0[pos], r1 = NRHS == 2 ? sh.x1[pos] : 0.0;
            double mc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) mc[u] = (c0 + u < k) ? M[(size_t)(c0 + u) * lu.pf_ld + s] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p0[u] += r0 * mc[u];
                if (NRHS == 2) p1[u] += r1 * mc[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (c0 + u >= k) break;  // wave-uniform
            const double w0 = wave_sum(p0[u]);
            const double w1 = NRHS == 2 ? wave_sum(p1[u]) : 0.0;
            if (lane == LAST) {
                sh.pfpart[wave * LU_MAX_SLOTS + c0 + u] = w0;
                if (NRHS == 2) sh.pfpart[(nwaves + wave) * LU_MAX_SLOTS + c0 + u] = w1;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < k) {
        const int c = threadIdx.x;
        double t0 = 0.0, t1 = 0.0;
        for (int w = 0; w < nwaves; ++w) {
            t0 += sh.pfpart[w * LU_MAX_SLOTS + c];
            if (NRHS == 2) t1 += sh.pfpart[(nwaves + w) * LU_MAX_SLOTS + c];
        }
        const int pos = lu.colpos[lu.pf_slot[c]];
        sh.x0[pos] = t0;
        if (NRHS == 2) sh.x1[pos] = t1;
    }
    __syncthreads();
}

// ---- eta files, applied in parallel -------------------------------------------------------------------------------------------
// FTRAN direction (eta_file.rs:72-105): for each update j in order  v[t_j] -= sum_k r_jk v[k].  With V_j the value of v[t_j]
// right after step j:  V_j = base_j - sum_{i<j} MF[j][i] V_i,  base_j = (v[t_j] unless an earlier eta pivots there) - the dot
// product over the arena entries (positions no earlier eta pivots on: their v is still the input).  The dot products are
// independent -- one wave per eta, all sixteen waves --; the k x k chain is solved by one wave with readlane broadcasts; the
// last V of every position is written back.  (Round 2: one wave, one dependent dot product per eta, 11 k cycles at k ~ 16.)
// x0 complete on entry; ends with a barrier.
__device__ __forceinline__ void lu_etas_forward(const DeviceLU& lu, const LuShared& sh, const int n_updates) {
    if (n_updates <= 0) return;
    volatile lds_f64* x0 = sh.x0;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE, nwaves = blockDim.x / WAVE;
    for (int j = wave; j < n_updates; j += nwaves) {
        const int start = lu.eta_start[j], end = lu.eta_start[j + 1];
        double partial = 0.0;
        for (int e = start + lane; e < end; e += WAVE) partial += lu.eta_val[e] * x0[lu.eta_idx[e]];
        const double total = wave_sum(partial);
        if (lane == LAST) sh.st0[j] = (lu.eta_prev[j] < 0 ? x0[lu.eta_pivot[j]] : 0.0) - total;
    }
    __syncthreads();
    if (threadIdx.x < WAVE) {
        const bool mine = lane < n_updates;
        double v = mine ? sh.st0[lane] : 0.0;
        const int t = mine ? lu.eta_pivot[lane] : 0;
        const bool latest = mine && lu.eta_of_pos[t] == lane;
        for (int i = 0; i + 1 < n_updates; ++i) {
            const double vi = lane_value(v, i);
            if (lane > i && mine) v -= sh.MF[lane * lu.ldt + i] * vi;
        }
        if (latest) x0[t] = v;
    }
    __syncthreads();
}
// BTRAN direction (eta_file.rs:49-65): for each update j in reverse  v[k] -= r_jk v[t_j].  With c_j the value of v[t_j] when
// step j is applied:  c_j = (v[t_j] unless a later eta pivots there) - sum_{l>j} MF[l][j] c_l  (one wave, readlane chain), and
// then every position gathers  v[k] = (c of the first eta that pivots on k, else v[k]) - sum over its arena entries r_lk c_l.
template <int NRHS>
__device__ __forceinline__ void lu_etas_backward(const DeviceLU& lu, const LuShared& sh, const int n_updates) {
    if (n_updates <= 0) return;
    volatile lds_f64* x0 = sh.x0;
    volatile lds_f64* x1 = sh.x1;
    const int lane = threadIdx.x & (WAVE - 1);
    if (threadIdx.x < WAVE) {
        const bool mine = lane < n_updates;
        const int t = mine ? lu.eta_pivot[lane] : 0;
        const bool last_of_position = mine && lu.eta_of_pos[t] == lane;
        double c0 = last_of_position ? x0[t] : 0.0;
        double c1 = (NRHS == 2 && last_of_position) ? x1[t] : 0.0;
        for (int l = n_updates - 1; l > 0; --l) {
            const double cl0 = lane_value(c0, l);
            const double cl1 = NRHS == 2 ? lane_value(c1, l) : 0.0;
            if (lane < l) {
                const double f = sh.MF[l * lu.ldt + lane];
                c0 -= f * cl0;
                if (NRHS == 2) c1 -= f * cl1;
            }
        }
        if (mine) {
            sh.st0[lane] = c0;
            if (NRHS == 2) sh.st1[lane] = c1;
        }
    }
    __syncthreads();
    const int stride = lu.max_updates;
    for (int k = threadIdx.x; k < lu.m; k += blockDim.x) {
        const int n = lu.eapp_len[k];
        const int first = lu.eta_first[k];
        if (n <= 0 && first < 0) continue;
        double v0 = first >= 0 ? sh.st0[first] : x0[k];
        double v1 = NRHS == 2 ? (first >= 0 ? sh.st1[first] : x1[k]) : 0.0;
        for (int e = 0; e < n; ++e) {
            const int l = lu.eapp_eta[k * stride + e];
            if (l >= n_updates) break;  // (the eta being built in this very pivot: not part of the current basis yet)
            const double r = lu.eapp_val[k * stride + e];
            v0 -= r * sh.st0[l];
            if (NRHS == 2) v1 -= r * sh.st1[l];
        }
        x0[k] = v0;
        if (NRHS == 2) x1[k] = v1;
    }
    __syncthreads();
}

// FTRAN on the vector in sh.x0 (position space, P already applied): L solve, etas, [spike], U solve (trailing block by one
// wave, spike contributions, then the base rows with the replaced positions held at zero).  Ends with a barrier.
template <bool INV>
__device__ __forceinline__ void lu_ftran_block(const DeviceLU& lu, const LuShared& sh, const int n_updates, int& epoch,
                                               double* spike_out, const LuSlot& lower_record) {
    (void)epoch;
    const int m = lu.m;
    const LuSlot upper_record = INV ? LuSlot{} : lu_load_slot(lu, 1, threadIdx.x);  // (lands while L and the etas are done)
    if constexpr (INV) {  // x1 <- L^-1 x0, x0 <- U^-1 x1, then the product-form updates (x1 is scratch)
        (void)lower_record;
        (void)upper_record;
        lui_apply<1, true>(lu, 0, lui_load_head(lu, 0), lui_load_slot(lu, 0, threadIdx.x), sh.x0, sh.x0, sh.x1, sh.x1);
        lu_stamp(sh, 2);
        lui_apply<1, false>(lu, 1, lui_load_head(lu, 1), lui_load_slot(lu, 1, threadIdx.x), sh.x1, sh.x1, sh.x0, sh.x0);
        lu_stamp(sh, 3);
        lui_apply_updates_forward(lu, sh, lu.state[LU_PF_COUNT]);
        lu_stamp(sh, 4);
        return;
    }
    lu_solve_tasks<1, false>(lu, sh, 0, lower_record);
    lu_stamp(sh, 2);
    lu_etas_forward(lu, sh, n_updates);
    lu_stamp(sh, 3);
    if (spike_out) {
        for (int i = threadIdx.x; i < m; i += blockDim.x) spike_out[i] = sh.x0[i];
        __syncthreads();  // (the spike is read from x0 before anybody rewrites it: the trailing solve, or the U solve's first pass)
    }
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
            if (pos >= 0) sh.x0[pos] = 0.0;  // masked: the stale entries of its old column multiply nothing in the base solve
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
        __syncthreads();
    }
    lu_solve_tasks<1, true>(lu, sh, 1, upper_record);
    if (n_updates > 0) {
        if ((int)threadIdx.x < n_updates) {
            const int pos = lu.trail_pos[threadIdx.x];
            if (pos >= 0) sh.x0[pos] = sh.xt0[threadIdx.x];
        }
        __syncthreads();
    }
    lu_stamp(sh, 4);
}

// BTRAN on the vectors in sh.x0 (and sh.x1), position space with Q applied.  `after_upper` runs between the U solve and the
// etas, when x0 = (e_t' U^-1) for a unit input (the Forrest-Tomlin row comes from there).  Ends with a barrier.
template <int NRHS, bool INV, class AfterUpper>
__device__ __forceinline__ void lu_btran_block(const DeviceLU& lu, const LuShared& sh, const int n_updates, int& epoch,
                                               AfterUpper after_upper, const LuSlot& upper_record) {
    (void)epoch;
    const LuSlot lower_record = INV ? LuSlot{} : lu_load_slot(lu, 3, threadIdx.x);  // (lands while U' and the etas are done)
    if constexpr (INV) {  // r <- r M, then (x2, x3) <- (x0, x1) U^-1, (x0, x1) <- (x2, x3) L^-1
        lui_apply_updates_backward<NRHS>(lu, sh, lu.state[LU_PF_COUNT]);
        lu_stamp(sh, 7);
        (void)lower_record;
        (void)upper_record;
        if (NRHS == 2 && lu.inverse_factors == 3) {  // three vectors: the right-hand sides one after the other
            lui_apply<1, false>(lu, 2, lui_load_head(lu, 2), lui_load_slot(lu, 2, threadIdx.x), sh.x0, sh.x0, sh.x2, sh.x2);
            lui_apply<1, true>(lu, 3, lui_load_head(lu, 3), lui_load_slot(lu, 3, threadIdx.x), sh.x2, sh.x2, sh.x0, sh.x0);
            lui_apply<1, false>(lu, 2, lui_load_head(lu, 2), lui_load_slot(lu, 2, threadIdx.x), sh.x1, sh.x1, sh.x2, sh.x2);
            lu_stamp(sh, 8);
            after_upper();
            lui_apply<1, true>(lu, 3, lui_load_head(lu, 3), lui_load_slot(lu, 3, threadIdx.x), sh.x2, sh.x2, sh.x1, sh.x1);
            lu_stamp(sh, 10);
            return;
        }
        lui_apply<NRHS, false>(lu, 2, lui_load_head(lu, 2), lui_load_slot(lu, 2, threadIdx.x), sh.x0, sh.x1, sh.x2, sh.x3);
        lu_stamp(sh, 8);
        after_upper();
        lui_apply<NRHS, true>(lu, 3, lui_load_head(lu, 3), lui_load_slot(lu, 3, threadIdx.x), sh.x2, sh.x3, sh.x0, sh.x1);
        lu_stamp(sh, 10);
        return;
    }
    if (n_updates > 0) {
        // the replaced positions leave the base solve: their right-hand sides wait in xt0 / xt1, their components read as zero
        if ((int)threadIdx.x < n_updates) {
            const int pos = lu.trail_pos[threadIdx.x];
            sh.xt0[threadIdx.x] = pos >= 0 ? sh.x0[pos] : 0.0;
            if (NRHS == 2) sh.xt1[threadIdx.x] = pos >= 0 ? sh.x1[pos] : 0.0;
            if (pos >= 0) {
                sh.x0[pos] = 0.0;
                if (NRHS == 2) sh.x1[pos] = 0.0;
            }
        }
        __syncthreads();
    }
    lu_solve_tasks<NRHS, true>(lu, sh, 2, upper_record);
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
                    p0 += v * sh.x0[i];  // (a row replaced since reads as zero)
                    if (NRHS == 2) p1 += v * sh.x1[i];
                }
            }
            p0 = wave_sum(p0);
            if (NRHS == 2) p1 = wave_sum(p1);
            if (lane == LAST) {
                sh.st0[k] = pos >= 0 ? sh.xt0[k] - p0 : 0.0;
                if (NRHS == 2) sh.st1[k] = pos >= 0 ? sh.xt1[k] - p1 : 0.0;
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
    lu_etas_backward<NRHS>(lu, sh, n_updates);
    lu_stamp(sh, 9);
    lu_solve_tasks<NRHS, false>(lu, sh, 3, lower_record);
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
    const int j = lu.state[LU_N_UPDATES];  // the eta being built
    const int stride = lu.max_updates;
    const double yt = sh.x0[t];
    double dot = 0.0;
    // entries at positions no eta pivots on yet: the arena, and the by-position lists of the BTRAN gather
    const int count = ordered_compact(
        m, sh.group_count, [&](int i) { return i != t && sh.x0[i] != 0.0 && lu.eta_of_pos[i] < 0; },
        [&](int i, int slot) {
            const double r = -sh.x0[i] / yt;
            lu.eta_idx[eta_top + slot] = i;
            lu.eta_val[eta_top + slot] = r;
            const int at = i * stride + lu.eapp_len[i];
            lu.eapp_eta[at] = j;
            lu.eapp_val[at] = r;
            lu.eapp_len[i] += 1;
            dot += r * spike[i];
        });
    // entries at positions an earlier eta pivots on: row j of the chaining matrix (lu.hpp), against the LATEST such eta; -1 where
    // that eta pivots on this eta's own position (its value is what this step starts from)
    if ((int)threadIdx.x < j) {
        const int i = threadIdx.x;
        const int pos = lu.eta_pivot[i];
        double f = 0.0;
        if (lu.eta_of_pos[pos] == i) {
            if (pos == t) {
                f = -1.0;
            } else if (sh.x0[pos] != 0.0) {
                f = -sh.x0[pos] / yt;
                dot += f * spike[pos];
            }
        }
        lu.eta_mf[j * lu.ldt + i] = f;
    }
    const double total = block_reduce<0>(dot, sh.red);
    *new_diag = spike[t] - total;
    return count;
}

// Structural part of the Forrest-Tomlin update (mod.rs:127-176): row t leaves U, column t becomes the spike, position t
// moves to the end of the logical order -- i.e. it takes the next slot of the trailing block.  The base part of U is not
// rewritten: from now on position t is masked in every solve (lu.hpp).  `eta_count` entries were already written by
// lu_build_eta.
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
    if (old_slot >= 0) {
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
            lu.app_len[t] = 0;  // row t leaves: its spike entries are not applied any more (the column copies of S are masked)
        } else {
            lu.s_clen[old_slot] = 0;
            lu.trail_pos[old_slot] = -1;
        }
    }
    // the spike becomes the column of the new slot: base rows -> S (column arena + one appended entry per row),
    // live trailing positions -> T
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
        lu.eta_prev[n_updates] = lu.eta_of_pos[t];
        if (lu.eta_first[t] < 0) lu.eta_first[t] = n_updates;
        lu.eta_of_pos[t] = n_updates;
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
template <bool INV>
__global__ void __launch_bounds__(LU_THREADS) lu_ftran_kernel(DeviceLU lu, const int* rows, const double* vals, int nnz,
                                                               const double* dense, double* out, int keep_spike) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuSlot lower_record = INV ? LuSlot{} : lu_load_slot(lu, 0, threadIdx.x);  // the first solve's record starts travelling at once
    const LuShared sh = lu_shared(smem, m, lu.max_updates, lu.inverse_factors);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear<INV>(lu, sh, n_updates, false);
    if (dense) {
        for (int i = threadIdx.x; i < m; i += blockDim.x) sh.x0[lu.rowpos[i]] = dense[i];
    } else {
        for (int e = threadIdx.x; e < nnz; e += blockDim.x) sh.x0[lu.rowpos[rows[e]]] = vals[e];
    }
    __syncthreads();
    int epoch = 0;
    lu_ftran_block<INV>(lu, sh, n_updates, epoch, keep_spike ? lu.spike : nullptr, lower_record);
    for (int s = threadIdx.x; s < m; s += blockDim.x) out[s] = sh.x0[lu.colpos[s]];
}

template <bool INV>
__global__ void __launch_bounds__(LU_THREADS) lu_btran_kernel(DeviceLU lu, const int* slots, const double* vals, int nnz,
                                                               const double* dense, double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuSlot upper_record = INV ? LuSlot{} : lu_load_slot(lu, 2, threadIdx.x);  // the first solve's record starts travelling at once
    const LuShared sh = lu_shared(smem, m, lu.max_updates, lu.inverse_factors);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear<INV>(lu, sh, n_updates, false);
    if (dense) {
        for (int s = threadIdx.x; s < m; s += blockDim.x) sh.x0[lu.colpos[s]] = dense[s];
    } else {
        for (int e = threadIdx.x; e < nnz; e += blockDim.x) sh.x0[lu.colpos[slots[e]]] = vals[e];
    }
    __syncthreads();
    int epoch = 0;
    lu_btran_block<1, INV>(lu, sh, n_updates, epoch, [] {}, upper_record);
    for (int i = threadIdx.x; i < m; i += blockDim.x) out[i] = sh.x0[lu.rowpos[i]];
}

__global__ void __launch_bounds__(LU_THREADS) lu_update_kernel(DeviceLU lu, int p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuSlot upper_record = lu_load_slot(lu, 2, threadIdx.x);  // the first solve's record starts travelling at once
    const LuShared sh = lu_shared(smem, m, lu.max_updates, lu.inverse_factors);
    const int t = lu.colpos[p];
    const int n_updates = lu.state[LU_N_UPDATES];
    if (n_updates >= lu.max_updates) {  // no room for another eta: the caller has to refactor
        if (threadIdx.x == 0) lu.state[LU_FLAGS] |= LU_FLAG_OVERFLOW;
        return;
    }
    lu_clear<false>(lu, sh, n_updates, false);
    if (threadIdx.x == 0) sh.x0[t] = 1.0;
    __syncthreads();
    // y = e_t' U^-1: the U stage of a BTRAN (mod.rs:373-397), then the eta from it.  (The etas and the L stage run too --
    // they cost nothing next to a second code path -- but the eta is taken right behind the U stage.)
    int epoch = 0;
    int eta_count = 0;
    double new_diag = 0.0;
    lu_btran_block<1, false>(lu, sh, n_updates, epoch, [&] { eta_count = lu_build_eta(lu, sh, t, lu.spike, &new_diag); }, upper_record);
    lu_ft_update_block(lu, sh, t, eta_count, new_diag, lu.spike);
}

static PerDeviceOnce g_lu_lds_configured;  // (per device: solver.hpp)
static void allow_full_lds(const void* kernel);
static void configure_lu_lds() {
    g_lu_lds_configured.run([] {
        allow_full_lds(reinterpret_cast<const void*>(&lu_ftran_kernel<false>));
        allow_full_lds(reinterpret_cast<const void*>(&lu_ftran_kernel<true>));
        allow_full_lds(reinterpret_cast<const void*>(&lu_btran_kernel<false>));
        allow_full_lds(reinterpret_cast<const void*>(&lu_btran_kernel<true>));
        allow_full_lds(reinterpret_cast<const void*>(&lu_update_kernel));
    });
}
static void check_launch(const char* what) {
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) throw DeviceError(std::string(what) + ": " + hipGetErrorString(err));
}

void launch_lu_ftran(const DeviceLU& lu, const int* rows, const double* vals, int nnz, double* out, int keep_spike, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL((lu.inverse_factors ? lu_ftran_kernel<true> : lu_ftran_kernel<false>), dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, lu, rows, vals, nnz, (const double*)nullptr, out, keep_spike);
    check_launch("lu_ftran_kernel");
}
void launch_lu_ftran_dense(const DeviceLU& lu, const double* rhs, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL((lu.inverse_factors ? lu_ftran_kernel<true> : lu_ftran_kernel<false>), dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, lu, (const int*)nullptr, (const double*)nullptr, 0, rhs, out, 0);
    check_launch("lu_ftran_kernel");
}
void launch_lu_btran(const DeviceLU& lu, const int* slots, const double* vals, int nnz, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL((lu.inverse_factors ? lu_btran_kernel<true> : lu_btran_kernel<false>), dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, lu, slots, vals, nnz, (const double*)nullptr, out);
    check_launch("lu_btran_kernel");
}
void launch_lu_btran_dense(const DeviceLU& lu, const double* in_slots, double* out, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL((lu.inverse_factors ? lu_btran_kernel<true> : lu_btran_kernel<false>), dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, lu, (const int*)nullptr, (const double*)nullptr, 0, in_slots, out);
    check_launch("lu_btran_kernel");
}
void launch_lu_update(const DeviceLU& lu, int p, hipStream_t s) {
    configure_lu_lds();
    hipLaunchKernelGGL(lu_update_kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, lu, p);
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
template <int RULE, bool INV>
__global__ void __launch_bounds__(LU_THREADS) lu_pivot_kernel(DeviceLP lp, DeviceLU lu, int n_price_blocks, double tol_pivot,
                                                               double harris_delta, int skip_artificial_rows, int mode,
                                                               int refactor_period) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_akey[LU_THREADS / WAVE];
    __shared__ unsigned long long s_arank[LU_THREADS / WAVE];
    __shared__ double s_bcast[4];
    Ctl* ctl = lp.ctl;
    const LuSlot lower_record = INV ? LuSlot{} : lu_load_slot(lu, 0, threadIdx.x);  // FTRAN's first solve: lands while the entering column is chosen
    // (the inverse-factor form: its compact records, the same way)
    LuiSlot inv_lower_rows{}, inv_upper_cols{};
    LuiHead inv_head[4] = {};
    if constexpr (INV) {
        inv_lower_rows = lui_load_slot(lu, 0, threadIdx.x);
        for (int k = 0; k < 4; ++k) inv_head[k] = lui_load_head(lu, k);
    }
    const int tid = threadIdx.x, T = blockDim.x;
    const int m = lp.m;
    LuShared sh = lu_shared(smem, m, lu.max_updates, lu.inverse_factors);
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
    const int rho_buf = ctl->rho_buf ^ 1;  // generated columns: the half of rho_bits this pivot marks (published with the pivot)
    const long long iters = ctl->iters;
    const long long budget = ctl->budget;
    const int forced_q = ctl->forced_q;
    const int forced_p = ctl->forced_p;
    const double minus_obj = ctl->minus_obj;
    const int n_updates = lu.state[LU_N_UPDATES];
    // inverse-factor form: the kept columns of M (lu.hpp) -- their basis slots and positions go to LDS now, so that no phase below
    // starts with this chain of three dependent loads
    __shared__ int s_pf_slot[LU_MAX_SLOTS], s_pf_pos[LU_MAX_SLOTS];
    const int pf_k = INV ? lu.state[LU_PF_COUNT] : 0;
    if (INV && tid < pf_k) {
        const int slot = lu.pf_slot[tid];
        s_pf_slot[tid] = slot;
        s_pf_pos[tid] = lu.colpos[slot];
    }
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
    // Implicit upper bounds (lp.ub): a column held in complemented form (x_j = u_j - x'_j) enters, and sits in the basis, with
    // the opposite sign -- the factors hold sgn * a_j, so the sign goes into the FTRAN's input and the spike inherits it.
    const bool bounded = lp.ub != nullptr;
    double sgn_q = 1.0, ub_q = INFINITY;
    if (bounded) {
        sgn_q = lp.flipped[q] ? -1.0 : 1.0;
        ub_q = lp.ub[q];
        if (forced_q >= 0) cbar_q *= sgn_q;  // the candidates of the pricing pass carry the sign already
    }
    lu_stamp(sh, 0);
    lu_clear<INV>(lu, sh, n_updates, false);
    for (int e = lp.col_start[q] + tid; e < lp.col_start[q + 1]; e += T) sh.x0[lu.rowpos[lp.row_index[e]]] = sgn_q * lp.value[e];
    __syncthreads();
    lu_stamp(sh, 1);
    int epoch = 0;
    if constexpr (INV) {
        // x1 <- L^-1 x0, x0 <- U^-1 x1, alpha = M (that), the operands y[slot_c] read first
        const LuiSlot upper_rows = lui_load_slot(lu, 1, tid);
        lui_apply<1, true>(lu, 0, inv_head[0], inv_lower_rows, sh.x0, sh.x0, sh.x1, sh.x1);
        lu_stamp(sh, 2);
        lui_apply<1, false>(lu, 1, inv_head[1], upper_rows, sh.x1, sh.x1, sh.x0, sh.x0);
        lu_stamp(sh, 3);
        if (pf_k > 0) {
            if (tid < pf_k) sh.xt0[tid] = sh.x0[s_pf_pos[tid]];
            __syncthreads();
            // alpha_s = y_s + sum_c M[s][c] y[slot_c]  -  (y_s when s is a kept slot: done by the k owners afterwards).  The k operands
            // sit one per lane and reach the multiply-adds through readlane -- as LDS broadcasts they were two reads per term.
            const gptr_f64 M = (gptr_f64)lu.pf_M;
            const int lane = tid & (WAVE - 1);
            const double y_lane = lane < pf_k ? sh.xt0[lane] : 0.0;
            for (int s = tid; s < m; s += 2 * T) {  // two of a thread's rows at a time: sixteen loads in flight (m > 1024 only matters)
                const int s2 = s + T;
                const bool has2 = s2 < m;
                double acc = 0.0, acc2 = 0.0;
                for (int c0 = 0; c0 < pf_k; c0 += 8) {
                    double mc[8], mc2[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        mc[u] = (c0 + u < pf_k) ? M[(size_t)(c0 + u) * lu.pf_ld + s] : 0.0;
                        mc2[u] = (has2 && c0 + u < pf_k) ? M[(size_t)(c0 + u) * lu.pf_ld + s2] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (c0 + u < pf_k) {
                            const double y = lane_value(y_lane, c0 + u);
                            acc += mc[u] * y;
                            acc2 += mc2[u] * y;
                        }
                }
                const int pos = lu.colpos[s];
                sh.x0[pos] = sh.x0[pos] + acc;
                if (has2) {
                    const int pos2 = lu.colpos[s2];
                    sh.x0[pos2] = sh.x0[pos2] + acc2;
                }
            }
            __syncthreads();
            if (tid < pf_k) sh.x0[s_pf_pos[tid]] = sh.x0[s_pf_pos[tid]] - sh.xt0[tid];
            __syncthreads();
        }
        lu_stamp(sh, 4);
    } else {
        lu_ftran_block<INV>(lu, sh, n_updates, epoch, lu.spike, lower_record);
    }
    const LuSlot upper_record = INV ? LuSlot{} : lu_load_slot(lu, 2, tid);  // BTRAN's first solve: lands during the ratio test
    if constexpr (INV) inv_upper_cols = lui_load_slot(lu, 2, tid);
    // ---- alpha per basis slot (kept in x1), gamma_q, Harris pass 1 ----------------------------------------------------------
    // harris_delta < 0: the reference's ratio test (exact minimum, ties to the lowest leaving column; tableau/mod.rs:287-313).
    // With implicit bounds a basic variable may also leave at its upper bound -- rows with alpha_i < 0 whose basic variable has
    // one -- and the entering variable may run into its own bound first (a bound flip, no basis change): the rules of
    // ftran_ratio_fast_kernel (kernels.hip).
    const bool textbook = harris_delta < 0.0;
    const double harris_slack = textbook ? 0.0 : harris_delta;
    // eligibility of row s and the distance of its basic variable to the bound it moves towards
    auto row_room = [&](int s, double a, double* room) {
        const bool allowed = !(skip_artificial_rows && lp.basis[s] < lp.n_art);
        const double xs = lp.xB[s];
        *room = fmax(xs, 0.0);
        bool eligible = allowed && a > tol_pivot;
        if (bounded && allowed && a < -tol_pivot) {
            const double up = lp.xub[s];
            if (up < INFINITY) {
                eligible = true;
                *room = fmax(up - xs, 0.0);
            }
        }
        return eligible;
    };
    double sumsq = 0.0, theta = INFINITY;
    for (int s = tid; s < m; s += T) {
        const double a = sh.x0[lu.colpos[s]];
        sh.x1[s] = a;
        lp.alpha[s] = a;
        sumsq += a * a;
        double room;
        if (row_room(s, a, &room)) theta = fmin(theta, (room + harris_slack) / fabs(a));
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
            double room;
            if (!row_room(s, a, &room)) continue;
            const double mag = fabs(a);
            if (room / mag <= theta_max) {
                const unsigned long long rk = ((unsigned long long)(unsigned)lp.basis[s] << 32) | (unsigned)s;
                const double key = textbook ? 1.0 : mag;
                if (hrank == RANK_NONE || key > hkey || (key == hkey && rk < hrank)) {
                    hkey = key;
                    hrank = rk;
                }
            }
        }
        block_argbest(hkey, hrank, s_akey, s_arank);
        p = hrank == RANK_NONE ? -1 : (int)(hrank & 0xffffffffu);
    }
    const double alpha_pq = p >= 0 ? sh.x1[p] : 1.0;
    double room_p = 0.0;
    if (p >= 0) (void)row_room(p, alpha_pq, &room_p);
    // step length: to the bound of the leaving variable, or (forced zero-level pivots) as the reference computes it
    double xp = p < 0 ? INFINITY : ((forced_p >= 0 || !bounded) ? fmax(lp.xB[p], 0.0) / alpha_pq : room_p / fabs(alpha_pq));
    const bool leaves_at_upper = bounded && forced_p < 0 && p >= 0 && alpha_pq < 0.0;
    const bool flip = bounded && forced_p < 0 && ub_q < INFINITY && (p < 0 || ub_q <= xp);
    if (p < 0 && !flip) {
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
    if (mode == 2) {
        if (tid == 0) {
            ctl->q = q;
            ctl->p = flip ? -1 : p;
            ctl->cbar_q = cbar_q;
            ctl->gamma_q = gamma_q;
            ctl->pending = 0;
            ctl->forced_q = -1;
            ctl->forced_p = -1;
        }
        return;
    }
    if (flip) {
        // ---- bound flip: x_q runs from 0 to ub_q, the basis does not change; x_q is complemented so that it sits at 0 again ------
        __syncthreads();
        for (int s = tid; s < m; s += T) lp.xB[s] = lp.xB[s] - sh.x1[s] * ub_q;
        for (int e = lp.col_start[q] + tid; e < lp.col_start[q + 1]; e += T) lp.rhs[lp.row_index[e]] -= ub_q * sgn_q * lp.value[e];
        if (tid == 0) {
            const int now_flipped = lp.flipped[q] ^ 1;
            lp.flipped[q] = now_flipped;
            lp.pos[q] = now_flipped ? -2 : -1;
            ctl->flip_cost += (now_flipped ? 1.0 : -1.0) * ub_q * lp.cost[q];
            ctl->q = q;
            ctl->p = -1;
            ctl->cbar_q = cbar_q;
            ctl->minus_obj = minus_obj - cbar_q * ub_q;
            ctl->iters = iters + 1;
            ctl->bound_flips += 1;
            ctl->pending = 0;  // no basis change: no update of the factors, no weight update
            ctl->forced_q = -1;
            ctl->forced_p = -1;
            ctl->last_selected = q;
        }
        return;
    }
    if (alpha_pq == 0.0) return;  // (a forced pivot on a zero element: the host sees that nothing happened)
    const int leaving = lp.basis[p];
    const double xb_p = lp.xB[p];
    const int leaving_flipped = bounded ? lp.flipped[leaving] : 0;
    __syncthreads();  // every thread has read basis[p] / xB[p] / flipped[leaving] before they change
    // ---- x_B update (carry/mod.rs:295-325) ------------------------------------------------------------------------------
    for (int s = tid; s < m; s += T) lp.xB[s] = (s == p) ? xp : lp.xB[s] - sh.x1[s] * xp;
    if (leaves_at_upper) {  // the leaving variable reached its upper bound: it is held in complemented form from now on
        const double ub_l = xb_p + room_p;
        const double sgn_l = leaving_flipped ? -1.0 : 1.0;
        for (int e = lp.col_start[leaving] + tid; e < lp.col_start[leaving + 1]; e += T) lp.rhs[lp.row_index[e]] -= ub_l * sgn_l * lp.value[e];
    }
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
    const double diag_t = INV ? 1.0 : lu.diag[t];
    int eta_count = 0;
    double new_diag = 0.0;
    if constexpr (INV) {
        // The two row vectors of the BTRAN in front of the factors (old basis):  e_p' M  (row p of M: its k kept entries, and 1 at p
        // when p has no kept column) and  alpha' M  (alpha off the kept slots, the k sums  sum_s alpha_s M[s][c]  on them) -- and,
        // in the same pass over M, the eta of this pivot folded in:  M[s][c] -= (alpha_s - [s == p]) M[p][c] / alpha_p, plus a new
        // column for p when it had none.  One read of M serves both; row p of the old M is read first (st0).
        typedef __attribute__((address_space(1))) double* gmut_f64;
        const gmut_f64 M = (gmut_f64)lu.pf_M;
        const int have = lu.pf_col_of[p];
        const int k_new = (do_update && have < 0) ? pf_k + 1 : pf_k;
        if (tid < pf_k) sh.st0[tid] = M[(size_t)tid * lu.pf_ld + p];
        if (tid == pf_k) sh.st0[pf_k] = 1.0;  // (the new column starts as e_p)
        __syncthreads();
        lu_stamp(sh, 13);
        const double inv_ap = 1.0 / alpha_pq;
        const int lane = tid & (WAVE - 1), wave = tid / WAVE, nwaves = T / WAVE;
        const double mp_lane = lane < k_new ? sh.st0[lane] : 0.0;  // row p of the old M, one kept column per lane (readlane below)
        // alpha per basis slot into LDS (x2 is free until the products): the k sums run over it a wave per kept column
        for (int s = tid; s < m; s += T) sh.x2[s] = lp.alpha[s];
        __syncthreads();
        lu_stamp(sh, 14);
        if (lu.inverse_factors == 4) {  // (the four-vector layout: x3 is a vector of its own)
            // ONE wave per kept column: it walks the column once (coalesced, all its loads in flight), adds up  sum_s alpha_s M[s][c]
            // (one wave reduction per column -- a thread per row with eight sums at a time has every wave reduce every sum: 18
            // instructions per sum and wave) and writes the column back with the eta folded in; a column belongs to one wave, so
            // the sums and the fold need no barrier between them.  The row factors (alpha_s - [s == p]) / alpha_p wait in x3.
            // (the eta is logged for every pivot made on the BASIS while the next factors are on their way -- also for the one that
            //  finds the update slots full: it changes the basis and leaves the factors as they are, lu.hpp)
            const bool logging = lu.state[LU_LOG_ON] != 0 && lu.state[LU_LOG_COUNT] < LU_LOG_CAPACITY;
            if (do_update || logging)
                for (int s = tid; s < m; s += T) {
                    const double factor = (sh.x2[s] - (s == p ? 1.0 : 0.0)) * inv_ap;
                    if (do_update) sh.x3[s] = factor;
                    if (logging) lu.log_factor[(size_t)lu.state[LU_LOG_COUNT] * lu.pf_ld + s] = factor;
                }
            __syncthreads();
            for (int c = wave; c < k_new; c += nwaves) {
                const gmut_f64 column = M + (size_t)c * lu.pf_ld;
                const bool fresh = c >= pf_k;  // (the new column starts as e_p)
                const double mp_c = sh.st0[c];
                double part = 0.0;
                // (twelve loads first, then their uses and the stores: with loads and stores of one array in one loop body the
                //  compiler keeps their order, and every load became a round trip of its own -- 12 k cycles per column.  Two columns
                //  of a wave at a time, so that their loads travel together, was measured too: slower, 17 k against 14.5 k cycles)
                for (int s0 = lane; s0 < m; s0 += 12 * WAVE) {
                    double old[12];
#pragma unroll
                    for (int u = 0; u < 12; ++u) {
                        const int s = s0 + u * WAVE;
                        old[u] = (s < m && !fresh) ? column[s] : ((s == p && fresh) ? 1.0 : 0.0);
                    }
#pragma unroll
                    for (int u = 0; u < 12; ++u) {
                        const int s = s0 + u * WAVE;
                        if (s < m) {
                            part += sh.x2[s] * old[u];
                            if (do_update) column[s] = old[u] - sh.x3[s] * mp_c;
                        }
                    }
                }
                if (!fresh) {
                    part = wave_sum(part);
                    if (lane == LAST) sh.xt1[c] = part;
                }
            }
            __syncthreads();
            lu_stamp(sh, 15);
        } else {
            // More than one row per thread and some: the sums and the fold in ONE pass over M, eight kept columns at a time (a column
            // walked by a single wave is 40+ dependent iterations there; the wave reductions are shared by a thread's rows here)
            for (int c0 = 0; c0 < k_new; c0 += 8) {
                double part[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) part[u] = 0.0;
                for (int s = tid; s < m; s += T) {
                    const double a = sh.x2[s];
                    const double factor = (a - (s == p ? 1.0 : 0.0)) * inv_ap;
                    double old[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) old[u] = (c0 + u < pf_k) ? M[(size_t)(c0 + u) * lu.pf_ld + s] : (s == p ? 1.0 : 0.0);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        part[u] += a * old[u];
                        if (do_update && c0 + u < k_new) M[(size_t)(c0 + u) * lu.pf_ld + s] = old[u] - factor * lane_value(mp_lane, c0 + u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (c0 + u >= pf_k) break;  // wave-uniform (the new column's sum is alpha_p: already at p's position)
                    const double w0 = wave_sum(part[u]);
                    if (lane == LAST) sh.pfpart[wave * LU_MAX_SLOTS + c0 + u] = w0;
                }
            }
            __syncthreads();
            if (tid < pf_k) {
                double sum = 0.0;
                for (int w = 0; w < nwaves; ++w) sum += sh.pfpart[w * LU_MAX_SLOTS + tid];
                sh.xt1[tid] = sum;
            }
        }
        if (tid < pf_k) {
            const int pos = s_pf_pos[tid];
            sh.x1[pos] = sh.xt1[tid];
            sh.x0[pos] = sh.st0[tid];  // (row p of M at its kept slots; p's own position holds the 1 set above unless p is kept)
        }
        __syncthreads();
        lu_stamp(sh, 7);
        const LuiSlot lower_cols = lui_load_slot(lu, 3, tid);
        if (lu.inverse_factors == 4) {
            lui_apply<2, false>(lu, 2, inv_head[2], inv_upper_cols, sh.x0, sh.x1, sh.x2, sh.x3);
            lu_stamp(sh, 8);
            lui_apply<2, true>(lu, 3, inv_head[3], lower_cols, sh.x2, sh.x3, sh.x0, sh.x1);
        } else {  // three vectors in LDS (more than ~4300 rows): the two right-hand sides through the factors one after the other
            lui_apply<1, false>(lu, 2, inv_head[2], inv_upper_cols, sh.x0, sh.x0, sh.x2, sh.x2);
            lui_apply<1, true>(lu, 3, inv_head[3], lower_cols, sh.x2, sh.x2, sh.x0, sh.x0);
            lui_apply<1, false>(lu, 2, inv_head[2], lui_load_slot(lu, 2, tid), sh.x1, sh.x1, sh.x2, sh.x2);
            lu_stamp(sh, 8);
            lui_apply<1, true>(lu, 3, inv_head[3], lui_load_slot(lu, 3, tid), sh.x2, sh.x2, sh.x1, sh.x1);
        }
        lu_stamp(sh, 10);
        if (do_update && tid == 0) {
            if (have < 0) {
                lu.pf_slot[pf_k] = p;
                lu.pf_col_of[p] = pf_k;
            }
            lu.state[LU_PF_COUNT] = k_new;
            lu.state[LU_N_UPDATES] = n_updates + 1;
        }
        if (tid == 0 && lu.state[LU_LOG_ON] != 0) {  // (past the capacity the count keeps running: the host sees that the log is incomplete)
            const int logged = lu.state[LU_LOG_COUNT];
            if (logged < LU_LOG_CAPACITY) lu.log_p[logged] = p;
            lu.state[LU_LOG_COUNT] = logged + 1;
        }
    } else {
        lu_btran_block<2, INV>(lu, sh, n_updates, epoch, [&] {
            if (do_update) eta_count = lu_build_eta(lu, sh, t, lu.spike, &new_diag);
        }, upper_record);
    }
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
        mark_rho_row(lp, rho_buf, i, r);
    }
    lu_stamp(sh, 11);
    // ---- Forrest-Tomlin update, or hand the new basis to the host ---------------------------------------------------------------
    bool refactor = !do_update;
    if (!INV && do_update) {  // (inverse-factor form: the eta went into M with the BTRAN set-up above)
        lu_ft_update_block(lu, sh, t, eta_count, new_diag, lu.spike);
        // det(B_new) = alpha_pq det(B_old)  =>  the new diagonal element must equal alpha_pq * u_tt: a free accuracy check
        const double expect = alpha_pq * diag_t;
        if (!(fabs(new_diag - expect) <= 1e-7 * (fabs(new_diag) + fabs(expect)))) refactor = true;
    }
    if (tid == 0) {
        lp.basis[p] = q;
        lp.pos[q] = p;
        if (bounded) {
            int fl = leaving_flipped;
            if (leaves_at_upper) {
                fl ^= 1;
                lp.flipped[leaving] = fl;
                ctl->flip_cost += (fl ? 1.0 : -1.0) * (xb_p + room_p) * lp.cost[leaving];
            }
            lp.pos[leaving] = fl ? -2 : -1;
            lp.xub[p] = ub_q;
        } else {
            lp.pos[leaving] = -1;
        }
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
        ctl->rho_buf = rho_buf;
        ctl->forced_q = -1;
        ctl->forced_p = -1;
        ctl->last_selected = q;
        if (refactor) ctl->status = ST_REFACTOR;
    }
    lu_stamp(sh, 12);
}

// x_B = B^-1 b  (InverseMaintainer::from_basis, carry/mod.rs:452-463) through the resident factors
template <bool INV>
__global__ void __launch_bounds__(LU_THREADS) lu_xb_kernel(DeviceLP lp, DeviceLU lu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuSlot lower_record = INV ? LuSlot{} : lu_load_slot(lu, 0, threadIdx.x);  // the first solve's record starts travelling at once
    const LuShared sh = lu_shared(smem, m, lu.max_updates, lu.inverse_factors);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear<INV>(lu, sh, n_updates, false);
    for (int i = threadIdx.x; i < m; i += blockDim.x) sh.x0[lu.rowpos[i]] = lp.rhs[i];
    __syncthreads();
    int epoch = 0;
    lu_ftran_block<INV>(lu, sh, n_updates, epoch, nullptr, lower_record);
    for (int s = threadIdx.x; s < m; s += blockDim.x) lp.xB[s] = sh.x0[lu.colpos[s]];
}
// -pi = -c_B' B^-1 and -obj = -c_B' x_B  (carry/mod.rs:226-283: the reference forms all of B^-1 with m FTRANs)
template <bool INV>
__global__ void __launch_bounds__(LU_THREADS) lu_pi_kernel(DeviceLP lp, DeviceLU lu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuSlot upper_record = INV ? LuSlot{} : lu_load_slot(lu, 2, threadIdx.x);  // the first solve's record starts travelling at once
    const LuShared sh = lu_shared(smem, m, lu.max_updates, lu.inverse_factors);
    const int n_updates = lu.state[LU_N_UPDATES];
    lu_clear<INV>(lu, sh, n_updates, false);
    double obj = 0.0;
    for (int s = threadIdx.x; s < m; s += blockDim.x) {
        const int bj = lp.basis[s];
        const double c = (lp.flipped && lp.flipped[bj]) ? -lp.cost[bj] : lp.cost[bj];  // a complemented basic column: B holds -a_j
        sh.x0[lu.colpos[s]] = c;
        obj += c * lp.xB[s];
    }
    if (lp.flipped)  // constant of the complemented variables: sum ub_j c_j (cb_kernel, kernels.hip)
        for (int j = threadIdx.x; j < lp.n; j += blockDim.x)
            if (lp.flipped[j]) obj += lp.ub[j] * lp.cost[j];
    __syncthreads();
    int epoch = 0;
    lu_btran_block<1, INV>(lu, sh, n_updates, epoch, [] {}, upper_record);
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
template <bool INV>
__global__ void __launch_bounds__(LU_THREADS) lu_gamma_kernel(DeviceLP lp, DeviceLU lu) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = lu.m;
    const LuSlot lower_record = INV ? LuSlot{} : lu_load_slot(lu, 0, threadIdx.x);  // the first solve's record starts travelling at once
    const LuShared sh = lu_shared(smem, m, lu.max_updates, lu.inverse_factors);
    const int n_updates = lu.state[LU_N_UPDATES];
    for (int j = lp.n_art + blockIdx.x; j < lp.n; j += gridDim.x) {
        if (lp.pos[j] >= 0) continue;
        __syncthreads();
        lu_clear<INV>(lu, sh, n_updates, false);
        for (int e = lp.col_start[j] + threadIdx.x; e < lp.col_start[j + 1]; e += blockDim.x) sh.x0[lu.rowpos[lp.row_index[e]]] = lp.value[e];
        __syncthreads();
        int epoch = 0;
        lu_ftran_block<INV>(lu, sh, n_updates, epoch, nullptr, lower_record);
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

static PerDeviceOnce g_lu_pivot_configured;
static void allow_full_lds(const void* kernel) {
    // dynamic + static LDS may reach the 160 KB of a CU; the attribute belongs to the kernel on the current device, so it is set
    // once per device to the maximum (not to the current LP's size: the last loaded handle would decide for every other one)
    hipFuncAttributes attr{};
    size_t fixed = 0;
    if (hipFuncGetAttributes(&attr, kernel) == hipSuccess) fixed = attr.sharedSizeBytes;
    const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - fixed));
    if (err != hipSuccess) {
        (void)hipGetLastError();  // not sticky: the launch itself reports a request that is too large
    }
}
static void configure_lu_pivot_lds() {
  g_lu_pivot_configured.run([] {
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_STEEPEST_EDGE, false>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_STEEPEST_EDGE, true>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_DANTZIG, false>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_DANTZIG, true>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_FIRST_PROFITABLE, false>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_FIRST_PROFITABLE, true>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_FIRST_PROFITABLE_MEMORY, false>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pivot_kernel<RELP_PIVOT_FIRST_PROFITABLE_MEMORY, true>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_xb_kernel<false>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_xb_kernel<true>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pi_kernel<false>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_pi_kernel<true>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_gamma_kernel<false>));
    allow_full_lds(reinterpret_cast<const void*>(&lu_gamma_kernel<true>));
  });
}
template <int RULE>
static void launch_lu_pivot_rule(const DeviceLP& d, const DeviceLU& lu, int n_price_blocks, double tol_pivot, double harris_delta,
                                 int skip_art, int mode, int refactor_period, hipStream_t s, hipEvent_t start, hipEvent_t stop) {
    auto kernel = lu.inverse_factors ? lu_pivot_kernel<RULE, true> : lu_pivot_kernel<RULE, false>;
    if (start)
        hipExtLaunchKernelGGL(kernel, dim3(1), dim3(LU_THREADS), (std::uint32_t)lu_lds_bytes_for(lu), s, start, stop, 0,
                              d, lu, n_price_blocks, tol_pivot, harris_delta, skip_art, mode, refactor_period);
    else
        hipLaunchKernelGGL(kernel, dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, d, lu, n_price_blocks, tol_pivot,
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
__global__ void lu_clear_refactor_status_kernel(Ctl* ctl) {
    if (ctl->status == ST_REFACTOR) ctl->status = ST_RUNNING;
}
void launch_clear_refactor_status(const DeviceLP& d, hipStream_t s) {
    hipLaunchKernelGGL(lu_clear_refactor_status_kernel, dim3(1), dim3(1), 0, s, d.ctl);
    check_launch("lu_clear_refactor_status_kernel");
}
void launch_lu_xb(const DeviceLP& d, const DeviceLU& lu, hipStream_t s) {
    configure_lu_pivot_lds();
    hipLaunchKernelGGL((lu.inverse_factors ? lu_xb_kernel<true> : lu_xb_kernel<false>), dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, d, lu);
    check_launch("lu_xb_kernel");
}
void launch_lu_pi(const DeviceLP& d, const DeviceLU& lu, hipStream_t s) {
    configure_lu_pivot_lds();
    hipLaunchKernelGGL((lu.inverse_factors ? lu_pi_kernel<true> : lu_pi_kernel<false>), dim3(1), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, d, lu);
    check_launch("lu_pi_kernel");
}
void launch_lu_gamma(const DeviceLP& d, const DeviceLU& lu, hipStream_t s) {
    configure_lu_pivot_lds();
    const int blocks = std::max(1, std::min(512, d.n - d.n_art));
    hipLaunchKernelGGL((lu.inverse_factors ? lu_gamma_kernel<true> : lu_gamma_kernel<false>), dim3(blocks), dim3(LU_THREADS), lu_lds_bytes_for(lu), s, d, lu);
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
    // `LUDecomposition::rows` runs as a KERNEL (lu_factor.hip): the factors this object solves with -- and hands out through
    // relp_bi_get_factors, where the reference's exact-factor and Forrest-Tomlin known answers are checked -- are the ones the device
    // produced.  Their level schedules and slot lists are still laid out by the host (LuFactors::upload) for this carry.  What the
    // kernel does not take (a row of more than 256 entries) is factorised by lu_factor on the host.
    HostLU f;
    if (!factor_on_device(cs, rows, vals, f)) f = lu_factor(m_, cs.data(), rows.data(), vals.data(), options_);
    if (f.singular) throw std::runtime_error("singular basis");
    lu_.upload(f, period_ + 1, stream_);
    RELP_HIP(hipStreamSynchronize(stream_));
    have_spike_ = false;
}
// false: the kernel gave up for a reason other than singularity (the caller factorises on the host)
bool LuBasis::factor_on_device(const std::vector<int>& cs, const std::vector<int>& rows, const std::vector<double>& vals, HostLU& f) {
    if (thread_tuning().has(RELP_SW_BI_FACTOR_HOST) || m_ > 65535) return false;
    const int m = m_;
    const size_t nnz = rows.size();
    size_t cap = 4 * nnz + 8 * (size_t)m + 4096;
    struct Buffers {
        std::vector<void*> owned;
        ~Buffers() { for (void* p : owned) (void)hipFree(p); }
        void* bytes(size_t n) {
            void* p = nullptr;
            RELP_HIP(hipMalloc(&p, std::max<size_t>(8, n)));
            owned.push_back(p);
            return p;
        }
    };
    for (int attempt = 0; attempt < 4; ++attempt, cap *= 4) {
        Buffers b;
        int* d_cs = (int*)b.bytes((size_t)(m + 1) * sizeof(int));
        int* d_ri = (int*)b.bytes((size_t)(nnz) * sizeof(int));
        double* d_va = (double*)b.bytes((size_t)(nnz) * sizeof(double));
        LuFactorOut out;
        out.rowpos = (int*)b.bytes((size_t)(m) * sizeof(int)); out.colpos = (int*)b.bytes((size_t)(m) * sizeof(int)); out.diag = (double*)b.bytes((size_t)(m) * sizeof(double));
        out.l_start = (int*)b.bytes((size_t)(m + 1) * sizeof(int)); out.l_col = (int*)b.bytes((size_t)(cap) * sizeof(int)); out.l_val = (double*)b.bytes((size_t)(cap) * sizeof(double));
        out.u_start = (int*)b.bytes((size_t)(m + 1) * sizeof(int)); out.u_col = (int*)b.bytes((size_t)(cap) * sizeof(int)); out.u_val = (double*)b.bytes((size_t)(cap) * sizeof(double));
        out.cap_l = out.cap_u = (int)std::min<size_t>(cap, (size_t)1 << 30);
        RELP_HIP(hipMemcpyAsync(d_cs, cs.data(), (m + 1) * sizeof(int), hipMemcpyHostToDevice, stream_));
        if (nnz) {
            RELP_HIP(hipMemcpyAsync(d_ri, rows.data(), nnz * sizeof(int), hipMemcpyHostToDevice, stream_));
            RELP_HIP(hipMemcpyAsync(d_va, vals.data(), nnz * sizeof(double), hipMemcpyHostToDevice, stream_));
        }
        factor_scratch_.reserve(m, nnz, cap, cap);
        LuFactorSource src;
        src.col_start = d_cs;
        src.row_index = d_ri;
        src.value = d_va;
        launch_lu_factor(src, factor_scratch_.work(), out, options_.threshold, options_.reference_ties ? 1 : 0, 32, stream_);
        int info[LUF_INFO_WORDS];
        RELP_HIP(hipMemcpyAsync(info, factor_scratch_.work().info, sizeof(info), hipMemcpyDeviceToHost, stream_));
        RELP_HIP(hipStreamSynchronize(stream_));
        f = HostLU{};
        f.m = m;
        if (info[LUF_STATUS] == LUF_ERR_SINGULAR) {
            f.singular = true;
            return true;
        }
        if (info[LUF_STATUS] == LUF_ERR_L_CAPACITY || info[LUF_STATUS] == LUF_ERR_U_CAPACITY || info[LUF_STATUS] == LUF_ERR_ARENA) continue;  // more room
        if (info[LUF_STATUS] != LUF_OK) return false;
        const size_t nl = (size_t)info[LUF_NNZ_L], nu = (size_t)info[LUF_NNZ_U];
        f.rowpos.resize(m); f.colpos.resize(m); f.diag.resize(m);
        f.l_start.resize(m + 1); f.u_start.resize(m + 1);
        f.l_col.resize(nl); f.l_val.resize(nl); f.u_col.resize(nu); f.u_val.resize(nu);
        auto get = [&](void* dst, const void* from, size_t bytes) {
            if (bytes) RELP_HIP(hipMemcpyAsync(dst, from, bytes, hipMemcpyDeviceToHost, stream_));
        };
        get(f.rowpos.data(), out.rowpos, m * sizeof(int));
        get(f.colpos.data(), out.colpos, m * sizeof(int));
        get(f.diag.data(), out.diag, m * sizeof(double));
        get(f.l_start.data(), out.l_start, (m + 1) * sizeof(int));
        get(f.u_start.data(), out.u_start, (m + 1) * sizeof(int));
        get(f.l_col.data(), out.l_col, nl * sizeof(int));
        get(f.l_val.data(), out.l_val, nl * sizeof(double));
        get(f.u_col.data(), out.u_col, nu * sizeof(int));
        get(f.u_val.data(), out.u_val, nu * sizeof(double));
        RELP_HIP(hipStreamSynchronize(stream_));
        return true;
    }
    return false;
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
    std::vector<int> ucs = geti(d.u_cstart, m + 1);
    const size_t used = (size_t)ucs[m];
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
            for (int e = ucs[j]; e < ucs[j + 1]; ++e)
                if (slot_of[ucrow[e]] < 0) col.push_back({rank[ucrow[e]], ucval[e]});  // (the row of a replaced position has left U: masked, not removed)
        } else {
            const int k = slot_of[j];
            for (int e = scs[k]; e < scs[k] + scl[k]; ++e)
                if (slot_of[scrow[e]] < 0) col.push_back({rank[scrow[e]], scval[e]});  // (a row replaced since: masked)
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
    std::vector<double> ev = getd(d.eta_val, state[LU_ETA_TOP]), mf = getd(d.eta_mf, (size_t)d.max_updates * d.ldt);
    std::vector<int> r(m);
    for (int i = 0; i < m; ++i) r[i] = i;
    f.eta_start.assign(1, 0);
    for (int k = 0; k < n_updates; ++k) {
        std::vector<std::pair<int, double>> entries;
        for (int e = es[k]; e < es[k + 1]; ++e) entries.push_back({r[ei[e]], ev[e]});
        for (int i = 0; i < k; ++i)  // the entries held in the chaining matrix: positions an earlier eta pivots on (not its own)
            if (mf[(size_t)k * d.ldt + i] != 0.0 && ep[i] != ep[k]) entries.push_back({r[ep[i]], mf[(size_t)k * d.ldt + i]});
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
