// HIP kernels of the revised-simplex inner loop for gfx950 (wave64, 256 CUs in 8 XCDs, 160 KB LDS per CU).
//
// One pivot = K1 price (+steepest-edge update) -> K2 entering column / FTRAN / ratio test / x_B update
//           -> K3 inverse update fused with w, rho_p and the -pi update.
// All state stays in HBM / L2; every kernel starts by reading the control word, so that a launch sequence
// enqueued past the end of a phase degenerates into no-ops (no host round trip per pivot).
//
// Layout of the explicit inverse: COLUMN-major.  `T[j*ld + i] = Binv(i, j)`.  With it
//   * FTRAN  alpha = sum_k v_k Binv(:, r_k)          reads nnz(a_q) contiguous columns      (K2, coalesced)
//   * the update of column j, w_j, rho_p[j], -pi_j   touch one contiguous column each       (K3, one wave per column)
// so no kernel gathers with a stride (the row-major variant spent 20 us per pivot in strided FTRAN gathers).
//
// Reference functions each kernel replaces are cited at the kernel; paths relative to
// /root/reference/src/algorithm/two_phase/.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdlib>
#include <string>
#include <utility>

#include "solver.hpp"
#include "wave_ops.hpp"

namespace relp {


// ---------------------------------------------------------------------------------------------------
// K0: budget for the next batch of pivots (first node of every batch / graph)
// ---------------------------------------------------------------------------------------------------
__global__ void budget_kernel(Ctl* ctl, long long add) {
    if (ctl->status == ST_BUDGET) ctl->status = ST_RUNNING;
    ctl->budget = ctl->iters + add;
}

// ---------------------------------------------------------------------------------------------------
// K1: fused pricing pass over the non-basic columns.
//   replaces  SteepestDescentAlongObjective::select_primal_pivot_column   strategy/pivot_rule.rs:221-241
//             Tableau::relative_cost / Carry::cost_difference             tableau/mod.rs:106-112, carry/mod.rs:606-611
//             SteepestDescentAlongObjective::after_basis_update            strategy/pivot_rule.rs:243-296
//             MatrixData::column (no per-column clone)                     matrix_provider/matrix_data.rs:291-329
// LPC lanes share one sparse column (entries strided over the lanes, shuffle-reduced), so that the dependent
// index -> LDS gather chain is 1-3 steps long whatever the column length; -pi, rho_p and w are staged in LDS once
// per workgroup.  Traffic is the column data itself: nnz*(8+4) + 3*8 bytes per column.
// The reference makes TWO passes over A per pivot (pricing, weight update) and clones every column twice.
// ---------------------------------------------------------------------------------------------------
typedef double f64x2_t __attribute__((ext_vector_type(2)));
// UNIT (graph LPs, LPC == 2): the columns are GENERATED from the arcs' endpoints instead of streamed -- MatrixProvider::column(j)
// of examples/max_flow.rs:174-200 evaluated on the device: `ell_rows` holds 8 bytes per arc (row of each end, bit 31 = the -1
// end, 0x7fffffff = that end is s or t: no entry), every value is +-1 and the cost a signed byte (-1 on the arcs leaving s).
template <int RULE, bool USE_LDS, int LPC, bool UNIT = false>
__global__ void __launch_bounds__(256) price_kernel(DeviceLP lp, int skip_weights, double tol_dual, int col_first,
                                                    int col_last, int cand_offset) {
    static_assert(!UNIT || LPC == 2, "generated columns are incidence columns");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ Cand s_cand[8];
    static_assert(LPC == ELL_W || LPC == 2, "one lane per padded entry (width 2 when no column has more than two entries: graph LPs)");
    constexpr int CPB = 256 / LPC;  // columns per workgroup pass
    Ctl* ctl = lp.ctl;
    const int m = lp.m;
    const int g = threadIdx.x / LPC, sub = threadIdx.x % LPC;
    // ---- ONE memory round trip: control word, this workgroup's first columns, and the three vectors ----------
    const int status = ctl->status;
    const int pending = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? (ctl->pending && !skip_weights) : 0;
    const double gamma_q = ctl->gamma_q;
    const double alpha_pq = ctl->alpha_pq;
    const int leaving = ctl->leaving;
    const int last = ctl->last_selected;
    int base = col_first + blockIdx.x * CPB;
    int a = 0, b = 0, r0 = 0;
    double v0 = 0.0, cost_j = 0.0, g_j = 1.0;
    bool nonbasic = false;
    double sgn_j = 1.0;  // -1: the column is held in complemented form (implicit upper bounds)
    auto load_column = [&](int j) {
        a = b = r0 = 0;
        v0 = cost_j = 0.0;
        g_j = 1.0;
        nonbasic = false;
        sgn_j = 1.0;
        if (j < col_last) {
            const int pos_j = lp.pos[j];
            nonbasic = pos_j == -1 || pos_j == -2;  // -3: fixed variable (implicit bounds), never priced
            sgn_j = pos_j == -2 ? -1.0 : 1.0;
            if (LPC != 2) {  // width 2: no column is longer than the padded copy, the CSC is not needed
                a = lp.col_start[j];
                b = lp.col_start[j + 1];
            }
            if (UNIT) {
                const unsigned code = (unsigned)lp.ell_rows[(size_t)j * LPC + sub];  // (non-temporal loads of the three streams: measured, no gain)
                const bool absent = code == 0x7fffffffu;
                r0 = absent ? 0 : (int)(code & 0x7fffffffu);
                v0 = absent ? 0.0 : ((code >> 31) ? -1.0 : 1.0);
                cost_j = (double)lp.cost8[j];
            } else {
                r0 = lp.ell_rows[(size_t)j * LPC + sub];
                v0 = lp.ell_vals[(size_t)j * LPC + sub];
                cost_j = lp.cost[j];
            }
            // (generated columns: the weight is fetched only by the columns that need it -- candidates and columns with an
            //  entry in the pivot row -- 16 of the 29 bytes per arc otherwise)
            if (RULE == RELP_PIVOT_STEEPEST_EDGE && !UNIT) g_j = lp.gamma[j];
        }
    };
    load_column(base + g);
    const double* v_pi = lp.minus_pi;
    const double* v_rho = lp.rho;
    const double* v_w = lp.w;
    if (USE_LDS) {
        double* s_pi = smem;
        double* s_rho = smem + m;
        double* s_w = smem + 2 * m;
        for (int i0 = threadIdx.x; i0 < m; i0 += 4 * 256) {
            double t_pi[4], t_rho[4], t_w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                t_pi[u] = i < m ? lp.minus_pi[i] : 0.0;
                t_rho[u] = i < m ? lp.rho[i] : 0.0;
                t_w[u] = i < m ? lp.w[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                if (i < m) {
                    s_pi[i] = t_pi[u];
                    s_rho[i] = t_rho[u];
                    s_w[i] = t_w[u];
                }
            }
        }
        v_pi = s_pi;
        v_rho = s_rho;
        v_w = s_w;
    }
    if (status != ST_RUNNING) return;  // uniform: every thread read the same word
    if (USE_LDS) __syncthreads();

    Cand best;
    best.key = 0.0;
    best.idx = -1;
    best.aux = 0;
    double best_cbar = 0.0;
    int best_row = 0, best_len = 0;  // this lane's padded entry of the group's best column so far
    double best_val = 0.0;
    while (base < col_last) {
        const int j = base + g;
        if (!nonbasic) { b = a; v0 = 0.0; }
        double d_pi, d_rho = 0.0, d_w = 0.0;
        if (UNIT) {
            // Generated columns (config 5: 65 534 rows, a million arcs): the 32-byte records are 2 MB that each of the eight L2s
            // pulls in a 128-byte line at a time -- more bytes than the arcs themselves (PMC 39 MB per pass for 13.6 MB of arcs).
            // -pi comes from its own 8-byte vector, and rho_p -- row p of the inverse, a handful of non-zeros on these bases --
            // is looked up in a byte per row first; rho and w are fetched only by the columns that have an entry there (both
            // entries of such a column fetch w: the sum is the one the packed gather made).
            d_pi = v0 * lp.minus_pi[r0];
            if (pending) {
                int hit = lp.rho_nz[r0] != 0 && v0 != 0.0;
                hit |= __shfl_xor(hit, 1);
                if (hit) {
                    d_rho = v0 * lp.rho[r0];
                    d_w = v0 * lp.w[r0];
                }
            }
        } else if (LPC == 2) {  // (-pi_r, rho_r, w_r) packed per row: one 32-byte gather instead of three from three cache lines
            const double* t = lp.prw + (size_t)4 * r0;
            if (pending) {
                const f64x2_t lo = *reinterpret_cast<const f64x2_t*>(t);
                d_w = v0 * t[2];
                d_pi = v0 * lo.x;
                d_rho = v0 * lo.y;
            } else {
                d_pi = v0 * t[0];
            }
        } else {
            d_pi = v0 * v_pi[r0];
            if (pending) {
                d_rho = v0 * v_rho[r0];
                d_w = v0 * v_w[r0];
            }
        }
        for (int e = a + LPC + sub; e < b; e += LPC) {  // columns longer than the padded width (rare)
            const int r = lp.row_index[e];
            const double v = lp.value[e];
            d_pi += v * v_pi[r];
            if (pending) {
                d_rho += v * v_rho[r];
                d_w += v * v_w[r];
            }
        }
        // 8-lane group sums with DPP moves (quad_perm, quad_perm, row_half_mirror): every lane of the group gets the total
        d_pi += dpp_f64<DPP_QUAD_1032, 0xF>(0.0, d_pi);
        if (LPC > 2) {
            d_pi += dpp_f64<DPP_QUAD_2301, 0xF>(0.0, d_pi);
            d_pi += dpp_f64<DPP_ROW_HALF_MIRROR, 0xF>(0.0, d_pi);
        }
        if (RULE == RELP_PIVOT_STEEPEST_EDGE) {
            d_rho += dpp_f64<DPP_QUAD_1032, 0xF>(0.0, d_rho);
            d_w += dpp_f64<DPP_QUAD_1032, 0xF>(0.0, d_w);
            if (LPC > 2) {
                d_rho += dpp_f64<DPP_QUAD_2301, 0xF>(0.0, d_rho);
                d_rho += dpp_f64<DPP_ROW_HALF_MIRROR, 0xF>(0.0, d_rho);
                d_w += dpp_f64<DPP_QUAD_2301, 0xF>(0.0, d_w);
                d_w += dpp_f64<DPP_ROW_HALF_MIRROR, 0xF>(0.0, d_w);
            }
        }
        int improved = 0;
        const double cbar = sgn_j * (cost_j + d_pi);
        // a weight with no entry in the pivot row does not change (gamma - 0 + 0, and gamma >= 1): not read, not written
        const bool weight_changes = RULE == RELP_PIVOT_STEEPEST_EDGE && pending && (!UNIT || j == leaving || d_rho != 0.0);
        if (sub == 0 && nonbasic && (!UNIT || weight_changes || cbar < -tol_dual)) {
            double gam = g_j;
            if (UNIT && RULE == RELP_PIVOT_STEEPEST_EDGE) gam = lp.gamma[j];
            if (weight_changes) {
                if (j == leaving) {
                    gam = gamma_q / (alpha_pq * alpha_pq);  // pivot_rule.rs:294-295
                } else {
                    const double sq = d_rho * d_rho;  // pivot_rule.rs:262-288 (Goldfarb-Reid)
                    gam = gam - 2.0 * d_rho * d_w + sq * gamma_q;
                    gam = fmax(gam, 1.0 + sq);
                }
                lp.gamma[j] = gam;
            }
            bool candidate = cbar < -tol_dual;
            Cand c;
            c.idx = j;
            c.aux = 0;
            c.key = 0.0;
            if (RULE == RELP_PIVOT_STEEPEST_EDGE) c.key = cbar * cbar / gam;
            else if (RULE == RELP_PIVOT_DANTZIG) c.key = -cbar;
            else if (RULE == RELP_PIVOT_FIRST_PROFITABLE) c.key = -(double)j;
            else {
                if (last >= 0 && j == last) candidate = false;
                const long long rank = (last < 0) ? j : (j > last ? (long long)j - last - 1 : (long long)j + lp.n - last);
                c.key = -(double)rank;
            }
            if (candidate) {
                Cand nb = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? better<TIE_LARGER_IDX>(best, c) : better<TIE_SMALLER_IDX>(best, c);
                if (nb.idx == j) {
                    best_cbar = cbar;
                    improved = 1;
                }
                best = nb;
            }
        }
        improved = __shfl(improved, threadIdx.x & (WAVE - 1) & ~(LPC - 1));  // the group's lane 0 decides
        if (improved) {
            best_row = r0;
            best_val = v0;
            best_len = LPC == 2 ? 2 : b - a;
        }
        base += gridDim.x * CPB;
        if (base < col_last) load_column(base + g);
    }
    Cand blk = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? block_best<TIE_LARGER_IDX>(best, s_cand)
                                                  : block_best<TIE_SMALLER_IDX>(best, s_cand);
    int winner = (sub == 0 && blk.idx >= 0 && blk.idx == best.idx) ? 1 : 0;
    if (winner) {  // the winning thread publishes
        lp.cand_key[cand_offset + blockIdx.x] = blk.key;
        lp.cand_j[cand_offset + blockIdx.x] = blk.idx;
        lp.cand_cbar[cand_offset + blockIdx.x] = best_cbar;
        lp.cand_len[cand_offset + blockIdx.x] = best_len;
    }
    winner = __shfl(winner, threadIdx.x & (WAVE - 1) & ~(LPC - 1));
    if (winner) {  // its group publishes the column's padded entries
        lp.cand_rows[(size_t)(cand_offset + blockIdx.x) * ELL_W + sub] = best_row;
        lp.cand_vals[(size_t)(cand_offset + blockIdx.x) * ELL_W + sub] = best_val;
    }
    if (blk.idx < 0 && threadIdx.x == 0) lp.cand_j[cand_offset + blockIdx.x] = -1;
}

constexpr int PRICE_UNIT_ARCS = 4;  // arcs per lane of price_unit_kernel
// ---------------------------------------------------------------------------------------------------
// K1u: the pricing pass over GENERATED incidence columns (graph providers: examples/max_flow.rs:174-200, every value +-1, integer
// costs), one LANE per arc and PRICE_UNIT_ARCS arcs per lane.  The two-lanes-per-arc form above kept one arc per lane pair in
// flight and walked four dependent passes per workgroup (14 us for 13.6 MB on config 5: 12 % of HBM, all of it latency); here a
// lane loads the 8 bytes of four arcs, their positions and cost bytes in one round trip, gathers the eight -pi entries in the
// next, and a million arcs are ONE pass of 1024 workgroups.  Same arithmetic, same order of every sum (a column's two
// products are added as the lane pair added them), same total order on the candidates: the pivot sequence does not change.
// rho_p and w are fetched only where row p of the inverse is non-zero: a bit per row, copied into LDS (DeviceLP::rho_bits; the
// byte table rho_nz of price_kernel when the LP has more rows than bits fit).  A gather costs the L1 a cycle per lane whatever it
// brings: with the byte table beside -pi the pass spent 6-7 of its 14 us on 4 M lane gathers (tools/stamps_maxflow.py).
// ---------------------------------------------------------------------------------------------------
template <int RULE>
__global__ void __launch_bounds__(256) price_unit_kernel(DeviceLP lp, int skip_weights, double tol_dual, int col_first,
                                                         int col_last, int cand_offset) {
    constexpr int U = PRICE_UNIT_ARCS;
    extern __shared__ __attribute__((aligned(16))) unsigned s_rho_bits[];  // lp.rho_words words (none: the byte table is gathered)
    __shared__ Cand s_cand[8];
    Ctl* ctl = lp.ctl;
#ifdef RELP_STAMPS  // (tools/stamps_maxflow.py: wall-clock ticks of 10 ns, first workgroup in dbg[32..], last in dbg[40..])
    unsigned long long t_prev__ = wall_clock64();
    const int stamp_base__ = blockIdx.x == 0 ? 32 : (blockIdx.x == gridDim.x - 1 ? 40 : -1);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        lp.dbg[48] += 1;
        *(volatile unsigned long long*)(lp.dbg + 49) = t_prev__;
    }
#define USTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0 && stamp_base__ >= 0) { unsigned long long t__ = wall_clock64(); lp.dbg[stamp_base__ + (k)] += t__ - t_prev__; t_prev__ = t__; } } while (0)
#else
#define USTAMP(k) do {} while (0)
#endif
    const int status = ctl->status;
    const int pending = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? (ctl->pending && !skip_weights) : 0;
    const double gamma_q = ctl->gamma_q;
    const double alpha_pq = ctl->alpha_pq;
    const int leaving = ctl->leaving;
    const int last = ctl->last_selected;
    // ---- ONE memory round trip: the control word and this workgroup's first arcs, positions and cost bytes ----------
    const uint2* arcs = reinterpret_cast<const uint2*>(lp.ell_rows);
    unsigned ca[U], cb[U];
    int pos[U];
    double cost[U];
    auto load_arcs = [&](int base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u * 256 + threadIdx.x;
            ca[u] = cb[u] = 0x7fffffffu;
            pos[u] = 0;
            cost[u] = 0.0;
            if (j < col_last) {
                const uint2 c = arcs[j];
                ca[u] = c.x;
                cb[u] = c.y;
                pos[u] = lp.pos[j];
                cost[u] = (double)lp.cost8[j];
            }
        }
    };
    load_arcs(col_first + blockIdx.x * (256 * U));
    // ... and the bits of rho_p's non-zero rows (see DeviceLP::rho_bits): this pivot's half into LDS, the other half cleared for the next
    const int rho_words = lp.rho_words;
    if (rho_words) {
        const uint4* mine = reinterpret_cast<const uint4*>(lp.rho_bits + (size_t)ctl->rho_buf * rho_words);
        uint4* other = reinterpret_cast<uint4*>(lp.rho_bits + (size_t)(ctl->rho_buf ^ 1) * rho_words);
        if (pending)
            for (int i = threadIdx.x; i < rho_words / 4; i += 256) reinterpret_cast<uint4*>(s_rho_bits)[i] = mine[i];
        for (int i = blockIdx.x * 256 + threadIdx.x; i < rho_words / 4; i += gridDim.x * 256) other[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (status != ST_RUNNING) return;  // uniform
    if (rho_words && pending) __syncthreads();
    USTAMP(0);
    Cand best;
    best.key = 0.0;
    best.idx = -1;
    best.aux = 0;
    double best_cbar = 0.0;
    int best_ra = 0, best_rb = 0;
    double best_va = 0.0, best_vb = 0.0;
    for (int base = col_first + blockIdx.x * (256 * U); base < col_last; base += gridDim.x * (256 * U)) {
        if (base != col_first + (int)blockIdx.x * (256 * U)) load_arcs(base);
        USTAMP(1);
        int ra[U], rb[U];
        double va[U], vb[U], pa[U], pb[U];
        bool nonbasic[U];
        int hit[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            nonbasic[u] = pos[u] == -1 || pos[u] == -2;  // -3: fixed variable (implicit bounds), never priced
            const bool has_a = nonbasic[u] && ca[u] != 0x7fffffffu, has_b = nonbasic[u] && cb[u] != 0x7fffffffu;
            ra[u] = ca[u] == 0x7fffffffu ? 0 : (int)(ca[u] & 0x7fffffffu);
            rb[u] = cb[u] == 0x7fffffffu ? 0 : (int)(cb[u] & 0x7fffffffu);
            va[u] = ca[u] == 0x7fffffffu ? 0.0 : ((ca[u] >> 31) ? -1.0 : 1.0);
            vb[u] = cb[u] == 0x7fffffffu ? 0.0 : ((cb[u] >> 31) ? -1.0 : 1.0);
            pa[u] = has_a ? lp.minus_pi[ra[u]] : 0.0;
            pb[u] = has_b ? lp.minus_pi[rb[u]] : 0.0;
            hit[u] = 0;
            if (pending && rho_words)
                hit[u] = (has_a && ((s_rho_bits[ra[u] >> 5] >> (ra[u] & 31)) & 1u)) | (has_b && ((s_rho_bits[rb[u] >> 5] >> (rb[u] & 31)) & 1u));
            else if (pending)
                hit[u] = (has_a && lp.rho_nz[ra[u]] != 0) | (has_b && lp.rho_nz[rb[u]] != 0);
        }
        USTAMP(2);
        double cbar[U], d_rho[U], d_w[U], gam[U];
        bool weight_changes[U], wanted[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u * 256 + threadIdx.x;
            const bool has_a = nonbasic[u] && ca[u] != 0x7fffffffu, has_b = nonbasic[u] && cb[u] != 0x7fffffffu;
            const double sgn = pos[u] == -2 ? -1.0 : 1.0;  // the column is held in complemented form (implicit upper bounds)
            cbar[u] = sgn * (cost[u] + ((has_a ? va[u] * pa[u] : 0.0) + (has_b ? vb[u] * pb[u] : 0.0)));
            d_rho[u] = d_w[u] = 0.0;
            if (hit[u]) {  // (rare: the columns with an entry where rho_p is non-zero)
                d_rho[u] = (has_a ? va[u] * lp.rho[ra[u]] : 0.0) + (has_b ? vb[u] * lp.rho[rb[u]] : 0.0);
                d_w[u] = (has_a ? va[u] * lp.w[ra[u]] : 0.0) + (has_b ? vb[u] * lp.w[rb[u]] : 0.0);
            }
            // a weight with no entry in the pivot row does not change (gamma - 0 + 0, and gamma >= 1): not read, not written
            weight_changes[u] = RULE == RELP_PIVOT_STEEPEST_EDGE && pending && (j == leaving || d_rho[u] != 0.0);
            wanted[u] = nonbasic[u] && (weight_changes[u] || cbar[u] < -tol_dual);
            gam[u] = 1.0;
            if (RULE == RELP_PIVOT_STEEPEST_EDGE && wanted[u]) gam[u] = lp.gamma[j];
        }
        USTAMP(3);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u * 256 + threadIdx.x;
            if (!wanted[u]) continue;
            double g = gam[u];
            if (weight_changes[u]) {
                if (j == leaving) {
                    g = gamma_q / (alpha_pq * alpha_pq);  // pivot_rule.rs:294-295
                } else {
                    const double sq = d_rho[u] * d_rho[u];  // pivot_rule.rs:262-288 (Goldfarb-Reid)
                    g = g - 2.0 * d_rho[u] * d_w[u] + sq * gamma_q;
                    g = fmax(g, 1.0 + sq);
                }
                lp.gamma[j] = g;
            }
            bool candidate = cbar[u] < -tol_dual;
            Cand c;
            c.idx = j;
            c.aux = 0;
            c.key = 0.0;
            if (RULE == RELP_PIVOT_STEEPEST_EDGE) c.key = cbar[u] * cbar[u] / g;
            else if (RULE == RELP_PIVOT_DANTZIG) c.key = -cbar[u];
            else if (RULE == RELP_PIVOT_FIRST_PROFITABLE) c.key = -(double)j;
            else {
                if (last >= 0 && j == last) candidate = false;
                const long long rank = (last < 0) ? j : (j > last ? (long long)j - last - 1 : (long long)j + lp.n - last);
                c.key = -(double)rank;
            }
            if (candidate) {
                const Cand nb = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? better<TIE_LARGER_IDX>(best, c) : better<TIE_SMALLER_IDX>(best, c);
                if (nb.idx == j) {
                    best_cbar = cbar[u];
                    best_ra = ra[u];
                    best_rb = rb[u];
                    best_va = va[u];
                    best_vb = vb[u];
                }
                best = nb;
            }
        }
    }
    USTAMP(4);
    const Cand blk = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? block_best<TIE_LARGER_IDX>(best, s_cand)
                                                        : block_best<TIE_SMALLER_IDX>(best, s_cand);
    if (blk.idx >= 0 && blk.idx == best.idx) {  // the winning thread publishes the candidate and its column's two entries
        const size_t slot = (size_t)(cand_offset + blockIdx.x);
        lp.cand_key[slot] = blk.key;
        lp.cand_j[slot] = blk.idx;
        lp.cand_cbar[slot] = best_cbar;
        lp.cand_len[slot] = 2;
        lp.cand_rows[slot * ELL_W] = best_ra;
        lp.cand_rows[slot * ELL_W + 1] = best_rb;
        lp.cand_vals[slot * ELL_W] = best_va;
        lp.cand_vals[slot * ELL_W + 1] = best_vb;
    }
    if (blk.idx < 0 && threadIdx.x == 0) lp.cand_j[cand_offset + blockIdx.x] = -1;
    USTAMP(5);
#ifdef RELP_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) lp.dbg[50] += wall_clock64() - *(volatile unsigned long long*)(lp.dbg + 49);
#endif
#undef USTAMP
}

// ---------------------------------------------------------------------------------------------------
// K1d: the same pricing pass for DENSE columns (BASELINE config 3: m = 4096, n = 8192, 268 MB of f64 per pass).
// Columns are stored dense, column-major, without row indices (8 B per entry instead of 12); one wave streams one
// column with 16-byte loads (1 KiB per wave instruction), 4 loads in flight per lane; -pi, rho_p, w live in LDS
// (3 x 32 KB at m = 4096).  This is the HBM-roofline kernel: algorithmic bytes = non-basic dense columns * m * 8.
// ---------------------------------------------------------------------------------------------------
constexpr int K1D_THREADS = 1024;
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// F32: the dense block is stored as float because every entry is exactly representable in it (checked at upload); the
// widening back is exact and all arithmetic stays f64, so results are bit-identical while the pass streams half the bytes.
// I8: the block as signed bytes when every entry is an integer in [-128, 127] (a quarter of the f32 stream).  Stored in
// chunks of 1024 rows, bytes t, t+1 (t even) of lane l's 16-byte piece holding rows (t/2)*128 + 2l + {0, 1} of the chunk: one
// 16-byte load per lane brings 16 entries, and for every byte pair the 64 lanes read 64 consecutive double2 of -pi / rho / w
// from LDS (16-byte reads, no bank conflicts).
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int K1D_I8_CHUNK = 1024;
template <bool F32, bool I8 = false>
__global__ void __launch_bounds__(K1D_THREADS) price_dense_kernel(DeviceLP lp, int skip_weights, double tol_dual, int cand_offset) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ Cand s_cand[K1D_THREADS / WAVE + 2];
    __shared__ double s_cbar[K1D_THREADS / WAVE];
    Ctl* ctl = lp.ctl;
    if (ctl->status != ST_RUNNING) return;
    const int m = lp.m;
    const int mp = lp.dense_ld;  // padded to a multiple of 2
    const int pending = ctl->pending && !skip_weights;
    const double gamma_q = ctl->gamma_q;
    const double alpha_pq = ctl->alpha_pq;
    const int leaving = ctl->leaving;
    double* s_pi = smem;
    double* s_rho = smem + mp;
    double* s_w = smem + 2 * mp;
    for (int i = threadIdx.x; i < mp; i += K1D_THREADS) {
        s_pi[i] = i < m ? lp.minus_pi[i] : 0.0;
        s_rho[i] = (pending && i < m) ? lp.rho[i] : 0.0;
        s_w[i] = (pending && i < m) ? lp.w[i] : 0.0;
    }
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int waves_total = gridDim.x * (K1D_THREADS / WAVE);
    Cand best;
    best.key = 0.0;
    best.idx = -1;
    best.aux = 0;
    double best_cbar = 0.0;
    const int half = mp / 2;
    auto finish_column = [&](int j, int pos_j, double gamma_j, double cost_j, double d_pi, double d_rho, double d_w) {
        d_pi = wave_sum(d_pi);
        d_rho = wave_sum(d_rho);
        d_w = wave_sum(d_w);
        if (lane == LAST) {
            double gam = gamma_j;
            if (pending) {
                if (j == leaving) {
                    gam = gamma_q / (alpha_pq * alpha_pq);
                } else {
                    const double sq = d_rho * d_rho;
                    gam = gam - 2.0 * d_rho * d_w + sq * gamma_q;
                    gam = fmax(gam, 1.0 + sq);
                }
                lp.gamma[j] = gam;
            }
            const double cbar = (pos_j == -2 ? -1.0 : 1.0) * (cost_j + d_pi);
            if (cbar < -tol_dual) {
                Cand c;
                c.idx = j;
                c.aux = 0;
                c.key = cbar * cbar / gam;
                Cand nb = better<TIE_LARGER_IDX>(best, c);
                if (nb.idx == j) best_cbar = cbar;
                best = nb;
            }
        }
    };
    if (I8) {
        // One 16-byte load per lane = 16 entries; per entry one widening and three f64 FMAs against -pi / rho / w from LDS
        // (16-byte LDS reads).  33 MB instead of 131 MB per pass at 4096 x 8192, and the pass is then bound by the 800 MB
        // that come out of LDS, not by HBM.  Measured and dropped (all spill at the 128 VGPRs a 1024-thread workgroup
        // allows, or lose the loads in flight): two columns per wave to halve the LDS reads (33.8 us), prefetching the wave's
        // next column (25.6 us), a rolled chunk loop (23.4 us), the 2^52 widening trick (22.3 us) -- against 19.2 us.
        const int chunks = mp / K1D_I8_CHUNK;
        const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
        const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
        const double2* w2 = reinterpret_cast<const double2*>(s_w);
        for (int jd = blockIdx.x * (K1D_THREADS / WAVE) + wave; jd < lp.n_dense; jd += waves_total) {
            const int j = lp.dense_first + jd;
            const int pos_j = lp.pos[j];
            const double gamma_j = lp.gamma[j];
            const double cost_j = lp.cost[j];
            if (pos_j >= 0) continue;  // wave-uniform
            const i32x4* col = reinterpret_cast<const i32x4*>(lp.dense_val8 + (size_t)jd * mp);
            double p0 = 0.0, p1 = 0.0, r0 = 0.0, r1 = 0.0, w0 = 0.0, w1 = 0.0;  // two chains per sum
            for (int c0 = 0; c0 < chunks; c0 += 4) {
                i32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = c0 + u < chunks ? __builtin_nontemporal_load(col + (size_t)(c0 + u) * WAVE + lane) : i32x4{0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c0 + u >= chunks) continue;
                    const int base2 = (c0 + u) * (K1D_I8_CHUNK / 2) + lane;  // in double2 units
#pragma unroll
                    for (int t = 0; t < 16; t += 2) {  // bytes t, t+1 <-> rows (t/2)*128 + 2*lane + {0, 1}: one 16-byte LDS read each
                        const double x0 = (double)((int)((unsigned)v[u][t / 4] << (24 - 8 * (t % 4))) >> 24);  // sign-extended byte
                        const double x1 = (double)((int)((unsigned)v[u][(t + 1) / 4] << (24 - 8 * ((t + 1) % 4))) >> 24);
                        const int at = base2 + (t / 2) * WAVE;
                        const double2 vp = pi2[at];
                        p0 += x0 * vp.x;
                        p1 += x1 * vp.y;
                        if (pending) {
                            const double2 vr = rho2[at], vw = w2[at];
                            r0 += x0 * vr.x;
                            r1 += x1 * vr.y;
                            w0 += x0 * vw.x;
                            w1 += x1 * vw.y;
                        }
                    }
                }
            }
            finish_column(j, pos_j, gamma_j, cost_j, p0 + p1, r0 + r1, w0 + w1);
        }
    } else
    for (int jd = blockIdx.x * (K1D_THREADS / WAVE) + wave; jd < lp.n_dense; jd += waves_total) {
        const int j = lp.dense_first + jd;
        const int pos_j = lp.pos[j];
        const double gamma_j = lp.gamma[j];  // issued with the first column loads; used only in the tail
        const double cost_j = lp.cost[j];
        if (pos_j >= 0) continue;  // wave-uniform
        // the column stream is read once per pass and is larger than the Infinity Cache: non-temporal loads keep it from
        // competing with the lines other kernels left there (measured: 5.3 TB/s against 4.0 TB/s behind the inverse update)
        const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
        const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
        const double2* w2 = reinterpret_cast<const double2*>(s_w);
        double d_pi = 0.0, d_rho = 0.0, d_w = 0.0;
        if (I8) {
            // (handled by the two-column loop above)
        } else if (F32) {
            const f32x4* col = reinterpret_cast<const f32x4*>(lp.dense_val32 + (size_t)jd * mp);
            const int quarter = mp / 4;
            for (int k0 = lane; k0 < quarter; k0 += 8 * WAVE) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * WAVE;
                    v[u] = k < quarter ? __builtin_nontemporal_load(col + k) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * WAVE;
                    if (k < quarter) {
                        const double x0 = v[u].x, x1 = v[u].y, x2 = v[u].z, x3 = v[u].w;
                        const double2 a0 = pi2[2 * k], a1 = pi2[2 * k + 1];
                        d_pi += (x0 * a0.x + x1 * a0.y) + (x2 * a1.x + x3 * a1.y);
                        if (pending) {
                            const double2 b0 = rho2[2 * k], b1 = rho2[2 * k + 1], c0 = w2[2 * k], c1 = w2[2 * k + 1];
                            d_rho += (x0 * b0.x + x1 * b0.y) + (x2 * b1.x + x3 * b1.y);
                            d_w += (x0 * c0.x + x1 * c0.y) + (x2 * c1.x + x3 * c1.y);
                        }
                    }
                }
            }
        } else {
            const f64x2* col = reinterpret_cast<const f64x2*>(lp.dense_val + (size_t)jd * mp);
            for (int k0 = lane; k0 < half; k0 += 8 * WAVE) {
                f64x2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * WAVE;
                    v[u] = k < half ? __builtin_nontemporal_load(col + k) : f64x2{0.0, 0.0};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u * WAVE;
                    if (k < half) {
                        const double2 a = pi2[k];
                        d_pi += v[u].x * a.x + v[u].y * a.y;
                        if (pending) {
                            const double2 b = rho2[k], c = w2[k];
                            d_rho += v[u].x * b.x + v[u].y * b.y;
                            d_w += v[u].x * c.x + v[u].y * c.y;
                        }
                    }
                }
            }
        }
        d_pi = wave_sum(d_pi);
        d_rho = wave_sum(d_rho);
        d_w = wave_sum(d_w);
        if (lane == LAST) {
            double gam = gamma_j;
            if (pending) {
                if (j == leaving) {
                    gam = gamma_q / (alpha_pq * alpha_pq);
                } else {
                    const double sq = d_rho * d_rho;
                    gam = gam - 2.0 * d_rho * d_w + sq * gamma_q;
                    gam = fmax(gam, 1.0 + sq);
                }
                lp.gamma[j] = gam;
            }
            const double cbar = (pos_j == -2 ? -1.0 : 1.0) * (cost_j + d_pi);
            if (cbar < -tol_dual) {
                Cand c;
                c.idx = j;
                c.aux = 0;
                c.key = cbar * cbar / gam;
                Cand nb = better<TIE_LARGER_IDX>(best, c);
                if (nb.idx == j) best_cbar = cbar;
                best = nb;
            }
        }
    }
    if (lane == LAST) s_cbar[wave] = best_cbar;
    best.aux = wave;
    if (lane != LAST) best.idx = -1;
    Cand blk = block_best<TIE_LARGER_IDX>(best, s_cand);
    if (threadIdx.x == 0) {
        lp.cand_j[cand_offset + blockIdx.x] = blk.idx;
        lp.cand_len[cand_offset + blockIdx.x] = -1;  // column not inlined with the candidate: K2 reads it from the CSC
        if (blk.idx >= 0) {
            lp.cand_key[cand_offset + blockIdx.x] = blk.key;
            lp.cand_cbar[cand_offset + blockIdx.x] = s_cbar[blk.aux];
        }
    }
}

// K1c: the int8 dense block priced with ONE COLUMN PER LANE.  price_dense_kernel<.., I8> reads 24 B of -pi / rho / w out of LDS
// for every 1-byte entry (805 MB a pass at 4096 x 8192) and is bound by that, not by the 33 MB that come from HBM.  Here the
// block is stored in tiles of 16 columns x 64 rows (1 KiB): lane l's 16 bytes are rows 64B + 16(l >> 4) .. + 15 of column
// 16G + (l & 15).  A wave streams a run of such tiles of one column group, one 16-byte load per lane and tile, and the 16
// lanes of a DPP row work on the SAME 16 matrix rows: lane l holds -pi / rho / w of row 64B + l in a register pair (one
// coalesced 512-byte load), and entry t of the piece is multiplied by lane t's value through the DPP row broadcast of the f64 FMA
// (v_fmac_f64_dpp row_newbcast:t) -- no LDS traffic, no cross-lane reduction, 5 VALU instructions per entry (extract,
// convert, three FMAs; tools/micro/valu_rates.hip: all five issue at the full rate, DPP included).  A workgroup owns one
// group of 16 columns, its waves split the rows; their partial sums meet in LDS, and a thread per column adds them in a fixed
// order, updates the column's weight and tests it: one candidate slot per workgroup, 512 workgroups of 8 waves at n = 8192
// (two per CU, so that one's tail overlaps the other's stream; 32 columns per group and 16 waves: 12.5 us against 9.6 + 4.8).
// Measured on the way (4096 x 8192, per pass): the vectors in registers with a wave per 1024 rows and DPP wave sums per
// column, 20.5 us; 64 columns per wave with the vectors as SGPR operands from scalar loads (s_load latency exposed, and rows
// split over workgroups that meet behind a per-group counter: the __threadfence() on either side writes back and
// invalidates the XCD's whole L2, 3x slower per pivot), 11.8 us + 5.3 us for a second kernel that adds the splits.
constexpr int K1C_MAX_THREADS = 512, K1C_U = 4, K1C_COLS = 16, K1C_TILE_ROWS = 64;
template <int T>
__device__ __forceinline__ void fmac_row_broadcast(double& acc, double a, double x) {  // acc += (a of lane T of this lane's row of 16) * x
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(x), "n"(T));
}
template <int T>
__device__ __forceinline__ void dense_lane_entry(const i32x4& v, double a_pi, double a_rho, double a_w, double (&acc)[6]) {
    const double x = (double)((int)((unsigned)v[T / 4] << (24 - 8 * (T % 4))) >> 24);  // sign-extended byte T
    fmac_row_broadcast<T>(acc[T & 1], a_pi, x);  // two chains per sum
    fmac_row_broadcast<T>(acc[2 + (T & 1)], a_rho, x);
    fmac_row_broadcast<T>(acc[4 + (T & 1)], a_w, x);
}
template <int... T>
__device__ __forceinline__ void dense_lane_block(const i32x4& v, double a_pi, double a_rho, double a_w, double (&acc)[6], std::integer_sequence<int, T...>) {
    (dense_lane_entry<T>(v, a_pi, a_rho, a_w, acc), ...);
}
// The same with the block as float (data that float holds exactly, all arithmetic f64): a tile of 16 columns x 64 rows is
// 4 KiB, FOUR 16-byte pieces per lane -- piece u of lane l holds rows 64B + 16(l >> 4) + 4u .. + 3 of column 16G + (l & 15) --
// and entry e of piece u meets lane 4u + e of the lane's DPP row.  4 VALU instructions per entry; the pass is bound by HBM.
template <int T>
__device__ __forceinline__ void dense_lane_entry_f32(float f, double a_pi, double a_rho, double a_w, double (&acc)[6]) {
    const double x = (double)f;
    fmac_row_broadcast<T>(acc[T & 1], a_pi, x);
    fmac_row_broadcast<T>(acc[2 + (T & 1)], a_rho, x);
    fmac_row_broadcast<T>(acc[4 + (T & 1)], a_w, x);
}
template <int U4>
__device__ __forceinline__ void dense_lane_piece_f32(const f32x4& v, double a_pi, double a_rho, double a_w, double (&acc)[6]) {
    dense_lane_entry_f32<U4>(v[0], a_pi, a_rho, a_w, acc);
    dense_lane_entry_f32<U4 + 1>(v[1], a_pi, a_rho, a_w, acc);
    dense_lane_entry_f32<U4 + 2>(v[2], a_pi, a_rho, a_w, acc);
    dense_lane_entry_f32<U4 + 3>(v[3], a_pi, a_rho, a_w, acc);
}
// ... and as double: EIGHT pieces of two rows per lane and tile (8 KiB), entry e of piece u meets lane 2u + e; 3 VALU per entry.
template <int U2>
__device__ __forceinline__ void dense_lane_piece_f64(const f64x2& v, double a_pi, double a_rho, double a_w, double (&acc)[6]) {
    fmac_row_broadcast<U2>(acc[0], a_pi, v[0]);
    fmac_row_broadcast<U2>(acc[2], a_rho, v[0]);
    fmac_row_broadcast<U2>(acc[4], a_w, v[0]);
    fmac_row_broadcast<U2 + 1>(acc[1], a_pi, v[1]);
    fmac_row_broadcast<U2 + 1>(acc[3], a_rho, v[1]);
    fmac_row_broadcast<U2 + 1>(acc[5], a_w, v[1]);
}
template <int BYTES>  // bytes per entry: 1 (signed bytes), 4 (float), 8 (double)
struct DenseLaneBatch {  // the tiles of one lane and batch: the column pieces and the lane's -pi / rho / w of each
    static constexpr int TILES = BYTES == 1 ? K1C_U : BYTES == 4 ? 2 : 1, PIECES = BYTES;  // (a tile is BYTES KiB: BYTES pieces per lane)
    i32x4 v[TILES][PIECES];
    double vp[TILES], vr[TILES], vw[TILES];
};
template <int BYTES>
__device__ __forceinline__ void dense_lane_load(DenseLaneBatch<BYTES>& t, const i32x4* piece, const double* a_pi, const double* a_rho, const double* a_w, int b0) {
    // (33 MB of bytes at 4096 x 8192: non-temporal loads make no difference there, 18.05k against 18.13k pivots/s)
    constexpr int TILES = DenseLaneBatch<BYTES>::TILES, PIECES = DenseLaneBatch<BYTES>::PIECES;
#pragma unroll
    for (int u = 0; u < TILES; ++u)
#pragma unroll
        for (int q = 0; q < PIECES; ++q)
            t.v[u][q] = BYTES > 1 ? __builtin_nontemporal_load(piece + ((size_t)(b0 + u) * PIECES + q) * WAVE) : piece[((size_t)(b0 + u) * PIECES + q) * WAVE];
#pragma unroll
    for (int u = 0; u < TILES; ++u) {
        t.vp[u] = a_pi[(b0 + u) * K1C_TILE_ROWS];
        t.vr[u] = a_rho[(b0 + u) * K1C_TILE_ROWS];
        t.vw[u] = a_w[(b0 + u) * K1C_TILE_ROWS];
    }
}
// dense_ld is a multiple of K1C_TILE_ROWS * (blockDim.x / 64) * K1C_U; the next batch's loads are in flight while one is worked on
template <int BYTES>
__global__ void __launch_bounds__(K1C_MAX_THREADS) price_dense_lane_kernel(DeviceLP lp, int skip_weights, double tol_dual, int cand_offset) {
    __shared__ double s_part[K1C_MAX_THREADS / WAVE][3][WAVE];
    const int mp = lp.dense_ld;
    const int group = blockIdx.x;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    const int ntiles = mp / (K1C_TILE_ROWS * nw);  // tiles of 64 rows per wave
    const int first_tile = wave * ntiles;
    constexpr int TILES = DenseLaneBatch<BYTES>::TILES, PIECES = DenseLaneBatch<BYTES>::PIECES;
    const i32x4* piece = (BYTES == 8 ? reinterpret_cast<const i32x4*>(lp.dense_val) : BYTES == 4 ? reinterpret_cast<const i32x4*>(lp.dense_val32) : reinterpret_cast<const i32x4*>(lp.dense_val8)) +
                         ((size_t)group * (mp / K1C_TILE_ROWS) + first_tile) * PIECES * WAVE + lane;
    const size_t my_row = (size_t)first_tile * K1C_TILE_ROWS + lane;
    const double* a_pi = lp.minus_pi + my_row;
    const double* a_rho = lp.rho + my_row;
    const double* a_w = lp.w + my_row;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    DenseLaneBatch<BYTES> A, B;  // two batches alternate: the loads of one are in flight while the other is worked on
    dense_lane_load(A, piece, a_pi, a_rho, a_w, 0);  // requested before the control block is looked at
    // the thread that finishes column 16 * group + threadIdx.x asks for what it needs there now
    const int jd = group * K1C_COLS + threadIdx.x;
    const bool finisher = threadIdx.x < K1C_COLS && jd < lp.n_dense;
    const int j = lp.dense_first + jd;
    const int pos_j = finisher ? lp.pos[j] : 0;
    const double gamma_j = finisher ? lp.gamma[j] : 1.0, cost_j = finisher ? lp.cost[j] : 0.0;
    const Ctl* ctl = lp.ctl;
    if (ctl->status != ST_RUNNING) return;
    const int pending = ctl->pending && !skip_weights;
    const int leaving = ctl->leaving;
    const double alpha_pq = ctl->alpha_pq, gamma_q = ctl->gamma_q;
    // (all three sums whether or not a weight update is pending -- two code paths make the compiler hoist the conversions they
    // share above the branch, and spill)
    auto work = [&](DenseLaneBatch<BYTES>& t) {
#pragma unroll
        for (int u = 0; u < TILES; ++u) {
            // A VGPR written by a VALU instruction must not be read through DPP for two wait states, and the compiler does not
            // see the DPP reads inside the asm statements: whatever it does to the three vector registers (copies) happens
            // before this statement, which owns them and waits.
            asm volatile("s_nop 1" : "+v"(t.vp[u]), "+v"(t.vr[u]), "+v"(t.vw[u]));
            if constexpr (BYTES == 8) {
                dense_lane_piece_f64<0>(__builtin_bit_cast(f64x2, t.v[u][0]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f64<2>(__builtin_bit_cast(f64x2, t.v[u][1]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f64<4>(__builtin_bit_cast(f64x2, t.v[u][2]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f64<6>(__builtin_bit_cast(f64x2, t.v[u][3]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f64<8>(__builtin_bit_cast(f64x2, t.v[u][4]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f64<10>(__builtin_bit_cast(f64x2, t.v[u][5]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f64<12>(__builtin_bit_cast(f64x2, t.v[u][6]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f64<14>(__builtin_bit_cast(f64x2, t.v[u][7]), t.vp[u], t.vr[u], t.vw[u], acc);
            } else if constexpr (BYTES == 4) {
                dense_lane_piece_f32<0>(__builtin_bit_cast(f32x4, t.v[u][0]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f32<4>(__builtin_bit_cast(f32x4, t.v[u][1]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f32<8>(__builtin_bit_cast(f32x4, t.v[u][2]), t.vp[u], t.vr[u], t.vw[u], acc);
                dense_lane_piece_f32<12>(__builtin_bit_cast(f32x4, t.v[u][3]), t.vp[u], t.vr[u], t.vw[u], acc);
            } else {
                dense_lane_block(t.v[u][0], t.vp[u], t.vr[u], t.vw[u], acc, std::make_integer_sequence<int, 16>{});
            }
        }
    };
    for (int b0 = 0; b0 < ntiles; b0 += 2 * TILES) {
        const bool second = b0 + TILES < ntiles;
        if (second) dense_lane_load(B, piece, a_pi, a_rho, a_w, b0 + TILES);
        work(A);
        if (b0 + 2 * TILES < ntiles) dense_lane_load(A, piece, a_pi, a_rho, a_w, b0 + 2 * TILES);
        if (second) work(B);
    }
    s_part[wave][0][lane] = acc[0] + acc[1];
    s_part[wave][1][lane] = acc[2] + acc[3];
    s_part[wave][2][lane] = acc[4] + acc[5];
    __syncthreads();
    if (wave != 0) return;
    // wave 0: lane 16 * which + c adds sum `which` of column c (wave order, and the four row quarters of a tile in theirs);
    // lanes 0 .. 15 then hold -pi.a_j, fetch rho.a_j and w.a_j from lanes 16 + c and 32 + c, and finish their column
    double sum = 0.0;
    {
        const int which = min(lane >> 4, 2), c = lane & 15;
        for (int k = 0; k < nw; ++k) {
            const double* part = &s_part[k][which][c];
            sum += (part[0] + part[16]) + (part[32] + part[48]);
        }
    }
    const double d_pi = sum, d_rho = __shfl(sum, (lane & 15) + 16), d_w = __shfl(sum, (lane & 15) + 32);
    Cand best;
    best.key = 0.0;
    best.idx = -1;
    best.aux = lane;
    double best_cbar = 0.0;
    if (finisher && pos_j < 0) {
        double gam = gamma_j;
        if (pending) {
            if (j == leaving) {
                gam = gamma_q / (alpha_pq * alpha_pq);
            } else {
                const double sq = d_rho * d_rho;
                gam = gam - 2.0 * d_rho * d_w + sq * gamma_q;
                gam = fmax(gam, 1.0 + sq);
            }
            lp.gamma[j] = gam;
        }
        const double cbar = (pos_j == -2 ? -1.0 : 1.0) * (cost_j + d_pi);
        if (cbar < -tol_dual) {
            best.idx = j;
            best.key = cbar * cbar / gam;
            best_cbar = cbar;
        }
    }
    const Cand blk = wave_best<TIE_LARGER_IDX>(best);  // lane 63
    const int winner = __shfl(blk.aux, LAST);
    best_cbar = __shfl(best_cbar, winner);
    if (lane == LAST) {
        lp.cand_j[cand_offset + blockIdx.x] = blk.idx;
        lp.cand_len[cand_offset + blockIdx.x] = -1;  // column not inlined with the candidate: K2 reads it from the CSC
        if (blk.idx >= 0) {
            lp.cand_key[cand_offset + blockIdx.x] = blk.key;
            lp.cand_cbar[cand_offset + blockIdx.x] = best_cbar;
        }
    }
}
int dense_lane_slots(int n_dense) { return (n_dense + K1C_COLS - 1) / K1C_COLS; }
int dense_lane_threads(int m) { return m > 1024 ? 512 : 256; }
int dense_lane_ld(int m) { const int unit = K1C_TILE_ROWS * (dense_lane_threads(m) / WAVE) * K1C_U; return (m + unit - 1) / unit * unit; }

// Multi-block FTRAN for long entering columns: partial[c][i] = sum over the c-th slice of the entries of a_q of
// v_e * Binv(i, r_e).  Grid (row tiles of 256, slices); coalesced over i; fixed slice order => deterministic.
// The entering-column choice is folded in: every workgroup reduces the pricing candidates itself (same data, same fixed
// order, same result -- cheaper than a one-workgroup kernel plus a kernel boundary); workgroup (0, 0) publishes q and
// c_q, handles the iteration budget and the "no entering column" exit (the reference's tie rule: last maximum).
__global__ void __launch_bounds__(256) ftran_partial_kernel(DeviceLP lp, int n_slices, int n_price_blocks, int rule) {
    __shared__ int s_rows[256];
    __shared__ double s_vals[256];
    __shared__ double s_unit[256];
    __shared__ Cand s_cand[8];
    Ctl* ctl = lp.ctl;
    // round trip 1: the control block AND this thread's first four candidate slots (they do not depend on it)
    constexpr int PRE = 4;
    int pre_j[PRE];
    double pre_key[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int b = threadIdx.x + u * 256;
        pre_j[u] = b < n_price_blocks ? lp.cand_j[b] : -1;
        pre_key[u] = b < n_price_blocks ? lp.cand_key[b] : 0.0;  // (not written when the slot is empty: ignored below)
    }
    const int status = ctl->status;
    const long long iters_now = ctl->iters, budget = ctl->budget;
    int q = ctl->forced_q;
    if (status != ST_RUNNING) return;
    const bool structured = lp.eta_cap > 0;  // unit columns of the stored inverse are known: skip their loads
    s_unit[threadIdx.x] = 0.0;
    const bool publisher = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    if (iters_now >= budget) {
        if (publisher) {
            ctl->status = ST_BUDGET;
            ctl->pending = 0;
        }
        return;
    }
    if (q >= 0) {
        if (publisher) ctl->q = q;
    } else {
        Cand c;
        c.key = 0.0;
        c.idx = -1;
        c.aux = 0;
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int b = threadIdx.x + u * 256;
            if (b < n_price_blocks) {
                Cand o;
                o.idx = pre_j[u];
                o.key = o.idx >= 0 ? pre_key[u] : 0.0;
                o.aux = b;
                c = (rule == RELP_PIVOT_STEEPEST_EDGE) ? better<TIE_LARGER_IDX>(c, o) : better<TIE_SMALLER_IDX>(c, o);
            }
        }
        for (int b = threadIdx.x + PRE * 256; b < n_price_blocks; b += 256) {
            Cand o;
            o.idx = lp.cand_j[b];
            o.key = o.idx >= 0 ? lp.cand_key[b] : 0.0;
            o.aux = b;
            c = (rule == RELP_PIVOT_STEEPEST_EDGE) ? better<TIE_LARGER_IDX>(c, o) : better<TIE_SMALLER_IDX>(c, o);
        }
        c = (rule == RELP_PIVOT_STEEPEST_EDGE) ? block_best<TIE_LARGER_IDX>(c, s_cand) : block_best<TIE_SMALLER_IDX>(c, s_cand);
        q = c.idx;
        if (publisher) {
            ctl->q = q;
            if (q >= 0) {
                ctl->cbar_q = lp.cand_cbar[c.aux];
            } else {
                ctl->status = ST_NO_ENTERING;
                ctl->pending = 0;
                ctl->last_selected = -1;
            }
        }
    }
    if (q < 0) return;
    const int m = lp.m, ld = lp.ld;
    // a column of a dense block whose columns all have m entries: its place in the CSC and its row indices are known
    const bool full = lp.dense_full && q >= lp.dense_first && q < lp.dense_first + lp.n_dense;
    const int ca = full ? lp.dense_csc_start + (q - lp.dense_first) * m : lp.col_start[q];
    const int cb = full ? ca + m : lp.col_start[q + 1];
    const int len = (cb - ca + n_slices - 1) / n_slices;
    const int e0 = ca + blockIdx.y * len, e1 = min(cb, e0 + len);
    const int tile0 = blockIdx.x * 256;
    const int i = tile0 + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    __shared__ int s_wave_count[4];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    double unit = 0.0;  // contributions of unit columns e_row (rows are unique within a column)
    for (int c0 = e0; c0 < e1; c0 += 256) {
        int cnt = min(256, e1 - c0);
        __syncthreads();
        if (structured) {
            // keep only the entries whose column of the stored inverse carries information; a unit column e_row adds its
            // value to row `row` alone.  Order-preserving compaction (ballot + prefix) keeps the sums deterministic.
            int row = -1;
            double val = 0.0;
            bool keep = false;
            if (threadIdx.x < cnt) {
                row = full ? c0 + (int)threadIdx.x - ca : lp.row_index[c0 + threadIdx.x];
                val = lp.value[c0 + threadIdx.x];
                keep = lp.touched[row] != 0;
            }
            const unsigned long long mask = __ballot(keep);
            if (lane == 0) s_wave_count[wave] = __popcll(mask);
            __syncthreads();
            int base = 0;
            for (int wv = 0; wv < wave; ++wv) base += s_wave_count[wv];
            cnt = s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
            if (keep) {
                const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                s_rows[slot] = row;
                s_vals[slot] = val;
            } else if (row >= tile0 && row < tile0 + 256) {
                s_unit[row - tile0] = val;
            }
        } else if (threadIdx.x < cnt) {
            s_rows[threadIdx.x] = lp.row_index[c0 + threadIdx.x];
            s_vals[threadIdx.x] = lp.value[c0 + threadIdx.x];
        }
        __syncthreads();
        if (i < m) {
            const double* col = lp.Binv + i;
            int e = 0;
            for (; e + 4 <= cnt; e += 4) {
                a0 += col[(size_t)s_rows[e] * ld] * s_vals[e];
                a1 += col[(size_t)s_rows[e + 1] * ld] * s_vals[e + 1];
                a2 += col[(size_t)s_rows[e + 2] * ld] * s_vals[e + 2];
                a3 += col[(size_t)s_rows[e + 3] * ld] * s_vals[e + 3];
            }
            for (; e < cnt; ++e) a0 += col[(size_t)s_rows[e] * ld] * s_vals[e];
        }
        if (structured) {
            unit += s_unit[threadIdx.x];
            s_unit[threadIdx.x] = 0.0;
        }
    }
    a0 += unit;
    if (i < m) lp.alpha_part[(size_t)blockIdx.y * m + i] = (a0 + a1) + (a2 + a3);
}

// alpha_in[i] = sum_c alpha_part[c][i] (fixed order), so that the fused kernel reads one vector.  In the deferred product
// form the pending etas are applied here, all at once and in parallel:  alpha = M y = y + sum_c (M[:, P_c] - e_{P_c}) y[P_c].
constexpr int ETA_MAX = 32;
__device__ __forceinline__ double sum_slices(const DeviceLP& lp, int n_slices, int i) {
    double acc = 0.0;
    for (int c0 = 0; c0 < n_slices; c0 += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (c0 + u < n_slices) ? lp.alpha_part[(size_t)(c0 + u) * lp.m + i] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    return acc;
}
// the kept columns of M exist twice; ctl->eta_version (one more per pivot, counted by K2) says which copy is current
__device__ __forceinline__ double* eta_columns(const DeviceLP& lp, int version) { return lp.eta_cols + (size_t)(version & 1) * lp.eta_cap * lp.ld; }
constexpr int AR_ROWS = 64, AR_GROUPS = 4;
__global__ void __launch_bounds__(AR_ROWS * AR_GROUPS) alpha_reduce_kernel(DeviceLP lp, int n_slices) {
    __shared__ double s_y[ETA_MAX];
    __shared__ int s_p[ETA_MAX];
    __shared__ double s_part[AR_GROUPS][AR_ROWS];
    const int m = lp.m;
    // round trip 1: everything that needs no other result (control word, kept rows, this thread's slices and eta entries)
    const int status = lp.ctl->status, q = lp.ctl->q;
    const int k = lp.eta_cap > 0 ? lp.ctl->eta_count : 0;
    const double* kept = lp.eta_cap > 0 ? eta_columns(lp, lp.ctl->eta_version) : nullptr;
    const int c8 = threadIdx.x / 8, sub = threadIdx.x % 8;
    const int kept_row = (lp.eta_cap > 0 && c8 < lp.eta_cap) ? lp.eta_rows[c8] : 0;
    const int g = threadIdx.x / AR_ROWS, r = threadIdx.x % AR_ROWS;
    const int i = blockIdx.x * AR_ROWS + r;
    double acc = 0.0;
    if (i < m)
        for (int sl = g; sl < n_slices; sl += AR_GROUPS) acc += lp.alpha_part[(size_t)sl * m + i];
    constexpr int EPT = ETA_MAX / AR_GROUPS;  // eta columns per thread
    double mic[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int c = g + u * AR_GROUPS;
        mic[u] = (lp.eta_cap > 0 && i < m) ? kept[(size_t)c * lp.ld + i] : 0.0;  // entries of unused slots are ignored below
    }
    if (status != ST_RUNNING || q < 0) return;
    if (k > 0) {  // round trip 2: y at the kept rows: 8 threads per row, 1/8 of the slices each, fixed-order combine
        double part = 0.0;
        if (c8 < k)
            for (int sl = sub; sl < n_slices; sl += 8) part += lp.alpha_part[(size_t)sl * m + kept_row];
        part += __shfl_xor(part, 4, 8);
        part += __shfl_xor(part, 2, 8);
        part += __shfl_xor(part, 1, 8);
        if (c8 < k && sub == 0) {
            s_y[c8] = part;
            s_p[c8] = kept_row;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int c = g + u * AR_GROUPS;
            if (c < k) acc += (mic[u] - (i == s_p[c] ? 1.0 : 0.0)) * s_y[c];
        }
    }
    s_part[g][r] = acc;
    __syncthreads();
    const double alpha_i = (s_part[0][r] + s_part[1][r]) + (s_part[2][r] + s_part[3][r]);
    if (g == 0 && i < m) lp.alpha_in[i] = alpha_i;
    // this block of rows' share of alpha' M[:, c] for every kept column (the BTRAN pass needs the whole products: they are
    // the entries of the row vector behind w at the kept rows); wave g takes the columns g, g + 4, ...
    if (k > 0) {
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int c = g + u * AR_GROUPS;
            if (c >= k) break;  // wave-uniform
            const double share = wave_sum(i < m ? alpha_i * mic[u] : 0.0);
            if (r == WAVE - 1) lp.eta_dot_part[(size_t)c * gridDim.x + blockIdx.x] = share;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Deferred product form of the inverse (dense pipeline; DeviceLP::eta_cap > 0).
//   replaces  BasisInverse::change_basis + should_refactor/invert       lower_upper/mod.rs:94-178,249-252,78-92
//                 (the reference keeps <= 30 Forrest-Tomlin etas beside L U and refactors; here <= eta_cap
//                  product-form etas are kept beside the explicit inverse and folded in by one rank-k update)
//             BasisInverse::basis_inverse_row, right_multiply_by_basis_inverse (w)    lower_upper/mod.rs:254-272,212-237
//             Carry::update_minus_pi_and_obj                                           carry/mod.rs:338-349
// With the etas kept as the columns M[:, P] of their product, applying them is a parallel (m x k) mat-vec (no
// sequential eta loop), and a pivot costs two READ-ONLY passes over the stored inverse (FTRAN; rho_p and w together)
// instead of a read pass plus a read-modify-write pass: 2 m^2 instead of 3 m^2 doubles of HBM traffic, no dirty lines
// in front of the next pricing pass.
// ---------------------------------------------------------------------------------------------------
// After K2 chose (q, p) the new eta E = I - (alpha - e_p) e_p'/alpha_p has to be folded into the kept columns, and the BTRAN
// pass needs two row vectors:
//   rvec2 = alpha' M_old   (w   = rvec2 * Binv, old basis:  carry/mod.rs:575)
//   rvec1 = e_p'  M_new    (rho = rvec1 * Binv, new basis:  lower_upper/mod.rs:254-272)
// Off the kept rows they are alpha and 0; at kept row eta_rows[c] they are alpha' M_old[:, c] and M_old[p, c] / alpha_p; at p,
// when p had no kept column, alpha_p and 1 / alpha_p.  Round 1 had a kernel of its own for this between K2 and the BTRAN
// pass; now alpha_reduce_kernel leaves the products alpha' M_old[:, c] behind (per block of 64 rows), K2 does the
// bookkeeping of the new kept column, and every workgroup of the BTRAN pass builds the two vectors in LDS itself (32 KB of
// alpha and 16 KB of partial products out of L2 instead of the 64 KB of the two vectors) and rewrites its share of the kept
// columns into the OTHER copy of them -- its neighbours still read M_old[p, c] from the current one.
// The BTRAN pass: rho[j] = rvec1 . Binv(:, j), w[j] = rvec2 . Binv(:, j), -pi[j] -= cbar_q rho[j]; one wave per column,
// 16-byte loads, both row vectors in LDS (built by every workgroup, see above).
constexpr int BT_THREADS = 1024;
// When the sparse part of the LP is one unit-like column per row (the slack columns of the dense LP), their pricing for the
// NEXT pivot is done here as well: (-pi_j, rho_j, w_j) of row j are in registers when they are written, and the slack column of
// row j needs nothing else -- the separate pricing launch over the slack columns disappears (`lp.slack_of_row`; same
// arithmetic as price_kernel: weight update, reduced cost, key, last-maximum tie rule).
__global__ void __launch_bounds__(BT_THREADS) btran_pass_kernel(DeviceLP lp, double tol_dual) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ Cand s_cand[BT_THREADS / WAVE + 2];
    Ctl* ctl = lp.ctl;
    const bool price_slacks = lp.slack_of_row != nullptr;
    const int m = lp.m, ld = lp.ld;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    // Round trip 1 carries, beside the control block, everything whose ADDRESS does not depend on it: this thread's unit-column
    // candidate (column j_u: is it touched, its -pi, its slack column) and the entries of the touched-column
    // list its wave will visit (the list only grows; entries past touched_count are not used).
    constexpr int GROUPS = BT_THREADS / WAVE / 4;  // touched columns per workgroup pass: four waves each
    const int part = wave & 3, group = wave >> 2;
    const int j_u = blockIdx.x * BT_THREADS + threadIdx.x;
    const bool has_u = j_u < m;
    const int touched_u = has_u ? lp.touched[j_u] : 1;
    const double pi_u = has_u ? lp.minus_pi[j_u] : 0.0;
    const int sor_u = (has_u && price_slacks) ? lp.slack_of_row[j_u] : -1;
    constexpr int TL = 4;
    int tl[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) {
        const int idx = (blockIdx.x + t * gridDim.x) * GROUPS + group;
        tl[t] = idx < m ? lp.tlist[idx] : 0;
    }
    const int status = ctl->status, pending_now = ctl->pending;
    const double gamma_q = ctl->gamma_q, alpha_pq_c = ctl->alpha_pq;
    const int leaving = ctl->leaving;
    const int n_touched = ctl->touched_count;
    if (status != ST_RUNNING || !pending_now) return;
    Cand best;
    best.key = 0.0;
    best.idx = -1;
    best.aux = 0;
    double best_cbar = 0.0, best_val = 0.0;
    int best_row = 0;
    // the slack column's own data is fetched BEFORE the column sweep it belongs to (no dependent round trip at the end)
    struct SlackData {
        int js;
        int pos;
        double v, gam, cost;
    };
    auto slack_data = [&](int js) {
        SlackData d;
        d.js = js;
        const int jj = js < 0 ? 0 : js;
        d.pos = js < 0 ? 0 : lp.pos[jj];
        d.v = lp.ell_vals[(size_t)jj * ELL_W];
        d.gam = lp.gamma[jj];
        d.cost = lp.cost[jj];
        return d;
    };
    auto slack_fetch = [&](int j) { return slack_data(lp.slack_of_row[j]); };
    auto slack = [&](const SlackData& d, int j, double pi_new, double rho_j, double w_j) {
        const int js = d.js;
        if (js < 0 || d.pos != -1) return;
        const double v = d.v;
        const double d_pi = v * pi_new, d_rho = v * rho_j, d_w = v * w_j;
        double gam = d.gam;
        if (js == leaving) {
            gam = gamma_q / (alpha_pq_c * alpha_pq_c);  // pivot_rule.rs:294-295
        } else {
            const double sq = d_rho * d_rho;  // pivot_rule.rs:262-288 (Goldfarb-Reid)
            gam = gam - 2.0 * d_rho * d_w + sq * gamma_q;
            gam = fmax(gam, 1.0 + sq);
        }
        lp.gamma[js] = gam;
        const double cbar = d.cost + d_pi;
        if (cbar < -tol_dual) {
            Cand c;
            c.idx = js;
            c.aux = 0;
            c.key = cbar * cbar / gam;
            const Cand nb = better<TIE_LARGER_IDX>(best, c);
            if (nb.idx == js) {
                best_cbar = cbar;
                best_row = j;
                best_val = v;
            }
            best = nb;
        }
    };
    const int mp = (m + 1) & ~1;
    double* s_r1 = smem;
    double* s_r2 = smem + mp;
    const double cbar_q = ctl->cbar_q;
    const int half = m / 2;  // ld is even for the dense pipeline (m even is required by the caller)
    const int quarter = (half + 3) / 4;
    const int k_first = part * quarter, k_last = min(half, k_first + quarter);
    // round trip 2, all of it requested before anything of it is used: the unit-column candidate's slack column, the first
    // touched column of this wave (the first batch of its quarter) with its slack column, and the prologue's loads below
    SlackData sd_u;
    sd_u.js = -1;
    if (has_u && !touched_u && price_slacks) sd_u = slack_data(sor_u);
    struct Visit {  // one touched column of this wave: the first batch of its rows, and what the wave that finishes it needs
        bool active;
        int j;
        double2 v[8];
        SlackData sd;
        double pi_old;
    };
    auto issue = [&](int t, int base, Visit& x) {
        const int idx = base + group;
        x.active = idx < n_touched;
        const int listed = t == 0 ? tl[0] : t == 1 ? tl[1] : t == 2 ? tl[2] : t == 3 ? tl[3] : (x.active ? lp.tlist[idx] : 0);
        x.j = x.active ? listed : 0;
        const double2* col = reinterpret_cast<const double2*>(lp.Binv + (size_t)x.j * ld);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k_first + lane + u * WAVE;
            x.v[u] = (x.active && kk < k_last) ? col[kk] : make_double2(0.0, 0.0);
        }
        x.sd.js = -1;
        x.pi_old = 0.0;
        if (x.active && part == 0 && lane == WAVE - 1) {
            if (price_slacks) x.sd = slack_fetch(x.j);
            x.pi_old = lp.minus_pi[x.j];
        }
    };
    Visit cur;
    issue(0, blockIdx.x * GROUPS, cur);
    {
        __shared__ double s_dot[ETA_MAX][32];
        const int version = ctl->eta_version;  // K2 counted this pivot already: the columns as they were are in copy version - 1
        const int is_new = ctl->eta_new;
        const int k_old = ctl->eta_count - is_new;
        const int p = ctl->p;
        const double* kept_old = eta_columns(lp, version - 1);
        double* kept_new = eta_columns(lp, version);
        // this thread's element of the kept columns (column c_el, row i_el), requested first
        const int el = blockIdx.x * BT_THREADS + threadIdx.x;
        const int c_el = el / m, i_el = el - c_el * m;
        const bool el_kept = c_el < k_old, el_new = is_new && c_el == k_old;
        const double el_old = el_kept ? kept_old[(size_t)c_el * ld + i_el] : 0.0;
        const double el_at_p = el_kept ? kept_old[(size_t)c_el * ld + p] : 0.0;
        const double el_alpha = (el_kept || el_new) ? lp.alpha[i_el] : 0.0;
        // thread (c, s) of the first 32 * k_old: two of the 64-row blocks' shares of alpha' M_old[:, c]
        const int n_shares = (m + AR_ROWS - 1) / AR_ROWS;  // <= 64
        const int dc = threadIdx.x / 32, ds = threadIdx.x % 32;
        double share = 0.0;
        if (dc < k_old) {
            const double* part = lp.eta_dot_part + (size_t)dc * n_shares;
            share = (ds < n_shares ? part[ds] : 0.0) + (ds + 32 < n_shares ? part[ds + 32] : 0.0);
        }
        const double at_p = (int)threadIdx.x < k_old ? kept_old[(size_t)threadIdx.x * ld + p] : 0.0;
        const int kept_row = (int)threadIdx.x < k_old ? lp.eta_rows[threadIdx.x] : 0;
        for (int i = threadIdx.x; i < mp; i += BT_THREADS) {
            s_r1[i] = 0.0;
            s_r2[i] = i < m ? lp.alpha[i] : 0.0;
        }
        if (dc < ETA_MAX) s_dot[dc][ds] = share;
        __syncthreads();
        if ((int)threadIdx.x < k_old) {
            double dot = 0.0;
            for (int e = 0; e < 32; ++e) dot += s_dot[threadIdx.x][e];  // fixed order
            s_r2[kept_row] = dot;
            s_r1[kept_row] = at_p / alpha_pq_c;
        }
        if (threadIdx.x == 0 && is_new) {  // p was not a kept column: M_old[:, p] = e_p
            s_r2[p] = alpha_pq_c;
            s_r1[p] = 1.0 / alpha_pq_c;
        }
        if (el_kept) {
            const double t = el_at_p / alpha_pq_c;
            kept_new[(size_t)c_el * ld + i_el] = (i_el == p) ? t : el_old - el_alpha * t;
        } else if (el_new) {  // column p of E itself
            kept_new[(size_t)c_el * ld + i_el] = (i_el == p) ? 1.0 / alpha_pq_c : -el_alpha / alpha_pq_c;
        }
        __syncthreads();
    }
    const double2* r1 = reinterpret_cast<const double2*>(s_r1);
    const double2* r2 = reinterpret_cast<const double2*>(s_r2);
    // unit columns: rho_j = rvec1[j], w_j = rvec2[j]
    if (has_u && !touched_u) {
        const double d1 = s_r1[j_u];
        const double pi_new = pi_u - cbar_q * d1;
        lp.rho[j_u] = d1;
        lp.w[j_u] = s_r2[j_u];
        lp.minus_pi[j_u] = pi_new;
        if (price_slacks) slack(sd_u, j_u, pi_new, d1, s_r2[j_u]);
    }
    for (int j = j_u + gridDim.x * BT_THREADS; j < m; j += gridDim.x * BT_THREADS) {  // (m beyond the grid: not the case today)
        if (!lp.touched[j]) {
            SlackData sd;
            sd.js = -1;
            if (price_slacks) sd = slack_fetch(j);
            const double d1 = s_r1[j];
            const double pi_new = lp.minus_pi[j] - cbar_q * d1;
            lp.rho[j] = d1;
            lp.w[j] = s_r2[j];
            lp.minus_pi[j] = pi_new;
            if (price_slacks) slack(sd, j, pi_new, d1, s_r2[j]);
        }
    }
    // FOUR waves per touched column (a quarter of its rows each: one batch of loads per lane at m = 4096 instead of four dependent
    // ones), the four partial sums added in a fixed order by the first of them.  Early in a solve there are far fewer touched
    // columns than waves, so the pass was a handful of waves each waiting on four round trips.
    __shared__ double s_quarter[BT_THREADS / WAVE][2];
    int visit = 0;
    for (int base = blockIdx.x * GROUPS; base < n_touched; base += gridDim.x * GROUPS, ++visit) {
        if (visit > 0) issue(visit, base, cur);  // (requesting a wave's next column before the sums and barriers of the current one
                                                 // was measured: the second set of registers spills at 1024 threads, 11.3 us against 10.3)
        const bool active = cur.active;
        const int j = cur.j;
        const double2* col = reinterpret_cast<const double2*>(lp.Binv + (size_t)j * ld);
        double d1 = 0.0, d2 = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k_first + lane + u * WAVE;
            if (active && kk < k_last) {
                const double2 x = r1[kk], y = r2[kk];
                d1 += cur.v[u].x * x.x + cur.v[u].y * x.y;
                d2 += cur.v[u].x * y.x + cur.v[u].y * y.y;
            }
        }
        for (int k0 = k_first + lane + 8 * WAVE; active && k0 < k_last; k0 += 8 * WAVE) {  // (quarters beyond one batch: m > 4096)
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k0 + u * WAVE;
                v[u] = kk < k_last ? col[kk] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k0 + u * WAVE;
                if (kk < k_last) {
                    const double2 x = r1[kk], y = r2[kk];
                    d1 += v[u].x * x.x + v[u].y * x.y;
                    d2 += v[u].x * y.x + v[u].y * y.y;
                }
            }
        }
        d1 = wave_sum(d1);
        d2 = wave_sum(d2);
        if (lane == WAVE - 1) {
            s_quarter[wave][0] = d1;
            s_quarter[wave][1] = d2;
        }
        __syncthreads();
        if (active && part == 0 && lane == WAVE - 1) {
            d1 = (s_quarter[wave][0] + s_quarter[wave + 1][0]) + (s_quarter[wave + 2][0] + s_quarter[wave + 3][0]);
            d2 = (s_quarter[wave][1] + s_quarter[wave + 1][1]) + (s_quarter[wave + 2][1] + s_quarter[wave + 3][1]);
            const double pi_new = cur.pi_old - cbar_q * d1;
            lp.rho[j] = d1;
            lp.w[j] = d2;
            lp.minus_pi[j] = pi_new;
            if (price_slacks) slack(cur.sd, j, pi_new, d1, d2);
        }
        __syncthreads();
    }
    if (price_slacks) {  // this workgroup's best slack column for the next pivot, in price_kernel's candidate format
        const Cand blk = block_best<TIE_LARGER_IDX>(best, s_cand);
        if (blk.idx >= 0 && blk.idx == best.idx) {
            lp.cand_key[blockIdx.x] = blk.key;
            lp.cand_j[blockIdx.x] = blk.idx;
            lp.cand_cbar[blockIdx.x] = best_cbar;
            lp.cand_len[blockIdx.x] = 1;
#pragma unroll
            for (int e = 0; e < ELL_W; ++e) {
                lp.cand_rows[(size_t)blockIdx.x * ELL_W + e] = e == 0 ? best_row : 0;
                lp.cand_vals[(size_t)blockIdx.x * ELL_W + e] = e == 0 ? best_val : 0.0;
            }
        }
        if (blk.idx < 0 && threadIdx.x == 0) lp.cand_j[blockIdx.x] = -1;
    }
}

// Consolidation, step 0: the pivot rows of the pending etas become touched columns.
__global__ void eta_mark_kernel(DeviceLP lp) {
    if (threadIdx.x != 0) return;
    Ctl* ctl = lp.ctl;
    const int k = ctl->eta_count;
    int count = ctl->touched_count;
    for (int c = 0; c < k; ++c) {
        const int row = lp.eta_rows[c];
        if (!lp.touched[row]) {
            lp.touched[row] = 1;
            lp.tlist[count++] = row;
        }
    }
    ctl->touched_count = count;
}
__global__ void __launch_bounds__(256) mark_all_touched_kernel(DeviceLP lp) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < lp.m) {
        lp.touched[j] = 1;
        lp.tlist[j] = j;
    }
    if (j == 0) lp.ctl->touched_count = lp.m;
}
// Step 1: gather the rows P of the stored inverse (they are overwritten by step 2); touched columns only (the others
// are unit vectors of rows outside P: zero there).
__global__ void __launch_bounds__(256) eta_gather_kernel(DeviceLP lp) {
    const int k = lp.ctl->eta_count;
    const int c = blockIdx.y;
    if (c >= k) return;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= lp.ctl->touched_count) return;
    const int j = lp.tlist[idx];
    lp.eta_gather[(size_t)c * lp.m + idx] = lp.Binv[(size_t)j * lp.ld + lp.eta_rows[c]];
}
// Step 2: Binv <- M Binv = Binv + (M[:, P] - I[:, P]) * Binv[P, :], a rank-k update of the touched columns;
// 128 x 64 tiles, 4 x 8 per thread.
constexpr int EA_TI = 128, EA_TJ = 64;
__global__ void __launch_bounds__(256) eta_apply_kernel(DeviceLP lp) {
    __shared__ double s_a[ETA_MAX][EA_TI];  // (M - I)[rows of the tile, P_c]
    __shared__ double s_b[ETA_MAX][EA_TJ];  // Binv_old[P_c, columns of the tile]
    __shared__ int s_j[EA_TJ];
    const int k = lp.ctl->eta_count;
    const int n_touched = lp.ctl->touched_count;
    const int i0 = blockIdx.x * EA_TI, j0 = blockIdx.y * EA_TJ;
    if (k == 0 || j0 >= n_touched) return;
    const int m = lp.m, ld = lp.ld;
    const double* kept = eta_columns(lp, lp.ctl->eta_version);
    for (int e = threadIdx.x; e < k * EA_TI; e += 256) {
        const int c = e / EA_TI, r = e % EA_TI;
        const int i = i0 + r;
        s_a[c][r] = i < m ? kept[(size_t)c * ld + i] - (i == lp.eta_rows[c] ? 1.0 : 0.0) : 0.0;
    }
    for (int e = threadIdx.x; e < k * EA_TJ; e += 256) {
        const int c = e / EA_TJ, r = e % EA_TJ;
        s_b[c][r] = j0 + r < n_touched ? lp.eta_gather[(size_t)c * m + j0 + r] : 0.0;
    }
    if (threadIdx.x < EA_TJ) s_j[threadIdx.x] = j0 + threadIdx.x < n_touched ? lp.tlist[j0 + threadIdx.x] : -1;
    __syncthreads();
    const int ri = (threadIdx.x % 32) * 4, cj = (threadIdx.x / 32) * 8;
    double acc[8][4];
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[y][x] = 0.0;
    for (int c = 0; c < k; ++c) {
        double a[4], b[8];
#pragma unroll
        for (int x = 0; x < 4; ++x) a[x] = s_a[c][ri + x];
#pragma unroll
        for (int y = 0; y < 8; ++y) b[y] = s_b[c][cj + y];
#pragma unroll
        for (int y = 0; y < 8; ++y)
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[y][x] += a[x] * b[y];
    }
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const int j = s_j[cj + y];
        if (j < 0) continue;
        double* col = lp.Binv + (size_t)j * ld + i0 + ri;
#pragma unroll
        for (int x = 0; x < 4; ++x)
            if (i0 + ri + x < m) col[x] += acc[y][x];
    }
}
// Step 3: forget the etas.
__global__ void eta_reset_kernel(DeviceLP lp) {
    const int k = lp.ctl->eta_count;
    if ((int)threadIdx.x < k) lp.eta_slot[lp.eta_rows[threadIdx.x]] = -1;
    __syncthreads();
    if (threadIdx.x == 0) lp.ctl->eta_count = 0;
}

// ---------------------------------------------------------------------------------------------------
// K2: entering column choice, FTRAN, ratio test, x_B update and the compacted non-zero list of alpha, in ONE
// workgroup (the data it touches is nnz(a_q) columns of the inverse + O(m) vectors).
//   replaces  Tableau::generate_column -> BasisInverse::left_multiply_by_basis_inverse   tableau/mod.rs:126-130,
//                 lower_upper/mod.rs:180-210 (explicit inverse: basis_inverse_rows.rs:139-152)
//             Tableau::select_primal_pivot_row                                           tableau/mod.rs:287-313
//             Carry::update_b and the basis bookkeeping                                  carry/mod.rs:295-325,561-604
// The ratio test is the two-pass Harris variant (f64 needs a pivot-size preference the exact reference does
// not); ties keep the reference's Bland rule (lowest leaving column).
// mode 0: full iteration | 1: stop after the entering-column choice | 2: stop after the ratio test (no update)
// ---------------------------------------------------------------------------------------------------
constexpr int K2_THREADS = 1024;
constexpr int K2_COL_CHUNK = 1024;  // entries of the entering column staged per pass
template <int RULE>
__global__ void __launch_bounds__(K2_THREADS) ftran_ratio_kernel(DeviceLP lp, int n_price_blocks, double tol_pivot,
                                                               double harris_delta, int skip_artificial_rows, int mode) {
    __shared__ Cand s_cand[18];
    __shared__ double s_red[18];
    __shared__ int s_q;
    __shared__ double s_cbar;
    __shared__ int s_rows[K2_COL_CHUNK];
    __shared__ double s_vals[K2_COL_CHUNK];
    Ctl* ctl = lp.ctl;
    if (ctl->status != ST_RUNNING) return;
    if (mode == 0 && ctl->iters >= ctl->budget) {
        if (threadIdx.x == 0) {
            ctl->status = ST_BUDGET;
            ctl->pending = 0;
        }
        return;
    }
    const int m = lp.m;
    const int ld = lp.ld;
    const int forced_q = ctl->forced_q;
    const int forced_p = ctl->forced_p;

    // ---- entering column ------------------------------------------------------------------------
    if (forced_q < 0) {
        Cand c;
        c.key = 0.0;
        c.idx = -1;
        c.aux = 0;
        for (int b = threadIdx.x; b < n_price_blocks; b += blockDim.x) {
            Cand o;
            o.idx = lp.cand_j[b];
            o.key = o.idx >= 0 ? lp.cand_key[b] : 0.0;
            o.aux = b;
            c = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? better<TIE_LARGER_IDX>(c, o) : better<TIE_SMALLER_IDX>(c, o);
        }
        c = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? block_best<TIE_LARGER_IDX>(c, s_cand) : block_best<TIE_SMALLER_IDX>(c, s_cand);
        if (threadIdx.x == 0) {
            s_q = c.idx;
            s_cbar = c.idx >= 0 ? lp.cand_cbar[c.aux] : 0.0;
        }
    } else if (threadIdx.x == 0) {
        s_q = forced_q;
        double cb = lp.cost[forced_q];
        for (int e = lp.col_start[forced_q]; e < lp.col_start[forced_q + 1]; ++e) cb += lp.value[e] * lp.minus_pi[lp.row_index[e]];
        s_cbar = cb;
    }
    __syncthreads();
    const int q = s_q;
    const double cbar_q = s_cbar;
    if (q < 0) {
        if (threadIdx.x == 0) {
            if (mode == 0) ctl->status = ST_NO_ENTERING;
            ctl->q = -1;
            ctl->pending = 0;
            if (mode == 0) ctl->last_selected = -1;
        }
        return;
    }
    if (mode == 1) {
        if (threadIdx.x == 0) {
            ctl->q = q;
            ctl->cbar_q = cbar_q;
            ctl->pending = 0;
        }
        return;
    }

    // ---- FTRAN: alpha = sum_k v_k Binv(:, r_k): contiguous column reads, the column staged through LDS ----
    // Every pass over the m rows handles K2_U rows per thread at a time so that their loads are in flight together (a
    // one-row-per-iteration loop is a chain of ~m / 1024 dependent memory latencies: 190 us at m = 65 534).
    constexpr int K2_U = 8;
    const int ca = lp.col_start[q], cb_ = lp.col_start[q + 1];
    const bool single = (cb_ - ca) <= K2_COL_CHUNK;
    // Implicit upper bounds: see ftran_ratio_fast_kernel (same rules, alpha kept in global memory here).
    const bool bounded = lp.ub != nullptr;
    const double sgn_q = (bounded && lp.flipped[q]) ? -1.0 : 1.0;
    const double ub_q = bounded ? lp.ub[q] : INFINITY;
    const double cbar_signed = (bounded && forced_q >= 0) ? cbar_q * sgn_q : cbar_q;
    double sumsq = 0.0;
    double theta = INFINITY;
    if (!single)
        for (int i = threadIdx.x; i < m; i += blockDim.x) lp.alpha[i] = 0.0;
    for (int c0 = ca; c0 < cb_; c0 += K2_COL_CHUNK) {
        const int cnt = min(K2_COL_CHUNK, cb_ - c0);
        const bool last_chunk = c0 + K2_COL_CHUNK >= cb_;
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
            s_rows[e] = lp.row_index[c0 + e];
            s_vals[e] = lp.value[c0 + e];
        }
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < m; i0 += K2_U * blockDim.x) {
            double acc[K2_U], xbv[K2_U], upv[K2_U];
            int basv[K2_U];
#pragma unroll
            for (int u = 0; u < K2_U; ++u) {
                const int i = i0 + u * blockDim.x;
                acc[u] = (!single && i < m) ? lp.alpha[i] : 0.0;
                xbv[u] = (last_chunk && i < m) ? lp.xB[i] : 0.0;
                basv[u] = (last_chunk && i < m) ? lp.basis[i] : 0;
                upv[u] = (last_chunk && bounded && i < m) ? lp.xub[i] : INFINITY;
            }
            for (int e = 0; e < cnt; ++e) {
                const size_t off = (size_t)s_rows[e] * ld;
                const double v = s_vals[e];
#pragma unroll
                for (int u = 0; u < K2_U; ++u) {
                    const int i = i0 + u * blockDim.x;
                    if (i < m) acc[u] += lp.Binv[off + i] * v;
                }
            }
#pragma unroll
            for (int u = 0; u < K2_U; ++u) {
                const int i = i0 + u * blockDim.x;
                if (i >= m) continue;
                if (!last_chunk) {
                    lp.alpha[i] = acc[u];
                    continue;
                }
                const double a = acc[u] * sgn_q;  // gamma_q and Harris pass 1 fused into the last FTRAN pass
                lp.alpha[i] = a;
                sumsq += a * a;
                if (skip_artificial_rows && basv[u] < lp.n_art) continue;
                if (a > tol_pivot) theta = fmin(theta, (fmax(xbv[u], 0.0) + harris_delta) / a);
                else if (bounded && a < -tol_pivot && upv[u] < INFINITY) theta = fmin(theta, (fmax(upv[u] - xbv[u], 0.0) + harris_delta) / -a);
            }
        }
    }
    const double gamma_q = 1.0 + block_reduce<0>(sumsq, s_red);  // pivot_rule.rs:258 (1 + ||alpha_q||^2)
    int p = forced_p;
    if (forced_p < 0) {
        // ---- Harris ratio test ---------------------------------------------------------------------
        const double theta_max = block_reduce<1>(theta, s_red);
        Cand c;
        c.key = 0.0;
        c.idx = -1;
        c.aux = 0;
        for (int i0 = threadIdx.x; i0 < m; i0 += K2_U * blockDim.x) {
            double av[K2_U], xbv[K2_U], upv[K2_U];
            int basv[K2_U];
#pragma unroll
            for (int u = 0; u < K2_U; ++u) {
                const int i = i0 + u * blockDim.x;
                av[u] = i < m ? lp.alpha[i] : 0.0;
                xbv[u] = i < m ? lp.xB[i] : 0.0;
                basv[u] = i < m ? lp.basis[i] : 0;
                upv[u] = (bounded && i < m) ? lp.xub[i] : INFINITY;
            }
#pragma unroll
            for (int u = 0; u < K2_U; ++u) {
                const int i = i0 + u * blockDim.x;
                if (i >= m || (skip_artificial_rows && basv[u] < lp.n_art)) continue;
                const double a = av[u];
                double room = -1.0;
                if (a > tol_pivot) room = fmax(xbv[u], 0.0);
                else if (bounded && a < -tol_pivot && upv[u] < INFINITY) room = fmax(upv[u] - xbv[u], 0.0);
                if (room >= 0.0 && room / fabs(a) <= theta_max) {
                    Cand o;
                    o.key = fabs(a);
                    o.idx = i;
                    o.aux = basv[u];
                    c = better<TIE_SMALLER_AUX>(c, o);
                }
            }
        }
        c = block_best<TIE_SMALLER_AUX>(c, s_cand);
        p = c.idx;
    }
    const double alpha_pq = p >= 0 ? lp.alpha[p] : 1.0;
    const double xb_p = p >= 0 ? lp.xB[p] : 0.0;
    const double up_p = (bounded && p >= 0) ? lp.xub[p] : INFINITY;
    const bool leaves_at_upper = bounded && forced_p < 0 && p >= 0 && alpha_pq < 0.0;
    const double xp = (forced_p >= 0 || !bounded) ? fmax(xb_p, 0.0) / alpha_pq
                                                  : (leaves_at_upper ? fmax(up_p - xb_p, 0.0) : fmax(xb_p, 0.0)) / fabs(alpha_pq);
    const bool flip = bounded && forced_p < 0 && ub_q < INFINITY && (p < 0 || ub_q <= xp);
    if (p < 0 && !flip) {
        if (threadIdx.x == 0) {
            if (mode == 0) ctl->status = ST_UNBOUNDED;
            ctl->q = q;
            ctl->p = -1;
            ctl->pending = 0;
            ctl->forced_q = -1;
            ctl->forced_p = -1;
        }
        return;
    }
    if (mode == 2) {
        if (threadIdx.x == 0) {
            ctl->q = q;
            ctl->p = flip ? -1 : p;
            ctl->cbar_q = cbar_signed;
            ctl->gamma_q = gamma_q;
            ctl->pending = 0;
            ctl->forced_q = -1;
            ctl->forced_p = -1;
        }
        return;
    }
    const int leaving = p >= 0 ? lp.basis[p] : -1;
    const int leaving_flipped = (bounded && leaving >= 0) ? lp.flipped[leaving] : 0;
    __syncthreads();  // every thread has read xB[p], basis[p], flipped[..] before they are overwritten
    if (flip) {
        // bound flip: x_q runs from 0 to ub_q, the basis does not change; x_q is complemented so that it sits at 0 again
        for (int i = threadIdx.x; i < m; i += blockDim.x) lp.xB[i] -= lp.alpha[i] * ub_q;
        for (int e = lp.col_start[q] + threadIdx.x; e < lp.col_start[q + 1]; e += blockDim.x)
            lp.rhs[lp.row_index[e]] -= ub_q * sgn_q * lp.value[e];
        if (threadIdx.x == 0) {
            const int now_flipped = (sgn_q < 0.0) ? 0 : 1;
            lp.flipped[q] = now_flipped;
            lp.pos[q] = now_flipped ? -2 : -1;
            ctl->flip_cost += (now_flipped ? 1.0 : -1.0) * ub_q * lp.cost[q];
            ctl->q = q;
            ctl->p = -1;
            ctl->cbar_q = cbar_signed;
            ctl->minus_obj -= cbar_signed * ub_q;
            ctl->iters += 1;
            ctl->bound_flips += 1;
            ctl->pending = 0;
            ctl->forced_q = -1;
            ctl->forced_p = -1;
            ctl->last_selected = q;
        }
        return;
    }

    // ---- x_B update (carry/mod.rs:295-325) and the ordered non-zero list of alpha for K3 ----------------
    // ordered list of the rows K3 has to touch (alpha_i != 0, plus p): ballots + one prefix per round of K2_U * blockDim rows
    constexpr int K2_NW = K2_THREADS / WAVE;
    __shared__ int s_nz_wave[K2_U * K2_NW + 1];
    int total = 0;
    for (int i0 = 0; i0 < m; i0 += K2_U * blockDim.x) {
        const int lane_k2 = threadIdx.x & (WAVE - 1), wave_k2 = threadIdx.x / WAVE;
        double av[K2_U], xbv[K2_U];
        unsigned long long masks[K2_U];
#pragma unroll
        for (int u = 0; u < K2_U; ++u) {
            const int i = i0 + u * blockDim.x + threadIdx.x;
            av[u] = i < m ? lp.alpha[i] : 0.0;
            xbv[u] = i < m ? lp.xB[i] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < K2_U; ++u) {
            const int i = i0 + u * blockDim.x + threadIdx.x;
            masks[u] = __ballot(i < m && (av[u] != 0.0 || i == p));
            if (lane_k2 == 0) s_nz_wave[u * K2_NW + wave_k2] = __popcll(masks[u]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {  // exclusive prefix over the K2_U * K2_NW wave counts (rows ascend with (u, wave, lane))
            int running = 0;
            for (int e = 0; e < K2_U * K2_NW; ++e) {
                const int c = s_nz_wave[e];
                s_nz_wave[e] = running;
                running += c;
            }
            s_nz_wave[K2_U * K2_NW] = running;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < K2_U; ++u) {
            const int i = i0 + u * blockDim.x + threadIdx.x;
            if (i >= m) continue;
            if (av[u] != 0.0 || i == p) {
                const int slot = total + s_nz_wave[u * K2_NW + wave_k2] + __popcll(masks[u] & ((1ull << lane_k2) - 1ull));
                lp.nz_index[slot] = i;
                lp.nz_alpha[slot] = av[u];
            }
            lp.xB[i] = (i == p) ? xp : xbv[u] - av[u] * xp;
        }
        total += s_nz_wave[K2_U * K2_NW];
    }
    if (leaves_at_upper) {  // the leaving variable reached its upper bound: hold it in complemented form from now on
        const double sgn_l = leaving_flipped ? -1.0 : 1.0;
        for (int e = lp.col_start[leaving] + threadIdx.x; e < lp.col_start[leaving + 1]; e += blockDim.x)
            lp.rhs[lp.row_index[e]] -= up_p * sgn_l * lp.value[e];
    }
    if (threadIdx.x == 0) {
        lp.basis[p] = q;
        if (lp.track_touched && lp.eta_cap == 0 && !lp.touched[p]) {  // column p of the inverse stops being a unit vector
            const int count = ctl->touched_count;
            lp.touched[p] = 1;
            lp.tlist[count] = p;
            ctl->touched_count = count + 1;
        }
        lp.pos[q] = p;
        if (bounded) {
            int fl = leaving_flipped;
            if (leaves_at_upper) {
                fl ^= 1;
                lp.flipped[leaving] = fl;
                ctl->flip_cost += (fl ? 1.0 : -1.0) * up_p * lp.cost[leaving];
            }
            lp.pos[leaving] = fl ? -2 : -1;
            lp.xub[p] = ub_q;
        } else {
            lp.pos[leaving] = -1;
        }
        ctl->q = q;
        ctl->p = p;
        ctl->leaving = leaving;
        ctl->cbar_q = cbar_signed;
        ctl->alpha_pq = alpha_pq;
        ctl->gamma_q = gamma_q;
        ctl->xp = xp;
        ctl->nz_count = total;
        ctl->minus_obj -= cbar_signed * xp;
        ctl->iters += 1;
        ctl->pending = 1;
        ctl->rho_buf ^= 1;  // (the update of this pivot marks the other half of rho_bits)
        ctl->forced_q = -1;
        ctl->forced_p = -1;
        ctl->last_selected = q;
    }
}

// ---------------------------------------------------------------------------------------------------
// K2 for large m (beyond the register-resident kernel): the same steps spread over ceil(m / 1024) workgroups.  One
// workgroup sustains ~60 GB/s, and the entering-column / ratio-test step moves ~8 MB at m = 65 534 (150 us in
// ftran_ratio_kernel); three short kernels exchange per-workgroup partials instead:
//   k2l_ftran   entering column (every workgroup reduces the candidates itself), alpha for its rows, partial
//               |alpha|^2 and Harris pass-1 minimum
//   k2l_harris  fold the partials, Harris pass 2 on its rows -> per-workgroup candidate, count of non-zero alpha
//   k2l_apply   every workgroup takes the decision (pivot row, step length, bound flip or pivot) from those candidates,
//               updates x_B on its rows and writes its piece of the ordered list of touched rows for K3;
//               workgroup 0 does the O(1) bookkeeping
// Same rules and tie-breaks as ftran_ratio_kernel (mode 0 only; the fine-grained operations keep that kernel).
// ---------------------------------------------------------------------------------------------------
constexpr int K2L_THREADS = 1024;
constexpr int K2L_PD = 8;  // doubles per workgroup in k2_partd: |alpha|^2 sum, pass-1 minimum, candidate key, its alpha, x_B, upper bound
template <int RULE>
__global__ void __launch_bounds__(K2L_THREADS) k2l_ftran_kernel(DeviceLP lp, int n_price_blocks, double tol_pivot,
                                                              double harris_delta, int skip_artificial_rows) {
    __shared__ Cand s_cand[18];
    __shared__ double s_red[18];
    __shared__ int s_q, s_inline, s_short;
    __shared__ double s_cbar;
    __shared__ int s_rows[K2_COL_CHUNK];
    __shared__ double s_vals[K2_COL_CHUNK];
    Ctl* ctl = lp.ctl;
    // ---- ONE memory round trip: the control word, this thread's row, and the candidate of "its" pricing workgroup with everything
    //      the winner will need (reduced cost, length, the first two entries of the column: all of an incidence column).  Round 3
    //      read them one after the other -- control word, candidates, the winner's reduced cost and length, its entries, the
    //      inverse: five dependent trips of ~1.5 us in a 9.7 us kernel (config 5).
    const int m = lp.m, ld = lp.ld;
    const bool bounded = lp.ub != nullptr;
    const int i = blockIdx.x * K2L_THREADS + threadIdx.x;
    const double xb = i < m ? lp.xB[i] : 0.0;
    const int bas = i < m ? lp.basis[i] : 0;
    const double up = (bounded && i < m) ? lp.xub[i] : INFINITY;
    const int b_mine = threadIdx.x;
    int mine_j = -1, mine_len = -1, mine_row0 = 0, mine_row1 = 0;
    double mine_key = 0.0, mine_cbar = 0.0, mine_val0 = 0.0, mine_val1 = 0.0;
    if (b_mine < n_price_blocks) {
        mine_j = lp.cand_j[b_mine];
        mine_key = lp.cand_key[b_mine];
        mine_cbar = lp.cand_cbar[b_mine];
        mine_len = lp.cand_len[b_mine];
        mine_row0 = lp.cand_rows[(size_t)b_mine * ELL_W];
        mine_row1 = lp.cand_rows[(size_t)b_mine * ELL_W + 1];
        mine_val0 = lp.cand_vals[(size_t)b_mine * ELL_W];
        mine_val1 = lp.cand_vals[(size_t)b_mine * ELL_W + 1];
    }
    const int status = ctl->status;
    const long long iters = ctl->iters, budget = ctl->budget;
    const int forced_q = ctl->forced_q;
    if (status != ST_RUNNING) return;
    const bool publisher = blockIdx.x == 0 && threadIdx.x == 0;
    if (iters >= budget) {
        if (publisher) {
            ctl->status = ST_BUDGET;
            ctl->pending = 0;
        }
        return;
    }
    if (forced_q < 0) {
        Cand c;
        c.key = 0.0;
        c.idx = -1;
        c.aux = 0;
        for (int b = threadIdx.x; b < n_price_blocks; b += blockDim.x) {
            Cand o;
            o.idx = b == b_mine ? mine_j : lp.cand_j[b];
            o.key = o.idx >= 0 ? (b == b_mine ? mine_key : lp.cand_key[b]) : 0.0;
            o.aux = b;
            c = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? better<TIE_LARGER_IDX>(c, o) : better<TIE_SMALLER_IDX>(c, o);
        }
        c = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? block_best<TIE_LARGER_IDX>(c, s_cand) : block_best<TIE_SMALLER_IDX>(c, s_cand);
        if (c.idx < 0) {
            if (threadIdx.x == 0) {
                s_q = -1;
                s_cbar = 0.0;
                s_inline = -1;
                s_short = 0;
            }
        } else if (c.aux < (int)blockDim.x) {  // the winner is some thread's preloaded candidate: that thread publishes it
            if ((int)threadIdx.x == c.aux) {
                s_q = c.idx;
                s_cbar = mine_cbar;
                s_inline = (mine_len >= 0 && mine_len <= ELL_W) ? c.aux : -1;
                s_short = mine_len >= 0 && mine_len <= 2;
                s_rows[0] = mine_row0;
                s_rows[1] = mine_row1;
                s_vals[0] = mine_len >= 1 ? mine_val0 : 0.0;
                s_vals[1] = mine_len >= 2 ? mine_val1 : 0.0;
            }
        } else if (threadIdx.x == 0) {  // (more pricing workgroups than threads here: the later ones are read now)
            s_q = c.idx;
            s_cbar = lp.cand_cbar[c.aux];
            s_inline = (lp.cand_len[c.aux] >= 0 && lp.cand_len[c.aux] <= ELL_W) ? c.aux : -1;
            s_short = 0;
        }
    } else if (threadIdx.x == 0) {
        s_inline = -1;
        s_short = 0;
        s_q = forced_q;
        double cb = lp.cost[forced_q];
        for (int e = lp.col_start[forced_q]; e < lp.col_start[forced_q + 1]; ++e) cb += lp.value[e] * lp.minus_pi[lp.row_index[e]];
        if (lp.ub && lp.flipped[forced_q]) cb = -cb;
        s_cbar = cb;
    }
    __syncthreads();
    const int q = s_q;
    if (publisher) {
        ctl->q = q;
        ctl->cbar_q = s_cbar;
        if (q < 0) {
            ctl->status = ST_NO_ENTERING;
            ctl->pending = 0;
            ctl->last_selected = -1;
        }
    }
    if (q < 0) return;
    const double sgn_q = (bounded && lp.flipped[q]) ? -1.0 : 1.0;
    double acc = 0.0;
    const int inline_block = s_inline;
    if (inline_block >= 0 && s_short) {
        // at most two entries, already here (every column of a graph provider): straight to the two columns of the inverse
        const int r0 = s_rows[0], r1 = s_rows[1];
        const double v0 = s_vals[0], v1 = s_vals[1];
        const double t0 = (i < m && v0 != 0.0) ? lp.Binv[(size_t)r0 * ld + i] : 0.0;
        const double t1 = (i < m && v1 != 0.0) ? lp.Binv[(size_t)r1 * ld + i] : 0.0;
        acc += t0 * v0;
        acc += t1 * v1;
    } else if (inline_block >= 0) {
        // the winning pricing workgroup published the column's padded entries: no col_start -> row_index -> value chain
        int rows[ELL_W];
        double vals[ELL_W], t[ELL_W];
#pragma unroll
        for (int e = 0; e < ELL_W; ++e) {
            rows[e] = lp.cand_rows[(size_t)inline_block * ELL_W + e];
            vals[e] = lp.cand_vals[(size_t)inline_block * ELL_W + e];
        }
#pragma unroll
        for (int e = 0; e < ELL_W; ++e) t[e] = (i < m && vals[e] != 0.0) ? lp.Binv[(size_t)rows[e] * ld + i] : 0.0;
#pragma unroll
        for (int e = 0; e < ELL_W; ++e) acc += t[e] * vals[e];
    }
    const int ca = inline_block >= 0 ? 0 : lp.col_start[q], cb_ = inline_block >= 0 ? 0 : lp.col_start[q + 1];
    for (int c0 = ca; c0 < cb_; c0 += K2_COL_CHUNK) {
        const int cnt = min(K2_COL_CHUNK, cb_ - c0);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
            s_rows[e] = lp.row_index[c0 + e];
            s_vals[e] = lp.value[c0 + e];
        }
        __syncthreads();
        if (i < m) {
            double a0 = 0.0, a1 = 0.0;
            int e = 0;
            for (; e + 2 <= cnt; e += 2) {
                a0 += lp.Binv[(size_t)s_rows[e] * ld + i] * s_vals[e];
                a1 += lp.Binv[(size_t)s_rows[e + 1] * ld + i] * s_vals[e + 1];
            }
            for (; e < cnt; ++e) a0 += lp.Binv[(size_t)s_rows[e] * ld + i] * s_vals[e];
            acc += a0 + a1;
        }
    }
    const double a = acc * sgn_q;
    double sumsq = 0.0, theta = INFINITY;
    if (i < m) {
        lp.alpha[i] = a;
        sumsq = a * a;
        if (!(skip_artificial_rows && bas < lp.n_art)) {
            if (a > tol_pivot) theta = (fmax(xb, 0.0) + harris_delta) / a;
            else if (bounded && a < -tol_pivot && up < INFINITY) theta = (fmax(up - xb, 0.0) + harris_delta) / -a;
        }
    }
    sumsq = block_reduce<0>(sumsq, s_red);
    __syncthreads();
    theta = block_reduce<1>(theta, s_red);
    if (threadIdx.x == 0) {
        lp.k2_partd[K2L_PD * blockIdx.x] = sumsq;
        lp.k2_partd[K2L_PD * blockIdx.x + 1] = theta;
    }
}

__global__ void __launch_bounds__(K2L_THREADS) k2l_harris_kernel(DeviceLP lp, int n_blocks, double tol_pivot, int skip_artificial_rows) {
    __shared__ Cand s_cand[18];
    __shared__ double s_red[18];
    Ctl* ctl = lp.ctl;
    const int m = lp.m;
    // this thread's row and its first partials in the same round trip as the control word (round 3: after it)
    const bool bounded = lp.ub != nullptr;
    const int i = blockIdx.x * K2L_THREADS + threadIdx.x;
    const double a = i < m ? lp.alpha[i] : 0.0;
    const double xb = i < m ? lp.xB[i] : 0.0;
    const double up = (bounded && i < m) ? lp.xub[i] : INFINITY;
    const int bas = i < m ? lp.basis[i] : 0;
    const bool has_first = (int)threadIdx.x < n_blocks;
    const double first_sum = has_first ? lp.k2_partd[K2L_PD * threadIdx.x] : 0.0;
    const double first_min = has_first ? lp.k2_partd[K2L_PD * threadIdx.x + 1] : INFINITY;
    const int status = ctl->status, q_now = ctl->q;
    const int forced_p = ctl->forced_p;
    if (status != ST_RUNNING || q_now < 0) return;
    double v1 = 0.0, v2 = INFINITY;
    if (has_first) {
        v1 += first_sum;
        v2 = fmin(v2, first_min);
    }
    for (int b = threadIdx.x + blockDim.x; b < n_blocks; b += blockDim.x) {  // fixed order: deterministic
        v1 += lp.k2_partd[K2L_PD * b];
        v2 = fmin(v2, lp.k2_partd[K2L_PD * b + 1]);
    }
    const double gamma_q = 1.0 + block_reduce<0>(v1, s_red);
    __syncthreads();
    const double theta_max = block_reduce<1>(v2, s_red);
    __syncthreads();
    Cand c;
    c.key = 0.0;
    c.idx = -1;
    c.aux = 0;
    if (forced_p >= 0) {  // Carry::bring_into_basis with a given row: that row is the only candidate
        if (i == forced_p) {
            c.key = 1.0;
            c.idx = i;
            c.aux = bas;
        }
    } else if (i < m && !(skip_artificial_rows && bas < lp.n_art)) {
        double room = -1.0;
        if (a > tol_pivot) room = fmax(xb, 0.0);
        else if (bounded && a < -tol_pivot && up < INFINITY) room = fmax(up - xb, 0.0);
        if (room >= 0.0 && room / fabs(a) <= theta_max) {
            c.key = fabs(a);
            c.idx = i;
            c.aux = bas;
        }
    }
    c = block_best<TIE_SMALLER_AUX>(c, s_cand);
    __syncthreads();
    const double count = block_reduce<0>((i < m && a != 0.0) ? 1.0 : 0.0, s_red);
    if (c.idx >= 0 && c.idx == i) {  // the row that won brings everything the decision needs
        lp.k2_partd[K2L_PD * blockIdx.x + 2] = c.key;
        lp.k2_partd[K2L_PD * blockIdx.x + 3] = a;
        lp.k2_partd[K2L_PD * blockIdx.x + 4] = xb;
        lp.k2_partd[K2L_PD * blockIdx.x + 5] = up;
    }
    if (threadIdx.x == 0) {
        lp.k2_parti[4 * blockIdx.x] = c.idx;
        lp.k2_parti[4 * blockIdx.x + 1] = c.aux;
        lp.k2_parti[4 * blockIdx.x + 2] = (int)count;
        if (blockIdx.x == 0) {
            ctl->gamma_q = gamma_q;
            ctl->k2_forced = forced_p >= 0 ? 1 : 0;  // k2l_apply resets forced_p while other workgroups still decide
        }
    }
}

// Every workgroup takes the same decision from the per-workgroup candidates (no row data is read again, so the x_B
// writes of one workgroup cannot be seen by another's decision); workgroup 0 alone does the bookkeeping.
__global__ void __launch_bounds__(K2L_THREADS) k2l_apply_kernel(DeviceLP lp, int n_blocks) {
    __shared__ Cand s_cand[18];
    __shared__ double s_red[18];
    __shared__ int s_count[K2L_THREADS / WAVE + 1];
    Ctl* ctl = lp.ctl;
    // (this thread's row and its first candidate in the same round trip as the control word)
    const bool bounded = lp.ub != nullptr;
    const int m = lp.m;
    const int i = blockIdx.x * K2L_THREADS + threadIdx.x;
    const double a = i < m ? lp.alpha[i] : 0.0;  // in flight while the decision is taken
    const double xb_i = i < m ? lp.xB[i] : 0.0;
    const bool has_first = (int)threadIdx.x < n_blocks;
    const int first_idx = has_first ? lp.k2_parti[4 * threadIdx.x] : -1;
    const int first_aux = has_first ? lp.k2_parti[4 * threadIdx.x + 1] : 0;
    const int first_count = has_first ? lp.k2_parti[4 * threadIdx.x + 2] : 0;
    const double first_key = has_first ? lp.k2_partd[K2L_PD * threadIdx.x + 2] : 0.0;
    const int status = ctl->status;
    const int q = ctl->q;
    const bool forced = ctl->k2_forced != 0;
    if (status != ST_RUNNING) return;  // (workgroup 0 may set UNBOUNDED below: the others then have nothing to apply either)
    if (q < 0) return;
    Cand c;
    c.key = 0.0;
    c.idx = -1;
    c.aux = 0;
    double before = 0.0, all = 0.0;  // list entries of the workgroups ahead of this one / of all of them
    for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) {
        const bool first = b == (int)threadIdx.x;
        Cand o;
        o.idx = first ? first_idx : lp.k2_parti[4 * b];
        o.key = o.idx >= 0 ? (first ? first_key : lp.k2_partd[K2L_PD * b + 2]) : 0.0;
        o.aux = first ? first_aux : lp.k2_parti[4 * b + 1];  // the basic column of that row: ties go to the smaller one
        const int count = first ? first_count : lp.k2_parti[4 * b + 2];
        all += count;
        if (b < (int)blockIdx.x) before += count;
        if (o.idx >= 0) c = better<TIE_SMALLER_AUX>(c, o);
    }
    c = block_best<TIE_SMALLER_AUX>(c, s_cand);
    __syncthreads();
    const int offset = (int)block_reduce<0>(before, s_red);
    __syncthreads();
    const int total = (int)block_reduce<0>(all, s_red);
    const int p = c.idx;
    const int wb = p >= 0 ? p / K2L_THREADS : 0;  // the workgroup that owns row p published its data
    const double alpha_pq = p >= 0 ? lp.k2_partd[K2L_PD * wb + 3] : 1.0;
    const double xb_p = p >= 0 ? lp.k2_partd[K2L_PD * wb + 4] : 0.0;
    const double up_p = p >= 0 ? lp.k2_partd[K2L_PD * wb + 5] : INFINITY;
    const int leaving = c.aux;
    const double ub_q = bounded ? lp.ub[q] : INFINITY;
    const bool leaves_at_upper = bounded && !forced && p >= 0 && alpha_pq < 0.0;
    const double xp = (forced || !bounded) ? fmax(xb_p, 0.0) / alpha_pq
                                           : (leaves_at_upper ? fmax(up_p - xb_p, 0.0) : fmax(xb_p, 0.0)) / fabs(alpha_pq);
    const bool flip = bounded && !forced && ub_q < INFINITY && (p < 0 || ub_q <= xp);
    if (p >= 0 || flip) {
        if (flip) {
            if (i < m) lp.xB[i] = xb_i - a * ub_q;
        } else {
            const bool keep = i < m && a != 0.0;
            const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
            const unsigned long long mask = __ballot(keep);
            if (lane == 0) s_count[wave] = __popcll(mask);
            __syncthreads();
            int base = offset;
            for (int wv = 0; wv < wave; ++wv) base += s_count[wv];
            if (keep) {
                const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                lp.nz_index[slot] = i;
                lp.nz_alpha[slot] = a;
            }
            if (i < m) lp.xB[i] = (i == p) ? xp : xb_i - a * xp;
        }
    }
    if (blockIdx.x != 0) return;
    // ---- bookkeeping (workgroup 0): same statements as the tail of ftran_ratio_kernel ----------------------------
    __shared__ int s_toggle;
    __shared__ double s_amount;
    if (threadIdx.x == 0) {
        const double cbar_q = ctl->cbar_q;
        int toggle = -1;
        double amount = 0.0;
        if (p < 0 && !flip) {
            ctl->status = ST_UNBOUNDED;
            ctl->p = -1;
            ctl->pending = 0;
        } else if (flip) {
            const int was = lp.flipped[q];
            toggle = q;
            amount = ub_q * (was ? -1.0 : 1.0);  // rhs -= ub * (signed column)
            lp.flipped[q] = was ^ 1;
            lp.pos[q] = (was ^ 1) ? -2 : -1;
            ctl->flip_cost += ((was ^ 1) ? 1.0 : -1.0) * ub_q * lp.cost[q];
            ctl->p = -1;
            ctl->xp = ub_q;
            ctl->minus_obj -= cbar_q * ub_q;
            ctl->iters += 1;
            ctl->bound_flips += 1;
            ctl->pending = 0;
            ctl->last_selected = q;
        } else {
            lp.basis[p] = q;
            if (lp.track_touched && lp.eta_cap == 0 && !lp.touched[p]) {  // column p of the inverse stops being a unit vector
                const int count = ctl->touched_count;
                lp.touched[p] = 1;
                lp.tlist[count] = p;
                ctl->touched_count = count + 1;
            }
            lp.pos[q] = p;
            if (bounded) {
                int fl = lp.flipped[leaving];
                if (leaves_at_upper) {
                    toggle = leaving;
                    amount = up_p * (fl ? -1.0 : 1.0);
                    fl ^= 1;
                    lp.flipped[leaving] = fl;
                    ctl->flip_cost += (fl ? 1.0 : -1.0) * up_p * lp.cost[leaving];
                }
                lp.pos[leaving] = fl ? -2 : -1;
                lp.xub[p] = ub_q;
            } else {
                lp.pos[leaving] = -1;
            }
            ctl->p = p;
            ctl->leaving = leaving;
            ctl->alpha_pq = alpha_pq;
            ctl->xp = xp;
            ctl->nz_count = total;
            ctl->minus_obj -= cbar_q * xp;
            ctl->iters += 1;
            ctl->pending = 1;
            ctl->rho_buf ^= 1;  // (the update of this pivot marks the other half of rho_bits)
            ctl->last_selected = q;
        }
        ctl->forced_q = -1;
        ctl->forced_p = -1;
        s_toggle = toggle;
        s_amount = amount;
    }
    __syncthreads();
    const int toggle = s_toggle;
    if (toggle >= 0) {  // the complemented column moves u_j a_j to the right-hand side
        const double amount = s_amount;
        for (int e = lp.col_start[toggle] + threadIdx.x; e < lp.col_start[toggle + 1]; e += blockDim.x)
            lp.rhs[lp.row_index[e]] -= amount * lp.value[e];
    }
}

// ---------------------------------------------------------------------------------------------------
// K2, register-resident variant for m <= R*K2F_THREADS (the common case).  Same contract as ftran_ratio_kernel, but
// the dependent chain of global-memory round trips (~1.2 k cycles each when the data was produced by the previous
// kernel on another XCD) is cut to four: {control word, candidates, own x_B/basis rows} -> {column extent} ->
// {column entries} -> {inverse columns}.  alpha_i, x_B,i and basis_i of the rows a thread owns stay in registers;
// scalars are exchanged through LDS.
// ---------------------------------------------------------------------------------------------------
constexpr int K2F_THREADS = 512;
constexpr int K2F_MAX_BLOCKS = 2048;
constexpr int K2F_INLINE_BLOCKS = 128;  // candidate columns staged with the candidates when there are at most this many
template <int RULE, int R>
__global__ void __launch_bounds__(K2F_THREADS) ftran_ratio_fast_kernel(DeviceLP lp, int n_price_blocks, double tol_pivot,
                                                                     double harris_delta, int skip_artificial_rows,
                                                                     int mode, int n_alpha_slices) {
    // n_alpha_slices > 0: q was chosen by select_kernel and alpha comes from ftran_partial_kernel's slices
    __shared__ double s_akey[K2F_THREADS / WAVE];
    __shared__ unsigned long long s_arank[K2F_THREADS / WAVE];
    __shared__ double s_red[K2F_THREADS / WAVE + 2];
    __shared__ double s_red2[K2F_THREADS / WAVE + 2];
    __shared__ double s_cbarv[K2F_MAX_BLOCKS];
    __shared__ int s_crows[K2F_INLINE_BLOCKS * ELL_W];
    __shared__ double s_cvals[K2F_INLINE_BLOCKS * ELL_W];
    __shared__ int s_clen[K2F_INLINE_BLOCKS];
    __shared__ int s_rows[K2_COL_CHUNK];
    __shared__ double s_vals[K2_COL_CHUNK];
    __shared__ double s_bcast[4];
    __shared__ int s_ibcast[2];
    Ctl* ctl = lp.ctl;
    STAMP_INIT;
    const int tid = threadIdx.x;
    const int m = lp.m, ld = lp.ld;
    // ---- round trip 1: everything that does not depend on q -----------------------------------------
    const int status = ctl->status;
    const long long iters = ctl->iters;
    const long long budget = ctl->budget;
    const int forced_q = ctl->forced_q;
    const int forced_p = ctl->forced_p;
    const double minus_obj = ctl->minus_obj;
    double xb[R];
    int bas[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * K2F_THREADS;
        xb[r] = i < m ? lp.xB[i] : 0.0;
        bas[r] = i < m ? lp.basis[i] : 0x7fffffff;
    }
    // candidate of this thread: key, rank = (tie rule on the column index) << 16 | pricing workgroup
    double ckey = 0.0;
    unsigned long long crank = RANK_NONE;
    const bool preselected = n_alpha_slices > 0;
    // preselected: q, its reduced cost and the whole column alpha_in were left by earlier kernels -- requested in this round trip too
    const int q_pre = preselected ? ctl->q : -1;
    const double cbar_pre = preselected ? ctl->cbar_q : 0.0;
    double ain[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * K2F_THREADS;
        ain[r] = (preselected && i < m) ? lp.alpha_in[i] : 0.0;
    }
    if (!preselected) {  // NOT conditional on forced_q (a value still in flight): that would put these loads a round trip later
        for (int b = tid; b < n_price_blocks; b += K2F_THREADS) {
            const int j = lp.cand_j[b];
            const double k = lp.cand_key[b];
            s_cbarv[b] = lp.cand_cbar[b];
            if (j >= 0) {
                const unsigned long long order = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? (unsigned long long)(0x7fffffff - j) : (unsigned long long)j;
                const unsigned long long r = (order << 16) | (unsigned long long)b;
                if (crank == RANK_NONE || k > ckey || (k == ckey && r < crank)) {
                    ckey = k;
                    crank = r;
                }
            }
        }
        if (n_price_blocks <= K2F_INLINE_BLOCKS) {
            for (int e = tid; e < n_price_blocks * ELL_W; e += K2F_THREADS) {
                s_crows[e] = lp.cand_rows[e];
                s_cvals[e] = lp.cand_vals[e];
            }
            for (int b = tid; b < n_price_blocks; b += K2F_THREADS) s_clen[b] = lp.cand_len[b];
        }
    }
    const bool inline_column = forced_q < 0 && !preselected && n_price_blocks <= K2F_INLINE_BLOCKS;
    if (status != ST_RUNNING) return;
    STAMP(0);
    if (mode == 0 && iters >= budget) {
        if (tid == 0) {
            ctl->status = ST_BUDGET;
            ctl->pending = 0;
        }
        return;
    }
    // ---- entering column --------------------------------------------------------------------------
    int q;
    double cbar_q;
    int winner_block = 0;
    if (preselected && forced_q < 0) {
        q = q_pre;
        cbar_q = cbar_pre;
    } else if (forced_q < 0) {
        block_argbest(ckey, crank, s_akey, s_arank);
        if (crank == RANK_NONE) {
            q = -1;
            cbar_q = 0.0;
        } else {
            winner_block = (int)(crank & 0xffff);
            const int order = (int)(crank >> 16);
            q = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? 0x7fffffff - order : order;
            cbar_q = s_cbarv[winner_block];
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
    STAMP(1);
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
    // ---- FTRAN (round trips 2-4) --------------------------------------------------------------------
    double al[R];
#pragma unroll
    for (int r = 0; r < R; ++r) al[r] = 0.0;
    int ca = 0, cb_ = 0;
    if (inline_column) {
        // the padded entries of the winning column arrived with the candidates: FTRAN starts without another fetch
        const int len = s_clen[winner_block];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * K2F_THREADS;
            if (i >= m || len < 0) continue;
            const double* col = lp.Binv + i;
            double t[ELL_W];
#pragma unroll
            for (int e = 0; e < ELL_W; ++e) t[e] = col[(size_t)s_crows[winner_block * ELL_W + e] * ld];  // padding: row 0, value 0
            double a0 = 0.0;
#pragma unroll
            for (int e = 0; e < ELL_W; ++e) a0 += t[e] * s_cvals[winner_block * ELL_W + e];
            al[r] = a0;
        }
        if (len > ELL_W || len < 0) {  // rare: the rest of a long column (or all of a dense one) from the CSC
            ca = lp.col_start[q] + (len < 0 ? 0 : ELL_W);
            cb_ = lp.col_start[q + 1];
        }
    } else if (!preselected) {
        ca = lp.col_start[q];
        cb_ = lp.col_start[q + 1];
    }
    if (preselected) {  // alpha_reduce_kernel left the whole column (pending etas applied) in alpha_in
#pragma unroll
        for (int r = 0; r < R; ++r) al[r] += ain[r];
    }
    for (int c0 = ca; c0 < cb_; c0 += K2_COL_CHUNK) {
        const int cnt = min(K2_COL_CHUNK, cb_ - c0);
        __syncthreads();
        for (int e = tid; e < cnt; e += K2F_THREADS) {
            s_rows[e] = lp.row_index[c0 + e];
            s_vals[e] = lp.value[c0 + e];
        }
        __syncthreads();
        // Per row: four chains over the entries e = 0, 1, 2, 3 (mod 4), leftovers onto the first, (a0 + a1) + (a2 + a3) -- the
        // arithmetic of pivot_fused_kernel, bit for bit.  The rows of a thread go through the entry loop TOGETHER, up to eight at a
        // time: 32 loads in flight per thread instead of 4 (a row at a time, this one-workgroup FTRAN was a chain of round trips:
        // 7 of the 17 us of this kernel on GREENBEA).
        constexpr int RG = R <= 8 ? R : 1;
        if constexpr (R <= 8) {
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += RG) {
            double acc[RG][4];
            int row[RG];
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int i = tid + (r0 + r) * K2F_THREADS;
                row[r] = i < m ? i : 0;  // (rows past m read row 0 and are dropped below)
                acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0;
            }
            const double* T = lp.Binv;
            int e = 0;
            for (; e + 4 <= cnt; e += 4) {
                const size_t o0 = (size_t)s_rows[e] * ld, o1 = (size_t)s_rows[e + 1] * ld, o2 = (size_t)s_rows[e + 2] * ld, o3 = (size_t)s_rows[e + 3] * ld;
                const double v0 = s_vals[e], v1 = s_vals[e + 1], v2 = s_vals[e + 2], v3 = s_vals[e + 3];
                double x[RG][4];
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    x[r][0] = T[o0 + row[r]];
                    x[r][1] = T[o1 + row[r]];
                    x[r][2] = T[o2 + row[r]];
                    x[r][3] = T[o3 + row[r]];
                }
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    acc[r][0] += x[r][0] * v0;
                    acc[r][1] += x[r][1] * v1;
                    acc[r][2] += x[r][2] * v2;
                    acc[r][3] += x[r][3] * v3;
                }
            }
            for (; e < cnt; ++e) {
                const size_t o = (size_t)s_rows[e] * ld;
                const double v = s_vals[e];
                double x[RG];
#pragma unroll
                for (int r = 0; r < RG; ++r) x[r] = T[o + row[r]];
#pragma unroll
                for (int r = 0; r < RG; ++r) acc[r][0] += x[r] * v;
            }
#pragma unroll
            for (int r = 0; r < RG; ++r)
                if (tid + (r0 + r) * K2F_THREADS < m) al[r0 + r] += (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
        }
        } else {  // (16 rows per thread: the grouped form spills)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = tid + r * K2F_THREADS;
                if (i >= m) continue;
                const double* col = lp.Binv + i;
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                int e = 0;
                for (; e + 4 <= cnt; e += 4) {
                    a0 += col[(size_t)s_rows[e] * ld] * s_vals[e];
                    a1 += col[(size_t)s_rows[e + 1] * ld] * s_vals[e + 1];
                    a2 += col[(size_t)s_rows[e + 2] * ld] * s_vals[e + 2];
                    a3 += col[(size_t)s_rows[e + 3] * ld] * s_vals[e + 3];
                }
                for (; e < cnt; ++e) a0 += col[(size_t)s_rows[e] * ld] * s_vals[e];
                al[r] += (a0 + a1) + (a2 + a3);
            }
        }
    }
    STAMP(2);
    // ---- gamma_q and Harris pass 1, one combined block reduction ----------------------------------------
    // Implicit upper bounds (lp.ub): a basic variable may also leave at its upper bound -- rows with alpha_i < 0 whose
    // basic variable has one -- and the entering variable may run into its own bound first (a "bound flip", no basis
    // change).  Complemented columns enter with the opposite sign.
    const bool bounded = lp.ub != nullptr;
    double sgn_q = 1.0, ub_q = INFINITY;
    if (bounded) {
        sgn_q = lp.flipped[q] ? -1.0 : 1.0;
        ub_q = lp.ub[q];
        if (forced_q >= 0) cbar_q *= sgn_q;  // the candidates of the pricing pass carry the sign already
    }
    // harris_delta < 0 selects the reference's ratio test (tableau/mod.rs:287-313): the exact minimum ratio, ties to the lowest
    // leaving column (Bland) -- for data on which f64 is exact; the default is the Harris two-pass test f64 needs in general.
    const bool textbook = harris_delta < 0.0;
    const double harris_slack = textbook ? 0.0 : harris_delta;
    double sumsq = 0.0, theta = INFINITY;
    bool eligible[R];
    double room[R];  // distance of the basic variable to the bound it moves towards
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * K2F_THREADS;
        al[r] *= sgn_q;
        const double a = al[r];
        sumsq += a * a;
        const bool allowed = i < m && !(skip_artificial_rows && bas[r] < lp.n_art);
        room[r] = fmax(xb[r], 0.0);
        eligible[r] = allowed && a > tol_pivot;
        if (bounded && allowed && a < -tol_pivot) {
            const double up = lp.xub[i];
            if (up < INFINITY) {
                eligible[r] = true;
                room[r] = fmax(up - xb[r], 0.0);
            }
        }
        if (eligible[r]) theta = fmin(theta, (room[r] + harris_slack) / fabs(a));
    }
    {
        const int lane = tid & (WAVE - 1), wave = tid / WAVE;
        sumsq = wave_sum(sumsq);
        theta = wave_min(theta);
        if (lane == LAST) {
            s_red[wave] = sumsq;
            s_red2[wave] = theta;
        }
        __syncthreads();
        if (wave == 0) {
            double t1 = lane < K2F_THREADS / WAVE ? s_red[lane] : 0.0;
            double t2 = lane < K2F_THREADS / WAVE ? s_red2[lane] : INFINITY;
            t1 = wave_sum(t1);
            t2 = wave_min(t2);
            if (lane == LAST) {
                s_red[K2F_THREADS / WAVE] = t1;
                s_red2[K2F_THREADS / WAVE] = t2;
            }
        }
        __syncthreads();
    }
    const double gamma_q = 1.0 + s_red[K2F_THREADS / WAVE];  // pivot_rule.rs:258
    const double theta_max = s_red2[K2F_THREADS / WAVE];
    STAMP(3);
    // ---- Harris pass 2 ---------------------------------------------------------------------------------
    int p = forced_p;
    if (forced_p < 0) {
        // largest eligible pivot, ties by the lowest leaving column (Bland, tableau/mod.rs:295), then the lowest row
        double hkey = 0.0;
        unsigned long long hrank = RANK_NONE;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double mag = fabs(al[r]);
            const double key = textbook ? 1.0 : mag;
            if (eligible[r] && room[r] / mag <= theta_max) {
                const unsigned long long rk = ((unsigned long long)(unsigned)bas[r] << 32) | (unsigned)(tid + r * K2F_THREADS);
                if (hrank == RANK_NONE || key > hkey || (key == hkey && rk < hrank)) {
                    hkey = key;
                    hrank = rk;
                }
            }
        }
        block_argbest(hkey, hrank, s_akey, s_arank);
        p = hrank == RANK_NONE ? -1 : (int)(hrank & 0xffffffffu);
    }
    STAMP(4);
    // ---- broadcast the pivot row's scalars (its owner has them in registers) ------------------------------
    if (tid == 0) {
        s_bcast[1] = 1.0;
        s_bcast[2] = 0.0;
        s_bcast[3] = INFINITY;
        s_ibcast[0] = -1;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (tid + r * K2F_THREADS == p) {
            s_bcast[1] = al[r];
            s_bcast[2] = xb[r];
            s_bcast[3] = room[r];
            s_ibcast[0] = bas[r];
        }
    }
    __syncthreads();
    const double alpha_pq = s_bcast[1];
    const int leaving = s_ibcast[0];
    // step length: to the bound of the leaving variable, or (forced zero-level pivots) as the reference computes it
    double xp = (forced_p >= 0 || !bounded) ? fmax(s_bcast[2], 0.0) / alpha_pq : s_bcast[3] / fabs(alpha_pq);
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
    if (mode == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * K2F_THREADS;
            if (i < m) lp.alpha[i] = al[r];
        }
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
        // ---- bound flip: x_q runs from 0 to ub_q, the basis does not change; x_q is complemented so that it sits at 0 again
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * K2F_THREADS;
            if (i < m) lp.xB[i] = xb[r] - al[r] * ub_q;
        }
        for (int e = lp.col_start[q] + tid; e < lp.col_start[q + 1]; e += K2F_THREADS)
            lp.rhs[lp.row_index[e]] -= ub_q * sgn_q * lp.value[e];
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
            ctl->pending = 0;  // no basis change: no inverse update, no weight update
            ctl->forced_q = -1;
            ctl->forced_p = -1;
            ctl->last_selected = q;
        }
        return;
    }
    // ---- x_B update (carry/mod.rs:295-325) and alpha for K3 ------------------------------------------------
    const int eta_slot_p = (lp.eta_cap > 0 && tid == 0) ? lp.eta_slot[p] : 0;  // (requested here, used at the end)
    int total = 0;
    if (R >= 8 && lp.eta_cap == 0) {  // ordered list of the rows K3 has to touch (alpha_i != 0, plus p); rows ascend with (r, tid)
        constexpr int NW = K2F_THREADS / WAVE;
        __shared__ int s_nz_count[R * NW + 1];
        const int lane_k2 = tid & (WAVE - 1), wave_k2 = tid / WAVE;
        unsigned long long masks[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * K2F_THREADS;
            masks[r] = __ballot(i < m && (al[r] != 0.0 || i == p));
            if (lane_k2 == 0) s_nz_count[r * NW + wave_k2] = __popcll(masks[r]);
        }
        __syncthreads();
        if (wave_k2 == 0) {  // exclusive prefix over the R * NW wave counts: wave 0, 64 counts at a time (a serial loop of one
                             // thread over 64 - 128 LDS reads was 2 - 4 us of this kernel on the mid-size LPs)
            int carry = 0;
#pragma unroll
            for (int base = 0; base < R * NW; base += WAVE) {
                const int c = base + lane_k2 < R * NW ? s_nz_count[base + lane_k2] : 0;
                int incl = c;
#pragma unroll
                for (int d = 1; d < WAVE; d <<= 1) {
                    const int up = __shfl_up(incl, d);
                    if (lane_k2 >= d) incl += up;
                }
                if (base + lane_k2 < R * NW) s_nz_count[base + lane_k2] = carry + incl - c;
                carry += __shfl(incl, WAVE - 1);
            }
            if (lane_k2 == 0) s_nz_count[R * NW] = carry;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * K2F_THREADS;
            if (i < m && (al[r] != 0.0 || i == p)) {
                const int slot = s_nz_count[r * NW + wave_k2] + __popcll(masks[r] & ((1ull << lane_k2) - 1ull));
                lp.nz_index[slot] = i;
                lp.nz_alpha[slot] = al[r];
            }
        }
        total = s_nz_count[R * NW];
    }
    // (the stores come after the barriers of the list above: a barrier waits for every store in flight)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * K2F_THREADS;
        if (i < m) {
            lp.alpha[i] = al[r];
            lp.xB[i] = (i == p) ? xp : xb[r] - al[r] * xp;
        }
    }
    STAMP(5);
    const int leaving_flipped = bounded ? lp.flipped[leaving] : 0;
    if (leaves_at_upper) {  // the leaving variable reached its upper bound: hold it in complemented form from now on
        const double ub_l = s_bcast[2] + s_bcast[3];  // x_p + room = its upper bound
        const double sgn_l = leaving_flipped ? -1.0 : 1.0;
        for (int e = lp.col_start[leaving] + tid; e < lp.col_start[leaving + 1]; e += K2F_THREADS)
            lp.rhs[lp.row_index[e]] -= ub_l * sgn_l * lp.value[e];
    }
    if (bounded) __syncthreads();  // every thread has read flipped[leaving] before thread 0 rewrites it
    if (tid == 0) {
        lp.basis[p] = q;
        if (lp.track_touched && lp.eta_cap == 0 && !lp.touched[p]) {  // column p of the inverse stops being a unit vector
            const int count = ctl->touched_count;
            lp.touched[p] = 1;
            lp.tlist[count] = p;
            ctl->touched_count = count + 1;
        }
        lp.pos[q] = p;
        if (lp.eta_cap > 0) {  // deferred product form: row p gets a kept column of M unless it has one; one more version of them
            int is_new = 0;
            if (eta_slot_p < 0) {
                const int k = ctl->eta_count;
                lp.eta_slot[p] = k;
                lp.eta_rows[k] = p;
                ctl->eta_count = k + 1;
                is_new = 1;
            }
            ctl->eta_new = is_new;
            ctl->eta_version = ctl->eta_version + 1;
        }
        if (bounded) {
            int fl = leaving_flipped;
            if (leaves_at_upper) {
                fl ^= 1;
                lp.flipped[leaving] = fl;
                ctl->flip_cost += (fl ? 1.0 : -1.0) * (s_bcast[2] + s_bcast[3]) * lp.cost[leaving];
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
        ctl->nz_count = total;
        ctl->minus_obj = minus_obj - cbar_q * xp;
        ctl->iters = iters + 1;
        ctl->pending = 1;
        ctl->rho_buf ^= 1;  // (the update of this pivot marks the other half of rho_bits)
        ctl->forced_q = -1;
        ctl->forced_p = -1;
        ctl->last_selected = q;
    }
    STAMP(6);
}

// ---------------------------------------------------------------------------------------------------
// K23: ratio test AND inverse update in ONE launch (m <= 2048, explicit carry, no implicit bounds, no forced pivot): two
// kernels per pivot instead of three.  Every workgroup repeats the whole of K2 -- entering column, FTRAN, ratio test: 60 KB
// of L2 reads and three block reductions, identical bits in every workgroup -- and then its eight waves update one column of
// the inverse each, so the kernel boundary between K2 and K3 (a cold start on data written by another XCD, 1.5-2 us) and
// K3's own first round trip disappear.  Workgroups are not synchronised, so nothing a workgroup reads at its start may be
// written by another one before its end:
//   * x_B, the basis and the control block exist twice (`lp.state[2]`); pivot k of a batch reads copy k & 1 (the pricing pass
//     before it too) and workgroup 0 writes copy (k + 1) & 1 -- also when nothing happens (status != running: copied through);
//   * the inverse is updated OUT OF PLACE (T_old = buffer ctl->t_buf, T_new = the other one; every entry is written);
//   * -pi_j, rho_j, w_j are touched by the owner of column j only; column positions are written by workgroup 0 and read by the
//     next pricing pass;
//   * `begin_batch_kernel` / `commit_kernel` move the canonical arrays into copy 0 and the last copy (and the inverse) back, so
//     everything outside a batch of pivots sees the canonical arrays only.
// Same arithmetic in the same order as ftran_ratio_fast_kernel<RULE, R> + update_kernel<true>: the pivot sequence and every
// number are BIT-IDENTICAL with the three-kernel pivot (tests/test_gpu_fused.py).
// Measured and dropped: one wave per workgroup with all rows in registers (a wave cannot keep enough loads in
// flight: 25 us), wave-local decisions out of LDS (16 rows and 32 divisions per lane: 14 us), and ONE launch per pivot with
// every workgroup pricing all columns (360 KB of conflicting LDS gathers per workgroup: 21 us).
// ---------------------------------------------------------------------------------------------------
constexpr int KF_THREADS = 512;
constexpr int KF_NW = KF_THREADS / WAVE;  // waves = columns of the inverse per workgroup
constexpr int KF_MAX_R = 4;               // rows per thread: 2 up to 1024 rows, 4 up to 2048 (as ftran_ratio_fast_kernel<RULE, R>)
constexpr int KF_MAX_M = KF_MAX_R * KF_THREADS;
template <int RULE, int KF_R>
__global__ void __launch_bounds__(KF_THREADS) pivot_fused_kernel(DeviceLP lp, DeviceLP::State in, DeviceLP::State out, int n_price_blocks,
                                                                 double tol_pivot, double harris_delta, int skip_artificial_rows) {
    constexpr int R = KF_R;
    constexpr int KF_U = KF_R * KF_THREADS / WAVE;  // rows of a column per lane
    constexpr int KF_M = KF_R * KF_THREADS;
    __shared__ double s_akey[KF_NW];
    __shared__ unsigned long long s_arank[KF_NW];
    __shared__ double s_red[KF_NW + 2];
    __shared__ double s_red2[KF_NW + 2];
    __shared__ double s_cbarv[K2F_MAX_BLOCKS];
    __shared__ int s_crows[K2F_INLINE_BLOCKS * ELL_W];
    __shared__ double s_cvals[K2F_INLINE_BLOCKS * ELL_W];
    __shared__ int s_clen[K2F_INLINE_BLOCKS];
    __shared__ int s_rows[K2_COL_CHUNK];
    __shared__ double s_vals[K2_COL_CHUNK];
    __shared__ double s_alpha[KF_M];
    __shared__ double s_bcast[4];
    __shared__ int s_ibcast[2];
    const Ctl* ctl = in.ctl;  // (the two copies are kernel arguments: no dependent load to find them)
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1), wave = tid / WAVE;
#ifdef RELP_STAMPS
    unsigned long long t_prev__ = clock64();
    if (tid == 0 && blockIdx.x == 0) lp.dbg[63] += 1;
#define FSTAMP(k) do { if (tid == 0 && blockIdx.x == 0) { unsigned long long t__ = clock64(); lp.dbg[(k)] += t__ - t_prev__; t_prev__ = t__; } } while (0)
#else
#define FSTAMP(k) do {} while (0)
#endif
    const int m = lp.m, ld = lp.ld;
    const bool writer = blockIdx.x == 0;
    // ---- round trip 1: everything that does not depend on q -----------------------------------------
    const int status = ctl->status;
    const long long iters = ctl->iters;
    const long long budget = ctl->budget;
    const double minus_obj = ctl->minus_obj;
    const int t_buf = ctl->t_buf;
    double xb[R];
    int bas[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * KF_THREADS;
        xb[r] = i < m ? in.xB[i] : 0.0;
        bas[r] = i < m ? in.basis[i] : 0x7fffffff;
    }
    double ckey = 0.0;
    unsigned long long crank = RANK_NONE;
    for (int b = tid; b < n_price_blocks; b += KF_THREADS) {
        const int j = lp.cand_j[b];
        const double k = lp.cand_key[b];
        s_cbarv[b] = lp.cand_cbar[b];
        if (j >= 0) {
            const unsigned long long order = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? (unsigned long long)(0x7fffffff - j) : (unsigned long long)j;
            const unsigned long long r = (order << 16) | (unsigned long long)b;
            if (crank == RANK_NONE || k > ckey || (k == ckey && r < crank)) {
                ckey = k;
                crank = r;
            }
        }
    }
    const bool inline_column = n_price_blocks <= K2F_INLINE_BLOCKS;
    if (inline_column) {
        for (int e = tid; e < n_price_blocks * ELL_W; e += KF_THREADS) {
            s_crows[e] = lp.cand_rows[e];
            s_cvals[e] = lp.cand_vals[e];
        }
        for (int b = tid; b < n_price_blocks; b += KF_THREADS) s_clen[b] = lp.cand_len[b];
    }
    // workgroup 0 hands the state on when no pivot is made: x_B and the basis unchanged, the control block edited
    auto hand_on = [&](int new_status, int new_q, bool clear_last) {
        if (!writer) return;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * KF_THREADS;
            if (i < m) {
                out.xB[i] = xb[r];
                out.basis[i] = bas[r];
            }
        }
        if (tid == 0) {
            Ctl c = *ctl;  // (read again by one thread: preloading the whole block in round trip 1 was measured slower)
            if (new_status >= 0) {
                c.status = new_status;
                c.pending = 0;
                c.q = new_q;
                if (new_status == ST_UNBOUNDED) c.p = -1;
                if (clear_last) c.last_selected = -1;
            }
            c.forced_q = -1;
            c.forced_p = -1;
            *out.ctl = c;
        }
    };
    if (status != ST_RUNNING) {
        hand_on(-1, 0, false);
        return;
    }
    FSTAMP(0);
    if (iters >= budget) {
        hand_on(ST_BUDGET, ctl->q, false);
        return;
    }
    // ---- entering column --------------------------------------------------------------------------
    block_argbest(ckey, crank, s_akey, s_arank);
    if (crank == RANK_NONE) {
        hand_on(ST_NO_ENTERING, -1, true);
        return;
    }
    const int winner_block = (int)(crank & 0xffff);
    const int order = (int)(crank >> 16);
    const int q = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? 0x7fffffff - order : order;
    const double cbar_q = s_cbarv[winner_block];
    FSTAMP(1);
    // ---- this wave's column of the inverse: issued together with the FTRAN loads (both only needed round trip 1) ---
    const int j_own = blockIdx.x * KF_NW + wave;
    const bool own = j_own < m;
    const double* t_old = (t_buf ? lp.Binv2 : lp.Binv);
    double* t_new = (t_buf ? lp.Binv : lp.Binv2);
    const double* c_old = t_old + (size_t)(own ? j_own : 0) * ld;
    double o[KF_U];
#pragma unroll
    for (int u = 0; u < KF_U; ++u) {
        const int i = lane + u * WAVE;
        o[u] = (own && i < m) ? c_old[i] : 0.0;
    }
    const double pi_old = (own && lane == LAST) ? lp.minus_pi[j_own] : 0.0;
    // ---- FTRAN -------------------------------------------------------------------------------------
    double al[R];
#pragma unroll
    for (int r = 0; r < R; ++r) al[r] = 0.0;
    int ca = 0, cb_ = 0;
    if (inline_column) {
        const int len = s_clen[winner_block];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * KF_THREADS;
            if (i >= m || len < 0) continue;
            const double* col = t_old + i;
            double t[ELL_W];
#pragma unroll
            for (int e = 0; e < ELL_W; ++e) t[e] = col[(size_t)s_crows[winner_block * ELL_W + e] * ld];  // padding: row 0, value 0
            double a0 = 0.0;
#pragma unroll
            for (int e = 0; e < ELL_W; ++e) a0 += t[e] * s_cvals[winner_block * ELL_W + e];
            al[r] = a0;
        }
        if (len > ELL_W || len < 0) {
            ca = lp.col_start[q] + (len < 0 ? 0 : ELL_W);
            cb_ = lp.col_start[q + 1];
        }
    } else {
        ca = lp.col_start[q];
        cb_ = lp.col_start[q + 1];
    }
    for (int c0_ = ca; c0_ < cb_; c0_ += K2_COL_CHUNK) {
        const int cnt = min(K2_COL_CHUNK, cb_ - c0_);
        __syncthreads();
        for (int e = tid; e < cnt; e += KF_THREADS) {
            s_rows[e] = lp.row_index[c0_ + e];
            s_vals[e] = lp.value[c0_ + e];
        }
        __syncthreads();
        // (the rows of a thread share the entry loop -- 4 R loads in flight -- with the per-row arithmetic of
        // ftran_ratio_fast_kernel, bit for bit: four chains over e mod 4, leftovers onto the first, (a0 + a1) + (a2 + a3))
        double acc[R][4];
        int row[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * KF_THREADS;
            row[r] = i < m ? i : 0;  // (rows past m read row 0 and are dropped below)
            acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0;
        }
        int e = 0;
        for (; e + 4 <= cnt; e += 4) {
            const size_t o0 = (size_t)s_rows[e] * ld, o1 = (size_t)s_rows[e + 1] * ld, o2 = (size_t)s_rows[e + 2] * ld, o3 = (size_t)s_rows[e + 3] * ld;
            const double v0 = s_vals[e], v1 = s_vals[e + 1], v2 = s_vals[e + 2], v3 = s_vals[e + 3];
            double x[R][4];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                x[r][0] = t_old[o0 + row[r]];
                x[r][1] = t_old[o1 + row[r]];
                x[r][2] = t_old[o2 + row[r]];
                x[r][3] = t_old[o3 + row[r]];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[r][0] += x[r][0] * v0;
                acc[r][1] += x[r][1] * v1;
                acc[r][2] += x[r][2] * v2;
                acc[r][3] += x[r][3] * v3;
            }
        }
        for (; e < cnt; ++e) {
            const size_t o_e = (size_t)s_rows[e] * ld;
            const double v = s_vals[e];
            double x[R];
#pragma unroll
            for (int r = 0; r < R; ++r) x[r] = t_old[o_e + row[r]];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][0] += x[r] * v;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (tid + r * KF_THREADS < m) al[r] += (acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3]);
    }
    FSTAMP(2);
    // ---- gamma_q and Harris pass 1, one combined block reduction (see ftran_ratio_fast_kernel) ---------------------
    const bool textbook = harris_delta < 0.0;
    const double harris_slack = textbook ? 0.0 : harris_delta;
    double sumsq = 0.0, theta = INFINITY;
    bool eligible[R];
    double room[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * KF_THREADS;
        const double a = al[r];
        sumsq += a * a;
        const bool allowed = i < m && !(skip_artificial_rows && bas[r] < lp.n_art);
        room[r] = fmax(xb[r], 0.0);
        eligible[r] = allowed && a > tol_pivot;
        if (eligible[r]) theta = fmin(theta, (room[r] + harris_slack) / fabs(a));
        s_alpha[i] = a;
    }
    sumsq = wave_sum(sumsq);
    theta = wave_min(theta);
    if (lane == LAST) {
        s_red[wave] = sumsq;
        s_red2[wave] = theta;
    }
    __syncthreads();
    if (wave == 0) {
        double t1 = lane < KF_NW ? s_red[lane] : 0.0;
        double t2 = lane < KF_NW ? s_red2[lane] : INFINITY;
        t1 = wave_sum(t1);
        t2 = wave_min(t2);
        if (lane == LAST) {
            s_red[KF_NW] = t1;
            s_red2[KF_NW] = t2;
        }
    }
    __syncthreads();
    const double gamma_q = 1.0 + s_red[KF_NW];  // pivot_rule.rs:258
    const double theta_max = s_red2[KF_NW];
    FSTAMP(3);
    // ---- Harris pass 2: largest eligible pivot, ties by the lowest leaving column (Bland), then the lowest row ----------
    double hkey = 0.0;
    unsigned long long hrank = RANK_NONE;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double mag = fabs(al[r]);
        const double key = textbook ? 1.0 : mag;
        if (eligible[r] && room[r] / mag <= theta_max) {
            const unsigned long long rk = ((unsigned long long)(unsigned)bas[r] << 32) | (unsigned)(tid + r * KF_THREADS);
            if (hrank == RANK_NONE || key > hkey || (key == hkey && rk < hrank)) {
                hkey = key;
                hrank = rk;
            }
        }
    }
    block_argbest(hkey, hrank, s_akey, s_arank);
    const int p = hrank == RANK_NONE ? -1 : (int)(hrank & 0xffffffffu);
    if (p < 0) {
        hand_on(ST_UNBOUNDED, q, false);
        return;
    }
    FSTAMP(4);
    // ---- the pivot row's scalars (its owner has them in registers) --------------------------------------------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (tid + r * KF_THREADS == p) {
            s_bcast[1] = al[r];
            s_bcast[2] = xb[r];
            s_ibcast[0] = bas[r];
        }
    }
    __syncthreads();
    const double alpha_pq = s_bcast[1];
    const int leaving = s_ibcast[0];
    const double xp = fmax(s_bcast[2], 0.0) / alpha_pq;
    // ---- workgroup 0: x_B update (carry/mod.rs:295-325), basis bookkeeping, control block -----------------------
    if (writer) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid + r * KF_THREADS;
            if (i < m) {
                lp.alpha[i] = al[r];
                out.xB[i] = (i == p) ? xp : xb[r] - al[r] * xp;
                out.basis[i] = (i == p) ? q : bas[r];
            }
        }
        if (tid == 0) {
            lp.pos[q] = p;
            lp.pos[leaving] = -1;
            Ctl c = *ctl;  // (read again by one thread: preloading the whole block in round trip 1 was measured slower)
            c.q = q;
            c.p = p;
            c.leaving = leaving;
            c.cbar_q = cbar_q;
            c.alpha_pq = alpha_pq;
            c.gamma_q = gamma_q;
            c.xp = xp;
            c.nz_count = 0;
            c.minus_obj = minus_obj - cbar_q * xp;
            c.iters = iters + 1;
            c.pending = 1;
            c.forced_q = -1;
            c.forced_p = -1;
            c.last_selected = q;
            c.t_buf = t_buf ^ 1;
            *out.ctl = c;
        }
    }
    FSTAMP(5);
    // ---- this wave's column: rho_p[j], w_j, -pi_j and the rank-one update, written to the other buffer ------------
    if (!own) return;
    const int u_p = p >> 6, lane_p = p & (WAVE - 1);
    double o_p = 0.0;
#pragma unroll
    for (int u = 0; u < KF_U; ++u) o_p = (u == u_p) ? o[u] : o_p;
    const double r_j = __shfl(o_p, lane_p) / alpha_pq;  // row p of the new inverse
    double* c_new = t_new + (size_t)j_own * ld;
    double w_j = 0.0;
#pragma unroll
    for (int u = 0; u < KF_U; ++u) {
        const int i = lane + u * WAVE;
        const double a = s_alpha[i];
        w_j += a * o[u];
        if (i < m) c_new[i] = (i == p) ? r_j : ((a != 0.0) ? o[u] - a * r_j : o[u]);
    }
    w_j = wave_sum(w_j);
    if (lane == LAST) {
        lp.w[j_own] = w_j;
        lp.rho[j_own] = r_j;
        lp.minus_pi[j_own] = pi_old - cbar_q * r_j;
    }
    FSTAMP(6);
}
#undef FSTAMP
// Batch start in the fused mode: the budget, and copy 0 of the state made equal to the canonical arrays.
__global__ void __launch_bounds__(256) begin_batch_kernel(DeviceLP lp, long long add) {
    const DeviceLP::State s0 = lp.state[0];
    for (int i = threadIdx.x; i < lp.m; i += 256) {
        s0.xB[i] = lp.xB[i];
        s0.basis[i] = lp.basis[i];
    }
    if (threadIdx.x == 0) {
        Ctl c = *lp.ctl;
        if (c.status == ST_BUDGET) c.status = ST_RUNNING;
        c.budget = c.iters + add;
        *lp.ctl = c;
        *s0.ctl = c;
    }
}
// Batch end: copy `parity` of the state back into the canonical arrays, the inverse back into the first buffer.
__global__ void __launch_bounds__(256) commit_kernel(DeviceLP lp, int parity) {
    const DeviceLP::State last = lp.state[parity];
    const int t_buf = last.ctl->t_buf;
    const int m = lp.m, ld = lp.ld;
    if (t_buf) {
        const int lane = threadIdx.x & (WAVE - 1);
        for (int j = blockIdx.x * 4 + threadIdx.x / WAVE; j < m; j += gridDim.x * 4) {
            const double* src = lp.Binv2 + (size_t)j * ld;
            double* dst = lp.Binv + (size_t)j * ld;
            for (int i = lane; i < m; i += WAVE) dst[i] = src[i];
        }
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < m; i += 256) {
            lp.xB[i] = last.xB[i];
            lp.basis[i] = last.basis[i];
        }
        if (threadIdx.x == 0) {
            Ctl c = *last.ctl;
            c.t_buf = 0;
            *lp.ctl = c;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K3: product-form update of the explicit inverse, fused with everything that is "per column j of Binv":
//     rho_p[j] (row p of the NEW inverse), -pi_j update, w_j = alpha_q' Binv_old(:, j).
//   replaces  BasisInverse::change_basis                       basis_inverse_rows.rs:36-70,123-137
//                 (role of the Forrest-Tomlin update            lower_upper/mod.rs:94-178)
//             BasisInverse::basis_inverse_row                  lower_upper/mod.rs:254-272
//             Carry::update_minus_pi_and_obj                   carry/mod.rs:338-349
//             BasisInverse::right_multiply_by_basis_inverse    lower_upper/mod.rs:212-237 (work vector, carry/mod.rs:575)
// One wave owns CPW contiguous columns; lanes sweep the ordered non-zero list of alpha (rows with alpha_i == 0 are
// never read or written).  Every touched element is read once and written once, coalesced; w_j is a wave-shuffle
// sum in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------------
constexpr int K3_THREADS = 256;
constexpr int K3_CPW = 2;  // columns per wave
constexpr int K3_SMALL_NZ = 32;  // alpha with at most this many non-zeros: eight lanes per column (large sparse LPs)
// EAGER: alpha and both columns are loaded in the same memory round trip, unconditionally (m small: the kernel is
// latency bound, the extra reads of rows with alpha_i == 0 hit L2); otherwise the column loads are predicated on
// alpha_i != 0 (m large: HBM bound, untouched rows cost no traffic).
template <bool EAGER>
__global__ void __launch_bounds__(K3_THREADS) update_kernel(DeviceLP lp) {
    Ctl* ctl = lp.ctl;
    const int m = lp.m, ld = lp.ld;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    int slot0 = (blockIdx.x * (K3_THREADS / WAVE) + wave) * K3_CPW;
    int j0 = slot0, j1 = slot0 + 1;
    bool two = slot0 + 1 < m;
    int row_lo = 0, row_hi = m;  // rows this wave sweeps
    bool split = false;
    if (!EAGER && lp.track_touched) {
        // Only the columns that are not unit vectors any more need the update (E e_j = e_j for j != p; K2 has put p on
        // the list).  For a unit column e_j:  w_j = alpha_j, rho_j = 0 and -pi_j does not change.
        if (ctl->status != ST_RUNNING || !ctl->pending) return;
        for (int j = blockIdx.x * K3_THREADS + threadIdx.x; j < m; j += gridDim.x * K3_THREADS)
            if (!lp.touched[j]) {
                const double a_j = lp.alpha[j];
                lp.w[j] = a_j;
                lp.rho[j] = 0.0;
                if (lp.prw) {
                    lp.prw[(size_t)4 * j + 1] = 0.0;
                    lp.prw[(size_t)4 * j + 2] = a_j;
                }
                if (lp.rho_nz) lp.rho_nz[j] = 0;
            }
        const int n_touched = ctl->touched_count;
        const int nz_small = ctl->nz_count;
        if (nz_small > 0 && nz_small <= K3_SMALL_NZ) {
            // Short alpha (network bases: a path): EIGHT lanes per column, eight columns per wave, every entry of the list in
            // flight at once.  With one wave per two columns 95 % of the lanes idle and the pass is a queue of waves that each
            // wait on three dependent round trips (config 5: 19.6 us on average for ~32 k touched columns).
            const int slot = (blockIdx.x * (K3_THREADS / WAVE) + wave) * 8 + (lane >> 3);
            if ((slot & ~7) >= n_touched) return;  // wave-uniform
            const bool active = slot < n_touched;
            const int sub = lane & 7;
            const int j = lp.tlist[active ? slot : 0];
            double* c = lp.Binv + (size_t)j * ld;
            const int p = ctl->p;
            const double alpha_pq = ctl->alpha_pq;
            const double cbar_q = ctl->cbar_q;
            const double pi_old = sub == 0 ? lp.minus_pi[j] : 0.0;
            int idx[K3_SMALL_NZ / 8];
            double a[K3_SMALL_NZ / 8], o[K3_SMALL_NZ / 8];
#pragma unroll
            for (int u = 0; u < K3_SMALL_NZ / 8; ++u) {
                const int e = u * 8 + sub;
                idx[u] = e < nz_small ? lp.nz_index[e] : -1;
                a[u] = e < nz_small ? lp.nz_alpha[e] : 0.0;
            }
            const double r = c[p] / alpha_pq;
#pragma unroll
            for (int u = 0; u < K3_SMALL_NZ / 8; ++u) o[u] = idx[u] >= 0 ? c[idx[u]] : 0.0;
            double w = 0.0;
#pragma unroll
            for (int u = 0; u < K3_SMALL_NZ / 8; ++u) {
                w += a[u] * o[u];
                if (active && idx[u] >= 0 && r != 0.0) c[idx[u]] = (idx[u] == p) ? r : o[u] - a[u] * r;
            }
            w += dpp_f64<DPP_QUAD_1032, 0xF>(0.0, w);
            w += dpp_f64<DPP_QUAD_2301, 0xF>(0.0, w);
            w += dpp_f64<DPP_ROW_HALF_MIRROR, 0xF>(0.0, w);
            if (active && sub == 0) {
                const double pi_new = pi_old - cbar_q * r;
                lp.w[j] = w;
                lp.rho[j] = r;
                lp.minus_pi[j] = pi_new;
                if (lp.prw) {
                    lp.prw[(size_t)4 * j] = pi_new;
                    lp.prw[(size_t)4 * j + 1] = r;
                    lp.prw[(size_t)4 * j + 2] = w;
                }
                mark_rho_row(lp, ctl->rho_buf, j, r);
            }
            return;
        }
        if (!(nz_small > 0 && nz_small * 6 < m) && (int)gridDim.x * K3_CPW >= m) {  // (the launch sized the grid for it)
            // Full sweep (alpha not sparse): ONE WORKGROUP per column pair, its four waves take a quarter of the rows each.
            // With a wave per pair, m = 2785 means 1400 waves on 1024 SIMDs, each walking 6 chunks of two dependent round
            // trips; split four ways the chip holds them all and a wave walks 1-2 chunks (GREENBEA: 49.1 -> 45.5 us per pivot).
            split = true;
            slot0 = blockIdx.x * K3_CPW;
            const int quarter = (((m + 3) / 4) + 7) & ~7;
            row_lo = wave * quarter;
            row_hi = min(m, row_lo + quarter);
        }
        if (slot0 >= n_touched) return;  // (workgroup-uniform when split)
        two = slot0 + 1 < n_touched;
        j0 = lp.tlist[slot0];
        j1 = two ? lp.tlist[slot0 + 1] : j0;
    } else if (slot0 >= m) {
        return;
    }
    double* c0 = lp.Binv + (size_t)j0 * ld;
    double* c1 = lp.Binv + (size_t)(two ? j1 : j0) * ld;
    // everything below is issued before the first use: one round trip
    const int status = ctl->status;
    const int pending = ctl->pending;
    const int p = ctl->p;
    const double alpha_pq = ctl->alpha_pq;
    const double cbar_q = ctl->cbar_q;
    double pi0 = 0.0, pi1 = 0.0;
    if (lane == LAST) {
        pi0 = lp.minus_pi[j0];
        pi1 = two ? lp.minus_pi[j1] : 0.0;
    }
    constexpr int U = EAGER ? 16 : 8;  // EAGER: 1024 rows in the first round trip
    double a[U], o0[U], o1[U];
    if (EAGER) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = lane + u * WAVE;
            a[u] = i < m ? lp.alpha[i] : 0.0;
            o0[u] = i < m ? c0[i] : 0.0;
            o1[u] = i < m ? c1[i] : 0.0;
        }
    }
    if (status != ST_RUNNING || !pending) return;
    const double r0 = c0[p] / alpha_pq;  // row p of the new inverse
    const double r1 = c1[p] / alpha_pq;
    if (split) __syncthreads();  // every wave of the pair has read row p before the wave that owns it overwrites it
    double w0 = 0.0, w1 = 0.0;
    const int nz = EAGER ? 0 : ctl->nz_count;
    if (!EAGER && nz > 0 && nz * 6 < m) {
        // sparse alpha: lanes walk K2's ordered list of touched rows instead of sweeping all m of them
        for (int base = 0; base < nz; base += U * WAVE) {
            int idx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + lane + u * WAVE;
                idx[u] = e < nz ? lp.nz_index[e] : -1;
                a[u] = e < nz ? lp.nz_alpha[e] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                o0[u] = idx[u] >= 0 ? c0[idx[u]] : 0.0;
                o1[u] = idx[u] >= 0 ? c1[idx[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                w0 += a[u] * o0[u];
                w1 += a[u] * o1[u];
                if (idx[u] >= 0) {  // r == 0: column j has no entry in row p and does not change (sparse inverses: most columns)
                    if (r0 != 0.0) c0[idx[u]] = (idx[u] == p) ? r0 : o0[u] - a[u] * r0;
                    if (two && r1 != 0.0) c1[idx[u]] = (idx[u] == p) ? r1 : o1[u] - a[u] * r1;
                }
            }
        }
    } else
    for (int base = row_lo; base < row_hi; base += U * WAVE) {
        if (!EAGER || base > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + lane + u * WAVE;
                a[u] = i < row_hi ? lp.alpha[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + lane + u * WAVE;
                const bool touch = i < row_hi && (EAGER || a[u] != 0.0 || i == p);
                o0[u] = touch ? c0[i] : 0.0;
                o1[u] = touch ? c1[i] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + lane + u * WAVE;
            w0 += a[u] * o0[u];
            w1 += a[u] * o1[u];
            if (i < row_hi && (a[u] != 0.0 || i == p)) {  // (r == 0: the column has no entry in row p and does not change)
                if (r0 != 0.0) c0[i] = (i == p) ? r0 : o0[u] - a[u] * r0;
                if (two && r1 != 0.0) c1[i] = (i == p) ? r1 : o1[u] - a[u] * r1;
            }
        }
    }
    w0 = wave_sum(w0);
    w1 = wave_sum(w1);
    if (split) {  // the four row quarters, added in a fixed order
        __shared__ double s_w[K3_THREADS / WAVE][2];
        if (lane == LAST) {
            s_w[wave][0] = w0;
            s_w[wave][1] = w1;
        }
        __syncthreads();
        if (wave != 0) return;
        w0 = (s_w[0][0] + s_w[1][0]) + (s_w[2][0] + s_w[3][0]);
        w1 = (s_w[0][1] + s_w[1][1]) + (s_w[2][1] + s_w[3][1]);
    }
    if (lane == LAST) {
        lp.w[j0] = w0;
        lp.rho[j0] = r0;
        lp.minus_pi[j0] = pi0 - cbar_q * r0;
        if (lp.prw) {  // packed copy for the width-2 pricing kernel
            lp.prw[(size_t)4 * j0] = pi0 - cbar_q * r0;
            lp.prw[(size_t)4 * j0 + 1] = r0;
            lp.prw[(size_t)4 * j0 + 2] = w0;
        }
        mark_rho_row(lp, ctl->rho_buf, j0, r0);
        if (two) {
            lp.w[j1] = w1;
            lp.rho[j1] = r1;
            lp.minus_pi[j1] = pi1 - cbar_q * r1;
            if (lp.prw) {
                lp.prw[(size_t)4 * j1] = pi1 - cbar_q * r1;
                lp.prw[(size_t)4 * j1 + 1] = r1;
                lp.prw[(size_t)4 * j1 + 2] = w1;
            }
            mark_rho_row(lp, ctl->rho_buf, j1, r1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Phase set-up
// ---------------------------------------------------------------------------------------------------
// -pi_j = -sum_i c_{basis[i]} Binv(i, j)   (Carry::create_minus_pi_from_artificial, carry/mod.rs:226-260: the reference
// forms all of B^-1 with m FTRANs; here B^-1 is resident).  Three kernels:
//   cb_kernel         c_B once (the costs of the basic columns, complemented ones negated) and the ORDERED list of its
//                     non-zeros; also -obj = -sum_i xB_i c_{basis[i]} (carry/mod.rs:270-283)
//   pi_kernel         one wave per column j of the inverse (contiguous), c_B streamed beside it
//   pi_sparse_kernel  when c_B has few non-zeros (max-flow: only the arcs leaving s cost anything): one THREAD per column
//                     gathers those rows -- k cache lines per column instead of the whole column (34 GB at m = 65 534)
constexpr int CB_THREADS = 1024;
__global__ void __launch_bounds__(CB_THREADS) cb_kernel(DeviceLP lp) {
    __shared__ double s_red[CB_THREADS / WAVE + 2];
    __shared__ int s_wave_count[CB_THREADS / WAVE];
    __shared__ int s_base;
    const int m = lp.m;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    double acc = 0.0;
    for (int i0 = 0; i0 < m; i0 += CB_THREADS) {
        const int i = i0 + threadIdx.x;
        double c = 0.0;
        if (i < m) {
            const int bj = lp.basis[i];
            c = (lp.flipped && lp.flipped[bj]) ? -lp.cost[bj] : lp.cost[bj];
            lp.cb[i] = c;
            acc += lp.xB[i] * c;
        }
        const unsigned long long mask = __ballot(c != 0.0);
        if (lane == 0) s_wave_count[wave] = __popcll(mask);
        __syncthreads();
        int before = s_base;
        for (int w2 = 0; w2 < wave; ++w2) before += s_wave_count[w2];
        if (c != 0.0) lp.cb_idx[before + __popcll(mask & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = s_base;
            for (int w2 = 0; w2 < CB_THREADS / WAVE; ++w2) total += s_wave_count[w2];
            s_base = total;
        }
        __syncthreads();
    }
    if (lp.flipped)  // constant of the complemented variables: sum ub_j c_j
        for (int j = threadIdx.x; j < lp.n; j += CB_THREADS)
            if (lp.flipped[j]) acc += lp.ub[j] * lp.cost[j];
    acc = block_reduce<0>(acc, s_red);
    if (threadIdx.x == 0) {
        lp.ctl->minus_obj = -acc;
        lp.cb_idx[m] = s_base;  // number of non-zeros of c_B
    }
}
__device__ __forceinline__ bool pi_takes_sparse_path(int k, int m) { return k * 16 <= m && k <= 4096; }
__global__ void __launch_bounds__(256) pi_kernel(DeviceLP lp) {
    const int m = lp.m, ld = lp.ld;
    if (pi_takes_sparse_path(lp.cb_idx[m], m)) return;
    const int lane = threadIdx.x & (WAVE - 1);
    const int j = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (j >= m) return;
    if (lp.track_touched && lp.eta_cap == 0 && !lp.touched[j]) {  // stored column j is still the unit vector e_j
        if (lane == LAST) {
            const double v = -lp.cb[j];
            lp.minus_pi[j] = v;
            if (lp.prw) lp.prw[(size_t)4 * j] = v;
        }
        return;
    }
    const double* col = lp.Binv + (size_t)j * ld;
    double acc = 0.0;
    int i = lane;
    for (; i + 3 * WAVE < m; i += 4 * WAVE) {  // four independent loads of the column in flight
        const double t0 = col[i], t1 = col[i + WAVE], t2 = col[i + 2 * WAVE], t3 = col[i + 3 * WAVE];
        const double c0 = lp.cb[i], c1 = lp.cb[i + WAVE], c2 = lp.cb[i + 2 * WAVE], c3 = lp.cb[i + 3 * WAVE];
        acc += c0 * t0;
        acc += c1 * t1;
        acc += c2 * t2;
        acc += c3 * t3;
    }
    for (; i < m; i += WAVE) acc += lp.cb[i] * col[i];
    acc = wave_sum(acc);
    if (lane == LAST) {
        lp.minus_pi[j] = -acc;
        if (lp.prw) lp.prw[(size_t)4 * j] = -acc;
    }
}
__global__ void __launch_bounds__(256) pi_sparse_kernel(DeviceLP lp) {
    const int m = lp.m, ld = lp.ld;
    const int k = lp.cb_idx[m];
    if (!pi_takes_sparse_path(k, m)) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const double* col = lp.Binv + (size_t)j * ld;
    double acc = 0.0;
    int t = 0;
    for (; t + 3 < k; t += 4) {
        const int i0 = lp.cb_idx[t], i1 = lp.cb_idx[t + 1], i2 = lp.cb_idx[t + 2], i3 = lp.cb_idx[t + 3];
        const double v0 = col[i0], v1 = col[i1], v2 = col[i2], v3 = col[i3];
        acc += lp.cb[i0] * v0;
        acc += lp.cb[i1] * v1;
        acc += lp.cb[i2] * v2;
        acc += lp.cb[i3] * v3;
    }
    for (; t < k; ++t) {
        const int i = lp.cb_idx[t];
        acc += lp.cb[i] * col[i];
    }
    lp.minus_pi[j] = -acc;
    if (lp.prw) lp.prw[(size_t)4 * j] = -acc;
}

// xB = Binv rhs  (Carry::from_basis, carry/mod.rs:452-463): xB_i = sum_j T[j*ld+i] rhs_j.  A workgroup owns 64 rows;
// its four waves split the columns (rhs_j == 0 skipped, wave-uniform), 8 independent loads in flight, LDS-combined
// in a fixed order.
__global__ void __launch_bounds__(256) xb_kernel(DeviceLP lp) {
    __shared__ double s_part[4][WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int i = blockIdx.x * WAVE + lane;
    const int m = lp.m;
    double acc = 0.0;
    for (int j0 = wave * 8; j0 < m; j0 += 32) {
        double r[8], t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = (j0 + u < m) ? lp.rhs[j0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (r[u] != 0.0 && i < m) ? lp.Binv[(size_t)(j0 + u) * lp.ld + i] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += r[u] * t[u];
    }
    s_part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && i < m) lp.xB[i] = (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
}

// gamma_j = 1 + ||Binv a_j||^2 for the non-basic, non-artificial columns (pivot_rule.rs:202-219, 299-305:
// the reference does n-m FTRANs).  One workgroup per column.
__global__ void __launch_bounds__(256) gamma_init_kernel(DeviceLP lp, int identity) {
    __shared__ double s_red[6];
    const int j = lp.n_art + blockIdx.x;
    if (j >= lp.n) return;
    if (lp.pos[j] >= 0) {
        if (threadIdx.x == 0) lp.gamma[j] = 1.0;
        return;
    }
    const int a = lp.col_start[j], b = lp.col_start[j + 1];
    double acc = 0.0;
    if (identity) {
        for (int e = a + threadIdx.x; e < b; e += blockDim.x) acc += lp.value[e] * lp.value[e];
    } else {
        for (int i = threadIdx.x; i < lp.m; i += blockDim.x) {
            double v = 0.0;
            for (int e = a; e < b; ++e) v += lp.Binv[(size_t)lp.row_index[e] * lp.ld + i] * lp.value[e];
            acc += v * v;
        }
    }
    acc = block_reduce<0>(acc, s_red);
    if (threadIdx.x == 0) lp.gamma[j] = 1.0 + acc;
}

__global__ void identity_kernel(double* X, int m, int ld) {  // grid-stride over the columns: gridDim.y is capped at 65535
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    for (int i = blockIdx.y; i < m; i += gridDim.y) X[(size_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------------
// Polish: Newton-Schulz  X <- X + (I - X B) X, i.e. in the stored transpose  T <- T + T S,  S = I - B' T.
// Plays the role of BasisInverse::should_refactor + invert (lower_upper/mod.rs:78-92,249-252; carry/mod.rs:584-591)
// for the explicit inverse: it removes the drift of the product-form updates with GEMM-shaped work only.
// ---------------------------------------------------------------------------------------------------
// S[k][:] = e_k - sum_{(r,v) in column basis[k] of A} v * T[r][:]   (CSC column of the basis, contiguous rows of T)
__global__ void __launch_bounds__(256) residual_kernel(DeviceLP lp, const double* T, double* S) {
    __shared__ double s_red[6];
    const int k = blockIdx.x;
    const int m = lp.m, ld = lp.ld;
    const int col = lp.basis[k];
    const int a = lp.col_start[col], b = lp.col_start[col + 1];
    const double sgn = (lp.flipped && lp.flipped[col]) ? -1.0 : 1.0;  // complemented basic column: B' holds -a_j
    double local_max = 0.0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        double acc = (i == k) ? 1.0 : 0.0;
        for (int e = a; e < b; ++e) acc -= sgn * lp.value[e] * T[(size_t)lp.row_index[e] * ld + i];
        if (S) S[(size_t)k * ld + i] = acc;  // S == nullptr: only the norm is wanted (polish of an inverse that may be exact)
        local_max = fmax(local_max, fabs(acc));
    }
    const double blk = -block_reduce<1>(-local_max, s_red);  // max via min of negatives
    if (threadIdx.x == 0) {
        // non-negative doubles order like their bit patterns
        atomicMax(reinterpret_cast<unsigned long long*>(&lp.ctl->residual),
                  (unsigned long long)__double_as_longlong(blk));
    }
}

// C = X + X R  (m x m, f64, row-major with leading dimension ld).  LDS-tiled 64x64 per workgroup, 4x4 per thread.
// MODE 1: C = I - X R (the residual S = I - B' T for DENSE bases, X = B' gathered dense).
constexpr int GT = 64, GK = 16;
template <int MODE>
__global__ void __launch_bounds__(256) gemm_polish_kernel(const double* __restrict__ X, const double* __restrict__ R,
                                                        double* __restrict__ C, int m, int ld, double* residual_max) {
    __shared__ double sA[GK][GT + 1];
    __shared__ double sB[GK][GT + 1];
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    const int row0 = blockIdx.y * GT, col0 = blockIdx.x * GT;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < m; k0 += GK) {
        for (int t = threadIdx.x; t < GT * GK; t += 256) {
            const int r = t / GK, k = t % GK;  // A tile: rows row0.., cols k0..
            const int gr = row0 + r, gk = k0 + k;
            sA[k][r] = (gr < m && gk < m) ? X[(size_t)gr * ld + gk] : 0.0;
            const int kk = t / GT, c = t % GT;  // B tile: rows k0.., cols col0..
            const int gk2 = k0 + kk, gc = col0 + c;
            sB[kk][c] = (gk2 < m && gc < m) ? R[(size_t)gk2 * ld + gc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = sA[k][ty * 4 + u];
                b[u] = sB[k][tx * 4 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
        }
        __syncthreads();
    }
    double local_max = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int gr = row0 + ty * 4 + u;
        if (gr >= m) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int gc = col0 + tx * 4 + v;
            if (gc < m) {
                if (MODE == 0) {
                    C[(size_t)gr * ld + gc] = X[(size_t)gr * ld + gc] + acc[u][v];
                } else {
                    const double r = (gr == gc ? 1.0 : 0.0) - acc[u][v];
                    C[(size_t)gr * ld + gc] = r;
                    local_max = fmax(local_max, fabs(r));
                }
            }
        }
    }
    if (MODE == 1) {
        __shared__ double s_red[6];
        const double blk = -block_reduce<1>(-local_max, s_red);
        if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(residual_max), (unsigned long long)__double_as_longlong(blk));
    }
}

// The same two GEMMs on the matrix cores: v_mfma_f64_16x16x4_f64 (one f64 of A and of B per lane: A[row l&15][k l>>4],
// B[k l>>4][col l&15]; D: col = l&15, row = (l>>4) + 4*reg).  Workgroup tile 64x64 (4 waves, 2x2), wave tile 32x32
// (2x2 MFMA tiles), K staged 16 at a time through LDS stored k-major with an 80-double row pitch, so that the four
// k-groups of a wave read disjoint bank halves (conflict-free ds_read_b64).
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));
constexpr int MT = 64, MK = 16, MPITCH = 80;
template <int MODE>
__global__ void __launch_bounds__(256) gemm_mfma_kernel(const double* __restrict__ X, const double* __restrict__ R,
                                                      double* __restrict__ C, int m, int ld, double* residual_max,
                                                      const int* __restrict__ row_list, int n_rows) {
    // row_list != nullptr: only the storage rows row_list[0 .. n_rows) are computed (the columns of the stored inverse that
    // are not unit vectors; the other rows of S are zero and the other rows of the polished inverse do not change)
    __shared__ double sA[MK][MPITCH];
    __shared__ double sB[MK][MPITCH];
    __shared__ double s_red[6];
    __shared__ int s_row[MT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * MT, col0 = blockIdx.x * MT;
    if (tid < MT) s_row[tid] = row_list ? (row0 + tid < n_rows ? row_list[row0 + tid] : -1) : (row0 + tid < m ? row0 + tid : -1);
    __syncthreads();
    mfma_f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (mfma_f64x4){0.0, 0.0, 0.0, 0.0};
    const int ar = tid >> 2, ak = (tid & 3) * 4;   // A tile: row ar, k ak..ak+3
    const int bk = tid >> 4, bc = (tid & 15) * 4;  // B tile: k bk, col bc..bc+3
    for (int k0 = 0; k0 < m; k0 += MK) {
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gr = s_row[ar], gk = k0 + ak + u;
            av[u] = (gr >= 0 && gk < m) ? X[(size_t)gr * ld + gk] : 0.0;
            const int gk2 = k0 + bk, gc = col0 + bc + u;
            bv[u] = (gk2 < m && gc < m) ? R[(size_t)gk2 * ld + gc] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sA[ak + u][ar] = av[u];
            sB[bk][bc + u] = bv[u];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < MK; kk += 4) {
            double a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = sA[kk + (lane >> 4)][wr * 32 + t * 16 + (lane & 15)];
                b[t] = sB[kk + (lane >> 4)][wc * 32 + t * 16 + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    double local_max = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gr = s_row[wr * 32 + i * 16 + (lane >> 4) + 4 * reg];
                const int gc = col0 + wc * 32 + j * 16 + (lane & 15);
                if (gr >= 0 && gc < m) {
                    if (MODE == 0) {
                        C[(size_t)gr * ld + gc] = X[(size_t)gr * ld + gc] + acc[i][j][reg];
                    } else {
                        const double r = (gr == gc ? 1.0 : 0.0) - acc[i][j][reg];
                        C[(size_t)gr * ld + gc] = r;
                        local_max = fmax(local_max, fabs(r));
                    }
                }
            }
    if (MODE == 1) {
        const double blk = -block_reduce<1>(-local_max, s_red);
        if (tid == 0) atomicMax(reinterpret_cast<unsigned long long*>(residual_max), (unsigned long long)__double_as_longlong(blk));
    }
}

// Bd[k][r] = B[r][k]: row k of Bd is column basis[k] of A, dense (zero-filled).
__global__ void gather_basis_kernel(DeviceLP lp, double* Bd) {
    const int k = blockIdx.x;
    const int col = lp.basis[k];
    for (int r = threadIdx.x; r < lp.m; r += blockDim.x) Bd[(size_t)k * lp.ld + r] = 0.0;
    __syncthreads();
    const double sgn = (lp.flipped && lp.flipped[col]) ? -1.0 : 1.0;
    for (int e = lp.col_start[col] + threadIdx.x; e < lp.col_start[col + 1]; e += blockDim.x)
        Bd[(size_t)k * lp.ld + lp.row_index[e]] = sgn * lp.value[e];
}

// T0 = s * B  (T0[r][k] = s B[r][k]; X0 = T0' = s B'): start of a from-scratch Newton-Schulz inversion.
__global__ void scaled_basis_kernel(DeviceLP lp, double* T, double scale) {
    const int k = blockIdx.x;
    const int col = lp.basis[k];
    for (int r = threadIdx.x; r < lp.m; r += blockDim.x) T[(size_t)r * lp.ld + k] = 0.0;
    __syncthreads();
    const double sgn = (lp.flipped && lp.flipped[col]) ? -1.0 : 1.0;
    for (int e = lp.col_start[col] + threadIdx.x; e < lp.col_start[col + 1]; e += blockDim.x)
        T[(size_t)lp.row_index[e] * lp.ld + k] = scale * sgn * lp.value[e];
}

// ---------------------------------------------------------------------------------------------------
// Zero-level pivots (phase_one.rs:232-278): first non-basic, non-artificial column with a non-zero entry in
// tableau row r (generate_element, lower_upper/mod.rs:239-247, without a full FTRAN per candidate).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) row_scan_kernel(DeviceLP lp, int r, double tol) {
    for (int j = lp.n_art + blockIdx.x * blockDim.x + threadIdx.x; j < lp.n; j += gridDim.x * blockDim.x) {
        if (lp.pos[j] >= 0) continue;
        double acc = 0.0;
        for (int e = lp.col_start[j]; e < lp.col_start[j + 1]; ++e)
            acc += lp.value[e] * lp.Binv[(size_t)lp.row_index[e] * lp.ld + r];
        if (fabs(acc) > tol) atomicMin(&lp.ctl->scan_column, j);
    }
}

// ---------------------------------------------------------------------------------------------------
// Fine-grained trait ops (tests, Rust shim)
// ---------------------------------------------------------------------------------------------------
// out = Binv v (FTRAN of a sparse column given as (rows, values) in device scratch)
__global__ void ftran_vec_kernel(DeviceLP lp, const int* rows, const double* vals, int nnz, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lp.m) return;
    double acc = 0.0;
    for (int e = 0; e < nnz; ++e) acc += lp.Binv[(size_t)rows[e] * lp.ld + i] * vals[e];
    out[i] = acc;
}
// out = v' Binv (BTRAN): out_j = sum_e v_e Binv(r_e, j)
__global__ void btran_vec_kernel(DeviceLP lp, const int* rows, const double* vals, int nnz, double* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lp.m) return;
    double acc = 0.0;
    for (int e = 0; e < nnz; ++e) acc += vals[e] * lp.Binv[(size_t)j * lp.ld + rows[e]];
    out[j] = acc;
}
// cbar_j for every column (Tableau::relative_cost)
__global__ void relative_cost_kernel(DeviceLP lp, double* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lp.n) return;
    double acc = lp.cost[j];
    for (int e = lp.col_start[j]; e < lp.col_start[j + 1]; ++e) acc += lp.value[e] * lp.minus_pi[lp.row_index[e]];
    out[j] = acc;
}

// ---------------------------------------------------------------------------------------------------
// launch helpers used by solver.hip
// ---------------------------------------------------------------------------------------------------
// Measurement hook (bench.py roofline leg): when armed, the NEXT launch of the named hot-loop kernel is bracketed by a
// start/stop event pair through hipExtLaunchKernelGGL, i.e. its own execution time inside the real pivot sequence.
struct LaunchTimer {
    int which = -1;  // 0 price (sparse or dense), 1 fused ftran/ratio, 2 update
    hipEvent_t start = nullptr, stop = nullptr;
};
static thread_local LaunchTimer g_timer;
void arm_launch_timer(int which, hipEvent_t start, hipEvent_t stop) {
    g_timer.which = which;
    g_timer.start = start;
    g_timer.stop = stop;
}
// (kernels of another file: hand the armed event pair over, if it is meant for `which`)
void take_launch_timer(int which, hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (g_timer.which == which) {
        g_timer.which = -1;
        *start = g_timer.start;
        *stop = g_timer.stop;
    }
}
#define RELP_LAUNCH(WHICH, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                            \
    do {                                                                                                     \
        if (g_timer.which == (WHICH)) {                                                                      \
            g_timer.which = -1;                                                                              \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, (std::uint32_t)(LDS), STREAM, g_timer.start, g_timer.stop, 0, __VA_ARGS__); \
        } else {                                                                                             \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                               \
        }                                                                                                    \
    } while (0)
constexpr int PRICE_LPC = 8;  // lanes per sparse column in the pricing kernel
int price_columns_per_block(int ell_w, bool generated) { return generated ? 256 * PRICE_UNIT_ARCS : 256 / ell_w; }

template <int RULE>
static void launch_price_rule(const DeviceLP& d, int blocks, size_t lds, bool use_lds, int skip_weights, double tol,
                              int first, int last, int cand_offset, hipStream_t s) {
    const bool timed_elsewhere = d.n_dense > 0;  // with a dense block the dense kernel is the one that is timed
    if (d.ell_w == 2 && d.cost8 && d.price_unit_pairs)  // (the round-2 form, two lanes per arc: A/B)
        RELP_LAUNCH(timed_elsewhere ? -2 : 0, (price_kernel<RULE, false, 2, true>), dim3(blocks), dim3(256), 0, s, d, skip_weights, tol, first, last, cand_offset);
    else if (d.ell_w == 2 && d.cost8)  // incidence columns generated from the arcs' endpoints (8 B per arc)
        RELP_LAUNCH(timed_elsewhere ? -2 : 0, (price_unit_kernel<RULE>), dim3(blocks), dim3(256), (size_t)d.rho_words * sizeof(unsigned), s, d, skip_weights, tol, first, last, cand_offset);
    else if (d.ell_w == 2)  // graph LPs: two entries per column, 128 columns per workgroup pass (large m: vectors gathered from L2)
        RELP_LAUNCH(timed_elsewhere ? -2 : 0, (price_kernel<RULE, false, 2>), dim3(blocks), dim3(256), 0, s, d, skip_weights, tol, first, last, cand_offset);
    else if (use_lds)
        RELP_LAUNCH(timed_elsewhere ? -2 : 0, (price_kernel<RULE, true, PRICE_LPC>), dim3(blocks), dim3(256), lds, s, d, skip_weights, tol, first, last, cand_offset);
    else
        RELP_LAUNCH(timed_elsewhere ? -2 : 0, (price_kernel<RULE, false, PRICE_LPC>), dim3(blocks), dim3(256), 0, s, d, skip_weights, tol, first, last, cand_offset);
}

// sparse (CSC) pricing over the device columns [first, last)
void launch_price(const DeviceLP& d, int rule, int blocks, size_t lds, bool use_lds, int skip_weights, double tol,
                  int first, int last, int cand_offset, hipStream_t s) {
    switch (rule) {
        case RELP_PIVOT_DANTZIG: launch_price_rule<RELP_PIVOT_DANTZIG>(d, blocks, lds, use_lds, skip_weights, tol, first, last, cand_offset, s); break;
        case RELP_PIVOT_FIRST_PROFITABLE: launch_price_rule<RELP_PIVOT_FIRST_PROFITABLE>(d, blocks, lds, use_lds, skip_weights, tol, first, last, cand_offset, s); break;
        case RELP_PIVOT_FIRST_PROFITABLE_MEMORY: launch_price_rule<RELP_PIVOT_FIRST_PROFITABLE_MEMORY>(d, blocks, lds, use_lds, skip_weights, tol, first, last, cand_offset, s); break;
        default: launch_price_rule<RELP_PIVOT_STEEPEST_EDGE>(d, blocks, lds, use_lds, skip_weights, tol, first, last, cand_offset, s); break;
    }
}

void launch_price_dense(const DeviceLP& d, int blocks, int skip_weights, double tol, int cand_offset, hipStream_t s) {
    const size_t lds = (size_t)3 * d.dense_ld * sizeof(double);
    if (d.dense_lane) {
        if (d.dense_val) RELP_LAUNCH(0, price_dense_lane_kernel<8>, dim3(blocks), dim3(dense_lane_threads(d.m)), 0, s, d, skip_weights, tol, cand_offset);
        else if (d.dense_val32) RELP_LAUNCH(0, price_dense_lane_kernel<4>, dim3(blocks), dim3(dense_lane_threads(d.m)), 0, s, d, skip_weights, tol, cand_offset);
        else RELP_LAUNCH(0, price_dense_lane_kernel<1>, dim3(blocks), dim3(dense_lane_threads(d.m)), 0, s, d, skip_weights, tol, cand_offset);
    } else if (d.dense_val8) RELP_LAUNCH(0, (price_dense_kernel<false, true>), dim3(blocks), dim3(K1D_THREADS), lds, s, d, skip_weights, tol, cand_offset);
    else if (d.dense_val32) RELP_LAUNCH(0, price_dense_kernel<true>, dim3(blocks), dim3(K1D_THREADS), lds, s, d, skip_weights, tol, cand_offset);
    else RELP_LAUNCH(0, price_dense_kernel<false>, dim3(blocks), dim3(K1D_THREADS), lds, s, d, skip_weights, tol, cand_offset);
}
// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel on the CURRENT device: setting it to the current LP's size would
// let the last loaded handle decide for every other one.  It is set once per device (PerDeviceOnce, solver.hpp), to what the CU has
// (160 KB minus the kernel's static LDS).
static void allow_full_lds(const void* kernel) {
    hipFuncAttributes attr{};
    size_t fixed = 0;
    if (hipFuncGetAttributes(&attr, kernel) == hipSuccess) fixed = attr.sharedSizeBytes;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - fixed)) != hipSuccess) (void)hipGetLastError();
}
void configure_dense_lds(size_t) {
    static PerDeviceOnce once;
    once.run([] {
        allow_full_lds(reinterpret_cast<const void*>(&price_dense_kernel<false>));
        allow_full_lds(reinterpret_cast<const void*>(&price_dense_kernel<true>));
        allow_full_lds(reinterpret_cast<const void*>(&price_dense_kernel<false, true>));
    });
}
void launch_ftran_partial(const DeviceLP& d, int n_slices, int n_price_blocks, int rule, hipStream_t s) {
    hipLaunchKernelGGL(ftran_partial_kernel, dim3((d.m + 255) / 256, n_slices), dim3(256), 0, s, d, n_slices, n_price_blocks, rule);
}

void configure_lds(size_t) {
    // opt in to > 64 KB of dynamic LDS (160 KB per CU on gfx950), once per device
    static PerDeviceOnce once;
    once.run([] {
        allow_full_lds(reinterpret_cast<const void*>(&price_kernel<RELP_PIVOT_STEEPEST_EDGE, true, PRICE_LPC>));
        allow_full_lds(reinterpret_cast<const void*>(&price_kernel<RELP_PIVOT_DANTZIG, true, PRICE_LPC>));
        allow_full_lds(reinterpret_cast<const void*>(&price_kernel<RELP_PIVOT_FIRST_PROFITABLE, true, PRICE_LPC>));
        allow_full_lds(reinterpret_cast<const void*>(&price_kernel<RELP_PIVOT_FIRST_PROFITABLE_MEMORY, true, PRICE_LPC>));
    });
}

template <int RULE>
static void launch_ftran_ratio_rule(const DeviceLP& d, int n_price_blocks, double tol_pivot, double harris_delta,
                                    int skip_artificial_rows, int mode, int n_alpha_slices, hipStream_t s) {
    const bool fits = n_price_blocks <= K2F_MAX_BLOCKS;
    if (fits && d.m <= 2 * K2F_THREADS)
        RELP_LAUNCH(1, (ftran_ratio_fast_kernel<RULE, 2>), dim3(1), dim3(K2F_THREADS), 0, s, d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode, n_alpha_slices);
    else if (fits && d.m <= 4 * K2F_THREADS)
        RELP_LAUNCH(1, (ftran_ratio_fast_kernel<RULE, 4>), dim3(1), dim3(K2F_THREADS), 0, s, d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode, n_alpha_slices);
    else if (fits && d.m <= 8 * K2F_THREADS)
        RELP_LAUNCH(1, (ftran_ratio_fast_kernel<RULE, 8>), dim3(1), dim3(K2F_THREADS), 0, s, d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode, n_alpha_slices);
    else if (fits && d.m <= 16 * K2F_THREADS)
        RELP_LAUNCH(1, (ftran_ratio_fast_kernel<RULE, 16>), dim3(1), dim3(K2F_THREADS), 0, s, d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode, n_alpha_slices);
    else if (mode == 0 && d.k2_partd != nullptr) {
        const int blocks = (d.m + K2L_THREADS - 1) / K2L_THREADS;
        RELP_LAUNCH(1, (k2l_ftran_kernel<RULE>), dim3(blocks), dim3(K2L_THREADS), 0, s, d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows);
        hipLaunchKernelGGL(k2l_harris_kernel, dim3(blocks), dim3(K2L_THREADS), 0, s, d, blocks, tol_pivot, skip_artificial_rows);
        hipLaunchKernelGGL(k2l_apply_kernel, dim3(blocks), dim3(K2L_THREADS), 0, s, d, blocks);
    } else
        RELP_LAUNCH(1, (ftran_ratio_kernel<RULE>), dim3(1), dim3(K2_THREADS), 0, s, d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode);
}

// n_alpha_slices > 0 requires the register-resident kernel (m <= 8192 and <= 2048 pricing workgroups)
bool fast_k2_available(const DeviceLP& d, int n_price_blocks) { return n_price_blocks <= K2F_MAX_BLOCKS && d.m <= 16 * K2F_THREADS; }

void launch_ftran_ratio(const DeviceLP& d, int rule, int n_price_blocks, double tol_pivot, double harris_delta,
                        int skip_artificial_rows, int mode, int n_alpha_slices, hipStream_t s) {
    if (rule == RELP_PIVOT_STEEPEST_EDGE)
        launch_ftran_ratio_rule<RELP_PIVOT_STEEPEST_EDGE>(d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode, n_alpha_slices, s);
    else
        launch_ftran_ratio_rule<RELP_PIVOT_DANTZIG>(d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode, n_alpha_slices, s);
}

bool fused_pivot_available(const DeviceLP& d, int n_price_blocks) { return d.m <= KF_MAX_M && n_price_blocks <= K2F_MAX_BLOCKS; }
void launch_pivot_fused(const DeviceLP& d, int rule, int parity, int n_price_blocks, double tol_pivot, double harris_delta,
                        int skip_artificial_rows, hipStream_t s) {
    const dim3 grid((d.m + KF_NW - 1) / KF_NW);
    const bool small = d.m <= 2 * KF_THREADS;
    if (rule == RELP_PIVOT_STEEPEST_EDGE) {
        if (small) RELP_LAUNCH(1, (pivot_fused_kernel<RELP_PIVOT_STEEPEST_EDGE, 2>), grid, dim3(KF_THREADS), 0, s, d, d.state[parity], d.state[parity ^ 1], n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows);
        else RELP_LAUNCH(1, (pivot_fused_kernel<RELP_PIVOT_STEEPEST_EDGE, 4>), grid, dim3(KF_THREADS), 0, s, d, d.state[parity], d.state[parity ^ 1], n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows);
    } else {
        if (small) RELP_LAUNCH(1, (pivot_fused_kernel<RELP_PIVOT_DANTZIG, 2>), grid, dim3(KF_THREADS), 0, s, d, d.state[parity], d.state[parity ^ 1], n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows);
        else RELP_LAUNCH(1, (pivot_fused_kernel<RELP_PIVOT_DANTZIG, 4>), grid, dim3(KF_THREADS), 0, s, d, d.state[parity], d.state[parity ^ 1], n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows);
    }
}
void launch_begin_batch(const DeviceLP& d, long long add, hipStream_t s) {
    hipLaunchKernelGGL(begin_batch_kernel, dim3(1), dim3(256), 0, s, d, add);
}
void launch_commit(const DeviceLP& d, int parity, hipStream_t s) {
    hipLaunchKernelGGL(commit_kernel, dim3(std::min(256, (d.m + 3) / 4)), dim3(256), 0, s, d, parity);
}

void launch_update(const DeviceLP& d, hipStream_t s) {
    const int cols_per_block = (K3_THREADS / WAVE) * K3_CPW;
    const dim3 grid((d.m + cols_per_block - 1) / cols_per_block);
    // m > 2048: a full sweep runs one workgroup per column pair (the list-driven modes use the first quarter of the grid)
    const dim3 grid_split((d.m + K3_CPW - 1) / K3_CPW);
    if (d.m <= 2048) RELP_LAUNCH(2, (update_kernel<true>), grid, dim3(K3_THREADS), 0, s, d);
    else RELP_LAUNCH(2, (update_kernel<false>), (d.track_touched && d.m <= 16384) ? grid_split : grid, dim3(K3_THREADS), 0, s, d);  // larger m: the list-driven modes dominate (graph LPs) and 4x the workgroups only cost launch time
}

void launch_budget(const DeviceLP& d, long long add, hipStream_t s) {
    hipLaunchKernelGGL(budget_kernel, dim3(1), dim3(1), 0, s, d.ctl, add);
}

void launch_pi(const DeviceLP& d, hipStream_t s) {
    hipLaunchKernelGGL(cb_kernel, dim3(1), dim3(CB_THREADS), 0, s, d);
    hipLaunchKernelGGL(pi_kernel, dim3((d.m + 3) / 4), dim3(256), 0, s, d);       // (one of the two returns at once:
    hipLaunchKernelGGL(pi_sparse_kernel, dim3((d.m + 255) / 256), dim3(256), 0, s, d);  //  decided on the device by nnz(c_B))
}
void launch_xb(const DeviceLP& d, hipStream_t s) {
    hipLaunchKernelGGL(xb_kernel, dim3((d.m + WAVE - 1) / WAVE), dim3(256), 0, s, d);
}
void launch_gamma_init(const DeviceLP& d, int identity, hipStream_t s) {
    hipLaunchKernelGGL(gamma_init_kernel, dim3(d.n - d.n_art), dim3(256), 0, s, d, identity);
}
void launch_identity(double* X, int m, int ld, hipStream_t s) {
    hipLaunchKernelGGL(identity_kernel, dim3((m + 255) / 256, std::min(m, 65535)), dim3(256), 0, s, X, m, ld);
}
__global__ void __launch_bounds__(256) scatter_kernel(double* X, const long long* index, const double* value, long long count) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += stride) X[index[e]] = value[e];
}
// sparse entries into a resident matrix (the crash basis' inverse into the identity)
void launch_scatter(double* X, const long long* index, const double* value, long long count, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)std::min<long long>(4096, (count + 255) / 256)), dim3(256), 0, s, X, index, value, count);
}
void launch_residual(const DeviceLP& d, const double* T, double* S, hipStream_t s) {
    hipLaunchKernelGGL(residual_kernel, dim3(d.m), dim3(256), 0, s, d, T, S);
}
bool gemm_row_lists_supported();
static bool use_mfma_gemm() { return !thread_tuning().has(RELP_SW_GEMM_VECTOR); }  // (the switch selects the plain-FMA kernel: A/B measurements)
__global__ void __launch_bounds__(256) copy_rows_kernel(const double* src, double* dst, int m, int ld, const int* row_list, int n_rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    for (int idx = blockIdx.y; idx < n_rows; idx += gridDim.y) {  // (gridDim.y is capped at 65535)
        const size_t row = (size_t)row_list[idx] * ld;
        dst[row + i] = src[row + i];
    }
}
bool gemm_row_lists_supported() { return use_mfma_gemm(); }  // the plain-FMA fallback kernels compute every row
void launch_copy_rows(const double* src, double* dst, int m, int ld, const int* row_list, int n_rows, hipStream_t s) {
    if (n_rows > 0) hipLaunchKernelGGL(copy_rows_kernel, dim3((m + 255) / 256, std::min(n_rows, 65535)), dim3(256), 0, s, src, dst, m, ld, row_list, n_rows);
}
// row_list (device, n_rows entries) restricts the computed storage rows; nullptr = all m
void launch_gemm_polish(const double* X, const double* R, double* C, int m, int ld, const int* row_list, int n_rows, hipStream_t s) {
    const int rows = row_list ? n_rows : m;
    if (rows <= 0) return;
    dim3 grid((m + GT - 1) / GT, (rows + GT - 1) / GT);
    if (use_mfma_gemm()) hipLaunchKernelGGL((gemm_mfma_kernel<0>), grid, dim3(256), 0, s, X, R, C, m, ld, (double*)nullptr, row_list, n_rows);
    else hipLaunchKernelGGL((gemm_polish_kernel<0>), grid, dim3(256), 0, s, X, R, C, m, ld, (double*)nullptr);
}
// S = I - B' T for a dense basis: gather B' into `Bd`, then one GEMM (also records max |S|)
void launch_residual_dense(const DeviceLP& d, double* Bd, const double* T, double* S, const int* row_list, int n_rows, hipStream_t s) {
    hipLaunchKernelGGL(gather_basis_kernel, dim3(d.m), dim3(256), 0, s, d, Bd);
    const int rows = row_list ? n_rows : d.m;
    if (rows <= 0) return;
    dim3 grid((d.m + GT - 1) / GT, (rows + GT - 1) / GT);
    if (use_mfma_gemm()) hipLaunchKernelGGL((gemm_mfma_kernel<1>), grid, dim3(256), 0, s, Bd, T, S, d.m, d.ld, &d.ctl->residual, row_list, n_rows);
    else hipLaunchKernelGGL((gemm_polish_kernel<1>), grid, dim3(256), 0, s, Bd, T, S, d.m, d.ld, &d.ctl->residual);
}
void launch_alpha_reduce(const DeviceLP& d, int n_slices, hipStream_t s) {
    hipLaunchKernelGGL(alpha_reduce_kernel, dim3((d.m + AR_ROWS - 1) / AR_ROWS), dim3(AR_ROWS * AR_GROUPS), 0, s, d, n_slices);
}
int eta_max() { return ETA_MAX; }
void configure_btran_lds(size_t) {
    static PerDeviceOnce once;
    once.run([] { allow_full_lds(reinterpret_cast<const void*>(&btran_pass_kernel)); });
}
// deferred product form: fold the new eta into the kept columns, then one read-only pass for rho_p, w and -pi
int btran_pass_blocks() { return 256; }
void launch_eta_update(const DeviceLP& d, double tol_dual, hipStream_t s) {
    const size_t lds = (size_t)2 * ((d.m + 1) & ~1) * sizeof(double);
    RELP_LAUNCH(2, btran_pass_kernel, dim3(btran_pass_blocks()), dim3(BT_THREADS), lds, s, d, tol_dual);
}
void launch_mark_all_touched(const DeviceLP& d, hipStream_t s) {
    hipLaunchKernelGGL(mark_all_touched_kernel, dim3((d.m + 255) / 256), dim3(256), 0, s, d);
}
void launch_eta_consolidate(const DeviceLP& d, hipStream_t s) {
    hipLaunchKernelGGL(eta_mark_kernel, dim3(1), dim3(64), 0, s, d);
    hipLaunchKernelGGL(eta_gather_kernel, dim3((d.m + 255) / 256, d.eta_cap), dim3(256), 0, s, d);
    hipLaunchKernelGGL(eta_apply_kernel, dim3((d.m + EA_TI - 1) / EA_TI, (d.m + EA_TJ - 1) / EA_TJ), dim3(256), 0, s, d);
    hipLaunchKernelGGL(eta_reset_kernel, dim3(1), dim3(ETA_MAX), 0, s, d);
}
void launch_scaled_basis(const DeviceLP& d, double* T, double scale, hipStream_t s) {
    hipLaunchKernelGGL(scaled_basis_kernel, dim3(d.m), dim3(64), 0, s, d, T, scale);
}
void launch_row_scan(const DeviceLP& d, int r, double tol, hipStream_t s) {
    int blocks = (d.n - d.n_art + 255) / 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(row_scan_kernel, dim3(blocks), dim3(256), 0, s, d, r, tol);
}
void launch_ftran_vec(const DeviceLP& d, const int* rows, const double* vals, int nnz, double* out, hipStream_t s) {
    hipLaunchKernelGGL(ftran_vec_kernel, dim3((d.m + 255) / 256), dim3(256), 0, s, d, rows, vals, nnz, out);
}
void launch_btran_vec(const DeviceLP& d, const int* rows, const double* vals, int nnz, double* out, hipStream_t s) {
    hipLaunchKernelGGL(btran_vec_kernel, dim3((d.m + 255) / 256), dim3(256), 0, s, d, rows, vals, nnz, out);
}
void launch_relative_cost(const DeviceLP& d, double* out, hipStream_t s) {
    hipLaunchKernelGGL(relative_cost_kernel, dim3((d.n + 255) / 256), dim3(256), 0, s, d, out);
}

}  // namespace relp
