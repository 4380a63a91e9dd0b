// HIP kernels of the revised-simplex inner loop for gfx950 (wave64, 256 CUs in 8 XCDs, 160 KB LDS per CU).
//
// One pivot = price -> ftran+ratio(+O(m) updates) -> inverse update (+ partial w) -> w reduce.
// All state stays in HBM / L2; every kernel starts by reading the control word, so that a launch sequence
// enqueued past the end of a phase degenerates into no-ops (no host round trip per pivot).
//
// Reference functions each kernel replaces are cited at the kernel; paths relative to
// /root/reference/src/algorithm/two_phase/.
#include <hip/hip_runtime.h>

#include <cmath>

#include "solver.hpp"

namespace relp {

constexpr int WAVE = 64;

// ---------------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------------
struct Cand {
    double key;
    int idx;  // -1: empty
    int aux;
};

enum : int { TIE_LARGER_IDX = 0, TIE_SMALLER_IDX = 1, TIE_SMALLER_AUX = 2 };

template <int TIE>
__device__ __forceinline__ Cand better(const Cand& a, const Cand& b) {
    if (a.idx < 0) return b;
    if (b.idx < 0) return a;
    if (a.key > b.key) return a;
    if (a.key < b.key) return b;
    if (TIE == TIE_LARGER_IDX) return a.idx > b.idx ? a : b;
    if (TIE == TIE_SMALLER_IDX) return a.idx < b.idx ? a : b;
    return a.aux < b.aux ? a : (a.aux > b.aux ? b : (a.idx < b.idx ? a : b));
}

template <int TIE>
__device__ __forceinline__ Cand wave_best(Cand v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        Cand o;
        o.key = __shfl_down(v.key, off);
        o.idx = __shfl_down(v.idx, off);
        o.aux = __shfl_down(v.aux, off);
        v = better<TIE>(v, o);
    }
    return v;
}

// Block-wide argmax; result valid in every thread.  `s` holds at least blockDim.x/64 + 1 entries.
template <int TIE>
__device__ __forceinline__ Cand block_best(Cand v, Cand* s) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int nwaves = (blockDim.x + WAVE - 1) / WAVE;
    v = wave_best<TIE>(v);
    __syncthreads();
    if (lane == 0) s[wave] = v;
    __syncthreads();
    if (wave == 0) {
        Cand t;
        t.key = 0.0;
        t.idx = -1;
        t.aux = 0;
        if (lane < nwaves) t = s[lane];
        t = wave_best<TIE>(t);
        if (lane == 0) s[nwaves] = t;
    }
    __syncthreads();
    return s[nwaves];
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off));
    return v;
}

// op: 0 sum, 1 min.  Deterministic order (fixed tree).  `s` holds blockDim.x/64 + 1 doubles.
template <int OP>
__device__ __forceinline__ double block_reduce(double v, double* s) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    const int nwaves = (blockDim.x + WAVE - 1) / WAVE;
    v = OP == 0 ? wave_sum(v) : wave_min(v);
    __syncthreads();
    if (lane == 0) s[wave] = v;
    __syncthreads();
    if (wave == 0) {
        double t = OP == 0 ? 0.0 : INFINITY;
        if (lane < nwaves) t = s[lane];
        t = OP == 0 ? wave_sum(t) : wave_min(t);
        if (lane == 0) s[nwaves] = t;
    }
    __syncthreads();
    return s[nwaves];
}

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
// One thread per column (sparse CSC columns of ~5-10 entries); -pi, rho_p and w are staged in LDS once per
// workgroup, so HBM/L2 traffic is the column data itself: nnz*(8+4) + 3*8 bytes per column.
// The reference makes TWO passes over A per pivot (pricing, weight update) and clones every column twice.
// ---------------------------------------------------------------------------------------------------
template <int RULE, bool USE_LDS>
__global__ void __launch_bounds__(256) price_kernel(DeviceLP lp, int skip_weights, double tol_dual, int n_chunks) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ Cand s_cand[8];
    Ctl* ctl = lp.ctl;
    if (ctl->status != ST_RUNNING) return;
    const int m = lp.m;
    const int pending = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? (ctl->pending && !skip_weights) : 0;
    const double* v_pi = lp.minus_pi;
    const double* v_rho = lp.rho;
    const double* v_w = lp.w;
    if (USE_LDS) {
        double* s_pi = smem;
        double* s_rho = smem + m;
        double* s_w = smem + 2 * m;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            s_pi[i] = lp.minus_pi[i];
            if (pending) {
                s_rho[i] = lp.rho[i];
                s_w[i] = lp.w[i];
            }
        }
        __syncthreads();
        v_pi = s_pi;
        v_rho = s_rho;
        v_w = s_w;
    }
    const double gamma_q = ctl->gamma_q;
    const double alpha_pq = ctl->alpha_pq;
    const int leaving = ctl->leaving;
    const int last = ctl->last_selected;

    Cand best;
    best.key = 0.0;
    best.idx = -1;
    best.aux = 0;
    double best_cbar = 0.0;
    for (int j = lp.n_art + blockIdx.x * blockDim.x + threadIdx.x; j < lp.n; j += gridDim.x * blockDim.x) {
        if (lp.pos[j] >= 0) continue;  // basic
        const int a = lp.col_start[j], b = lp.col_start[j + 1];
        double d_pi = 0.0, d_rho = 0.0, d_w = 0.0;
        if (pending) {
            for (int e = a; e < b; ++e) {
                const int r = lp.row_index[e];
                const double v = lp.value[e];
                d_pi += v * v_pi[r];
                d_rho += v * v_rho[r];
                d_w += v * v_w[r];
            }
        } else {
            for (int e = a; e < b; ++e) d_pi += lp.value[e] * v_pi[lp.row_index[e]];
        }
        double g = 1.0;
        if (RULE == RELP_PIVOT_STEEPEST_EDGE) {
            g = lp.gamma[j];
            if (pending) {
                if (j == leaving) {
                    g = gamma_q / (alpha_pq * alpha_pq);  // pivot_rule.rs:294-295
                } else {
                    // pivot_rule.rs:262-288 (Goldfarb-Reid)
                    const double sq = d_rho * d_rho;
                    g = g - 2.0 * d_rho * d_w + sq * gamma_q;
                    g = fmax(g, 1.0 + sq);
                }
                lp.gamma[j] = g;
            }
        }
        const double cbar = lp.cost[j] + d_pi;
        if (cbar < -tol_dual) {
            Cand c;
            c.idx = j;
            c.aux = 0;
            if (RULE == RELP_PIVOT_STEEPEST_EDGE) c.key = cbar * cbar / g;
            else if (RULE == RELP_PIVOT_DANTZIG) c.key = -cbar;
            else if (RULE == RELP_PIVOT_FIRST_PROFITABLE) c.key = -(double)j;
            else {
                if (last >= 0 && j == last) continue;
                const long long rank = (last < 0) ? j : (j > last ? (long long)j - last - 1 : (long long)j + lp.n - last);
                c.key = -(double)rank;
            }
            Cand nb = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? better<TIE_LARGER_IDX>(best, c) : better<TIE_SMALLER_IDX>(best, c);
            if (nb.idx == j) best_cbar = cbar;
            best = nb;
        }
    }
    Cand blk = (RULE == RELP_PIVOT_STEEPEST_EDGE) ? block_best<TIE_LARGER_IDX>(best, s_cand)
                                                  : block_best<TIE_SMALLER_IDX>(best, s_cand);
    if (blk.idx >= 0 && blk.idx == best.idx) {  // the winning thread publishes
        lp.cand_key[blockIdx.x] = blk.key;
        lp.cand_j[blockIdx.x] = blk.idx;
        lp.cand_cbar[blockIdx.x] = best_cbar;
    }
    if (blk.idx < 0 && threadIdx.x == 0) lp.cand_j[blockIdx.x] = -1;
}

// ---------------------------------------------------------------------------------------------------
// K2: entering column choice, FTRAN, ratio test and every O(m) state update, in ONE workgroup.
//   replaces  Tableau::generate_column -> BasisInverse::left_multiply_by_basis_inverse   tableau/mod.rs:126-130,
//                 lower_upper/mod.rs:180-210 (explicit inverse: basis_inverse_rows.rs:139-152)
//             Tableau::select_primal_pivot_row                                           tableau/mod.rs:287-313
//             Carry::update_b, update_minus_pi_and_obj, basis bookkeeping                carry/mod.rs:295-349,561-604
//             BasisInverse::basis_inverse_row (row p of the NEW inverse)                 lower_upper/mod.rs:254-272
// The ratio test is the two-pass Harris variant (f64 needs a pivot-size preference the exact reference does
// not); ties keep the reference's Bland rule (lowest leaving column).
// ---------------------------------------------------------------------------------------------------
template <int RULE>
__global__ void __launch_bounds__(1024) ftran_ratio_kernel(DeviceLP lp, int n_price_blocks, double tol_pivot,
                                                         double harris_delta, int skip_artificial_rows, int mode) {
    // mode 0: full iteration | 1: stop after the entering-column choice | 2: stop after the ratio test (no update)
    __shared__ Cand s_cand[18];
    __shared__ double s_red[18];
    __shared__ int s_q;
    __shared__ double s_cbar;
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
    // ---- FTRAN: alpha = Binv a_q --------------------------------------------------------------------
    const int ca = lp.col_start[q], cb_ = lp.col_start[q + 1];
    double sumsq = 0.0;
    double theta = INFINITY;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double* row = lp.Binv + (size_t)i * ld;
        double a = 0.0;
        for (int e = ca; e < cb_; ++e) a += row[lp.row_index[e]] * lp.value[e];
        lp.alpha[i] = a;
        sumsq += a * a;
        const bool skip = skip_artificial_rows && lp.basis[i] < lp.n_art;
        if (a > tol_pivot && !skip) theta = fmin(theta, (fmax(lp.xB[i], 0.0) + harris_delta) / a);
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
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            const double a = lp.alpha[i];
            const bool skip = skip_artificial_rows && lp.basis[i] < lp.n_art;
            if (a > tol_pivot && !skip && fmax(lp.xB[i], 0.0) / a <= theta_max) {
                Cand o;
                o.key = a;
                o.idx = i;
                o.aux = lp.basis[i];
                c = better<TIE_SMALLER_AUX>(c, o);
            }
        }
        c = block_best<TIE_SMALLER_AUX>(c, s_cand);
        p = c.idx;
    }
    if (p < 0) {
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
            ctl->p = p;
            ctl->cbar_q = cbar_q;
            ctl->gamma_q = gamma_q;
            ctl->pending = 0;
            ctl->forced_q = -1;
            ctl->forced_p = -1;
        }
        return;
    }
    // ---- O(m) updates (carry/mod.rs:295-349) --------------------------------------------------------
    const double alpha_pq = lp.alpha[p];
    const double xp = fmax(lp.xB[p], 0.0) / alpha_pq;
    const double* row_p = lp.Binv + (size_t)p * ld;
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double r = row_p[i] / alpha_pq;  // row p of the new inverse
        lp.rho[i] = r;
        lp.minus_pi[i] -= cbar_q * r;
        lp.xB[i] = (i == p) ? xp : lp.xB[i] - lp.alpha[i] * xp;
    }
    if (threadIdx.x == 0) {
        const int leaving = lp.basis[p];
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
        ctl->minus_obj -= cbar_q * xp;
        ctl->iters += 1;
        ctl->pending = 1;
        ctl->forced_q = -1;
        ctl->forced_p = -1;
        ctl->last_selected = q;
    }
}

// ---------------------------------------------------------------------------------------------------
// K3: product-form update of the explicit inverse, fused with w = alpha_q' Binv_old (the BTRAN the
// steepest-edge update needs).
//   replaces  BasisInverse::change_basis                       basis_inverse_rows.rs:36-70,123-137
//                 (role of the Forrest-Tomlin update            lower_upper/mod.rs:94-178)
//             BasisInverse::right_multiply_by_basis_inverse    lower_upper/mod.rs:212-237 (work vector, carry/mod.rs:575)
// Grid: (column strips of 256) x (row chunks).  Each element of Binv is read once and written at most once per
// pivot; rows with alpha_i == 0 are skipped (no traffic).  Partial column sums go to wpart[chunk][j].
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) update_kernel(DeviceLP lp, int rows_per_chunk, long long iters_expected_parity) {
    Ctl* ctl = lp.ctl;
    if (ctl->status != ST_RUNNING || !ctl->pending) return;
    (void)iters_expected_parity;
    const int m = lp.m, ld = lp.ld;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(m, r0 + rows_per_chunk);
    const int p = ctl->p;
    const double alpha_pq = ctl->alpha_pq;
    double wacc = 0.0;
    if (j < m) {
        const double rj = lp.rho[j];
        for (int i = r0; i < r1; ++i) {
            const double a = lp.alpha[i];  // wave-uniform
            if (i == p) {
                wacc += a * (rj * alpha_pq);  // old row p = rho * alpha_pq
                lp.Binv[(size_t)i * ld + j] = rj;
            } else if (a != 0.0) {
                const double old = lp.Binv[(size_t)i * ld + j];
                wacc += a * old;
                lp.Binv[(size_t)i * ld + j] = old - a * rj;
            }
        }
        lp.wpart[(size_t)blockIdx.y * m + j] = wacc;
    }
}

// K4: w[j] = sum over chunks (fixed order => deterministic).
__global__ void __launch_bounds__(256) wreduce_kernel(DeviceLP lp, int n_chunks) {
    Ctl* ctl = lp.ctl;
    if (ctl->status != ST_RUNNING || !ctl->pending) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lp.m) return;
    double acc = 0.0;
    for (int c = 0; c < n_chunks; ++c) acc += lp.wpart[(size_t)c * lp.m + j];
    lp.w[j] = acc;
}

// ---------------------------------------------------------------------------------------------------
// Phase set-up
// ---------------------------------------------------------------------------------------------------
// -pi_j = -sum_i c_{basis[i]} Binv[i][j]   (Carry::create_minus_pi_from_artificial, carry/mod.rs:226-260: the reference
// forms all of B^-1 with m FTRANs; here B^-1 is resident).  Also -obj = -sum_i xB_i c_{basis[i]} (carry/mod.rs:270-283).
__global__ void __launch_bounds__(256) pi_kernel(DeviceLP lp) {
    __shared__ double s_red[6];
    const int m = lp.m, ld = lp.ld;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) {
        double acc = 0.0;
        for (int i = 0; i < m; ++i) {
            const double c = lp.cost[lp.basis[i]];
            if (c != 0.0) acc += c * lp.Binv[(size_t)i * ld + j];
        }
        lp.minus_pi[j] = -acc;
    }
    if (blockIdx.x == 0) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) acc += lp.xB[i] * lp.cost[lp.basis[i]];
        acc = block_reduce<0>(acc, s_red);
        if (threadIdx.x == 0) lp.ctl->minus_obj = -acc;
    }
}

// xB = Binv rhs  (Carry::from_basis, carry/mod.rs:452-463): one wave per row, coalesced row reads.
__global__ void __launch_bounds__(256) xb_kernel(DeviceLP lp) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int row = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (row >= lp.m) return;
    const double* r = lp.Binv + (size_t)row * lp.ld;
    double acc = 0.0;
    for (int j = lane; j < lp.m; j += WAVE) acc += r[j] * lp.rhs[j];
    acc = wave_sum(acc);
    if (lane == 0) lp.xB[row] = acc;
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
            const double* row = lp.Binv + (size_t)i * lp.ld;
            double v = 0.0;
            for (int e = a; e < b; ++e) v += row[lp.row_index[e]] * lp.value[e];
            acc += v * v;
        }
    }
    acc = block_reduce<0>(acc, s_red);
    if (threadIdx.x == 0) lp.gamma[j] = 1.0 + acc;
}

__global__ void identity_kernel(double* X, int m, int ld) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < m) X[(size_t)i * ld + j] = (i == j) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------------
// Polish: Newton-Schulz  X <- X + X (I - B X).  Plays the role of BasisInverse::should_refactor + invert
// (lower_upper/mod.rs:78-92,249-252; carry/mod.rs:584-591) for the explicit inverse: it removes the drift of
// the product-form updates with GEMM-shaped work only.
// ---------------------------------------------------------------------------------------------------
// R[i][:] = e_i - sum_{k} B[i][k] X[k][:] ; row i of B comes from the CSR of A restricted to basic columns.
__global__ void __launch_bounds__(256) residual_kernel(DeviceLP lp, const double* X, double* R) {
    __shared__ double s_red[6];
    const int i = blockIdx.x;
    const int m = lp.m, ld = lp.ld;
    const int a = lp.row_start[i], b = lp.row_start[i + 1];
    double local_max = 0.0;
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        double acc = (i == j) ? 1.0 : 0.0;
        for (int e = a; e < b; ++e) {
            const int k = lp.pos[lp.col_index[e]];
            if (k >= 0) acc -= lp.row_value[e] * X[(size_t)k * ld + j];
        }
        R[(size_t)i * ld + j] = acc;
        local_max = fmax(local_max, fabs(acc));
    }
    // max via min of negatives
    const double blk = -block_reduce<1>(-local_max, s_red);
    if (threadIdx.x == 0) {
        // non-negative doubles order like their bit patterns
        atomicMax(reinterpret_cast<unsigned long long*>(&lp.ctl->residual),
                  (unsigned long long)__double_as_longlong(blk));
    }
}

// C = X + X R  (m x m, f64).  LDS-tiled 64x64 per workgroup, 4x4 per thread.
constexpr int GT = 64, GK = 16;
__global__ void __launch_bounds__(256) gemm_polish_kernel(const double* __restrict__ X, const double* __restrict__ R,
                                                        double* __restrict__ C, int m, int ld) {
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
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int gr = row0 + ty * 4 + u;
        if (gr >= m) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int gc = col0 + tx * 4 + v;
            if (gc < m) C[(size_t)gr * ld + gc] = X[(size_t)gr * ld + gc] + acc[u][v];
        }
    }
}

// X0 = s * B'  (row k of X0 = column basis[k] of A, scaled): start of a from-scratch Newton-Schulz inversion.
__global__ void transpose_basis_kernel(DeviceLP lp, double* X, double scale) {
    const int k = blockIdx.x;
    const int col = lp.basis[k];
    for (int j = threadIdx.x; j < lp.m; j += blockDim.x) X[(size_t)k * lp.ld + j] = 0.0;
    __syncthreads();
    for (int e = lp.col_start[col] + threadIdx.x; e < lp.col_start[col + 1]; e += blockDim.x)
        X[(size_t)k * lp.ld + lp.row_index[e]] = scale * lp.value[e];
}

// ---------------------------------------------------------------------------------------------------
// Zero-level pivots (phase_one.rs:232-278): first non-basic, non-artificial column with a non-zero entry in
// tableau row r (generate_element, lower_upper/mod.rs:239-247, without a full FTRAN per candidate).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) row_scan_kernel(DeviceLP lp, int r, double tol) {
    const double* row = lp.Binv + (size_t)r * lp.ld;
    for (int j = lp.n_art + blockIdx.x * blockDim.x + threadIdx.x; j < lp.n; j += gridDim.x * blockDim.x) {
        if (lp.pos[j] >= 0) continue;
        double acc = 0.0;
        for (int e = lp.col_start[j]; e < lp.col_start[j + 1]; ++e) acc += lp.value[e] * row[lp.row_index[e]];
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
    const double* row = lp.Binv + (size_t)i * lp.ld;
    double acc = 0.0;
    for (int e = 0; e < nnz; ++e) acc += row[rows[e]] * vals[e];
    out[i] = acc;
}
// out = v' Binv (BTRAN)
__global__ void btran_vec_kernel(DeviceLP lp, const int* rows, const double* vals, int nnz, double* out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= lp.m) return;
    double acc = 0.0;
    for (int e = 0; e < nnz; ++e) acc += vals[e] * lp.Binv[(size_t)rows[e] * lp.ld + j];
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
template <int RULE>
static void launch_price_rule(const DeviceLP& d, int blocks, size_t lds, bool use_lds, int skip_weights, double tol,
                              int n_chunks, hipStream_t s) {
    if (use_lds)
        hipLaunchKernelGGL((price_kernel<RULE, true>), dim3(blocks), dim3(256), lds, s, d, skip_weights, tol, n_chunks);
    else
        hipLaunchKernelGGL((price_kernel<RULE, false>), dim3(blocks), dim3(256), 0, s, d, skip_weights, tol, n_chunks);
}

void launch_price(const DeviceLP& d, int rule, int blocks, size_t lds, bool use_lds, int skip_weights, double tol,
                  int n_chunks, hipStream_t s) {
    switch (rule) {
        case RELP_PIVOT_DANTZIG: launch_price_rule<RELP_PIVOT_DANTZIG>(d, blocks, lds, use_lds, skip_weights, tol, n_chunks, s); break;
        case RELP_PIVOT_FIRST_PROFITABLE: launch_price_rule<RELP_PIVOT_FIRST_PROFITABLE>(d, blocks, lds, use_lds, skip_weights, tol, n_chunks, s); break;
        case RELP_PIVOT_FIRST_PROFITABLE_MEMORY: launch_price_rule<RELP_PIVOT_FIRST_PROFITABLE_MEMORY>(d, blocks, lds, use_lds, skip_weights, tol, n_chunks, s); break;
        default: launch_price_rule<RELP_PIVOT_STEEPEST_EDGE>(d, blocks, lds, use_lds, skip_weights, tol, n_chunks, s); break;
    }
}

void configure_price_lds(size_t lds) {
    // opt in to > 64 KB of dynamic LDS (160 KB per CU on gfx950)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&price_kernel<RELP_PIVOT_STEEPEST_EDGE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&price_kernel<RELP_PIVOT_DANTZIG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&price_kernel<RELP_PIVOT_FIRST_PROFITABLE, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&price_kernel<RELP_PIVOT_FIRST_PROFITABLE_MEMORY, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

void launch_ftran_ratio(const DeviceLP& d, int rule, int n_price_blocks, double tol_pivot, double harris_delta,
                        int skip_artificial_rows, int mode, hipStream_t s) {
    if (rule == RELP_PIVOT_STEEPEST_EDGE)
        hipLaunchKernelGGL((ftran_ratio_kernel<RELP_PIVOT_STEEPEST_EDGE>), dim3(1), dim3(1024), 0, s, d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode);
    else
        hipLaunchKernelGGL((ftran_ratio_kernel<RELP_PIVOT_DANTZIG>), dim3(1), dim3(1024), 0, s, d, n_price_blocks, tol_pivot, harris_delta, skip_artificial_rows, mode);
}

void launch_update(const DeviceLP& d, int rows_per_chunk, int n_chunks, hipStream_t s) {
    dim3 grid((d.m + 255) / 256, n_chunks);
    hipLaunchKernelGGL(update_kernel, grid, dim3(256), 0, s, d, rows_per_chunk, 0LL);
    hipLaunchKernelGGL(wreduce_kernel, dim3((d.m + 255) / 256), dim3(256), 0, s, d, n_chunks);
}

void launch_budget(const DeviceLP& d, long long add, hipStream_t s) {
    hipLaunchKernelGGL(budget_kernel, dim3(1), dim3(1), 0, s, d.ctl, add);
}

void launch_pi(const DeviceLP& d, hipStream_t s) {
    hipLaunchKernelGGL(pi_kernel, dim3((d.m + 255) / 256), dim3(256), 0, s, d);
}
void launch_xb(const DeviceLP& d, hipStream_t s) {
    hipLaunchKernelGGL(xb_kernel, dim3((d.m + 3) / 4), dim3(256), 0, s, d);
}
void launch_gamma_init(const DeviceLP& d, int identity, hipStream_t s) {
    hipLaunchKernelGGL(gamma_init_kernel, dim3(d.n - d.n_art), dim3(256), 0, s, d, identity);
}
void launch_identity(double* X, int m, int ld, hipStream_t s) {
    hipLaunchKernelGGL(identity_kernel, dim3((m + 255) / 256, m), dim3(256), 0, s, X, m, ld);
}
void launch_residual(const DeviceLP& d, const double* X, double* R, hipStream_t s) {
    hipLaunchKernelGGL(residual_kernel, dim3(d.m), dim3(256), 0, s, d, X, R);
}
void launch_gemm_polish(const double* X, const double* R, double* C, int m, int ld, hipStream_t s) {
    dim3 grid((m + GT - 1) / GT, (m + GT - 1) / GT);
    hipLaunchKernelGGL(gemm_polish_kernel, grid, dim3(256), 0, s, X, R, C, m, ld);
}
void launch_transpose_basis(const DeviceLP& d, double* X, double scale, hipStream_t s) {
    hipLaunchKernelGGL(transpose_basis_kernel, dim3(d.m), dim3(64), 0, s, d, X, scale);
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
