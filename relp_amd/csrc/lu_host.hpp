// Host side of the LU basis factorisation `P B Q = L U`: sparse right-looking Gaussian elimination with Markowitz pivoting.
//
// Replaces (paths relative to /root/reference/src/algorithm/two_phase/tableau/inverse_maintenance/carry/lower_upper/):
//   LUDecomposition::invert / LUDecomposition::rows     mod.rs:78-92, decomposition/mod.rs:27-143
//   Markowitz::choose_pivot                             decomposition/pivoting.rs:45-81
//   subtract_multiple_of_row_from_other_row             decomposition/mod.rs:146-210
//
// The reference eliminates in exact arithmetic, so any non-zero pivot is acceptable and its rule is purely structural:
// the minimum of (r_i - 1)(c_j - 1) over the remaining entries, ties to the first entry in (column, row) order of the
// CURRENT (swapped) positions.  `reference_ties` reproduces that choice exactly (with threshold 0 the factors are the
// reference's, entry for entry: tests/test_lu_host.py replays its exact-factor known-answer tests).  The f64 product adds
// what floating point needs and the reference does not: a relative row threshold on the pivot magnitude and a limited
// search (the rows / columns of the lowest counts), the standard sparse-LU practice.
//
// The symbolic work is inherently sequential and tiny (25FV47: m = 821, a few 10^4 operations), so it runs on the host once
// per refactorisation; the numeric factors go to the device in one transfer (DeviceLU, lu.hip) where the per-pivot FTRAN /
// BTRAN / Forrest-Tomlin update run.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace relp {

struct LuOptions {
    double threshold = 0.1;        // accept a_ij only if |a_ij| >= threshold * max_k |a_ik| (0: any non-zero, the reference)
    bool reference_ties = false;   // full search with the reference's tie rule (pivoting.rs:60-80)
    int search_limit = 4;          // rows + columns examined with an acceptable candidate before the search stops
};

// Arithmetic of the factorisation: f64 for the solver's carry, Z_p for the exact certificate (certify.hip), where the
// same elimination runs modulo a 31-bit prime and any non-zero pivot is as good as another.
struct LuRealOps {
    using value = double;
    bool exact() const { return false; }
    bool is_zero(double v) const { return v == 0.0; }
    double magnitude(double v) const { return std::fabs(v); }
    double div(double a, double b) const { return a / b; }
    double mul(double a, double b) const { return a * b; }
    double sub(double a, double b) const { return a - b; }
    double neg(double a) const { return -a; }
};
struct LuModOps {
    using value = uint32_t;
    uint32_t p;
    bool exact() const { return true; }
    bool is_zero(uint32_t v) const { return v == 0; }
    double magnitude(uint32_t v) const { return v ? 1.0 : 0.0; }
    uint32_t inverse(uint32_t a) const {  // extended Euclid, a in [1, p)
        int64_t t = 0, nt = 1, r = p, nr = a;
        while (nr != 0) {
            const int64_t q = r / nr;
            int64_t tmp = t - q * nt; t = nt; nt = tmp;
            tmp = r - q * nr; r = nr; nr = tmp;
        }
        if (t < 0) t += p;
        return (uint32_t)t;
    }
    uint32_t mul(uint32_t a, uint32_t b) const { return (uint32_t)(((uint64_t)a * b) % p); }
    uint32_t div(uint32_t a, uint32_t b) const { return mul(a, inverse(b)); }
    uint32_t sub(uint32_t a, uint32_t b) const { return a >= b ? a - b : a + p - b; }
    uint32_t neg(uint32_t a) const { return a ? p - a : 0; }
};

template <class V>
struct HostLUT {
    int m = 0;
    bool singular = false;
    std::vector<int> rowpos;  // P.forward: original row  -> position in L / U   (decomposition/mod.rs:129-133)
    std::vector<int> colpos;  // Q.forward: basis slot    -> position in L / U
    // strictly lower part of L by rows of the position space (unit diagonal implied), entries (j < i, l_ij)
    std::vector<int> l_start, l_col;
    std::vector<V> l_val;
    // strictly upper part of U by rows, entries (j > i, u_ij); separate diagonal (mod.rs:36-58 `upper_diagonal`)
    std::vector<int> u_start, u_col;
    std::vector<V> u_val;
    std::vector<V> diag;
    // Level schedules of the four triangular solves (the DAG of each is static between refactorisations: Forrest-Tomlin
    // updates only ever REMOVE entries of the refactorised U, see lu.hpp).  sched[k]: 0 = L by rows (FTRAN), 1 = U by rows
    // (FTRAN), 2 = U by columns (BTRAN), 3 = L by columns (BTRAN).  Rows of one level are independent of each other.
    std::vector<int> lev_start[4], lev_row[4];
    long long nnz_l() const { return (long long)l_col.size(); }
    long long nnz_u() const { return (long long)u_col.size(); }
};
using HostLU = HostLUT<double>;

namespace lu_detail {
template <class V>
struct EntryT {
    int col;
    V val;
};
// doubly linked bucket lists of the active rows (or columns) by their current count
struct Buckets {
    std::vector<int> head, next, prev, count;
    void init(int m) {
        head.assign(m + 2, -1);
        next.assign(m, -1);
        prev.assign(m, -1);
        count.assign(m, 0);
    }
    void insert(int x, int c) {
        count[x] = c;
        prev[x] = -1;
        next[x] = head[c];
        if (head[c] >= 0) prev[head[c]] = x;
        head[c] = x;
    }
    void remove(int x) {
        const int c = count[x];
        if (prev[x] >= 0) next[prev[x]] = next[x];
        else head[c] = next[x];
        if (next[x] >= 0) prev[next[x]] = prev[x];
        next[x] = prev[x] = -1;
    }
    void move(int x, int c) {
        if (count[x] == c) return;
        remove(x);
        insert(x, c);
    }
};
}  // namespace lu_detail

// Basis columns in slot order, CSC (rows of a column in any order, no duplicates).
template <class Ops>
inline HostLUT<typename Ops::value> lu_factor_t(int m, const int* col_start, const int* row_index, const typename Ops::value* value,
                                                   const LuOptions& opt, const Ops& ops) {
    using V = typename Ops::value;
    using Entry = lu_detail::EntryT<V>;
    using lu_detail::Buckets;
    HostLUT<V> f;
    f.m = m;
    f.rowpos.assign(m, -1);
    f.colpos.assign(m, -1);
    f.diag.assign(m, V(0));
    // Working storage is kept per host thread and per arithmetic: a solve refactorises every few dozen pivots, and the ~4 m
    // small vectors below cost more to allocate than the elimination of a Netlib basis costs to run (their capacity survives).
    struct Workspace {
        std::vector<std::vector<Entry>> R, Lrow, Urow;
        std::vector<std::vector<int>> C;
    };
    static thread_local Workspace ws;
    auto fresh = [m](auto& vv) {
        if ((int)vv.size() < m) vv.resize(m);
        for (int i = 0; i < m; ++i) vv[i].clear();
    };
    fresh(ws.R);
    fresh(ws.C);
    fresh(ws.Lrow);
    fresh(ws.Urow);
    std::vector<std::vector<Entry>>& R = ws.R;   // active entries of the active rows
    std::vector<std::vector<int>>& C = ws.C;     // active rows of every active column (pattern)
    for (int j = 0; j < m; ++j)
        for (int e = col_start[j]; e < col_start[j + 1]; ++e) {
            if (ops.is_zero(value[e])) continue;
            R[row_index[e]].push_back({j, value[e]});
            C[j].push_back(row_index[e]);
        }
    Buckets rb, cb;
    rb.init(m);
    cb.init(m);
    for (int i = 0; i < m; ++i) {
        if (R[i].empty() || C[i].empty()) f.singular = true;
        rb.insert(i, (int)R[i].size());
        cb.insert(i, (int)C[i].size());
    }
    if (f.singular) return f;
    // current positions of the not yet pivoted rows / columns (the reference swaps the pivot to (k, k) at every step,
    // decomposition/mod.rs:224-273; only the tie rule looks at them)
    std::vector<int> rpos(m), cpos(m), row_at(m), col_at(m);
    for (int i = 0; i < m; ++i) rpos[i] = cpos[i] = row_at[i] = col_at[i] = i;
    std::vector<char> row_done(m, 0), col_done(m, 0);
    // L by (row, step, ratio) and U rows by step, in original indices until the end
    std::vector<std::vector<Entry>>& Lrow = ws.Lrow;      // Lrow[orig row] = (step k, ratio)
    std::vector<std::vector<Entry>>& Urow = ws.Urow;      // Urow[k] = (orig col, value)
    std::vector<int> where(m, -1);                        // scatter workspace: column -> index in the row being edited

    auto row_max = [&](int i) {
        double mx = 0.0;
        for (const Entry& e : R[i]) mx = std::max(mx, ops.magnitude(e.val));
        return mx;
    };

    for (int k = 0; k < m; ++k) {
        // ---- Markowitz search --------------------------------------------------------------------------------
        long long best_score = std::numeric_limits<long long>::max();
        int bi = -1, bj = -1, bjp = 0, bip = 0;
        V bval = V(0);
        int examined = 0;
        auto consider = [&](int i, int j, V v, long long score, double rmax) {
            if (ops.is_zero(v)) return;
            if (ops.magnitude(v) < opt.threshold * rmax) return;
            bool take;
            if (opt.reference_ties) {
                take = score < best_score || (score == best_score && (cpos[j] < bjp || (cpos[j] == bjp && rpos[i] < bip)));
            } else {
                take = score < best_score || (score == best_score && ops.magnitude(v) > ops.magnitude(bval));
            }
            if (bi < 0 || take) {
                best_score = score;
                bi = i;
                bj = j;
                bjp = cpos[j];
                bip = rpos[i];
                bval = v;
            }
        };
        for (int nz = 1; nz <= m - k; ++nz) {
            const long long bound = (long long)(nz - 1) * (nz - 1);
            if (bi >= 0 && (opt.reference_ties ? best_score < bound : best_score <= bound)) break;
            for (int j = cb.head[nz]; j >= 0; j = cb.next[j]) {
                for (int i : C[j]) {
                    V v = V(0);
                    for (const Entry& e : R[i])
                        if (e.col == j) { v = e.val; break; }
                    // a column singleton needs no elimination: any non-zero is a stable pivot
                    consider(i, j, v, (long long)(rb.count[i] - 1) * (nz - 1), nz == 1 ? 0.0 : row_max(i));
                }
                if (!opt.reference_ties && bi >= 0 && (best_score == 0 || ++examined >= opt.search_limit)) goto found;
            }
            for (int i = rb.head[nz]; i >= 0; i = rb.next[i]) {
                const double rmax = row_max(i);
                for (const Entry& e : R[i]) consider(i, e.col, e.val, (long long)(nz - 1) * (cb.count[e.col] - 1), rmax);
                if (!opt.reference_ties && bi >= 0 && (best_score == 0 || ++examined >= opt.search_limit)) goto found;
            }
        }
    found:
        if (bi < 0) {
            f.singular = true;
            return f;
        }
        const int pi = bi, pj = bj;
        const V pv = bval;
        // ---- swap to (k, k): positions only (decomposition/mod.rs:224-273) ------------------------------------
        {
            const int other_row = row_at[k], pr = rpos[pi];
            row_at[pr] = other_row;
            rpos[other_row] = pr;
            row_at[k] = pi;
            rpos[pi] = k;
            const int other_col = col_at[k], pc = cpos[pj];
            col_at[pc] = other_col;
            cpos[other_col] = pc;
            col_at[k] = pj;
            cpos[pj] = k;
        }
        f.rowpos[pi] = k;
        f.colpos[pj] = k;
        f.diag[k] = pv;
        row_done[pi] = 1;
        col_done[pj] = 1;
        rb.remove(pi);
        cb.remove(pj);
        // ---- the pivot row becomes row k of U; it leaves the column patterns ----------------------------------
        std::vector<Entry>& prow = Urow[k];
        for (const Entry& e : R[pi]) {
            if (e.col == pj) continue;
            prow.push_back(e);
            std::vector<int>& cj = C[e.col];
            for (size_t t = 0; t < cj.size(); ++t)
                if (cj[t] == pi) { cj[t] = cj.back(); cj.pop_back(); break; }
            cb.move(e.col, (int)cj.size());
        }
        // ---- eliminate column pj from the other rows (decomposition/mod.rs:71-100) ----------------------------
        for (int i2 : C[pj]) {
            if (i2 == pi) continue;
            std::vector<Entry>& row = R[i2];
            V a = V(0);
            for (size_t t = 0; t < row.size(); ++t)
                if (row[t].col == pj) { a = row[t].val; row[t] = row.back(); row.pop_back(); break; }
            const V ratio = ops.div(a, pv);
            Lrow[i2].push_back({k, ratio});
            if (!prow.empty()) {
                for (size_t t = 0; t < row.size(); ++t) where[row[t].col] = (int)t;
                for (const Entry& e : prow) {
                    const V product = ops.mul(ratio, e.val);
                    const int t = where[e.col];
                    if (t >= 0) {
                        const V old = row[t].val;
                        V updated = ops.sub(old, product);
                        if (!ops.exact() && !ops.is_zero(updated) && !opt.reference_ties &&
                            ops.magnitude(updated) <= 1e-15 * (ops.magnitude(old) + ops.magnitude(product))) updated = V(0);
                        row[t].val = updated;  // zeros are swept below
                    } else {
                        row.push_back({e.col, ops.neg(product)});
                        C[e.col].push_back(i2);
                        cb.move(e.col, (int)C[e.col].size());
                    }
                }
                for (size_t t = 0; t < row.size(); ++t) where[row[t].col] = -1;
                for (size_t t = 0; t < row.size();) {  // exact cancellations leave the pattern (decomposition/mod.rs:176-186)
                    if (ops.is_zero(row[t].val)) {
                        std::vector<int>& cj = C[row[t].col];
                        for (size_t s = 0; s < cj.size(); ++s)
                            if (cj[s] == i2) { cj[s] = cj.back(); cj.pop_back(); break; }
                        cb.move(row[t].col, (int)cj.size());
                        row[t] = row.back();
                        row.pop_back();
                    } else {
                        ++t;
                    }
                }
            }
            if (row.empty()) {
                f.singular = true;
                return f;
            }
            rb.move(i2, (int)row.size());
        }
        C[pj].clear();
        R[pi].clear();
    }
    // ---- position space, row-major L and U ----------------------------------------------------------------------
    f.l_start.assign(m + 1, 0);
    f.u_start.assign(m + 1, 0);
    for (int i = 0; i < m; ++i) f.l_start[f.rowpos[i] + 1] = (int)Lrow[i].size();
    for (int k = 0; k < m; ++k) f.u_start[k + 1] = (int)Urow[k].size();
    for (int k = 0; k < m; ++k) {
        f.l_start[k + 1] += f.l_start[k];
        f.u_start[k + 1] += f.u_start[k];
    }
    f.l_col.resize(f.l_start[m]);
    f.l_val.resize(f.l_start[m]);
    f.u_col.resize(f.u_start[m]);
    f.u_val.resize(f.u_start[m]);
    for (int i = 0; i < m; ++i) {
        int dst = f.l_start[f.rowpos[i]];
        std::sort(Lrow[i].begin(), Lrow[i].end(), [](const Entry& a, const Entry& b) { return a.col < b.col; });
        for (const Entry& e : Lrow[i]) {
            f.l_col[dst] = e.col;
            f.l_val[dst++] = e.val;
        }
    }
    for (int k = 0; k < m; ++k) {
        int dst = f.u_start[k];
        for (Entry& e : Urow[k]) e.col = f.colpos[e.col];
        std::sort(Urow[k].begin(), Urow[k].end(), [](const Entry& a, const Entry& b) { return a.col < b.col; });
        for (const Entry& e : Urow[k]) {
            f.u_col[dst] = e.col;
            f.u_val[dst++] = e.val;
        }
    }
    return f;
}

inline HostLU lu_factor(int m, const int* col_start, const int* row_index, const double* value, const LuOptions& opt) {
    return lu_factor_t<LuRealOps>(m, col_start, row_index, value, opt, LuRealOps{});
}

// Level sets: level(i) = 1 + max level of the rows i reads (0 when it reads none); rows sorted by level.
template <class V>
inline void lu_schedules(HostLUT<V>& f) {
    const int m = f.m;
    std::vector<int> lev(m);
    auto finish = [&](int k) {
        int depth = 0;
        for (int i = 0; i < m; ++i) depth = std::max(depth, lev[i]);
        f.lev_start[k].assign(depth + 2, 0);
        for (int i = 0; i < m; ++i) f.lev_start[k][lev[i] + 1]++;
        for (int l = 0; l <= depth; ++l) f.lev_start[k][l + 1] += f.lev_start[k][l];
        f.lev_row[k].resize(m);
        std::vector<int> fill(f.lev_start[k].begin(), f.lev_start[k].end() - 1);
        for (int i = 0; i < m; ++i) f.lev_row[k][fill[lev[i]]++] = i;
    };
    // 0: L by rows, ascending (row i reads columns j < i)
    for (int i = 0; i < m; ++i) {
        int l = 0;
        for (int e = f.l_start[i]; e < f.l_start[i + 1]; ++e) l = std::max(l, lev[f.l_col[e]] + 1);
        lev[i] = l;
    }
    finish(0);
    // 1: U by rows, descending (row i reads columns j > i)
    for (int i = m - 1; i >= 0; --i) {
        int l = 0;
        for (int e = f.u_start[i]; e < f.u_start[i + 1]; ++e) l = std::max(l, lev[f.u_col[e]] + 1);
        lev[i] = l;
    }
    finish(1);
    // 2: U by columns, ascending (column j reads rows i < j): level(j) = 1 + max level(i) over the entries (i, j)
    std::fill(lev.begin(), lev.end(), 0);
    for (int i = 0; i < m; ++i)
        for (int e = f.u_start[i]; e < f.u_start[i + 1]; ++e) lev[f.u_col[e]] = std::max(lev[f.u_col[e]], lev[i] + 1);
    finish(2);
    // 3: L by columns, descending (column j reads rows i > j)
    std::fill(lev.begin(), lev.end(), 0);
    for (int i = m - 1; i >= 0; --i)
        for (int e = f.l_start[i]; e < f.l_start[i + 1]; ++e) lev[f.l_col[e]] = std::max(lev[f.l_col[e]], lev[i] + 1);
    finish(3);
}

// The triangular factors INVERTED, as sparse matrices in the same position space (the inverse-factor carry, lu.hpp):
//   L^-1 by rows, strictly lower part (unit diagonal implied):  row_i(L^-1) = e_i - sum_{j < i} l_ij row_j(L^-1)
//   U^-1 by rows, diagonal included:                            row_i(U^-1) = (e_i - sum_{j > i} u_ij row_j(U^-1)) / u_ii
// Each row is accumulated in a dense work vector with a list of the touched columns (exact zeros from cancellation are dropped).
// On the bases of a simplex run the two inverses together hold 3-10 x the entries of L + U (25FV47 23 k against 5.5 k, GREENBEA
// 47-84 k against 14-16 k) where B^-1 itself holds 300 k-1.7 M: the fill stays inside each triangle.  Returns false (and leaves
// `out` unspecified) when the entries exceed `max_entries`: the caller falls back or reports.  out.diag is all ones, out.lev_*
// describe ONE level holding every row (a product with an explicit inverse has no dependencies between rows).
inline bool lu_invert_factors(const HostLU& f, size_t max_entries, HostLU& out) {
    const int m = f.m;
    // (`out` is reused from one refactorisation to the next: every vector keeps its capacity)
    out.m = m;
    out.singular = false;
    out.rowpos = f.rowpos;
    out.colpos = f.colpos;
    out.diag.assign(m, 1.0);
    thread_local std::vector<double> work;
    thread_local std::vector<char> seen;
    thread_local std::vector<int> touched, rev_start, rev_col;
    thread_local std::vector<double> rev_val;
    work.assign(m, 0.0);
    seen.assign(m, 0);
    // ---- L^-1, rows ascending (entries in the order they are first touched: deterministic, and a product does not care) -------------
    out.l_start.assign(m + 1, 0);
    out.l_col.clear();
    out.l_val.clear();
    for (int i = 0; i < m; ++i) {
        touched.clear();
        for (int e = f.l_start[i]; e < f.l_start[i + 1]; ++e) {
            const int j = f.l_col[e];
            const double lij = f.l_val[e];
            // - l_ij * row_j(L^-1): its unit diagonal and its strict part
            if (!seen[j]) { seen[j] = 1; touched.push_back(j); }
            work[j] -= lij;
            for (int t = out.l_start[j]; t < out.l_start[j + 1]; ++t) {
                const int c = out.l_col[t];
                if (!seen[c]) { seen[c] = 1; touched.push_back(c); }
                work[c] -= lij * out.l_val[t];
            }
        }
        for (int c : touched) {
            if (work[c] != 0.0) {
                out.l_col.push_back(c);
                out.l_val.push_back(work[c]);
            }
            work[c] = 0.0;
            seen[c] = 0;
        }
        out.l_start[i + 1] = (int)out.l_col.size();
        if (out.l_col.size() > max_entries) return false;
    }
    // ---- U^-1, rows descending into one array in the order they are made (row i is the (m - 1 - i)-th), then laid out ascending -----
    rev_start.assign(m + 1, 0);
    rev_col.clear();
    rev_val.clear();
    for (int i = m - 1; i >= 0; --i) {
        touched.clear();
        seen[i] = 1;
        touched.push_back(i);
        work[i] = 1.0;
        for (int e = f.u_start[i]; e < f.u_start[i + 1]; ++e) {
            const int j = f.u_col[e];
            const double uij = f.u_val[e];
            const int r = m - 1 - j;
            for (int t = rev_start[r]; t < rev_start[r + 1]; ++t) {
                const int c = rev_col[t];
                if (!seen[c]) { seen[c] = 1; touched.push_back(c); }
                work[c] -= uij * rev_val[t];
            }
        }
        const double inv = 1.0 / f.diag[i];
        for (int c : touched) {
            if (work[c] != 0.0) {
                rev_col.push_back(c);
                rev_val.push_back(work[c] * inv);
            }
            work[c] = 0.0;
            seen[c] = 0;
        }
        rev_start[m - i] = (int)rev_col.size();
        if (rev_col.size() + out.l_col.size() > max_entries) return false;
    }
    out.u_start.assign(m + 1, 0);
    out.u_col.resize(rev_col.size());
    out.u_val.resize(rev_col.size());
    {
        int at = 0;
        for (int i = 0; i < m; ++i) {
            const int r = m - 1 - i;
            const int n = rev_start[r + 1] - rev_start[r];
            std::copy(rev_col.begin() + rev_start[r], rev_col.begin() + rev_start[r + 1], out.u_col.begin() + at);
            std::copy(rev_val.begin() + rev_start[r], rev_val.begin() + rev_start[r + 1], out.u_val.begin() + at);
            at += n;
            out.u_start[i + 1] = at;
        }
    }
    for (int k = 0; k < 4; ++k) {
        if ((int)out.lev_row[k].size() != m) {
            out.lev_start[k] = {0, m};
            out.lev_row[k].resize(m);
            for (int i = 0; i < m; ++i) out.lev_row[k][i] = i;
        }
    }
    return true;
}

// Longest dependency chain of the two triangular solves (each hop is one LDS round trip on the device, lu.hip).
template <class V>
inline void lu_depths(const HostLUT<V>& f, int* depth_l, int* depth_u) {
    std::vector<int> lev(f.m, 0);
    int dl = 0, du = 0;
    for (int i = 0; i < f.m; ++i) {
        int l = 0;
        for (int e = f.l_start[i]; e < f.l_start[i + 1]; ++e) l = std::max(l, lev[f.l_col[e]] + 1);
        lev[i] = l;
        dl = std::max(dl, l);
    }
    for (int i = f.m - 1; i >= 0; --i) {
        int l = 0;
        for (int e = f.u_start[i]; e < f.u_start[i + 1]; ++e) l = std::max(l, lev[f.u_col[e]] + 1);
        lev[i] = l;
        du = std::max(du, l);
    }
    // (the second loop reuses lev: rows above i were overwritten only after they were read as columns > i)
    if (depth_l) *depth_l = dl;
    if (depth_u) *depth_u = du;
}

}  // namespace relp
