// C++ CPU oracle: exact-rational restatement of relp's two-phase revised simplex (TEST INFRASTRUCTURE ONLY).
//
// Same algorithm, data structures and asymptotics as the reference's `Carry<RationalBig, LUDecomposition<_>>` path;
// it is the compiled twin of oracle/relp_oracle/*.py (which the reference's known-answer tests pin) and must produce
// the same pivot sequence, basis and exact optimum -- tests/test_oracle_cpp.py checks that against tests/golden/.
// It is what bench.py times as `cpu_baseline` (kind "port"): relp itself needs nightly Rust, which this image lacks.
// Nothing here is linked into librelp_amd.so, and nothing in the product may call it.
//
// Reference files followed (relative to src/algorithm/two_phase/ unless stated), each function cites file:line:
//   tableau/inverse_maintenance/carry/lower_upper/{mod.rs, eta_file.rs, decomposition/mod.rs, decomposition/pivoting.rs,
//   permutation/*.rs}, tableau/inverse_maintenance/carry/mod.rs, tableau/mod.rs, tableau/kind/**, strategy/pivot_rule.rs,
//   phase_one.rs, phase_two.rs, mod.rs, matrix_provider/filter/generic_wrapper.rs.
//
// Input: a problem dump written by oracle/relp_oracle/dump.py (the provider's columns, costs, right-hand side and
// initial pivots, all exact).  Output: one JSON object on stdout.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <vector>

#include "rational.hpp"

using oracle::Rat;

typedef std::pair<int, Rat> Entry;
typedef std::vector<Entry> SV;  // sorted by index, unique, zero-free (data/linear_algebra/vector/sparse.rs:90-95)

static const Rat ZERO(0), ONE(1);
static bool g_tuned = false;  // row-indexed BTRAN instead of the reference's column scans (results identical)

static bool by_index(const Entry& a, const Entry& b) { return a.first < b.first; }
static void sort_sv(SV& v) { std::sort(v.begin(), v.end(), by_index); }
static int sorted_find(const SV& v, int index) {
    size_t lo = 0, hi = v.size();
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (v[mid].first < index) lo = mid + 1;
        else hi = mid;
    }
    return lo < v.size() && v[lo].first == index ? (int)lo : -1;
}
static size_t lower_bound_sv(const SV& v, int index) {
    size_t lo = 0, hi = v.size();
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        if (v[mid].first < index) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
// index_utils::inner_product_slice_iter (sorted merge dot; sparse.rs:105-111)
static Rat sparse_dot(const SV& a, const SV& b) {
    Rat total;
    size_t i = 0, j = 0;
    while (i < a.size() && j < b.size()) {
        if (a[i].first < b[j].first) ++i;
        else if (a[i].first > b[j].first) ++j;
        else {
            total += a[i].second * b[j].second;
            ++i;
            ++j;
        }
    }
    return total;
}

// ---- permutations (permutation/{full,rotate_to_back,swap}.rs) ------------------------------------------------------
struct FullPermutation {  // full.rs:15-110
    std::vector<int> fwd, bwd;
    FullPermutation() {}
    explicit FullPermutation(const std::vector<int>& forward) : fwd(forward), bwd(forward.size()) {
        for (size_t i = 0; i < fwd.size(); ++i) bwd[fwd[i]] = (int)i;
    }
    static FullPermutation identity(int n) {
        std::vector<int> f(n);
        for (int i = 0; i < n; ++i) f[i] = i;
        return FullPermutation(f);
    }
    void invert() { std::swap(fwd, bwd); }  // full.rs:60-62
    int forward(int i) const { return fwd[i]; }
    int backward(int i) const { return bwd[i]; }
    int len() const { return (int)fwd.size(); }
};
struct RotateToBack {  // rotate_to_back.rs:15-63
    int index, len;
    int forward(int i) const { return i < index ? i : (i == index ? len - 1 : i - 1); }
    int backward(int i) const { return i < index ? i : (i < len - 1 ? i + 1 : index); }
};
template <class P>
static SV forward_unsorted(const P& p, const SV& items) {  // permutation/mod.rs:96-102
    SV out;
    out.reserve(items.size());
    for (const Entry& e : items) out.push_back(Entry(p.forward(e.first), e.second));
    return out;
}
template <class P>
static SV backward_unsorted(const P& p, const SV& items) {  // permutation/mod.rs:113-119
    SV out;
    out.reserve(items.size());
    for (const Entry& e : items) out.push_back(Entry(p.backward(e.first), e.second));
    return out;
}
template <class P>
static SV forward_sorted(const P& p, const SV& items) {  // permutation/mod.rs:58-66
    SV out = forward_unsorted(p, items);
    sort_sv(out);
    return out;
}
template <class P>
static SV backward_sorted(const P& p, const SV& items) {  // permutation/mod.rs:73-81
    SV out = backward_unsorted(p, items);
    sort_sv(out);
    return out;
}

// ---- eta file (eta_file.rs) ----------------------------------------------------------------------------------------
static void update_value(const Rat& difference, SV& vector, int index) {  // eta_file.rs:137-157
    if (difference.is_zero()) return;
    size_t pos = lower_bound_sv(vector, index);
    if (pos < vector.size() && vector[pos].first == index) {
        Rat updated = vector[pos].second - difference;
        if (updated.is_zero()) vector.erase(vector.begin() + pos);
        else vector[pos].second = updated;
    } else {
        vector.insert(vector.begin() + pos, Entry(index, -difference));
    }
}
struct EtaFile {  // eta_file.rs:14-18: R = I + e_pivot r'
    SV values;
    int pivot, len;
    void apply_left(SV& vector) const {  // eta_file.rs:49-65
        int pos = sorted_find(vector, pivot);
        if (pos < 0) return;
        Rat pivot_value = vector[pos].second;
        for (const Entry& e : values) update_value(e.second * pivot_value, vector, e.first);
    }
    void apply_right(SV& vector) const {  // eta_file.rs:72-105
        update_value(sparse_dot(values, vector), vector, pivot);
    }
    void update_spike_pivot_value(SV& spike) const {  // eta_file.rs:112-134 (values all lie right of the pivot)
        update_value(sparse_dot(values, spike), spike, pivot);
    }
};

// ---- LU decomposition with Forrest-Tomlin updates (lower_upper/mod.rs) ---------------------------------------------
typedef std::map<int, Rat> Worklist;  // BTreeMap<usize, F>
static void insert_or_shift_maybe_remove(Worklist& work, int index, const Rat& change) {  // mod.rs:400-415
    Worklist::iterator it = work.find(index);
    if (it == work.end()) work.emplace(index, -change);
    else {
        Rat updated = it->second - change;
        if (updated.is_zero()) work.erase(it);
        else it->second = updated;
    }
}
static Worklist to_worklist(const SV& items) {
    Worklist work;
    for (const Entry& e : items) work.emplace(e.first, e.second);
    return work;
}

struct ColumnAndSpike {  // mod.rs:417-432
    SV column, spike;
};

static void subtract_multiple_of_row(SV& to_edit, const Rat& ratio, const SV& being_removed, size_t skip,
                                     std::vector<int>& nnz_column);

struct LU {
    FullPermutation row_permutation, column_permutation;
    std::vector<SV> lower, upper;  // column major, m-1 columns each
    std::vector<Rat> diagonal;
    struct Update {
        EtaFile eta;
        RotateToBack q;
    };
    std::vector<Update> updates;
    // tuned mode only: row-wise copies of L and U
    mutable bool rows_valid = false;
    mutable std::vector<SV> upper_rows, lower_rows;

    int m() const { return row_permutation.len(); }
    bool should_refactor() const { return updates.size() > 30; }  // mod.rs:249-252

    static LU identity(int m) {  // mod.rs:67-76
        LU lu;
        lu.row_permutation = FullPermutation::identity(m);
        lu.column_permutation = FullPermutation::identity(m);
        lu.lower.assign(m - 1, SV());
        lu.upper.assign(m - 1, SV());
        lu.diagonal.assign(m, ONE);
        return lu;
    }
    static LU invert(const std::vector<SV>& columns) {  // mod.rs:78-92
        const int m = (int)columns.size();
        std::vector<SV> rows(m);
        for (int j = 0; j < m; ++j)
            for (const Entry& e : columns[j]) rows[e.first].push_back(Entry(j, e.second));
        return from_rows(rows);
    }

    // decomposition/pivoting.rs:45-81: first minimum of (r_i - 1)(c_j - 1) over the remaining entries in (j, i) order
    static void markowitz(const std::vector<int>& nnz_row, const std::vector<int>& nnz_column, const std::vector<SV>& rows,
                          int k, int& pivot_row, int& pivot_column) {
        bool have = false;
        long best_score = 0;
        for (int i = k; i < (int)rows.size(); ++i) {
            const SV& row = rows[i];
            for (size_t t = lower_bound_sv(row, k); t < row.size(); ++t) {
                const int j = row[t].first;
                const long score = (long)(nnz_row[i] - 1) * (nnz_column[j] - 1);
                if (!have || score < best_score || (score == best_score && (j < pivot_column || (j == pivot_column && i < pivot_row)))) {
                    have = true;
                    best_score = score;
                    pivot_row = i;
                    pivot_column = j;
                }
            }
        }
        if (!have) throw std::runtime_error("singular basis");
    }

    static LU from_rows(std::vector<SV> rows) {  // decomposition/mod.rs:27-143
        const int m = (int)rows.size();
        if (m <= 1) throw std::runtime_error("LU needs m > 1 (decomposition/mod.rs:32)");
        std::vector<int> row_permutation(m), column_permutation(m);
        for (int i = 0; i < m; ++i) row_permutation[i] = column_permutation[i] = i;
        std::vector<SV> lower_row_major(m - 1);
        std::vector<int> nnz_row(m), nnz_column(m, 0);  // decomposition/mod.rs:278-304
        for (int i = 0; i < m; ++i) {
            nnz_row[i] = (int)rows[i].size();
            for (const Entry& e : rows[i]) ++nnz_column[e.first];
        }
        for (int k = 0; k < m; ++k) {
            int pivot_row = -1, pivot_column = -1;
            markowitz(nnz_row, nnz_column, rows, k, pivot_row, pivot_column);
            // decomposition/mod.rs:224-273
            if (pivot_row != k) {
                std::swap(row_permutation[pivot_row], row_permutation[k]);
                std::swap(nnz_row[pivot_row], nnz_row[k]);
                std::swap(rows[pivot_row], rows[k]);
                if (pivot_row > 0 && k > 0) std::swap(lower_row_major[pivot_row - 1], lower_row_major[k - 1]);
            }
            if (pivot_column != k) {
                std::swap(column_permutation[pivot_column], column_permutation[k]);
                std::swap(nnz_column[pivot_column], nnz_column[k]);
                for (SV& row : rows) {
                    bool touched = false;
                    for (Entry& e : row) {
                        if (e.first == pivot_column) { e.first = k; touched = true; }
                        else if (e.first == k) { e.first = pivot_column; touched = true; }
                    }
                    if (touched) sort_sv(row);
                }
            }
            for (const Entry& e : rows[k]) {  // decomposition/mod.rs:57-60
                --nnz_row[k];
                --nnz_column[e.first];
            }
            const SV& current = rows[k];
            const Rat pivot_value = current[0].second;
            std::vector<std::pair<int, Rat>> ratios;  // decomposition/mod.rs:71-79
            for (int i = k + 1; i < m; ++i) {
                SV& row = rows[i];
                if (row.empty()) throw std::runtime_error("singular basis (empty row)");
                if (row[0].first == k) {
                    ratios.push_back(std::make_pair(i, row[0].second / pivot_value));
                    row.erase(row.begin());
                    --nnz_row[i];
                    --nnz_column[k];
                }
            }
            for (const std::pair<int, Rat>& r : ratios) {  // decomposition/mod.rs:82-100
                SV& row = rows[r.first];
                const int old_len = (int)row.size();
                subtract_multiple_of_row(row, r.second, current, 1, nnz_column);
                nnz_row[r.first] += (int)row.size() - old_len;
                lower_row_major[r.first - 1].push_back(Entry(k, r.second));
            }
        }
        LU lu;  // decomposition/mod.rs:108-133
        lu.upper.assign(m - 1, SV());
        lu.lower.assign(m - 1, SV());
        lu.diagonal.reserve(m);
        for (int i = 0; i < m; ++i) {
            lu.diagonal.push_back(rows[i][0].second);
            for (size_t t = 1; t < rows[i].size(); ++t) lu.upper[rows[i][t].first - 1].push_back(Entry(i, rows[i][t].second));
        }
        for (int idx = 0; idx < m - 1; ++idx)
            for (const Entry& e : lower_row_major[idx]) lu.lower[e.first].push_back(Entry(idx + 1, e.second));
        lu.row_permutation = FullPermutation(row_permutation);
        lu.row_permutation.invert();
        lu.column_permutation = FullPermutation(column_permutation);
        lu.column_permutation.invert();
        return lu;
    }

    // ---- FTRAN ---------------------------------------------------------------------------------------------------
    SV left_multiply_by_lower_inverse(Worklist work) const {  // mod.rs:286-305
        SV result;
        const int last = m() - 1;
        while (!work.empty()) {
            Worklist::iterator first = work.begin();
            const int row = first->first;
            const Rat value = first->second;
            work.erase(first);
            if (row != last)
                for (const Entry& e : lower[row]) insert_or_shift_maybe_remove(work, e.first, value * e.second);
            result.push_back(Entry(row, value));
        }
        return result;
    }
    SV left_multiply_by_upper_inverse(Worklist work) const {  // mod.rs:307-321, update_rhs :339-345
        SV result;
        while (!work.empty()) {
            Worklist::iterator back = std::prev(work.end());
            const int row = back->first;
            const Rat x = back->second / diagonal[row];
            work.erase(back);
            if (row > 0)
                for (const Entry& e : upper[row - 1]) insert_or_shift_maybe_remove(work, e.first, x * e.second);
            result.push_back(Entry(row, x));
        }
        std::reverse(result.begin(), result.end());
        return result;
    }
    ColumnAndSpike left_multiply_by_basis_inverse(const SV& column) const {  // mod.rs:180-210
        Worklist rhs;
        for (const Entry& e : column) rhs.emplace(row_permutation.forward(e.first), e.second);
        SV w = left_multiply_by_lower_inverse(rhs);
        for (const Update& u : updates) {
            u.eta.apply_right(w);
            w = forward_sorted(u.q, w);
        }
        ColumnAndSpike out;
        out.spike = w;
        SV result = left_multiply_by_upper_inverse(to_worklist(w));
        for (size_t t = updates.size(); t-- > 0;) result = backward_unsorted(updates[t].q, result);
        result = backward_unsorted(column_permutation, result);
        sort_sv(result);
        out.column = result;
        return out;
    }
    bool generate_element(int i, const SV& column, Rat& out) const {  // mod.rs:239-247
        SV result = left_multiply_by_basis_inverse(column).column;
        int pos = sorted_find(result, i);
        if (pos < 0) return false;
        out = result[pos].second;
        return true;
    }

    // ---- BTRAN ---------------------------------------------------------------------------------------------------
    void build_rows() const {
        const int mm = m();
        upper_rows.assign(mm, SV());
        lower_rows.assign(mm, SV());
        for (int j = 0; j < mm - 1; ++j) {
            for (const Entry& e : upper[j]) upper_rows[e.first].push_back(Entry(j + 1, e.second));
            for (const Entry& e : lower[j]) lower_rows[e.first].push_back(Entry(j, e.second));
        }
        rows_valid = true;
    }
    SV right_multiply_by_upper_inverse(Worklist work) const {  // mod.rs:373-397
        SV result;
        const int mm = m();
        if (g_tuned && !rows_valid) build_rows();
        while (!work.empty()) {
            Worklist::iterator first = work.begin();
            const int column = first->first;
            const Rat x = first->second / diagonal[column];
            work.erase(first);
            if (g_tuned) {
                for (const Entry& e : upper_rows[column]) insert_or_shift_maybe_remove(work, e.first, x * e.second);
            } else {
                for (int j = column + 1; j < mm; ++j) {  // the reference's scan of every later column (mod.rs:381-389)
                    int pos = sorted_find(upper[j - 1], column);
                    if (pos >= 0) insert_or_shift_maybe_remove(work, j, x * upper[j - 1][pos].second);
                }
            }
            result.push_back(Entry(column, x));
        }
        return result;
    }
    SV right_multiply_by_lower_inverse(Worklist work) const {  // mod.rs:347-371
        SV result;
        if (g_tuned && !rows_valid) build_rows();
        while (!work.empty()) {
            Worklist::iterator back = std::prev(work.end());
            const int column = back->first;
            const Rat value = back->second;
            work.erase(back);
            if (g_tuned) {
                for (const Entry& e : lower_rows[column]) insert_or_shift_maybe_remove(work, e.first, value * e.second);
            } else {
                for (int j = 0; j < column; ++j) {  // mod.rs:355-362
                    int pos = sorted_find(lower[j], column);
                    if (pos >= 0) insert_or_shift_maybe_remove(work, j, value * lower[j][pos].second);
                }
            }
            result.push_back(Entry(column, value));
        }
        std::reverse(result.begin(), result.end());
        return result;
    }
    SV btran_tail(SV lhs) const {
        lhs = right_multiply_by_upper_inverse(to_worklist(lhs));
        for (size_t t = updates.size(); t-- > 0;) {
            lhs = backward_sorted(updates[t].q, lhs);
            updates[t].eta.apply_left(lhs);
        }
        lhs = right_multiply_by_lower_inverse(to_worklist(lhs));
        return backward_sorted(row_permutation, lhs);
    }
    SV right_multiply_by_basis_inverse(const SV& row) const {  // mod.rs:212-237
        SV lhs;
        for (const Entry& e : row) lhs.push_back(Entry(column_permutation.forward(e.first), e.second));
        for (const Update& u : updates) lhs = forward_unsorted(u.q, lhs);
        return btran_tail(lhs);
    }
    SV basis_inverse_row(int row) const {  // mod.rs:254-272
        row = column_permutation.forward(row);
        for (const Update& u : updates) row = u.q.forward(row);
        return btran_tail(SV(1, Entry(row, ONE)));
    }

    // ---- Forrest-Tomlin (mod.rs:94-178) ----------------------------------------------------------------------------
    void change_basis(int pivot_row_index, const ColumnAndSpike& info) {
        const int mm = m();
        int t = column_permutation.forward(pivot_row_index);
        for (const Update& u : updates) t = u.q.forward(t);
        SV u_bar;  // mod.rs:112-125
        std::vector<std::pair<int, int>> to_zero;
        for (int j = t + 1; j < mm; ++j) {
            int pos = sorted_find(upper[j - 1], t);
            if (pos >= 0) {
                u_bar.push_back(Entry(j, upper[j - 1][pos].second));
                to_zero.push_back(std::make_pair(j, pos));
            }
        }
        EtaFile eta;
        eta.values = right_multiply_by_upper_inverse(to_worklist(u_bar));
        eta.pivot = t;
        eta.len = mm;
        for (const std::pair<int, int>& z : to_zero) upper[z.first - 1].erase(upper[z.first - 1].begin() + z.second);  // :129-131
        SV spike = info.spike;
        eta.update_spike_pivot_value(spike);  // :135
        if (sorted_find(spike, t) < 0) throw std::runtime_error("singular basis after update");
        const int disappearing = t == 0 ? 0 : t - 1;  // :141-149
        upper[disappearing] = spike;
        std::rotate(upper.begin() + disappearing, upper.begin() + disappearing + 1, upper.end());
        std::rotate(diagonal.begin() + t, diagonal.begin() + t + 1, diagonal.end());
        RotateToBack q = {t, mm};  // :158-161
        for (int j = std::max(t, 1); j < mm; ++j) upper[j - 1] = forward_sorted(q, upper[j - 1]);
        Entry corner = upper.back().back();  // :162-164
        if (corner.first != mm - 1) throw std::runtime_error("spike corner missing");
        upper.back().pop_back();
        diagonal.back() = corner.second;
        Update update = {eta, q};
        updates.push_back(update);
        rows_valid = false;
    }
};

// decomposition/mod.rs:146-210: to_edit -= ratio * being_removed[skip..] (sorted merge, zeros dropped), keeping the
// column counts current.
static void subtract_multiple_of_row(SV& to_edit, const Rat& ratio, const SV& being_removed, size_t skip,
                                     std::vector<int>& nnz_column) {
    SV merged;
    merged.reserve(to_edit.size() + being_removed.size());
    size_t index = skip;
    const size_t n = being_removed.size();
    for (const Entry& old : to_edit) {
        while (index < n && being_removed[index].first < old.first) {
            merged.push_back(Entry(being_removed[index].first, -(ratio * being_removed[index].second)));
            ++nnz_column[being_removed[index].first];
            ++index;
        }
        if (index < n && being_removed[index].first == old.first) {
            Rat product = ratio * being_removed[index].second;
            if (product != old.second) merged.push_back(Entry(old.first, old.second - product));
            else --nnz_column[old.first];
            ++index;
        } else {
            merged.push_back(old);
        }
    }
    for (; index < n; ++index) {
        merged.push_back(Entry(being_removed[index].first, -(ratio * being_removed[index].second)));
        ++nnz_column[being_removed[index].first];
    }
    to_edit.swap(merged);
}

// ---- provider: explicit columns (what MatrixData::column(j) yields, matrix_data.rs:291-329) ---------------------------
struct Provider {
    int nr_rows = 0;
    std::vector<SV> columns;
    std::vector<Rat> cost, rhs;
    std::vector<std::pair<int, int>> pivots;  // (row, column) sorted by row (matrix_data.rs:419-445)
    std::string route;                        // "partial" | "fully" | "full_basis"
    std::vector<int> filtered_rows;           // set on a RemoveRows view

    int nr_columns() const { return (int)columns.size(); }
    SV column(int j) const { return columns[j]; }  // a clone, like matrix_data.rs:302
    // filter/generic_wrapper.rs:27-205
    Provider remove_rows(const std::vector<int>& rows_to_skip) const {
        Provider out;
        std::vector<int> relabel(nr_rows, -1);
        size_t s = 0;
        int next = 0;
        for (int i = 0; i < nr_rows; ++i) {
            if (s < rows_to_skip.size() && rows_to_skip[s] == i) ++s;
            else relabel[i] = next++;
        }
        out.nr_rows = next;
        out.columns.resize(columns.size());
        for (size_t j = 0; j < columns.size(); ++j)
            for (const Entry& e : columns[j])
                if (relabel[e.first] >= 0) out.columns[j].push_back(Entry(relabel[e.first], e.second));
        out.cost = cost;
        for (int i = 0; i < nr_rows; ++i)
            if (relabel[i] >= 0) out.rhs.push_back(rhs[i]);
        out.filtered_rows = rows_to_skip;
        out.route = route;
        return out;
    }
};

// ---- tableau kinds (tableau/kind/**) ---------------------------------------------------------------------------------
struct Kind {
    enum Type { FULLY, PARTIALLY, NON_ARTIFICIAL } type;
    const Provider* provider;
    std::vector<int> column_to_row;  // PARTIALLY: artificial k sits on this row (partially.rs:17-107)

    int nr_artificial() const {
        return type == FULLY ? provider->nr_rows : (type == PARTIALLY ? (int)column_to_row.size() : 0);
    }
    int nr_rows() const { return provider->nr_rows; }
    int nr_columns() const { return nr_artificial() + provider->nr_columns(); }
    Rat initial_cost_value(int j) const {  // fully.rs:27-33, partially.rs:42-50, non_artificial.rs
        if (type == NON_ARTIFICIAL) return provider->cost[j];
        return j < nr_artificial() ? ONE : ZERO;
    }
    SV original_column(int j) const {  // fully.rs:35-43, partially.rs:52-60
        const int a = nr_artificial();
        if (j < a) return SV(1, Entry(type == FULLY ? j : column_to_row[j], ONE));
        return provider->column(j - a);
    }
};

// ---- Carry (tableau/inverse_maintenance/carry/mod.rs) -----------------------------------------------------------------
struct BasisChange {  // tableau/mod.rs:205-234
    int pivot_row_index, pivot_column_index, leaving_column_index;
    SV column_before_change, work_vector, basis_inverse_row;
};
struct Carry {
    Rat minus_objective;
    std::vector<Rat> minus_pi, b;
    std::vector<int> basis_indices;
    LU basis_inverse;

    int m() const { return (int)b.size(); }

    static std::vector<Rat> minus_pi_from_artificial(const LU& bi, const Provider& provider, const std::vector<int>& basis) {
        const int m = bi.m();  // carry/mod.rs:226-260: all of B^-1 by m FTRANs
        std::vector<Rat> pi(m);
        for (int j = 0; j < m; ++j) {
            SV column = bi.left_multiply_by_basis_inverse(SV(1, Entry(j, ONE))).column;
            for (const Entry& e : column) pi[j] += e.second * provider.cost[basis[e.first]];
        }
        for (Rat& v : pi) v = -v;
        return pi;
    }
    static Rat minus_obj_from_artificial(const Provider& provider, const std::vector<int>& basis, const std::vector<Rat>& b) {
        Rat total;  // carry/mod.rs:270-283
        for (int row = 0; row < provider.nr_rows; ++row) total += b[row] * provider.cost[basis[row]];
        return -total;
    }
    static Carry from_basis(const std::vector<int>& basis, const Provider& provider) {  // carry/mod.rs:444-478
        std::vector<SV> columns;
        for (int j : basis) columns.push_back(provider.column(j));
        Carry c;
        c.basis_inverse = LU::invert(columns);
        SV rhs;
        for (int i = 0; i < provider.nr_rows; ++i)
            if (!provider.rhs[i].is_zero()) rhs.push_back(Entry(i, provider.rhs[i]));
        c.b.assign(provider.nr_rows, ZERO);
        for (const Entry& e : c.basis_inverse.left_multiply_by_basis_inverse(rhs).column) c.b[e.first] = e.second;
        c.minus_objective = minus_obj_from_artificial(provider, basis, c.b);
        c.minus_pi = minus_pi_from_artificial(c.basis_inverse, provider, basis);
        c.basis_indices = basis;
        return c;
    }

    void update_b(int p, const SV& column) {  // carry/mod.rs:295-325
        int pos = sorted_find(column, p);
        if (pos < 0) throw std::runtime_error("Pivot value can't be zero.");
        b[p] /= column[pos].second;
        const Rat pivot_b = b[p];
        for (const Entry& e : column)
            if (e.first != p) b[e.first] -= e.second * pivot_b;
    }
    BasisChange change_basis(int p, int q, const ColumnAndSpike& info, const Rat& relative_cost, const Kind& kind) {
        // carry/mod.rs:561-604
        BasisChange change;
        const SV& column = info.column;
        change.work_vector = basis_inverse.right_multiply_by_basis_inverse(column);
        update_b(p, column);
        change.leaving_column_index = basis_indices[p];
        basis_indices[p] = q;
        if (basis_inverse.should_refactor()) {
            std::vector<SV> columns;
            for (int j : basis_indices) columns.push_back(kind.original_column(j));
            basis_inverse = LU::invert(columns);
        } else {
            basis_inverse.change_basis(p, info);
        }
        change.column_before_change = column;
        change.basis_inverse_row = basis_inverse.basis_inverse_row(p);
        for (const Entry& e : change.basis_inverse_row) minus_pi[e.first] -= relative_cost * e.second;  // :338-349
        minus_objective -= relative_cost * b[p];
        change.pivot_row_index = p;
        change.pivot_column_index = q;
        return change;
    }
    Rat cost_difference(const SV& column) const {  // carry/mod.rs:606-611
        Rat total;
        for (const Entry& e : column) total += minus_pi[e.first] * e.second;
        return total;
    }
};

// ---- Tableau (tableau/mod.rs) -------------------------------------------------------------------------------------------
struct Tableau {
    Carry im;
    std::set<int> basis_columns;  // HashSet<usize> in the reference
    Kind kind;

    int nr_rows() const { return kind.nr_rows(); }
    int nr_columns() const { return kind.nr_columns(); }
    int start_index() const { return kind.nr_artificial(); }  // pivot_rule.rs:57-80
    bool is_in_basis(int j) const { return basis_columns.count(j) != 0; }
    Rat relative_cost(int j) const { return im.cost_difference(kind.original_column(j)) + kind.initial_cost_value(j); }  // :106-112
    ColumnAndSpike generate_column(int j) const { return im.basis_inverse.left_multiply_by_basis_inverse(kind.original_column(j)); }
    BasisChange bring_into_basis(int q, int p, const ColumnAndSpike& info, const Rat& cost) {  // :48-64, :76-88
        BasisChange change = im.change_basis(p, q, info, cost, kind);
        basis_columns.erase(change.leaving_column_index);
        basis_columns.insert(q);
        return change;
    }
    int select_primal_pivot_row(const SV& column) const {  // :287-313 (ties: lowest leaving column)
        int best_row = -1, best_leaving = 0;
        Rat best_ratio;
        for (const Entry& e : column) {
            if (e.second.sign() <= 0) continue;
            Rat ratio = im.b[e.first] / e.second;
            const int leaving = im.basis_indices[e.first];
            if (best_row < 0) {
                best_row = e.first; best_ratio = ratio; best_leaving = leaving;
            } else {
                int c = compare(ratio, best_ratio);
                if (c == 0 && leaving < best_leaving) { best_row = e.first; best_leaving = leaving; }
                else if (c < 0) { best_row = e.first; best_ratio = ratio; best_leaving = leaving; }
            }
        }
        return best_row;
    }
    Rat variable_value(int column) const {  // :164-176
        if (!is_in_basis(column)) return ZERO;
        for (int i = 0; i < nr_rows(); ++i)
            if (im.basis_indices[i] == column) return im.b[i];
        return ZERO;
    }
    SV current_bfs() const {  // carry/mod.rs:636-645
        SV out;
        for (int i = 0; i < nr_rows(); ++i)
            if (!im.b[i].is_zero()) out.push_back(Entry(im.basis_indices[i], im.b[i]));
        sort_sv(out);
        return out;
    }
};

// ---- pivot rules (strategy/pivot_rule.rs) -------------------------------------------------------------------------------
struct PivotRule {
    enum Type { SE, DANTZIG, FIRST, MEMORY } type = SE;
    std::vector<Rat> gamma;
    std::vector<char> has_gamma;
    int last_selected = -1;

    static Rat initial_gamma(int j, const Tableau& t) {  // :299-305
        Rat total = ONE;
        for (const Entry& e : t.generate_column(j).column) total += e.second * e.second;
        return total;
    }
    void init(const Tableau& t) {  // :202-219
        last_selected = -1;
        if (type != SE) return;
        gamma.assign(t.nr_columns(), ZERO);
        has_gamma.assign(t.nr_columns(), 0);
        for (int j = t.start_index(); j < t.nr_columns(); ++j)
            if (!t.is_in_basis(j)) {
                gamma[j] = initial_gamma(j, t);
                has_gamma[j] = 1;
            }
    }
    bool find_first(const Tableau& t, int from, int to, int& q, Rat& cost) const {
        for (int j = from; j < to; ++j) {
            if (t.is_in_basis(j)) continue;
            Rat c = t.relative_cost(j);
            if (c.sign() < 0) { q = j; cost = c; return true; }
        }
        return false;
    }
    bool select(const Tableau& t, int& q, Rat& cost) {
        const int start = t.start_index(), n = t.nr_columns();
        if (type == FIRST) return find_first(t, start, n, q, cost);  // :86-109
        if (type == MEMORY) {  // :113-150
            bool found = last_selected < 0 ? find_first(t, start, n, q, cost)
                                           : (find_first(t, last_selected + 1, n, q, cost) || find_first(t, start, last_selected, q, cost));
            last_selected = found ? q : -1;
            return found;
        }
        bool found = false;
        Rat best_key;
        for (int j = start; j < n; ++j) {
            if (t.is_in_basis(j)) continue;
            Rat c = t.relative_cost(j);
            if (c.sign() >= 0) continue;
            if (type == DANTZIG) {  // :153-187: first minimum
                if (!found || c < cost) { found = true; q = j; cost = c; }
            } else {  // :221-241: last maximum of c^2 / gamma
                Rat key = c * c / gamma[j];
                if (!found || key >= best_key) { found = true; best_key = key; q = j; cost = c; }
            }
        }
        return found;
    }
    void after_basis_update(const BasisChange& info, const Tableau& t) {  // :243-296
        if (type != SE) return;
        has_gamma[info.pivot_column_index] = 0;
        Rat gamma_q = ONE;
        for (const Entry& e : info.column_before_change) gamma_q += e.second * e.second;
        const Rat two(2);
        for (int j = t.start_index(); j < (int)gamma.size(); ++j) {
            if (!has_gamma[j]) continue;
            SV column = t.kind.original_column(j);
            Rat alpha_j_bar = sparse_dot(info.basis_inverse_row, column);
            Rat g = gamma[j], alternative = ONE;
            if (!alpha_j_bar.is_zero()) {
                Rat squared = alpha_j_bar * alpha_j_bar;
                Rat inner = sparse_dot(info.work_vector, column);
                if (!inner.is_zero()) g -= two * alpha_j_bar * inner;
                g += squared * gamma_q;
                alternative = ONE + squared;
            }
            if (g < alternative) g = alternative;
            gamma[j] = g;
        }
        int pos = sorted_find(info.column_before_change, info.pivot_row_index);
        const Rat w_p = info.column_before_change[pos].second;
        gamma[info.leaving_column_index] = gamma_q / (w_p * w_p);
        has_gamma[info.leaving_column_index] = 1;
    }
};

// ---- driver (phase_one.rs, phase_two.rs, two_phase/mod.rs) -----------------------------------------------------------------
struct PivotLimit {};
struct Trace {
    long limit = -1;
    double max_seconds = -1;
    std::chrono::steady_clock::time_point start;
    long count[3] = {0, 0, 0};
    int phase = 1;
    size_t keep = 64;
    size_t max_bits = 0;
    std::vector<std::vector<int>> head;
    long stamp_every = 100;                               // cumulative seconds after every `stamp_every` pivots (bench.py: same-work ratios)
    std::vector<std::pair<long, double>> stamps;
    void record(int q, int p, int leaving, const Tableau& t) {
        ++count[phase];
        if (stamp_every > 0 && (count[1] + count[2]) % stamp_every == 0)
            stamps.push_back(std::make_pair(count[1] + count[2], std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count()));
        if (head.size() < keep) head.push_back({phase, q, p, leaving});
        if (t.im.minus_objective.is_big()) max_bits = std::max(max_bits, t.im.minus_objective.bits());
        if (limit >= 0 && count[1] + count[2] >= limit) throw PivotLimit();
        if (max_seconds >= 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() >= max_seconds) throw PivotLimit();
    }
};

// the loop shared by phase_one.rs:134-178 and phase_two.rs:36-58; returns false when the ratio test fails (unbounded)
static bool simplex_loop(Tableau& t, PivotRule& rule, Trace& trace) {
    for (;;) {
        int q = -1;
        Rat cost;
        if (!rule.select(t, q, cost)) return true;
        ColumnAndSpike info = t.generate_column(q);
        int p = t.select_primal_pivot_row(info.column);
        if (p < 0) return false;
        BasisChange change = t.bring_into_basis(q, p, info, cost);
        trace.record(q, p, change.leaving_column_index, t);
        rule.after_basis_update(change, t);
    }
}

// phase_one.rs:232-278
static std::vector<int> remove_artificial_basis_variables(Tableau& t, Trace& trace) {
    std::vector<int> rows_to_remove;
    const int a = t.kind.nr_artificial();
    std::vector<std::pair<int, int>> artificial_rows;  // kind/artificial/mod.rs:38-43
    for (int i = 0; i < t.nr_rows(); ++i)
        if (t.im.basis_indices[i] < a) artificial_rows.push_back(std::make_pair(i, t.im.basis_indices[i]));
    for (const std::pair<int, int>& ra : artificial_rows) {
        const int pivot_row = ra.first;
        const Rat constraint_value = t.variable_value(ra.second);
        int found = -1;
        Rat found_cost;
        for (int j = a; j < t.nr_columns() && found < 0; ++j) {
            if (t.is_in_basis(j)) continue;
            Rat cost = t.relative_cost(j);
            Rat element;
            if (!constraint_value.is_zero()) {
                if (!cost.is_zero()) continue;
                if (t.im.basis_inverse.generate_element(pivot_row, t.kind.original_column(j), element) && element.sign() > 0) {
                    found = j; found_cost = cost;
                }
            } else if (t.im.basis_inverse.generate_element(pivot_row, t.kind.original_column(j), element) && !element.is_zero()) {
                found = j; found_cost = cost;
            }
        }
        if (found >= 0) {
            ColumnAndSpike column = t.generate_column(found);
            BasisChange change = t.bring_into_basis(found, pivot_row, column, found_cost);
            trace.record(found, pivot_row, change.leaving_column_index, t);
        } else {
            rows_to_remove.push_back(pivot_row);
        }
    }
    return rows_to_remove;
}

static Provider read_problem(const char* path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error(std::string("cannot open ") + path);
    std::string word;
    int version = 0, m = 0, n = 0, k = 0;
    Provider p;
    in >> word >> version;
    if (word != "relp-problem" || version != 1) throw std::runtime_error("not a relp-problem dump");
    in >> word >> p.route >> word >> m >> word >> n >> word >> k;
    p.nr_rows = m;
    for (int i = 0; i < k; ++i) {
        int r, c;
        in >> r >> c;
        p.pivots.push_back(std::make_pair(r, c));
    }
    in >> word;  // rhs
    for (int i = 0; i < m; ++i) { in >> word; p.rhs.push_back(Rat::parse(word)); }
    in >> word;  // cost
    for (int j = 0; j < n; ++j) { in >> word; p.cost.push_back(Rat::parse(word)); }
    in >> word;  // columns
    p.columns.resize(n);
    for (int j = 0; j < n; ++j) {
        int nnz = 0;
        in >> nnz;
        for (int t = 0; t < nnz; ++t) {
            int i;
            in >> i >> word;
            p.columns[j].push_back(Entry(i, Rat::parse(word)));
        }
    }
    if (!in) throw std::runtime_error("truncated problem dump");
    return p;
}

int main(int argc, char** argv) {
    const char* path = nullptr;
    Trace trace;
    PivotRule rule;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--max-pivots" && i + 1 < argc) trace.limit = atol(argv[++i]);
        else if (a == "--max-seconds" && i + 1 < argc) trace.max_seconds = atof(argv[++i]);
        else if (a == "--trace" && i + 1 < argc) trace.keep = (size_t)atol(argv[++i]);
        else if (a == "--stamp-every" && i + 1 < argc) trace.stamp_every = atol(argv[++i]);
        else if (a == "--tuned") g_tuned = true;
        else if (a == "--rule" && i + 1 < argc) {
            std::string r = argv[++i];
            rule.type = r == "dantzig" ? PivotRule::DANTZIG : r == "first" ? PivotRule::FIRST : r == "memory" ? PivotRule::MEMORY : PivotRule::SE;
        } else path = argv[i];
    }
    if (!path) {
        fprintf(stderr, "usage: relp_cpu problem.txt [--max-pivots K] [--max-seconds S] [--trace N] [--rule se|dantzig|first|memory] [--tuned]\n");
        return 2;
    }
    try {
        Provider provider = read_problem(path);
        const auto start = std::chrono::steady_clock::now();
        trace.start = start;
        std::string status = "optimal";
        Tableau final_tableau;
        Provider reduced;  // outlives the tableau that borrows it
        const int m = provider.nr_rows;
        bool have_final = false;
        try {
            Tableau artificial;
            bool feasible = true;
            std::vector<int> rows_to_remove;
            if (provider.route == "full_basis") {
                // two_phase/mod.rs:80-109 with carry/mod.rs:480-497
                std::vector<int> basis;
                for (const std::pair<int, int>& rc : provider.pivots) basis.push_back(rc.second);
                final_tableau.kind = Kind{Kind::NON_ARTIFICIAL, &provider, {}};
                final_tableau.im = Carry::from_basis(basis, provider);
                final_tableau.basis_columns = std::set<int>(basis.begin(), basis.end());
            } else {
                Carry& im = artificial.im;
                im.b = provider.rhs;
                im.basis_inverse = LU::identity(m);
                if (provider.route == "partial") {  // partially.rs:125-205, carry/mod.rs:397-442
                    std::vector<int> real_column(m, -1), column_to_row;
                    for (const std::pair<int, int>& rc : provider.pivots) real_column[rc.first] = rc.second;
                    for (int i = 0; i < m; ++i)
                        if (real_column[i] < 0) column_to_row.push_back(i);
                    const int a = (int)column_to_row.size();
                    im.minus_pi.assign(m, ZERO);
                    im.basis_indices.resize(m);
                    Rat objective;
                    int next = 0;
                    for (int i = 0; i < m; ++i) {
                        if (real_column[i] < 0) {
                            im.basis_indices[i] = next++;
                            im.minus_pi[i] = -ONE;
                            objective += provider.rhs[i];
                        } else {
                            im.basis_indices[i] = a + real_column[i];
                        }
                    }
                    im.minus_objective = -objective;
                    artificial.kind = Kind{Kind::PARTIALLY, &provider, column_to_row};
                } else {  // fully.rs:82-98, carry/mod.rs:374-395
                    im.minus_pi.assign(m, -ONE);
                    im.basis_indices.resize(m);
                    Rat objective;
                    for (int i = 0; i < m; ++i) {
                        im.basis_indices[i] = i;
                        objective += provider.rhs[i];
                    }
                    im.minus_objective = -objective;
                    artificial.kind = Kind{Kind::FULLY, &provider, {}};
                }
                artificial.basis_columns = std::set<int>(im.basis_indices.begin(), im.basis_indices.end());
                trace.phase = 1;
                rule.init(artificial);  // phase_one.rs:123-179
                if (!simplex_loop(artificial, rule, trace)) throw std::runtime_error("Artificial cost can not be unbounded.");
                if (!artificial.im.minus_objective.is_zero()) feasible = false;
                if (feasible) {
                    const int a = artificial.kind.nr_artificial();
                    bool has_artificial = false;
                    for (int j : artificial.basis_columns) has_artificial |= j < a;
                    if (has_artificial) rows_to_remove = remove_artificial_basis_variables(artificial, trace);
                    // non_artificial.rs:99-165, carry/mod.rs:499-559
                    const Provider* target = &provider;
                    std::vector<int> basis;
                    std::vector<Rat> b;
                    if (!rows_to_remove.empty()) {
                        reduced = provider.remove_rows(rows_to_remove);
                        target = &reduced;
                        size_t s = 0;
                        for (int i = 0; i < m; ++i) {
                            if (s < rows_to_remove.size() && rows_to_remove[s] == i) { ++s; continue; }
                            basis.push_back(artificial.im.basis_indices[i] - a);
                            b.push_back(artificial.im.b[i]);
                        }
                        std::vector<SV> columns;
                        for (int j : basis) columns.push_back(target->column(j));
                        final_tableau.im.basis_inverse = LU::invert(columns);
                    } else {
                        for (int i = 0; i < m; ++i) basis.push_back(artificial.im.basis_indices[i] - a);
                        b = artificial.im.b;
                        final_tableau.im.basis_inverse = artificial.im.basis_inverse;
                    }
                    final_tableau.kind = Kind{Kind::NON_ARTIFICIAL, target, {}};
                    final_tableau.im.minus_pi = Carry::minus_pi_from_artificial(final_tableau.im.basis_inverse, *target, basis);
                    final_tableau.im.minus_objective = Carry::minus_obj_from_artificial(*target, basis, b);
                    final_tableau.im.b = b;
                    final_tableau.im.basis_indices = basis;
                    final_tableau.basis_columns = std::set<int>(basis.begin(), basis.end());
                } else {
                    status = "infeasible";
                }
            }
            if (status == "optimal") {
                have_final = true;
                trace.phase = 2;
                rule.init(final_tableau);  // phase_two.rs:22-59
                if (!simplex_loop(final_tableau, rule, trace)) status = "unbounded";
            }
        } catch (const PivotLimit&) {
            status = "pivot_limit";
        }
        const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        std::ostringstream out;
        out << "{\"status\":\"" << status << "\",\"pivots_phase1\":" << trace.count[1] << ",\"pivots_phase2\":" << trace.count[2]
            << ",\"seconds\":" << seconds << ",\"max_objective_bits\":" << trace.max_bits << ",\"tuned\":" << (g_tuned ? "true" : "false")
            << ",\"trace_head\":[";
        for (size_t i = 0; i < trace.head.size(); ++i)
            out << (i ? "," : "") << "[" << trace.head[i][0] << "," << trace.head[i][1] << "," << trace.head[i][2] << "," << trace.head[i][3] << "]";
        out << "],\"stamps\":[";
        for (size_t i = 0; i < trace.stamps.size(); ++i) out << (i ? "," : "") << "[" << trace.stamps[i].first << "," << trace.stamps[i].second << "]";
        out << "]";
        if (status == "optimal" && have_final) {
            out << ",\"objective\":\"" << (-final_tableau.im.minus_objective).to_string() << "\",\"basis\":[";
            for (size_t i = 0; i < final_tableau.im.basis_indices.size(); ++i) out << (i ? "," : "") << final_tableau.im.basis_indices[i];
            out << "],\"solution\":[";
            SV bfs = final_tableau.current_bfs();
            for (size_t i = 0; i < bfs.size(); ++i) out << (i ? "," : "") << "[" << bfs[i].first << ",\"" << bfs[i].second.to_string() << "\"]";
            out << "]";
        }
        out << "}";
        puts(out.str().c_str());
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "relp_cpu: %s\n", e.what());
        return 1;
    }
}
