// Graph providers: the `MatrixProvider`s of the reference's examples/max_flow.rs and examples/shortest_path.rs, built on
// the incidence matrix of data/linear_program/network/representation.rs:24-100 (all under /root/reference).
//
// Both are instances of `MatrixData` (matrix_provider/matrix_data.rs:63-102):
//   max flow       rows  = (V-2) conservation equalities (s and t removed) | one VariableBound row per arc
//                  cols  = arcs (cost -1 when the arc leaves s, upper bound = capacity) | one bound slack per arc
//                  which is exactly examples/max_flow.rs:141-223: column(j) = incidence column + (V-2+j, 1), slack
//                  column E+j = (V-2+j, 1), right-hand side = 0 | capacities, initial pivots (V-2+j, E+j).
//   shortest path  rows  = (V-1) conservation equalities (s removed, b = e_t), cols = arcs with cost = length, no bounds
//                  (examples/shortest_path.rs:67-118).
// Arcs are given as the reference's column-major adjacency matrix enumerates them: sorted by (tail, head), no self arcs.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "model.hpp"

namespace relp {

struct Arc {
    int tail, head;
    Rat value;  // capacity or length
};

// representation.rs:29-91: the incidence column of one arc over the vertices that are not removed (removed: sorted)
inline SparseColumn incidence_column(const Arc& arc, const std::vector<int>& removed) {
    auto shift = [&](int v, bool& deleted) {
        auto it = std::lower_bound(removed.begin(), removed.end(), v);
        deleted = it != removed.end() && *it == v;
        return (int)(it - removed.begin());
    };
    bool tail_deleted, head_deleted;
    const int tail_shift = shift(arc.tail, tail_deleted), head_shift = shift(arc.head, head_deleted);
    SparseColumn column;
    const int tail_row = arc.tail - tail_shift, head_row = arc.head - head_shift;
    if (tail_deleted && head_deleted) return column;
    if (tail_deleted) {
        column.push(head_row, Rat(1));  // ArcDirection::Incoming
    } else if (head_deleted) {
        column.push(tail_row, Rat(-1));  // ArcDirection::Outgoing
    } else if (tail_row < head_row) {
        column.push(tail_row, Rat(-1));
        column.push(head_row, Rat(1));
    } else {
        column.push(head_row, Rat(1));
        column.push(tail_row, Rat(-1));
    }
    return column;
}

inline void check_arcs(int nr_vertices, const std::vector<Arc>& arcs, int s, int t) {
    if (nr_vertices < 2 || s < 0 || t < 0 || s >= nr_vertices || t >= nr_vertices || s == t)
        throw std::invalid_argument("graph provider: bad vertex count or terminals");
    for (size_t k = 0; k < arcs.size(); ++k) {
        const Arc& a = arcs[k];
        if (a.tail < 0 || a.head < 0 || a.tail >= nr_vertices || a.head >= nr_vertices || a.tail == a.head)
            throw std::invalid_argument("graph provider: arc endpoints out of range or a self arc (representation.rs:38)");
        if (k > 0 && !(arcs[k - 1].tail < a.tail || (arcs[k - 1].tail == a.tail && arcs[k - 1].head < a.head)))
            throw std::invalid_argument("graph provider: arcs must be sorted by (tail, head) without duplicates");
    }
}

// examples/max_flow.rs:53-75 (`Primal::new`) and :141-223 (the provider)
inline StandardForm make_max_flow(int nr_vertices, const std::vector<Arc>& arcs, int s, int t) {
    check_arcs(nr_vertices, arcs, s, t);
    StandardForm form;
    form.name = "max_flow";
    MatrixData& data = form.data;
    std::vector<int> removed = {std::min(s, t), std::max(s, t)};
    data.nr_equality = nr_vertices - 2;
    data.b.assign(nr_vertices - 2, Rat(0));
    for (const Arc& arc : arcs) {
        if (arc.value.sign() < 0) throw std::invalid_argument("max flow: negative capacity");
        data.constraints.push_back(incidence_column(arc, removed));
        Variable variable;
        variable.cost = arc.tail == s ? Rat(-1) : Rat(0);  // max_flow.rs:164-172 (`s_arc_range`)
        variable.has_upper = true;
        variable.upper = arc.value;
        data.variables.push_back(variable);
        form.column_names.push_back("arc_" + std::to_string(arc.tail) + "_" + std::to_string(arc.head));
    }
    data.finalize();
    form.nr_original = (int)arcs.size();
    form.free_negative_part.assign(arcs.size(), -1);
    return form;
}

// examples/shortest_path.rs:34-58 (`Primal::new`) and :67-118 (the provider)
inline StandardForm make_shortest_path(int nr_vertices, const std::vector<Arc>& arcs, int s, int t) {
    check_arcs(nr_vertices, arcs, s, t);
    StandardForm form;
    form.name = "shortest_path";
    MatrixData& data = form.data;
    std::vector<int> removed = {s};
    data.nr_equality = nr_vertices - 1;
    data.b.assign(nr_vertices - 1, Rat(0));
    data.b[t < s ? t : t - 1] = Rat(1);  // shortest_path.rs:87-93
    for (const Arc& arc : arcs) {
        data.constraints.push_back(incidence_column(arc, removed));
        Variable variable;
        variable.cost = arc.value;
        data.variables.push_back(variable);
        form.column_names.push_back("arc_" + std::to_string(arc.tail) + "_" + std::to_string(arc.head));
    }
    data.finalize();
    form.nr_original = (int)arcs.size();
    form.free_negative_part.assign(arcs.size(), -1);
    return form;
}

}  // namespace relp
