// Host-side mirror of the reference's `MatrixProvider` boundary for this path.
//
// `MatrixData` replaces `MatrixData<'a, F>` (/root/reference/src/algorithm/two_phase/matrix_provider/matrix_data.rs:63-102):
// a column-major constraint matrix plus *virtual* slack columns in six column groups / six row groups.
// Same names, same index arithmetic, same argument meaning, so that a caller of the reference's provider finds
// `column(j)`, `cost_value(j)`, `right_hand_side()`, `bound_row_index(j)`, `nr_rows()`, `nr_columns()`,
// `pivot_element_indices()` and `reconstruct_solution()` with identical results.
#pragma once
#include <cstdlib>
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "rat.hpp"

namespace relp {
// Diagnostics on stderr (timelines, the presolve's overflow report): the ONE place the library reads the environment.  They print;
// they never change what is computed.  RELP_TIME_SOLVE, RELP_TIME_UPLOAD, RELP_TIME_CERTIFY, RELP_TIME_REFACTOR, RELP_PRESOLVE_DEBUG,
// RELP_EXACT_PROFILE.
inline bool diagnostic(const char* name) { return std::getenv(name) != nullptr; }

struct SparseColumn {
    std::vector<int> index;  // sorted, unique rows
    std::vector<Rat> value;  // no explicit zeros
    size_t nnz() const { return index.size(); }
    void push(int i, const Rat& v) { index.push_back(i); value.push_back(v); }
};

// general_form/mod.rs `struct Variable` (only what the provider reads).
struct Variable {
    Rat cost;
    bool has_upper = false;
    Rat upper;
    // bookkeeping of `standardize()` for the solution back-mapping (general_form/mod.rs:808-825)
    Rat shift;
    bool flipped = false;
};

struct MatrixData {
    // --- inputs (matrix_data.rs:172-182) ---
    std::vector<SparseColumn> constraints;  // n structural columns over the constraint rows
    std::vector<Rat> b;                     // one per constraint row
    std::vector<Rat> ranges;                // one per range row
    int nr_equality = 0, nr_range = 0, nr_upper = 0, nr_lower = 0;
    std::vector<Variable> variables;

    // --- derived (matrix_data.rs:184-232) ---
    std::vector<int> variable_to_bound;  // -1 when the variable has no upper bound
    std::vector<int> bound_to_variable;
    int row_end[6] = {0, 0, 0, 0, 0, 0};  // Equality | Range | UpperIneq | LowerIneq | VariableBound | SlackBound
    int col_end[6] = {0, 0, 0, 0, 0, 0};  // Normal | RangeSlack | UpperIneqSlack | LowerIneqSlack | VarBoundSlack | SlackBoundSlack

    void finalize() {
        variable_to_bound.assign(variables.size(), -1);
        bound_to_variable.clear();
        for (size_t j = 0; j < variables.size(); ++j) {
            if (variables[j].has_upper) {
                variable_to_bound[j] = (int)bound_to_variable.size();
                bound_to_variable.push_back((int)j);
            }
        }
        int nb = (int)bound_to_variable.size();
        int rows[6] = {nr_equality, nr_range, nr_upper, nr_lower, nb, nr_range};
        int cols[6] = {(int)variables.size(), nr_range, nr_upper, nr_lower, nb, nr_range};
        int r = 0, c = 0;
        for (int k = 0; k < 6; ++k) {
            r += rows[k];
            c += cols[k];
            row_end[k] = r;
            col_end[k] = c;
        }
    }

    int nr_constraints() const { return row_end[3]; }                                   // matrix_data.rs:380-386
    int nr_variable_bounds() const { return (int)bound_to_variable.size() + nr_range; }  // :388-390
    int nr_rows() const { return nr_constraints() + nr_variable_bounds(); }
    int nr_columns() const { return col_end[5]; }                                       // :392-400
    int nr_normal_variables() const { return (int)constraints.size(); }

    // matrix_data.rs:253-273
    std::pair<int, int> column_type(int j) const {
        int previous = 0;
        for (int g = 0; g < 6; ++g) {
            if (j < col_end[g]) return {g, j - previous};
            previous = col_end[g];
        }
        return {-1, -1};
    }

    // matrix_data.rs:355-378 (upper direction; the lower direction is always None)
    int bound_row_index(int j) const {
        auto [g, k] = column_type(j);
        if (g == 0) return variable_to_bound[k] < 0 ? -1 : row_end[3] + variable_to_bound[k];
        if (g == 1) return row_end[4] + k;
        return -1;
    }

    // matrix_data.rs:291-329: constraint values, then the optional bound-row one.
    SparseColumn column(int j) const {
        auto [g, k] = column_type(j);
        SparseColumn out;
        switch (g) {
            case 0: {
                out = constraints[k];
                int br = bound_row_index(j);
                if (br >= 0) out.push(br, Rat(1));
                break;
            }
            case 1: out.push(row_end[0] + k, Rat(1)); out.push(row_end[4] + k, Rat(1)); break;
            case 2: out.push(row_end[1] + k, Rat(1)); break;
            case 3: out.push(row_end[2] + k, Rat(-1)); break;
            case 4: out.push(row_end[3] + k, Rat(1)); break;
            default: out.push(row_end[4] + k, Rat(1)); break;
        }
        return out;
    }

    // matrix_data.rs:331-339
    Rat cost_value(int j) const {
        auto [g, k] = column_type(j);
        return g == 0 ? variables[k].cost : Rat(0);
    }

    // matrix_data.rs:341-353
    std::vector<Rat> right_hand_side() const {
        std::vector<Rat> out = b;
        for (int j : bound_to_variable) out.push_back(variables[j].upper);
        for (const Rat& r : ranges) out.push_back(r);
        return out;
    }

    // A provider that is not a `MatrixData` (relp_model_from_provider): its columns are held as structural columns over
    // equality rows, and the unit columns its `PartialInitialBasis::pivot_element_indices` names are listed here.
    std::vector<std::pair<int, int>> provider_pivots;

    // matrix_data.rs:419-445: (row, column) pairs, sorted by row.
    std::vector<std::pair<int, int>> pivot_element_indices() const {
        std::vector<std::pair<int, int>> out;
        for (int j = 0; j < nr_upper; ++j) out.push_back({row_end[1] + j, col_end[1] + j});
        for (int j = 0; j < (int)bound_to_variable.size(); ++j) out.push_back({row_end[3] + j, col_end[3] + j});
        for (int j = 0; j < nr_range; ++j) out.push_back({row_end[4] + j, col_end[4] + j});
        if (!provider_pivots.empty()) {
            out.insert(out.end(), provider_pivots.begin(), provider_pivots.end());
            std::sort(out.begin(), out.end());
        }
        return out;
    }
};

// What the steps either side of the path keep (general_form/mod.rs:41-81), reduced to what the product needs to
// report the reference's objective: `objective = sum_j x_j c_j + fixed_cost` (general_form/mod.rs:840-851).
// general_form/mod.rs `RemovedVariable` (filled by the presolve): a value, or an affine function of other ORIGINAL
// variables, x = constant - sum_k coefficient_k x_k.
struct RemovedOriginal {
    bool function_of_others = false;
    Rat constant;
    std::vector<std::pair<int, Rat>> coefficients;
};

struct StandardForm {
    std::string name;
    MatrixData data;
    Rat fixed_cost;
    std::vector<std::string> column_names;   // variables still in the problem (all of them without presolve)
    std::vector<int> free_negative_part;     // per such variable: index of its negated twin, or -1
    int nr_original = 0;                     // their number (the first nr_original columns of `data`)
    // presolve bookkeeping (general_form/mod.rs `from_active_to_original`, `original_variables`)
    std::vector<std::string> all_column_names;  // every variable of the file
    std::vector<int> active_to_original;        // index into all_column_names per remaining variable
    std::vector<std::pair<int, RemovedOriginal>> removed;  // (original index, how to recover its value)
    bool presolve_dropped = false;  // the presolve was asked for but its result did not fit the 128-bit host model: loaded as in the file
    // what became of a requested presolve: applied as the reference does it | applied without the implied bounds of more than 126
    // bits (the OVERFLOW escalation of the host model) | dropped (the LP of the file)
    enum PresolveState { PRESOLVE_OFF = 0, PRESOLVE_APPLIED = 1, PRESOLVE_BOUNDED = 2, PRESOLVE_DROPPED = 3 };
    int presolve_state = PRESOLVE_OFF;

    // Values of the file's variables from the values of the standardised columns (general_form/mod.rs:753-771, 840-934):
    // un-shift, un-flip, recombine free variables, then evaluate the variables the presolve removed.
    int nr_file_variables() const { return all_column_names.empty() ? nr_original : (int)all_column_names.size(); }
    std::vector<double> original_solution(const std::vector<double>& standardised) const {
        const bool identity = active_to_original.empty();  // providers built without a file: no presolve, no renumbering
        std::vector<double> out((size_t)nr_file_variables(), 0.0);
        std::vector<char> known(out.size(), 0);
        for (int j = 0; j < nr_original; ++j) {
            double x = standardised[j];
            if (j < (int)free_negative_part.size() && free_negative_part[j] >= 0) x -= standardised[free_negative_part[j]];
            x -= data.variables[j].shift.to_double();
            if (data.variables[j].flipped) x = -x;
            const int original = identity ? j : active_to_original[j];
            out[original] = x;
            known[original] = 1;
        }
        // removed variables may refer to each other: resolve until nothing changes (the dependency graph is acyclic)
        bool progress = true;
        while (progress) {
            progress = false;
            for (const auto& [original, how] : removed) {
                if (known[original]) continue;
                bool ready = true;
                double value = how.constant.to_double();
                if (how.function_of_others)
                    for (const auto& [k, c] : how.coefficients) {
                        if (!known[k]) { ready = false; break; }
                        value -= c.to_double() * out[k];
                    }
                if (ready) {
                    out[original] = value;
                    known[original] = 1;
                    progress = true;
                }
            }
        }
        return out;
    }
};

// mps.cpp
StandardForm load_mps(const std::string& text, bool fixed_format, bool presolve = false);

}  // namespace relp
