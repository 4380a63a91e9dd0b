// Presolve of the general form (host side, exact rationals), applied between the MPS conversion and `standardize()` as the
// reference's Netlib harness does (tests/netlib/mod.rs:58).
//
// Mirrors /root/reference/src/data/linear_program/general_form/presolve/{mod.rs, counters.rs, queues.rs, updates.rs,
// rule/fixed_variable.rs, rule/bound_constraint.rs, rule/slack.rs, rule/domain_propagation.rs} and the application of the
// changes in general_form/mod.rs:335-463: same four rules, same queue disciplines (LIFO vectors, a FIFO set for the
// activity queue), same counters, so that the presolved problem equals the reference's.  tests/test_host_presolve.py
// compares it with the oracle (which the reference's 34 known-answer tests pin) on every shipped problem file.
#pragma once
#include <algorithm>
#include <deque>
#include <map>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>

#include "bigrat.hpp"
#include "model.hpp"

namespace relp {

enum RowKind { EQUAL = 0, RANGE = 1, LESS = 2, GREATER = 3 };

// the variable of the general form as mps.cpp keeps it (fixed-width exact numbers)
struct GeneralVariable {
    Rat cost;
    bool has_lower = false, has_upper = false;
    Rat lower, upper, shift;
    bool flipped = false;
};

// GeneralForm::new (general_form/mod.rs:211-237): what a caller hands over before any transformation.  Columns are sparse
// with ascending row indices and no explicit zeros; `range[i]` is read for RANGE rows only (b - range <= a x <= b).
struct GeneralInput {
    std::string name;
    bool maximize = false;
    std::vector<GeneralVariable> variables;
    std::vector<SparseColumn> columns;
    std::vector<RowKind> kind;
    std::vector<Rat> range;
    std::vector<Rat> b;
    std::vector<std::string> column_names;
    Rat fixed_cost;
};
// [presolve ->] standardize -> derive_matrix_data (general_form/mod.rs:335-463, 325-332, 262-304); defined in mps.cpp
StandardForm standardize_general_form(GeneralInput general, bool presolve_first);

// The presolve computes in arbitrary precision (bigrat.hpp): bound tightening leaves the 128-bit range on larger LPs.
typedef BigRat Num;

struct PVariable {
    Num cost;
    bool has_lower = false, has_upper = false;
    Num lower, upper;
};
struct PColumn {
    std::vector<int> index;
    std::vector<Num> value;
    size_t nnz() const { return index.size(); }
    void push(int i, const Num& v) { index.push_back(i); value.push_back(v); }
};

struct ConstraintKind {
    RowKind kind = EQUAL;
    Num range;  // RANGE only
    bool operator==(const ConstraintKind& o) const { return kind == o.kind && (kind != RANGE || range == o.range); }
    bool operator!=(const ConstraintKind& o) const { return !(*this == o); }
};

// general_form/mod.rs `RemovedVariable`: a value, or an affine function of other ORIGINAL variables
// (x = constant - sum_k coefficient_k x_k).
struct RemovedVariable {
    bool function_of_others = false;
    Num constant;  // the value when !function_of_others
    std::vector<std::pair<int, Num>> coefficients;
};

struct PresolveInfeasible : std::runtime_error {
    PresolveInfeasible() : std::runtime_error("presolve: the problem is infeasible") {}
};
struct PresolveUnbounded : std::runtime_error {
    PresolveUnbounded() : std::runtime_error("presolve: the problem is unbounded") {}
};

// The general form as mps.cpp holds it between conversion and standardisation.
struct GeneralProblem {
    bool maximize = false;
    std::vector<PVariable> variables;
    std::vector<PColumn> columns;
    std::vector<ConstraintKind> kinds;
    std::vector<Num> b;
    Num fixed_cost;
    std::vector<int> active_to_original;            // general_form/mod.rs `from_active_to_original`
    std::map<int, RemovedVariable> removed;          // original index -> how to recover its value
};

namespace presolve_detail {

enum Direction { LOWER = 0, UPPER = 1 };
enum Change { MEANINGFUL, NOT_MEANINGFUL, NO_CHANGE };
inline Direction flip(Direction d) { return d == LOWER ? UPPER : LOWER; }
inline Direction times_sign(Direction d, const Num& coefficient) { return coefficient.sign() > 0 ? d : flip(d); }

struct Optional {
    bool has = false;
    Num value;
    Optional() {}
    explicit Optional(const Num& v) : has(true), value(v) {}
};

inline bool is_empty_constraint_feasible(const Num& rhs, const ConstraintKind& kind) {  // presolve/mod.rs:403-425
    switch (kind.kind) {
        case EQUAL: return rhs.is_zero();
        case LESS: return rhs.sign() >= 0;
        case GREATER: return rhs.sign() <= 0;
        default: return rhs.sign() >= 0 && (rhs - kind.range).sign() <= 0;
    }
}
inline Num optimize_independent_column(bool maximize, const Num& cost, const Optional& lower, const Optional& upper) {  // updates.rs:368-389
    const Optional& bound = ((!maximize && cost.sign() > 0) || (maximize && cost.sign() < 0)) ? lower : upper;
    if (!bound.has) throw PresolveUnbounded();
    return bound.value;
}

struct BoundChange {
    enum Type { NONE, NEW_BOUND, SHIFT } type = NONE;
    Num shift;
};

class Index {  // presolve/mod.rs:31-45 with Counters, Queues and Updates inlined
public:
    typedef std::pair<int, int> Key;  // (variable, direction)
    const GeneralProblem& gf;
    std::vector<std::vector<std::pair<int, Num>>> rows;
    std::vector<int> count_constraint, count_variable;
    std::vector<std::pair<int, int>> count_activity;
    std::map<int, Num> b_changes;
    std::map<int, ConstraintKind> constraint_changes;
    Num fixed_cost;
    std::map<Key, Num> bounds, activity_variable_bounds;
    std::vector<std::pair<int, RemovedVariable>> removed_variables;
    std::vector<int> constraints_marked_removed;
    std::vector<int> q_substitution, q_bound, q_slack;
    std::deque<Key> q_activity;  // FIFOSet<(constraint, direction)> (crate fifo-set 1.0.0): each item at most once
    std::set<Key> q_activity_members;
    std::vector<std::pair<Optional, Optional>> activity_bounds;

    explicit Index(const GeneralProblem& problem) : gf(problem) {
        const int nr_rows = (int)gf.b.size(), nr_vars = (int)gf.variables.size();
        rows.resize(nr_rows);  // counters.rs:33-62
        for (int j = 0; j < nr_vars; ++j)
            for (size_t k = 0; k < gf.columns[j].nnz(); ++k) rows[gf.columns[j].index[k]].push_back({j, gf.columns[j].value[k]});
        for (int i = 0; i < nr_rows; ++i) count_constraint.push_back((int)rows[i].size());
        for (int j = 0; j < nr_vars; ++j) count_variable.push_back((int)gf.columns[j].nnz());
        for (int i = 0; i < nr_rows; ++i) {
            int lower_missing = 0, upper_missing = 0;
            for (auto& [j, coefficient] : rows[i]) {
                const PVariable& v = gf.variables[j];
                const bool has_l = coefficient.sign() > 0 ? v.has_lower : v.has_upper;
                const bool has_u = coefficient.sign() > 0 ? v.has_upper : v.has_lower;
                lower_missing += !has_l;
                upper_missing += !has_u;
            }
            count_activity.push_back({lower_missing, upper_missing});
        }
        for (int j = 0; j < nr_vars; ++j) {  // updates.rs:44-98
            if (count_variable[j] != 0) continue;
            const PVariable& v = gf.variables[j];
            RemovedVariable solved;
            if (v.cost.is_zero()) {
                solved.constant = feasible_value(original(j, LOWER), original(j, UPPER)).value;
            } else {
                solved.constant = optimize_independent_column(gf.maximize, v.cost, original(j, LOWER), original(j, UPPER));
                fixed_cost = fixed_cost + v.cost * solved.constant;
            }
            removed_variables.push_back({j, solved});
        }
        for (int i = 0; i < nr_rows; ++i) {
            if (count_constraint[i] != 0) continue;
            if (!is_empty_constraint_feasible(gf.b[i], gf.kinds[i])) throw PresolveInfeasible();
            constraints_marked_removed.push_back(i);
        }
        for (int i = 0; i < nr_rows; ++i)  // queues.rs:33-63
            if (count_constraint[i] == 1) q_bound.push_back(i);
        for (int i = 0; i < nr_rows; ++i) {
            if (count_constraint[i] <= 1) continue;
            if (count_activity[i].first <= 1) push_activity(i, LOWER);
            if (count_activity[i].second <= 1) push_activity(i, UPPER);
        }
        for (int j = 0; j < nr_vars; ++j)
            if (count_variable[j] == 1 && gf.variables[j].cost.is_zero()) q_slack.push_back(j);
        for (int j = 0; j < nr_vars; ++j) {
            const PVariable& v = gf.variables[j];
            if (count_variable[j] > 0 && v.has_lower && v.has_upper && v.lower == v.upper) q_substitution.push_back(j);
        }
        activity_bounds.resize(nr_rows);
    }

    // ---- counters.rs:64-88 ------------------------------------------------------------------------
    bool constraint_active(int i) const { return count_constraint[i] > 0; }
    bool variable_active(int j) const { return count_variable[j] > 0; }
    std::vector<std::pair<int, Num>> active_column(int j) const {
        std::vector<std::pair<int, Num>> out;
        for (size_t k = 0; k < gf.columns[j].nnz(); ++k)
            if (constraint_active(gf.columns[j].index[k])) out.push_back({gf.columns[j].index[k], gf.columns[j].value[k]});
        return out;
    }
    std::vector<std::pair<int, Num>> active_row(int i) const {
        std::vector<std::pair<int, Num>> out;
        for (auto& e : rows[i])
            if (variable_active(e.first)) out.push_back(e);
        return out;
    }

    // ---- updates.rs accessors --------------------------------------------------------------------
    Optional original(int j, Direction d) const {
        const PVariable& v = gf.variables[j];
        if (d == LOWER) return v.has_lower ? Optional(v.lower) : Optional();
        return v.has_upper ? Optional(v.upper) : Optional();
    }
    static Optional feasible_value(const Optional& lower, const Optional& upper) {  // updates.rs:133-152
        if (!lower.has && !upper.has) return Optional(Num(0));
        if (!lower.has) return upper;
        if (!upper.has) return lower;
        return lower.value <= upper.value ? upper : Optional();
    }
    const Num& b(int i) const {
        auto it = b_changes.find(i);
        return it == b_changes.end() ? gf.b[i] : it->second;
    }
    void change_b(int i, const Num& change) { b_changes[i] = b(i) + change; }
    const ConstraintKind& constraint_type(int i) const {
        auto it = constraint_changes.find(i);
        return it == constraint_changes.end() ? gf.kinds[i] : it->second;
    }
    Optional variable_bound(int j, Direction d) const {  // updates.rs:154-173
        auto it = activity_variable_bounds.find({j, d});
        if (it != activity_variable_bounds.end()) return Optional(it->second);
        it = bounds.find({j, d});
        if (it != bounds.end()) return Optional(it->second);
        return original(j, d);
    }
    Optional is_variable_fixed(int j) const {
        Optional lower = variable_bound(j, LOWER), upper = variable_bound(j, UPPER);
        return lower.has && upper.has && lower.value == upper.value ? lower : Optional();
    }
    Optional variable_feasible_value(int j) const { return feasible_value(variable_bound(j, LOWER), variable_bound(j, UPPER)); }
    static BoundChange compare_and_update(const Key& key, const Num& fresh, const Num& existing, std::map<Key, Num>& table) {
        BoundChange out;  // updates.rs:336-357
        if ((key.second == LOWER && fresh > existing) || (key.second == UPPER && fresh < existing)) {
            out.type = BoundChange::SHIFT;
            out.shift = fresh - existing;
            table[key] = fresh;
        }
        return out;
    }
    BoundChange update_bound(int j, Direction d, const Num& fresh) {  // updates.rs:175-210
        const Key key{j, d};
        Num compare_with;
        auto it = bounds.find(key);
        if (it != bounds.end()) {
            compare_with = it->second;
        } else {
            auto act = activity_variable_bounds.find(key);
            if (act != activity_variable_bounds.end()) {
                compare_with = act->second;
                bounds[key] = act->second;
                activity_variable_bounds.erase(act);
            } else {
                Optional orig = original(j, d);
                if (!orig.has) {
                    bounds[key] = fresh;
                    BoundChange out;
                    out.type = BoundChange::NEW_BOUND;
                    return out;
                }
                compare_with = orig.value;
            }
        }
        return compare_and_update(key, fresh, compare_with, bounds);
    }
    // 0: none (the reference's behaviour).  Otherwise a bound that domain propagation IMPLIES from a constraint which stays in the
    // problem is not recorded when its numerator or denominator needs more bits than this: skipping an implied tightening keeps
    // every later step valid (nothing was removed because of it), and it is how a presolved LP stays inside the fixed-width
    // host model when the exact tightenings of a few variables run to thousands of bits (BORE3D, CYCLE, GREENBEB).
    size_t activity_bound_bit_limit = 0;
    BoundChange update_activity_variable_bound(int j, Direction d, const Num& fresh) {  // updates.rs:212-253
        if (activity_bound_bit_limit > 0 && fresh.bits() > activity_bound_bit_limit) return BoundChange();
        const Key key{j, d};
        auto act = activity_variable_bounds.find(key);
        if (act != activity_variable_bounds.end()) return compare_and_update(key, fresh, Num(act->second), activity_variable_bounds);
        auto it = bounds.find(key);
        if (it != bounds.end()) return compare_and_update(key, fresh, Num(it->second), bounds);
        Optional orig = original(j, d);
        if (!orig.has) {
            activity_variable_bounds[key] = fresh;
            BoundChange out;
            out.type = BoundChange::NEW_BOUND;
            return out;
        }
        return compare_and_update(key, fresh, orig.value, activity_variable_bounds);
    }
    RemovedVariable optimize_column_independently(int j) {  // updates.rs:255-276
        RemovedVariable solved;
        solved.constant = optimize_independent_column(gf.maximize, gf.variables[j].cost, variable_bound(j, LOWER), variable_bound(j, UPPER));
        fixed_cost = fixed_cost + gf.variables[j].cost * solved.constant;
        return solved;
    }
    size_t nr_variables_remaining() const { return gf.variables.size() - removed_variables.size(); }
    size_t nr_constraints_remaining() const { return gf.b.size() - constraints_marked_removed.size(); }
    bool queues_empty() const { return q_activity.empty() && q_slack.empty() && q_bound.empty() && q_substitution.empty(); }
    void push_activity(int constraint, Direction d) {
        if (q_activity_members.insert({constraint, d}).second) q_activity.push_back({constraint, d});
    }

    // ---- presolve/mod.rs:129-167 ---------------------------------------------------------------------
    Change presolve_step() {
        if (!q_substitution.empty()) {
            const int variable = q_substitution.back();
            q_substitution.pop_back();
            if (variable_active(variable)) {
                presolve_fixed_variable(variable);
                return MEANINGFUL;
            }
        }
        while (!q_bound.empty()) {
            const int constraint = q_bound.back();
            q_bound.pop_back();
            if (constraint_active(constraint)) {
                presolve_bound_constraint(constraint);
                return MEANINGFUL;
            }
        }
        while (!q_slack.empty()) {
            const int variable = q_slack.back();
            q_slack.pop_back();
            if (variable_active(variable)) {
                presolve_slack(variable);
                return MEANINGFUL;
            }
        }
        while (!q_activity.empty()) {
            const Key item = q_activity.front();
            q_activity.pop_front();
            q_activity_members.erase(item);
            if (constraint_active(item.first)) return presolve_domain_propagation(item.first, (Direction)item.second);
        }
        return NOT_MEANINGFUL;
    }

    // ---- presolve/mod.rs:182-268 ---------------------------------------------------------------------
    void after_bound_change(int variable, Direction d, const Optional& change) {
        if (is_variable_fixed(variable).has && variable_active(variable)) q_substitution.push_back(variable);
        if (change.has) update_activity_bounds(variable, d, change.value);
        else update_activity_counters(variable, d);
    }
    void update_activity_bounds(int variable, Direction d, const Num& by_how_much) {
        for (auto& [row, coefficient] : active_column(variable)) {
            const Direction edit = times_sign(d, coefficient);
            Optional& bound = edit == LOWER ? activity_bounds[row].first : activity_bounds[row].second;
            if (bound.has) {
                bound.value = bound.value + by_how_much * coefficient;
                push_activity(row, edit);
            }
        }
    }
    void update_activity_counters(int variable, Direction d) {
        for (auto& [constraint, coefficient] : active_column(variable)) {
            const Direction activity_direction = times_sign(d, coefficient);
            int& counter = activity_direction == LOWER ? count_activity[constraint].first : count_activity[constraint].second;
            counter -= 1;
            if (counter <= 1) push_activity(constraint, activity_direction);
        }
    }

    // ---- presolve/mod.rs:279-374 ---------------------------------------------------------------------
    void remove_constraint_values(int constraint) {
        for (auto& [variable, coefficient] : active_row(constraint)) {
            (void)coefficient;
            count_constraint[constraint] -= 1;
            count_variable[variable] -= 1;
            queue_variable_by_counter(variable);
        }
    }
    void queue_variable_by_counter(int variable) {
        const int count = count_variable[variable];
        if (count == 0) {
            RemovedVariable value;
            if (gf.variables[variable].cost.is_zero()) value.constant = variable_feasible_value(variable).value;
            else value = optimize_column_independently(variable);
            removed_variables.push_back({variable, value});
        } else if (count == 1 && gf.variables[variable].cost.is_zero()) {
            q_slack.push_back(variable);
        }
    }
    Change queue_constraint_by_counter(int constraint) {
        const int count = count_constraint[constraint];
        if (count == 0) {
            if (!is_empty_constraint_feasible(b(constraint), constraint_type(constraint))) throw PresolveInfeasible();
            constraints_marked_removed.push_back(constraint);
            return MEANINGFUL;
        }
        if (count == 1) q_bound.push_back(constraint);
        return NO_CHANGE;
    }

    // ---- rule/fixed_variable.rs ------------------------------------------------------------------------
    void presolve_fixed_variable(int variable) {
        const Num value = is_variable_fixed(variable).value;
        const auto column = active_column(variable);
        for (auto& [constraint, coefficient] : column) change_b(constraint, -(coefficient * value));
        fixed_cost = fixed_cost + gf.variables[variable].cost * value;
        for (auto& [constraint, coefficient] : column) {
            (void)coefficient;
            count_variable[variable] -= 1;
            count_constraint[constraint] -= 1;
            queue_constraint_by_counter(constraint);
        }
        RemovedVariable solved;
        solved.constant = value;
        removed_variables.push_back({variable, solved});
    }

    // ---- rule/bound_constraint.rs ----------------------------------------------------------------------
    void presolve_bound_constraint(int constraint) {
        const auto row = active_row(constraint);
        const int variable = row[0].first;
        const Num coefficient = row[0].second;
        const Num bound_value = b(constraint) / coefficient;
        const ConstraintKind kind = constraint_type(constraint);
        const bool positive = coefficient.sign() > 0;
        std::vector<std::pair<Direction, Num>> changes;
        if ((kind.kind == GREATER && positive) || (kind.kind == LESS && !positive)) changes.push_back({LOWER, bound_value});
        else if ((kind.kind == LESS && positive) || (kind.kind == GREATER && !positive)) changes.push_back({UPPER, bound_value});
        else if (kind.kind == EQUAL) { changes.push_back({LOWER, bound_value}); changes.push_back({UPPER, bound_value}); }
        else {
            const Num bound1 = (b(constraint) - kind.range) / coefficient;
            if (positive) { changes.push_back({LOWER, bound1}); changes.push_back({UPPER, bound_value}); }
            else { changes.push_back({LOWER, bound_value}); changes.push_back({UPPER, bound1}); }
        }
        count_variable[variable] -= 1;
        count_constraint[constraint] -= 1;
        constraints_marked_removed.push_back(constraint);
        for (auto& [d, value] : changes) {
            BoundChange change = update_bound(variable, d, value);
            if (change.type == BoundChange::NEW_BOUND) after_bound_change(variable, d, Optional());
            else if (change.type == BoundChange::SHIFT) after_bound_change(variable, d, Optional(change.shift));
        }
        if (!variable_feasible_value(variable).has) throw PresolveInfeasible();
        queue_variable_by_counter(variable);
    }

    // ---- rule/slack.rs -------------------------------------------------------------------------------------
    RemovedVariable compute_removed_variable_solution(int constraint, int variable, const Num& coefficient) const {
        RemovedVariable out;  // slack.rs:149-165
        out.function_of_others = true;
        out.constant = b(constraint) / coefficient;
        for (auto& [j, other] : active_row(constraint))
            if (j != variable) out.coefficients.push_back({gf.active_to_original[j], other / coefficient});
        return out;
    }
    void presolve_slack(int variable) {
        const auto column = active_column(variable);
        const int constraint = column[0].first;
        const Num coefficient = column[0].second;
        const ConstraintKind kind = constraint_type(constraint);
        const Optional lower = variable_bound(variable, LOWER), upper = variable_bound(variable, UPPER);
        const bool positive = coefficient.sign() > 0;
        const bool hl = lower.has, hu = upper.has;
        const RowKind k = kind.kind;
        const bool removable = (k == GREATER && hl && !hu && positive) || (k == LESS && !hl && hu && positive) ||
                               (k == LESS && hl && !hu && !positive) || (k == GREATER && !hl && hu && !positive) || (!hl && !hu);
        if (removable) {  // slack.rs:46-66
            RemovedVariable solution = compute_removed_variable_solution(constraint, variable, coefficient);
            for (auto& [other, c] : active_row(constraint)) {
                (void)c;
                count_constraint[constraint] -= 1;
                count_variable[other] -= 1;
                if (other != variable) queue_variable_by_counter(other);
            }
            removed_variables.push_back({variable, solution});
            constraints_marked_removed.push_back(constraint);
            return;
        }
        ConstraintKind fresh;
        Num bound;
        if (k == EQUAL && hl && hu) {
            fresh.kind = RANGE;
            fresh.range = positive ? coefficient * (upper.value - lower.value) : coefficient * (lower.value - upper.value);
            bound = positive ? lower.value : upper.value;
        } else if (k == RANGE && hl && hu) {
            fresh.kind = RANGE;
            fresh.range = kind.range + (positive ? coefficient * (upper.value - lower.value) : coefficient * (lower.value - upper.value));
            bound = positive ? lower.value : upper.value;
        } else if (positive && ((hl && !hu && (k == LESS || k == EQUAL || k == RANGE)) || (k == LESS && hl && hu))) {
            fresh.kind = LESS;
            bound = lower.value;
        } else if (positive && ((!hl && hu && (k == EQUAL || k == GREATER || k == RANGE)) || (k == GREATER && hl && hu))) {
            fresh.kind = GREATER;
            bound = upper.value;
        } else if (!positive && ((hl && !hu && (k == EQUAL || k == GREATER || k == RANGE)) || (k == GREATER && hl && hu))) {
            fresh.kind = GREATER;
            bound = lower.value;
        } else if (!positive && ((!hl && hu && (k == LESS || k == EQUAL || k == RANGE)) || (k == LESS && hl && hu))) {
            fresh.kind = LESS;
            bound = upper.value;
        } else {
            throw std::logic_error("presolve_slack: unreachable combination");
        }
        const Num change = -(coefficient * bound);
        RemovedVariable removed;
        if (k == EQUAL || k == RANGE) removed = compute_removed_variable_solution(constraint, variable, coefficient);
        else removed.constant = bound;
        count_variable[variable] -= 1;
        removed_variables.push_back({variable, removed});
        // slack.rs:127-147
        if ((!hl && positive) || (!hu && !positive)) {
            count_activity[constraint].first -= 1;
            if (count_activity[constraint].first <= 1) push_activity(constraint, LOWER);
        }
        if ((!hu && positive) || (!hl && !positive)) {
            count_activity[constraint].second -= 1;
            if (count_activity[constraint].second <= 1) push_activity(constraint, UPPER);
        }
        count_constraint[constraint] -= 1;
        queue_constraint_by_counter(constraint);
        change_b(constraint, change);
        constraint_changes[constraint] = fresh;
    }

    // ---- rule/domain_propagation.rs --------------------------------------------------------------------------
    Change presolve_domain_propagation(int constraint, Direction d) {
        const int counter = d == LOWER ? count_activity[constraint].first : count_activity[constraint].second;
        if (counter == 0) return for_entire_constraint(constraint, d);
        if (counter == 1) return create_variable_bound(constraint, d);
        throw std::logic_error("activity queue entry with more than one missing bound");
    }
    Optional can_variable_rule_be_applied(int constraint, Direction activity_direction) const {
        const Num rhs = b(constraint);
        const ConstraintKind& kind = constraint_type(constraint);
        switch (kind.kind) {
            case EQUAL: return Optional(rhs);
            case LESS: return activity_direction == LOWER ? Optional(rhs) : Optional();
            case GREATER: return activity_direction == LOWER ? Optional() : Optional(rhs);
            default: return activity_direction == LOWER ? Optional(rhs) : Optional(rhs - kind.range);
        }
    }
    Num compute_activity_bound_if_needed(int constraint, Direction d) {
        Optional& bound = d == LOWER ? activity_bounds[constraint].first : activity_bounds[constraint].second;
        if (!bound.has) {
            Num total(0);
            for (auto& [j, c] : active_row(constraint)) total = total + c * variable_bound(j, times_sign(d, c)).value;
            bound = Optional(total);
        }
        return bound.value;
    }
    enum UpdateType { U_NONE, U_REMOVE, U_REPLACE, U_SET_TO_BOUND };
    struct ConstraintUpdate {
        UpdateType type = U_NONE;
        RowKind replace_with = LESS;
        Num shift;
    };
    ConstraintUpdate constraint_update(int constraint, const Num& bound_value, Direction d) const {  // :166-228
        const Num rhs = b(constraint);
        const ConstraintKind& kind = constraint_type(constraint);
        const RowKind k = kind.kind;
        const int order = (rhs > bound_value) - (rhs < bound_value);
        ConstraintUpdate out;
        if (d == LOWER) {
            if (order < 0 && (k == EQUAL || k == LESS || k == RANGE)) throw PresolveInfeasible();
            if (order == 0 && (k == EQUAL || k == LESS)) { out.type = U_SET_TO_BOUND; return out; }
            if (k == GREATER && order <= 0) { out.type = U_REMOVE; return out; }
            if (k == RANGE && order > 0) {
                if (bound_value >= rhs - kind.range) { out.type = U_REPLACE; out.replace_with = LESS; out.shift = Num(0); }
                return out;
            }
            if (k == RANGE && order == 0) throw std::logic_error("range of zero");
            return out;
        }
        if (order > 0 && (k == EQUAL || k == GREATER)) throw PresolveInfeasible();
        if (order == 0 && (k == EQUAL || k == GREATER)) { out.type = U_SET_TO_BOUND; return out; }
        if (k == LESS && order >= 0) { out.type = U_REMOVE; return out; }
        if (k == RANGE && order == 0) { out.type = U_REPLACE; out.replace_with = GREATER; out.shift = -kind.range; return out; }
        if (k == RANGE && order > 0) {
            const Num lower_bound = rhs - kind.range;
            if (bound_value < lower_bound) throw PresolveInfeasible();
            if (bound_value == lower_bound) { out.type = U_SET_TO_BOUND; return out; }
            out.type = U_REPLACE; out.replace_with = GREATER; out.shift = -kind.range;
            return out;
        }
        return out;
    }
    Change for_entire_constraint(int constraint, Direction d) {
        Change change = NO_CHANGE;
        const Num activity_bound = compute_activity_bound_if_needed(constraint, d);
        bool remove_constraint = false, apply_variable_part = true;
        const ConstraintUpdate update = constraint_update(constraint, activity_bound, d);
        if (update.type != U_NONE) {
            if (update.type == U_REMOVE) {
                remove_constraint = true;
            } else if (update.type == U_SET_TO_BOUND) {
                std::vector<std::pair<int, Direction>> to_update;
                for (auto& [variable, coefficient] : active_row(constraint)) {
                    const Direction variable_direction = times_sign(d, coefficient);
                    const Num value = variable_bound(variable, variable_direction).value;
                    auto act = activity_variable_bounds.find({variable, variable_direction});
                    if (act != activity_variable_bounds.end()) {
                        bounds[{variable, variable_direction}] = act->second;
                        activity_variable_bounds.erase(act);
                    }
                    if (update_bound(variable, flip(variable_direction), value).type == BoundChange::NEW_BOUND)
                        to_update.push_back({variable, flip(variable_direction)});
                    q_substitution.push_back(variable);
                }
                for (auto& [variable, direction] : to_update) update_activity_counters(variable, direction);
                remove_constraint = true;
                apply_variable_part = false;
            } else {
                ConstraintKind fresh;
                fresh.kind = update.replace_with;
                constraint_changes[constraint] = fresh;
                change_b(constraint, update.shift);
            }
            change = MEANINGFUL;
        }
        if (apply_variable_part) {
            Optional rhs = can_variable_rule_be_applied(constraint, d);
            if (rhs.has) variable_part(constraint, rhs.value, activity_bound, d, change);
        }
        if (remove_constraint) {
            remove_constraint_values(constraint);
            constraints_marked_removed.push_back(constraint);
        }
        return change;
    }
    void variable_part(int constraint, const Num& rhs, const Num& activity_bound, Direction activity_direction, Change& change) {
        for (auto& [variable, coefficient] : active_row(constraint)) {
            const Direction fresh_direction = times_sign(flip(activity_direction), coefficient);
            const Num value = variable_bound(variable, times_sign(activity_direction, coefficient)).value;
            const Num residual = activity_bound - coefficient * value;
            const Num fresh_value = (rhs - residual) / coefficient;
            BoundChange result = update_activity_variable_bound(variable, fresh_direction, fresh_value);
            if (result.type == BoundChange::NEW_BOUND) {
                after_bound_change(variable, fresh_direction, Optional());
                change = MEANINGFUL;
            } else if (result.type == BoundChange::SHIFT) {
                after_bound_change(variable, fresh_direction, Optional(result.shift));
                if (change != MEANINGFUL) change = NOT_MEANINGFUL;
            }
        }
    }
    Change create_variable_bound(int constraint, Direction activity_direction) {
        Optional rhs = can_variable_rule_be_applied(constraint, activity_direction);
        if (!rhs.has) return NO_CHANGE;
        Num total(0);
        int target = -1;
        Num target_coefficient;
        for (auto& [variable, coefficient] : active_row(constraint)) {
            Optional bound = variable_bound(variable, times_sign(activity_direction, coefficient));
            if (!bound.has) {
                if (target < 0) { target = variable; target_coefficient = coefficient; }
            } else {
                total = total + coefficient * bound.value;
            }
        }
        const Num value = (rhs.value - total) / target_coefficient;
        const Direction bound_direction = times_sign(flip(activity_direction), target_coefficient);
        BoundChange result = update_activity_variable_bound(target, bound_direction, value);
        if (result.type == BoundChange::NONE) return NO_CHANGE;
        if (result.type == BoundChange::NEW_BOUND) {
            after_bound_change(target, bound_direction, Optional());
            return MEANINGFUL;
        }
        after_bound_change(target, bound_direction, Optional(result.shift));
        return NOT_MEANINGFUL;
    }
};

}  // namespace presolve_detail

// `GeneralForm::presolve` (general_form/mod.rs:335-463): compute the changes, apply them, drop rows and columns.
// Throws PresolveInfeasible / PresolveUnbounded.
inline void presolve(GeneralProblem& gf, size_t activity_bound_bit_limit = 0) {
    using namespace presolve_detail;
    std::map<int, Num> b_changes;
    std::map<int, ConstraintKind> constraint_changes;
    std::map<Index::Key, Num> bounds;
    std::vector<std::pair<int, RemovedVariable>> removed_variables;
    std::vector<int> removed_rows;
    Num fixed_cost;
    {
        Index index(gf);  // compute_presolve_changes (mod.rs:360-386)
        index.activity_bound_bit_limit = activity_bound_bit_limit;
        size_t without_change = 0;
        while (!index.queues_empty() && without_change < index.nr_variables_remaining() + index.nr_constraints_remaining()) {
            const Change change = index.presolve_step();
            if (change == MEANINGFUL) without_change = 0;
            else if (change == NOT_MEANINGFUL) without_change += 1;
        }
        // Updates::into_changes (updates.rs:278-323)
        std::set<int> rows_gone(index.constraints_marked_removed.begin(), index.constraints_marked_removed.end());
        std::set<int> vars_gone;
        for (auto& rv : index.removed_variables) vars_gone.insert(rv.first);
        for (auto& [i, v] : index.b_changes)
            if (!rows_gone.count(i) && v != gf.b[i]) b_changes[i] = v;
        for (auto& [i, k] : index.constraint_changes)
            if (!rows_gone.count(i) && k != gf.kinds[i]) constraint_changes[i] = k;
        for (auto& [key, v] : index.bounds)
            if (!vars_gone.count(key.first)) bounds[key] = v;
        std::set<int> restrict;
        for (auto& [key, v] : index.activity_variable_bounds) {
            (void)v;
            if (vars_gone.count(key.first)) continue;
            const PVariable& var = gf.variables[key.first];
            if (!var.has_lower && !var.has_upper && !bounds.count({key.first, LOWER}) && !bounds.count({key.first, UPPER}))
                restrict.insert(key.first);
        }
        for (auto& [key, v] : index.activity_variable_bounds)
            if (restrict.count(key.first)) bounds[key] = v;
        removed_variables = index.removed_variables;
        removed_rows = index.constraints_marked_removed;
        fixed_cost = index.fixed_cost;
    }
    // update_values_that_remain (mod.rs:388-421)
    for (auto& [i, v] : b_changes) gf.b[i] = v;
    for (auto& [i, k] : constraint_changes) gf.kinds[i] = k;
    gf.fixed_cost = gf.fixed_cost + fixed_cost;
    for (auto& [j, solution] : removed_variables) gf.removed[gf.active_to_original[j]] = solution;
    for (auto& [key, v] : bounds) {
        PVariable& var = gf.variables[key.first];
        if (key.second == LOWER) { var.has_lower = true; var.lower = v; }
        else { var.has_upper = true; var.upper = v; }
    }
    // remove_rows_and_columns (mod.rs:423-463)
    std::set<int> vars_gone;
    for (auto& rv : removed_variables) vars_gone.insert(rv.first);
    std::set<int> rows_gone(removed_rows.begin(), removed_rows.end());
    std::vector<int> new_row(gf.b.size(), -1);
    int next = 0;
    for (size_t i = 0; i < gf.b.size(); ++i)
        if (!rows_gone.count((int)i)) new_row[i] = next++;
    std::vector<PColumn> columns;
    std::vector<PVariable> variables;
    std::vector<int> active_to_original;
    for (size_t j = 0; j < gf.variables.size(); ++j) {
        if (vars_gone.count((int)j)) continue;
        PColumn column;
        for (size_t k = 0; k < gf.columns[j].nnz(); ++k)
            if (new_row[gf.columns[j].index[k]] >= 0) column.push(new_row[gf.columns[j].index[k]], gf.columns[j].value[k]);
        columns.push_back(column);
        variables.push_back(gf.variables[j]);
        active_to_original.push_back(gf.active_to_original[j]);
    }
    std::vector<Num> b;
    std::vector<ConstraintKind> kinds;
    for (size_t i = 0; i < gf.b.size(); ++i)
        if (new_row[i] >= 0) { b.push_back(gf.b[i]); kinds.push_back(gf.kinds[i]); }
    gf.columns.swap(columns);
    gf.variables.swap(variables);
    gf.active_to_original.swap(active_to_original);
    gf.b.swap(b);
    gf.kinds.swap(kinds);
}

}  // namespace relp
