"""Presolve of a ``GeneralForm`` (oracle; test infrastructure only).

Follows ``data/linear_program/general_form/presolve/{mod.rs, counters.rs, queues.rs, updates.rs,
rule/{fixed_variable, bound_constraint, slack, domain_propagation}.rs}`` and the application of the changes in
``general_form/mod.rs:335-505``.  Directions are the strings ``"L"`` (lower) and ``"U"`` (upper); constraint types are
``"Equal" | "Less" | "Greater" | ("Range", r)`` as in ``mps.GeneralForm``.
"""
from collections import deque
from fractions import Fraction

ZERO = Fraction(0)

MEANINGFUL, NOT_MEANINGFUL, NONE = "Meaningful", "NotMeaningful", "None"


class Infeasible(Exception):
    """LinearProgramType::Infeasible raised during presolve."""


class Unbounded(Exception):
    """LinearProgramType::Unbounded raised during presolve."""


def flip(direction):
    return "U" if direction == "L" else "L"


def times_sign(direction, coefficient):
    """``BoundDirection * NonZeroSign`` (elements.rs): a negative coefficient swaps the direction."""
    return direction if coefficient > 0 else flip(direction)


def is_empty_constraint_feasible(rhs, kind):
    """presolve/mod.rs:403-425."""
    if kind == "Equal":
        return rhs == 0
    if kind == "Less":
        return rhs >= 0
    if kind == "Greater":
        return rhs <= 0
    return rhs >= 0 and rhs - kind[1] <= 0


def optimize_independent_column(objective, cost, lower, upper):
    """updates.rs:368-389."""
    if (objective == "Minimize" and cost > 0) or (objective == "Maximize" and cost < 0):
        bound = lower
    else:
        bound = upper
    if bound is None:
        raise Unbounded()
    return bound


class FIFOSet:
    """``fifo_set::FIFOSet`` (crate fifo-set 1.0.0, not vendored): a queue that holds each item at most once."""

    def __init__(self, items=()):
        self.queue = deque()
        self.members = set()
        for item in items:
            self.push(item)

    def push(self, item):
        if item not in self.members:
            self.members.add(item)
            self.queue.append(item)

    def pop(self):
        if not self.queue:
            return None
        item = self.queue.popleft()
        self.members.discard(item)
        return item

    def __bool__(self):
        return bool(self.queue)


class Index:
    """presolve/mod.rs:31-45 with Counters (counters.rs), Queues (queues.rs) and Updates (updates.rs) inlined."""

    def __init__(self, gf):
        self.gf = gf
        nr_rows, nr_vars = len(gf.b), len(gf.variables)
        # ---- counters.rs:33-62 ----
        self.rows = [[] for _ in range(nr_rows)]
        for j, column in enumerate(gf.columns):
            for i, value in column:
                self.rows[i].append((j, value))
        self.count_constraint = [len(r) for r in self.rows]
        self.count_variable = [len(c) for c in gf.columns]
        self.count_activity = []
        for row in self.rows:
            lower_missing = upper_missing = 0
            for j, coefficient in row:
                v = gf.variables[j]
                lower, upper = (v.lower_bound, v.upper_bound) if coefficient > 0 else (v.upper_bound, v.lower_bound)
                lower_missing += lower is None
                upper_missing += upper is None
            self.count_activity.append([lower_missing, upper_missing])
        # ---- updates.rs:44-98 ----
        self.b_changes = {}
        self.constraint_changes = {}
        self.fixed_cost = ZERO
        self.bounds = {}
        self.activity_variable_bounds = {}
        self.removed_variables = []
        for j, count in enumerate(self.count_variable):
            if count == 0:
                v = gf.variables[j]
                if v.cost == 0:
                    value = self._feasible_value(v.lower_bound, v.upper_bound)
                else:
                    value = optimize_independent_column(gf.objective, v.cost, v.lower_bound, v.upper_bound)
                    self.fixed_cost += v.cost * value
                self.removed_variables.append((j, ("Solved", value)))
        self.constraints_marked_removed = []
        for i, count in enumerate(self.count_constraint):
            if count == 0:
                if not is_empty_constraint_feasible(gf.b[i], gf.constraint_types[i]):
                    raise Infeasible()
                self.constraints_marked_removed.append(i)
        # ---- queues.rs:33-63 ----
        self.q_bound = [i for i, c in enumerate(self.count_constraint) if c == 1]
        activity = []
        for i, (lower_count, upper_count) in enumerate(self.count_activity):
            if self.count_constraint[i] > 1:
                if lower_count <= 1:
                    activity.append((i, "L"))
                if upper_count <= 1:
                    activity.append((i, "U"))
        self.q_activity = FIFOSet(activity)
        self.q_slack = [j for j, c in enumerate(self.count_variable) if c == 1 and gf.variables[j].cost == 0]
        self.q_substitution = [j for j, v in enumerate(gf.variables)
                               if self.count_variable[j] > 0 and v.lower_bound is not None and v.lower_bound == v.upper_bound]
        # presolve/mod.rs:40
        self.activity_bounds = [[None, None] for _ in range(nr_rows)]

    # ---- counters.rs:64-88 ---------------------------------------------------------------------------
    def constraint_active(self, i):
        return self.count_constraint[i] > 0

    def variable_active(self, j):
        return self.count_variable[j] > 0

    def active_column(self, j):
        return [(i, v) for i, v in self.gf.columns[j] if self.constraint_active(i)]

    def active_row(self, i):
        return [(j, v) for j, v in self.rows[i] if self.variable_active(j)]

    # ---- updates.rs accessors ------------------------------------------------------------------------
    @staticmethod
    def _feasible_value(lower, upper):
        """general_form Variable::get_feasible_value / updates.rs:133-152."""
        if lower is None and upper is None:
            return ZERO
        if lower is None:
            return upper
        if upper is None:
            return lower
        return upper if lower <= upper else None

    def b(self, i):
        return self.b_changes.get(i, self.gf.b[i])

    def change_b(self, i, change):
        self.b_changes[i] = self.b(i) + change

    def constraint_type(self, i):
        return self.constraint_changes.get(i, self.gf.constraint_types[i])

    def variable_bound(self, j, direction):
        """updates.rs:154-173: activity-derived, then derived, then original."""
        key = (j, direction)
        if key in self.activity_variable_bounds:
            return self.activity_variable_bounds[key]
        if key in self.bounds:
            return self.bounds[key]
        v = self.gf.variables[j]
        return v.lower_bound if direction == "L" else v.upper_bound

    def is_variable_fixed(self, j):
        lower, upper = self.variable_bound(j, "L"), self.variable_bound(j, "U")
        return lower if lower is not None and upper is not None and lower == upper else None

    def variable_feasible_value(self, j):
        return self._feasible_value(self.variable_bound(j, "L"), self.variable_bound(j, "U"))

    @staticmethod
    def _compare_and_update(key, new, existing, table):
        """updates.rs:336-357: returns None | ("shift", difference)."""
        direction = key[1]
        if (direction == "L" and new > existing) or (direction == "U" and new < existing):
            table[key] = new
            return ("shift", new - existing)
        return None

    def update_bound(self, j, direction, new):
        """updates.rs:175-210.  Returns None | "new" | ("shift", d)."""
        key = (j, direction)
        if key in self.bounds:
            compare_with = self.bounds[key]
        elif key in self.activity_variable_bounds:
            compare_with = self.bounds[key] = self.activity_variable_bounds.pop(key)
        else:
            v = self.gf.variables[j]
            original = v.lower_bound if direction == "L" else v.upper_bound
            if original is None:
                self.bounds[key] = new
                return "new"
            compare_with = original
        return self._compare_and_update(key, new, compare_with, self.bounds)

    def update_activity_variable_bound(self, j, direction, new):
        """updates.rs:212-253."""
        key = (j, direction)
        if key in self.activity_variable_bounds:
            return self._compare_and_update(key, new, self.activity_variable_bounds[key], self.activity_variable_bounds)
        if key in self.bounds:
            return self._compare_and_update(key, new, self.bounds[key], self.bounds)
        v = self.gf.variables[j]
        original = v.lower_bound if direction == "L" else v.upper_bound
        if original is None:
            self.activity_variable_bounds[key] = new
            return "new"
        return self._compare_and_update(key, new, original, self.activity_variable_bounds)

    def optimize_column_independently(self, j):
        """updates.rs:255-276."""
        cost = self.gf.variables[j].cost
        value = optimize_independent_column(self.gf.objective, cost, self.variable_bound(j, "L"), self.variable_bound(j, "U"))
        self.fixed_cost += cost * value
        return ("Solved", value)

    def nr_variables_remaining(self):
        return len(self.gf.variables) - len(self.removed_variables)

    def nr_constraints_remaining(self):
        return len(self.gf.b) - len(self.constraints_marked_removed)

    def queues_empty(self):
        return not (self.q_activity or self.q_slack or self.q_bound or self.q_substitution)

    # ---- presolve/mod.rs:129-167 ------------------------------------------------------------------------
    def presolve_step(self):
        if self.q_substitution:
            variable = self.q_substitution.pop()
            if self.variable_active(variable):
                self.presolve_fixed_variable(variable)
                return MEANINGFUL
        while self.q_bound:
            constraint = self.q_bound.pop()
            if self.constraint_active(constraint):
                self.presolve_bound_constraint(constraint)
                return MEANINGFUL
        while self.q_slack:
            variable = self.q_slack.pop()
            if self.variable_active(variable):
                self.presolve_slack(variable)
                return MEANINGFUL
        while self.q_activity:
            constraint, direction = self.q_activity.pop()
            if self.constraint_active(constraint):
                assert self.count_constraint[constraint] > 1
                return self.presolve_domain_propagation(constraint, direction)
        return NOT_MEANINGFUL

    # ---- presolve/mod.rs:182-268 ------------------------------------------------------------------------
    def after_bound_change(self, variable, direction, change):
        if self.is_variable_fixed(variable) is not None and self.variable_active(variable):
            self.q_substitution.append(variable)
        if change is not None:
            self.update_activity_bounds(variable, direction, change)
        else:
            self.update_activity_counters(variable, direction)

    def update_activity_bounds(self, variable, direction, by_how_much):
        for row, coefficient in self.active_column(variable):
            bound_to_edit = times_sign(direction, coefficient)
            slot = 0 if bound_to_edit == "L" else 1
            if self.activity_bounds[row][slot] is not None:
                self.activity_bounds[row][slot] += by_how_much * coefficient
                self.q_activity.push((row, bound_to_edit))

    def update_activity_counters(self, variable, direction):
        for constraint, coefficient in self.active_column(variable):
            activity_direction = times_sign(direction, coefficient)
            slot = 0 if activity_direction == "L" else 1
            self.count_activity[constraint][slot] -= 1
            if self.count_activity[constraint][slot] <= 1:
                self.q_activity.push((constraint, activity_direction))

    # ---- presolve/mod.rs:279-374 ------------------------------------------------------------------------
    def remove_constraint_values(self, constraint):
        for variable, _ in self.active_row(constraint):
            self.count_constraint[constraint] -= 1
            self.count_variable[variable] -= 1
            self.queue_variable_by_counter(variable)
        assert self.count_constraint[constraint] == 0

    def queue_variable_by_counter(self, variable):
        count = self.count_variable[variable]
        if count == 0:
            if self.gf.variables[variable].cost == 0:
                value = ("Solved", self.variable_feasible_value(variable))
            else:
                value = self.optimize_column_independently(variable)
            self.remove_variable(variable, value)
        elif count == 1 and self.gf.variables[variable].cost == 0:
            self.q_slack.append(variable)

    def queue_constraint_by_counter(self, constraint):
        count = self.count_constraint[constraint]
        if count == 0:
            if not is_empty_constraint_feasible(self.b(constraint), self.constraint_type(constraint)):
                raise Infeasible()
            self.remove_constraint(constraint)
            return MEANINGFUL
        if count == 1:
            self.q_bound.append(constraint)
        return NONE

    def remove_constraint(self, constraint):
        assert self.count_constraint[constraint] == 0
        self.constraints_marked_removed.append(constraint)

    def remove_variable(self, variable, solution):
        assert self.count_variable[variable] == 0
        self.removed_variables.append((variable, solution))

    # ---- rule/fixed_variable.rs ---------------------------------------------------------------------------
    def presolve_fixed_variable(self, variable):
        value = self.is_variable_fixed(variable)
        column = self.active_column(variable)
        for constraint, coefficient in column:
            self.change_b(constraint, -coefficient * value)
        self.fixed_cost += self.gf.variables[variable].cost * value
        for constraint, _ in column:
            self.count_variable[variable] -= 1
            self.count_constraint[constraint] -= 1
            self.queue_constraint_by_counter(constraint)
        self.remove_variable(variable, ("Solved", value))

    # ---- rule/bound_constraint.rs -------------------------------------------------------------------------
    def presolve_bound_constraint(self, constraint):
        (variable, coefficient), = self.active_row(constraint)
        bound_value = self.b(constraint) / coefficient
        kind = self.constraint_type(constraint)
        positive = coefficient > 0
        if (kind == "Greater" and positive) or (kind == "Less" and not positive):
            changes = [("L", bound_value)]
        elif (kind == "Less" and positive) or (kind == "Greater" and not positive):
            changes = [("U", bound_value)]
        elif kind == "Equal":
            changes = [("L", bound_value), ("U", bound_value)]
        else:
            bound1 = (self.b(constraint) - kind[1]) / coefficient
            changes = [("L", bound1), ("U", bound_value)] if positive else [("L", bound_value), ("U", bound1)]
        self.count_variable[variable] -= 1
        self.count_constraint[constraint] -= 1
        self.remove_constraint(constraint)
        for direction, value in changes:
            change = self.update_bound(variable, direction, value)
            if change == "new":
                self.after_bound_change(variable, direction, None)
            elif change is not None:
                self.after_bound_change(variable, direction, change[1])
        if self.variable_feasible_value(variable) is None:
            raise Infeasible()
        self.queue_variable_by_counter(variable)

    # ---- rule/slack.rs ----------------------------------------------------------------------------------------
    def presolve_slack(self, variable):
        (constraint, coefficient), = self.active_column(variable)
        kind = self.constraint_type(constraint)
        lower, upper = self.variable_bound(variable, "L"), self.variable_bound(variable, "U")
        none = (lower is None, upper is None)
        positive = coefficient > 0
        is_range = isinstance(kind, tuple)
        has = (lower is not None, upper is not None)
        removable = (
            (kind == "Greater" and has == (True, False) and positive) or (kind == "Less" and has == (False, True) and positive)
            or (kind == "Less" and has == (True, False) and not positive) or (kind == "Greater" and has == (False, True) and not positive)
            or has == (False, False))
        if removable:  # slack.rs:46-66: the constraint can always be satisfied through this variable
            solution = self.compute_removed_variable_solution(constraint, variable, coefficient)
            for other, _ in self.active_row(constraint):
                self.count_constraint[constraint] -= 1
                self.count_variable[other] -= 1
                if other != variable:
                    self.queue_variable_by_counter(other)
            self.remove_variable(variable, solution)
            self.remove_constraint(constraint)
            return
        if kind == "Equal" and has == (True, True):
            new_kind, bound = (("Range", coefficient * (upper - lower)), lower) if positive else (("Range", coefficient * (lower - upper)), upper)
        elif is_range and has == (True, True):
            new_kind, bound = ((("Range", kind[1] + coefficient * (upper - lower)), lower) if positive
                               else (("Range", kind[1] + coefficient * (lower - upper)), upper))
        elif positive and ((has == (True, False) and (kind in ("Less", "Equal") or is_range)) or (kind == "Less" and has == (True, True))):
            new_kind, bound = "Less", lower
        elif positive and ((has == (False, True) and (kind in ("Equal", "Greater") or is_range)) or (kind == "Greater" and has == (True, True))):
            new_kind, bound = "Greater", upper
        elif not positive and ((has == (True, False) and (kind in ("Equal", "Greater") or is_range)) or (kind == "Greater" and has == (True, True))):
            new_kind, bound = "Greater", lower
        elif not positive and ((has == (False, True) and (kind in ("Less", "Equal") or is_range)) or (kind == "Less" and has == (True, True))):
            new_kind, bound = "Less", upper
        else:
            raise AssertionError("slack.rs match is exhaustive")
        change = -coefficient * bound
        if kind == "Equal" or is_range:
            removed = self.compute_removed_variable_solution(constraint, variable, coefficient)
        else:
            removed = ("Solved", bound)
        self.count_variable[variable] -= 1
        self.remove_variable(variable, removed)
        self.update_activity_queues_if_needed(constraint, none, positive)
        self.count_constraint[constraint] -= 1
        self.queue_constraint_by_counter(constraint)
        self.change_b(constraint, change)
        self.constraint_changes[constraint] = new_kind

    def update_activity_queues_if_needed(self, constraint, none, positive):
        """slack.rs:127-147."""
        lower_none, upper_none = none
        if (lower_none and positive) or (upper_none and not positive):
            self.count_activity[constraint][0] -= 1
            if self.count_activity[constraint][0] <= 1:
                self.q_activity.push((constraint, "L"))
        if (upper_none and positive) or (lower_none and not positive):
            self.count_activity[constraint][1] -= 1
            if self.count_activity[constraint][1] <= 1:
                self.q_activity.push((constraint, "U"))

    def compute_removed_variable_solution(self, constraint, variable, coefficient):
        """slack.rs:149-165: x = constant - sum coefficients_k x_k, in ORIGINAL variable indices."""
        constant = self.b(constraint) / coefficient
        coefficients = [(self.gf.active_to_original[j], other / coefficient)
                        for j, other in self.active_row(constraint) if j != variable]
        return ("FunctionOfOthers", constant, coefficients)

    # ---- rule/domain_propagation.rs ------------------------------------------------------------------------------
    def presolve_domain_propagation(self, constraint, direction):
        counter = self.count_activity[constraint][0 if direction == "L" else 1]
        missing = sum(1 for j, c in self.active_row(constraint) if self.variable_bound(j, times_sign(direction, c)) is None)
        assert missing == counter, (missing, counter)
        if counter == 0:
            return self.for_entire_constraint(constraint, direction)
        assert counter == 1
        return self.create_variable_bound(constraint, direction)

    def for_entire_constraint(self, constraint, direction):
        change = [NONE]
        activity_bound = self.compute_activity_bound_if_needed(constraint, direction)
        remove_constraint, apply_variable_part = self.constraint_part(constraint, activity_bound, direction, change)
        if apply_variable_part:
            rhs = self.can_variable_rule_be_applied(constraint, direction)
            if rhs is not None:
                self.variable_part(constraint, rhs, activity_bound, direction, change)
        if remove_constraint:
            self.remove_constraint_values(constraint)
            self.remove_constraint(constraint)
        return change[0]

    def compute_activity_bound_if_needed(self, constraint, direction):
        slot = 0 if direction == "L" else 1
        if self.activity_bounds[constraint][slot] is None:
            self.activity_bounds[constraint][slot] = sum(
                (c * self.variable_bound(j, times_sign(direction, c)) for j, c in self.active_row(constraint)), ZERO)
        return self.activity_bounds[constraint][slot]

    def constraint_part(self, constraint, bound, direction, change):
        update = self.constraint_update(constraint, bound, direction)
        if update is None:
            return False, True
        if update == "Remove":
            result = (True, True)
        elif update == "SetVariablesToBound":
            to_update = []
            for variable, coefficient in self.active_row(constraint):
                variable_direction = times_sign(direction, coefficient)
                value = self.variable_bound(variable, variable_direction)
                key = (variable, variable_direction)
                if key in self.activity_variable_bounds:
                    self.bounds[key] = self.activity_variable_bounds.pop(key)
                if self.update_bound(variable, flip(variable_direction), value) == "new":
                    to_update.append((variable, flip(variable_direction)))
                assert self.is_variable_fixed(variable) is not None
                self.q_substitution.append(variable)
            for variable, d in to_update:
                self.update_activity_counters(variable, d)
            result = (True, False)
        else:
            _, new_inequality, shift = update
            self.constraint_changes[constraint] = new_inequality
            self.change_b(constraint, shift)
            result = (False, True)
        change[0] = MEANINGFUL
        return result

    def constraint_update(self, constraint, bound_value, direction):
        """domain_propagation.rs:166-228."""
        rhs = self.b(constraint)
        kind = self.constraint_type(constraint)
        is_range = isinstance(kind, tuple)
        order = (rhs > bound_value) - (rhs < bound_value)  # rhs.cmp(bound_value)
        if direction == "L":
            if order < 0 and (kind in ("Equal", "Less") or is_range):
                raise Infeasible()
            if order == 0 and kind in ("Equal", "Less"):
                return "SetVariablesToBound"
            if kind == "Greater" and order <= 0:
                return "Remove"
            if is_range and order > 0:
                return ("Replace", "Less", ZERO) if bound_value >= rhs - kind[1] else None
            if is_range and order == 0:
                raise AssertionError("range of zero")
            return None
        if order > 0 and kind in ("Equal", "Greater"):
            raise Infeasible()
        if order == 0 and kind in ("Equal", "Greater"):
            return "SetVariablesToBound"
        if kind == "Less" and order >= 0:
            return "Remove"
        if is_range and order == 0:
            return ("Replace", "Greater", -kind[1])
        if is_range and order > 0:
            lower_bound = rhs - kind[1]
            if bound_value < lower_bound:
                raise Infeasible()
            if bound_value == lower_bound:
                return "SetVariablesToBound"
            return ("Replace", "Greater", -kind[1])
        return None

    def variable_part(self, constraint, rhs, activity_bound, activity_direction, change):
        for variable, coefficient in self.active_row(constraint):
            new_direction = times_sign(flip(activity_direction), coefficient)
            value = self.variable_bound(variable, times_sign(activity_direction, coefficient))
            residual = activity_bound - coefficient * value
            new_value = (rhs - residual) / coefficient
            result = self.update_activity_variable_bound(variable, new_direction, new_value)
            if result == "new":
                self.after_bound_change(variable, new_direction, None)
                change[0] = MEANINGFUL
            elif result is not None:
                self.after_bound_change(variable, new_direction, result[1])
                if change[0] != MEANINGFUL:
                    change[0] = NOT_MEANINGFUL

    def create_variable_bound(self, constraint, activity_direction):
        rhs = self.can_variable_rule_be_applied(constraint, activity_direction)
        if rhs is None:
            return NONE
        total, target = ZERO, None
        for variable, coefficient in self.active_row(constraint):
            bound = self.variable_bound(variable, times_sign(activity_direction, coefficient))
            if bound is None:
                if target is None:
                    target = (variable, coefficient)
            else:
                total += coefficient * bound
        target_column, target_coefficient = target
        value = (rhs - total) / target_coefficient
        bound_direction = times_sign(flip(activity_direction), target_coefficient)
        result = self.update_activity_variable_bound(target_column, bound_direction, value)
        if result is None:
            return NONE
        if result == "new":
            self.after_bound_change(target_column, bound_direction, None)
            return MEANINGFUL
        self.after_bound_change(target_column, bound_direction, result[1])
        return NOT_MEANINGFUL

    def can_variable_rule_be_applied(self, constraint, activity_direction):
        rhs = self.b(constraint)
        kind = self.constraint_type(constraint)
        if kind == "Equal":
            return rhs
        if kind == "Less":
            return rhs if activity_direction == "L" else None
        if kind == "Greater":
            return None if activity_direction == "L" else rhs
        return rhs if activity_direction == "L" else rhs - kind[1]

    # ---- updates.rs:278-323 ---------------------------------------------------------------------------------------
    def into_changes(self):
        removed_constraints = set(self.constraints_marked_removed)
        b = {i: v for i, v in self.b_changes.items() if i not in removed_constraints and v != self.gf.b[i]}
        constraints = {i: k for i, k in self.constraint_changes.items()
                       if i not in removed_constraints and k != self.gf.constraint_types[i]}
        removed_vars = {j for j, _ in self.removed_variables}
        bounds = {k: v for k, v in self.bounds.items() if k[0] not in removed_vars}
        activity = {k: v for k, v in self.activity_variable_bounds.items() if k[0] not in removed_vars}
        restrict = {variable for (variable, _d) in activity
                    if self.gf.variables[variable].lower_bound is None and self.gf.variables[variable].upper_bound is None
                    and (variable, "L") not in bounds and (variable, "U") not in bounds}
        for (variable, direction), value in activity.items():
            if variable in restrict:
                bounds[(variable, direction)] = value
        return {
            "b": b, "constraints": constraints, "fixed_cost": self.fixed_cost, "bounds": bounds,
            "removed_variables": sorted(self.removed_variables, key=lambda t: t[0]),
            "constraints_marked_removed": sorted(self.constraints_marked_removed),
        }


def compute_presolve_changes(gf):
    """general_form/mod.rs:360-386."""
    index = Index(gf)
    without_change = 0
    while not index.queues_empty() and without_change < index.nr_variables_remaining() + index.nr_constraints_remaining():
        change = index.presolve_step()
        if change == MEANINGFUL:
            without_change = 0
        elif change == NOT_MEANINGFUL:
            without_change += 1
    return index.into_changes()


def presolve(gf):
    """``GeneralForm::presolve`` (general_form/mod.rs:335-358): compute the changes and apply them in place.

    Raises ``Infeasible`` / ``Unbounded``; returns the changes (for tests).  Removed variables are recorded in
    ``gf.removed`` (original index -> RemovedVariable) for the solution back-mapping.
    """
    changes = compute_presolve_changes(gf)
    # update_values_that_remain (mod.rs:388-421)
    for i, value in changes["b"].items():
        gf.b[i] = value
    for i, kind in changes["constraints"].items():
        gf.constraint_types[i] = kind
    gf.fixed_cost += changes["fixed_cost"]
    for j, solution in changes["removed_variables"]:
        gf.removed[gf.active_to_original[j]] = solution
    for (j, direction), value in changes["bounds"].items():
        if direction == "L":
            gf.variables[j].lower_bound = value
        else:
            gf.variables[j].upper_bound = value
    # remove_rows_and_columns (mod.rs:423-463)
    removed_vars = {j for j, _ in changes["removed_variables"]}
    removed_rows = set(changes["constraints_marked_removed"])
    new_row = {}
    for i in range(len(gf.b)):
        if i not in removed_rows:
            new_row[i] = len(new_row)
    gf.columns = [[(new_row[i], v) for i, v in column if i in new_row]
                  for j, column in enumerate(gf.columns) if j not in removed_vars]
    gf.variables = [v for j, v in enumerate(gf.variables) if j not in removed_vars]
    gf.active_to_original = [o for j, o in enumerate(gf.active_to_original) if j not in removed_vars]
    gf.b = [v for i, v in enumerate(gf.b) if i not in removed_rows]
    gf.constraint_types = [k for i, k in enumerate(gf.constraint_types) if i not in removed_rows]
    return changes
