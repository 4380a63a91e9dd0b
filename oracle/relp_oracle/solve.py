"""Two-phase driver (oracle; test infrastructure only).

Follows ``phase_one.rs``, ``phase_two.rs`` and ``two_phase/mod.rs``.
"""
from .carry import Carry
from .lu import LUDecomposition
from .pivot_rule import SteepestDescentAlongObjective
from .provider import RemoveRows
from .tableau import Tableau


class Infeasible:
    def __eq__(self, other):
        return isinstance(other, Infeasible)

    def __repr__(self):
        return "Infeasible"


class Unbounded:
    def __eq__(self, other):
        return isinstance(other, Unbounded)

    def __repr__(self):
        return "Unbounded"


class FiniteOptimum:
    """algorithm/mod.rs:43-47: sparse optimal vertex over the provider's columns."""

    def __init__(self, solution, objective=None, basis=None):
        self.solution = solution
        self.objective = objective
        self.basis = basis

    def __eq__(self, other):
        return isinstance(other, FiniteOptimum) and self.solution == other.solution

    def __repr__(self):
        return "FiniteOptimum(%r)" % (self.solution,)


class Trace:
    """Optional recorder of ``(phase, q, p, leaving, cbar_q)`` per pivot (for golden fixtures)."""

    def __init__(self, limit=None):
        self.pivots = []
        self.limit = limit
        self.phase = 1

    def record(self, q, p, leaving, cost):
        self.pivots.append((self.phase, q, p, leaving, cost))
        if self.limit is not None and len(self.pivots) >= self.limit:
            raise PivotLimit()


class PivotLimit(Exception):
    pass


def _simplex_loop(tableau, rule, trace, check):
    """The loop shared by phase_one.rs:134-178 and phase_two.rs:36-58.

    Returns ``None`` when no entering column exists, ``"unbounded"`` when the ratio test fails.
    """
    while True:
        if check:
            tableau.check_bfs_state()
        selected = rule.select_primal_pivot_column(tableau)
        if selected is None:
            return None
        q, cost = selected
        info = tableau.generate_column(q)
        p = tableau.select_primal_pivot_row(info.into_column())
        if p is None:
            return "unbounded"
        change = tableau.bring_into_basis(q, p, info, cost)
        if trace is not None:
            trace.record(q, p, change.leaving_column_index, cost)
        rule.after_basis_update(change, tableau)


def phase_one_primal(tableau, rule_cls=SteepestDescentAlongObjective, trace=None, check=False):
    """phase_one.rs:123-179.  Returns ``None`` (infeasible) or ``(rank_rows, nr_artificial, im, basis)``."""
    rule = rule_cls(tableau)
    if _simplex_loop(tableau, rule, trace, check) == "unbounded":
        raise RuntimeError("Artificial cost can not be unbounded.")  # phase_one.rs:151
    if tableau.objective_function_value() != 0:
        return None
    rows_to_remove = []
    if tableau.has_artificial_in_basis():
        rows_to_remove = remove_artificial_basis_variables(tableau, trace)
    im, nr_artificial, basis = tableau.into_basis()
    return rows_to_remove, nr_artificial, im, basis


def remove_artificial_basis_variables(tableau, trace=None):
    """phase_one.rs:232-278: zero-level pivots; rows that cannot be pivoted are redundant."""
    rows_to_remove = []
    for pivot_row, artificial in tableau.artificial_basis_columns():
        constraint_value = tableau.variable_value(artificial)
        found = None
        for j in range(tableau.nr_artificial_variables(), tableau.nr_columns()):
            if tableau.is_in_basis(j):
                continue
            cost = tableau.relative_cost(j)
            if constraint_value != 0:
                if cost != 0:
                    continue
                element = tableau.generate_element(pivot_row, j)
                if element is not None and element > 0:
                    found = (j, cost)
                    break
            else:
                element = tableau.generate_element(pivot_row, j)
                if element is not None and element != 0:
                    found = (j, cost)
                    break
        if found is not None:
            pivot_column, cost = found
            column = tableau.generate_column(pivot_column)
            change = tableau.bring_into_basis(pivot_column, pivot_row, column, cost)
            if trace is not None:
                trace.record(pivot_column, pivot_row, change.leaving_column_index, cost)
        else:
            rows_to_remove.append(pivot_row)
    return rows_to_remove


def phase_two_primal(tableau, rule_cls=SteepestDescentAlongObjective, trace=None, check=False):
    """phase_two.rs:22-59."""
    rule = rule_cls(tableau)
    if _simplex_loop(tableau, rule, trace, check) == "unbounded":
        return Unbounded()
    return FiniteOptimum(tableau.current_bfs(), tableau.objective_function_value(),
                         list(tableau.inverse_maintainer.basis_indices))


def solve_relaxation(provider, bi_cls=LUDecomposition, rule_cls=SteepestDescentAlongObjective,
                     trace=None, check=False):
    """two_phase/mod.rs:30-73 with ``compute_bfs_giving_im`` (phase_one.rs:44-99).

    Providers with ``pivot_element_indices`` take the partially artificial route
    (``PartialInitialBasis`` specialisation), others the fully artificial one.
    """
    if hasattr(provider, "pivot_element_indices"):
        artificial = Tableau.new_partially_artificial(provider, bi_cls)
    else:
        artificial = Tableau.new_fully_artificial(provider, bi_cls)
    if trace is not None:
        trace.phase = 1
    result = phase_one_primal(artificial, rule_cls, trace, check)
    if result is None:
        return Infeasible()
    rows_to_remove, nr_artificial, im, basis = result
    if trace is not None:
        trace.phase = 2
    if rows_to_remove:
        rows_removed = RemoveRows(provider, rows_to_remove)
        tableau = Tableau.from_artificial_removing_rows(im, nr_artificial, basis, rows_removed)
    else:
        tableau = Tableau.from_artificial(im, nr_artificial, basis, provider)
    return phase_two_primal(tableau, rule_cls, trace, check)


def solve_relaxation_full_basis(provider, bi_cls=LUDecomposition, rule_cls=SteepestDescentAlongObjective,
                                trace=None, check=False):
    """two_phase/mod.rs:80-109 (``FullInitialBasis``): skip phase one, start from the given pivots."""
    pivots = provider.pivot_element_indices()
    im = Carry.from_basis_pivots(bi_cls, pivots, provider)
    tableau = Tableau.new_with_inverse_maintainer(provider, im, [column for _, column in pivots])
    if trace is not None:
        trace.phase = 2
    return phase_two_primal(tableau, rule_cls, trace, check)
