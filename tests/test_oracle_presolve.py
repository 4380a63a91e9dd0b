"""Presolve restatement (oracle/relp_oracle/presolve.py) against the reference's own known-answer tests.

The vectors below are the data of ``data/linear_program/general_form/presolve/test/{changes.rs, per_rule.rs,
with_application.rs}`` (problem, expected ``Changes`` / counters / removed variables), transcribed as data.
Constraint types: "Equal" | "Less" | "Greater" | ("Range", r); directions "L" / "U".
"""
import glob
import json
import os
from fractions import Fraction as F

import pytest

from relp_oracle import FiniteOptimum, solve_relaxation
from relp_oracle.mps import GeneralForm, load_problem
from relp_oracle.presolve import Index, Infeasible, compute_presolve_changes, presolve
from relp_oracle.provider import Variable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, E, G = "Less", "Equal", "Greater"
BASE_ROWS = [[1, 1], [F(4, 5), F(3, 2)], [F(3, 2), F(4, 5)]]
BASE_B = [1, F(6, 5), F(6, 5)]


def general_form(objective, rows, types, b, variables):
    n = len(variables)
    columns = [[(i, F(rows[i][j])) for i in range(len(rows)) if rows[i][j] != 0] for j in range(n)]
    vs = [Variable(F(c), lower_bound=None if lo is None else F(lo), upper_bound=None if up is None else F(up)) for c, lo, up in variables]
    return GeneralForm(objective, columns, list(types), [F(v) for v in b], vs, ["x%d" % j for j in range(n)])


def changes(b=None, constraints=None, fixed_cost=0, bounds=None, removed=(), rows_removed=()):
    return {"b": {k: F(v) for k, v in (b or {}).items()}, "constraints": dict(constraints or {}), "fixed_cost": F(fixed_cost),
            "bounds": {k: F(v) for k, v in (bounds or {}).items()}, "removed_variables": list(removed),
            "constraints_marked_removed": list(rows_removed)}


def solved(value):
    return ("Solved", F(value))


def function_of_others(constant, coefficients):
    return ("FunctionOfOthers", F(constant), [(j, F(c)) for j, c in coefficients])


V3 = [(1, 0, F(3, 4))] * 3
ROWS3 = [[1, 1, 0], [F(4, 5), F(3, 2), 0], [F(3, 2), F(4, 5), 0]]
ROWS3_SLACK = [[1, 1, 1], [F(4, 5), F(3, 2), 0], [F(3, 2), F(4, 5), 0]]

# (name, objective, rows, types, b, variables, expected)  -- presolve/test/changes.rs
CASES = [
    ("no_changes", "Maximize", BASE_ROWS, [L, L, L], BASE_B, [(1, 0, F(3, 4))] * 2, changes()),
    ("empty_column", "Maximize", ROWS3, [L, L, L], BASE_B, V3, changes(fixed_cost=F(3, 4), removed=[(2, solved(F(3, 4)))])),
    ("empty_column_no_cost", "Maximize", ROWS3, [L, L, L], BASE_B, [(1, 0, F(3, 4))] * 2 + [(0, 0, F(3, 4))],
     changes(removed=[(2, solved(F(3, 4)))])),
    ("empty_column_no_cost_no_bound", "Maximize", ROWS3, [L, L, L], BASE_B, [(1, None, None)] * 2 + [(0, None, None)],
     changes(removed=[(2, solved(0))])),
    ("infeasible_equality_bound", "Maximize", BASE_ROWS + [[0, 0]], [L, L, L, E], BASE_B + [123], [(1, 0, F(3, 4))] * 2, "Infeasible"),
    ("infeasible_inequality_bound", "Maximize", BASE_ROWS + [[0, 0]], [L, L, L, L], BASE_B + [-123], [(1, None, F(3, 4))] * 2, "Infeasible"),
    ("feasible_multiple", "Maximize", [[1, 1], [F(4, 5), F(3, 2)], [0, 0], [F(3, 2), F(4, 5)], [0, 0], [0, 0]], [L, L, E, L, G, L],
     [1, F(6, 5), 0, F(6, 5), -1234, 0], [(1, 0, None)] * 2, changes(rows_removed=[2, 4, 5])),
    ("infeasible_multiple", "Maximize", [[1, 1], [F(4, 5), F(3, 2)], [0, 0], [F(3, 2), F(4, 5)], [0, 0], [0, 0], [0, 0]],
     [L, L, E, L, L, G, L], [1, F(6, 5), 0, F(6, 5), -5678, -1234, 0], [(1, 0, None)] * 2, "Infeasible"),
    ("shift_bounds", "Maximize", [[1, 1, F(1, 1000)], [F(4, 5), F(3, 2), 0], [F(3, 2), F(4, 5), 0]], [L, L, L], BASE_B,
     [(1, 0, F(3, 4))] * 2 + [(0, 1, 1)], changes(b={0: F(999, 1000)}, removed=[(2, solved(1))])),
    ("shift_costs", "Maximize", ROWS3, [L, L, L], BASE_B, [(1, 0, F(3, 4))] * 2 + [(10, 3, 3)],
     changes(fixed_cost=30, removed=[(2, solved(3))])),
    ("trigger_empty_row", "Maximize", [[0, -51, 0], [1, F(1, 100), 1], [F(4, 5), 0, F(3, 2)], [F(3, 2), 0, F(4, 5)]], [L, L, L, L],
     [-20, 1, F(6, 5), F(6, 5)], [(1, 0, F(3, 4)), (15, F(1, 2), F(1, 2)), (1, 0, F(3, 4))],
     changes(b={1: F(199, 200)}, fixed_cost=F(15, 2), removed=[(1, solved(F(1, 2)))], rows_removed=[0])),
    ("trigger_bound_rule", "Maximize", [[0, F(1, 2), 1], [1, 0, 1], [F(4, 5), F(1, 20), F(3, 2)], [F(3, 2), 0, F(4, 5)]], [L, L, L, L],
     [1, 1, F(6, 5), F(6, 5)], [(-12, 0, None), (-11, F(1, 2), F(1, 2)), (-12, 0, None)],
     changes(b={2: F(47, 40)}, fixed_cost=F(-11, 2), bounds={(2, "U"): F(3, 4)}, removed=[(1, solved(F(1, 2)))], rows_removed=[0])),
    ("feasible_inequality", "Maximize", [[1, 1], [F(4, 5), F(3, 2)], [0, 1], [F(3, 2), F(4, 5)]], [L, L, L, L],
     [1, F(6, 5), F(3, 4), F(6, 5)], [(-2, None, None)] * 2, changes(bounds={(1, "U"): F(3, 4)}, rows_removed=[2])),
    ("feasible_inequality_negative_sign", "Maximize", [[1, 1], [F(4, 5), F(3, 2)], [0, -1], [F(3, 2), F(4, 5)]], [L, L, L, L],
     [1, F(6, 5), -F(3, 4), F(6, 5)], [(-2, None, None)] * 2,
     changes(bounds={(0, "U"): F(3, 32), (1, "L"): F(3, 4)}, rows_removed=[2])),
    ("feasible_equality_unsolved_no_substitution", "Maximize", [[1, 1, 0], [F(4, 5), F(3, 2), 0], [0, 0, 1], [F(3, 2), F(4, 5), 0]],
     [L, L, E, L], [1, F(6, 5), 10, F(6, 5)], [(2, None, None)] * 3,
     changes(fixed_cost=20, removed=[(2, solved(10))], rows_removed=[2])),
    ("feasible_equality_unsolved_substitution", "Maximize", [[1, 1, 0], [F(4, 5), F(3, 2), 0], [0, 0, -1], [F(3, 2), F(4, 5), F(1, 100)]],
     [L, L, E, L], [1, F(6, 5), -3, F(6, 5)], [(2, None, None)] * 3,
     changes(b={3: F(117, 100)}, fixed_cost=6, removed=[(2, solved(3))], rows_removed=[2])),
    ("feasible_equality_solved", "Maximize", [[1, 1], [F(4, 5), F(3, 2)], [1, 0], [F(3, 2), F(4, 5)]], [L, L, E, L],
     [1, F(6, 5), F(3, 4), F(6, 5)], [(2, None, None)] * 2,
     changes(fixed_cost=(F(3, 4) + F(3, 32)) * 2, removed=[(0, solved(F(3, 4))), (1, solved(F(3, 32)))], rows_removed=[0, 1, 2, 3])),
    ("feasible_equality_unsolved_independent_column", "Minimize", [[1, 1, 0], [F(4, 5), F(3, 2), 0], [0, 0, 1], [F(3, 2), F(4, 5), 0]],
     [L, L, G, L], [1, F(6, 5), -3, F(6, 5)], [(-3, None, None)] * 2 + [(0, None, None)],
     changes(removed=[(2, solved(-3))], rows_removed=[2])),
    ("feasible_equality_unsolved_slack", "Minimize", [[1, 1, 1], [F(4, 5), F(3, 2), 0], [0, 0, 1], [F(3, 2), F(4, 5), 0]],
     [E, L, G, L], [2, F(6, 5), 1, F(6, 5)], [(-3, None, None)] * 2 + [(0, None, None)],
     changes(b={0: 1}, constraints={0: L}, removed=[(2, function_of_others(2, [(0, 1), (1, 1)]))], rows_removed=[2])),
    ("remove_slack", "Maximize", ROWS3_SLACK, [E, L, L], BASE_B, [(1, 0, F(3, 4))] * 2 + [(0, 0, None)],
     changes(constraints={0: L}, removed=[(2, function_of_others(1, [(0, 1), (1, 1)]))])),
    ("remove_range_slack", "Maximize", ROWS3_SLACK, [E, L, L], BASE_B, [(1, 0, F(3, 4))] * 2 + [(0, 0, F(3, 4))],
     changes(constraints={0: ("Range", F(3, 4))}, removed=[(2, function_of_others(1, [(0, 1), (1, 1)]))])),
    ("free_variable", "Maximize", ROWS3_SLACK, [E, L, L], BASE_B, [(1, 0, F(3, 4))] * 2 + [(0, None, None)],
     changes(removed=[(2, function_of_others(1, [(0, 1), (1, 1)]))], rows_removed=[0])),
    ("range_slack_for_range_bound", "Maximize", ROWS3_SLACK, [("Range", F(1, 2)), L, L], BASE_B,
     [(14, None, None)] * 2 + [(0, 0, F(1, 2))],
     changes(constraints={0: ("Range", F(1))}, removed=[(2, function_of_others(1, [(0, 1), (1, 1)]))])),
    ("range_slack_inequality_bound", "Maximize", [[1, 0, 1], [1, 1, 0], [F(4, 5), F(3, 2), 0], [F(3, 2), F(4, 5), 0]], [L, L, L, L],
     [1, 1, F(6, 5), F(6, 5)], [(1, None, None)] * 2 + [(0, 0, 1)],
     changes(bounds={(0, "U"): 1}, removed=[(2, solved(0))], rows_removed=[0])),
    ("remove_redundant_constraint", "Maximize", BASE_ROWS, [L, L, L], [F(3, 2), F(6, 5), F(6, 5)], [(1, 0, F(3, 4))] * 2,
     changes(rows_removed=[0])),
    ("trigger_fixed_unfeasible", "Maximize", BASE_ROWS, [E, L, L], BASE_B, [(14, F(1, 4), None), (14, F(3, 4), None)], "Infeasible"),
    ("trigger_substitution_feasible", "Maximize", BASE_ROWS, [E, L, L], BASE_B, [(14, F(1, 2), None)] * 2,
     changes(fixed_cost=14, removed=[(0, solved(F(1, 2))), (1, solved(F(1, 2)))], rows_removed=[0, 1, 2])),
    ("trigger_substitution_infeasible", "Maximize", BASE_ROWS, [L, L, L], BASE_B, [(14, F(2, 3), None)] * 2, "Infeasible"),
    ("all_constraints_redundant", "Maximize", BASE_ROWS, [L, L, L], [F(3, 2), F(6, 5), F(6, 5)], [(1, 0, F(1, 4))] * 2,
     changes(fixed_cost=F(1, 2), removed=[(0, solved(F(1, 4))), (1, solved(F(1, 4)))], rows_removed=[0, 1, 2])),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_compute_presolve_changes(case):
    _, objective, rows, types, b, variables, expected = case
    gf = general_form(objective, rows, types, b, variables)
    if expected == "Infeasible":
        with pytest.raises(Infeasible):
            compute_presolve_changes(gf)
    else:
        assert compute_presolve_changes(gf) == expected


# ---- presolve/test/per_rule.rs -----------------------------------------------------------------------------------------
def test_presolve_fixed_variable_feasible():
    index = Index(general_form("Minimize", [[1], [2]], [E, G], [1, 1], [(1, 1, 1)]))
    index.presolve_fixed_variable(0)
    assert index.count_constraint == [0, 0] and index.count_variable == [0]
    assert index.constraints_marked_removed == [0, 1]
    assert index.removed_variables == [(0, solved(1))]
    assert index.b_changes == {0: 0, 1: -1} and index.fixed_cost == 1
    assert not index.queues_empty()


def test_presolve_fixed_variable_infeasible():
    index = Index(general_form("Minimize", [[1], [2]], [E, E], [1, 1], [(1, 1, 1)]))
    with pytest.raises(Infeasible):
        index.presolve_fixed_variable(0)


def test_presolve_simple_bound_constraint():
    index = Index(general_form("Minimize", [[1], [1]], [E, E], [2, 2], [(1, None, None)]))
    index.presolve_bound_constraint(0)
    assert index.count_constraint == [0, 1] and index.count_variable == [1]
    assert index.constraints_marked_removed == [0] and index.removed_variables == []
    assert index.bounds == {(0, "L"): 2, (0, "U"): 2}
    assert not index.queues_empty()


def test_presolve_constraint_if_slack_with_suitable_bounds():
    def create(kind, lower, upper):
        return Index(general_form("Minimize", [[2, 2, 2]], [kind], [3], [(0, lower, upper)] * 3))

    expected = (0, function_of_others(F(3, 2), [(1, 1), (2, 1)]))
    index = create(E, 1, 2)
    index.presolve_slack(0)
    assert index.count_constraint == [2] and index.count_variable == [0, 1, 1]
    assert index.constraints_marked_removed == [] and index.removed_variables == [expected]

    index = create(E, 1, None)
    index.presolve_slack(0)
    assert index.count_constraint == [2] and index.count_variable == [0, 1, 1]
    assert index.constraint_type(0) == L and index.removed_variables == [expected] and index.b(0) == 1

    index = create(E, None, 1)
    index.presolve_slack(0)
    assert index.constraint_type(0) == G and index.removed_variables == [expected] and index.b(0) == 1

    index = create(G, 1, None)
    index.presolve_slack(0)
    assert index.count_constraint == [0] and index.count_variable == [0, 0, 0]
    assert index.constraints_marked_removed == [0]
    assert index.removed_variables == [(1, solved(1)), (2, solved(1)), expected]


# ---- presolve/test/with_application.rs ------------------------------------------------------------------------------------
def test_presolve_solves_the_whole_problem():
    rows = [[2, 0, 0, 0, 0, 0], [3, 5, 0, 0, 0, 0], [7, 11, 13, 0, 0, 0], [17, 19, 23, 0, 29, 31]]
    variables = [(211, None, None), (223, (F(103) - F(101, 2) * 3) / 5, None), (227, None, None), (-229, None, 131),
                 (233, F(-30736, 65 * 29), 123), (0, 5, None)]
    gf = general_form("Minimize", rows, [E, L, G, E], [101, 103, 107, 109], variables)
    gf.fixed_cost = F(1)
    presolve(gf)
    assert gf.variables == [] and gf.b == []  # everything is removed: the reference returns FiniteOptimum from presolve
    assert gf.fixed_cost == (F(1) + F(211 * 101, 2) + F(223 * -97, 10) + F(227 * -699, 65) + F(-229 * 131) + F(233 * -30736, 1885))
    solution = gf.full_solution([])
    assert solution == {"x0": F(101, 2), "x1": (F(103) - F(101, 2) * 3) / 5, "x2": (F(-3601, 5) + F(29 * 30736, 1885)) / 23,
                        "x3": F(131), "x4": F(-30736, 65 * 29), "x5": F(5)}


# ---- the presolved LP has the same optimum ----------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["AFIRO", "ADLITTLE", "BLEND", "KB2", "SCAGR7", "SC50B", "BOEING2", "RECIPELP", "STOCFOR1", "VTP-BASE"])
def test_presolved_netlib_problem_keeps_its_exact_optimum(name):
    with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as handle:
        golden = json.load(handle)
    general, data = load_problem(os.path.join(ROOT, "data", "netlib", name + ".SIF"), presolve=True)
    assert data.nr_rows() <= golden["m"] and data.nr_columns() <= golden["n"]
    result = solve_relaxation(data)
    assert isinstance(result, FiniteOptimum)
    objective = general.objective_of(data.reconstruct_solution(result.solution))
    assert "%d/%d" % (objective.numerator, objective.denominator) == golden["objective"]
