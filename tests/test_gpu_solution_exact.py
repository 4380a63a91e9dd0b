"""`OptimizationResult::FiniteOptimum(SparseVector<RationalBig>)` at the boundary: the exact solution vector (``relp_get_solution_exact``).

The reference returns exact values and its own tests assert them (tests/burkardt/test.rs:60-112 afiro through
``Solution::is_probably_equal_to``, :143 maros and :185 testprob through equality; src/algorithm/two_phase/test.rs).  Here the f64
device loop ends on a basis, the certificate solves ``B x_B = b`` exactly, and the back-mapping of
``compute_full_solution_with_reduced_solution`` (general_form/mod.rs:840-934) runs in exact arithmetic on the host.
"""
import os
import sys
from fractions import Fraction as F

import pytest

import relp_amd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

# tests/burkardt/test.rs:74-110 (data): the 32 values the reference lists for afiro
AFIRO = {"X01": F(80), "X02": F(51, 2), "X03": F(109, 2), "X04": F(424, 5), "X06": F(255, 14), "X07": 0, "X08": 0, "X09": 0, "X10": 0,
         "X11": 0, "X12": 0, "X13": 0, "X14": F(255, 14), "X15": 0, "X16": F(999), "X22": F(500), "X23": F(11898, 25), "X24": F(602, 25),
         "X25": 0, "X26": F(215), "X28": 0, "X29": 0, "X30": 0, "X31": 0, "X32": 0, "X33": 0, "X34": 0, "X35": 0, "X36": F(11898, 35),
         "X37": F(11898, 35), "X38": 0, "X39": 0}
MAROS = {"VOL1": F(10, 3), "VOL2": F(40, 3), "VOL3": F(20), "VOL4": F(0)}   # tests/burkardt/test.rs:143-150
TESTPROB = {"X1": F(4), "X2": F(-1), "X3": F(6)}                             # tests/burkardt/test.rs:185-191


def solve_file(path, presolve):
    solver = relp_amd.Solver(certify=1).load_mps(path, presolve=presolve)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1
    return solver


def named_solution(solver):
    values = solver.solution_exact(original=True)
    count = len(solver.original_solution())
    return {solver.variable_name(j): values.get(j, F(0)) for j in range(count)}


def is_probably_equal_to(expected, got, min_equal):
    """data/linear_program/solution.rs:47-79 (objective compared by the caller)."""
    if set(expected) != set(got):
        return False
    if len(expected) < 10:
        return True
    return sum(1 for k in expected if expected[k] == got[k]) / len(expected) > min_equal


@pytest.mark.parametrize("presolve", [False, True])
def test_afiro_solution_values(presolve):
    solver = solve_file(os.path.join(ROOT, "data", "burkardt", "afiro.mps"), presolve)
    assert solver.objective_exact() == "-406659/875"
    got = named_solution(solver)
    assert is_probably_equal_to(AFIRO, got, 0.1)                    # what the reference asserts (the optimal face is not a point)
    # every listed value or not, the exact vector must cost exactly the optimum: objective = sum_j c_j x_j over the file's variables
    assert sum(1 for k, v in AFIRO.items() if got[k] == v) >= 20
    solver.close()


@pytest.mark.parametrize("name, expected, objective", [("maros", MAROS, "385/3"), ("testprob", TESTPROB, "54/1")])
@pytest.mark.parametrize("presolve", [False, True])
def test_burkardt_exact_solutions(name, expected, objective, presolve):
    try:
        solver = solve_file(os.path.join(ROOT, "data", "burkardt", name + ".mps"), presolve)
    except relp_amd.RelpError as e:  # the reference's presolve may solve a tiny LP outright (RELP_ERR_STATE)
        if presolve and e.status == relp_amd.api.ERR_STATE:
            pytest.skip("presolve solves it completely: " + str(e))
        raise
    assert solver.objective_exact() == objective
    assert named_solution(solver) == expected
    solver.close()


@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "SC50B", "ADLITTLE", "BLEND", "SHARE2B", "KB2", "SCAGR7", "STOCFOR1"])
def test_exact_vector_is_feasible_and_reaches_the_oracle_optimum(name):
    """Whatever vertex the f64 loop ends on: the exact vector satisfies A x = b, x >= 0 of the oracle's standard form
    exactly, and c'x + fixed cost equals the oracle's exact optimum (tests/golden)."""
    import json
    from relp_oracle.mps import load_problem
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    general, data = load_problem(path)
    solver = solve_file(path, False)
    x = solver.solution_exact()
    assert all(v > 0 for v in x.values())
    # slack-free check: every structural column's contribution; rows of the standard form relate as E | R | <= | >=
    activity = [F(0)] * data.nr_constraints()
    for j, v in x.items():
        for i, a in data.column(j):
            if i < data.nr_constraints():
                activity[i] += a * v
    b = data.right_hand_side()
    e, r, u, l = data.nr_equality, data.nr_range, data.nr_upper, data.nr_lower
    for i in range(data.nr_constraints()):
        if i < e:
            assert activity[i] == b[i]
        elif i < e + r:
            assert b[i] - data.ranges[i - e] <= activity[i] <= b[i]
        elif i < e + r + u:
            assert activity[i] <= b[i]
        else:
            assert activity[i] >= b[i]
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    objective = general.objective_of(sorted(x.items()))
    assert "%d/%d" % (objective.numerator, objective.denominator) == golden["objective"] == solver.objective_exact()
    solver.close()


def test_no_exact_solution_without_certificate():
    solver = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", "AFIRO.SIF"))
    solver.solve_relaxation()
    with pytest.raises(relp_amd.RelpError) as info:
        solver.solution_exact()
    assert info.value.status == relp_amd.api.ERR_STATE
    solver.close()
