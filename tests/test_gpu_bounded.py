"""Implicit upper bounds (``relp_options.implicit_bounds``; ``-m gpu``): the VariableBound / SlackBound rows of MatrixData
are handled by the bounded-variable ratio test instead of as rows.  The optimum must not change: every case is compared
with the exact oracle (which works on the reference's explicit formulation), and the final state is mapped back to the
explicit basis and proved optimal by the exact certificate."""
import json
import os
import random
from fractions import Fraction

import numpy as np
import pytest

import relp_amd
from relp_oracle import FiniteOptimum, Infeasible, MatrixData, Unbounded, Variable, solve_relaxation

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")) as handle:
    EXPECTED = json.load(handle)
NAMES = sorted(n for n, e in EXPECTED.items() if os.path.exists(os.path.join(ROOT, "data", "netlib", n + ".SIF"))
               and (not e["ignored"] or "intensive" in e["ignored"]))


def bounded_lp(rng):
    n = rng.randint(2, 8)
    counts = [rng.randint(0, 3), rng.randint(0, 2), rng.randint(0, 3), rng.randint(0, 2)]  # E, R, <=, >=
    if sum(counts) < 2:
        counts[2] += 2
    m = sum(counts)
    dense = [[rng.choice([0, 0, 1, 2, 3, -1, -2, 5]) for _ in range(n)] for _ in range(m)]
    columns = [[(i, dense[i][j]) for i in range(m) if dense[i][j] != 0] for j in range(n)]
    b = [rng.randint(0, 12) for _ in range(m)]
    ranges = [rng.randint(1, 6) for _ in range(counts[1])]
    cost = [rng.randint(-6, 4) for _ in range(n)]
    upper = [rng.choice([None, rng.randint(1, 9), rng.randint(1, 4)]) for _ in range(n)]  # most variables bounded
    return n, counts, columns, b, ranges, cost, upper


@pytest.mark.parametrize("seed", range(100))
def test_random_bounded_lp_matches_oracle_exactly(seed):
    rng = random.Random(5000 + seed)
    n, counts, columns, b, ranges, cost, upper = bounded_lp(rng)
    if seed >= 80:  # variables FIXED at zero (both bounds at one point): never priced on the device, resolved at the end
        upper = [0 if rng.random() < 0.35 else u for u in upper]
    data = MatrixData(columns, b, ranges, counts[0], counts[1], counts[2], counts[3],
                      [Variable(c, upper_bound=u) for c, u in zip(cost, upper)])
    try:
        expected = solve_relaxation(data)
    except AssertionError:
        pytest.skip("the reference's LU cannot factor a 1 x 1 basis")
    column_start, rows, nums = [0], [], []
    for col in columns:
        for i, v in col:
            rows.append(i)
            nums.append(v)
        column_start.append(len(rows))
    solver = relp_amd.Solver(certify=1, implicit_bounds=1)
    solver.load_matrix_data(column_start, rows or [0], nums or [0], [1] * max(1, len(nums)), b=b, cost=cost, upper=upper,
                            ranges=ranges, counts=tuple(counts))
    result = solver.solve_relaxation()
    if isinstance(expected, Infeasible):
        assert result.kind == relp_amd.INFEASIBLE
    elif isinstance(expected, Unbounded):
        assert result.kind == relp_amd.UNBOUNDED
    else:
        assert isinstance(expected, FiniteOptimum)
        assert result.kind == relp_amd.FINITE_OPTIMUM
        objective = sum((Fraction(cost[j]) * v for j, v in data.reconstruct_solution(expected.solution)), Fraction(0))
        assert abs(result.objective - float(objective)) <= 1e-9 * max(1.0, abs(float(objective)))
        assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
        assert solver.objective_exact() == "%d/%d" % (objective.numerator, objective.denominator)
        x = solver.solution()
        assert abs(float(np.dot(x, cost)) - float(objective)) <= 1e-8 * max(1.0, abs(float(objective)))
        for j, u in enumerate(upper):
            assert x[j] >= -1e-9 and (u is None or x[j] <= u + 1e-9)
    solver.close()


@pytest.mark.parametrize("name", NAMES)
def test_netlib_with_implicit_bounds(name):
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    golden_path = os.path.join(ROOT, "tests", "golden", name + ".json")
    solver = relp_amd.Solver(certify=1 if os.path.exists(golden_path) else 0, implicit_bounds=1).load_mps(path)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    entry = EXPECTED[name]
    tolerance = max(entry["tolerance"], 2e-5 if name == "25FV47" else 0.0)
    assert abs(result.objective - entry["expected"]) <= tolerance
    if os.path.exists(golden_path):
        with open(golden_path) as handle:
            golden = json.load(handle)
        assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
        assert Fraction(solver.objective_exact()) == Fraction(golden["objective"])
    solver.close()


@pytest.mark.parametrize("name", ["AFIRO", "BOEING2", "KB2", "RECIPELP", "SCTAP1", "ETAMACRO", "80BAU3B"])
def test_basis_round_trip_between_the_two_formulations(name):
    """``relp_get_basis`` / ``relp_set_basis`` speak the reference's formulation in both modes: the optimal basis found with
    the bound rows explicit warm-starts the implicit-bounds solver (and the other way round) with nothing left to pivot."""
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    solvers = {mode: relp_amd.Solver(implicit_bounds=mode, certify=0).load_mps(path) for mode in (0, 1)}
    results = {mode: s.solve_relaxation() for mode, s in solvers.items()}
    bases = {mode: s.basis() for mode, s in solvers.items()}
    fixed_cost = relp_amd.Model(path).fixed_cost()  # the handle-level objective of the fine-grained operations excludes it
    assert len(bases[0]) == len(bases[1]) == solvers[0].m
    for source, target in ((0, 1), (1, 0), (1, 1)):
        fresh = relp_amd.Solver(implicit_bounds=target, certify=0).load_mps(path)
        fresh.set_basis(bases[source])
        done, _ = fresh.iterate(1000)
        objective = fresh.objective_function_value()
        expected = results[source].objective
        assert abs(objective + fixed_cost - expected) <= 1e-7 * max(1.0, abs(expected)), (source, target)
        assert done <= 2, (source, target, done)  # optimal already (a tie in the tolerances may cost a degenerate pivot)
        # (the basis read back may differ from the one given on the bound row of a FIXED variable: with both bounds at the same
        # point the variable and its bound slack are interchangeable there, and the sign of its reduced cost decides)
        again = relp_amd.Solver(implicit_bounds=target, certify=0).load_mps(path)
        again.set_basis(fresh.basis())
        assert again.iterate(1000)[0] <= 2
        assert abs(again.objective_function_value() + fixed_cost - expected) <= 1e-7 * max(1.0, abs(expected))
        again.close()
        fresh.close()
    for s in solvers.values():
        s.close()

