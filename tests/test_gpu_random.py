"""Random small LPs in `MatrixData` form: the HIP path (with the exact certificate) against the exact oracle (``-m gpu``).

Covers every row group (equality, range, <=, >=), variable upper bounds, infeasible and unbounded programs, redundant
rows -- the edge cases the reference's unit tests exercise one at a time (two_phase/test.rs:96-212).
Bit-exact comparison of the optimal objective; result kinds must agree.
"""
import random
from fractions import Fraction

import numpy as np
import pytest

import relp_amd
from relp_oracle import FiniteOptimum, Infeasible, MatrixData, Unbounded, Variable, solve_relaxation

pytestmark = pytest.mark.gpu


def random_lp(rng):
    n = rng.randint(2, 7)
    counts = [rng.randint(0, 3), rng.randint(0, 2), rng.randint(0, 3), rng.randint(0, 2)]  # E, R, <=, >=
    if sum(counts) < 2:  # the reference's LU needs m >= 2 (lower_upper/mod.rs:67-76 allocates m - 1 columns)
        counts[2] += 2
    m = sum(counts)
    dense = [[rng.choice([0, 0, 1, 2, 3, -1, -2, 5]) for _ in range(n)] for _ in range(m)]
    columns = [[(i, dense[i][j]) for i in range(m) if dense[i][j] != 0] for j in range(n)]
    b = [rng.randint(0, 12) for _ in range(m)]
    ranges = [rng.randint(1, 6) for _ in range(counts[1])]
    cost = [rng.randint(-5, 5) for _ in range(n)]
    upper = [rng.choice([None, None, rng.randint(1, 9)]) for _ in range(n)]
    if rng.random() < 0.2 and m >= 2:  # a duplicated row: rank deficiency
        dense[1] = list(dense[0]); b[1] = b[0]
        columns = [[(i, dense[i][j]) for i in range(m) if dense[i][j] != 0] for j in range(n)]
    return n, counts, columns, b, ranges, cost, upper


@pytest.mark.parametrize("seed", range(60))
def test_random_lp_matches_oracle_exactly(seed):
    rng = random.Random(1000 + seed)
    n, counts, columns, b, ranges, cost, upper = random_lp(rng)
    data = MatrixData(columns, b, ranges, counts[0], counts[1], counts[2], counts[3],
                      [Variable(c, upper_bound=u) for c, u in zip(cost, upper)])
    try:
        expected = solve_relaxation(data)
    except AssertionError:
        # the reference's LU needs m >= 2 (decomposition/mod.rs:32); removing redundant rows can leave a single row
        pytest.skip("the reference's LU cannot factor a 1 x 1 basis")

    column_start = [0]
    rows, nums = [], []
    for col in columns:
        for i, v in col:
            rows.append(i)
            nums.append(v)
        column_start.append(len(rows))
    solver = relp_amd.Solver(certify=1)
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
        assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
        assert solver.objective_exact() == "%d/%d" % (objective.numerator, objective.denominator)
        assert abs(result.objective - float(objective)) <= 1e-9 * max(1.0, abs(float(objective)))
        x = solver.solution()
        assert abs(float(np.dot(x, cost)) - float(objective)) <= 1e-8 * max(1.0, abs(float(objective)))
    solver.close()


def test_iteration_limit_is_reported():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    solver = relp_amd.Solver(max_pivots=10).load_mps(os.path.join(root, "data", "netlib", "ADLITTLE.SIF"))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.ITERATION_LIMIT
    assert result.pivots_phase_one + result.pivots_phase_two <= 10 + 64
    solver.close()


def dense_mixed_lp(rng, m, n):
    """Feasible, bounded LP with a dense coefficient block and all four row kinds (equality, range, <=, >=): built from a
    primal point x0 >= 0 (feasibility) and dual multipliers of the right signs (boundedness)."""
    counts = [m // 4, m // 8, m - m // 4 - m // 8 - m // 5, m // 5]  # E, R, <=, >=
    a = [[rng.randint(1, 9) * rng.choice([1, 1, 1, -1]) for _ in range(n)] for _ in range(m)]
    x0 = [rng.randint(0, 3) for _ in range(n)]
    activity = [sum(a[i][j] * x0[j] for j in range(n)) for i in range(m)]
    b, ranges = [], []
    row = 0
    for _ in range(counts[0]):
        b.append(activity[row]); row += 1
    for _ in range(counts[1]):
        width = rng.randint(1, 9)
        b.append(activity[row] + rng.randint(0, width)); ranges.append(width); row += 1   # b - range <= a x <= b
    for _ in range(counts[2]):
        b.append(activity[row] + rng.randint(0, 9)); row += 1
    for _ in range(counts[3]):
        b.append(activity[row] - rng.randint(0, 9)); row += 1
    # the standard form needs b >= 0: flip equality / inequality rows with a negative right-hand side where allowed
    for i in range(m):
        if b[i] < 0 and i < counts[0]:
            b[i] = -b[i]; a[i] = [-v for v in a[i]]
    if any(v < 0 for v in b):
        return None
    # dual point: y free on E, <= 0 on '<=', >= 0 on '>=' (minimisation); c = A'y + s, s >= 0
    y = [rng.randint(-2, 2) if i < counts[0] else (0 if i < counts[0] + counts[1] else (-rng.randint(0, 2) if i < m - counts[3] else rng.randint(0, 2)))
         for i in range(m)]
    cost = [sum(a[i][j] * y[i] for i in range(m)) + rng.randint(0, 4) for j in range(n)]
    return counts, a, b, ranges, cost


@pytest.mark.parametrize("storage", ["bytes", "float", "f64"])  # the dense block in each of its exact storage types
@pytest.mark.parametrize("seed", range(6))
def test_dense_block_with_mixed_rows_exact(seed, storage, monkeypatch):
    """Dense pipeline (forced by the test hook) on LPs with artificials, range rows and zero-level pivots.  The device's
    exact certificate must hold (an independent proof of optimality in rational arithmetic), the optimum must agree with
    HiGHS, and on the smallest size with the exact optimum of the C++ oracle (its rationals make larger dense LPs slow)."""
    from scipy.optimize import linprog
    monkeypatch.setenv("RELP_FTRAN_MIN_NNZ", "16")
    if storage != "bytes":  # coefficients here are integers in [-9, 9]: signed bytes unless told otherwise
        monkeypatch.setenv("RELP_DENSE_F32" if storage == "float" else "RELP_DENSE_F64", "1")
    rng = random.Random(7000 + seed)
    m, n = rng.choice([(64, 96), (80, 128), (97, 130), (120, 200)])
    lp = None
    while lp is None:
        lp = dense_mixed_lp(rng, m, n)
    counts, a, b, ranges, cost = lp
    columns = [[(i, a[i][j]) for i in range(m) if a[i][j] != 0] for j in range(n)]
    column_start, rows, nums = [0], [], []
    for col in columns:
        for i, v in col:
            rows.append(i); nums.append(v)
        column_start.append(len(rows))
    solver = relp_amd.Solver(certify=1, polish_period=32)
    solver.load_matrix_data(column_start, rows, nums, [1] * len(nums), b=b, cost=cost, upper=[None] * n, ranges=ranges,
                            counts=tuple(counts))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    exact = Fraction(solver.objective_exact())
    # HiGHS on the same LP
    am = np.array(a, dtype=float)
    e, r = counts[0], counts[1]
    le0, le1 = e + r, e + r + counts[2]
    a_ub = np.vstack([am[e:e + r], -am[e:e + r], am[le0:le1], -am[le1:]])
    b_ub = np.concatenate([np.array(b[e:e + r], float), -(np.array(b[e:e + r], float) - np.array(ranges, float)),
                           np.array(b[le0:le1], float), -np.array(b[le1:], float)])
    highs = linprog(cost, A_ub=a_ub, b_ub=b_ub, A_eq=am[:e], b_eq=np.array(b[:e], float), bounds=(0, None), method="highs")
    assert highs.status == 0
    assert abs(float(exact) - highs.fun) <= 1e-7 * max(1.0, abs(highs.fun))
    if (m, n) == (64, 96):
        from relp_oracle import cpu
        data = MatrixData(columns, b, ranges, counts[0], counts[1], counts[2], counts[3], [Variable(c) for c in cost])
        record = cpu.solve_provider(data, trace=0)
        objective = sum((Fraction(cost[j]) * v for j, v in data.reconstruct_solution(record["solution"])), Fraction(0))
        assert exact == objective
    solver.close()
