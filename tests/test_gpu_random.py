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
