"""`relp_options.crash` on general LPs (``-m gpu``): the triangular crash basis is an optional START of phase one -- kept only
when it is primal feasible, otherwise the reference's artificial start is used -- so whatever it does, the result reported must
be the reference's: the exact optimum (certificate), the same verdict on infeasible / unbounded LPs."""
import json
import os
import random
from fractions import Fraction

import pytest

import relp_amd
from relp_oracle import FiniteOptimum, Infeasible, MatrixData, Unbounded, Variable, solve_relaxation

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["AFIRO", "SC50A", "SC105", "ADLITTLE", "SHARE2B", "BLEND", "SCAGR7", "E226", "BANDM", "SCFXM1", "STOCFOR1", "25FV47"]


@pytest.mark.parametrize("implicit", [0, 1], ids=["bound-rows", "implicit-bounds"])
@pytest.mark.parametrize("name", NAMES)
def test_netlib_with_crash_reaches_the_exact_optimum(name, implicit):
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    solver = relp_amd.Solver(certify=1, crash=1, implicit_bounds=implicit).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
    assert Fraction(solver.objective_exact()) == Fraction(golden["objective"])
    solver.close()


def random_lp(rng):
    """Equality-heavy LPs (they need artificials, so the crash has rows to cover) with every verdict."""
    n = rng.randint(4, 12)
    m_eq, m_le, m_ge = rng.randint(1, 5), rng.randint(0, 3), rng.randint(0, 2)
    if m_eq + m_le + m_ge < 2:
        m_le += 1  # (a single row: `LUDecomposition::change_basis` of the reference indexes an empty Vec, lower_upper/mod.rs:150 --
                   #  the oracle restates that faithfully, so such an LP has no reference answer through the LU carry)
    m = m_eq + m_le + m_ge
    dense = [[rng.choice([1, -1, 2, 3, -2]) if rng.random() < 0.35 else 0 for _ in range(n)] for _ in range(m)]
    for i in range(m):
        if not any(dense[i]):
            dense[i][rng.randrange(n)] = 1
    columns = [[(i, dense[i][j]) for i in range(m) if dense[i][j]] for j in range(n)]
    b = [rng.choice([0, 0, 1, 2, 5]) for _ in range(m)]
    cost = [rng.randint(-4, 6) for _ in range(n)]
    return columns, b, cost, (m_eq, 0, m_le, m_ge)


@pytest.mark.parametrize("seed", range(60))
def test_random_lps_with_crash_match_the_oracle(seed):
    rng = random.Random(4100 + seed)
    columns, b, cost, counts = random_lp(rng)
    data = MatrixData(columns, b, [], counts[0], counts[1], counts[2], counts[3], [Variable(c) for c in cost])
    exact = solve_relaxation(data)
    column_start, rows, nums = [0], [], []
    for column in columns:
        for i, v in column:
            rows.append(i)
            nums.append(v)
        column_start.append(len(rows))
    solver = relp_amd.Solver(certify=1, crash=1)
    solver.load_matrix_data(column_start, rows, nums, [1] * len(nums), b=b, cost=cost, counts=tuple(counts))
    result = solver.solve_relaxation()
    if isinstance(exact, FiniteOptimum):
        assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
        objective = sum((Fraction(cost[j]) * v for j, v in data.reconstruct_solution(exact.solution)), Fraction(0))
        assert Fraction(solver.objective_exact()) == objective
    elif isinstance(exact, Infeasible):
        assert result.kind == relp_amd.INFEASIBLE
    else:
        assert isinstance(exact, Unbounded) and result.kind == relp_amd.UNBOUNDED
    solver.close()
