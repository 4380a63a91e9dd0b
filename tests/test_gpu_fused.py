"""The fused pivot kernel (ratio test + inverse update in one launch, relp_amd/csrc/kernels.hip `pivot_fused_kernel`) against
the three-kernel pivot it replaces for m <= 1024: same arithmetic in the same order, so the two must agree BIT FOR BIT --
pivot counts, objective, every basic value, the basis, every row of the inverse -- and both must agree with the oracle's exact
optimum through the certificate (``-m gpu``)."""
import os

import numpy as np
import pytest

import relp_amd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["AFIRO", "SC50A", "ADLITTLE", "SHARE2B", "BLEND", "SCAGR7", "BRANDY", "E226", "SCFXM1", "BANDM", "25FV47",
         "CZPROB", "CYCLE"]  # (the last two: 1024 < m <= 2048, four rows per thread)


def load(name, fused, **options):
    if not fused:
        os.environ["RELP_NO_FUSED"] = "1"
    try:
        return relp_amd.Solver(**options).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    finally:
        os.environ.pop("RELP_NO_FUSED", None)


@pytest.mark.parametrize("use_graph", [1, 0], ids=["graph", "plain"])
@pytest.mark.parametrize("name", NAMES)
def test_fused_and_three_kernel_pivots_agree_bit_for_bit(name, use_graph):
    a = load(name, True, use_graph=use_graph)
    b = load(name, False, use_graph=use_graph)
    ra, rb = a.solve_relaxation(), b.solve_relaxation()
    assert ra.kind == rb.kind == relp_amd.FINITE_OPTIMUM
    assert (ra.pivots_phase_one, ra.pivots_phase_two) == (rb.pivots_phase_one, rb.pivots_phase_two)
    assert ra.objective == rb.objective
    assert np.array_equal(a.basis(), b.basis())
    assert np.array_equal(a.solution(), b.solution())
    for r in range(0, a.m, max(1, a.m // 7)):
        assert np.array_equal(a.basis_inverse_row(r), b.basis_inverse_row(r))
    a.close()
    b.close()


@pytest.mark.parametrize("name", ["AFIRO", "SC105", "SHARE1B"])
def test_fused_loop_one_pivot_at_a_time_equals_the_batch(name):
    """`relp_iterate(1)` repeatedly (a batch of one: budget, pricing, fused kernel, commit -- an odd number of buffer swaps each
    time) ends in the same state as one long batch."""
    a = load(name, True)
    b = load(name, True)
    ra = a.solve_relaxation()
    b.begin_phase_one()
    total = 0
    for phase in (1, 2):
        while True:
            done, reason = b.iterate(1)
            total += done
            if done == 0:
                break
            assert total < 100000
        if phase == 1:
            # (drive-out of zero-level artificials is part of solve_relaxation; these LPs have none left after phase one)
            b.begin_phase_two()
    assert total == ra.pivots_phase_one + ra.pivots_phase_two
    assert b.objective_function_value() == ra.objective - 0.0 or abs(b.objective_function_value() - ra.objective) <= 1e-9 * max(1.0, abs(ra.objective))
    a.close()
    b.close()


def test_fused_path_certifies_the_exact_optimum():
    import json
    from fractions import Fraction
    for name in ("AFIRO", "SC50B", "ADLITTLE", "25FV47"):
        golden = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
        s = load(name, True, certify=1)
        r = s.solve_relaxation()
        assert r.kind == relp_amd.FINITE_OPTIMUM and r.certified
        assert Fraction(s.objective_exact()) == Fraction(golden["objective"])
        s.close()


@pytest.mark.parametrize("seed", range(3))
def test_fused_pivot_with_more_than_128_pricing_workgroups(seed):
    """> 4096 columns: the candidates' columns are not staged with them and the entering column comes from the CSC (the other
    FTRAN path of the fused kernel); columns of up to 12 entries also take the long-column tail.  Same bits as the
    three-kernel pivot, and the exact optimum through the certificate of both."""
    import random
    rng = random.Random(31000 + seed)
    m, n = 48, 5000 + 37 * seed
    columns, column_start, rows, nums = [], [0], [], []
    for j in range(n):
        support = sorted(rng.sample(range(m), rng.randint(1, 12)))
        for i in support:
            rows.append(i)
            nums.append(rng.choice([1, 2, 3, 5, 7]))  # non-negative rows: bounded
        column_start.append(len(rows))
    b = [rng.randint(1, 40) for _ in range(m)]
    cost = [rng.randint(-30, 5) for _ in range(n)]
    results = []
    for fused in (True, False):
        if not fused:
            os.environ["RELP_NO_FUSED"] = "1"
        try:
            solver = relp_amd.Solver(certify=1)
            solver.load_matrix_data(column_start, rows, nums, [1] * len(nums), b=b, cost=cost, counts=(0, 0, m, 0))
        finally:
            os.environ.pop("RELP_NO_FUSED", None)
        r = solver.solve_relaxation()
        results.append((r.kind, r.pivots_phase_one, r.pivots_phase_two, r.objective, tuple(solver.basis()),
                        solver.objective_exact() if r.kind == relp_amd.FINITE_OPTIMUM else None, r.certified))
        solver.close()
    assert results[0] == results[1]
    assert results[0][0] == relp_amd.FINITE_OPTIMUM and results[0][6] and results[0][2] > 10


def test_fused_path_is_the_one_that_runs():
    """Two kernel launches per pivot (pricing, fused pivot kernel) instead of three: counted by the library itself."""
    a = load("25FV47", True, use_graph=0)
    b = load("25FV47", False, use_graph=0)
    ra, rb = a.solve_relaxation(), b.solve_relaxation()
    pivots = ra.pivots_phase_one + ra.pivots_phase_two
    assert pivots == rb.pivots_phase_one + rb.pivots_phase_two
    launches_a, launches_b = a.stats().launches, b.stats().launches
    # (batches run past the end of a phase: at most one batch of 64 no-op pivots per phase and per polish interval)
    assert 2 * pivots <= launches_a <= 2 * (pivots + 64 * 12) + 200
    assert 3 * pivots <= launches_b <= 3 * (pivots + 64 * 12) + 200
    assert launches_a < launches_b
    a.close()
    b.close()


@pytest.mark.parametrize("cap", [1, 7, 64, 65, 200])
def test_iteration_cap_and_partial_batches_agree(cap):
    """`max_pivots` smaller than a batch, equal to one, one more: the budget logic of the fused batches (begin / commit kernels,
    copies alternating per pivot) ends in the same state as the three-kernel pivot."""
    a = load("SHARE2B", True, max_pivots=cap)
    b = load("SHARE2B", False, max_pivots=cap)
    ra, rb = a.solve_relaxation(), b.solve_relaxation()
    assert ra.kind == rb.kind
    assert (ra.pivots_phase_one, ra.pivots_phase_two) == (rb.pivots_phase_one, rb.pivots_phase_two)
    assert ra.pivots_phase_one + ra.pivots_phase_two <= cap or ra.kind == relp_amd.FINITE_OPTIMUM
    assert np.array_equal(a.basis(), b.basis())
    a.close()
    b.close()
