"""Parity of the HIP path (through the C ABI) with the exact oracle, on a real MI355X (``-m gpu``).

f64 path: every vector the fine-grained trait ops return is compared with the oracle's exact rationals on the same
state (tolerance stated per assert); whole solves are compared with the exact golden optimum at 1e-9 relative
(the bound ``north_star`` states for the f64 path).
"""
import glob
import json
import os
from fractions import Fraction

import numpy as np
import pytest

import relp_amd
from reference_expectations import NETLIB
from relp_oracle import (Carry, LUDecomposition, SteepestDescentAlongObjective, Tableau, FirstProfitable,
                         SteepestDescentAlongVariable, FirstProfitableWithMemory)
from relp_oracle.mps import load_problem

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = {os.path.basename(p)[:-5]: json.load(open(p)) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.json"))}
GOLDEN = {name: g for name, g in GOLDEN.items() if "status" in g}  # per-LP fixtures only
REL = 1e-9  # north_star: "f64 path within 1e-9 rel"


def dense(sparse, m):
    out = np.zeros(m)
    for i, v in sparse:
        out[i] = float(v)
    return out


def close(got, exact_sparse, m):
    """f64 vector vs exact rationals: 1e-8 relative to the entry or to the vector's largest entry (no polish is
    run in the step-by-step tests, so rounding of up to 40 product-form updates accumulates)."""
    want = dense(exact_sparse, m)
    scale = max(1.0, float(np.abs(want).max()))
    return np.allclose(got, want, rtol=1e-8, atol=1e-9 * scale)


def exact_objective(name):
    num, den = GOLDEN[name]["objective"].split("/")
    return Fraction(int(num), int(den))


@pytest.fixture(scope="module")
def library():
    return relp_amd.lib()


# ---------------------------------------------------------------------------------------------------------
# whole solves
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(n for n, g in GOLDEN.items() if g["status"] == "optimal"))
def test_solve_relaxation_matches_exact_optimum(name):
    golden = GOLDEN[name]
    solver = relp_amd.Solver().load_mps(os.path.join(ROOT, golden["file"]))
    assert (solver.m, solver.n_provider) == (golden["m"], golden["n"])
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    expected = float(exact_objective(name))
    assert abs(result.objective - expected) <= REL * max(1.0, abs(expected)), (result.objective, expected)
    assert result.max_residual < 1e-3  # |I - B Binv| before a polish; absolute, so it scales with |B|
    # the reported vertex is feasible and reproduces the objective (general_form/mod.rs:840-851)
    model = relp_amd.Model(os.path.join(ROOT, golden["file"]))
    x = solver.solution()
    cost = np.array([model.cost_value(j) for j in range(model.nr_structural)])
    assert abs(cost @ x + model.fixed_cost() - expected) <= 1e-8 * max(1.0, abs(expected))
    assert x.min() >= -1e-7
    solver.close()


@pytest.mark.parametrize("name", ["25FV47", "CZPROB", "BNL1"])
def test_headline_problems_meet_reference_tolerance(name):
    """tests/netlib/test.rs: 25FV47 = 5.5018459e+03 within 1e-5 (ignored there as too intensive)."""
    expected, tolerance, _ = NETLIB[name]
    solver = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    if name == "25FV47":
        # The constant in tests/netlib/test.rs:10 (5.5018459e+03, tol 1e-5) is the Netlib optimum rounded to 8 digits;
        # the true optimum 5.5018458883E+03 (tests/netlib/problem_files/README:85) is 1.2e-5 away from it, so that
        # (ignored) reference test cannot pass as written.  The README value is the expectation here.
        assert abs(result.objective - 5.5018458883e+03) < 1e-6, result.objective
        assert abs(result.objective - expected) < 2e-5
    else:
        assert abs(result.objective - expected) < max(tolerance, REL * abs(expected)), result.objective
    solver.close()


@pytest.mark.parametrize("carry", [relp_amd.api.CARRY_EXPLICIT, relp_amd.api.CARRY_LU, relp_amd.api.CARRY_LU_INVERSE], ids=["explicit", "lu", "lu_inverse"])
def test_unbounded_and_infeasible_are_reported(carry):
    solver = relp_amd.Solver(carry=carry).load_mps(os.path.join(ROOT, "data", "burkardt", "nazareth.mps"))
    assert solver.solve_relaxation().kind == relp_amd.UNBOUNDED  # tests/burkardt/test.rs:157-167
    solver.close()
    # x0 >= 2 and x0 <= 1: infeasible (phase_one.rs:171-173)
    solver = relp_amd.Solver(carry=carry)
    solver.load_matrix_data([0, 2], [0, 1], [1, 1], [1, 1], b=[1, 2], cost=[1], counts=(0, 0, 1, 1))
    assert solver.solve_relaxation().kind == relp_amd.INFEASIBLE
    solver.close()


@pytest.mark.parametrize("rule", [relp_amd.DANTZIG, relp_amd.FIRST_PROFITABLE, relp_amd.FIRST_PROFITABLE_MEMORY])
@pytest.mark.parametrize("carry", [relp_amd.api.CARRY_EXPLICIT, relp_amd.api.CARRY_LU, relp_amd.api.CARRY_LU_INVERSE], ids=["explicit", "lu", "lu_inverse"])
@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "ADLITTLE", "BLEND"])
def test_other_pivot_rules_reach_the_optimum(name, rule, carry):
    solver = relp_amd.Solver(pivot_rule=rule, carry=carry).load_mps(os.path.join(ROOT, GOLDEN[name]["file"]))
    result = solver.solve_relaxation()
    expected = float(exact_objective(name))
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective - expected) <= REL * max(1.0, abs(expected))
    solver.close()


def test_graph_and_plain_launches_agree():
    path = os.path.join(ROOT, "data", "netlib", "SHARE2B.SIF")
    a = relp_amd.Solver(use_graph=1).load_mps(path).solve_relaxation()
    b = relp_amd.Solver(use_graph=0).load_mps(path).solve_relaxation()
    assert a.objective == b.objective
    assert (a.pivots_phase_one, a.pivots_phase_two) == (b.pivots_phase_one, b.pivots_phase_two)


@pytest.mark.parametrize("carry", [relp_amd.api.CARRY_EXPLICIT, relp_amd.api.CARRY_LU, relp_amd.api.CARRY_LU_INVERSE], ids=["explicit", "lu", "lu_inverse"])
def test_redundant_rows_and_empty_rows(carry):
    """two_phase/test.rs:96-212: redundant_row, empty_row_at_eq, empty_row_at_ineq -> x = (3/4, 1/4, ...)."""
    # three identical equality rows x0 + x1 = 1, x0 <= 3/4, min -2 x0 - x1
    solver = relp_amd.Solver(carry=carry)
    solver.load_matrix_data([0, 3, 6], [0, 1, 2, 0, 1, 2], [1] * 6, [1] * 6, b=[1, 1, 1], cost=[-2, -1],
                            upper=[(3, 4), None], counts=(3, 0, 0, 0))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert np.allclose(solver.solution(), [0.75, 0.25], atol=1e-12)
    solver.close()
    for b, counts in (([1, 0], (2, 0, 0, 0)), ([1, 1], (1, 0, 1, 0))):
        solver = relp_amd.Solver(carry=carry)
        solver.load_matrix_data([0, 1, 2], [0, 0], [1, 1], [1, 1], b=b, cost=[-2, -1], upper=[(3, 4), None], counts=counts)
        result = solver.solve_relaxation()
        assert result.kind == relp_amd.FINITE_OPTIMUM
        assert np.allclose(solver.solution(), [0.75, 0.25], atol=1e-12)
        solver.close()


@pytest.mark.parametrize("carry", [relp_amd.api.CARRY_EXPLICIT, relp_amd.api.CARRY_LU, relp_amd.api.CARRY_LU_INVERSE], ids=["explicit", "lu", "lu_inverse"])
def test_set_basis_warm_start(carry):
    """`InverseMaintainer::from_basis` (carry/mod.rs:444-478): restart from the optimal basis => zero pivots."""
    path = os.path.join(ROOT, "data", "netlib", "SC50A.SIF")
    solver = relp_amd.Solver(carry=carry).load_mps(path)
    first = solver.solve_relaxation()
    basis = solver.basis()
    assert (basis >= 0).all()
    other = relp_amd.Solver(carry=carry).load_mps(path)
    other.set_basis(basis)
    done, reason = other.iterate(1000)
    assert (done, reason) == (0, relp_amd.STOP_NO_ENTERING)
    assert abs(other.objective_function_value() + relp_amd.Model(path).fixed_cost() - first.objective) < 1e-9 * abs(first.objective)


# ---------------------------------------------------------------------------------------------------------
# fine-grained trait ops, step by step against the oracle on the same basis
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("carry", [relp_amd.api.CARRY_EXPLICIT, relp_amd.api.CARRY_LU, relp_amd.api.CARRY_LU_INVERSE], ids=["explicit", "lu", "lu_inverse"])
@pytest.mark.parametrize("name, steps", [("AFIRO", 19), ("SC50A", 30), ("ADLITTLE", 40), ("SHARE2B", 40)])
def test_trait_ops_follow_the_oracle(name, steps, carry):
    path = os.path.join(ROOT, GOLDEN[name]["file"])
    general, data = load_problem(path)
    tableau = Tableau.new_partially_artificial(data, LUDecomposition)
    rule = SteepestDescentAlongObjective(tableau)
    # (refactor_period 7: the LU carry goes through several refactorisations inside the compared steps)
    solver = relp_amd.Solver(polish_period=0, use_graph=0, carry=carry, refactor_period=7).load_mps(path)
    solver.begin_phase_one()
    m, n = solver.m, solver.n
    assert n == tableau.nr_columns() and solver.n_art == tableau.nr_artificial_variables()
    rng = np.random.default_rng(7)
    for step in range(steps):
        # Tableau::relative_cost for every column (tableau/mod.rs:106-112)
        cbar = solver.relative_costs()
        exact_cbar = np.array([float(tableau.relative_cost(j)) for j in range(n)])
        assert np.allclose(cbar, exact_cbar, rtol=1e-9, atol=1e-9)
        # steepest-edge weights (pivot_rule.rs:202-219, 243-296)
        selected = solver.select_primal_pivot_column()  # applies the pending weight update
        gamma = solver.gamma()
        for j in range(solver.n_art, n):  # weights of artificial columns are never read (pivot_rule.rs:57-80)
            if rule.gamma[j] is not None and not tableau.is_in_basis(j):
                assert gamma[j] == pytest.approx(float(rule.gamma[j]), rel=1e-8), (step, j)
        expected = rule.select_primal_pivot_column(tableau)
        if expected is None:
            assert selected is None
            break
        q, cost_q = selected
        key = float(tableau.relative_cost(q) ** 2 / rule.gamma[q])
        best = float(expected[1] ** 2 / rule.gamma[expected[0]])
        assert key == pytest.approx(best, rel=1e-9)      # same maximum (ties may pick another column)
        assert cost_q == pytest.approx(float(tableau.relative_cost(q)), rel=1e-9, abs=1e-12)
        # generate_column / FTRAN (tableau/mod.rs:126-130, lower_upper/mod.rs:180-210)
        info = tableau.generate_column(q)
        p, alpha = solver.select_primal_pivot_row(q)
        assert close(alpha, info.column, m)
        exact_p = tableau.select_primal_pivot_row(info.column)
        assert (p is None) == (exact_p is None)
        # ratio test: the chosen row attains the exact minimum ratio (tableau/mod.rs:287-313)
        col = dict(info.column)
        assert col[p] > 0
        assert tableau.inverse_maintainer.b[p] / col[p] == tableau.inverse_maintainer.b[exact_p] / col[exact_p]
        # BTRAN of a random sparse row and a basis inverse row (lower_upper/mod.rs:212-272)
        rows = np.sort(rng.choice(m, size=min(3, m), replace=False)).astype(np.int32)
        vals = rng.integers(1, 9, size=len(rows)).astype(np.float64)
        exact = tableau.inverse_maintainer.basis_inverse.right_multiply_by_basis_inverse(
            [(int(i), Fraction(int(v))) for i, v in zip(rows, vals)])
        assert close(solver.right_multiply_by_basis_inverse(rows, vals), exact, m)
        exact = tableau.inverse_maintainer.basis_inverse.basis_inverse_row(int(rows[0]))
        assert close(solver.basis_inverse_row(int(rows[0])), exact, m)
        exact = tableau.inverse_maintainer.basis_inverse.left_multiply_by_basis_inverse(
            [(int(i), Fraction(int(v))) for i, v in zip(rows, vals)]).column
        assert close(solver.left_multiply_by_basis_inverse(rows, vals), exact, m)
        # one device iteration; the oracle follows the device's (q, p) so that both stay on the same basis
        done, reason = solver.iterate(1)
        assert done == 1
        basis = solver.basis()
        device_q = basis[p] + solver.n_art if basis[p] >= 0 else -1 - basis[p]
        assert device_q == q
        change = tableau.bring_into_basis(q, p, info, tableau.relative_cost(q))
        rule.after_basis_update(change, tableau)
        assert np.allclose(solver.b(), [float(v) for v in tableau.inverse_maintainer.b], rtol=1e-9, atol=1e-10)
        assert solver.objective_function_value() == pytest.approx(float(tableau.objective_function_value()), rel=1e-9, abs=1e-10)
    solver.close()


@pytest.mark.parametrize("fused", [True, False], ids=["fused-pivot", "three-kernel-pivot"])
def test_profile_hook_runs(fused):
    if not fused:
        os.environ["RELP_NO_FUSED"] = "1"
    try:
        solver = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", "SC105.SIF"))
    finally:
        os.environ.pop("RELP_NO_FUSED", None)
    solver.begin_phase_one()
    solver.iterate(5)
    for which in (0, 1, 2):
        if fused and which == 2:  # small LPs: the update is part of kernel 1 (pivot_fused_kernel)
            with pytest.raises(relp_amd.api.RelpError):
                solver.profile_kernel(which, 10)
            continue
        seconds = solver.profile_kernel(which, 10)
        assert 0 < seconds < 1e-2
    # state is restored: the solve still reaches the optimum
    result_after = solver.iterate(10 ** 6)
    assert result_after[1] == relp_amd.STOP_NO_ENTERING
    solver.close()


ALL_NETLIB = sorted(name for name in NETLIB if os.path.exists(os.path.join(ROOT, "data", "netlib", name + ".SIF")) and name != "GROW7")


@pytest.mark.parametrize("name", ALL_NETLIB)
def test_every_netlib_problem_of_the_reference_suite(name):
    """All 47 solvable problems of tests/netlib/test.rs -- including the 9 it ignores as "too computationally intensive"
    and the 2 it suspects of cycling -- within the tolerance that test states, and with the exact certificate."""
    expected, tolerance, _ = NETLIB[name]
    solver = relp_amd.Solver(certify=1).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    limit = 2e-5 if name == "25FV47" else tolerance  # see test_headline_problems_meet_reference_tolerance
    assert abs(result.objective - expected) < limit, (result.objective, expected)
    num, den = solver.objective_exact().split("/")
    assert abs(Fraction(int(num), int(den)) - Fraction(expected)) < Fraction(limit)
    solver.close()


@pytest.mark.parametrize("name, expected, tolerance", [("50v-10", 2879.065687, 1e-3), ("acc-tight4", 0.0, 1e-3)])
def test_miplib_relaxations(name, expected, tolerance):
    """tests/miplib/test.rs:3-18 (free-format `import`, `Carry<RationalBig, BasisInverseRows<_>>`); acc-tight4 is ignored
    there as "too computationally expensive"."""
    solver = relp_amd.Solver(certify=1).load_mps(os.path.join(ROOT, "data", "miplib", name + ".mps"), fixed=False)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective - expected) < tolerance, result.objective
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    solver.close()


@pytest.mark.parametrize("carry", [relp_amd.api.CARRY_EXPLICIT, relp_amd.api.CARRY_LU, relp_amd.api.CARRY_LU_INVERSE], ids=["explicit", "lu", "lu_inverse"])
@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "ADLITTLE"])
def test_loop_driven_from_outside_through_bring_into_basis(name, carry):
    """The reference's loop (phase_one.rs:134-178 / phase_two.rs:36-58) written by the caller with the fine-grained operations
    -- select_primal_pivot_column, generate_column + select_primal_pivot_row, bring_into_basis -- reaches the same optimum
    as ``solve_relaxation``; ``refactor`` in the middle leaves the state where it was."""
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    whole = relp_amd.Solver(certify=0).load_mps(path)
    expected = whole.solve_relaxation()
    assert expected.kind == relp_amd.FINITE_OPTIMUM
    fixed_cost = relp_amd.Model(path).fixed_cost()
    solver = relp_amd.Solver(certify=0, carry=carry, refactor_period=9).load_mps(path)
    solver.begin_phase_one()
    pivots = 0
    for phase in (1, 2):
        while True:
            selected = solver.select_primal_pivot_column()
            if selected is None:
                break
            column, _ = selected
            row, _ = solver.select_primal_pivot_row(column)
            assert row >= 0
            solver.bring_into_basis(column, row)
            pivots += 1
            if pivots == 7:
                before = solver.objective_function_value()
                assert solver.refactor() < 1e-9
                assert solver.objective_function_value() == pytest.approx(before, rel=1e-12, abs=1e-12)
            assert pivots < 5000
        if phase == 1:
            assert abs(solver.objective_function_value()) <= 1e-7  # feasible: the artificial cost is zero
            solver.begin_phase_two()  # artificials still basic at level zero stay where they are: phase two never prices them
    assert solver.objective_function_value() + fixed_cost == pytest.approx(expected.objective, rel=1e-9, abs=1e-9)
    with pytest.raises(relp_amd.RelpError):
        solver.bring_into_basis(-1, 0)
    whole.close()
    solver.close()


@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "ADLITTLE", "BLEND", "25FV47"])
def test_steepest_edge_weights_carried_from_phase_one(name, monkeypatch):
    """Large LPs keep the steepest-edge weights of phase one for phase two instead of recomputing them (the recurrences are
    exact, so they are what `SteepestDescentAlongObjective::new` would compute: pivot_rule.rs:202-219).  Forced here on small
    LPs: same exact optimum, and weights that agree with a fresh computation at the hand-over."""
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    fresh = relp_amd.Solver(certify=1).load_mps(path)
    expected = fresh.solve_relaxation()
    monkeypatch.setenv("RELP_CARRY_WEIGHTS_MIN", "0")
    carried = relp_amd.Solver(certify=1).load_mps(path)
    result = carried.solve_relaxation()
    assert result.kind == expected.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1
    assert carried.objective_exact() == fresh.objective_exact()
    assert abs(result.pivots_phase_two - expected.pivots_phase_two) <= max(20, expected.pivots_phase_two // 5)
    fresh.close()
    carried.close()


@pytest.mark.parametrize("carry", [relp_amd.api.CARRY_EXPLICIT, relp_amd.api.CARRY_LU, relp_amd.api.CARRY_LU_INVERSE], ids=["explicit", "lu", "lu_inverse"])
def test_after_basis_update_applies_the_pending_weight_update(carry):
    """`PivotRule::after_basis_update` (pivot_rule.rs:243-296) as an entry of its own: the Goldfarb-Reid update is applied once,
    whether the caller asks for it right after the pivot or lets the next pricing pass do it; asking twice changes nothing."""
    path = os.path.join(ROOT, "data", "netlib", "ADLITTLE.SIF")
    general, data = load_problem(path)
    a = relp_amd.Solver(carry=carry, use_graph=0).load_mps(path)
    b = relp_amd.Solver(carry=carry, use_graph=0).load_mps(path)
    a.begin_phase_one()
    b.begin_phase_one()
    tableau = Tableau.new_partially_artificial(data, LUDecomposition)
    rule = SteepestDescentAlongObjective(tableau)
    for step in range(25):
        qa, qb = a.select_primal_pivot_column(), b.select_primal_pivot_column()
        assert qa == qb
        assert np.array_equal(a.gamma(), b.gamma(), equal_nan=True)  # same weights, bit for bit, either way
        if qa is None:
            break
        q = qa[0]
        p, _ = a.select_primal_pivot_row(q)
        pb, _ = b.select_primal_pivot_row(q)
        assert p == pb
        a.bring_into_basis(q, p)
        b.bring_into_basis(q, p)
        a.after_basis_update()
        g1 = a.gamma()
        a.after_basis_update()  # nothing pending any more
        assert np.array_equal(g1, a.gamma(), equal_nan=True)
        # and they are the reference's weights (exact oracle on the same pivots)
        info = tableau.generate_column(q)
        change = tableau.bring_into_basis(q, p, info, tableau.relative_cost(q))
        rule.after_basis_update(change, tableau)
        for j in range(a.n_art, a.n):
            if rule.gamma[j] is not None and not tableau.is_in_basis(j):
                assert g1[j] == pytest.approx(float(rule.gamma[j]), rel=1e-8), (step, j)
    a.close()
    b.close()
