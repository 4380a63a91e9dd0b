"""The LU carries (`Carry<F, LUDecomposition<F>>`) inside the device-resident loop (``-m gpu``): relp_options.carry =
RELP_CARRY_LU (level-by-level triangular solves, Forrest-Tomlin updates) and RELP_CARRY_LU_INVERSE (the same factorisation applied
through the sparse inverses of its triangles, product-form updates on top; round 3).

Whole solves through the update / refactorisation cycle on every golden LP, with the exact certificate; the step-by-step
comparisons with the oracle are the `lu` parametrisations in tests/test_gpu_parity.py.
"""
import glob
import json
import os
from fractions import Fraction

import numpy as np
import pytest

import relp_amd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = {os.path.basename(p)[:-5]: json.load(open(p)) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.json"))}
GOLDEN = {name: g for name, g in GOLDEN.items() if g.get("status") == "optimal"}
REL = 1e-9
LU = relp_amd.api.CARRY_LU
LU_INVERSE = relp_amd.api.CARRY_LU_INVERSE
BOTH = pytest.mark.parametrize("carry", [LU, LU_INVERSE], ids=["lu", "lu_inverse"])


@BOTH
@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_lu_carry_reaches_the_exact_optimum(name, carry):
    golden = GOLDEN[name]
    solver = relp_amd.Solver(carry=carry, certify=1).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    num, den = golden["objective"].split("/")
    exact = Fraction(int(num), int(den))
    assert abs(result.objective - float(exact)) <= REL * max(1.0, abs(float(exact))), (result.objective, float(exact))
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    assert solver.objective_exact() == golden["objective"]  # bit-exact with the reference's RationalBig optimum
    pivots = result.pivots_phase_one + result.pivots_phase_two
    # `should_refactor` every 31 updates (lower_upper/mod.rs:249-252); the inverse-factor form's default is 47
    assert result.refactors >= pivots // (32 if carry == LU else 48)
    solver.close()


@BOTH
@pytest.mark.parametrize("period", [1, 5, 31, 63])
def test_refactor_period(period, carry):
    """Any period gives the same optimum; period p means p updates between refactorisations."""
    path = os.path.join(ROOT, "data", "netlib", "SHARE2B.SIF")
    solver = relp_amd.Solver(carry=carry, refactor_period=period).load_mps(path)
    result = solver.solve_relaxation()
    expected = GOLDEN["SHARE2B"]["objective_float"]
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective - expected) <= REL * abs(expected)
    pivots = result.pivots_phase_one + result.pivots_phase_two
    assert result.refactors >= pivots // (period + 1)
    solver.close()


@BOTH
def test_graph_and_plain_launches_agree_with_the_lu_carry(carry):
    path = os.path.join(ROOT, "data", "netlib", "SHARE2B.SIF")
    a = relp_amd.Solver(use_graph=1, carry=carry).load_mps(path).solve_relaxation()
    b = relp_amd.Solver(use_graph=0, carry=carry).load_mps(path).solve_relaxation()
    assert a.objective == b.objective  # deterministic: fixed summation orders everywhere
    assert (a.pivots_phase_one, a.pivots_phase_two) == (b.pivots_phase_one, b.pivots_phase_two)


BOUNDED = ["BOEING1", "BOEING2", "ETAMACRO", "FINNIS", "GFRD-PNC", "STANDATA", "STANDMPS", "VTP-BASE", "RECIPELP", "CAPRI", "80BAU3B", "BORE3D"]


@BOTH
@pytest.mark.parametrize("name", BOUNDED)
def test_lu_carry_with_implicit_bounds(name, carry):
    """The bounded ratio test (leaving at the upper bound, bound flips, complemented columns with the opposite sign in the
    factors) inside the LU pivot kernel: the reference's optimum within its tolerance, certified on the basis mapped back to
    the reference's formulation."""
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))[name]
    solver = relp_amd.Solver(carry=carry, implicit_bounds=1, certify=1).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective - expected["expected"]) <= max(expected["tolerance"], 1e-9 * abs(expected["expected"])), (result.objective, expected)
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    plain = relp_amd.Solver(carry=relp_amd.api.CARRY_EXPLICIT, certify=1).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    plain.solve_relaxation()
    assert solver.objective_exact() == plain.objective_exact()
    solver.close()
    plain.close()


def test_lu_carry_takes_the_rows_the_old_lds_limit_excluded():
    """80BAU3B in the reference's formulation: 5746 rows (round 2's LU kernels stopped at about 3200)."""
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))["80BAU3B"]
    solver = relp_amd.Solver(carry=LU).load_mps(os.path.join(ROOT, "data", "netlib", "80BAU3B.SIF"))
    assert solver.m > 5000
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and abs(result.objective - expected["expected"]) <= expected["tolerance"]
    solver.close()


def test_lu_carry_rejects_what_does_not_fit_its_lds():
    from relp_amd.workloads import max_flow_graph
    tail, head, capacity = max_flow_graph(2048, 16384)
    model = relp_amd.Model.max_flow(2048, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, 2047)  # 18 k rows
    with pytest.raises(relp_amd.RelpError) as e:
        relp_amd.Solver(carry=LU).load_model(model)
    assert e.value.status == relp_amd.api.ERR_ARGUMENT


@BOTH
def test_both_carries_walk_the_same_vertices_on_a_small_lp(carry):
    """Same pricing, same ratio test, same tie rules: on an LP where f64 leaves no room for different choices the two
    `BasisInverse`s make the same pivots."""
    path = os.path.join(ROOT, "data", "netlib", "AFIRO.SIF")
    a = relp_amd.Solver(carry=relp_amd.api.CARRY_EXPLICIT).load_mps(path)
    b = relp_amd.Solver(carry=carry).load_mps(path)
    ra, rb = a.solve_relaxation(), b.solve_relaxation()
    assert (ra.pivots_phase_one, ra.pivots_phase_two) == (rb.pivots_phase_one, rb.pivots_phase_two)
    assert np.array_equal(a.basis(), b.basis())
    assert ra.objective == pytest.approx(rb.objective, rel=1e-13)


@BOTH
@pytest.mark.parametrize("name", ["BNL2", "CYCLE", "CZPROB", "GREENBEA", "GREENBEB", "MODSZK1"])
def test_lu_carry_on_the_largest_lps_it_takes(name, carry):
    """The Netlib LPs with 600 < m <= 2800 rows (hundreds of refactorisation cycles each): the reference's expected optimum within
    its tolerance, and the exact certificate.  (`tools/lu_netlib_scan.py` runs all 47 shipped LPs that fit the LDS-resident size.)"""
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))[name]
    solver = relp_amd.Solver(carry=carry, certify=1).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1
    assert abs(result.objective - expected["expected"]) <= max(expected["tolerance"], REL * abs(expected["expected"]))
    assert result.refactors >= (result.pivots_phase_one + result.pivots_phase_two) // (32 if carry == LU else 48)
    solver.close()


def test_inverse_factor_carry_with_three_vectors_in_lds():
    """Beyond about 4300 rows the four LDS vectors of the inverse-factor form do not fit; up to about 5800 it runs with three (the
    two right-hand sides of the BTRAN go through the factors one after the other): 80BAU3B in the reference's formulation, 5746 rows."""
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))["80BAU3B"]
    solver = relp_amd.Solver(carry=LU_INVERSE, certify=1).load_mps(os.path.join(ROOT, "data", "netlib", "80BAU3B.SIF"))
    assert solver.m > 5000
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1
    assert abs(result.objective - expected["expected"]) <= expected["tolerance"]
    solver.close()


def test_inverse_factor_carry_rejects_what_does_not_fit_its_lds():
    """... and beyond that: an error, no silent switch."""
    from relp_amd.workloads import max_flow_graph
    tail, head, capacity = max_flow_graph(1024, 8192)
    model = relp_amd.Model.max_flow(1024, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, 1023)  # ~9 k rows
    with pytest.raises(relp_amd.RelpError) as e:
        relp_amd.Solver(carry=LU_INVERSE).load_model(model)
    assert e.value.status == relp_amd.api.ERR_ARGUMENT
