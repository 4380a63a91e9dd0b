"""The LU carry (`Carry<F, LUDecomposition<F>>`, relp_options.carry = RELP_CARRY_LU) inside the device-resident loop (``-m gpu``).

Whole solves through the Forrest-Tomlin / refactorisation cycle on every golden LP, with the exact certificate; the
step-by-step comparisons with the oracle are the `lu` parametrisations in tests/test_gpu_parity.py.
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


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_lu_carry_reaches_the_exact_optimum(name):
    golden = GOLDEN[name]
    solver = relp_amd.Solver(carry=LU, certify=1).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    num, den = golden["objective"].split("/")
    exact = Fraction(int(num), int(den))
    assert abs(result.objective - float(exact)) <= REL * max(1.0, abs(float(exact))), (result.objective, float(exact))
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    assert solver.objective_exact() == golden["objective"]  # bit-exact with the reference's RationalBig optimum
    pivots = result.pivots_phase_one + result.pivots_phase_two
    assert result.refactors >= pivots // 32  # `should_refactor` every 31 updates (lower_upper/mod.rs:249-252)
    solver.close()


@pytest.mark.parametrize("period", [1, 5, 31, 63])
def test_refactor_period(period):
    """Any period gives the same optimum; period p means p Forrest-Tomlin updates between refactorisations."""
    path = os.path.join(ROOT, "data", "netlib", "SHARE2B.SIF")
    solver = relp_amd.Solver(carry=LU, refactor_period=period).load_mps(path)
    result = solver.solve_relaxation()
    expected = GOLDEN["SHARE2B"]["objective_float"]
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective - expected) <= REL * abs(expected)
    pivots = result.pivots_phase_one + result.pivots_phase_two
    assert result.refactors >= pivots // (period + 1)
    solver.close()


def test_graph_and_plain_launches_agree_with_the_lu_carry():
    path = os.path.join(ROOT, "data", "netlib", "SHARE2B.SIF")
    a = relp_amd.Solver(use_graph=1, carry=LU).load_mps(path).solve_relaxation()
    b = relp_amd.Solver(use_graph=0, carry=LU).load_mps(path).solve_relaxation()
    assert a.objective == b.objective  # deterministic: fixed summation orders everywhere
    assert (a.pivots_phase_one, a.pivots_phase_two) == (b.pivots_phase_one, b.pivots_phase_two)


def test_lu_carry_rejects_what_it_does_not_support():
    with pytest.raises(relp_amd.RelpError) as e:
        relp_amd.Solver(carry=LU, implicit_bounds=1).load_mps(os.path.join(ROOT, "data", "netlib", "BOEING1.SIF"))
    assert e.value.status == relp_amd.api.ERR_ARGUMENT


def test_both_carries_walk_the_same_vertices_on_a_small_lp():
    """Same pricing, same ratio test, same tie rules: on an LP where f64 leaves no room for different choices the two
    `BasisInverse`s make the same pivots."""
    path = os.path.join(ROOT, "data", "netlib", "AFIRO.SIF")
    a = relp_amd.Solver(carry=relp_amd.api.CARRY_EXPLICIT).load_mps(path)
    b = relp_amd.Solver(carry=LU).load_mps(path)
    ra, rb = a.solve_relaxation(), b.solve_relaxation()
    assert (ra.pivots_phase_one, ra.pivots_phase_two) == (rb.pivots_phase_one, rb.pivots_phase_two)
    assert np.array_equal(a.basis(), b.basis())
    assert ra.objective == pytest.approx(rb.objective, rel=1e-13)


@pytest.mark.parametrize("name", ["BNL2", "CYCLE", "CZPROB", "GREENBEA", "GREENBEB", "MODSZK1"])
def test_lu_carry_on_the_largest_lps_it_takes(name):
    """The Netlib LPs with 600 < m <= 2800 rows (hundreds of refactorisation cycles each): the reference's expected optimum within
    its tolerance, and the exact certificate.  (`tools/lu_netlib_scan.py` runs all 47 shipped LPs that fit the LDS-resident size.)"""
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))[name]
    solver = relp_amd.Solver(carry=LU, certify=1).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1
    assert abs(result.objective - expected["expected"]) <= max(expected["tolerance"], REL * abs(expected["expected"]))
    assert result.refactors >= (result.pivots_phase_one + result.pivots_phase_two) // 32
    solver.close()
