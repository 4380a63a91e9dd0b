"""Bit-exact rational optimum through the exact certificate (``options.certify``), on a real MI355X (``-m gpu``).

The expected strings are the oracle's exact optima (tests/golden/*.json, produced by the Fraction restatement of
relp's RationalBig path) and the exact values the reference's own tests assert (tests/reference_expectations.py).
Bit-exact comparison of "numerator/denominator".
"""
import glob
import json
import os

import pytest

import relp_amd
from reference_expectations import EXACT

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = {os.path.basename(p)[:-5]: json.load(open(p)) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.json"))}
GOLDEN = {name: g for name, g in GOLDEN.items() if "status" in g}  # per-LP fixtures only


@pytest.mark.parametrize("name", sorted(n for n, g in GOLDEN.items() if g["status"] == "optimal"))
def test_certified_objective_is_bit_exact(name):
    golden = GOLDEN[name]
    solver = relp_amd.Solver(certify=1).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    assert solver.objective_exact() == golden["objective"]
    if name in EXACT:  # the reference's own exact assertions
        value = EXACT[name]
        assert solver.objective_exact() == "%d/%d" % (value.numerator, value.denominator)
    solver.close()


def test_25fv47_exact_optimum_has_1791_bits():
    golden = GOLDEN["25FV47"]
    solver = relp_amd.Solver(certify=1).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    assert result.certified == 1
    exact = solver.objective_exact()
    assert exact == golden["objective"]
    num, den = exact.split("/")
    assert max(int(num).bit_length(), int(den).bit_length()) == golden["objective_bits"] == 1791
    assert result.certify_seconds < 30
    solver.close()


def test_uncertified_solve_has_no_exact_objective():
    solver = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", "AFIRO.SIF"))
    solver.solve_relaxation()
    with pytest.raises(relp_amd.RelpError):
        solver.objective_exact()


@pytest.mark.parametrize("name, tol", [("BLEND", 0.05), ("E226", 0.01), ("SC50A", 0.01), ("ADLITTLE", 0.01), ("SCAGR7", 0.01), ("BRANDY", 0.001)])
def test_exact_repair_pivots_fix_a_suboptimal_f64_basis(name, tol):
    """A deliberately sloppy f64 solve (huge dual tolerance) stops on a non-optimal basis; the certificate must notice
    (exact signs) and repair it with exact simplex pivots, still returning the bit-exact optimum."""
    golden = GOLDEN[name]
    solver = relp_amd.Solver(certify=1, tol_dual=tol).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    if result.kind == relp_amd.INFEASIBLE:
        pytest.skip("the sloppy tolerance already stops phase one")
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    assert solver.objective_exact() == golden["objective"]
    expected = float(int(golden["objective"].split("/")[0])) / float(int(golden["objective"].split("/")[1]))
    if abs(result.objective - expected) > 1e-9 * abs(expected):
        assert result.exact_repair_pivots > 0  # the f64 vertex was not optimal, so exact pivots were needed
    solver.close()


def test_unbounded_is_certified_by_an_exact_ray():
    """The reference decides `Unbounded` exactly (phase_two.rs:53; tests/burkardt/test.rs:157-167).  With `certify` the f64
    verdict is proved: x_B >= 0, cbar_q < 0 and B^-1 a_q <= 0 in exact arithmetic."""
    solver = relp_amd.Solver(certify=1).load_mps(os.path.join(ROOT, "data", "burkardt", "nazareth.mps"))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.UNBOUNDED
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    assert solver.objective_exact() == "-inf"
    solver.close()


@pytest.mark.parametrize("b, expected", [([1, 2], "1/1"), ([3, 10], "7/1"), ([(1, 3), (5, 7)], "8/21")])
def test_infeasible_is_certified_by_a_farkas_vector(b, expected):
    """x0 <= b0 and x0 >= b1 > b0: infeasible (phase_one.rs:171-173, decided exactly by the reference).  With `certify` the
    phase-one dual solution is checked exactly -- y'A <= 0, y'b > 0 -- and the exact phase-one optimum (the distance b1 - b0)
    comes back."""
    solver = relp_amd.Solver(certify=1)
    solver.load_matrix_data([0, 2], [0, 1], [1, 1], [1, 1], b=b, cost=[1], counts=(0, 0, 1, 1))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.INFEASIBLE
    assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h)
    assert solver.objective_exact() == expected
    solver.close()


def test_a_wrong_infeasible_verdict_is_not_certified():
    """A sloppy feasibility tolerance turns a feasible LP into an f64 `Infeasible`; the exact check refuses to confirm it."""
    golden = GOLDEN["SC50A"]
    solver = relp_amd.Solver(certify=1, tol_dual=50.0).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    if result.kind == relp_amd.INFEASIBLE:
        assert result.certified == 0
    solver.close()
