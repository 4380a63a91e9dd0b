"""Presolve + device solve (``-m gpu``): the harness order of the reference (tests/netlib/mod.rs:55-71: parse, presolve,
standardize, solve, back-map).  The presolved LP must reach the reference's optimum; where a golden exact optimum exists
the certificate must reproduce it bit for bit; removed variables must come back in the original solution."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

import relp_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")) as handle:
    EXPECTED = json.load(handle)
NAMES = sorted(n for n, e in EXPECTED.items() if os.path.exists(os.path.join(ROOT, "data", "netlib", n + ".SIF"))
               and (not e["ignored"] or "intensive" in e["ignored"]))


@pytest.mark.parametrize("name", NAMES)
def test_presolved_netlib_problem_reaches_the_reference_optimum(name):
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    golden_path = os.path.join(ROOT, "tests", "golden", name + ".json")
    solver = relp_amd.Solver(certify=1 if os.path.exists(golden_path) else 0)
    # (BORE3D, CYCLE, GREENBEB: the reference's bound tightening leaves the 128-bit host model; the product presolves them again
    #  without the implied bounds that need more than 126 / 60 bits -- still presolved, same optimum -- instead of failing with
    #  RELP_ERR_OVERFLOW)
    solver.load_mps(path, presolve=True)
    plain = relp_amd.Model(path)
    assert solver.m <= plain.nr_rows
    if name in ("BORE3D", "CYCLE", "GREENBEB"):
        assert solver.m < plain.nr_rows
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    entry = EXPECTED[name]
    tolerance = max(entry["tolerance"], 2e-5 if name == "25FV47" else 0.0)
    assert abs(result.objective - entry["expected"]) <= tolerance
    if os.path.exists(golden_path):
        with open(golden_path) as handle:
            golden = json.load(handle)
        # tightened bounds carry ~100-bit rationals into the right-hand side (BANDM): the certificate lifts such a
        # right-hand side limb by limb
        assert result.certified == 1, relp_amd.lib().relp_last_error(solver._h).decode()
        assert Fraction(solver.objective_exact()) == Fraction(golden["objective"])
    solver.close()


@pytest.mark.parametrize("name", ["AFIRO", "ADLITTLE", "BLEND", "KB2", "BOEING2", "VTP-BASE", "RECIPELP"])
def test_original_solution_is_feasible_and_optimal_in_the_file(name):
    """The solution mapped back to the file's variables (shifts, flips, free splits and presolve removals undone) must
    satisfy the file's rows and bounds and have the optimal objective, with and without presolve."""
    from relp_oracle.mps import GeneralForm, parse
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    with open(path) as handle:
        general = GeneralForm.from_mps(parse(handle.read(), fixed=True))  # the general form before any transformation
    n = len(general.variables)
    cost = np.array([float(v.cost) for v in general.variables])
    for presolve in (False, True):
        solver = relp_amd.Solver().load_mps(path, presolve=presolve)
        result = solver.solve_relaxation()
        assert result.kind == relp_amd.FINITE_OPTIMUM
        x = solver.original_solution()
        assert len(x) == n
        assert abs(float(cost @ x) - result.objective) <= 1e-7 * max(1.0, abs(result.objective))
        for j, v in enumerate(general.variables):
            assert v.lower_bound is None or x[j] >= float(v.lower_bound) - 1e-7
            assert v.upper_bound is None or x[j] <= float(v.upper_bound) + 1e-7
        activity = np.zeros(len(general.b))
        for j, column in enumerate(general.columns):
            for i, value in column:
                activity[i] += float(value) * x[j]
        for i, kind in enumerate(general.constraint_types):
            b = float(general.b[i])
            slack = 1e-6 * max(1.0, abs(b))
            if kind == "Equal":
                assert abs(activity[i] - b) <= slack
            elif kind == "Less":
                assert activity[i] <= b + slack
            elif kind == "Greater":
                assert activity[i] >= b - slack
            else:
                assert b - float(kind[1]) - slack <= activity[i] <= b + slack
        solver.close()


@pytest.mark.parametrize("name", ["AFIRO", "ADLITTLE", "BOEING2", "KB2", "RECIPELP", "SCTAP1", "ETAMACRO", "FINNIS", "GFRD-PNC", "80BAU3B"])
def test_presolve_and_implicit_bounds_together(name):
    """The two options combined (the presolve tightens and adds bounds, the bounded-variable simplex then takes them out of the
    rows): same certified rational optimum as without either."""
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    plain = relp_amd.Solver(certify=1).load_mps(path)
    expected = plain.solve_relaxation()
    both = relp_amd.Solver(certify=1, implicit_bounds=1).load_mps(path, presolve=True)
    result = both.solve_relaxation()
    assert expected.kind == result.kind == relp_amd.FINITE_OPTIMUM and expected.certified == result.certified == 1
    assert both.objective_exact() == plain.objective_exact()
    assert both.m <= plain.m
    plain.close()
    both.close()


@pytest.mark.gpu
def test_per_lp_record():
    """One JSON object per solved LP: dimensions, pivots per phase, times, exact objective."""
    path = os.path.join(ROOT, "data", "netlib", "AFIRO.SIF")
    solver = relp_amd.Solver(certify=1).load_mps(path)
    result = solver.solve_relaxation()
    record = solver.record()
    assert (record["m"], record["n"], record["nnz"]) == (27, 51, 102)
    assert record["result"] == "finite_optimum" and record["certified"] is True
    assert record["pivots_phase_one"] == result.pivots_phase_one and record["pivots_phase_two"] == result.pivots_phase_two
    assert record["objective_exact"] == "-406659/875"           # tests/burkardt/test.rs:75
    assert record["objective"] == pytest.approx(-464.753142857, rel=1e-9)
    assert record["pivots_per_second"] > 0 and record["solve_seconds"] > 0
    assert record["presolve"] == "off"
    solver.close()
    for name, state in (("AFIRO", "applied"), ("BORE3D", "applied without the implied bounds beyond 126 bits")):
        solver = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"), presolve=True)
        solver.solve_relaxation()
        assert solver.record()["presolve"] == state
        solver.close()
