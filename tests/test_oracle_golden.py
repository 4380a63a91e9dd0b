"""The oracle against the reference's end-to-end expectations and the committed golden fixtures (CPU only)."""
import glob
import json
import os
from fractions import Fraction

import pytest

from reference_expectations import EXACT, NETLIB
from relp_oracle import FiniteOptimum, Unbounded, solve_relaxation, BasisInverseRows, SteepestDescentAlongVariable
from relp_oracle.mps import load_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = {os.path.basename(p)[:-5]: json.load(open(p)) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.json"))}
GOLDEN = {name: g for name, g in GOLDEN.items() if "status" in g}  # per-LP fixtures only
FAST = [name for name, g in GOLDEN.items() if g.get("oracle_seconds", 1e9) < 1.0]


def objective_of(name):
    general, data = load_problem(os.path.join(ROOT, GOLDEN[name]["file"]))
    result = solve_relaxation(data)
    assert isinstance(result, FiniteOptimum)
    return general.objective_of(data.reconstruct_solution(result.solution)), result


@pytest.mark.parametrize("name", sorted(EXACT))
def test_exact_optima_of_the_reference(name):
    objective, _ = objective_of(name)
    assert objective == EXACT[name]


@pytest.mark.parametrize("name", sorted(FAST))
def test_fixture_regenerates(name):
    objective, result = objective_of(name)
    golden = GOLDEN[name]
    assert "%d/%d" % (objective.numerator, objective.denominator) == golden["objective"]
    assert result.basis == golden["basis"]


@pytest.mark.parametrize("name", sorted(n for n in GOLDEN if n in NETLIB))
def test_golden_objectives_meet_reference_tolerances(name):
    expected, tolerance, _ = NETLIB[name]
    num, den = GOLDEN[name]["objective"].split("/")
    value = Fraction(int(num), int(den))
    if name == "25FV47":
        # tests/netlib/test.rs:10 holds the optimum rounded to 8 digits (5.5018459e+03) with tolerance 1e-5, but the true
        # optimum 5.5018458883E+03 (tests/netlib/problem_files/README:85) is 1.2e-5 away: that ignored test cannot pass as
        # written.  The README value is the expectation here.
        assert abs(value - Fraction("5501.8458883")) < Fraction(1, 10 ** 6)
        assert abs(value - Fraction(expected)) < Fraction(2, 10 ** 5)
        return
    assert abs(value - Fraction(expected)) < Fraction(tolerance)


def test_unbounded_nazareth():  # tests/burkardt/test.rs:157-167
    _, data = load_problem(os.path.join(ROOT, "data", "burkardt", "nazareth.mps"))
    assert isinstance(solve_relaxation(data), Unbounded)


AFIRO_VALUES = {  # tests/burkardt/test.rs:77-109 (non-zero entries; every other listed variable is zero)
    "X01": Fraction(80), "X02": Fraction(51, 2), "X03": Fraction(109, 2), "X04": Fraction(424, 5),
    "X06": Fraction(255, 14), "X14": Fraction(255, 14), "X16": Fraction(999), "X22": Fraction(500),
    "X23": Fraction(11898, 25), "X24": Fraction(602, 25), "X26": Fraction(215), "X36": Fraction(11898, 35),
    "X37": Fraction(11898, 35),
}


def test_afiro_solution_values():
    """tests/burkardt/test.rs:74-112: exact objective and `is_probably_equal_to(.., 0.1)`
    (data/linear_program/solution.rs:47-79: more than 10 % of the values equal)."""
    general, data = load_problem(os.path.join(ROOT, "data", "burkardt", "afiro.mps"))
    result = solve_relaxation(data)
    assert general.objective_of(data.reconstruct_solution(result.solution)) == Fraction(-406659, 875)
    x = general.full_solution(data.reconstruct_solution(result.solution))
    assert len(x) == 32
    equal = sum(1 for name, value in x.items() if AFIRO_VALUES.get(name, Fraction(0)) == value)
    assert equal / len(x) > 0.1


@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "burkardt_afiro"])
def test_other_inverse_and_rule_reach_the_same_optimum(name):
    general, data = load_problem(os.path.join(ROOT, GOLDEN[name]["file"]))
    result = solve_relaxation(data, BasisInverseRows, SteepestDescentAlongVariable)
    objective = general.objective_of(data.reconstruct_solution(result.solution))
    assert "%d/%d" % (objective.numerator, objective.denominator) == GOLDEN[name]["objective"]
