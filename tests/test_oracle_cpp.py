"""The C++ CPU oracle (oracle/cpp) against the Fraction oracle and the committed golden vectors.

The C++ restatement is what bench.py times as ``cpu_baseline``; it must walk exactly the reference's pivot sequence,
so it is pinned to the same fixtures as the Python oracle (which the reference's known-answer tests pin).
"""
import glob
import json
import math
import os
import random
import subprocess
from fractions import Fraction

import pytest

from relp_oracle import FiniteOptimum, Infeasible, Unbounded, solve_relaxation, solve_relaxation_full_basis
from relp_oracle import cpu
from relp_oracle.mps import load_problem
from relp_oracle.pivot_rule import (FirstProfitable, FirstProfitableWithMemory, SteepestDescentAlongVariable)
from relp_oracle.provider import MatrixData, Variable
from relp_oracle.solve import Trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_rational_arithmetic_matches_python():
    cpu.ensure_built()
    rng = random.Random(20261003)
    sizes = [1, 8, 30, 61, 62, 63, 64, 65, 100, 127, 128, 129, 200, 500, 1000, 2000]

    def natural():
        return rng.getrandbits(rng.choice(sizes))

    def fraction():
        return Fraction(natural() * rng.choice([1, -1]), natural() + 1)

    lines, expected = [], []
    for _ in range(3000):
        op = rng.choice("+-*/cg")
        if op == "g":
            g = natural() + 1
            a, b = (natural() + 1) * g, (natural() + 1) * g
            if rng.random() < 0.1:
                b = a * rng.randint(1, 5)
            lines.append("g %d %d" % (a, b))
            expected.append(str(math.gcd(a, b)))
            continue
        x, y = fraction(), fraction()
        if rng.random() < 0.1:
            y = x * rng.choice([1, -1, 2])
        if rng.random() < 0.1:
            y = Fraction(y.numerator, x.denominator * rng.randint(1, 3))
        if op == "/" and y == 0:
            y = Fraction(1)
        lines.append("%s %d/%d %d/%d" % (op, x.numerator, x.denominator, y.numerator, y.denominator))
        if op == "c":
            expected.append(str((x > y) - (x < y)))
        else:
            r = x + y if op == "+" else x - y if op == "-" else x * y if op == "*" else x / y
            expected.append("%d/%d" % (r.numerator, r.denominator))
    done = subprocess.run([cpu.RATIONAL_CHECK], input="\n".join(lines) + "\n", capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    assert done.stdout.split() == expected


def _fixtures(max_seconds):
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN, "*.json"))):
        with open(path) as handle:
            g = json.load(handle)
        if isinstance(g, dict) and "status" in g and g.get("oracle_seconds", 1e9) <= max_seconds:
            out.append(g)
    return out


@pytest.mark.parametrize("golden", _fixtures(1.5), ids=lambda g: g["name"])
def test_cpp_oracle_reproduces_golden_fixture(golden):
    """Same status, pivot counts, pivot sequence, final basis and exact optimum as the committed vectors."""
    general, data = load_problem(os.path.join(ROOT, golden["file"]))
    record = cpu.solve_provider(data)
    assert record["status"] == golden["status"]
    assert (record["pivots_phase1"], record["pivots_phase2"]) == (golden["pivots_phase1"], golden["pivots_phase2"])
    assert record["trace_head"] == golden["trace_head"]
    if golden["status"] == "optimal":
        assert record["basis"] == golden["basis"]
        objective = general.objective_of(data.reconstruct_solution(record["solution"]))
        assert "%d/%d" % (objective.numerator, objective.denominator) == golden["objective"]


def test_tuned_mode_walks_the_same_path():
    general, data = load_problem(os.path.join(ROOT, "data", "netlib", "SHARE2B.SIF"))
    faithful = cpu.solve_provider(data, trace=1000)
    tuned = cpu.solve_provider(data, trace=1000, tuned=True)
    for key in ("status", "trace_head", "basis", "solution", "objective"):
        assert faithful[key] == tuned[key]


@pytest.mark.parametrize("rule,rule_cls", [("dantzig", SteepestDescentAlongVariable), ("first", FirstProfitable),
                                           ("memory", FirstProfitableWithMemory)])
@pytest.mark.parametrize("name", ["AFIRO", "SC50B", "KB2"])
def test_other_pivot_rules_match_python_oracle(name, rule, rule_cls):
    general, data = load_problem(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    trace = Trace()
    exact = solve_relaxation(data, rule_cls=rule_cls, trace=trace)
    assert isinstance(exact, FiniteOptimum)
    record = cpu.solve_provider(data, rule=rule, trace=100000)
    assert record["trace_head"] == [[ph, q, p, lv] for ph, q, p, lv, _ in trace.pivots]
    assert record["solution"] == exact.solution
    assert record["basis"] == exact.basis


def _tiny(rows, b, costs, nr_upper, nr_lower=0, nr_equality=0):
    n = len(costs)
    columns = [[(i, Fraction(rows[i][j])) for i in range(len(rows)) if rows[i][j] != 0] for j in range(n)]
    return MatrixData(columns, b, [], nr_equality, 0, nr_upper, nr_lower, [Variable(c) for c in costs])


def test_infeasible_unbounded_and_full_basis_routes():
    # x1 + x2 <= 1 and x1 + x2 >= 2: infeasible
    infeasible = _tiny([[1, 1], [1, 1]], [1, 2], [1, 1], nr_upper=1, nr_lower=1)
    assert isinstance(solve_relaxation(infeasible), Infeasible)
    assert cpu.solve_provider(infeasible)["status"] == "infeasible"
    # min -x1 with x1 - x2 <= 1: unbounded
    unbounded = _tiny([[1, -1], [-1, -1]], [1, 0], [-1, 0], nr_upper=2)
    assert isinstance(solve_relaxation(unbounded), Unbounded)
    assert cpu.solve_provider(unbounded)["status"] == "unbounded"
    # all-slack start (FullInitialBasis, two_phase/mod.rs:80-109)
    rng = random.Random(3)
    m, n = 6, 9
    rows = [[rng.randint(1, 9) for _ in range(n)] for _ in range(m)]
    dense = _tiny(rows, [rng.randint(50, 90) for _ in range(m)], [-rng.randint(1, 9) for _ in range(n)], nr_upper=m)
    trace = Trace()
    exact = solve_relaxation_full_basis(dense, trace=trace)
    record = cpu.solve_provider(dense, route="full_basis", trace=1000)
    assert record["status"] == "optimal" and record["pivots_phase1"] == 0
    assert record["trace_head"] == [[ph, q, p, lv] for ph, q, p, lv, _ in trace.pivots]
    assert record["solution"] == exact.solution and record["objective"] == exact.objective


def test_redundant_rows_are_removed():
    """phase_one.rs:232-278 / generic_wrapper.rs: duplicate equality rows leave an artificial that cannot be pivoted out."""
    redundant = _tiny([[1, 1, 0], [1, 1, 0], [0, 1, 1]], [2, 2, 3], [1, 2, 3], nr_upper=0, nr_equality=3)
    exact = solve_relaxation(redundant)
    record = cpu.solve_provider(redundant)
    assert isinstance(exact, FiniteOptimum) and record["status"] == "optimal"
    assert record["solution"] == exact.solution and record["objective"] == exact.objective


def test_pivot_limit_stops_early():
    general, data = load_problem(os.path.join(ROOT, "data", "netlib", "ADLITTLE.SIF"))
    record = cpu.solve_provider(data, max_pivots=10)
    assert record["status"] == "pivot_limit" and record["pivots_phase1"] + record["pivots_phase2"] == 10
