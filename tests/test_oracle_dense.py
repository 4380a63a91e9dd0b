"""The numpy f64 restatement for the dense workloads (oracle/f64_dense.py, bench.py's dense cpu_baseline) against the
committed optima of the scaled-down twins of BASELINE config 3 (exact Fraction optimum / HiGHS)."""
import json
import os
import sys
from fractions import Fraction

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from f64_dense import DenseModel  # noqa: E402  (oracle/ is on sys.path via conftest)
from relp_amd.workloads import dense_lp  # noqa: E402


@pytest.mark.parametrize("key", ["16x32", "64x128", "256x512"])
def test_dense_f64_model_reaches_the_committed_optimum(key):
    with open(os.path.join(ROOT, "tests", "golden", "dense_lp.json")) as handle:
        golden = json.load(handle)[key]
    m, n = (int(v) for v in key.split("x"))
    model = DenseModel(*dense_lp(m, n))
    assert model.solve() == "optimal"
    expected = float(Fraction(golden["exact"])) if golden.get("exact") else golden["highs"]
    assert abs(model.objective() - expected) <= 1e-9 * abs(expected)


def test_dense_f64_model_honours_limits():
    model = DenseModel(*dense_lp(64, 128))
    assert model.solve(max_pivots=5) == "limit" and model.pivots == 5
