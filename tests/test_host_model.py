"""Host logic of the product (C++ MPS reader, standardisation, MatrixData) against the oracle's restatement.

CPU only: ``relp_model_*`` needs no device.  Every data file is read by both and compared column by column, exactly.
"""
import glob
import os
from fractions import Fraction

import pytest

import relp_amd
from relp_oracle.mps import load_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "data", "*", "*.SIF")) + glob.glob(os.path.join(ROOT, "data", "*", "*.mps")))
UNSUPPORTED = set()
SKIP_ORACLE = {"quadratic_model_data_1.mps", "quadratic_model_data_2.mps", "mixed_model_data_1.mps",
               "mixed_model_data_2.mps", "basis_data_1.mps"}


def name_of(path):
    return os.path.basename(path)


@pytest.mark.parametrize("path", [p for p in FILES if name_of(p) not in UNSUPPORTED | SKIP_ORACLE], ids=name_of)
def test_model_equals_oracle_standard_form(path):
    try:
        general, data = load_problem(path)
    except Exception as error:  # files the reference's parser rejects as well
        with pytest.raises(relp_amd.RelpError):
            relp_amd.Model(path)
        pytest.skip("rejected by both: %s" % error)
    model = relp_amd.Model(path)
    assert model.nr_rows == data.nr_rows()
    assert model.nr_columns == data.nr_columns()
    assert model.nr_constraints == data.nr_constraints()
    assert model.nr_structural == data.nr_normal_variables()
    assert model.group_counts == [data.nr_equality, data.nr_range, data.nr_upper, data.nr_lower]
    assert model.pivot_element_indices() == data.pivot_element_indices()
    step = max(1, model.nr_columns // 400)
    for j in list(range(0, model.nr_columns, step)) + [model.nr_columns - 1]:
        expected = data.column(j)
        got = model.column_exact(j)
        assert [(i, Fraction(n, d)) for i, n, d in got] == expected, j
        assert model.cost_value(j) == pytest.approx(float(data.cost_value(j)), rel=1e-15, abs=0)
    rhs = model.right_hand_side()
    assert list(rhs) == [float(v) for v in data.right_hand_side()]
    assert model.fixed_cost() == pytest.approx(float(general.fixed_cost), rel=1e-15, abs=0)
