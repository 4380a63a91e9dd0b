"""Expected optima asserted by the reference's own integration tests (DATA, with citations).

``(value, tolerance)`` pairs from /root/reference/tests/netlib/test.rs (``(result - RB!(expected)).abs() < RB!(tol)``),
exact rationals from tests/burkardt/test.rs, tests/unicamp/test.rs and tests/cook/test.rs.
"""
import json
import os
from fractions import Fraction as F

def _load_netlib():
    """tests/netlib/test.rs, extracted by tests/golden/make_reference_expectations.py into netlib_expected.json."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "netlib_expected.json")
    return {name: (entry["expected"], entry["tolerance"], entry["ignored"]) for name, entry in json.load(open(path)).items()}


NETLIB = _load_netlib()

EXACT = {  # fixture name -> exact optimal objective
    "burkardt_afiro": F(-406659, 875),                                              # tests/burkardt/test.rs:75
    "burkardt_adlittle": F(24975305659811992079614961229, 120651674036153428931840),  # tests/burkardt/test.rs:53
    "burkardt_maros": F(385, 3),                                                    # tests/burkardt/test.rs:143
    "burkardt_testprob": F(54),                                                     # tests/burkardt/test.rs:185
    "cook_small_example": F(-143, 2),                                               # tests/cook/test.rs:18-37
    "unicamp_model_data_1": F(123, 38), "unicamp_model_data_3_1": F(70), "unicamp_model_data_3_2": F(180),
    "unicamp_model_data_3_3": F(245), "unicamp_model_data_3_4": F(2250), "unicamp_model_data_4": F(7),
    "unicamp_model_data_6": F(28),                                                  # tests/unicamp/test.rs:8-130
}
