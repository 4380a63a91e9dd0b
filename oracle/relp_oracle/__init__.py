"""Exact CPU oracle: a restatement of relp's two-phase revised simplex hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import anything below ``oracle/``.  The product
(``relp_amd``) never imports it and has no CPU fallback.

Arithmetic: ``fractions.Fraction`` over Python integers stands in for ``relp_num::RationalBig``
(crate ``relp-num = "0.1.13"``, un-vendored).  Exact rational equality is
representation independent, so the known-answer tests of the reference (``#[test]`` functions
under ``/root/reference/src``) pin this restatement directly: see ``tests/test_oracle_kat.py``.

Parity status: PINNED by the reference's own known-answer tests (exact LU factors, FTRAN/BTRAN,
Forrest-Tomlin 4x4/5x5, eta files, permutations, tableau, pivot rule, phase one, six end-to-end
LPs, Burkardt/Unicamp exact optima, Netlib objectives within the reference's tolerances).  The
reference itself cannot be built here (Rust nightly; no toolchain).

Every function cites the reference ``file:line`` it follows (paths relative to
``/root/reference/src/algorithm/two_phase`` unless stated otherwise).
"""
from fractions import Fraction

from .lu import EtaFile, LUDecomposition, ColumnAndSpike  # noqa: F401
from .permutation import FullPermutation, RotateToBack, Swap  # noqa: F401
from .inverse_rows import BasisInverseRows  # noqa: F401
from .carry import Carry  # noqa: F401
from .provider import MatrixData, RemoveRows, Variable  # noqa: F401
from .tableau import Tableau, Fully, Partially, NonArtificial  # noqa: F401
from .pivot_rule import (  # noqa: F401
    FirstProfitable, FirstProfitableWithMemory, SteepestDescentAlongVariable,
    SteepestDescentAlongObjective,
)
from .solve import (  # noqa: F401
    phase_one_primal, phase_two_primal, solve_relaxation, solve_relaxation_full_basis,
    Infeasible, Unbounded, FiniteOptimum,
)

F = Fraction
