"""Host side of the LU basis factorisation (relp_amd/csrc/lu_host.hpp) -- CPU tests, no device.

* the reference's exact-factor known-answer tests (lower_upper/decomposition/mod.rs:319-438): with ``reference_ties`` and
  threshold 0 the C++ factorisation must produce the reference's L, U, diagonal and both permutations (all values are dyadic,
  so f64 is exact);
* the same pivot ORDER as the exact oracle (``oracle/relp_oracle/lu.py``, itself pinned by those tests) on the reference's
  13 inverse-property matrices (:454-651) and on random sparse matrices;
* ``P B Q = L U`` for the threshold-pivoting mode the product uses.
"""
import json
import os
import random
import sys
from fractions import Fraction as F

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from relp_amd.basis_inverse import lu_factor_host  # noqa: E402
from relp_oracle import LUDecomposition  # noqa: E402


def columns_of_rows(rows, m):
    columns = [[] for _ in range(m)]
    for i, row in enumerate(rows):
        for j, v in row:
            columns[j].append((i, v))
    return columns


def reference_layout(f):
    """lower_triangular / upper_triangular by column, as lower_upper/mod.rs:36-58 stores them."""
    m = len(f["diag"])
    lower = [[] for _ in range(m - 1)]
    upper = [[] for _ in range(m - 1)]
    for i, row in enumerate(f["lower_rows"]):
        for j, v in row:
            lower[j].append((i, v))
    for i, row in enumerate(f["upper_rows"]):
        for j, v in row:
            upper[j - 1].append((i, v))
    return lower, upper


# decomposition/mod.rs:319-438 as data: (rows, row_permutation, column_permutation, lower, upper, diagonal)
EXACT_FACTOR_CASES = [
    ("identity_2", [[(0, 1)], [(1, 1)]], [0, 1], [0, 1], [[]], [[]], [1, 1]),
    ("identity_3", [[(0, 1)], [(1, 1)], [(2, 1)]], [0, 1, 2], [0, 1, 2], [[], []], [[], []], [1, 1, 1]),
    ("offdiagonal_2_upper", [[(0, 1), (1, 1)], [(1, 1)]], [0, 1], [0, 1], [[]], [[(0, 1)]], [1, 1]),
    ("offdiagonal_2_lower", [[(0, 1)], [(0, 1), (1, 1)]], [0, 1], [0, 1], [[(1, 1)]], [[]], [1, 1]),
    ("offdiagonal_2_both", [[(0, 1), (1, 1)], [(0, 1)]], [1, 0], [0, 1], [[(1, 1)]], [[]], [1, 1]),
    ("wikipedia_example", [[(0, 4), (1, 3)], [(0, 6), (1, 3)]], [0, 1], [0, 1], [[(1, 1.5)]], [[(0, 3)]], [4, -1.5]),
    ("wikipedia_example2", [[(0, -1), (1, 1.5)], [(0, 1), (1, -1)]], [0, 1], [0, 1], [[(1, -1)]], [[(0, 1.5)]], [-1, 0.5]),
]


@pytest.mark.parametrize("case", EXACT_FACTOR_CASES, ids=[c[0] for c in EXACT_FACTOR_CASES])
def test_reference_exact_factors(case):
    _, rows, rp, cp, lower, upper, diag = case
    m = len(rows)
    f = lu_factor_host(columns_of_rows(rows, m), pivot_threshold=0.0, reference_ties=True)
    assert f["rowpos"] == rp and f["colpos"] == cp
    got_lower, got_upper = reference_layout(f)
    assert got_lower == [[(i, float(v)) for i, v in c] for c in lower]
    assert got_upper == [[(i, float(v)) for i, v in c] for c in upper]
    assert f["diag"] == [float(v) for v in diag]


def reconstruct(f, m):
    """Dense P' L U Q' from the factors."""
    L = np.eye(m)
    U = np.diag(f["diag"])
    for i, row in enumerate(f["lower_rows"]):
        for j, v in row:
            assert j < i
            L[i, j] = v
    for i, row in enumerate(f["upper_rows"]):
        for j, v in row:
            assert j > i
            U[i, j] = v
    LU = L @ U
    B = np.zeros((m, m))
    for i in range(m):
        for j in range(m):
            B[i, j] = LU[f["rowpos"][i], f["colpos"][j]]
    return B


def random_sparse(rng, m, density):
    """A non-singular sparse integer matrix: a permuted triangular backbone plus random entries."""
    while True:
        A = np.zeros((m, m))
        perm = list(range(m))
        rng.shuffle(perm)
        for i in range(m):
            A[i, perm[i]] = rng.choice([-3, -2, -1, 1, 2, 3, 4])
        for _ in range(int(density * m * m)):
            A[rng.randrange(m), rng.randrange(m)] = rng.choice([-2, -1, 1, 2, 5])
        if abs(np.linalg.det(A)) > 1e-6:
            return A


def columns_of_dense(A):
    m = A.shape[0]
    return [[(i, float(A[i, j])) for i in range(m) if A[i, j] != 0] for j in range(m)]


@pytest.mark.parametrize("seed", range(12))
def test_same_pivot_order_as_the_oracle(seed):
    rng = random.Random(seed)
    m = rng.choice([3, 5, 8, 13, 21, 34])
    A = random_sparse(rng, m, 0.08)
    f = lu_factor_host(columns_of_dense(A), pivot_threshold=0.0, reference_ties=True)
    rows = [[(j, F(int(A[i, j]))) for j in range(m) if A[i, j] != 0] for i in range(m)]
    oracle = LUDecomposition.rows(rows)
    assert f["rowpos"] == list(oracle.row_permutation.fwd)
    assert f["colpos"] == list(oracle.column_permutation.fwd)
    got_lower, got_upper = reference_layout(f)
    for got, want in zip(got_lower, oracle.lower_triangular):
        assert [i for i, _ in got] == [i for i, _ in want]
        assert np.allclose([v for _, v in got], [float(v) for _, v in want], rtol=1e-12, atol=0)
    for got, want in zip(got_upper, oracle.upper_triangular):
        assert [i for i, _ in got] == [i for i, _ in want]
        assert np.allclose([v for _, v in got], [float(v) for _, v in want], rtol=1e-12, atol=0)
    assert np.allclose(f["diag"], [float(v) for v in oracle.upper_diagonal], rtol=1e-12, atol=0)


@pytest.mark.parametrize("threshold,ties", [(0.1, False), (0.01, False), (0.0, True), (1.0, False)])
@pytest.mark.parametrize("seed", range(6))
def test_factors_multiply_back(seed, threshold, ties):
    rng = random.Random(100 + seed)
    m = rng.choice([4, 17, 60, 150])
    A = random_sparse(rng, m, 0.03)
    f = lu_factor_host(columns_of_dense(A), pivot_threshold=threshold, reference_ties=ties)
    assert sorted(f["rowpos"]) == list(range(m)) and sorted(f["colpos"]) == list(range(m))
    assert np.allclose(reconstruct(f, m), A, rtol=0, atol=1e-9 * np.abs(A).max())


def test_singular_is_reported():
    from relp_amd.api import RelpError
    with pytest.raises(RelpError):
        lu_factor_host([[(0, 1.0), (1, 2.0)], [(0, 2.0), (1, 4.0)]])
    with pytest.raises(RelpError):
        lu_factor_host([[(0, 1.0)], []])


def test_netlib_basis_factor_is_small_and_shallow():
    """The optimal basis of SCFXM1 (330 rows): a few hundred non-zeros in each factor and a dependency depth far below m --
    the numbers the device design (one workgroup, factor in L2/LDS, one LDS round trip per dependency) rests on."""
    from relp_amd import api
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "SCFXM1.json")))
    model = api.Model(os.path.join(ROOT, g["file"]))
    columns = []
    for j in g["basis"]:
        r, v = model.column(j)
        columns.append(list(zip(r.tolist(), v.tolist())))
    assert len(columns) == model.nr_rows
    f = lu_factor_host(columns)
    A = np.zeros((model.nr_rows, model.nr_rows))
    for j, column in enumerate(columns):
        for i, v in column:
            A[i, j] = v
    assert np.allclose(reconstruct(f, model.nr_rows), A, atol=1e-9 * np.abs(A).max())
    assert f["nnz_lower"] + f["nnz_upper"] < 4 * sum(len(c) for c in columns)
    assert f["depth_lower"] + f["depth_upper"] < model.nr_rows // 4


def dense_of(rows, m, unit_diagonal=False, diag=None):
    a = np.eye(m) if unit_diagonal else np.zeros((m, m))
    for i, row in enumerate(rows):
        for j, v in row:
            a[i, j] = v
    if diag is not None:
        a[np.arange(m), np.arange(m)] = diag
    return a


@pytest.mark.parametrize("seed", range(12))
def test_inverted_triangles_are_the_inverses_of_the_factors(seed):
    """What the inverse-factor carry uploads at a refactorisation (`relp_lu_invert_host` = `lu_invert_factors`, lu_host.hpp): the
    strict part of L^-1 and U^-1 with its diagonal, as sparse rows of the position space.  On random sparse non-singular matrices --
    near-triangular ones like simplex bases, and denser ones -- the products with the factors are the identity, the permutations are
    those of the factorisation, and nothing above / below the diagonal appears in the wrong triangle."""
    rng = random.Random(4100 + seed)
    m = rng.choice([1, 2, 7, 40, 150, 400])
    density = rng.choice([1.5, 3.0, 6.0]) / max(m, 1)
    columns = []
    for j in range(m):
        column = {j: rng.choice([-3.0, -1.0, 0.5, 1.0, 2.0, 7.0])}  # a non-zero diagonal keeps it non-singular with high probability
        for i in range(m):
            if i != j and rng.random() < density:
                column[i] = rng.choice([-2.0, -1.0, 1.0, 3.0, 0.25])
        columns.append(sorted(column.items()))
    dense = np.zeros((m, m))
    for j, column in enumerate(columns):
        for i, v in column:
            dense[i, j] = v
    if abs(np.linalg.det(dense)) < 1e-9:
        pytest.skip("singular sample")
    f = lu_factor_host(columns)
    g = lu_factor_host(columns, inverted=True)
    assert g["rowpos"] == f["rowpos"] and g["colpos"] == f["colpos"]
    lower = dense_of(f["lower_rows"], m, unit_diagonal=True)
    upper = dense_of(f["upper_rows"], m, diag=f["diag"])
    lower_inverse = dense_of(g["lower_rows"], m, unit_diagonal=True)
    upper_inverse = dense_of(g["upper_rows"], m)
    assert all(j < i for i, row in enumerate(g["lower_rows"]) for j, _ in row)    # strictly lower
    assert all(j >= i for i, row in enumerate(g["upper_rows"]) for j, _ in row)   # upper, diagonal included
    assert all(any(j == i for j, _ in row) for i, row in enumerate(g["upper_rows"]))
    assert g["diag"] == [1.0] * m
    scale = max(1.0, np.abs(lower_inverse).max(), np.abs(upper_inverse).max())
    assert np.allclose(lower @ lower_inverse, np.eye(m), atol=1e-10 * scale)
    assert np.allclose(upper @ upper_inverse, np.eye(m), atol=1e-10 * scale)
    # ... and together they are the inverse of the permuted matrix:  P B Q = L U  =>  (P B Q)^-1 = U^-1 L^-1
    permuted = np.zeros((m, m))
    for i in range(m):
        for j in range(m):
            permuted[f["rowpos"][i], f["colpos"][j]] = dense[i, j]
    assert np.allclose(upper_inverse @ lower_inverse @ permuted, np.eye(m), atol=1e-8 * scale * scale)
