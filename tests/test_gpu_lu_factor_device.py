"""`BasisInverse::invert` as a KERNEL (relp_amd/csrc/lu_factor.hip through `relp_lu_factor_device`; -m gpu):

* with ``reference_ties`` the device kernel must produce the reference's exact-factor known answers
  (lower_upper/decomposition/mod.rs:319-438) and, on the reference's inverse-property matrices and random ones, the SAME pivot
  order and factors as the exact oracle and as the host code (one pivot per round, the reference's tie rule);
* in the product's mode (independent pivots per round, threshold 0.1, dense tail) ``P B Q = L U`` on random sparse matrices and on
  bases taken from real solves, rows sorted, triangles clean, fill close to the sequential Markowitz rule's, deterministic;
* singular matrices are reported.
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

from relp_amd.basis_inverse import lu_factor_device, lu_factor_host  # noqa: E402
from relp_oracle import LUDecomposition  # noqa: E402
from test_lu_host import EXACT_FACTOR_CASES, columns_of_dense, columns_of_rows, random_sparse, reconstruct, reference_layout  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", EXACT_FACTOR_CASES, ids=[c[0] for c in EXACT_FACTOR_CASES])
def test_reference_exact_factors_on_the_device(case):
    _, rows, rp, cp, lower, upper, diag = case
    m = len(rows)
    f = lu_factor_device(columns_of_rows(rows, m), pivot_threshold=0.0, reference_ties=True)
    assert f["rowpos"] == rp and f["colpos"] == cp
    got_lower, got_upper = reference_layout(f)
    assert got_lower == [[(i, float(v)) for i, v in c] for c in lower]
    assert got_upper == [[(i, float(v)) for i, v in c] for c in upper]
    assert f["diag"] == [float(v) for v in diag]
    assert f["info"][3] == m and f["info"][4] == 0  # one pivot per round, no dense tail


@pytest.mark.parametrize("seed", range(12))
def test_device_follows_the_oracle_pivot_for_pivot(seed):
    rng = random.Random(seed)
    m = rng.choice([3, 5, 8, 13, 21, 34])
    A = random_sparse(rng, m, 0.08)
    f = lu_factor_device(columns_of_dense(A), pivot_threshold=0.0, reference_ties=True)
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
    host = lu_factor_host(columns_of_dense(A), pivot_threshold=0.0, reference_ties=True)
    for key in ("rowpos", "colpos", "lower_rows", "upper_rows", "diag"):
        assert f[key] == host[key], key  # the same operations in the same order: bit for bit


def check_factors(f, A, columns):
    m = A.shape[0]
    assert f["info"][0] == 0
    assert sorted(f["rowpos"]) == list(range(m)) and sorted(f["colpos"]) == list(range(m))
    for i, row in enumerate(f["lower_rows"]):
        assert [j for j, _ in row] == sorted(j for j, _ in row) and all(j < i for j, _ in row) and all(v != 0 for _, v in row)
    for i, row in enumerate(f["upper_rows"]):
        assert [j for j, _ in row] == sorted(j for j, _ in row) and all(j > i for j, _ in row) and all(v != 0 for _, v in row)
    assert all(d != 0 for d in f["diag"])
    assert np.allclose(reconstruct(f, m), A, rtol=0, atol=1e-9 * np.abs(A).max())
    assert f["info"][6] == sum(len(c) for c in columns)


@pytest.mark.parametrize("dense_tail", [0, 16, 32])
@pytest.mark.parametrize("threshold", [0.1, 0.01, 1.0])
@pytest.mark.parametrize("seed", range(6))
def test_device_factors_multiply_back(seed, threshold, dense_tail):
    rng = random.Random(100 + seed)
    m = rng.choice([1, 2, 4, 17, 60, 150, 400])
    A = random_sparse(rng, m, 0.03 if m < 200 else 0.008)
    columns = columns_of_dense(A)
    f = lu_factor_device(columns, pivot_threshold=threshold, dense_tail=dense_tail)
    check_factors(f, A, columns)
    again = lu_factor_device(columns, pivot_threshold=threshold, dense_tail=dense_tail)
    for key in ("rowpos", "colpos", "lower_rows", "upper_rows", "diag"):
        assert f[key] == again[key], key  # deterministic: no result depends on the order in which atomics land


def test_device_reports_singular_matrices():
    from relp_amd.api import RelpError
    with pytest.raises(RelpError):
        lu_factor_device([[(0, 1.0), (1, 2.0)], [(0, 2.0), (1, 4.0)]])
    with pytest.raises(RelpError):
        lu_factor_device([[(0, 1.0)], []])
    with pytest.raises(RelpError):
        lu_factor_device([[(0, 1.0), (1, 2.0)], [(0, 2.0), (1, 4.0)]], reference_ties=True, pivot_threshold=0.0)


def basis_columns(name, fraction):
    """The basis a solve of `name` holds after `fraction` of its pivots (explicit carry), columns in slot order."""
    import relp_amd
    model = relp_amd.Model(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    solver = relp_amd.Solver(certify=0).load_model(model)
    total = solver.solve_relaxation()
    count = total.pivots_phase_one + total.pivots_phase_two
    solver.close()
    solver = relp_amd.Solver(certify=0, max_pivots=max(1, int(count * fraction))).load_model(model)
    solver.solve_relaxation()
    basis = solver.basis()
    solver.close()
    pivots = model.pivot_element_indices()
    art_rows = sorted(set(range(model.nr_rows)) - {r for r, _ in pivots})
    columns = []
    for c in basis:
        if c >= 0:
            r, v = model.column(int(c))
            columns.append(list(zip(r.tolist(), v.tolist())))
        else:
            columns.append([(art_rows[-1 - int(c)], 1.0)])
    return columns


@pytest.mark.parametrize("name,fraction", [("AFIRO", 1.0), ("SC205", 0.5), ("SCFXM1", 1.0), ("25FV47", 0.5), ("25FV47", 1.0), ("BNL2", 0.75),
                                           ("GREENBEA", 0.5)])
def test_device_factorises_real_bases(name, fraction):
    """Bases of real solves: P B Q = L U, fill within 1.35 x of the sequential Markowitz rule's (host code), a few dozen rounds
    where the sequential rule makes m steps."""
    columns = basis_columns(name, fraction)
    m = len(columns)
    A = np.zeros((m, m))
    for j, column in enumerate(columns):
        for i, v in column:
            A[i, j] = v
    f = lu_factor_device(columns)
    check_factors(f, A, columns)
    host = lu_factor_host(columns)
    fill_device, fill_host = f["nnz_lower"] + f["nnz_upper"], host["nnz_lower"] + host["nnz_upper"]
    print("%s at %.0f %%: m %d nnz(B) %d | device: %d rounds + %d dense rows, nnz(L) + nnz(U) %d, %.1f us | host %d" % (
        name, 100 * fraction, m, f["info"][6], f["info"][3], f["info"][4], fill_device, f["info"][31] / 10.0, fill_host) + " (%d rounds in LDS)" % f["info"][10])
    print("   kcycles: load %d | candidates %d | competition %d | conflicts %d | accept %d | U rows + targets %d | layout %d | copy + eliminate %d | "
          "reset %d | dense tail %d | finalisation %d" % tuple(16 * v // 1000 for v in f["info"][12:23]))
    print("   eliminating waves, kcycles summed: preamble %d | pivot set-up %d | entry loop %d | write-out %d; %d (target, pivot) pairs, %d targets" % tuple(f["info"][23:29]))
    assert fill_device <= 1.35 * fill_host + 64
    assert f["info"][3] <= max(8, m // 8)


def dense_triangles(f, m):
    L, U = np.eye(m), np.diag(np.array(f["diag"], dtype=float))
    for i, row in enumerate(f["lower_rows"]):
        for j, v in row:
            L[i, j] = v
    for i, row in enumerate(f["upper_rows"]):
        for j, v in row:
            U[i, j] = v
    return L, U


@pytest.mark.parametrize("seed", range(10))
def test_device_inverts_both_triangles(seed):
    """`lu_invert_kernel` (lu_device_tasks.hip): the strict part of L^-1 and U^-1 with its diagonal, rows sorted by column, from the
    factors the factorisation kernel left on the device -- products with those factors are the identity; small matrices take the
    LDS accumulators, m = 1200 the global ones."""
    rng = random.Random(5200 + seed)
    m = rng.choice([1, 2, 7, 40, 150, 400, 1200])
    A = random_sparse(rng, m, (3.0 if m < 500 else 1.5) / m)
    columns = columns_of_dense(A)
    f = lu_factor_device(columns)
    inv = lu_factor_device(columns, inverted=True)
    assert inv["rowpos"] == f["rowpos"] and inv["colpos"] == f["colpos"] and inv["diag"] == [1.0] * m
    L, U = dense_triangles(f, m)
    Li, Ui = np.eye(m), np.zeros((m, m))
    for i, row in enumerate(inv["lower_rows"]):
        assert [j for j, _ in row] == sorted(j for j, _ in row) and all(j < i for j, _ in row)
        for j, v in row:
            Li[i, j] = v
    for i, row in enumerate(inv["upper_rows"]):
        assert [j for j, _ in row] == sorted(j for j, _ in row) and all(j >= i for j, _ in row) and row[0][0] == i
        for j, v in row:
            Ui[i, j] = v
    scale = max(1.0, np.abs(Li).max(), np.abs(Ui).max())
    assert np.allclose(Li @ L, np.eye(m), atol=1e-9 * scale) and np.allclose(Ui @ U, np.eye(m), atol=1e-9 * scale)
    assert inv["info"][7] == sum(len(r) for r in inv["lower_rows"]) and inv["info"][8] == sum(len(r) for r in inv["upper_rows"])
    again = lu_factor_device(columns, inverted=True)
    assert again["lower_rows"] == inv["lower_rows"] and again["upper_rows"] == inv["upper_rows"]


@pytest.mark.parametrize("name,fraction", [("25FV47", 1.0), ("GREENBEA", 0.5)])
def test_device_inverts_the_triangles_of_real_bases(name, fraction):
    columns = basis_columns(name, fraction)
    m = len(columns)
    f = lu_factor_device(columns)
    inv = lu_factor_device(columns, inverted=True)
    L, U = dense_triangles(f, m)
    Li, Ui = np.eye(m), np.zeros((m, m))
    for i, row in enumerate(inv["lower_rows"]):
        for j, v in row:
            Li[i, j] = v
    for i, row in enumerate(inv["upper_rows"]):
        for j, v in row:
            Ui[i, j] = v
    assert np.abs(Li @ L - np.eye(m)).max() < 1e-7 and np.abs(Ui @ U - np.eye(m)).max() < 1e-7
    print("%s at %.0f %%: nnz(L) + nnz(U) + m %d -> nnz(L^-1) + nnz(U^-1) %d, factorisation + inversion %.1f us" % (
        name, 100 * fraction, f["nnz_lower"] + f["nnz_upper"] + m, inv["info"][7] + inv["info"][8], inv["info"][31] / 10.0))
    print("   inversion, kcycles: L^-1 block %d wall (waves summed: waiting %d | streaming + accumulating %d | emitting %d); U^-1 block %d wall (%d | %d | %d)" % (
        tuple(inv["info"][23:27]) + tuple(inv["info"][19:23])))


@pytest.mark.parametrize("arena", [1024, 1536, 2048])
def test_the_lds_arena_spills_back_to_global_memory(monkeypatch, arena):
    """The active sub-matrix moves into LDS once it fits (3/4 of the arena) and back to global memory when a round's fill-in bound
    outgrows the arena: with a small arena (RELP_LUF_LDS_ARENA) both moves happen, repeatedly; the factors are the same bits as with
    the sub-matrix in global memory throughout (RELP_LUF_NO_LDS_ARENA)."""
    columns = basis_columns("25FV47", 1.0)
    m = len(columns)
    A = np.zeros((m, m))
    for j, column in enumerate(columns):
        for i, v in column:
            A[i, j] = v
    monkeypatch.setenv("RELP_LUF_NO_LDS_ARENA", "1")
    plain = lu_factor_device(columns)
    assert plain["info"][10] == 0
    monkeypatch.delenv("RELP_LUF_NO_LDS_ARENA")
    monkeypatch.setenv("RELP_LUF_LDS_ARENA", str(arena))
    f = lu_factor_device(columns)
    check_factors(f, A, columns)
    assert 0 < f["info"][10] <= f["info"][3]
    for key in ("rowpos", "colpos", "lower_rows", "upper_rows", "diag"):
        assert f[key] == plain[key], key
