"""The stand-alone device ``BasisInverse`` (``relp_bi_*``; LU + Forrest-Tomlin, relp_amd/csrc/lu.hip) against the reference.

* the reference's exact-factor known-answer tests (lower_upper/decomposition/mod.rs:319-438) on the RESIDENT factors;
* its Forrest-Tomlin known-answer tests (lower_upper/mod.rs:688-940): eta file, rotated U, diagonal, every column and row of
  the updated inverse -- all values are small rationals, compared at 1e-15 relative;
* FTRAN / BTRAN known answers (lower_upper/mod.rs:536-685);
* random bases with random column replacements against a dense numpy inverse, through refactorisations;
* the ORACLE'S OWN ``Carry`` solving AFIRO / SC50A / ADLITTLE with the device object mirrored behind every ``BasisInverse``
  call of the exact ``BasisInverseRows``: each FTRAN, BTRAN, row, element and refactor decision is compared step for step
  along the reference's pivot sequence.
Everything goes through the C ABI; nothing is computed on the CPU but the expectations.
"""
import os
import random
import sys
from fractions import Fraction as F

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from relp_amd.api import ERR_STATE, RelpError  # noqa: E402
from relp_amd.basis_inverse import BasisInverse  # noqa: E402

pytestmark = pytest.mark.gpu

REF = dict(pivot_threshold=0.0, reference_ties=1)


def columns_of_rows(rows, m):
    columns = [[] for _ in range(m)]
    for i, row in enumerate(rows):
        for j, v in row:
            columns[j].append((i, float(v)))
    return columns


def fl(pairs):
    return [(i, float(v)) for i, v in pairs]


def close_pairs(got, want, tol=1e-15):
    assert [i for i, _ in got] == [i for i, _ in want], (got, want)
    for (_, a), (_, b) in zip(got, want):
        assert abs(a - float(b)) <= tol * max(1.0, abs(float(b))), (got, want)


def dense_of(pairs, m):
    out = np.zeros(m)
    for i, v in pairs:
        out[i] = float(v)
    return out


EXACT_FACTOR_CASES = [  # decomposition/mod.rs:319-438
    ("identity_2", [[(0, 1)], [(1, 1)]], [0, 1], [0, 1], [[]], [[]], [1, 1]),
    ("identity_3", [[(0, 1)], [(1, 1)], [(2, 1)]], [0, 1, 2], [0, 1, 2], [[], []], [[], []], [1, 1, 1]),
    ("offdiagonal_2_upper", [[(0, 1), (1, 1)], [(1, 1)]], [0, 1], [0, 1], [[]], [[(0, 1)]], [1, 1]),
    ("offdiagonal_2_lower", [[(0, 1)], [(0, 1), (1, 1)]], [0, 1], [0, 1], [[(1, 1)]], [[]], [1, 1]),
    ("offdiagonal_2_both", [[(0, 1), (1, 1)], [(0, 1)]], [1, 0], [0, 1], [[(1, 1)]], [[]], [1, 1]),
    ("wikipedia_example", [[(0, 4), (1, 3)], [(0, 6), (1, 3)]], [0, 1], [0, 1], [[(1, 1.5)]], [[(0, 3)]], [4, -1.5]),
    ("wikipedia_example2", [[(0, -1), (1, 1.5)], [(0, 1), (1, -1)]], [0, 1], [0, 1], [[(1, -1)]], [[(0, 1.5)]], [-1, 0.5]),
]


@pytest.mark.parametrize("case", EXACT_FACTOR_CASES, ids=[c[0] for c in EXACT_FACTOR_CASES])
def test_reference_exact_factors_on_device(case):
    _, rows, rp, cp, lower, upper, diag = case
    m = len(rows)
    bi = BasisInverse.invert(columns_of_rows(rows, m), **REF)
    f = bi.factors()
    assert f["row_permutation"] == rp and f["column_permutation"] == cp
    assert f["lower_triangular"] == [fl(c) for c in lower]
    assert f["upper_triangular"] == [fl(c) for c in upper]
    assert f["upper_diagonal"] == [float(v) for v in diag]
    assert f["updates"] == []
    # B^-1 B = I through the device solves
    B = np.zeros((m, m))
    for i, row in enumerate(rows):
        for j, v in row:
            B[i, j] = v
    for j in range(m):
        assert np.allclose(bi.left_multiply_by_basis_inverse([(i, B[i, j]) for i in range(m) if B[i, j]]), np.eye(m)[j], atol=1e-15)
    for i in range(m):
        assert np.allclose(bi.basis_inverse_row(i) @ B, np.eye(m)[i], atol=1e-15)


def test_wikipedia_example2_columns():  # decomposition/mod.rs:427-437
    bi = BasisInverse.invert(columns_of_rows([[(0, -1), (1, 1.5)], [(0, 1), (1, -1)]], 2), **REF)
    assert np.array_equal(bi.left_multiply_by_basis_inverse([(0, 1)]), [2, 2])
    assert np.array_equal(bi.left_multiply_by_basis_inverse([(1, 1)]), [3, 2])


def test_matmul_known_answers():  # lower_upper/mod.rs:536-685
    ident = BasisInverse.identity(2)
    for column in ([], [(0, 1)], [(1, 1)], [(0, 1), (1, 1)]):
        assert np.array_equal(ident.left_multiply_by_basis_inverse(column), dense_of(column, 2))
        assert np.array_equal(ident.right_multiply_by_basis_inverse(column), dense_of(column, 2))
    off = BasisInverse.invert(columns_of_rows([[(0, 1)], [(0, 1), (1, 1)]], 2), **REF)  # L = [[1,0],[1,1]]
    assert np.array_equal(off.left_multiply_by_basis_inverse([]), [0, 0])
    assert np.array_equal(off.left_multiply_by_basis_inverse([(0, 1)]), [1, -1])
    assert np.array_equal(off.left_multiply_by_basis_inverse([(1, 1)]), [0, 1])
    dense = BasisInverse.invert(columns_of_rows([[(0, 1), (1, 2)], [(0, 3), (1, 4)]], 2), **REF)  # the `dense` fixture's B
    assert np.allclose(dense.left_multiply_by_basis_inverse([(0, 1)]), [-2, 1.5], rtol=1e-15)
    assert np.allclose(dense.left_multiply_by_basis_inverse([(1, 1)]), [1, -0.5], rtol=1e-15)
    assert np.allclose(dense.right_multiply_by_basis_inverse([(0, 1)]), [-2, 1], rtol=1e-15)
    assert np.allclose(dense.right_multiply_by_basis_inverse([(1, 1)]), [1.5, -0.5], rtol=1e-15)
    assert dense.generate_element(1, [(0, 1)]) == pytest.approx(1.5, rel=1e-15)
    assert ident.generate_element(1, [(0, 1)]) is None


def upper_triangular_matrix(upper, diag):
    m = len(diag)
    rows = [[(i, diag[i])] for i in range(m)]
    for c, column in enumerate(upper):
        for i, v in column:
            rows[i].append((c + 1, v))
    return columns_of_rows(rows, m)


def check_ft(bi, upper, diag, eta_pivot, eta, cancellation=1.0):
    """`cancellation`: the new diagonal element is spike_t - r.spike (eta_file.rs:112-134); when that difference cancels, f64
    loses that factor of relative accuracy (5x5 example: 22 - 22.592... = -0.592..., a factor 37)."""
    f = bi.factors()
    m = len(diag)
    assert f["row_permutation"] == list(range(m)) and f["column_permutation"] == list(range(m))
    assert f["lower_triangular"] == [[] for _ in range(m - 1)]
    for got, want in zip(f["upper_triangular"], upper):
        close_pairs(got, want)
    assert np.allclose(f["upper_diagonal"], [float(v) for v in diag], rtol=1e-15 * cancellation, atol=0)
    assert len(f["updates"]) == 1
    assert f["updates"][0][0] == eta_pivot
    close_pairs(f["updates"][0][1], eta)


def test_ft_no_change():  # lower_upper/mod.rs:695-706
    bi = BasisInverse.identity(3)
    bi.left_multiply_by_basis_inverse([(1, 1)])
    bi.change_basis(1)
    check_ft(bi, [[], []], [1, 1, 1], 1, [])


def test_ft_from_identity_2():  # lower_upper/mod.rs:708-727
    bi = BasisInverse.identity(2)
    bi.left_multiply_by_basis_inverse([(0, 1), (1, 1)])
    bi.change_basis(0)
    check_ft(bi, [[(0, 1)]], [1, 1], 0, [])


def test_ft_from_5x5_identity_no_r():  # lower_upper/mod.rs:729-760
    bi = BasisInverse.identity(5)
    bi.left_multiply_by_basis_inverse([(0, 2), (1, 3), (2, 5), (3, 7)])
    bi.change_basis(1)
    check_ft(bi, [[], [], [], [(0, 2), (1, 5), (2, 7)]], [1, 1, 1, 1, 3], 1, [])


def test_ft_from_4x4_identity():  # lower_upper/mod.rs:762-839
    m = 4
    bi = BasisInverse.invert(upper_triangular_matrix([[], [], [(1, 5)]], [1, 1, 4, 6]), **REF)
    assert bi.factors()["upper_triangular"] == [[], [], [(1, 5.0)]]
    bi.left_multiply_by_basis_inverse([(1, 2), (2, 3), (3, 4)])  # L = I, no etas: the spike is the column itself
    bi.change_basis(1)
    check_ft(bi, [[], [], [(1, 3), (2, 4)]], [1, 4, 6, F(-8, 6)], 1, [(3, F(5, 6))])
    cols = [[(0, 1)],
            [(1, F(-3, 4)), (2, F(9, 16)), (3, F(1, 2))],
            [(2, F(1, 4))],
            [(1, F(5, 8)), (2, F(-15, 32)), (3, F(-1, 4))]]
    for j in range(m):
        assert np.allclose(bi.left_multiply_by_basis_inverse([(j, 1)]), dense_of(cols[j], m), rtol=1e-15, atol=1e-17)
    rows = [[(0, 1)],
            [(1, F(-3, 4)), (3, F(5, 8))],
            [(1, F(9, 16)), (2, F(1, 4)), (3, F(-15, 32))],
            [(1, F(1, 2)), (3, F(-1, 4))]]
    for i in range(m):
        assert np.allclose(bi.basis_inverse_row(i), dense_of(rows[i], m), rtol=1e-15, atol=1e-17)


def test_ft_from_5x5_identity_elble_sahinidis():  # lower_upper/mod.rs:841-939
    m = 5
    upper = [[(0, 12)], [(0, 13), (1, 23)], [(0, 14), (1, 24), (2, 34)], [(0, 15), (1, 25), (2, 35), (3, 45)]]
    bi = BasisInverse.invert(upper_triangular_matrix(upper, [11, 22, 33, 44, 55]), **REF)
    bi.left_multiply_by_basis_inverse([(0, 12), (1, 22), (2, 32), (3, 42)])
    bi.change_basis(1)
    eta = [(2, F(23, 33)), (3, F(24 * 33 - 34 * 23, 33 * 44)), (4, F(43, 7986))]
    check_ft(bi, [[(0, 13)], [(0, 14), (1, 34)], [(0, 15), (1, 35), (2, 45)], [(0, 12), (1, 32), (2, 42)]],
             [11, 33, 44, 55, F(-215, 363)], 1, eta, cancellation=22 * 363 / 215)
    cols = [[(0, F(1, 11))],
            [(0, F(-2, 11)), (1, F(-363, 215)), (2, F(-1, 43)), (3, F(693, 430))],
            [(0, F(1, 11)), (1, F(253, 215)), (2, F(2, 43)), (3, F(-483, 430))],
            [(1, F(1, 86)), (2, F(-1, 43)), (3, F(1, 86))],
            [(1, F(1, 110)), (3, F(-3, 110)), (4, F(1, 55))]]
    for j in range(m):
        assert np.allclose(bi.left_multiply_by_basis_inverse([(j, 1)]), dense_of(cols[j], m), rtol=1e-15 * 22 * 363 / 215, atol=1e-15)
    assert np.allclose(bi.left_multiply_by_basis_inverse([(0, 1), (1, 1)]),
                       dense_of([(0, F(-1, 11)), (1, F(-363, 215)), (2, F(-1, 43)), (3, F(693, 430))], m), rtol=1e-15 * 22 * 363 / 215, atol=1e-15)
    rows = [[(0, F(1, 11)), (1, F(-2, 11)), (2, F(1, 11))],
            [(1, F(-363, 215)), (2, F(253, 215)), (3, F(1, 86)), (4, F(1, 110))],
            [(1, F(-1, 43)), (2, F(2, 43)), (3, F(-1, 43))],
            [(1, F(693, 430)), (2, F(-483, 430)), (3, F(1, 86)), (4, F(-3, 110))],
            [(4, F(1, 55))]]
    for i in range(m):
        assert np.allclose(bi.basis_inverse_row(i), dense_of(rows[i], m), rtol=1e-15 * 22 * 363 / 215, atol=1e-15)


def test_change_basis_needs_the_column():
    bi = BasisInverse.identity(3)
    with pytest.raises(RelpError) as e:
        bi.change_basis(0)
    assert e.value.status == ERR_STATE


# ---- random bases, random replacements, dense numpy inverse as the expectation --------------------------------------
def random_basis(rng, m, extra):
    while True:
        A = np.zeros((m, m))
        perm = list(range(m))
        rng.shuffle(perm)
        for i in range(m):
            A[i, perm[i]] = rng.choice([-3, -2, -1, 1, 2, 3, 4])
        for _ in range(extra):
            A[rng.randrange(m), rng.randrange(m)] = rng.choice([-2, -1, 1, 2, 5])
        if np.linalg.cond(A) < 1e8:
            return A


def sparse_column(A, j):
    return [(i, float(A[i, j])) for i in range(A.shape[0]) if A[i, j] != 0]


@pytest.mark.parametrize("m,extra,period", [(7, 10, 5), (40, 60, 31), (200, 500, 31), (821, 3500, 31), (1500, 5000, 16)])
def test_random_updates_against_numpy(m, extra, period):
    rng = random.Random(m * 1000 + extra)
    B = random_basis(rng, m, extra)
    bi = BasisInverse.invert([sparse_column(B, j) for j in range(m)], refactor_period=period)
    refactors = 0
    steps = 3 * period + 7 if m <= 200 else period + 9

    def check(k):
        Binv = np.linalg.inv(B)
        scale = np.abs(Binv).max()
        for _ in range(3):
            c = [(i, rng.choice([-2.0, -1.0, 1.0, 3.0])) for i in sorted(rng.sample(range(m), min(m, rng.randint(1, 6))))]
            assert np.allclose(bi.left_multiply_by_basis_inverse(c), Binv @ dense_of(c, m), rtol=1e-9, atol=1e-9 * scale), k
            assert np.allclose(bi.right_multiply_by_basis_inverse(c), dense_of(c, m) @ Binv, rtol=1e-9, atol=1e-9 * scale), k
        r = rng.randrange(m)
        assert np.allclose(bi.basis_inverse_row(r), Binv[r], rtol=1e-9, atol=1e-9 * scale), k

    check(-1)
    for k in range(steps):
        # a new sparse column and a pivot row with a healthy pivot element
        while True:
            c = [(i, rng.choice([-2.0, -1.0, 1.0, 2.0, 3.0])) for i in sorted(rng.sample(range(m), min(m, rng.randint(1, 5))))]
            alpha = np.linalg.solve(B, dense_of(c, m))
            candidates = [i for i in range(m) if abs(alpha[i]) > 0.2 * np.abs(alpha).max()]
            p = rng.choice(candidates)
            trial = B.copy()
            trial[:, p] = dense_of(c, m)
            if np.linalg.cond(trial) < 1e8:
                break
        got = bi.left_multiply_by_basis_inverse(c)
        assert np.allclose(got, alpha, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(alpha).max()))
        if bi.should_refactor():  # carry/mod.rs:584-591: polled before the update; the new basis is inverted from scratch
            B = trial
            bi = BasisInverse.invert([sparse_column(B, j) for j in range(m)], refactor_period=period)
            refactors += 1
        else:
            bi.change_basis(p)
            B = trial
        assert bi.statistics()["updates"] == (k + 1) % (period + 1)
        if k % 5 == 4 or k == steps - 1:
            check(k)
    assert refactors == steps // (period + 1)


def test_remove_basis_part():  # carry/mod.rs:176-180; basis_inverse_rows.rs:212-229
    rng = random.Random(5)
    m = 12
    drop = [2, 7]
    keep = [i for i in range(m) if i not in drop]
    small = random_basis(rng, m - len(drop), 20)
    B = np.zeros((m, m))
    B[np.ix_(keep, keep)] = small
    for i in drop:  # what leaves: the unit column of an artificial variable that stayed basic on a redundant row
        B[i, i] = 1.0
    bi = BasisInverse.invert([sparse_column(B, j) for j in range(m)])
    bi.remove_basis_part(drop)
    assert bi.m() == m - len(drop)
    inv = np.linalg.inv(small)
    for r in range(m - len(drop)):
        assert np.allclose(bi.basis_inverse_row(r), inv[r], rtol=1e-10, atol=1e-12)


# ---- the oracle's own Carry, every BasisInverse call mirrored on the device -------------------------------------------
def make_mirrored(log):
    from relp_oracle import BasisInverseRows

    class Info:
        def __init__(self, exact, original):
            self.exact = exact
            self.original = original
            self.column = exact.column
            self.spike = None

        def into_column(self):
            return self.exact.into_column()

    class Mirrored:
        """`BasisInverse` (carry/mod.rs:69-169): answers come from the exact `BasisInverseRows`, every call is repeated on the
        device `LUDecomposition` and compared."""
        tol = 1e-9

        def __init__(self, exact, device):
            self.exact = exact
            self.device = device
            self.last = None

        @classmethod
        def identity(cls, m):
            log["identity"] += 1
            return cls(BasisInverseRows.identity(m), BasisInverse.identity(m))

        @classmethod
        def invert(cls, columns):
            columns = [list(c) for c in columns]
            log["invert"] += 1
            return cls(BasisInverseRows.invert(columns), BasisInverse.invert([[(i, float(v)) for i, v in c] for c in columns]))

        def m(self):
            assert self.device.m() == self.exact.m()
            return self.exact.m()

        def _compare(self, got, want, what):
            expect = dense_of(want, self.exact.m())
            scale = max(1.0, np.abs(expect).max())
            assert np.allclose(got, expect, rtol=self.tol, atol=self.tol * scale), what

        def left_multiply_by_basis_inverse(self, column):
            column = list(column)
            exact = self.exact.left_multiply_by_basis_inverse(column)
            got = self.device.left_multiply_by_basis_inverse([(i, float(v)) for i, v in column])
            self._compare(got, exact.into_column(), "FTRAN")
            self.last = column
            log["ftran"] += 1
            return Info(exact, column)

        def right_multiply_by_basis_inverse(self, row):
            row = list(row)
            exact = self.exact.right_multiply_by_basis_inverse(row)
            self._compare(self.device.right_multiply_by_basis_inverse([(i, float(v)) for i, v in row]), exact, "BTRAN")
            log["btran"] += 1
            return exact

        def generate_element(self, i, column):
            column = list(column)
            exact = self.exact.generate_element(i, column)
            got = self.device.generate_element(i, [(r, float(v)) for r, v in column])
            assert abs((got or 0.0) - float(exact or 0)) <= self.tol * max(1.0, abs(float(exact or 0))), "generate_element"
            log["element"] += 1
            return exact

        def should_refactor(self):
            log["should_refactor"] += 1
            return self.device.should_refactor()

        def change_basis(self, pivot_row_index, info):
            if self.last is not info.original:  # the device keeps the spike of its LAST left_multiply
                self.device.left_multiply_by_basis_inverse([(i, float(v)) for i, v in info.original])
            self.device.change_basis(pivot_row_index)
            log["change_basis"] += 1
            return self.exact.change_basis(pivot_row_index, info.exact)

        def basis_inverse_row(self, row):
            exact = self.exact.basis_inverse_row(row)
            self._compare(self.device.basis_inverse_row(row), exact, "row")
            log["row"] += 1
            return exact

        def remove_basis_part(self, indices):
            self.exact.remove_basis_part(indices)
            self.device.remove_basis_part(list(indices))

    return Mirrored


@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "ADLITTLE"])
def test_oracle_carry_drives_the_device_basis_inverse(name):
    import json
    from collections import Counter

    from relp_oracle import FiniteOptimum
    from relp_oracle import solve as S
    from relp_oracle.mps import load_problem
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    general, provider = load_problem(os.path.join(ROOT, golden["file"]))
    log = Counter()
    result = S.solve_relaxation(provider, make_mirrored(log))
    assert isinstance(result, FiniteOptimum)
    objective = general.objective_of(provider.reconstruct_solution(result.solution))
    assert "%d/%d" % (objective.numerator, objective.denominator) == golden["objective"]
    assert result.basis == golden["basis"]  # the reference's pivot path, undisturbed by the mirror
    pivots = golden["pivots_phase1"] + golden["pivots_phase2"]
    assert log["change_basis"] + log["invert"] >= pivots  # every pivot went through the device object
    assert log["btran"] >= pivots and log["row"] >= pivots and log["ftran"] >= pivots
    if pivots > 40:
        assert log["invert"] >= 1  # the refactorisation path (should_refactor -> invert) was taken
