"""``relp_bix_*``: the reference's ``BasisInverse`` (carry/mod.rs:69-169) over EXACT rationals on the device (relp_amd/csrc/exact_bi.hip; ``-m gpu``).

Round-5 review, item 6: the exact path could only be called as a whole solve.  Here the nine trait methods one at a time, through the C ABI,
against the reference's own known answers -- with ``==`` on ``Fraction``s, not ``approx``:

* FTRAN / BTRAN / generate_element known answers (lower_upper/mod.rs:536-685) and the inverse property on the reference's test matrices
  (decomposition/mod.rs:319-438, 480-651);
* the Forrest-Tomlin known-answer tests (lower_upper/mod.rs:688-940: every column and row of the updated inverse of the 4 x 4 and the
  Elble-Sahinidis 5 x 5 example) -- the update itself is an integer-preserving pivot here, the INVERSE it leaves is what the trait promises;
* random rational bases with random column replacements against an exact Fraction inverse, through widenings of the integers;
* the ORACLE'S OWN ``Carry`` solving AFIRO / SC50A / ADLITTLE with this object mirrored behind every ``BasisInverse`` call, every answer equal;
* at the size of whole LPs: ``invert`` of the reference's optimal bases (223 to 1158 rows), B^-1 B = I, x_B >= 0, c_B' x_B = the exact optimum.
"""
import os
import random
import sys
from fractions import Fraction as F

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from relp_amd.api import ERR_STATE, RelpError  # noqa: E402
from relp_amd.basis_inverse import ExactBasisInverse  # noqa: E402

pytestmark = pytest.mark.gpu


def columns_of_rows(rows, m):
    columns = [[] for _ in range(m)]
    for i, row in enumerate(rows):
        for j, v in row:
            columns[j].append((i, F(v)))
    return columns


def dense_of(pairs, m):
    out = [F(0)] * m
    for i, v in pairs:
        out[i] = F(v)
    return out


def exact_inverse(B):
    """Gauss-Jordan over Fractions (the expectation of the random tests)."""
    m = len(B)
    A = [[F(v) for v in row] + [F(int(i == j)) for j in range(m)] for i, row in enumerate(B)]
    for c in range(m):
        pivot = next(r for r in range(c, m) if A[r][c] != 0)
        A[c], A[pivot] = A[pivot], A[c]
        inv = 1 / A[c][c]
        A[c] = [v * inv for v in A[c]]
        for r in range(m):
            if r != c and A[r][c] != 0:
                f = A[r][c]
                A[r] = [a - f * b for a, b in zip(A[r], A[c])]
    return [row[m:] for row in A]


INVERSE_CASES = [  # decomposition/mod.rs:319-438 (the matrices of the exact-factor tests), by rows
    ("identity_2", [[(0, 1)], [(1, 1)]]),
    ("offdiagonal_2_upper", [[(0, 1), (1, 1)], [(1, 1)]]),
    ("offdiagonal_2_lower", [[(0, 1)], [(0, 1), (1, 1)]]),
    ("offdiagonal_2_both", [[(0, 1), (1, 1)], [(0, 1)]]),
    ("wikipedia_example", [[(0, 4), (1, 3)], [(0, 6), (1, 3)]]),
    ("wikipedia_example2", [[(0, -1), (1, F(3, 2))], [(0, 1), (1, -1)]]),
]


@pytest.mark.parametrize("case", INVERSE_CASES, ids=[c[0] for c in INVERSE_CASES])
def test_inverse_property_on_the_reference_matrices(case):
    _, rows = case
    m = len(rows)
    B = [[F(0)] * m for _ in range(m)]
    for i, row in enumerate(rows):
        for j, v in row:
            B[i][j] = F(v)
    bi = ExactBasisInverse.invert(columns_of_rows(rows, m))
    for j in range(m):  # B^-1 B = I
        assert bi.left_multiply_by_basis_inverse([(i, B[i][j]) for i in range(m) if B[i][j]]) == dense_of([(j, 1)], m)
    inverse = exact_inverse(B)
    for i in range(m):
        assert bi.basis_inverse_row(i) == inverse[i]
    assert not bi.should_refactor() and bi.m() == m


def test_wikipedia_example2_columns():  # decomposition/mod.rs:427-437
    bi = ExactBasisInverse.invert(columns_of_rows([[(0, -1), (1, F(3, 2))], [(0, 1), (1, -1)]], 2))
    assert bi.left_multiply_by_basis_inverse([(0, 1)]) == [2, 2]
    assert bi.left_multiply_by_basis_inverse([(1, 1)]) == [3, 2]


def test_matmul_known_answers():  # lower_upper/mod.rs:536-685
    ident = ExactBasisInverse.identity(2)
    for column in ([], [(0, 1)], [(1, 1)], [(0, 1), (1, 1)]):
        assert ident.left_multiply_by_basis_inverse(column) == dense_of(column, 2)
        assert ident.right_multiply_by_basis_inverse(column) == dense_of(column, 2)
    off = ExactBasisInverse.invert(columns_of_rows([[(0, 1)], [(0, 1), (1, 1)]], 2))
    assert off.left_multiply_by_basis_inverse([]) == [0, 0]
    assert off.left_multiply_by_basis_inverse([(0, 1)]) == [1, -1]
    assert off.left_multiply_by_basis_inverse([(1, 1)]) == [0, 1]
    dense = ExactBasisInverse.invert(columns_of_rows([[(0, 1), (1, 2)], [(0, 3), (1, 4)]], 2))
    assert dense.left_multiply_by_basis_inverse([(0, 1)]) == [-2, F(3, 2)]
    assert dense.left_multiply_by_basis_inverse([(1, 1)]) == [1, F(-1, 2)]
    assert dense.right_multiply_by_basis_inverse([(0, 1)]) == [-2, 1]
    assert dense.right_multiply_by_basis_inverse([(1, 1)]) == [F(3, 2), F(-1, 2)]
    assert dense.generate_element(1, [(0, 1)]) == F(3, 2)
    assert ident.generate_element(1, [(0, 1)]) is None


def upper_triangular_matrix(upper, diag):
    m = len(diag)
    rows = [[(i, diag[i])] for i in range(m)]
    for c, column in enumerate(upper):
        for i, v in column:
            rows[i].append((c + 1, v))
    return columns_of_rows(rows, m)


def test_change_basis_from_4x4():  # lower_upper/mod.rs:762-839: every column and row of the updated inverse
    m = 4
    bi = ExactBasisInverse.invert(upper_triangular_matrix([[], [], [(1, 5)]], [1, 1, 4, 6]))
    bi.left_multiply_by_basis_inverse([(1, 2), (2, 3), (3, 4)])
    bi.change_basis(1)
    cols = [[(0, 1)],
            [(1, F(-3, 4)), (2, F(9, 16)), (3, F(1, 2))],
            [(2, F(1, 4))],
            [(1, F(5, 8)), (2, F(-15, 32)), (3, F(-1, 4))]]
    for j in range(m):
        assert bi.left_multiply_by_basis_inverse([(j, 1)]) == dense_of(cols[j], m)
    rows = [[(0, 1)],
            [(1, F(-3, 4)), (3, F(5, 8))],
            [(1, F(9, 16)), (2, F(1, 4)), (3, F(-15, 32))],
            [(1, F(1, 2)), (3, F(-1, 4))]]
    for i in range(m):
        assert bi.basis_inverse_row(i) == dense_of(rows[i], m)


def test_change_basis_elble_sahinidis_5x5():  # lower_upper/mod.rs:841-939
    m = 5
    upper = [[(0, 12)], [(0, 13), (1, 23)], [(0, 14), (1, 24), (2, 34)], [(0, 15), (1, 25), (2, 35), (3, 45)]]
    bi = ExactBasisInverse.invert(upper_triangular_matrix(upper, [11, 22, 33, 44, 55]))
    bi.left_multiply_by_basis_inverse([(0, 12), (1, 22), (2, 32), (3, 42)])
    bi.change_basis(1)
    cols = [[(0, F(1, 11))],
            [(0, F(-2, 11)), (1, F(-363, 215)), (2, F(-1, 43)), (3, F(693, 430))],
            [(0, F(1, 11)), (1, F(253, 215)), (2, F(2, 43)), (3, F(-483, 430))],
            [(1, F(1, 86)), (2, F(-1, 43)), (3, F(1, 86))],
            [(1, F(1, 110)), (3, F(-3, 110)), (4, F(1, 55))]]
    for j in range(m):
        assert bi.left_multiply_by_basis_inverse([(j, 1)]) == dense_of(cols[j], m)
    assert bi.left_multiply_by_basis_inverse([(0, 1), (1, 1)]) == dense_of([(0, F(-1, 11)), (1, F(-363, 215)), (2, F(-1, 43)), (3, F(693, 430))], m)
    rows = [[(0, F(1, 11)), (1, F(-2, 11)), (2, F(1, 11))],
            [(1, F(-363, 215)), (2, F(253, 215)), (3, F(1, 86)), (4, F(1, 110))],
            [(1, F(-1, 43)), (2, F(2, 43)), (3, F(-1, 43))],
            [(1, F(693, 430)), (2, F(-483, 430)), (3, F(1, 86)), (4, F(-3, 110))],
            [(4, F(1, 55))]]
    for i in range(m):
        assert bi.basis_inverse_row(i) == dense_of(rows[i], m)


def test_small_updates_from_the_identity():  # lower_upper/mod.rs:695-760 (what the updates leave, as an inverse)
    bi = ExactBasisInverse.identity(3)
    bi.left_multiply_by_basis_inverse([(1, 1)])
    bi.change_basis(1)
    for i in range(3):
        assert bi.basis_inverse_row(i) == dense_of([(i, 1)], 3)
    bi = ExactBasisInverse.identity(5)
    assert bi.left_multiply_by_basis_inverse([(0, 2), (1, 3), (2, 5), (3, 7)]) == [2, 3, 5, 7, 0]
    bi.change_basis(1)
    # B = I with column 1 = (2, 3, 5, 7, 0): B^-1 e_1 = (-2/3, 1/3, -5/3, -7/3, 0)
    assert bi.left_multiply_by_basis_inverse([(1, 1)]) == [F(-2, 3), F(1, 3), F(-5, 3), F(-7, 3), 0]
    assert bi.left_multiply_by_basis_inverse([(0, 2), (1, 3), (2, 5), (3, 7)]) == [0, 1, 0, 0, 0]


def test_change_basis_needs_the_column_and_a_nonzero_pivot():
    bi = ExactBasisInverse.identity(3)
    with pytest.raises(RelpError) as e:
        bi.change_basis(0)
    assert e.value.status == ERR_STATE
    bi.left_multiply_by_basis_inverse([(1, 1)])
    with pytest.raises(RelpError):
        bi.change_basis(0)  # alpha_0 = 0: the new basis would be singular
    with pytest.raises(RelpError):
        ExactBasisInverse.invert([[(0, 1), (1, 2)], [(0, 2), (1, 4)]])  # singular columns


@pytest.mark.parametrize("m,steps,seed", [(4, 12, 1), (9, 30, 2), (25, 40, 3), (60, 25, 4)])
def test_random_rational_updates_against_an_exact_inverse(m, steps, seed):
    """Rational entries with denominators up to 9, numerators up to 12: the common denominator D grows past one, two, four words on the way."""
    rng = random.Random(seed)

    def random_column():
        return [(i, F(rng.randint(-12, 12) or 1, rng.randint(1, 9))) for i in sorted(rng.sample(range(m), min(m, rng.randint(1, 4))))]

    while True:
        B = [[F(0)] * m for _ in range(m)]
        for i, j in enumerate(rng.sample(range(m), m)):
            B[i][j] = F(rng.choice([-3, -2, -1, 1, 2, 5]), rng.randint(1, 4))
        for _ in range(2 * m):
            B[rng.randrange(m)][rng.randrange(m)] = F(rng.randint(-9, 9), rng.randint(1, 6))
        try:
            inverse = exact_inverse(B)
            break
        except StopIteration:
            continue
    bi = ExactBasisInverse.invert([[(i, B[i][j]) for i in range(m) if B[i][j]] for j in range(m)])
    words = [bi.result_words()]
    for step in range(steps):
        c = random_column()
        dense_c = dense_of(c, m)
        alpha = [sum(inverse[i][k] * dense_c[k] for k in range(m)) for i in range(m)]
        assert bi.left_multiply_by_basis_inverse(c) == alpha
        r = random_column()
        dense_r = dense_of(r, m)
        assert bi.right_multiply_by_basis_inverse(r) == [sum(dense_r[i] * inverse[i][k] for i in range(m)) for k in range(m)]
        row = rng.randrange(m)
        assert bi.basis_inverse_row(row) == inverse[row]
        # a row vector that is itself a result (what `Carry::change_basis` multiplies from the left): multi-word numerators over one denominator
        big = [(i, alpha[i] * F(10 ** 25 + 7, 3)) for i in range(m) if alpha[i] != 0]
        assert bi.right_multiply_by_basis_inverse(big) == [sum(v * inverse[i][k] for i, v in big) for k in range(m)]
        element = bi.generate_element(row, c)
        assert (element or 0) == alpha[row] and (element is None) == (alpha[row] == 0)
        candidates = [i for i in range(m) if alpha[i] != 0]
        if not candidates:
            continue
        p = rng.choice(candidates)
        bi.left_multiply_by_basis_inverse(c)
        bi.change_basis(p)
        for i in range(m):
            B[i][p] = dense_c[i]
        inverse = exact_inverse(B)
        words.append(bi.result_words())
    for i in range(m):
        assert bi.basis_inverse_row(i) == inverse[i]
    print("m = %d: words per integer %d -> %d over %d changes of basis" % (m, words[0], words[-1], steps))
    assert words[-1] >= words[0]


def test_remove_basis_part():  # carry/mod.rs:176-180; basis_inverse_rows.rs:212-229
    rng = random.Random(5)
    m, drop = 12, [2, 7]
    keep = [i for i in range(m) if i not in drop]
    while True:
        small = [[F(rng.randint(-5, 5), rng.randint(1, 3)) if rng.random() < 0.4 else F(0) for _ in keep] for _ in keep]
        for d in range(len(keep)):
            small[d][d] += F(7)
        try:
            inverse = exact_inverse(small)
            break
        except StopIteration:
            continue
    B = [[F(0)] * m for _ in range(m)]
    for a, i in enumerate(keep):
        for b, j in enumerate(keep):
            B[i][j] = small[a][b]
    for i in drop:  # what leaves: the unit column of an artificial variable that stayed basic on a redundant row ...
        B[i][i] = F(1)
        B[i][keep[0]] = F(3, 2)  # (... whose row may hold entries of the columns that stay)
    bi = ExactBasisInverse.invert([[(i, B[i][j]) for i in range(m) if B[i][j]] for j in range(m)])
    bi.remove_basis_part(drop)
    assert bi.m() == m - len(drop)
    for r in range(len(keep)):
        assert bi.basis_inverse_row(r) == inverse[r]
    assert bi.left_multiply_by_basis_inverse([(0, 1)]) == [inverse[r][0] for r in range(len(keep))]


# ---- the oracle's own Carry, every BasisInverse call mirrored on the device with == ------------------------------------------------
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
        """`BasisInverse` (carry/mod.rs:69-169): answers come from the exact `BasisInverseRows`; every call is repeated on the device
        object and compared with == (Fractions on both sides)."""

        def __init__(self, exact, device):
            self.exact = exact
            self.device = device
            self.last = None

        @classmethod
        def identity(cls, m):
            log["identity"] += 1
            return cls(BasisInverseRows.identity(m), ExactBasisInverse.identity(m))

        @classmethod
        def invert(cls, columns):
            columns = [list(c) for c in columns]
            log["invert"] += 1
            return cls(BasisInverseRows.invert(columns), ExactBasisInverse.invert(columns))

        def m(self):
            assert self.device.m() == self.exact.m()
            return self.exact.m()

        def left_multiply_by_basis_inverse(self, column):
            column = list(column)
            exact = self.exact.left_multiply_by_basis_inverse(column)
            assert self.device.left_multiply_by_basis_inverse(column) == dense_of(exact.into_column(), self.exact.m()), "FTRAN"
            self.last = column
            log["ftran"] += 1
            return Info(exact, column)

        def right_multiply_by_basis_inverse(self, row):
            row = list(row)
            exact = self.exact.right_multiply_by_basis_inverse(row)
            assert self.device.right_multiply_by_basis_inverse(row) == dense_of(exact, self.exact.m()), "BTRAN"
            log["btran"] += 1
            return exact

        def generate_element(self, i, column):
            column = list(column)
            exact = self.exact.generate_element(i, column)
            assert self.device.generate_element(i, column) == exact, "generate_element"
            log["element"] += 1
            return exact

        def should_refactor(self):
            log["should_refactor"] += 1
            assert self.device.should_refactor() is False
            return self.exact.should_refactor()

        def change_basis(self, pivot_row_index, info):
            if self.last is not info.original:  # the device keeps the column of its LAST left_multiply
                self.device.left_multiply_by_basis_inverse(info.original)
            self.device.change_basis(pivot_row_index)
            log["change_basis"] += 1
            return self.exact.change_basis(pivot_row_index, info.exact)

        def basis_inverse_row(self, row):
            exact = self.exact.basis_inverse_row(row)
            assert self.device.basis_inverse_row(row) == dense_of(exact, self.exact.m()), "row"
            log["row"] += 1
            return exact

        def remove_basis_part(self, indices):
            log["remove"] += 1
            self.exact.remove_basis_part(indices)
            self.device.remove_basis_part(list(indices))

    return Mirrored


@pytest.mark.parametrize("name", ["AFIRO", "SC50A", "ADLITTLE", "SC50B", "KB2", "SC105", "SHARE2B", "BLEND", "STOCFOR1", "ISRAEL"])
def test_the_oracles_carry_with_the_device_object_behind_every_call(name):
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
    assert log["change_basis"] + log["invert"] >= pivots and log["btran"] >= pivots and log["row"] >= pivots and log["ftran"] >= pivots


# (up to 1158 rows: larger than the metric's LP, whose own fixture holds the basis over the 820 rows its phase one keeps.  CZPROB is what
#  made `invert` scale the ROWS to integers: with every column scaled by its own lcm its 1158 columns outgrew 8192 bits on the way)
@pytest.mark.parametrize("name", ["E226", "SCFXM1", "BANDM", "STAIR", "ETAMACRO", "25FV47", "GFRD-PNC", "CZPROB"])
def test_invert_the_optimal_basis_of_a_netlib_lp(name):
    """At the size of a whole LP, through properties that need no second implementation: `invert` of the reference's OPTIMAL basis (the
    golden fixture's) -- 223 to 1158 columns of decimal data -- then B^-1 B = I column by column,
    x_B = B^-1 b >= 0, and c_B' x_B equal to the reference's exact optimum, digit for digit."""
    import json

    from relp_oracle.mps import load_problem
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
    general, provider = load_problem(os.path.join(ROOT, golden["file"]))
    m = provider.nr_rows()
    basis = golden["basis"]
    columns = [[(i, F(v)) for i, v in provider.column(j)] for j in basis]
    right_hand_side = list(provider.right_hand_side())
    if len(basis) == m - 1:
        # 25FV47, BASELINE configs[1]: one row is redundant, the reference's phase one removes it (`RemoveRows`) and its basis spans the 820
        # others.  The rows that may go are those with a non-zero entry in the left null vector of the 821 x 820 basis; a consistent
        # system gives the same x_B whichever of them goes.
        import numpy as np
        dense = np.zeros((m, m - 1))
        for slot, column in enumerate(columns):
            for i, v in column:
                dense[i, slot] = float(v)
        null_vector = np.linalg.svd(dense.T)[2][-1]  # (B' y = 0)
        gone = int(np.argmax(np.abs(null_vector)))
        columns = [[(i - (i > gone), v) for i, v in column if i != gone] for column in columns]
        del right_hand_side[gone]
        m -= 1
    assert len(basis) == m
    bi = ExactBasisInverse.invert(columns)
    for slot in list(range(0, m, 7)) + [m - 1]:  # (every seventh column: a left multiply is a launch and m Fractions back)
        assert bi.left_multiply_by_basis_inverse(columns[slot]) == dense_of([(slot, 1)], m), slot
    b = [(i, F(v)) for i, v in enumerate(right_hand_side) if v != 0]
    x_basic = bi.left_multiply_by_basis_inverse(b)
    assert all(v >= 0 for v in x_basic)
    solution = sorted((j, x_basic[slot]) for slot, j in enumerate(basis) if x_basic[slot] != 0)  # (sparse, by column: what the oracle's solve returns)
    objective = general.objective_of(provider.reconstruct_solution(solution))
    assert "%d/%d" % (objective.numerator, objective.denominator) == golden["objective"]
    # ... and a row of the inverse times the basis is a unit vector too (BTRAN side): row r of B^-1 against column `slot` of B
    row = bi.basis_inverse_row(m // 2)
    for slot in range(0, m, 11):
        assert sum(row[i] * v for i, v in columns[slot]) == (1 if slot == m // 2 else 0)
