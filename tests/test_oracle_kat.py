"""Known-answer tests of the reference, replayed against the oracle (pins the oracle; CPU only).

Every expected value below is DATA taken from a ``#[test]`` of the reference; the citation names it.
Paths are relative to /root/reference/src/algorithm/two_phase/tableau/inverse_maintenance/carry/.
"""
from fractions import Fraction as F

import pytest

from relp_oracle import (BasisInverseRows, Carry, ColumnAndSpike, EtaFile, FullPermutation,
                         LUDecomposition, MatrixData, RotateToBack, Swap, Tableau, Variable)
from relp_oracle.lu import subtract_multiple_of_row_from_other_row
from relp_oracle import FirstProfitable, SteepestDescentAlongObjective
from relp_oracle import solve as S


def sv(*pairs):
    return [(i, F(v)) for i, v in pairs]


def lu(rp, cp, lower, upper, diag, updates=()):
    return LUDecomposition(FullPermutation(rp), FullPermutation(cp),
                           [sv(*c) for c in lower], [sv(*c) for c in upper], [F(d) for d in diag], list(updates))


def rows_of(matrix):
    return [[(j, F(v)) for j, v in enumerate(row) if v != 0] for row in matrix]


# ---- lower_upper/decomposition/mod.rs:319-438 -------------------------------------------------------
@pytest.mark.parametrize("rows, expected", [
    ([[(0, 1)], [(1, 1)]], lu([0, 1], [0, 1], [[]], [[]], [1, 1])),                                  # identity_2
    ([[(0, 1)], [(1, 1)], [(2, 1)]], lu([0, 1, 2], [0, 1, 2], [[], []], [[], []], [1, 1, 1])),     # identity_3
    ([[(0, 1), (1, 1)], [(1, 1)]], lu([0, 1], [0, 1], [[]], [[(0, 1)]], [1, 1])),                   # offdiagonal_2_upper
    ([[(0, 1)], [(0, 1), (1, 1)]], lu([0, 1], [0, 1], [[(1, 1)]], [[]], [1, 1])),                   # offdiagonal_2_lower
    ([[(0, 1), (1, 1)], [(0, 1)]], lu([1, 0], [0, 1], [[(1, 1)]], [[]], [1, 1])),                   # offdiagonal_2_both
    ([[(0, 4), (1, 3)], [(0, 6), (1, 3)]], lu([0, 1], [0, 1], [[(1, F(3, 2))]], [[(0, 3)]], [4, F(-3, 2)])),  # wikipedia_example
    ([[(0, -1), (1, F(3, 2))], [(0, 1), (1, -1)]],
     lu([0, 1], [0, 1], [[(1, -1)]], [[(0, F(3, 2))]], [-1, F(1, 2)])),                             # wikipedia_example2
])
def test_decomposition_exact_factors(rows, expected):
    assert LUDecomposition.rows([sv(*r) for r in rows]) == expected


def test_wikipedia_example2_columns():  # decomposition/mod.rs:427-437
    d = lu([0, 1], [0, 1], [[(1, -1)]], [[(0, F(3, 2))]], [-1, F(1, 2)])
    assert d.left_multiply_by_basis_inverse(sv((0, 1))).column == sv((0, 2), (1, 2))
    assert d.left_multiply_by_basis_inverse(sv((1, 1))).column == sv((0, 3), (1, 2))


MATRICES = {  # decomposition/mod.rs:480-651 (test_matrix inputs)
    "3x3": [[2, 3, 0], [5, 0, 11], [23, 29, 0]],
    "4x4_1": [[2, 3, 0, 5], [5, 0, 11, 13], [23, 29, 0, 57], [31, 37, 41, 0]],
    "4x4_2": [[-101, 0, 0, -5], [-110, -81, 0, 0], [0, 0, 1, -111], [0, 93, 69, 0]],
    "4x4_3": [[0, 0, -84, 122], [0, 9, 0, 0], [-39, 115, 0, 57], [0, -12, 121, 0]],
    "5x5_banded": [[2, 3, 0, 0, 0], [5, 7, 11, 0, 0], [0, 29, 13, 57, 0], [0, 0, 41, 17, 0], [0, 0, 0, 53, 51]],
    "5x5_1": [[29, 23, 0, 19, 0], [0, 0, 17, 13, 0], [0, 0, 7, 0, 0], [5, 0, 0, 3, 0], [0, 0, 0, 0, 2]],
    "5x5_2": [[29, 23, 0, 19, 0], [0, 0, 17, 13, 0], [0, 11, 7, 0, 0], [5, 0, 0, 3, 0], [0, 0, 0, 0, 2]],
    "5x5_3": [[2, 3, 0, 5, 7], [5, 0, 11, 13, 17], [23, 29, 0, 57, 59], [31, 37, 41, 0, 0], [43, 0, 47, 53, 51]],
    "5x5_5": [[0, 54, 43, 0, 84], [4, 0, 0, 0, 0], [0, -111, -27, 0, -86], [-6, 0, 0, 17, -62], [-109, 0, 0, 0, -104]],
    "5x5_6": [[-71, -124, 0, 0, -108], [0, 66, -121, -74, -53], [0, 104, 0, 0, 0], [0, 55, 0, 1, -3], [93, 0, 0, 0, 104]],
    "6x6_1": [[0, 0, 0, -25, 0, 0], [-15, 79, 0, 0, 0, 0], [0, 0, 0, 0, 0, 14], [0, 0, 0, -114, -61, 0],
              [0, 0, 109, 0, 0, -126], [46, 0, 0, 50, 21, 0]],
    "6x6_2": [[0, 0, -26, -68, 84, 0], [-125, 43, 0, 0, 0, -63], [0, 0, 1, 90, 0, 0], [0, -81, 0, 0, 0, 0],
              [-15, 0, 0, -81, 0, 0], [0, -12, 0, 0, 0, 1]],
    "10x10_1": [
        [0, 0, 0, 0, 0, 0, -60, 0, -10, 0], [0, 0, 0, 0, 0, 0, 0, -84, 0, 0], [0, -105, 0, 0, 0, 0, 0, 0, 0, 0],
        [0, 0, 0, -25, 0, 0, 0, 0, 116, 0], [0, 0, 0, 0, -18, 0, 0, 0, 0, 0], [0, 0, 0, -72, 0, 0, 0, 0, 0, 0],
        [0, 0, 16, 48, 0, 0, 0, 0, 0, 0], [-57, 0, 0, -88, 107, 0, 0, 0, 0, 0],
        [-122, -108, 0, 0, 0, 91, 0, 0, -127, 0], [0, 85, 0, 0, 106, 0, 0, 0, 0, -121]],
    "11x11_1": [
        [0, 0, 0, 0, 0, 0, 0, 0, 0, -13, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -122],
        [0, 0, 0, 0, 0, 102, 82, 0, 0, 0, 13], [0, 0, 0, 0, 0, -107, -39, 0, 0, 0, 0],
        [0, 0, 0, 0, 0, 0, -39, 48, -113, 0, 0], [24, 0, 0, 0, 0, 0, 0, 0, 0, -93, -120],
        [-111, 0, 0, -81, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 82, 0, 0, 76, 0, 0],
        [0, 0, -51, 0, 0, 0, 126, 0, 0, 0, -105], [0, 118, 0, 0, 0, 0, 0, 0, 0, 0, 27],
        [0, 0, 120, 0, -31, 0, 0, 0, 0, 0, 0]],
}


@pytest.mark.parametrize("name", sorted(MATRICES))
def test_matrix_inverse_property(name):
    """decomposition/mod.rs:454-478: B^-1 B[:,j] == e_j and B[i,:] B^-1 == e_i."""
    matrix = MATRICES[name]
    m = len(matrix)
    d = LUDecomposition.rows(rows_of(matrix))
    for j in range(m):
        column = [(i, F(matrix[i][j])) for i in range(m) if matrix[i][j] != 0]
        assert d.left_multiply_by_basis_inverse(column).column == sv((j, 1)), j
    for i in range(m):
        row = [(j, F(v)) for j, v in enumerate(matrix[i]) if v != 0]
        assert d.right_multiply_by_basis_inverse(row) == sv((i, 1)), i
    # cross-check the alternative implementation (basis_inverse_rows.rs:98-121)
    columns = [[(i, F(matrix[i][j])) for i in range(m) if matrix[i][j] != 0] for j in range(m)]
    rows = BasisInverseRows.invert(columns)
    for j in range(m):
        assert rows.left_multiply_by_basis_inverse(columns[j]).column == sv((j, 1))
        assert rows.basis_inverse_row(j) == d.basis_inverse_row(j)


@pytest.mark.parametrize("a, ratio, b, expected", [  # decomposition/mod.rs:653-713
    ([], 1, [], []),
    ([], 1, [(1, 1)], [(1, -1)]),
    ([(1, 1)], 1, [], [(1, 1)]),
    ([(1, 1)], 1, [(2, 3)], [(1, 1), (2, -3)]),
    ([(1, 1)], 1, [(1, 3)], [(1, -2)]),
    ([(1, 1)], F(1, 3), [(1, 3)], []),
    ([(1, 1)], 1, [(0, 3)], [(0, -3), (1, 1)]),
])
def test_subtract_multiple_of_row(a, ratio, b, expected):
    new, _, _ = subtract_multiple_of_row_from_other_row(sv(*a), F(ratio), sv(*b))
    assert new == sv(*expected)


# ---- lower_upper/mod.rs:536-685 (matmul) -------------------------------------------------------------
def test_matmul_identity():
    ident = LUDecomposition.identity(2)
    for column in ([], sv((0, 1)), sv((1, 1)), sv((0, 1), (1, 1))):
        assert ident._left_multiply_by_upper_inverse(list(column)) == column
        assert ident._right_multiply_by_upper_inverse(list(column)) == column
        assert ident._left_multiply_by_lower_inverse(list(column)) == column
        assert ident._right_multiply_by_lower_inverse(list(column)) == column


def test_matmul_offdiagonal():
    off = lu([0, 1], [0, 1], [[(1, 1)]], [[]], [1, 1])
    assert off.left_multiply_by_basis_inverse([]).column == []
    assert off.left_multiply_by_basis_inverse(sv((0, 1))).column == sv((0, 1), (1, -1))
    assert off.left_multiply_by_basis_inverse(sv((1, 1))).column == sv((1, 1))


def test_matmul_dense():
    dense = lu([1, 0], [0, 1], [[(1, F(1, 3))]], [[(0, 4)]], [3, F(2, 3)])
    assert dense.left_multiply_by_basis_inverse(sv((0, 1))).column == sv((0, -2), (1, F(3, 2)))
    assert dense.left_multiply_by_basis_inverse(sv((1, 1))).column == sv((0, 1), (1, F(-1, 2)))
    assert dense.right_multiply_by_basis_inverse(sv((0, 1))) == sv((0, -2), (1, 1))
    assert dense.right_multiply_by_basis_inverse(sv((1, 1))) == sv((0, F(3, 2)), (1, F(-1, 2)))


# ---- lower_upper/mod.rs:688-940 (change_basis, Forrest-Tomlin) ------------------------------------
def info(spike, m):
    return ColumnAndSpike(sv(*spike), sv(*spike))


def test_ft_no_change():
    d = LUDecomposition.identity(3)
    d.change_basis(1, info([(1, 1)], 3))
    expected = LUDecomposition.identity(3)
    expected.updates.append((EtaFile([], 1, 3), RotateToBack(1, 3)))
    assert d == expected


def test_ft_from_identity_2():
    d = LUDecomposition.identity(2)
    d.change_basis(0, info([(0, 1), (1, 1)], 2))
    assert d == lu([0, 1], [0, 1], [[]], [[(0, 1)]], [1, 1], [(EtaFile([], 0, 2), RotateToBack(0, 2))])


def test_ft_from_5x5_identity_no_r():
    d = LUDecomposition.identity(5)
    d.change_basis(1, info([(0, 2), (1, 3), (2, 5), (3, 7)], 5))
    assert d == lu(range(5), range(5), [[]] * 4, [[], [], [], [(0, 2), (1, 5), (2, 7)]], [1, 1, 1, 1, 3],
                   [(EtaFile([], 1, 5), RotateToBack(1, 5))])


def test_ft_from_4x4_identity():
    m = 4
    d = lu(range(m), range(m), [[]] * 3, [[], [], [(1, 5)]], [1, 1, 4, 6])
    d.change_basis(1, info([(1, 2), (2, 3), (3, 4)], m))
    assert d == lu(range(m), range(m), [[]] * 3, [[], [], [(1, 3), (2, 4)]], [1, 4, 6, F(-8, 6)],
                   [(EtaFile(sv((3, F(5, 6))), 1, m), RotateToBack(1, m))])
    cols = [sv((0, 1)),
            sv((1, F(-3, 4)), (2, F(9, 16)), (3, F(1, 2))),
            sv((2, F(1, 4))),
            sv((1, F(5, 8)), (2, F(-15, 32)), (3, F(-1, 4)))]
    for j in range(m):
        assert d.left_multiply_by_basis_inverse(sv((j, 1))).column == cols[j]
    rows = [sv((0, 1)),
            sv((1, F(-3, 4)), (3, F(5, 8))),
            sv((1, F(9, 16)), (2, F(1, 4)), (3, F(-15, 32))),
            sv((1, F(1, 2)), (3, F(-1, 4)))]
    for i in range(m):
        assert d.basis_inverse_row(i) == rows[i]


def test_ft_from_5x5_identity_elble_sahinidis():
    m = 5
    d = lu(range(m), range(m), [[]] * 4,
           [[(0, 12)], [(0, 13), (1, 23)], [(0, 14), (1, 24), (2, 34)], [(0, 15), (1, 25), (2, 35), (3, 45)]],
           [11, 22, 33, 44, 55])
    d.change_basis(1, info([(0, 12), (1, 22), (2, 32), (3, 42)], m))
    eta = EtaFile(sv((2, F(23, 33)), (3, F(24 * 33 - 34 * 23, 33 * 44)), (4, F(43, 7986))), 1, m)
    assert d == lu(range(m), range(m), [[]] * 4,
                   [[(0, 13)], [(0, 14), (1, 34)], [(0, 15), (1, 35), (2, 45)], [(0, 12), (1, 32), (2, 42)]],
                   [11, 33, 44, 55, F(-215, 363)], [(eta, RotateToBack(1, m))])
    cols = [sv((0, F(1, 11))),
            sv((0, F(-2, 11)), (1, F(-363, 215)), (2, F(-1, 43)), (3, F(693, 430))),
            sv((0, F(1, 11)), (1, F(253, 215)), (2, F(2, 43)), (3, F(-483, 430))),
            sv((1, F(1, 86)), (2, F(-1, 43)), (3, F(1, 86))),
            sv((1, F(1, 110)), (3, F(-3, 110)), (4, F(1, 55)))]
    for j in range(m):
        assert d.left_multiply_by_basis_inverse(sv((j, 1))).column == cols[j]
    assert d.left_multiply_by_basis_inverse(sv((0, 1), (1, 1))).column == \
        sv((0, F(-1, 11)), (1, F(-363, 215)), (2, F(-1, 43)), (3, F(693, 430)))
    rows = [sv((0, F(1, 11)), (1, F(-2, 11)), (2, F(1, 11))),
            sv((1, F(-363, 215)), (2, F(253, 215)), (3, F(1, 86)), (4, F(1, 110))),
            sv((1, F(-1, 43)), (2, F(2, 43)), (3, F(-1, 43))),
            sv((1, F(693, 430)), (2, F(-483, 430)), (3, F(1, 86)), (4, F(-3, 110))),
            sv((4, F(1, 55)))]
    for i in range(m):
        assert d.basis_inverse_row(i) == rows[i]


# ---- lower_upper/eta_file.rs:160-263 -------------------------------------------------------------------
@pytest.mark.parametrize("values, pivot, n, vec, right, left", [
    ([], 0, 1, [(0, 1)], [(0, 1)], [(0, 1)]),
    ([], 0, 2, [(0, 1)], [(0, 1)], [(0, 1)]),
    ([], 1, 2, [(0, 1)], [(0, 1)], [(0, 1)]),
    ([(1, 1)], 0, 2, [(0, 13), (1, 17)], [(0, 13 - 17), (1, 17)], [(0, 13), (1, 17 - 13)]),
    ([(1, 1)], 0, 2, [], [], []),
    ([(1, 5), (2, 7)], 0, 3, [(0, 13), (1, 17), (2, 19)],
     [(0, 13 - 5 * 17 - 7 * 19), (1, 17), (2, 19)], [(0, 13), (1, -5 * 13 + 17), (2, -7 * 13 + 19)]),
    ([(1, 5)], 0, 3, [(0, 13), (1, 17), (2, 19)], [(0, 13 - 5 * 17), (1, 17), (2, 19)], [(0, 13), (1, -5 * 13 + 17), (2, 19)]),
    ([(2, 5)], 0, 3, [(0, 13), (1, 17), (2, 19)], [(0, 13 - 5 * 19), (1, 17), (2, 19)], [(0, 13), (1, 17), (2, -5 * 13 + 19)]),
])
def test_eta_file(values, pivot, n, vec, right, left):
    eta = EtaFile(sv(*values), pivot, n)
    v = sv(*vec)
    eta.apply_right(v)
    assert v == sv(*right)
    v = sv(*vec)
    eta.apply_left(v)
    assert v == sv(*left)


def test_eta_file_many():  # eta_file.rs:251-262
    eta = EtaFile(sv((1, 2), (2, 3), (5, 5), (7, 7), (11, 11), (12, 13)), 0, 14)
    v = sv((0, 17), (1, 19), (3, 23), (5, 29), (6, 31), (9, 37), (11, 41))
    eta.apply_right(v)
    assert v == sv((0, 17 - 2 * 19 - 5 * 29 - 11 * 41), (1, 19), (3, 23), (5, 29), (6, 31), (9, 37), (11, 41))
    v = sv((0, 13), (1, 19), (3, 23), (5, 29), (6, 31), (9, 37), (11, 41))
    eta.apply_left(v)
    assert v == sv((0, 13), (1, 19 - 2 * 13), (2, -3 * 13), (3, 23), (5, 29 - 5 * 13), (6, 31), (7, -7 * 13),
                   (9, 37), (11, 41 - 11 * 13), (12, -13 * 13))


# ---- permutations (full.rs:136-259, rotate_to_back.rs:130-209, swap.rs:86-167) ---------------------
def test_permutations():
    p = FullPermutation([2, 0, 1])
    assert [p.forward(i) for i in range(3)] == [2, 0, 1]
    assert [p.backward(i) for i in range(3)] == [1, 2, 0]
    assert all(p.backward(p.forward(i)) == i for i in range(3))
    p.invert()
    assert [p.forward(i) for i in range(3)] == [1, 2, 0]
    r = RotateToBack(1, 4)
    assert [r.forward(i) for i in range(4)] == [0, 3, 1, 2]
    assert [r.backward(i) for i in range(4)] == [0, 2, 3, 1]
    assert r.forward_sorted([(0, "a"), (1, "b"), (3, "c")]) == [(0, "a"), (2, "c"), (3, "b")]
    assert r.backward_sorted([(0, "a"), (2, "c"), (3, "b")]) == [(0, "a"), (1, "b"), (3, "c")]
    assert r.backward_sorted([]) == []
    s = Swap((0, 2), 3)
    assert s.forward_sorted([(0, "a"), (1, "b")]) == [(1, "b"), (2, "a")]
    assert s.forward_sorted([(0, "a"), (2, "c")]) == [(0, "c"), (2, "a")]
    q = FullPermutation.identity(4)
    q.swap(0, 3)
    assert q.fwd == [3, 1, 2, 0] and q.bwd == [3, 1, 2, 0]
    q = FullPermutation([1, 2, 0, 3])
    q.rotate_right_from(1)
    assert q.fwd == [1, 3, 2, 0]


# ---- basis_inverse_rows.rs:282-331 ------------------------------------------------------------------
def test_basis_inverse_rows():
    assert BasisInverseRows.invert([sv((0, 1)), sv((1, 1))]) == BasisInverseRows.identity(2)
    assert BasisInverseRows.invert([sv((0, 1), (1, 1)), sv((1, 1))]) == \
        BasisInverseRows([sv((0, 1)), sv((0, -1), (1, 1))])
    bi = BasisInverseRows.identity(2)
    bi.remove_basis_part([1])
    assert bi == BasisInverseRows.identity(1)


# ---- src/tests/problem_2.rs fixture ---------------------------------------------------------------
def problem_2():
    rows = [[3, 2, 1, 0, 0], [5, 1, 1, 1, 0], [2, 5, 1, 0, 1]]
    constraints = [[(i, F(rows[i][j])) for i in range(3) if rows[i][j] != 0] for j in range(5)]
    variables = [Variable(1) for _ in range(5)]
    return MatrixData(constraints, [1, 3, 4], [], 3, 0, 0, 0, variables)


def artificial_tableau(data):  # problem_2.rs:126-147
    return Tableau.new_partially_artificial(data, BasisInverseRows)


def phase2_tableau(data):  # problem_2.rs:149-181
    rows = [sv((0, F(1, 2))), sv((0, F(-1, 2)), (1, 1)), sv((0, F(-5, 2)), (2, 1))]
    carry = Carry(F(-9, 2), [F(5, 2), -1, -1], [F(1, 2), F(5, 2), F(3, 2)], [1, 3, 4], BasisInverseRows(rows))
    return Tableau.new_with_inverse_maintainer(data, carry, {1, 3, 4})


def test_problem_2_artificial_tableau_form():  # problem_2.rs:29-36
    t = artificial_tableau(problem_2())
    im = t.inverse_maintainer
    assert im.minus_objective == -8 and im.minus_pi == [-1, -1, -1] and im.b == [1, 3, 4]
    assert im.basis_indices == [0, 1, 2] and t.basis_columns == {0, 1, 2}
    assert t.kind.column_to_row == [0, 1, 2]


def test_tableau_ops():  # tableau/mod.rs:491-603
    data = problem_2()
    art = artificial_tableau(data)
    assert art.objective_function_value() == 8
    assert art.relative_cost(0) == 0
    assert art.relative_cost(art.nr_artificial_variables()) == -10
    carry = Carry(-6, [1, -1, -1], [1, 2, 3], [2, 3, 4],
                  BasisInverseRows([sv((0, 1)), sv((0, -1), (1, 1)), sv((0, -1), (2, 1))]))
    t = Tableau.new_with_inverse_maintainer(data, carry, {2, 3, 4})
    assert t.objective_function_value() == 6
    assert [t.relative_cost(j) for j in range(3)] == [-3, -3, 0]
    assert art.generate_column(3).into_column() == sv((0, 3), (1, 5), (2, 2))
    assert t.generate_column(0).into_column() == sv((0, 3), (1, 2), (2, -1))
    # bring_into_basis
    column = art.generate_column(3)
    row = art.select_primal_pivot_row(column.into_column())
    art.bring_into_basis(3, row, column, art.relative_cost(3))
    assert art.is_in_basis(3) and not art.is_in_basis(0)
    assert art.objective_function_value() == F(14, 3)
    column = t.generate_column(1)
    row = t.select_primal_pivot_row(column.into_column())
    t.bring_into_basis(1, row, column, t.relative_cost(1))
    assert t.is_in_basis(1) and t.objective_function_value() == F(9, 2)


def test_create_tableau_no_profitable_column():  # tableau/mod.rs:594-602
    m = 3
    carry = Carry(0, [1, 1, 1], [1, 2, 3], [m + 2, m + 3, m + 4],
                  BasisInverseRows([sv((0, 1)), sv((0, -1), (1, 1)), sv((0, -1), (2, 1))]))
    t = Tableau.new_with_inverse_maintainer(problem_2(), carry, {m + 2, m + 3, m + 4})
    # NOTE: the reference builds this tableau on a 5-column provider with out-of-range basis ids;
    # only "no column is selected" is asserted.
    assert FirstProfitable(t).select_primal_pivot_column(t) is None


def test_pivot_rule_and_ratio():  # strategy/pivot_rule.rs:314-344
    data = problem_2()
    art = artificial_tableau(data)
    assert FirstProfitable(art).select_primal_pivot_column(art)[0] == 3
    t = phase2_tableau(data)
    assert FirstProfitable(t).select_primal_pivot_column(t) is None
    assert art.select_primal_pivot_row(sv((0, 3), (1, 5), (2, 2))) == 0
    assert art.select_primal_pivot_row(sv((0, 2), (1, 1), (2, 5))) == 0
    assert t.select_primal_pivot_row(sv((0, 3), (1, 2), (2, -1))) == 0
    assert t.select_primal_pivot_row(sv((0, 2), (1, -1), (2, 3))) == 0


def test_conversion_pipeline():  # problem_2.rs:29-67 and phase_one.rs:293-305
    data = problem_2()
    result = S.phase_one_primal(artificial_tableau(data), FirstProfitable)
    assert result is not None
    rows_to_remove, nr_artificial, im, basis = result
    assert rows_to_remove == []
    t = Tableau.from_artificial(im, nr_artificial, basis, data)
    assert t.inverse_maintainer == phase2_tableau(data).inverse_maintainer
    assert t.basis_columns == {1, 3, 4}
    out = S.phase_two_primal(t, FirstProfitable)
    assert out.solution == sv((1, F(1, 2)), (3, F(5, 2)), (4, F(3, 2)))
    assert t.objective_function_value() == F(9, 2)  # two_phase/test.rs:19-29


# ---- two_phase/test.rs:31-212 ---------------------------------------------------------------------
@pytest.mark.parametrize("bi", [LUDecomposition, BasisInverseRows])
@pytest.mark.parametrize("rule", [SteepestDescentAlongObjective, FirstProfitable])
def test_solve_matrix(bi, rule):
    out = S.solve_relaxation(problem_2(), bi, rule, check=True)
    assert out.solution == sv((1, F(1, 2)), (3, F(5, 2)), (4, F(3, 2)))


def test_solve_relaxation_1():
    constraints = [sv((0, 1), (1, 1)), sv((1, 1))]
    data = MatrixData(constraints, [F(3, 2), F(5, 2)], [], 0, 0, 2, 0, [Variable(-2), Variable(-1)])
    out = S.solve_relaxation(data, LUDecomposition, check=True)
    assert out.solution == sv((0, F(3, 2)), (1, 1))


def _bounded_variables():
    return [Variable(-2, upper_bound=F(3, 4)), Variable(-1)]


@pytest.mark.parametrize("bi", [BasisInverseRows, LUDecomposition])
def test_redundant_row(bi):
    constraints = [sv((0, 1), (1, 1), (2, 1)), sv((0, 1), (1, 1), (2, 1))]
    data = MatrixData(constraints, [1, 1, 1], [], 3, 0, 0, 0, _bounded_variables())
    out = S.solve_relaxation(data, bi)
    assert out.solution == sv((0, F(3, 4)), (1, F(1, 4)))  # dense [3/4, 1/4, 0]


@pytest.mark.parametrize("bi", [BasisInverseRows, LUDecomposition])
def test_empty_row_at_eq(bi):
    constraints = [sv((0, 1)), sv((0, 1))]
    data = MatrixData(constraints, [1, 0], [], 2, 0, 0, 0, _bounded_variables())
    out = S.solve_relaxation(data, bi)
    assert out.solution == sv((0, F(3, 4)), (1, F(1, 4)))


@pytest.mark.parametrize("bi", [BasisInverseRows, LUDecomposition])
def test_empty_row_at_ineq(bi):
    constraints = [sv((0, 1)), sv((0, 1))]
    data = MatrixData(constraints, [1, 1], [], 1, 0, 1, 0, _bounded_variables())
    out = S.solve_relaxation(data, bi)
    assert out.solution == sv((0, F(3, 4)), (1, F(1, 4)), (2, 1))  # dense [3/4, 1/4, 1, 0]
