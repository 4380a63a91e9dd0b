"""Explicit sparse rows of ``B^-1`` (oracle; test infrastructure only).

Follows ``tableau/inverse_maintenance/carry/basis_inverse_rows.rs``.
"""
from fractions import Fraction

from .lu import LUDecomposition
from .permutation import sorted_get

ZERO = Fraction(0)
ONE = Fraction(1)


def sparse_dot(a, b):
    """``index_utils::inner_product_slice_iter`` (sorted merge dot; data/linear_algebra/vector/sparse.rs:105-111)."""
    if len(a) > len(b):
        a, b = b, a
    lookup = dict(b)
    total = ZERO
    for i, v in a:
        other = lookup.get(i)
        if other is not None:
            total += v * other
    return total


def add_multiple_of_row(target, multiple, other):
    """data/linear_algebra/vector/sparse.rs:258-291: ``target += multiple * other`` dropping zeros."""
    values = dict(target)
    for j, v in other:
        new = values.get(j, ZERO) + multiple * v
        if new == 0:
            values.pop(j, None)
        else:
            values[j] = new
    return sorted(values.items())


class BasisInverseRows:
    """basis_inverse_rows.rs:21-23."""

    def __init__(self, rows):
        self.rows = [list(r) for r in rows]

    @classmethod
    def identity(cls, m):  # basis_inverse_rows.rs:92-96
        return cls([[(i, ONE)] for i in range(m)])

    @classmethod
    def invert(cls, columns):
        """basis_inverse_rows.rs:98-121: LU, then m FTRANs of the unit vectors."""
        columns = list(columns)
        m = len(columns)
        lu = LUDecomposition.invert(columns)
        row_major = [[] for _ in range(m)]
        for j in range(m):
            for i, value in lu.left_multiply_by_basis_inverse([(j, ONE)]).column:
                row_major[i].append((j, value))
        return cls(row_major)

    def m(self):
        return len(self.rows)

    def should_refactor(self):  # basis_inverse_rows.rs:197-201
        return False

    def change_basis(self, pivot_row_index, column):
        """basis_inverse_rows.rs:123-137 with normalize_pivot_row :36-45 and row_reduce :47-70."""
        column = column.into_column() if hasattr(column, "into_column") else column
        pos = sorted_get(column, pivot_row_index)
        assert pos is not None, "Pivot value can't be zero."
        pivot_value = column[pos][1]
        self.rows[pivot_row_index] = [(j, v / pivot_value) for j, v in self.rows[pivot_row_index]]
        pivot_row = self.rows[pivot_row_index]
        for i, value in column:
            if i != pivot_row_index:
                self.rows[i] = add_multiple_of_row(self.rows[i], -value, pivot_row)
        return column

    def left_multiply_by_basis_inverse(self, column):
        """basis_inverse_rows.rs:139-152."""
        column = [(i, Fraction(v)) for i, v in column]
        result = []
        for i in range(self.m()):
            value = self.generate_element(i, column)
            if value is not None:
                result.append((i, value))
        return _PlainColumn(result)

    def right_multiply_by_basis_inverse(self, row):
        """basis_inverse_rows.rs:154-170."""
        total = []
        for index, factor in row:
            total = add_multiple_of_row(total, Fraction(factor), self.rows[index])
        return total

    def generate_element(self, i, column):
        """basis_inverse_rows.rs:172-189."""
        value = sparse_dot(self.rows[i], list(column))
        return value if value != 0 else None

    def basis_inverse_row(self, row):  # basis_inverse_rows.rs:203-205
        return list(self.rows[row])

    def remove_basis_part(self, indices):
        """basis_inverse_rows.rs:212-229: drop the given rows and columns."""
        drop = set(indices)
        relabel = {}
        new_index = 0
        for i in range(self.m()):
            if i not in drop:
                relabel[i] = new_index
                new_index += 1
        self.rows = [[(relabel[j], v) for j, v in row if j not in drop]
                     for i, row in enumerate(self.rows) if i not in drop]

    def __eq__(self, other):
        return isinstance(other, BasisInverseRows) and self.rows == other.rows


class _PlainColumn:
    """``ColumnComputationInfo`` for ``SparseVector`` (basis_inverse_rows.rs:231-239)."""

    def __init__(self, column):
        self.column = column
        self.spike = None

    def into_column(self):
        return self.column
