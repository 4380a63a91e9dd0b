"""The "carry" inverse maintainer (oracle; test infrastructure only).

Follows ``tableau/inverse_maintenance/carry/mod.rs``: state ``{-objective, -pi, b, basis_indices, B^-1}``.
"""
from fractions import Fraction

from .permutation import sorted_get

ZERO = Fraction(0)
ONE = Fraction(1)


class BasisChangeComputationInfo:
    """tableau/mod.rs:205-234."""

    def __init__(self, pivot_row_index, pivot_column_index, leaving_column_index,
                 column_before_change, work_vector, basis_inverse_row):
        self.pivot_row_index = pivot_row_index
        self.pivot_column_index = pivot_column_index
        self.leaving_column_index = leaving_column_index
        self.column_before_change = column_before_change  # alpha_q = B^-1 a_q (old basis)
        self.work_vector = work_vector                    # w = alpha_q' B^-1 (old basis)
        self.basis_inverse_row = basis_inverse_row        # rho_p = e_p' B^-1 (new basis)


class Carry:
    """carry/mod.rs:46-66."""

    def __init__(self, minus_objective, minus_pi, b, basis_indices, basis_inverse):
        self.minus_objective = Fraction(minus_objective)
        self.minus_pi = [Fraction(v) for v in minus_pi]
        self.b = [Fraction(v) for v in b]
        self.basis_indices = list(basis_indices)
        self.basis_inverse = basis_inverse

    def m(self):
        return len(self.b)

    # ---- constructors -------------------------------------------------------------------------
    @classmethod
    def create_for_fully_artificial(cls, bi_cls, b):
        """carry/mod.rs:374-395."""
        m = len(b)
        return cls(-sum(b, ZERO), [-ONE] * m, b, list(range(m)), bi_cls.identity(m))

    @classmethod
    def create_for_partially_artificial(cls, bi_cls, artificial_rows, free_basis_values, b, basis_indices):
        """carry/mod.rs:397-442."""
        m = len(b)
        assert len(artificial_rows) + len(free_basis_values) == m
        objective = sum((b[i] for i in artificial_rows), ZERO)
        artificial = set(artificial_rows)
        minus_pi = [-ONE if row in artificial else ZERO for row in range(m)]
        return cls(-objective, minus_pi, b, basis_indices, bi_cls.identity(m))

    @classmethod
    def from_basis(cls, bi_cls, basis, provider):
        """carry/mod.rs:444-478."""
        basis_inverse = bi_cls.invert(provider.column(j) for j in basis)
        rhs = [(i, v) for i, v in enumerate(provider.right_hand_side()) if v != 0]
        b = [ZERO] * provider.nr_rows()
        for i, v in basis_inverse.left_multiply_by_basis_inverse(rhs).into_column():
            b[i] = v
        minus_objective = cls._minus_obj_from_artificial(provider, basis, b)
        minus_pi = cls._minus_pi_from_artificial(basis_inverse, provider, basis)
        return cls(minus_objective, minus_pi, b, list(basis), basis_inverse)

    @classmethod
    def from_basis_pivots(cls, bi_cls, basis_columns, provider):
        """carry/mod.rs:480-497: sort the ``(row, column)`` pivots by row."""
        columns = [column for _, column in sorted(basis_columns, key=lambda t: t[0])]
        return cls.from_basis(bi_cls, columns, provider)

    @staticmethod
    def _minus_pi_from_artificial(basis_inverse, provider, basis):
        """carry/mod.rs:226-260: all of ``B^-1`` by m FTRANs, then ``pi_j = sum_i Binv[i][j] c_{basis[i]}``."""
        m = basis_inverse.m()
        pi = [ZERO] * m
        for j in range(m):
            for i, value in basis_inverse.left_multiply_by_basis_inverse([(j, ONE)]).into_column():
                pi[j] += value * provider.cost_value(basis[i])
        return [-v for v in pi]

    @staticmethod
    def _minus_obj_from_artificial(provider, basis, b):
        """carry/mod.rs:270-283."""
        return -sum((b[row] * provider.cost_value(basis[row]) for row in range(provider.nr_rows())), ZERO)

    @classmethod
    def from_artificial(cls, artificial, provider, nr_artificial):
        """carry/mod.rs:499-525."""
        basis_indices = [index - nr_artificial for index in artificial.basis_indices]
        minus_pi = cls._minus_pi_from_artificial(artificial.basis_inverse, provider, basis_indices)
        minus_objective = cls._minus_obj_from_artificial(provider, basis_indices, artificial.b)
        return cls(minus_objective, minus_pi, artificial.b, basis_indices, artificial.basis_inverse)

    @classmethod
    def from_artificial_remove_rows(cls, artificial, rows_removed, nr_artificial):
        """carry/mod.rs:527-559 (default: re-invert) and :663-708 (``RemoveBasisPart`` specialisation)."""
        skip = set(rows_removed.filtered_rows())
        basis_indices = [j - nr_artificial for i, j in enumerate(artificial.basis_indices) if i not in skip]
        if hasattr(artificial.basis_inverse, "remove_basis_part"):
            basis_inverse = artificial.basis_inverse
            basis_inverse.remove_basis_part(rows_removed.filtered_rows())
        else:
            basis_inverse = type(artificial.basis_inverse).invert(rows_removed.column(j) for j in basis_indices)
        minus_pi = cls._minus_pi_from_artificial(basis_inverse, rows_removed, basis_indices)
        b = [v for i, v in enumerate(artificial.b) if i not in skip]
        minus_objective = cls._minus_obj_from_artificial(rows_removed, basis_indices, b)
        return cls(minus_objective, minus_pi, b, basis_indices, basis_inverse)

    # ---- operations -----------------------------------------------------------------------------
    def change_basis(self, pivot_row_index, pivot_column_index, info, relative_cost, kind):
        """carry/mod.rs:561-604."""
        column = info.into_column()
        work_vector = self.basis_inverse.right_multiply_by_basis_inverse(column)
        self._update_b(pivot_row_index, column)
        leaving = self.basis_indices[pivot_row_index]
        self.basis_indices[pivot_row_index] = pivot_column_index
        if self.basis_inverse.should_refactor():
            self.basis_inverse = type(self.basis_inverse).invert(
                kind.original_column(j) for j in self.basis_indices)
            column_before_change = column
        else:
            column_before_change = self.basis_inverse.change_basis(pivot_row_index, info)
        basis_inverse_row = self.basis_inverse.basis_inverse_row(pivot_row_index)
        self._update_minus_pi_and_obj(pivot_row_index, relative_cost, basis_inverse_row)
        return BasisChangeComputationInfo(pivot_row_index, pivot_column_index, leaving,
                                          column_before_change, work_vector, basis_inverse_row)

    def _update_b(self, pivot_row_index, column):
        """carry/mod.rs:295-325."""
        pos = sorted_get(column, pivot_row_index)
        assert pos is not None, "Pivot value can't be zero."
        self.b[pivot_row_index] /= column[pos][1]
        pivot_b = self.b[pivot_row_index]
        for i, value in column:
            if i != pivot_row_index:
                self.b[i] -= value * pivot_b

    def _update_minus_pi_and_obj(self, pivot_row_index, relative_cost, basis_inverse_row):
        """carry/mod.rs:338-349."""
        for j, value in basis_inverse_row:
            self.minus_pi[j] -= relative_cost * value
        self.minus_objective -= relative_cost * self.b[pivot_row_index]

    def cost_difference(self, column):
        """carry/mod.rs:606-611 -> dense.rs:101-112 (gather dot with ``-pi``)."""
        total = ZERO
        for i, v in column:
            total += self.minus_pi[i] * v
        return total

    def generate_column(self, column):
        """carry/mod.rs:613-621."""
        return self.basis_inverse.left_multiply_by_basis_inverse(column)

    def generate_element(self, i, column):
        """carry/mod.rs:623-634."""
        return self.basis_inverse.generate_element(i, column)

    def current_bfs(self):
        """carry/mod.rs:636-645."""
        return sorted((self.basis_indices[i], v) for i, v in enumerate(self.b) if v != 0)

    def basis_column_index_for_row(self, row):
        return self.basis_indices[row]

    def get_objective_function_value(self):
        return -self.minus_objective

    def get_constraint_value(self, i):
        return self.b[i]

    def __eq__(self, other):
        return (isinstance(other, Carry) and self.minus_objective == other.minus_objective
                and self.minus_pi == other.minus_pi and self.b == other.b
                and self.basis_indices == other.basis_indices
                and self.basis_inverse == other.basis_inverse)
