"""Tableau and its kinds (oracle; test infrastructure only).

Follows ``tableau/mod.rs`` and ``tableau/kind/{mod.rs, artificial/*.rs, non_artificial.rs}``.
"""
from fractions import Fraction

from .carry import Carry

ZERO = Fraction(0)
ONE = Fraction(1)


class Fully:
    """kind/artificial/fully.rs:14-98: one artificial identity column per row, placed first."""

    def __init__(self, provider):
        self.provider = provider

    def nr_artificial_variables(self):
        return self.provider.nr_rows()

    def pivot_row_from_artificial(self, artificial_index):
        return artificial_index

    def initial_cost_value(self, j):  # fully.rs:27-33 (Binary cost)
        return ONE if j < self.nr_artificial_variables() else ZERO

    def original_column(self, j):  # fully.rs:35-43
        if j < self.nr_rows():
            return [(j, ONE)]
        return self.provider.column(j - self.nr_rows())

    def nr_rows(self):
        return self.provider.nr_rows()

    def nr_columns(self):
        return self.nr_rows() + self.provider.nr_columns()


class Partially:
    """kind/artificial/partially.rs:17-107: artificials only on rows without a free slack pivot."""

    def __init__(self, provider, column_to_row):
        self.provider = provider
        self.column_to_row = list(column_to_row)

    def nr_artificial_variables(self):
        return len(self.column_to_row)

    def pivot_row_from_artificial(self, artificial_index):
        return self.column_to_row[artificial_index]

    def initial_cost_value(self, j):  # partially.rs:42-50
        return ONE if j < self.nr_artificial_variables() else ZERO

    def original_column(self, j):  # partially.rs:52-60
        if j < self.nr_artificial_variables():
            return [(self.column_to_row[j], ONE)]
        return self.provider.column(j - self.nr_artificial_variables())

    def nr_rows(self):
        return self.provider.nr_rows()

    def nr_columns(self):
        return self.nr_artificial_variables() + self.provider.nr_columns()


class NonArtificial:
    """kind/non_artificial.rs:18-56: pass-through to the provider."""

    def __init__(self, provider):
        self.provider = provider

    def nr_artificial_variables(self):
        return 0

    def initial_cost_value(self, j):
        return self.provider.cost_value(j)

    def original_column(self, j):
        return self.provider.column(j)

    def nr_rows(self):
        return self.provider.nr_rows()

    def nr_columns(self):
        return self.provider.nr_columns()


class Tableau:
    """tableau/mod.rs:25-39."""

    def __init__(self, inverse_maintainer, basis_columns, kind):
        self.inverse_maintainer = inverse_maintainer
        self.basis_columns = set(basis_columns)
        self.kind = kind

    # ---- constructors -------------------------------------------------------------------------
    @classmethod
    def new_fully_artificial(cls, provider, bi_cls):
        """fully.rs:82-98."""
        m = provider.nr_rows()
        im = Carry.create_for_fully_artificial(bi_cls, provider.right_hand_side())
        return cls(im, range(m), Fully(provider))

    @classmethod
    def new_partially_artificial(cls, provider, bi_cls):
        """partially.rs:125-205."""
        m = provider.nr_rows()
        real = provider.pivot_element_indices()
        assert real == sorted(real, key=lambda t: t[0])
        real_rows = {row for row, _ in real}
        artificial = [row for row in range(m) if row not in real_rows]  # partially.rs:137-148
        nr_artificial = len(artificial)
        real_by_row = dict(real)
        artificial_by_row = {row: k for k, row in enumerate(artificial)}
        # partially.rs:156-185: merge in row order
        basis_indices = [artificial_by_row[i] if i in artificial_by_row else nr_artificial + real_by_row[i]
                         for i in range(m)]
        im = Carry.create_for_partially_artificial(
            bi_cls, artificial, real, provider.right_hand_side(), basis_indices)
        return cls(im, basis_indices, Partially(provider, artificial))

    @classmethod
    def new_with_inverse_maintainer(cls, provider, inverse_maintainer, basis_columns):
        """non_artificial.rs:59-73."""
        return cls(inverse_maintainer, basis_columns, NonArtificial(provider))

    @classmethod
    def new_with_basis(cls, provider, basis, bi_cls):
        """non_artificial.rs:75-97 (basis order is arbitrary in the reference; sorted here)."""
        order = sorted(basis)
        return cls(Carry.from_basis(bi_cls, order, provider), order, NonArtificial(provider))

    @classmethod
    def from_artificial(cls, inverse_maintainer, nr_artificial, basis_indices, provider):
        """non_artificial.rs:99-120."""
        im = Carry.from_artificial(inverse_maintainer, provider, nr_artificial)
        return cls(im, {c - nr_artificial for c in basis_indices}, NonArtificial(provider))

    @classmethod
    def from_artificial_removing_rows(cls, inverse_maintainer, nr_artificial, basis, provider):
        """non_artificial.rs:128-165."""
        basis = set(basis)
        for row in provider.filtered_rows():
            basis.remove(inverse_maintainer.basis_column_index_for_row(row))
        basis_columns = {j - nr_artificial for j in basis}
        im = Carry.from_artificial_remove_rows(inverse_maintainer, provider, nr_artificial)
        return cls(im, basis_columns, NonArtificial(provider))

    # ---- operations -----------------------------------------------------------------------------
    def nr_rows(self):
        return self.kind.nr_rows()

    def nr_columns(self):
        return self.kind.nr_columns()

    def nr_artificial_variables(self):
        return self.kind.nr_artificial_variables()

    def start_index(self):
        """strategy/pivot_rule.rs:57-80."""
        return self.kind.nr_artificial_variables()

    def bring_into_basis(self, pivot_column_index, pivot_row_index, info, cost):
        """tableau/mod.rs:48-64 and update_basis_indices :76-88."""
        change = self.inverse_maintainer.change_basis(
            pivot_row_index, pivot_column_index, info, cost, self.kind)
        self.basis_columns.remove(change.leaving_column_index)
        assert pivot_column_index not in self.basis_columns
        self.basis_columns.add(pivot_column_index)
        return change

    def relative_cost(self, j):
        """tableau/mod.rs:106-112."""
        return (self.inverse_maintainer.cost_difference(self.kind.original_column(j))
                + self.kind.initial_cost_value(j))

    def generate_column(self, j):
        """tableau/mod.rs:126-130."""
        return self.inverse_maintainer.generate_column(self.kind.original_column(j))

    def generate_element(self, i, j):
        """tableau/mod.rs:141-146."""
        return self.inverse_maintainer.generate_element(i, self.kind.original_column(j))

    def original_column(self, j):
        return self.kind.original_column(j)

    def is_in_basis(self, column):
        return column in self.basis_columns

    def variable_value(self, column):
        """tableau/mod.rs:164-176."""
        if self.is_in_basis(column):
            row = next(i for i in range(self.nr_rows())
                       if self.inverse_maintainer.basis_column_index_for_row(i) == column)
            return self.inverse_maintainer.get_constraint_value(row)
        return ZERO

    def current_bfs(self):
        return self.inverse_maintainer.current_bfs()

    def objective_function_value(self):
        return self.inverse_maintainer.get_objective_function_value()

    def select_primal_pivot_row(self, column):
        """tableau/mod.rs:287-313: ratio test, ties broken by the lowest leaving column (Bland)."""
        best = None  # (row, ratio, leaving column)
        for row, xij in column:
            if xij > 0:
                ratio = self.inverse_maintainer.get_constraint_value(row) / xij
                leaving = self.inverse_maintainer.basis_column_index_for_row(row)
                if best is None:
                    best = (row, ratio, leaving)
                elif ratio == best[1] and leaving < best[2]:
                    best = (row, best[1], leaving)
                elif ratio < best[1]:
                    best = (row, ratio, leaving)
        return None if best is None else best[0]

    # ---- artificial only ------------------------------------------------------------------------
    def has_artificial_in_basis(self):
        """kind/artificial/mod.rs:34-36."""
        return any(c < self.nr_artificial_variables() for c in self.basis_columns)

    def artificial_basis_columns(self):
        """kind/artificial/mod.rs:38-43: ``(row, artificial column)`` pairs sorted by row."""
        out = []
        for i in range(self.nr_rows()):
            j = self.inverse_maintainer.basis_column_index_for_row(i)
            if j < self.nr_artificial_variables():
                out.append((i, j))
        return out

    def into_basis(self):
        """kind/artificial/mod.rs:54-57."""
        return self.inverse_maintainer, self.nr_artificial_variables(), self.basis_columns

    def check_bfs_state(self):
        """tableau/mod.rs:319-357 (the reference's debug-build invariant check)."""
        m = self.nr_rows()
        assert len(self.basis_columns) == m
        for i in range(m):
            j = self.inverse_maintainer.basis_column_index_for_row(i)
            assert self.generate_column(j).into_column() == [(i, ONE)], (i, j)
            assert self.relative_cost(j) == 0, j
            assert self.inverse_maintainer.b[i] >= 0, i
