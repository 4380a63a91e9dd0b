"""Pivot (pricing) rules (oracle; test infrastructure only).

Follows ``strategy/pivot_rule.rs``.
"""
from fractions import Fraction

from .inverse_rows import sparse_dot
from .permutation import sorted_get

ONE = Fraction(1)


def _candidates(tableau, columns):
    for j in columns:
        if not tableau.is_in_basis(j):
            yield j, tableau.relative_cost(j)


class FirstProfitable:
    """pivot_rule.rs:86-109: first column with a negative relative cost."""

    def __init__(self, tableau):
        pass

    def select_primal_pivot_column(self, tableau):
        for j, cost in _candidates(tableau, range(tableau.start_index(), tableau.nr_columns())):
            if cost < 0:
                return j, cost
        return None

    def after_basis_update(self, info, tableau):
        pass


class FirstProfitableWithMemory:
    """pivot_rule.rs:113-150: continue the scan after the previously selected column."""

    def __init__(self, tableau):
        self.last_selected = None

    def select_primal_pivot_column(self, tableau):
        def find(columns):
            for j, cost in _candidates(tableau, columns):
                if cost < 0:
                    return j, cost
            return None
        if self.last_selected is None:
            potential = find(range(tableau.start_index(), tableau.nr_columns()))
        else:
            last = self.last_selected
            potential = find(range(last + 1, tableau.nr_columns())) or find(range(tableau.start_index(), last))
        self.last_selected = None if potential is None else potential[0]
        return potential

    def after_basis_update(self, info, tableau):
        pass


class SteepestDescentAlongVariable:
    """pivot_rule.rs:153-187 (Dantzig): most negative cost, *first* minimum on ties (strict ``<``)."""

    def __init__(self, tableau):
        pass

    def select_primal_pivot_column(self, tableau):
        smallest = None
        for j, cost in _candidates(tableau, range(tableau.start_index(), tableau.nr_columns())):
            if cost < 0 and (smallest is None or cost < smallest[1]):
                smallest = (j, cost)
        return smallest

    def after_basis_update(self, info, tableau):
        pass


class SteepestDescentAlongObjective:
    """pivot_rule.rs:190-305 (Goldfarb-Reid steepest edge); the reference's hard-wired default."""

    def __init__(self, tableau, check=False):
        """pivot_rule.rs:202-219: ``gamma_j = 1 + ||B^-1 a_j||^2`` for every non-basic non-artificial j."""
        self.check = check
        self.gamma = [
            initial_gamma(j, tableau) if j >= tableau.start_index() and not tableau.is_in_basis(j) else None
            for j in range(tableau.nr_columns())
        ]

    def select_primal_pivot_column(self, tableau):
        """pivot_rule.rs:221-241: max of ``cost^2 / gamma`` over negative costs; *last* maximum on ties."""
        best = None
        best_key = None
        for j, cost in _candidates(tableau, range(tableau.start_index(), tableau.nr_columns())):
            if cost < 0:
                key = cost * cost / self.gamma[j]
                if best_key is None or key >= best_key:  # Iterator::max_by_key keeps the last maximum
                    best_key = key
                    best = (j, cost)
        return best

    def after_basis_update(self, info, tableau):
        """pivot_rule.rs:243-296."""
        self.gamma[info.pivot_column_index] = None
        gamma_q = ONE + sum((v * v for _, v in info.column_before_change), Fraction(0))
        for j in range(tableau.start_index(), len(self.gamma)):
            gamma = self.gamma[j]
            if gamma is None:
                continue
            column = tableau.original_column(j)
            alpha_j_bar = sparse_dot(info.basis_inverse_row, column)
            if alpha_j_bar != 0:
                squared = alpha_j_bar * alpha_j_bar
                inner = sparse_dot(info.work_vector, column)
                if inner != 0:
                    gamma -= 2 * alpha_j_bar * inner
                gamma += squared * gamma_q
                alternative = ONE + squared
            else:
                alternative = ONE
            if gamma < alternative:
                gamma = alternative
            self.gamma[j] = gamma
            if self.check:  # pivot_rule.rs:290 (debug builds only)
                assert gamma == initial_gamma(j, tableau), j
        pos = sorted_get(info.column_before_change, info.pivot_row_index)
        w_p = info.column_before_change[pos][1]
        self.gamma[info.leaving_column_index] = gamma_q / (w_p * w_p)


def initial_gamma(j, tableau):
    """pivot_rule.rs:299-305."""
    column = tableau.generate_column(j).into_column()
    return ONE + sum((v * v for _, v in column), Fraction(0))
