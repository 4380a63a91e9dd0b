"""Matrix providers (oracle; test infrastructure only).

``MatrixData`` follows ``matrix_provider/matrix_data.rs``; ``RemoveRows`` follows
``matrix_provider/filter/generic_wrapper.rs``.  A column is a list of ``(row, Fraction)`` in
iteration order (constraint values, then the optional bound-row entry; matrix_data.rs:563-603).
"""
from fractions import Fraction

ZERO = Fraction(0)
ONE = Fraction(1)


class Variable:
    """data/linear_program/general_form/mod.rs (struct Variable): cost, bounds, shift, flipped."""

    def __init__(self, cost, lower_bound=ZERO, upper_bound=None, shift=ZERO, flipped=False,
                 variable_type="Continuous"):
        self.cost = Fraction(cost)
        self.lower_bound = None if lower_bound is None else Fraction(lower_bound)
        self.upper_bound = None if upper_bound is None else Fraction(upper_bound)
        self.shift = Fraction(shift)
        self.flipped = flipped
        self.variable_type = variable_type


class MatrixData:
    """matrix_data.rs:63-102: constraint matrix plus virtual slack columns in 6 column / 6 row groups."""

    def __init__(self, constraints, b, ranges, nr_equality, nr_range, nr_upper, nr_lower, variables):
        """matrix_data.rs:172-248.  ``constraints``: column-major list of sorted ``(row, value)`` lists."""
        self.constraints = [[(i, Fraction(v)) for i, v in col] for col in constraints]
        self.b = [Fraction(v) for v in b]
        self.ranges = [Fraction(r) for r in ranges]
        assert len(self.ranges) == nr_range
        self.nr_equality, self.nr_range, self.nr_upper, self.nr_lower = nr_equality, nr_range, nr_upper, nr_lower
        self.variables = list(variables)

        self.bound_to_variable = []
        self.variable_to_bound = []
        for j, variable in enumerate(self.variables):
            if variable.upper_bound is not None:
                self.variable_to_bound.append(len(self.bound_to_variable))
                self.bound_to_variable.append(j)
            else:
                self.variable_to_bound.append(None)
        nr_bounds = len(self.bound_to_variable)

        def cumsum(values):
            out, total = [], 0
            for v in values:
                total += v
                out.append(total)
            return out
        # RowType: Equality, Range, UpperIneq, LowerIneq, VariableBound, SlackBound (matrix_data.rs:104-112)
        self.row_end = cumsum([nr_equality, nr_range, nr_upper, nr_lower, nr_bounds, nr_range])
        assert self.row_end[3] == len(self.b)
        # ColumnType: Normal, RangeSlack, UpperIneqSlack, LowerIneqSlack, VariableBoundSlack, SlackBoundSlack
        self.col_end = cumsum([len(self.variables), nr_range, nr_upper, nr_lower, nr_bounds, nr_range])

    def column_type(self, j):
        """matrix_data.rs:253-273: ``(group, index in group)``."""
        previous = 0
        for group, end in enumerate(self.col_end):
            if j < end:
                return group, j - previous
            previous = end
        raise IndexError(j)

    def column(self, j):
        """matrix_data.rs:291-329."""
        group, k = self.column_type(j)
        if group == 0:
            column = list(self.constraints[k])
            bound = self.bound_row_index(k)
            if bound is not None:
                column.append((bound, ONE))
            return column
        if group == 1:
            return [(self.row_end[0] + k, ONE), (self.row_end[4] + k, ONE)]
        if group == 2:
            return [(self.row_end[1] + k, ONE)]
        if group == 3:
            return [(self.row_end[2] + k, -ONE)]
        if group == 4:
            return [(self.row_end[3] + k, ONE)]
        return [(self.row_end[4] + k, ONE)]

    def cost_value(self, j):
        """matrix_data.rs:331-339: cost only on Normal columns (``None`` acts as zero)."""
        group, k = self.column_type(j)
        return self.variables[k].cost if group == 0 else ZERO

    def right_hand_side(self):
        """matrix_data.rs:341-353: ``b || upper bounds || ranges``."""
        return (list(self.b) + [self.variables[j].upper_bound for j in self.bound_to_variable]
                + list(self.ranges))

    def bound_row_index(self, j):
        """matrix_data.rs:355-378 (upper direction only)."""
        group, k = self.column_type(j)
        if group == 0:
            index = self.variable_to_bound[k]
            return None if index is None else self.row_end[3] + index
        if group == 1:
            return self.row_end[4] + k
        return None

    def nr_constraints(self):
        return self.row_end[3]

    def nr_variable_bounds(self):
        return len(self.bound_to_variable) + self.nr_range

    def nr_rows(self):
        return self.nr_constraints() + self.nr_variable_bounds()

    def nr_columns(self):
        return self.col_end[5]

    def nr_normal_variables(self):
        return len(self.constraints)

    def reconstruct_solution(self, column_values):
        """matrix_data.rs:402-411: drop the slack entries."""
        n = self.nr_normal_variables()
        return [(j, v) for j, v in column_values if j < n]

    def pivot_element_indices(self):
        """matrix_data.rs:419-445: free slack pivots ``(row, column)`` sorted by row."""
        upper = [(self.row_end[1] + j, self.col_end[1] + j) for j in range(self.nr_upper)]
        variable = [(self.row_end[3] + j, self.col_end[3] + j) for j in range(len(self.bound_to_variable))]
        slack = [(self.row_end[4] + j, self.col_end[4] + j) for j in range(self.nr_range)]
        return upper + variable + slack

    def nr_initial_elements(self):
        return self.nr_upper + self.nr_variable_bounds()


class RemoveRows:
    """filter/generic_wrapper.rs:27-205: a provider view with some constraint rows removed."""

    def __init__(self, provider, rows_to_skip):
        assert list(rows_to_skip) == sorted(set(rows_to_skip))
        self.provider = provider
        self.rows_to_skip = list(rows_to_skip)
        skip = set(self.rows_to_skip)
        self.relabel = {}
        new = 0
        for i in range(provider.nr_rows()):
            if i not in skip:
                self.relabel[i] = new
                new += 1

    def filtered_rows(self):
        return self.rows_to_skip

    def column(self, j):  # generic_wrapper.rs:145-149 with IntoFilteredColumn
        return [(self.relabel[i], v) for i, v in self.provider.column(j) if i in self.relabel]

    def cost_value(self, j):
        return self.provider.cost_value(j)

    def right_hand_side(self):  # generic_wrapper.rs:157-161
        skip = set(self.rows_to_skip)
        return [v for i, v in enumerate(self.provider.right_hand_side()) if i not in skip]

    def nr_constraints(self):
        return self.provider.nr_constraints() - len(self.rows_to_skip)

    def nr_variable_bounds(self):
        return self.provider.nr_variable_bounds()

    def nr_rows(self):
        return self.provider.nr_rows() - len(self.rows_to_skip)

    def nr_columns(self):
        return self.provider.nr_columns()

    def reconstruct_solution(self, column_values):
        return self.provider.reconstruct_solution(column_values)
