"""Graph providers (oracle; test infrastructure only).

``IncidenceMatrix`` follows ``data/linear_program/network/representation.rs:24-100``; ``MaxFlowPrimal`` and
``ShortestPathPrimal`` follow the ``MatrixProvider`` implementations of ``examples/max_flow.rs:31-223`` and
``examples/shortest_path.rs:20-118``.  A graph is given like the reference's adjacency matrix: ``arcs[from]`` is the
list of ``(to, value)`` leaving ``from``, sorted by ``to`` (``ColumnMajor`` data: column = tail, row = head).
"""
from bisect import bisect_left
from fractions import Fraction

ZERO = Fraction(0)
ONE = Fraction(1)


def adjacency_from_rows(rows):
    """``ColumnMajor::from_test_data`` as the examples use it: ``rows[to][from]`` = value (0 = no arc)."""
    n = len(rows)
    return [[(to, Fraction(rows[to][frm])) for to in range(n) if rows[to][frm] != 0] for frm in range(n)]


class IncidenceMatrix:
    """representation.rs:24-100: one column per arc over the vertices that were not removed."""

    def __init__(self, arcs, removed):
        nr_vertices = len(arcs)
        removed = sorted(removed)
        self.removed = removed
        self.columns = []
        self.values = []
        self.tails = []
        for frm, outgoing in enumerate(arcs):
            for to, value in outgoing:
                assert to != frm, "no self-arcs (representation.rs:38)"
                from_shift = bisect_left(removed, frm)
                to_shift = bisect_left(removed, to)
                from_deleted = from_shift < len(removed) and removed[from_shift] == frm
                to_deleted = to_shift < len(removed) and removed[to_shift] == to
                if from_deleted and to_deleted:
                    column = []
                elif from_deleted:
                    column = [(to - to_shift, ONE)]                 # ArcDirection::Incoming = +1
                elif to_deleted:
                    column = [(frm - from_shift, -ONE)]             # ArcDirection::Outgoing = -1
                else:
                    column = sorted([(frm - from_shift, -ONE), (to - to_shift, ONE)])
                self.columns.append(column)
                self.values.append(Fraction(value))
                self.tails.append(frm)
        self.nr_rows = nr_vertices - len(removed)
        self._nr_vertices = nr_vertices

    def column(self, j):
        return list(self.columns[j])

    def nr_vertices(self):
        return self._nr_vertices

    def nr_edges(self):
        return len(self.columns)


class MaxFlowPrimal:
    """examples/max_flow.rs:31-223: maximise the flow out of ``s``; one bound row + slack per arc."""

    def __init__(self, arcs, s, t):
        self.s, self.t = s, t
        self.incidence = IncidenceMatrix(arcs, [s, t])          # max_flow.rs:66
        self.capacity = self.incidence.values
        before = sum(len(arcs[v]) for v in range(s))            # max_flow.rs:63-65
        self.s_arc_range = range(before, before + len(arcs[s]))

    def nr_vertices(self):
        return self.incidence.nr_vertices()

    def nr_edges(self):
        return self.incidence.nr_edges()

    def column(self, j):                                       # max_flow.rs:148-162
        if j < self.nr_edges():
            return self.incidence.column(j) + [(self.nr_constraints() + j, ONE)]
        return [(self.nr_constraints() + j - self.nr_edges(), ONE)]

    def cost_value(self, j):                                   # max_flow.rs:164-172
        return -ONE if j in self.s_arc_range else ZERO

    def right_hand_side(self):                                 # max_flow.rs:174-178
        return [ZERO] * self.nr_constraints() + list(self.capacity)

    def bound_row_index(self, j):                              # max_flow.rs:180-191 (upper direction)
        return self.nr_constraints() + j if j < self.nr_edges() else None

    def nr_constraints(self):                                  # max_flow.rs:193-195
        return self.nr_vertices() - 2

    def nr_variable_bounds(self):
        return self.nr_edges()

    def nr_rows(self):
        return self.nr_constraints() + self.nr_variable_bounds()

    def nr_columns(self):                                      # max_flow.rs:201-204
        return 2 * self.nr_edges()

    def pivot_element_indices(self):                           # max_flow.rs:215-218
        return [(j + self.nr_constraints(), self.nr_edges() + j) for j in range(self.nr_edges())]

    def reconstruct_solution(self, column_values):
        return list(column_values)


class ShortestPathPrimal:
    """examples/shortest_path.rs:20-118: unit flow from ``s`` to ``t`` at minimum length; the row of ``s`` is dropped."""

    def __init__(self, arcs, s, t):
        self.s, self.t = s, t
        self.incidence = IncidenceMatrix(arcs, [s])             # shortest_path.rs:44-46
        self.cost = self.incidence.values

    def nr_vertices(self):
        return self.incidence.nr_vertices()

    def nr_edges(self):
        return self.incidence.nr_edges()

    def column(self, j):                                       # shortest_path.rs:75-79
        return self.incidence.column(j)

    def cost_value(self, j):                                   # shortest_path.rs:81-85
        return self.cost[j]

    def right_hand_side(self):                                 # shortest_path.rs:87-93
        b = [ZERO] * self.nr_rows()
        b[self.t if self.t < self.s else self.t - 1] = ONE
        return b

    def nr_constraints(self):                                  # shortest_path.rs:102-105
        return self.nr_vertices() - 1

    def nr_variable_bounds(self):
        return 0

    def nr_rows(self):
        return self.nr_constraints()

    def nr_columns(self):
        return self.nr_edges()

    def reconstruct_solution(self, column_values):
        return list(column_values)
