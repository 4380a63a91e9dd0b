"""Problem dump for the C++ CPU oracle ``oracle/cpp/relp_cpu.cpp`` (oracle; test infrastructure only).

Writes what a ``MatrixProvider`` hands the simplex (``matrix_provider/mod.rs:37-134``): every column as the provider
yields it (``column(j)``), the costs, the right-hand side and the initial pivots, all exact (``num/den``).
"""
from fractions import Fraction


def _text(value):
    value = Fraction(value)
    return "%d/%d" % (value.numerator, value.denominator)


def dump_provider(provider, path, route=None):
    """``route``: ``partial`` (``PartialInitialBasis``, phase_one.rs:66-80), ``fully`` or ``full_basis``
    (``FullInitialBasis``, two_phase/mod.rs:80-109); default: ``partial`` when the provider has pivots."""
    has_pivots = hasattr(provider, "pivot_element_indices")
    if route is None:
        route = "partial" if has_pivots else "fully"
    pivots = provider.pivot_element_indices() if has_pivots and route != "fully" else []
    m, n = provider.nr_rows(), provider.nr_columns()
    with open(path, "w") as out:
        out.write("relp-problem 1\nroute %s\nm %d\nn %d\npivots %d\n" % (route, m, n, len(pivots)))
        for row, column in pivots:
            out.write("%d %d\n" % (row, column))
        out.write("rhs\n")
        out.write("\n".join(_text(v) for v in provider.right_hand_side()))
        out.write("\ncost\n")
        out.write("\n".join(_text(provider.cost_value(j)) for j in range(n)))
        out.write("\ncolumns\n")
        for j in range(n):
            column = sorted(provider.column(j), key=lambda t: t[0])
            out.write("%d %s\n" % (len(column), " ".join("%d %s" % (i, _text(v)) for i, v in column)))
