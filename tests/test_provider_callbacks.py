"""``relp_model_from_provider``: any ``MatrixProvider`` through the calls the reference's loops make on it (column, cost_value,
right_hand_side, pivot_element_indices).  The providers here are the oracle's restatements of the reference's example
providers (examples/max_flow.rs, examples/shortest_path.rs) and plain ``MatrixData`` objects; the host must hand back exactly
their columns, and the GPU must reach the oracle's exact optimum on them."""
import random
from fractions import Fraction

import pytest

import relp_amd
from relp_oracle import FiniteOptimum, Infeasible, MatrixData, Unbounded, Variable, solve_relaxation
from relp_oracle.network import MaxFlowPrimal, ShortestPathPrimal


def graph(rng, nr_vertices, nr_arcs):
    arcs = {}
    while len(arcs) < nr_arcs:
        a, b = rng.randrange(nr_vertices), rng.randrange(nr_vertices)
        if a != b and b != 0 and a != nr_vertices - 1:
            arcs[(a, b)] = Fraction(rng.randint(1, 9))
    return [sorted((b, v) for (a, b), v in arcs.items() if a == frm) for frm in range(nr_vertices)]  # arcs[from] = [(to, value)]


def providers(seed):
    rng = random.Random(4200 + seed)
    nr_vertices = rng.randint(4, 9)
    arcs = graph(rng, nr_vertices, rng.randint(nr_vertices, 2 * nr_vertices))
    yield MaxFlowPrimal(arcs, 0, nr_vertices - 1)
    yield ShortestPathPrimal(arcs, 0, nr_vertices - 1)
    n, m = rng.randint(2, 6), rng.randint(2, 5)
    dense = [[rng.choice([0, 1, 2, -1, 3]) for _ in range(n)] for _ in range(m)]
    columns = [[(i, dense[i][j]) for i in range(m) if dense[i][j] != 0] for j in range(n)]
    yield MatrixData(columns, [rng.randint(0, 9) for _ in range(m)], [], 1, 0, m - 1, 0,
                     [Variable(rng.randint(-4, 4), upper_bound=rng.choice([None, 5])) for _ in range(n)])


@pytest.mark.parametrize("seed", range(12))
def test_host_model_holds_the_providers_columns(seed):
    for provider in providers(seed):
        model = relp_amd.Model.from_provider(provider)
        assert (model.nr_rows, model.nr_columns) == (provider.nr_rows(), provider.nr_columns())
        for j in range(provider.nr_columns()):
            assert [(i, Fraction(n, d)) for i, n, d in model.column_exact(j)] == [(i, Fraction(v)) for i, v in provider.column(j)]
            assert model.cost_value(j) == float(provider.cost_value(j))
        assert list(model.right_hand_side()) == [float(v) for v in provider.right_hand_side()]
        expected = sorted(provider.pivot_element_indices()) if hasattr(provider, "pivot_element_indices") else []
        assert model.pivot_element_indices() == expected


def test_argument_errors():
    class Bad:
        def __init__(self, column, pivots=None):
            self._column, self._pivots = column, pivots
        def nr_rows(self): return 2
        def nr_columns(self): return 2
        def column(self, j): return self._column
        def cost_value(self, j): return 1
        def right_hand_side(self): return [1, 1]
    relp_amd.Model.from_provider(Bad([(0, 1)]))
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.from_provider(Bad([(2, 1)]))          # row out of range
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.from_provider(Bad([(1, 1), (0, 1)]))  # rows must ascend
    bad = Bad([(0, 2)])
    bad.pivot_element_indices = lambda: [(0, 0)]
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.from_provider(bad)                    # a pivot column must be a unit vector


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_solves_providers_like_the_oracle(seed):
    for provider in providers(seed):
        try:
            expected = solve_relaxation(provider)
        except AssertionError:
            continue  # the reference's LU cannot factor a 1 x 1 basis
        solver = relp_amd.Solver(certify=1).load_model(relp_amd.Model.from_provider(provider))
        result = solver.solve_relaxation()
        if isinstance(expected, Infeasible):
            assert result.kind == relp_amd.INFEASIBLE
        elif isinstance(expected, Unbounded):
            assert result.kind == relp_amd.UNBOUNDED
        else:
            assert isinstance(expected, FiniteOptimum) and result.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1
            objective = sum((Fraction(provider.cost_value(j)) * v for j, v in expected.solution), Fraction(0))
            assert Fraction(solver.objective_exact()) == objective
        solver.close()
