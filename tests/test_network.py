"""Graph providers: the `MatrixProvider`s of the reference's examples/max_flow.rs and
examples/shortest_path.rs.  CPU part: the oracle reproduces the optima the examples assert, the C++ host model equals the
oracle's provider column for column, the C++ oracle agrees.  GPU part: the device path solves them (exact certificate)."""
import os
import random
import sys
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import relp_amd  # noqa: E402
from relp_oracle import FiniteOptimum, solve_relaxation  # noqa: E402
from relp_oracle import cpu  # noqa: E402
from relp_oracle.inverse_rows import BasisInverseRows  # noqa: E402
from relp_oracle.lu import LUDecomposition  # noqa: E402
from relp_oracle.network import MaxFlowPrimal, ShortestPathPrimal, adjacency_from_rows  # noqa: E402

# examples/max_flow.rs:264-272 and examples/shortest_path.rs:156-163 (Papadimitriou's example; rows[to][from])
MAX_FLOW_ROWS = [[0, 0, 0, 0], [2, 0, 0, 0], [1, 1, 0, 0], [0, 1, 2, 0]]
SHORTEST_PATH_ROWS = [[0, 0, 0, 0], [1, 0, 0, 0], [2, 2, 0, 0], [0, 3, 1, 0]]


def _dense(solution, n):
    out = [Fraction(0)] * n
    for j, v in solution:
        out[j] = v
    return out


@pytest.mark.parametrize("bi_cls", [BasisInverseRows, LUDecomposition])
def test_oracle_reproduces_the_examples(bi_cls):
    problem = MaxFlowPrimal(adjacency_from_rows(MAX_FLOW_ROWS), 0, 3)
    result = solve_relaxation(problem, bi_cls=bi_cls)
    assert _dense(result.solution, 10) == [2, 1, 1, 1, 2, 0, 0, 0, 0, 0]       # examples/max_flow.rs:279-282
    problem = ShortestPathPrimal(adjacency_from_rows(SHORTEST_PATH_ROWS), 0, 3)
    result = solve_relaxation(problem, bi_cls=bi_cls)
    assert _dense(result.solution, 5) == [0, 1, 0, 0, 1]                       # examples/shortest_path.rs:165-168


def random_graph(rng, nr_vertices, nr_arcs, max_value):
    pairs = set()
    while len(pairs) < nr_arcs:
        a, b = rng.randrange(nr_vertices), rng.randrange(nr_vertices)
        if a != b:
            pairs.add((a, b))
    arcs = [[] for _ in range(nr_vertices)]
    for a, b in sorted(pairs):
        arcs[a].append((b, Fraction(rng.randint(1, max_value))))
    return arcs


def arc_list(arcs):
    return [(a, b, v) for a, outgoing in enumerate(arcs) for b, v in outgoing]


def _same_provider(model, provider):
    assert (model.nr_rows, model.nr_columns, model.nr_constraints) == (
        provider.nr_rows(), provider.nr_columns(), provider.nr_constraints())
    for j in range(provider.nr_columns()):
        expected = sorted((i, v.numerator, v.denominator) for i, v in provider.column(j))
        assert model.column_exact(j) == expected, j
        assert model.cost_value(j) == float(provider.cost_value(j))
    assert list(model.right_hand_side()) == [float(v) for v in provider.right_hand_side()]


@pytest.mark.parametrize("seed", range(4))
def test_host_model_equals_oracle_provider(seed):
    rng = random.Random(seed)
    nr_vertices = rng.randint(4, 9)
    arcs = random_graph(rng, nr_vertices, rng.randint(nr_vertices, 2 * nr_vertices), 9)
    s, t = rng.sample(range(nr_vertices), 2)
    flow = MaxFlowPrimal(arcs, s, t)
    model = relp_amd.Model.max_flow(nr_vertices, arc_list(arcs), s, t)
    _same_provider(model, flow)
    assert model.pivot_element_indices() == flow.pivot_element_indices()
    path = ShortestPathPrimal(arcs, s, t)
    model = relp_amd.Model.shortest_path(nr_vertices, arc_list(arcs), s, t)
    _same_provider(model, path)
    assert model.pivot_element_indices() == []


def test_graph_model_argument_errors():
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.max_flow(3, [(0, 0, 1)], 0, 2)            # self arc
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.max_flow(3, [(0, 1, 1), (0, 1, 2)], 0, 2)  # duplicate arc
    with pytest.raises(relp_amd.RelpError):
        relp_amd.Model.shortest_path(3, [(0, 1, 1)], 1, 1)        # s == t


def test_cpp_oracle_agrees_on_graph_providers():
    rng = random.Random(11)
    arcs = random_graph(rng, 7, 16, 9)
    for problem in (MaxFlowPrimal(arcs, 0, 6), ShortestPathPrimal(arcs, 0, 6)):
        exact = solve_relaxation(problem)
        record = cpu.solve_provider(problem)
        if isinstance(exact, FiniteOptimum):
            assert record["status"] == "optimal" and record["solution"] == exact.solution
        else:
            assert record["status"] == repr(exact).lower()


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_solves_the_examples_exactly():
    model = relp_amd.Model.max_flow(4, arc_list(adjacency_from_rows(MAX_FLOW_ROWS)), 0, 3)
    solver = relp_amd.Solver(certify=1).load_model(model)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
    assert Fraction(solver.objective_exact()) == -3
    assert np.allclose(solver.solution(), [2, 1, 1, 1, 2], atol=1e-9)          # the vertex examples/max_flow.rs asserts
    model = relp_amd.Model.shortest_path(4, arc_list(adjacency_from_rows(SHORTEST_PATH_ROWS)), 0, 3)
    solver = relp_amd.Solver(certify=1).load_model(model)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
    assert Fraction(solver.objective_exact()) == 3
    assert np.allclose(solver.solution(), [0, 1, 0, 0, 1], atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_gpu_random_graphs_match_oracle_exactly(seed):
    rng = random.Random(100 + seed)
    nr_vertices = rng.randint(5, 12)
    arcs = random_graph(rng, nr_vertices, rng.randint(nr_vertices + 2, 3 * nr_vertices), 20)
    s, t = 0, nr_vertices - 1
    for provider, model in ((MaxFlowPrimal(arcs, s, t), relp_amd.Model.max_flow(nr_vertices, arc_list(arcs), s, t)),
                            (ShortestPathPrimal(arcs, s, t), relp_amd.Model.shortest_path(nr_vertices, arc_list(arcs), s, t))):
        exact = solve_relaxation(provider)
        solver = relp_amd.Solver(certify=1).load_model(model)
        result = solver.solve_relaxation()
        if isinstance(exact, FiniteOptimum):
            assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
            assert Fraction(solver.objective_exact()) == exact.objective
        else:
            assert result.kind == relp_amd.INFEASIBLE      # no s-t path: the shortest-path LP is infeasible


@pytest.mark.gpu
def test_gpu_max_flow_matches_scipy_at_scale():
    """V = 256, E ~ 2030 (m ~ 2290 rows): the LP optimum equals the max-flow value of scipy's Dinic."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    from relp_amd.workloads import max_flow_graph
    nr_vertices, nr_arcs = 256, 2048
    tail, head, capacity = max_flow_graph(nr_vertices, nr_arcs)
    # the reference's objective is the GROSS flow on the arcs leaving s (examples/max_flow.rs:164-172) and the rows of s
    # and t are removed, so it equals the max-flow value only when nothing enters s or leaves t: drop those arcs here
    keep = (head != 0) & (tail != nr_vertices - 1)
    tail, head, capacity = tail[keep], head[keep], capacity[keep]
    graph = csr_matrix((capacity.astype(np.int32), (tail, head)), shape=(nr_vertices, nr_vertices))
    expected = maximum_flow(graph, 0, nr_vertices - 1).flow_value
    model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
    solver = relp_amd.Solver(certify=1).load_model(model)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
    assert Fraction(solver.objective_exact()) == -expected


@pytest.mark.gpu
def test_gpu_max_flow_m_9000():
    """V = 1024, E ~ 8170 (m ~ 9190 rows: the general ratio-test kernel, unit-column skipping in the inverse update):
    f64 optimum equals scipy's max-flow value."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    from relp_amd.workloads import max_flow_graph
    nr_vertices = 1024
    tail, head, capacity = max_flow_graph(nr_vertices, 8192)
    keep = (head != 0) & (tail != nr_vertices - 1)
    tail, head, capacity = tail[keep], head[keep], capacity[keep]
    graph = csr_matrix((capacity.astype(np.int32), (tail, head)), shape=(nr_vertices, nr_vertices))
    expected = maximum_flow(graph, 0, nr_vertices - 1).flow_value
    model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
    solver = relp_amd.Solver().load_model(model)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective + expected) <= 1e-9 * max(1.0, abs(expected))
    flow = solver.solution()
    assert np.all(flow >= -1e-9) and np.all(flow <= capacity + 1e-9)


@pytest.mark.gpu
def test_gpu_max_flow_131k_arcs_with_implicit_bounds():
    """V = 16 384, E ~ 131 000 (1/8 of BASELINE config 5): with the capacity rows handled as implicit bounds the device LP has
    V - 2 rows instead of V - 2 + E (147 436); the optimum equals scipy's max-flow value and every flow respects its capacity."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    from relp_amd.workloads import max_flow_graph
    nr_vertices = 16384
    tail, head, capacity = max_flow_graph(nr_vertices, 131072)
    keep = (head != 0) & (tail != nr_vertices - 1)
    tail, head, capacity = tail[keep], head[keep], capacity[keep]
    graph = csr_matrix((capacity.astype(np.int32), (tail, head)), shape=(nr_vertices, nr_vertices))
    expected = maximum_flow(graph, 0, nr_vertices - 1).flow_value
    model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
    solver = relp_amd.Solver(implicit_bounds=1).load_model(model)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective + expected) <= 1e-9 * max(1.0, abs(expected))
    flow = solver.solution()
    assert len(flow) == len(tail)
    assert np.all(flow >= -1e-9) and np.all(flow <= capacity + 1e-9)
    # conservation at every inner vertex
    net = np.zeros(nr_vertices)
    np.add.at(net, head, flow)
    np.subtract.at(net, tail, flow)
    assert np.max(np.abs(net[1:-1])) <= 1e-7
    assert abs(-net[0] - expected) <= 1e-7 and abs(net[-1] - expected) <= 1e-7


@pytest.mark.gpu
def test_gpu_generated_column_pricing_variants_make_the_same_pivots(monkeypatch):
    """The pricing pass over generated incidence columns (`price_unit_kernel`: a lane per arc, rho_p's non-zero rows as a bit
    table in LDS) against its two predecessors -- the byte table gathered from memory (RELP_NO_RHO_BITS) and the two-lanes-per-arc
    kernel (RELP_PRICE_UNIT_PAIRS): every sum is formed in the same order and the candidates have a total order, so the three
    make the same pivots and end on the same numbers, bit for bit (pivot_rule.rs:221-296 is one sequential pass)."""
    from relp_amd.workloads import max_flow_graph
    nr_vertices = 16384
    tail, head, capacity = max_flow_graph(nr_vertices, 131072)
    keep = (head != 0) & (tail != nr_vertices - 1)
    tail, head, capacity = tail[keep], head[keep], capacity[keep]
    model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
    outcomes = []
    for switch in (None, "RELP_NO_RHO_BITS", "RELP_PRICE_UNIT_PAIRS"):
        if switch:
            monkeypatch.setenv(switch, "1")
        solver = relp_amd.Solver(implicit_bounds=1).load_model(model)
        result = solver.solve_relaxation()
        outcomes.append((result.kind, result.pivots_phase_one, result.pivots_phase_two, result.objective, solver.solution().tobytes()))
        solver.close()
        if switch:
            monkeypatch.delenv(switch)
    assert outcomes[0][0] == relp_amd.FINITE_OPTIMUM
    assert outcomes[0][1] + outcomes[0][2] > 1000
    assert outcomes[1] == outcomes[0]
    assert outcomes[2] == outcomes[0]


@pytest.mark.gpu
def test_gpu_generated_columns_under_the_lu_carry():
    """Generated incidence columns priced by `price_unit_kernel` while the LU + Forrest-Tomlin carry maintains the basis: between
    6 800 rows (where the width-2 padded copy and the generated columns start) and 8 000 (the LU carry's LDS limit) it is
    `lu_pivot_kernel` that writes rho_p and marks its non-zero rows in the bit table the pricing pass reads.  Same optimum as the
    explicit carry and as scipy's max-flow."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    from relp_amd.workloads import max_flow_graph
    nr_vertices = 7400
    tail, head, capacity = max_flow_graph(nr_vertices, 40000)
    keep = (head != 0) & (tail != nr_vertices - 1)
    tail, head, capacity = tail[keep], head[keep], capacity[keep]
    expected = maximum_flow(csr_matrix((capacity.astype(np.int32), (tail, head)), shape=(nr_vertices, nr_vertices)), 0, nr_vertices - 1).flow_value
    model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
    objectives = []
    for carry in (0, 1):
        solver = relp_amd.Solver(implicit_bounds=1, carry=carry).load_model(model)
        result = solver.solve_relaxation()
        assert result.kind == relp_amd.FINITE_OPTIMUM, (carry, result.kind)
        objectives.append(result.objective)
        flow = solver.solution()
        assert np.all(flow >= -1e-9) and np.all(flow <= capacity + 1e-9)
        solver.close()
    assert abs(objectives[0] + expected) <= 1e-9 * max(1.0, abs(expected))
    assert abs(objectives[1] + expected) <= 1e-7 * max(1.0, abs(expected))


@pytest.mark.gpu
def test_gpu_shortest_path_12k_vertices_matches_dijkstra():
    """V = 12 000, E ~ 60 000 (one conservation row per vertex but the target, no bounds: the ratio test across workgroups
    without the bounded-variable rules): the LP optimum equals scipy's Dijkstra distance and the solution is a unit s-t flow."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import dijkstra
    from relp_amd.workloads import max_flow_graph
    nr_vertices = 12000
    tail, head, weight = max_flow_graph(nr_vertices, 60000)
    keep = tail != head
    tail, head, weight = tail[keep], head[keep], weight[keep]
    # parallel arcs: scipy's csr constructor would add their weights, the LP takes the cheapest
    order = np.lexsort((weight, head, tail))
    tail, head, weight = tail[order], head[order], weight[order]
    first = np.ones(len(tail), dtype=bool)
    first[1:] = (tail[1:] != tail[:-1]) | (head[1:] != head[:-1])
    tail, head, weight = tail[first], head[first], weight[first]
    graph = csr_matrix((weight.astype(np.float64), (tail, head)), shape=(nr_vertices, nr_vertices))
    expected = dijkstra(graph, directed=True, indices=0)[nr_vertices - 1]
    assert np.isfinite(expected)
    model = relp_amd.Model.shortest_path(nr_vertices, list(zip(tail.tolist(), head.tolist(), weight.tolist())), 0, nr_vertices - 1)
    solver = relp_amd.Solver().load_model(model)
    assert solver.m > 8192
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective - expected) <= 1e-9 * max(1.0, abs(expected))
    flow = solver.solution()
    net = np.zeros(nr_vertices)
    np.add.at(net, head, flow)
    np.subtract.at(net, tail, flow)
    assert abs(net[0] + 1.0) <= 1e-7 and abs(net[-1] - 1.0) <= 1e-7 and np.max(np.abs(net[1:-1])) <= 1e-7
    solver.close()


@pytest.mark.gpu
@pytest.mark.parametrize("implicit", [0, 1], ids=["bound-rows", "implicit-bounds"])
@pytest.mark.parametrize("seed", range(6))
def test_gpu_crash_basis_same_exact_optimum_on_random_graphs(seed, implicit):
    """`relp_options.crash`: the spanning-forest start must not change what is reported -- the certified exact optimum of
    the oracle -- and on a max-flow LP (b = 0 on every conservation row) it leaves nothing to do for phase one."""
    rng = random.Random(900 + seed)
    nr_vertices = rng.randint(6, 14)
    arcs = random_graph(rng, nr_vertices, rng.randint(nr_vertices + 4, 3 * nr_vertices), 20)
    s, t = 0, nr_vertices - 1
    for provider, model, is_flow in ((MaxFlowPrimal(arcs, s, t), relp_amd.Model.max_flow(nr_vertices, arc_list(arcs), s, t), True),
                                     (ShortestPathPrimal(arcs, s, t), relp_amd.Model.shortest_path(nr_vertices, arc_list(arcs), s, t), False)):
        exact = solve_relaxation(provider)
        solver = relp_amd.Solver(certify=1, crash=1, implicit_bounds=implicit).load_model(model)
        plain = relp_amd.Solver(certify=1, implicit_bounds=implicit).load_model(model)
        result, reference = solver.solve_relaxation(), plain.solve_relaxation()
        assert result.kind == reference.kind
        if isinstance(exact, FiniteOptimum):
            assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
            assert Fraction(solver.objective_exact()) == exact.objective
            if is_flow:
                assert result.pivots_phase_one <= reference.pivots_phase_one
        else:
            assert result.kind == relp_amd.INFEASIBLE
        solver.close()
        plain.close()


@pytest.mark.gpu
def test_gpu_crash_basis_and_generated_columns_131k_arcs():
    """V = 16 384, E ~ 131 000 with the crash basis: every conservation row is covered by a tree arc, phase one makes no
    pivot, the pricing pass generates the incidence columns from the arcs' endpoints, and the optimum is scipy's."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    from relp_amd.workloads import max_flow_graph
    nr_vertices = 16384
    tail, head, capacity = max_flow_graph(nr_vertices, 131072)
    keep = (head != 0) & (tail != nr_vertices - 1)
    tail, head, capacity = tail[keep], head[keep], capacity[keep]
    graph = csr_matrix((capacity.astype(np.int32), (tail, head)), shape=(nr_vertices, nr_vertices))
    expected = maximum_flow(graph, 0, nr_vertices - 1).flow_value
    model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
    solver = relp_amd.Solver(implicit_bounds=1, crash=1).load_model(model)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    assert abs(result.objective + expected) <= 1e-9 * max(1.0, abs(expected))
    assert result.pivots_phase_one == 0
    flow = solver.solution()
    assert np.all(flow >= -1e-9) and np.all(flow <= capacity + 1e-9)
    net = np.zeros(nr_vertices)
    np.add.at(net, head, flow)
    np.subtract.at(net, tail, flow)
    assert np.max(np.abs(net[1:-1])) <= 1e-7
    solver.close()


@pytest.mark.gpu
def test_gpu_max_flow_config5_full_size():
    """BASELINE configs[4] at its stated size: V = 65 536, E = 1 048 576 (splitmix64 seed 0x5EED0005).  The capacity rows are
    implicit bounds (65 534 conservation rows on the device); the optimum equals scipy's max-flow value, every flow respects
    its capacity and conservation holds at every inner vertex."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    from relp_amd.workloads import max_flow_graph
    nr_vertices, nr_arcs = 65536, 1048576
    tail, head, capacity = max_flow_graph(nr_vertices, nr_arcs)
    assert len(tail) == nr_arcs
    graph = csr_matrix((capacity.astype(np.int32), (tail, head)), shape=(nr_vertices, nr_vertices))
    expected = maximum_flow(graph, 0, nr_vertices - 1).flow_value
    model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
    solver = relp_amd.Solver(implicit_bounds=1).load_model(model)
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM
    # (the example's objective is the gross flow out of s; it equals the max-flow value when the optimal flow sends nothing
    #  back into s, which holds for this graph: checked through conservation below)
    flow = solver.solution()
    assert len(flow) == nr_arcs
    order = np.lexsort((head, tail))  # the model's arc order: sorted by (tail, head)
    t_sorted, h_sorted, c_sorted = tail[order], head[order], capacity[order]
    assert np.all(flow >= -1e-9) and np.all(flow <= c_sorted + 1e-9)
    net = np.zeros(nr_vertices)
    np.add.at(net, h_sorted, flow)
    np.subtract.at(net, t_sorted, flow)
    assert np.max(np.abs(net[1:-1])) <= 1e-6
    # the example's objective is the GROSS flow on the arcs leaving s (examples/max_flow.rs:141-223); on this graph it equals
    # the max-flow value (arcs into s or out of t carry nothing useful at the optimum)
    assert abs(result.objective + expected) <= 1e-9 * expected
    assert result.solve_seconds < 30
    solver.close()
    # the same LP from the crash basis (spanning forest grown from s and t): no phase-one pivot, same optimum
    solver = relp_amd.Solver(implicit_bounds=1, crash=1).load_model(model)
    crashed = solver.solve_relaxation()
    assert crashed.kind == relp_amd.FINITE_OPTIMUM and crashed.pivots_phase_one == 0
    assert abs(crashed.objective + expected) <= 1e-9 * expected
    flow = solver.solution()
    assert np.all(flow >= -1e-9) and np.all(flow <= c_sorted + 1e-9)
    net = np.zeros(nr_vertices)
    np.add.at(net, h_sorted, flow)
    np.subtract.at(net, t_sorted, flow)
    assert np.max(np.abs(net[1:-1])) <= 1e-6
    assert crashed.solve_seconds < result.solve_seconds
    solver.close()


@pytest.mark.gpu
def test_bench_rccl_path_runs_at_world_size_one():
    """`bench.py` with RELP_FORCE_DISTRIBUTED=1: init_process_group("nccl") (RCCL), the barrier and the all-reduce of time and
    pivot counts run with one rank -- the code path the 8-GPU run takes, started as a child process BEFORE this process's
    GPU state matters to it (no exec from a process that touched the GPU)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RELP_FORCE_DISTRIBUTED="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from bench_support import run_bench
    short, line, _ = run_bench(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-configs", "--no-concurrency-probe"], env=env, timeout=600)
    assert short["n_gpus"] == 1 and short["config"]["certified"] is True
    assert line["n_gpus"] == 1 and line["steps"] == 2
    assert line["config"]["exact"]["certified"] is True
    assert abs(line["config"]["objective"] - 5501.8458883) < 1e-6
    assert line["value"] > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["25fv47", "netlib"])
def test_bench_two_ranks_over_rccl(workload):
    """Two ranks, one per GPU, under `torch.distributed.run` exactly as the driver launches the scaling bench (started as a
    child process; this process has made no GPU call that the child could inherit).  Needs two GPUs: skipped on a 1-GPU box
    (`device_count` does not initialise the GPU on this image)."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from bench_support import run_bench
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    short, line, _ = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--no-cpu-baseline",
                                "--no-dense-roofline", "--no-concurrency-probe"], env=env, launcher=launcher)
    assert short["n_gpus"] == 2 and short["value"] > 0
    assert line["n_gpus"] == 2 and line["steps"] == 2
    config = line["config"]
    assert config["makespan_s"] > 0
    if workload == "netlib":
        assert len(config["tickets_per_rank"]) == 2 and sum(config["tickets_per_rank"]) == 2 * 45
        assert config["objectives_outside_reference_tolerance"] == []
    else:
        assert [r["rank"] for r in config["per_rank"]] == [0, 1]
        assert all(r["solves"] == 2 and r["pivots"] > 4000 for r in config["per_rank"])


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["25fv47", "netlib"])
def test_bench_two_ranks_without_a_launcher(workload):
    """`python bench.py --gpus 2 ...` with nothing around it: bench.py starts its two ranks itself (before it imports torch) and relays
    rank 0's line with n_gpus == 2.  On a 1-GPU box the two ranks share device 0 and talk over gloo (RELP_BENCH_SHARED_DEVICE=1);
    with two GPUs they take one each over RCCL."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if torch.cuda.device_count() < 2:
        env["RELP_BENCH_SHARED_DEVICE"] = "1"
    from bench_support import run_bench
    short, line, _ = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--no-cpu-baseline",
                                "--no-concurrency-probe"], env=env)
    assert short["n_gpus"] == 2 and short["value"] > 0
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["makespan_s"] > 0
    if workload == "netlib":
        assert len(line["config"]["tickets_per_rank"]) == 2 and sum(line["config"]["tickets_per_rank"]) == 2 * 45
    else:
        assert [r["rank"] for r in line["config"]["per_rank"]] == [0, 1]


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_device_print_the_eight_gpu_line():
    """Round-5 review, item 9: when the driver does get an 8-GPU node, `bench.py --gpus 8` must need no code change.  Here the eight ranks of
    the driver's invocation share the one device of the box (RELP_BENCH_SHARED_DEVICE=1, gloo for the reductions): BASELINE config 4 -- the
    Netlib list served from ONE ticket queue in the store -- and the line says n_gpus 8 with the records of all eight ranks."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if torch.cuda.device_count() < 8:
        env["RELP_BENCH_SHARED_DEVICE"] = "1"
    from bench_support import run_bench
    short, line, _ = run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0", "--workload", "netlib", "--no-cpu-baseline", "--no-concurrency-probe"], env=env)
    assert short["n_gpus"] == 8 and line["n_gpus"] == 8 and line["value"] > 0 and line["config"]["makespan_s"] > 0
    tickets = line["config"]["tickets_per_rank"]
    assert len(tickets) == 8 and sum(tickets) == 45 and len(line["config"]["pivots_per_rank"]) == 8
    assert line["config"]["objectives_outside_reference_tolerance"] == []


@pytest.mark.gpu
def test_bench_more_ranks_than_devices_fails_loudly():
    """`--gpus N` on a box with fewer devices: non-zero exit, no line that claims N GPUs."""
    import subprocess
    import sys
    import torch
    wanted = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "RELP_BENCH_SHARED_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(wanted), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                          "--no-configs", "--no-concurrency-probe"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not any(row.startswith("{") and '"n_gpus"' in row for row in out.stdout.splitlines())


@pytest.mark.gpu
@pytest.mark.parametrize("crash", [1, 0], ids=["crash-basis", "reference-start"])
def test_bench_max_flow_workload_line(crash):
    """`bench.py --workload maxflow64k` (the 64 k-arc twin of BASELINE config 5) end to end as a child process: one JSON line with the
    roofline of every kernel of the pivot, with and without the crash basis (the roofline leg profiles phase two when phase one has
    nothing to do)."""
    import json
    import subprocess
    import sys
    from bench_support import run_bench
    short, line, _ = run_bench(["--workload", "maxflow64k", "--steps", "1", "--warmup", "1", "--crash", str(crash), "--no-cpu-baseline"], timeout=600)
    assert abs(short["config"]["objective"] + 421.0) < 1e-9 and short["roofline"]["frac"] > 0
    assert line["unit"] == "pivots/s" and line["n_gpus"] == 1 and line["value"] > 0
    assert abs(line["config"]["objective"] + 421.0) < 1e-9            # scipy's max-flow value on this graph
    assert set(line["roofline"]["kernels"]) == {"price", "ftran_ratio", "update"}
    assert all(k["seconds_per_launch"] > 0 and k["contract_bytes_per_launch"] > 0 and k["kernel_bytes_per_launch"] > 0
               for k in line["roofline"]["kernels"].values())
    if crash:
        assert line["config"]["pivots_per_solve"] < 1000
    else:
        assert line["config"]["pivots_per_solve"] > 8000


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["25fv47", "netlib", "dense4096"])
def test_bench_two_ranks_sharing_one_device(workload):
    """The N > 1 logic of `bench.py` on a 1-GPU box: two ranks under `torch.distributed.run` as the driver launches them, both solving
    on device 0 (RELP_BENCH_SHARED_DEVICE=1) and talking over gloo -- RCCL refuses two ranks on one device, everything above the
    collectives (barriers, MAX / SUM reductions, the ticket queue shared through the store, the gathered records, rank 0's ONE
    line) is the code the multi-GPU run executes."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RELP_BENCH_SHARED_DEVICE="1")
    from bench_support import run_bench
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29537"]
    short, line, out = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload], env=env, launcher=launcher)
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints the one line
    assert short["n_gpus"] == 2 and short["value"] > 0 and "configs_summary" not in short and "cpu_baseline" not in short
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert "configs" not in line and "cpu_baseline" not in line  # N = 1 only
    config = line["config"]
    if workload == "netlib":
        assert line["scaling"] == "strong"
        assert len(config["tickets_per_rank"]) == 2 and sum(config["tickets_per_rank"]) == 2 * 45
        assert all(t > 0 for t in config["tickets_per_rank"])  # one queue, both ranks drew from it
        assert config["objectives_outside_reference_tolerance"] == []
    else:
        assert line["scaling"] == "weak"
        assert [r["rank"] for r in config["per_rank"]] == [0, 1]
        assert all(r["solves"] == 2 for r in config["per_rank"])
        total = sum(r["pivots"] for r in config["per_rank"])
        assert abs(line["value"] - total / config["makespan_s"]) <= 1e-6 * line["value"]  # whole-job aggregate over max-over-ranks time
