"""Probe: max-flow LPs (BASELINE config 5 generator) of growing size through the device path; prints pivots, seconds."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import relp_amd
from relp_amd.workloads import max_flow_graph
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import maximum_flow

sizes = [(256, 2048), (512, 4096), (1024, 8192), (2048, 16384)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for nr_vertices, nr_arcs in sizes:
    tail, head, capacity = max_flow_graph(nr_vertices, nr_arcs)
    keep = (head != 0) & (tail != nr_vertices - 1)
    tail, head, capacity = tail[keep], head[keep], capacity[keep]
    expected = maximum_flow(csr_matrix((capacity.astype(np.int32), (tail, head)), shape=(nr_vertices, nr_vertices)), 0, nr_vertices - 1).flow_value
    t0 = time.time()
    model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
    t1 = time.time()
    solver = relp_amd.Solver(certify=0, implicit_bounds=int(os.environ.get("RELP_IMPLICIT_BOUNDS", "0")), use_graph=int(os.environ.get("RELP_GRAPH", "1")), crash=int(os.environ.get("RELP_CRASH", "0"))).load_model(model)
    t2 = time.time()
    r = solver.solve_relaxation()
    t3 = time.time()
    print("V=%d E=%d m=%d n=%d: kind=%d pivots=%d+%d solve=%.3fs (%.0f pivots/s) objective=%.6f expected=%d build=%.2fs upload=%.2fs resid=%.1e" % (
        nr_vertices, len(tail), model.nr_rows, model.nr_columns, r.kind, r.pivots_phase_one, r.pivots_phase_two, r.solve_seconds,
        (r.pivots_phase_one + r.pivots_phase_two) / max(r.solve_seconds, 1e-9), r.objective, -expected, t1 - t0, t2 - t1, r.max_residual), flush=True)
    solver.close()
