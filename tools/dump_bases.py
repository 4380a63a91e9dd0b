"""Diagnostic: bases along a solve, written as .npz (CSC of the basis columns in slot order) for offline work on the LU carry.

    python tools/dump_bases.py 25FV47 GREENBEA 80BAU3B maxflow:8192x65536

For every LP: the explicit carry runs to 25 / 50 / 75 / 100 % of its pivots (`max_pivots`), the basis is read back
(`relp_get_basis`) and its columns are taken from the host model.  Output: gpurun_out/bases/<name>_<percent>.npz with
m, col_start, row_index, value (artificial k = unit column of its row).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd  # noqa: E402
from relp_amd.workloads import max_flow_graph  # noqa: E402

out_dir = os.path.join(ROOT, "gpurun_out", "bases")
os.makedirs(out_dir, exist_ok=True)


def model_of(name):
    if name.startswith("maxflow:"):
        v, e = (int(t) for t in name.split(":")[1].split("x"))
        tail, head, capacity = max_flow_graph(v, e)
        keep = (head != 0) & (tail != v - 1)
        tail, head, capacity = tail[keep], head[keep], capacity[keep]
        return relp_amd.Model.max_flow(v, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, v - 1)
    return relp_amd.Model(os.path.join(ROOT, "data", "netlib", name + ".SIF"))


for name in sys.argv[1:]:
    model = model_of(name)
    solver = relp_amd.Solver(certify=0).load_model(model)
    full = solver.solve_relaxation()
    total = full.pivots_phase_one + full.pivots_phase_two
    print(name, "m", solver.m, "pivots", total, "seconds", full.solve_seconds, flush=True)
    solver.close()
    pivots = model.pivot_element_indices()
    buf_rows, buf_vals = np.zeros(model.nr_rows, np.int32), np.zeros(model.nr_rows, np.float64)  # (Model.column allocates per call)
    count = C.c_int32()

    def column(j):
        relp_amd.lib().relp_model_column(model._h, j, model.nr_rows, C.byref(count), buf_rows.ctypes.data_as(C.POINTER(C.c_int32)),
                                         buf_vals.ctypes.data_as(C.POINTER(C.c_double)))
        return buf_rows[:count.value], buf_vals[:count.value]

    art_rows = sorted(set(range(model.nr_rows)) - {r for r, _ in pivots})  # artificial k sits on the k-th row without a slack pivot
    for percent in (25, 50, 75, 100):
        s = relp_amd.Solver(certify=0, max_pivots=max(1, total * percent // 100) if percent < 100 else 0).load_model(model)
        s.solve_relaxation()
        basis = s.basis()
        s.close()
        col_start, rows, vals = [0], [], []
        for c in basis:
            if c >= 0:
                r, v = column(int(c))
                rows.extend(int(x) for x in r)
                vals.extend(float(x) for x in v)
            else:
                rows.append(art_rows[-1 - int(c)])
                vals.append(1.0)
            col_start.append(len(rows))
        tag = name.replace(":", "_")
        np.savez_compressed(os.path.join(out_dir, "%s_%03d.npz" % (tag, percent)), m=np.int32(len(basis)),
                            col_start=np.array(col_start, np.int64), row_index=np.array(rows, np.int32), value=np.array(vals))
        print("  ", percent, "% nnz", len(rows), flush=True)
