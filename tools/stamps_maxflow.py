"""Diagnostic: wall-clock stamps inside price_unit_kernel on a max-flow LP (needs librelp_amd_stamps.so: `make -C relp_amd/csrc stamps`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RELP_AMD_LIB"] = os.path.join(ROOT, "relp_amd", "librelp_amd_stamps.so")
sys.path.insert(0, ROOT)
import relp_amd
from relp_amd.workloads import max_flow_graph

nr_vertices, nr_arcs = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "65536x1048576").split("x"))
tail, head, capacity = max_flow_graph(nr_vertices, nr_arcs)
keep = (head != 0) & (tail != nr_vertices - 1)
tail, head, capacity = tail[keep], head[keep], capacity[keep]
model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
solver = relp_amd.Solver(certify=0, implicit_bounds=1, use_graph=0, crash=0).load_model(model)
r = solver.solve_relaxation()
d = solver.debug_stamps()
n = max(int(d[48]), 1)
print("pivots", r.pivots_phase_one + r.pivots_phase_two, "launches stamped", n, "seconds", r.solve_seconds)
names = ["control word", "arcs, positions, costs", "-pi gathers, rho bytes", "rho/w of the hits, weights of the wanted", "weights, candidates", "workgroup arg-max, publish"]
for which, base in (("first workgroup", 32), ("last workgroup", 40)):
    print(which)
    for k, name in enumerate(names):
        print("  %-44s %7.2f us" % (name, d[base + k] / n * 0.01))
print("first workgroup's entry -> last workgroup's exit  %7.2f us" % (d[50] / n * 0.01))
