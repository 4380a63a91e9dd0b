"""First pivot at which relp_solve_exact leaves the golden trace (diagnostic).  python3 tools/exact_trace_check.py LP [max_pivots]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
name = sys.argv[1]
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 400
golden = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))
solver = relp_amd.Solver().load_mps(os.path.join(ROOT, golden["file"]))
got = solver.solve_exact(first_limbs=4, max_limbs=64, max_pivots=cap)
n_art = solver.n_art
want = [(ph, q + n_art if True else q, p, lv) for ph, q, p, lv in golden.get("trace", golden["trace_head"])]
trace = got["trace"]
print(name, "status", got["status"], "limbs", got["limbs"], "pivots", len(trace), "survived", got["survived"])
sys.path.insert(0, os.path.join(ROOT, "tests"))
for k, t in enumerate(trace):
    if k >= len(want):
        print("ran past the golden trace at", k); break
    g = golden.get("trace", golden["trace_head"])[k]
    if (t[0], t[2]) != (g[0], g[2]):
        print("first difference at pivot", k, "device", t, "golden", g); break
else:
    print("no difference in (phase, row) over", min(len(trace), len(want)), "pivots")
