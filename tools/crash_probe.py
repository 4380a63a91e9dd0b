import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, relp_amd
from relp_amd.workloads import max_flow_graph
for V, E in ((8192, 65536), (65536, 1048576)):
    tail, head, cap = max_flow_graph(V, E)
    model = relp_amd.Model.max_flow(V, list(zip(tail.tolist(), head.tolist(), cap.tolist())), 0, V - 1)
    for crash in (0, 1):
        os.environ["RELP_TIME_SOLVE"] = "1"
        t0 = time.time()
        s = relp_amd.Solver(implicit_bounds=1, crash=crash).load_model(model)
        t1 = time.time()
        r = s.solve_relaxation()
        r2 = s.solve_relaxation()
        print(V, E, "crash", crash, "load %.2f s" % (t1 - t0), "solve %.3f s (second %.3f)" % (r.solve_seconds, r2.solve_seconds), "pivots", r.pivots_phase_one, r.pivots_phase_two, "obj", r.objective, flush=True)
        s.close()
