import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import relp_amd
for name in sys.argv[1:]:
    s = relp_amd.Solver(carry=int(os.environ.get("RELP_PROBE_CARRY", "1"))).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    r = s.solve_relaxation()
    print(name, "pivots", r.pivots_phase_one + r.pivots_phase_two, "solve %.1f ms" % (r.solve_seconds * 1e3), "refactors", r.refactors, "%.1f ms" % (r.refactor_seconds * 1e3), flush=True)
    s.close()
