import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import relp_amd
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for name in sys.argv[1:]:
    for ib in (0, 1):
        s = relp_amd.Solver(implicit_bounds=ib).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
        s.solve_relaxation()
        r = s.solve_relaxation()
        piv = r.pivots_phase_one + r.pivots_phase_two
        print("%-9s ib %d m %5d kind %d obj %.9g pivots %6d %8.2f ms %6.1f us/pivot" % (name, ib, s.m, r.kind, r.objective, piv, r.solve_seconds * 1e3, r.solve_seconds * 1e6 / piv), flush=True)
        s.close()
