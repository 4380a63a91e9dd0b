import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import relp_amd
for name in sys.argv[1:]:
    for carry in [int(c) for c in os.environ.get("RELP_PROBE_CARRIES", "0,1,2").split(",")]:
        try:
            s = relp_amd.Solver(carry=carry, refactor_period=int(os.environ.get("RELP_PROBE_PERIOD", "0"))).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
            s.solve_relaxation()
            r = s.solve_relaxation()
            piv = r.pivots_phase_one + r.pivots_phase_two
            print("%-9s carry %d kind %d obj %.10g pivots %6d %8.2f ms %6.1f us/pivot refactors %d (%.1f ms)" % (name, carry, r.kind, r.objective, piv, r.solve_seconds * 1e3, r.solve_seconds * 1e6 / max(1, piv), r.refactors, r.refactor_seconds * 1e3), flush=True)
            s.close()
        except Exception as e:
            print(name, carry, "FAILED", str(e)[:200], flush=True)
