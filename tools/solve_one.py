"""Diagnostic: solve one shipped Netlib file twice (for profiling under rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
name = sys.argv[1]
s = relp_amd.Solver(use_graph=int(os.environ.get("RELP_GRAPH", "1"))).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
for _ in range(2):
    r = s.solve_relaxation()
print(name, s.m, s.n_provider, r.pivots_phase_one, r.pivots_phase_two, r.solve_seconds, r.objective, r.polishes, r.max_residual)
s.close()
