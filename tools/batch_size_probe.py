"""Diagnostic: wall time of one 25FV47 solve against the number of pivots captured per hipGraph launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
path = os.path.join(ROOT, "data", "netlib", (sys.argv[1] if len(sys.argv) > 1 else "25FV47") + ".SIF")
for batch in (16, 32, 64, 96, 128, 256):
    s = relp_amd.Solver(certify=0, pivots_per_launch=batch).load_mps(path)
    s.solve_relaxation()
    best = 1e9
    for _ in range(5):
        r = s.solve_relaxation()
        best = min(best, r.solve_seconds)
    print("pivots_per_launch %4d: %.2f ms (%d pivots, %.0f pivots/s)" % (batch, best * 1e3, r.pivots_phase_one + r.pivots_phase_two,
                                                                       (r.pivots_phase_one + r.pivots_phase_two) / best))
    s.close()
