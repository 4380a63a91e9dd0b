"""Parameter sweep on the GPU (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
path = os.path.join(ROOT, "data", "netlib", "25FV47.SIF")
for period in (64, 128, 256, 512, 1024):
    for ppl in (32, 64, 128):
        s = relp_amd.Solver(polish_period=period, pivots_per_launch=ppl).load_mps(path)
        s.solve_relaxation()
        t = time.perf_counter(); n = 0
        for _ in range(3):
            r = s.solve_relaxation(); n += r.pivots_phase_one + r.pivots_phase_two
        dt = time.perf_counter() - t
        print("period %4d ppl %3d: %7.0f pivots/s  pivots %d obj %.9f maxres %.2e polishes %d" % (period, ppl, n / dt, n // 3, r.objective, r.max_residual, r.polishes), flush=True)
        s.close()
