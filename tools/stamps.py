"""Diagnostic: per-segment cycle sums of the fused kernel (needs librelp_amd_stamps.so built with -DRELP_STAMPS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RELP_AMD_LIB"] = os.path.join(ROOT, "relp_amd", "librelp_amd_stamps.so")
sys.path.insert(0, ROOT)
import relp_amd
s = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", sys.argv[1] if len(sys.argv) > 1 else "25FV47.SIF"))
r = s.solve_relaxation()
d = s.debug_stamps()
n = int(d[63])
print("pivots", r.pivots_phase_one + r.pivots_phase_two, "launches stamped", n, "seconds", r.solve_seconds)
names = ["status", "select q", "ftran", "sumsq+theta", "harris", "xB+compact", "ctl write"]
for k, name in enumerate(names):
    print("%-12s %8.0f cycles/launch" % (name, d[k] / max(n, 1)))
