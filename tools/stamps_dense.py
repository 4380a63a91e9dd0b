"""Diagnostic: per-segment cycle sums of the one-workgroup ratio-test kernel (K2) on the dense LP of BASELINE config 3
(needs librelp_amd_stamps.so: make -C relp_amd/csrc stamps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RELP_AMD_LIB"] = os.path.join(ROOT, "relp_amd", "librelp_amd_stamps.so")
sys.path.insert(0, ROOT)
import relp_amd
from relp_amd.workloads import dense_lp
m, n = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096x8192").split("x"))
a, b, c = dense_lp(m, n)
s = relp_amd.Solver(polish_period=512).load_dense_le(a, b, c)
r = s.solve_relaxation()
d = s.debug_stamps()
count = int(d[63])
print("pivots", r.pivots_phase_one + r.pivots_phase_two, "launches stamped", count, "seconds", r.solve_seconds)
names = ["status", "select q", "ftran", "sumsq+theta", "harris", "xB+compact", "ctl write"]
for k, name in enumerate(names):
    print("%-12s %8.0f cycles/launch" % (name, d[k] / max(count, 1)))
