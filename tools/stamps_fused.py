"""Diagnostic: per-segment cycle sums of workgroup 0 of the fused pivot kernel (librelp_amd_stamps.so, `make -C relp_amd/csrc stamps`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RELP_AMD_LIB"] = os.path.join(ROOT, "relp_amd", "librelp_amd_stamps.so")
sys.path.insert(0, ROOT)
import relp_amd
s = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", (sys.argv[1] if len(sys.argv) > 1 else "25FV47") + ".SIF"))
r = s.solve_relaxation()
d = s.debug_stamps()
n = int(d[63])
print("pivots", r.pivots_phase_one + r.pivots_phase_two, "launches stamped", n, "seconds", r.solve_seconds)
total = 0
for k, name in enumerate(["round trip 1", "select q", "ftran + own column", "sumsq + theta", "harris 2", "bcast + writer", "column update"]):
    print("%-20s %8.0f ticks/launch" % (name, d[k] / max(n, 1)))
    total += d[k] / max(n, 1)
print("%-20s %8.0f ticks/launch (shader clock, about 2.1 GHz => %.2f us)" % ("sum", total, total / 2100.0))
