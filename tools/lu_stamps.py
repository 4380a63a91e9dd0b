"""Diagnostic: per-segment cycle sums of the fused LU pivot kernel (librelp_amd_stamps.so, `make -C relp_amd/csrc stamps`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RELP_AMD_LIB"] = os.path.join(ROOT, "relp_amd", "librelp_amd_stamps.so")
sys.path.insert(0, ROOT)
import relp_amd
name = sys.argv[1] if len(sys.argv) > 1 else "25FV47"
period = int(sys.argv[2]) if len(sys.argv) > 2 else 31
carry = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # 2: the inverse-factor form (the segments are then: L^-1, U^-1, M; M', U^-1', -, -, L^-1')
s = relp_amd.Solver(carry=carry, refactor_period=period).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
r = s.solve_relaxation()
d = s.debug_stamps()
n = int(d[63])
print(name, "pivots", r.pivots_phase_one + r.pivots_phase_two, "launches stamped", n, "seconds", r.solve_seconds, "refactors", r.refactors)
names = ["candidates+q", "clear+scatter", "FTRAN L", "FTRAN etas", "FTRAN U", "alpha+ratio", "xB+BTRAN setup", "BTRAN U", "eta build",
         "BTRAN etas", "BTRAN L", "rho/w/pi out", "FT update+ctl"]
total = 0
for k, nm in enumerate(names):
    print("%-16s %9.0f cycles/launch" % (nm, d[k] / max(n, 1)))
    total += d[k] / max(n, 1)
print("%-16s %9.0f cycles/launch" % ("sum", total))
if carry == 2:
    print("inside the pass over M (before its end stamp):  row p of M staged %.0f, alpha to LDS %.0f, the k sums %.0f cycles/launch (the rest of 'BTRAN U': the fold and the set-up of the two row vectors)" % (d[13] / max(n, 1), d[14] / max(n, 1), d[15] / max(n, 1)))
for k, nm in enumerate(["L rows (FTRAN)", "U rows (FTRAN)", "U cols (BTRAN)", "L cols (BTRAN)"]):
    print("%-16s preamble %7.0f  level loop %7.0f cycles/launch  levels %.1f  slots %.0f  -> %.0f cycles per level" % (
        nm, d[32 + k] / max(n, 1), d[44 + k] / max(n, 1), d[36 + k] / max(n, 1), d[40 + k] / max(n, 1), d[44 + k] / max(d[36 + k], 1)))
