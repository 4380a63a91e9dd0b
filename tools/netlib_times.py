"""Diagnostic: per-LP solve time of the Netlib batch (no certificate), sorted; shows the makespan bound of config 4."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
exp = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
names = sorted(n for n, e in exp.items() if os.path.exists(os.path.join(ROOT, "data", "netlib", n + ".SIF")) and (not e["ignored"] or "intensive" in e["ignored"]))
rows = []
for name in names:
    s = relp_amd.Solver(implicit_bounds=int(os.environ.get("RELP_IMPLICIT_BOUNDS", "0"))).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    s.solve_relaxation()
    r = s.solve_relaxation()
    rows.append((r.solve_seconds, name, s.m, s.n_provider, r.pivots_phase_one + r.pivots_phase_two))
    s.close()
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
print("total %.3f s over %d LPs" % (total, len(rows)))
for sec, name, m, n, piv in rows[:12]:
    print("%-9s m=%5d n=%5d pivots=%6d %.3f s (%.0f pivots/s, %.1f us/pivot)" % (name, m, n, piv, sec, piv / sec, 1e6 * sec / piv))
