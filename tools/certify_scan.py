"""Diagnostic: exact certificate over the whole Netlib batch (no presolve); prints failures and the total time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
exp = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
names = sorted(n for n, e in exp.items() if os.path.exists(os.path.join(ROOT, "data", "netlib", n + ".SIF")) and (not e["ignored"] or "intensive" in e["ignored"]))
ok, total_solve, total_cert, worst = 0, 0.0, 0.0, ("", 0.0)
mode = int(os.environ.get("RELP_IMPLICIT_BOUNDS", "0"))
for name in names:
    s = relp_amd.Solver(certify=1, implicit_bounds=mode).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    r = s.solve_relaxation()
    total_solve += r.solve_seconds
    total_cert += r.certify_seconds
    if r.certify_seconds > worst[1]:
        worst = (name, r.certify_seconds)
    if r.kind == relp_amd.FINITE_OPTIMUM and r.certified:
        ok += 1
    else:
        print("%-9s kind=%d certified=%d repairs=%d %s" % (name, r.kind, r.certified, r.exact_repair_pivots, relp_amd.lib().relp_last_error(s._h).decode()))
    s.close()
print("%d of %d certified bit-exact; solve %.2f s, certificates %.2f s (slowest %s %.2f s)" % (ok, len(names), total_solve, total_cert, worst[0], worst[1]))
