"""Diagnostic: which presolved Netlib LPs load, solve and certify."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
exp = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
names = sorted(n for n, e in exp.items() if os.path.exists(os.path.join(ROOT, "data", "netlib", n + ".SIF")) and (not e["ignored"] or "intensive" in e["ignored"]))
for name in names:
    s = relp_amd.Solver(certify=1)
    try:
        s.load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"), presolve=True)
    except relp_amd.RelpError as e:
        print(name, "LOAD:", e); continue
    r = s.solve_relaxation()
    msg = relp_amd.lib().relp_last_error(s._h).decode() if not r.certified else ""
    print("%-9s m=%5d kind=%d pivots=%6d certified=%d %s" % (name, s.m, r.kind, r.pivots_phase_one + r.pivots_phase_two, r.certified, msg))
    s.close()
