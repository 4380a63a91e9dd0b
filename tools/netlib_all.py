"""Diagnostic: solve every shipped Netlib file, compare with the reference's expected value."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
exp = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
names = sorted(f[:-4] for f in os.listdir(os.path.join(ROOT, "data", "netlib")) if f.endswith(".SIF"))
for name in names:
    try:
        s = relp_amd.Solver(certify=1, max_pivots=200000).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    except relp_amd.RelpError as e:
        print(name, "LOAD ERROR", e); continue
    t = time.time()
    r = s.solve_relaxation()
    e = exp.get(name)
    diff = abs(r.objective - e["expected"]) if e and r.kind == 1 else None
    print("%-9s m=%5d n=%5d kind=%d pivots=%6d+%6d obj=%.10g diff=%s tol=%s cert=%d rep=%d maxres=%.1e %.2fs %s" % (
        name, s.m, s.n_provider, r.kind, r.pivots_phase_one, r.pivots_phase_two, r.objective, diff, e and e["tolerance"],
        r.certified, r.exact_repair_pivots, r.max_residual, time.time() - t, (e and e["ignored"]) or ""), flush=True)
    s.close()
