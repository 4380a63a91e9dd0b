"""Diagnostic: limb counts the exact device simplex needs per LP, and how far each width gets on 25FV47."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
names = sys.argv[1:] or ["AFIRO", "SC50A", "SC50B", "KB2", "SC105", "SCAGR7", "ADLITTLE", "SHARE2B", "BLEND", "SC205", "LOTFI", "STOCFOR1", "ISRAEL", "SHARE1B", "E226", "BRANDY"]
for name in names:
    golden = os.path.join(ROOT, "tests", "golden", name + ".json")
    g = json.load(open(golden)) if os.path.exists(golden) else None
    s = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    t0 = time.time()
    max_limbs = int(os.environ.get("RELP_PROBE_MAX_LIMBS", "128" if name == "25FV47" else "32"))
    r = s.solve_exact(first_limbs=int(os.environ.get("RELP_PROBE_FIRST_LIMBS", "1")), max_limbs=max_limbs, max_pivots=20000)
    dt = time.time() - t0
    ok = g is not None and r["status"] == 1 and r["objective"] == g["objective"] and (r["pivots_phase_one"], r["pivots_phase_two"]) == (g["pivots_phase1"], g["pivots_phase2"])
    print("%-9s m %4d status %d limbs %2d pivots %5d+%5d  %.2f s  survived %s  %s" % (
        name, s.m, r["status"], r["limbs"], r["pivots_phase_one"], r["pivots_phase_two"], dt, r["survived"],
        "== golden (optimum, pivot counts)" if ok else ("" if g is None else "DIFFERS from golden" if r["status"] == 1 else "")), flush=True)
    s.close()
