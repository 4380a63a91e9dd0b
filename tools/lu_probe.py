"""Diagnostic: solve Netlib LPs with the LU carry and with the explicit inverse, print objective, pivots, time per pivot."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd  # noqa: E402

names = sys.argv[1:] or ["AFIRO", "SC50A", "ADLITTLE", "SHARE2B", "SCFXM1", "25FV47"]
for name in names:
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    golden = os.path.join(ROOT, "tests", "golden", name + ".json")
    expected = json.load(open(golden))["objective_float"] if os.path.exists(golden) else float("nan")
    for carry, period in ((relp_amd.api.CARRY_EXPLICIT, 0), (relp_amd.api.CARRY_LU, 31), (relp_amd.api.CARRY_LU, 64), (relp_amd.api.CARRY_LU, 100)):
        try:
            s = relp_amd.Solver(carry=carry, refactor_period=period).load_mps(path)
            s.solve_relaxation()  # warm-up (graph capture, LDS attribute, first touch)
            t0 = time.time()
            r = s.solve_relaxation()
            wall = time.time() - t0
            pivots = r.pivots_phase_one + r.pivots_phase_two
            print("%-9s %-8s period %3d kind %d obj %.10g (expected %.10g) pivots %5d+%5d  %.2f ms  %.1f us/pivot  refactors %d (%.2f ms)" % (
                name, "LU" if carry else "explicit", period, r.kind, r.objective, expected, r.pivots_phase_one, r.pivots_phase_two,
                r.solve_seconds * 1e3, r.solve_seconds * 1e6 / max(1, pivots), r.refactors, r.refactor_seconds * 1e3), flush=True)
            s.close()
        except Exception as e:  # noqa: BLE001
            print(name, "carry", carry, "FAILED:", e, flush=True)
