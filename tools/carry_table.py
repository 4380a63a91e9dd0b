"""Diagnostic: microseconds per pivot under both carries.  LPs the LU carry does not take (more
than ~3300 rows, implicit bounds) are reported as such -- no silent switch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
extra = {}
if os.environ.get("RELP_TABLE_PERIOD"):  # A/B: another refactor period for the LU carries
    extra["refactor_period"] = int(os.environ["RELP_TABLE_PERIOD"])
names = sys.argv[1:] or ["25FV47", "GREENBEA", "80BAU3B", "SCFXM2", "PILOT4", "BNL1"]
for name in names:
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    for label, options in (("explicit", dict(carry=0)), ("explicit+implicit bounds", dict(carry=0, implicit_bounds=1)), ("lu", dict(carry=1)), ("lu inverse factors", dict(carry=2))):
        try:
            s = relp_amd.Solver(**dict(options, **(extra if options["carry"] else {}))).load_mps(path)
        except relp_amd.api.RelpError as e:
            print("%-9s %-26s not available: %s" % (name, label, str(e)[:110]), flush=True)
            continue
        s.solve_relaxation()
        r = s.solve_relaxation()
        pivots = r.pivots_phase_one + r.pivots_phase_two
        print("%-9s %-26s m %5d kind %d obj %.9g pivots %6d  %8.2f ms  %6.1f us/pivot  refactors %d (%.1f ms on the host)" % (
            name, label, s.m, r.kind, r.objective, pivots, r.solve_seconds * 1e3, r.solve_seconds * 1e6 / max(1, pivots), r.refactors, r.refactor_seconds * 1e3), flush=True)
        s.close()
