"""Diagnostic: every shipped problem file in the four mode combinations (presolve x implicit bounds), exact certificate on;
the certified optimum must be the same rational in all of them (files both parsers reject, and presolve overflows, are skipped)."""
import glob, os, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
files = sorted(glob.glob(os.path.join(ROOT, "data", "*", "*.SIF")) + glob.glob(os.path.join(ROOT, "data", "*", "*.mps")))
bad, checked = [], 0
start = time.time()
for path in files:
    name = os.path.basename(path)
    values = {}
    for presolve in (0, 1):
        for bounded in (0, 1):
            s = relp_amd.Solver(certify=1, implicit_bounds=bounded)
            try:
                s.load_mps(path, presolve=bool(presolve))
            except relp_amd.RelpError as error:
                values[(presolve, bounded)] = "load: " + str(error)[:40]
                continue
            r = s.solve_relaxation()
            if r.kind == relp_amd.FINITE_OPTIMUM:
                values[(presolve, bounded)] = Fraction(s.objective_exact()) if r.certified else "uncertified: " + relp_amd.lib().relp_last_error(s._h).decode()[:40]
            else:
                values[(presolve, bounded)] = "kind %d" % r.kind
            s.close()
    solved = [v for v in values.values() if isinstance(v, Fraction)]
    kinds = {v for v in values.values() if not isinstance(v, Fraction) and not str(v).startswith("load")}
    checked += 1
    if len(set(solved)) > 1 or (solved and kinds) or len(kinds) > 1:
        bad.append(name)
        print(name, {k: (float(v) if isinstance(v, Fraction) else v) for k, v in values.items()}, flush=True)
print("%d files, %d with differing outcomes across the four modes %s (%.0f s)" % (checked, len(bad), bad, time.time() - start))
