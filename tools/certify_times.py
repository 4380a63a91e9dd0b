"""Diagnostic: time of the exact certificate on the larger Netlib LPs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
for name in sys.argv[1:] or ["25FV47", "CZPROB", "BNL2", "CYCLE", "GREENBEA", "80BAU3B"]:
    s = relp_amd.Solver(certify=1).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    t0 = time.time()
    r = s.solve_relaxation()
    bits = len(s.objective_exact()) if r.certified else 0
    print("%-9s m=%5d solve %.3f s  certify %.3f s  certified=%d repairs=%d  (objective text %d chars) %s" % (
        name, s.m, r.solve_seconds, r.certify_seconds, r.certified, r.exact_repair_pivots, bits,
        "" if r.certified else relp_amd.lib().relp_last_error(s._h).decode()), flush=True)
    s.close()
