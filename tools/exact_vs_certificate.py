"""Diagnostic: the exact device simplex (relp_solve_exact, the reference's pivot rule in fixed-width integers) against the exact
certificate of the f64 loop on LPs that have no golden file (the Fraction oracle is too slow for them): the two optima must be the
same rational."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
for name in sys.argv[1:] or ["GROW7", "STAIR", "MODSZK1", "BNL1"]:
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    s = relp_amd.Solver(certify=1).load_mps(path)
    r = s.solve_relaxation()
    certified = s.objective_exact()
    t0 = time.time()
    e = s.solve_exact(first_limbs=2, max_limbs=128, max_pivots=100000)
    dt = time.time() - t0
    print("%-8s m %5d f64+certificate: kind %d %s | exact simplex: status %d limbs %d pivots %d+%d %.1f s | same optimum: %s" % (
        name, s.m, r.kind, str(certified)[:40], e["status"], e["limbs"], e["pivots_phase_one"], e["pivots_phase_two"], dt,
        certified is not None and e["status"] == 1 and e["objective"] == certified), flush=True)
    s.close()
