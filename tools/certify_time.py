"""Diagnostic: timeline of the exact certificate (RELP_TIME_CERTIFY=1)."""
import os, sys
os.environ["RELP_TIME_CERTIFY"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
for name in sys.argv[1:] or ["25FV47"]:
    s = relp_amd.Solver(certify=1).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    s.solve_relaxation()
    r = s.solve_relaxation()
    print(name, "solve %.2f ms  certify %.2f ms  certified %d" % (r.solve_seconds * 1e3, r.certify_seconds * 1e3, r.certified))
