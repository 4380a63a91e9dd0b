"""Diagnostic: microseconds per pivot of the LU carry for library variants built side by side (RELP_AMD_LIB)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys; sys.path.insert(0, %r)
import relp_amd
for name in %r:
    s = relp_amd.Solver(carry=1, refactor_period=%d).load_mps(%r + "/data/netlib/" + name + ".SIF")
    s.solve_relaxation(); r = s.solve_relaxation()
    p = r.pivots_phase_one + r.pivots_phase_two
    print("%%-10s %%-9s period %%2d pivots %%6d  %%8.2f ms  %%6.1f us/pivot  (refactors %%d: %%.1f ms)" %% (%r, name, %d, p, r.solve_seconds*1e3, r.solve_seconds*1e6/p, r.refactors, r.refactor_seconds*1e3), flush=True)
'''
names = ["25FV47", "GREENBEA", "80BAU3B", "BNL2"]
for lib in sys.argv[1:]:
    for period in (31, 63):
        env = dict(os.environ, RELP_AMD_LIB=os.path.join(ROOT, "relp_amd", lib))
        subprocess.run([sys.executable, "-c", code % (ROOT, names, period, ROOT, lib, period)], env=env)
