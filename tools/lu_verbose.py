import os, sys
sys.path.insert(0, '/root/repo')
import relp_amd
s = relp_amd.Solver(carry=1, refactor_period=31, verbose=2).load_mps('/root/repo/data/netlib/25FV47.SIF')
r = s.solve_relaxation()
print(r.pivots_phase_one, r.pivots_phase_two, r.solve_seconds)
