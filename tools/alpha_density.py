"""Diagnostic: density of the FTRAN result alpha along a solve (decides whether a list-driven inverse update can pay)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
for name in sys.argv[1:]:
    s = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    s.begin_phase_one()
    out = []
    for step in range(30):
        done, reason = s.iterate(400)
        sel = s.select_primal_pivot_column()
        if sel is None:
            break
        q, _ = sel
        row, alpha = s.select_primal_pivot_row(q)
        out.append(int(np.count_nonzero(alpha)))
        if done < 400:
            break
    print(name, s.m, out)
    s.close()
