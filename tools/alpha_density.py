"""Diagnostic: density of the FTRAN result alpha along a solve (decides whether a list-driven inverse update can pay),
with the count of entries above 1e-11 * max|alpha| beside it (noise left by the polish in structurally zero positions)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
for name in sys.argv[1:]:
    s = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    s.begin_phase_one()
    out = []
    for step in range(40):
        done, reason = s.iterate(200)
        sel = s.select_primal_pivot_column()
        if sel is None:
            break
        q, _ = sel
        row, alpha = s.select_primal_pivot_row(q)
        big = float(np.max(np.abs(alpha))) if len(alpha) else 0.0
        out.append((int(np.count_nonzero(alpha)), int(np.count_nonzero(np.abs(alpha) > 1e-11 * big))))
        if done < 200:
            break
    print(name, s.m, out)
    s.close()
