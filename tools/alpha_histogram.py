"""Diagnostic: histogram of log10(|alpha_i| / max|alpha|) late in a solve: real entries against the noise a polish leaves in
structurally zero positions of the inverse (input for a drop tolerance)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd
for name in sys.argv[1:]:
    s = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    s.begin_phase_one()
    hist = np.zeros(40, dtype=np.int64)
    for step in range(12):
        done, reason = s.iterate(300)
        sel = s.select_primal_pivot_column()
        if sel is None:
            break
        q, _ = sel
        row, alpha = s.select_primal_pivot_row(q)
        a = np.abs(alpha[alpha != 0])
        if len(a):
            e = np.clip(-np.floor(np.log10(a / a.max())).astype(int), 0, 39)
            hist += np.bincount(e, minlength=40)
        if done < 300:
            break
    print(name, s.m, "decades below the maximum -> count:", {int(k): int(v) for k, v in enumerate(hist) if v})
    s.close()
