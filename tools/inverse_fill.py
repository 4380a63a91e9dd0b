"""Diagnostic (CPU): how many entries do the INVERSES of the two triangular factors of a basis hold, against L + U and against B^-1?
Input: bases written by tools/dump_bases.py (gpurun_out/bases/<name>.npz).  The numbers behind the inverse-factor carry.

    python tools/inverse_fill.py 25FV47_050 GREENBEA_100 ...
"""
import os
import sys

import numpy as np
import scipy.linalg as la

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relp_amd.basis_inverse import lu_factor_host  # noqa: E402

for name in sys.argv[1:]:
    d = np.load(os.path.join(ROOT, "gpurun_out", "bases", name + ".npz"))
    m = int(d["m"])
    cs, ri, va = d["col_start"], d["row_index"], d["value"]
    cols = [[(int(ri[e]), float(va[e])) for e in range(cs[j], cs[j + 1])] for j in range(m)]
    f = lu_factor_host(cols)
    L, U = np.eye(m), np.diag(f["diag"])
    for i, row in enumerate(f["lower_rows"]):
        for j, v in row:
            L[i, j] = v
    for i, row in enumerate(f["upper_rows"]):
        for j, v in row:
            U[i, j] = v
    Li = la.solve_triangular(L, np.eye(m), lower=True, unit_diagonal=True)
    Ui = la.solve_triangular(U, np.eye(m), lower=False)
    nl, nu = int(np.count_nonzero(Li)) - m, int(np.count_nonzero(Ui))
    nb = int(np.count_nonzero(Ui @ Li))
    print("%-22s m %5d nnz(B) %6d nnz(L) %6d nnz(U)+m %6d depth %d/%d | nnz(L^-1)-m %7d nnz(U^-1) %7d  sum %7d = %.1f x (L+U)  | nnz(B^-1) %8d (%.0f%% of m^2)" % (
        name, m, len(ri), f["nnz_lower"], f["nnz_upper"] + m, f["depth_lower"], f["depth_upper"], nl, nu, nl + nu,
        (nl + nu) / (f["nnz_lower"] + f["nnz_upper"] + m), nb, 100.0 * nb / m / m), flush=True)
