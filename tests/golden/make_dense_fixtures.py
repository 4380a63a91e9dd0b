"""Golden optima of the synthetic dense LPs (BASELINE config 3 and its scaled-down twins).

Small twins: exact optimum from the oracle (FullInitialBasis route).  All sizes: HiGHS f64 objective via scipy.
Run in the build container; writes tests/golden/dense_lp.json.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from scipy.optimize import linprog  # noqa: E402

from relp_amd.workloads import dense_lp  # noqa: E402
from relp_oracle import MatrixData, Variable, solve_relaxation_full_basis  # noqa: E402

out = {}
for m, n in [(16, 32), (64, 128), (256, 512), (1024, 2048), (4096, 8192)]:
    a, b, c = dense_lp(m, n)
    entry = {"m": m, "n": n, "seed": "0x5EED0001"}
    if m <= 64:
        cols = [[(i, int(a[j, i])) for i in range(m)] for j in range(n)]
        data = MatrixData(cols, [int(v) for v in b], [], 0, 0, m, 0, [Variable(int(v)) for v in c])
        result = solve_relaxation_full_basis(data)
        entry["exact"] = "%d/%d" % (result.objective.numerator, result.objective.denominator)
        entry["exact_pivots"] = None
    start = time.time()
    hs = linprog(c, A_ub=a.T, b_ub=b, method="highs")
    entry["highs"] = float(hs.fun)
    entry["highs_seconds"] = round(time.time() - start, 2)
    entry["highs_iterations"] = int(hs.nit)
    out["%dx%d" % (m, n)] = entry
    print(entry, flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "dense_lp.json"), "w"), indent=1, sort_keys=True)
