"""The refactorisation beside the pivots (relp_options.lu_refactor = 3, RELP_CARRY_LU_INVERSE) against the synchronous paths: solve
time, refactorisations taken asynchronously / abandoned by the guard, worst residual the guard saw, per batch size.

    python3 tools/async_refactor_probe.py [LP ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd  # noqa: E402

for name in sys.argv[1:] or ["25FV47"]:
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    for where, batch in ((2, 64), (1, 64), (3, 8), (3, 16), (3, 32)):
        solver = relp_amd.Solver(carry=2, lu_refactor=where, certify=1, pivots_per_launch=batch).load_mps(path)
        solver.solve_relaxation()
        best = None
        for _ in range(3):
            start = time.perf_counter()
            result = solver.solve_relaxation()
            elapsed = time.perf_counter() - start
            best = elapsed if best is None else min(best, elapsed)
        record = solver.record()
        pivots = result.pivots_phase_one + result.pivots_phase_two
        print("%-8s lu_refactor %d batch %2d: %.1f ms, %d pivots, %.1f us per pivot, refactors %d (async %d, abandoned %d, worst residual %.1e), certified %s" % (
            name, where, batch, 1e3 * best, pivots, 1e6 * best / pivots, result.refactors, record["async_refactors"], record["async_refactors_abandoned"],
            record["async_worst_residual"], result.certified), flush=True)
        solver.close()
