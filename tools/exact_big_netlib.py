"""Round 6: the Netlib LPs the reference's own suite ignores as too intensive (tests/netlib/test.rs: 80BAU3B, BNL2, CYCLE, GREENBEA, GREENBEB, ...)
on the exact device path: relp_solve_exact up to 128 limbs against the f64 loop's certified optimum -- two independent exact computations.
    python tools/exact_big_netlib.py [NAME ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd  # noqa: E402


def main():
    for name in sys.argv[1:] or ["BNL2", "CYCLE", "80BAU3B", "GREENBEA", "GREENBEB"]:
        solver = relp_amd.Solver(certify=1).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
        relaxed = solver.solve_relaxation()
        certified = solver.objective_exact() if relaxed.certified else None
        start = time.perf_counter()
        try:
            got = solver.solve_exact(first_limbs=4, max_limbs=128, max_pivots=400000, trace_capacity=1)
            seconds = time.perf_counter() - start
            print("%-9s m %5d: status %d, %d + %d pivots, %d limbs, widths %s, %.1f s, objective %s the certified one (%d bits)" % (
                name, solver.m, got["status"], got["pivots_phase_one"], got["pivots_phase_two"], got["limbs"], got["survived"], seconds,
                "==" if got["objective"] == certified else "!=", max(len(bin(int(t))) - 2 for t in (certified or "0/1").split("/"))), flush=True)
        except relp_amd.api.RelpError as error:
            print("%-9s m %5d: %s (%.1f s)" % (name, solver.m, str(error)[:160], time.perf_counter() - start), flush=True)
        solver.close()


if __name__ == "__main__":
    main()
