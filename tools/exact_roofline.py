"""The exact fixed-width simplex against its integer-multiply roofline: per width, seconds per step of the loop and the word products
(64 x 64 -> 128 bit) of the update of N, needed by the entries and issued by the waves (`relp_get_exact_counters`).

    python3 tools/exact_roofline.py [LP ...]        (default 25FV47; peak from tools/micro/intmul_rates.hip, see profiles/)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd  # noqa: E402


def main():
    names = sys.argv[1:] or ["25FV47"]
    for name in names:
        solver = relp_amd.Solver().load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
        start = time.perf_counter()
        got = solver.solve_exact(first_limbs=4, max_limbs=128)
        seconds = time.perf_counter() - start
        records = solver.exact_counters()
        solver.close()
        print(json.dumps({"lp": name, "status": got["status"], "limbs": got["limbs"], "pivots": got["pivots_phase_one"] + got["pivots_phase_two"],
                          "seconds": seconds, "survived": got["survived"]}))
        for r in records:
            update = r["step_seconds"]["update of N"]
            r["update_word_products_per_second_issued"] = r["update_word_products_issued"] / update if update > 0 else None
            print(json.dumps(r))


if __name__ == "__main__":
    main()
