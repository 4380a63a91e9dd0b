"""Every shipped Netlib LP under both ratio rules, certificate on: result, pivots, repair pivots, seconds (round 6: can the reference's textbook
rule -- tableau/mod.rs:287-313 -- be the f64 default beyond small-integer data?).    python tools/ratio_rule_scan.py [NAME ...]"""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd  # noqa: E402


def main():
    names = sys.argv[1:] or sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "data", "netlib", "*.SIF")))
    totals = {0: [0, 0, 0.0], 1: [0, 0, 0.0]}
    for name in names:
        row = []
        for rule in (1, 0):
            try:
                solver = relp_amd.Solver(certify=1, ratio_rule=rule).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
            except relp_amd.api.RelpError as error:
                row.append("load: %s" % str(error)[:60])
                continue
            start = time.perf_counter()
            try:
                r = solver.solve_relaxation()
                seconds = time.perf_counter() - start
                ok = r.kind == relp_amd.FINITE_OPTIMUM and r.certified
                totals[rule][0] += 1 if ok else 0
                totals[rule][1] += int(r.pivots_phase_one + r.pivots_phase_two)
                totals[rule][2] += seconds
                row.append("%s pivots %6d repairs %3d polishes %3d %.3f s" % ("certified" if ok else "kind %d certified %d" % (r.kind, r.certified),
                                                                              r.pivots_phase_one + r.pivots_phase_two, r.exact_repair_pivots, r.polishes, seconds))
            except relp_amd.api.RelpError as error:
                row.append("FAILED: %s" % str(error)[:80])
            solver.close()
        print("%-10s textbook: %-62s | harris: %s" % (name, row[0], row[1]), flush=True)
    print("certified optima: textbook %d, harris %d of %d; pivots %d / %d; seconds %.2f / %.2f" % (totals[1][0], totals[0][0], len(names), totals[1][1], totals[0][1],
                                                                                                 totals[1][2], totals[0][2]))


if __name__ == "__main__":
    main()
