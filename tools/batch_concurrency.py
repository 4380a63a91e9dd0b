"""Diagnostic (GPU): BASELINE config 4 -- the Netlib batch on ONE GPU -- against the carry and the number of LPs in flight.

    python tools/batch_concurrency.py [--presolve] [--passes 2] [--carries 0,1,2] [--workers 4,8,16,32,64] [--mixed 5]

The explicit carry streams the whole inverse per pivot (11.6 MB on 25FV47) and four LPs in flight already slow each other down;
an LU pivot kernel occupies ONE compute unit and moves < 1 MB.  One line per (carry, LPs in flight): seconds per pass over the
45 LPs, pivots/s, objectives outside the reference's tolerance.  `--mixed K`: the K costliest LPs under the explicit carry (4 in
flight) beside the rest under the inverse-factor LU carry (the given workers), two batches running concurrently.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import relp_amd  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument("--presolve", action="store_true")
parser.add_argument("--passes", type=int, default=2)
parser.add_argument("--carries", default="0,1,2")
parser.add_argument("--workers", default="4,8,16,32,64")
parser.add_argument("--mixed", type=int, default=5)
parser.add_argument("--certify", type=int, default=1)
parser.add_argument("--lu-refactor", type=int, default=0, help="relp_options.lu_refactor for the LU carries: 1 = the refactorisation kernels on the device")
args = parser.parse_args()

relp_amd.Solver().close()  # (HIP comes up on this thread: under rocprofv3 the workers' concurrent first calls crash the tool)
expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
names = sorted(n for n, e in expected.items() if os.path.exists(os.path.join(ROOT, "data", "netlib", n + ".SIF"))
               and (not e["ignored"] or "intensive" in e["ignored"]))
models = [relp_amd.Model(os.path.join(ROOT, "data", "netlib", n + ".SIF"), presolve=args.presolve) for n in names]
costs = [float(m.nr_rows) * float(m.nnz + m.nr_columns) for m in models]
order = sorted(range(len(names)), key=lambda k: (-costs[k], names[k]))


def check(entries, subset):
    wrong = []
    for e in entries:
        name = names[subset[e.model]]
        tol = max(expected[name]["tolerance"], 2e-5 if name == "25FV47" else 0.0)
        if e.status != 0 or abs(e.result.objective - expected[name]["expected"]) > tol:
            wrong.append(name)
    return wrong


def run_batch(subset, carry, workers, passes, out):
    """subset: model indices (cost-sorted).  Runs one warm-up pass and `passes` timed passes as one queue."""
    try:
        pool = relp_amd.Batch([models[k] for k in subset], devices=(0,), workers_per_device=workers, carry=carry, certify=args.certify,
                              lu_refactor=args.lu_refactor if carry == 2 else 0)
    except relp_amd.api.RelpError as error:
        out.update({"error": str(error)})
        return
    local = list(range(len(subset)))
    pool.run(local)
    t0 = time.perf_counter()
    entries, stats, _ = pool.run(local * passes)
    elapsed = time.perf_counter() - t0
    served = [e for e in entries if e.status == 0]
    out.update({"seconds_per_pass": elapsed / passes, "pivots": sum(e.result.pivots_phase_one + e.result.pivots_phase_two for e in served) / passes,
                "wrong": check(entries, subset), "refactor_seconds": sum(e.result.refactor_seconds for e in served) / passes,
                "solve_seconds": sum(e.result.solve_seconds for e in served) / passes,
                "longest": max((e.result.solve_seconds, names[subset[e.model]]) for e in served) if served else None})
    pool.close()


print("Netlib batch, %d LPs%s, one GPU, %d timed passes per line, certificate %s%s" % (
    len(names), " (presolved)" if args.presolve else "", args.passes, "inside" if args.certify else "off",
    ", LU-inverse carry refactorised on the device" if args.lu_refactor == 1 else ""), flush=True)
print("%-28s %8s %10s %12s %10s %10s  %s" % ("carry", "in flight", "s / pass", "pivots/s", "sum solve", "refactor", "longest LP; wrong"), flush=True)
label = {0: "explicit", 1: "LU + Forrest-Tomlin", 2: "LU inverse factors"}
for carry in [int(c) for c in args.carries.split(",") if c != ""]:
    for workers in [int(w) for w in args.workers.split(",")]:
        if carry == 0 and workers > 16:
            continue  # (every worker holds every LP's explicit inverse resident)
        out = {}
        run_batch(order, carry, workers, args.passes, out)
        if "error" in out:
            print("%-28s %8d  %s" % (label[carry], workers, out["error"]), flush=True)
            continue
        print("%-28s %8d %10.3f %12.0f %10.2f %10.2f  %s %.3f s; %s" % (label[carry], workers, out["seconds_per_pass"], out["pivots"] / out["seconds_per_pass"],
                                                                       out["solve_seconds"], out["refactor_seconds"], out["longest"][1], out["longest"][0], out["wrong"]), flush=True)
if args.mixed > 0:
    big, rest = order[:args.mixed], order[args.mixed:]
    for workers in [int(w) for w in args.workers.split(",")]:
        a, b = {}, {}
        ta = threading.Thread(target=run_batch, args=(big, 0, min(4, len(big)), args.passes, a))
        tb = threading.Thread(target=run_batch, args=(rest, 2, workers, args.passes, b))
        t0 = time.perf_counter()
        ta.start(); tb.start(); ta.join(); tb.join()
        if "error" in a or "error" in b:
            print("mixed: %s %s" % (a.get("error"), b.get("error")), flush=True)
            continue
        per_pass = max(a["seconds_per_pass"], b["seconds_per_pass"])
        print("%-28s %8s %10.3f %12.0f %10.2f %10.2f  explicit part %.3f s (%s), LU part %.3f s (%s); %s" % (
            "mixed: %d explicit + LU inv" % args.mixed, "4+%d" % workers, per_pass, (a["pivots"] + b["pivots"]) / per_pass, a["solve_seconds"] + b["solve_seconds"],
            b["refactor_seconds"], a["seconds_per_pass"], a["longest"][1], b["seconds_per_pass"], b["longest"][1], a["wrong"] + b["wrong"]), flush=True)
