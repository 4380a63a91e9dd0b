"""Add the WHOLE pivot sequence to the golden Netlib fixtures (oracle; test infrastructure only; run in the build container).

    python oracle/gen_full_traces.py [NAME ...]      # default: every tests/golden/<NETLIB NAME>.json

`gen_golden.py` (the Python Fraction oracle) stored the first 64 pivots of each trace; the compiled twin `oracle/cpp/relp_cpu` -- pinned
to the same golden vectors pivot for pivot (tests/test_oracle_cpp.py) -- is fast enough to write all of them: `trace` = every
(phase, q, p, leaving) of the reference's algorithm on the file (25FV47: 1133 + 1259 pivots, 17 minutes on one core here).  A fixture is
only rewritten when the compiled oracle reproduces what it already holds (pivot counts, the first 64 pivots, final basis, exact optimum).
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def one(name):
    from relp_oracle import cpu
    from relp_oracle.mps import load_problem
    path = os.path.join(GOLDEN, name + ".json")
    golden = json.load(open(path))
    general, data = load_problem(os.path.join(ROOT, golden["file"]))
    record = cpu.solve_provider(data, trace=1000000)
    trace = [list(t) for t in record["trace_head"]]
    assert record["status"] == golden["status"] == "optimal", (name, record["status"])
    assert (record["pivots_phase1"], record["pivots_phase2"]) == (golden["pivots_phase1"], golden["pivots_phase2"]), name
    assert len(trace) == golden["pivots_phase1"] + golden["pivots_phase2"], name
    assert trace[:len(golden["trace_head"])] == golden["trace_head"], name
    assert list(record["basis"]) == list(golden["basis"]), name
    objective = general.objective_of(data.reconstruct_solution(record["solution"]))
    assert Fraction(golden["objective"]) == objective, name
    golden["trace"] = trace
    golden["trace_source"] = "oracle/cpp/relp_cpu (oracle/gen_full_traces.py)"
    with open(path, "w") as handle:
        json.dump(golden, handle, indent=None, separators=(",", ":"))
        handle.write("\n")
    return name, len(trace), record["seconds"]


def main(names):
    if not names:
        names = sorted(f[:-5] for f in os.listdir(GOLDEN)
                       if f.endswith(".json") and (f[0].isupper() or f[0].isdigit()) and "file" in json.load(open(os.path.join(GOLDEN, f))))
    with ProcessPoolExecutor(max_workers=4) as pool:
        for name, pivots, seconds in pool.map(one, names):
            print(name, pivots, "%.1f s" % seconds, flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
