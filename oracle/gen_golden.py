"""Generate golden fixtures with the exact oracle (run in the build container; outputs are committed).

    python oracle/gen_golden.py NAME [NAME ...]      # writes tests/golden/<NAME>.json

Each fixture holds: dimensions of the standard form, exact optimal objective (``num/den``), pivot
counts, the first pivots of the reference-rule trace ``(phase, q, p, leaving)`` and the final basis.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from relp_oracle import solve_relaxation, FiniteOptimum  # noqa: E402
from relp_oracle.mps import load_problem  # noqa: E402
from relp_oracle.solve import Trace  # noqa: E402


def find(name):
    for sub, ext in (("netlib", ".SIF"), ("burkardt", ".mps"), ("unicamp", ".mps"), ("cook", ".mps")):
        path = os.path.join(ROOT, "data", sub, name + ext)
        if os.path.exists(path):
            return path
    raise FileNotFoundError(name)


def main(names):
    for name in names:
        path = find(name)
        start = time.time()
        general, data = load_problem(path)
        trace = Trace()
        result = solve_relaxation(data, trace=trace)
        record = {
            "name": name, "file": os.path.relpath(path, ROOT),
            "m": data.nr_rows(), "n": data.nr_columns(),
            "nnz": sum(len(c) for c in data.constraints),
            "pivots_phase1": sum(1 for p in trace.pivots if p[0] == 1),
            "pivots_phase2": sum(1 for p in trace.pivots if p[0] == 2),
            "trace_head": [[ph, q, p, lv] for ph, q, p, lv, _ in trace.pivots[:64]],
            "oracle_seconds": None,
        }
        if isinstance(result, FiniteOptimum):
            objective = general.objective_of(data.reconstruct_solution(result.solution))
            record.update(status="optimal", objective="%d/%d" % (objective.numerator, objective.denominator),
                          objective_float=float(objective), basis=result.basis,
                          objective_bits=max(objective.numerator.bit_length(), objective.denominator.bit_length()))
        else:
            record.update(status=repr(result).lower())
        record["oracle_seconds"] = round(time.time() - start, 2)
        prefix = "" if "/netlib/" in path else os.path.basename(os.path.dirname(path)) + "_"
        out = os.path.join(ROOT, "tests", "golden", prefix + name + ".json")
        with open(out, "w") as handle:
            json.dump(record, handle, indent=None, separators=(",", ":"))
            handle.write("\n")
        print(name, record["m"], record["n"], record["pivots_phase1"], record["pivots_phase2"],
              record.get("objective_float"), record["oracle_seconds"], flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
