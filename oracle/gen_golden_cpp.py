"""A golden Netlib fixture written by the COMPILED oracle alone (oracle/cpp/relp_cpu; test infrastructure only; run in the build container).

    python oracle/gen_golden_cpp.py NAME [NAME ...]      # writes tests/golden/<NAME>.json

For LPs the Python Fraction oracle (oracle/gen_golden.py) takes too long for: the compiled twin is pinned to the Fraction oracle pivot for
pivot on every other fixture (tests/test_oracle_cpp.py); a fixture written here says so in `trace_source`.  The optimum is checked against the
tolerance the reference's own test holds for the LP (tests/netlib/test.rs, tests/golden/netlib_expected.json).
"""
import json
import os
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main(names):
    from relp_oracle import cpu
    from relp_oracle.mps import load_problem
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
    for name in names:
        path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
        general, data = load_problem(path)
        record = cpu.solve_provider(data, trace=1000000)
        assert record["status"] == "optimal", (name, record["status"])
        objective = general.objective_of(data.reconstruct_solution(record["solution"]))
        want = expected[name]
        assert abs(objective - Fraction(str(want["expected"]))) < Fraction(str(want["tolerance"])), (name, float(objective), want)
        trace = [list(t) for t in record["trace_head"]]
        fixture = {"name": name, "file": os.path.relpath(path, ROOT), "m": data.nr_rows(), "n": data.nr_columns(),
                   "nnz": sum(len(c) for c in data.constraints), "pivots_phase1": record["pivots_phase1"], "pivots_phase2": record["pivots_phase2"],
                   "status": "optimal", "objective": "%d/%d" % (objective.numerator, objective.denominator),
                   "objective_bits": max(objective.numerator.bit_length(), objective.denominator.bit_length()),
                   "trace_head": trace[:64], "basis": list(record["basis"]), "oracle_seconds": record["seconds"], "trace": trace,
                   "trace_source": "oracle/cpp/relp_cpu alone (oracle/gen_golden_cpp.py): not cross-checked by the Fraction oracle; the optimum meets "
                                   "the reference's tolerance (tests/netlib/test.rs)"}
        with open(os.path.join(ROOT, "tests", "golden", name + ".json"), "w") as handle:
            json.dump(fixture, handle, indent=None, separators=(",", ":"))
            handle.write("\n")
        print(name, record["pivots_phase1"], record["pivots_phase2"], "%.1f s" % record["seconds"], float(objective), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
