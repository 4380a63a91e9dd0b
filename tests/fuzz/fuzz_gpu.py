"""One-off fuzzing on the GPU box: random small LPs with every row kind and bounded variables, both formulations (bound rows
explicit / implicit bounds), against the exact oracle.  Prints the seeds that disagree.

    python tests/fuzz/fuzz_gpu.py [first_seed] [count] [zero]
"""
import os, random, sys, time
from fractions import Fraction
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import relp_amd
from relp_oracle import FiniteOptimum, Infeasible, MatrixData, Unbounded, Variable, solve_relaxation


ZERO_BOUNDS = len(sys.argv) > 3 and sys.argv[3] == "zero"  # also variables fixed at zero (upper bound 0)


def make(rng):
    n = rng.randint(2, 12)
    counts = [rng.randint(0, 4), rng.randint(0, 3), rng.randint(0, 4), rng.randint(0, 3)]
    if sum(counts) < 2:
        counts[2] += 2
    m = sum(counts)
    density = rng.choice([0.3, 0.5, 0.8])
    dense = [[rng.choice([1, 2, 3, -1, -2, 5, 7, -4]) if rng.random() < density else 0 for _ in range(n)] for _ in range(m)]
    if rng.random() < 0.25 and m >= 2:
        dense[1] = list(dense[0])
    columns = [[(i, dense[i][j]) for i in range(m) if dense[i][j] != 0] for j in range(n)]
    b = [rng.randint(0, 15) for _ in range(m)]
    if rng.random() < 0.25 and m >= 2:
        b[1] = b[0]
    ranges = [rng.randint(1, 6) for _ in range(counts[1])]
    cost = [rng.randint(-6, 5) for _ in range(n)]
    upper = [rng.choice([None, None, rng.randint(1, 9), rng.randint(1, 3)] + ([0] if ZERO_BOUNDS else [])) for _ in range(n)]
    return n, counts, columns, b, ranges, cost, upper


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad = []
start = time.time()
for seed in range(first, first + count):
    rng = random.Random(900000 + seed)
    n, counts, columns, b, ranges, cost, upper = make(rng)
    data = MatrixData(columns, b, ranges, counts[0], counts[1], counts[2], counts[3],
                      [Variable(c, upper_bound=u) for c, u in zip(cost, upper)])
    try:
        expected = solve_relaxation(data)
    except AssertionError:
        continue
    column_start, rows, nums = [0], [], []
    for col in columns:
        for i, v in col:
            rows.append(i); nums.append(v)
        column_start.append(len(rows))
    CARRY = int(os.environ.get("RELP_FUZZ_CARRY", "0"))  # 1: the LU + Forrest-Tomlin carry (bound rows explicit only)
    for mode in ((0,) if CARRY else (0, 1)):
        solver = relp_amd.Solver(certify=1, implicit_bounds=mode, carry=CARRY)
        try:
            solver.load_matrix_data(column_start, rows or [0], nums or [0], [1] * max(1, len(nums)), b=b, cost=cost, upper=upper,
                                    ranges=ranges, counts=tuple(counts))
            result = solver.solve_relaxation()
            if isinstance(expected, Infeasible):
                ok = result.kind == relp_amd.INFEASIBLE
            elif isinstance(expected, Unbounded):
                ok = result.kind == relp_amd.UNBOUNDED
            else:
                objective = sum((Fraction(cost[j]) * v for j, v in data.reconstruct_solution(expected.solution)), Fraction(0))
                ok = (result.kind == relp_amd.FINITE_OPTIMUM and result.certified == 1
                      and solver.objective_exact() == "%d/%d" % (objective.numerator, objective.denominator))
        except relp_amd.RelpError as error:
            ok = False
            print("seed", seed, "mode", mode, "error", error)
        if not ok:
            bad.append((seed, mode))
            print("MISMATCH seed", seed, "mode", mode, "expected", type(expected).__name__, "got kind", getattr(result, "kind", None),
                  "certified", getattr(result, "certified", None), flush=True)
        solver.close()
print("checked %d seeds in %.1f s: %d mismatches %s" % (count, time.time() - start, len(bad), bad[:20]))
