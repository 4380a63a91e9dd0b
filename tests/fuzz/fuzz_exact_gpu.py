"""One-off fuzzing of the exact device path on the GPU box (round 6: the update's second pass inside the MFMA tiles, the double-buffered N):
random LPs with every row kind and rational data, larger than the suite's, started at 16 or 32 limbs so that every pivot runs on the matrix
cores -- verdict, whole pivot sequence and exact optimum against the Fraction oracle.  Prints the seeds that disagree.

    python tests/fuzz/fuzz_exact_gpu.py [first_seed] [count]
"""
import os
import random
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import relp_amd  # noqa: E402
from relp_oracle import FiniteOptimum, Infeasible, MatrixData, Unbounded, Variable, solve_relaxation  # noqa: E402


class Trace:
    def __init__(self):
        self.phase = 1
        self.pivots = []

    def record(self, q, p, leaving, cost):
        self.pivots.append((self.phase, q, p, leaving))


def one(seed):
    rng = random.Random(910000 + seed)
    n = rng.randint(6, 40)
    counts = [rng.randint(0, 8), 0, rng.randint(0, 14), rng.randint(0, 8)]
    if sum(counts) < 3:
        counts[2] += 3
    m = sum(counts)
    density = rng.choice([0.15, 0.3, 0.6])
    dense = [[rng.choice([1, 2, 3, -1, -2, 5, 7, -4, 11, 13]) if rng.random() < density else 0 for _ in range(n)] for _ in range(m)]
    for i in range(m):
        if not any(dense[i]):
            dense[i][rng.randrange(n)] = 1
    if rng.random() < 0.3 and m >= 2:  # a dependent row: zero-level pivots, a redundant row
        dense[1] = [2 * v for v in dense[0]]
    dens = [[rng.choice([1, 1, 1, 2, 3, 7]) for _ in range(n)] for _ in range(m)]
    columns = [[(i, Fraction(dense[i][j], dens[i][j])) for i in range(m) if dense[i][j]] for j in range(n)]
    b = [Fraction(rng.randint(0, 40), rng.choice([1, 1, 2, 3])) for _ in range(m)]
    if rng.random() < 0.3 and m >= 2:
        b[1] = 2 * b[0]
    cost = [Fraction(rng.randint(-9, 12), rng.choice([1, 1, 2, 5])) for _ in range(n)]
    data = MatrixData(columns, b, [], counts[0], counts[1], counts[2], counts[3], [Variable(c) for c in cost])
    trace = Trace()
    try:
        exact = solve_relaxation(data, trace=trace)
    except IndexError:
        return "skip"
    column_start, rows, nums, dnms = [0], [], [], []
    for column in columns:
        for i, v in column:
            rows.append(i)
            nums.append(v.numerator)
            dnms.append(v.denominator)
        column_start.append(len(rows))
    first = rng.choice([16, 32])
    problems = []
    for mode in (0, 4):
        solver = relp_amd.Solver(exact_update=mode)
        solver.load_matrix_data(column_start, rows, nums, dnms, b=[(v.numerator, v.denominator) for v in b],
                                cost=[(v.numerator, v.denominator) for v in cost], counts=tuple(counts))
        got = solver.solve_exact(first_limbs=first, max_limbs=128)
        n_art = solver.n_art
        want = [(ph, q + (n_art if ph == 2 else 0), p, lv + (n_art if ph == 2 else 0)) for ph, q, p, lv in trace.pivots]
        if isinstance(exact, FiniteOptimum):
            objective = sum((cost[j] * v for j, v in data.reconstruct_solution(exact.solution)), Fraction(0))
            ok = got["status"] == 1 and Fraction(got["objective"]) == objective and got["trace"] == want
        elif isinstance(exact, Infeasible):
            ok = got["status"] == 2 and got["trace"] == [t for t in want if t[0] == 1]
        else:
            ok = isinstance(exact, Unbounded) and got["status"] == 3 and got["trace"] == want
        if not ok:
            problems.append("mode %d from %d limbs: status %d, %d pivots against %d" % (mode, first, got["status"], len(got["trace"]), len(want)))
        solver.close()
    return problems or "ok"


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    start = time.time()
    tally = {"ok": 0, "skip": 0, "bad": 0}
    for seed in range(first, first + count):
        result = one(seed)
        if result in ("ok", "skip"):
            tally[result] += 1
        else:
            tally["bad"] += 1
            print("seed", seed, result, flush=True)
    print("seeds %d..%d: %s in %.0f s" % (first, first + count - 1, tally, time.time() - start))


if __name__ == "__main__":
    main()
