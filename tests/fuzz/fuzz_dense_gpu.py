"""One-off fuzzing of the dense pipeline on the GPU box: random sizes (rows not multiples of the 64-row tile, columns not multiples
of the 16-column group, odd m = no deferred product form), the three storage types of the dense block, the multi-block FTRAN /
deferred-product-form pipeline forced on small LPs -- against the numpy f64 restatement (objective within 1e-9 relative).

    python tests/fuzz/fuzz_dense_gpu.py [first_seed] [count]
"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import relp_amd
from f64_dense import DenseModel
from relp_amd.workloads import dense_lp

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = []
start = time.time()
for seed in range(first, first + count):
    rng = random.Random(770000 + seed)
    m = rng.randint(64, 700)
    n = rng.randint(max(64, m // 2), 2 * m + 64)
    storage = rng.choice(["bytes", "f32", "f64"])
    pipeline = rng.random() < 0.7
    for name in ("RELP_DENSE_F32", "RELP_DENSE_F64", "RELP_FTRAN_MIN_NNZ"):
        os.environ.pop(name, None)
    if storage != "bytes":
        os.environ["RELP_DENSE_" + storage.upper()] = "1"
    if pipeline:
        os.environ["RELP_FTRAN_MIN_NNZ"] = "16"
    a, b, c = dense_lp(m, n, seed=0x5EED1000 + seed)
    model = DenseModel(a, b, c)
    status = model.solve()
    solver = relp_amd.Solver(polish_period=64).load_dense_le(a, b, c)
    result = solver.solve_relaxation()
    ok = status == "optimal" and result.kind == relp_amd.FINITE_OPTIMUM and abs(result.objective - model.objective()) <= 1e-9 * abs(model.objective())
    if not ok:
        bad.append(seed)
        print("MISMATCH seed", seed, m, n, storage, pipeline, status, result.kind, result.objective, model.objective() if status == "optimal" else None, flush=True)
    solver.close()
print("checked %d seeds in %.1f s: %d mismatches %s" % (count, time.time() - start, len(bad), bad[:20]))
