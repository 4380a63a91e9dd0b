"""bench.py -- simplex pivots/sec on Netlib 25FV47 (BASELINE.json configs[1]) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 25fv47]

One "step" = one complete ``solve_relaxation`` of the workload, LP resident in HBM when the timed region starts
(MPS parsing, standardisation and the H2D upload happen before it).  ``value`` = pivots of all ranks / wall time.
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank solves its own copy -- independent LPs shard
one per GPU with no data-path collective (weak scaling); the barrier + max-over-ranks timing is the only exchange.

The JSON line also carries ``roofline`` (pricing kernel: algorithmic bytes / HIP-event time per launch vs 8 TB/s HBM)
and ``cpu_baseline`` (the exact-rational restatement of relp's own algorithm on one host core, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "25fv47": os.path.join(ROOT, "data", "netlib", "25FV47.SIF"),   # BASELINE configs[1]: the default, the metric's config
    "dense4096": (4096, 8192),                                          # BASELINE configs[2]: the HBM-roofline config
    "dense1024": (1024, 2048),
    "maxflow": ("maxflow", 65536, 1048576),                          # BASELINE configs[4]: 1 M-arc max-flow LP, implicit capacity bounds
    "maxflow64k": ("maxflow", 8192, 65536),
    "netlib": "batch",   # BASELINE configs[3]: the Netlib problems the reference's suite enables, one LP per GPU at a time
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def emit(text):
    """Print the ONE JSON line as the last line of stdout: RCCL writes a banner through C stdio, which is block buffered
    when stdout is a pipe and would otherwise be flushed at exit, after the line."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    print(text, flush=True)


def cpu_baseline(path, budget_seconds):
    """relp-equivalent exact CPU path (the oracle: kind "port"), first pivots of the same workload, one core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from relp_oracle import cpu
    from relp_oracle.mps import load_problem

    _, data = load_problem(path)
    record = cpu.solve_provider(data, max_seconds=budget_seconds, trace=0)  # oracle/cpp/relp_cpu.cpp, g++ -O2, one thread
    pivots = record["pivots_phase1"] + record["pivots_phase2"]
    elapsed = record["seconds"]
    full = ""
    measured = os.path.join(ROOT, "profiles", "r1_cpu_oracle_full_solve.json")
    if os.path.exists(measured) and path.endswith("25FV47.SIF"):
        g = json.load(open(measured))
        full = "; the full exact solve took %d pivots in %.0f s = %.2f pivots/s on %s" % (
            g["pivots"], g["seconds"], g["pivots"] / g["seconds"], g["host"])
    return {"value": pivots / elapsed if elapsed > 0 else 0.0, "unit": "pivots/s", "cores": 1, "kind": "port",
            "sample": "first %d pivots (%.1f s) of the same LP with exact rationals: C++ restatement of relp's "
                      "Carry<RationalBig, LUDecomposition> steepest-edge path (oracle/cpp, same pivot sequence as the "
                      "reference's algorithm; early pivots are the cheap ones, numbers grow to ~1800 bits%s)" % (pivots, elapsed, full)}


def cpu_baseline_f64(path, budget_seconds):
    """The SAME f64 algorithm on the CPU (oracle/f64_model.py: numpy twin of the device loop -- explicit inverse, steepest edge,
    Harris ratio test, Newton-Schulz polish), bounded sample; and, as context, a tuned CPU f64 simplex (HiGHS through scipy)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from f64_model import Model, Options
    from relp_oracle.mps import load_problem
    try:
        from threadpoolctl import threadpool_info
        threads = max([pool.get("num_threads", 1) for pool in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    _, data = load_problem(path)
    options = Options()
    options.max_seconds = budget_seconds
    model = Model(data, options)
    start = time.perf_counter()
    status = model.solve()
    elapsed = time.perf_counter() - start
    pivots = int(sum(model.iters))
    record = {"value": pivots / elapsed if elapsed > 0 else 0.0, "unit": "pivots/s", "cores": threads, "kind": "port",
              "sample": "%s of the same LP in f64 on the CPU: numpy restatement of the device algorithm (explicit inverse, steepest "
                        "edge, polish), BLAS on %d threads: %d pivots in %.1f s (%s)" % (
                            "the whole solve" if status == "optimal" else "the first pivots", threads, pivots, elapsed, status)}
    try:  # context only: a production CPU simplex with its own presolve and pivoting rules (a different algorithm)
        import numpy as np
        from scipy.optimize import linprog
        import relp_amd
        mdl = relp_amd.Model(path)
        m, n = mdl.nr_rows, mdl.nr_columns
        import scipy.sparse as sp
        rows, cols, vals = [], [], []
        for j in range(n):
            r, v = mdl.column(j)
            rows.extend(r.tolist())
            cols.extend([j] * len(r))
            vals.extend(v.tolist())
        a = sp.csc_matrix((vals, (rows, cols)), shape=(m, n))
        c = np.array([mdl.cost_value(j) for j in range(n)])
        t0 = time.perf_counter()
        res = linprog(c, A_eq=a, b_eq=mdl.right_hand_side(), bounds=(0, None), method="highs-ds", options={"presolve": True})
        seconds = time.perf_counter() - t0
        record["tuned_cpu_solver"] = {"name": "HiGHS dual simplex (scipy.optimize.linprog, method highs-ds, presolve on)",
                                      "seconds": seconds, "iterations": int(res.nit), "objective": float(res.fun) + mdl.fixed_cost(),
                                      "status": int(res.status)}
    except Exception as error:  # noqa: BLE001
        record["tuned_cpu_solver"] = {"error": str(error)}
    return record


def cpu_baseline_dense(dims, budget_seconds):
    """f64 CPU restatement of the same loop for the dense workloads (oracle/f64_dense.py, numpy + its threaded BLAS);
    the exact-rational path is infeasible at this size."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from f64_dense import DenseModel
    from relp_amd.workloads import dense_lp
    try:
        from threadpoolctl import threadpool_info
        threads = max([pool.get("num_threads", 1) for pool in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    model = DenseModel(*dense_lp(*dims))
    start = time.perf_counter()
    model.solve(max_seconds=budget_seconds)
    elapsed = time.perf_counter() - start
    return {"value": model.pivots / elapsed if elapsed > 0 else 0.0, "unit": "pivots/s", "cores": threads, "kind": "port",
            "sample": "first %d pivots (%.1f s) of the same dense LP in f64: numpy restatement of the same steepest-edge "
                      "explicit-inverse loop, BLAS on %d threads" % (model.pivots, elapsed, threads)}


def dense_roofline(device):
    """The pricing pass of BASELINE configs[2] (dense 4096 x 8192), timed with HIP events inside the pivot loop: the
    HBM-bound kernel of the path, reported beside the default workload's (latency-bound) figure."""
    import relp_amd
    from relp_amd.workloads import dense_lp
    a, b, c = dense_lp(4096, 8192)
    solver = relp_amd.Solver(device=device, polish_period=512).load_dense_le(a, b, c)
    solver.begin_phase_one()
    solver.iterate(300)
    seconds = solver.profile_kernel(0, 100)
    bytes_per_launch = solver.stats().price_bytes
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r2_dense4096_pmc_traffic.json")
    if os.path.exists(pmc):
        for name, entry in json.load(open(pmc)).items():
            if "price_dense_lane_kernel" in name:
                traffic = entry["hbm_bytes_corrected"]
    solver.close()
    achieved = bytes_per_launch / seconds / 1e9
    # The coefficients of this workload are integers in [1, 100]: the block is held as signed bytes (1 B per entry; 4 B as
    # float measured 131.4 MB in 25.1 us = 5.2 TB/s = 65 % of the HBM peak, 8 B as f64 262.7 MB in 45.9 us = 5.7 TB/s = 72 %).  With a quarter
    # of the bytes the pass is no longer bound by HBM but by f64 issue: 5 VALU instructions per entry (extract, convert, three
    # FMAs whose second operand comes through the DPP row broadcast), 1.9 ns per wave instruction per SIMD.
    valu_seconds = bytes_per_launch * 5 * 1.9e-9 / 64 / 1024
    return {"workload": "synthetic dense random LP m=4096 n=8192 (python bench.py --workload dense4096 for its pivots/s)",
            "bound": "hbm", "kernel": "price (dense block, 1 B per entry)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "seconds_per_launch": seconds,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "valu": {"instructions_per_entry": 5, "issue_seconds_per_launch": valu_seconds, "frac_of_launch": valu_seconds / seconds},
            "note": "the same pass with the block held as float streams 131.4 MB in 25.1 us (5.2 TB/s, frac 0.65 of HBM); as bytes "
                    "it streams 33 MB in less time, one column per lane with -pi / rho / w broadcast through DPP inside the f64 FMA "
                    "(no LDS traffic), and is bound by f64 issue and the launch's fixed latencies"}


def netlib_batch(args, rank, local_rank, world, distributed):
    """Config 4: independent LPs shard across ranks -- by default through a dynamic ticket queue over the cost-sorted list
    (relp_amd.batch.TicketQueue), or the static longest-first partition (relp_amd.batch.assign); every LP is resident
    in HBM before the timed region; value = pivots of all ranks / makespan (max over ranks)."""
    import glob
    import torch
    import relp_amd
    from relp_amd import batch
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
    names = sorted(n for n, e in expected.items() if os.path.exists(os.path.join(ROOT, "data", "netlib", n + ".SIF"))
                   and (not e["ignored"] or "intensive" in e["ignored"]))
    models = {}
    costs = []
    for name in names:
        model = relp_amd.Model(os.path.join(ROOT, "data", "netlib", name + ".SIF"), presolve=args.presolve)  # --presolve: the reference's harness order
        models[name] = model
        costs.append((name, float(model.nr_rows) * float(model.nnz + model.nr_columns)))
    dynamic = args.schedule == "dynamic"
    ordered = [name for name, _ in sorted(costs, key=lambda item: (-item[1], item[0]))]  # longest estimated first
    # dynamic: any rank may draw any LP, so every rank keeps the whole suite resident (< 2 GB of 288 GB HBM)
    mine = ordered if dynamic else batch.assign(costs, world)[rank]
    workers = max(1, args.concurrency)
    solvers = {name: relp_amd.Solver(device=local_rank).load_model(models[name]) for name in mine}
    import threading
    handle_locks = {name: threading.Lock() for name in solvers}  # a handle is single-threaded
    records = []
    lp_records = []   # one JSON object per solved LP (--records FILE)
    record_file = args.records
    passes = [0]
    # Order of the K x 45 tickets.  1-2 GPUs: pass after pass, each longest-first.  4+ GPUs: the cost-sorted list in chunks of
    # eight LPs, all K passes of a chunk before the next chunk -- close to longest-first over everything (the long solves of
    # the last pass do not start late) while consecutive tickets are still different LPs, so the host threads of one rank
    # do not queue on one handle.  (Tried at 1 GPU and dropped: longest-first over all passes, 1.07 s per pass against 0.95 s --
    # copies of the same long LP side by side on one GPU; one set of handles per host thread, 1.23 s.)
    chunk = 8 if world >= 4 else len(ordered)
    chunk = int(os.environ.get("RELP_BATCH_CHUNK", chunk))  # diagnostic override

    def ticket_to_index(ticket, repeat, count):
        chunk_index, within = divmod(ticket, chunk * repeat)
        size = min(chunk, count - chunk_index * chunk)  # the last chunk may be short
        if size <= 0:
            return None
        return chunk_index * chunk + within % size if within < size * repeat else None

    def run_all(repeat=1):
        """`repeat` passes over the suite as ONE queue of tickets (order: see `chunk` above): no rank or thread waits at a
        pass boundary.  `--concurrency K` keeps K LPs in flight on this GPU (K host threads, one stream each): the small LPs
        are latency bound and use a fraction of the chip, so their kernels overlap."""
        import threading
        passes[0] += 1
        def slots(count):  # tickets of a short last chunk that fall outside it are skipped by the workers
            return (count + chunk - 1) // chunk * chunk * repeat
        tickets = batch.TicketQueue(slots(len(ordered)), tag="pass%d" % passes[0]) if dynamic else None
        static_tickets = batch.TicketQueue(slots(len(mine)), tag="static%d" % passes[0]) if not dynamic else None
        if static_tickets is not None:
            static_tickets.store = None  # a rank-local counter over this rank's own share
        totals = []

        def worker():
            pivots = 0
            while True:
                index = tickets.next() if dynamic else static_tickets.next()
                if index is None:
                    break
                position = ticket_to_index(index, repeat, len(ordered) if dynamic else len(mine))
                if position is None:
                    continue
                name = ordered[position] if dynamic else mine[position]
                with handle_locks[name]:
                    r = solvers[name].solve_relaxation()
                    if record_file is not None:
                        lp_records.append(solvers[name].record())
                pivots += r.pivots_phase_one + r.pivots_phase_two
                records.append((name, r.objective, r.pivots_phase_one + r.pivots_phase_two, r.solve_seconds))
            totals.append(pivots)

        threads = [threading.Thread(target=worker) for _ in range(workers - 1)]
        for t in threads:
            t.start()
        worker()
        for t in threads:
            t.join()
        return sum(totals)

    for _ in range(args.warmup):
        run_all()
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    start = time.perf_counter()
    del records[:]
    pivots = run_all(repeat=args.steps)
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed, pivots = batch.aggregate(time.perf_counter() - start, pivots, device="cuda" if distributed else None)
    gathered = batch.gather_records(list(records))
    if rank == 0:
        wrong = []
        for rank_records in gathered:
            for name, objective, _, _ in rank_records:
                e = expected[name]
                tolerance = max(e["tolerance"], 2e-5 if name == "25FV47" else 0.0)
                if abs(objective - e["expected"]) > tolerance:
                    wrong.append(name)
        summary = json.dumps({
            "metric": "simplex pivots/sec, Netlib suite batched one LP per GPU", "value": pivots / elapsed, "unit": "pivots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "%d Netlib .SIF files shipped under data/netlib" % len(names),
            "config": {"workload": "Netlib batch (%d LPs%s), %s, independent LPs sharded over the GPUs" % (
                len(names), ", after the reference's presolve" if args.presolve else "",
                "dynamic ticket queue over the cost-sorted list" if dynamic else "static longest-first assignment"),
                       "lps_in_flight_per_gpu": max(1, args.concurrency),
                       "problems_per_rank": [len(r) for r in gathered],
                       "pivots_per_rank": [sum(entry[2] for entry in r) for r in gathered],
                       "solve_seconds_per_rank": [sum(entry[3] for entry in r) for r in gathered],
                       "makespan_s": elapsed, "objectives_outside_reference_tolerance": wrong}})
    if record_file is not None:
        with open(record_file if world == 1 else "%s.rank%d" % (record_file, rank), "w") as handle:
            for entry in lp_records:
                handle.write(json.dumps(entry) + "\n")
    if distributed:
        torch.distributed.destroy_process_group()
    if rank == 0:
        emit(summary)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=5)
    parser.add_argument("--warmup", type=int, default=1)
    parser.add_argument("--workload", default="25fv47")
    parser.add_argument("--crash", type=int, default=1, help="graph workloads: start phase one from the spanning-forest crash basis")
    parser.add_argument("--dense-storage", choices=["narrowest", "f32", "f64"], default="narrowest",
                        help="dense workloads: storage type of the dense block (narrowest exact type: signed bytes for this generator)")
    parser.add_argument("--cpu-seconds", type=float, default=15.0)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-concurrency-probe", action="store_true")
    parser.add_argument("--no-dense-roofline", action="store_true")
    parser.add_argument("--no-certify", action="store_true", help="25fv47: leave the exact certificate out of the timed step (A/B only)")
    parser.add_argument("--carry", type=int, default=0, choices=[0, 1], help="0 explicit inverse, 1 LU + Forrest-Tomlin (relp_options.carry)")
    parser.add_argument("--presolve", action="store_true", help="apply the reference's presolve before standardisation (its harness order)")
    parser.add_argument("--concurrency", type=int, default=4, help="netlib batch: LPs in flight per GPU")
    parser.add_argument("--records", default=None, help="netlib batch: write one JSON line per solved LP to this file")
    parser.add_argument("--schedule", default="dynamic", choices=["dynamic", "static"], help="netlib batch: work distribution")
    args = parser.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("RELP_FORCE_DISTRIBUTED") == "1"  # the env switch exercises the RCCL path at N=1
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    if world > 1:
        # the exact certificate assembles its digits on host threads (up to 32 per process): share the cores between the ranks
        os.environ.setdefault("RELP_CERTIFY_THREADS", str(max(2, (os.cpu_count() or 8) // world)))
    import relp_amd
    path = WORKLOADS[args.workload]
    if path == "batch":
        return netlib_batch(args, rank, local_rank, world, distributed)
    graph = isinstance(path, tuple) and path[0] == "maxflow"
    dense = not isinstance(path, str) and not graph
    if graph:
        # the MatrixProvider of examples/max_flow.rs on the random graph of SURVEY.md section 8(d); the capacity rows are
        # handled as implicit bounds (the explicit formulation has V - 2 + E rows: no inverse of that size fits)
        from relp_amd.workloads import max_flow_graph
        _, nr_vertices, nr_arcs = path
        tail, head, capacity = max_flow_graph(nr_vertices, nr_arcs)
        model = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
        solver = relp_amd.Solver(device=local_rank, implicit_bounds=1, crash=args.crash).load_model(model)
    elif dense:
        from relp_amd.workloads import dense_lp
        a, b, c = dense_lp(*path)
        if args.dense_storage != "narrowest":  # generic data: the block as float / double (the library reads this at load time)
            os.environ["RELP_DENSE_F32" if args.dense_storage == "f32" else "RELP_DENSE_F64"] = "1"
        solver = relp_amd.Solver(device=local_rank, polish_period=int(os.environ.get("RELP_POLISH", "512"))).load_dense_le(a, b, c)
    else:
        # the step is `solve_relaxation` with the exact certificate INSIDE: the f64 loop alone is narrower arithmetic than the
        # reference's, the bit-exact optimum is part of the job (BASELINE.json north_star)
        solver = relp_amd.Solver(device=local_rank, certify=0 if args.no_certify else 1, carry=args.carry).load_mps(path, presolve=args.presolve)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.solve_relaxation()
    barrier()
    start = time.perf_counter()
    pivots = 0
    last = None
    loop_seconds = certify_seconds = 0.0
    all_certified = True
    for _ in range(args.steps):
        last = solver.solve_relaxation()
        pivots += last.pivots_phase_one + last.pivots_phase_two
        loop_seconds += last.solve_seconds
        certify_seconds += last.certify_seconds
        all_certified = all_certified and bool(last.certified)
    barrier()
    elapsed = time.perf_counter() - start
    from relp_amd import batch
    # one record per rank (what each GPU did); the makespan is the max over ranks of the barrier-to-barrier time
    per_rank = batch.gather_records({"rank": rank, "solves": args.steps, "pivots": int(pivots), "busy_seconds": loop_seconds + certify_seconds,
                                     "elapsed_seconds": elapsed})
    elapsed, pivots = batch.aggregate(elapsed, pivots, device="cuda" if distributed else None)

    in_flight = None
    if rank == 0 and world == 1 and not dense and not graph and not args.no_concurrency_probe:
        # headroom: the same LP, 4 independent copies in flight on this GPU (one host thread and stream each); a single
        # latency-bound solve uses a fraction of the chip.  Reported beside `value`, never as `value`.
        import threading
        copies = [solver] + [relp_amd.Solver(device=local_rank, certify=0 if args.no_certify else 1, carry=args.carry).load_mps(path, presolve=args.presolve) for _ in range(3)]
        for extra in copies[1:]:
            extra.solve_relaxation()
        counts = [0] * len(copies)

        def run(k):
            for _ in range(args.steps):
                r = copies[k].solve_relaxation()
                counts[k] += r.pivots_phase_one + r.pivots_phase_two

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        threads = [threading.Thread(target=run, args=(k,)) for k in range(1, len(copies))]
        for t in threads:
            t.start()
        run(0)
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        in_flight = {"copies": len(copies), "pivots_per_s": sum(counts) / (time.perf_counter() - t0)}
        for extra in copies[1:]:
            extra.close()

    exact = None
    if rank == 0 and not dense and not graph and not args.no_certify:
        # the exact optimum of the LAST TIMED solve (the certificate ran inside every timed step)
        if all_certified:
            text = solver.objective_exact()
            num, den = text.split("/")
            exact = {"certified": True, "objective_bits": max(int(num).bit_length(), int(den).bit_length()),
                     "certify_seconds_per_solve": certify_seconds / args.steps, "objective_exact_head": text[:40] + "..."}
        else:
            exact = {"certified": False}
    if rank == 0:
        # roofline of the dominant kernel (pricing pass), measured live with HIP events on the solver's stream
        solver.begin_phase_one()
        # (phase one of the flow LP has only a few hundred ordinary pivots before the zero-level ones: short samples there)
        _, reason = solver.iterate(20 if graph else (300 if dense else 200))
        if graph and reason == relp_amd.STOP_NO_ENTERING:  # crash basis: phase one ends without a pivot -- profile phase two
            solver.begin_phase_two()
            solver.iterate(20)
        reps = 40 if graph else (100 if dense else 200)  # further real pivots, the profiled kernel of each bracketed by its own event pair
        solver.profile_kernel(0, 10 if graph else 50)   # discarded: brings clocks and caches to the state of a running solve
        lu_carry = args.carry == 1 and not dense and not graph
        kernels = ["price", "lu_pivot"] if lu_carry else ["price", "ftran_ratio", "update"]
        try:
            seconds = {name: solver.profile_kernel(which, reps) for which, name in enumerate(kernels)}
        except relp_amd.api.RelpError:
            # m <= 1024: ratio test and inverse update are ONE launch (pivot_fused_kernel): two kernels per pivot
            kernels = ["price", "pivot_fused"]
            seconds = {name: solver.profile_kernel(which, reps) for which, name in enumerate(kernels)}
        stats = solver.stats()
        # Algorithmic bytes per launch: pricing = the non-basic columns (exact, counted at the profiled
        # state); K2 = nnz(a_q) columns of the inverse + six m-vectors; K3 = read + write of the touched part of the inverse
        # (upper bound: all of it); the LU kernel = both orientations of the factors once each + twelve m-vectors.
        m_rows = solver.m
        mean_column = 2.0 if graph else (max(1.0, float(relp_amd.Model(path).nnz) / max(1, solver.n_provider)) if not dense else float(m_rows))
        algorithmic = {"price": stats.price_bytes,
                       "ftran_ratio": int(mean_column * m_rows * 8 + 6 * m_rows * 8),
                       "update": stats.update_bytes,
                       "lu_pivot": int(12 * m_rows * 8 + 2 * 12 * 3 * m_rows),  # (factor entries: about 3 per row and triangle)
                       # fused: one workgroup's FTRAN + ratio test, and ONE read + one write of the whole inverse (out of place)
                       "pivot_fused": int(mean_column * m_rows * 8 + 6 * m_rows * 8 + 2 * m_rows * m_rows * 8)}
        per_kernel = {name: {"seconds_per_launch": seconds[name], "algorithmic_bytes_per_launch": algorithmic[name],
                             "achieved_gb_s": algorithmic[name] / seconds[name] / 1e9,
                             "frac": algorithmic[name] / seconds[name] / 1e9 / HBM_PEAK_GBS,
                             "share_of_pivot_time": seconds[name] / sum(seconds.values())} for name in kernels}
        dominant = max(kernels, key=lambda name: seconds[name])  # by MEASURED time share, not by assumption
        bytes_per_launch = algorithmic[dominant]
        achieved = bytes_per_launch / seconds[dominant] / 1e9
        # HBM traffic per launch from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE on gfx950, MI355X_MICROARCH.md
        # section HBM); null when not collected for this kernel
        traffic = None
        dense_price = "price_dense_lane_kernel" if args.dense_storage == "narrowest" else "price_dense_kernel"
        pmc_names = {"price": dense_price if dense else "relp::price_kernel<", "ftran_ratio": "ftran_ratio", "update": "update_kernel",
                     "lu_pivot": "lu_pivot_kernel", "pivot_fused": "pivot_fused_kernel"}
        for candidate in ("r2_%s_pmc_traffic.json" % args.workload, "r1_%s_pmc_traffic.json" % args.workload):
            pmc = os.path.join(ROOT, "profiles", candidate)
            if traffic is None and os.path.exists(pmc):
                for name, entry in json.load(open(pmc)).items():
                    if pmc_names[dominant] in name:
                        traffic = entry["hbm_bytes_corrected"]
                        break
        if graph:
            workload = ("max-flow LP (examples/max_flow.rs provider) on a random graph V=%d E=%d (splitmix64 seed 0x5EED0005): "
                        "%d conservation rows on the device, the %d capacity rows as implicit bounds, %s" % (
                            path[1], path[2], path[1] - 2, path[2],
                            "phase one from the spanning-forest crash basis" if args.crash else "artificial start as in the reference"))
            data = "synthetic"
        elif dense:
            workload = "synthetic dense random LP m=%d n=%d f64 (splitmix64 seed 0x5EED0001), steepest-edge pricing, dense block stored as %s" % (
                path + ({"narrowest": "signed bytes (narrowest exact type)", "f32": "float", "f64": "double"}[args.dense_storage],))
            data = "synthetic"
        else:
            workload = ("Netlib 25FV47 %dx%d, steepest-edge pricing, %s carry, exact certificate %s the timed step, %s" % (
                solver.m, solver.n_provider, "LU + Forrest-Tomlin" if args.carry == 1 else "explicit-inverse",
                "outside" if args.no_certify else "inside",
                "after the reference's presolve" if args.presolve else "no presolve (+520 virtual artificials)"))
            data = "Netlib 25FV47.SIF (shipped problem file), one copy per GPU"
        line = {
            "metric": "simplex pivots/sec + wall-clock to optimal, Netlib 25fv47 @1 GPU",
            "value": pivots / elapsed, "unit": "pivots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": data,
            "config": {"workload": workload,
                       "pivots_per_solve": int(last.pivots_phase_one + last.pivots_phase_two),
                       "objective": last.objective,
                       "wall_clock_to_exact_optimum_s": (loop_seconds + certify_seconds) / args.steps,
                       "wall_clock_f64_loop_s": loop_seconds / args.steps,
                       "pivots_per_s_f64_loop_only": pivots / world / loop_seconds if loop_seconds > 0 else None,
                       "carry": "lu" if args.carry == 1 else "explicit", "refactors": int(last.refactors),
                       "polishes": int(last.polishes), "max_residual_before_polish": last.max_residual,
                       "parallelism": "1 LP per GPU x%d" % world,
                       "makespan_s": elapsed, "per_rank": sorted(per_rank, key=lambda r: r["rank"]), "exact": exact,
                       "aggregate_with_copies_in_flight": in_flight},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "seconds_per_launch": seconds, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernels": per_kernel},
        }
        if graph:
            line["metric"] = "simplex pivots/sec + wall-clock to optimal, max-flow LP @1 GPU"
            line["roofline"]["note"] = ("the dominant kernel by measured time is listed first; 'price' generates each incidence column from the 8 bytes of "
                                        "its arc's endpoints (+ 1 B cost, 4 B basis position; a steepest-edge weight is read/written only by candidates "
                                        "and by columns with an entry in the pivot row) and gathers one packed 32-byte (-pi, rho, w) record per entry from "
                                        "L2; 'update' walks the non-zero rows of alpha in every column of the inverse that is not a unit vector (8-byte "
                                        "gathers, one cache line each: PMC traffic is line-granular)")
        elif dense:
            line["metric"] = "simplex pivots/sec + wall-clock to optimal, dense LP @1 GPU"
            if args.dense_storage == "narrowest":
                line["roofline"]["note"] = ("dense block held in the narrowest exact type (this workload: signed bytes, 1 B per entry; as float the "
                                            "pass streams 4x the bytes at 5.2 TB/s = 0.65 of the HBM peak): one column per lane, -pi / rho / w "
                                            "broadcast through DPP inside the f64 FMA (no LDS traffic); 5 VALU instructions per entry = %.1f us "
                                            "of issue time on 1024 SIMDs (tools/micro/valu_rates.hip: 1.9 ns each), the rest of the launch is "
                                            "the first load's latency and the per-workgroup tail" % (bytes_per_launch * 5 * 1.9e-3 / 64 / 1024))
            else:
                line["roofline"]["note"] = {
                    "f32": "dense block streamed as float (exact for this data; all arithmetic f64): one column per lane, 16-byte "
                           "non-temporal loads, -pi / rho / w broadcast through DPP inside the f64 FMA (no LDS traffic); bound by HBM",
                    "f64": "dense block streamed as double: one column per lane, 16-byte non-temporal loads, -pi / rho / w broadcast "
                           "through DPP inside the f64 FMA (no LDS traffic); bound by HBM"}[args.dense_storage]
        if not dense and not graph:
            line["roofline"]["note"] = ("latency bound by construction: the dominant kernel BY MEASURED TIME is '%s' (%d KB of algorithmic "
                                        "bytes per launch, all of it resident in L2 / Infinity Cache; SURVEY.md section 8(d)); every kernel of "
                                        "the pivot is listed under 'kernels' with its share of the pivot time; traffic = 2 x FETCH_SIZE + "
                                        "WRITE_SIZE of the committed PMC passes for that kernel; the HBM-roofline configuration is BASELINE "
                                        "configs[2], measured below") % (dominant, bytes_per_launch // 1024)
            if world == 1 and not args.no_dense_roofline:
                line["roofline_config3"] = dense_roofline(local_rank)
        if world == 1 and not args.no_cpu_baseline and not graph:  # reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline_dense(path, args.cpu_seconds) if dense else cpu_baseline(path, args.cpu_seconds)
            if not dense:
                line["cpu_baseline_f64"] = cpu_baseline_f64(path, args.cpu_seconds)
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        emit(json.dumps(line))


if __name__ == "__main__":
    main()
