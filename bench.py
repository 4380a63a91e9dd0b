"""bench.py -- simplex pivots/sec on Netlib 25FV47 (BASELINE.json configs[1]) on MI355X, and every other BASELINE config beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 25fv47]

One "step" = one complete ``solve_relaxation`` of the workload, LP resident in HBM when the timed region starts
(MPS parsing, standardisation and the H2D upload happen before it).  ``value`` = pivots of all ranks / wall time.
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank solves its own copy -- independent LPs shard
one per GPU with no data-path collective (weak scaling); the barrier + max-over-ranks timing is the only exchange.

The JSON line carries ``roofline`` (the dominant kernel by measured time: SURVEY.md section 8(d)'s algorithmic bytes of that
kernel / its HIP-event time per launch vs 8 TB/s HBM; the kernel's own byte count beside it) and ``cpu_baseline`` (the
exact-rational restatement of relp's own algorithm on one host core, bounded sample).  The default run (N = 1) also measures
the other BASELINE configs in the same process and reports them under ``configs``: the LU carry on 25FV47, the dense LP of
config 3 with the block stored as double and in the narrowest exact type, the Netlib batch of config 4 without and with the
reference's presolve, and the max-flow LP of config 5 from the reference's artificial start and from the crash basis --
each with its own ``value``, ``ms_per_step``, ``roofline`` and ``cpu_baseline``.  CPU legs run as child processes, started after
the headline's timed region (one core each; the all-cores Netlib leg runs alone at the end).

Output: the LAST stdout line is ONE COMPACT JSON object (< 4 KB: the driver's capture is bounded) with every contract field, the
roofline of the dominant kernel, the CPU baselines, ``value_lu_carry`` / ``value_lu_inverse_carry`` (the same LP and step under the
LU carries: BASELINE configs[1] as written) and ``configs_summary`` {name: [value, ms_per_step, roofline.frac]}; the full record --
per-kernel tables, every config object, CPU samples, per-rank records -- is written to ``bench_configs.json`` (``--detail``).
"""
import argparse
import copy
import json
import os
import subprocess
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
I8_MFMA_PEAK_TMACS = 2500.0     # dense i8 MFMA: 2x the bf16 rate (same guide, matrix cores table) = 5 POP/s = 2.5 P MAC/s; measured 2.39 here
VALU_WORD_PRODUCT_PEAK = 2.1    # T word products/s: the 4 x 4-word block product of exact.hip on the vector multiplier, operands in registers (measured)


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


def _round(value, digits=6):
    """Floats of the compact line: six significant digits (the full precision is in the detail file)."""
    if isinstance(value, float):
        return float("%.*g" % (digits, value))
    if isinstance(value, dict):
        return {k: _round(v, 12 if k == "objective" else digits) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_round(v, digits) for v in value]
    return value


def _short(text, limit):
    text = "" if text is None else str(text)
    return text if len(text) <= limit else text[:limit - 3] + "..."


def compact_cpu(record, sample_limit=200):
    """A CPU baseline as it appears in the compact line."""
    if not isinstance(record, dict):
        return None
    if "error" in record:
        return {"error": _short(record["error"], 120)}
    out = {k: record.get(k) for k in ("value", "unit", "cores", "kind", "mode", "cpu_model", "nproc") if k in record}
    out["sample"] = _short(record.get("sample"), sample_limit)
    return out


def summary_triple(entry):
    """[value, ms_per_step, roofline.frac] of one config of the detail file."""
    if not isinstance(entry, dict) or "error" in entry:
        return {"error": _short((entry or {}).get("error"), 80)}
    return [entry.get("value"), entry.get("ms_per_step"), (entry.get("roofline") or {}).get("frac")]


COMPACT_LIMIT = 4000  # bytes: the driver keeps the last 8 KB of stdout; the LAST line must parse on its own from its final 4 KB


def compact_line(line, detail_name):
    """The ONE line the driver parses: every contract field, the roofline of the dominant kernel and the CPU baseline in
    short form, the other BASELINE configs as `configs_summary` {name: [value, ms_per_step, roofline.frac]} and the LU carries'
    figures on the same LP at top level.  Everything else (per-kernel tables, per-rank records, notes, the full config
    objects) goes to `detail_name` (written next to bench.py)."""
    config, roofline = line.get("config", {}), line.get("roofline")
    exact = config.get("exact") or {}
    out = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    keep = ("carry", "pivots_per_solve", "objective", "wall_clock_to_exact_optimum_s", "wall_clock_f64_loop_s", "refactors",
            "refactor_seconds_per_solve", "parallelism", "makespan_s", "throughput_kind", "lps_in_flight_per_gpu", "single_pass_makespan_s",
            "longest_lp", "tickets_per_rank", "pivots_per_rank", "objectives_outside_reference_tolerance", "carry_per_lp")
    out["config"] = {"workload": _short(config.get("workload"), 260)}
    out["config"].update({k: config[k] for k in keep if k in config and config[k] is not None})
    if exact:
        out["config"]["certified"] = bool(exact.get("certified"))
        if "objective_bits" in exact:
            out["config"]["objective_bits"] = exact["objective_bits"]
    if "ratio_rule" in config:
        out["config"]["ratio_rule"] = _short(config["ratio_rule"], 40)
    if config.get("aggregate_with_copies_in_flight"):
        out["config"]["pivots_per_s_4_copies_in_flight"] = config["aggregate_with_copies_in_flight"]["pivots_per_s"]
    if roofline:
        seconds = roofline.get("seconds_per_launch")
        if isinstance(seconds, dict):
            seconds = seconds.get(roofline.get("kernel"))
        out["roofline"] = {k: roofline.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                        "algorithmic_bytes_per_launch", "kernel_bytes_per_launch") if k in roofline}
        out["roofline"]["seconds_per_launch"] = seconds
        per_pivot = roofline.get("per_pivot") or {}
        if per_pivot.get("contract_frac") is not None:
            out["roofline"]["per_pivot_contract_frac"] = per_pivot["contract_frac"]
    for key in ("cpu_baseline", "cpu_baseline_tuned"):
        if key in line:
            out[key] = compact_cpu(line[key], 220 if key == "cpu_baseline" else 60)
    f64 = line.get("cpu_baseline_f64")
    if isinstance(f64, dict) and "error" not in f64:
        out["cpu_baseline_f64_port"] = {"value": f64.get("value"), "unit": f64.get("unit"), "cores": f64.get("cores"),
                                        "what": "numpy twin of the device's f64 loop, whole solve"}
        tuned = f64.get("tuned_cpu_solver") or {}
        if "seconds" in tuned:
            gpu_seconds = config.get("wall_clock_to_exact_optimum_s")
            out["cpu_baseline_f64_tuned"] = {"name": "HiGHS dual simplex (scipy highs-ds, presolve on), same LP", "seconds": tuned["seconds"],
                                             "iterations": tuned.get("iterations"), "cores": 1,
                                             "gpu_seconds_to_exact_optimum": gpu_seconds,
                                             "cpu_over_gpu": tuned["seconds"] / gpu_seconds if gpu_seconds else None}
    for key in ("value_lu_carry", "value_lu_inverse_carry", "value_lu_inverse_carry_device_refactor", "same_work_exact", "same_work_exact_25fv47"):
        if key in line:
            out[key] = line[key]
    if "configs" in line:
        out["configs_summary"] = {name: summary_triple(entry) for name, entry in line["configs"].items()}
        out["configs_summary_fields"] = ["value (pivots/s)", "ms_per_step", "roofline.frac"]
    if "host" in line:
        out["host"] = line["host"]
    out["detail_file"] = detail_name
    out = _round(out)
    if out.get("roofline") and out["roofline"].get("peak") and out["roofline"].get("achieved") is not None:
        out["roofline"]["frac"] = out["roofline"]["achieved"] / out["roofline"]["peak"]  # consistent after the rounding: frac IS achieved / peak
    text = json.dumps(out, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:  # never let the line outgrow the driver's capture again: drop the prose first, then the summaries
        for victim in (("cpu_baseline", "sample"), ("cpu_baseline_tuned", None), ("config", "workload"), ("configs_summary", None),
                       ("configs_summary_fields", None), ("cpu_baseline_f64_tuned", None), ("cpu_baseline_f64_port", None),
                       ("same_work_exact", None), ("same_work_exact_25fv47", None), ("host", None)):
            if victim[1] is None:
                out.pop(victim[0], None)
            elif isinstance(out.get(victim[0]), dict):
                out[victim[0]][victim[1]] = _short(out[victim[0]].get(victim[1]), 60)
            text = json.dumps(out, separators=(",", ":"))
            if len(text) <= COMPACT_LIMIT:
                break
    if len(text) > COMPACT_LIMIT:  # last resort: the contract's fields and nothing else
        contract = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                            "vs_baseline", "dtype", "data", "detail_file")}
        contract["metric"] = _short(contract["metric"], 200)
        contract["data"] = _short(contract["data"], 200)
        contract["config"] = {"workload": _short((out.get("config") or {}).get("workload"), 60)}
        for key in ("roofline", "cpu_baseline"):
            if isinstance(out.get(key), dict):
                contract[key] = {k: (v if not isinstance(v, str) else _short(v, 60)) for k, v in out[key].items() if not isinstance(v, (dict, list))}
        text = json.dumps(contract, separators=(",", ":"))
    assert len(text) <= COMPACT_LIMIT, len(text)
    return text


def write_detail(line, path):
    """The full record (what round 3 printed as one 35 KB line): every config with its per-kernel roofline table, CPU legs with
    their samples, per-rank records."""
    try:
        with open(path, "w") as handle:
            json.dump(line, handle, indent=1)
            handle.write("\n")
    except OSError as error:  # a read-only checkout must not take the bench line down
        sys.stderr.write("bench.py: cannot write %s: %s\n" % (path, error))


def host_description():
    """CPU model and core count of the bench host (stated in every report)."""
    model = "unknown"
    try:
        for row in open("/proc/cpuinfo"):
            if row.startswith("model name"):
                model = row.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count() or 1}


# =====================================================================================================================
# CPU legs (child processes: `python bench.py --cpu-leg NAME --cpu-seconds S`; each prints one JSON object)
# =====================================================================================================================
def cpu_leg_exact(path, budget_seconds, tuned):
    """relp-equivalent exact CPU path (the oracle: kind "port"), first pivots of the same workload, one core.  Faithful mode
    keeps the reference's data structures and asymptotics; `tuned` only adds a row index to the BTRAN scans."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from relp_oracle import cpu
    from relp_oracle.mps import load_problem

    _, data = load_problem(path)
    record = cpu.solve_provider(data, max_seconds=budget_seconds, trace=0, tuned=tuned)  # oracle/cpp/relp_cpu.cpp, g++ -O2, one thread
    pivots = record["pivots_phase1"] + record["pivots_phase2"]
    elapsed = record["seconds"]
    full = ""
    measured = os.path.join(ROOT, "profiles", "r1_cpu_oracle_full_solve.json")
    if os.path.exists(measured) and path.endswith("25FV47.SIF") and not tuned:
        g = json.load(open(measured))
        full = "; the full exact solve took %d pivots in %.0f s = %.2f pivots/s on %s" % (
            g["pivots"], g["seconds"], g["pivots"] / g["seconds"], g["host"])
    return {"value": pivots / elapsed if elapsed > 0 else 0.0, "unit": "pivots/s", "cores": 1, "kind": "port", "mode": "tuned" if tuned else "faithful",
            "sample": "first %d pivots (%.1f s) of the same LP with exact rationals: C++ restatement of relp's "
                      "Carry<RationalBig, LUDecomposition> steepest-edge path (oracle/cpp%s, same pivot sequence as the "
                      "reference's algorithm; early pivots are the cheap ones, numbers grow to ~1800 bits%s)" % (
                          pivots, elapsed, " --tuned: row index for the BTRAN scans" if tuned else ", faithful data structures", full)}


def cpu_leg_f64(path, budget_seconds):
    """The SAME f64 algorithm on the CPU (oracle/f64_model.py: numpy twin of the device loop -- explicit inverse, steepest edge,
    Harris ratio test, Newton-Schulz polish), bounded sample; and, as context, a tuned CPU f64 simplex (HiGHS through scipy)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from f64_model import Model, Options
    from relp_oracle.mps import load_problem
    threads = blas_threads()
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


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([pool.get("num_threads", 1) for pool in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        return os.cpu_count() or 1


def cpu_leg_dense(dims, budget_seconds):
    """f64 CPU restatement of the same loop for the dense workloads (oracle/f64_dense.py, numpy + its threaded BLAS);
    the exact-rational path is infeasible at this size."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from f64_dense import DenseModel
    from relp_amd.workloads import dense_lp
    import numpy  # noqa: F401  (so that the BLAS pool exists when its size is read)
    threads = blas_threads()
    model = DenseModel(*dense_lp(*dims))
    start = time.perf_counter()
    model.solve(max_seconds=budget_seconds)
    elapsed = time.perf_counter() - start
    return {"value": model.pivots / elapsed if elapsed > 0 else 0.0, "unit": "pivots/s", "cores": threads, "kind": "port",
            "sample": "first %d pivots (%.1f s) of the same dense LP in f64: numpy restatement of the same steepest-edge "
                      "explicit-inverse loop, BLAS on %d threads" % (model.pivots, elapsed, threads)}


def netlib_names(expected):
    return sorted(n for n, e in expected.items() if os.path.exists(os.path.join(ROOT, "data", "netlib", n + ".SIF"))
                  and (not e["ignored"] or "intensive" in e["ignored"]))


def _netlib_one(job):
    name, budget = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from relp_oracle import cpu
    from relp_oracle.mps import load_problem
    _, data = load_problem(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
    t0 = time.perf_counter()
    record = cpu.solve_provider(data, max_seconds=budget, trace=0)
    return name, record["pivots_phase1"] + record["pivots_phase2"], record["seconds"], time.perf_counter() - t0, record.get("status", "")


def cpu_leg_netlib(budget_seconds):
    """Config 4 on the host: the exact C++ restatement, ONE LP PER CORE over all host cores (the reference itself is single-threaded;
    this is what its harness could do with a process per LP), every LP bounded to `budget_seconds` of solve time -- the long
    ones (25FV47: ~1000 s to the optimum) are sampled over their first pivots.  value = pivots of all LPs / wall time of the pool."""
    from concurrent.futures import ProcessPoolExecutor
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
    names = netlib_names(expected)
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, len(names)))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from relp_oracle import cpu
    cpu.ensure_built()
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=workers) as pool:
        done = list(pool.map(_netlib_one, [(n, budget_seconds) for n in names]))
    wall = time.perf_counter() - t0
    pivots = sum(d[1] for d in done)
    solve_seconds = sum(d[2] for d in done)
    finished = sum(1 for d in done if d[4] in ("optimal", "infeasible", "unbounded"))
    return {"value": pivots / wall if wall > 0 else 0.0, "unit": "pivots/s", "cores": workers, "kind": "port",
            "sample": "the %d LPs of the batch with exact rationals (oracle/cpp, faithful), one LP per core on %d cores, each bounded to %.0f s: "
                      "%d pivots in %.1f s of wall time (%.1f core-seconds of solving; %d LPs reached their optimum inside the bound, the rest "
                      "are sampled over their first pivots, the cheap ones)" % (len(names), workers, budget_seconds, pivots, wall, solve_seconds, finished)}


def cpu_leg_maxflow(budget_seconds):
    """Config 5 on the host: scipy's max-flow (a combinatorial algorithm on the same graph: the value the LP must reach) at the
    full size, and the exact C++ restatement of relp's path on a 16 k-arc twin of the LP (the provider of examples/max_flow.rs)."""
    import numpy as np
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_flow
    from relp_amd.workloads import max_flow_graph
    out = {}
    tail, head, capacity = max_flow_graph(65536, 1048576)
    keep = (head != 0) & (tail != 65535)
    t0 = time.perf_counter()
    flow = maximum_flow(csr_matrix((capacity[keep].astype(np.int32), (tail[keep], head[keep])), shape=(65536, 65536)), 0, 65535).flow_value
    out["scipy_max_flow"] = {"seconds": time.perf_counter() - t0, "flow_value": int(flow), "cores": 1,
                             "name": "scipy.sparse.csgraph.maximum_flow on the same 1 M-arc graph (combinatorial, not an LP solve)"}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from fractions import Fraction
    from relp_oracle import cpu
    from relp_oracle.network import MaxFlowPrimal
    twin_v, twin_e = 2048, 16384
    tail, head, capacity = max_flow_graph(twin_v, twin_e)
    keep = (head != 0) & (tail != twin_v - 1)
    arcs = [[] for _ in range(twin_v)]
    for u, v, c in zip(tail[keep].tolist(), head[keep].tolist(), capacity[keep].tolist()):
        arcs[u].append((v, Fraction(c)))
    provider = MaxFlowPrimal(arcs, 0, twin_v - 1)
    record = cpu.solve_provider(provider, max_seconds=budget_seconds, trace=0)
    pivots = record["pivots_phase1"] + record["pivots_phase2"]
    out.update({"value": pivots / record["seconds"] if record["seconds"] > 0 else 0.0, "unit": "pivots/s", "cores": 1, "kind": "port",
                "sample": "first %d pivots (%.1f s) of a 16 k-arc twin of the LP (V = %d, E = %d: %d rows in the reference's formulation; same generator) "
                          "with exact rationals, oracle/cpp faithful -- at 1 M arcs the reference's ordered-map LU does not finish a pivot in the budget" % (
                              pivots, record["seconds"], twin_v, int(keep.sum()), provider.nr_rows())})
    return out


def cpu_leg_exact_full(lp_name, budget_seconds):
    """The exact CPU restatement run TO COMPLETION on one mid-size LP: the same pivots the device's fixed-width exact simplex makes
    (both walk the reference's pivot sequence), so pivots/s on both sides is the same work."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from relp_oracle import cpu
    from relp_oracle.mps import load_problem
    _, data = load_problem(os.path.join(ROOT, "data", "netlib", lp_name + ".SIF"))
    record = cpu.solve_provider(data, max_seconds=budget_seconds, trace=0)
    pivots = record["pivots_phase1"] + record["pivots_phase2"]
    return {"value": pivots / record["seconds"] if record["seconds"] > 0 else 0.0, "unit": "pivots/s", "cores": 1, "kind": "port", "mode": "faithful",
            "pivots": pivots, "seconds": record["seconds"], "status": record.get("status", ""),
            "sample": "the WHOLE exact solve of %s (oracle/cpp, faithful): %d pivots in %.2f s (%s)" % (lp_name, pivots, record["seconds"], record.get("status", ""))}


def cpu_leg_exact_prefix(lp_name, budget_seconds):
    """The exact CPU restatement on the metric's own LP for `budget_seconds` on THIS host (or to the optimum if that comes first): the
    pivot count P it reached is what the device's exact path is then timed on (`relp_solve_exact(max_pivots = P)`), so the ratio compares
    the same first P pivots of the reference's sequence, measured in the same run on the same machine."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from relp_oracle import cpu
    from relp_oracle.mps import load_problem
    _, data = load_problem(os.path.join(ROOT, "data", "netlib", lp_name + ".SIF"))
    record = cpu.solve_provider(data, max_seconds=budget_seconds, trace=0)
    pivots = record["pivots_phase1"] + record["pivots_phase2"]
    return {"value": pivots / record["seconds"] if record["seconds"] > 0 else 0.0, "unit": "pivots/s", "cores": 1, "kind": "port", "mode": "faithful",
            "pivots": pivots, "seconds": record["seconds"], "status": record.get("status", ""), "stamps": record.get("stamps", []),
            "sample": "the first %d pivots of %s's exact solve (oracle/cpp, faithful, %s) in %.1f s on this host" % (
                pivots, lp_name, record.get("status", ""), record["seconds"])}


def run_cpu_leg(name, seconds):
    if name.startswith("exact_prefix:"):
        return cpu_leg_exact_prefix(name.split(":")[1], seconds)
    if name.startswith("exact_full:"):
        return cpu_leg_exact_full(name.split(":")[1], max(seconds, 120.0))
    if name == "exact_faithful":
        return cpu_leg_exact(WORKLOADS["25fv47"], seconds, False)
    if name == "exact_tuned":
        return cpu_leg_exact(WORKLOADS["25fv47"], seconds, True)
    if name == "f64":
        return cpu_leg_f64(WORKLOADS["25fv47"], seconds)
    if name.startswith("dense:"):
        m, n = (int(v) for v in name.split(":")[1].split("x"))
        return cpu_leg_dense((m, n), seconds)
    if name == "netlib":
        return cpu_leg_netlib(seconds)
    if name == "maxflow":
        return cpu_leg_maxflow(seconds)
    raise SystemExit("unknown cpu leg " + name)


class CpuLegs:
    """CPU baselines as child processes beside the GPU measurements (bounded BLAS pools, so that they do not crowd the host threads
    that feed the GPU); `collect` waits for them."""

    def __init__(self):
        self.children = {}
        self.done = {}

    def start(self, key, name, seconds, blas=8):
        env = dict(os.environ)
        for var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
            env[var] = str(blas)
        env["HIP_VISIBLE_DEVICES"] = ""  # a CPU leg never touches the GPU
        env["ROCR_VISIBLE_DEVICES"] = ""
        self.children[key] = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-leg", name, "--cpu-seconds", str(seconds)],
                                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)

    def peek(self, key, timeout=600):
        """Like collect, but the record stays available to a later collect."""
        if key not in self.done:
            self.done[key] = self.collect(key, timeout)
        return self.done[key]

    def collect(self, key, timeout=600):
        if key in self.done:
            return self.done[key]
        child = self.children.pop(key, None)
        if child is None:
            return None
        try:
            out, err = child.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            child.kill()
            return {"error": "timed out"}
        if child.returncode != 0:
            return {"error": err.strip()[-400:]}
        record = json.loads(out.strip().splitlines()[-1])
        record.update(host_description())
        return record


# =====================================================================================================================
# section 8(d) of SURVEY.md: algorithmic bytes per pivot, the figure the roofline fraction is quoted on
# =====================================================================================================================
def factor_nonzeros(model, basis, art_rows, inverse_entries=None):
    """nnzF = nnz(L) + nnz(U) + m of the basis the solve ended on (host factorisation, relp_lu_factor_host)."""
    import ctypes as C
    import numpy as np
    import relp_amd
    from relp_amd.basis_inverse import lu_factor_host
    buf_rows, buf_vals = np.zeros(model.nr_rows, np.int32), np.zeros(model.nr_rows, np.float64)
    count = C.c_int32()
    columns = []
    for c in basis:
        if c >= 0:
            relp_amd.lib().relp_model_column(model._h, int(c), model.nr_rows, C.byref(count), buf_rows.ctypes.data_as(C.POINTER(C.c_int32)),
                                             buf_vals.ctypes.data_as(C.POINTER(C.c_double)))
            columns.append([(int(buf_rows[e]), float(buf_vals[e])) for e in range(count.value) if buf_rows[e] < len(basis)])
        else:
            columns.append([(art_rows[-1 - int(c)], 1.0)])
    factors = lu_factor_host(columns)
    if inverse_entries is not None:  # the inverse-factor carry streams L^-1 and U^-1: their entries, counted on the dense inverses
        import scipy.linalg
        m = len(basis)
        lower, upper = np.eye(m), np.diag(np.asarray(factors["diag"], dtype=np.float64))
        for i, row in enumerate(factors["lower_rows"]):
            for j, v in row:
                lower[i, j] = v
        for i, row in enumerate(factors["upper_rows"]):
            for j, v in row:
                upper[i, j] = v
        inverse_entries.append(int(np.count_nonzero(scipy.linalg.solve_triangular(lower, np.eye(m), lower=True, unit_diagonal=True))) - m +
                               int(np.count_nonzero(scipy.linalg.solve_triangular(upper, np.eye(m), lower=False))))
    return factors["nnz_lower"] + factors["nnz_upper"] + len(basis)


def contract_bytes(price_bytes, nnz_f, m):
    """one fused pricing + steepest-edge pass, one FTRAN + one shared-pass two-RHS BTRAN over the factor
    (12 B per entry: f64 value + 32-bit index), twelve m-vectors."""
    solve = 2 * nnz_f * 12 + 12 * m * 8
    return {"pricing": int(price_bytes), "ftran_btran_vectors": int(solve), "per_pivot": int(price_bytes + solve)}


# =====================================================================================================================
# one LP resident on the GPU: 25FV47, the dense LPs, the max-flow LPs
# =====================================================================================================================
def single_lp(args, ctx):
    """Times `steps` solves of one workload (barrier + synchronize on both sides, max over ranks) and measures the kernels of a
    pivot with HIP events inside the real pivot sequence.  Returns the line (rank 0) or None."""
    import torch
    import relp_amd
    from relp_amd import batch
    rank, local_rank, world, distributed = ctx["rank"], ctx["local_rank"], ctx["world"], ctx["distributed"]
    path = WORKLOADS[args.workload]
    graph = isinstance(path, tuple) and path[0] == "maxflow"
    dense = not isinstance(path, str) and not graph
    model = None
    if graph:
        # the MatrixProvider of examples/max_flow.rs on the random graph of SURVEY.md section 8(d); the capacity rows are
        # handled as implicit bounds (the explicit formulation has V - 2 + E rows: no inverse of that size fits)
        from relp_amd.workloads import max_flow_graph
        _, nr_vertices, nr_arcs = path
        key = ("maxflow_model", nr_vertices, nr_arcs)
        if key not in ctx["cache"]:
            tail, head, capacity = max_flow_graph(nr_vertices, nr_arcs)
            ctx["cache"][key] = relp_amd.Model.max_flow(nr_vertices, list(zip(tail.tolist(), head.tolist(), capacity.tolist())), 0, nr_vertices - 1)
        model = ctx["cache"][key]
        solver = relp_amd.Solver(device=local_rank, implicit_bounds=1, crash=args.crash).load_model(model)
    elif dense:
        from relp_amd.workloads import dense_lp
        key = ("dense_lp",) + tuple(path)
        if key not in ctx["cache"]:
            ctx["cache"][key] = dense_lp(*path)
        a, b, c = ctx["cache"][key]
        # generic data: the block as float / double through relp_options.dense_storage (the narrowest exact type is the default)
        solver = relp_amd.Solver(device=local_rank, polish_period=int(os.environ.get("RELP_POLISH", "512")),
                                 dense_storage={"narrowest": 0, "f32": 1, "f64": 2}[args.dense_storage]).load_dense_le(a, b, c)
    else:
        # the step is `solve_relaxation` with the exact certificate INSIDE: the f64 loop alone is narrower arithmetic than the
        # reference's, the bit-exact optimum is part of the job (BASELINE.json north_star)
        model = relp_amd.Model(path, presolve=args.presolve)
        solver = relp_amd.Solver(device=local_rank, certify=0 if args.no_certify else 1, carry=args.carry, lu_refactor=args.lu_refactor).load_model(model)

    def barrier():
        if distributed:
            torch.distributed.barrier()
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
    final_basis = solver.basis() if rank == 0 else None
    # one record per rank (what each GPU did); the makespan is the max over ranks of the barrier-to-barrier time
    per_rank = batch.gather_records({"rank": rank, "solves": args.steps, "pivots": int(pivots), "busy_seconds": loop_seconds + certify_seconds,
                                     "elapsed_seconds": elapsed})
    elapsed, pivots = batch.aggregate(elapsed, pivots, device=ctx["reduce_device"])

    in_flight = None
    if rank == 0 and world == 1 and not dense and not graph and not args.no_concurrency_probe:
        # headroom: the same LP, 4 independent copies in flight on this GPU (one host thread and stream each); a single
        # latency-bound solve uses a fraction of the chip.  Reported beside `value`, never as `value`.
        import threading
        copies = [solver] + [relp_amd.Solver(device=local_rank, certify=0 if args.no_certify else 1, carry=args.carry).load_model(model) for _ in range(3)]
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

    if rank != 0:
        solver.close()
        return None
    exact = None
    if not dense and not graph and not args.no_certify:
        # the exact optimum of the LAST TIMED solve (the certificate ran inside every timed step)
        if all_certified:
            text = solver.objective_exact()
            num, den = text.split("/")
            exact = {"certified": True, "objective_bits": max(int(num).bit_length(), int(den).bit_length()),
                     "certify_seconds_per_solve": certify_seconds / args.steps, "objective_exact_head": text[:40] + "...",
                     "solution_exact_nonzeros": len(solver.solution_exact())}
        else:
            exact = {"certified": False}
    # ---- roofline of the dominant kernel, measured live with HIP events on the solver's stream -------------------------------
    solver.begin_phase_one()
    # (phase one of the flow LP has only a few hundred ordinary pivots before the zero-level ones: short samples there)
    _, reason = solver.iterate(20 if graph else (300 if dense else 200))
    if graph and reason == relp_amd.STOP_NO_ENTERING:  # crash basis: phase one ends without a pivot -- profile phase two
        solver.begin_phase_two()
        solver.iterate(20)
    reps = 40 if graph else (100 if dense else 200)  # further real pivots, the profiled kernel of each bracketed by its own event pair
    solver.profile_kernel(0, 10 if graph else 50)   # discarded: brings clocks and caches to the state of a running solve
    lu_carry = args.carry in (1, 2) and not dense and not graph
    kernels = ["price", "lu_pivot"] if lu_carry else ["price", "ftran_ratio", "update"]
    try:
        seconds = {name: solver.profile_kernel(which, reps) for which, name in enumerate(kernels)}
    except relp_amd.api.RelpError:
        # m <= 2048: ratio test and inverse update are ONE launch (pivot_fused_kernel): two kernels per pivot
        kernels = ["price", "pivot_fused"]
        seconds = {name: solver.profile_kernel(which, reps) for which, name in enumerate(kernels)}
    stats = solver.stats()
    m_rows = solver.m
    # ---- algorithmic bytes: the contract's and each kernel's own ---------------------
    # the kernel's own: pricing = the non-basic columns (exact, counted at the profiled state); K2 = nnz(a_q) columns of the inverse
    # + six m-vectors; K3 = read + write of the touched part of the inverse (upper bound: all of it); the LU kernel = both
    # orientations of the factors once each + twelve m-vectors.
    if dense:
        basic_structurals = int(sum(1 for c in final_basis if 0 <= c < path[1]))
        nnz_f = basic_structurals * m_rows + m_rows  # SURVEY.md section 8(d), worked example: a dense k x k block plus k (m - k) entries
        mean_column = float(m_rows)
    else:
        pivots_initial = model.pivot_element_indices()
        art_rows = sorted(set(range(model.nr_rows)) - {r for r, _ in pivots_initial})
        try:
            # (graph LPs: a tree basis -- at most two entries per arc column -- is counted, not factorised on the host, at a million rows)
            inverse_entries = [] if (lu_carry and args.carry == 2) else None
            nnz_f = 3 * m_rows if graph else factor_nonzeros(model, final_basis, art_rows, inverse_entries)
        except Exception:  # noqa: BLE001  (a basis the host factorisation rejects: fall back to 3 entries per row and triangle)
            nnz_f = 7 * m_rows
        mean_column = 2.0 if graph else max(1.0, float(model.nnz) / max(1, solver.n_provider))
    contract = contract_bytes(stats.price_bytes, nnz_f, m_rows)
    own = {"price": stats.price_bytes,
           "ftran_ratio": int(mean_column * m_rows * 8 + 6 * m_rows * 8),
           "update": stats.update_bytes,
           # (the inverse-factor form: both inverted triangles twice, 11 B per entry in its compact records, the kept columns of M --
           #  about half a period of them -- read twice and written once, the same twelve vectors)
           "lu_pivot": int(2 * nnz_f * 12 + 12 * m_rows * 8) if not (lu_carry and args.carry == 2 and not graph)
                       else int(2 * inverse_entries[0] * 11 + 3 * 16 * m_rows * 8 + 12 * m_rows * 8),
           # fused: one workgroup's FTRAN + ratio test, and ONE read + one write of the whole inverse (out of place)
           "pivot_fused": int(mean_column * m_rows * 8 + 6 * m_rows * 8 + 2 * m_rows * m_rows * 8)}
    # the contract's share per kernel: the pricing pass is the pricing kernel's; everything else belongs to the kernel(s) that do
    # the FTRAN / ratio test / BTRAN / update of a pivot, split by their measured time
    rest = [k for k in kernels if k != "price"]
    rest_seconds = sum(seconds[k] for k in rest)
    contract_share = {"price": contract["pricing"]}
    for k in rest:
        contract_share[k] = int(contract["ftran_btran_vectors"] * (seconds[k] / rest_seconds if rest_seconds > 0 else 1.0 / len(rest)))
    per_kernel = {name: {"seconds_per_launch": seconds[name],
                         "contract_bytes_per_launch": contract_share[name], "kernel_bytes_per_launch": own[name],
                         "achieved_gb_s": contract_share[name] / seconds[name] / 1e9, "frac": contract_share[name] / seconds[name] / 1e9 / HBM_PEAK_GBS,
                         "kernel_bytes_gb_s": own[name] / seconds[name] / 1e9, "kernel_bytes_frac": own[name] / seconds[name] / 1e9 / HBM_PEAK_GBS,
                         "share_of_pivot_time": seconds[name] / sum(seconds.values())} for name in kernels}
    dominant = max(kernels, key=lambda name: seconds[name])  # by MEASURED time share, not by assumption
    if graph and "price" in kernels:
        # Graph LPs: the pricing pass and the update of the inverse take about the same time per launch and swapped places from run to
        # run (frac 0.10 / 0.94 for the same code).  The contract's bytes describe the pricing pass -- 13 B per arc, generated columns --
        # while the update's contract share is bytes that kernel never moves (a tree basis has almost no inverse to stream): the
        # roofline is the pricing pass's, always; every kernel of the pivot is in `roofline.per_kernel` of the detail file.
        dominant = "price"
    achieved = per_kernel[dominant]["achieved_gb_s"]
    # HBM traffic per launch from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE on gfx950, MI355X_MICROARCH.md
    # section HBM); null when not collected for this kernel
    traffic = None
    # (the dense pricing kernel is one template per storage type: bytes, float, double -- the traffic of another instance is not this one's)
    dense_price = "price_dense_lane_kernel<%d>" % {"narrowest": 1, "f32": 4, "f64": 8}[args.dense_storage]
    pmc_names = {"price": dense_price if dense else ("relp::price_unit_kernel<" if graph else "relp::price_kernel<"), "ftran_ratio": "ftran_ratio", "update": "update_kernel",
                 "lu_pivot": "lu_pivot_kernel", "pivot_fused": "pivot_fused_kernel"}
    tag = args.workload + (("_lu" if args.carry == 1 else "_lui") if lu_carry else "")
    if lu_carry and args.lu_refactor == 1:
        tag += "_device_refactor"
    # (round 6: one file per configuration as measured -- the dense block's storage, the max-flow start -- at this round's kernels; the older
    #  files remain for the configurations not profiled again)
    this_round = "r6_%s_%s_pmc_traffic.json" % (args.workload, args.dense_storage) if dense else \
        "r6_maxflow_%s_pmc_traffic.json" % ("crash" if args.crash else "reference_start") if graph and args.workload == "maxflow" else "r6_%s_pmc_traffic.json" % tag
    for candidate in (this_round, "r4_%s_pmc_traffic.json" % tag, "r3_%s_pmc_traffic.json" % tag, "r2_%s_pmc_traffic.json" % tag, "r2_%s_pmc_traffic.json" % args.workload,
                      "r1_%s_pmc_traffic.json" % args.workload):
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
            solver.m, solver.n_provider, {1: "LU + Forrest-Tomlin", 2: "LU inverse factors + product-form updates"}.get(args.carry, "explicit-inverse"),
            "outside" if args.no_certify else "inside",
            "after the reference's presolve" if args.presolve else "no presolve (+520 virtual artificials)"))
        data = "Netlib 25FV47.SIF (shipped problem file), one copy per GPU"
    options = solver.options
    seconds_per_pivot = loop_seconds / max(1, pivots // max(1, world))
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
                   "carry": {1: "lu", 2: "lu_inverse"}.get(args.carry, "explicit"), "refactors": int(last.refactors),
                   "lu_refactor": ("device kernels (lu_factor.hip, lu_device_tasks.hip)" if args.lu_refactor == 1 else "host core") if lu_carry else None,
                   "refactor_seconds_per_solve": float(last.refactor_seconds),
                   "polishes": int(last.polishes), "max_residual_before_polish": last.max_residual,
                   # the reference is exact and has neither tolerances nor a Harris test: what f64 adds, and what it changes
                   # (relp_options.ratio_rule = AUTO resolves against the data at load: the per-LP record says which test ran)
                   "ratio_rule": "harris two-pass (what AUTO -- the default -- resolves to on decimal data; the reference's textbook rule with Bland ties "
                                 "runs on small-integer data and under relp_options.ratio_rule = 1)"
                                 if solver.record().get("ratio_rule") == "harris" else "textbook minimum ratio, Bland ties (the reference's)",
                   "tolerances": {"tol_dual": options.tol_dual, "tol_pivot": options.tol_pivot, "harris_delta": options.harris_delta,
                                  "tol_feasible": options.tol_feasible},
                   "pivot_sequence": "f64 + Harris: 2383 pivots on 25FV47 where the exact reference rule makes 2392 (tests/golden/25FV47.json); "
                                     "the certified optimum is the same rational" if not dense and not graph else None,
                   "parallelism": "1 LP per GPU x%d" % world,
                   "makespan_s": elapsed, "per_rank": sorted(per_rank, key=lambda r: r["rank"]), "exact": exact,
                   "aggregate_with_copies_in_flight": in_flight},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "seconds_per_launch": seconds,
                     "algorithmic_bytes_per_launch": per_kernel[dominant]["contract_bytes_per_launch"],
                     "kernel_bytes_per_launch": per_kernel[dominant]["kernel_bytes_per_launch"],
                     "kernel_bytes_frac": per_kernel[dominant]["kernel_bytes_frac"],
                     "contract_bytes_per_pivot": contract, "factor_nonzeros": int(nnz_f),
                     "per_pivot": {"seconds": seconds_per_pivot, "contract_gb_s": contract["per_pivot"] / seconds_per_pivot / 1e9 if seconds_per_pivot > 0 else None,
                                   "contract_frac": contract["per_pivot"] / seconds_per_pivot / 1e9 / HBM_PEAK_GBS if seconds_per_pivot > 0 else None},
                     "kernels": per_kernel},
    }
    if graph:
        line["metric"] = "simplex pivots/sec + wall-clock to optimal, max-flow LP @1 GPU"
        line["roofline"]["note"] = ("the dominant kernel by measured time is listed first; 'price' generates each incidence column from the 8 bytes of "
                                    "its arc's endpoints (+ 1 B cost, 4 B basis position: 13 B per arc; a steepest-edge weight is read/written only by candidates "
                                    "and by columns with an entry in the pivot row) and gathers one packed 32-byte (-pi, rho, w) record per entry from "
                                    "L2; 'update' walks the non-zero rows of alpha in every column of the inverse that is not a unit vector (8-byte "
                                    "gathers, one cache line each: PMC traffic is line-granular)")
    elif dense:
        line["metric"] = "simplex pivots/sec + wall-clock to optimal, dense LP @1 GPU"
        if args.dense_storage == "narrowest":
            line["roofline"]["note"] = ("dense block held in the narrowest exact type (this workload: signed bytes, 1 B per entry -- data specific; the "
                                        "generic figure is the f64-storage config): one column per lane, -pi / rho / w "
                                        "broadcast through DPP inside the f64 FMA (no LDS traffic); 5 VALU instructions per entry = %.1f us "
                                        "of issue time on 1024 SIMDs (tools/micro/valu_rates.hip: 1.9 ns each), the rest of the launch is "
                                        "the first load's latency and the per-workgroup tail" % (stats.price_bytes * 5 * 1.9e-3 / 64 / 1024))
        else:
            line["roofline"]["note"] = {
                "f32": "dense block streamed as float (exact for this data; all arithmetic f64): one column per lane, 16-byte "
                       "non-temporal loads, -pi / rho / w broadcast through DPP inside the f64 FMA (no LDS traffic); bound by HBM",
                "f64": "dense block streamed as double (SURVEY.md section 8(d)'s bytes: 8 B per entry): one column per lane, 16-byte non-temporal "
                       "loads, -pi / rho / w broadcast through DPP inside the f64 FMA (no LDS traffic); bound by HBM"}[args.dense_storage]
    else:
        line["roofline"]["note"] = ("latency bound by construction: the dominant kernel BY MEASURED TIME is '%s'; `frac` prices it on the contract's "
                                    "bytes (SURVEY.md section 8(d): the factor entries twice + twelve m-vectors, %d KB per pivot outside pricing), "
                                    "`kernel_bytes_frac` on what the kernel itself streams (%d KB per launch, resident in L2 / Infinity Cache); every kernel of "
                                    "the pivot is listed under 'kernels' with its share of the pivot time; traffic = 2 x FETCH_SIZE + "
                                    "WRITE_SIZE of the committed PMC passes for that kernel; the HBM-roofline configuration is BASELINE "
                                    "configs[2], under configs.dense4096_f64") % (dominant, contract["ftran_btran_vectors"] // 1024,
                                                                                  per_kernel[dominant]["kernel_bytes_per_launch"] // 1024)
    solver.close()
    return line


# =====================================================================================================================
# config 4: the Netlib batch through the library's batch entry (relp_batch_*)
# =====================================================================================================================
def netlib_batch(args, ctx):
    """Independent LPs shard across ranks through ONE ticket queue over K passes of the cost-sorted suite: inside a process the
    library's worker threads draw tickets (relp_batch_run), across processes the queue is an atomic counter on the process
    group's store (relp_amd.batch.TicketQueue handed to the library as `next_ticket`).  Every LP is resident in HBM on every
    worker before the timed region.  value = pivots of all ranks / makespan (max over ranks) = SUITE THROUGHPUT; a single pass
    is bounded below by its longest LP, reported as `single_pass_makespan_s`."""
    import torch
    import relp_amd
    from relp_amd import batch
    rank, local_rank, world, distributed = ctx["rank"], ctx["local_rank"], ctx["world"], ctx["distributed"]
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "netlib_expected.json")))
    names = netlib_names(expected)
    key = ("netlib_models", bool(args.presolve))
    if key not in ctx["cache"]:
        ctx["cache"][key] = [relp_amd.Model(os.path.join(ROOT, "data", "netlib", name + ".SIF"), presolve=args.presolve) for name in names]
    models = ctx["cache"][key]
    costs = [float(mdl.nr_rows) * float(mdl.nnz + mdl.nr_columns) for mdl in models]
    ordered = sorted(range(len(names)), key=lambda k: (-costs[k], names[k]))  # longest estimated first
    workers = max(1, args.concurrency)
    static = args.schedule == "static"
    if static:  # longest-first partition: this rank's batch holds (and serves) its own share only
        mine = set(batch.assign([(names[k], costs[k]) for k in range(len(names))], world)[rank])
        ordered = [k for k in ordered if names[k] in mine]
    pool = relp_amd.Batch(models, devices=(local_rank,), workers_per_device=workers)
    # Order of the K x 45 tickets.  1-2 GPUs: pass after pass, each longest-first.  4+ GPUs: the cost-sorted list in chunks of
    # eight LPs, all K passes of a chunk before the next chunk -- close to longest-first over everything (the long solves of
    # the last pass do not start late) while consecutive tickets are still different LPs.
    chunk = 8 if world >= 4 else len(ordered)
    chunk = max(1, int(os.environ.get("RELP_BATCH_CHUNK", chunk)))  # diagnostic override

    def schedule(repeat):
        out = []
        for first in range(0, len(ordered), chunk):
            part = ordered[first:first + chunk]
            for _ in range(repeat):
                out.extend(part)
        return out

    counter = [0]

    def run(repeat):
        counter[0] += 1
        sched = schedule(repeat)
        queue = None
        if distributed and world > 1 and not static:
            tickets = batch.TicketQueue(len(sched), tag="r3pass%d" % counter[0])
            queue = lambda: (lambda t: len(sched) if t is None else t)(tickets.next())  # noqa: E731
        return sched, pool.run(sched, next_ticket=queue)

    for _ in range(args.warmup):
        run(1)
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    start = time.perf_counter()
    sched, (entries, worker_stats, _) = run(args.steps)
    if distributed:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - start
    served = [e for e in entries if e.status == 0]
    pivots_local = sum(e.result.pivots_phase_one + e.result.pivots_phase_two for e in served)
    elapsed, pivots = batch.aggregate(elapsed_local, pivots_local, device=ctx["reduce_device"])
    record = {"rank": rank, "tickets": len(served), "pivots": int(pivots_local), "solve_seconds": sum(e.result.solve_seconds for e in served),
              "elapsed_seconds": elapsed_local,
              "workers": [{"tickets": int(w.tickets), "pivots": int(w.pivots), "busy_seconds": w.busy_seconds, "idle_seconds": w.idle_seconds,
                           "queue_seconds": w.queue_seconds} for w in worker_stats],
              "results": [(names[e.model], e.result.objective, int(e.result.pivots_phase_one + e.result.pivots_phase_two), e.result.solve_seconds)
                          for e in served]}
    lp_records = []
    if args.records is not None:
        import ctypes as C
        for e in served:  # the resident handle's record of its last solve
            h = C.c_void_p()
            relp_amd.lib().relp_batch_handle(pool._h, e.worker, e.model, C.byref(h))
            length = C.c_int32()
            relp_amd.lib().relp_get_record_json(h, None, 0, C.byref(length))
            buf = C.create_string_buffer(length.value + 1)
            relp_amd.lib().relp_get_record_json(h, buf, length.value + 1, C.byref(length))
            lp_records.append(json.loads(buf.value.decode()))
        with open(args.records if world == 1 else "%s.rank%d" % (args.records, rank), "w") as handle:
            for entry in lp_records:
                handle.write(json.dumps(entry) + "\n")
    # bytes the suite moves on the contract's terms: the per-LP record's pricing bytes per pivot x its pivots (one pass each)
    contract_total = 0
    if rank == 0 and world == 1:
        import ctypes as C
        for e in served:
            h = C.c_void_p()
            relp_amd.lib().relp_batch_handle(pool._h, e.worker, e.model, C.byref(h))
            stats = relp_amd.api.Stats()
            relp_amd.lib().relp_get_stats(h, C.byref(stats))
            m_rows = models[e.model].nr_rows
            contract_total += (e.result.pivots_phase_one + e.result.pivots_phase_two) * (stats.price_bytes + 2 * 7 * m_rows * 12 + 12 * m_rows * 8)
    gathered = batch.gather_records(record)
    pool.close()
    if rank != 0:
        return None
    wrong = []
    longest = {}
    for rank_record in gathered:
        for name, objective, _, seconds in rank_record["results"]:
            e = expected[name]
            tolerance = max(e["tolerance"], 2e-5 if name == "25FV47" else 0.0)
            if abs(objective - e["expected"]) > tolerance:
                wrong.append(name)
            longest[name] = max(longest.get(name, 0.0), seconds)
    slowest = max(longest, key=longest.get) if longest else None
    line = {
        "metric": "simplex pivots/sec, Netlib suite batched one LP per GPU", "value": pivots / elapsed, "unit": "pivots/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        # suite throughput over ONE queue of K passes: the work is fixed as N grows (strong scaling of the K x 45 tickets); a single
        # pass cannot scale past its longest LP (`single_pass_makespan_s`)
        "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "%d Netlib .SIF files shipped under data/netlib" % len(names),
        "config": {"workload": "Netlib batch (%d LPs%s), %s, independent LPs sharded over the GPUs through relp_batch_run" % (
            len(names), ", after the reference's presolve" if args.presolve else "",
            "static longest-first assignment" if static else "dynamic ticket queue over the cost-sorted list"),
                   "throughput_kind": "suite throughput: %d passes over the suite as one ticket queue" % args.steps,
                   "lps_in_flight_per_gpu": workers,
                   "single_pass_makespan_s": longest.get(slowest) if slowest else None, "longest_lp": slowest,
                   "tickets_per_rank": [r["tickets"] for r in gathered],
                   "pivots_per_rank": [r["pivots"] for r in gathered],
                   "solve_seconds_per_rank": [r["solve_seconds"] for r in gathered],
                   "elapsed_seconds_per_rank": [r["elapsed_seconds"] for r in gathered],
                   "workers_per_rank": [r["workers"] for r in gathered],
                   "makespan_s": elapsed, "objectives_outside_reference_tolerance": wrong}}
    if world == 1 and contract_total:
        achieved = contract_total / elapsed / 1e9
        line["roofline"] = {"bound": "hbm", "kernel": "whole batch", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                            "note": "latency / L2 bound (SURVEY.md section 8(d)): contract bytes of every pivot of the run (pricing pass of the "
                                    "LP + 2 x ~7 factor entries per row x 12 B + twelve m-vectors) / makespan; not roofline-meaningful"}
    return line


# =====================================================================================================================
# the loop in fixed-width exact integers on the device (relp_solve_exact): the reference's own pivot sequence
# =====================================================================================================================
EXACT_SAME_WORK_LP = "E226"   # mid-size: the exact CPU restatement finishes it in seconds, so both sides run the SAME pivots to completion


def exact_lp(lp_name, ctx, first_limbs, max_limbs, max_pivots=0):
    """One solve of `lp_name` with `relp_solve_exact` (LIMBS x 64-bit integers, widened where a value might not fit), timed: value =
    pivots of the reference's sequence per second.  The roofline entry prices the integer update of the m x m numerator matrix."""
    import relp_amd
    golden_path = os.path.join(ROOT, "tests", "golden", lp_name + ".json")
    golden = json.load(open(golden_path)) if os.path.exists(golden_path) else None
    solver = relp_amd.Solver(device=ctx["local_rank"]).load_mps(os.path.join(ROOT, "data", "netlib", lp_name + ".SIF"))
    start = time.perf_counter()
    got = solver.solve_exact(first_limbs=first_limbs, max_limbs=max_limbs, max_pivots=max_pivots)
    elapsed = time.perf_counter() - start
    pivots = got["pivots_phase_one"] + got["pivots_phase_two"]
    m = solver.m
    widths = solver.exact_counters()
    solver.close()
    # The roofline of the dominant step at the final width: the integer-preserving update of N = D B^-1.  Its work is counted in the
    # kernel in 64 x 64 -> 128-bit word products: `needed` by the entries' bit bounds (the algorithmic count) and `issued` by the waves
    # (whole 64-byte blocks, the widest entry of a tile for all sixteen).  From 32 limbs on the products run on the matrix cores as byte
    # products (v_mfma_i32_16x16x64_i8: 64 byte MACs = one word product), so the peak is the chip's dense i8 MFMA rate; below 32 limbs
    # they run on the vector multiplier (v_mad_u64_u32; the compiled 4 x 4-word block product reaches 2.1 T word products/s:
    # profiles/r5_micro_intmul_rates.txt).
    last = widths[-1] if widths else None
    roofline = {"bound": "mfma", "kernel": "exact_simplex_kernel<%d>: update of N" % got["limbs"], "achieved": None, "peak": None, "unit": "T word products/s",
                "frac": None, "traffic": None}
    if last:
        update_seconds = last["step_seconds"]["update of N"]
        on_matrix_cores = got["limbs"] >= 32
        peak = I8_MFMA_PEAK_TMACS / 64.0 if on_matrix_cores else VALU_WORD_PRODUCT_PEAK
        achieved = last["update_word_products_needed"] / update_seconds / 1e12 if update_seconds > 0 else 0.0
        roofline.update({
            "bound": "mfma" if on_matrix_cores else "int-mul (valu)", "achieved": achieved, "peak": peak, "frac": achieved / peak,
            "word_products_needed": last["update_word_products_needed"], "word_products_issued": last["update_word_products_issued"],
            "issued_per_second_T": last["update_word_products_issued"] / update_seconds / 1e12 if update_seconds > 0 else None,
            "update_seconds_at_final_width": update_seconds, "seconds_at_final_width": last["seconds"],
            "step_seconds_at_final_width": last["step_seconds"],
            "note": "achieved = word products the entries need / seconds of the update step at the final width (%d limbs); peak = %s"
                    % (got["limbs"], "dense i8 MFMA, 2.5 P MAC/s / 64 (measured 2.39: profiles/r5_micro_intmul_rates.txt)" if on_matrix_cores
                       else "the compiled block product on v_mad_u64_u32, measured")})
    matches = None
    if golden is not None:
        matches = (got["status"] == 1 and got["objective"] == golden["objective"] and
                   (got["pivots_phase_one"], got["pivots_phase_two"]) == (golden["pivots_phase1"], golden["pivots_phase2"]))
    if max_pivots and golden is not None:  # a prefix of the solve: the pivots made must be the golden trace's first ones
        reference = [tuple(t) for t in golden.get("trace", golden["trace_head"])]
        shared = min(len(reference), len(got["trace"]))
        matches = got["trace"][:shared] == reference[:shared]
    return {"metric": "simplex pivots/sec, exact fixed-width integers on the device, Netlib %s" % lp_name, "value": pivots / elapsed if elapsed > 0 else 0.0,
            "unit": "pivots/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1e3 * elapsed, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int%d" % (64 * got["limbs"]), "data": "Netlib %s.SIF" % lp_name,
            "config": {"workload": "Netlib %s %d rows, relp_solve_exact: the reference's pivot sequence (steepest edge, Bland ratio ties) in %d x 64-bit "
                                   "integers over a common denominator, widths tried %s" % (lp_name, m, got["limbs"], got["survived"]),
                       "pivots_per_solve": pivots, "status": got["status"], "limbs": got["limbs"], "widths_and_pivots_survived": got["survived"],
                       "objective_exact": got["objective"], "objective": None, "matches_golden_optimum_and_pivot_counts": matches,
                       "seconds_per_width": [[w["limbs"], w["seconds"]] for w in widths]},
            "roofline": roofline}


EXACT_25FV47_CPU = {"value": 2392 / 1026.0, "unit": "pivots/s", "cores": 1, "kind": "port", "mode": "faithful", "recorded": True, "seconds": 1026.0,
                    "recorded_on_other_host": True,
                    "sample": "the whole exact solve on the CPU restatement: 2392 pivots in 1026 s (profiles/r1_cpu_oracle_full_solve.json, build container, "
                              "NOT the bench host: see same_work for the measured pair)"}


def exact_25fv47_live(ctx):
    """BASELINE configs[1] pivot for pivot in fixed-width integers, measured in this process.  The CPU side of `same_work` is measured in the
    same run on the same host (`exact_same_work_25fv47`); the whole CPU solve recorded in round 1 on another machine is kept only as a
    labelled extra."""
    entry = exact_lp("25FV47", ctx, 4, 128)
    entry["recorded"] = False
    entry["cpu_baseline"] = dict(EXACT_25FV47_CPU)
    return entry


def exact_same_work_25fv47(ctx, entry, cpu):
    """`same_work_exact_25fv47`: the exact CPU port ran P pivots of 25FV47 on this host in this run (`cpu`, the exact_prefix leg); the
    device path is timed on the same first P pivots (`relp_solve_exact(max_pivots = P)`, widths escalating from 4 limbs as in the whole
    solve) -- or the whole solve on both sides when the CPU leg reached the optimum."""
    whole = cpu.get("status") == "optimal"
    pivots = cpu["pivots"]
    if whole:
        seconds, limbs, got_pivots = entry["ms_per_step"] / 1e3, entry["config"]["limbs"], entry["config"]["pivots_per_solve"]
    else:
        prefix = exact_lp("25FV47", ctx, 4, 128, max_pivots=pivots)
        seconds, limbs, got_pivots = prefix["ms_per_step"] / 1e3, prefix["config"]["limbs"], prefix["config"]["pivots_per_solve"]
    return {"lp": "25FV47", "pivots": pivots, "pivots_gpu": got_pivots, "whole_solve": whole, "gpu_seconds": seconds, "cpu_seconds_measured": cpu["seconds"],
            "cpu_over_gpu": cpu["seconds"] / seconds if seconds > 0 else None, "limbs": limbs, "host": cpu.get("cpu_model"), "cpu_cores": 1,
            "same_run_same_host": True, "cpu_stamps_pivot_seconds": cpu.get("stamps", [])[-6:],
            "gpu_whole_solve_seconds": entry["ms_per_step"] / 1e3, "matches_golden": entry["config"]["matches_golden_optimum_and_pivot_counts"],
            "cpu_whole_solve_seconds_recorded_other_host": EXACT_25FV47_CPU["seconds"]}


def exact_25fv47_recorded():
    """The committed measurement of the same call (profiles/r4_exact_25fv47_128_limbs.txt; also the gpu test
    tests/test_gpu_exact.py::test_25fv47_whole_reference_pivot_sequence_in_fixed_width_integers), for `--recorded-exact-25fv47`."""
    path = os.path.join(ROOT, "profiles", "r4_exact_25fv47_128_limbs.txt")
    row = [r for r in open(path) if r.startswith("25FV47")][-1].split()
    seconds = float(row[row.index("s") - 1])
    pivots = 1133 + 1259
    return {"metric": "simplex pivots/sec, exact fixed-width integers on the device, Netlib 25FV47", "value": pivots / seconds, "unit": "pivots/s", "n_gpus": 1,
            "steps": 1, "warmup": 0, "ms_per_step": 1e3 * seconds, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int8192",
            "data": "Netlib 25FV47.SIF", "recorded": True,
            "config": {"workload": "Netlib 25FV47 821 rows, relp_solve_exact, 4 -> 128 limbs: RECORDED run (profiles/r4_exact_25fv47_128_limbs.txt), not measured "
                                   "in this process (the default run measures it: leave out --recorded-exact-25fv47)",
                       "pivots_per_solve": pivots, "limbs": 128, "matches_golden_optimum_and_pivot_counts": True},
            "roofline": {"bound": "hbm", "kernel": "exact_simplex_kernel<128>", "achieved": 2 * 821 * 821 * 128 * 8 * pivots / seconds / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": 2 * 821 * 821 * 128 * 8 * pivots / seconds / 1e9 / HBM_PEAK_GBS, "traffic": None},
            "cpu_baseline": dict(EXACT_25FV47_CPU)}


# =====================================================================================================================
def all_configs(args, ctx, legs):
    """The other BASELINE configs, measured in this process after the headline (default run, N = 1)."""
    out = {}

    def variant(**overrides):
        v = copy.copy(args)
        v.no_concurrency_probe = True
        for k, val in overrides.items():
            setattr(v, k, val)
        return v

    def attempt(key, fn):
        t0 = time.perf_counter()
        try:
            line = fn()
            line["bench_wall_seconds"] = time.perf_counter() - t0
            out[key] = line
        except Exception as error:  # noqa: BLE001  (one config must not take the headline down)
            out[key] = {"error": "%s: %s" % (type(error).__name__, error)}

    steps = max(1, min(args.steps, 3))
    if not args.no_cpu_baseline:
        # the longest CPU leg first: the exact port on the metric's LP, for --exact-cpu-seconds on this host, beside all the GPU work below
        legs.start("exact_prefix", "exact_prefix:25FV47", args.exact_cpu_seconds)
        legs.start("exact_full", "exact_full:" + EXACT_SAME_WORK_LP, args.cpu_seconds)
    attempt("exact_" + EXACT_SAME_WORK_LP.lower(), lambda: exact_lp(EXACT_SAME_WORK_LP, ctx, 2, 32))
    attempt("exact_25fv47", lambda: exact_25fv47_recorded() if args.recorded_exact_25fv47 else exact_25fv47_live(ctx))
    attempt("lu_inverse_carry_25fv47_device_refactor", lambda: single_lp(variant(carry=2, lu_refactor=1, steps=steps, warmup=1), ctx))
    attempt("lu_carry_25fv47", lambda: single_lp(variant(carry=1, steps=steps, warmup=1), ctx))
    attempt("lu_inverse_carry_25fv47", lambda: single_lp(variant(carry=2, steps=steps, warmup=1), ctx))
    attempt("dense4096_f64", lambda: single_lp(variant(workload="dense4096", dense_storage="f64", steps=steps, warmup=1), ctx))
    attempt("dense4096_narrowest", lambda: single_lp(variant(workload="dense4096", dense_storage="narrowest", steps=steps, warmup=1), ctx))
    attempt("maxflow_reference_start", lambda: single_lp(variant(workload="maxflow", crash=0, steps=1, warmup=1), ctx))
    attempt("maxflow_crash", lambda: single_lp(variant(workload="maxflow", crash=1, steps=steps, warmup=1), ctx))
    if not args.no_cpu_baseline:
        key = "exact_" + EXACT_SAME_WORK_LP.lower()
        if "error" not in out[key]:
            full = legs.collect("exact_full", timeout=900)
            out[key]["cpu_baseline"] = full
            if full and "error" not in full:
                gpu = out[key]["config"]
                out[key]["same_work"] = {"lp": EXACT_SAME_WORK_LP, "pivots_gpu": gpu["pivots_per_solve"], "pivots_cpu": full.get("pivots"),
                                         "gpu_seconds": out[key]["ms_per_step"] * 1e-3, "cpu_seconds": full.get("seconds"),
                                         "cpu_over_gpu": full.get("seconds") / (out[key]["ms_per_step"] * 1e-3) if out[key]["ms_per_step"] > 0 else None,
                                         "same_pivot_count": full.get("pivots") == gpu["pivots_per_solve"] and full.get("status") == "optimal"}
        for key in ("lu_carry_25fv47", "lu_inverse_carry_25fv47", "lu_inverse_carry_25fv47_device_refactor"):  # the same LP as the headline: the same exact CPU path beside it
            if "error" not in out[key]:
                out[key]["cpu_baseline"] = legs.peek("exact_faithful")
        dense_cpu = legs.collect("dense")
        for key in ("dense4096_f64", "dense4096_narrowest"):
            if "error" not in out[key]:
                out[key]["cpu_baseline"] = dense_cpu
        flow_cpu = legs.collect("maxflow")
        for key in ("maxflow_reference_start", "maxflow_crash"):
            if "error" not in out[key]:
                out[key]["cpu_baseline"] = flow_cpu
        if "error" not in out["exact_25fv47"] and not out["exact_25fv47"].get("recorded"):
            prefix = legs.collect("exact_prefix", timeout=args.exact_cpu_seconds + 600)
            if prefix and "error" not in prefix and prefix.get("pivots"):
                try:
                    out["exact_25fv47"]["same_work"] = exact_same_work_25fv47(ctx, out["exact_25fv47"], prefix)
                    out["exact_25fv47"]["cpu_baseline_same_host"] = {k: v for k, v in prefix.items() if k != "stamps"}
                except Exception as error:  # noqa: BLE001
                    out["exact_25fv47"]["same_work"] = {"error": "%s: %s" % (type(error).__name__, error)}
            else:
                out["exact_25fv47"]["same_work"] = {"error": (prefix or {}).get("error", "no CPU leg")}
    # The batch lines last, when the CPU legs above have finished: a batch has four host threads that factorise, certify and feed
    # four streams, and shared the cores with three busy CPU legs before (0.59-0.75 s per presolved pass from run to run).
    attempt("netlib_batch", lambda: netlib_batch(variant(workload="netlib", presolve=False, steps=2, warmup=1), ctx))
    attempt("netlib_batch_presolve", lambda: netlib_batch(variant(workload="netlib", presolve=True, steps=2, warmup=1), ctx))
    if not args.no_cpu_baseline:
        # the all-cores leg runs alone, after every other CPU leg and GPU measurement
        legs.start("netlib", "netlib", max(2.0, args.cpu_seconds), blas=1)
        netlib_cpu = legs.collect("netlib", timeout=900)
        for key in ("netlib_batch", "netlib_batch_presolve"):
            if "error" not in out[key]:
                out[key]["cpu_baseline"] = netlib_cpu
    return out


def free_port():
    import socket
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        return probe.getsockname()[1]


def launch_ranks(gpus, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process (which never initialises the GPU:
    no torch import, no HIP call, no exec), each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment exactly as
    `torch.distributed.run --nnodes=1 --nproc-per-node N` would set them, wait for all of them, and relay rank 0's stdout (its last line
    is the bench line).  Returns the exit code: non-zero when any rank failed -- a run that cannot give N ranks must not print a line."""
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(free_port())
    children = []
    with tempfile.TemporaryFile(mode="w+") as rank0_stdout:
        for rank in range(gpus):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus), GROUP_RANK="0",
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                             stdout=rank0_stdout if rank == 0 else subprocess.DEVNULL))
        codes = [None] * gpus
        while any(code is None for code in codes):
            for k, child in enumerate(children):
                if codes[k] is None:
                    codes[k] = child.poll()
            if any(code for code in codes):  # one rank died: the others would wait for it in the rendezvous -- stop them (by their own pids)
                for k, child in enumerate(children):
                    if codes[k] is None:
                        child.terminate()
                        try:
                            codes[k] = child.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            child.kill()
                            codes[k] = child.wait()
                break
            time.sleep(0.05)
        rank0_stdout.seek(0)
        out = rank0_stdout.read()
    lines = [row for row in (out or "").splitlines() if row.strip()]
    if any(codes):  # (nothing of rank 0's goes to stdout: a rank that died late -- in destroy_process_group, say -- must not leave a bench line behind)
        sys.stderr.write("bench.py: --gpus %d: exit codes of the ranks %s\n" % (gpus, codes))
        for row in lines:
            sys.stderr.write("[rank 0] " + row + "\n")
        return next(code for code in codes if code) or 1
    for row in lines[:-1]:
        print(row)
    try:
        line = json.loads(lines[-1])
    except (IndexError, ValueError):
        sys.stderr.write("bench.py: --gpus %d: rank 0 printed no JSON line\n" % gpus)
        return 1
    if line.get("n_gpus") != gpus:
        sys.stderr.write("bench.py: --gpus %d but the ranks report n_gpus %s\n" % (gpus, line.get("n_gpus")))
        return 1
    emit(lines[-1])
    return 0


def launch_check(rank, local_rank, world):
    """`--launch-check`: the rendezvous, a barrier and a SUM of the ranks -- what every workload's timing does around its solves -- and
    nothing else.  gloo where there is no device per rank (the CPU tests), RCCL otherwise."""
    import torch
    shared = os.environ.get("RELP_BENCH_SHARED_DEVICE") == "1" or torch.cuda.device_count() < world
    ranks = [0]
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group(backend="gloo")
            device = "cpu"
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            device = "cuda"
        dist.barrier()
        seen = torch.zeros(world, dtype=torch.int64, device=device)
        seen[rank] = rank + 1
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        ranks = [int(v) - 1 for v in seen.tolist()]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(json.dumps({"launch_check": True, "n_gpus": world, "ranks": ranks, "backend": "gloo" if shared else "nccl"}))


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
    parser.add_argument("--exact-cpu-seconds", type=float, default=150.0,
                        help="budget of the exact CPU port on 25FV47 (same_work_exact_25fv47: the device is timed on the pivots it reached; "
                             "about 400 s reach the optimum on the bench host)")
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-concurrency-probe", action="store_true")
    parser.add_argument("--no-dense-roofline", action="store_true", help="(kept for compatibility: the dense roofline now lives under configs)")
    parser.add_argument("--no-configs", action="store_true", help="default workload only: skip the other BASELINE configs")
    parser.add_argument("--no-certify", action="store_true", help="25fv47: leave the exact certificate out of the timed step (A/B only)")
    parser.add_argument("--carry", type=int, default=0, choices=[0, 1, 2],
                        help="0 explicit inverse, 1 LU + Forrest-Tomlin, 2 LU through the inverses of its triangles + product-form updates (relp_options.carry)")
    parser.add_argument("--lu-refactor", type=int, default=0, choices=[0, 1, 2, 3], help="LU carries: 0 automatic, 1 refactorisation kernels on the device, 2 host core (relp_options.lu_refactor)")
    parser.add_argument("--recorded-exact-25fv47", action="store_true", help="quote the recorded 25FV47 exact run (profiles/r4_exact_25fv47_128_limbs.txt) instead of measuring it")
    parser.add_argument("--presolve", action="store_true", help="apply the reference's presolve before standardisation (its harness order)")
    parser.add_argument("--concurrency", type=int, default=4, help="netlib batch: LPs in flight per GPU")
    parser.add_argument("--records", default=None, help="netlib batch: write one JSON line per solved LP to this file")
    parser.add_argument("--schedule", default="dynamic", choices=["dynamic", "static"], help="netlib batch: work distribution")
    parser.add_argument("--detail", default=os.path.join(ROOT, "bench_configs.json"),
                        help="file that receives the full record (per-kernel tables, every config, CPU samples); the stdout line is compact")
    parser.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)
    parser.add_argument("--launch-check", action="store_true",
                        help="no solve: the ranks rendezvous, reduce their ranks and rank 0 prints {n_gpus, ranks} (runs without a GPU over gloo)")
    args = parser.parse_args()

    if args.cpu_leg:  # child process: one CPU baseline, no GPU
        print(json.dumps(run_cpu_leg(args.cpu_leg, args.cpu_seconds)), flush=True)
        return

    if args.gpus < 1:
        parser.error("--gpus must be at least 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher -- it has not imported torch or touched HIP -- and the N
        # ranks are its children (one per GPU, rendezvous on 127.0.0.1); rank 0's line is relayed as the last line of stdout
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # never report n_gpus of a job that is not the one asked for
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d: launch with --nproc-per-node %d (or without a launcher)\n" % (args.gpus, world, args.gpus))
        sys.exit(2)
    if args.launch_check:
        launch_check(rank, local_rank, world)
        return
    distributed = world > 1 or os.environ.get("RELP_FORCE_DISTRIBUTED") == "1"  # the env switch exercises the RCCL path at N=1
    # RELP_BENCH_SHARED_DEVICE=1 (tests on a 1-GPU box): every rank solves on device 0 and the ranks talk over gloo -- RCCL refuses two
    # ranks on one device.  Everything above the collectives (barriers, MAX / SUM reductions, the shared ticket queue, the gathered
    # records) is the code of the real multi-GPU run.
    shared_device = os.environ.get("RELP_BENCH_SHARED_DEVICE") == "1"
    if shared_device:
        local_rank = 0
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if shared_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if world > 1:
        # the exact certificate assembles its digits on host threads (up to 32 per process): share the cores between the ranks
        os.environ.setdefault("RELP_CERTIFY_THREADS", str(max(2, (os.cpu_count() or 8) // world)))
    ctx = {"rank": rank, "local_rank": local_rank, "world": world, "distributed": distributed, "cache": {},
           "reduce_device": None if (shared_device or not distributed) else "cuda"}
    path = WORKLOADS[args.workload]
    batch_workload = path == "batch"
    graph = isinstance(path, tuple) and path[0] == "maxflow"
    dense = not isinstance(path, str) and not graph
    full = rank == 0 and world == 1 and args.workload == "25fv47" and not args.no_configs and not args.presolve and args.carry == 0

    # CPU legs (reported at N = 1 only) start AFTER the headline's timed region -- nothing competes with the host thread that feeds the
    # GPU while `value` is measured -- and run beside the other configs
    legs = CpuLegs()
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline

    line = netlib_batch(args, ctx) if batch_workload else single_lp(args, ctx)

    if want_cpu and not batch_workload and not graph and not dense:
        legs.start("exact_faithful", "exact_faithful", args.cpu_seconds)
        legs.start("exact_tuned", "exact_tuned", args.cpu_seconds)
        legs.start("f64", "f64", args.cpu_seconds)
    if want_cpu and (dense or full):
        dims = path if dense else WORKLOADS["dense4096"]
        legs.start("dense", "dense:%dx%d" % dims, args.cpu_seconds)
    if want_cpu and (graph or full):
        legs.start("maxflow", "maxflow", args.cpu_seconds)
    if rank == 0:
        if full:
            line["configs"] = all_configs(args, ctx, legs)
            # BASELINE configs[1] as written ("LU carry BasisInverse"): the same LP, same step, under the two LU carries
            for key, name in (("value_lu_carry", "lu_carry_25fv47"), ("value_lu_inverse_carry", "lu_inverse_carry_25fv47"),
                              ("value_lu_inverse_carry_device_refactor", "lu_inverse_carry_25fv47_device_refactor")):
                if "error" not in line["configs"].get(name, {"error": 1}):
                    line[key] = line["configs"][name]["value"]
            same = line["configs"].get("exact_" + EXACT_SAME_WORK_LP.lower(), {}).get("same_work")
            if same:
                line["same_work_exact"] = same
            same_25 = line["configs"].get("exact_25fv47", {}).get("same_work")
            if same_25:  # the metric's LP itself, the reference's 2392 pivots in exact arithmetic on both sides (the CPU side recorded)
                line["same_work_exact_25fv47"] = same_25
        if want_cpu:
            if batch_workload:
                legs.start("netlib", "netlib", max(2.0, args.cpu_seconds), blas=1)
                line["cpu_baseline"] = legs.collect("netlib", timeout=900)
            elif dense:
                line["cpu_baseline"] = legs.collect("dense")
            elif graph:
                line["cpu_baseline"] = legs.collect("maxflow")
            else:
                line["cpu_baseline"] = legs.collect("exact_faithful")
                line["cpu_baseline_tuned"] = legs.collect("exact_tuned")
                line["cpu_baseline_f64"] = legs.collect("f64")
        line["host"] = host_description()
    if distributed:
        torch.distributed.destroy_process_group()
    if rank == 0:
        write_detail(line, args.detail)
        emit(compact_line(line, os.path.relpath(args.detail, ROOT) if args.detail.startswith(ROOT) else args.detail))


if __name__ == "__main__":
    main()
