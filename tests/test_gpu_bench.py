"""The driver's contract for ``bench.py`` (``-m gpu``): one JSON line as the LAST line of stdout with the named fields, the
roofline object of the dominant kernel and the CPU baseline -- the default workload, end to end as a child process."""
import os

import pytest

from bench_support import run_bench

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_contract():
    short, line, _ = run_bench(["--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--no-configs", "--no-concurrency-probe"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
        assert key in short, key
    # the compact line (what the driver parses) agrees with the detail file
    assert abs(short["value"] - line["value"]) <= 1e-5 * line["value"] and short["unit"] == "pivots/s" and short["dtype"] == "f64"
    assert short["config"]["carry"] == "explicit" and short["config"]["certified"] is True and short["config"]["objective_bits"] == 1791
    assert abs(short["config"]["objective"] - 5501.8458883) < 1e-6 and short["config"]["pivots_per_solve"] == line["config"]["pivots_per_solve"]
    assert short["roofline"]["kernel"] == line["roofline"]["kernel"] and short["roofline"]["peak"] == 8000.0
    assert abs(short["roofline"]["frac"] - short["roofline"]["achieved"] / 8000.0) < 1e-9 * short["roofline"]["frac"] + 1e-15
    assert short["roofline"]["seconds_per_launch"] > 0 and short["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert short["cpu_baseline"]["kind"] == "port" and short["cpu_baseline"]["mode"] == "faithful" and short["cpu_baseline"]["cores"] == 1
    assert short["cpu_baseline"]["value"] > 0 and short["cpu_baseline"]["nproc"] >= 1 and short["cpu_baseline"]["cpu_model"] and short["cpu_baseline"]["sample"]
    assert short["cpu_baseline_f64_tuned"]["seconds"] > 0 and "HiGHS" in short["cpu_baseline_f64_tuned"]["name"]
    assert line["unit"] == "pivots/s" and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "f64"
    assert "25FV47" in line["config"]["workload"] and "model" not in line["config"]
    # the certificate is inside the timed step: value = pivots / (f64 loop + certificate), and it holds
    exact = line["config"]["exact"]
    assert exact["certified"] is True and exact["objective_bits"] == 1791
    assert line["config"]["wall_clock_to_exact_optimum_s"] >= line["config"]["wall_clock_f64_loop_s"] > 0
    assert abs(line["value"] - line["config"]["pivots_per_solve"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    roofline = line["roofline"]
    assert roofline["bound"] in ("hbm", "mfma") and roofline["unit"] in ("GB/s", "TFLOP/s")
    assert roofline["peak"] == 8000.0 and abs(roofline["frac"] - roofline["achieved"] / roofline["peak"]) < 1e-12
    assert roofline["kernel"] == max(roofline["kernels"], key=lambda k: roofline["kernels"][k]["seconds_per_launch"])  # by measured time
    assert roofline["traffic"] is None or roofline["traffic"] > 0
    # `frac` is priced on the contract's bytes; the kernel's own bytes are reported beside it
    assert roofline["contract_bytes_per_pivot"]["per_pivot"] == roofline["contract_bytes_per_pivot"]["pricing"] + roofline["contract_bytes_per_pivot"]["ftran_btran_vectors"]
    assert roofline["kernel_bytes_per_launch"] > 0 and roofline["factor_nonzeros"] > 821
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert cpu["nproc"] >= 1 and cpu["cpu_model"]  # stated in every report
    assert line["cpu_baseline_tuned"]["mode"] == "tuned" and line["cpu_baseline_tuned"]["value"] > 0
    assert line["value"] > cpu["value"]  # (no fixed factor: the suite may share the GPU with another test process)
    assert line["config"]["tolerances"]["tol_dual"] == 1e-9 and "harris" in line["config"]["ratio_rule"]
    assert line["config"]["exact"]["solution_exact_nonzeros"] > 0


def test_netlib_batch_line_through_the_library_batch_entry():
    """`bench.py --workload netlib --steps 1`: the batch runner (relp_batch_run, four LPs in flight) on one GPU."""
    short, line, _ = run_bench(["--workload", "netlib", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    config = line["config"]
    assert short["config"]["tickets_per_rank"] == [45] and short["config"]["objectives_outside_reference_tolerance"] == []
    assert short["config"]["lps_in_flight_per_gpu"] == 4 and short["value"] > 1000
    assert line["n_gpus"] == 1 and config["tickets_per_rank"] == [45] and config["objectives_outside_reference_tolerance"] == []
    assert config["lps_in_flight_per_gpu"] == 4 and len(config["workers_per_rank"][0]) == 4
    assert sum(w["tickets"] for w in config["workers_per_rank"][0]) == 45
    assert 0 < config["single_pass_makespan_s"] <= config["makespan_s"] and config["longest_lp"]
    assert line["value"] > 1000 and "suite throughput" in config["throughput_kind"]


def test_default_bench_compact_line_and_detail_file_with_every_baseline_config():
    """The driver's default invocation: the LAST stdout line is compact (parses from the final 4096 bytes: checked by run_bench) and
    carries the headline, its roofline and CPU baseline, the LU carries' figures on the same LP at top level and a summary of every
    other BASELINE config; the detail file carries every config with its own value, roofline and CPU baseline (short CPU legs here)."""
    short, line, _ = run_bench(["--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--exact-cpu-seconds", "20"], timeout=1500)
    configs = line["configs"]
    assert set(short["configs_summary"]) == set(configs)
    for name, triple in short["configs_summary"].items():
        assert isinstance(triple, list) and len(triple) == 3 and triple[0] > 0 and triple[1] > 0 and triple[2] > 0, (name, triple)
        assert abs(triple[0] - configs[name]["value"]) <= 1e-5 * triple[0]
    assert abs(short["value_lu_carry"] - configs["lu_carry_25fv47"]["value"]) <= 1e-5 * short["value_lu_carry"]
    assert abs(short["value_lu_inverse_carry"] - configs["lu_inverse_carry_25fv47"]["value"]) <= 1e-5 * short["value_lu_inverse_carry"]
    assert short["roofline"]["frac"] > 0 and short["cpu_baseline"]["value"] > 0 and short["value"] > short["cpu_baseline"]["value"]
    assert set(configs) == {"lu_carry_25fv47", "lu_inverse_carry_25fv47", "lu_inverse_carry_25fv47_device_refactor", "dense4096_f64", "dense4096_narrowest",
                            "netlib_batch", "netlib_batch_presolve", "maxflow_reference_start", "maxflow_crash", "exact_e226", "exact_25fv47"}
    for name, entry in configs.items():
        assert "error" not in entry, (name, entry)
        assert entry["value"] > 0 and entry["ms_per_step"] > 0 and entry["unit"] == "pivots/s", name
        if name.startswith("exact_"):  # the update of N on the matrix cores: priced in word products against the dense i8 MFMA rate
            assert entry["roofline"]["bound"] == "mfma" and entry["roofline"]["unit"] == "T word products/s" and entry["roofline"]["peak"] == 2500.0 / 64
            assert 0 < entry["roofline"]["frac"] < 1 and entry["roofline"]["word_products_issued"] >= entry["roofline"]["word_products_needed"] > 0
        else:
            assert entry["roofline"]["frac"] > 0 and entry["roofline"]["peak"] == 8000.0, name
        assert entry["cpu_baseline"]["value"] > 0 and (entry.get("recorded") or entry["cpu_baseline"].get("recorded") or entry["cpu_baseline"]["nproc"] >= 1), name
    assert configs["lu_carry_25fv47"]["config"]["carry"] == "lu" and configs["lu_carry_25fv47"]["config"]["exact"]["certified"] is True
    assert configs["lu_inverse_carry_25fv47"]["config"]["carry"] == "lu_inverse" and configs["lu_inverse_carry_25fv47"]["config"]["exact"]["certified"] is True
    assert configs["lu_inverse_carry_25fv47"]["roofline"]["kernel"] == "lu_pivot"
    device = configs["lu_inverse_carry_25fv47_device_refactor"]["config"]
    assert device["carry"] == "lu_inverse" and "device" in device["lu_refactor"] and device["exact"]["certified"] is True and device["refactors"] > 0
    assert abs(short["value_lu_inverse_carry_device_refactor"] - configs["lu_inverse_carry_25fv47_device_refactor"]["value"]) <= 1e-5 * short["value_lu_inverse_carry_device_refactor"]
    # the same work on both sides: the reference's pivot sequence of E226 in exact arithmetic, device and CPU restatement, to completion
    exact = configs["exact_e226"]
    assert exact["config"]["matches_golden_optimum_and_pivot_counts"] is True and exact["same_work"]["same_pivot_count"] is True
    assert short["same_work_exact"]["lp"] == "E226" and short["same_work_exact"]["cpu_over_gpu"] > 0
    # ... and of the metric's LP itself, measured in this run: 1133 + 1259 pivots at 128 limbs, optimum and counts those of the golden file
    whole = configs["exact_25fv47"]
    assert whole["recorded"] is False and whole["config"]["limbs"] == 128 and whole["config"]["pivots_per_solve"] == 2392
    assert whole["config"]["matches_golden_optimum_and_pivot_counts"] is True
    same = short["same_work_exact_25fv47"]  # CPU port and device on the same first P pivots, same host, same run
    assert same["cpu_over_gpu"] > 2.0 and same["matches_golden"] is True and same["same_run_same_host"] is True
    assert same["pivots"] == same["pivots_gpu"] > 100 and same["cpu_seconds_measured"] >= 19.0
    assert abs(configs["dense4096_f64"]["config"]["objective"] + 202885.40946447) < 1e-4
    assert abs(configs["dense4096_narrowest"]["config"]["objective"] + 202885.40946447) < 1e-4
    assert configs["dense4096_f64"]["roofline"]["kernel"] == "price" and configs["dense4096_f64"]["roofline"]["frac"] > 0.4
    assert configs["netlib_batch"]["config"]["objectives_outside_reference_tolerance"] == []
    assert configs["netlib_batch_presolve"]["config"]["objectives_outside_reference_tolerance"] == []
    assert abs(configs["maxflow_reference_start"]["config"]["objective"] + 778.0) < 1e-9
    assert abs(configs["maxflow_crash"]["config"]["objective"] + 778.0) < 1e-9
    assert configs["maxflow_crash"]["cpu_baseline"]["scipy_max_flow"]["flow_value"] == 778
