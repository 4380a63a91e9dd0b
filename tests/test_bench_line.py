"""bench.py's compact line (CPU): round 3's full record -- 35 KB, which the driver's 8 KB capture could not hold -- must come out
under 4000 bytes with every contract field, the roofline, the CPU baseline and a summary of every config."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("relp_bench", os.path.join(ROOT, "bench.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def test_round_three_record_fits_the_compact_line():
    bench = load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_default.json")))
    assert len(json.dumps(full)) > 30000
    full["value_lu_carry"] = full["configs"]["lu_carry_25fv47"]["value"]
    full["value_lu_inverse_carry"] = full["configs"]["lu_inverse_carry_25fv47"]["value"]
    text = bench.compact_line(full, "bench_configs.json")
    assert "\n" not in text and len(text.encode()) < bench.COMPACT_LIMIT <= 4000
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert "model" not in line["config"] and "25FV47" in line["config"]["workload"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "seconds_per_launch"):
        assert key in line["roofline"], key
    assert isinstance(line["roofline"]["seconds_per_launch"], float)
    for key in ("value", "unit", "cores", "kind", "mode", "cpu_model", "nproc", "sample"):
        assert key in line["cpu_baseline"], key
    assert set(line["configs_summary"]) == set(full["configs"])
    assert line["value_lu_carry"] > 0 and line["value_lu_inverse_carry"] > 0
    assert abs(line["config"]["objective"] - full["config"]["objective"]) < 1e-6


def test_oversized_fields_are_cut_not_the_line():
    bench = load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_default.json")))
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["configs"] = {("config_%03d" % k): full["configs"]["dense4096_f64"] for k in range(80)}
    text = bench.compact_line(full, "bench_configs.json")
    assert len(text.encode()) < bench.COMPACT_LIMIT
    line = json.loads(text)
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_batch_record_fits_the_compact_line():
    bench = load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_default.json")))["configs"]["netlib_batch"]
    line = json.loads(bench.compact_line(full, "bench_configs.json"))
    assert line["scaling"] == "strong" and line["config"]["tickets_per_rank"] == [90] and line["config"]["longest_lp"]
