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


def test_gpus_flag_launches_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with NO torch.distributed.run around it (the shape of the driver's N = 1 command with another N): the
    process starts the two ranks itself and rank 0 reports n_gpus == 2.  `--launch-check` is the rendezvous + barrier + reduction of every
    workload without the solves, so it runs here over gloo; the GPU twin with a real workload is
    tests/test_network.py::test_bench_two_ranks_without_a_launcher."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [0, 1]


def test_gpus_flag_that_disagrees_with_the_launcher_fails_loudly():
    """A launcher that gave another world size than --gpus asks for: non-zero exit and no bench line (round 4 printed n_gpus 1)."""
    import subprocess
    import sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr and out.stdout.strip() == ""


def test_compact_line_is_bounded_whatever_the_record_holds():
    """Fields the trimming loop never looked at (config values, host, the f64 baselines) blown up: the line still fits and still
    carries the contract's fields; a roofline with a peak and no achieved figure does not raise."""
    bench = load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r3_bench_default.json")))
    full["host"] = {"cpu_model": "z" * 6000, "nproc": 256}
    full["metric"] = "m" * 3000
    full["config"]["carry_per_lp"] = {("LP%04d" % k): "explicit" for k in range(600)}
    full["roofline"]["achieved"] = None
    text = bench.compact_line(full, "bench_configs.json")
    assert len(text.encode()) <= bench.COMPACT_LIMIT
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
