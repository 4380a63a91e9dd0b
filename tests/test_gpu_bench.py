"""The driver's contract for ``bench.py`` (``-m gpu``): one JSON line as the LAST line of stdout with the named fields, the
roofline object of the dominant kernel and the CPU baseline -- the default workload, end to end as a child process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "2",
                          "--no-dense-roofline", "--no-concurrency-probe"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
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
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert line["value"] > cpu["value"]  # (no fixed factor: the suite may share the GPU with another test process)
