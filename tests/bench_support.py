"""Running ``bench.py`` the way the driver does: the LAST stdout line must be one compact JSON object that parses on its own from
the final 4096 bytes of stdout (the driver's capture is bounded; round 3's 35 KB line did not fit it), and the full record lives in
the detail file named by ``detail_file``."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAIL_BYTES = 4096


def run_bench(arguments, env=None, timeout=900, launcher=None):
    """Returns (compact line, full detail record, completed process)."""
    with tempfile.TemporaryDirectory() as scratch:
        detail = os.path.join(scratch, "bench_configs.json")
        command = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + list(arguments) + ["--detail", detail]
        out = subprocess.run(command, env=env, capture_output=True, text=True, timeout=timeout)
        assert out.returncode == 0, out.stderr[-2000:]
        tail = out.stdout.encode()[-TAIL_BYTES:].decode(errors="replace")
        last = tail.strip().splitlines()[-1]
        assert last == out.stdout.strip().splitlines()[-1], "the last line does not fit the final %d bytes of stdout" % TAIL_BYTES
        assert len(last.encode()) < 4000
        line = json.loads(last)
        assert isinstance(line, dict) and line["detail_file"]
        full = json.load(open(detail))
    return line, full, out
