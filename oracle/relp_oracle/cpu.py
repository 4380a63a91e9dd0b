"""Python front end of the C++ CPU oracle ``oracle/cpp/relp_cpu.cpp`` (oracle; test infrastructure only).

Used by ``tests/`` (parity of the compiled twin with the Fraction oracle and the golden vectors) and by ``bench.py``'s
``cpu_baseline`` leg.  The product never imports this.
"""
import json
import os
import subprocess
import tempfile
from fractions import Fraction

from .dump import dump_provider

CPP_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp")
BINARY = os.path.join(CPP_DIR, "_build", "relp_cpu")
RATIONAL_CHECK = os.path.join(CPP_DIR, "_build", "rational_check")


def ensure_built():
    """Build with g++ when the binaries are missing or older than their sources."""
    sources = [os.path.join(CPP_DIR, name) for name in ("relp_cpu.cpp", "rational.hpp", "rational_check.cpp")]
    newest = max(os.path.getmtime(path) for path in sources)
    if all(os.path.exists(b) and os.path.getmtime(b) >= newest for b in (BINARY, RATIONAL_CHECK)):
        return
    subprocess.check_call(["make", "-C", CPP_DIR, "-s"])


def solve_dump(path, max_pivots=None, max_seconds=None, rule=None, tuned=False, trace=64, timeout=None):
    """Run the binary on a problem dump; returns its JSON record."""
    ensure_built()
    command = [BINARY, path, "--trace", str(trace)]
    if max_pivots is not None:
        command += ["--max-pivots", str(max_pivots)]
    if max_seconds is not None:
        command += ["--max-seconds", str(max_seconds)]
    if rule is not None:
        command += ["--rule", rule]
    if tuned:
        command.append("--tuned")
    done = subprocess.run(command, capture_output=True, text=True, timeout=timeout)
    if done.returncode != 0:
        raise RuntimeError("relp_cpu failed: %s" % done.stderr.strip())
    record = json.loads(done.stdout)
    if "solution" in record:
        record["solution"] = [(j, Fraction(v)) for j, v in record["solution"]]
        record["objective"] = Fraction(record["objective"])
    return record


def solve_provider(provider, route=None, **options):
    """Dump ``provider`` (see ``dump.dump_provider``) and solve it with the C++ oracle."""
    with tempfile.TemporaryDirectory() as directory:
        path = os.path.join(directory, "problem.txt")
        dump_provider(provider, path, route=route)
        return solve_dump(path, **options)
