"""Diagnostic: does the engine clock drop while the one-workgroup LU kernel is the only thing running?  Solves 25FV47 under the LU
carry alone, then beside a 'heater' (another stream kept busy), and samples rocm-smi's sclk meanwhile."""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import relp_amd

def sample_clocks(stop, out):
    while not stop.is_set():
        try:
            text = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            for line in text.splitlines():
                if "sclk" in line:
                    out.append(line.strip())
        except Exception as e:  # noqa: BLE001
            out.append("rocm-smi failed: %r" % e)
        time.sleep(0.05)

def solve(carry, repeats=6):
    s = relp_amd.Solver(carry=carry).load_mps(os.path.join(ROOT, "data", "netlib", "25FV47.SIF"))
    s.solve_relaxation()
    best = 1e9
    for _ in range(repeats):
        r = s.solve_relaxation()
        best = min(best, r.solve_seconds * 1e6 / (r.pivots_phase_one + r.pivots_phase_two))
    s.close()
    return best

def heater(kind, stop):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        if kind == "matmul":
            a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
            while not stop.is_set():
                for _ in range(20):
                    a @ a
                stream.synchronize()
        else:
            a = torch.zeros(1 << 24, device="cuda")
            while not stop.is_set():
                for _ in range(50):
                    a.add_(1.0)
                stream.synchronize()

for kind in (None, "elementwise", "matmul"):
    stop = threading.Event()
    clocks = []
    threads = [threading.Thread(target=sample_clocks, args=(stop, clocks))]
    if kind:
        threads.append(threading.Thread(target=heater, args=(kind, stop)))
    for t in threads:
        t.start()
    time.sleep(0.5)
    lu = solve(1)
    explicit = solve(0)
    stop.set()
    for t in threads:
        t.join()
    print("heater %-12s LU %.1f us/pivot   explicit %.1f us/pivot   sclk samples: %s" % (kind, lu, explicit, sorted(set(clocks))[:6]), flush=True)
