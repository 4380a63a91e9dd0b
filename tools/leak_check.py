import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, relp_amd
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]
base = None
for rnd in range(6):
    for name in ("AFIRO", "SC105", "ADLITTLE", "25FV47"):
        for carry in (0, 1, 2):
            s = relp_amd.Solver(certify=1, carry=carry).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
            for _ in range(3):
                s.solve_relaxation()
            if name == "AFIRO":
                s.solve_exact(first_limbs=1, max_limbs=4)
            s.close()
    f = free()
    if base is None:
        base = f
    print("round", rnd, "free MB", f / 1e6, "delta vs first", (f - base) / 1e6, flush=True)
