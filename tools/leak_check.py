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
            if name == "SC105" and carry == 0:  # (round 6: the wide types -- matrix-core update, both buffers of N, widening)
                s.solve_exact(first_limbs=16, max_limbs=64)
            s.close()
    from relp_amd.basis_inverse import ExactBasisInverse  # (round 6: relp_bix_* through its widenings)
    from fractions import Fraction
    import random
    rng = random.Random(rnd)
    bix = ExactBasisInverse.invert([[(j, Fraction(5 + j, 3))] + [(i, Fraction(rng.randint(-9, 9), rng.randint(1, 5))) for i in range(12) if i != j and rng.random() < 0.3]
                                    for j in range(12)])
    for step in range(12):
        alpha = bix.left_multiply_by_basis_inverse([(i, Fraction(10 ** 9 + 7 * i + step, 3 + i)) for i in range(12)])
        bix.change_basis(next(i for i in range(12) if alpha[i] != 0))
    bix.close()
    f = free()
    if base is None:
        base = f
    print("round", rnd, "free MB", f / 1e6, "delta vs first", (f - base) / 1e6, flush=True)
