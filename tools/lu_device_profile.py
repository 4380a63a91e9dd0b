"""Diagnostic (GPU): the refactorisation kernels on bases taken from real solves -- time, rounds, fill and the in-kernel cycle sums.

    python tools/lu_device_profile.py [LP[:fraction] ...] [--dense-tail 0,8,32]

`relp_lu_factor_device` runs lu_factor_kernel (and, `inverted`, lu_invert_kernel) twice and times the second run with HIP events.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from relp_amd.basis_inverse import lu_factor_device, lu_factor_host  # noqa: E402
from test_gpu_lu_factor_device import basis_columns  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument("lps", nargs="*", default=["25FV47:1.0", "BNL1:1.0", "GREENBEA:0.5"])
parser.add_argument("--dense-tail", default="0,8,32")
args = parser.parse_args()
for spec in args.lps:
    name, _, fraction = spec.partition(":")
    columns = basis_columns(name, float(fraction or 1.0))
    m = len(columns)
    host = lu_factor_host(columns)
    host_inverse = lu_factor_host(columns, inverted=True)
    print("%s at %.0f %%: m %d, nnz(B) %d; host Markowitz: nnz(L) + nnz(U) %d, inverted triangles %d entries" % (
        name, 100 * float(fraction or 1.0), m, sum(len(c) for c in columns), host["nnz_lower"] + host["nnz_upper"], host_inverse["nnz_lower"] + host_inverse["nnz_upper"]))
    for tail in [int(t) for t in args.dense_tail.split(",")]:
        f = lu_factor_device(columns, dense_tail=tail)
        inv = lu_factor_device(columns, dense_tail=tail, inverted=True)
        stamps = [16 * v // 1000 for v in f["info"][12:23]]
        print("  dense tail %2d: %2d rounds (%d in LDS) + %2d dense rows, nnz(L) + nnz(U) %5d, factorisation %7.1f us, with the inversion %7.1f us, inverted triangles %6d entries" % (
            tail, f["info"][3], f["info"][10], f["info"][4], f["nnz_lower"] + f["nnz_upper"], f["info"][31] / 10.0, inv["info"][31] / 10.0, inv["info"][7] + inv["info"][8]))
        print("     factorisation kcycles: load %d | candidates %d | competition %d | conflicts %d | accept %d | U rows + targets %d | layout %d | copy + eliminate %d | "
              "reset %d | dense tail %d | finalisation %d" % tuple(stamps))
        print("     inversion kcycles: L^-1 block %d wall (waves summed: waiting %d | streaming + accumulating %d | emitting %d); U^-1 block %d wall (%d | %d | %d)" % (
            tuple(inv["info"][23:27]) + tuple(inv["info"][19:23])))
