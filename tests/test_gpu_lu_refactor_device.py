"""The LU refactorisation as kernels inside the solve loop (`relp_options.lu_refactor = RELP_REFACTOR_DEVICE`; -m gpu):
`BasisInverse::invert` (lower_upper/mod.rs:78-92) -- Markowitz factorisation with independent pivots per round, inversion of the
two triangles, compact slot records -- runs on the handle's stream with no basis read-back and no upload, and the inverse-factor
carry solves with what the kernels left in device memory.

* every golden LP reaches its bit-exact certified optimum with the refactorisation on the device, refactorisations counted;
* the device path and the host path agree on the optimum (the pivot sequences may differ: different but equivalent factors);
* what the kernels cannot take -- here a dense basis, rows of more than 256 entries -- comes back as ST_REFACTOR_FAILED, the host
  factorises instead and the solve goes on (counted in the handle's record);
* two solves on the device path are identical (nothing depends on the order in which atomics land).
"""
import glob
import json
import os

import pytest

import relp_amd
from relp_amd.api import CARRY_LU_INVERSE

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = {os.path.basename(p)[:-5]: json.load(open(p)) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.json"))}
OPTIMAL = sorted(name for name, g in GOLDEN.items() if g.get("status") == "optimal" and "file" in g)
REFACTOR_DEVICE, REFACTOR_HOST, REFACTOR_DEVICE_ASYNC = 1, 2, 3


@pytest.mark.parametrize("name", OPTIMAL)
def test_certified_optimum_with_the_refactorisation_on_the_device(name):
    golden = GOLDEN[name]
    solver = relp_amd.Solver(carry=CARRY_LU_INVERSE, lu_refactor=REFACTOR_DEVICE, certify=1, refactor_period=7).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
    assert solver.objective_exact() == golden["objective"]
    record = solver.record()
    assert record["lu_refactor"] == "device"
    pivots = result.pivots_phase_one + result.pivots_phase_two
    if pivots > 16:
        assert result.refactors >= pivots // 8 - 1 and record["device_refactor_fallbacks"] == 0
    solver.close()


@pytest.mark.parametrize("name", ["25FV47", "BNL1", "SCFXM1", "E226", "SHARE1B"])
def test_device_and_host_refactorisation_agree(name):
    path = os.path.join(ROOT, "data", "netlib", name + ".SIF")
    results = {}
    for where in (REFACTOR_DEVICE, REFACTOR_HOST):
        solver = relp_amd.Solver(carry=CARRY_LU_INVERSE, lu_refactor=where, certify=1).load_mps(path)
        first = solver.solve_relaxation()
        again = solver.solve_relaxation()
        assert first.kind == relp_amd.FINITE_OPTIMUM and first.certified
        # deterministic: the second solve of the handle repeats the first pivot for pivot
        assert (first.pivots_phase_one, first.pivots_phase_two, first.objective) == (again.pivots_phase_one, again.pivots_phase_two, again.objective)
        results[where] = (solver.objective_exact(), first.refactors, solver.record()["lu_refactor"])
        solver.close()
    assert results[REFACTOR_DEVICE][0] == results[REFACTOR_HOST][0]
    assert results[REFACTOR_DEVICE][2] == "device" and results[REFACTOR_HOST][2] == "host"
    assert results[REFACTOR_DEVICE][1] > 0 and results[REFACTOR_HOST][1] > 0


def test_a_basis_the_kernels_do_not_take_falls_back_to_the_host(monkeypatch):
    """The kernels give up on what does not fit them (a work arena the active sub-matrix outgrows, a row of more than 256 entries, a
    factor beyond its capacity): the pack kernel then leaves ST_REFACTOR_FAILED in the control block, the pivots enqueued behind it
    are no-ops, the host factorises the same basis at the next read of the control block and the solve goes on.  Forced here by an
    arena of 64 entries (RELP_LUF_ARENA_CAP, read when the handle's work memory is sized): every refactorisation falls back."""
    monkeypatch.setenv("RELP_LUF_ARENA_CAP", "64")
    golden = GOLDEN["SCFXM1"]
    solver = relp_amd.Solver(carry=CARRY_LU_INVERSE, lu_refactor=REFACTOR_DEVICE, certify=1).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    record = solver.record()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified and solver.objective_exact() == golden["objective"]
    assert record["lu_refactor"] == "device" and record["device_refactor_fallbacks"] == result.refactors > 0
    solver.close()


def test_auto_is_the_host_path_today():
    solver = relp_amd.Solver(carry=CARRY_LU_INVERSE).load_mps(os.path.join(ROOT, "data", "netlib", "AFIRO.SIF"))
    solver.solve_relaxation()
    assert solver.record()["lu_refactor"] == "host"
    solver.close()


@pytest.mark.parametrize("name", OPTIMAL)
def test_certified_optimum_with_the_refactorisation_beside_the_pivots(name):
    """Round 5 (`RELP_REFACTOR_DEVICE_ASYNC`): the next factors are built on a second stream from a snapshot of the basis while the
    pivots go on, the etas made meanwhile are logged and replayed onto the new factors, the handle swaps sets.  Every golden LP
    reaches its bit-exact certified optimum; on the LPs long enough to need it, refactorisations really were taken that way."""
    golden = GOLDEN[name]
    solver = relp_amd.Solver(carry=CARRY_LU_INVERSE, lu_refactor=REFACTOR_DEVICE_ASYNC, certify=1).load_mps(os.path.join(ROOT, golden["file"]))
    result = solver.solve_relaxation()
    assert result.kind == relp_amd.FINITE_OPTIMUM and result.certified
    assert solver.objective_exact() == golden["objective"]
    record = solver.record()
    assert "beside the pivots" in record["lu_refactor"]
    pivots = result.pivots_phase_one + result.pivots_phase_two
    if pivots > 200 and solver.m <= 4000:
        assert record["async_refactors"] >= 2, record
    again = solver.solve_relaxation()  # the handle is reusable, and what it does is deterministic in its result
    assert again.kind == relp_amd.FINITE_OPTIMUM and again.certified and solver.objective_exact() == golden["objective"]
    solver.close()
