"""The grid barrier behind the exact simplex's one-launch solve (relp_amd/csrc/grid_barrier.hpp; ``-m gpu``).

Round-5 review, weak #1: the barrier's correctness rests on "a die's last arrival releases for every workgroup of that die" and no
test in the suite checked it; a 16-limb run with the update on the matrix cores had hung on ISRAEL, unexplained.  Here:

* the exchange test through ``relp_debug_grid_barrier``: >= 10^5 barriers at 512 / 256 / 7 workgroups, every thread of every workgroup
  writes a fresh value per round and reads other workgroups' (every round from every die, rotating through all pairs): zero stale reads;
* the watchdog: a launch whose workgroups make different numbers of barriers ends itself and says so (no hung device);
* the whole 25FV47 golden trace at ``exact_grid`` in {64, 256, 512} (grid independence on the metric's LP);
* ISRAEL and the other 16-limb LPs pivot for pivot with the update on the matrix cores AT 16 limbs (the width that hung).
"""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

import relp_amd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = {os.path.basename(p)[:-5]: json.load(open(p)) for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.json"))}


def barrier_run(grid, rounds, reads=8, mode=0, limit_ticks=0):
    out = np.zeros(8, dtype=np.int64)
    fn = relp_amd.lib().relp_debug_grid_barrier
    fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int64)]
    fn.restype = C.c_int32
    status = fn(0, grid, rounds, reads, mode, limit_ticks, out.ctypes.data_as(C.POINTER(C.c_int64)))
    assert status == 0, status
    return {"stale": int(out[0]), "first": int(out[1]), "dies": int(out[2]), "ticks": int(out[3]), "rounds": int(out[4]),
            "finished": int(out[5]), "abort": int(out[6]), "waiting": int(out[7])}


@pytest.mark.parametrize("grid", [512, 256, 7])
def test_a_hundred_thousand_barriers_and_no_stale_value(grid):
    got = barrier_run(grid, 100_000)
    assert got["abort"] == 0 and got["finished"] == grid, got
    assert got["stale"] == 0, "stale or wrong values behind the barrier: %r" % (got,)
    assert got["dies"] == min(8, grid)  # (consecutive workgroups go to different dies: the test reads across all of them)
    microseconds = got["ticks"] / 100.0 / got["rounds"]
    print("grid %d: %.2f us per round (store, barrier, %d reads)" % (grid, microseconds, 8))
    assert microseconds < 60.0  # (6.4 us at 512 workgroups by itself; cooperative_groups' grid.sync() was 53)


def test_every_pair_of_workgroups_exchanges():
    """With reads = grid - 1 every round reads EVERY other workgroup's value (fewer rounds: 31 reads a thread)."""
    got = barrier_run(32, 20_000, reads=31)
    assert got["abort"] == 0 and got["stale"] == 0 and got["finished"] == 32, got


def test_the_watchdog_ends_a_launch_whose_barrier_counts_differ():
    got = barrier_run(64, 1000, mode=1, limit_ticks=20_000_000)  # 0.2 s
    assert got["abort"] != 0, got           # somebody gave up ...
    assert got["waiting"] == 63, got        # ... everybody but the workgroup that left a barrier out was found waiting
    assert got["finished"] == 1, got
    # ... and the device is fine afterwards
    again = barrier_run(64, 1000)
    assert again["abort"] == 0 and again["stale"] == 0 and again["finished"] == 64, again


def device_indices(pivots, n_art):
    return [(ph, q + (n_art if ph == 2 else 0), p, leaving + (n_art if ph == 2 else 0)) for ph, q, p, leaving in pivots]


@pytest.mark.parametrize("grid", [64, 256, 512])
def test_25fv47_whole_trace_at_any_grid(grid):
    golden = GOLDEN["25FV47"]
    solver = relp_amd.Solver(exact_grid=grid).load_mps(os.path.join(ROOT, golden["file"]))
    got = solver.solve_exact(first_limbs=4, max_limbs=128)
    assert got["status"] == 1, (got["status"], got["survived"])
    assert got["objective"] == golden["objective"]
    assert got["trace"] == device_indices([tuple(t) for t in golden["trace"]], solver.n_art)
    assert all(r["grid"] <= grid for r in solver.exact_counters())
    solver.close()


# the LPs whose solve ends at (or passes through) 16 limbs -- ISRAEL is the one the round-5 matrix-core path hung on at that width
AT_16_LIMBS = ["ISRAEL", "BLEND", "STOCFOR1", "LOTFI", "BEACONFD", "BOEING1", "STANDMPS", "SHARE1B", "E226", "BRANDY"]


@pytest.mark.parametrize("name", AT_16_LIMBS)
@pytest.mark.parametrize("update", [0, 4, 1])
def test_the_update_on_the_matrix_cores_at_16_limbs(name, update):
    """Started AT 16 limbs so that every pivot the LP makes at that width runs there; exact_update 0 = matrix cores (fused epilogue), 4 = matrix cores in two passes, 1 = vector unit."""
    golden = GOLDEN[name]
    if golden.get("status") != "optimal":
        pytest.skip("no optimum in the fixture")
    solver = relp_amd.Solver(exact_update=update).load_mps(os.path.join(ROOT, golden["file"]))
    got = solver.solve_exact(first_limbs=16, max_limbs=64)
    assert got["status"] == 1, (got["status"], got["survived"])
    assert (got["pivots_phase_one"], got["pivots_phase_two"]) == (golden["pivots_phase1"], golden["pivots_phase2"])
    assert got["objective"] == golden["objective"]
    head = device_indices([tuple(t) for t in golden.get("trace", golden["trace_head"])], solver.n_art)
    assert got["trace"][:len(head)] == head
    assert got["survived"][0][0] == 16 and got["survived"][0][1] > 0  # (pivots were made at 16 limbs)
    solver.close()
