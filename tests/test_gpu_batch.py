"""``relp_batch_*``: independent LPs served from one ticket queue by host threads inside the library."""
import json
import os
import threading

import pytest

import relp_amd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["AFIRO", "SC50A", "ADLITTLE", "BLEND", "SHARE2B", "KB2", "SC105", "LOTFI"]


def golden(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", name + ".json")))


def test_batch_solves_every_ticket_with_the_exact_optimum():
    models = [relp_amd.Model(os.path.join(ROOT, "data", "netlib", n + ".SIF")) for n in NAMES]
    batch = relp_amd.Batch(models, devices=(0,), workers_per_device=3, certify=1)
    assert batch.n_workers == 3
    schedule = list(range(len(NAMES))) * 2  # two passes over the suite, one queue
    entries, workers, makespan = batch.run(schedule)
    assert makespan > 0 and sum(w.tickets for w in workers) == len(schedule)
    for t, e in enumerate(entries):
        assert e.status == 0 and e.model == schedule[t] and 0 <= e.worker < 3
        assert e.result.kind == relp_amd.FINITE_OPTIMUM and e.result.certified == 1
        assert batch.objective_exact(t) == golden(NAMES[e.model])["objective"]
        assert 0 <= e.start_seconds <= e.end_seconds <= makespan + 1e-3
    assert sum(w.pivots for w in workers) == sum(e.result.pivots_phase_one + e.result.pivots_phase_two for e in entries)
    batch.close()


def test_batch_with_an_external_ticket_source_serves_only_its_share():
    """Several processes share one queue through `next_ticket`; here: a counter that hands this batch every second ticket."""
    models = [relp_amd.Model(os.path.join(ROOT, "data", "netlib", n + ".SIF")) for n in NAMES[:4]]
    batch = relp_amd.Batch(models, devices=(0,), workers_per_device=2)
    lock, state = threading.Lock(), {"next": 0}

    def next_ticket():
        with lock:
            t = state["next"]
            state["next"] += 2
            return t
    entries, workers, _ = batch.run([0, 1, 2, 3, 0, 1, 2, 3], next_ticket=next_ticket)
    assert [e.status for e in entries] == [0, -1, 0, -1, 0, -1, 0, -1]
    assert sum(w.tickets for w in workers) == 4
    batch.close()


def test_batch_rejects_a_bad_schedule():
    models = [relp_amd.Model(os.path.join(ROOT, "data", "netlib", "AFIRO.SIF"))]
    batch = relp_amd.Batch(models)
    with pytest.raises(relp_amd.RelpError):
        batch.run([0, 1])
    batch.close()


def test_a_failing_ticket_source_ends_the_run_and_reaches_the_caller():
    """An exception inside the `next_ticket` callback is not swallowed into "ticket 0 forever": the workers stop and it is re-raised."""
    models = [relp_amd.Model(os.path.join(ROOT, "data", "netlib", n + ".SIF")) for n in NAMES[:2]]
    batch = relp_amd.Batch(models, devices=(0,), workers_per_device=2)

    def broken():
        raise KeyError("no tickets today")
    with pytest.raises(KeyError):
        batch.run([0, 1, 0, 1], next_ticket=broken)
    batch.close()


def test_a_ticket_handed_out_twice_is_served_once():
    """A source that repeats a ticket: the second draw ends its worker, the run reports the misuse, no entry is written twice."""
    models = [relp_amd.Model(os.path.join(ROOT, "data", "netlib", n + ".SIF")) for n in NAMES[:2]]
    batch = relp_amd.Batch(models, devices=(0,), workers_per_device=2)
    with pytest.raises(relp_amd.RelpError):
        batch.run([0, 1, 0, 1], next_ticket=lambda: 0)
    batch.close()
