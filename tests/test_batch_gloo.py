"""N > 1 host logic on CPU: world_size-2 gloo process group (no GPU)."""
import os
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, queue):
    sys.path.insert(0, ROOT)
    from relp_amd import batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        costs = [("A", 9.0), ("B", 7.0), ("C", 4.0), ("D", 3.0), ("E", 1.0)]
        mine = batch.assign(costs, world)[rank]
        # fake per-LP results: elapsed proportional to cost, 10 pivots per unit of cost
        elapsed = sum(dict(costs)[name] for name in mine)
        pivots = int(10 * elapsed)
        total_time, total_pivots = batch.aggregate(elapsed, pivots)
        records = batch.gather_records({"rank": rank, "problems": mine})
        queue.put((rank, mine, total_time, total_pivots, records))
    finally:
        dist.destroy_process_group()


def test_assign_is_a_balanced_partition():
    sys.path.insert(0, ROOT)
    from relp_amd import batch
    costs = [("p%d" % i, float((i * 37) % 11 + 1)) for i in range(23)]
    parts = batch.assign(costs, 4)
    assert sorted(name for part in parts for name in part) == sorted(name for name, _ in costs)
    loads = [sum(dict(costs)[n] for n in part) for part in parts]
    assert max(loads) - min(loads) <= max(c for _, c in costs)
    assert batch.assign(costs, 4) == parts  # deterministic
    assert batch.aggregate(1.5, 7) == (1.5, 7)  # no process group: N = 1


def test_two_ranks_aggregate_with_gloo():
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(rank, 2, port, queue)) for rank in range(2)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    (r0, mine0, t0, p0, rec0), (r1, mine1, t1, p1, rec1) = results
    assert sorted(mine0 + mine1) == ["A", "B", "C", "D", "E"]
    assert mine0 == ["A", "D"] and mine1 == ["B", "C", "E"]      # LPT: 9+3 | 7+4+1
    assert t0 == t1 == 12.0                                      # max over ranks
    assert p0 == p1 == 240                                       # sum over ranks
    assert rec0 == rec1 and [r["rank"] for r in rec0] == [0, 1]


def _queue_worker(rank, world, port, queue):
    sys.path.insert(0, ROOT)
    import time
    from relp_amd import batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        durations = [0.20, 0.02, 0.02, 0.02, 0.02, 0.02, 0.02]   # the first (largest) LP occupies one rank for a while
        taken = []
        for epoch in range(2):
            tickets = batch.TicketQueue(len(durations), tag="epoch%d" % epoch)
            mine = []
            while True:
                index = tickets.next()
                if index is None:
                    break
                mine.append(index)
                time.sleep(durations[index])
            taken.append(mine)
            dist.barrier()
        queue.put((rank, taken, batch.gather_records(taken)))
    finally:
        dist.destroy_process_group()


def test_ticket_queue_hands_out_every_lp_exactly_once():
    sys.path.insert(0, ROOT)
    from relp_amd import batch
    local = batch.TicketQueue(3, tag="local")   # no process group: plain counter
    assert [local.next() for _ in range(5)] == [0, 1, 2, None, None]

    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = 30500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_queue_worker, args=(rank, 2, port, queue)) for rank in range(2)]
    for p in procs:
        p.start()
    results = sorted(queue.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, taken0, gathered0), (_, taken1, gathered1) = results
    assert gathered0 == gathered1 == [taken0, taken1]
    for epoch in range(2):
        assert sorted(taken0[epoch] + taken1[epoch]) == list(range(7))      # a partition, every pass
        # dynamic balancing: whoever drew the long LP got few others
        long_owner = taken0[epoch] if 0 in taken0[epoch] else taken1[epoch]
        other = taken1[epoch] if long_owner is taken0[epoch] else taken0[epoch]
        assert len(long_owner) < len(other)
