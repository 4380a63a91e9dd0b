"""Multi-GPU host logic: independent LPs shard one per GPU; no data-path collective.

`assign` is the static work distribution of a batch of LPs (longest estimated cost first, deterministic);
`aggregate` is the only exchange of the bench: MAX of the per-rank wall time and SUM of the per-rank pivot counts
(RCCL all-reduce on GPUs, gloo in the CPU tests).  Nothing here computes an LP.
"""
import threading

import torch
import torch.distributed as dist


def assign(costs, world_size):
    """Longest-processing-time-first partition.  `costs`: list of (name, estimated cost).  Returns one list per rank."""
    ranks = [[] for _ in range(world_size)]
    loads = [0.0] * world_size
    for name, cost in sorted(costs, key=lambda item: (-item[1], item[0])):
        target = min(range(world_size), key=lambda r: (loads[r], r))
        ranks[target].append(name)
        loads[target] += cost
    return ranks


def aggregate(elapsed_seconds, pivots, device=None):
    """(max elapsed over ranks, total pivots over ranks).  Works without an initialised process group (N = 1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_seconds), int(pivots)
    t = torch.tensor([float(elapsed_seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    p = torch.tensor([float(pivots)], dtype=torch.float64, device=device)
    dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(p.item()))


def gather_records(record):
    """All ranks' result records on every rank (fixed-size python objects; tiny)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [record]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, record)
    return out


class TicketQueue:
    """Dynamic work queue over the LP ids 0..count-1 ("served by an atomic ticket").

    Every rank holds the same cost-sorted list; `next()` hands out the next unclaimed index, or None when the list is
    exhausted.  The ticket is an atomic fetch-add on the process group's key-value store (owned by rank 0; one host
    round trip of ~0.1 ms per LP against solves of milliseconds to seconds) -- RCCL has no one-sided atomic, so the
    collectives carry only the result gather and the time / pivot reductions.  `tag` separates successive passes over
    the list (one per bench step), so no barrier is needed between them.  Without a process group it is a local counter.
    """

    def __init__(self, count, tag):
        self.count = int(count)
        self.key = "relp_amd/ticket/%s" % tag
        self.local = 0
        self.lock = threading.Lock()  # several host threads of one rank may draw tickets (one LP in flight per thread)
        self.store = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.store = dist.distributed_c10d._get_default_store()

    def next(self):
        with self.lock:
            if self.store is None:
                ticket = self.local
                self.local += 1
            else:
                ticket = self.store.add(self.key, 1) - 1
        return ticket if ticket < self.count else None
