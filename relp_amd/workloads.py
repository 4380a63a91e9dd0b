"""Synthetic inputs of the BASELINE configs (input generators only; nothing here solves anything).

Dense random LP of config 3: splitmix64(seed 0x5EED0001); row-major draws
A[i][j] = 1 + (x mod 100), then b[i] = 100000 + (x mod 100000), then c[j] = -(1 + (x mod 100));
`A x <= b, x >= 0` (all-slack initial basis).
"""
import numpy as np

MASK = (1 << 64) - 1


def splitmix64_stream(seed, count):
    """`count` successive outputs of splitmix64 (vectorised: the state is an arithmetic progression)."""
    gamma = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + gamma * np.arange(1, count + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def dense_lp(m, n, seed=0x5EED0001):
    """Returns (A column-major as an (n, m) int64 array, b (m,), c (n,))."""
    x = splitmix64_stream(seed, m * n + m + n)
    a_row_major = (1 + (x[:m * n] % np.uint64(100))).astype(np.int64).reshape(m, n)
    b = (100000 + (x[m * n:m * n + m] % np.uint64(100000))).astype(np.int64)
    c = -(1 + (x[m * n + m:] % np.uint64(100))).astype(np.int64)
    return np.ascontiguousarray(a_row_major.T), b, c


def max_flow_graph(nr_vertices, nr_arcs, seed=0x5EED0005):
    """Random directed graph of BASELINE config 5: splitmix64(seed); endpoints
    ``u = x mod V``, ``v = x' mod V`` (self arcs and duplicates rejected), capacity ``1 + (x'' mod 100)``;
    source 0, sink V-1.  Returns (tail, head, capacity) sorted by (tail, head)."""
    seen = set()
    tails, heads, capacities = [], [], []
    offset = 0
    while len(tails) < nr_arcs:
        need = nr_arcs - len(tails)
        x = splitmix64_stream(seed, offset + 3 * need)[offset:]
        offset += 3 * need
        for k in range(need):
            u, v = int(x[3 * k] % np.uint64(nr_vertices)), int(x[3 * k + 1] % np.uint64(nr_vertices))
            if u == v or (u, v) in seen:
                continue
            seen.add((u, v))
            tails.append(u)
            heads.append(v)
            capacities.append(1 + int(x[3 * k + 2] % np.uint64(100)))
    order = sorted(range(nr_arcs), key=lambda k: (tails[k], heads[k]))
    return (np.array([tails[k] for k in order], dtype=np.int32), np.array([heads[k] for k in order], dtype=np.int32),
            np.array([capacities[k] for k in order], dtype=np.int64))
