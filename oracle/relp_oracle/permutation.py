"""Permutations used by the LU decomposition (oracle; test infrastructure only).

Follows ``tableau/inverse_maintenance/carry/lower_upper/permutation/{mod,full,rotate_to_back,swap}.rs``.
Items are Python lists of ``[index, value]`` pairs (or tuples); *sorted* variants keep them ordered.
"""
from bisect import bisect_left


class Permutation:
    """permutation/mod.rs:17-124 -- default method bodies."""

    def forward(self, i):
        raise NotImplementedError

    def backward(self, i):
        raise NotImplementedError

    def forward_unsorted(self, items):  # mod.rs:96-102
        return [(self.forward(i), v) for i, v in items]

    def backward_unsorted(self, items):  # mod.rs:113-119
        return [(self.backward(i), v) for i, v in items]

    def forward_sorted(self, items):  # mod.rs:58-66
        return sorted(self.forward_unsorted(items), key=lambda t: t[0])

    def backward_sorted(self, items):  # mod.rs:73-81
        return sorted(self.backward_unsorted(items), key=lambda t: t[0])


class FullPermutation(Permutation):
    """permutation/full.rs:15-110: explicit forward and backward arrays."""

    def __init__(self, forward):
        self.fwd = list(forward)
        self.bwd = [0] * len(self.fwd)
        for i, j in enumerate(self.fwd):  # full.rs:26-41 (inverse by sorting)
            self.bwd[j] = i

    @classmethod
    def identity(cls, n):  # full.rs:50-55
        return cls(range(n))

    def invert(self):  # full.rs:60-62
        self.fwd, self.bwd = self.bwd, self.fwd

    def swap(self, i, j):  # full.rs:64-73
        it, jt = self.fwd[i], self.fwd[j]
        self.fwd[i], self.fwd[j] = self.fwd[j], self.fwd[i]
        self.bwd[it], self.bwd[jt] = self.bwd[jt], self.bwd[it]

    def swap_inverse(self, i, j):  # full.rs:75-84
        it, jt = self.bwd[i], self.bwd[j]
        self.fwd[it], self.fwd[jt] = self.fwd[jt], self.fwd[it]
        self.bwd[i], self.bwd[j] = self.bwd[j], self.bwd[i]

    def rotate_right_from(self, i):  # full.rs:86-91
        self.fwd[i:] = self.fwd[-1:] + self.fwd[i:-1]
        self.bwd[i:] = self.bwd[i + 1:] + self.bwd[i:i + 1]

    def forward(self, i):
        return self.fwd[i]

    def backward(self, i):
        return self.bwd[i]

    def __getitem__(self, i):  # full.rs:111-119 (Index == forward)
        return self.fwd[i]

    def __len__(self):
        return len(self.fwd)

    def __eq__(self, other):
        return isinstance(other, FullPermutation) and self.fwd == other.fwd and self.bwd == other.bwd

    def __repr__(self):
        return "Full(%r)" % (self.fwd,)


class RotateToBack(Permutation):
    """permutation/rotate_to_back.rs:15-122: ``index`` goes to ``len-1``; larger ones shift down."""

    def __init__(self, index, length):
        assert index < length
        self.index = index
        self.len = length

    def forward(self, i):  # rotate_to_back.rs:43-51
        if i < self.index:
            return i
        if i == self.index:
            return self.len - 1
        return i - 1

    def backward(self, i):  # rotate_to_back.rs:53-63
        if i < self.index:
            return i
        if i < self.len - 1:
            return i + 1
        return self.index

    # The specialised *_sorted bodies (rotate_to_back.rs:65-117) produce exactly the generic result
    # (a relabel followed by a sort); the generic versions above are used.

    def __eq__(self, other):
        return isinstance(other, RotateToBack) and (self.index, self.len) == (other.index, other.len)

    def __repr__(self):
        return "RotateToBack(%d, %d)" % (self.index, self.len)


class Swap(Permutation):
    """permutation/swap.rs:9-84."""

    def __init__(self, indices, length):
        self.indices = tuple(indices)
        self.len = length

    def forward(self, i):  # swap.rs:24-34
        a, b = self.indices
        if i == a:
            return b
        if i == b:
            return a
        return i

    backward = forward  # swap.rs:36-40


def sorted_get(items, index):
    """Binary search of a sorted ``(index, value)`` list; returns position or ``None``."""
    pos = bisect_left(items, index, key=lambda t: t[0])
    if pos < len(items) and items[pos][0] == index:
        return pos
    return None
