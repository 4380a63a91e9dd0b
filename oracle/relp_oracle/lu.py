"""LU decomposition ``P B Q = L U`` with Forrest-Tomlin updates (oracle; test infrastructure only).

Follows ``tableau/inverse_maintenance/carry/lower_upper/{mod.rs, eta_file.rs,
decomposition/mod.rs, decomposition/pivoting.rs}``.  Sparse vectors are sorted lists of
``(index, Fraction)`` with no explicit zeros, as in ``data/linear_algebra/vector/sparse.rs:90-95``.
"""
import heapq
from bisect import bisect_left
from fractions import Fraction

from .permutation import FullPermutation, RotateToBack, Swap, sorted_get

ZERO = Fraction(0)
ONE = Fraction(1)


def _update_value(difference, vector, index):
    """eta_file.rs:137-157: ``vector[index] -= difference`` on a sorted sparse list."""
    if difference == 0:
        return
    pos = bisect_left(vector, index, key=lambda t: t[0])
    if pos < len(vector) and vector[pos][0] == index:
        new = vector[pos][1] - difference
        if new == 0:
            del vector[pos]
        else:
            vector[pos] = (index, new)
    else:
        vector.insert(pos, (index, -difference))


class EtaFile:
    """eta_file.rs:14-18: row eta ``R = I + e_p r'``; ``values`` are the entries of ``r`` right of ``pivot``."""

    def __init__(self, values, pivot, length):
        self.values = list(values)
        self.pivot = pivot
        self.len = length
        assert all(a[0] < b[0] for a, b in zip(self.values, self.values[1:]))
        assert not self.values or self.values[0][0] > pivot
        assert not self.values or self.values[-1][0] < length

    def apply_left(self, vector):
        """eta_file.rs:49-65 (BTRAN direction, ``x M``): ``v[j] -= r_j * v[pivot]``."""
        pos = sorted_get(vector, self.pivot)
        if pos is not None:
            pivot_value = vector[pos][1]
            for j, value in self.values:
                _update_value(value * pivot_value, vector, j)

    def apply_right(self, vector):
        """eta_file.rs:72-105 (FTRAN direction, ``M x``): ``v[pivot] -= sum_k r_k v_k``."""
        lookup = dict(vector)
        total = ZERO
        for k, value in self.values:
            other = lookup.get(k)
            if other is not None:
                total += value * other
        _update_value(total, vector, self.pivot)

    def update_spike_pivot_value(self, spike):
        """eta_file.rs:112-134: same as ``apply_right`` but only uses spike entries right of the pivot."""
        lookup = {i: v for i, v in spike if i > self.pivot}
        difference = ZERO
        for row, value in self.values:
            other = lookup.get(row)
            if other is not None:
                difference += value * other
        _update_value(difference, spike, self.pivot)

    def __eq__(self, other):
        return (isinstance(other, EtaFile) and self.values == other.values
                and self.pivot == other.pivot and self.len == other.len)

    def __repr__(self):
        return "EtaFile(%r, pivot=%d, len=%d)" % (self.values, self.pivot, self.len)


class ColumnAndSpike:
    """lower_upper/mod.rs:417-432: FTRAN result plus the spike (the vector before the U solve)."""

    def __init__(self, column, spike):
        self.column = column  # sorted sparse list
        self.spike = spike    # sorted sparse list

    def into_column(self):
        return self.column


class _Worklist:
    """Ordered-map work list standing in for ``BTreeMap<usize, F>`` (mod.rs:286-415)."""

    def __init__(self, items, descending):
        self.values = {}
        self.sign = -1 if descending else 1
        self.heap = []
        for i, v in items:
            assert i not in self.values
            self.values[i] = v
            self.heap.append(self.sign * i)
        heapq.heapify(self.heap)

    def pop(self):
        while self.heap:
            i = self.sign * heapq.heappop(self.heap)
            v = self.values.pop(i, None)
            if v is not None:
                return i, v
        return None

    def insert_or_shift_maybe_remove(self, index, change):
        """lower_upper/mod.rs:400-415."""
        existing = self.values.get(index)
        if existing is None:
            self.values[index] = -change
            heapq.heappush(self.heap, self.sign * index)
        else:
            new = existing - change
            if new == 0:
                del self.values[index]
            else:
                self.values[index] = new


class LUDecomposition:
    """lower_upper/mod.rs:36-58."""

    REFACTOR_AFTER = 30  # lower_upper/mod.rs:249-252: ``updates.len() > 30``

    def __init__(self, row_permutation, column_permutation, lower_triangular, upper_triangular,
                 upper_diagonal, updates=None):
        self.row_permutation = row_permutation
        self.column_permutation = column_permutation
        self.lower_triangular = [list(c) for c in lower_triangular]
        self.upper_triangular = [list(c) for c in upper_triangular]
        self.upper_diagonal = list(upper_diagonal)
        self.updates = list(updates or [])
        self._row_index_cache = None

    # ---- constructors -------------------------------------------------------------------------
    @classmethod
    def identity(cls, m):
        """lower_upper/mod.rs:67-76."""
        return cls(FullPermutation.identity(m), FullPermutation.identity(m),
                   [[] for _ in range(m - 1)], [[] for _ in range(m - 1)], [ONE] * m)

    @classmethod
    def invert(cls, columns):
        """lower_upper/mod.rs:78-92: gather the basis columns row-major, then factorise."""
        columns = list(columns)
        m = len(columns)
        rows = [[] for _ in range(m)]
        for j, column in enumerate(columns):
            for i, value in column:
                rows[i].append((j, Fraction(value)))
        return cls.rows(rows)

    @classmethod
    def rows(cls, rows):
        """decomposition/mod.rs:27-143: right-looking sparse Gaussian elimination with Markowitz pivoting."""
        rows = [list(r) for r in rows]
        m = len(rows)
        assert m > 1  # decomposition/mod.rs:32
        row_permutation = list(range(m))
        column_permutation = list(range(m))
        lower_row_major = [[] for _ in range(m - 1)]

        nnz_row = [len(r) for r in rows]  # decomposition/mod.rs:278-304
        nnz_column = [0] * m
        for r in rows:
            for j, _ in r:
                nnz_column[j] += 1

        for k in range(m):
            pivot_row, pivot_column = _markowitz(nnz_row, nnz_column, rows, k)
            _swap(pivot_row, pivot_column, k, row_permutation, column_permutation,
                  nnz_row, nnz_column, rows, lower_row_major)

            for j, _ in rows[k]:  # decomposition/mod.rs:57-60
                nnz_row[k] -= 1
                nnz_column[j] -= 1

            current_row = rows[k]
            pivot_value = current_row[0][1]
            assert current_row[0][0] == k

            ratios = []  # decomposition/mod.rs:71-79
            for i in range(k + 1, m):
                row = rows[i]
                assert row, "The first item exists (invertibility)."
                if row[0][0] == k:
                    ratios.append((i, row.pop(0)[1] / pivot_value))
                    nnz_row[i] -= 1
                    nnz_column[k] -= 1

            for i, ratio in ratios:  # decomposition/mod.rs:82-100
                old_len = len(rows[i])
                new_row, removed, added = subtract_multiple_of_row_from_other_row(
                    rows[i], ratio, current_row[1:])
                rows[i] = new_row
                nnz_row[i] += len(new_row) - old_len
                for c in removed:
                    nnz_column[c] -= 1
                for c in added:
                    nnz_column[c] += 1
                lower_row_major[i - 1].append((k, ratio))

        # decomposition/mod.rs:108-127: transpose into column major
        upper_triangular = [[] for _ in range(m - 1)]
        upper_diagonal = []
        for i, row in enumerate(rows):
            assert row[0][0] == i
            upper_diagonal.append(row[0][1])
            for j, value in row[1:]:
                upper_triangular[j - 1].append((i, value))
        lower_triangular = [[] for _ in range(m - 1)]
        for idx, row in enumerate(lower_row_major):
            i = idx + 1
            for j, v in row:
                lower_triangular[j].append((i, v))

        # decomposition/mod.rs:129-133
        rp = FullPermutation(row_permutation)
        rp.invert()
        cp = FullPermutation(column_permutation)
        cp.invert()
        return cls(rp, cp, lower_triangular, upper_triangular, upper_diagonal)

    # ---- queries --------------------------------------------------------------------------------
    def m(self):
        return len(self.row_permutation)

    def should_refactor(self):
        """lower_upper/mod.rs:249-252."""
        return len(self.updates) > self.REFACTOR_AFTER

    # ---- FTRAN ----------------------------------------------------------------------------------
    def left_multiply_by_basis_inverse(self, column):
        """lower_upper/mod.rs:180-210: ``B^-1 c``; also returns the spike."""
        rhs = [(self.row_permutation[i], Fraction(v)) for i, v in column]
        w = self._left_multiply_by_lower_inverse(rhs)
        for eta, q in self.updates:
            eta.apply_right(w)
            w = q.forward_sorted(w)
        spike = list(w)
        column = self._left_multiply_by_upper_inverse(w)
        for _, q in reversed(self.updates):
            column = q.backward_unsorted(column)
        column = self.column_permutation.backward_unsorted(column)
        column.sort(key=lambda t: t[0])
        return ColumnAndSpike(column, spike)

    def generate_element(self, i, column):
        """lower_upper/mod.rs:239-247: a full FTRAN followed by a lookup."""
        result = self.left_multiply_by_basis_inverse(column).column
        pos = sorted_get(result, i)
        return None if pos is None else result[pos][1]

    def _left_multiply_by_lower_inverse(self, rhs):
        """lower_upper/mod.rs:286-305."""
        work = _Worklist(rhs, descending=False)
        result = []
        m = self.m()
        while True:
            item = work.pop()
            if item is None:
                break
            row, value = item
            if row != m - 1:
                for i, l in self.lower_triangular[row]:
                    work.insert_or_shift_maybe_remove(i, value * l)
            result.append((row, value))
        return result

    def _left_multiply_by_upper_inverse(self, rhs):
        """lower_upper/mod.rs:307-321 and update_rhs :339-345."""
        work = _Worklist(rhs, descending=True)
        result = []
        while True:
            item = work.pop()
            if item is None:
                break
            row, value = item
            x = value / self.upper_diagonal[row]
            if row > 0:
                for i, u in self.upper_triangular[row - 1]:
                    work.insert_or_shift_maybe_remove(i, x * u)
            result.append((row, x))
        result.reverse()
        return result

    # ---- BTRAN ----------------------------------------------------------------------------------
    def right_multiply_by_basis_inverse(self, row):
        """lower_upper/mod.rs:212-237: ``r B^-1``."""
        lhs = [(self.column_permutation[i], Fraction(v)) for i, v in row]
        for _, q in self.updates:
            lhs = q.forward_unsorted(lhs)
        lhs = self._right_multiply_by_upper_inverse(lhs)
        for eta, q in reversed(self.updates):
            lhs = q.backward_sorted(lhs)
            eta.apply_left(lhs)
        lhs = self._right_multiply_by_lower_inverse(lhs)
        return self.row_permutation.backward_sorted(lhs)

    def basis_inverse_row(self, row):
        """lower_upper/mod.rs:254-272: ``e_row' B^-1``."""
        row = self.column_permutation.forward(row)
        for _, q in self.updates:
            row = q.forward(row)
        w = self._right_multiply_by_upper_inverse([(row, ONE)])
        for eta, q in reversed(self.updates):
            w = q.backward_sorted(w)
            eta.apply_left(w)
        tuples = self._right_multiply_by_lower_inverse(w)
        return self.row_permutation.backward_sorted(tuples)

    def _row_index(self):
        """Row-wise view of L and U.

        The reference scans every column with a binary search per popped entry
        (lower_upper/mod.rs:355-362, 381-389); the set of entries found is the same.
        """
        if self._row_index_cache is None:
            m = self.m()
            upper_rows = [[] for _ in range(m)]
            for jm1, column in enumerate(self.upper_triangular):
                for i, v in column:
                    upper_rows[i].append((jm1 + 1, v))
            lower_rows = [[] for _ in range(m)]
            for j, column in enumerate(self.lower_triangular):
                for i, v in column:
                    lower_rows[i].append((j, v))
            self._row_index_cache = (upper_rows, lower_rows)
        return self._row_index_cache

    def _right_multiply_by_upper_inverse(self, rhs):
        """lower_upper/mod.rs:373-397."""
        upper_rows, _ = self._row_index()
        work = _Worklist(rhs, descending=False)
        result = []
        while True:
            item = work.pop()
            if item is None:
                break
            column, value = item
            x = value / self.upper_diagonal[column]
            for j, u in upper_rows[column]:
                work.insert_or_shift_maybe_remove(j, x * u)
            result.append((column, x))
        return result

    def _right_multiply_by_lower_inverse(self, rhs):
        """lower_upper/mod.rs:347-371."""
        _, lower_rows = self._row_index()
        work = _Worklist(rhs, descending=True)
        result = []
        while True:
            item = work.pop()
            if item is None:
                break
            column, value = item
            for j, l in lower_rows[column]:
                work.insert_or_shift_maybe_remove(j, value * l)
            result.append((column, value))
        result.reverse()
        return result

    # ---- Forrest-Tomlin update ------------------------------------------------------------------
    def change_basis(self, pivot_row_index, info):
        """lower_upper/mod.rs:94-178.  Returns the FTRAN column unmodified."""
        m = self.m()
        t = self.column_permutation.forward(pivot_row_index)
        for _, q in self.updates:
            t = q.forward(t)

        # mod.rs:112-125: row t of U right of the diagonal, then r = u_bar U^-1
        u_bar = []
        to_zero = []
        for j in range(t + 1, m):
            column = self.upper_triangular[j - 1]
            pos = sorted_get(column, t)
            if pos is not None:
                u_bar.append((j, column[pos][1]))
                to_zero.append((j, pos))
        r = self._right_multiply_by_upper_inverse(u_bar)
        eta = EtaFile(r, t, m)

        for j, pos in to_zero:  # mod.rs:129-131
            del self.upper_triangular[j - 1][pos]
        spike = list(info.spike)
        eta.update_spike_pivot_value(spike)  # mod.rs:135
        assert sorted_get(spike, t) is not None, "singular basis after update"

        disappearing = 0 if t == 0 else t - 1  # mod.rs:141-149
        self.upper_triangular[disappearing] = spike
        ut = self.upper_triangular
        ut[disappearing:] = ut[disappearing + 1:] + ut[disappearing:disappearing + 1]  # rotate_left(1)
        ud = self.upper_diagonal
        ud[t:] = ud[t + 1:] + ud[t:t + 1]

        q = RotateToBack(t, m)  # mod.rs:158-161
        for j in range(max(t, 1), m):
            ut[j - 1] = q.forward_sorted(ut[j - 1])
        corner_index, corner_value = ut[-1].pop()  # mod.rs:162-164
        assert corner_index == m - 1
        ud[-1] = corner_value

        self.updates.append((eta, q))
        self._row_index_cache = None
        return info.column

    def __eq__(self, other):
        return (isinstance(other, LUDecomposition)
                and self.row_permutation == other.row_permutation
                and self.column_permutation == other.column_permutation
                and self.lower_triangular == other.lower_triangular
                and self.upper_triangular == other.upper_triangular
                and self.upper_diagonal == other.upper_diagonal
                and self.updates == other.updates)

    def __repr__(self):
        return ("LU(P=%r, Q=%r, L=%r, U=%r, D=%r, updates=%r)" % (
            self.row_permutation, self.column_permutation, self.lower_triangular,
            self.upper_triangular, self.upper_diagonal, self.updates))


def _markowitz(nnz_row, nnz_column, rows, k):
    """decomposition/pivoting.rs:45-81.

    All remaining ``(i, j)`` pairs in (row, column) order, stably sorted by ``j``; the *first*
    minimum of ``(nnz_row[i]-1)*(nnz_column[j]-1)`` wins (Rust ``Iterator::min_by_key``).
    """
    best = None
    best_key = None
    for i in range(k, len(rows)):
        row = rows[i]
        start = bisect_left(row, k, key=lambda t: t[0])
        for j, _ in row[start:]:
            score = (nnz_row[i] - 1) * (nnz_column[j] - 1)
            key = (score, j, i)
            if best_key is None or key < best_key:
                best_key = key
                best = (i, j)
    return best


def _swap(pivot_row, pivot_column, k, row_permutation, column_permutation, nnz_row, nnz_column,
          rows, lower_row_major):
    """decomposition/mod.rs:224-273."""
    if pivot_row != k:
        row_permutation[pivot_row], row_permutation[k] = row_permutation[k], row_permutation[pivot_row]
        nnz_row[pivot_row], nnz_row[k] = nnz_row[k], nnz_row[pivot_row]
        rows[pivot_row], rows[k] = rows[k], rows[pivot_row]
        if pivot_row > 0 and k > 0:
            a, b = pivot_row - 1, k - 1
            lower_row_major[a], lower_row_major[b] = lower_row_major[b], lower_row_major[a]
    if pivot_column != k:
        column_permutation[pivot_column], column_permutation[k] = \
            column_permutation[k], column_permutation[pivot_column]
        nnz_column[pivot_column], nnz_column[k] = nnz_column[k], nnz_column[pivot_column]
        swap = Swap((pivot_column, k), len(rows))
        for idx, row in enumerate(rows):
            if row:
                rows[idx] = swap.forward_sorted(row)


def subtract_multiple_of_row_from_other_row(to_edit, ratio, being_removed):
    """decomposition/mod.rs:146-210: ``to_edit -= ratio * being_removed`` (sorted merge, zeros dropped).

    Returns ``(new_row, columns_removed, columns_added)``.
    """
    new = []
    removed = []
    added = []
    index = 0
    n = len(being_removed)
    for j, old_value in to_edit:
        while index < n and being_removed[index][0] < j:
            new.append((being_removed[index][0], -ratio * being_removed[index][1]))
            added.append(being_removed[index][0])
            index += 1
        if index < n and being_removed[index][0] == j:
            product = ratio * being_removed[index][1]
            if product != old_value:
                new.append((j, old_value - product))
            else:
                removed.append(j)
            index += 1
        else:
            new.append((j, old_value))
    while index < n:
        new.append((being_removed[index][0], -ratio * being_removed[index][1]))
        added.append(being_removed[index][0])
        index += 1
    return new, removed, added
