"""ctypes binding of the stand-alone ``BasisInverse`` object (``relp_bi_*`` in ``include/relp_amd.h``).

Mirrors the reference's trait ``BasisInverse`` (tableau/inverse_maintenance/carry/mod.rs:69-169) as implemented by
``LUDecomposition`` (carry/lower_upper/mod.rs): every method goes through the C ABI into the HIP kernels of
``relp_amd/csrc/lu.hip``.  ``lu_factor_host`` is the host-only factorisation step (no device needed).
"""
import ctypes as C

import numpy as np

from .api import OK, RelpError, _ptr, lib


class BiOptions(C.Structure):
    _fields_ = [("device", C.c_int32), ("refactor_period", C.c_int32), ("pivot_threshold", C.c_double),
                ("reference_ties", C.c_int32), ("switches", C.c_int32)]


def default_bi_options(**overrides):
    options = BiOptions()
    lib().relp_bi_options_default(C.byref(options))
    for key, value in overrides.items():
        if not hasattr(options, key):
            raise AttributeError(key)
        setattr(options, key, value)
    if "switches" not in overrides:  # (the binding maps the old environment hooks; the library reads none)
        from .api import default_options
        options.switches = int(default_options().switches)
    return options


def _sparse(pairs):
    pairs = list(pairs)
    index = np.array([i for i, _ in pairs] or [0], dtype=np.int32)
    value = np.array([float(v) for _, v in pairs] or [0.0], dtype=np.float64)
    return len(pairs), index, value


def _csc(columns):
    start = np.zeros(len(columns) + 1, dtype=np.int64)
    rows, vals = [], []
    for j, column in enumerate(columns):
        for i, v in column:
            rows.append(int(i))
            vals.append(float(v))
        start[j + 1] = len(rows)
    return start, np.array(rows or [0], dtype=np.int32), np.array(vals or [0.0], dtype=np.float64)


class BasisInverse:
    """``LUDecomposition`` on the device.  Construct with :meth:`identity` or :meth:`invert`."""

    def __init__(self, handle):
        self._h = handle
        lib().relp_bi_last_error.restype = C.c_char_p
        lib().relp_bi_last_error.argtypes = [C.c_void_p]

    # ---- constructors (carry/mod.rs:83-92) ----------------------------------------------------------------------
    @classmethod
    def identity(cls, m, **options):
        handle = C.c_void_p()
        opts = default_bi_options(**options)
        status = lib().relp_bi_identity(C.byref(opts), int(m), C.byref(handle))
        cls._check_static(status)
        return cls(handle)

    @classmethod
    def invert(cls, columns, **options):
        """``columns``: m sparse columns ``[(row, value), ...]`` in basis order."""
        start, rows, vals = _csc(columns)
        handle = C.c_void_p()
        opts = default_bi_options(**options)
        status = lib().relp_bi_invert(C.byref(opts), len(columns), _ptr(start, C.c_int64), _ptr(rows, C.c_int32),
                                      _ptr(vals, C.c_double), C.byref(handle))
        cls._check_static(status)
        return cls(handle)

    @staticmethod
    def _check_static(status):
        if status != OK:
            lib().relp_bi_last_error.restype = C.c_char_p
            lib().relp_bi_last_error.argtypes = [C.c_void_p]
            raise RelpError(status, (lib().relp_bi_last_error(None) or b"").decode())

    def _check(self, status):
        if status != OK:
            raise RelpError(status, (lib().relp_bi_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            lib().relp_bi_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the trait's operations -----------------------------------------------------------------------------------
    def m(self):
        m = C.c_int32()
        self._check(lib().relp_bi_m(self._h, C.byref(m)))
        return m.value

    def left_multiply_by_basis_inverse(self, column):
        nnz, index, value = _sparse(column)
        out = np.zeros(self.m())
        self._check(lib().relp_bi_left_multiply(self._h, nnz, _ptr(index, C.c_int32), _ptr(value, C.c_double), _ptr(out, C.c_double)))
        return out

    def right_multiply_by_basis_inverse(self, row):
        nnz, index, value = _sparse(row)
        out = np.zeros(self.m())
        self._check(lib().relp_bi_right_multiply(self._h, nnz, _ptr(index, C.c_int32), _ptr(value, C.c_double), _ptr(out, C.c_double)))
        return out

    def basis_inverse_row(self, row):
        out = np.zeros(self.m())
        self._check(lib().relp_bi_basis_inverse_row(self._h, int(row), _ptr(out, C.c_double)))
        return out

    def generate_element(self, i, column):
        nnz, index, value = _sparse(column)
        element, some = C.c_double(), C.c_int32()
        self._check(lib().relp_bi_generate_element(self._h, int(i), nnz, _ptr(index, C.c_int32), _ptr(value, C.c_double),
                                                   C.byref(element), C.byref(some)))
        return element.value if some.value else None

    def change_basis(self, pivot_row_index):
        self._check(lib().relp_bi_change_basis(self._h, int(pivot_row_index)))

    def should_refactor(self):
        flag = C.c_int32()
        self._check(lib().relp_bi_should_refactor(self._h, C.byref(flag)))
        return bool(flag.value)

    def remove_basis_part(self, indices):
        idx_list = list(indices)  # (a generator must be walked once only)
        idx = np.array(idx_list or [0], dtype=np.int32)
        self._check(lib().relp_bi_remove_basis_part(self._h, len(idx_list), _ptr(idx, C.c_int32)))

    def statistics(self):
        nl, nu = C.c_int64(), C.c_int64()
        dl, du, upd = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(lib().relp_bi_statistics(self._h, C.byref(nl), C.byref(nu), C.byref(dl), C.byref(du), C.byref(upd)))
        return {"nnz_lower": nl.value, "nnz_upper": nu.value, "depth_lower": dl.value, "depth_upper": du.value, "updates": upd.value}

    def factors(self):
        """The factors in the reference's layout (lower_upper/mod.rs:36-58) as plain Python lists."""
        m = self.m()
        stats = self.statistics()
        cap = 4 * (stats["nnz_lower"] + stats["nnz_upper"] + (stats["updates"] + 2) * (m + 2)) + 64
        rp, cp = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        ls, us, es = np.zeros(cap, np.int64), np.zeros(cap, np.int64), np.zeros(cap, np.int64)
        lr, ur, ep, ei = (np.zeros(cap, np.int32) for _ in range(4))
        lv, uv, ud, ev = (np.zeros(cap, np.float64) for _ in range(4))
        k = C.c_int32()
        self._check(lib().relp_bi_get_factors(
            self._h, cap, _ptr(rp, C.c_int32), _ptr(cp, C.c_int32), _ptr(ls, C.c_int64), _ptr(lr, C.c_int32), _ptr(lv, C.c_double),
            _ptr(us, C.c_int64), _ptr(ur, C.c_int32), _ptr(uv, C.c_double), _ptr(ud, C.c_double), C.byref(k),
            _ptr(es, C.c_int64), _ptr(ep, C.c_int32), _ptr(ei, C.c_int32), _ptr(ev, C.c_double)))
        lower = [[(int(lr[e]), float(lv[e])) for e in range(ls[j], ls[j + 1])] for j in range(m)]
        upper = [[(int(ur[e]), float(uv[e])) for e in range(us[j], us[j + 1])] for j in range(m)]
        etas = [(int(ep[q]), [(int(ei[e]), float(ev[e])) for e in range(es[q], es[q + 1])]) for q in range(k.value)]
        return {"row_permutation": [int(v) for v in rp[:m]], "column_permutation": [int(v) for v in cp[:m]],
                "lower_triangular": lower[:m - 1], "upper_triangular": upper[1:], "upper_diagonal": [float(v) for v in ud[:m]],
                "updates": etas}


def lu_factor_device(columns, pivot_threshold=0.1, reference_ties=False, inverted=False, dense_tail=32, device=0):
    """The same factorisation as the DEVICE runs it (``relp_lu_factor_device``: the kernels of lu_factor.hip that the LU carries'
    refactorisation launches).  Same dict as ``lu_factor_host`` plus ``info`` (status, nnz(L), nnz(U), rounds, dense-tail rows, ...)."""
    return lu_factor_host(columns, pivot_threshold, reference_ties, inverted, _device=(device, dense_tail))


def lu_factor_host(columns, pivot_threshold=0.1, reference_ties=False, inverted=False, _device=None):
    """Host-only ``LUDecomposition::rows`` (decomposition/mod.rs:27-143).  Returns a dict: ``rowpos``, ``colpos``, L and U as
    lists of rows ``[(column, value)]`` of the position space, ``diag`` and the dependency depths.  ``inverted``: the two triangles
    inverted as sparse matrices instead (``relp_lu_invert_host``: what the inverse-factor carry uploads) -- ``lower_rows`` the strict
    part of L^-1, ``upper_rows`` U^-1 with its diagonal, ``diag`` ones."""
    m = len(columns)
    start, rows, vals = _csc(columns)
    nnz = int(start[-1])
    cap = 64 + 4 * m + 16 * nnz
    while True:
        rp, cp = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        ls, us = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
        lc, uc = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        lv, uv, ud = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        dl, du = C.c_int32(), C.c_int32()
        info = np.zeros(32, np.int32)
        if _device is not None:
            from .api import default_options
            lib().relp_debug_set_tuning(C.byref(default_options()))  # (the RELP_LUF_* test hooks reach the kernels as options)
            status = lib().relp_lu_factor_device(
                int(_device[0]), m, _ptr(start, C.c_int64), _ptr(rows, C.c_int32), _ptr(vals, C.c_double), C.c_double(pivot_threshold),
                int(bool(reference_ties)), int(_device[1]), int(bool(inverted)), C.c_int64(cap), _ptr(rp, C.c_int32), _ptr(cp, C.c_int32),
                _ptr(ls, C.c_int64), _ptr(lc, C.c_int32), _ptr(lv, C.c_double), _ptr(us, C.c_int64), _ptr(uc, C.c_int32), _ptr(uv, C.c_double),
                _ptr(ud, C.c_double), _ptr(info, C.c_int32))
        elif inverted:
            status = lib().relp_lu_invert_host(
                m, _ptr(start, C.c_int64), _ptr(rows, C.c_int32), _ptr(vals, C.c_double), C.c_double(pivot_threshold), cap,
                _ptr(rp, C.c_int32), _ptr(cp, C.c_int32), _ptr(ls, C.c_int64), _ptr(lc, C.c_int32), _ptr(lv, C.c_double),
                _ptr(us, C.c_int64), _ptr(uc, C.c_int32), _ptr(uv, C.c_double), _ptr(ud, C.c_double))
        else:
            status = lib().relp_lu_factor_host(
                m, _ptr(start, C.c_int64), _ptr(rows, C.c_int32), _ptr(vals, C.c_double), C.c_double(pivot_threshold),
                int(bool(reference_ties)), cap, _ptr(rp, C.c_int32), _ptr(cp, C.c_int32), _ptr(ls, C.c_int64), _ptr(lc, C.c_int32),
                _ptr(lv, C.c_double), _ptr(us, C.c_int64), _ptr(uc, C.c_int32), _ptr(uv, C.c_double), _ptr(ud, C.c_double),
                C.byref(dl), C.byref(du))
        if status == OK:
            break
        lib().relp_bi_last_error.restype = C.c_char_p
        lib().relp_bi_last_error.argtypes = [C.c_void_p]
        message = (lib().relp_bi_last_error(None) or b"").decode()
        if "capacity" in message and cap < (1 << 28):
            cap *= 4
            continue
        raise RelpError(status, message)
    return {
        "rowpos": [int(v) for v in rp[:m]], "colpos": [int(v) for v in cp[:m]],
        "lower_rows": [[(int(lc[e]), float(lv[e])) for e in range(ls[i], ls[i + 1])] for i in range(m)],
        "upper_rows": [[(int(uc[e]), float(uv[e])) for e in range(us[i], us[i + 1])] for i in range(m)],
        "diag": [float(v) for v in ud[:m]], "depth_lower": dl.value, "depth_upper": du.value,
        "nnz_lower": int(ls[m]), "nnz_upper": int(us[m]), "info": [int(v) for v in info],
    }


# ---- `BasisInverse` over exact rationals (relp_bix_*; relp_amd/csrc/exact_bi.hip) -----------------------------------------------
def _exact_sparse(pairs):
    from fractions import Fraction
    pairs = [(int(i), Fraction(v)) for i, v in pairs]
    index = np.array([i for i, _ in pairs] or [0], dtype=np.int32)
    num = np.array([v.numerator for _, v in pairs] or [0], dtype=np.int64)
    den = np.array([v.denominator for _, v in pairs] or [1], dtype=np.int64)
    return len(pairs), index, num, den


def _integer(words):
    """Little-endian two's complement 64-bit words -> int."""
    return int.from_bytes(np.ascontiguousarray(words, dtype=np.uint64).tobytes(), "little", signed=True)


class ExactBasisInverse:
    """The reference's ``BasisInverse`` (carry/mod.rs:69-169) for ``F = RationalBig`` on the device: N = D B^-1 in fixed-width integers,
    ``change_basis`` by Edmonds' integer-preserving pivot.  Vectors in: sparse ``(index, Fraction-like)`` pairs; vectors out: lists of
    ``Fraction`` (numerators and the one denominator cross the C ABI as 64-bit words; the reduction is Python's)."""

    def __init__(self, handle):
        self._h = handle
        lib().relp_bix_last_error.restype = C.c_char_p
        lib().relp_bix_last_error.argtypes = [C.c_void_p]

    @classmethod
    def identity(cls, m, device=0):
        handle = C.c_void_p()
        status = lib().relp_bix_identity(int(device), int(m), C.byref(handle))
        if status != OK:
            lib().relp_bix_last_error.restype = C.c_char_p
            raise RelpError(status, (lib().relp_bix_last_error(None) or b"").decode())
        return cls(handle)

    @classmethod
    def invert(cls, columns, device=0):
        from fractions import Fraction
        columns = [[(int(i), Fraction(v)) for i, v in column] for column in columns]
        start = np.zeros(len(columns) + 1, dtype=np.int64)
        rows, nums, dens = [], [], []
        for j, column in enumerate(columns):
            for i, v in column:
                rows.append(i)
                nums.append(v.numerator)
                dens.append(v.denominator)
            start[j + 1] = len(rows)
        handle = C.c_void_p()
        status = lib().relp_bix_invert(int(device), len(columns), _ptr(start, C.c_int64), _ptr(np.array(rows or [0], dtype=np.int32), C.c_int32),
                                       _ptr(np.array(nums or [0], dtype=np.int64), C.c_int64), _ptr(np.array(dens or [1], dtype=np.int64), C.c_int64),
                                       C.byref(handle))
        if status != OK:
            lib().relp_bix_last_error.restype = C.c_char_p
            raise RelpError(status, (lib().relp_bix_last_error(None) or b"").decode())
        return cls(handle)

    def close(self):
        if getattr(self, "_h", None):
            lib().relp_bix_free(self._h)
            self._h = None

    __del__ = close

    def _check(self, status):
        if status != OK:
            raise RelpError(status, (lib().relp_bix_last_error(self._h) or b"").decode())

    def m(self):
        out = C.c_int32()
        self._check(lib().relp_bix_m(self._h, C.byref(out)))
        return out.value

    def result_words(self):
        out = C.c_int32()
        self._check(lib().relp_bix_result_words(self._h, C.byref(out)))
        return out.value

    def _vector(self, call, count):
        """Run `call(capacity, numerators, denominator, words)`; retry once when the object widened meanwhile."""
        from fractions import Fraction
        for _ in range(3):
            capacity = self.result_words() + 2
            numerators = np.zeros(count * capacity, dtype=np.uint64)
            denominator = np.zeros(capacity, dtype=np.uint64)
            words = C.c_int32()
            status = call(capacity, _ptr(numerators, C.c_uint64), _ptr(denominator, C.c_uint64), C.byref(words))
            if status == OK:
                w = words.value
                d = _integer(denominator[:w])
                assert d > 0
                return [Fraction(_integer(numerators[e * w:(e + 1) * w]), d) for e in range(count)]
            if words.value <= capacity:
                self._check(status)
        self._check(status)

    def left_multiply_by_basis_inverse(self, column):
        nnz, index, num, den = _exact_sparse(column)
        return self._vector(lambda cap, n, d, w: lib().relp_bix_left_multiply(self._h, nnz, _ptr(index, C.c_int32), _ptr(num, C.c_int64),
                                                                              _ptr(den, C.c_int64), cap, n, d, w), self.m())

    def right_multiply_by_basis_inverse(self, row):
        """int64 (numerator, denominator) pairs where the vector fits them -- the reference's `Rational64` inputs -- else multi-word integers
        over one denominator (``relp_bix_right_multiply_words``: the row vectors `Carry::change_basis` forms are `RationalBig`)."""
        import math
        from fractions import Fraction
        pairs = [(int(i), Fraction(v)) for i, v in row]
        common = 1
        for _, v in pairs:
            common = common * v.denominator // math.gcd(common, v.denominator)
        scaled = [v.numerator * (common // v.denominator) for _, v in pairs]
        if common < (1 << 61) and all(abs(n) < (1 << 61) for n in scaled):
            nnz, index, num, den = _exact_sparse(pairs)
            return self._vector(lambda cap, n, d, w: lib().relp_bix_right_multiply(self._h, nnz, _ptr(index, C.c_int32), _ptr(num, C.c_int64),
                                                                                   _ptr(den, C.c_int64), cap, n, d, w), self.m())
        bits = max([common.bit_length()] + [abs(n).bit_length() for n in scaled]) + 2
        vw = (bits + 63) // 64
        mask = (1 << (64 * vw)) - 1

        def words_of(value):
            value &= mask
            return [(value >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(vw)]
        values = np.array([w for n in scaled for w in words_of(n)] or [0], dtype=np.uint64)
        denominator_in = np.array(words_of(common), dtype=np.uint64)
        index = np.array([i for i, _ in pairs] or [0], dtype=np.int32)
        m = self.m()
        capacity = self.result_words() + vw + 2
        numerators = np.zeros(m * capacity, dtype=np.uint64)
        denominator = np.zeros(capacity, dtype=np.uint64)
        words = C.c_int32()
        self._check(lib().relp_bix_right_multiply_words(self._h, len(pairs), _ptr(index, C.c_int32), vw, _ptr(values, C.c_uint64),
                                                        _ptr(denominator_in, C.c_uint64), capacity, _ptr(numerators, C.c_uint64),
                                                        _ptr(denominator, C.c_uint64), C.byref(words)))
        w = words.value
        d = _integer(denominator[:w])
        assert d > 0
        return [Fraction(_integer(numerators[e * w:(e + 1) * w]), d) for e in range(m)]

    def basis_inverse_row(self, row):
        return self._vector(lambda cap, n, d, w: lib().relp_bix_basis_inverse_row(self._h, int(row), cap, n, d, w), self.m())

    def generate_element(self, i, column):
        """``None`` when the element is zero (the reference returns ``Option``)."""
        nnz, index, num, den = _exact_sparse(column)
        some = C.c_int32()
        value = self._vector(lambda cap, n, d, w: lib().relp_bix_generate_element(self._h, int(i), nnz, _ptr(index, C.c_int32), _ptr(num, C.c_int64),
                                                                                  _ptr(den, C.c_int64), cap, n, d, w, C.byref(some)), 1)[0]
        return value if some.value else None

    def change_basis(self, pivot_row_index):
        self._check(lib().relp_bix_change_basis(self._h, int(pivot_row_index)))

    def should_refactor(self):
        out = C.c_int32()
        self._check(lib().relp_bix_should_refactor(self._h, C.byref(out)))
        return bool(out.value)

    def remove_basis_part(self, indices):
        idx = np.ascontiguousarray(list(indices), dtype=np.int32)
        self._check(lib().relp_bix_remove_basis_part(self._h, len(idx), _ptr(idx, C.c_int32)))
