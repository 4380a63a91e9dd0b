"""The exact simplex's update of N on the matrix cores (round 5, relp_amd/csrc/exact.hip): the pass that turns what the MFMA tiles leave
(pairs of words plus a carry per pair) into the entries of N, checked against Python integers -- carries of both signs that ripple
through words and pairs, shifts of 64 bits and more, negation, the bit length of every magnitude (zero, -1, -(2^k) included).  The
tiles themselves are pinned end to end by the golden pivot sequences (tests/test_gpu_exact.py runs on the matrix cores from 32 limbs on,
and on the vector unit only under exact_update = 1: test_whole_traces_on_the_matrix_cores_and_on_the_vector_unit below)."""
import ctypes as C
import json
import os
import random
from fractions import Fraction

import numpy as np
import pytest

import relp_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def decompose(value, words, rng, ripple):
    """`value` (signed, fits 64 * words bits) as the tiles would leave it: unsigned 128-bit pairs X_P and carries c_P with
    sum_P (X_P + c_(P-1)) 2^(128 P) == value modulo 2^(64 words), built pair by pair with random carries."""
    modulus = 1 << (64 * words)
    digits = [(value % modulus >> (128 * p)) & ((1 << 128) - 1) for p in range(words // 2)]
    pairs, carries, carry_in = [], [], 0
    for p, digit in enumerate(digits):
        x = (digit - carry_in) % (1 << 128)
        ripple_out = (x + carry_in) >> 128  # floor: -1, 0 or 1
        assert (x + carry_in) % (1 << 128) == digit and ripple_out in (-1, 0, 1)
        stored = rng.choice([0, 1, -1, rng.randrange(-(1 << 20), 1 << 20)]) if ripple else rng.randrange(-(1 << 20), 1 << 20)
        pairs.append(x)
        carries.append(stored)
        carry_in = stored + ripple_out
    return pairs, carries


def run_finish(limbs, values, words, shift, flip, rng, ripple):
    count = len(values)
    T = np.zeros((limbs, count), dtype=np.uint64)
    carry = np.zeros((limbs // 2, count), dtype=np.int32)
    for e, (value, w) in enumerate(zip(values, words)):
        pairs, carries = decompose(value, w, rng, ripple)
        for p, (x, c) in enumerate(zip(pairs, carries)):
            T[2 * p, e] = x & ((1 << 64) - 1)
            T[2 * p + 1, e] = x >> 64
            carry[p, e] = c
        # words above the valid ones hold whatever an earlier, wider numerator left there
        for k in range(w, limbs):
            T[k, e] = rng.getrandbits(64)
        for p in range(w // 2, limbs // 2):
            carry[p, e] = rng.randrange(-5, 5)
    words_arr = np.array(words, dtype=np.int32)
    N = np.zeros((limbs, count), dtype=np.uint64)
    bits = np.zeros(count, dtype=np.int32)
    status = relp_amd.lib().relp_debug_exact_finish(0, limbs, count, T.ctypes.data_as(C.POINTER(C.c_uint64)), carry.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    words_arr.ctypes.data_as(C.POINTER(C.c_int32)), shift, int(flip),
                                                    N.ctypes.data_as(C.POINTER(C.c_uint64)), bits.ctypes.data_as(C.POINTER(C.c_int32)))
    assert status == 0
    out = []
    for e in range(count):
        v = sum(int(N[k, e]) << (64 * k) for k in range(limbs))
        if v >> (64 * limbs - 1):
            v -= 1 << (64 * limbs)
        out.append(v)
    return out, [int(b) for b in bits]


@pytest.mark.parametrize("limbs", [16, 32, 64, 128])
@pytest.mark.parametrize("shift", [0, 1, 7, 63, 64, 65, 130, 200])
@pytest.mark.parametrize("flip", [False, True])
def test_finish_pass_against_python_integers(limbs, shift, flip):
    rng = random.Random(1000 * limbs + 10 * shift + int(flip))
    values, words = [], []
    special = [0, -1, 1, -(1 << 64), 1 << 64, -(1 << 63), (1 << 63), -(1 << 127), (1 << 128) - 1, -(1 << 128), (1 << 200), -(1 << 200) + 1, -(1 << 255)]
    for trial in range(400):
        w = 8 * rng.randrange(1, limbs // 8 + 1)
        if trial < len(special) * 2:
            value = special[trial // 2] << (shift if trial % 2 else 0)
            if not -(1 << (64 * w - 1)) <= value < (1 << (64 * w - 1)):
                value = special[trial // 2]
                w = max(w, 8 * ((value.bit_length() + 1 + 511) // 512))
                if w > limbs:
                    value, w = 0, 8
        else:
            bits = rng.randrange(1, 64 * w - 1)
            value = rng.getrandbits(bits) * rng.choice([1, -1])
            if rng.random() < 0.3:  # long runs of zeros / ones: carries that travel
                value = (value >> (bits // 2)) << (bits // 2)
            if rng.random() < 0.2:
                value = -(1 << rng.randrange(0, 64 * w - 1))
        values.append(value)
        words.append(w)
    for ripple in (False, True):
        got, bits = run_finish(limbs, values, words, shift, flip, rng, ripple)
        for value, g, b in zip(values, got, bits):
            expected = value >> shift  # floor = arithmetic shift
            if flip:
                expected = -expected
            assert g == expected, (value, shift, flip)
            assert b == abs(expected).bit_length(), (value, shift, flip, b)


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("name", ["STOCFOR1", "SHARE1B", "E226", "BANDM", "SCSD1"])
def test_whole_traces_on_the_matrix_cores_and_on_the_vector_unit(name, monkeypatch):
    """The same golden pivot sequences with the update on the matrix cores (from 32 limbs on: these LPs need 32 or 64) and on the vector
    unit only (`relp_options.exact_update = 1`): identical traces, pivot counts, final bases and optima."""
    golden = json.load(open(os.path.join(GOLDEN, name + ".json")))
    results = []
    for mode in (0, 4, 1):  # round 6: 0 = the tiles finish their entries themselves (N double-buffered), 4 = the two passes of round 5, 1 = vector unit
        solver = relp_amd.Solver(exact_update=mode).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
        got = solver.solve_exact(first_limbs=4, max_limbs=64)
        counters = solver.exact_counters()
        solver.close()
        assert got["status"] == 1 and got["limbs"] >= 32 and Fraction(got["objective"]) == Fraction(golden["objective"])
        assert (got["pivots_phase_one"], got["pivots_phase_two"]) == (golden["pivots_phase1"], golden["pivots_phase2"])
        results.append((got["trace"], list(got["basis"]), got["objective"]))
        assert counters and counters[-1]["update_word_products_issued"] > 0
    assert results[0] == results[1] == results[2]


@pytest.mark.parametrize("name", ["STOCFOR1", "SHARE1B", "E226", "BANDM", "SCSD1", "ISRAEL", "AFIRO", "ADLITTLE", "BLEND", "SCAGR7", "SHARE2B", "KB2",
                                  "SCRS8", "CZPROB", "GFRD-PNC", "SC205", "LOTFI", "BRANDY", "SCTAP1", "AGG"])
def test_weight_estimates_from_leading_words_and_exact_products_choose_the_same_pivots(name):
    """The pricing pass estimates the steepest-edge weights from the four leading words of every operand of N a_j, with an error bound per
    term, and forms a column's products exactly only where the bound asks (price_estimates / price_products in exact.hip).  Bit 2 of
    `relp_options.exact_update` sends EVERY column that can enter down the exact path: same golden trace, basis and optimum either way."""
    golden = json.load(open(os.path.join(GOLDEN, name + ".json")))
    results = []
    for mode in (0, 2):
        solver = relp_amd.Solver(exact_update=mode).load_mps(os.path.join(ROOT, "data", "netlib", name + ".SIF"))
        got = solver.solve_exact(first_limbs=4, max_limbs=64)
        solver.close()
        assert got["status"] == 1 and Fraction(got["objective"]) == Fraction(golden["objective"])
        assert (got["pivots_phase_one"], got["pivots_phase_two"]) == (golden["pivots_phase1"], golden["pivots_phase2"])
        results.append((got["trace"], list(got["basis"]), got["objective"]))
    assert results[0] == results[1]


def _words_of(value, limbs):
    return [(value >> (64 * k)) & ((1 << 64) - 1) for k in range(limbs)]


@pytest.mark.parametrize("limbs", [16, 32, 64, 128])
def test_word_arithmetic_of_the_pivots_scalars_against_python_integers(limbs):
    """`wave_inverse_odd` (1 / D_odd modulo 2^(64 L): Newton's doublings by one wave on blocks of four words) and `wave_mul_lo_negated` (the
    rows' factors -alpha_i u and y's factor, every lane at work, carries from lane to lane, the two's complement by the lanes together)
    through `relp_debug_exact_words`, against Python's integers: random operands and the ones that stress the carries (0, 1, -1, 2^k,
    all ones, runs of zero words)."""
    rng = np.random.default_rng(limbs)
    modulus = 1 << (64 * limbs)
    specials = [0, 1, modulus - 1, 1 << 63, (1 << 64) - 1, 1 << 64, 1 << (64 * limbs - 1), (1 << (64 * (limbs // 2))) - 1,
                ((1 << 64) - 1) << (64 * (limbs - 1)), modulus - (1 << 64), 3, (1 << 192) + 1]
    randoms = [int.from_bytes(rng.bytes(8 * limbs), "little") for _ in range(20)]
    sparse = [sum(int(rng.integers(0, 1 << 63)) << (64 * int(k)) for k in rng.choice(limbs, size=3, replace=False)) for _ in range(6)]
    operands = specials + randoms + sparse
    pairs = [(x, y) for x in operands for y in (operands[:6] + randoms[:4])]
    for mode in (1, 2):
        a = np.array([w for x, _ in pairs for w in _words_of(x, limbs)], dtype=np.uint64)
        b = np.array([w for _, y in pairs for w in _words_of(y, limbs)], dtype=np.uint64)
        out = np.zeros_like(a)
        status = relp_amd.lib().relp_debug_exact_words(0, limbs, mode, len(pairs), a.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                       b.ctypes.data_as(C.POINTER(C.c_uint64)), out.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert status == 0
        for e, (x, y) in enumerate(pairs):
            got = sum(int(out[e * limbs + k]) << (64 * k) for k in range(limbs))
            want = (x * y) % modulus if mode == 2 else (-(x * y)) % modulus
            assert got == want, (limbs, mode, hex(x)[:40], hex(y)[:40])
    odd = [x | 1 for x in operands]
    a = np.array([w for x in odd for w in _words_of(x, limbs)], dtype=np.uint64)
    out = np.zeros_like(a)
    status = relp_amd.lib().relp_debug_exact_words(0, limbs, 0, len(odd), a.ctypes.data_as(C.POINTER(C.c_uint64)), a.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   out.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert status == 0
    for e, x in enumerate(odd):
        got = sum(int(out[e * limbs + k]) << (64 * k) for k in range(limbs))
        assert (got * x) % modulus == 1, (limbs, hex(x)[:40])
