"""Host big-integer arithmetic of the exact certificate (relp_amd/csrc/bigint.hpp) against Python integers (CPU only)."""
import math
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bigint") / "bigint_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "support", "bigint_check.cpp"), "-o", exe])
    return exe


def trunc_div(a, b):
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def test_bigint_matches_python(driver):
    rng = random.Random(12345)
    cases = []
    for _ in range(600):
        bits_a = rng.choice([1, 31, 32, 33, 63, 64, 65, 127, 500, 1791, 4000])
        bits_b = rng.choice([1, 31, 32, 33, 63, 64, 65, 127, 500, 1791])
        a = rng.getrandbits(bits_a) * rng.choice([1, -1])
        b = (rng.getrandbits(bits_b) or 1) * rng.choice([1, -1])
        op = rng.choice(["add", "sub", "mul", "div", "mod", "gcd", "cmp"])
        cases.append((op, a, b))
    cases += [("div", 2 ** 64, 2 ** 32), ("mod", 2 ** 96 - 1, 2 ** 64 - 1), ("div", (2 ** 64 - 1) * (2 ** 64 - 1), 2 ** 64 - 1),
              ("mul", 0, 5), ("sub", 7, 7), ("div", 5, 7), ("mod", -5, 7), ("div", -(2 ** 70), 3)]
    text = "".join("%s %d %d\n" % c for c in cases)
    out = subprocess.run([driver], input=text, capture_output=True, text=True, check=True).stdout.split()
    assert len(out) == len(cases)
    for (op, a, b), got in zip(cases, out):
        if op == "add":
            want = a + b
        elif op == "sub":
            want = a - b
        elif op == "mul":
            want = a * b
        elif op == "div":
            want = trunc_div(a, b)
        elif op == "mod":
            want = a - trunc_div(a, b) * b
        elif op == "gcd":
            want = math.gcd(a, b)
        else:
            want = (a > b) - (a < b)
        assert int(got) == want, (op, a, b)


def test_bigrat_matches_python_fractions(tmp_path):
    """Arbitrary-precision rationals of the presolve (relp_amd/csrc/bigrat.hpp) against fractions.Fraction."""
    from fractions import Fraction
    exe = str(tmp_path / "bigrat_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "support", "bigrat_check.cpp"), "-o", exe])
    rng = random.Random(777)
    sizes = [1, 8, 30, 61, 62, 63, 64, 65, 100, 126, 127, 128, 200, 500, 2000]

    def fraction():
        return Fraction(rng.getrandbits(rng.choice(sizes)) * rng.choice([1, -1]), rng.getrandbits(rng.choice(sizes)) + 1)

    lines, expected = [], []
    for _ in range(2500):
        op = rng.choice("+-*/cr")
        x, y = fraction(), fraction()
        if rng.random() < 0.1:
            y = x * rng.choice([1, -1, 3])
        if op == "/" and y == 0:
            y = Fraction(1)
        lines.append("%s %d/%d %d/%d" % (op, x.numerator, x.denominator, y.numerator, y.denominator))
        if op == "c":
            expected.append(str((x > y) - (x < y)))
        elif op == "r":
            fits = x.numerator.bit_length() <= 126 and x.denominator.bit_length() <= 126
            expected.append("%d/%d" % (x.numerator, x.denominator) if fits else "overflow")
        else:
            r = x + y if op == "+" else x - y if op == "-" else x * y if op == "*" else x / y
            expected.append("%d/%d" % (r.numerator, r.denominator))
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.split()
    assert out == expected


def reconstruct_reference(a, M):
    """The definition relp_amd/csrc/rational_reconstruct.hpp implements: the extended Euclidean sequence of (M, a mod M), stopped at
    the first remainder r with 2 bits(r) + 2 <= bits(M); the cofactor must satisfy the same bound."""
    small = lambda v: 2 * abs(v).bit_length() + 2 <= M.bit_length()
    r0, r1, t0, t1 = M, a % M, 0, 1
    while not small(r1):
        q = r0 // r1
        r0, r1, t0, t1 = r1, r0 - q * r1, t1, t0 - q * t1
    if t1 == 0 or not small(t1):
        return None
    n, d = (-r1 if t1 < 0 else r1), abs(t1)
    g = math.gcd(n, d)
    return (n // g, d // g) if g else None


def test_rational_reconstruction_lehmer_equals_plain_equals_python(driver):
    """Lehmer's batched sequence is the plain sequence: same answer on every input, including the ones with no answer."""
    rng = random.Random(2024)
    p = 2147483647
    cases = []
    for k in (2, 4, 8, 9, 16, 32, 33, 64, 128, 130):
        M = p ** k
        for _ in range(12):
            bits = rng.randrange(1, max(2, (M.bit_length() - 2) // 2))
            n = rng.getrandbits(bits) * rng.choice([1, -1])
            d = rng.getrandbits(rng.randrange(1, bits + 1)) | 1
            while math.gcd(d, p) != 1:
                d += 2
            cases.append((n * pow(d, -1, M) % M, M))      # a reconstructible residue
            cases.append((rng.randrange(M), M))            # a random one (almost surely none)
        cases.append((0, M))
        cases.append((1, M))
        cases.append((M - 1, M))
    for _ in range(60):                                      # moduli that are not prime powers, odd sizes
        M = rng.getrandbits(rng.choice([70, 129, 131, 200, 257, 1000])) | (1 << 69) | 1
        cases.append((rng.randrange(M), M))
        n, d = rng.getrandbits(30), rng.getrandbits(28) | 1
        if math.gcd(d, M) == 1:
            cases.append((n * pow(d, -1, M) % M, M))
    is_prime_power = lambda M: M % p == 0 or M == 1
    text = "".join("%s %d %d\n" % (op if op != "rrprime" or is_prime_power(M) else "rr", a, M) for a, M in cases for op in ("rr", "rrplain", "rrprime"))
    out = subprocess.run([driver], input=text, capture_output=True, text=True, check=True).stdout.split()
    assert len(out) == 3 * len(cases)
    found = 0
    for k, (a, M) in enumerate(cases):
        want = reconstruct_reference(a, M)
        want_text = "none" if want is None else "%d/%d" % want
        assert out[3 * k] == want_text, ("lehmer", a, M)
        assert out[3 * k + 1] == want_text, ("plain", a, M)
        # with the prime given the gcd is replaced by a divisibility test of the cofactor: "none" where the gcd would have been > 1
        assert out[3 * k + 2] == want_text or (out[3 * k + 2] == "none" and is_prime_power(M)), ("prime", a, M)
        if is_prime_power(M) and out[3 * k + 2] == "none" and want is not None:
            assert want[1] % p == 0 or (a * want[1] - want[0]) % M != 0  # what was dropped was not a fraction of a (mod M)
        found += want is not None
    assert found >= 100
