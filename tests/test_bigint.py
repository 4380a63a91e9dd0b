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
