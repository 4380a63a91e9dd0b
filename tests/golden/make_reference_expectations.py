"""Extract the expected optima the reference's Netlib tests assert (run in the build container only).

Reads /root/reference/tests/netlib/test.rs and writes tests/golden/netlib_expected.json:
{name: {"expected": float, "tolerance": float, "ignored": reason-or-null}} -- data, not code.
"""
import json
import os
import re

SRC = "/root/reference/tests/netlib/test.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "netlib_expected.json")

text = open(SRC).read()
out = {}
pattern = re.compile(
    r'(#\[ignore\s*=\s*"(?P<why>[^"]*)"\]\s*)?fn test_\w+\(\) \{\s*let result = solve\("(?P<name>[^"]+)"\);\s*'
    r'let expected = (?P<value>[-0-9.e+]+);.*?abs\(\) < RB!\((?P<tol>[-0-9.e+]+)\)', re.S)
for match in pattern.finditer(text):
    out[match.group("name")] = {"expected": float(match.group("value")), "tolerance": float(match.group("tol")),
                                "ignored": match.group("why")}
json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
print(len(out), "expectations written")
