#!/usr/bin/env python
"""Extracts the literal vectors of the reference's NumberParsingTest (inputs and the value / message each test
asserts) into tests/golden/number_vectors.json.  Run in the build container, where /root/reference exists; the JSON
travels with the repo (the GPU box has no reference).  Only test DATA is taken: the strings of @ValueSource /
@CsvSource / toUtf8("...") and the asserted constants.

usage: python tests/golden/make_number_vectors.py [/root/reference]"""
import json
import os
import re
import struct
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
path = os.path.join(ref, "src", "test", "java", "org", "simdjson", "NumberParsingTest.java")
lines = open(path, encoding="utf-8").read().split("\n")


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


# what a test asserts for every input of its source annotation (the assert lines are cited in the output)
FIXED = {
    "positiveInfinity": float("inf"), "negativeInfinity": float("-inf"), "positiveZero": 0.0, "negativeZero": -0.0,
    "roundingOverflow": 7.2057594037927936e16, "minNormalDouble": float.fromhex("0x1p-1022"),
    "maxSubnormalDouble": float.fromhex("0x0.fffffffffffffp-1022"), "minSubnormalDouble": float.fromhex("0x0.0000000000001p-1022"),
    "maxDouble": float.fromhex("0x1.fffffffffffffp+1023"),
}
out = []
i = 0
while i < len(lines):
    m = re.search(r"@(ValueSource|CsvSource)\(", lines[i])
    if not m:
        i += 1
        continue
    start = i
    block = lines[i]
    while not re.search(r"public void (\w+)\(", lines[i]):
        i += 1
        block += "\n" + lines[i]
    method = re.search(r"public void (\w+)\(", lines[i]).group(1)
    cite = "NumberParsingTest.java:%d-%d" % (start + 1, i + 1)
    strings = [bytes(s, "utf-8").decode("unicode_escape") for s in re.findall(r'"((?:[^"\\]|\\.)*)"', block.split("public void")[0])]
    # the message of assertThrows tests: the next .hasMessage("...") inside the method body
    body_end = i
    while not lines[body_end].startswith("    }"):
        body_end += 1
    body = "\n".join(lines[i:body_end])
    msg = re.search(r'\.hasMessage\("((?:[^"\\]|\\.)*)"\)', body)
    if m.group(1) == "ValueSource" and "longs" in block:
        i += 1
        continue  # Long.MIN/MAX: already among the hand-written range vectors
    for s in strings:
        if m.group(1) == "CsvSource":
            inp, exp = [t.strip() for t in s.split(",")]
            out.append({"input": inp, "double_bits": bits(float(exp)), "cite": cite, "test": method})
        elif msg:
            out.append({"input": s, "message": bytes(msg.group(1), "utf-8").decode("unicode_escape"), "cite": cite, "test": method})
        elif method == "doubleAtRoot":
            out.append({"input": s, "double_bits": bits(float(s)), "cite": cite, "test": method})  # Double.parseDouble(input)
        else:
            out.append({"input": s, "double_bits": bits(FIXED[method]), "cite": cite, "test": method})
    i += 1
# the three @Test methods with a literal document
out.append({"input": "-0", "long": 0, "cite": "NumberParsingTest.java:177-189", "test": "minusZeroIsTreatedAsIntegerZero"})
out.append({"input": "+1", "message": "Unrecognized primitive. Expected: string, number, 'true', 'false' or 'null'.",
            "cite": "NumberParsingTest.java:191-203", "test": "startingWithPlusIsNotAllowed"})
out.append({"input": "1234", "length": 2, "long": 12, "cite": "NumberParsingTest.java:634-646", "test": "passedLengthSmallerThanNumberLength"})
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "number_vectors.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=0)
print("%d vectors from %d tests -> %s" % (len(out), len({v["test"] for v in out}), dst))
