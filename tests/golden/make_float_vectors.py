#!/usr/bin/env python
"""Extracts every literal input of the reference's FloatingPointNumberSchemaBasedParsingTest whose test method is about
binary32 (method name contains "Float") into tests/golden/float_vectors.json, with the expectation the test asserts where
it is a constant (0.0f / -0.0f / infinities / Float.MIN_NORMAL ...), else null (the tests then require the correctly
rounded binary32 value, computed exactly with Python's Decimal).  Run in the build container (reads /root/reference);
the JSON it writes is committed."""
import json
import os
import re

SRC = "/root/reference/src/test/java/org/simdjson/FloatingPointNumberSchemaBasedParsingTest.java"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "float_vectors.json")
CONST = {"0.0f": "0x00000000", "-0.0f": "0x80000000", "Float.POSITIVE_INFINITY": "0x7f800000", "Float.NEGATIVE_INFINITY": "0xff800000",
         "Float.MIN_NORMAL": "0x00800000", "Float.MIN_VALUE": "0x00000001", "Float.MAX_VALUE": "0x7f7fffff", "0x0.fffffep-126f": "0x007fffff"}

text = open(SRC, encoding="utf-8").read()
methods = re.split(r"\n    @(?:CartesianTest|ParameterizedTest|Test)", text)
vectors = []
for m in methods:
    name = re.search(r"public void (\w+)\(", m)
    if not name or not re.search(r"(Float(?!ingPoint)|^float)", name.group(1)):
        continue
    vals = re.search(r"@Values\(strings = \{(.*?)\}\)", m, re.S)
    if not vals:
        continue
    lits = re.findall(r'"((?:[^"\\]|\\.)*)"', vals.group(1))
    exp = re.search(r"isEqualTo\(([^)]*)\)", m)
    want = CONST.get(exp.group(1).strip()) if exp else None
    line = text[:text.index(m[:60])].count("\n") + 1 if m[:60] in text else 0
    for lit in lits:
        vectors.append({"input": lit, "bits": want, "test": name.group(1), "line": line})
json.dump(vectors, open(OUT, "w"), indent=0)
print(len(vectors), "float vectors ->", OUT)
