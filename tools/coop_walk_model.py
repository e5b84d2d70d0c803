#!/usr/bin/env python
"""Model of the COOPERATIVE stage 2 planned in DESIGN.md 4.4 (one wave per document, lane = structural): every grammar
test of JsonIterator.walkDocument as a LOCAL predicate over (this structural, its predecessor, the kind of the
container it sits in), the document's error = the predicate failing at the lowest position, tape positions / container
words / element counts from prefix sums and bracket matching.  Design validation only (tests/test_coop_walk_model.py
checks it against the oracle's sequential walker on the grammar vectors and fuzzed documents); nothing ships from here.

The per-structural quantities a kernel would get from scans are computed here with plain loops -- the point of the
model is WHICH quantities are needed and that the local rules reproduce the reference's first error exactly."""
import struct

import numpy as np

E = dict(NO_STRUCTURAL=9, UNCLOSED_OBJECT=10, UNCLOSED_ARRAY=11, OBJECT_NO_KEY=12, MISSING_COLON=13, KEY_MISSING=14,
         NO_COMMA_OBJECT=15, NO_COMMA_ARRAY=16, TRAILING=17, UNRECOGNIZED=18, TRUE=19, FALSE=20, NULL=21, MINUS=22,
         LEADING=23, DEC=24, EXP=25, FOLLOWED=26, LONG=27, DEPTH=28)
SOW = b" \n\r\t,:[]{}"


def parse_number(buf, p, limit):
    """NumberParser.parseNumber (NumberParser.java:23-74): ('err', code) | ('l', u64) | ('d', bits)"""
    def B(q):
        return buf[q] if q < limit else 0x20
    start = p
    neg = B(p) == 0x2D
    if neg:
        p += 1
    ds = p
    while 0x30 <= B(p) <= 0x39:
        p += 1
    if p == ds:
        return ("err", E["MINUS"])
    if B(ds) == 0x30 and p - ds > 1:
        return ("err", E["LEADING"])
    digits = int(bytes(buf[ds:p])) if p <= limit else 0
    ndig = p - ds
    floating = False
    if B(p) == 0x2E:
        floating = True
        p += 1
        a = p
        while 0x30 <= B(p) <= 0x39:
            p += 1
        if p == a:
            return ("err", E["DEC"])
    if B(p) in b"eE":
        floating = True
        p += 1
        if B(p) in b"+-":
            p += 1
        a = p
        while 0x30 <= B(p) <= 0x39:
            p += 1
        if p == a:
            return ("err", E["EXP"])
    if B(p) not in SOW:
        return ("err", E["FOLLOWED"])
    if floating:
        v = float(bytes(buf[start:p]).decode())
        return ("d", struct.unpack("<Q", struct.pack("<d", v))[0])
    if ndig > 19 or (ndig == 19 and not (digits < 2 ** 63 or (neg and digits == 2 ** 63))):
        return ("err", E["LONG"])
    return ("l", (-digits if neg else digits) & (2 ** 64 - 1))


def primitive(buf, idx, end):
    """visitRootPrimitive's form (TapeBuilder.java:59-68, :100-168), which also serves inside containers:
    ('err', code) | list of tape words (type char, payload)"""
    c = buf[idx]
    for lit, code in ((b"true", E["TRUE"]), (b"false", E["FALSE"]), (b"null", E["NULL"])):
        if c == lit[0]:
            n = len(lit)
            ok = idx + n <= end and bytes(buf[idx:idx + n]) == lit and (idx + n == end or buf[idx + n] in SOW)
            return [(chr(c), 0)] if ok else ("err", code)
    if c == 0x2D or 0x30 <= c <= 0x39:
        r = parse_number(buf, idx, end)
        return r if r[0] == "err" else [(r[0], 0), (None, r[1])]
    return ("err", E["UNRECOGNIZED"])


def walk(buf, doc_start, doc_end, idx, string_sizes, string_errors, max_depth=1024):
    """buf: bytes-like of the batch; idx: this document's structural positions; string_sizes[i]: 4 + length of the
    record of structural i (0 if it is no quote); string_errors[i]: SJMI_E_* of a failing string or 0.
    -> (error code, tape as np.uint64 array or None)"""
    n = len(idx)
    if n == 0:
        return E["NO_STRUCTURAL"], None
    c = [buf[p] for p in idx]
    OPEN, CLOSE = b"{[", b"}]"
    # ---- (1) empty containers: an opening bracket directly followed by its closing bracket is ONE value
    empty_open = [i + 1 < n and c[i] in OPEN and c[i + 1] == c[i] + 2 for i in range(n)]
    empty_close = [i > 0 and empty_open[i - 1] for i in range(n)]
    # ---- (2) depth, enclosing container, matching bracket (scans / bracket matching in a kernel)
    depth_before = [0] * (n + 1)   # open containers in front of structural i
    parent = [-1] * (n + 1)        # the opening bracket of the container structural i sits in
    match = [-1] * n
    stack = []
    for i in range(n):
        depth_before[i] = len(stack)
        parent[i] = stack[-1] if stack else -1
        if c[i] in OPEN and not empty_open[i]:
            stack.append(i)
        elif c[i] in CLOSE and not empty_close[i] and stack:
            match[i] = stack.pop()
            match[match[i]] = i
    depth_before[n] = len(stack)
    parent[n] = stack[-1] if stack else -1
    # the root value ends at `root_end` (exclusive): structurals behind it are never looked at (only counted: TRAILING)
    if c[0] in OPEN:
        root_end = 2 if empty_open[0] else (match[0] + 1 if match[0] >= 0 else n + 1)
    else:
        root_end = 1
    # ---- (3) local predicates.  role of a structural = f(predecessor's class and role, kind of its container)
    sentinel = buf[doc_start]
    errors = [0] * (n + 1)

    def in_array(i):
        return c[parent[i]] == 0x5B

    is_key = [False] * (n + 1)
    for i in range(min(root_end, n + 1)):
        ci = c[i] if i < n else sentinel   # past the end: BitIndexes' sentinel = the document's first byte
        if i < n and empty_close[i]:
            continue  # consumed together with its opening bracket
        if i == 0:
            if ci in OPEN and c[n - 1] != ci + 2:
                errors[0] = E["UNCLOSED_OBJECT"] if ci == 0x7B else E["UNCLOSED_ARRAY"]
                break
            expect = "value"
        else:
            j = i - 1  # predecessor (an empty container's closing bracket counts as the end of a value)
            pj = c[j]
            if pj in OPEN and not empty_open[j]:
                expect = "value" if pj == 0x5B else "first_key"
            elif pj == 0x2C:
                expect = "value" if in_array(i) else "key"
            elif pj == 0x3A:
                expect = "value"
            elif is_key[j]:
                expect = "colon"
            else:
                expect = "sep"  # the predecessor ended a value
        if expect in ("first_key", "key"):
            if ci != 0x22:
                errors[i] = E["OBJECT_NO_KEY"] if expect == "first_key" else E["KEY_MISSING"]
            else:
                is_key[i] = True
                errors[i] = string_errors[i] if i < n else 0
        elif expect == "colon":
            if ci != 0x3A:
                errors[i] = E["MISSING_COLON"]
        elif expect == "sep":
            arr = in_array(i)
            if ci != 0x2C and ci != (0x5D if arr else 0x7D):
                errors[i] = E["NO_COMMA_ARRAY"] if arr else E["NO_COMMA_OBJECT"]
        else:  # value
            if ci in OPEN:
                if not (i < n and empty_open[i]) and depth_before[i] + 1 >= max_depth:
                    errors[i] = E["DEPTH"]  # JsonIterator.java:69-70 (an empty container spends no depth)
            elif ci == 0x22:
                errors[i] = string_errors[i] if i < n else 0
            else:
                r = primitive(buf, idx[i] if i < n else doc_start, doc_end)
                if isinstance(r, tuple):
                    errors[i] = r[1]
        if errors[i]:
            break
    first = next((e for e in errors if e), 0)
    if first:
        return first, None
    if root_end < n:
        return E["TRAILING"], None
    # ---- (4) the tape: positions from a prefix sum of words per structural, container words from match / counts
    words = []
    commas = [0] * n  # commas directly inside the container opened at i
    for i in range(n):
        if c[i] == 0x2C:
            commas[parent[i]] += 1
    str_off = 0
    out = [None]  # tape[0]: root
    pos = [0] * n
    for i in range(n):
        pos[i] = len(out)
        ci = c[i]
        if empty_close[i] or ci in b",:":
            continue
        if empty_open[i]:
            out.append((chr(ci), pos[i] + 2))
            out.append((chr(ci + 2), pos[i] + 1))
        elif ci in OPEN:
            out.append(("open", i))
        elif ci in CLOSE:
            out.append((chr(ci), pos[match[i]]))
        elif ci == 0x22:
            out.append(('"', str_off))
            str_off += string_sizes[i]
        else:
            out.extend(primitive(buf, idx[i], doc_end))
    for i in range(n):
        if c[i] in OPEN and not empty_open[i]:
            cnt = min(commas[i] + 1, 0xFFFFFF)
            out[pos[i]] = (chr(c[i]), (pos[match[i]] + 1) | (cnt << 32))
    out.append(("r", 0))
    out[0] = ("r", len(out))
    tape = np.zeros(len(out), dtype=np.uint64)
    for k, (t, v) in enumerate(out):
        tape[k] = np.uint64(v if t is None else (v | (ord(t) << 56)))
    return 0, tape
