"""Document generators of tools/soak_strings.py / tools/soak_batch.py (escape-heavy strings, every error family)."""
import random

ESC = ['\\"', "\\\\", "\\/", "\\b", "\\f", "\\n", "\\r", "\\t"]
BAD = ["\\q", "\\u12G4", "\\u12", "\\uD800", "\\uD800\\n", "\\uD800\\u0041", "\\uDC00", "\\uDFFF\\uD800", "\\x41", "\\U0041", "\\ ", "\\u"]


def string(rng, target, p_bad):
    out = []
    n = 0
    while n < target:
        r = rng.random()
        if r < 0.45:
            k = rng.randint(1, 40)
            t = "".join(rng.choice("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 ,:[]{}/+-_.") for _ in range(k))
        elif r < 0.60:
            t = rng.choice(ESC)
        elif r < 0.72:
            cp = rng.choice([rng.randrange(0x20, 0x80), rng.randrange(0x80, 0x800), rng.randrange(0x800, 0xD800), rng.randrange(0xE000, 0x10000)])
            t = "\\u%04x" % cp if rng.random() < 0.5 else "\\u%04X" % cp
        elif r < 0.80:
            cp = rng.randrange(0x10000, 0x110000) - 0x10000
            t = "\\u%04x\\u%04X" % (0xD800 + (cp >> 10), 0xDC00 + (cp & 0x3FF))
        elif r < 0.92:
            t = rng.choice(["é", "€", "😀", "ü", "漢", "߿", "￿"])
        elif r < 0.96:
            t = "\\\\" * rng.randint(1, 40)
        elif r < 0.96 + p_bad:
            t = rng.choice(BAD)
        else:
            t = "x"
        out.append(t)
        n += len(t)
    return '"' + "".join(out) + '"'


def document(rng, size=None):
    if size is None:
        size = rng.choice([100, 1000, 4000, 4200, 16000, 17000, 70000, 300000])
    p_bad = rng.choice([0, 0, 0, 0.02])
    parts, n = [], 2
    while n < size:
        r = rng.random()
        if r < 0.6:
            v = string(rng, rng.choice([0, 1, 3, 8, 20, 60, 64, 200, 4096, 5000]) if rng.random() < 0.9 else rng.randint(0, 20000), p_bad)
        elif r < 0.8:
            v = rng.choice(["1", "-12", "3.25", "1e5", "true", "false", "null", "12345678901234567", "0.1", "-0"])
        elif r < 0.9:
            v = "[" + ",".join(string(rng, rng.randint(0, 30), p_bad) for _ in range(rng.randint(0, 5))) + "]"
        else:
            v = "{" + ",".join(string(rng, rng.randint(1, 10), 0) + ":" + string(rng, rng.randint(0, 30), p_bad) for _ in range(rng.randint(0, 4))) + "}"
        parts.append(" " * rng.choice([0, 0, 0, 1, 7]) + v)
        n += len(parts[-1].encode()) + 1
    return ("[" + ",".join(parts) + "]").encode()


