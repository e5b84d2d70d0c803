"""Deterministic synthetic JSON generators for BASELINE.json configs[2] and [3].

synth_tile(seed, target_bytes): one self-contained JSON array of records terminated by '\\n':
  ~50 % of the bytes inside string literals, ~10 % of string characters are escape sequences
  (simple escapes 70 %, \\uXXXX BMP 25 %, surrogate pairs 5 %), ~10 % of string characters non-ASCII
  (2/3/4-byte UTF-8 in ratio 6:3:1); the rest = keys' punctuation, integers, floats, true/false/null,
  2-space indentation.  The 4 GiB input of configs[2] is this tile repeated (closed-form parity check).
small_docs(seed, n): n flat-ish records of 768..1280 bytes (configs[3]), returned as a list of bytes.
"""
import random

_SIMPLE = ['\\"', "\\\\", "\\/", "\\b", "\\f", "\\n", "\\r", "\\t"]
_ASCII = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 _-.,:;!?()[]{}@#$%&*+=<>|~'"


def _string(rng, n_chars, esc=0.10, nonascii=0.10):
    out = []
    for _ in range(n_chars):
        r = rng.random()
        if r < esc:
            k = rng.random()
            if k < 0.70:
                out.append(rng.choice(_SIMPLE))
            elif k < 0.95:
                cp = rng.randrange(0x20, 0xD800) if rng.random() < 0.8 else rng.randrange(0xE000, 0x10000)
                out.append("\\u%04x" % cp)
            else:
                v = rng.randrange(0x10000, 0x110000) - 0x10000
                out.append("\\u%04X\\u%04X" % (0xD800 + (v >> 10), 0xDC00 + (v & 0x3FF)))
        elif r < esc + nonascii:
            k = rng.random()
            if k < 0.6:
                out.append(chr(rng.randrange(0x80, 0x800)))
            elif k < 0.9:
                cp = rng.randrange(0x800, 0x10000)
                out.append(chr(cp if not 0xD800 <= cp <= 0xDFFF else 0x4E2D))
            else:
                out.append(chr(rng.randrange(0x10000, 0x110000)))
        else:
            out.append(rng.choice(_ASCII))
    return '"' + "".join(out) + '"'


def _record(rng, indent="  "):
    fields = []
    for i in range(rng.randint(4, 9)):
        k = rng.random()
        if k < 0.55:
            v = _string(rng, rng.randint(4, 120))
        elif k < 0.75:
            v = str(rng.randrange(-10**9, 10**12))
        elif k < 0.85:
            v = repr(rng.uniform(-1e6, 1e6))
        elif k < 0.95:
            v = rng.choice(["true", "false", "null"])
        else:
            v = "[" + ", ".join(str(rng.randrange(1000)) for _ in range(rng.randint(0, 8))) + "]"
        fields.append('%s%s"f%d": %s' % (indent, indent, i, v))
    return indent + "{\n" + ",\n".join(fields) + "\n" + indent + "}"


def synth_tile(seed=20250824, target_bytes=4 << 20):
    rng = random.Random(seed)
    recs, size = [], 2
    while size < target_bytes:
        r = _record(rng)
        recs.append(r)
        size += len(r.encode("utf-8")) + 2
    return ("[\n" + ",\n".join(recs) + "\n]\n").encode("utf-8")


def small_docs(seed=20250825, n=1000, lo=768, hi=1280, same_schema=False):
    """same_schema: every document has the same sequence of field types (records of one log / table, the usual NDJSON
    case) instead of a random type per field."""
    rng = random.Random(seed)
    schema_rng = random.Random(seed + 1)
    kinds = [schema_rng.random() for _ in range(256)]
    docs = []
    for _ in range(n):
        target = rng.randint(lo, hi)
        fields = []
        size = 2
        i = 0
        while size < target - 60:
            k = kinds[i % 256] if same_schema else rng.random()
            if k < 0.40:
                v = _string(rng, rng.randint(8, 60), esc=0.05, nonascii=0.05)
            elif k < 0.70:
                v = str(rng.randrange(-10**6, 10**9))
            elif k < 0.80:
                v = rng.choice(["true", "false", "null"])
            elif k < 0.90:
                v = "[" + ",".join(str(rng.randrange(100)) for _ in range(rng.randint(0, 8))) + "]"
            else:
                v = '{"x":%d,"y":%s}' % (rng.randrange(100), _string(rng, 6, 0, 0))
            f = '"k%d":%s' % (i, v)
            fields.append(f)
            size += len(f.encode("utf-8")) + 1
            i += 1
        docs.append(("{" + ",".join(fields) + "}").encode("utf-8"))
    return docs
