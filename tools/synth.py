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


def synth_tile_spec(seed=20250824, target_bytes=16 << 20):
    """The configs[2] tile as SURVEY.md 8(d) specifies it: one JSON array of records terminated by '\\n', 50 % of the BYTES
    inside string literals (keys and values; the generator steers the field kinds towards that fraction), 10 % of the
    string CHARACTERS escape sequences (simple 70 %, \\uXXXX BMP 25 %, surrogate pairs 5 %), 10 % non-ASCII (2 / 3 / 4-byte
    UTF-8 6:3:1), the rest integers, floats, true / false / null, keys' punctuation and 2-space indentation.
    numpy's PCG64 instead of the survey's xorshift64* (deterministic for a seed either way); string characters are drawn
    from pools of pre-built pieces so that 16 MiB take seconds, not minutes."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    pr = random.Random(seed)
    simple = [p.encode() for p in _SIMPLE]
    bmp = [("\\u%04x" % (pr.randrange(0x20, 0xD800) if pr.random() < 0.8 else pr.randrange(0xE000, 0x10000))).encode() for _ in range(2048)]
    pairs = []
    for _ in range(512):
        v = pr.randrange(0x10000, 0x110000) - 0x10000
        pairs.append(("\\u%04X\\u%04X" % (0xD800 + (v >> 10), 0xDC00 + (v & 0x3FF))).encode())
    u2 = [chr(pr.randrange(0x80, 0x800)).encode() for _ in range(1024)]
    u3 = [chr(c if not 0xD800 <= c <= 0xDFFF else 0x4E2D).encode() for c in (pr.randrange(0x800, 0x10000) for _ in range(1024))]
    u4 = [chr(pr.randrange(0x10000, 0x110000)).encode() for _ in range(512)]
    ascii_ = [c.encode() for c in _ASCII]
    n_chars = int(target_bytes * 0.55)
    cls = rng.random(n_chars)
    sub = rng.random(n_chars)
    pick = rng.integers(0, 1 << 30, n_chars)
    pieces = np.empty(n_chars, dtype=object)

    def fill(mask, pool):
        idx = np.nonzero(mask)[0]
        pl = np.array(pool, dtype=object)
        pieces[idx] = pl[pick[idx] % len(pool)]

    esc, na = cls < 0.10, (cls >= 0.10) & (cls < 0.20)
    fill(esc & (sub < 0.70), simple)
    fill(esc & (sub >= 0.70) & (sub < 0.95), bmp)
    fill(esc & (sub >= 0.95), pairs)
    fill(na & (sub < 0.6), u2)
    fill(na & (sub >= 0.6) & (sub < 0.9), u3)
    fill(na & (sub >= 0.9), u4)
    fill(cls >= 0.20, ascii_)
    out = [b"[\n"]
    size, in_str, pos, rec = 2, 0, 0, 0
    while size < target_bytes and pos + 200 < n_chars:
        fields = []
        nf = pr.randint(4, 9)
        for i in range(nf):
            key = b'    "f%d": ' % i
            want_string = in_str < 0.5 * (size + 40)
            if want_string:
                n = pr.randint(4, 120)
                body = b"".join(pieces[pos:pos + n])
                pos += n
                v = b'"' + body + b'"'
                in_str += len(body) + 1
            else:
                k = pr.random()
                if k < 0.45:
                    v = str(pr.randrange(-10**9, 10**12)).encode()
                elif k < 0.70:
                    v = repr(pr.uniform(-1e6, 1e6)).encode()
                elif k < 0.88:
                    v = pr.choice([b"true", b"false", b"null"])
                else:
                    v = b"[" + b", ".join(str(pr.randrange(1000)).encode() for _ in range(pr.randint(0, 8))) + b"]"
            in_str += len(key) - 4 - 3  # the key's opening quote and its characters ("fN)
            f = key + v + (b",\n" if i + 1 < nf else b"\n")
            fields.append(f)
            size += len(f)
        r = b"  {\n" + b"".join(fields) + b"  }"
        out.append(r if rec == 0 else b",\n" + r)
        size += 8
        rec += 1
    out.append(b"\n]\n")
    return b"".join(out)


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
