"""Shared by the CPU and GPU tests of the GPU walker (csrc/walk_doc.h / walk.hip)."""
NEEDS_HOST = -1


def exact_range(lit):
    """Independent statement of what the device converts itself (csrc/sj_number.h): every literal of at most 19 significant
    digits (Clinger's exact range + Eisel-Lemire), and every longer one whose two 19-digit neighbours w * 10^q and
    (w + 1) * 10^q are the same double (Python's float() is correctly rounded).  False: handed back (the reference's slow
    path, DoubleParser.java:205-330)."""
    s = lit.lstrip("-").lower()
    mant, _, e = s.partition("e")
    ip, _, fp = mant.partition(".")
    digits = ip + fp
    stripped = digits.lstrip("0")
    if len(stripped.rstrip("0")) <= 19:
        return True
    lead = len(digits) - len(stripped)
    exp = max(-10 ** 6, min(10 ** 6, int(e) if e else 0))
    w, q = int(stripped[:19]), exp + len(ip) - lead - 19
    return float("%de%d" % (w, q)) == float("%de%d" % (w + 1, q))


# more than 19 significant digits AND closer than 10^-19 (relative) to the midpoint of two doubles: neither 19-digit
# neighbour decides the rounding
AMBIGUOUS = ["9007199254740993.00000000000000000001", "9007199254740994.99999999999999999999", "1.00000000000000011102230246251565404236316680908203125",
             "1.00000000000000011102230246251565404236316680908203124", "2.4703282292062327208051355972e-324",
             "0.500000000000000166533453693773481063544750213623046875",
             str(((1 << 54) - 1) << 970) + ".0"]  # the midpoint of the largest double and 2^1024


def random_number_literal(rng):
    k = rng.random()
    if k > 0.97:  # around a rounding boundary: an exact midpoint of two doubles, nudged (or not) far behind the 19th digit
        from decimal import Decimal, getcontext
        import struct
        getcontext().prec = 1200
        bits = rng.getrandbits(52) | (rng.choice([1, 500, 1022, 1023, 1024, 1075, 1500, 2045]) << 52)
        lo = Decimal(struct.unpack("<d", struct.pack("<Q", bits))[0])
        hi = Decimal(struct.unpack("<d", struct.pack("<Q", bits + 1))[0])
        mid = (lo + hi) / 2
        text = format(mid, "f")
        if "." not in text:
            text += ".0"
        tweak = rng.choice(["", "1", "000000000001", "9"])
        if tweak == "9":  # just below: decrement the last digit, append nines
            text = text.rstrip("0")
            text = text[:-1] + str(int(text[-1]) - 1) + "9" * 12 if text[-1] not in ".0" else text + "1"
        else:
            text += tweak
        return ("-" if rng.random() < 0.3 else "") + text
    sign = "-" if rng.random() < 0.3 else ""
    if k < 0.25:
        return sign + str(rng.randrange(10 ** rng.randint(1, 18)))
    ip = str(rng.randrange(10 ** rng.randint(1, rng.choice([1, 3, 8, 16, 21, 30])))) if rng.random() < 0.8 else "0"
    fp = ""
    if rng.random() < 0.8:
        fp = "." + "".join(rng.choice("0000123456789") for _ in range(rng.randint(1, rng.choice([1, 2, 6, 12, 20, 40]))))
    ex = ""
    if rng.random() < 0.4 or not fp:
        ex = rng.choice("eE") + rng.choice(["", "+", "-"]) + str(rng.randrange(rng.choice([3, 10, 25, 40, 330, 400])))
    return sign + ip + fp + ex


def number_documents(rng, n):
    """-> (documents, set of indexes that must be handed back, set of indexes where either answer is right)"""
    docs, hard, either = [], set(), set()
    for k in range(n):
        lits = [random_number_literal(rng) for _ in range(rng.randint(1, 6))]
        docs.append(("[" + ", ".join(lits) + "]").encode())
        verdicts = [exact_range(x) for x in lits if any(c in x for c in ".eE")]
        if any(v is False for v in verdicts):
            hard.add(k)
    return docs, hard, either
