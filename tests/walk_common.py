"""Shared by the CPU and GPU tests of the GPU walker (csrc/walk_doc.h / walk.hip)."""
NEEDS_HOST = -1


def exact_range(lit):
    """Independent statement of what the device converts itself (csrc/sj_number.h: Clinger's exact range + Eisel-Lemire):
    every literal whose significand, zeros at either end stripped, has at most 19 digits.  False: handed back (the
    reference's slow path, DoubleParser.java:205-330)."""
    s = lit.lstrip("-").lower()
    mant, _, e = s.partition("e")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0").rstrip("0")
    return len(digits) <= 19


def random_number_literal(rng):
    k = rng.random()
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
        elif any(v is None for v in verdicts):
            either.add(k)
    return docs, hard, either
