#!/usr/bin/env python
"""How often could k_strings skip its bit-plane transposition if stage 1 handed over its quote / in-string masks?
The planes are needed for everything escape-related (which escaped characters are n / t / r / b / f / u, hex digits, the \\uXXXX
look-back), and a wave = one 4 KiB granule runs ONE path for its 64 lanes: the transposition can only be skipped for a granule that
holds no backslash at all.  Counted on the bytes of the two workloads (no GPU needed)."""
import gzip, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tools"))
import workloads as W


def stats(name, b):
    g, k = len(b) // 4096, len(b) // 64
    fg = sum(1 for i in range(g) if b"\\" not in b[i * 4096:(i + 1) * 4096])
    fk = sum(1 for i in range(k) if b"\\" not in b[i * 64:(i + 1) * 64])
    print("%-28s 4 KiB granules without a backslash: %5d of %6d = %5.1f %%   64-byte blocks without: %5.1f %%" % (name, fg, g, 100.0 * fg / g, 100.0 * fk / k))


stats("twitter.json", gzip.open(os.path.join(R, "tests", "golden", "data", "twitter.json.gz")).read())
stats("configs[3], 20,000 documents", bytes(W.unique_docs(0, 20000)[0]))
