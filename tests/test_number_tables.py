"""The 5^q table of the Eisel-Lemire conversion (simdjson-java_amd/csrc/sj_pow5_table.h) is GENERATED
(tools/gen_pow5_table.py, the published construction + the reference's one quirk) and was compared, in the build
container, with the reference's NumberParserTables.POWERS_OF_FIVE (`--check-reference`: identical, 651 entries).
This test pins the generator's output and the committed header to the digest of that comparison."""
import os
import re
import sys

from tests.conftest import ROOT

PINNED_SHA256 = "73146b610549bd2e37ce0c2d4c4f751da7395fc0e70c0ba0063ceab6bb3ee89d"


def test_generated_table_is_the_pinned_one():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_pow5_table as G
    tab = G.entries()
    assert len(tab) == 651 and G.digest(tab) == PINNED_SHA256
    text = open(os.path.join(ROOT, "simdjson-java_amd", "csrc", "sj_pow5_table.h")).read()
    vals = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{16})ull", text)]
    assert [(vals[2 * i] << 64) | vals[2 * i + 1] for i in range(651)] == tab
    # spot values anyone can check by hand: 5^0 and 5^1 normalised to bit 127, 5^-1 = ceil(2^130 / 5)
    assert tab[342] == 1 << 127 and tab[343] == 5 << 125 and tab[341] == (1 << 130) // 5 + 1
