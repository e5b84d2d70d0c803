"""Static instruction statistics of one kernel from hipcc's --save-temps assembly: per basic block VALU / SALU / memory counts,
lane read/write (SGPR spill traffic) and the source lines that contribute most (compile with -gline-tables-only for those).
usage: python tools/isa_blocks.py <file.s> <kernel name substring> [--lines]"""
import collections
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    show_lines = "--lines" in sys.argv
    cur, blocks, b, line = None, [], None, None
    per_line = collections.defaultdict(collections.Counter)
    for l in open(path).read().split("\n"):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            cur = m.group(1)
            if want in cur:
                b = {"name": "entry", "V": 0, "S": 0, "M": 0, "wl": 0, "lines": collections.Counter(), "br": []}
                blocks.append(b)
            continue
        if cur is None or want not in cur:
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            b = {"name": m.group(1), "V": 0, "S": 0, "M": 0, "wl": 0, "lines": collections.Counter(), "br": []}
            blocks.append(b)
            continue
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            line = (int(m.group(1)), int(m.group(2)))
            continue
        if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
            op = l.split()[0]
            k = "V" if op.startswith("v_") else "S" if op.startswith("s_") else "M"
            b[k] += 1
            if k == "V":
                # tools/microbench/valu_rates.hip: the fast class of gfx950 (2.1 cycles per wave64 instruction) -- plain 32-bit
                # logic / add / sub / mov, shifts by a constant, v_bitop3_b32 -- with VGPR / literal / inline operands only
                base = op.replace("_e32", "").replace("_e64", "")
                ops = [x.strip(",") for x in l.split()[1:]]
                fast_op = base in ("v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_bitop3_b32") or \
                    (base in ("v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32") and len(ops) > 1 and re.match(r"^-?\d+$|^0x", ops[1]) is not None)
                sgpr_src = any(re.match(r"^(s\d+|s\[|vcc|exec)", x) for x in ops[1:])
                b["fast"] = b.get("fast", 0) + (1 if fast_op and not sgpr_src else 0)
            if op in ("v_writelane_b32", "v_readlane_b32"):
                b["wl"] += 1
            b["lines"][line] += 1
            per_line[line][k] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                b["br"].append(op.replace("s_cbranch_", "").replace("s_branch", "jmp") + "->" + l.split()[-1])
    tot = collections.Counter()
    for b in blocks:
        top = ", ".join("%d:%d" % (k[1], v) for k, v in b["lines"].most_common(4) if k)
        print("%-11s V%4d (fast %4d) S%4d M%3d rl/wl%3d | %s | %s" % (b["name"], b["V"], b.get("fast", 0), b["S"], b["M"], b["wl"], top, " ".join(b["br"])))
        tot["fast"] += b.get("fast", 0)
        for k in "VSM":
            tot[k] += b[k]
        tot["wl"] += b["wl"]
    print("total", dict(tot))
    if show_lines:
        for (f, ln), c in sorted(per_line.items(), key=lambda x: (x[0] or (0, 0))):
            print(f, ln, c["V"], c["S"], c["M"])


if __name__ == "__main__":
    main()
