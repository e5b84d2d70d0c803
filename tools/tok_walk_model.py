#!/usr/bin/env python
"""Lane-level model of round 5's batch walker (k_tok_walk: one document per wave at a time; replaced in round 6 by the stream form,
k_tok_stream -- tools/tok_stream_model.py -- whose TOKEN STEP is this one's: the same scans, level words, stack, counters and
grammar table; this model stays as the executable description of that step and as the base tok_stream_model.py imports): one
"wave" of 64 lanes walks one document exactly the way that kernel did -- chunks of 64 structurals ingested into a ring of 256 tokens (separators folded
into the token behind them by shifted lane masks), token steps of up to 64 tokens with the trim in front of an opening bracket
whose successor is not at hand, neighbours by wave shifts with the previous step's last token carried in, depth / tape position
from one two-field scan, containers from a word of lanes per level + a stack by level + comma counters (the LDS arrays of the
kernel, atomics and all), the grammar from the SAME table the kernel reads (csrc/sj_tokens.h through tests/host_sim/tok_sim.cpp),
literals queued and parsed behind the steps.  Lane masks are Python ints of 64 bits.  -> (kept, tape): kept = False is "the exact
walker takes this document".  Design validation (tests/test_tok_walk_model.py checks it against the oracle); nothing ships from
here."""
import struct

from coop_walk_model import E, SOW, parse_number  # the literal parsers of the cooperative walker's model

M64 = (1 << 64) - 1
TK_OPEN_A, TK_OPEN_O, TK_CLOSE_A, TK_CLOSE_O, TK_STRING, TK_NONE, TK_ATOM, TK_NUMBER = range(8)
TOK_COMMA, TOK_COLON, TOK_SCAN_FIELDS, TOK_RING, LEVELS = 8, 16, 0x00600060, 256, 64


def first(n):
    return M64 >> (64 - n)


def lanes(mask):
    return [(mask >> i) & 1 for i in range(64)]


def ballot(bits):
    m = 0
    for i, b in enumerate(bits):
        if b:
            m |= 1 << i
    return m


def below(mask, lane):
    return bin(mask & ((1 << lane) - 1)).count("1")


def atom(buf, p):
    for word, t in ((b"true", ord("t")), (b"false", ord("f")), (b"null", ord("n"))):
        if bytes(buf[p:p + len(word)]) == word and buf[p + len(word)] in SOW:
            return t
    return None


def walk(tables, buf, structurals, record_offsets, string_base=0, max_depth=1024):
    """buf: the document + padding; structurals: stage 1's positions; record_offsets[k]: the k-th string's record in the string buffer"""
    tok_of_first_byte, grammar = tables
    n = len(structurals)
    if n == 0 or n >= 1 << 30:
        return False, None
    nchunks = (n + 63) // 64
    depth_limit = min(max_depth, LEVELS) - 1
    ring = [(0, TK_NONE)] * TOK_RING
    open_words, stk, cnt = [0] * 64, [(0, 0)] * 64, [0] * 64
    tape, queue = {}, []
    H0, T0, S0 = 0, 1, 0
    c_token, c_empty_open, root_closed = TK_NONE, 0, False
    c = head = tail = 0
    SEPp = COLp = sep_twice = 0

    def ingest():
        nonlocal c, tail, SEPp, COLp, sep_twice
        nvl = min(n - c * 64, 64)
        VL = first(nvl)
        pos = [structurals[min(c * 64 + l, n - 1)] for l in range(64)]
        b0 = [buf[p] for p in pos]
        COL = ballot(b == 0x3A for b in b0) & VL
        SEP = (ballot(b == 0x2C for b in b0) & VL) | COL
        S1 = ((SEP << 1) | (SEPp >> 63)) & M64
        C1 = ((COL << 1) | (COLp >> 63)) & M64
        sep_twice |= SEP & S1
        TOK = VL & ~SEP & M64
        for l in range(64):
            if (TOK >> l) & 1:
                pre = TOK_COLON if (C1 >> l) & 1 else (TOK_COMMA if (S1 >> l) & 1 else 0)
                ring[(tail + below(TOK, l)) % TOK_RING] = (pos[l], tok_of_first_byte(b0[l]) | pre)
        tail += bin(TOK).count("1")
        SEPp, COLp = SEP, COL
        c += 1
        if c == nchunks:
            sep_twice |= SEP >> (nvl - 1)

    for _ in range(min(nchunks, 4)):
        ingest()
    ok = True
    while ok:
        while c < nchunks and tail - head <= TOK_RING - 64:
            ingest()
        avail = tail - head
        if sep_twice:
            ok = False
        if avail == 0 or not ok:
            break
        if root_closed:
            ok = False
            break
        na = min(avail, 64)
        re = [ring[(head + l) % TOK_RING] for l in range(64)]
        more = c < nchunks or avail > 64
        nv = 63 if (more and (re[63][1] & 7) <= TK_OPEN_O) else na
        V = first(nv)
        p = [r[0] for r in re]
        token = [re[l][1] if (V >> l) & 1 else TK_NONE for l in range(64)]
        tk = [t & 7 for t in token]
        OPEN = ballot(t <= TK_OPEN_O for t in tk)
        CLOSE = ballot(t <= TK_CLOSE_O for t in tk) & ~OPEN & M64
        Q = ballot(t == TK_STRING for t in tk)
        PRIM = ballot(t >= TK_ATOM for t in tk)
        prev = [c_token] + token[:63]                 # wave_shr:1, the previous step's last token in lane 0
        nxt = token[1:] + [TK_NONE]                   # wave_shl:1
        EO = ballot(((nxt[l] ^ (tk[l] + 2)) & (7 | TOK_COMMA | TOK_COLON)) == 0 for l in range(64)) & OPEN
        EC = CLOSE & (((EO << 1) | c_empty_open) & M64)
        inc = [(t & TOK_SCAN_FIELDS) >> 5 for t in token]
        scan, run = [], 0
        for v in inc:
            run += v
            scan.append(run)
        tot = scan[63]
        excl = [scan[l] - inc[l] for l in range(64)]
        h = [H0 + (excl[l] & 0xFFFF) - l for l in range(64)]
        tpos = [T0 + (excl[l] >> 16) for l in range(64)]
        sord = [S0 + below(Q, l) for l in range(64)]
        DEEP = ballot(x >= depth_limit for x in h)
        ROOT_END = ballot(x == 1 for x in h) & CLOSE
        # containers
        lvl = [(x - 1) & 63 for x in h]
        for l in range(64):
            open_words[l] = 0
            cnt[l] = 0
        for l in range(64):
            if (OPEN >> l) & 1:
                open_words[h[l] & 63] |= 1 << l
        par_lane, in_step = [0] * 64, [False] * 64
        for l in range(64):
            om = open_words[lvl[l]] & ((1 << l) - 1)
            in_step[l] = om != 0
            par_lane[l] = om.bit_length() - 1 if om else 0
        IN_STEP = ballot(in_step)
        closer = CLOSE & ~EC & IN_STEP & M64
        for l in range(64):                            # one atomic add per lane (most add 0)
            add = ((token[l] >> 3) & 1) | (0x80000000 if (closer >> l) & 1 else 0)
            if in_step[l]:
                cnt[par_lane[l]] = (cnt[par_lane[l]] + add) & 0xFFFFFFFF
            else:
                stk[lvl[l]] = (stk[lvl[l]][0], (stk[lvl[l]][1] + add) & 0xFFFFFFFF)
        se_x = [stk[lvl[l]][0] for l in range(64)]
        pcnt = [cnt[par_lane[l]] if in_step[l] else stk[lvl[l]][1] for l in range(64)]
        own = list(cnt)
        opener = [tpos[l] | (0x80000000 if tk[l] == TK_OPEN_A else 0) for l in range(64)]
        for l in range(64):
            if (OPEN >> l) & 1 and not (EO >> l) & 1 and not own[l] & 0x80000000:
                stk[h[l] & 63] = (opener[l], own[l])
        par = [opener[par_lane[l]] if in_step[l] else se_x[l] for l in range(64)]
        par_tpos = [x & 0x7FFFFFFF for x in par]
        par_cnt = [x & 0x3FFFFFFF for x in pcnt]
        gi = [(token[l] & 0x1F) | ((prev[l] & 0x17) << 5) | ((par[l] >> 21) & 0x400) for l in range(64)]
        BAD = ballot(grammar(g) == 0 for g in gi) | (OPEN & ~EO & DEEP & M64)
        if ROOT_END:
            low = ROOT_END & -ROOT_END
            BAD |= V & ~((low << 1) - 1) & M64
        if BAD:
            ok = False
            break
        if ROOT_END:
            root_closed = True
        for l in range(64):
            if (Q >> l) & 1:
                tape[tpos[l]] = (ord('"') << 56) | (string_base + record_offsets[sord[l]])
            if (PRIM >> l) & 1:
                queue.append((p[l], tpos[l]))
            ch = (token[l] >> 8) & 0xFF
            if ((EO | CLOSE) >> l) & 1:
                pay = tpos[l] + 2 if (EO >> l) & 1 else (tpos[l] if (EC >> l) & 1 else par_tpos[l])
                tape[tpos[l]] = (ch << 56) | pay
            if ((CLOSE & ~EC) >> l) & 1:
                tape[par_tpos[l]] = ((ch - 2) << 56) | (min(par_cnt[l] + 1, 0xFFFFFF) << 32) | (tpos[l] + 1)
        H0 += (tot & 0xFFFF) - nv
        T0 += tot >> 16
        S0 += bin(Q).count("1")
        c_token = token[nv - 1]
        c_empty_open = (EO >> (nv - 1)) & 1
        head += nv
    if ok and not root_closed:
        ok = False
    if not ok:
        return False, None
    for p, t in queue:                                  # the dense literal parser behind the steps
        a = atom(buf, p)
        if a is not None:
            tape[t] = a << 56
            continue
        if not (buf[p] == 0x2D or 0x30 <= buf[p] <= 0x39):
            return False, None
        r = parse_number(buf, p, len(buf))
        if r[0] == "err":
            return False, None
        tape[t] = ord(r[0]) << 56
        tape[t + 1] = r[1]
    tlen = T0 + 1
    tape[T0] = ord("r") << 56
    tape[0] = (ord("r") << 56) | tlen
    return True, [tape[i] for i in range(tlen)]
