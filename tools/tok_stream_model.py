#!/usr/bin/env python
"""Model of the STREAM form of the batch walker, written at the end of round 5 as the design for round 6 and built then:
simdjson-java_amd/csrc/coop_walk.hip k_tok_stream.  The kernel differs from this model in how it gets a token's
document-relative depth / ordinal / tape position -- three run-wide counters and per-document bases in LDS instead of a max-scan
and a shuffle per step -- and in ruling out "something follows the root's end" by ONE compare (only a document's first token may
stand at depth 0) instead of interval masks; the rest is this.  One "wave"
walks a RUN of consecutive documents as one token stream -- the structurals of the run are contiguous in stage 1's index array, so
the ingest never stops at a document, and a token step of 64 tokens may hold the end of one document, a whole small one and the
beginning of a third.  What round 5 measured made this the thing to try: the per-document scaffolding of k_tok_walk (four
ingest trips, prefetch arrays, prologue, epilogue, a second half-empty step) is 56 % of its scalar instructions.

What the model pins down (tests/test_tok_stream_model.py checks it against the oracle, document by document, on runs that mix valid
and broken documents):
  * a token carries its document's number in the run; the separator folding and the "what stands in front of me" carry do not
    cross a document start; a token whose predecessor belongs to another document is a document's first (DOCSTART);
  * depth, tape position and string ordinal are ONE three-field scan over the step, made document-relative by subtracting the scan
    at the segment's first lane (a max-scan of "lane if DOCSTART" gives that lane; one shuffle fetches the three bases); the first
    segment of a step continues the previous step's document (carried bases);
  * containers: the per-level words and the stack of k_tok_walk unchanged -- a document's containers all lie behind its DOCSTART,
    so the last opening bracket of my level in front of me is mine whenever my document is well formed so far;
  * the grammar table unchanged, the previous token of a DOCSTART token is "nothing";
  * nothing may follow a root's end inside its document (interval masks by a borrow), a document must end on its root's end
    (DOCSTART lanes look at the token in front of them, the run's end at the last token), a document with no token at all fails;
  * a failing document is isolated: it is listed for the exact walker, a bit per document of the run keeps its later tokens from
    storing anything, and the next DOCSTART re-bases everything, whatever state the broken document left behind;
  * both root words of a document are written at its DOCSTART from its predicted length (exact for a well-formed document).
-> per document: kept (bool), tape (list) -- tape payloads of strings are string_base + record_offsets[ordinal]."""
from coop_walk_model import parse_number
from tok_walk_model import (M64, TK_OPEN_A, TK_OPEN_O, TK_CLOSE_O, TK_STRING, TK_NONE, TK_ATOM, TOK_COMMA, TOK_COLON, TOK_SCAN_FIELDS,
                            LEVELS, atom, ballot, below, first)

RING = 512


def walk_run(tables, buf, structurals, docs, record_offsets, string_base=0, max_depth=1024):
    """docs: list of dicts {from, to, dso, toff, room} -- structural index range, ordinal of its first string, tape word offset and
    predicted length; consecutive (docs[j].to == docs[j + 1].from).  -> [(kept, tape words or None)] per document"""
    tok_of_first_byte, grammar = tables
    R = len(docs)
    assert R <= 64
    I0, I1 = docs[0]["from"], docs[-1]["to"]
    n = I1 - I0
    depth_limit = min(max_depth, LEVELS) - 1
    failed = 0                                  # bit j: document j of the run goes to the exact walker
    for j, d in enumerate(docs):
        if d["to"] == d["from"]:
            failed |= 1 << j                    # (no structural at all: nothing of it ever reaches a step)
    tape = {}
    if n == 0:
        return [(False, None)] * R
    nchunks = (n + 63) // 64
    ring = [(0, TK_NONE, 0)] * RING             # (position, token, document)
    open_words, stk, cnt = [0] * 64, [(0, 0)] * 64, [0] * 64
    queue = []
    c = head = tail = 0
    SEPp = COLp = 0
    doc_of_last = 0                             # document of the previous chunk's last structural
    sep_last_doc = None
    # carried across steps: bases of the continuing document, the previous step's last token
    Hb = Wb = Qb = 0                            # scan values at the continuing document's start (relative to this step: negative growth)
    c_token, c_empty_open, c_root_end, c_doc = TK_NONE, 0, 1, 0
    c_seen = False                              # has any token been walked yet

    def ingest():
        nonlocal c, tail, SEPp, COLp, doc_of_last, failed
        i0 = I0 + c * 64
        nvl = min(I1 - i0, 64)
        VL = first(nvl)
        pos = [structurals[min(i0 + l, I1 - 1)] for l in range(64)]
        b0 = [buf[p] for p in pos]
        # document starts of this chunk (the kernel: the lanes that hold the run's `from` values write their number into an LDS byte
        # per structural of the chunk; a max-scan hands every structural its document)
        DS, start_doc = 0, {}
        for j, d in enumerate(docs):
            if d["to"] > d["from"] and i0 <= d["from"] < i0 + 64:
                DS |= 1 << (d["from"] - i0)
                start_doc[d["from"] - i0] = j
        doc = []
        cur = doc_of_last
        for l in range(64):
            cur = start_doc.get(l, cur)
            doc.append(cur)
        doc_of_last = doc[nvl - 1]
        COL = ballot(b == 0x3A for b in b0) & VL
        SEP = (ballot(b == 0x2C for b in b0) & VL) | COL
        S1 = ((SEP << 1) | (SEPp >> 63)) & M64 & ~DS      # nothing stands in front of a document's first structural
        C1 = ((COL << 1) | (COLp >> 63)) & M64 & ~DS
        TOK = VL & ~SEP & M64
        twice = SEP & S1
        # a separator behind a document's last structural: the structural in front of a document start (or the run's last one)
        ends = ((DS >> 1) | (1 << (nvl - 1) if c == nchunks - 1 else 0)) & VL
        bad_sep = twice | (SEP & ends)
        if c > 0 and (DS & 1) and (SEPp >> 63):
            failed |= 1 << prev_chunk_last_doc[0]         # the previous chunk ended a document on a separator
        for l in range(64):
            if (bad_sep >> l) & 1:
                failed |= 1 << doc[l]
        for l in range(64):
            if (TOK >> l) & 1:
                pre = TOK_COLON if (C1 >> l) & 1 else (TOK_COMMA if (S1 >> l) & 1 else 0)
                ring[(tail + below(TOK, l)) % RING] = (pos[l], tok_of_first_byte(b0[l]) | pre, doc[l])
        tail += bin(TOK).count("1")
        SEPp, COLp = SEP, COL
        prev_chunk_last_doc[0] = doc[nvl - 1]
        c += 1

    prev_chunk_last_doc = [0]
    while True:
        while c < nchunks and tail - head <= RING - 64:
            ingest()
        avail = tail - head
        if avail == 0:
            break
        na = min(avail, 64)
        re = [ring[(head + l) % RING] for l in range(64)]
        more = c < nchunks or avail > 64
        nv = 63 if (more and (re[63][1] & 7) <= TK_OPEN_O) else na
        V = first(nv)
        p = [r[0] for r in re]
        token = [re[l][1] if (V >> l) & 1 else TK_NONE for l in range(64)]
        docj = [re[l][2] for l in range(64)]
        # a document starts where the token in front belongs to another one (a document without any token never shows up here)
        DS = ballot((docj[l] != (docj[l - 1] if l else c_doc) or (l == 0 and not c_seen)) and (V >> l) & 1 for l in range(64))
        tk = [t & 7 for t in token]
        OPEN = ballot(t <= TK_OPEN_O for t in tk)
        CLOSE = ballot(t <= TK_CLOSE_O for t in tk) & ~OPEN & M64
        Q = ballot(t == TK_STRING for t in tk)
        PRIM = ballot(t >= TK_ATOM for t in tk)
        prev = [c_token] + token[:63]
        prev = [TK_NONE if (DS >> l) & 1 else prev[l] for l in range(64)]      # nothing in front of a document's first token
        nxt = token[1:] + [TK_NONE]
        nxt = [TK_NONE if (DS >> (l + 1)) & 1 else nxt[l] for l in range(64)]  # ... and its predecessor has no successor
        EO = ballot(((nxt[l] ^ (tk[l] + 2)) & (7 | TOK_COMMA | TOK_COLON)) == 0 for l in range(64)) & OPEN
        EC = CLOSE & (((EO << 1) | (c_empty_open & ~DS & 1)) & M64)
        # one scan, three fields: 1 + up - down | strings | words
        inc = [((t & TOK_SCAN_FIELDS) >> 5) for t in token]
        incd = [(inc[l] & 0xFFFF) | ((1 if (Q >> l) & 1 else 0) << 8) | ((inc[l] >> 16) << 16) for l in range(64)]
        scan, run = [], 0
        for v in incd:
            run += v
            scan.append(run)
        excl = [scan[l] - incd[l] for l in range(64)]
        # the segment's first lane (max-scan of "lane if DOCSTART"), its three bases by one shuffle
        src, cur = [], -1
        for l in range(64):
            if (DS >> l) & 1:
                cur = l
            src.append(cur)
        dep = [(excl[l] & 0xFF) - l for l in range(64)]                         # running depth in front of the lane, step-relative
        h, tpos, sord = [], [], []
        for l in range(64):
            if src[l] < 0:                                                        # the continuing document
                h.append(Hb + dep[l])
                tpos.append(Wb + (excl[l] >> 16))
                sord.append(Qb + ((excl[l] >> 8) & 0xFF))
            else:
                s_ = src[l]
                h.append(dep[l] - dep[s_])
                tpos.append(1 + (excl[l] >> 16) - (excl[s_] >> 16))
                sord.append(docs[docj[l]]["dso"] + ((excl[l] >> 8) & 0xFF) - ((excl[s_] >> 8) & 0xFF))
        toff = [docs[docj[l]]["toff"] for l in range(64)]
        room = [docs[docj[l]]["room"] for l in range(64)]
        words = [incd[l] >> 16 for l in range(64)]
        DEEP = ballot(x >= depth_limit for x in h)
        ROOT_END = ballot(x == 1 for x in h) & CLOSE
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
        for l in range(64):
            add = ((token[l] >> 3) & 1) | (0x80000000 if (closer >> l) & 1 else 0)
            if in_step[l]:
                cnt[par_lane[l]] = (cnt[par_lane[l]] + add) & 0xFFFFFFFF
            else:
                stk[lvl[l]] = (stk[lvl[l]][0], (stk[lvl[l]][1] + add) & 0xFFFFFFFF)
        se_x = [stk[lvl[l]][0] for l in range(64)]
        pcnt = [cnt[par_lane[l]] if in_step[l] else stk[lvl[l]][1] for l in range(64)]
        own = list(cnt)
        opener = [(tpos[l] & 0x7FFFFFFF) | (0x80000000 if tk[l] == TK_OPEN_A else 0) for l in range(64)]
        for l in range(64):
            if (OPEN >> l) & 1 and not (EO >> l) & 1 and not own[l] & 0x80000000:
                stk[h[l] & 63] = (opener[l], own[l])
        par = [opener[par_lane[l]] if in_step[l] else se_x[l] for l in range(64)]
        par_tpos = [x & 0x7FFFFFFF for x in par]
        par_cnt = [x & 0x3FFFFFFF for x in pcnt]
        gi = [(token[l] & 0x1F) | ((prev[l] & 0x17) << 5) | ((par[l] >> 21) & 0x400) for l in range(64)]
        BAD = ballot(grammar(g) == 0 for g in gi) & V
        BAD |= OPEN & ~EO & DEEP & M64
        BAD |= ballot(tpos[l] + words[l] + 1 > room[l] for l in range(64)) & V       # (+ the closing root word)
        # nothing may follow a root's end inside its document: the lanes behind a root's end up to the next document start
        stops = (DS | (1 << nv)) & ((1 << 65) - 1)
        A = (ROOT_END << 1) & ((1 << 65) - 1) & ~stops                                # (a root's end directly in front of a stop: nothing behind it)
        BAD |= ((stops - A) & ~stops) & V & M64 if A else 0
        # ... and the first segment inherits "my root has ended" from the previous step
        if c_root_end and c_seen:
            first_stop = (DS & -DS).bit_length() - 1 if DS else nv
            BAD |= first(first_stop) & V if first_stop else 0
        # a document must end on its root's end: the token in front of every document start
        ended_ok = ((ROOT_END << 1) | (1 if c_root_end else 0)) & M64
        NOT_CLOSED = DS & ~ended_ok & M64
        for l in range(64):
            if (NOT_CLOSED >> l) & 1 and (l > 0 or c_seen):
                failed |= 1 << (docj[l - 1] if l > 0 else c_doc)
            if (BAD >> l) & 1:
                failed |= 1 << docj[l]
        live = ballot(not (failed >> docj[l]) & 1 for l in range(64)) & V
        for l in range(64):
            if not (live >> l) & 1:
                continue
            T = toff[l]
            if (DS >> l) & 1:                                                        # both root words, from the predicted length
                tape[T] = (ord("r") << 56) | room[l]
                tape[T + room[l] - 1] = ord("r") << 56
            if (Q >> l) & 1:
                tape[T + tpos[l]] = (ord('"') << 56) | (string_base + record_offsets[sord[l]])
            if (PRIM >> l) & 1:
                queue.append((p[l], T + tpos[l], docj[l]))
            ch = (token[l] >> 8) & 0xFF
            if ((EO | CLOSE) >> l) & 1:
                pay = tpos[l] + 2 if (EO >> l) & 1 else (tpos[l] if (EC >> l) & 1 else par_tpos[l])
                tape[T + tpos[l]] = (ch << 56) | pay
            if ((CLOSE & ~EC) >> l) & 1:
                tape[T + par_tpos[l]] = ((ch - 2) << 56) | (min(par_cnt[l] + 1, 0xFFFFFF) << 32) | (tpos[l] + 1)
        # carries: the bases of the document that continues = its values at the end of this step, relative to the next step's lane 0
        last = nv - 1
        tot = scan[last]
        s_ = src[last]
        if s_ < 0:
            Hb = Hb + (tot & 0xFF) - nv
            Wb = Wb + (tot >> 16)
            Qb = Qb + ((tot >> 8) & 0xFF)
        else:
            Hb = (tot & 0xFF) - nv - dep[s_]
            Wb = 1 + (tot >> 16) - (excl[s_] >> 16)
            Qb = docs[docj[last]]["dso"] + ((tot >> 8) & 0xFF) - ((excl[s_] >> 8) & 0xFF)
        c_token = token[last]
        c_empty_open = (EO >> last) & 1
        c_root_end = (ROOT_END >> last) & 1
        c_doc = docj[last]
        c_seen = True
        head += nv
    if c_seen and not c_root_end:
        failed |= 1 << c_doc                                                       # the run's last document never closed its root
    for p, t, j in queue:                                                           # the dense literal parser behind the steps
        if (failed >> j) & 1:
            continue
        a = atom(buf, p)
        if a is not None:
            tape[t] = a << 56
            continue
        r = parse_number(buf, p, len(buf)) if (buf[p] == 0x2D or 0x30 <= buf[p] <= 0x39) else ("err", 0)
        if r[0] == "err":
            failed |= 1 << j
            continue
        tape[t] = ord(r[0]) << 56
        tape[t + 1] = r[1]
    out = []
    for j, d in enumerate(docs):
        if (failed >> j) & 1:
            out.append((False, None))
        else:
            out.append((True, [tape.get(d["toff"] + i) for i in range(d["room"])]))
    return out
