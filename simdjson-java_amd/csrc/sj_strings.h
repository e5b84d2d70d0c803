// sj_strings.h -- per-64-byte-block algebra of the streaming string pass (csrc/strings.hip), bit-plane form.
//
// Replaces the per-string calls StringParser.parseString(buf, idx, stringBuffer, stringBufferIdx)
// (/root/reference/src/main/java/org/simdjson/StringParser.java:18-68, helpers :112-153,
//  CharacterUtils.escape CharacterUtils.java:52-83, hexToInt :241-247) that the reference's stage 2 makes once per
// string (TapeBuilder.visitString, TapeBuilder.java:174-177) by ONE pass over the document: the string buffer --
// records [be32 length][unescaped bytes], in document order -- is a byte compaction of the document with a 4-byte
// header inserted at every opening quote, so a record's offset is a prefix sum over bytes and its length is the
// difference of two such sums.  One GPU lane owns one 64-byte block (as in stage 1, sj_block.h); this header holds
// everything a lane decides on its own from the block's bit planes:
//   K   the source bytes that go to the string buffer unchanged ("kept")
//   O   opening quotes (each makes a header), CL closing quotes
//   escaped b f n r t: kept, and patched to the control character afterwards (XOR with a constant)
//   \uXXXX: decided at the LAST hex digit by looking back (so a lane never needs bytes behind its block: the
//   16 bytes in front of it, the "halo", are classified with the same algebra); the UTF-8 bytes replace the last
//   1..4 hex digits of the sequence ("slots"), everything else of the sequence is dropped
//   the error StringParser would throw, by position (the lowest position of a string is its first error).
// Compiles as plain C++ (tests/host_sim/str_sim.cpp runs it against the oracle without a GPU) and as HIP device code.
#pragma once
#include "sj_block.h"

// ---- byte classes from bit planes; T = sj_u64 (a block) or uint32_t (the 16-byte halo, low 16 bits valid) ----
template <class T>
struct SjStrClasses {
    T bs, rawquote;          // '\\'  '"'
    T isu;                   // 'u'
    T esc_ok;                // " / \ b f n r t u   (CharacterUtils.java:52-72 plus 'u')
    T cn, ct, cr, cbf;       // n, t, r, b|f
    T hv;                    // hex digit (CharacterUtils.java:241-247 accepts exactly 0-9 a-f A-F)
    T d0, d1, d2, d3;        // its value, bit planes
};

template <class T>
SJ_HD void sj_str_quote_bs(const T p[8], T* bs, T* rawquote) {
    const T a = ~p[7] & ~p[6], b = ~p[7] & p[6];
    *bs = b & ~p[5] & p[4] & p[3] & p[2] & ~p[1] & ~p[0];        // 0x5C
    *rawquote = a & p[5] & ~p[4] & ~p[3] & ~p[2] & p[1] & ~p[0];  // 0x22
}

template <class T>
SJ_HD SjStrClasses<T> sj_str_classes(const T p[8], bool want_hex) {
    const T p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6], p7 = p[7];
    const T a = ~p7 & ~p6, b = ~p7 & p6;
    SjStrClasses<T> c;
    // shared decodes: high nibble 2 / 5 / 6 / 7, low nibble by halves
    const T h2 = a & p5 & ~p4, h5 = b & ~p5 & p4, h6 = b & p5 & ~p4, h7 = b & p5 & p4;
    const T n00 = ~p3 & ~p2, n01 = ~p3 & p2, n11 = p3 & p2;
    const T x00 = ~p1 & ~p0, x01 = ~p1 & p0, x10 = p1 & ~p0;
    const T l0010 = n00 & x10;
    c.bs = h5 & n11 & x00;                  // 0x5C
    c.rawquote = h2 & l0010;                // 0x22
    const T slash = h2 & n11 & p1 & p0;     // 0x2F
    c.cn = h6 & n11 & x10;                  // 0x6E
    c.cr = h7 & l0010;                      // 0x72
    c.ct = h7 & n01 & x00;                  // 0x74
    c.isu = h7 & n01 & x01;                 // 0x75
    c.cbf = h6 & ~p3 & x10;                 // 0x62, 0x66
    c.esc_ok = c.rawquote | slash | c.bs | c.cbf | c.cn | c.cr | c.ct | c.isu;
    c.hv = c.d0 = c.d1 = c.d2 = c.d3 = 0;
    if (want_hex) {
        const T dec = a & p5 & p4 & (~p3 | (~p2 & ~p1));                              // 0x30..0x39
        const T alpha = b & ~p4 & ~p3 & (p2 | p1 | p0) & ~(p2 & p1 & p0);             // 0x41..0x46, 0x61..0x66
        c.hv = dec | alpha;
        // value of a letter = low nibble + 9 (low nibble 1..6)
        c.d0 = (dec & p0) | (alpha & ~p0);
        c.d1 = (dec & p1) | (alpha & (p1 ^ p0));
        c.d2 = (dec & p2) | (alpha & (p2 ^ (p1 & p0)));
        c.d3 = (dec & p3) | alpha;
    }
    return c;
}

// ---- 80-bit masks: the block (positions 0..63) with its halo in front (positions -16..-1 = bits 0..15 of h) ----
struct SjX {
    sj_u64 c;
    uint32_t h;
};
SJ_HD SjX operator&(SjX a, SjX b) { return SjX{a.c & b.c, a.h & b.h}; }
SJ_HD SjX operator|(SjX a, SjX b) { return SjX{a.c | b.c, a.h | b.h}; }
SJ_HD SjX operator~(SjX a) { return SjX{~a.c, ~a.h & 0xFFFFu}; }
// the value K positions back / ahead
template <int K>
SJ_HD SjX sjx_back(SjX x) { return SjX{(x.c << K) | (sj_u64)(x.h >> (16 - K)), (x.h << K) & 0xFFFFu}; }
template <int K>
SJ_HD SjX sjx_ahead(SjX x) { return SjX{x.c >> K, ((x.h >> K) | (uint32_t)(x.c << (16 - K))) & 0xFFFFu}; }

// StructuralIndexer.java:211-229 (the odd/even backslash-run carry): which bytes are escaped
template <class T>
SJ_HD T sj_escaped_mask(T bsraw, uint32_t e_in) {
    const T EVEN = (T)0x5555555555555555ull;
    const T bs = bsraw & ~(T)e_in;
    const T follows = (T)(bs << 1) | (T)e_in;
    const T odd_starts = bs & ~EVEN & ~follows;
    const T seq_even = odd_starts + bs;
    return (EVEN ^ (T)(seq_even << 1)) & follows;
}

// The strings OPENED in the first `upto` bytes (< 64) of the 64-byte block at buf + start, given whether the block is entered
// inside a string and whether its first byte is escaped: quotes and backslashes of the block as 64-bit masks (a SWAR compare
// per dword, a multiply gathers the four flag bits), then StructuralIndexer.java:211-234 on them.  NOT the '"' structurals of the
// index array: a quote directly behind a primitive (1"abc") opens a string for the string pass without being a structural.
SJ_HD uint32_t sj_str_opens_before(const uint8_t* buf, sj_u64 start, uint32_t upto, uint32_t in_str, uint32_t e_in) {
    sj_u64 qm = 0, bm = 0;
    uint32_t w16[16];  // (the block is 16-byte aligned: four 16-byte loads on the device)
    for (int v = 0; v < 4; ++v) {
        struct alignas(16) Q4 { uint32_t a, b, c, d; };
        const Q4 q4 = reinterpret_cast<const Q4*>(buf + start)[v];
        w16[4 * v] = q4.a;
        w16[4 * v + 1] = q4.b;
        w16[4 * v + 2] = q4.c;
        w16[4 * v + 3] = q4.d;
    }
    for (int i = 0; i < 16; ++i) {
        const uint32_t w = w16[i];
        const uint32_t zq = w ^ 0x22222222u, zb = w ^ 0x5C5C5C5Cu;
        const uint32_t fq = ~(((zq & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | zq) & 0x80808080u;  // 0x80 where the byte matches
        const uint32_t fb = ~(((zb & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | zb) & 0x80808080u;
        const sj_u64 nq = (((fq >> 7) * 0x00204081u) >> 21) & 0xFu, nb = (((fb >> 7) * 0x00204081u) >> 21) & 0xFu;
        qm |= nq << (4 * i);
        bm |= nb << (4 * i);
    }
    const sj_u64 quote = qm & ~sj_escaped_mask<sj_u64>(bm, e_in);
    const sj_u64 in0 = sj_prefix_xor(quote);
    const sj_u64 opens = quote & (in_str ? ~in0 : in0);
    const sj_u64 below = upto >= 64u ? ~0ull : ((1ull << upto) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(opens & below);
#else
    return (uint32_t)__builtin_popcountll(opens & below);
#endif
}

struct SjStrBlock {
    sj_u64 K;            // kept bytes (content that is copied, slots of \u sequences included)
    sj_u64 O, CL;        // opening / closing quotes
    sj_u64 pn, pt, pr, pbf;  // escaped n / t / r / b|f inside strings: patched after the copy
    sj_u64 l1, l2, l3, pair;  // \u sequences by the position of their last hex digit: 1 / 2 / 3 UTF-8 bytes, surrogate pair (4)
    sj_u64 e4, e5, e6, e7, e8;  // error positions by SJMI_E_* code (4 ESCAPE_UNEXPECTED .. 8 LOW_SURROGATE_RANGE)
    uint32_t head;       // slots of a sequence that ends here but began in the previous block: bytes in front of the block's own
    uint32_t exit_in;    // in-string parity after the block
    uint32_t any_escape; // the block has an escaped character inside a string
};

// Stage A: quotes, escapes, content.  e_in: byte 0 is escaped; pin: the block is entered inside a string.
struct SjStrBase {
    sj_u64 bsraw, escaped, quote, in, C, ES, ED;
};
SJ_HD SjStrBase sj_str_base(const sj_u64 p[8], uint32_t e_in, uint32_t pin) {
    SjStrBase s;
    sj_u64 rawquote;
    sj_str_quote_bs<sj_u64>(p, &s.bsraw, &rawquote);
    s.escaped = sj_escaped_mask<sj_u64>(s.bsraw, e_in);
    s.quote = rawquote & ~s.escaped;                       // StructuralIndexer.java:232
    const sj_u64 in0 = sj_prefix_xor(s.quote);             // :233 (includes the opening, excludes the closing quote)
    s.in = pin ? ~in0 : in0;
    s.C = s.in & ~s.quote;                                 // strictly inside a string
    s.ES = s.bsraw & ~s.escaped & s.C;                     // a backslash that starts an escape (StringParser.java:42)
    s.ED = s.escaped & s.C;                                // the character it escapes
    return s;
}

// The halo's planes: hp[k] bit t = bit k of halo byte t (t = 0..15 <-> positions -16..-1).
struct SjStrHalo {
    uint32_t hp[8];
    uint32_t e_in;  // halo byte 0 is escaped (only needed when the halo begins with four backslashes: see sj_str_halo_unresolved)
};
// positions >= -12 of the halo are classified exactly unless a backslash run covers its first four bytes
SJ_HD bool sj_str_halo_unresolved(const uint32_t hp[8]) {
    uint32_t bs, rq;
    sj_str_quote_bs<uint32_t>(hp, &bs, &rq);
    return (bs & 0xFu) == 0xFu;
}

// The whole block.  do_escapes = false: the caller knows that no lane of the wave has an escaped character inside a
// string (then K = C).  do_u = false: ... that no \u sequence ends in or reaches into any block of the wave (then the halo
// is not looked at).  Both are pure shortcuts.
SJ_HD SjStrBlock sj_str_block(const sj_u64 p[8], const SjStrBase& s, uint32_t pin, bool do_escapes, bool do_u,
                              const SjStrHalo* halo) {
    SjStrBlock r;
    r.O = s.quote & s.in;
    r.CL = s.quote & ~s.in;
    r.exit_in = (uint32_t)(s.in >> 63);
    r.any_escape = s.ED != 0;
    r.K = s.C & ~s.ES;
    r.pn = r.pt = r.pr = r.pbf = 0;
    r.l1 = r.l2 = r.l3 = r.pair = 0;
    r.e4 = r.e5 = r.e6 = r.e7 = r.e8 = 0;
    r.head = 0;
    if (!do_escapes) return r;
    const SjStrClasses<sj_u64> c = sj_str_classes<sj_u64>(p, do_u);
    r.pn = s.ED & c.cn;
    r.pt = s.ED & c.ct;
    r.pr = s.ED & c.cr;
    r.pbf = s.ED & c.cbf;
    r.e4 = s.ED & ~c.esc_ok;  // CharacterUtils.java:74-83 (a non-ASCII byte is not in the table either: StringParser.java:58)
    if (!do_u) return r;

    // ---- \uXXXX (StringParser.java:45-57, :112-153), on the block with its halo in front ----
    const SjStrClasses<uint32_t> hc = sj_str_classes<uint32_t>(halo->hp, true);
    const uint32_t hesc = sj_escaped_mask<uint32_t>(hc.bs & 0xFFFFu, halo->e_in) & 0xFFFFu;
    const uint32_t hquote = hc.rawquote & ~hesc & 0xFFFFu;
    // in-string state after each halo byte, backwards from the state at the block's entry
    uint32_t sfx = hquote;
    sfx ^= sfx >> 1;
    sfx ^= sfx >> 2;
    sfx ^= sfx >> 4;
    sfx ^= sfx >> 8;  // sfx bit t = parity of the quotes at halo positions >= t
    const uint32_t hin = ((pin ? 0xFFFFu : 0u) ^ sfx ^ hquote) & 0xFFFFu;
    const uint32_t hC = hin & ~hquote;
    const uint32_t hEU = hesc & hC & hc.isu & 0xFFF0u;  // (positions -16..-13 are only there to resolve the runs)

    const SjX EU{s.ED & c.isu, hEU};
    const SjX Q{s.quote, hquote};
    const SjX BS{c.bs, hc.bs & 0xFFFFu};
    const SjX U{c.isu, hc.isu & 0xFFFFu};
    const SjX HV{c.hv, hc.hv & 0xFFFFu};
    const SjX D0{c.d0, hc.d0 & 0xFFFFu}, D1{c.d1, hc.d1 & 0xFFFFu}, D2{c.d2, hc.d2 & 0xFFFFu}, D3{c.d3, hc.d3 & 0xFFFFu};
    const SjX NQ = ~Q;
    // the u and the (up to) four bytes behind it emit nothing -- a closing quote cuts the sequence short
    const SjX g1 = sjx_back<1>(EU) & NQ, g2 = sjx_back<1>(g1) & NQ, g3 = sjx_back<1>(g2) & NQ, g4 = sjx_back<1>(g3) & NQ;
    const SjX DROP = EU | g1 | g2 | g3 | g4;
    // (what StringParser reads as hex digits is not looked at as an escape: "\u\q.." is an invalid unicode escape)
    r.e4 &= ~(g1 | g2 | g3).c;
    const SjX SEQ = sjx_back<4>(EU);  // a sequence's last hex digit is here
    const SjX hexok = sjx_back<3>(HV) & sjx_back<2>(HV) & sjx_back<1>(HV) & HV;
    // code point bits 15..7 at the last digit: digit 1 is three back, digit 2 two, digit 3 one
    const SjX c15 = sjx_back<3>(D3), c14 = sjx_back<3>(D2), c13 = sjx_back<3>(D1), c12 = sjx_back<3>(D0);
    const SjX c11 = sjx_back<2>(D3), c10 = sjx_back<2>(D2), c9 = sjx_back<2>(D1), c8 = sjx_back<2>(D0);
    const SjX c7 = sjx_back<1>(D3);
    const SjX sur = c15 & c14 & ~c13 & c12 & c11;  // D800..DFFF
    const SjX V = SEQ & hexok;
    const SjX HS = V & sur & ~c10;  // D800..DBFF (StringParser.java:50)
    const SjX LS = V & sur & c10;   // DC00..DFFF (:53)
    const SjX BMP = V & ~sur;
    const SjX z1 = ~(c15 | c14 | c13 | c12 | c11);  // < 0x800
    const SjX L1 = BMP & z1 & ~(c10 | c9 | c8 | c7);
    const SjX L2 = BMP & z1 & ~L1;
    const SjX L3 = BMP & ~z1;
    const SjX HS6 = sjx_back<6>(HS);
    const SjX PAIR = LS & HS6;  // :112-124
    // errors, at a position inside the string (content or its closing quote); a sequence cut by the closing quote is
    // reported AT the quote, so that nothing is ever reported behind the string it belongs to
    const SjX q1 = sjx_back<1>(Q), q2 = sjx_back<2>(Q), q3 = sjx_back<3>(Q);
    const SjX uncut = ~(q1 | q2 | q3);
    const SjX lowu = sjx_back<5>(BS) & sjx_back<4>(U);  // "\u" right behind a high surrogate, seen from its last digit
    const sj_u64 hex_e = (SEQ & ~hexok & ~HS6 & uncut).c;                          // :127-129
    const sj_u64 rng_e = (HS6 & lowu & ~LS & uncut).c;                             // :118-122 (also a malformed low half)
    const sj_u64 lone_e = (LS & ~HS6).c;                                           // :53-55
    const sj_u64 nou_e = (sjx_back<2>(HS) & ~(sjx_back<1>(BS) & U) & ~q1).c;       // :113-115
    // (a closing quote one, two or three bytes behind an escaped u, with no other quote in between: g1 / g2 above)
    const SjX EUH = EU & sjx_back<2>(HS);  // the "\u" right behind a high surrogate
    const SjX h1 = sjx_back<1>(EUH) & NQ, h2 = sjx_back<1>(h1) & NQ;
    const sj_u64 cut_rng = r.CL & (sjx_back<1>(EUH) | sjx_back<1>(h1) | sjx_back<1>(h2)).c;
    const sj_u64 cut_hex = r.CL & (sjx_back<1>(EU) | sjx_back<1>(g1) | sjx_back<1>(g2)).c & ~cut_rng;
    const sj_u64 cut_nou = r.CL & sjx_back<1>(HS).c;
    const sj_u64 inside = s.C | r.CL;
    r.e5 = (hex_e & inside) | cut_hex;
    r.e6 = lone_e & inside;
    r.e7 = (nou_e & inside) | cut_nou;
    r.e8 = (rng_e & inside) | cut_rng;
    r.l1 = L1.c;
    r.l2 = L2.c;
    r.l3 = L3.c;
    r.pair = PAIR.c;
    // slots: the last 1..4 digit positions of a sequence that ends in THIS block
    const SjX T1{r.l1 | r.l2 | r.l3 | r.pair, 0}, T2{r.l2 | r.l3 | r.pair, 0}, T3{r.l3 | r.pair, 0}, T4{r.pair, 0};
    const SjX KU = T1 | sjx_ahead<1>(T2) | sjx_ahead<2>(T3) | sjx_ahead<3>(T4);
    r.K = (s.C & ~s.ES & ~DROP.c) | KU.c;
    r.head = (uint32_t)__builtin_popcount(KU.h);
    // (an escaped character that a malformed \u swallowed as a digit is not copied, so there is nothing to patch)
    r.pn &= r.K;
    r.pt &= r.K;
    r.pr &= r.K;
    r.pbf &= r.K;
    return r;
}

// offset of block position `pos` (0..64) in the block's output: bytes kept in front of it + 4 per header
SJ_HD uint32_t sj_str_offset(const SjStrBlock& b, uint32_t pos) {
    const sj_u64 lt = pos >= 64 ? ~0ull : ((1ull << pos) - 1ull);
    return b.head + (uint32_t)__builtin_popcountll(b.K & lt) + 4u * (uint32_t)__builtin_popcountll(b.O & lt);
}
SJ_HD uint32_t sj_str_out_bytes(const SjStrBlock& b) { return sj_str_offset(b, 64); }

// the SJMI_E_* code of the lowest error position in m-masked block positions (0 = none); ties: see DESIGN.md 4.2
SJ_HD uint32_t sj_str_first_error(const SjStrBlock& b, sj_u64 m, uint32_t* pos) {
    const sj_u64 any = (b.e4 | b.e5 | b.e6 | b.e7 | b.e8) & m;
    if (!any) return 0;
    const sj_u64 low = any & (0 - any);
    *pos = (uint32_t)__builtin_ctzll(any);
    if (b.e7 & low) return 7;
    if (b.e8 & low) return 8;
    if (b.e6 & low) return 6;
    if (b.e5 & low) return 5;
    return 4;
}

// kept bytes of one dword, packed to the front (v_perm_b32 with a selector from a 16-entry table)
SJ_HD uint32_t sj_str_pack_selector(uint32_t nibble) {
    uint32_t sel = 0, n = 0;
    for (uint32_t i = 0; i < 4; ++i)
        if ((nibble >> i) & 1u) sel |= i << (8 * n++);
    for (; n < 4; ++n) sel |= 0x0Cu << (8 * n);  // 0x0C = constant zero byte
    return sel;
}

// opening quotes that shift the kept bytes of their own dword (no kept byte in front of them inside the dword), and the
// one pattern a dword cannot be copied in one piece for: kept, closing quote, opening quote, kept ("a""b")
struct SjStrGroups {
    sj_u64 oshift;  // O & ~(kept before in the same dword)
    sj_u64 bad;     // O with kept bytes on both sides inside the dword (always byte 2 of its dword)
};
SJ_HD SjStrGroups sj_str_groups(sj_u64 K, sj_u64 O) {
    const sj_u64 kb = ((K & 0x7777777777777777ull) << 1) | ((K & 0x3333333333333333ull) << 2) | ((K & 0x1111111111111111ull) << 3);
    const sj_u64 ka = ((K & 0xEEEEEEEEEEEEEEEEull) >> 1) | ((K & 0xCCCCCCCCCCCCCCCCull) >> 2) | ((K & 0x8888888888888888ull) >> 3);
    SjStrGroups g;
    g.oshift = O & ~kb;
    g.bad = O & kb & ka;
    return g;
}

SJ_HD int32_t sj_hex4_word(uint32_t w) {  // four hex digits, first in the low byte (CharacterUtils.java:241-247)
    uint32_t v = 0, bad = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = (w >> (8 * i)) & 0xFFu;
        const uint32_t dec = c - '0', alpha = (c | 0x20u) - 'a';
        const uint32_t d = dec <= 9u ? dec : alpha + 10u;
        bad |= (dec > 9u) & (alpha > 5u);
        v = (v << 4) | (d & 15u);
    }
    return bad ? -1 : (int32_t)v;
}

// the same for four bytes that are KNOWN to be hex digits (the plane algebra has validated the sequence: sj_str_block's hexok): all
// four nibbles at once -- low nibble of the byte, + 9 for a letter (bit 6 is set in A-F / a-f and clear in 0-9) -- about a dozen
// operations instead of four rounds of range tests (the \u patch loop of strings.hip runs once per sequence of the busiest lane)
SJ_HD uint32_t sj_hex4_valid_word(uint32_t w) {
    const uint32_t letter = (w >> 6) & 0x01010101u;
    const uint32_t n = (w & 0x0F0F0F0Fu) + letter * 9u;  // (a byte's nibble value: at most 15, no carry between bytes)
    return ((n & 0xFu) << 12) | ((n & 0xF00u)) | ((n >> 12) & 0xF0u) | (n >> 24);
}

// UTF-8 of a code point, first byte in the low byte (StringParser.java:126-153)
SJ_HD uint32_t sj_utf8_bytes(uint32_t cp, uint32_t* n) {
    if (cp <= 0x7F) {
        *n = 1;
        return cp;
    }
    if (cp <= 0x7FF) {
        *n = 2;
        return ((cp >> 6) + 192) | (((cp & 63) + 128) << 8);
    }
    if (cp <= 0xFFFF) {
        *n = 3;
        return ((cp >> 12) + 224) | ((((cp >> 6) & 63) + 128) << 8) | (((cp & 63) + 128) << 16);
    }
    *n = 4;
    return ((cp >> 18) + 240) | ((((cp >> 12) & 63) + 128) << 8) | ((((cp >> 6) & 63) + 128) << 16) | (((cp & 63) + 128) << 24);
}
