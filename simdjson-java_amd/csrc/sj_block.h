// sj_block.h -- per-64-byte-block stage-1 algebra, bit-plane ("transposed") form.
//
// One GPU lane owns one 64-byte block of the document, exactly the unit one iteration of
// the reference's StructuralIndexer.index512 loop consumes
// (/root/reference/src/main/java/org/simdjson/StructuralIndexer.java:206-253), but instead of
// the reference's 5 vector compares + 2 nibble shuffles the block is first transposed into
// 8 bit planes (plane k, bit i = bit k of byte i).  Every character class of
// StructuralIndexer (:210,:231-232,:237-240) and the whole of Utf8Validator.validate
// (Utf8Validator.java:54-168) then become 64-bit boolean algebra on the planes, which is what
// a CDNA4 lane is good at (no byte shuffles / byte compares in the VALU).
//
// Everything cross-block is passed in explicitly:
//   e_in  -- is byte 0 of the block escaped   (reference: prevEscaped,  :197-201)
//   p_in  -- was the previous byte a non-quote scalar (reference: prevScalar)
//   Utf8Carry -- what the previous 3 bytes expect from this block (reference: previous 4 bytes
//            + previousIncomplete, Utf8Validator.java:55-57)
// and the in-string parity (reference: prevInString) is NOT an input: the block returns masks
// for an incoming parity of 0; the caller flips them once the parity prefix is known
// (structurals(p=1) = pot & sm0, structurals(p=0) = pot & ~sm0).
//
// Compiles as plain C++ (host fuzzing against the oracle: tests/host_sim) and as HIP device code.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SJ_HD __host__ __device__ __forceinline__
#else
#define SJ_HD inline
#endif

typedef unsigned long long sj_u64;

struct SjUtf8Carry {
    uint32_t c1;   // bit0: position 0 must be a continuation because byte -1 is a 2/3/4-byte lead
    uint32_t c2;   // bit0: byte -2 is a 3/4-byte lead; bit1: byte -1 is a 3/4-byte lead
    uint32_t c3;   // bit0: byte -3 is a 4-byte lead; bit1: byte -2 is; bit2: byte -1 is
    uint32_t sec;  // byte -1 is: bit0 0xE0, bit1 0xED, bit2 0xF0, bit3 0xF4 (second-byte range checks)
};

struct SjBlockMasks {
    sj_u64 pot;      // potential structural starts: op | scalar start        (StructuralIndexer.java:243-248)
    sj_u64 sm0;      // (inString ^ quote) for incoming parity 0              (:233,:251)
    uint32_t qpar;   // parity of unescaped quotes in the block               (:234)
    uint32_t ue0;    // unescaped control char inside a string, if parity 0   (:231,:252)
    uint32_t ue1;    // ... if incoming parity is 1
    uint32_t utf8;   // block contains a UTF-8 error                          (Utf8Validator.java:109-110,165)
    uint32_t words;  // (want_words) tape words the block's structurals make, entered outside a string | inside << 8: see sj_block
};

// The reference's own per-block locals that the kernels never materialise (they work with pot / sm0): filled in on
// request for the bit-mask parity entry point (sjmi_stage1_masks), see sj_reference_masks below.
struct SjBlockDetail {
    sj_u64 escaped;  // StructuralIndexer.java:213-228
    sj_u64 quote;    // :232
    sj_u64 in0;      // inString for incoming parity 0 (:233)
    sj_u64 op;       // :239-240
    sj_u64 ws;       // :237-238
};

SJ_HD sj_u64 sj_prefix_xor(sj_u64 m) {  // StructuralIndexer.java:311-319
    m ^= m << 1;
    m ^= m << 2;
    m ^= m << 4;
    m ^= m << 8;
    m ^= m << 16;
    m ^= m << 32;
    return m;
}

// Bytes at index >= valid are invisible: the reference copies the tail into a space-filled
// block for indexing (StructuralIndexer.java:305-309) and zero-pads it for UTF-8
// (Utf8Validator.java:115-117); both views agree on "ASCII, not a continuation", which is all
// the UTF-8 algebra below looks at, so one space-padded view serves both.
SJ_HD void sj_mask_tail(sj_u64 p[8], uint32_t valid) {
    if (valid >= 64) return;
    const sj_u64 vm = (1ull << valid) - 1ull;
    for (int k = 0; k < 8; ++k) p[k] &= vm;
    p[5] |= ~vm;  // 0x20
}

// do_utf8 = false skips the UTF-8 algebra; only legal when the caller knows the block is pure ASCII and uc is
// all zero (the kernel decides per wave with a ballot), in which case the result is identical.
// want_words: also count the TAPE WORDS the block's structurals make (Tape.java:33-43, TapeBuilder.java:41-48,191-208: a
// bracket, a string or an atom one word, a number two, ',' and ':' none) for both entry parities (<= 128 each).  Exact for a
// well-formed document -- there a primitive that begins with '-' or a digit IS a number -- which is all the batch pipeline uses
// it for (it lays the tapes out before the documents are walked; a malformed document's slot is merely unused).
SJ_HD SjBlockMasks sj_block(const sj_u64 p[8], uint32_t e_in, uint32_t p_in, SjUtf8Carry uc, bool do_utf8 = true,
                            SjBlockDetail* det = nullptr, bool want_words = false) {
    const sj_u64 p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6], p7 = p[7];
    const sj_u64 a = ~p7 & ~p6;  // 0x00..0x3F
    const sj_u64 b = ~p7 & p6;   // 0x40..0x7F
    // low-nibble decodes
    const sj_u64 n00 = ~p3 & ~p2, n11 = p3 & p2, n10 = p3 & ~p2;
    const sj_u64 n_0000 = n00 & ~p1 & ~p0;
    const sj_u64 n_0010 = n00 & p1 & ~p0;
    const sj_u64 n_1100 = n11 & ~p1 & ~p0;
    const sj_u64 n_1101 = n11 & ~p1 & p0;
    const sj_u64 n_1001 = n10 & ~p1 & p0;
    const sj_u64 n_1010 = n10 & p1 & ~p0;
    const sj_u64 n_1011 = n10 & p1 & p0;

    // ---- character classes (StructuralIndexer.java:210,231-232,237-240) ---------------------
    sj_u64 bs = b & ~p5 & p4 & n_1100;            // '\\' 0x5C
    const sj_u64 rawquote = a & p5 & ~p4 & n_0010; // '"'  0x22
    const sj_u64 ctrl = a & ~p5;                   // <= 0x1F
    // whitespace table (:23-25) matches exactly {0x20, 0x09, 0x0A, 0x0D}
    const sj_u64 ws = a & ~p4 & ((p5 & n_0000) | (~p5 & (n_1001 | n_1010 | n_1101)));
    // (c|0x20) == OP_TABLE[c&15] (:26-28,:239-240): { , : [ ] { } } and also 0x0C, 0x1A
    const sj_u64 op = (a & ~p4 & n_1100) | (a & p4 & n_1010) | (b & p4 & (n_1011 | n_1101));

    // ---- escapes (:211-229); prevEscaped = e_in. The bs==0 branch is the same formula. ----
    bs &= ~(sj_u64)e_in;
    const sj_u64 follows_escape = (bs << 1) | e_in;
    const sj_u64 EVEN = 0x5555555555555555ull;
    const sj_u64 odd_starts = bs & ~EVEN & ~follows_escape;
    const sj_u64 seq_even = odd_starts + bs;
    const sj_u64 escaped = (EVEN ^ (seq_even << 1)) & follows_escape;

    // ---- strings (:232-234) ------------------------------------------------------------------
    const sj_u64 quote = rawquote & ~escaped;
    const sj_u64 in0 = sj_prefix_xor(quote);  // in-string mask for incoming parity 0

    // ---- scalars / structural starts (:243-248) -------------------------------------------
    const sj_u64 scalar = ~(op | ws);
    const sj_u64 nqs = scalar & ~quote;
    const sj_u64 follows_nqs = (nqs << 1) | p_in;
    const sj_u64 pot = op | (scalar & ~follows_nqs);

    // ---- UTF-8 (Utf8Validator.java:54-168 as plane algebra; == strict RFC 3629) ------------
    sj_u64 err = 0;
    if (do_utf8) {
    const sj_u64 cont = p7 & ~p6;
    const sj_u64 lead = p7 & p6;
    const sj_u64 L2 = lead & ~p5;
    const sj_u64 L3 = lead & p5 & ~p4;
    const sj_u64 L4 = lead & p5 & p4 & ~p3;
    err = lead & p5 & p4 & p3;                             // 0xF8..0xFF
    const sj_u64 E1 = ((L2 | L3 | L4) << 1) | uc.c1;       // expected 1st continuation
    const sj_u64 E2 = ((L3 | L4) << 2) | uc.c2;            // expected 2nd
    const sj_u64 E3 = (L4 << 3) | uc.c3;                   // expected 3rd
    err |= cont ^ (E1 | E2 | E3);                          // TOO_SHORT / TOO_LONG / TWO_CONTINUATIONS
    err |= L2 & ~p4 & n00 & ~p1;                           // 0xC0, 0xC1            (OVERLONG_2BYTE)
    err |= L4 & p2 & (p1 | p0);                            // 0xF5..0xF7            (TOO_LARGE)
    const sj_u64 sE0 = ((L3 & n_0000) << 1) | (uc.sec & 1u);
    const sj_u64 sED = ((L3 & n_1101) << 1) | ((uc.sec >> 1) & 1u);
    const sj_u64 sF0 = ((L4 & ~p2 & ~p1 & ~p0) << 1) | ((uc.sec >> 2) & 1u);
    const sj_u64 sF4 = ((L4 & p2 & ~p1 & ~p0) << 1) | ((uc.sec >> 3) & 1u);
    err |= sE0 & ~p5;                                      // E0 80..9F             (OVERLONG_3BYTE)
    err |= sED & p5;                                       // ED A0..BF             (SURROGATE)
    err |= sF0 & ~p5 & ~p4;                                // F0 80..8F             (OVERLONG_4BYTE)
    err |= sF4 & (p5 | p4);                                // F4 90..BF             (TOO_LARGE)
    }

    if (det) {
        det->escaped = escaped;
        det->quote = quote;
        det->in0 = in0;
        det->op = op;
        det->ws = ws;
    }
    SjBlockMasks r;
    r.words = 0;
    if (want_words) {
        // (from the planes themselves, which are live to the end of the UTF-8 algebra anyway: re-using the nibble decodes of
        //  the top would keep five more masks alive across it -- the FAST S = 4 kernel then needs 132 VGPRs instead of 128)
        const sj_u64 a5 = ~p7 & ~p6 & p5;                                            // 0x20..0x3F
        const sj_u64 hi3 = p3 & p2, x1100 = hi3 & ~p1;                               // low nibble 110x
        const sj_u64 comma_colon = a5 & ((~p4 & x1100 & ~p0) | (p4 & p3 & ~p2 & p1 & ~p0));   // 0x2C, 0x3A (not the op table's 0x0C, 0x1A)
        const sj_u64 number = a5 & ((p4 & (~p3 | (~p2 & ~p1))) | (~p4 & x1100 & p0));          // 0x30..0x39, 0x2D
        const sj_u64 sm0_ = in0 ^ quote, s0 = pot & ~sm0_, s1 = pot & sm0_;
        auto pc = [](sj_u64 m) -> uint32_t {
#if defined(__HIP_DEVICE_COMPILE__)
            return (uint32_t)__popcll(m);
#else
            return (uint32_t)__builtin_popcountll(m);
#endif
        };
        r.words = (pc(s0 & ~comma_colon) + pc(s0 & number)) | ((pc(s1 & ~comma_colon) + pc(s1 & number)) << 8);
    }
    r.pot = pot;
    r.sm0 = in0 ^ quote;
    r.qpar = (uint32_t)(in0 >> 63);
    r.ue0 = (ctrl & in0) != 0;
    r.ue1 = (ctrl & ~in0) != 0;
    r.utf8 = err != 0;
    return r;
}

// The six masks one iteration of the reference's loop holds (StructuralIndexer.java:210-252), in the order
// {escaped, quote, inString, op, whitespace, structurals}, from the kernel's own formulation: the block's masks for an
// incoming in-string parity of 0 plus the parity resolved by the prefix scan.  inString(p=1) = ~in0 (prefixXor ^
// all-ones, :233-234); structurals(p=1) = pot & ~(~in0 ^ quote) = pot & sm0 (:251).
SJ_HD void sj_reference_masks(const SjBlockMasks& bm, const SjBlockDetail& det, uint32_t parity_in, sj_u64 out[6]) {
    out[0] = det.escaped;
    out[1] = det.quote;
    out[2] = parity_in ? ~det.in0 : det.in0;
    out[3] = det.op;
    out[4] = det.ws;
    out[5] = parity_in ? (bm.pot & bm.sm0) : (bm.pot & ~bm.sm0);
}

// ---- carries from the bytes before the block -------------------------------------------------
// halo = the 8 bytes preceding the block, little-endian (byte -1 is bits 56..63).  Branch-free SWAR: this runs
// once per block in every lane, so every instruction counts.

SJ_HD SjUtf8Carry sj_utf8_carry(sj_u64 halo) {
    const uint32_t hh = (uint32_t)(halo >> 32);  // bytes -4..-1; only bit 7 of each byte is meaningful below
    const uint32_t geC0 = hh & (hh << 1);        // byte >= 0xC0: a 2/3/4-byte lead
    const uint32_t geE0 = geC0 & (hh << 2);      // byte >= 0xE0: a 3/4-byte lead
    const uint32_t geF0 = geE0 & (hh << 3);      // byte >= 0xF0: a 4-byte lead
    const uint32_t h1 = hh >> 24;
    SjUtf8Carry c;
    c.c1 = geC0 >> 31;
    c.c2 = ((geE0 >> 23) & 1u) | ((geE0 >> 30) & 2u);
    c.c3 = ((geF0 >> 15) & 1u) | ((geF0 >> 22) & 2u) | ((geF0 >> 29) & 4u);
    c.sec = (uint32_t)(h1 == 0xE0) | ((uint32_t)(h1 == 0xED) << 1) | ((uint32_t)(h1 == 0xF0) << 2) |
            ((uint32_t)(h1 == 0xF4) << 3);
    return c;
}

SJ_HD uint32_t sj_is_ws_or_op(uint32_t c) {  // { 09 0A 0D 20 , : [ ] { } } and 0C, 1A (StructuralIndexer.java:23-28)
    // 128-bit membership bitmap, looked up with 32-bit operations only (64-bit shifts are half rate on gfx950)
    const uint32_t M0 = (1u << 0x09) | (1u << 0x0A) | (1u << 0x0C) | (1u << 0x0D) | (1u << 0x1A);  // 0x00..0x1F
    const uint32_t M1 = (1u << (0x20 - 32)) | (1u << (0x2C - 32)) | (1u << (0x3A - 32));            // 0x20..0x3F
    const uint32_t M2 = (1u << (0x5B - 64)) | (1u << (0x5D - 64));                                   // 0x40..0x5F
    const uint32_t M3 = (1u << (0x7B - 96)) | (1u << (0x7D - 96));                                   // 0x60..0x7F
    const uint32_t m = (c & 64u) ? ((c & 32u) ? M3 : M2) : ((c & 32u) ? M1 : M0);
    return (c < 128u) & ((m >> (c & 31u)) & 1u);
}

// Resolve e_in / p_in from the halo.  Returns false when the backslash run reaches past its 8 bytes (then the
// caller counts the run from memory: sj_carry_slow).
//   e_in = (length of the backslash run ending at byte -1) is odd
//   p_in = byte -1 is a scalar that is not an (unescaped) quote
// The common case needs only the top 4 bytes and 32-bit operations; a run that fills them (4 backslashes, or a quote
// preceded by 3) continues into the low 4 bytes -- rare, but with 10 % escapes in the strings it happened in 1 % of
// the 64-block wave-steps, and one such block sends the whole wave through the slow path.
SJ_HD bool sj_carry_from_halo(sj_u64 halo, uint32_t* e_in, uint32_t* p_in) {
    const uint32_t hh = (uint32_t)(halo >> 32);                   // bytes -4..-1, byte -1 on top
    const uint32_t z = hh ^ 0x5C5C5C5Cu;                          // zero bytes = backslashes
    const uint32_t h1 = hh >> 24;
    uint32_t run1 = z ? (uint32_t)__builtin_clz(z) >> 3 : 4u;               // backslashes ending at byte -1 (0..4)
    uint32_t run2 = (uint32_t)__builtin_clz((z << 8) | 0xFFu) >> 3;         // ... ending at byte -2 (0..3)
    const bool is_q = h1 == 0x22;
    bool resolved = true;
    if ((run1 == 4u) | (is_q & (run2 == 3u))) {
        const uint32_t zl = (uint32_t)halo ^ 0x5C5C5C5Cu;                   // bytes -8..-5
        const uint32_t more = zl ? (uint32_t)__builtin_clz(zl) >> 3 : 4u;   // backslashes ending at byte -5 (0..4)
        if (run1 == 4u) run1 += more;  // the run ending at byte -1 goes on
        else run2 += more;             // (byte -1 is the quote) the run ending at byte -2 goes on
        resolved = more != 4u;
    }
    *e_in = run1 & 1u;
    *p_in = run1 ? 1u : (is_q ? (run2 & 1u) : (sj_is_ws_or_op(h1) ^ 1u));
    return resolved;
}

// Slow path (backslash run longer than the halo): parity of the run of backslashes that ends
// right before buf[pos] and does not extend below buf[lo].
SJ_HD uint32_t sj_backslash_run_parity(const uint8_t* buf, sj_u64 lo, sj_u64 pos) {
    // whole aligned 8-byte groups of backslashes do not change the parity: skip them with one load each
    while (pos >= lo + 8 && (pos & 7) == 0 &&
           *reinterpret_cast<const sj_u64*>(buf + pos - 8) == 0x5C5C5C5C5C5C5C5Cull)
        pos -= 8;
    uint32_t par = 0;
    while (pos > lo && buf[pos - 1] == 0x5C) {
        par ^= 1u;
        --pos;
        while (pos >= lo + 8 && (pos & 7) == 0 &&
               *reinterpret_cast<const sj_u64*>(buf + pos - 8) == 0x5C5C5C5C5C5C5C5Cull)
            pos -= 8;
    }
    return par;
}

// does the backslash run that ends right before buf[pos] reach down to buf[lo] (so that it may go on in front of what is
// readable: a shard's / stream chunk's left halo)?
SJ_HD bool sj_backslash_run_reaches(const uint8_t* buf, sj_u64 lo, sj_u64 pos) {
    while (pos > lo && buf[pos - 1] == 0x5C) --pos;
    return pos == lo && lo < (sj_u64)-1 && buf[lo] == 0x5C;
}

SJ_HD void sj_carry_slow(const uint8_t* buf, sj_u64 doc_lo, sj_u64 start, uint32_t* e_in, uint32_t* p_in) {
    const uint32_t h1 = buf[start - 1];
    if (h1 == 0x5C) {
        *e_in = sj_backslash_run_parity(buf, doc_lo, start);
        *p_in = 1;
    } else if (h1 == 0x22) {
        *e_in = 0;
        *p_in = sj_backslash_run_parity(buf, doc_lo, start - 1);
    } else {
        *e_in = 0;
        *p_in = !sj_is_ws_or_op(h1);
    }
}

// ---- bit-plane transposition, butterfly form ------------------------------------------------------
// Per 32-byte half: 2 x (4x4 byte transpose with v_perm_b32) gather byte 8c+r of the half into byte c of register r,
// then ONE 8x8 bit-matrix transpose ACROSS the 8 registers (register index <-> bit index inside the byte, all four
// bytes of a register at once): 3 butterfly stages x 4 register pairs x 6 VOP2 ops.  Register k then IS the half's
// 32 bits of plane k.  176 instructions per block (32 VOP3 + 144 VOP2); the first butterfly version transposed
// 8 bytes at a time inside register pairs (Hacker's Delight 7-3) and needed 256, the v_and/v_msad_u8 form 480 issue
// units (VOP3 ops cost double on gfx950).
SJ_HD uint32_t sj_perm(uint32_t hi, uint32_t lo, uint32_t sel) {  // v_perm_b32 D = bytes of {hi:lo} picked by sel
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const sj_u64 v = ((sj_u64)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= (uint32_t)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xFF) << (8 * i);
    return r;
#endif
}

// one butterfly stage between two registers: x's bits at positions with (bit & D) set <-> y's bits with it clear
template <int D, uint32_t LOW_MASK>
SJ_HD void sj_bit_swap(uint32_t& x, uint32_t& y) {
    const uint32_t t = ((x >> D) ^ y) & LOW_MASK;
    y ^= t;
    x ^= t << D;
}

SJ_HD void sj_transpose4x4_bytes(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t b[4]) {
    const uint32_t t0 = sj_perm(a1, a0, 0x05010400u), t1 = sj_perm(a1, a0, 0x07030602u);
    const uint32_t t2 = sj_perm(a3, a2, 0x05010400u), t3 = sj_perm(a3, a2, 0x07030602u);
    b[0] = sj_perm(t2, t0, 0x05040100u);
    b[1] = sj_perm(t2, t0, 0x07060302u);
    b[2] = sj_perm(t3, t1, 0x05040100u);
    b[3] = sj_perm(t3, t1, 0x07060302u);
}

// one 32-byte half: w8 = its 8 dwords -> x[k] = its 32 bits of plane k
SJ_HD void sj_transpose_half(const uint32_t w8[8], uint32_t x[8]) {
    // x[r] byte c = byte 8c + r of this half
    sj_transpose4x4_bytes(w8[0], w8[2], w8[4], w8[6], x);
    sj_transpose4x4_bytes(w8[1], w8[3], w8[5], w8[7], x + 4);
    for (int r = 0; r < 4; ++r) sj_bit_swap<4, 0x0F0F0F0Fu>(x[r], x[r + 4]);
    for (int r = 0; r < 8; r += 4) {
        sj_bit_swap<2, 0x33333333u>(x[r], x[r + 2]);
        sj_bit_swap<2, 0x33333333u>(x[r + 1], x[r + 3]);
    }
    for (int r = 0; r < 8; r += 2) sj_bit_swap<1, 0x55555555u>(x[r], x[r + 1]);
    // now x[k] bit (8c + r) = bit k of byte 8c + r
}

SJ_HD void sj_transpose_butterfly(const uint32_t w[16], sj_u64 p[8]) {
    uint32_t y[2][8];
    for (int h = 0; h < 2; ++h) sj_transpose_half(w + 8 * h, y[h]);
    for (int k = 0; k < 8; ++k) p[k] = (sj_u64)y[0][k] | ((sj_u64)y[1][k] << 32);
}

// portable transposition (host + reference for the device fast path's self-test)
SJ_HD void sj_transpose_ref(const uint32_t w[16], sj_u64 p[8]) {
    for (int k = 0; k < 8; ++k) p[k] = 0;
    for (int i = 0; i < 64; ++i) {
        const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFF;
        for (int k = 0; k < 8; ++k) p[k] |= (sj_u64)((c >> k) & 1u) << i;
    }
}
