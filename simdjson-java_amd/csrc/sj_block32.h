// sj_block32.h -- the per-block stage-1 algebra of sj_block.h written for the FEWEST VALU instructions on gfx950.
//
// k_stage1 is bound by VALU issue (one wave64 instruction per ~4.2 cycles per SIMD in its instruction mix), so what counts is
// the NUMBER of instructions.  Against sj_block.h (which hipcc lowers 64-bit operation by 64-bit operation):
//   * every 64-bit mask is a {lo, hi} pair of 32-bit registers, and every three-input boolean function is ONE explicit
//     v_bitop3_b32 (the instruction selector forms them only where two-input operations happen to line up);
//   * the transposition is two byte stages of v_perm_b32 (as before) and three bit stages of {shift, shift, v_bitop3, v_bitop3}
//     per register pair -- 4 instead of 5-6 instructions: 128 per block instead of 152;
//   * shifts across the halves are v_alignbit_b32 / v_lshl_or_b32, the one addition and the prefix XOR stay 64-bit
//     (v_add_co / v_addc, v_lshlrev_b64 + two v_xor per stage), the second-byte checks of UTF-8 shift the two CONDITION planes
//     right instead of four lead masks left.
// What was measured on the way (tools/microbench/valu_rates.hip, valu_mix.hip, profiles/r4/valu_*.jsonl; tools/ab_stage1.sh):
// gfx950 has a FAST class of VALU instructions -- v_and / v_or / v_xor / v_not / v_add_u32 / v_sub_u32 / v_mov_b32 / shifts by a
// constant / v_bitop3_b32, with VGPR or literal operands -- that a SIMD retires every ~2.1 cycles when two or more waves issue
// them, against ~4.3 for everything else (v_perm, v_bfi, v_or3, v_and_or, v_lshl_or, v_alignbit, v_bfe, 64-bit shifts and
// adds, DPP, v_cmp, v_bcnt, v_readlane, any SGPR operand; LDS and memory instructions count too).  But one slow instruction
// among sixteen fast ones already costs the fast ones half their advantage (2.1 -> 3.0 cycles; 1 in 8: 3.6), for EVERY wave on
// the SIMD.  A first version of this file used fast-class instructions only (five butterfly stages, 32-bit shifts spelt out,
// masks in vector registers: 85 % of the classification fast-class): +3 % -- and replacing its fast sequences by FEWER slow
// instructions (this version) gained another 8 %.  Synchronising the phases of the sixteen waves of a CU (1024-thread
// workgroups, barriers between classification and expansion) so that fast code meets fast code: -7 % to -13 %.  The fast
// class is out of reach for a kernel that also loads, scans and scatters; the count is what pays.
// Results are bit-identical to sj_block (tests/host_sim/block32.cpp fuzzes one against the other; the kernels' parity tests
// cover the device code).  Reference lines as in sj_block.h.
#pragma once
#include "sj_block.h"

struct SjPair {  // a 64-bit mask: bytes 0..31 of the block in lo, 32..63 in hi
    uint32_t lo, hi;
};

// f(a, b, c) by truth table, a = 0xF0, b = 0xCC, c = 0xAA (v_bitop3_b32)
template <uint32_t TT>
SJ_HD uint32_t sj_bop(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i)
        if ((TT >> i) & 1u) r |= ((i & 4) ? a : ~a) & ((i & 2) ? b : ~b) & ((i & 1) ? c : ~c);
    return r;
#endif
}
enum : uint32_t {
    SJ_TT_MUX_C = 0xE4,      // c ? a : b
    SJ_TT_OR3 = 0xFE,        // a | b | c
    SJ_TT_A_NB_NC = 0x10,    // a & ~b & ~c
    SJ_TT_A_B_NC = 0x40,     // a & b & ~c
    SJ_TT_A_NB_C = 0x20,     // a & ~b & c
    SJ_TT_A_B_C = 0x80,      // a & b & c
    SJ_TT_NOR3 = 0x01,       // ~(a | b | c)
    SJ_TT_A_OR_BC = 0xF8,    // a | (b & c)
    SJ_TT_A_AND_BORC = 0xE0, // a & (b | c)
    SJ_TT_A_OR_NBORC = 0xF1, // a | ~(b | c)
    SJ_TT_A_OR_BXC = 0xF6,   // a | (b ^ c)
    SJ_TT_AB_OR_C = 0xEA,    // (a & b) | c
    SJ_TT_ANB_OR_C = 0xBA,   // (a & ~b) | c
    SJ_TT_A_N_BC = 0x70,     // a & ~(b & c)
    SJ_TT_ANC_OR_BC = 0xD8,  // (a & ~c) | (b & c)
};

// one butterfly stage between two registers: afterwards x holds the elements whose position bit (of weight S) was clear, y
// those whose bit was set, and the position bit tells which register an element came from.  m = positions with the bit clear
// (a scalar register: v_bitop3_b32 takes no literal).
template <int S>
SJ_HD void sj_butterfly(uint32_t& x, uint32_t& y, uint32_t m) {
    const uint32_t xs = x >> S, ys = y << S;
    const uint32_t nx = sj_bop<SJ_TT_MUX_C>(x, ys, m);
    y = sj_bop<SJ_TT_MUX_C>(xs, y, m);
    x = nx;
}
// one 32-byte half: w8 = its 8 dwords -> x[k] = its 32 bits of plane k (bit j = bit k of byte j).  16 v_perm_b32 gather byte
// 8c + r of the half into byte c of register r (sj_transpose4x4_bytes), three butterfly stages exchange the register index with
// the bit index inside the bytes: 64 instructions.
SJ_HD void sj_transpose_half32_bytes(const uint32_t w8[8], uint32_t x[8]) {
    sj_transpose4x4_bytes(w8[0], w8[2], w8[4], w8[6], x);
    sj_transpose4x4_bytes(w8[1], w8[3], w8[5], w8[7], x + 4);
}
SJ_HD void sj_transpose_half32_bits(uint32_t x[8]) {
    for (int r = 0; r < 4; ++r) sj_butterfly<4>(x[r], x[r + 4], 0x0F0F0F0Fu);
    for (int r = 0; r < 8; r += 4) {
        sj_butterfly<2>(x[r], x[r + 2], 0x33333333u);
        sj_butterfly<2>(x[r + 1], x[r + 3], 0x33333333u);
    }
    for (int r = 0; r < 8; r += 2) sj_butterfly<1>(x[r], x[r + 1], 0x55555555u);
}
SJ_HD void sj_transpose_half32(const uint32_t w8[8], uint32_t x[8]) {
    sj_transpose_half32_bytes(w8, x);
    sj_transpose_half32_bits(x);
}

SJ_HD void sj_transpose32(const uint32_t w[16], uint32_t lo[8], uint32_t hi[8]) {
    sj_transpose_half32(w, lo);
    sj_transpose_half32(w + 8, hi);
}

// ({hi, lo} >> n)[31:0], 0 < n < 32: v_alignbit_b32
SJ_HD uint32_t sj_funnel(uint32_t hi, uint32_t lo, uint32_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, n);
#else
    return (uint32_t)(((((sj_u64)hi) << 32) | lo) >> n);
#endif
}
// (x << 1) | cin, cin = 0 / 1
SJ_HD SjPair sj_shl1(SjPair x, uint32_t cin) {
    SjPair r;
    r.lo = (x.lo << 1) | cin;
    r.hi = sj_funnel(x.hi, x.lo, 31);
    return r;
}
// (x << 1) | y
SJ_HD SjPair sj_shl1_or(SjPair x, SjPair y) {
    SjPair r;
    r.lo = (x.lo << 1) | y.lo;
    r.hi = sj_funnel(x.hi, x.lo, 31) | y.hi;
    return r;
}
SJ_HD SjPair sj_shr1(SjPair x) {
    SjPair r;
    r.lo = sj_funnel(x.hi, x.lo, 1);
    r.hi = x.hi >> 1;
    return r;
}
// prefix XOR over the 64 bits (StructuralIndexer.java:311-319)
SJ_HD SjPair sj_prefix_xor32(SjPair m) {
    sj_u64 v = ((sj_u64)m.hi << 32) | m.lo;
    v ^= v << 1;
    v ^= v << 2;
    v ^= v << 4;
    v ^= v << 8;
    v ^= v << 16;
    v ^= v << 32;
    SjPair r = {(uint32_t)v, (uint32_t)(v >> 32)};
    return r;
}

struct SjBlockMasks32 {
    SjPair pot, sm0;  // as SjBlockMasks
    uint32_t qpar;    // 0 / 1
    uint32_t ue0, ue1, utf8;  // NONZERO = set (not normalised to 0 / 1: the consumer ballots / ORs them)
    uint32_t words;   // want_words, as SjBlockMasks
};

// tail of the document: bytes at index >= valid read as spaces (sj_mask_tail)
SJ_HD void sj_mask_tail32(uint32_t lo[8], uint32_t hi[8], uint32_t valid) {
    if (valid >= 64) return;
    const uint32_t vl = valid >= 32 ? 0xFFFFFFFFu : (1u << valid) - 1u;
    const uint32_t vh = valid <= 32 ? 0u : (1u << (valid - 32)) - 1u;
    for (int k = 0; k < 8; ++k) {
        lo[k] &= vl;
        hi[k] &= vh;
    }
    lo[5] |= ~vl;
    hi[5] |= ~vh;
}

// per-half character classes (StructuralIndexer.java:210,231-232,237-240)
struct SjHalfClasses {
    uint32_t bs, rawquote, ctrl, ws, op;
    uint32_t n_0000, n_1101, n00;  // low-nibble decodes the UTF-8 algebra uses again
};
SJ_HD SjHalfClasses sj_classes32(const uint32_t p[8]) {
    const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4], p5 = p[5], p6 = p[6], p7 = p[7];
    SjHalfClasses c;
    const uint32_t a = ~(p7 | p6);  // 0x00..0x3F
    const uint32_t b = p6 & ~p7;    // 0x40..0x7F
    const uint32_t n00 = ~(p3 | p2), n11 = p3 & p2, n10 = p3 & ~p2;
    const uint32_t n_0000 = sj_bop<SJ_TT_A_NB_NC>(n00, p1, p0);
    const uint32_t n_0010 = sj_bop<SJ_TT_A_B_NC>(n00, p1, p0);
    const uint32_t n_1100 = sj_bop<SJ_TT_A_NB_NC>(n11, p1, p0);
    const uint32_t n_1101 = sj_bop<SJ_TT_A_NB_C>(n11, p1, p0);
    const uint32_t n_1001 = sj_bop<SJ_TT_A_NB_C>(n10, p1, p0);
    const uint32_t n_1010 = sj_bop<SJ_TT_A_B_NC>(n10, p1, p0);
    const uint32_t n_1011 = sj_bop<SJ_TT_A_B_C>(n10, p1, p0);
    c.bs = sj_bop<SJ_TT_A_NB_C>(b, p5, p4) & n_1100;               // '\\' 0x5C
    c.rawquote = sj_bop<SJ_TT_A_B_NC>(a, p5, p4) & n_0010;         // '"'  0x22
    c.ctrl = a & ~p5;                                              // <= 0x1F
    const uint32_t wsl = sj_bop<SJ_TT_OR3>(n_1001, n_1010, n_1101);
    c.ws = sj_bop<SJ_TT_A_NB_C>(a, p4, sj_bop<SJ_TT_MUX_C>(n_0000, wsl, p5));     // {0x20, 0x09, 0x0A, 0x0D}
    const uint32_t opa = sj_bop<SJ_TT_MUX_C>(n_1010, n_1100, p4);                  // a: p4 ? 1010 : 1100
    const uint32_t opb = sj_bop<SJ_TT_A_AND_BORC>(p4, n_1011, n_1101);             // b: p4 & (1011 | 1101)
    c.op = sj_bop<SJ_TT_MUX_C>(opa, opb, a) & ~p7;  // (a & opa) | (b & opb): a, b disjoint, a | b = ~p7
    c.n_0000 = n_0000;
    c.n_1101 = n_1101;
    c.n00 = n00;
    return c;
}


// The UTF-8 carries as the kernel wants them: the three "must be a continuation" bits merged (sj_block ORs them anyway), and of the
// lead in byte -1 only whether it is one of the FOUR whose second byte has a restricted range -- 0xE0, 0xED, 0xF0, 0xF4: in
// twitter.json 13 bytes of 631,515, so that whole wave-steps go without the second-byte checks (sj_block32's `any`).
struct SjUtf8Lazy {
    uint32_t c123;     // c1 | c2 | c3 of SjUtf8Carry
    uint32_t special;  // byte -1 if it is 0xE0 / 0xED / 0xF0 / 0xF4, else 0
};
SJ_HD uint32_t sj_utf8_special_lead(uint32_t c) {  // c = a byte value: is it one of the four?
    const uint32_t M = (1u << 0x00) | (1u << 0x0D) | (1u << 0x10) | (1u << 0x14);  // by c - 0xE0
    return (uint32_t)(c >= 0xE0u) & ((M >> (c & 31u)) & 1u);
}
SJ_HD SjUtf8Lazy sj_utf8_carry_lazy(sj_u64 halo) {
    const uint32_t hh = (uint32_t)(halo >> 32);  // bytes -4..-1; only bit 7 of each byte is meaningful below
    const uint32_t geC0 = hh & (hh << 1);        // byte >= 0xC0: a 2/3/4-byte lead
    const uint32_t geE0 = geC0 & (hh << 2);      // byte >= 0xE0: a 3/4-byte lead
    const uint32_t geF0 = geE0 & (hh << 3);      // byte >= 0xF0: a 4-byte lead
    SjUtf8Lazy c;
    // byte 0 continues a lead in byte -1, a 3/4-byte lead in byte -2 or a 4-byte lead in byte -3; byte 1 a 3/4-byte lead in byte -1
    // or a 4-byte lead in byte -2; byte 2 a 4-byte lead in byte -1
    c.c123 = ((geC0 >> 31) | ((geE0 >> 23) & 1u) | ((geF0 >> 15) & 1u)) | (((geE0 >> 30) | (geF0 >> 22)) & 2u) | ((geF0 >> 29) & 4u);
    const uint32_t h1 = hh >> 24;
    c.special = sj_utf8_special_lead(h1) ? h1 : 0u;
    return c;
}
SJ_HD SjUtf8Lazy sj_utf8_lazy_of(SjUtf8Carry uc) {  // (tests, the slow path: from the full form)
    SjUtf8Lazy c;
    c.c123 = uc.c1 | uc.c2 | uc.c3;
    c.special = (uc.sec & 1u) ? 0xE0u : (uc.sec & 2u) ? 0xEDu : (uc.sec & 4u) ? 0xF0u : (uc.sec & 8u) ? 0xF4u : 0u;
    return c;
}

// sj_block with {lo, hi} planes
// any(x): is x nonzero in ANY block the caller processes together (a ballot over the wave in the kernels; x != 0 for one block)
template <class ANY>
SJ_HD SjBlockMasks32 sj_block32(const uint32_t lo[8], const uint32_t hi[8], uint32_t e_in, uint32_t p_in, SjUtf8Lazy uc,
                                bool do_utf8, bool want_words, ANY&& any) {
    const SjHalfClasses cl = sj_classes32(lo), ch = sj_classes32(hi);
    // ---- escapes (:211-229) ----
    SjPair bs = {cl.bs & ~e_in, ch.bs};
    const SjPair fe = sj_shl1(bs, e_in);  // follows_escape
    const uint32_t ODD = 0xAAAAAAAAu, EVEN = 0x55555555u;
    const SjPair os = {(bs.lo & ODD) & ~fe.lo, (bs.hi & ODD) & ~fe.hi};  // odd_starts
    const sj_u64 sum = (((sj_u64)os.hi << 32) | os.lo) + (((sj_u64)bs.hi << 32) | bs.lo);  // seq_even = odd_starts + bs
    const SjPair se = {(uint32_t)sum, (uint32_t)(sum >> 32)};
    const SjPair se1 = sj_shl1(se, 0);
    const SjPair escaped = {(EVEN ^ se1.lo) & fe.lo, (EVEN ^ se1.hi) & fe.hi};
    // ---- strings (:232-234) ----
    const SjPair quote = {cl.rawquote & ~escaped.lo, ch.rawquote & ~escaped.hi};
    const SjPair in0 = sj_prefix_xor32(quote);
    // ---- scalars / structural starts (:243-248): pot = op | (scalar & ~follows_nqs) = op | ~(ws | follows_nqs) ----
    const SjPair nqs = {sj_bop<SJ_TT_NOR3>(cl.op, cl.ws, quote.lo), sj_bop<SJ_TT_NOR3>(ch.op, ch.ws, quote.hi)};
    const SjPair fn = sj_shl1(nqs, p_in);
    SjBlockMasks32 r;
    r.pot.lo = sj_bop<SJ_TT_A_OR_NBORC>(cl.op, cl.ws, fn.lo);
    r.pot.hi = sj_bop<SJ_TT_A_OR_NBORC>(ch.op, ch.ws, fn.hi);
    r.sm0.lo = in0.lo ^ quote.lo;
    r.sm0.hi = in0.hi ^ quote.hi;
    r.qpar = in0.hi >> 31;
    r.ue0 = sj_bop<SJ_TT_AB_OR_C>(cl.ctrl, in0.lo, ch.ctrl & in0.hi);
    r.ue1 = sj_bop<SJ_TT_ANB_OR_C>(cl.ctrl, in0.lo, ch.ctrl & ~in0.hi);
    // ---- UTF-8 (Utf8Validator.java:54-168) ----
    uint32_t err = 0;
    if (do_utf8) {
        const uint32_t* P[2] = {lo, hi};
        const SjHalfClasses* C[2] = {&cl, &ch};
        uint32_t cont[2], L2[2], L3[2], L4[2], L34[2], L234[2], e[2], l4a[2];
        for (int h = 0; h < 2; ++h) {
            const uint32_t* p = P[h];
            const uint32_t lead = p[7] & p[6];
            cont[h] = p[7] & ~p[6];
            L2[h] = lead & ~p[5];
            const uint32_t l5 = lead & p[5];
            L3[h] = l5 & ~p[4];
            L4[h] = sj_bop<SJ_TT_A_B_NC>(l5, p[4], p[3]);
            e[h] = sj_bop<SJ_TT_A_B_C>(l5, p[4], p[3]);            // 0xF8..0xFF
            L34[h] = sj_bop<SJ_TT_A_N_BC>(l5, p[4], p[3]);
            L234[h] = lead ^ e[h];
            const uint32_t ovl2 = sj_bop<SJ_TT_A_NB_NC>(L2[h], p[4], p[1]);   // 0xC0, 0xC1 (OVERLONG_2BYTE) with n00
            e[h] = sj_bop<SJ_TT_A_OR_BC>(e[h], ovl2, C[h]->n00);
            const uint32_t big = sj_bop<SJ_TT_A_AND_BORC>(p[2], p[1], p[0]);  // 0xF5..0xF7 (TOO_LARGE) with L4
            e[h] = sj_bop<SJ_TT_A_OR_BC>(e[h], L4[h], big);
            // second-byte range checks, looked at from the LEAD: the condition planes are shifted right instead of four lead
            // masks left (bit i of q5 = p5 of byte i + 1); the lead in byte 63 is checked by the next block (uc.sec)
            l4a[h] = sj_bop<SJ_TT_A_NB_NC>(p[2], p[1], p[0]);  // low nibble x100 (with L4: 0xF4)
        }
        // expected continuations: E = (L4 << 3) | (L34 << 2) | (L234 << 1) | carries = (((L4 << 1 | L34) << 1) | L234) << 1 | carries
        const SjPair l4 = {L4[0], L4[1]}, l34 = {L34[0], L34[1]}, l234 = {L234[0], L234[1]};
        const SjPair E = sj_shl1(sj_shl1_or(sj_shl1_or(l4, l34), l234), 0);
        e[0] = sj_bop<SJ_TT_A_OR_BXC>(e[0], cont[0], E.lo | uc.c123);  // TOO_SHORT / TOO_LONG / TWO_CONTINUATIONS
        e[1] = sj_bop<SJ_TT_A_OR_BXC>(e[1], cont[1], E.hi);
        err = e[0] | e[1];
        // the leads whose second byte has a restricted range: 0xE0 (L3, low nibble 0000), 0xED (L3, 1101), 0xF0 (L4, 0000),
        // 0xF4 (L4, x100) -- in this block or in front of it?  Mostly not (see SjUtf8Lazy): a wave-uniform skip
        const uint32_t sp0 = sj_bop<SJ_TT_A_AND_BORC>(L3[0], cl.n_0000, cl.n_1101) | sj_bop<SJ_TT_A_AND_BORC>(L4[0], cl.n_0000, l4a[0]);
        const uint32_t sp1 = sj_bop<SJ_TT_A_AND_BORC>(L3[1], ch.n_0000, ch.n_1101) | sj_bop<SJ_TT_A_AND_BORC>(L4[1], ch.n_0000, l4a[1]);
        if (any(sj_bop<SJ_TT_OR3>(sp0, sp1, uc.special))) {
        const SjPair p5 = {lo[5], hi[5]}, p4 = {lo[4], hi[4]};
        const SjPair q5 = sj_shr1(p5), q4 = sj_shr1(p4);
        const uint32_t Q5[2] = {q5.lo, q5.hi}, Q4[2] = {q4.lo, q4.hi};
        for (int h = 0; h < 2; ++h) {
            // E0 80..9F (OVERLONG_3BYTE), ED A0..BF (SURROGATE): L3 & (n_0000 ? ~q5 : n_1101 ? q5 : 0)
            const uint32_t w1 = sj_bop<SJ_TT_ANC_OR_BC>(C[h]->n_0000, C[h]->n_1101, Q5[h]);
            uint32_t part = L3[h] & w1;
            // F0 80..8F (OVERLONG_4BYTE): n_0000 & ~q5 & ~q4;  F4 90..BF (TOO_LARGE): x100 & (q5 | q4)
            const uint32_t w2 = sj_bop<SJ_TT_A_NB_NC>(C[h]->n_0000, Q5[h], Q4[h]);
            const uint32_t w3 = sj_bop<SJ_TT_A_AND_BORC>(l4a[h], Q5[h], Q4[h]);
            part = sj_bop<SJ_TT_OR3>(part, L4[h] & w2, L4[h] & w3);
            if (h == 1) part &= 0x7FFFFFFFu;
            e[h] |= part;
        }
        // ... and the lead in byte -1 against this block's byte 0
        const uint32_t b5 = lo[5] & 1u, b4 = lo[4] & 1u;
        const uint32_t k0 = uc.special == 0xE0u, k1 = uc.special == 0xEDu, k2 = uc.special == 0xF0u, k3 = uc.special == 0xF4u;
        const uint32_t first = (k0 & ~b5) | (k1 & b5) | (k2 & ~b5 & ~b4) | (k3 & (b5 | b4));
        err = e[0] | e[1] | first;
        }
    }
    r.utf8 = err;
    r.words = 0;
    if (want_words) {
        uint32_t w0 = 0, w1 = 0;
        const uint32_t* P[2] = {lo, hi};
        const uint32_t POT[2] = {r.pot.lo, r.pot.hi}, SM[2] = {r.sm0.lo, r.sm0.hi};
        for (int h = 0; h < 2; ++h) {
            const uint32_t* p = P[h];
            const uint32_t a5 = sj_bop<SJ_TT_NOR3>(p[7], p[6], ~p[5]);                         // 0x20..0x3F
            const uint32_t x1100 = sj_bop<SJ_TT_A_B_NC>(p[3], p[2], p[1]);                     // low nibble 110x
            const uint32_t cc = a5 & ((~p[4] & x1100 & ~p[0]) | (p[4] & p[3] & ~p[2] & p[1] & ~p[0]));  // 0x2C, 0x3A
            const uint32_t num = a5 & ((p[4] & (~p[3] | (~p[2] & ~p[1]))) | (~p[4] & x1100 & p[0]));     // 0x30..0x39, 0x2D
            const uint32_t s0 = POT[h] & ~SM[h], s1 = POT[h] & SM[h];
#if defined(__HIP_DEVICE_COMPILE__)
            w0 += (uint32_t)__popc(s0 & ~cc) + (uint32_t)__popc(s0 & num);
            w1 += (uint32_t)__popc(s1 & ~cc) + (uint32_t)__popc(s1 & num);
#else
            w0 += (uint32_t)__builtin_popcount(s0 & ~cc) + (uint32_t)__builtin_popcount(s0 & num);
            w1 += (uint32_t)__builtin_popcount(s1 & ~cc) + (uint32_t)__builtin_popcount(s1 & num);
#endif
        }
        r.words = w0 | (w1 << 8);
    }
    return r;
}

// ... one block on its own, carries in the full form (tests, the slow path of the kernels)
SJ_HD SjBlockMasks32 sj_block32(const uint32_t lo[8], const uint32_t hi[8], uint32_t e_in, uint32_t p_in, SjUtf8Carry uc,
                                bool do_utf8 = true, bool want_words = false) {
    return sj_block32(lo, hi, e_in, p_in, sj_utf8_lazy_of(uc), do_utf8, want_words, [](uint32_t x) { return x != 0; });
}
