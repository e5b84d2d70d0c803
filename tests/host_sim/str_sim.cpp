// Host simulation of the streaming string pass (simdjson-java_amd/csrc/strings.hip): the per-block algebra of
// sj_strings.h, chained block by block the way the kernel chains lanes, waves and granules (offsets by prefix sum, the
// pending opening quote and the pending error carried forward, headers written by the closing quote's block).
// TEST ONLY: lets the CPU suite check the exact device algebra against the oracle's StringParser without a GPU.
// Built by tests/test_host_strings.py with g++.
#include <string.h>
#include <vector>
#include "../../simdjson-java_amd/csrc/sj_strings.h"

static void put_or(uint8_t* sb, uint64_t cap, uint64_t at, uint32_t bytes, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i)
        if (at + i < cap) sb[at + i] |= (uint8_t)(bytes >> (8 * i));
}
static void put_xor(uint8_t* sb, uint64_t cap, uint64_t at, uint32_t bytes, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i)
        if (at + i < cap) sb[at + i] ^= (uint8_t)(bytes >> (8 * i));
}

// force_u: 0 = the shortcuts as the kernel takes them, 1 = always the full algebra (must give the same bytes)
extern "C" int sim_strings(const uint8_t* buf, uint64_t len, uint8_t* sb, uint64_t cap, uint64_t* total, uint64_t* nstrings,
                           uint64_t* soff, uint64_t soff_cap, uint64_t* first_error, int force_u) {
    memset(sb, 0, cap);
    const uint64_t nblocks = len / 64 + 1;
    uint64_t out = 0, ns = 0;
    uint32_t parity = 0;
    uint64_t pendD = ~0ull;   // absolute offset of the header of the string that is open
    uint32_t pend_err = 0;    // its first error so far
    uint64_t fe = ~0ull;
    for (uint64_t b = 0; b < nblocks; ++b) {
        const uint64_t start = b * 64;
        const uint32_t valid = (uint32_t)(len - start < 64 ? len - start : 64);
        uint32_t w[16];
        memset(w, 0xA5, sizeof w);  // bytes past the end are garbage on the device too
        memcpy(w, buf + start, valid);
        sj_u64 p[8];
        sj_transpose_butterfly(w, p);
        sj_mask_tail(p, valid);
        uint32_t e_in = 0, p_in = 0;
        uint8_t hb[16];
        memset(hb, 0x20, 16);
        if (b > 0) {
            sj_u64 halo8;
            memcpy(&halo8, buf + start - 8, 8);
            if (!sj_carry_from_halo(halo8, &e_in, &p_in)) sj_carry_slow(buf, 0, start, &e_in, &p_in);
            memcpy(hb, buf + start - 16, 16);
        }
        const SjStrBase s = sj_str_base(p, e_in, parity);
        SjStrHalo halo;
        {
            uint32_t w8[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // the halo as the first 16 bytes of a 32-byte half (as the kernel does)
            memcpy(w8, hb, 16);
            sj_transpose_half(w8, halo.hp);
            for (int k = 0; k < 8; ++k) {
                uint32_t ref = 0;
                for (int t = 0; t < 16; ++t) ref |= (uint32_t)((hb[t] >> k) & 1u) << t;
                if ((halo.hp[k] & 0xFFFFu) != ref) return -5;
            }
        }
        halo.e_in = 0;
        if (b > 0 && sj_str_halo_unresolved(halo.hp)) halo.e_in = sj_backslash_run_parity(buf, 0, start - 16);
        bool halo_bs = false;
        for (int t = 4; t < 16; ++t) halo_bs |= hb[t] == 0x5C;
        uint32_t bsq;
        {
            sj_u64 bs, rq;
            sj_str_quote_bs<sj_u64>(p, &bs, &rq);
            bsq = bs != 0;
        }
        const bool do_escapes = force_u || s.ED != 0 || halo_bs;
        const bool do_u = force_u || halo_bs || bsq;  // (the kernel's trigger is sharper; any superset gives the same bytes)
        const SjStrBlock m = sj_str_block(p, s, parity, do_escapes, do_u, &halo);
        const SjStrGroups g = sj_str_groups(m.K, m.O);
        const sj_u64 Kc = m.K & ~(g.bad << 1);
        // ---- copy: kept bytes dword by dword ----
        for (uint32_t i = 0; i < 16; ++i) {
            const sj_u64 lt = (1ull << (4 * i)) - 1ull, grp = 0xFull << (4 * i);
            const uint32_t nib = (uint32_t)(Kc >> (4 * i)) & 15u;
            const uint64_t dest = out + m.head + (uint32_t)__builtin_popcountll(m.K & lt) +
                                  4u * ((uint32_t)__builtin_popcountll(m.O & lt) + (uint32_t)__builtin_popcountll(g.oshift & grp));
            const uint32_t packed = sj_perm(0, w[i], sj_str_pack_selector(nib));
            put_or(sb, cap, dest, packed, (uint32_t)__builtin_popcount(nib));
        }
        for (sj_u64 bad = g.bad; bad; bad &= bad - 1) {
            const uint32_t o = (uint32_t)__builtin_ctzll(bad);  // byte 2 of its dword; byte 3 follows the new header
            put_or(sb, cap, out + sj_str_offset(m, o + 1), w[o >> 2] >> 24, 1);
        }
        // ---- patches ----
        const sj_u64 pm[4] = {m.pn, m.pt, m.pr, m.pbf};
        const uint32_t pd[4] = {0x6E ^ 0x0A, 0x74 ^ 0x09, 0x72 ^ 0x0D, 0x6A};  // b -> 08, f -> 0C: both ^ 0x6A
        for (int q = 0; q < 4; ++q)
            for (sj_u64 x = pm[q]; x; x &= x - 1) put_xor(sb, cap, out + sj_str_offset(m, (uint32_t)__builtin_ctzll(x)), pd[q], 1);
        for (sj_u64 x = m.l1 | m.l2 | m.l3 | m.pair; x; x &= x - 1) {
            const uint32_t e = (uint32_t)__builtin_ctzll(x);
            uint32_t lo;
            memcpy(&lo, buf + start + e - 3, 4);
            uint32_t cp = sj_hex4_valid_word(lo);
            if (cp != (uint32_t)sj_hex4_word(lo)) return -7;  // (the fast form on digits the planes found valid == the checking form)
            if ((m.pair >> e) & 1) {
                uint32_t hi;
                memcpy(&hi, buf + start + e - 9, 4);
                if (sj_hex4_valid_word(hi) != (uint32_t)sj_hex4_word(hi)) return -7;
                cp = (((sj_hex4_valid_word(hi) - 0xD800u) << 10) | (cp - 0xDC00u)) + 0x10000u;
            }
            uint32_t L;
            const uint32_t nb = sj_utf8_bytes(cp, &L);
            const uint32_t want = ((m.l1 >> e) & 1) ? 1u : ((m.l2 >> e) & 1) ? 2u : ((m.l3 >> e) & 1) ? 3u : 4u;
            if (L != want) return -3;  // the planes and the bytes disagree about the length
            uint32_t old = L == 4 ? lo : (lo >> (8 * (4 - L)));
            // slots in front of the block (the previous block dropped those digits, nothing was copied there)
            const uint32_t spilled = L - 1 > e ? L - 1 - e : 0;
            if (spilled) old &= ~0u << (8 * spilled);
            put_xor(sb, cap, out + sj_str_offset(m, e) - (L - 1), old ^ nb, L);
        }
        // ---- headers: written by the closing quote ----
        uint32_t epos = 0;
        for (sj_u64 cl = m.CL; cl; cl &= cl - 1) {
            const uint32_t c = (uint32_t)__builtin_ctzll(cl);
            const sj_u64 lt_c = (1ull << c) - 1ull;
            const sj_u64 le_c = c == 63 ? ~0ull : ((2ull << c) - 1ull);
            uint64_t Do;
            uint32_t err;
            if (m.O & lt_c) {
                const uint32_t o = 63u - (uint32_t)__builtin_clzll(m.O & lt_c);
                Do = out + sj_str_offset(m, o);
                err = sj_str_first_error(m, le_c & ~((2ull << o) - 1ull), &epos);
            } else {
                if (pendD == ~0ull) return -4;  // a closing quote without an opening one: the parity is broken
                Do = pendD;
                err = pend_err ? pend_err : sj_str_first_error(m, le_c, &epos);
            }
            const uint64_t Dc = out + sj_str_offset(m, c);
            const uint32_t n = (uint32_t)(Dc - Do - 4);
            const uint32_t hdr = err ? (0x00FFFFFFu | (err << 24)) : __builtin_bswap32(n);
            put_or(sb, cap, Do, hdr, 4);
        }
        {
            uint32_t pos;
            const uint32_t code = sj_str_first_error(m, ~0ull, &pos);
            if (code) {
                const uint64_t v = ((start + pos) << 8) | code;
                if (v < fe) fe = v;
            }
        }
        for (sj_u64 o = m.O; o; o &= o - 1) {
            if (ns < soff_cap) soff[ns] = out + sj_str_offset(m, (uint32_t)__builtin_ctzll(o));
            ++ns;
        }
        if (m.exit_in) {
            if (m.O) {
                const uint32_t o = 63u - (uint32_t)__builtin_clzll(m.O);
                pendD = out + sj_str_offset(m, o);
                pend_err = sj_str_first_error(m, o == 63 ? 0ull : ~((2ull << o) - 1ull), &epos);
            } else if (!pend_err) {
                pend_err = sj_str_first_error(m, ~0ull, &epos);
            }
        } else {
            pendD = ~0ull;
            pend_err = 0;
        }
        out += sj_str_out_bytes(m);
        parity = m.exit_in;
    }
    *total = out;
    *nstrings = ns;
    *first_error = fe;
    return parity ? 1 : 0;
}
