// Host fuzz of sj_block32.h (the fast-VALU-class formulation of the block algebra) against sj_block.h: transposition vs the
// bit loop, every output mask of sj_block32 vs sj_block on biased random blocks with every combination of carries, tails of
// every length.  TEST ONLY.  Built and run by tests/test_host_sim.py with g++.  returns 0 or the number of the first check
// that failed.
#include <string.h>
#include <stdint.h>
#include "../../simdjson-java_amd/csrc/sj_block32.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t x = rng_state;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    rng_state = x;
    return x * 0x2545F4914F6CDD1Dull;
}

extern "C" int block32_fuzz(uint64_t seed, uint64_t rounds, uint64_t* fails_at) {
    rng_state = seed * 2 + 1;
    static const uint8_t interesting[] = {'\\', '"', ' ', '\t', '\n', '\r', ',', ':', '[', ']', '{', '}', 0x0C, 0x1A, 0x1F, 0x00, 0x7F,
                                          0x80, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1, 0xED, 0xEE, 0xEF, 0xF0, 0xF1, 0xF4, 0xF5, 0xF8,
                                          0xFF, 0x9F, 0xA0, 0x8F, 0x90, 'a', '1', '-', '0', '9', 't', 'n'};
    for (uint64_t it = 0; it < rounds; ++it) {
        uint8_t blk[64];
        const uint32_t mode = (uint32_t)(rnd() % 4);
        for (int i = 0; i < 64; ++i) {
            const uint64_t r = rnd();
            if (mode == 0) blk[i] = (uint8_t)r;
            else if (mode == 1) blk[i] = interesting[r % sizeof interesting];
            else if (mode == 2) blk[i] = (r & 3) ? interesting[(r >> 8) % 14] : (uint8_t)('a' + (r >> 8) % 26);
            else blk[i] = (r & 1) ? '\\' : ((r & 6) ? '"' : interesting[(r >> 8) % sizeof interesting]);
        }
        uint32_t w[16];
        memcpy(w, blk, 64);
        sj_u64 p[8], q[8];
        uint32_t lo[8], hi[8];
        sj_transpose_ref(w, p);
        sj_transpose32(w, lo, hi);
        for (int k = 0; k < 8; ++k)
            if ((((sj_u64)hi[k] << 32) | lo[k]) != p[k]) { *fails_at = it; return 1; }
        sj_transpose_butterfly(w, q);
        for (int k = 0; k < 8; ++k)
            if (q[k] != p[k]) { *fails_at = it; return 2; }
        const uint32_t valid = (rnd() % 4) ? 64u : (uint32_t)(rnd() % 65);
        sj_mask_tail(p, valid);
        sj_mask_tail32(lo, hi, valid);
        for (int k = 0; k < 8; ++k)
            if ((((sj_u64)hi[k] << 32) | lo[k]) != p[k]) { *fails_at = it; return 3; }
        const uint32_t e_in = (uint32_t)(rnd() & 1), p_in = (uint32_t)(rnd() & 1);
        SjUtf8Carry uc = {0, 0, 0, 0};
        if (rnd() & 1) {  // any halo, through the real carry function
            sj_u64 halo = rnd();
            if (rnd() & 1) {
                uint8_t hb[8];
                for (int i = 0; i < 8; ++i) hb[i] = interesting[rnd() % sizeof interesting];
                memcpy(&halo, hb, 8);
            }
            uc = sj_utf8_carry(halo);
            const SjUtf8Lazy a = sj_utf8_carry_lazy(halo), b = sj_utf8_lazy_of(uc);
            if (a.c123 != b.c123 || a.special != b.special) { *fails_at = it; return 12; }
        }
        const bool words = (rnd() & 1) != 0;
        const SjBlockMasks a = sj_block(p, e_in, p_in, uc, true, nullptr, words);
        const SjBlockMasks32 b = sj_block32(lo, hi, e_in, p_in, uc, true, words);
        if (a.pot != (((sj_u64)b.pot.hi << 32) | b.pot.lo)) { *fails_at = it; return 4; }
        if (a.sm0 != (((sj_u64)b.sm0.hi << 32) | b.sm0.lo)) { *fails_at = it; return 5; }
        if (a.qpar != b.qpar) { *fails_at = it; return 6; }
        if (a.ue0 != (uint32_t)(b.ue0 != 0)) { *fails_at = it; return 7; }
        if (a.ue1 != (uint32_t)(b.ue1 != 0)) { *fails_at = it; return 8; }
        if (a.utf8 != (uint32_t)(b.utf8 != 0)) { *fails_at = it; return 9; }
        if (words && a.words != b.words) { *fails_at = it; return 10; }
        // the ASCII shortcut: same results without the UTF-8 algebra when the block is ASCII and nothing is pending
        bool ascii = (uc.c1 | uc.c2 | uc.c3 | uc.sec) == 0;
        for (int i = 0; i < 64 && ascii; ++i) ascii = i >= (int)valid || blk[i] < 0x80;
        if (ascii) {
            const SjBlockMasks32 c = sj_block32(lo, hi, e_in, p_in, uc, false, words);
            if (c.pot.lo != b.pot.lo || c.pot.hi != b.pot.hi || c.sm0.lo != b.sm0.lo || c.sm0.hi != b.sm0.hi || c.qpar != b.qpar ||
                (c.ue0 != 0) != (b.ue0 != 0) || (c.ue1 != 0) != (b.ue1 != 0) || c.utf8 != 0 || b.utf8 != 0) { *fails_at = it; return 11; }
        }
    }
    return 0;
}
