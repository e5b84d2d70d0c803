// Host simulation of the GPU stage-1 algorithm's per-block algebra (sj_block.h), chained
// sequentially.  TEST ONLY: lets the CPU test-suite fuzz the exact device algebra (plane
// classification, halo carries, UTF-8 planes, parity flip) against the oracle without a GPU.
// Built by tests/test_host_sim.py with g++.
#include <string.h>
#include <vector>
#include "../../simdjson-java_amd/csrc/sj_block.h"

extern "C" int sim_stage1(const uint8_t* buf, uint64_t len, uint32_t* idx, uint64_t cap, uint64_t* count,
                          uint32_t* status) {
    const uint64_t nblocks = len / 64 + 1;
    uint64_t n = 0;
    uint32_t parity = 0, st = 0;
    for (uint64_t b = 0; b < nblocks; ++b) {
        const uint64_t start = b * 64;
        const uint32_t valid = (uint32_t)(len - start < 64 ? len - start : 64);
        uint32_t w[16];
        memset(w, 0, sizeof w);
        memcpy(w, buf + start, valid);  // device loads all 64 and masks; garbage beyond valid is masked too
        sj_u64 p[8];
        sj_transpose_butterfly(w, p);
        {
            sj_u64 q[8];
            sj_transpose_ref(w, q);
            for (int k = 0; k < 8; ++k)
                if (p[k] != q[k]) return -2;  // butterfly transposition disagrees with the bit loop
        }
        sj_mask_tail(p, valid);
        uint32_t e_in = 0, p_in = 0;
        SjUtf8Carry uc = {0, 0, 0, 0};
        if (b > 0) {
            sj_u64 halo;
            memcpy(&halo, buf + start - 8, 8);
            uc = sj_utf8_carry(halo);
            if (!sj_carry_from_halo(halo, &e_in, &p_in)) sj_carry_slow(buf, 0, start, &e_in, &p_in);
        }
        SjBlockMasks m = sj_block(p, e_in, p_in, uc);
        const sj_u64 s = parity ? (m.pot & m.sm0) : (m.pot & ~m.sm0);
        if (parity ? m.ue1 : m.ue0) st |= 4;
        if (m.utf8) st |= 1;
        parity ^= m.qpar;
        sj_u64 bits = s;
        while (bits) {
            if (n >= cap) return -1;
            idx[n++] = (uint32_t)(start + __builtin_ctzll(bits));
            bits &= bits - 1;
        }
    }
    if (parity) st |= 2;
    if (n >= cap) return -1;
    idx[n] = 0;
    *count = n;
    *status = st;
    return 0;
}

// The reference's six per-block masks, rebuilt exactly as csrc/masks.hip does on the device: the kernel-side block
// algebra for an incoming parity of 0 (sj_block) + the XOR-prefix of the block quote parities + sj_reference_masks.
extern "C" int sim_masks(const uint8_t* buf, uint64_t len, uint64_t* masks) {
    const uint64_t nblocks = len / 64 + 1;
    uint32_t parity = 0;
    for (uint64_t b = 0; b < nblocks; ++b) {
        const uint64_t start = b * 64;
        const uint32_t valid = (uint32_t)(len - start < 64 ? len - start : 64);
        uint32_t w[16];
        memset(w, 0xA5, sizeof w);  // bytes past the end are garbage on the device too: they must be invisible
        memcpy(w, buf + start, valid);
        sj_u64 p[8];
        sj_transpose_butterfly(w, p);
        sj_mask_tail(p, valid);
        uint32_t e_in = 0, p_in = 0;
        SjUtf8Carry uc = {0, 0, 0, 0};
        if (b > 0) {
            sj_u64 halo;
            memcpy(&halo, buf + start - 8, 8);
            uc = sj_utf8_carry(halo);
            if (!sj_carry_from_halo(halo, &e_in, &p_in)) sj_carry_slow(buf, 0, start, &e_in, &p_in);
        }
        SjBlockDetail det;
        const SjBlockMasks m = sj_block(p, e_in, p_in, uc, true, &det);
        sj_u64 out[6];
        sj_reference_masks(m, det, parity, out);
        for (int k = 0; k < 6; ++k) masks[6 * b + k] = out[k];
        parity ^= m.qpar;
    }
    return 0;
}
