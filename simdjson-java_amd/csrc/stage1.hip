// stage1.hip -- MI355X (gfx950) stage-1 kernel: fused UTF-8 validation + structural indexing +
// index compaction in ONE pass over the document.
//
// Replaces SimdJsonParser.stage1 = Utf8Validator.validate + StructuralIndexer.index + BitIndexes.write
// (/root/reference/src/main/java/org/simdjson/SimdJsonParser.java:55-58, Utf8Validator.java:54-168,
//  StructuralIndexer.java:196-303, BitIndexes.java:14-41,82-96).
//
// Mapping to the hardware:
//   * one lane  = one 64-byte block (the reference's loop step), loaded as 4 x dwordx4;
//   * the block is transposed to 8 bit planes with v_and + v_msad_u8 (4 mask bits per op), and all
//     classification / escape / string / UTF-8 logic is 64-bit boolean algebra in VGPRs (sj_block.h);
//   * the three serial carries of the reference loop (prevEscaped, prevScalar, previous 4 UTF-8
//     bytes) are LOCAL: each lane re-derives them from the 8 bytes before its block;
//   * the two truly global carries -- in-string parity (XOR scan) and the output offset (+ scan of
//     popcounts) -- are resolved inside the wave by ballot/shuffle, inside the workgroup through
//     LDS, and across workgroups by a single-pass DECOUPLED LOOK-BACK over 8-byte {state,payload}
//     granules (agent-scope relaxed atomics, one granule per tile), so the input is read from HBM
//     exactly once.  Tiles are handed out by an atomic ticket so that a tile only ever waits for
//     tiles whose workgroups have already started (no dependence on dispatch order);
//   * structurals depend on the incoming parity only through a complement
//     (structurals(p) = p ? pot & sm : pot & ~sm), so each tile publishes counts for BOTH parities
//     and the look-back composes functions {0,1} -> (parity, count).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sj_block.h"
#include "stage1.h"

namespace sjmi {

// ---------------------------------------------------------------------------------------------
// bit-plane transposition: 16 dwords (64 bytes) -> 8 x u64
// v_msad_u8 D = S2 + sum_i (S1.byte[i] != 0 ? |S0.byte[i] - S1.byte[i]| : 0)
// With S1 = w & (0x01010101 << k) (each byte 0 or 2^k) and S0.byte[i] = 2^k +- weight_i the sum is
// the weighted popcount = 4 mask bits per instruction; two chained ops give one mask byte.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr uint32_t msad_const(int k, int base_weight_log2) {
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const int wgt = 1 << (base_weight_log2 + i);
        const int v = (1 << k) + wgt <= 255 ? (1 << k) + wgt : (1 << k) - wgt;
        r |= (uint32_t)v << (8 * i);
    }
    return r;
}

template <int K>
__device__ __forceinline__ sj_u64 plane_msad(const uint32_t w[16]) {
    constexpr uint32_t bit = 0x01010101u << K;
    constexpr uint32_t cA = msad_const(K, 0), cB = msad_const(K, 4);
    uint32_t half[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t v = __builtin_amdgcn_msad_u8(cA, w[8 * h + 2 * j] & bit, 0u);
            v = __builtin_amdgcn_msad_u8(cB, w[8 * h + 2 * j + 1] & bit, v);
            acc |= v << (8 * j);
        }
        half[h] = acc;
    }
    return (sj_u64)half[0] | ((sj_u64)half[1] << 32);
}

__device__ __forceinline__ void transpose_fast(const uint32_t w[16], sj_u64 p[8]) {
    p[0] = plane_msad<0>(w);
    p[1] = plane_msad<1>(w);
    p[2] = plane_msad<2>(w);
    p[3] = plane_msad<3>(w);
    p[4] = plane_msad<4>(w);
    p[5] = plane_msad<5>(w);
    p[6] = plane_msad<6>(w);
    p[7] = plane_msad<7>(w);
}

// self-test of the msad transposition against the portable loop (run once per context on the GPU)
__global__ void k_transpose_selftest(const uint32_t* __restrict__ words, uint32_t nblocks, uint32_t* mismatches) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = words[b * 16 + i];
    sj_u64 pf[8], pr[8];
    transpose_fast(w, pf);
    sj_transpose_ref(w, pr);
    uint32_t bad = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) bad += pf[k] != pr[k];
    if (bad) atomicAdd(mismatches, bad);
}

// ---------------------------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---------------------------------------------------------------------------------------------
// inter-tile chain: 8-byte granules, bit 63 = valid (both arrays zeroed by hipMemsetAsync per launch)
//   agg[t] (written by the tile's worker wave): [19:0] structurals if the tile is entered with
//          in-string parity 0, [39:20] with parity 1, [40] quote parity of the tile
//   pfx[t] (written by the scanner wave): [39:0] structurals before tile t, [40] parity entering t
// A granule is one naturally aligned 8-byte relaxed agent-scope store/load: the data is the flag.
//
// One elected SCANNER wave (raised priority) walks agg[] in order, 64 tiles per step, and turns it
// into pfx[]; it keeps the running (parity, count) in registers, so the chip-wide serial chain costs
// ~100 instructions per 64 tiles instead of one cross-chip round trip per look-back window.  (A
// classic per-tile decoupled look-back with a 64-tile window was measured first: it bounds the tile
// rate by 64 / poll-latency and floods the fabric with 512-byte polls from every resident wave.)
// Worker waves never wait before publishing agg[t], so the scanner always makes progress; a worker
// then polls ONE word (pfx[t]) from one lane.
// ---------------------------------------------------------------------------------------------
constexpr sj_u64 TS_VALID = 1ull << 63;
constexpr uint32_t SPIN_LIMIT = 1u << 21;  // ~ seconds; a healthy chain needs a handful of polls

__device__ __forceinline__ void ts_store(sj_u64* p, sj_u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ sj_u64 ts_load(const sj_u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The scanner wave: agg[] -> pfx[], then the document totals (count, unclosed string, sentinel).
// Each poll looks at a window of 64*SCAN_K tiles starting at the first unresolved tile (lane l owns
// SCAN_K consecutive tiles) and resolves exactly the contiguous run of published aggregates at its
// front -- never waiting for a later tile before serving an earlier one (forward progress: the
// lowest unfinished tile only ever needs aggregates of lower tiles).
constexpr int SCAN_K = 8;

__device__ __noinline__ void scanner_wave(const sj_u64* agg, sj_u64* pfx, uint32_t ntiles, uint32_t* __restrict__ out,
                                          sj_u64 out_cap, Stage1Result* res) {
    const int lane = threadIdx.x & 63;
    const sj_u64 lt_mask = (1ull << lane) - 1ull;
    __builtin_amdgcn_s_setprio(3);
    uint32_t P = 0;
    sj_u64 C = 0;
    bool dead = false;
    uint32_t base = 0, idle = 0;
    while (base < ntiles) {
        const uint32_t t0 = base + lane * SCAN_K;
        sj_u64 v[SCAN_K];
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j) v[j] = (t0 + j) < ntiles ? ts_load(&agg[t0 + j]) : 0ull;
        // r = number of leading published tiles of this lane; tiles past the end count as published
        uint32_t r = 0;
        bool open = true;
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j) {
            open = open && (v[j] != 0 || (t0 + j) >= ntiles);
            r += open ? 1u : 0u;
        }
        const sj_u64 full = __ballot(r == SCAN_K);
        const int L = ~full ? __builtin_ctzll(~full) : 64;  // first lane with a hole
        const uint32_t rL = L < 64 ? (uint32_t)__shfl((int)r, L) : 0u;
        const uint32_t take = lane < L ? SCAN_K : (lane == L ? rL : 0u);  // tiles of this lane resolved now
        uint32_t adv = (uint32_t)L * SCAN_K + rL;
        if (base + adv > ntiles) adv = ntiles - base;
        if (adv == 0) {
            if (++idle > SPIN_LIMIT) {  // a worker never published: give up loudly instead of hanging
                dead = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        idle = 0;
        // lane summary as a function of the parity entering the lane's first tile
        uint32_t lp = 0, s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j) {
            const bool on = (uint32_t)j < take && (t0 + j) < ntiles;
            const uint32_t a0 = on ? (uint32_t)v[j] & 0xFFFFFu : 0u, a1 = on ? (uint32_t)(v[j] >> 20) & 0xFFFFFu : 0u;
            s0 += lp ? a1 : a0;
            s1 += lp ? a0 : a1;
            lp ^= on ? (uint32_t)(v[j] >> 40) & 1u : 0u;
        }
        const sj_u64 pb = __ballot(lp);
        uint32_t pin = P ^ ((uint32_t)__popcll(pb & lt_mask) & 1u);  // parity entering this lane's first tile
        const uint32_t c = pin ? s1 : s0;
        const uint32_t incl = wave_incl_scan(c, lane);
        sj_u64 run = C + (incl - c);
#pragma unroll
        for (int j = 0; j < SCAN_K; ++j) {
            if ((uint32_t)j < take && (t0 + j) < ntiles) {
                ts_store(&pfx[t0 + j], TS_VALID | ((sj_u64)pin << 40) | run);
                run += pin ? (uint32_t)(v[j] >> 20) & 0xFFFFFu : (uint32_t)v[j] & 0xFFFFFu;
                pin ^= (uint32_t)(v[j] >> 40) & 1u;
            }
        }
        C += __shfl(incl, 63);
        P ^= (uint32_t)__popcll(pb) & 1u;
        base += adv;
    }
    if (lane == 0) {
        uint32_t e = dead ? SJMI_ST_INTERNAL : 0u;
        res->count = C;
        if (P) e |= SJMI_ST_UNCLOSED;        // StructuralIndexer.java:297-299
        if (C < out_cap) out[C] = 0;         // BitIndexes.finish :82-96
        else e |= SJMI_ST_CAPACITY;
        if (e) __hip_atomic_fetch_or(&res->status, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_setprio(0);
}

// worker side: publish this tile's aggregate, then wait for the scanner's prefix
__device__ __forceinline__ void tile_exchange(sj_u64* agg, const sj_u64* pfx, uint32_t tile, int lane, uint32_t T0,
                                              uint32_t T1, uint32_t tpar, uint32_t* par_in, sj_u64* cnt_in,
                                              Stage1Result* res) {
    uint32_t lo = 0, hi = 0;
    if (lane == 0) {
        ts_store(&agg[tile], TS_VALID | ((sj_u64)tpar << 40) | ((sj_u64)T1 << 20) | (sj_u64)T0);
        sj_u64 v = 0;
        for (uint32_t spins = 0;; ++spins) {
            v = ts_load(&pfx[tile]);
            if (v) break;
            if (spins > SPIN_LIMIT) {
                __hip_atomic_fetch_or(&res->status, SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        lo = (uint32_t)v;
        hi = (uint32_t)(v >> 32);
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    *par_in = (hi >> 8) & 1u;
    *cnt_in = (((sj_u64)(hi & 0xFFu)) << 32) | lo;
}

// ---------------------------------------------------------------------------------------------
// the stage-1 kernel.  WAVE-AUTONOMOUS tiles: one tile = one wave x S steps = S * 4 KiB of input.
// A wave never synchronises with its workgroup siblings: the in-tile parity / offset scans are
// ballot + shuffle, the staging buffer for coalesced index stores is a wave-private LDS slice, and
// every wave runs its own look-back.  (A first version used 256-lane tiles with 5 workgroup
// barriers per tile and one wave doing the look-back while three waited: 55 % of all wave-cycles
// were spent in s_waitcnt / s_barrier.)
//
// Tiles are handed out DYNAMICALLY to persistent waves: global wave g owns ticket counter g % 8 and
// its n-th ticket is tile 8*n + (g % 8).  One relaxed atomicAdd per tile, spread over 8 words (a
// single word saturates at ~88 tickets/us).  Because a wave takes its next tile only when it is
// ready to start it, tile order ~ start order: a tile's predecessors were started earlier, so the
// look-back rarely waits and the waves of a CU drift apart into different phases (loads of one
// wave overlap the ALU / look-back / stores of the others).  A static round-robin assignment was
// measured first: it locks all waves into the same phase and every iteration ends with a
// chip-wide wait on the slowest tile.
// Forward progress does not depend on dispatch order, placement or on how many workgroups are
// resident (>= 2): the lowest unfinished tile is either held by a running wave (its predecessors
// are done, so it finishes) or is the next ticket of a counter whose waves are running tiles with
// lower ids (which finish by induction).  Workgroups 0 and 1 alone cover all 8 counters.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
    // LDS accesses of one wave execute in order; this only stops the compiler from reordering them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The 64 bytes of one lane's block plus the 8 bytes before it (carries), as loaded from HBM.
struct StepData {
    uint4 q0, q1, q2, q3;
    sj_u64 halo;
};

// Branch-free on purpose (a block index past the end is clamped and its data ignored by the caller):
// a conditional load makes hipcc drain the whole load queue (s_waitcnt vmcnt(0)) at the join.
__device__ __forceinline__ void load_step(StepData& d, const uint8_t* __restrict__ buf, sj_u64 blk, sj_u64 nblocks) {
    const sj_u64 b = blk < nblocks ? blk : nblocks - 1;
    const uint4* src = reinterpret_cast<const uint4*>(buf + b * 64);
    d.q0 = src[0];
    d.q1 = src[1];
    d.q2 = src[2];
    d.q3 = src[3];
    d.halo = *reinterpret_cast<const sj_u64*>(buf + (b > 0 ? b * 64 - 8 : 0));  // unused for block 0
}

// phase 1 of a tile: transpose + classify every step (everything that needs no cross-lane data).
// Step 0 arrives preloaded (it was requested while the previous tile was being finished); the
// loads of steps 1..S-1 are issued up front and land while step 0 is being classified.
template <int S>
__device__ __forceinline__ void classify_tile(const uint32_t tile, StepData (&d)[2], const uint8_t* __restrict__ buf,
                                              const sj_u64 len, sj_u64 (&pot)[S], sj_u64 (&m0)[S], uint32_t (&fl)[S],
                                              const uint32_t dbg) {
    const int lane = threadIdx.x & 63;
    const sj_u64 nblocks = len / 64 + 1;  // the reference always processes one tail block (:255-294)
    const sj_u64 blk0 = (sj_u64)tile * (64 * S);
    uint32_t slow = 0;  // steps whose carries could not be resolved from the 8-byte halo (backslash run > 7)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        // software pipeline, depth 1: request step s+1, then classify step s.  The compiler barrier keeps
        // hipcc from hoisting every step's loads to the top (which costs 16 VGPRs per step in flight).
        // Ping-pong buffers with compile-time indices (S is even): step s lives in d[s & 1], so step 0 of
        // every tile is d[0] -- which is where the kernel loop prefetches the next tile's first step.
        if (s + 1 < S) load_step(d[(s + 1) & 1], buf, blk0 + (sj_u64)(s + 1) * 64 + lane, nblocks);
        asm volatile("" ::: "memory");
        const StepData& cur = d[s & 1];
        const sj_u64 blk = blk0 + (sj_u64)s * 64 + lane;
        pot[s] = 0;
        m0[s] = 0;
        fl[s] = 0;
        bool unresolved = false;
        if (blk < nblocks) {
            const sj_u64 start = blk * 64;
            const uint32_t w[16] = {cur.q0.x, cur.q0.y, cur.q0.z, cur.q0.w, cur.q1.x, cur.q1.y, cur.q1.z, cur.q1.w,
                                    cur.q2.x, cur.q2.y, cur.q2.z, cur.q2.w, cur.q3.x, cur.q3.y, cur.q3.z, cur.q3.w};
            uint32_t e_in = 0, p_in = 0;
            SjUtf8Carry uc = {0, 0, 0, 0};
            if (blk > 0) {
                uc = sj_utf8_carry(cur.halo);
                unresolved = !sj_carry_from_halo(cur.halo, &e_in, &p_in);
            }
            sj_u64 p[8];
            transpose_fast(w, p);
            const sj_u64 rem = len - start;
            sj_mask_tail(p, rem < 64 ? (uint32_t)rem : 64u);
            const SjBlockMasks bm = sj_block(p, e_in, p_in, uc);
            pot[s] = bm.pot;
            m0[s] = bm.sm0;
            fl[s] = bm.qpar | (bm.ue0 << 1) | (bm.ue1 << 2) | (bm.utf8 << 3);
        }
        slow |= unresolved ? (1u << s) : 0u;
    }
    // Rare: a backslash run longer than the halo reaches a block boundary.  Kept out of the streaming
    // loop (its dependent byte loads would make hipcc drain the load queue there): redo those blocks.
    if (__ballot(slow != 0)) {
        for (int s = 0; s < S; ++s) {
            if (!((slow >> s) & 1u)) continue;
            const sj_u64 blk = blk0 + (sj_u64)s * 64 + lane;
            const sj_u64 start = blk * 64;
            uint32_t w[16];
            for (int i = 0; i < 16; ++i) w[i] = reinterpret_cast<const uint32_t*>(buf + start)[i];
            sj_u64 halo = *reinterpret_cast<const sj_u64*>(buf + start - 8);
            uint32_t e_in = 0, p_in = 0;
            sj_carry_slow(buf, 0, start, &e_in, &p_in);
            sj_u64 p[8];
            sj_transpose_ref(w, p);
            const sj_u64 rem = len - start;
            sj_mask_tail(p, rem < 64 ? (uint32_t)rem : 64u);
            const SjBlockMasks bm = sj_block(p, e_in, p_in, sj_utf8_carry(halo));
            pot[s] = bm.pot;
            m0[s] = bm.sm0;
            fl[s] = bm.qpar | (bm.ue0 << 1) | (bm.ue1 << 2) | (bm.utf8 << 3);
        }
    }
}

// phases 2..6 of a tile.  Returns the tile's SJMI_ST_* error bits (per lane; the caller ORs them
// over the wave's lifetime).
template <int S>
__device__ __forceinline__ uint32_t finish_tile(uint32_t* __restrict__ stage /* wave-private, STAGE_CAP entries */,
                                                const uint32_t tile, sj_u64 (&pot)[S], sj_u64 (&m0)[S],
                                                uint32_t (&fl)[S], const sj_u64 len, uint32_t* __restrict__ out,
                                                const sj_u64 out_cap, sj_u64* agg, const sj_u64* pfx,
                                                Stage1Result* res, const uint32_t dbg) {
    const int lane = threadIdx.x & 63;
    const sj_u64 blk0 = (sj_u64)tile * (64 * S);
    const sj_u64 lt_mask = (1ull << lane) - 1ull;
    const bool timing = dbg & DBG_TIMING;
    sj_u64 tm[6];
    if (timing) tm[1] = __builtin_amdgcn_s_memtime();
    // ---- phase 2: in-string parity prefix inside the tile (order: step, lane) ----
    uint32_t lpar[S];
    uint32_t tpar = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const sj_u64 bal = __ballot(fl[s] & 1u);
        lpar[s] = ((uint32_t)__popcll(bal & lt_mask) & 1u) ^ tpar;
        tpar ^= (uint32_t)__popcll(bal) & 1u;
    }

    // ---- phase 3: structurals for tile-entry parity 0, counts for both parities, offsets ----
    uint32_t ex0[S], exp_[S];
    uint32_t T0 = 0, TP = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        m0[s] = lpar[s] ? (pot[s] & m0[s]) : (pot[s] & ~m0[s]);  // StructuralIndexer.java:251
        const uint32_t c0 = (uint32_t)__popcll(m0[s]), cp = (uint32_t)__popcll(pot[s]);
        const uint32_t packed = wave_incl_scan(c0 | (cp << 16), lane);  // both <= 4096 per step: no carry
        const uint32_t tot = __shfl(packed, 63);
        ex0[s] = T0 + (packed & 0xFFFFu) - c0;
        exp_[s] = TP + (packed >> 16) - cp;
        T0 += tot & 0xFFFFu;
        TP += tot >> 16;
    }
    const uint32_t T1 = TP - T0;

    if (timing) tm[2] = __builtin_amdgcn_s_memtime();
    // ---- phase 4: decoupled look-back, by the wave itself ----
    uint32_t par_in;
    sj_u64 cnt_in;
    if (dbg & DBG_NO_LOOKBACK) {  // ablation: no inter-tile chain (indexes land at fake offsets)
        par_in = 0;
        cnt_in = ((sj_u64)tile * 1430ull * S) % (out_cap / 2);
    } else {
        tile_exchange(agg, pfx, tile, lane, T0, T1, tpar, &par_in, &cnt_in, res);
    }
    const uint32_t T = par_in ? T1 : T0;
    if (timing) tm[3] = __builtin_amdgcn_s_memtime();

    // ---- phase 5: final masks + error flags ----
    uint32_t err = 0;
    uint32_t pos[S];  // tile-relative position of the lane's next index
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t in_par = lpar[s] ^ par_in;  // parity entering this block
        if (in_par ? (fl[s] & 4u) : (fl[s] & 2u)) err |= SJMI_ST_UNESCAPED;  // :252,:300-302
        if (fl[s] & 8u) err |= SJMI_ST_UTF8;
        m0[s] = par_in ? (pot[s] ^ m0[s]) : m0[s];
        pos[s] = par_in ? (exp_[s] - ex0[s]) : ex0[s];
    }
    const bool fits = cnt_in + T < out_cap;  // strict: keeps room for the sentinel
    if (!fits && T) err |= SJMI_ST_CAPACITY;

    // ---- phase 6: index emission (BitIndexes.write :14-41): expand the masks into the wave's LDS
    //      slice at their tile-relative positions, then store them with coalesced 4-byte stores ----
    if (fits && !(dbg & DBG_NO_WRITE)) {
        for (uint32_t base = 0; base < T; base += STAGE_CAP) {
            const uint32_t lim = base + STAGE_CAP;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t bstart = (uint32_t)((blk0 + (sj_u64)s * 64 + lane) * 64);
                uint32_t lo = (uint32_t)m0[s], hi = (uint32_t)(m0[s] >> 32);
                uint32_t ps = pos[s];
                while (lo && ps < lim) {
                    stage[ps - base] = bstart + (uint32_t)__builtin_ctz(lo);
                    lo &= lo - 1;
                    ++ps;
                }
                while (!lo && hi && ps < lim) {
                    stage[ps - base] = bstart + 32u + (uint32_t)__builtin_ctz(hi);
                    hi &= hi - 1;
                    ++ps;
                }
                m0[s] = (sj_u64)lo | ((sj_u64)hi << 32);
                pos[s] = ps;
            }
            wave_lds_fence();
            const uint32_t n = (T - base) < STAGE_CAP ? (T - base) : STAGE_CAP;
            uint32_t* dst = out + cnt_in + base;
            for (uint32_t i = lane; i < n; i += 64) dst[i] = stage[i];
            wave_lds_fence();
        }
    }
    if (timing) {
        tm[4] = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            sj_u64* acc = reinterpret_cast<sj_u64*>(reinterpret_cast<uint8_t*>(res) - WS_RESULT_OFFSET + WS_TIMING_OFFSET);
            atomicAdd(&acc[1], tm[2] - tm[1]);  // in-tile scans
            atomicAdd(&acc[2], tm[3] - tm[2]);  // look-back
            atomicAdd(&acc[3], tm[4] - tm[3]);  // expand + store
            atomicAdd(&acc[4], 1ull);
        }
    }
    return err;
}

// register budget: waves per SIMD the allocator must leave room for (= workgroups per CU at 256 threads)
template <int S> struct WavesPerSimd { static constexpr int v = S <= 4 ? 4 : 3; };

template <int S>
__global__ void __launch_bounds__(256)
k_stage1(const uint8_t* __restrict__ buf, sj_u64 len, uint32_t* __restrict__ out, sj_u64 out_cap, sj_u64* agg,
         sj_u64* pfx, uint32_t* tickets /* NUM_TICKETS tile counters + 1 election word, 16 dwords apart */,
         Stage1Result* res, uint32_t ntiles, uint32_t dbg) {
    __shared__ uint32_t s_stage[4][STAGE_CAP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform => tile math stays scalar
    // elect the scanner: the first wave to arrive (placement / dispatch-order independent)
    uint32_t arrival = 0;
    if (lane == 0) arrival = atomicAdd(tickets + NUM_TICKETS * 16, 1u);
    arrival = __builtin_amdgcn_readfirstlane(arrival);
    if (arrival == 0) {
        if (!(dbg & DBG_NO_LOOKBACK)) scanner_wave(agg, pfx, ntiles, out, out_cap, res);
        return;
    }
    uint32_t* stage = s_stage[wave];
    const uint32_t shard = (arrival - 1) % NUM_TICKETS;  // arrivals 1..8 cover all counters
    uint32_t* my_ticket = tickets + shard * 16;
    const sj_u64 nblocks = len / 64 + 1;
    uint32_t err = 0;
    uint32_t n = 0;
    if (lane == 0) n = atomicAdd(my_ticket, 1u);
    n = __builtin_amdgcn_readfirstlane(n);
    sj_u64 tile = (sj_u64)n * NUM_TICKETS + shard;
    if (tile < ntiles) {
        StepData d[2];
        load_step(d[0], buf, tile * (64 * S) + lane, nblocks);
        for (;;) {
            // the ticket for the NEXT tile is drawn now and only looked at after this tile's phase 1
            uint32_t nn = 0;
            if (lane == 0) nn = atomicAdd(my_ticket, 1u);
            sj_u64 pot[S], m0[S];
            uint32_t fl[S];
            classify_tile<S>((uint32_t)tile, d, buf, len, pot, m0, fl, dbg);
            nn = __builtin_amdgcn_readfirstlane(nn);
            const sj_u64 next = (sj_u64)nn * NUM_TICKETS + shard;
            // prefetch step 0 of the next tile: it flies while this tile waits for its prefix and stores
            // (unconditional: past the last tile the clamped loads are simply discarded; a branch here makes
            // hipcc wait for the prefetch right away)
            load_step(d[0], buf, next * (64 * S) + lane, nblocks);
            asm volatile("" ::: "memory");
            err |= finish_tile<S>(stage, (uint32_t)tile, pot, m0, fl, len, out, out_cap, agg, pfx, res, dbg);
            if (next >= ntiles) break;
            tile = next;
        }
    }
    // one status update per wave, after its last tile
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) err |= __shfl_xor(err, d);
    if (lane == 0 && err) __hip_atomic_fetch_or(&res->status, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static uint64_t tiles_for(uint64_t len, int steps) {
    const uint64_t nblocks = len / 64 + 1;
    return (nblocks + 64ull * steps - 1) / (64ull * steps);
}

size_t stage1_workspace_bytes(uint64_t len, int steps) {
    return WS_TILE_STATE_OFFSET + 2 * (size_t)tiles_for(len, steps) * sizeof(sj_u64);
}

int stage1_pick_steps(uint64_t len) {
    // small documents: small tiles so that more CUs get work; large: bigger tiles (fewer granules)
    return len <= (4u << 20) ? 2 : 4;
}

template <int S>
static hipError_t occupancy_of(int* blocks_per_cu) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_stage1<S>, 256, 0);
}

hipError_t stage1_resident_blocks(int steps, int* blocks_per_cu) {
    switch (steps) {
    case 2: return occupancy_of<2>(blocks_per_cu);
    case 4: return occupancy_of<4>(blocks_per_cu);
    case 8: return occupancy_of<8>(blocks_per_cu);
    default: return hipErrorInvalidValue;
    }
}

hipError_t stage1_launch(const uint8_t* d_buf, uint64_t len, uint32_t* d_out, uint64_t out_cap, void* d_ws,
                         int steps, uint32_t max_grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop,
                         uint32_t dbg) {
    const uint64_t tiles = tiles_for(len, steps);
    const size_t ws_bytes = WS_TILE_STATE_OFFSET + 2 * (size_t)tiles * sizeof(sj_u64);
    hipError_t e = hipMemsetAsync(d_ws, 0, ws_bytes, stream);
    if (e != hipSuccess) return e;
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    uint32_t* tickets = reinterpret_cast<uint32_t*>(ws + WS_TICKET_OFFSET);
    Stage1Result* res = reinterpret_cast<Stage1Result*>(ws + WS_RESULT_OFFSET);
    sj_u64* agg = reinterpret_cast<sj_u64*>(ws + WS_TILE_STATE_OFFSET);
    sj_u64* pfx = agg + tiles;
    // persistent grid: as many workgroups as fit on the chip (speed only; any grid >= 3 is correct:
    // 1 scanner wave + 8 worker waves to own the 8 ticket counters)
    uint64_t wgs = (tiles + 1 + 3) / 4;
    if (wgs < 3) wgs = 3;
    const unsigned g = (unsigned)(wgs < max_grid ? wgs : max_grid);
    const dim3 block(256);
    if (ev_start && (e = hipEventRecord(ev_start, stream)) != hipSuccess) return e;
#define SJMI_LAUNCH(S_)                                                                                          \
    hipLaunchKernelGGL((k_stage1<S_>), dim3(g), block, 0, stream, d_buf, (sj_u64)len, d_out, (sj_u64)out_cap, agg, \
                       pfx, tickets, res, (uint32_t)tiles, dbg)
    switch (steps) {
    case 2: SJMI_LAUNCH(2); break;
    case 4: SJMI_LAUNCH(4); break;
    case 8: SJMI_LAUNCH(8); break;
    default: return hipErrorInvalidValue;
    }
#undef SJMI_LAUNCH
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_stop) return hipEventRecord(ev_stop, stream);
    return hipSuccess;
}

hipError_t transpose_selftest_launch(const uint32_t* d_words, uint32_t nblocks, uint32_t* d_mismatches,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(k_transpose_selftest, dim3((nblocks + 255) / 256), dim3(256), 0, stream, d_words, nblocks,
                       d_mismatches);
    return hipGetLastError();
}

}  // namespace sjmi
