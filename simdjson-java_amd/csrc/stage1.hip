// stage1.hip -- MI355X (gfx950) stage-1 kernel: fused UTF-8 validation + structural indexing +
// index compaction in ONE pass over the document.
//
// Replaces SimdJsonParser.stage1 = Utf8Validator.validate + StructuralIndexer.index + BitIndexes.write
// (/root/reference/src/main/java/org/simdjson/SimdJsonParser.java:55-58, Utf8Validator.java:54-168,
//  StructuralIndexer.java:196-303, BitIndexes.java:14-41,82-96).
//
// Mapping to the hardware:
//   * one lane  = one 64-byte block (the reference's loop step), loaded as 4 x dwordx4; loads are
//     software-pipelined one step ahead of the ALU work;
//   * the block is transposed to 8 bit planes (8x8 bit-matrix butterflies in VOP2 ops + v_perm_b32 byte
//     transposes, sj_block.h; an earlier v_and + v_msad_u8 form is kept for the self-test) and all
//     classification / escape / string / UTF-8 logic is 64-bit boolean algebra in VGPRs;
//   * the three serial carries of the reference loop (prevEscaped, prevScalar, previous 4 UTF-8
//     bytes) are LOCAL: each lane re-derives them from the 8 bytes before its block;
//   * the two truly global carries -- in-string parity (XOR scan) and the output offset (+ scan of
//     popcounts) -- are resolved inside the wave by ballot/shuffle, inside the workgroup through
//     LDS, and across workgroups by a single-pass DECOUPLED LOOK-BACK over 8-byte {state,payload}
//     granules (agent-scope relaxed atomics, one granule per 64 KiB tile), so the input is read
//     from HBM exactly once.  Tiles are handed out by an atomic ticket so that a tile only ever
//     waits for tiles whose workgroups have already started (no dependence on dispatch order);
//   * structurals depend on the incoming parity only through a complement
//     (structurals(p) = p ? pot & sm : pot & ~sm), so each tile publishes counts for BOTH parities
//     and the look-back composes functions {0,1} -> (parity, count);
//   * indexes are expanded into a wave-private LDS slice and leave the CU as coalesced stores.
// Measured cost model on gfx950 (tools/ubench/valu_rate.hip): VOP2 integer ops issue in 2 cycles per
// wave, every VOP3 op (v_msad_u8, v_or3, v_lshl_or, v_bfi, v_bcnt, 64-bit shifts) in 4.  The kernel
// is VALU-bound (775 VALU instructions per 4 KiB wave-step after this round's reductions).
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
    sj_u64 pf[8], pb[8], pr[8];
    transpose_fast(w, pf);       // v_and + v_msad_u8 form
    sj_transpose_butterfly(w, pb);  // butterfly + v_perm_b32 form (the one the kernel uses)
    sj_transpose_ref(w, pr);
    uint32_t bad = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) bad += (pf[k] != pr[k]) + (pb[k] != pr[k]);
    if (bad) atomicAdd(mismatches, bad);
}

// ---------------------------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
// Inclusive + scan over the 64 lanes with DPP (no LDS traffic): Kogge-Stone inside each row of 16 lanes
// (row_shr 1/2/4/8, out-of-row reads give 0), then the row totals are carried across with row_bcast:15 / :31.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    (void)lane;
    v = dpp_add<0x111, 0xF>(v);  // row_shr:1
    v = dpp_add<0x112, 0xF>(v);  // row_shr:2
    v = dpp_add<0x114, 0xF>(v);  // row_shr:4
    v = dpp_add<0x118, 0xF>(v);  // row_shr:8
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---------------------------------------------------------------------------------------------
// tile-state granules (one u64 per tile; zeroed by hipMemsetAsync before every launch)
//   bits 63..62 : 0 = nothing yet, 1 = AGGREGATE, 2 = INCLUSIVE PREFIX
//   AGGREGATE   : [19:0] structurals if the tile is entered with parity 0, [39:20] with parity 1,
//                 [40] quote parity of the tile
//   PREFIX      : [39:0] structurals in tiles 0..t, [40] in-string parity after tile t
// A granule is one naturally aligned 8-byte relaxed agent-scope store/load: the data is the flag
// (cdna_hip_programming.md Guideline 16, form R2), so no fences are needed.
// ---------------------------------------------------------------------------------------------
constexpr sj_u64 TS_AGG = 1ull << 62, TS_PFX = 2ull << 62;
constexpr uint32_t SPIN_LIMIT = 1u << 21;  // ~ seconds; a healthy chain needs a handful of polls

__device__ __forceinline__ void ts_store(sj_u64* p, sj_u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ sj_u64 ts_load(const sj_u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Executed by all 64 lanes of wave 0.  Returns the parity / structural count entering `tile`.
__device__ __forceinline__ void tile_lookback(sj_u64* tile_state, uint32_t tile, int lane, uint32_t T0,
                                              uint32_t T1, uint32_t tpar, uint32_t* par_in, sj_u64* cnt_in,
                                              Stage1Result* res) {
    if (tile == 0) {
        *par_in = 0;
        *cnt_in = 0;
        if (lane == 0) ts_store(&tile_state[0], TS_PFX | ((sj_u64)tpar << 40) | (sj_u64)T0);
        return;
    }
    if (lane == 0) ts_store(&tile_state[tile], TS_AGG | ((sj_u64)tpar << 40) | ((sj_u64)T1 << 20) | (sj_u64)T0);

    sj_u64 g0 = 0, g1 = 0;  // structurals in the tiles already folded, if entered with parity 0 / 1
    uint32_t gpar = 0;      // their combined quote parity
    long long k = (long long)tile;  // lane i looks at tile k-1-i
    uint32_t P = 0;
    sj_u64 C = 0;
    for (;;) {
        const long long t = k - 1 - lane;
        sj_u64 v = 0;
        int J = 64;
        for (uint32_t spins = 0;; ++spins) {
            if (t >= 0) v = ts_load(&tile_state[t]);
            const bool is_pfx = (t < 0) || ((v >> 62) == 2);
            const bool ready = (t < 0) || (v != 0);
            const sj_u64 pm = __ballot(is_pfx);
            J = pm ? __builtin_ctzll(pm) : 64;
            const sj_u64 need = J >= 64 ? ~0ull : ((1ull << J) - 1ull);
            if ((__ballot(ready) & need) == need) break;
            if (spins > SPIN_LIMIT) {  // never expected: a predecessor tile did not publish
                if (lane == 0)
                    __hip_atomic_fetch_or(&res->status, SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const bool in_win = lane < J;  // AGGREGATE tiles newer than the nearest PREFIX
        const uint32_t apar = in_win ? (uint32_t)(v >> 40) & 1u : 0u;
        const sj_u64 pb = __ballot(apar);
        // parity accumulated by the window's tiles OLDER than this lane's tile (higher lanes)
        const uint32_t q = (lane < 63) ? (uint32_t)__popcll(pb >> (lane + 1)) & 1u : 0u;
        const uint32_t a0 = (uint32_t)v & 0xFFFFFu, a1 = (uint32_t)(v >> 20) & 0xFFFFFu;
        const uint32_t w0 = wave_sum(in_win ? (q ? a1 : a0) : 0u);  // window entered with parity 0
        const uint32_t w1 = wave_sum(in_win ? (q ? a0 : a1) : 0u);  // ... with parity 1
        const uint32_t wpar = (uint32_t)__popcll(pb) & 1u;
        if (J < 64) {
            const long long tj = k - 1 - J;
            const sj_u64 pv = __shfl(v, J);  // 64-bit shuffle of lane J's granule
            P = tj < 0 ? 0u : (uint32_t)(pv >> 40) & 1u;
            C = tj < 0 ? 0ull : (pv & ((1ull << 40) - 1ull));
            C += P ? w1 : w0;
            P ^= wpar;
            C += P ? g1 : g0;
            P ^= gpar;
            break;
        }
        // no prefix among these 64 tiles: fold the window in front of the suffix and keep walking
        const sj_u64 n0 = (sj_u64)w0 + (wpar ? g1 : g0);
        const sj_u64 n1 = (sj_u64)w1 + (wpar ? g0 : g1);
        g0 = n0;
        g1 = n1;
        gpar ^= wpar;
        k -= 64;
    }
    *par_in = P;
    *cnt_in = C;
    if (lane == 0)
        ts_store(&tile_state[tile], TS_PFX | ((sj_u64)(P ^ tpar) << 40) | (C + (P ? T1 : T0)));
}

// ---------------------------------------------------------------------------------------------
// the stage-1 kernel.  One workgroup (4 waves) = one tile of 4 * S * 4 KiB; wave w owns the w-th
// contiguous quarter of it (S steps of 64 blocks), so every wave's indexes are one contiguous run of
// the output and can be staged + stored without a workgroup barrier.  Three barriers per tile: the
// parity table, the count table, the look-back broadcast.
//
// Tile assignment: tile = blockIdx (fast, default) or an atomic ticket (safe fallback) -- see the kernel.
// Variants measured and dropped: 16 KiB workgroup tiles (ticket-bound), persistent
// waves with static striding (phase-locked waves, VGPR growth from loop-invariant hoisting),
// wave-autonomous 8-32 KiB tiles with per-wave look-back (chain rate ~100 tiles/us), a dedicated
// scanner wave (one cross-chip round trip per 256-1024 tiles, still ~100 tiles/us).
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

template <int S>
struct TileShared {
    uint32_t wpar[4], wc0[4], wcp[4];
    uint32_t par_in, tile;
    sj_u64 cnt_in;
    alignas(16) uint32_t stage[4][STAGE_CAP];  // per-wave staging of indexes for coalesced stores
};

template <int S>
__global__ void __launch_bounds__(256)
k_stage1(const uint8_t* __restrict__ buf, sj_u64 len, uint32_t* __restrict__ out, sj_u64 out_cap,
         sj_u64* tile_state, uint32_t* ticket, Stage1Result* res, uint32_t dbg) {
    __shared__ TileShared<S> sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform => scalar tile math
    uint32_t tile;
    if (dbg & FLAG_TICKET) {
        // SAFE mode: tiles handed out by an atomic ticket, so a tile only waits for tiles whose workgroups already
        // run, whatever the dispatch order.  Costs ~12 % (one exposed atomic round trip per workgroup).
        if (threadIdx.x == 0) sh.tile = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        tile = sh.tile;
    } else {
        // FAST mode (default): tile = blockIdx.  RESULTS never depend on dispatch order; LIVENESS does: a tile
        // spins on lower tiles, which is fine as long as lower-numbered workgroups are dispatched no later than
        // higher ones (what gfx950 is observed to do for 1-D grids).  If that ever fails, the bounded spin in
        // tile_lookback trips, the launch reports SJMI_ST_INTERNAL and the host re-runs it in SAFE mode.
        tile = blockIdx.x;
        if ((dbg & DBG_FAKE_TIMEOUT) && blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_fetch_or(&res->status, SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const sj_u64 nblocks = len / 64 + 1;  // the reference always processes one tail block (:255-294)
    const sj_u64 blk0 = (sj_u64)tile * (256 * S) + (sj_u64)wave * (64 * S);  // first block of this wave
    const sj_u64 lt_mask = (1ull << lane) - 1ull;

    sj_u64 pot[S], m0[S];
    uint32_t fl[S];  // bit0 quote parity, bit1 ue0, bit2 ue1, bit3 utf8 error

    // ---- phase 1: load, transpose, classify (everything that needs no cross-lane data) ----
    {
        StepData d[2];
        uint32_t slow = 0;  // steps whose carries the 8-byte halo could not resolve (backslash run > 7)
        load_step(d[0], buf, blk0 + lane, nblocks);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // software pipeline, depth 1, ping-pong buffers with compile-time indices; the compiler barrier
            // keeps hipcc from hoisting every step's loads to the top (16 VGPRs per step in flight)
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
                sj_transpose_butterfly(w, p);
                const sj_u64 rem = len - start;
                sj_mask_tail(p, rem < 64 ? (uint32_t)rem : 64u);
                // the UTF-8 algebra is skipped when no lane of the wave has a non-ASCII byte or a pending carry
                const bool need_utf8 = __ballot((p[7] != 0) | ((uc.c1 | uc.c2 | uc.c3 | uc.sec) != 0)) != 0;
                const SjBlockMasks bm = sj_block(p, e_in, p_in, uc, need_utf8);
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
                const sj_u64 halo = *reinterpret_cast<const sj_u64*>(buf + start - 8);
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

    // ---- phase 2: in-string parity prefix inside the tile (order: wave, step, lane) ----
    uint32_t lpar[S];
    uint32_t wpar = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const sj_u64 bal = __ballot(fl[s] & 1u);
        lpar[s] = ((uint32_t)__popcll(bal & lt_mask) & 1u) ^ wpar;
        wpar ^= (uint32_t)__popcll(bal) & 1u;
    }
    if (lane == 0) sh.wpar[wave] = wpar;
    __syncthreads();
    uint32_t tpar = 0;
    {
        uint32_t before = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j == wave) before = tpar;  // parity of the waves in front of this one
            tpar ^= sh.wpar[j];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) lpar[s] ^= before;
    }

    // ---- phase 3: structurals for tile-entry parity 0, counts for both parities, offsets ----
    uint32_t ex0[S], exp_[S];
    uint32_t W0 = 0, WP = 0;  // wave totals
#pragma unroll
    for (int s = 0; s < S; ++s) {
        m0[s] = lpar[s] ? (pot[s] & m0[s]) : (pot[s] & ~m0[s]);  // StructuralIndexer.java:251
        const uint32_t c0 = (uint32_t)__popcll(m0[s]), cp = (uint32_t)__popcll(pot[s]);
        const uint32_t packed = wave_incl_scan(c0 | (cp << 16), lane);  // both <= 4096 per step: no carry
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)packed, 63);
        ex0[s] = W0 + (packed & 0xFFFFu) - c0;   // wave-relative exclusive offsets
        exp_[s] = WP + (packed >> 16) - cp;
        W0 += tot & 0xFFFFu;
        WP += tot >> 16;
    }
    if (lane == 0) {
        sh.wc0[wave] = W0;
        sh.wcp[wave] = WP;
    }
    __syncthreads();
    uint32_t T0 = 0, TP = 0, base0 = 0, basep = 0;  // tile totals; offsets of this wave inside the tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j == wave) {
            base0 = T0;
            basep = TP;
        }
        T0 += sh.wc0[j];
        TP += sh.wcp[j];
    }
    const uint32_t T1 = TP - T0;

    // ---- phase 4: decoupled look-back (wave 0), broadcast through LDS ----
    if (wave == 0) {
        uint32_t P;
        sj_u64 C;
        if (dbg & DBG_NO_LOOKBACK) {  // ablation: no inter-tile chain (indexes land at fake offsets)
            P = 0;
            C = ((sj_u64)tile * 5700ull * S) % (out_cap / 2);
        } else {
            tile_lookback(tile_state, tile, lane, T0, T1, tpar, &P, &C, res);
        }
        if (lane == 0) {
            sh.par_in = P;
            sh.cnt_in = C;
        }
    }
    __syncthreads();
    const uint32_t par_in = sh.par_in;
    const sj_u64 cnt_in = sh.cnt_in;
    const uint32_t T = par_in ? T1 : T0;

    // ---- phase 5: final masks + error flags ----
    uint32_t err = 0;
    uint32_t pos[S];  // wave-relative position of the lane's next index
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

    // ---- phase 6: index emission (BitIndexes.write :14-41): expand the wave's masks into its LDS
    //      slice at their wave-relative positions, then store them with coalesced 4-byte stores ----
    if (fits && !(dbg & DBG_NO_WRITE)) {
        uint32_t* stage = sh.stage[wave];
        const uint32_t WT = par_in ? (WP - W0) : W0;                       // indexes of this wave
        uint32_t* dst0 = out + cnt_in + (par_in ? (basep - base0) : base0);  // its contiguous output run
        if (WT + 3 <= STAGE_CAP) {
            // common case: everything fits in one round, so the per-bit loops need no window test.  Entries are
            // staged at the same position modulo 4 as their final index, so that whole 16-byte quads of the LDS
            // slice go out as global_store_dwordx4 (the index array is 16-byte aligned); only the two boundary
            // quads of the wave's run need element-wise stores.
            const uint32_t g0 = (uint32_t)((cnt_in + (par_in ? (basep - base0) : base0)) & 3ull);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint32_t bstart = (uint32_t)((blk0 + (sj_u64)s * 64 + lane) * 64);
                uint32_t* q = stage + g0 + pos[s];
                for (uint32_t lo = (uint32_t)m0[s]; lo; lo &= lo - 1) *q++ = bstart | (uint32_t)__builtin_ctz(lo);
                for (uint32_t hi = (uint32_t)(m0[s] >> 32); hi; hi &= hi - 1) *q++ = bstart | 32u | (uint32_t)__builtin_ctz(hi);
            }
            wave_lds_fence();
            const uint32_t span = g0 + WT;
            uint32_t* gbase = dst0 - g0;  // 16-byte aligned
            for (uint32_t qi = lane; qi * 4 < span; qi += 64) {
                const uint4 v = reinterpret_cast<const uint4*>(stage)[qi];
                const uint32_t lo = qi * 4;
                if (lo >= g0 && lo + 4 <= span) {
                    reinterpret_cast<uint4*>(gbase)[qi] = v;
                } else {
                    if (lo + 0 >= g0 && lo + 0 < span) gbase[lo + 0] = v.x;
                    if (lo + 1 >= g0 && lo + 1 < span) gbase[lo + 1] = v.y;
                    if (lo + 2 >= g0 && lo + 2 < span) gbase[lo + 2] = v.z;
                    if (lo + 3 >= g0 && lo + 3 < span) gbase[lo + 3] = v.w;
                }
            }
            wave_lds_fence();
        } else {
            for (uint32_t base = 0; base < WT; base += STAGE_CAP) {
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
                const uint32_t n = (WT - base) < STAGE_CAP ? (WT - base) : STAGE_CAP;
                uint32_t* dst = dst0 + base;
                for (uint32_t i = lane; i < n; i += 64) dst[i] = stage[i];
                wave_lds_fence();
            }
        }
    }
    // one status update per wave
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) err |= __shfl_xor(err, d);
    if (lane == 0 && err) __hip_atomic_fetch_or(&res->status, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the tile that holds the final (tail) block finishes the job: count, sentinel, unclosed string
    if (threadIdx.x == 0 && tile == (uint32_t)((nblocks - 1) / (256 * S))) {
        const sj_u64 total = cnt_in + T;
        res->count = total;
        uint32_t e = 0;
        if (par_in ^ tpar) e |= SJMI_ST_UNCLOSED;  // StructuralIndexer.java:297-299
        if (total < out_cap) out[total] = 0;       // BitIndexes.finish :82-96
        else e |= SJMI_ST_CAPACITY;
        if (e) __hip_atomic_fetch_or(&res->status, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static uint64_t tiles_for(uint64_t len, int steps) {
    const uint64_t nblocks = len / 64 + 1;
    return (nblocks + 256ull * steps - 1) / (256ull * steps);
}

size_t stage1_workspace_bytes(uint64_t len, int steps) {
    return WS_TILE_STATE_OFFSET + (size_t)tiles_for(len, steps) * sizeof(sj_u64);
}

int stage1_pick_steps(uint64_t len) {
    // small documents: small tiles so that more CUs get work; large: 64 KiB tiles (fewer tickets / granules)
    return len <= (4u << 20) ? 1 : 4;
}

hipError_t stage1_launch(const uint8_t* d_buf, uint64_t len, uint32_t* d_out, uint64_t out_cap, void* d_ws,
                         int steps, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t dbg) {
    const uint64_t tiles = tiles_for(len, steps);
    const size_t ws_bytes = WS_TILE_STATE_OFFSET + (size_t)tiles * sizeof(sj_u64);
    hipError_t e = hipMemsetAsync(d_ws, 0, ws_bytes, stream);
    if (e != hipSuccess) return e;
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    uint32_t* ticket = reinterpret_cast<uint32_t*>(ws + WS_TICKET_OFFSET);
    Stage1Result* res = reinterpret_cast<Stage1Result*>(ws + WS_RESULT_OFFSET);
    sj_u64* ts = reinterpret_cast<sj_u64*>(ws + WS_TILE_STATE_OFFSET);
    const dim3 grid((unsigned)tiles), block(256);
    if (ev_start && (e = hipEventRecord(ev_start, stream)) != hipSuccess) return e;
#define SJMI_LAUNCH(S_)                                                                                           \
    hipLaunchKernelGGL((k_stage1<S_>), grid, block, 0, stream, d_buf, (sj_u64)len, d_out, (sj_u64)out_cap, ts, ticket, \
                       res, dbg)
    switch (steps) {
    case 1: SJMI_LAUNCH(1); break;
    case 2: SJMI_LAUNCH(2); break;
    case 4: SJMI_LAUNCH(4); break;
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
