// stage1.hip -- MI355X (gfx950) stage-1 kernel: fused UTF-8 validation + structural indexing +
// index compaction in ONE pass over the document.
//
// Replaces SimdJsonParser.stage1 = Utf8Validator.validate + StructuralIndexer.index + BitIndexes.write
// (/root/reference/src/main/java/org/simdjson/SimdJsonParser.java:55-58, Utf8Validator.java:54-168,
//  StructuralIndexer.java:196-303, BitIndexes.java:14-41,82-96).
//
// Mapping to the hardware:
//   * one lane  = one 64-byte block (the reference's loop step), loaded as 4 x dwordx4; the loads of a step are
//     requested as soon as the previous step has been transposed out of the registers they land in;
//   * the block is transposed to 8 bit planes (v_perm_b32 byte transposes + one 8x8 bit-matrix butterfly across
//     8 registers per 32-byte half; an earlier v_and + v_msad_u8 form is kept for the self-test) and all
//     classification / escape / string / UTF-8 logic is boolean algebra on the planes in VGPRs: sj_block.h states it
//     on 64-bit masks, sj_block32.h -- what the streaming loop runs since round 4 -- on 32-bit halves with one explicit
//     v_bitop3_b32 per three-input function (a sixth fewer instructions; bit-identical, fuzzed against each other);
//   * the three serial carries of the reference loop (prevEscaped, prevScalar, previous 4 UTF-8
//     bytes) are LOCAL: each lane re-derives them from the 8 bytes before its block;
//   * the two truly global carries -- in-string parity (XOR scan) and the output offset (+ scan of
//     popcounts) -- are resolved inside the wave by ballot / DPP scans and across waves by a single-pass chain
//     over 8-byte {state,payload} granules (agent-scope relaxed atomics, one granule per 16 KiB of input): every
//     worker wave publishes its granule's AGGREGATE, a scanner workgroup turns aggregates into PREFIXes, and the
//     worker picks its prefix up one classification later (software pipeline), so the input is read from HBM
//     exactly once and nobody waits for the chain;
//   * structurals depend on the incoming parity only through a complement
//     (structurals(p) = p ? pot & sm : pot & ~sm), so each granule publishes counts for BOTH parities
//     and the chain composes functions {0,1} -> (parity, count);
//   * indexes are expanded into a wave-private LDS slice and leave the CU as aligned 16-byte stores.
// Measured cost model on gfx950 (tools/microbench/valu_rates.hip, valu_mix.hip; profiles/r4/valu_*.jsonl): plain 32-bit
// logic / add / mov / shift-by-constant / v_bitop3_b32 retire in ~2.1 cycles per wave ONLY in streams of their own kind; one
// other instruction among sixteen (v_perm, v_bfi, v_or3, v_lshl_or, v_bcnt, v_cmp, DPP, 64-bit shifts, LDS, loads) already
// makes it 3.0, and this kernel's mix runs at ~4.2 -- so the instruction COUNT is what is tuned: 539 VALU instructions per
// 4 KiB wave-step on twitter.json (SQ_INSTS_VALU, tools/pmc_stage1.sh; 653 with sj_block.h in the loop).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "sj_block.h"
#include "sj_block32.h"
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
// chain granules (u64; one AGGREGATE and one PREFIX array entry per granule of input, zeroed by hipMemsetAsync
// before every launch)
//   bits 63..62 : 0 = nothing yet, 1 = AGGREGATE, 2 = INCLUSIVE PREFIX
//   AGGREGATE   : [19:0] structurals if the granule is entered with parity 0, [39:20] with parity 1,
//                 [40] quote parity of the granule, [41] UTF-8 error in the granule, [42] / [43] unescaped control
//                 character inside a string if the granule is entered with parity 0 / 1, [44] a shard's left halo was too
//                 short to resolve a backslash run
//   PREFIX      : [39:0] structurals in granules 0..t, [40] in-string parity after granule t
// A granule is one naturally aligned 8-byte relaxed agent-scope store/load: the data is the flag
// (cdna_hip_programming.md Guideline 16, form R2), so no fences are needed.
// ---------------------------------------------------------------------------------------------
constexpr sj_u64 TS_AGG = 1ull << 62, TS_PFX = 2ull << 62;
// bounded spins: ~2^19 polls of >= 100 cycles (s_sleep 1 + an uncached load) = tens of milliseconds; a healthy chain needs
// a handful of polls, a launch whose grid is not resident trips this, reports SJMI_ST_INTERNAL and is re-run in SAFE mode
constexpr uint32_t SPIN_LIMIT = 1u << 19;

__device__ __forceinline__ void ts_store(sj_u64* p, sj_u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ sj_u64 ts_load(const sj_u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void publish_aggregate(sj_u64* tile_state, uint32_t tile, uint32_t T0, uint32_t T1,
                                                  uint32_t tpar, uint32_t errbits = 0) {
    ts_store(&tile_state[tile], TS_AGG | ((sj_u64)errbits << 41) | ((sj_u64)tpar << 40) | ((sj_u64)T1 << 20) | (sj_u64)T0);
}
__device__ __forceinline__ void publish_prefix(sj_u64* tile_state, uint32_t tile, uint32_t par_after, sj_u64 cnt_after) {
    ts_store(&tile_state[tile], TS_PFX | ((sj_u64)par_after << 40) | cnt_after);
}


// Executed by all 64 lanes of a wave.  Returns the parity / structural count entering `tile` (> 0).  Reads only.
// The window is 64 * K tiles wide (lane i looks at the K tiles tile-1-i*K-j, j < K): the prefix frontier can
// only advance by one window per store-visibility + load round trip (~1.2 us under load), so the window width
// over that latency has to stay well above the rate at which tiles are produced (40-90 per us).
template <int K>
__device__ __forceinline__ void tile_lookback(const sj_u64* tile_state, uint32_t tile, int lane, uint32_t* par_in,
                                              sj_u64* cnt_in, Stage1Result* res) {
    sj_u64 g0 = 0, g1 = 0;  // structurals in the tiles already folded, if entered with parity 0 / 1
    uint32_t gpar = 0;      // their combined quote parity
    long long k = (long long)tile;
    uint32_t P = 0;
    sj_u64 C = 0;
    for (;;) {
        const long long hi = k - 1 - (long long)lane * K;  // this lane's newest tile
        sj_u64 v[K];
        int jp = K;  // this lane's newest PREFIX (K = none)
        int J = 64;  // first lane that holds a PREFIX
        for (uint32_t spins = 0;; ++spins) {
#pragma unroll
            for (int j = 0; j < K; ++j) v[j] = (hi - j >= 0) ? ts_load(&tile_state[hi - j]) : TS_PFX;  // "before tile 0": prefix (0, 0)
            jp = K;
#pragma unroll
            for (int j = K - 1; j >= 0; --j)
                if ((v[j] >> 62) == 2) jp = j;
            bool ready = true;
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < jp && v[j] == 0) ready = false;
            const sj_u64 pm = __ballot(jp < K);
            J = pm ? __builtin_ctzll(pm) : 64;
            const sj_u64 need = J >= 63 ? ~0ull : ((2ull << J) - 1ull);  // lanes 0..J
            if ((__ballot(ready) & need) == need) break;
            if (spins > SPIN_LIMIT) {  // never expected: a predecessor tile did not publish
                if (lane == 0)
                    __hip_atomic_fetch_or(&res->status, SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        // fold this lane's AGGREGATE tiles newer than the nearest PREFIX, oldest first
        const int nin = lane < J ? K : (lane == J ? jp : 0);
        uint32_t c0 = 0, c1 = 0, lpar = 0;
        sj_u64 pfxv = 0;
#pragma unroll
        for (int j = K - 1; j >= 0; --j) {
            if (j == jp) pfxv = v[j];
            if (j < nin) {
                const uint32_t a0 = (uint32_t)v[j] & 0xFFFFFu, a1 = (uint32_t)(v[j] >> 20) & 0xFFFFFu;
                const uint32_t n0 = c0 + (lpar ? a1 : a0), n1 = c1 + (lpar ? a0 : a1);
                c0 = n0;
                c1 = n1;
                lpar ^= (uint32_t)(v[j] >> 40) & 1u;
            }
        }
        const sj_u64 pb = __ballot(lpar);
        // parity accumulated by the window's tiles OLDER than this lane's tiles (higher lanes)
        const uint32_t q = (lane < 63) ? (uint32_t)__popcll(pb >> (lane + 1)) & 1u : 0u;
        const uint32_t w0 = wave_sum(q ? c1 : c0);  // window entered with parity 0
        const uint32_t w1 = wave_sum(q ? c0 : c1);  // ... with parity 1
        const uint32_t wpar = (uint32_t)__popcll(pb) & 1u;
        if (J < 64) {
            const sj_u64 pv = __shfl(pfxv, J);  // 64-bit shuffle of the PREFIX granule
            P = (uint32_t)(pv >> 40) & 1u;
            C = pv & ((1ull << 40) - 1ull);
            C += P ? w1 : w0;
            P ^= wpar;
            C += P ? g1 : g0;
            P ^= gpar;
            break;
        }
        // no prefix among these tiles: fold the window in front of the suffix and keep walking
        const sj_u64 n0 = (sj_u64)w0 + (wpar ? g1 : g0);
        const sj_u64 n1 = (sj_u64)w1 + (wpar ? g0 : g1);
        g0 = n0;
        g1 = n1;
        gpar ^= wpar;
        k -= 64 * K;
    }
    *par_in = P;
    *cnt_in = C;
}

// The SCANNER: workgroup 0 does nothing but turn the workers' per-granule AGGREGATEs, in order, into per-granule
// inclusive PREFIXes in a second array.  Every granule crosses the chip exactly twice (aggregate: worker ->
// scanner, prefix: scanner -> worker).  Measured alternative: every worker polling a 64..256-granule window of
// uncached granules itself, whose traffic and round trips (one per window of distance to the nearest prefix)
// were the bottleneck.  The workers need a granule's prefix one whole classification after they published its
// aggregate, so the scanner's latency (load + store visibility, ~3 us) is off the critical path; its THROUGHPUT
// is not (200+ granules per us): a single wave managed ~250/us, so the four waves take the windows of 256 granules
// round-robin, do everything that does not depend on the running state (polling, folding 4 granules per lane,
// cross-lane scans for both entry parities) in parallel, and pass the running (parity, count) from window to
// window through LDS, which is the only serial step.
constexpr int SCAN_K = 4;  // granules per lane
struct ScanHandoff {
    uint32_t seq;  // window whose entry state is in P / C; 0xFFFFFFFF = a scanner wave gave up
    uint32_t P;    // in-string parity after the windows scanned so far
    uint32_t err;  // SJMI_ST_UTF8 / SJMI_ST_UNESCAPED found in the granules scanned so far
    sj_u64 C;      // structurals in them
};

__device__ __forceinline__ void scanner_load(sj_u64 v[SCAN_K], const sj_u64* agg, sj_u64 first, uint32_t n) {
#pragma unroll
    for (int j = 0; j < SCAN_K; ++j) v[j] = first + j < n ? ts_load(&agg[first + j]) : TS_AGG;  // past the end: empty aggregates
}
__device__ __forceinline__ bool scanner_ready(const sj_u64 v[SCAN_K]) {
    bool ready = true;
#pragma unroll
    for (int j = 0; j < SCAN_K; ++j) ready &= v[j] != 0;
    return ready;
}
// the lane's granules as one function {entered outside, inside a string} -> (count, parity)
__device__ __forceinline__ void scanner_fold(const sj_u64 v[SCAN_K], uint32_t* c0, uint32_t* c1, uint32_t* par) {
    uint32_t a = 0, b = 0, lp = 0;
#pragma unroll
    for (int j = 0; j < SCAN_K; ++j) {
        const uint32_t a0 = (uint32_t)v[j] & 0xFFFFFu, a1 = (uint32_t)(v[j] >> 20) & 0xFFFFFu;
        const uint32_t n0 = a + (lp ? a1 : a0), n1 = b + (lp ? a0 : a1);
        a = n0;
        b = n1;
        lp ^= (uint32_t)(v[j] >> 40) & 1u;
    }
    *c0 = a;
    *c1 = b;
    *par = lp;
}
// SJMI_ST_* bits of the lane's granules, given the parity entering the first one (the aggregates carry "UTF-8 error" and
// "unescaped control character if entered outside / inside a string": the status of a launch is complete as soon as
// the scanner has seen every aggregate, no worker has to be waited for)
__device__ __forceinline__ uint32_t scanner_errors(const sj_u64 v[SCAN_K], uint32_t q) {
    uint32_t e = 0;
#pragma unroll
    for (int j = 0; j < SCAN_K; ++j) {
        if ((v[j] >> 41) & 1u) e |= SJMI_ST_UTF8;
        if ((v[j] >> 44) & 1u) e |= SJMI_ST_HALO;
        if ((v[j] >> (42 + q)) & 1u) e |= SJMI_ST_UNESCAPED;
        q ^= (uint32_t)(v[j] >> 40) & 1u;
    }
    return e;
}
__device__ __forceinline__ void scanner_report(ScanHandoff* hand, uint32_t e) {
    if (__ballot(e != 0)) {  // rare
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) e |= __shfl_xor(e, d);
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(&hand->err, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// prefixes of the lane's granules, given the parity entering the first one and the structurals before it
__device__ __forceinline__ void scanner_publish(const sj_u64 v[SCAN_K], sj_u64* pfx, sj_u64 first, uint32_t n, uint32_t q,
                                                sj_u64 run) {
#pragma unroll
    for (int j = 0; j < SCAN_K; ++j) {
        run += q ? (uint32_t)(v[j] >> 20) & 0xFFFFFu : (uint32_t)v[j] & 0xFFFFFu;
        q ^= (uint32_t)(v[j] >> 40) & 1u;
        if (first + j < n) publish_prefix(pfx, (uint32_t)(first + j), q, run);
    }
}

// (one lane, behind the result record: see Stage1Single)
__device__ __forceinline__ void single_doc_setup(const Stage1Single& ss, sj_u64 len, sj_u64 count, uint32_t status) {
    if (!ss.index_offsets) return;
    ss.doc_offsets[0] = 0;
    ss.doc_offsets[1] = len;
    ss.index_offsets[0] = 0;
    ss.index_offsets[1] = (status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL)) ? 0ull : count;
    ss.doc_status[0] = status & 0xFFu;
    ss.doc_str_offsets[0] = 0;
    ss.doc_str_offsets[1] = 0;
    for (int i = 0; i < (int)(sizeof(WalkResult) / 4); ++i) ss.walk_result[i] = 0;
    for (int i = 0; i < 16; ++i) ss.slow_header[i] = 0;
}

__device__ __forceinline__ void scanner_wave(ScanHandoff* hand, int wave, const sj_u64* agg, sj_u64* pfx, uint32_t n,
                                             int lane, uint32_t* out, sj_u64 out_cap, Stage1Result* res,
                                             Stage1Result* result_out, const Stage1Single& ss, sj_u64 len) {
    __builtin_amdgcn_s_setprio(3);  // everybody waits for these four waves
    constexpr uint32_t WIN = 64 * SCAN_K;
    const sj_u64 lt_mask = (1ull << lane) - 1ull;
    for (sj_u64 win = (sj_u64)wave; win * WIN < n; win += 4) {
        const sj_u64 first = win * WIN + (sj_u64)lane * SCAN_K;  // this lane's granules
        sj_u64 v[SCAN_K];
        uint32_t P2;
        sj_u64 C2;
        // poll the window until it is complete (then most of the work can be done before the running state arrives) or
        // until the running state has arrived (then the ready part cannot wait for the rest)
        bool full;
        for (;;) {
            scanner_load(v, agg, first, n);
            full = __ballot(scanner_ready(v)) == ~0ull;
            if (full) break;
            const uint32_t seq = __hip_atomic_load(&hand->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (seq == (uint32_t)win || seq == 0xFFFFFFFFu) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (full) {
            // ---- the scanner is behind the workers: the window is complete at the first look.  Everything that does
            //      not depend on the running state is done before waiting for it ----
            uint32_t c0, c1, lp;
            scanner_fold(v, &c0, &c1, &lp);
            const sj_u64 pb = __ballot(lp);
            const uint32_t qrel = (uint32_t)__popcll(pb & lt_mask) & 1u;      // parity of the lanes in front
            const uint32_t mine0 = qrel ? c1 : c0, mine1 = qrel ? c0 : c1;  // window entered outside / inside
            const uint32_t incl0 = wave_incl_scan(mine0, lane), incl1 = wave_incl_scan(mine1, lane);
            const uint32_t tot0 = (uint32_t)__builtin_amdgcn_readlane((int)incl0, 63);
            const uint32_t tot1 = (uint32_t)__builtin_amdgcn_readlane((int)incl1, 63);
            // the serial step: take the running state from the previous window's wave, pass it on
            uint32_t seq;
            do {
                seq = __hip_atomic_load(&hand->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } while (seq != (uint32_t)win && seq != 0xFFFFFFFFu);
            if (seq == 0xFFFFFFFFu) return;
            const uint32_t P = hand->P;
            const sj_u64 C = hand->C;
            P2 = P ^ ((uint32_t)__popcll(pb) & 1u);
            C2 = C + (P ? tot1 : tot0);
            scanner_report(hand, scanner_errors(v, P ^ qrel));
            if (lane == 0) {
                hand->P = P2;
                hand->C = C2;
                __hip_atomic_store(&hand->seq, (uint32_t)win + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            scanner_publish(v, pfx, first, n, P ^ qrel, C + (P ? incl1 - mine1 : incl0 - mine0));
        } else {
            // ---- the scanner is at the workers' frontier: take the running state first, then publish whatever
            //      becomes ready, lane by lane in order (a worker may be waiting for a prefix in the front part of this
            //      window while the back part has not even been handed out) ----
            uint32_t seq;
            do {
                seq = __hip_atomic_load(&hand->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } while (seq != (uint32_t)win && seq != 0xFFFFFFFFu);
            if (seq == 0xFFFFFFFFu) return;
            uint32_t P = hand->P;
            sj_u64 C = hand->C;
            int done = 0;  // lanes already turned into prefixes
            for (uint32_t spins = 0;; ++spins) {
                const sj_u64 rb = __ballot(scanner_ready(v));
                const int nr = ~rb ? __builtin_ctzll(~rb) : 64;  // lanes ready in a row from lane 0
                if (nr > done) {
                    const bool act = lane >= done && lane < nr;
                    uint32_t c0, c1, lp;
                    scanner_fold(v, &c0, &c1, &lp);
                    const sj_u64 pb = __ballot(act && lp);
                    const uint32_t q = P ^ ((uint32_t)__popcll(pb & lt_mask) & 1u);
                    const uint32_t mine = act ? (q ? c1 : c0) : 0u;
                    const uint32_t incl = wave_incl_scan(mine, lane);
                    if (act) scanner_publish(v, pfx, first, n, q, C + (incl - mine));
                    scanner_report(hand, act ? scanner_errors(v, q) : 0u);
                    C += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    P ^= (uint32_t)__popcll(pb) & 1u;
                    done = nr;
                    spins = 0;
                }
                if (done == 64) break;
                if (spins > SPIN_LIMIT) {  // never expected: a worker did not publish
                    if (lane == 0) {
                        __hip_atomic_fetch_or(&res->status, SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&hand->seq, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (result_out) {
                            result_out->count = 0;
                            result_out->status = SJMI_ST_INTERNAL;
                            result_out->reserved = 0;
                            single_doc_setup(ss, len, 0, SJMI_ST_INTERNAL);
                        }
                    }
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
                sj_u64 nv[SCAN_K];
                scanner_load(nv, agg, first, n);
#pragma unroll
                for (int j = 0; j < SCAN_K; ++j)
                    if (lane >= done) v[j] = nv[j];  // (finished lanes keep what their prefixes were computed from)
            }
            P2 = P;
            C2 = C;
            if (lane == 0) {
                hand->P = P2;
                hand->C = C2;
                __hip_atomic_store(&hand->seq, (uint32_t)win + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if ((win + 1) * WIN >= n && lane == 0) {
            // that was the last window: count, sentinel, unclosed string -- and the complete status of the launch
            // (every window's errors were reported before its wave passed the running state on)
            res->count = C2;
            uint32_t e = __hip_atomic_load(&hand->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (P2) e |= SJMI_ST_UNCLOSED;   // StructuralIndexer.java:297-299
            if (C2 < out_cap) out[C2] = 0;   // BitIndexes.finish :82-96
            else e |= SJMI_ST_CAPACITY;      // (== some granule did not fit: they are written in order)
            if (e) __hip_atomic_fetch_or(&res->status, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (result_out) {  // device-resident path: the caller's record, without a copy queued behind the kernel
                const uint32_t st_all = e | __hip_atomic_load(&res->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                result_out->count = C2;
                result_out->status = st_all;
                result_out->reserved = 0;
                single_doc_setup(ss, len, C2, st_all);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the stage-1 kernel: PERSISTENT, WAVE-AUTONOMOUS workers, software-pipelined across granules.
//
// A worker wave owns one granule of S * 4 KiB at a time (S steps of 64 blocks); its indexes are one contiguous run
// of the output.  Waves never synchronise with each other: no barrier, no workgroup-level table -- a workgroup is
// just four waves sharing an LDS allocation.  A per-tile timeline of the earlier one-tile-per-workgroup kernel
// (tools/trace.py) showed ~8 of every workgroup's ~20 us spent in two bubbles: the first load of a fresh
// workgroup (~2.7 us) and the wait between "aggregate published" and "prefix known" (~2 us for the slowest
// predecessor + ~3 us of cross-XCD visibility and load latency); that kernel's time followed
// T = 0.15 ms + 7 ns * tiles -- the bubbles over the ~1000 resident workgroups.  Here both are overlapped:
//
//   iteration k of a wave:   C(k)    classify granule k (its first step was loaded during E(k-2)), publish its
//                                    AGGREGATE; before the last step, request the PREFIX in front of granule k-1
//                            E(k-1)  expand granule k-1's masks (parked in LDS) and store its indexes
//                            park granule k's masks / offsets in the wave's LDS slice (20 bytes per block)
//
// so a granule's prefix is needed one whole classification after its aggregate went out, and it comes from the
// scanner workgroup (above), not from a look-back by the worker.
//
// Granule assignment, FAST mode (default): the first granule of a wave is its worker index, all further ones come
// from atomic tickets (8 counters, requested one step ahead; see the kernel).  Classification runs at raised wave
// priority, workers that share the scanner's CU retire after one granule.  RESULTS never depend on scheduling;
// LIVENESS does: the workers spin on prefixes, so every worker wave and the scanner have to be resident (the host
// sizes the grid with the occupancy API; the scanner is workgroup 0).  If that ever fails, the bounded spins trip,
// the launch reports SJMI_ST_INTERNAL and the host re-runs it in SAFE mode: a separate instantiation, every granule
// by one atomic ticket and a decoupled look-back by the worker itself (a granule then only waits for granules some
// running wave already holds), ~2.5x slower but free of residency assumptions.
// Variants measured and dropped: one tile per workgroup with the look-back on the critical
// path (2.8 TB/s), two chain granules per workgroup with a late look-back, persistent workgroups with a barrier
// per tile and look-back windows of 64..256 granules (uncached polling traffic), 16 KiB workgroup tiles
// (ticket-bound), wave-autonomous tiles with per-wave look-back (chain rate ~100 tiles/us).
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
// (Measured and dropped, round 4: 32-bit offsets, i.e. loads in the "scalar base + 32-bit vector offset" form.  They save a dozen
// slow-class VALU instructions of 64-bit address arithmetic per step -- and the kernel is 3 % SLOWER, 0.1907 against 0.1850 ms for
// twitter x1024, tools/ab_stage1.sh.)
__device__ __forceinline__ void load_step(StepData& d, const uint8_t* __restrict__ buf, sj_u64 blk, sj_u64 nblocks, bool left_halo = false) {
    const sj_u64 b = blk < nblocks ? blk : nblocks - 1;
    const uint4* src = reinterpret_cast<const uint4*>(buf + b * 64);
    d.q0 = src[0];
    d.q1 = src[1];
    d.q2 = src[2];
    d.q3 = src[3];
    // (unused for block 0 -- unless the buffer is a shard of a longer document: then the 8 bytes in front of it are real)
    d.halo = *reinterpret_cast<const sj_u64*>(buf + ((b > 0 || left_halo) ? (long long)(b * 64) - 8 : 0ll));
}

#ifdef SJMI_TRACE  // experiments only (tools/trace.py): per-granule timestamps behind the granule states
#define SJMI_TRACE_SLOTS 6
#define SJMI_TSTAMP(g_, k_)                                                                                    \
    do {                                                                                                       \
        if (lane == 0) (gstate + 2 * (sj_u64)ngran + (sj_u64)(g_) * SJMI_TRACE_SLOTS)[k_] = wall_clock64();     \
    } while (0)
#else
#define SJMI_TRACE_SLOTS 0
#define SJMI_TSTAMP(g_, k_) do { } while (0)
#endif

constexpr uint32_t NO_TILE = 0xFFFFFFFFu;

// Device-resident path: nothing but the kernel is queued per launch.  Every worker wave, on its way out, zeroes its
// slice of the OTHER workspace half, which the next launch of this context will use (the workspace memset was a
// separate 5 us kernel in front of every launch), and the scanner writes the caller's result record (that copy was a
// third queue entry behind it).  (A "last wave out copies the result" scheme was tried first: 4096 atomics on one
// counter at the end of the kernel cost 30 us, an acq_rel one each -- L2 write-back + invalidate -- 120 us.)
__device__ __forceinline__ void zero_next_workspace(uint4* zero_ptr, uint32_t zero_chunks, uint32_t slot, uint32_t nslots,
                                                    int lane) {
    if (zero_ptr && nslots) {
        const uint32_t per = (zero_chunks + nslots - 1) / nslots;
        const uint32_t b = slot * per, e = b + per < zero_chunks ? b + per : zero_chunks;
        for (uint32_t i = b + (uint32_t)lane; i < e; i += 64) zero_ptr[i] = make_uint4(0, 0, 0, 0);
    }
}

constexpr uint32_t TICKET_CLASSES = 8;
constexpr int LB_K = 1;  // SAFE mode's look-back window = 64 * LB_K granules

#ifndef SJMI_S1_SORT
#define SJMI_S1_SORT 1       // the expansion's half masks handed out by population (stage1_body, sorted_round)
#endif
#ifndef SJMI_S1_SORT_PLAIN
#define SJMI_S1_SORT_PLAIN 1 // ... in k_stage1 as well as in k_stage1_batch
#endif
#ifndef SJMI_S1_TWO_ENDED
#define SJMI_S1_TWO_ENDED 1  // the sorted round's per-bit loop takes the lowest AND the highest set bit a trip (k_stage1_batch 366 -> 356 us; the headline does not notice)
#endif
#ifndef SJMI_S1_NT_STORE
#define SJMI_S1_NT_STORE 4  // 0: plain index stores; 1: streaming stores everywhere; 4: in the plain 16 KiB-granule kernels only
#endif
#ifndef SJMI_S1_SORT_MIN
#define SJMI_S1_SORT_MIN 192 // ... for a round of more than this many indexes per 4 KiB step (below: the sort costs more than it saves)
#endif
// S = 4 KiB steps per granule, LDSW = bytes of LDS per wave
template <int S>
struct Parked {
    // per-block state of the granule awaiting its prefix, [step][lane]: every lane reads back exactly what it wrote
    // (LDS as a register file extension); lane 0's meta doubles as the step's base offset
    sj_u64 pot[S][64];     // potential structurals (StructuralIndexer.java:245-250 before the string mask)
    sj_u64 m0[S][64];      // structurals if the granule is entered outside a string; inside: pot ^ m0
    uint32_t meta[S][64];  // [13:0] granule-relative offset of m0's indexes, [27:14] of pot's, [28]/[29] unescaped-
                           // character error if the granule is entered outside / inside a string
};
template <int S, int LDSW>
union WaveShared {
    Parked<S> park;
    // staging of indexes for coalesced stores: overlays the parked state, which is dead from the moment the
    // expansion has read it into registers until the next granule is parked
    alignas(16) uint32_t stage[LDSW / 4];
    static_assert(sizeof(Parked<S>) <= LDSW, "LDS slice too small");
};

// BATCH: two more side outputs for the fused batch pipeline (batch.hip k_doc_prepare), per 64-byte block: blkidx = position in
// the index array of the block's first structural (what a document's index_offsets entry is read from instead of a binary
// search), blkw = the tape words its structurals make for both entry parities (sj_block_tape_words).
template <int S, int LDSW, bool SAFE, bool BATCH>
__device__ __forceinline__ void
stage1_body(const uint8_t* __restrict__ buf, sj_u64 len, uint32_t* __restrict__ out, sj_u64 out_cap,
            sj_u64* gstate, uint32_t* ticket, Stage1Result* res, uint32_t ngran, uint32_t dbg, uint4* zero_ptr,
            uint32_t zero_chunks, Stage1Result* result_out, sj_u64* __restrict__ blkpar, const uint32_t* __restrict__ skip,
            uint32_t* __restrict__ blkidx, uint16_t* __restrict__ blkw, uint4* zero2_ptr, uint32_t zero2_chunks, const Stage1Single& ss) {
    constexpr int E = S, CAP = LDSW / 4;
    if (skip && *skip) return;  // (fused batch pipeline: this pass is not needed; uniform for the whole launch)
    static_assert(S <= 4, "meta packs 14-bit offsets: at most 4 steps per granule");
    __shared__ WaveShared<S, LDSW> sh[4];
    __shared__ ScanHandoff hand;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool safe = SAFE;  // a separate instantiation: the look-back code costs the fast kernel 8+ VGPRs
    sj_u64* const agg = gstate;         // AGGREGATE per granule (SAFE mode: overwritten by its PREFIX)
    sj_u64* const pfx = gstate + ngran;  // the scanner's prefixes (FAST mode)
    // (xcc, se, sh, cu) of the CU this workgroup runs on, never 0
    const uint32_t my_cu = (((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xFFu) |
                           (((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 8) | 0x80000000u;
    uint32_t* const scanner_cu = reinterpret_cast<uint32_t*>(res) + WS_SCANNER_CU_WORD;
    if (!safe && blockIdx.x == 0) {
        if (threadIdx.x == 0) __hip_atomic_store(scanner_cu, my_cu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((dbg & DBG_FAKE_TIMEOUT) && threadIdx.x == 0)
            __hip_atomic_fetch_or(&res->status, SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            hand.seq = 0;
            hand.P = (dbg & FLAG_ENTRY_PARITY) ? 1u : 0u;  // (a shard of a longer document may start inside a string)
            hand.err = 0;
            hand.C = 0;
        }
        __syncthreads();
        if (!(dbg & DBG_NO_LOOKBACK)) scanner_wave(&hand, wave, agg, pfx, ngran, lane, out, out_cap, res, result_out, ss, len);
        return;
    }
    const uint32_t nworkers = (gridDim.x - (safe ? 0u : 1u)) * 4u;
    const uint32_t worker = (blockIdx.x - (safe ? 0u : 1u)) * 4u + (uint32_t)wave;
    WaveShared<S, LDSW>& ws = sh[wave];
    // the reference always processes one tail block (:255-294); a shard that is not the document's last one ends on a block
    // boundary and has none (its successor validates what straddles the boundary from its own left halo)
    const uint32_t nblocks = (uint32_t)(len / 64) + ((dbg & FLAG_NO_TAIL) ? 0u : 1u);  // (len < 4 GiB)
    const uint32_t halo_blocks = dbg >> 16;       // shard: readable 64-byte blocks in front of buf (0 = a whole document)
    const bool left_halo = halo_blocks != 0;
    const uint32_t entry_par = (dbg & FLAG_ENTRY_PARITY) ? 1u : 0u;
    const sj_u64 lt_mask = (1ull << lane) - 1ull;
    const uint32_t last = ngran - 1;

    // Granule assignment.  FAST: dynamic, so that a wave that is slowed down (three neighbours on its SIMD in their
    // ALU-heavy phase) simply takes fewer granules instead of holding up the chain for everybody (static striding
    // made every iteration a chip-wide barrier in effect).  One atomic counter saturates at ~88 tickets/us on
    // gfx950 and 200+ granules/us are needed, so the waves are split into NC classes by worker index, class c
    // owning the granules == c (mod NC) and its own counter (64 bytes apart); the classes are statistically
    // identical and the chain's back-pressure keeps them together (measured: 6 or 8 classes are equally fast, 4
    // are ticket-bound, 32 drift apart: +2 %).  Only the FIRST granule of every wave is static (= its worker index,
    // the counters start behind those): taking it by ticket as well would make a launch whose workgroups are not all
    // resident degrade gracefully instead of timing out into SAFE mode, but costs 6 % (A/B: tools/archive/abtest.sh).
    // SAFE: one counter, every granule by ticket, taken when needed (any running wave can take any granule).
    const uint32_t NC = safe ? 1u : (nworkers < TICKET_CLASSES ? nworkers : TICKET_CLASSES);
    const uint32_t cls = worker % NC;
    uint32_t* const my_ticket = ticket + cls * 16u;                 // one counter per 64-byte line
    // The scanner's high-priority waves slow the workers that share its CU to ~2/3 speed, and a slow worker holds
    // up the whole chain (per-CU timeline: tools/trace.py), so those workers retire after their first granule.
    uint32_t retire = 0;
    if (!safe && gridDim.x >= 64)
        retire = __hip_atomic_load(scanner_cu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ my_cu;  // 0 = same CU (used a granule later)
    else
        retire = 1;
    // FLAG_ALL_TICKETS (set by the host when several contexts may launch at the same time): the first granule comes
    // from a ticket as well, so that a workgroup that is not resident yet holds nothing anybody could wait for
    const bool first_by_ticket = safe || (dbg & FLAG_ALL_TICKETS) != 0;
    uint32_t cur;
    const uint32_t ticket_base = first_by_ticket ? 0u : (nworkers - cls + NC - 1u) / NC;
    if (first_by_ticket) {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(my_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) * NC + cls;
    } else cur = worker;
    uint32_t prev = NO_TILE, prev_par = 0, prev_c0 = 0, prev_c1 = 0;  // the parked granule and its aggregate
    uint32_t err = 0;
    StepData d;  // the step being loaded / classified (single buffer: re-used as soon as it has been transposed)
    load_step(d, buf, cur * (64u * S) + (uint32_t)lane, nblocks, left_halo);

    for (;;) {
        const bool have = cur < ngran;  // wave-uniform
        uint32_t tk = 0;  // FAST mode: the ticket of the next iteration, requested before the last step below
        sj_u64 pot[S], m[S];
        uint32_t meta[S];
        uint32_t wpar = 0, W0 = 0, WP = 0;
        sj_u64 pf = 0;  // FAST mode: the prefix in front of `prev`, requested before the last step below

        // =================== C: classify granule `cur`, publish its aggregate ===================
        if (have) {
            // a granule that is not classified yet holds up every granule behind it; one that is being expanded holds
            // up nobody: classification gets the SIMD first (equal priorities go oldest wave first, for ever the same
            // wave last in a persistent kernel: classifications of 25 us instead of 9 in the timeline)
            __builtin_amdgcn_s_setprio(2);
            SJMI_TSTAMP(cur, 0);
            const uint32_t blk0 = cur * (64u * S);  // (block numbers are 32-bit: load_step)
            sj_u64 sm[S];
            uint32_t fl[S];     // bit0 quote parity, bit1 ue0, bit2 ue1, bit3 utf8 error
            uint32_t slow = 0;  // steps whose carries the halo could not resolve (long backslash run)
            bool halo_short = false;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                // software pipeline with ONE buffer: step s was loaded during the algebra of step s-1 (step 0 before
                // the previous granule's expansion); as soon as it is transposed into planes, its registers take the
                // loads of step s+1.  (Ping-pong buffers cost 18 more VGPRs = one wave per SIMD at S = 4.)
                const uint32_t blk = blk0 + (uint32_t)s * 64u + (uint32_t)lane;
                const uint32_t w[16] = {d.q0.x, d.q0.y, d.q0.z, d.q0.w, d.q1.x, d.q1.y, d.q1.z, d.q1.w,
                                        d.q2.x, d.q2.y, d.q2.z, d.q2.w, d.q3.x, d.q3.y, d.q3.z, d.q3.w};
                const sj_u64 halo = d.halo;
                uint32_t plo[8], phi[8];  // the block's eight bit planes, bytes 0..31 / 32..63 (sj_block32.h)
                sj_transpose32(w, plo, phi);
                asm volatile("" ::: "memory");
                if (s + 1 < S) load_step(d, buf, blk0 + (uint32_t)(s + 1) * 64u + (uint32_t)lane, nblocks, left_halo);
                if (s == S - 1 && !safe) {
                    // requested one step ahead of their use: late enough that granules are started in ticket order
                    // (a ticket held through a whole slow iteration delays every granule behind it), early enough to
                    // hide the round trips
                    if (lane == 0 && retire != 0) tk = __hip_atomic_fetch_add(my_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (prev != NO_TILE && prev != 0) pf = ts_load(&pfx[prev - 1]);
                }
                asm volatile("" ::: "memory");
                pot[s] = 0;
                sm[s] = 0;
                fl[s] = 0;
                bool unresolved = false;
                if (blk < nblocks) {
                    const bool has_halo = blk > 0 || left_halo;
                    uint32_t e_in = 0, p_in = 0;
                    if (has_halo) unresolved = !sj_carry_from_halo(halo, &e_in, &p_in);
                    // (wave-uniform: only the wave-step that holds the document's last block masks a tail)
                    if (blk0 + (uint32_t)s * 64u + 63u >= nblocks - 1u) {
                        const sj_u64 rem = len - (sj_u64)blk * 64;
                        sj_mask_tail32(plo, phi, rem < 64 ? (uint32_t)rem : 64u);
                    }
                    // the UTF-8 algebra AND the carries into it are skipped when no lane of the wave has a non-ASCII byte in its
                    // block or in the four bytes before it (a pending carry needs a byte >= 0xC0 there)
                    const uint32_t halo_hi = has_halo ? (uint32_t)(halo >> 32) & 0x80808080u : 0u;
                    const bool need_utf8 = __ballot((plo[7] | phi[7] | halo_hi) != 0) != 0;
                    SjUtf8Lazy uc = {0, 0};
                    if (need_utf8 && has_halo) uc = sj_utf8_carry_lazy(halo);
                    const SjBlockMasks32 bm = sj_block32(plo, phi, e_in, p_in, uc, need_utf8, BATCH, [](uint32_t x) { return __ballot(x != 0) != 0; });
                    pot[s] = ((sj_u64)bm.pot.hi << 32) | bm.pot.lo;
                    sm[s] = ((sj_u64)bm.sm0.hi << 32) | bm.sm0.lo;
                    // (min: 0 / 1 from zero / nonzero in one instruction)
                    fl[s] = bm.qpar | (min(bm.ue0, 1u) << 1) | (min(bm.ue1, 1u) << 2) | (min(bm.utf8, 1u) << 3);
                    if (BATCH) blkw[blk] = (uint16_t)bm.words;
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
                    // (a shard: the run may reach into the left halo, whose first byte bounds the walk)
                    sj_carry_slow(buf - (sj_u64)halo_blocks * 64, 0, start + (sj_u64)halo_blocks * 64, &e_in, &p_in);
                    if (left_halo && !(dbg & FLAG_HALO_FROM_START)) {  // the run must begin inside the halo, or its parity is not known
                        const uint8_t* hb = buf - (sj_u64)halo_blocks * 64;
                        const sj_u64 hs = start + (sj_u64)halo_blocks * 64;
                        if (sj_backslash_run_reaches(hb, 0, hb[hs - 1] == 0x22 ? hs - 1 : hs)) halo_short = true;
                    }
                    sj_u64 p[8];
                    sj_transpose_butterfly(w, p);
                    const sj_u64 rem = len - start;
                    sj_mask_tail(p, rem < 64 ? (uint32_t)rem : 64u);
                    const SjBlockMasks bm = sj_block(p, e_in, p_in, sj_utf8_carry(halo), true, nullptr, BATCH);
                    pot[s] = bm.pot;
                    sm[s] = bm.sm0;
                    fl[s] = bm.qpar | (bm.ue0 << 1) | (bm.ue1 << 2) | (bm.utf8 << 3);
                    if (BATCH && blk < nblocks) blkw[blk] = (uint16_t)bm.words;
                }
            }
            // in-string parity, structurals and their offsets, all RELATIVE TO THE GRANULE being entered outside a
            // string (order inside the granule: step, lane)
            uint32_t gerr = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const sj_u64 bal = __ballot(fl[s] & 1u);
                const uint32_t lp = ((uint32_t)__popcll(bal & lt_mask) & 1u) ^ wpar;  // parity entering the block
                wpar ^= (uint32_t)__popcll(bal) & 1u;
                m[s] = lp ? (pot[s] & sm[s]) : (pot[s] & ~sm[s]);  // StructuralIndexer.java:251
                const uint32_t c0 = (uint32_t)__popcll(m[s]), cp = (uint32_t)__popcll(pot[s]);
                const uint32_t packed = wave_incl_scan(c0 | (cp << 16), lane);  // both <= 4096 per step: no carry
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)packed, 63);
                const uint32_t ex0 = W0 + (packed & 0xFFFFu) - c0;  // granule-relative exclusive offsets
                const uint32_t exp_ = WP + (packed >> 16) - cp;
                W0 += tot & 0xFFFFu;
                WP += tot >> 16;
                const uint32_t ue0 = (fl[s] >> 1) & 1u, ue1 = (fl[s] >> 2) & 1u;  // :252 for entry parity 0 / 1
                if (fl[s] & 8u) err |= SJMI_ST_UTF8;
                // ([30]: the block is entered inside a string if the granule is entered outside one -- the side output for
                //  the string pass, strings.hip, written once the granule's own entry parity is known)
                meta[s] = ex0 | (exp_ << 14) | ((lp ? ue1 : ue0) << 28) | ((lp ? ue0 : ue1) << 29) | (lp << 30);
                gerr |= ((fl[s] >> 3) & 1u) | ((meta[s] >> 27) & 6u);  // utf8, unescaped if entered outside / inside
            }
            // the granule's error bits travel with its aggregate (the scanner composes the launch's status from them)
            uint32_t gbits = (__ballot(gerr & 1u) ? 1u : 0u) | (__ballot(gerr & 2u) ? 2u : 0u) | (__ballot(gerr & 4u) ? 4u : 0u);
            if (left_halo && __ballot(halo_short)) {
                gbits |= 8u;
                err |= SJMI_ST_HALO;
            }
            if (lane == 0 && !(dbg & DBG_NO_LOOKBACK)) {
                if (safe && cur == 0) publish_prefix(agg, 0, wpar ^ entry_par, (sj_u64)(entry_par ? WP - W0 : W0));  // nothing to look back at
                else publish_aggregate(agg, cur, W0, WP - W0, wpar, gbits);
            }
            SJMI_TSTAMP(cur, 1);
            __builtin_amdgcn_s_setprio(0);
        }
        uint32_t nxt = NO_TILE;
        if (have) {
            if (safe) {
                uint32_t t = 0;
                if (lane == 0) t = __hip_atomic_fetch_add(my_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            } else {
                nxt = retire != 0 ? ((uint32_t)__builtin_amdgcn_readfirstlane((int)tk) + ticket_base) * NC + cls : NO_TILE;
            }
        }
        // first step of the next granule: in flight during the expansion below (clamped, so harmless without one)
        load_step(d, buf, nxt * (64u * S) + (uint32_t)lane, nblocks, left_halo);  // (NO_TILE wraps: any block will do)

        // =================== E: resolve granule `prev`'s prefix, expand and store its indexes ===================
        if (prev != NO_TILE) {
            uint32_t pe = entry_par;  // parity entering the granule (granule 0: the document's / shard's own)
            sj_u64 cnt_in = 0;
            SJMI_TSTAMP(prev, 2);
            if (dbg & DBG_NO_LOOKBACK) {  // ablation: no chain (indexes land at fake offsets)
                cnt_in = (sj_u64)prev * 512ull * S;  // 1 index slot per 8 input bytes (ablation buffers are sized for it)
            } else if (prev != 0) {
                if constexpr (safe) {
                    tile_lookback<LB_K>(agg, prev, lane, &pe, &cnt_in, res);
                } else {
                    for (uint32_t spins = 0; (pf >> 62) != 2; ++spins) {
                        if (spins > SPIN_LIMIT) {  // never expected: the scanner is not running
                            if (lane == 0)
                                __hip_atomic_fetch_or(&res->status, SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        if (spins) __builtin_amdgcn_s_sleep(1);
                        pf = ts_load(&pfx[prev - 1]);
                    }
                    pe = (uint32_t)(pf >> 40) & 1u;
                    cnt_in = pf & ((1ull << 40) - 1ull);
                }
            }
            SJMI_TSTAMP(prev, 3);
            const uint32_t WT_all = pe ? prev_c1 : prev_c0;  // indexes of the granule
            if (safe && lane == 0 && !(dbg & DBG_NO_LOOKBACK)) {
                if (prev != 0) publish_prefix(agg, prev, pe ^ prev_par, cnt_in + WT_all);
                if (prev == last) {
                    // the granule that holds the final (tail) block finishes the job: count, sentinel, unclosed string
                    const sj_u64 total = cnt_in + WT_all;
                    res->count = total;
                    uint32_t e = 0;
                    if (pe ^ prev_par) e |= SJMI_ST_UNCLOSED;  // StructuralIndexer.java:297-299
                    if (total < out_cap) out[total] = 0;       // BitIndexes.finish :82-96
                    else e |= SJMI_ST_CAPACITY;
                    if (e) __hip_atomic_fetch_or(&res->status, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const bool fits = cnt_in + WT_all < out_cap;  // strict: keeps room for the sentinel
            if (!fits && WT_all) err |= SJMI_ST_CAPACITY;
            const sj_u64 pblk0 = (sj_u64)prev * (64 * S);
            uint32_t* stage = ws.stage;
#pragma unroll
            for (int g = 0; g < S / E; ++g) {
                // ---- final masks + error flags of E steps ----
                sj_u64 mk[E];
                uint32_t pos[E];  // group-relative position of the lane's next index
                const uint32_t mb = ws.park.meta[g * E][0];
                const uint32_t gbase = pe ? ((mb >> 14) & 0x3FFFu) - (mb & 0x3FFFu) : (mb & 0x3FFFu);
                uint32_t gend = WT_all;
                if (g + 1 < S / E) {
                    const uint32_t me = ws.park.meta[(g + 1) * E][0];
                    gend = pe ? ((me >> 14) & 0x3FFFu) - (me & 0x3FFFu) : (me & 0x3FFFu);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int s = g * E + e;
                    const sj_u64 pt = ws.park.pot[s][lane], mm = ws.park.m0[s][lane];
                    const uint32_t mt = ws.park.meta[s][lane];
                    mk[e] = pe ? (pt ^ mm) : mm;
                    const uint32_t o0 = mt & 0x3FFFu, op = (mt >> 14) & 0x3FFFu;
                    pos[e] = (pe ? op - o0 : o0) - gbase;
                    if (BATCH) blkidx[(pblk0 + (sj_u64)s * 64) + lane] = (uint32_t)(cnt_in + gbase) + pos[e];  // (array padded to whole granules)
                    if ((mt >> (28 + pe)) & 1u) err |= SJMI_ST_UNESCAPED;  // :252,:300-302
                    if (blkpar) {  // (wave-uniform) StructuralIndexer.java:233-234's prevInString, for every block
                        const sj_u64 pm = __ballot(((mt >> 30) ^ pe) & 1u);
                        if (lane == 0) blkpar[(sj_u64)prev * S + s] = pm;
                    }
                }
                wave_lds_fence();  // the parked state is in registers now: its LDS becomes the staging buffer
                // ---- index emission (BitIndexes.write :14-41): expand the masks into the wave's LDS slice at
                //      their relative positions, then store them with coalesced stores ----
                if (fits && !(dbg & DBG_NO_WRITE)) {
                    const uint32_t WT = gend - gbase;
                    uint32_t* dst0 = out + cnt_in + gbase;
                    // One round of the fast form: steps [e0, e1) of the group, whose rcount indexes begin rbase entries into the
                    // group, fit the staging slots, so the per-bit loops need no window test.  Entries are staged at the same
                    // position modulo 4 as their final index, so that whole 16-byte quads of the LDS slice go out as
                    // global_store_dwordx4 (the index array is 16-byte aligned); only the two boundary quads of the run need
                    // element-wise stores.
                    auto flush_round = [&](uint32_t rbase, uint32_t rcount, uint32_t g0) {
                        const uint32_t span = g0 + rcount;
                        uint32_t* gbp = dst0 + rbase - g0;  // 16-byte aligned
                        for (uint32_t qi = lane; qi * 4 < span; qi += 64) {
                            const uint4 v = reinterpret_cast<const uint4*>(stage)[qi];
                            const uint32_t lo = qi * 4;
                            if (lo >= g0 && lo + 4 <= span) {
                                // (16 KiB granules = large inputs: the index array is a stream nobody reads before it has left the
                                //  caches anyway -- streaming stores: +1.4 % on the headline, +1.5 % on twitter x1024; small documents keep
                                //  theirs in L2 for the kernels behind this one, and the batch flavour is 3.5 % SLOWER with them:
                                //  profiles/r6/README.md)
                                if constexpr (SJMI_S1_NT_STORE == 1 || (SJMI_S1_NT_STORE == 4 && S == 4 && !BATCH)) {
                                    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                                    u32x4_t nv = {v.x, v.y, v.z, v.w};
                                    __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t*>(gbp) + qi);
                                } else {
                                    reinterpret_cast<uint4*>(gbp)[qi] = v;
                                }
                            } else {
                                if (lo + 0 >= g0 && lo + 0 < span) gbp[lo + 0] = v.x;
                                if (lo + 1 >= g0 && lo + 1 < span) gbp[lo + 1] = v.y;
                                if (lo + 2 >= g0 && lo + 2 < span) gbp[lo + 2] = v.z;
                                if (lo + 3 >= g0 && lo + 3 < span) gbp[lo + 3] = v.w;
                            }
                        }
                        wave_lds_fence();
                    };
                    auto fast_round = [&](int e0, int e1, uint32_t rbase, uint32_t rcount) {
                        const uint32_t g0 = (uint32_t)((cnt_in + gbase + rbase) & 3ull);
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            if (e < e0 || e >= e1) continue;  // (wave-uniform; e itself stays a constant: mk / pos are registers)
                            const uint32_t bstart = (uint32_t)((pblk0 + (sj_u64)(g * E + e) * 64 + lane) * 64);
                            uint32_t* q = stage + g0 + (pos[e] - rbase);
                            for (uint32_t lo = (uint32_t)mk[e]; lo; lo &= lo - 1) *q++ = bstart | (uint32_t)__builtin_ctz(lo);
                            for (uint32_t hi = (uint32_t)(mk[e] >> 32); hi; hi &= hi - 1)
                                *q++ = bstart | 32u | (uint32_t)__builtin_ctz(hi);
                        }
                        wave_lds_fence();
                        flush_round(rbase, rcount, g0);
                    };
                    // The per-bit loops above run as long as the BUSIEST lane of each of their 2 (e1 - e0) trips has bits: 151 trips of
                    // the loop body per 16 KiB of the configs[3] documents for a mean of 48 indexes per lane (56 for 22 on
                    // twitter.json) -- a third of the lanes at work.  The sorted form hands the round's 32-bit half masks out again by
                    // population: a counting sort in LDS (histogram by LDS atomics = the rank inside a bucket, one scan over the 33
                    // buckets, every half mask with its staging slot and its block stored at its rank), then lane l takes entries
                    // l, l + 64, ...: a trip's 64 half masks have (nearly) the same number of bits, the empty ones come last and
                    // their trips are not run (tools/README.md: expansion statistics; profiles/r6/README.md has the A/B).
                    auto sorted_round = [&](auto e0c, auto e1c, uint32_t rbase, uint32_t rcount) {
                        constexpr int E0 = decltype(e0c)::value, E1 = decltype(e1c)::value, NI = 2 * (E1 - E0);
                        static_assert(NI * 128 + 128 <= CAP, "the sort's tables overlay the staging slots");
                        const uint32_t g0 = (uint32_t)((cnt_in + gbase + rbase) & 3ull);
                        uint32_t* const hist = stage + NI * 128;  // [64]: half masks by population (0 .. 32)
                        uint32_t* const bbase = hist + 64;        // [64]: first rank of a bucket, the fullest bucket first
                        uint2* const items = reinterpret_cast<uint2*>(stage);  // [NI * 64]: .x = half mask, .y = byte offset of its first staging slot | (its first byte's offset in the granule) << 16
                        hist[lane] = 0u;
                        wave_lds_fence();
                        uint32_t hm[NI], cc[NI], rk[NI], inf[NI];
#pragma unroll
                        for (int e = E0; e < E1; ++e) {
                            const int i = 2 * (e - E0);
                            hm[i] = (uint32_t)mk[e];
                            hm[i + 1] = (uint32_t)(mk[e] >> 32);
                            cc[i] = (uint32_t)__popc(hm[i]);
                            cc[i + 1] = (uint32_t)__popc(hm[i + 1]);
                            const uint32_t p4 = (pos[e] - rbase + g0) * 4u;
                            const uint32_t ib = (uint32_t)(((g * E + e) * 64 + lane) * 64) << 16;
                            inf[i] = p4 | ib;
                            inf[i + 1] = (p4 + 4u * cc[i]) | (ib + (32u << 16));
                            rk[i] = atomicAdd(&hist[cc[i]], 1u);
                            rk[i + 1] = atomicAdd(&hist[cc[i + 1]], 1u);
                        }
                        wave_lds_fence();
                        const uint32_t hv = lane <= 32 ? hist[32 - lane] : 0u;
                        const uint32_t incl = wave_incl_scan(hv, lane);
                        if (lane <= 32) bbase[32 - lane] = incl - hv;
                        const uint32_t nz = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);  // half masks with a bit
                        wave_lds_fence();
#pragma unroll
                        for (int i = 0; i < NI; ++i) items[bbase[cc[i]] + rk[i]] = make_uint2(hm[i], inf[i]);
                        wave_lds_fence();
                        const uint32_t trips = (nz + 63u) / 64u;
                        uint2 it[NI];
#pragma unroll
                        for (int r = 0; r < NI; ++r) it[r] = items[r * 64 + lane];
                        wave_lds_fence();
                        const uint32_t b32 = (uint32_t)(pblk0 * 64);
#pragma unroll
                        for (int r = 0; r < NI; ++r) {
                            if ((uint32_t)r >= trips) break;  // (wave-uniform)
                            uint32_t* q = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(stage) + (it[r].y & 0xFFFFu));
                            const uint32_t bv = b32 + (it[r].y >> 16);
#if SJMI_S1_TWO_ENDED
                            // two indexes a trip: the lowest set bit to the front slot, the highest to the back one (a last single
                            // bit goes to the same slot twice)
                            uint32_t lo = it[r].x;
                            uint32_t* qh = q + __popc(lo) - 1;
                            while (lo) {
                                const uint32_t lz = (uint32_t)__builtin_clz(lo);
                                *q++ = bv | (uint32_t)__builtin_ctz(lo);
                                *qh-- = bv | (lz ^ 31u);
                                lo = lo & (lo - 1) & ~(0x80000000u >> lz);
                            }
#else
                            for (uint32_t lo = it[r].x; lo; lo &= lo - 1) *q++ = bv | (uint32_t)__builtin_ctz(lo);
#endif
                        }
                        wave_lds_fence();
                        flush_round(rbase, rcount, g0);
                    };
                    // (a structural every ~5 bytes -- the documents of configs[3] -- makes 3,000 per 16 KiB granule: more than
                    //  the 2,304 slots, but each half of the granule fits)
                    const uint32_t split = E >= 2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)pos[E / 2]) : 0u;
                    using std::integral_constant;
                    constexpr bool do_sort = SJMI_S1_SORT && (BATCH || SJMI_S1_SORT_PLAIN);
                    bool staged = false;
                    if (WT + 3 <= (uint32_t)CAP) {
                        if (do_sort && WT > (uint32_t)(SJMI_S1_SORT_MIN * E)) sorted_round(integral_constant<int, 0>{}, integral_constant<int, E>{}, 0, WT);
                        else fast_round(0, E, 0, WT);
                        staged = true;
                    }
                    if constexpr (E >= 2) {
                        if (!staged && split + 3 <= (uint32_t)CAP && WT - split + 3 <= (uint32_t)CAP) {
                            if (do_sort) {
                                sorted_round(integral_constant<int, 0>{}, integral_constant<int, E / 2>{}, 0, split);
                                sorted_round(integral_constant<int, E / 2>{}, integral_constant<int, E>{}, split, WT - split);
                            } else {
                                fast_round(0, E / 2, 0, split);
                                fast_round(E / 2, E, split, WT - split);
                            }
                            staged = true;
                        }
                    }
                    if (!staged) {
                        for (uint32_t base = 0; base < WT; base += CAP) {
                            const uint32_t lim = base + CAP;
#pragma unroll
                            for (int e = 0; e < E; ++e) {
                                const uint32_t bstart = (uint32_t)((pblk0 + (sj_u64)(g * E + e) * 64 + lane) * 64);
                                uint32_t lo = (uint32_t)mk[e], hi = (uint32_t)(mk[e] >> 32);
                                uint32_t p2 = pos[e];
                                while (lo && p2 < lim) {
                                    stage[p2 - base] = bstart + (uint32_t)__builtin_ctz(lo);
                                    lo &= lo - 1;
                                    ++p2;
                                }
                                while (!lo && hi && p2 < lim) {
                                    stage[p2 - base] = bstart + 32u + (uint32_t)__builtin_ctz(hi);
                                    hi &= hi - 1;
                                    ++p2;
                                }
                                mk[e] = (sj_u64)lo | ((sj_u64)hi << 32);
                                pos[e] = p2;
                            }
                            wave_lds_fence();
                            const uint32_t nn = (WT - base) < (uint32_t)CAP ? (WT - base) : (uint32_t)CAP;
                            uint32_t* dst = dst0 + base;
                            for (uint32_t i = lane; i < nn; i += 64) dst[i] = stage[i];
                            wave_lds_fence();
                        }
                    }
                }
            }
            SJMI_TSTAMP(prev, 4);
#ifdef SJMI_TRACE
            if (lane == 0)
                (gstate + 2 * (sj_u64)ngran + (sj_u64)prev * SJMI_TRACE_SLOTS)[5] =
                    (sj_u64)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((sj_u64)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
        }
        if (!have) break;
        // park granule `cur` until its prefix is known (next iteration)
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            ws.park.pot[s][lane] = pot[s];
            ws.park.m0[s][lane] = m[s];
            ws.park.meta[s][lane] = meta[s];
        }
        wave_lds_fence();
        prev = cur;
        prev_par = wpar;
        prev_c0 = W0;
        prev_c1 = WP - W0;
        cur = nxt;
    }
    // one status update per wave
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) err |= __shfl_xor(err, dd);
    if (lane == 0 && err && !(dbg & DBG_NO_LOOKBACK))  // (with fake prefixes every granule would report errors)
        __hip_atomic_fetch_or(&res->status, err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    zero_next_workspace(zero_ptr, zero_chunks, worker, nworkers, lane);
    zero_next_workspace(zero2_ptr, zero2_chunks, worker, nworkers, lane);
}

// The kernels proper.  The BATCH flavour is pinned to the occupancy of the plain one (four waves per SIMD = 128 VGPRs): its two
// side outputs cost the register allocator 4 VGPRs at S = 4, and 132 would mean three waves (the plain pass of a 1M-document
// batch: 0.51 ms instead of 0.43); spilling them is the cheaper way out.
template <int S, int LDSW, bool SAFE, bool BATCH>
struct Stage1Kernel;
// (round 6: the sorted expansion's entries -- 2 registers per half mask -- took k_stage1<4, ...> from 125 to 135 VGPRs = three waves
//  per SIMD; pinned to four like the batch flavour it allocates 128 without a spilled VGPR)
#ifndef SJMI_S1_PLAIN_WAVES
#define SJMI_S1_PLAIN_WAVES 4
#endif
template <int S, int LDSW, bool SAFE>
__global__ void __launch_bounds__(256)
#if SJMI_S1_PLAIN_WAVES
__attribute__((amdgpu_waves_per_eu(SJMI_S1_PLAIN_WAVES, SJMI_S1_PLAIN_WAVES)))
#endif
k_stage1(const uint8_t* __restrict__ buf, sj_u64 len, uint32_t* __restrict__ out, sj_u64 out_cap,
         sj_u64* gstate, uint32_t* ticket, Stage1Result* res, uint32_t ngran, uint32_t dbg, uint4* zero_ptr,
         uint32_t zero_chunks, Stage1Result* result_out, sj_u64* __restrict__ blkpar, const uint32_t* __restrict__ skip,
         uint32_t* __restrict__ blkidx, uint16_t* __restrict__ blkw, uint4* zero2_ptr, uint32_t zero2_chunks, Stage1Single ss) {
    stage1_body<S, LDSW, SAFE, false>(buf, len, out, out_cap, gstate, ticket, res, ngran, dbg, zero_ptr, zero_chunks, result_out, blkpar,
                                      skip, blkidx, blkw, zero2_ptr, zero2_chunks, ss);
}
#ifndef SJMI_S1_BATCH_WAVES
#define SJMI_S1_BATCH_WAVES 4
#endif
template <int S, int LDSW, bool SAFE>
__global__ void __launch_bounds__(256)
#if SJMI_S1_BATCH_WAVES
__attribute__((amdgpu_waves_per_eu(SJMI_S1_BATCH_WAVES, SJMI_S1_BATCH_WAVES)))
#endif
k_stage1_batch(const uint8_t* __restrict__ buf, sj_u64 len, uint32_t* __restrict__ out, sj_u64 out_cap,
               sj_u64* gstate, uint32_t* ticket, Stage1Result* res, uint32_t ngran, uint32_t dbg, uint4* zero_ptr,
               uint32_t zero_chunks, Stage1Result* result_out, sj_u64* __restrict__ blkpar, const uint32_t* __restrict__ skip,
               uint32_t* __restrict__ blkidx, uint16_t* __restrict__ blkw, uint4* zero2_ptr, uint32_t zero2_chunks, Stage1Single ss) {
    stage1_body<S, LDSW, SAFE, true>(buf, len, out, out_cap, gstate, ticket, res, ngran, dbg, zero_ptr, zero_chunks, result_out, blkpar,
                                     skip, blkidx, blkw, zero2_ptr, zero2_chunks, ss);
}
template <int S, int LDSW, bool SAFE>
struct Stage1Kernel<S, LDSW, SAFE, false> {
    static constexpr auto fn = k_stage1<S, LDSW, SAFE>;
};
template <int S, int LDSW, bool SAFE>
struct Stage1Kernel<S, LDSW, SAFE, true> {
    static constexpr auto fn = k_stage1_batch<S, LDSW, SAFE>;
};

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
// granules of steps x 4 KiB (one per worker wave and iteration)
static uint64_t granules_for(uint64_t len, int steps, bool no_tail = false) {
    const uint64_t nblocks = len / 64 + (no_tail ? 0 : 1);
    return (nblocks + 64ull * steps - 1) / (64ull * steps);
}

size_t stage1_workspace_bytes(uint64_t len, int steps) {
    return WS_TILE_STATE_OFFSET + (2 + SJMI_TRACE_SLOTS) * (size_t)granules_for(len, steps) * sizeof(sj_u64);
}
// The scanner's inclusive prefixes of a finished FAST launch over (len, steps), still in its workspace until the launch after
// next: entry g = [63:62] == 2, [39:0] structurals in granules 0..g (a granule = steps * 4 KiB of input).  For consumers that
// look structurals up by byte position (batch.hip: the binary search of a document's first index starts inside its granule).
Stage1Prefixes stage1_prefixes(const void* d_ws, uint64_t len, int steps) {
    Stage1Prefixes v;
    v.ngran = granules_for(len, steps);
    v.granule_bytes = (uint32_t)steps * 4096u;
    v.pfx = reinterpret_cast<const unsigned long long*>(static_cast<const uint8_t*>(d_ws) + WS_TILE_STATE_OFFSET) + v.ngran;
    return v;
}


int stage1_pick_steps(uint64_t len) {
    // small documents: small granules so that more waves get work (tools/size_sweep.py: 0.6 MB 8.9 us with 4 KiB
    // granules against 13.9 us with 16 KiB; 10 MB 14.5 us with 8 KiB; from ~20 MB on 16 KiB granules win)
    return len <= (4u << 20) ? 1 : (len <= (16u << 20) ? 2 : 4);
}

// workgroups of k_stage1<...> that are resident at the same time on the current device (fast mode's grid)
template <int S, int LDSW, bool SAFE, bool BATCH>
static hipError_t resident_workgroups(unsigned* out) {
    static std::atomic<unsigned> cached[16];  // (contexts on several host threads may get here together)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16 || !cached[dev]) {
        int per_cu = 0, cus = 0;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, Stage1Kernel<S, LDSW, SAFE, BATCH>::fn, 256, 0)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        const unsigned n = (unsigned)(per_cu > 0 ? per_cu : 1) * (unsigned)(cus > 0 ? cus : 1);
        if (dev < 0 || dev >= 16) {
            *out = n;
            return hipSuccess;
        }
        cached[dev].store(n, std::memory_order_relaxed);
    }
    *out = cached[dev];
    return hipSuccess;
}

template <int S, int LDSW, bool SAFE, bool BATCH>
static hipError_t launch_mode(const uint8_t* d_buf, uint64_t len, uint32_t* d_out, uint64_t out_cap, sj_u64* gs,
                                 uint32_t* ticket, Stage1Result* res, uint64_t ngran, hipStream_t stream,
                                 hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t dbg, const Stage1Extras& ex) {
    unsigned resident = 0;
    hipError_t e = resident_workgroups<S, LDSW, SAFE, BATCH>(&resident);
    if (e != hipSuccess) return e;
    const uint64_t want = (ngran + 3) / 4 + (SAFE ? 0 : 1);  // 4 worker waves each + the scanner workgroup
    if (dbg & DBG_SMALL_GRID) resident = SAFE ? 8 : 9;  // test hook: far fewer granules in flight than a scanner window
    const dim3 grid((unsigned)(want < resident ? want : resident)), block(256);
    uint4* zp = static_cast<uint4*>(ex.zero_next);
    const uint32_t zc = (uint32_t)(ex.zero_bytes / 16);
    Stage1Result* ro = static_cast<Stage1Result*>(ex.result_out);
    sj_u64* bp = static_cast<sj_u64*>(ex.blkpar);
    uint32_t* bi = static_cast<uint32_t*>(ex.blkidx);
    uint16_t* bw = static_cast<uint16_t*>(ex.blkw);
    uint4* z2 = static_cast<uint4*>(ex.zero2);
    const uint32_t z2c = (uint32_t)(ex.zero2_bytes / 16);
    if (ev_start && ev_stop) {
        // the events are attached to the dispatch itself (its start / end timestamps), not recorded around it:
        // hipEventRecord pairs added 10-25 us of queue latency to a 200 us kernel
        hipExtLaunchKernelGGL((Stage1Kernel<S, LDSW, SAFE, BATCH>::fn), grid, block, 0, stream, ev_start, ev_stop, 0, d_buf, (sj_u64)len,
                              d_out, (sj_u64)out_cap, gs, ticket, res, (uint32_t)ngran, dbg, zp, zc, ro, bp, ex.skip, bi, bw, z2, z2c, ex.single);
    } else {
        hipLaunchKernelGGL((Stage1Kernel<S, LDSW, SAFE, BATCH>::fn), grid, block, 0, stream, d_buf, (sj_u64)len, d_out, (sj_u64)out_cap,
                           gs, ticket, res, (uint32_t)ngran, dbg, zp, zc, ro, bp, ex.skip, bi, bw, z2, z2c, ex.single);
    }
    return hipGetLastError();
}

template <int S, int LDSW>
static hipError_t launch_variant(const uint8_t* d_buf, uint64_t len, uint32_t* d_out, uint64_t out_cap, sj_u64* gs,
                                 uint32_t* ticket, Stage1Result* res, uint64_t ngran, hipStream_t stream,
                                 hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t dbg, const Stage1Extras& ex) {
    const bool batch = ex.blkidx && ex.blkw;  // (both or neither)
    if (dbg & FLAG_SAFE)
        return batch ? launch_mode<S, LDSW, true, true>(d_buf, len, d_out, out_cap, gs, ticket, res, ngran, stream, ev_start, ev_stop, dbg, ex)
                     : launch_mode<S, LDSW, true, false>(d_buf, len, d_out, out_cap, gs, ticket, res, ngran, stream, ev_start, ev_stop, dbg, ex);
    return batch ? launch_mode<S, LDSW, false, true>(d_buf, len, d_out, out_cap, gs, ticket, res, ngran, stream, ev_start, ev_stop, dbg, ex)
                 : launch_mode<S, LDSW, false, false>(d_buf, len, d_out, out_cap, gs, ticket, res, ngran, stream, ev_start, ev_stop, dbg, ex);
}

// entries of the per-block side outputs of a BATCH launch over len bytes (whole granules: the kernel stores unguarded)
size_t stage1_block_entries(uint64_t len) { return (size_t)((len / 64 + 1 + 255) / 256 + 1) * 256; }

hipError_t stage1_launch(const uint8_t* d_buf, uint64_t len, uint32_t* d_out, uint64_t out_cap, void* d_ws,
                         int steps, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t dbg, const Stage1Extras& ex) {
    const uint64_t ngran = granules_for(len, steps, (dbg & FLAG_NO_TAIL) != 0);
    if (ngran == 0) return hipErrorInvalidValue;  // (an empty shard without a tail block: nothing to launch)
    const size_t ws_bytes = WS_TILE_STATE_OFFSET + (2 + SJMI_TRACE_SLOTS) * (size_t)ngran * sizeof(sj_u64);
    hipError_t e = hipSuccess;
    if (!ex.workspace_is_zero && (e = hipMemsetAsync(d_ws, 0, ws_bytes, stream)) != hipSuccess) return e;
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    uint32_t* ticket = reinterpret_cast<uint32_t*>(ws + WS_TICKET_OFFSET);
    Stage1Result* res = reinterpret_cast<Stage1Result*>(ws + WS_RESULT_OFFSET);
    sj_u64* gs = reinterpret_cast<sj_u64*>(ws + WS_TILE_STATE_OFFSET);
    switch (steps) {  // granule = steps x 4 KiB
    case 1: e = launch_variant<1, 6144>(d_buf, len, d_out, out_cap, gs, ticket, res, ngran, stream, ev_start, ev_stop, dbg, ex); break;
    case 2: e = launch_variant<2, 9216>(d_buf, len, d_out, out_cap, gs, ticket, res, ngran, stream, ev_start, ev_stop, dbg, ex); break;
    case 4: e = launch_variant<4, 9216>(d_buf, len, d_out, out_cap, gs, ticket, res, ngran, stream, ev_start, ev_stop, dbg, ex); break;
    default: return hipErrorInvalidValue;
    }
    return e;
}

hipError_t transpose_selftest_launch(const uint32_t* d_words, uint32_t nblocks, uint32_t* d_mismatches,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(k_transpose_selftest, dim3((nblocks + 255) / 256), dim3(256), 0, stream, d_words, nblocks,
                       d_mismatches);
    return hipGetLastError();
}

}  // namespace sjmi
