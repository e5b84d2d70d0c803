// strings.hip -- the string buffer of a document in ONE streaming pass (MI355X / gfx950).
//
// Replaces every StringParser.parseString(buf, idx, stringBuffer, stringBufferIdx) call of the reference's stage 2
// (/root/reference/src/main/java/org/simdjson/StringParser.java:18-68, TapeBuilder.visitString TapeBuilder.java:174-177):
// the string buffer -- records [be32 length][unescaped UTF-8] in document order -- is a byte compaction of the document
// (keep the bytes inside strings, drop escape backslashes, translate escapes) with a 4-byte header at every opening
// quote.  A record's offset is therefore a prefix sum over the document's bytes and its length the difference of two
// such sums: no per-string gather, no index array, no size array -- the document is read once, coalesced, and the
// string buffer written once, coalesced (round 1-2 had three index-driven kernels with two dependent gathers per
// structural: 11 % of the HBM roofline, 2-3 x the algorithmic traffic).
//
// Mapping to the hardware (the per-block algebra is in sj_strings.h, shared with the host simulation):
//   * one lane = one 64-byte block, loaded as 4 x dwordx4 (+ the 16 bytes in front of it), transposed to bit planes
//     like stage 1 (sj_block.h); which bytes are kept / dropped / patched is 64-bit boolean algebra in VGPRs.  The one
//     global input it cannot derive locally -- is the block entered inside a string -- comes from stage 1, which
//     records that bit for every block while it resolves its own parity chain (k_stage1's `blkpar` output);
//   * one wave = one GRANULE of 64 blocks (4 KiB) at a time, taken by atomic ticket (16 counters); the wave scans the lanes' byte
//     and string counts (DPP), publishes the granule's AGGREGATE {bytes, strings} and goes on; a scanner workgroup
//     turns aggregates into PREFIXes (the chain of stage 1, but a plain sum); the prefix of granule k is picked up
//     while granule k+1 is being classified, so nobody waits for the chain;
//   * the kept bytes of a block are packed dword by dword (v_perm_b32 with a selector from a 16-entry LDS table),
//     shifted to their byte position and OR-ed into a wave-private LDS tile (ds_or_b32: neighbouring lanes share
//     dwords); headers are OR-ed in by the lane that holds the CLOSING quote (length = offset at the closing quote -
//     offset at the opening quote - 4; the opening quote's offset travels to it through a DPP max-scan); escapes
//     that change a byte's value are XOR-ed in afterwards; the tile leaves one iteration later as 16-byte stores to
//     byte-granular addresses (gfx950 global memory runs in unaligned-access mode), its tail one byte per lane;
//   * a string that is still open at the end of a granule gets its header from the granule that closes it: every
//     granule publishes where its pending header sits (OPEN RECORD), the closing granule walks back to it and stores
//     the four bytes itself; the opening granule leaves those four bytes out of its stores, so nothing is written twice.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>

#include "sj_strings.h"
#include "sj_block32.h"
#include "stage1.h"

#ifndef SJMI_STR_ABL
#define SJMI_STR_ABL 0  // ablation experiments only (results invalid): 1 no headers, 2 no copy, 4 no stores, 8 no \\u patches
#endif

namespace sjmi {

constexpr uint32_t STR_NONE = 0xFFFFFFFFu;
// bytes a granule can produce: the kept bytes and slots are distinct source bytes of [start - 3, start + 4096), every
// header costs at least one quote byte of the granule and a second one for all but one of them: <= 4100 + 2 * 2048
constexpr int STR_TILE_BYTES = 8192 + 64;
constexpr int STR_TILE_DW = STR_TILE_BYTES / 4;
constexpr uint32_t STR_SPIN_LIMIT = 1u << 19;

// granule states (u64, relaxed agent-scope, the data is the flag: cdna_hip_programming.md Guideline 16 form R2)
//   AGGREGATE  [63:62] = 1, [36] a string error in the granule, [35:20] strings opened, [19:0] bytes produced
//   PREFIX     [63:62] = 2, [61:32] strings opened in granules 0..t, [31:0] bytes (saturating: the buffer is < 4 GiB)
//   OPEN REC   [63] valid, [62] the granule holds the opening quote of the string that is open at its end,
//              [23:16] first error of that string inside this granule, [15:0] granule-relative offset of its header
constexpr sj_u64 SG_AGG = 1ull << 62, SG_PFX = 2ull << 62;
constexpr sj_u64 OR_VALID = 1ull << 63, OR_HAS_OPEN = 1ull << 62;

__device__ __forceinline__ void sg_store(sj_u64* p, sj_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ sj_u64 sg_load(const sj_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t sdpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t str_incl_scan(uint32_t v) {
    v = sdpp_add<0x111, 0xF>(v);  // row_shr:1
    v = sdpp_add<0x112, 0xF>(v);
    v = sdpp_add<0x114, 0xF>(v);
    v = sdpp_add<0x118, 0xF>(v);
    v = sdpp_add<0x142, 0xA>(v);  // row_bcast:15
    v = sdpp_add<0x143, 0xC>(v);  // row_bcast:31
    return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t sdpp_max(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t str_incl_max(uint32_t v) {
    v = sdpp_max<0x111, 0xF>(v);
    v = sdpp_max<0x112, 0xF>(v);
    v = sdpp_max<0x114, 0xF>(v);
    v = sdpp_max<0x118, 0xF>(v);
    v = sdpp_max<0x142, 0xA>(v);
    v = sdpp_max<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ void str_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct StrArgs {
    const uint8_t* buf;
    sj_u64 len;
    const sj_u64* blkpar;  // bit l of word g: block 64 g + l is entered inside a string (stage 1's side output)
    uint8_t* sb;
    sj_u64 sb_cap;
    uint32_t* soff;        // optional: offset of every record, by string ordinal
    sj_u64 soff_cap;
    uint32_t* blk_ord;     // optional: ordinal of the first string opened at or behind every block's first byte
    sj_u64* gstate;        // aggregates[ngran] | prefixes[ngran] | open records[ngran]
    uint32_t* ticket;      // STR_TICKET_CLASSES counters, 64 bytes apart
    uint32_t* wsflags;     // [0] status bits (SJMI_ST_INTERNAL), [1] the scanner's CU, [2] the role ticket
    UnescapeResult* res;
    uint32_t ngran;
    uint32_t flags;
    // fused batch pipeline: *sel != 0 = the optimistic plain stage-1 pass was accepted (the batch itself + its parities);
    // *sel == 0 = the sanitized copy and the parities of the pass over it (see strings_sanitize_launch)
    const uint32_t* sel;
    const uint8_t* buf_alt;
    const sj_u64* blkpar_alt;
    const uint32_t* skip;  // != nullptr and *skip != 0: the launch does nothing (the batch pipeline's per-document string pass behind an accepted plain pass)
};

// ---------------------------------------------------------------------------------------------------------------------
// the scanner: workgroup 0 turns the workers' aggregates, in order, into inclusive prefixes (see stage1.hip's scanner for
// why it is built like this: four waves take windows of 64 * SSCAN_K granules round-robin, everything that does not depend on
// the running state is done before it arrives through LDS, and at the workers' frontier a wave publishes whatever
// is ready lane by lane so that a launch with few resident workgroups cannot deadlock on a half-handed-out window)
// ---------------------------------------------------------------------------------------------------------------------
// (windows of 128 granules: twitter x1024 0.400 ms against 0.406 with 256, 0.75 with 512 -- the frontier window is never
//  complete and goes lane by lane; 64: 0.60)
#ifndef SJMI_SSCAN_K
#define SJMI_SSCAN_K 2
#endif
constexpr int SSCAN_K = SJMI_SSCAN_K;
struct StrHand {
    uint32_t seq;  // window whose entry state is in ord / out; 0xFFFFFFFF = a scanner wave gave up
    uint32_t ord;
    sj_u64 out;
};
__device__ __forceinline__ void sscan_load(sj_u64 v[SSCAN_K], const sj_u64* agg, sj_u64 first, uint32_t n) {
#pragma unroll
    for (int j = 0; j < SSCAN_K; ++j) v[j] = first + j < n ? sg_load(&agg[first + j]) : SG_AGG;
}
__device__ __forceinline__ bool sscan_ready(const sj_u64 v[SSCAN_K]) {
    bool r = true;
#pragma unroll
    for (int j = 0; j < SSCAN_K; ++j) r &= v[j] != 0;
    return r;
}
__device__ __forceinline__ void sscan_fold(const sj_u64 v[SSCAN_K], uint32_t* s0, uint32_t* s1) {
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int j = 0; j < SSCAN_K; ++j) {
        a += (uint32_t)v[j] & 0xFFFFFu;
        b += (uint32_t)(v[j] >> 20) & 0xFFFFu;
    }
    *s0 = a;
    *s1 = b;
}
__device__ __forceinline__ void sscan_publish(const sj_u64 v[SSCAN_K], sj_u64* pfx, sj_u64 first, uint32_t n, sj_u64 out, uint32_t ord) {
#pragma unroll
    for (int j = 0; j < SSCAN_K; ++j) {
        out += (uint32_t)v[j] & 0xFFFFFu;
        ord += (uint32_t)(v[j] >> 20) & 0xFFFFu;
        if (first + j < n) {
            const sj_u64 o32 = out > 0xFFFFFFFFull ? 0xFFFFFFFFull : out;
            const sj_u64 r30 = ord > 0x3FFFFFFFu ? 0x3FFFFFFFu : ord;
            sg_store(&pfx[first + j], SG_PFX | (r30 << 32) | o32);
        }
    }
}

__device__ __forceinline__ void str_scanner_wave(StrHand* hand, int wave, const StrArgs& a, int lane) {
    __builtin_amdgcn_s_setprio(3);
    constexpr uint32_t WIN = 64 * SSCAN_K;
    const sj_u64* agg = a.gstate;
    sj_u64* pfx = a.gstate + a.ngran;
    const uint32_t n = a.ngran;
    for (sj_u64 win = (sj_u64)wave; win * WIN < n; win += 4) {
        const sj_u64 first = win * WIN + (sj_u64)lane * SSCAN_K;
        sj_u64 v[SSCAN_K];
        sj_u64 out2;
        uint32_t ord2;
        bool full;
        for (;;) {
            sscan_load(v, agg, first, n);
            full = __ballot(sscan_ready(v)) == ~0ull;
            if (full) break;
            const uint32_t seq = __hip_atomic_load(&hand->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (seq == (uint32_t)win || seq == 0xFFFFFFFFu) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (full) {
            uint32_t s0, s1;
            sscan_fold(v, &s0, &s1);
            const uint32_t i0 = str_incl_scan(s0), i1 = str_incl_scan(s1);
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
            uint32_t seq;
            do {
                seq = __hip_atomic_load(&hand->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } while (seq != (uint32_t)win && seq != 0xFFFFFFFFu);
            if (seq == 0xFFFFFFFFu) return;
            const sj_u64 out = hand->out;
            const uint32_t ord = hand->ord;
            out2 = out + t0;
            ord2 = ord + t1;
            if (lane == 0) {
                hand->out = out2;
                hand->ord = ord2;
                __hip_atomic_store(&hand->seq, (uint32_t)win + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            sscan_publish(v, pfx, first, n, out + (i0 - s0), ord + (i1 - s1));
        } else {
            uint32_t seq;
            do {
                seq = __hip_atomic_load(&hand->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } while (seq != (uint32_t)win && seq != 0xFFFFFFFFu);
            if (seq == 0xFFFFFFFFu) return;
            sj_u64 out = hand->out;
            uint32_t ord = hand->ord;
            int done = 0;
            for (uint32_t spins = 0;; ++spins) {
                const sj_u64 rb = __ballot(sscan_ready(v));
                const int nr = ~rb ? __builtin_ctzll(~rb) : 64;
                if (nr > done) {
                    const bool act = lane >= done && lane < nr;
                    uint32_t s0, s1;
                    sscan_fold(v, &s0, &s1);
                    if (!act) s0 = s1 = 0;
                    const uint32_t i0 = str_incl_scan(s0), i1 = str_incl_scan(s1);
                    if (act) sscan_publish(v, pfx, first, n, out + (i0 - s0), ord + (i1 - s1));
                    out += (uint32_t)__builtin_amdgcn_readlane((int)i0, 63);
                    ord += (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
                    done = nr;
                    spins = 0;
                }
                if (done == 64) break;
                if (spins > STR_SPIN_LIMIT) {  // never expected: a worker did not publish
                    if (lane == 0) {
                        __hip_atomic_fetch_or(&a.wsflags[0], SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&hand->seq, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_or(&a.res->flags, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
                sj_u64 nv[SSCAN_K];
                sscan_load(nv, agg, first, n);
#pragma unroll
                for (int j = 0; j < SSCAN_K; ++j)
                    if (lane >= done) v[j] = nv[j];
            }
            out2 = out;
            ord2 = ord;
            if (lane == 0) {
                hand->out = out2;
                hand->ord = ord2;
                __hip_atomic_store(&hand->seq, (uint32_t)win + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if ((win + 1) * WIN >= n && lane == 0) {  // the last window: the launch's totals
            a.res->total_bytes = out2;
            a.res->reserved = ord2;
            uint32_t f = 0;
            if (out2 > a.sb_cap || out2 > 0xFFFFFFFFull) f |= 1u;   // string buffer too small (nothing was written past its end)
            if (a.soff && (sj_u64)ord2 > a.soff_cap) f |= 2u;       // record table too small
            if (f) __hip_atomic_fetch_or(&a.res->flags, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// workers
// ---------------------------------------------------------------------------------------------------------------------
struct StrStep {
    uint4 q0, q1, q2, q3;  // the lane's 64 bytes
    uint4 h;               // the 16 bytes in front of them
};
__device__ __forceinline__ void str_load_step(StrStep& d, const uint8_t* __restrict__ buf, sj_u64 blk, sj_u64 nblocks) {
    const sj_u64 b = blk < nblocks ? blk : nblocks - 1;  // (branch-free: a block past the end is clamped and ignored)
    const uint4* src = reinterpret_cast<const uint4*>(buf + b * 64);
    d.q0 = src[0];
    d.q1 = src[1];
    d.q2 = src[2];
    d.q3 = src[3];
    d.h = *reinterpret_cast<const uint4*>(buf + (b > 0 ? b * 64 - 16 : 0));  // (unused for block 0)
}

__device__ __forceinline__ void tile_or(uint32_t* tile, uint32_t off, uint32_t v) {
    const sj_u64 x = (sj_u64)v << ((off & 3u) * 8u);  // (one v_lshlrev_b64: measured 1.6 % faster than two 32-bit shifts)
    atomicOr(&tile[off >> 2], (uint32_t)x);
    atomicOr(&tile[(off >> 2) + 1], (uint32_t)(x >> 32));
}
__device__ __forceinline__ void tile_xor(uint32_t* tile, uint32_t off, uint32_t v) {
    const sj_u64 x = (sj_u64)v << ((off & 3u) * 8u);
    atomicXor(&tile[off >> 2], (uint32_t)x);
    atomicXor(&tile[(off >> 2) + 1], (uint32_t)(x >> 32));
}
__device__ __forceinline__ bool swar_has_backslash(uint32_t w) {
    const uint32_t z = w ^ 0x5C5C5C5Cu;
    return (~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u) != 0;
}
struct __attribute__((packed, aligned(1))) StrU4B { uint32_t a; };

// ticket counters: class c = worker mod N owns the granules = c mod N.  A 4 KiB granule needs a ticket four times as often as
// stage 1's 16 KiB granule, and one counter saturates: twitter x1024 0.596 ms with 4 counters, 0.400 with 8, 0.385 with 16,
// 0.387 with 32
#ifndef SJMI_STR_CLASSES
#define SJMI_STR_CLASSES 16
#endif
constexpr uint32_t STR_TICKET_CLASSES = SJMI_STR_CLASSES;

constexpr uint32_t STR_UQ_ITEMS = 128;  // sequences of one granule that are decoded densely (more: the lane-local loop)

template <bool SOFF>
struct StrWaveLds {
    alignas(16) uint32_t tile[STR_TILE_DW];
    // Record offsets by ordinal (round 6).  A string's record begins at its header, and the header loop below knows where that is
    // when it writes the length -- so the offset of every string of the granule, relative to the granule's first output byte, is
    // parked HERE by ordinal inside the granule (one ds_write_b16 per header trip; at most 2048 strings in 4 KiB, offsets below
    // 8256), and the flush one iteration later, when the granule's prefix is known, stores them coalesced.  (Round 5 parked three
    // masks per lane and the flush re-derived every offset with two 64-bit popcounts and stored it by itself: a 4-byte store per
    // string to 64 different lines per trip -- 1.20 GB written for 0.86 GB of records and offsets on the configs[3] batch.)
    uint16_t so[SOFF ? 2048 : 2];
};

template <bool SOFF>
__global__ void __launch_bounds__(256)
k_strings(const StrArgs a0) {
    StrArgs a = a0;
    if (a.skip && *a.skip != 0) return;
    if (a.sel && *a.sel == 0) {  // (uniform for the whole launch)
        a.buf = a.buf_alt;
        a.blkpar = a.blkpar_alt;
    }
    __shared__ StrWaveLds<SOFF> sh[4];
    __shared__ uint32_t s_uq[4][STR_UQ_ITEMS];  // the \uXXXX sequences of a wave's granule: tile offset | position << 14 | pair << 26
    __shared__ uint32_t s_lut[16];
    __shared__ StrHand hand;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ngran = a.ngran;
    sj_u64* const agg = a.gstate;
    sj_u64* const pfx = a.gstate + ngran;
    sj_u64* const orec = a.gstate + 2 * (sj_u64)ngran;
    const uint32_t my_cu = (((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xFFu) |
                           (((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 8) | 0x80000000u;
    // Roles by ARRIVAL, not by workgroup number: the first workgroup that gets a CU is the scanner, so the scanner is resident
    // whenever a worker is.  (Workgroups are not placed in order when another process's persistent kernel holds part of the GPU:
    // with the scanner = workgroup 0, two processes alternating stage 1 and this pass left workers polling for a scanner that was
    // still waiting for a slot -- one tripped spin bound, flags 4, in about one of ten runs of tests/test_gpu_two_process.py.)
    __shared__ uint32_t s_role;
    if (threadIdx.x == 0) s_role = __hip_atomic_fetch_add(&a.wsflags[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t role = s_role;
    if (role == 0) {
        if (threadIdx.x == 0) {
            __hip_atomic_store(&a.wsflags[1], my_cu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hand.seq = 0;
            hand.ord = 0;
            hand.out = 0;
        }
        __syncthreads();
        str_scanner_wave(&hand, wave, a, lane);
        return;
    }
    if (threadIdx.x < 16) s_lut[threadIdx.x] = sj_str_pack_selector(threadIdx.x);
    __syncthreads();
    const uint32_t nworkers = (gridDim.x - 1u) * 4u;
    const uint32_t worker = (role - 1u) * 4u + (uint32_t)wave;
    uint32_t* const tile = sh[wave].tile;
    const sj_u64 nblocks = a.len / 64 + 1;
    const sj_u64 lt_lane = (1ull << lane) - 1ull;
    (void)lt_lane;
    const uint32_t NC = nworkers < STR_TICKET_CLASSES ? nworkers : STR_TICKET_CLASSES;
    const uint32_t cls = worker % NC;
    uint32_t* const my_ticket = a.ticket + cls * 16u;
    // workers that share the scanner's CU run at ~2/3 speed and would pace the chain: they retire after one granule
    uint32_t retire = 1;
    if (gridDim.x >= 64) retire = __hip_atomic_load(&a.wsflags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ my_cu;
    uint32_t cur;
    {
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(my_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) * NC + cls;
    }
    StrStep d;
    str_load_step(d, a.buf, (sj_u64)cur * 64 + lane, nblocks);

    // the granule whose tile awaits its prefix (flushed one iteration later)
    uint32_t prev = STR_NONE, prev_n = 0, prev_pend = STR_NONE, prev_fclose = STR_NONE, prev_fclose_err = 0, prev_nopen = 0;
    uint32_t prev_oexcl = 0;  // per lane
    sj_u64 pf = 0, pf_or = 0, pf_pp = 0;  // requested a classification ahead: pfx[prev-1], orec[prev-1], pfx[prev-2]
    uint32_t err_wave = 0;
#ifdef SJMI_STR_SPINSTAT
    unsigned long long stat_waits = 0, stat_spins = 0, stat_flushes = 0;
#endif

    for (;;) {
        const bool have = cur < ngran;
        uint32_t tk = 0;
        uint32_t w[16];
        SjStrBlock m;
        uint32_t base = 0, oexcl = 0, tot_out = 0, tot_open = 0, prevD = 0, xerr = 0;
        uint32_t entered_in = 0;  // the lane's block begins inside a string
        uint32_t pend_rel = STR_NONE, fclose_rel = STR_NONE, fclose_err = 0;
        bool any_err = false;
        sj_u64 blk = 0;
        if (have) {
            // =================== classify granule `cur`, publish its aggregate ===================
            __builtin_amdgcn_s_setprio(2);
            blk = (sj_u64)cur * 64 + lane;
            w[0] = d.q0.x; w[1] = d.q0.y; w[2] = d.q0.z; w[3] = d.q0.w;
            w[4] = d.q1.x; w[5] = d.q1.y; w[6] = d.q1.z; w[7] = d.q1.w;
            w[8] = d.q2.x; w[9] = d.q2.y; w[10] = d.q2.z; w[11] = d.q2.w;
            w[12] = d.q3.x; w[13] = d.q3.y; w[14] = d.q3.z; w[15] = d.q3.w;
            const uint4 hq = d.h;
            const sj_u64 parword = a.blkpar[cur];
            const bool active = blk < nblocks;
            sj_u64 p[8];
            {   // (sj_block32.h: 128 instead of 152 instructions per block)
                uint32_t plo[8], phi[8];
                sj_transpose32(w, plo, phi);
                for (int k = 0; k < 8; ++k) p[k] = ((sj_u64)phi[k] << 32) | plo[k];
            }
            uint32_t pin = 0, e_in = 0;
            bool unresolved = false;
            if (cur == ngran - 1) {  // (wave-uniform) the document's end: bytes behind it are spaces, blocks behind it empty
                if (active) {
                    const sj_u64 rem = a.len - blk * 64;
                    sj_mask_tail(p, rem < 64 ? (uint32_t)rem : 64u);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) p[k] = 0;
                    p[5] = ~0ull;
                }
            }
            if (active) {
                pin = (uint32_t)(parword >> lane) & 1u;
                if (blk > 0) {
                    uint32_t p_in;
                    unresolved = !sj_carry_from_halo((sj_u64)hq.z | ((sj_u64)hq.w << 32), &e_in, &p_in);
                }
            }
            if (__ballot(unresolved)) {  // rare: a backslash run longer than the 8 bytes in front of the block
                if (unresolved) {
                    uint32_t p_in;
                    sj_carry_slow(a.buf, 0, blk * 64, &e_in, &p_in);
                }
            }
            const SjStrBase s = sj_str_base(p, e_in, pin);
            // shortcuts: no escaped character inside a string anywhere in the wave's 4 KiB; no \u sequence that ends in or
            // reaches into one of its blocks (the previous lane's last 10 positions; lane 0: any backslash in its halo)
            const bool any_esc = __ballot(s.ED != 0) != 0;
            sj_u64 euc = 0;
            if (any_esc) euc = s.ED & sj_str_classes<sj_u64>(p, false).isu;  // (the same expressions as in sj_str_block: shared)
            const uint32_t euc_hi_prev = (uint32_t)__shfl_up((int)(uint32_t)(euc >> 32), 1);
            const bool halo_bs = blk > 0 && active && (swar_has_backslash(hq.y) | swar_has_backslash(hq.z) | swar_has_backslash(hq.w));
            const bool trig = euc != 0 || (lane > 0 ? (euc_hi_prev >> 22) != 0 : halo_bs);
#ifdef SJMI_STR_NO_U  /* experiment: how fast is the pass without the \u look-back in its register budget */
            const bool do_u = false;
            (void)trig;
#else
            const bool do_u = __ballot(trig) != 0;
#endif
            SjStrHalo halo;
            halo.e_in = 0;
            if (do_u) {
                uint32_t w8[8] = {hq.x, hq.y, hq.z, hq.w, 0u, 0u, 0u, 0u};
                if (blk == 0 || !active) w8[0] = w8[1] = w8[2] = w8[3] = 0x20202020u;
                sj_transpose_half(w8, halo.hp);
                const bool hun = blk > 0 && active && sj_str_halo_unresolved(halo.hp);
                if (__ballot(hun)) {
                    if (hun) halo.e_in = sj_backslash_run_parity(a.buf, 0, blk * 64 - 16);
                }
            }
            m = sj_str_block(p, s, pin, any_esc || do_u, do_u, &halo);
            entered_in = pin;
            const uint32_t nout = sj_str_out_bytes(m), nopen = (uint32_t)__popcll(m.O);
            const uint32_t packed = str_incl_scan(nout | (nopen << 16));
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)packed, 63);
            tot_out = tot & 0xFFFFu;
            tot_open = tot >> 16;
            base = (packed & 0xFFFFu) - nout;
            oexcl = (packed >> 16) - nopen;
            // the header of the string that is open where a lane's block begins: the last opening quote in front of it
            const uint32_t lastD = m.O ? base + sj_str_offset(m, 63u - (uint32_t)__builtin_clzll(m.O)) + 1u : 0u;
            const uint32_t lm = str_incl_max(lastD ? ((lastD << 6) | (uint32_t)lane) : 0u);
            uint32_t prevv = (uint32_t)__shfl_up((int)lm, 1);
            if (lane == 0) prevv = 0;
            prevD = prevv >> 6;
            const uint32_t prevL = prevv & 63u;
            const sj_u64 open_lanes = __ballot(m.O != 0), quote_lanes = __ballot((m.O | m.CL) != 0);
            const uint32_t entry_in = (uint32_t)__builtin_amdgcn_readlane((int)pin, 0);
            const uint32_t exit_in = (uint32_t)__builtin_amdgcn_readlane((int)m.exit_in, 63);
            const int last_open_lane = open_lanes ? 63 - __builtin_clzll(open_lanes) : -1;
            if (exit_in && open_lanes) pend_rel = (uint32_t)__builtin_amdgcn_readlane((int)lastD, last_open_lane) - 1u;
            const int first_quote_lane = quote_lanes ? __builtin_ctzll(quote_lanes) : -1;
            const uint32_t myfc = m.CL ? base + sj_str_offset(m, (uint32_t)__builtin_ctzll(m.CL)) : 0u;
            if (entry_in && quote_lanes) fclose_rel = (uint32_t)__builtin_amdgcn_readlane((int)myfc, first_quote_lane);
            // ---- errors (rare): the first error of a string travels to the lane that writes its header ----
            const sj_u64 e_all = m.e4 | m.e5 | m.e6 | m.e7 | m.e8;
            any_err = __ballot(e_all != 0) != 0;
            uint32_t pend_err = 0;
            if (any_err) {
                uint32_t epos = 0;
                const uint32_t e_any = sj_str_first_error(m, ~0ull, &epos);
                if (e_any) {  // the launch's first error by position
                    const sj_u64 v = ((blk * 64 + epos) << 8) | e_any;
                    atomicMax(reinterpret_cast<unsigned long long*>(&a.res->first_error_inv), ~v);
                }
                uint32_t e_tail = e_any, e_head = e_any, dummy;
                if (m.O) {
                    const uint32_t o = 63u - (uint32_t)__builtin_clzll(m.O);
                    e_tail = sj_str_first_error(m, o == 63 ? 0ull : ~((2ull << o) - 1ull), &dummy);
                }
                if (m.CL) {
                    const uint32_t c = (uint32_t)__builtin_ctzll(m.CL);
                    e_head = sj_str_first_error(m, c == 63 ? ~0ull : ((2ull << c) - 1ull), &dummy);
                }
                // lanes whose first closing quote closes a string opened in an earlier lane of this granule
                const bool cross = m.CL && !(m.O & ((1ull << __builtin_ctzll(m.CL)) - 1ull)) && prevD;
                for (sj_u64 todo = __ballot(cross); todo; todo &= todo - 1) {
                    const int L = __builtin_ctzll(todo);
                    const int Lo = __builtin_amdgcn_readlane((int)prevL, L);
                    const uint32_t cand = lane == Lo ? e_tail : ((lane > Lo && lane < L) ? e_any : (lane == L ? e_head : 0u));
                    const sj_u64 bal = __ballot(cand != 0);
                    const uint32_t code = bal ? (uint32_t)__builtin_amdgcn_readlane((int)cand, __builtin_ctzll(bal)) : 0u;
                    if (lane == L) xerr = code;
                }
                if (exit_in) {  // the string that stays open: its errors inside this granule
                    const uint32_t cand = lane == last_open_lane ? e_tail : (lane > last_open_lane ? e_any : 0u);
                    const sj_u64 bal = __ballot(cand != 0);
                    pend_err = bal ? (uint32_t)__builtin_amdgcn_readlane((int)cand, __builtin_ctzll(bal)) : 0u;
                }
                if (fclose_rel != STR_NONE) {  // the string that was open at the granule's entry: its errors up to its closing quote
                    const uint32_t cand = lane < first_quote_lane ? e_any : (lane == first_quote_lane ? e_head : 0u);
                    const sj_u64 bal = __ballot(cand != 0);
                    fclose_err = bal ? (uint32_t)__builtin_amdgcn_readlane((int)cand, __builtin_ctzll(bal)) : 0u;
                }
                err_wave = 1;
            }
            if (lane == 0) {
                sg_store(&orec[cur], OR_VALID | ((exit_in && open_lanes) ? OR_HAS_OPEN : 0ull) | ((sj_u64)pend_err << 16) |
                                         (sj_u64)(pend_rel == STR_NONE ? 0u : pend_rel));
                sg_store(&agg[cur], SG_AGG | ((sj_u64)(any_err ? 1u : 0u) << 36) | ((sj_u64)tot_open << 20) | (sj_u64)tot_out);
                if (retire != 0) tk = __hip_atomic_fetch_add(my_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        uint32_t nxt = STR_NONE;
        if (have && retire != 0) nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk) * NC + cls;
        // \uXXXX: the hex digits of a lane's first two sequences are requested now and used behind the flush and the copy
        // (a request per trip of the loop there costs a memory round trip per trip)
        // Round 5: the sequences are patched DENSELY.  A lane-local loop over its own sequences runs to the busiest lane's count (3-4
        // for a mean of 0.5 on the documents of configs[3]) at ~85 instructions a trip -- 12 % of the pass there.  Now each lane only
        // queues {where its UTF-8 bytes go in the tile, where its digits are, pair?} in LDS, and the decoding runs with one lane per
        // sequence (up to 64 at a time; their digits requested here and used behind the flush and the copy).
        uint32_t it_lo[2] = {0, 0}, it_hi[2] = {0, 0};
        uint32_t u_n = 0, u_ent = 0, u_lo = 0, u_hi = 0;
        const sj_u64 items = have ? (m.l1 | m.l2 | m.l3 | m.pair) : 0ull;
        const bool any_items = __ballot(items != 0) != 0;
        bool u_dense = false;
        if (any_items) {
            const uint32_t n_it = (uint32_t)__popcll(items);
            const uint32_t sc = str_incl_scan(n_it);
            const uint32_t tot_it = (uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
            u_dense = tot_it <= STR_UQ_ITEMS;
            if (u_dense) {
                uint32_t* const uq = s_uq[wave];
                uint32_t i = sc - n_it;
                for (sj_u64 x = items; x; x &= x - 1, ++i) {
                    const uint32_t e = (uint32_t)__builtin_ctzll(x);
                    const uint32_t dest = base + sj_str_offset(m, e);  // (< STR_TILE_BYTES: 14 bits)
                    uq[i] = dest | ((((uint32_t)lane << 6) | e) << 14) | ((uint32_t)((m.pair >> e) & 1ull) << 26);
                }
                str_lds_fence();
                u_n = tot_it;
                if ((uint32_t)lane < u_n) {
                    u_ent = uq[lane];
                    const uint8_t* src = a.buf + (sj_u64)cur * 4096 + ((u_ent >> 14) & 0xFFFu);
                    u_lo = reinterpret_cast<const StrU4B*>(src - 3)->a;
                    if (u_ent >> 26) u_hi = reinterpret_cast<const StrU4B*>(src - 9)->a;
                }
            } else {  // (more sequences than the queue holds: the lane-local loop, its first two items' digits requested here)
                sj_u64 x = items;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (x) {
                        const uint8_t* src = a.buf + blk * 64 + (uint32_t)__builtin_ctzll(x);
                        it_lo[t] = reinterpret_cast<const StrU4B*>(src - 3)->a;
                        if (m.pair & x & (0 - x)) it_hi[t] = reinterpret_cast<const StrU4B*>(src - 9)->a;
                        x &= x - 1;
                    }
                }
            }
        }

        // =================== flush granule `prev` ===================
        if (prev != STR_NONE) {
            sj_u64 outbase = 0;
            uint32_t ordbase = 0;
            if (prev != 0) {
#ifdef SJMI_STR_SPINSTAT
                if ((pf >> 62) != 2) ++stat_waits;
                ++stat_flushes;
#endif
                for (uint32_t spins = 0; (pf >> 62) != 2; ++spins) {
#ifdef SJMI_STR_SPINSTAT
                    ++stat_spins;
#endif
                    if (spins > STR_SPIN_LIMIT) {  // never expected: the scanner is not running
                        if (lane == 0) {
                            __hip_atomic_fetch_or(&a.wsflags[0], SJMI_ST_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_fetch_or(&a.res->flags, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        break;
                    }
                    if (spins) __builtin_amdgcn_s_sleep(1);  // (no sleep: 0.407 ms instead of 0.389 -- the polls compete with the stores; 2 .. 8: no change)
                    pf = sg_load(&pfx[prev - 1]);
                }
                outbase = pf & 0xFFFFFFFFull;
                ordbase = (uint32_t)(pf >> 32) & 0x3FFFFFFFu;
            }
            const bool fits = (pf >> 62) == 2 || prev == 0 ? (outbase != 0xFFFFFFFFull && outbase + prev_n <= a.sb_cap) : false;
            if (fits && !(SJMI_STR_ABL & 4)) {
                // 16-byte chunks of the tile, stored to byte-granular addresses (gfx950 global memory runs in unaligned access
                // mode; measured against an LDS funnel shift to 16-byte aligned stores: 3.6 % faster).  The two chunks around a
                // header that another granule will write, and the tail, go byte by byte, one byte per lane.
                struct __attribute__((packed, aligned(1))) StrU16B { uint32_t a, b, c, d; };
                const uint32_t pq = prev_pend != STR_NONE ? prev_pend >> 4 : 0x7FFFFFF0u;
                for (uint32_t q = (uint32_t)lane; 16u * q + 16u <= prev_n; q += 64) {
                    if (q - pq > 1u) {
                        const uint4 v = reinterpret_cast<const uint4*>(tile)[q];
                        StrU16B o = {v.x, v.y, v.z, v.w};
                        *reinterpret_cast<StrU16B*>(a.sb + outbase + 16u * q) = o;
                    }
                }
                {
                    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile);
                    int pos;
                    if (lane < 16) pos = (int)((prev_n & ~15u) + (uint32_t)lane);
                    else if (lane < 32) pos = -1;
                    else pos = prev_pend != STR_NONE ? (int)(16u * pq + ((uint32_t)lane - 32u)) : -1;
                    if (pos >= 0 && (uint32_t)pos < prev_n && !(prev_pend != STR_NONE && (uint32_t)pos - prev_pend < 4u))
                        a.sb[outbase + (uint32_t)pos] = tb[pos];
                }
            }
            // the string that was open when the granule began: its header lives in an earlier granule
            if (prev_fclose != STR_NONE && ((pf >> 62) == 2 || prev == 0)) {
                const sj_u64 absDc = outbase + prev_fclose;
                uint32_t g = prev - 1, err = prev_fclose_err;
                sj_u64 absDo = ~0ull;
                sj_u64 rec = pf_or, pp = pf_pp;
                for (uint32_t hops = 0; prev != 0; ++hops) {
                    if (hops) {
                        rec = sg_load(&orec[g]);
                        pp = g ? sg_load(&pfx[g - 1]) : SG_PFX;
                    }
                    for (uint32_t spins = 0; !(rec & OR_VALID) || (g && (pp >> 62) != 2); ++spins) {
                        if (spins > STR_SPIN_LIMIT) break;
                        __builtin_amdgcn_s_sleep(1);
                        rec = sg_load(&orec[g]);
                        pp = g ? sg_load(&pfx[g - 1]) : SG_PFX;
                    }
                    if (!(rec & OR_VALID)) break;
                    const uint32_t e2 = (uint32_t)(rec >> 16) & 0xFFu;
                    if (e2) err = e2;  // (an earlier granule: an earlier position)
                    if (rec & OR_HAS_OPEN) {
                        absDo = (g ? (pp & 0xFFFFFFFFull) : 0ull) + (rec & 0xFFFFull);
                        break;
                    }
                    if (g == 0) break;
                    --g;
                }
                if (absDo == ~0ull) {  // no opening quote in front of a closing one: the parities do not belong to this document
                    if (lane == 0) __hip_atomic_fetch_or(&a.res->flags, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (lane < 4 && absDo + 4 <= a.sb_cap && absDc >= absDo + 4) {
                    const uint32_t n = (uint32_t)(absDc - absDo - 4);
                    const uint32_t hdr = err ? (0x00FFFFFFu | (err << 24)) : __builtin_bswap32(n);
                    a.sb[absDo + (uint32_t)lane] = (uint8_t)(hdr >> (8 * lane));
                }
            }
            if constexpr (SOFF) {
                if ((pf >> 62) == 2 || prev == 0) {
                    if (a.blk_ord) {
                        const sj_u64 pb = (sj_u64)prev * 64 + lane;
                        if (pb < nblocks) a.blk_ord[pb] = ordbase + prev_oexcl;
                    }
                    if (a.soff && prev_nopen) {
                        const sj_u64 room = a.soff_cap > ordbase ? a.soff_cap - ordbase : 0;
                        const uint32_t nst = (sj_u64)prev_nopen < room ? prev_nopen : (uint32_t)room;
                        const uint16_t* const so = sh[wave].so;
                        for (uint32_t j = (uint32_t)lane; j < nst; j += 64) a.soff[ordbase + j] = (uint32_t)outbase + so[j];
                    }
                }
            }
            str_lds_fence();
        }
        if (!have) break;

        // =================== copy granule `cur` into the tile ===================
        {
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint32_t q = (uint32_t)lane; q * 16u < tot_out + 32u; q += 64) reinterpret_cast<uint4*>(tile)[q] = z;
            str_lds_fence();
            const SjStrGroups g = sj_str_groups(m.K, m.O);
            const sj_u64 Kc = m.K & ~(g.bad << 1);
            const uint32_t B = base + m.head;
            const uint32_t Klo = (uint32_t)m.K, Khi = (uint32_t)(m.K >> 32), Olo = (uint32_t)m.O, Ohi = (uint32_t)(m.O >> 32);
            const uint32_t Slo = (uint32_t)g.oshift, Shi = (uint32_t)(g.oshift >> 32);
            const uint32_t Clo = (uint32_t)Kc, Chi = (uint32_t)(Kc >> 32);
            const uint32_t Bhi = B + (uint32_t)__popc(Klo) + 4u * (uint32_t)__popc(Olo);
#pragma unroll
            for (int i = 0; i < ((SJMI_STR_ABL & 2) ? 0 : 16); ++i) {
                const int sft = 4 * (i & 7);
                const uint32_t ltm = (1u << sft) - 1u, grp = 0xFu << sft;
                const uint32_t Kh = i < 8 ? Klo : Khi, Oh = i < 8 ? Olo : Ohi, Sh = i < 8 ? Slo : Shi, Ch = i < 8 ? Clo : Chi;
                const uint32_t nib = (Ch >> sft) & 15u;
                const uint32_t dest = (i < 8 ? B : Bhi) + (uint32_t)__popc(Kh & ltm) + 4u * ((uint32_t)__popc(Oh & ltm) + (uint32_t)__popc(Sh & grp));
                const uint32_t packed = __builtin_amdgcn_perm(0u, w[i], s_lut[nib]);
                tile_or(tile, dest, packed);
            }
            if (__ballot(g.bad != 0)) {  // "a""b": the byte behind the opening quote goes behind the new header
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if ((g.bad >> (4 * i + 2)) & 1ull) tile_or(tile, base + sj_str_offset(m, 4u * i + 3u), w[i] >> 24);
            }
            // the next granule's bytes: requested as soon as this granule's are in the tile, in flight during the headers,
            // the patches and the first instructions of the next classification (a second register buffer for an earlier
            // request costs a wave per SIMD)
            asm volatile("" ::: "memory");
            str_load_step(d, a.buf, (sj_u64)nxt * 64 + lane, nblocks);
            asm volatile("" ::: "memory");
            str_lds_fence();
            // ---- headers, by the closing quotes: length = kept bytes between the quotes (32-bit halves: 64-bit shifts by
            //      a variable are slow) ----
            if (SJMI_STR_ABL & 1) {
            } else if (!any_err) {
                // The records of a block lie back to back -- [be32 length][bytes][be32 length][bytes] ...: nothing is kept between
                // a closing quote and the next opening one -- so the header of the NEXT string begins where the output offset of a
                // closing quote ends, and that offset is one v_bcnt (popcount + addend) once the header slots in front of it are
                // counted by the trip number: quotes alternate, so the k-th closing quote of the block has k openings in front of
                // it, one more if the block is entered outside a string.  (Round 3 located every header from scratch: two more
                // popcounts, a v_ffbh and the arithmetic around them per closing quote -- 34 instead of 19 instructions a trip, and
                // the trips are the busiest lane's.)
                uint32_t CLlo = (uint32_t)m.CL, CLhi = (uint32_t)(m.CL >> 32);
                // where the content of the string that is open (or opens first) begins = its header + 4
                uint32_t from = (entered_in ? prevD - 1u : B) + 4u;
                uint32_t slots = entered_in ? B : B + 4u;  // B + 4 x openings in front of the closing quote at hand
                // the ordinal (inside the granule) of the string the closing quote at hand closes: the lane's own strings from
                // oexcl on; the string it was entered in is the last one opened in front of the lane (parked by its own lane as
                // well: the same value)
                uint16_t* so = nullptr;
                if constexpr (SOFF) so = sh[wave].so + oexcl - ((entered_in && prevD) ? 1u : 0u);
                if (entered_in && !prevD) {
                    // opened in an earlier granule (the flush writes that header): pass over the block's first closing quote
                    const uint32_t inlo = CLlo != 0;
                    const uint32_t c = (uint32_t)__builtin_ctz(inlo ? CLlo : (CLhi ? CLhi : 1u));
                    const uint32_t ltc = (1u << c) - 1u;
                    const uint32_t end = inlo ? slots + (uint32_t)__popc(Klo & ltc) : slots + (uint32_t)__popc(Klo) + (uint32_t)__popc(Khi & ltc);
                    if (inlo) CLlo &= CLlo - 1;
                    else CLhi &= CLhi - 1;
                    from = end + 4u;
                    slots += 4u;
                }
                for (uint32_t cl = CLlo; cl; cl &= cl - 1) {
                    const uint32_t ltc = (1u << (uint32_t)__builtin_ctz(cl)) - 1u;
                    const uint32_t end = (uint32_t)__popc(Klo & ltc) + slots;
                    tile_or(tile, from - 4u, __builtin_bswap32(end - from));
                    if constexpr (SOFF) *so++ = (uint16_t)(from - 4u);
                    from = end + 4u;
                    slots += 4u;
                }
                slots += (uint32_t)__popc(Klo);
                for (uint32_t cl = CLhi; cl; cl &= cl - 1) {
                    const uint32_t ltc = (1u << (uint32_t)__builtin_ctz(cl)) - 1u;
                    const uint32_t end = (uint32_t)__popc(Khi & ltc) + slots;
                    tile_or(tile, from - 4u, __builtin_bswap32(end - from));
                    if constexpr (SOFF) *so++ = (uint16_t)(from - 4u);
                    from = end + 4u;
                    slots += 4u;
                }
                // the lane's last string stays open: its header is where the next content would begin
                if constexpr (SOFF) {
                    if (m.O && m.exit_in) *so = (uint16_t)(from - 4u);
                }
            } else {  // (rare: some string of the wave's 4 KiB has a malformed escape)
                if constexpr (SOFF) {  // record offsets: every opening quote by itself
                    uint16_t* so = sh[wave].so + oexcl;
                    for (sj_u64 o = m.O; o; o &= o - 1) {
                        const sj_u64 lt = (o & (0 - o)) - 1ull;
                        *so++ = (uint16_t)(B + (uint32_t)__popcll(m.K & lt) + 4u * (uint32_t)__popcll(m.O & lt));
                    }
                }
                for (sj_u64 cl = m.CL; cl; cl &= cl - 1) {
                    const uint32_t c = (uint32_t)__builtin_ctzll(cl);
                    const sj_u64 lt_c = (1ull << c) - 1ull;
                    const sj_u64 olt = m.O & lt_c;
                    uint32_t Do, err = 0;
                    if (olt) {
                        const uint32_t o = 63u - (uint32_t)__builtin_clzll(olt);
                        Do = base + sj_str_offset(m, o);
                        uint32_t dummy;
                        err = sj_str_first_error(m, (c == 63 ? ~0ull : ((2ull << c) - 1ull)) & ~((2ull << o) - 1ull), &dummy);
                    } else if (prevD) {
                        Do = prevD - 1u;
                        err = xerr;
                    } else {
                        continue;
                    }
                    const uint32_t n = base + sj_str_offset(m, c) - Do - 4u;
                    tile_or(tile, Do, err ? (0x00FFFFFFu | (err << 24)) : __builtin_bswap32(n));
                }
            }
            // ---- escapes that change the byte: XOR the difference in (n -> 0A, t -> 09, r -> 0D, b -> 08, f -> 0C) ----
            {
                const sj_u64 P = m.pn | m.pt | m.pr | m.pbf;
                if (__ballot(P != 0)) {
                    const uint32_t Plo = (uint32_t)P, Phi = (uint32_t)(P >> 32);
                    for (uint32_t x = Plo; x; x &= x - 1) {
                        const uint32_t bit = x & (0u - x), lt = bit - 1u;
                        const uint32_t dl = ((uint32_t)m.pn & bit) ? 0x64u : ((uint32_t)m.pt & bit) ? 0x7Du : ((uint32_t)m.pr & bit) ? 0x7Fu : 0x6Au;
                        tile_xor(tile, B + (uint32_t)__popc(Klo & lt) + 4u * (uint32_t)__popc(Olo & lt), dl);
                    }
                    for (uint32_t x = Phi; x; x &= x - 1) {
                        const uint32_t bit = x & (0u - x), lt = bit - 1u;
                        const uint32_t dl = ((uint32_t)(m.pn >> 32) & bit) ? 0x64u : ((uint32_t)(m.pt >> 32) & bit) ? 0x7Du : ((uint32_t)(m.pr >> 32) & bit) ? 0x7Fu : 0x6Au;
                        tile_xor(tile, Bhi + (uint32_t)__popc(Khi & lt) + 4u * (uint32_t)__popc(Ohi & lt), dl);
                    }
                }
            }
            if (any_items && u_dense && !(SJMI_STR_ABL & 8)) {  // \uXXXX: the UTF-8 bytes over the last hex digits (StringParser.java:126-153)
                const uint32_t* const uq = s_uq[wave];
                for (uint32_t c0 = 0; c0 < u_n; c0 += 64) {
                    const uint32_t it = c0 + (uint32_t)lane;
                    uint32_t ent = u_ent, lo = u_lo, hi = u_hi;
                    if (c0) {  // (more than 64 sequences in 4 KiB: their digits are fetched now)
                        ent = it < u_n ? uq[it] : 0u;
                        const uint8_t* src = a.buf + (sj_u64)cur * 4096 + ((ent >> 14) & 0xFFFu);
                        lo = it < u_n ? reinterpret_cast<const StrU4B*>(src - 3)->a : 0u;
                        hi = (it < u_n && (ent >> 26)) ? reinterpret_cast<const StrU4B*>(src - 9)->a : 0u;
                    }
                    if (it < u_n) {
                        const uint32_t e = (ent >> 14) & 63u;
                        const bool is_pair = (ent >> 26) != 0;
                        uint32_t cp = sj_hex4_valid_word(lo);  // (items exist only for sequences whose digits the plane algebra found valid)
                        if (is_pair) cp = (((sj_hex4_valid_word(hi) - 0xD800u) << 10) | (cp - 0xDC00u)) + 0x10000u;
                        uint32_t L;
                        const uint32_t nb = sj_utf8_bytes(cp, &L);
                        uint32_t old = L == 4 ? lo : (lo >> (8u * (4u - L)));
                        const uint32_t spilled = L - 1u > e ? L - 1u - e : 0u;  // slots in front of the block: nothing was copied there
                        if (spilled) old &= ~0u << (8u * spilled);
                        tile_xor(tile, (ent & 0x3FFFu) - (L - 1u), old ^ nb);
                    }
                }
            } else if (any_items && !(SJMI_STR_ABL & 8)) {
                int t = 0;
                for (sj_u64 x = items; x; x &= x - 1, ++t) {
                    const uint32_t e = (uint32_t)__builtin_ctzll(x);
                    const uint8_t* src = a.buf + blk * 64 + e;
                    const bool is_pair = (m.pair >> e) & 1ull;
                    uint32_t lo, hi = 0;
                    if (t == 0) {
                        lo = it_lo[0];
                        hi = it_hi[0];
                    } else if (t == 1) {
                        lo = it_lo[1];
                        hi = it_hi[1];
                    } else {
                        lo = reinterpret_cast<const StrU4B*>(src - 3)->a;
                        if (is_pair) hi = reinterpret_cast<const StrU4B*>(src - 9)->a;
                    }
                    uint32_t cp = sj_hex4_valid_word(lo);
                    if (is_pair) cp = (((sj_hex4_valid_word(hi) - 0xD800u) << 10) | (cp - 0xDC00u)) + 0x10000u;
                    uint32_t L;
                    const uint32_t nb = sj_utf8_bytes(cp, &L);
                    uint32_t old = L == 4 ? lo : (lo >> (8u * (4u - L)));
                    const uint32_t spilled = L - 1u > e ? L - 1u - e : 0u;
                    if (spilled) old &= ~0u << (8u * spilled);
                    tile_xor(tile, base + sj_str_offset(m, e) - (L - 1u), old ^ nb);
                }
            }
            str_lds_fence();
        }
        // one classification ahead of their use: the prefix in front of this granule (and what a closing quote needs)
        if (cur != 0) {
            pf = sg_load(&pfx[cur - 1]);
            pf_or = sg_load(&orec[cur - 1]);
            pf_pp = cur >= 2 ? sg_load(&pfx[cur - 2]) : SG_PFX;
        } else {
            pf = SG_PFX;
            pf_or = 0;
            pf_pp = SG_PFX;
        }
        prev = cur;
        prev_n = tot_out;
        prev_nopen = tot_open;
        prev_pend = pend_rel;
        prev_fclose = fclose_rel;
        prev_fclose_err = fclose_err;
        prev_oexcl = oexcl;
        cur = nxt;
    }
    (void)err_wave;
#ifdef SJMI_STR_SPINSTAT  // experiments only: how often and how long the flush waited for its prefix (in the result record)
    if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&a.res->first_error_inv), stat_waits | (stat_spins << 20) | (stat_flushes << 44));
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
constexpr size_t STR_WS_FLAGS_OFFSET = 0;     // u32 status, u32 scanner CU, u32 role ticket (workgroups in their order of arrival)
constexpr size_t STR_WS_RESULT_OFFSET = 32;   // a result record zeroed with the workspace (strings_workspace_result)
constexpr size_t STR_WS_TICKET_OFFSET = 64;   // the ticket counters, 64 bytes apart
constexpr size_t STR_WS_STATE_OFFSET = 64 + 32 * 64;  // (room for 32 ticket counters)

static uint64_t str_granules(uint64_t len) { return (len / 64 + 1 + 63) / 64; }
size_t strings_workspace_bytes(uint64_t len) { return STR_WS_STATE_OFFSET + 3 * (size_t)str_granules(len) * sizeof(sj_u64) + 64; }
size_t strings_parity_words(uint64_t len) { return (size_t)str_granules(len) + 4; }
// a result record inside the workspace: strings_launch zeroes it with the rest (one memset less on the latency path)
UnescapeResult* strings_workspace_result(void* d_ws) { return reinterpret_cast<UnescapeResult*>(static_cast<uint8_t*>(d_ws) + STR_WS_RESULT_OFFSET); }

template <bool SOFF>
static hipError_t str_resident(unsigned* out) {
    static std::atomic<unsigned> cached[16];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16 || !cached[dev]) {
        int per_cu = 0, cus = 0;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_strings<SOFF>, 256, 0)) != hipSuccess) return e;
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

// d_res must have been zeroed by the caller's stream order (first_error_inv / flags accumulate); the records of every
// string literal of buf[0, len) go to d_sb; optional: d_soff (offset of record k), d_blk_ord (see StrArgs)
hipError_t strings_launch(const uint8_t* d_buf, uint64_t len, const unsigned long long* d_blkpar, uint8_t* d_sb, uint64_t sb_cap,
                          uint32_t* d_soff, uint64_t soff_cap, uint32_t* d_blk_ord, void* d_ws, UnescapeResult* d_res,
                          hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, const StringsAlt& alt, bool workspace_is_zero,
                          const uint32_t* d_skip) {
    const uint64_t ngran = str_granules(len);
    hipError_t e = workspace_is_zero ? hipSuccess : hipMemsetAsync(d_ws, 0, strings_workspace_bytes(len), stream);
    if (e != hipSuccess) return e;
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    StrArgs a;
    a.buf = d_buf;
    a.len = len;
    a.blkpar = d_blkpar;
    a.sb = d_sb;
    a.sb_cap = sb_cap < 0xFFFFFFF0ull ? sb_cap : 0xFFFFFFF0ull;
    a.soff = d_soff;
    a.soff_cap = soff_cap;
    a.blk_ord = d_blk_ord;
    a.gstate = reinterpret_cast<sj_u64*>(ws + STR_WS_STATE_OFFSET);
    a.ticket = reinterpret_cast<uint32_t*>(ws + STR_WS_TICKET_OFFSET);
    a.wsflags = reinterpret_cast<uint32_t*>(ws + STR_WS_FLAGS_OFFSET);
    a.res = d_res;
    a.ngran = (uint32_t)ngran;
    a.flags = 0;
    a.sel = alt.d_sel;
    a.buf_alt = alt.d_buf;
    a.blkpar_alt = reinterpret_cast<const sj_u64*>(alt.d_blkpar);
    a.skip = d_skip;
    const bool soff = d_soff != nullptr || d_blk_ord != nullptr;
    unsigned resident = 0;
    e = soff ? str_resident<true>(&resident) : str_resident<false>(&resident);
    if (e != hipSuccess) return e;
    const uint64_t want = (ngran + 3) / 4 + 1;
    const dim3 grid((unsigned)(want < resident ? want : resident)), block(256);
    if (ev_start && ev_stop) {
        if (soff) hipExtLaunchKernelGGL((k_strings<true>), grid, block, 0, stream, ev_start, ev_stop, 0, a);
        else hipExtLaunchKernelGGL((k_strings<false>), grid, block, 0, stream, ev_start, ev_stop, 0, a);
    } else {
        if (soff) hipLaunchKernelGGL((k_strings<true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k_strings<false>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

// ---- batches ------------------------------------------------------------------------------------------------------------
// The string pass is one stream over the packed batch, so every document has to begin outside a string and behind an even
// backslash run.  That holds when all documents pass stage 1 and end in white space; a batch whose documents were indexed
// one by one (isolated mode: a broken document must not touch its neighbours) is therefore run on a SANITIZED COPY: the
// documents without structurals (the failed ones) blanked, a trailing odd backslash run shortened by one.  What survives is
// byte-identical to the batch where it matters (inside the strings of the surviving documents), at the same positions.
__global__ void __launch_bounds__(256)
k_batch_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t chunks, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// one wave per document
__global__ void __launch_bounds__(256)
k_batch_blank(uint8_t* __restrict__ copy, const unsigned long long* __restrict__ doc_offsets, const unsigned long long* __restrict__ index_offsets,
              uint64_t n_docs, const uint32_t* __restrict__ skip, const uint32_t* __restrict__ doc_status, unsigned long long total_len) {
    if (skip && *skip) return;
    const int lane = threadIdx.x & 63;
    for (uint64_t k = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < n_docs; k += (uint64_t)gridDim.x * 4) {
        // (device-resident offsets cannot be validated by the host: clamped to the copy)
        unsigned long long lo = doc_offsets[k], hi = doc_offsets[k + 1];
        if (hi > total_len) hi = total_len;
        if (lo > hi) lo = hi;
        if (doc_status ? doc_status[k] != 0 : index_offsets[k + 1] == index_offsets[k]) {
            for (unsigned long long p = lo + lane; p < hi; p += 64) copy[p] = 0x20;
        } else if (lane == 0 && hi > lo) {
            unsigned long long p = hi, run = 0;
            while (p > lo && copy[p - 1] == 0x5C) { --p; ++run; }
            if (run & 1) copy[hi - 1] = 0x20;
        }
    }
}
hipError_t strings_sanitize_launch(const uint8_t* d_buf, uint64_t total_len, const unsigned long long* d_doc_offsets,
                                   const unsigned long long* d_index_offsets, uint64_t n_docs, uint8_t* d_copy, const uint32_t* d_skip,
                                   hipStream_t stream, const uint32_t* d_doc_status) {
    const uint64_t chunks = (total_len + SJMI_PADDING + 15) / 16;
    const unsigned g1 = (unsigned)((chunks + 255) / 256 < 4096 ? (chunks + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_batch_copy, dim3(g1 ? g1 : 1), dim3(256), 0, stream, reinterpret_cast<const uint4*>(d_buf), reinterpret_cast<uint4*>(d_copy),
                       chunks, d_skip);
    const unsigned g2 = (unsigned)((n_docs + 3) / 4 < 8192 ? (n_docs + 3) / 4 : 8192);
    hipLaunchKernelGGL(k_batch_blank, dim3(g2 ? g2 : 1), dim3(256), 0, stream, d_copy, d_doc_offsets, d_index_offsets, n_docs, d_skip,
                       d_doc_status, (unsigned long long)total_len);
    return hipGetLastError();
}

// doc_ord[k] (n_docs + 1 entries) = ordinal of the first string opened at or behind document k's first byte: the ordinal of its
// block (blk_ord, left by the string pass) + the strings opened in that block in front of it (a byte loop over < 64 bytes from
// the block's entry state); doc_str_offsets[k] (optional) = offset of that string's record, or the total behind the last one.
__global__ void __launch_bounds__(256)
k_doc_str_ordinals(const uint8_t* __restrict__ buf0, const uint8_t* __restrict__ buf1, const sj_u64* __restrict__ par0,
                   const sj_u64* __restrict__ par1, const uint32_t* __restrict__ sel, uint64_t len,
                   const unsigned long long* __restrict__ doc_offsets, uint64_t n_docs, const uint32_t* __restrict__ blk_ord,
                   const uint32_t* __restrict__ soff, const UnescapeResult* __restrict__ res, unsigned long long* __restrict__ doc_ord,
                   unsigned long long* __restrict__ doc_str_offsets, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;  // (the fused pipeline's accepted plain pass: batch.hip k_doc_prepare computes these)
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_docs) return;
    const bool alt = sel && *sel == 0;
    const uint8_t* buf = alt ? buf1 : buf0;
    const sj_u64* par = alt ? par1 : par0;
    const unsigned long long nstr = res->reserved;
    unsigned long long ord = nstr;
    unsigned long long pos = doc_offsets[k];
    if (!(res->flags & 0xEu)) {
        if (pos > len) pos = len;
        const unsigned long long b = pos >> 6, start = b * 64;
        ord = blk_ord[b];
        if (pos > start) {
            // the strings opened in [start, pos) (sj_strings.h: counted on the bytes, from the block's entry state)
            const uint32_t in_str = (uint32_t)(par[b >> 6] >> (b & 63)) & 1u;
            const uint32_t e_in = b ? sj_backslash_run_parity(buf, 0, start) : 0u;
            ord += (unsigned long long)sj_str_opens_before(buf, start, (uint32_t)(pos - start), in_str, e_in);
        }
        if (ord > nstr) ord = nstr;
    }
    doc_ord[k] = ord;
    if (doc_str_offsets) doc_str_offsets[k] = ord < nstr ? soff[ord] : res->total_bytes;
}
hipError_t strings_doc_ordinals_launch(const uint8_t* d_buf, const unsigned long long* d_blkpar, const StringsAlt& alt, uint64_t len,
                                       const unsigned long long* d_doc_offsets, uint64_t n_docs, const uint32_t* d_blk_ord,
                                       const uint32_t* d_soff, const UnescapeResult* d_res, unsigned long long* d_doc_ord,
                                       unsigned long long* d_doc_str_offsets, hipStream_t stream, const uint32_t* d_skip) {
    hipLaunchKernelGGL(k_doc_str_ordinals, dim3((unsigned)((n_docs + 1 + 255) / 256)), dim3(256), 0, stream, d_buf, alt.d_buf,
                       reinterpret_cast<const sj_u64*>(d_blkpar), reinterpret_cast<const sj_u64*>(alt.d_blkpar), alt.d_sel, len, d_doc_offsets,
                       n_docs, d_blk_ord, d_soff, d_res, d_doc_ord, d_doc_str_offsets, d_skip);
    return hipGetLastError();
}

// Host forms of the C ABI report the first failing string by its position in indexes[]: the last structural at or in front
// of the error's byte position is that string's opening quote (nothing inside a string is a structural).  One thread,
// queued behind the string pass; *out = UINT64_MAX when there is no error (or no such structural).
__global__ void k_error_index(const uint32_t* __restrict__ idx, uint64_t count, const Stage1Result* __restrict__ dev_count,
                              const UnescapeResult* __restrict__ res, unsigned long long* __restrict__ out) {
    unsigned long long r = ~0ull;
    if (res->first_error_inv) {
        if (dev_count) count = (dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL)) ? 0 : dev_count->count;
        const unsigned long long pos = (~res->first_error_inv) >> 8;
        uint64_t lo = 0, hi = count;  // first index with idx[i] > pos
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (idx[mid] <= pos) lo = mid + 1; else hi = mid;
        }
        if (lo > 0) r = lo - 1;
    }
    *out = r;
}
// the same, plus the two result records behind the index: {index, stage-1 record, string record} in one place = ONE D2H for the
// drop-in call on a small document instead of three
struct __attribute__((aligned(8))) ParsePack {
    unsigned long long err_index;
    Stage1Result s1;
    UnescapeResult u;
};
__global__ void k_error_index_pack(const uint32_t* __restrict__ idx, const Stage1Result* __restrict__ dev_count,
                                   const UnescapeResult* __restrict__ res, ParsePack* __restrict__ out) {
    unsigned long long r = ~0ull;
    const Stage1Result s1 = *dev_count;
    const UnescapeResult u = *res;
    if (u.first_error_inv) {
        const uint64_t count = (s1.status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL)) ? 0 : s1.count;
        const unsigned long long pos = (~u.first_error_inv) >> 8;
        uint64_t lo = 0, hi = count;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (idx[mid] <= pos) lo = mid + 1; else hi = mid;
        }
        if (lo > 0) r = lo - 1;
    }
    out->err_index = r;
    out->s1 = s1;
    out->u = u;
}
size_t strings_parse_pack_bytes() { return sizeof(ParsePack); }
hipError_t strings_error_index_pack_launch(const uint32_t* d_idx, const Stage1Result* dev_count, const UnescapeResult* d_res, void* d_pack,
                                           hipStream_t stream) {
    hipLaunchKernelGGL(k_error_index_pack, dim3(1), dim3(1), 0, stream, d_idx, dev_count, d_res, static_cast<ParsePack*>(d_pack));
    return hipGetLastError();
}
hipError_t strings_error_index_launch(const uint32_t* d_idx, uint64_t count, const Stage1Result* dev_count, const UnescapeResult* d_res,
                                      unsigned long long* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(k_error_index, dim3(1), dim3(1), 0, stream, d_idx, count, dev_count, d_res, d_out);
    return hipGetLastError();
}

}  // namespace sjmi
