// coop_walk.hip -- stage 2 on the GPU, COOPERATIVE form: JsonIterator.walkDocument
// (/root/reference/src/main/java/org/simdjson/JsonIterator.java:26-200) + TapeBuilder (TapeBuilder.java:41-217) as scans
// and local predicates instead of a sequential automaton.
//
// One WAVE per document, lane j = structural (step * 64 + j) of that document; a document of any length is swept in
// steps of 64 structurals, so the same kernel builds the tape of one large document (twitter.json: 864 steps) and of
// every ~1 KB document of a batch (3 steps).  Per step, everything the sequential walker carries in its state is
// recovered with wave primitives (the formulation was validated first as a Python model against the oracle,
// tools/coop_walk_model.py):
//   * class of each structural from its first byte (a perfect hash + two byte-table lookups); coalesced loads of the
//     indexes, one 16-byte window of the document per structural (consecutive lanes share cache lines: the document is
//     read once);
//   * depth before each structural = running depth + exclusive DPP scan of (+1 open, -1 close);
//   * the container a structural sits in ("bracket matching") = the last opening bracket in front of it whose depth is
//     one less: found among the wave's own 64 structurals with one ballot per depth level present in the step, else in
//     a small per-wave stack (two VGPRs: lane L = level L: tape position, comma count; a 64-bit kind mask), which
//     the step then updates -- the only state carried from step to step besides four running sums; levels 64..1023
//     (the reference's default maxDepth is 1024) live in a per-wave overflow stack in global memory;
//   * the role of a structural (value / key / colon / separator) is a function of its predecessor's class, of whether
//     the predecessor was a key, and of the kind of its container: every grammar test of JsonIterator.java:68-193
//     becomes a local predicate, and the document's error is the failing predicate at the LOWEST position -- exactly
//     where the sequential walker stops;
//   * tape positions = running position + exclusive scan of words per structural (bracket / string / atom 1, number 2,
//     comma / colon 0; the same ladder as the depth scan, three fields of one dword); STRING payloads = the record offsets
//     the string pass (strings.hip) leaves BY STRING ORDINAL: ordinal = running count + popcount of the step's quotes in
//     front of the lane, one 4-byte gather, requested in one step and stored in the next; a closing bracket writes both
//     container words (its own and the opening one, TapeBuilder.java:197-203: element count = commas directly inside
//     + 1, saturated at 0xFFFFFF; empty-container quirk :205-208);
//   * atoms and numbers are parsed by the lane that owns them: atoms and integers of up to 15 digits branch-free out of
//     the window registers (cw_primitive), everything else through sj_number.h (Clinger / Eisel-Lemire on the device).
//     A floating literal of more than 19 significant digits whose two 19-digit neighbours round differently is listed
//     and decided behind the walk by an exact big-integer comparison with the midpoint (sj_bigdec.h, k_slow_doubles).
// Tape words leave as 8-byte stores at consecutive addresses across the lanes (whole lines per step); nothing is
// re-read: FETCH ~ document + 4 B per structural + 4 B per string, WRITE ~ tape.
//
// Handed back to the host walker (doc_errors[k] = SJMI_WALK_NEEDS_HOST) only beyond the reference's defaults: more than
// 1024 open containers (when the caller raised max_depth), more than 65,536 listed literals in one launch, a document of
// more than 2^31 - 256 structurals.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

#include "sj_number.h"
#include "sj_bigdec.h"
#include "sj_tokens.h"
#include "stage1.h"

namespace sjmi {

namespace {

constexpr int CW_LEVELS = 64;              // levels of the per-wave stack in registers
constexpr int CW_OVF_LEVELS = 960;         // deeper levels, in global memory (together: the reference's default maxDepth of 1024)

enum : uint32_t { K_OPEN_A = 0, K_OPEN_O = 1, K_CLOSE_A = 2, K_CLOSE_O = 3, K_COMMA = 4, K_COLON = 5, K_QUOTE = 6, K_PRIM = 7 };

struct __attribute__((packed, aligned(1))) CW16 { uint32_t a, b, c, d; };

// class of a structural from its first byte: (c * 25 >> 4) & 7 is a perfect hash of the seven structural characters
// ('{' 0, ']' 1, ':' 2, '}' 3, ',' 4, '"' 5, '[' 6); two byte-table lookups (v_perm_b32) and one compare instead of seven
__device__ __forceinline__ uint32_t class_of(uint32_t c) {
    const uint32_t h = ((c * 25u) >> 4) & 7u;
    const uint32_t sel = h | 0x0C0C0C00u;  // (selector 0x0C = constant zero byte)
    const uint32_t want = __builtin_amdgcn_perm(0x7B5B222Cu, 0x7D3A5D7Bu, sel);   // bytes 0..7: { ] : } , " [ {
    const uint32_t cls = __builtin_amdgcn_perm(0x07000604u, 0x03050201u, sel);    //             1 2 5 3 4 6 0 7
    return c == want ? cls : K_PRIM;
}

// The wave-wide ladders (inclusive sum, minimum, maximum) as ONE instruction per rung: the VOP2 form with a DPP source
// operand, written in place -- a lane whose source lane does not exist (bound_ctrl 0) or whose row is masked out is not
// written, i.e. keeps its own value, which is what every rung wants.  The compiler's own selection splits each rung into a
// v_mov_b32_dpp plus the operation (25 moves per step of the walker); "s_nop 1" = the two wait states a DPP read of a VGPR
// needs behind the VALU write of that VGPR.
#define CW_DPP_LADDER(OP)                                                         \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"       \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"       \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"       \
    "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"       \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"    \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
__device__ __forceinline__ uint32_t cw_incl_scan(uint32_t v) {
    asm volatile(CW_DPP_LADDER("v_add_u32_dpp") : "+v"(v));
    return v;
}
// wave-wide minimum / maximum of a signed value, uniform result: the full reduction arrives in lane 63 -- no LDS-crossbar
// shuffles on the step's dependency chain
template <bool IS_MAX>
__device__ __forceinline__ int cw_wave_minmax(int v) {
    if (IS_MAX) asm volatile(CW_DPP_LADDER("v_max_i32_dpp") : "+v"(v));
    else asm volatile(CW_DPP_LADDER("v_min_i32_dpp") : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t cw_last(uint32_t incl) { return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63); }

// The bytes of a primitive, for the lane that parses it: the 16-byte window that was loaded at the structural's own
// position serves offsets 0..15 out of registers (a whole atom, most numbers); a longer literal fetches the next 16 bytes
// once, and only literals beyond 32 bytes go to memory byte by byte.
struct CwBytes {
    const uint8_t* buf;
    uint32_t p;
    unsigned long long lo, hi, lo2, hi2;
    bool have2;
    __device__ __forceinline__ uint32_t at(uint32_t q) {
        const uint32_t o = q - p;
        if (o < 16u) return (uint32_t)((o < 8u ? lo >> (8u * o) : hi >> (8u * (o - 8u))) & 0xFFu);
        if (o < 32u) {
            if (!have2) {
                const CW16 v = *reinterpret_cast<const CW16*>(buf + p + 16);
                lo2 = (unsigned long long)v.a | ((unsigned long long)v.b << 32);
                hi2 = (unsigned long long)v.c | ((unsigned long long)v.d << 32);
                have2 = true;
            }
            return (uint32_t)((o < 24u ? lo2 >> (8u * (o - 16u)) : hi2 >> (8u * (o - 24u))) & 0xFFu);
        }
        return buf[q];
    }
};

constexpr uint32_t CW_TRUE = 0x65757274u, CW_FALS = 0x736c6166u, CW_NULL = 0x6c6c756eu;
// internal: a floating literal of more than 19 significant digits whose two 19-digit neighbours round to different doubles;
// the walker writes the lower candidate and lists the literal for k_slow_doubles (the exact comparison of sj_bigdec.h)
constexpr int CW_SLOW_DOUBLE = -2;

// TapeBuilder.visitPrimitive (TapeBuilder.java:70-79) / visitRootPrimitive (:59-68) for one lane; win = the 16 bytes at idx.
// -> 0 and (type, raw second word for numbers) or the SJMI_E_* / SJMI_WALK_NEEDS_HOST code
//
// The walker is VALU-bound and every lane of a step executes every path some lane takes, so the atoms and the common
// integers share ONE branch-free path out of the window registers (a step of 64 structurals nearly always has all of them):
//   * the digit run's length from a "byte > 9" mask (numbers: the digits start one byte up behind a '-');
//   * one terminator test for all: byte 4 of true / null, byte 5 of false, the byte behind the digits of a number;
//   * an integer of at most 15 digits: the digits are moved to the top of a 16-byte field (zeros = leading zeros below them,
//     the terminator and whatever follows shifted out) with dword selects + v_alignbyte_b32; two digits = one v_dot4_u32_u8,
//     the rest 24-bit multiply-adds (a 64-bit multiply chain costs a dozen quarter-rate v_mul_lo/hi_u32).
// Everything else (the root value, which must also end at the document's end; floats, exponents, longer integers, anything
// malformed) takes the scanner of sj_number.h behind a branch no lane of most steps enters.
__device__ int cw_primitive(const uint8_t* buf, const CW16& win, uint32_t idx, bool root, uint32_t end, uint32_t* type,
                            unsigned long long* raw) {
    const uint32_t c = win.a & 0xFFu;
    const bool is_t = c == 't', is_n = c == 'n', is_f = c == 'f';
    const bool is_atom = is_t || is_n || is_f, is_num = c == '-' || c - '0' <= 9u;
    int code = SJMI_E_UNRECOGNIZED_PRIMITIVE;
    bool slow_number = is_num;
    if (!root) {
        const uint32_t sgn = c == '-' ? 1u : 0u;  // (the vacated top byte of a negative number is 0: not a digit)
        const uint32_t x0 = __builtin_amdgcn_alignbyte(win.b, win.a, sgn), x1 = __builtin_amdgcn_alignbyte(win.c, win.b, sgn),
                       x2 = __builtin_amdgcn_alignbyte(win.d, win.c, sgn), x3 = __builtin_amdgcn_alignbyte(0u, win.d, sgn);
        const uint32_t t0 = x0 ^ 0x30303030u, t1 = x1 ^ 0x30303030u, t2 = x2 ^ 0x30303030u, t3 = x3 ^ 0x30303030u;  // digits: 0..9
        auto nondigit = [](uint32_t t) -> uint32_t { return ((t + 0x76767676u) | t) & 0x80808080u; };  // 0x80 in every byte > 9 up to the first
        auto ffbl = [](uint32_t v) -> uint32_t { return v ? (uint32_t)__builtin_ctz(v) : 0xFFFFFFFFu; };
        uint32_t z = ffbl(nondigit(t0));
        const uint32_t z1 = ffbl(nondigit(t1)) | 32u, z2 = ffbl(nondigit(t2)) | 64u, z3 = ffbl(nondigit(t3)) | 96u;
        z = z < z1 ? z : z1;
        z = z < z2 ? z : z2;
        z = z < z3 ? z : z3;
        const uint32_t nd = z >> 3;  // digits in front of the first non-digit (>= 16: none in the window)
        const uint32_t ti = is_num ? nd : (is_f ? 5u : 4u);  // where the terminator sits
        const uint32_t j = ti >> 2, bs = ti & 3u;
        const uint32_t xs = j == 0u ? x0 : j == 1u ? x1 : j == 2u ? x2 : x3;
        const bool sep = sjn_is_structural_or_ws((xs >> (8u * bs)) & 0xFFu);
        const bool atom_ok = win.a == (is_t ? CW_TRUE : is_n ? CW_NULL : CW_FALS) && (!is_f || (win.b & 0xFFu) == 'e') && sep;
        const bool leading_zero = (x0 & 0xFFu) == '0' && nd > 1u;
        const bool fast_number = is_num && nd >= 1u && nd <= 15u && sep && !leading_zero;
        // the 32 bytes {zeros, digits...} shifted down by nd bytes: the last digit lands in byte 15
        const uint32_t a4 = xs ^ 0x30303030u;
        const uint32_t a3 = j == 0u ? 0u : j == 1u ? t0 : j == 2u ? t1 : t2;
        const uint32_t a2 = j < 2u ? 0u : j == 2u ? t0 : t1;
        const uint32_t a1 = j == 3u ? t0 : 0u;
        const uint32_t r3 = __builtin_amdgcn_alignbyte(a4, a3, bs), r2 = __builtin_amdgcn_alignbyte(a3, a2, bs),
                       r1 = __builtin_amdgcn_alignbyte(a2, a1, bs), r0 = __builtin_amdgcn_alignbyte(a1, 0u, bs);
        // eight digits (0..9 per byte, the first in the low byte of the first dword) -> their value
        auto eight = [](uint32_t w0, uint32_t w1) -> uint32_t {
            uint32_t v = __builtin_amdgcn_udot4(w0, 0x0000010Au, 0u, false);          // 10 d0 + d1
            v = __builtin_amdgcn_udot4(w0, 0x010A0000u, __umul24(v, 100u), false);    // ... d3
            v = __builtin_amdgcn_udot4(w1, 0x0000010Au, __umul24(v, 100u), false);
            return __builtin_amdgcn_udot4(w1, 0x010A0000u, __umul24(v, 100u), false);  // < 10^8
        };
        const unsigned long long value = (unsigned long long)eight(r0, r1) * 100000000ull + eight(r2, r3);
        if (fast_number) {
            *type = 'l';
            *raw = sgn ? (~value + 1) : value;
            code = 0;
            slow_number = false;
        } else if (is_num && nd >= 16u - sgn) {
            // 16 .. 18 digits (ids, timestamps in nanoseconds: the window shows no terminator): the next 16 bytes of the document,
            // once, for the lanes that need them -- the first 16 digits are the whole field as it stands, one or two more and the
            // terminator come from the second window.  (19 digits may leave the long range: the scanner of sj_number.h decides.)
            const CW16 w2 = *reinterpret_cast<const CW16*>(buf + idx + 16);
            const uint32_t y3 = __builtin_amdgcn_alignbyte(w2.a, win.d, sgn), y4 = __builtin_amdgcn_alignbyte(w2.b, w2.a, sgn);
            const uint32_t u3 = y3 ^ 0x30303030u, u4 = y4 ^ 0x30303030u;
            const uint32_t rem = ffbl(nondigit(u4)) >> 3;  // digits 16, 17, ... in front of the first non-digit of the fifth dword
            const bool all16 = (nondigit(t0) | nondigit(t1) | nondigit(t2) | nondigit(u3)) == 0u;
            const bool term_ok = rem <= 2u && sjn_is_structural_or_ws((y4 >> (8u * rem)) & 0xFFu);
            if (all16 && term_ok && (x0 & 0xFFu) != '0') {
                const unsigned long long hi16 = (unsigned long long)eight(t0, t1) * 100000000ull + eight(t2, u3);
                const uint32_t d16 = u4 & 0xFFu, d17 = (u4 >> 8) & 0xFFu;
                const unsigned long long v = rem == 0u ? hi16 : rem == 1u ? hi16 * 10ull + d16 : hi16 * 100ull + (d16 * 10u + d17);
                *type = 'l';
                *raw = sgn ? (~v + 1) : v;
                code = 0;
                slow_number = false;
            }
        } else if (is_atom) {
            *type = c;
            code = atom_ok ? 0 : (is_t ? SJMI_E_INVALID_TRUE : is_n ? SJMI_E_INVALID_NULL : SJMI_E_INVALID_FALSE);
        }
    } else if (is_atom) {  // (rare) the document is one atom: it must also end where the document ends
        const uint32_t len = is_f ? 5u : 4u;
        const uint32_t behind = is_f ? (win.b >> 8) & 0xFFu : win.b & 0xFFu;
        const bool word_ok = win.a == (is_t ? CW_TRUE : is_n ? CW_NULL : CW_FALS) && (!is_f || (win.b & 0xFFu) == 'e');
        const bool ok = idx + len <= end && word_ok && (idx + len == end || sjn_is_structural_or_ws(behind));
        *type = c;
        code = ok ? 0 : (is_t ? SJMI_E_INVALID_TRUE : is_n ? SJMI_E_INVALID_NULL : SJMI_E_INVALID_FALSE);
    }
    if (slow_number) {
        CwBytes w = {buf, idx, (unsigned long long)win.a | ((unsigned long long)win.b << 32),
                     (unsigned long long)win.c | ((unsigned long long)win.d << 32), 0ull, 0ull, false};
        const uint32_t limit = root ? end : 0xFFFFFFFFu;  // the root number's padded copy (TapeBuilder.java:183-189)
        const SjNumber n = sj_scan_number([&](uint32_t q) -> uint32_t { return q < limit ? w.at(q) : 0x20u; }, idx);
        if (n.code) return n.code;
        if (n.floating) {
            *type = 'd';
            if (!sj_number_double_bits(n, raw)) return CW_SLOW_DOUBLE;  // DoubleParser's slow path (:205-330): decided behind the walk
        } else {
            if (sj_out_of_long_range(n.negative, n.digits, n.digit_count)) return SJMI_E_NUM_LONG_RANGE;
            *type = 'l';
            *raw = n.negative ? (~n.digits + 1) : n.digits;
        }
        return 0;
    }
    return code;
}

__device__ __forceinline__ unsigned long long tape_word(uint32_t type, unsigned long long payload) {
    return payload | ((unsigned long long)type << 56);
}
__device__ __forceinline__ int highest_bit_below(unsigned long long m, unsigned long long lt_mask) {
    const unsigned long long x = m & lt_mask;
    return x ? 63 - __builtin_clzll(x) : -1;
}

}  // namespace

// ---- one LARGE document over many waves ----------------------------------------------------------------------------------
// A wave needs, at the first structural it looks at, what the sequential walker would know there: depth, tape position,
// string offset, the open containers (their tape positions, comma counts, kinds), whether the root value has ended.  For
// chunks of 128 or 512 structurals these are obtained in a few launches:
//   k_chunk_summary  (parallel, a wave per chunk): depth profile of the chunk relative to its start -- net change, minimum,
//                    words, string bytes -- and its EXPORT: the containers it opens and leaves open (by relative level) and
//                    the commas it adds to the innermost container it leaves untouched;
//   k_group_summary / k_top_scan / k_group_replay: the scan of the summaries (applying one to a state is associative) that
//                    gives every chunk its entry state, two levels, groups of ~sqrt(chunks / 4) chunks;
//   k_coop_walk<true>(parallel, a wave per chunk): the walker proper, started from the chunk's entry state;
//   k_chunk_finish   (one wave): the document's first error by position, the root words.
// Relative levels live in the 64 lanes of the stack registers with a bias of 32; a chunk whose depth swings further, or a
// document deeper than the stack, raises the fall-back flag and the single-wave sweep takes the document.
// structurals per chunk and chunks per group are functions of the document's structural count n alone (every kernel
// computes them from index_offsets): short chunks while there are fewer chunks than SIMDs to put them on (a step costs a
// lone wave ~5 us, so a 128-structural chunk = 2 steps), longer ones after that; groups of ~sqrt(chunks / 4)
constexpr uint64_t CW_SMALL_N = 256u << 10;
constexpr uint64_t CW_SINGLE_N = 512;   // at most this many structurals (8 steps of ~6 us): the single-wave sweep is quicker than the chunk path's launches
__host__ __device__ inline uint32_t cw_chunk_of(uint64_t n) { return n <= CW_SMALL_N ? 128u : 512u; }
#ifndef SJMI_CW_GROUP_BIAS
#define SJMI_CW_GROUP_BIAS 4  // (groups of ~sqrt(chunks / 4): twitter.json 27 groups of 16 instead of 14 of 32 -- the two group kernels apply their summaries one after the other, the top scan its groups: 0.153 -> 0.147 ms)
#endif
__host__ __device__ inline uint32_t cw_group_of(uint64_t nchunks) {
    uint32_t g = 8;
    while ((uint64_t)g * g * SJMI_CW_GROUP_BIAS < nchunks) g <<= 1;
    return g;
}
constexpr int CW_BIAS = 32;
struct ChunkSum {            // what a chunk (or a group of chunks) does to the walker's state, relative to its start
    int32_t* delta;          // net depth change
    int32_t* min_after;      // minimum over the depths AFTER each structural
    uint32_t* words;         // tape words
    uint32_t* ssz;           // string-record bytes
    uint32_t* exp_tpos;      // [.][64] export: tape position (relative) of the open bracket left open at relative level lane - BIAS
    uint32_t* exp_cnt;       // [.][64] ... its comma count so far; lane BIAS + min - 1: commas added to the container below
    unsigned long long* exp_arr;  // [.] kinds (bit lane)
};
struct ChunkIn {             // the walker's state in front of a chunk (or group)
    uint32_t* H;
    uint32_t* T;
    unsigned long long* S;
    uint32_t* tpos;          // [.][64]
    uint32_t* cnt;           // [.][64]
    unsigned long long* arr;
    uint32_t* root_closed;
};
struct ChunkWs {
    ChunkSum sum, gsum;      // per chunk (k_chunk_summary), per group (k_group_summary)
    ChunkIn in, gin;         // per chunk (k_group_replay), per group (k_top_scan)
    // results per chunk (k_coop_walk<true>)
    uint32_t* err_pos;       // absolute structural position of the chunk's first error, 0xFFFFFFFF = none
    int32_t* err_code;
    // per document
    uint32_t* fallback;      // != 0: take the single-wave sweep
    uint32_t* fin;           // [0] final depth, [1] final tape position, [2] kind of the innermost open container (1 = array)
};

// the literals the walk could not decide (see CW_SLOW_DOUBLE): count, then per literal {address of the tape word that holds the
// lower candidate, position of the literal | limit of the readable bytes << 32}
struct SlowList {
    unsigned long long* count;
    unsigned long long* rec;
    unsigned long long cap;
};
constexpr unsigned long long CW_SLOW_CAP = 1ull << 16;

// One literal per workgroup, one thread of it at work (big-integer loops over two arrays in LDS): the rarest path there is.
__global__ void __launch_bounds__(64)
k_slow_doubles(const uint8_t* __restrict__ buf, SlowList sl) {
    __shared__ uint32_t wa[SJ_BIG_WORDS], wb[SJ_BIG_WORDS];
    unsigned long long n = *sl.count;
    if (n > sl.cap) n = sl.cap;
    if (threadIdx.x != 0) return;
    for (unsigned long long r = blockIdx.x; r < n; r += gridDim.x) {
        unsigned long long* const slot = reinterpret_cast<unsigned long long*>(sl.rec[2 * r]);
        const uint32_t p = (uint32_t)sl.rec[2 * r + 1], limit = (uint32_t)(sl.rec[2 * r + 1] >> 32);
        const unsigned long long cand = *slot;
        const bool neg = buf[p] == '-';
        const unsigned long long mag =
            sj_decide_double([&](uint32_t q) -> uint32_t { return q < limit ? (uint32_t)buf[q] : 0x20u; }, p + (neg ? 1u : 0u),
                             cand & ~(1ull << 63), wa, wb);
        *slot = mag | (neg ? 1ull << 63 : 0ull);
    }
}

// ONE document whose tape was written in place, executed by thread 0 of the walk's last launch: the listed literals, what
// k_tape_chunk_sums + k_tape_chunk_scan (walk.hip) would compute for it, and (sjmi_parse_document) the three stages' result
// records packed for one D2H
struct SingleFinish {
    SlowList slow;
    const uint32_t* tape_lens;
    const int32_t* doc_errors;
    uint64_t tape_capacity;
    unsigned long long* tape_offsets;
    WalkResult* res;
    const Stage1Result* s1;
    const UnescapeResult* u;
    SingleDocPack* pack;
};
__device__ void cw_single_finish(const uint8_t* __restrict__ buf, const SingleFinish& f, uint32_t tlen, int code, uint32_t* wa, uint32_t* wb) {
    unsigned long long n = *f.slow.count;
    if (n > f.slow.cap) n = f.slow.cap;
    for (unsigned long long r = 0; r < n; ++r) {
        unsigned long long* const slot = reinterpret_cast<unsigned long long*>(f.slow.rec[2 * r]);
        const uint32_t p = (uint32_t)f.slow.rec[2 * r + 1], limit = (uint32_t)(f.slow.rec[2 * r + 1] >> 32);
        const bool neg = buf[p] == '-';
        const unsigned long long mag =
            sj_decide_double([&](uint32_t q) -> uint32_t { return q < limit ? (uint32_t)buf[q] : 0x20u; }, p + (neg ? 1u : 0u),
                             *slot & ~(1ull << 63), wa, wb);
        *slot = mag | (neg ? 1ull << 63 : 0ull);
    }
    WalkResult w;
    w.tape_words = tlen;
    w.host_documents = code == SJMI_WALK_NEEDS_HOST ? 1 : 0;
    w.failed_documents = code > 0 ? 1 : 0;
    w.flags = f.res->flags | (tlen > f.tape_capacity ? 1u : 0u);
    w.reserved = 0;
    f.tape_offsets[0] = 0;
    f.tape_offsets[1] = tlen;
    *f.res = w;
    if (f.pack) {
        f.pack->s1 = *f.s1;
        f.pack->u = *f.u;
        f.pack->w = w;
        f.pack->to[0] = 0;
        f.pack->to[1] = tlen;
        f.pack->err = code;
        f.pack->fallback = 0;
    }
}
__global__ void __launch_bounds__(64)
k_single_finish(const uint8_t* __restrict__ buf, SingleFinish f) {
    __shared__ uint32_t wa[SJ_BIG_WORDS], wb[SJ_BIG_WORDS];
    if (threadIdx.x != 0) return;
    cw_single_finish(buf, f, f.tape_lens[0], f.doc_errors[0], wa, wb);
}

// k_coop_walk<false> as the EXACT walker behind the token walker (see k_tok_stream): which documents, and where their tapes go
struct ExactMode {
    const uint32_t* list = nullptr;      // [0] = how many, ids from [16]; nullptr: every document
    const DocMeta* metas = nullptr;      // != nullptr: document k's tape at metas[k].tape, room up to metas[k + 1].tape
    unsigned long long* tape = nullptr;  // *sel != 0 (or sel == nullptr): the final tape; else tape_alt (the scratch tape)
    unsigned long long* tape_alt = nullptr;
    const uint32_t* sel = nullptr;
    unsigned long long cap = 0;          // != 0 (one document written in place into the CALLER's tape): words of room there
};

// One wave per document (grid-stride over the documents).  Documents are delimited by index_offsets (n_docs + 1 entries)
// and doc_offsets; a single document is the batch of one.  Document k's tape is built in its slot of the scratch tape
// (2 words per structural + 2, walk.hip packs the tapes back to back afterwards); soff[] = the string pass's record offsets
// by string ordinal, doc_str_offsets[k] = ordinal of document k's first string, sb = the string buffer (only looked at when
// the pass reported a malformed escape: the failing record's header is FF FF FF <code>).
// (five waves per SIMD: 96 instead of 123 VGPRs and three spilled dwords, but the kernel is VALU-bound at 70 % issue
//  utilisation and the fifth wave fills bubbles: 4.91 -> 4.38 ms per million documents; six waves spill 26 dwords: 4.41)
template <bool CHUNKED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5)))
k_coop_walk(const uint8_t* __restrict__ buf, const unsigned long long* __restrict__ doc_offsets, uint64_t n_docs,
            const uint32_t* __restrict__ idx, const unsigned long long* __restrict__ index_offsets,
            const uint32_t* __restrict__ doc_status, const uint32_t* __restrict__ soff, const uint8_t* __restrict__ sb,
            const unsigned long long* __restrict__ doc_str_offsets, unsigned long long string_base, int max_depth,
            unsigned long long* __restrict__ scratch_tape, uint32_t* __restrict__ tape_lens, int32_t* __restrict__ doc_errors, const Stage1Result* __restrict__ dev_count,
            const UnescapeResult* __restrict__ dev_strings, WalkResult* res, uint32_t abl, ChunkWs cw, const uint32_t* run_only_if,
            unsigned long long* __restrict__ ovf, SlowList slow, ExactMode ex) {
    if (run_only_if && *run_only_if == 0) return;   // (the single-wave sweep behind a chunked launch: only on fall-back)
    if (CHUNKED && *cw.fallback != 0) return;
    if (!CHUNKED && ex.metas && ex.sel && *ex.sel == 0 && !ex.tape_alt) return;  // (optimistic pipeline only, plain pass rejected)
    // LIST MODE (the exact walker behind k_tok_stream): the documents are those the token walker listed; a document's tape goes
    // where its DocMeta says -- the final tape when *ex.sel != 0 (then this kernel also counts the failures), else the scratch tape
    const uint32_t* const list = CHUNKED ? nullptr : ex.list;
    const bool final_mode = !CHUNKED && ex.metas && ex.sel && *ex.sel != 0;  // (the tapes were laid out before the walk)
    if (!CHUNKED && ex.metas) scratch_tape = (!ex.sel || *ex.sel != 0) ? ex.tape : ex.tape_alt;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the per-wave stack of open containers lives in two VGPRs: LANE L holds level L (tape position of the opening word,
    // commas seen so far); a level is read with v_readlane and written with v_writelane -- no LDS round trip on the
    // step-to-step dependency chain
    uint32_t st_tpos = 0, st_cnt = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // (everything that indexes structurals is 32-bit: a launch has fewer than 2^32 of them -- its bytes are addressed by 32-bit
    //  indexes -- and a document of more than 2^31 - 256 is handed back below, so neither a position nor a tape offset wraps)
    const uint32_t nwaves = gridDim.x * 4u;
    // Levels CW_LEVELS and deeper (SimdJsonParser.java:7: the default maxDepth is 1024) live in global memory, one record per
    // level and wave: {commas so far | kind << 31, tape position of the opening word}.  Rare, so nothing about it is fast;
    // lane 0 writes and reads them (one thread, one address: program order), the value is broadcast.
    unsigned long long* const my_ovf = ovf ? ovf + ((uint64_t)blockIdx.x * 4 + wv) * CW_OVF_LEVELS : nullptr;
    const int level_cap = my_ovf ? CW_LEVELS + CW_OVF_LEVELS : CW_LEVELS;
    auto ovf_read = [&](int L) -> unsigned long long {
        unsigned long long v = 0;
        if (lane == 0) v = __hip_atomic_load(&my_ovf[L - CW_LEVELS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    auto ovf_write = [&](int L, uint32_t tpos_, uint32_t cnt_, bool arr_) {
        if (lane == 0)
            __hip_atomic_store(&my_ovf[L - CW_LEVELS], ((unsigned long long)tpos_ << 32) | (cnt_ & 0x7FFFFFFFu) | (arr_ ? 0x80000000u : 0u),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const bool upstream_failed = (dev_count && (dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL))) ||
                                 (dev_strings && (dev_strings->flags & 0xFu));
    // some string of the launch has a malformed escape (rare): then every string's record header is looked at
    const bool string_errors = dev_strings && dev_strings->first_error_inv != 0;
    unsigned long long n_host = 0, n_bad = 0;
    // A document costs three dependent round trips before its first step can run (its delimiters, then the positions of
    // its first structurals, then the bytes there): with ~3 steps per ~1 KB document that chain, not the steps, bounded
    // the kernel.  So the delimiters of the wave's NEXT document are requested when the current one starts, and the
    // positions of its first two steps when the current one ends.
    struct Meta {
        uint32_t from, to, dso;
        uint32_t doc_start, doc_end, st;
    };
    struct Head {
        uint32_t p_n, px_n, p_nn, px_nn;
    };
    auto load_meta = [&](uint32_t k) {
        Meta m;
        m.from = (uint32_t)index_offsets[k];
        m.to = (uint32_t)index_offsets[k + 1];
        m.dso = (uint32_t)doc_str_offsets[k];
        m.doc_start = (uint32_t)doc_offsets[k];
        m.doc_end = (uint32_t)doc_offsets[k + 1];
        m.st = doc_status ? doc_status[k] : 0u;
        return m;
    };
    // (wfrom, wto) = the structurals this wave walks: the whole document, or one chunk of it
    auto load_pos = [&](const Meta& m, uint32_t wfrom, uint32_t wto, uint32_t s, uint32_t* p, uint32_t* px) {
        const uint32_t i = wfrom + s * 64u + (uint32_t)lane;
        *p = i < wto ? idx[i] : m.doc_start;
        const uint32_t ix = wfrom + s * 64u + 64u;
        *px = ix < m.to ? idx[ix] : m.doc_start;
    };
    auto load_head = [&](const Meta& m, uint32_t wfrom, uint32_t wto) {
        Head h;
        load_pos(m, wfrom, wto, 0, &h.p_n, &h.px_n);
        h.p_nn = m.doc_start;
        h.px_nn = m.doc_start;
        if (wto - wfrom > 64) load_pos(m, wfrom, wto, 1, &h.p_nn, &h.px_nn);
        return h;
    };
    uint32_t k = blockIdx.x * 4u + (uint32_t)wv;
    Meta m = {0, 0, 0, 0, 0, 0}, m_next = m;
    Head hd = {0, 0, 0, 0};
    uint32_t n_items = (uint32_t)n_docs;
    if (list) n_items = list[0] < n_items ? list[0] : n_items;
    auto doc_id = [&](uint32_t kk) -> uint32_t { return list ? list[16 + kk] : kk; };
    uint32_t chunk = 0;
    if (CHUNKED) {  // the work items are the chunks of document 0
        m = load_meta(0);
        chunk = cw_chunk_of(m.to - m.from);
        n_items = (m.to - m.from + chunk - 1u) / chunk;
        if (k < n_items) {
            const uint32_t a = m.from + k * chunk, b = m.to - a > chunk ? a + chunk : m.to;
            hd = load_head(m, a, b);
        }
    } else if (k < n_items) {
        m = load_meta(doc_id(k));
        hd = load_head(m, m.from, m.to);
    }
    bool more = false;
    for (; k < n_items; k = more ? k + nwaves : n_items) {
        more = n_items - k > nwaves;  // (k + nwaves < n_items, without a sum that could wrap)
        if (!CHUNKED && more) m_next = load_meta(doc_id(k + nwaves));
        int code = 0;
        uint32_t tlen = 0, err_at = 0xFFFFFFFFu;
        const uint32_t st = m.st;
        const uint32_t from = m.from, to = m.to;
        const uint32_t wfrom = CHUNKED ? from + k * chunk : from;
        const uint32_t wto = CHUNKED ? (to - wfrom > chunk ? wfrom + chunk : to) : to;
        const uint32_t kdoc = CHUNKED ? 0u : doc_id(k);
        // SimdJsonParser.stage1 order: Utf8Validator.validate (:165-167), then StructuralIndexer.index (:297-302)
        if (upstream_failed) code = SJMI_E_CAPACITY;
        else if (st & SJMI_ST_UTF8) code = SJMI_E_UTF8;
        else if (st & SJMI_ST_UNCLOSED) code = SJMI_E_UNCLOSED_STRING;
        else if (st & SJMI_ST_UNESCAPED) code = SJMI_E_UNESCAPED_CHARS;
        else if (from == to) code = SJMI_E_NO_STRUCTURAL;  // JsonIterator.java:27-29
        else if (to - from > 0x7FFFFF00u || to > 0xFFFFFF00u) code = SJMI_WALK_NEEDS_HOST;  // (tape offsets are 32-bit here)
        if (code == 0) {
            const uint32_t doc_start = m.doc_start, doc_end = m.doc_end;
            const uint32_t n = to - from;
            // this document's slot (word 0 = root): two words per structural + 2 of the scratch tape, or what its DocMeta says
            unsigned long long t_off = 2ull * from + 2ull * kdoc, t_room = 2ull * n + 2ull;
            if (!CHUNKED && ex.metas) {
                t_off = ((unsigned long long)ex.metas[kdoc].tape_hi << 32) | ex.metas[kdoc].tape_lo;
                t_room = (((unsigned long long)ex.metas[kdoc + 1].tape_hi << 32) | ex.metas[kdoc + 1].tape_lo) - t_off;
            }
            if (ex.cap && t_room > ex.cap) t_room = ex.cap;  // (a tape that does not fit is reported, never overrun)
            unsigned long long* const T = scratch_tape + t_off;
            const uint32_t room = t_room > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t_room;
            // running state (wave-uniform)
            uint32_t H0 = 0;                 // open containers in front of the step
            uint32_t T0 = 1;                 // tape position of the step's first word (0 = the root word)
            uint32_t S0 = m.dso;             // ordinal of the first string at or behind the step (the record table soff[] is by ordinal)
            // a STRING word needs its record's offset, a gather by ordinal: requested in one step, stored in the next
            uint32_t pq_tpos = 0, pq_off = 0;
            bool pq_live = false;
            unsigned long long arr_mask = 0; // bit L: the open container of level L is an array
            uint32_t prev_cls = K_COMMA;     // class of the structural in front of the step (none at the start)
            bool prev_empty_open = false, prev_is_key = false, root_closed = false;
            uint32_t root_kind = 0, root_c = 0;
            if (CHUNKED) {  // start from the chunk's entry state (k_group_replay)
                H0 = cw.in.H[k];
                T0 = cw.in.T[k];
                S0 = (uint32_t)cw.in.S[k];
                st_tpos = cw.in.tpos[k * 64 + lane];
                st_cnt = cw.in.cnt[k * 64 + lane];
                arr_mask = cw.in.arr[k];
                root_closed = cw.in.root_closed[k] != 0;
                root_c = buf[idx[from]];
                const uint32_t rk = class_of(root_c);
                root_kind = rk <= K_OPEN_O ? 1u + rk : 0u;
                if (k > 0) {
                    // what the steps carry from their predecessor: its class, whether it was a key (a string directly
                    // behind an object's opening bracket or behind a comma inside an object; the container of a
                    // non-bracket structural in front of the chunk is the innermost open one)
                    const uint32_t c1 = class_of(buf[idx[wfrom - 1]]);
                    const uint32_t c2 = wfrom - 1 > from ? class_of(buf[idx[wfrom - 2]]) : K_COLON;
                    prev_cls = c1;
                    const bool in_obj = H0 >= 1 && !((arr_mask >> (H0 - 1)) & 1ull);
                    prev_is_key = c1 == K_QUOTE && (c2 == K_OPEN_O || (c2 == K_COMMA && in_obj));
                }
            }
            // positions (and sizes) are requested TWO steps ahead, the 16-byte windows they point at one step ahead: neither
            // round trip is on the step-to-step critical path
            const uint32_t nsteps = (wto - wfrom + 63u) / 64u;
            uint32_t p_n = hd.p_n, px_n = hd.px_n, p_nn = hd.p_nn, px_nn = hd.px_nn;
            CW16 win_n = *reinterpret_cast<const CW16*>(buf + p_n);
            uint32_t bx_n = buf[px_n];
            for (uint32_t s = 0; s < nsteps && code == 0; ++s) {
                const uint32_t p = p_n, c_extra = bx_n;
                const CW16 win = win_n;
                p_n = p_nn;
                px_n = px_nn;
                if (s + 1 < nsteps) {
                    win_n = *reinterpret_cast<const CW16*>(buf + p_n);
                    bx_n = buf[px_n];
                }
                if (s + 2 < nsteps) load_pos(m, wfrom, wto, s + 2, &p_nn, &px_nn);
                const uint32_t i = wfrom + s * 64u + (uint32_t)lane;
                const bool valid = i < wto;
                const unsigned long long vmask = __ballot(valid);
                if (root_closed) {  // JsonIterator.java:196-198: something follows the root value
                    code = SJMI_E_TRAILING_CONTENT;
                    err_at = wfrom + s * 64u;
                    break;
                }
                const uint32_t c = win.a & 0xFFu;
                const uint32_t cls = valid ? class_of(c) : K_COMMA;
                const bool is_open = valid && cls <= K_OPEN_O, is_close = valid && (cls == K_CLOSE_A || cls == K_CLOSE_O);
                // neighbours
                uint32_t cls_prev = (uint32_t)__shfl_up((int)cls, 1);
                if (lane == 0) cls_prev = prev_cls;
                uint32_t cls_next = (uint32_t)__shfl_down((int)cls, 1);
                const bool has_next = i + 1 < to;
                if (lane == 63) cls_next = class_of(c_extra);
                // (1) empty containers: an opening bracket directly followed by its closing bracket is ONE value (:205-208)
                const bool empty_open = is_open && has_next && cls_next == cls + 2;
                int eo_prev = __shfl_up((int)empty_open, 1);
                if (CHUNKED && s == 0 && k > 0) {  // was the structural in front of the chunk the opening half of an empty pair?
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cls);
                    prev_empty_open = prev_cls <= K_OPEN_O && c0 == prev_cls + 2;
                }
                if (lane == 0) eo_prev = prev_empty_open;
                const bool empty_close = valid && is_close && eo_prev && i > from;
                // (2) depth in front of every structural
                // ((2) and (3) share one ladder: opens <= 64, closes <= 64 and words <= 128 in the fields of one dword)
                const uint32_t up = is_open ? 1u : 0u, down = is_close ? 1u : 0u;
                const bool is_num = valid && cls == K_PRIM && (c == '-' || c - '0' <= 9u);
                const uint32_t words = !valid || cls == K_COMMA || cls == K_COLON ? 0u : (is_num ? 2u : 1u);
                const uint32_t scan3 = cw_incl_scan(up | (down << 8) | (words << 16));
                const uint32_t iu = scan3 & 0xFFu, id = (scan3 >> 8) & 0xFFu, iw = scan3 >> 16;
                const int h = (int)H0 + (int)(iu - up) - (int)(id - down);  // may go negative behind the root's end: never used there
                // (3) tape positions and string offsets
                const uint32_t tpos = T0 + iw - words;
                const bool is_str = valid && cls == K_QUOTE;
                const unsigned long long qm = __ballot(is_str);
                const uint32_t sord = S0 + (uint32_t)__popcll(qm & lt_mask);  // this string's ordinal
                const uint32_t rec_off = is_str ? soff[sord] : 0u;                                    // (used one step later)
                // (4) the container of every structural: level loop over the depths present in this step
                const int plevel = h - 1;  // level of the container this structural sits in
                int hmin = cw_wave_minmax<false>(valid ? plevel : 0x7FFF), hmax = cw_wave_minmax<true>(valid ? (is_open ? h : plevel) : -0x7FFF);
                if (hmin < 0) hmin = 0;  // (level -1 = in front of / behind the root: no container)
                // Per level only what needs the level's ballots: the lane of the nearest opener in front of every structural of
                // this level, the commas of the level in front of every lane (openers: of the level they open), and the
                // stack entry as it stood in front of the step.  Everything that needs a cross-lane read of the opener's
                // values happens ONCE behind the loop (the permutes were the loop's dependency chain).
                int par_lane = -1;
                uint32_t kc = 0, sk_t = 0, sk_c = 0;
                bool sk_arr = false;
                const int key = valid ? (is_open ? h : plevel) : -1;
                // (deeper than the device stack: the non-empty open at depth 63 is handed back below, at a lower position
                //  than anything that would need a level beyond the stack)
                if (hmax >= level_cap) hmax = level_cap - 1;
                if (abl & 2u) hmax = hmin - 1;
                for (int L = hmin; L <= hmax; ++L) {
                    const unsigned long long O = __ballot(is_open && h == L);                 // opens of level L
                    const unsigned long long C = __ballot(valid && cls == K_COMMA && plevel == L);  // commas directly inside level L
                    const unsigned long long Z = __ballot(is_close && plevel == L);            // closes of level-L containers
                    uint32_t sk_tpos, sk_cnt;
                    bool lv_arr;
                    if (L < CW_LEVELS) {
                        sk_tpos = (uint32_t)__builtin_amdgcn_readlane((int)st_tpos, L);
                        sk_cnt = (uint32_t)__builtin_amdgcn_readlane((int)st_cnt, L);
                        lv_arr = ((arr_mask >> L) & 1ull) != 0;
                    } else {  // (wave-uniform) a level beyond the stack registers
                        const unsigned long long v = ovf_read(L);
                        sk_tpos = (uint32_t)(v >> 32);
                        sk_cnt = (uint32_t)v & 0x7FFFFFFFu;
                        lv_arr = ((uint32_t)v >> 31) != 0;
                    }
                    if (key == L) kc = (uint32_t)__popcll(C & lt_mask);
                    if (valid && plevel == L) {
                        par_lane = highest_bit_below(O, lt_mask);
                        sk_t = sk_tpos;
                        sk_c = sk_cnt;
                        sk_arr = lv_arr;
                    }
                    // stack update for the next steps (wave-uniform)
                    if (O) {
                        const int al = 63 - __builtin_clzll(O);  // the last open of this level in the step
                        const unsigned long long above = al == 63 ? 0ull : ~((2ull << al) - 1ull);
                        if (!(Z & above)) {  // still open at the end of the step
                            const uint32_t tp = (uint32_t)__builtin_amdgcn_readlane((int)tpos, al);
                            const uint32_t kc_ = (uint32_t)__builtin_amdgcn_readlane((int)cls, al);
                            if (L < CW_LEVELS) {
                                st_tpos = lane == L ? tp : st_tpos;
                                st_cnt = lane == L ? (uint32_t)__popcll(C & above) : st_cnt;
                                arr_mask = kc_ == K_OPEN_A ? (arr_mask | (1ull << L)) : (arr_mask & ~(1ull << L));
                            } else {
                                ovf_write(L, tp, (uint32_t)__popcll(C & above), kc_ == K_OPEN_A);
                            }
                        }
                    } else if (!Z && C) {
                        if (L < CW_LEVELS) st_cnt = lane == L ? sk_cnt + (uint32_t)__popcll(C) : st_cnt;
                        else ovf_write(L, sk_tpos, sk_cnt + (uint32_t)__popcll(C), lv_arr);
                    }
                }
                const bool par_in_wave = par_lane >= 0;
                const int pl = par_in_wave ? par_lane : 0;
                const uint32_t a_tpos = (uint32_t)__shfl((int)tpos, pl), a_cls = (uint32_t)__shfl((int)cls, pl), a_kc = (uint32_t)__shfl((int)kc, pl);
                const bool par_is_array = par_in_wave ? a_cls == K_OPEN_A : sk_arr;
                const uint32_t par_tpos = par_in_wave ? a_tpos : sk_t;
                const uint32_t par_cnt = par_in_wave ? kc - a_kc : sk_c + kc;  // commas of my container in front of me
                // (5) roles and their local predicates (JsonIterator.java:68-193)
                const bool prev_open_nonempty = (cls_prev <= K_OPEN_O) && !eo_prev;
                const bool first_key = prev_open_nonempty && cls_prev == K_OPEN_O;
                const bool key_after_comma = cls_prev == K_COMMA && !par_is_array && i > from;
                const bool is_key = valid && (first_key || key_after_comma) && cls == K_QUOTE && i > from;
                int ik_prev = __shfl_up((int)is_key, 1);
                if (lane == 0) ik_prev = prev_is_key;
                int err = 0;
                uint32_t ptype = 0;
                unsigned long long praw = 0;
                bool want_value = false;
                if (valid && !empty_close) {
                    if (i == from) {
                        want_value = true;  // (the root bracket's "is the last structural my closing bracket" test: below)
                    } else if (prev_open_nonempty) {
                        if (cls_prev == K_OPEN_A) want_value = true;
                        else if (cls != K_QUOTE) err = SJMI_E_OBJECT_NO_KEY;  // :75-77
                    } else if (cls_prev == K_COMMA) {
                        if (par_is_array) want_value = true;
                        else if (cls != K_QUOTE) err = SJMI_E_KEY_MISSING;    // :121-123
                    } else if (cls_prev == K_COLON) {
                        want_value = true;
                    } else if (ik_prev) {
                        if (cls != K_COLON) err = SJMI_E_MISSING_COLON;       // :84-86
                    } else {  // the predecessor ended a value
                        const bool ok = cls == K_COMMA || cls == (par_is_array ? K_CLOSE_A : K_CLOSE_O);
                        if (!ok) err = par_is_array ? SJMI_E_NO_COMMA_ARRAY : SJMI_E_NO_COMMA_OBJECT;  // :131,:189
                    }
                    if (!err && cls == K_QUOTE && (is_key || want_value)) {
                        // a string the reference's StringParser would have thrown on: record header FF FF FF <code>
                        if (string_errors) {
                            const uint8_t* h = sb + rec_off;
                            if (h[0] == 0xFF && h[1] == 0xFF && h[2] == 0xFF) err = (int)h[3];
                        }
                    } else if (!err && want_value) {
                        if (cls <= K_OPEN_O) {
                            if (!empty_open) {
                                if (h + 1 >= max_depth) err = SJMI_E_DEPTH;                  // JsonIterator.java:69-70
                                else if (h + 1 >= level_cap) err = SJMI_WALK_NEEDS_HOST;     // deeper than the device stack
                            }
                        } else if (cls != K_QUOTE) {
                            if (!(abl & 1u)) err = cw_primitive(buf, win, p, i == from, doc_end, &ptype, &praw);
                            else ptype = 'n';
                            if (err == CW_SLOW_DOUBLE) {  // (rare) listed for k_slow_doubles, which overwrites the candidate
                                err = 0;
                                const unsigned long long slot = atomicAdd(slow.count, 1ull);
                                if (slot < slow.cap && tpos + 1 < room) {
                                    slow.rec[2 * slot] = reinterpret_cast<unsigned long long>(T + tpos + 1);
                                    slow.rec[2 * slot + 1] = (unsigned long long)p | ((unsigned long long)(i == from ? doc_end : 0xFFFFFFFFu) << 32);
                                } else {
                                    err = SJMI_WALK_NEEDS_HOST;  // (more than 65,536 such literals in one launch)
                                }
                            }
                        }
                    }
                }
                // (6) where the root value ends; the first error by position
                if (!CHUNKED && s == 0) {
                    root_kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cls <= K_OPEN_O ? 1u + cls : 0u));
                    root_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                }
                const bool closes_root = (is_close && h == 1 && root_kind != 0) || (valid && i == from && root_kind == 0);
                const unsigned long long rc = __ballot(closes_root);
                const int rc_lane = rc ? __builtin_ctzll(rc) : 64;
                if (lane == rc_lane + 1 && valid) err = SJMI_E_TRAILING_CONTENT;
                const unsigned long long em = __ballot(err != 0 && lane <= rc_lane + 1);
                if (em) {
                    code = __builtin_amdgcn_readlane(err, __builtin_ctzll(em));
                    err_at = wfrom + s * 64u + (uint32_t)__builtin_ctzll(em);
                    break;
                }
                if (rc) root_closed = true;
                // (7) the tape words of this step
                const bool live = valid && lane <= rc_lane && !(abl & 4u);
                if (pq_live && pq_tpos < room) T[pq_tpos] = tape_word('"', string_base + pq_off);  // the previous step's strings
                pq_live = live && cls == K_QUOTE;
                pq_tpos = tpos;
                pq_off = rec_off;
                if (live) {
                    if (cls == K_QUOTE) {
                    } else if (cls == K_PRIM) {
                        if (tpos < room) T[tpos] = tape_word(ptype, 0);
                        if (is_num && tpos + 1 < room) T[tpos + 1] = praw;
                    } else if (empty_open) {
                        if (tpos < room) T[tpos] = tape_word(c, tpos + 2);          // TapeBuilder.java:205-208
                    } else if (empty_close) {
                        if (tpos < room) T[tpos] = tape_word(c, tpos);              // (= position of the opening word + 1)
                    } else if (is_close) {
                        uint32_t cnt = par_cnt + 1u;
                        if (cnt > 0xFFFFFFu) cnt = 0xFFFFFFu;
                        if (tpos < room) T[tpos] = tape_word(c, par_tpos);                                                    // :197-203
                        if (par_tpos < room) T[par_tpos] = tape_word(c - 2, (unsigned long long)(tpos + 1) | ((unsigned long long)cnt << 32));
                    }
                }
                // (8) carries
                const uint32_t live_words = (uint32_t)__builtin_amdgcn_readlane((int)iw, rc_lane < 64 ? rc_lane : 63);
                const uint32_t tot3 = cw_last(scan3);
                H0 = (uint32_t)((int)H0 + (int)(tot3 & 0xFFu) - (int)((tot3 >> 8) & 0xFFu));
                T0 += rc_lane < 64 ? live_words : tot3 >> 16;
                S0 += (uint32_t)__popcll(qm);
                const int lastv = 63 - __builtin_clzll(vmask);
                prev_cls = (uint32_t)__builtin_amdgcn_readlane((int)cls, lastv);
                prev_empty_open = __builtin_amdgcn_readlane((int)empty_open, lastv) != 0;
                prev_is_key = __builtin_amdgcn_readlane((int)is_key, lastv) != 0;
            }
            if (pq_live && pq_tpos < room) T[pq_tpos] = tape_word('"', string_base + pq_off);  // the last step's strings
            if (CHUNKED) {  // the document's verdict is assembled by k_chunk_finish from the chunks' first errors
                if (lane == 0) {
                    cw.err_pos[k] = code ? err_at : 0xFFFFFFFFu;
                    cw.err_code[k] = code;
                }
                if (more) {
                    const uint32_t a = from + (k + nwaves) * chunk, b = to - a > chunk ? a + chunk : to;
                    hd = load_head(m, a, b);
                }
                continue;
            }
            // JsonIterator.java:39-41,:51-53: a root bracket whose closing bracket is not the document's LAST structural fails
            // before anything else is looked at (position 0 is the lowest there is).  The last structural's class is known
            // for free when the sweep reached the end; only a document that failed earlier has to go and look.
            if (root_kind != 0) {
                if (code != 0) {                      // failed (or handed back) on the way: go and look
                    if ((uint32_t)buf[idx[to - 1]] != root_c + 2) code = root_kind == 2 ? SJMI_E_UNCLOSED_OBJECT : SJMI_E_UNCLOSED_ARRAY;
                } else if (!root_closed) {            // swept to the end: prev_cls is the last structural's class
                    if (prev_cls != root_kind + 1) code = root_kind == 2 ? SJMI_E_UNCLOSED_OBJECT : SJMI_E_UNCLOSED_ARRAY;
                }                                     // (closed exactly at the end: the last structural IS the closing bracket)
            }
            if (code == 0 && !root_closed) {
                // the walker reads on past the last structural: BitIndexes' sentinel = the document's first byte, an opening
                // bracket where a separator is due (BitIndexes.java:82-96, JsonIterator.java:131,:189)
                const int top = (int)H0 - 1;
                bool top_arr = top >= 0 && top < CW_LEVELS && ((arr_mask >> top) & 1ull);
                if (top >= CW_LEVELS && top < level_cap) top_arr = ((uint32_t)ovf_read(top) >> 31) != 0;
                code = top_arr ? SJMI_E_NO_COMMA_ARRAY : SJMI_E_NO_COMMA_OBJECT;
            }
            if (code == 0) {
                tlen = T0 + 1;  // + the closing root word
                if (lane == 0) {
                    if (tlen <= room) {
                        T[T0] = tape_word('r', 0);      // visitDocumentEnd, TapeBuilder.java:45-48
                        T[0] = tape_word('r', tlen);
                    }
                }
                if (tlen > room && !ex.cap) {  // (cannot happen: two words per structural + 2; with ex.cap: the caller's tape is too small,
                    tlen = 0;                  //  reported through the length itself)
                    code = SJMI_E_INTERNAL;
                }
            }
        }
        if (CHUNKED) {  // (a document that failed stage 1 or has no structurals: decided by k_chunk_finish as well)
            if (lane == 0) {
                cw.err_pos[k] = code ? 0u : 0xFFFFFFFFu;
                cw.err_code[k] = code;
            }
            continue;
        }
        if (lane == 0) {
            if (tape_lens) tape_lens[kdoc] = tlen;
            doc_errors[kdoc] = code;
        }
        n_host += code == SJMI_WALK_NEEDS_HOST;
        n_bad += code > 0;
        m = m_next;
        if (more) hd = load_head(m, m.from, m.to);
    }
    // (host / failed documents: counted by the packing kernels from doc_errors -- except when the tapes were laid out before
    //  the walk, where nothing runs behind this kernel and only the documents of its list can have failed)
    if (final_mode && lane == 0) {
        if (n_host) atomicAdd(&res->host_documents, n_host);
        if (n_bad) atomicAdd(&res->failed_documents, n_bad);
    }
}

// ---- the BATCH walker: tokens, not structurals ---------------------------------------------------------------------------
// k_coop_walk above gives every structural a lane, and nearly half of a document's structurals are ',' and ':' -- they make no
// tape word, yet they occupy lanes of every scan, ballot and store of a step.  The batch walker (k_tok_stream below) walks
// TOKENS: the structurals are ingested 64 at a time (ONE byte each: the first byte says what a structural is), the separators
// are folded into a two-bit "what stands in front of me" field of the token behind them, and the tokens -- position, kind, that
// field -- are compacted into a per-wave ring in LDS; a TOKEN STEP then runs the scans of the cooperative walker over 64 tokens
// = ~116 structurals of a typical record.  Grammar in token form (JsonIterator.java:68-193, the same predicates, re-keyed):
//     first child of '['          no separator, a value            first child of '{'     no separator, a string (key)
//     behind a key                ':' and a value                  behind a value         ',' + value (array) / ',' + key (object)
//                                                                                          or no separator + the container's own close
//     two separators in a row, a separator in front of the root or behind the last token: never valid.
// The kernel is OPTIMISTIC: it builds the tape of a document that is well formed, and hands every other document -- any failing
// predicate, a stage-1 status, a root that is not a container, a malformed escape, a literal for k_slow_doubles, nesting beyond
// the 64 levels of the register stack -- to k_coop_walk (list mode), which walks it again and produces the reference's exact
// error code (or its tape).  A well-formed batch never gets there.
// (a ballot of a CONDITION: HIP's __ballot(int) compares its argument with zero in the VALU even when the condition already is
//  a lane mask in SGPRs -- v_cndmask + v_cmp per ballot; the builtin lets the compiler keep the mask)
__device__ __forceinline__ unsigned long long cw_ballot(bool c) { return __builtin_amdgcn_ballot_w64(c); }
struct TokArgs {
    const uint8_t* buf;
    const uint32_t* idx;
    const DocMeta* metas;
    uint32_t n_docs;
    int max_depth;
    const uint32_t* soff;
    const uint8_t* sb;
    unsigned long long string_base;
    unsigned long long* tape;       // *sel != 0 (or sel == nullptr): the tapes' final place; else tape_alt (the scratch tape)
    unsigned long long* tape_alt;
    const uint32_t* sel;
    int32_t* doc_errors;
    uint32_t* tape_lens;            // scratch mode only (walk.hip packs by them); may be nullptr
    uint32_t* list;                 // [0] = number of documents for the exact walker, their ids from [16]
    const Stage1Result* dev_count;
    const UnescapeResult* dev_strings;
    uint32_t run_docs;              // k_tok_stream: documents per run (1 .. TS_RUN), chosen per launch (tok_walk_launch)
};
// (token kinds, the first-byte table and the grammar table: sj_tokens.h, shared with the CPU test)
struct __attribute__((aligned(8))) TokLevels {
    unsigned long long open[64];  // per level: the lanes of this step's opening brackets with that depth in front of them
    uint2 stk[64];                // the open containers of the wave's document by level: .x = tape position of the opening word | is-array << 31, .y = commas so far
    uint32_t cnt[64];             // per token of the current step: an opening bracket's commas so far | closed-in-this-step << 31
};
constexpr int32_t CW_NEEDS_EXACT = -100;  // (internal, overwritten by the exact walker)
// Primitives are parsed DENSELY: a token step only queues its atoms and numbers (window, position, where the words go, which
// document), and whenever 64 are waiting they are parsed with every lane at work -- a quarter of a record's tokens are
// primitives, so parsing them inside the step ran the ~170 instructions of cw_primitive on 16 live lanes (and made every
// structural carry a 16-byte window through the ring: now the ingest loads ONE byte per structural).  The queue lives
// across documents; a literal that turns out malformed (or needs k_slow_doubles) sends ITS document to the exact walker,
// whatever the token walker thought of it.
struct __attribute__((aligned(16))) PrimQueue {
    uint4 e[128];  // .x = position (its 16-byte window is loaded when the queue is flushed: 64 dense loads), .y = document, .zw = where the words go
};
// LANE MASKS AND TABLES.  The round-4 walker was bound by VALU issue (91 % of the SIMD cycles, 887 instructions per document), the
// round-5 one -- with its predicates as 64-bit lane masks in SGPRs -- by the scalar unit (profiles/r5/README.md).  The idioms:
//   * a predicate that gates a store, an LDS operation or a select is a lane mask from ONE v_cmp; it comes back to the lanes
//     as the condition itself (inverse ballot: no instruction); "how many lanes below me" is v_mbcnt;
//   * "my predecessor / successor" is one DPP move of the token word (wave_shr / wave_shl), the carry of the neighbouring step
//     in the lane that has no neighbour;
//   * everything a token is by itself (kind, its fields of the scan, its byte) is ONE LDS read of a 256-entry table in the ingest;
//   * the token grammar (JsonIterator.java:68-193) is ONE LDS read of a 2048-entry table: token, what stands in front of it,
//     the previous token, is my container an array.
__device__ __forceinline__ bool cw_lanes(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ uint32_t cw_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ unsigned long long cw_first(uint32_t n) { return ~0ull >> (64u - n); }  // lanes [0, n), 1 <= n <= 64
__device__ __forceinline__ uint32_t cw_bit(unsigned long long m, uint32_t i) { return (uint32_t)(m >> i) & 1u; }
// ---- the batch walker in STREAM form (round 6) ------------------------------------------------------------------------------
// Round 5's k_tok_walk walked one document per wave at a time, and of its ~500 scalar + ~420 vector instructions per ~1 KB document
// more than half of the scalar ones were per-DOCUMENT scaffolding (profiles/r5/README.md: four ingest trips, a prefetch a document
// ahead, prologue, epilogue, a half-empty second token step); it was bound by scalar issue and spilled 62 SGPRs.  It is gone
// (profiles/r6/README.md has the A/B: 1.30 -> 1.12-1.20 ms per 1 M documents, 430 + 374 instructions).  k_tok_stream walks a RUN of TS_RUN
// consecutive documents as ONE token stream: the structurals of a run are contiguous in stage 1's index array, so the ingest never
// stops at a document, and a token step of 64 tokens may hold the end of one document, a whole small one and the beginning of a
// third.  Nothing in a step is per document:
//   * a token carries the run-local number of its document (8 bits of the token word); a token whose predecessor belongs to
//     another document is a document's first (DS), has "nothing" in front of it, and its predecessor has no successor;
//   * depth, string ordinal and tape position are ONE three-field scan over the step, continued over the run by three uniform
//     counters (A_d, A_q, A_w); a document's first token parks the scan values it sees in the document's LDS record, and every
//     token of the document subtracts them -- document-relative depth / ordinal / position whatever step the document began in;
//   * containers: every opening bracket of the step puts its lane into the LDS word of its level, a token finds its container as
//     the last such lane in front of it in the word of ITS level (two v_ffbh), else on the per-wave stack in LDS; commas and
//     "closed in this step" are one LDS atomic add; the grammar of JsonIterator.java:68-193 is one LDS read of a 2048-entry table.
//     A document's containers all lie behind its first token, so the last opening bracket of my level in front of me is mine;
//   * the only token that may stand at depth 0 is a document's first, and it must be an opening bracket: that is "nothing
//     follows the root's end", "the root is a container" and "a closing bracket too many" in one compare; a document must be
//     back at depth 0 where the next one starts (the next document's first token looks at it);
//   * a failing document is a bit in a mask per run: its tokens store nothing from the step on in which it fails, it is listed
//     for the exact walker at the run's end, and it cannot touch its neighbours (its tokens resolve against its own brackets
//     or, at depth < 1, against nothing: they are kept out of the LDS operations).
// The wave-uniform state is small on purpose (the first stream walker, round 5, died of 89-128 spilled SGPRs): three scan
// counters, the previous token, ring head / tail, the ingest's chunk and separator carries, the failed mask.
#ifndef SJMI_TS_RUN
#define SJMI_TS_RUN 16   // (measured: 8 -> 1.183 ms, 16 -> 1.126, 32 -> 1.147 per 1 M documents)
#endif
constexpr uint32_t TS_RUN = SJMI_TS_RUN;    // documents per run (their records live in LDS; a run is ~1,800 tokens = ~28 token steps)
constexpr uint32_t TS_RING = 128u;  // tokens between the ingest and the token steps
struct __attribute__((aligned(8))) TsRing {
    uint2 e[TS_RING];  // .x = position, .y = the token | run-local document << 24
};
struct __attribute__((aligned(16))) TsRun {
    uint4 rec[TS_RUN];       // per (non-empty) document of the run: tape offset lo, hi, room in words, its number in the batch
    uint4 dyn[TS_RUN];       // ... set by its first token: .x = depth base, .y = string base - its first ordinal, .z = word base - 1
    uint32_t dso[TS_RUN];    // ... the ordinal of its first string
    uint32_t from[TS_RUN + 1];  // ... its first structural (index into idx[]); [n] = the run's end
};
#ifndef SJMI_TS_WAVES
#define SJMI_TS_WAVES 6
#endif
#ifndef SJMI_TS_NT_IDX
#define SJMI_TS_NT_IDX 1  // the positions as streaming loads (the index array is read once, two whole lines an instruction): 1227 / 1231 -> 1212 / 1219 us
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SJMI_TS_WAVES, SJMI_TS_WAVES)))
k_tok_stream(TokArgs a_by_value) {
#if defined(__HIP_DEVICE_COMPILE__)
    const TokArgs& a = *(const TokArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // (the arguments are read from the kernarg segment where they are needed: as SGPRs held for the whole kernel they would spill -- a v_writelane / v_readlane pair each)
    (void)a_by_value;
#else
    const TokArgs& a = a_by_value;
#endif
    if (a.sel && *a.sel == 0 && !a.tape_alt) return;  // (only the optimistic pipeline was queued and its plain pass was rejected)
    __shared__ TsRing rings[4];
    __shared__ PrimQueue queues[4];
    __shared__ TokLevels levels[4];
    __shared__ TsRun runs[4];
    __shared__ uint32_t first_byte_token[256];
    __shared__ uint8_t grammar[TOK_GRAMMAR_ENTRIES];
    first_byte_token[threadIdx.x] = tok_of_first_byte(threadIdx.x);
    for (uint32_t i = threadIdx.x; i < TOK_GRAMMAR_ENTRIES; i += 256u) grammar[i] = (uint8_t)tok_grammar(i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TsRing& ring = rings[wv];
    PrimQueue& pq = queues[wv];
    TokLevels& lv = levels[wv];
    TsRun& run = runs[wv];
    uint32_t qhead = 0, qtail = 0;
    auto send_to_exact = [&](uint32_t doc) {
        if (atomicExch(&a.doc_errors[doc], CW_NEEDS_EXACT) != CW_NEEDS_EXACT) {
            const uint32_t slot = atomicAdd(&a.list[0], 1u);
            a.list[16 + slot] = doc;
        }
    };
    auto flush_primitives = [&](uint32_t nq) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool live = (uint32_t)lane < nq;
        const uint32_t e = (qhead + (uint32_t)lane) & 127u;
        const uint4 q = pq.e[e];
        const uint32_t p = live ? q.x : 0u, doc = q.y;
        unsigned long long* const dst = reinterpret_cast<unsigned long long*>(((unsigned long long)q.w << 32) | q.z);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const CW16 win = *reinterpret_cast<const CW16*>(a.buf + p);
        if (live) {
            uint32_t ptype = 0;
            unsigned long long praw = 0;
            if (cw_primitive(a.buf, win, p, false, 0u, &ptype, &praw) == 0) {
                dst[0] = tape_word(ptype, 0);
                if (ptype == 'l' || ptype == 'd') dst[1] = praw;
            } else {
                send_to_exact(doc);
            }
        }
        qhead += nq;
    };
    const unsigned long long lane_bit = 1ull << lane;
    const uint32_t below_lo = (uint32_t)(lane_bit - 1ull), below_hi = (uint32_t)((lane_bit - 1ull) >> 32);
    const uint32_t nwaves = gridDim.x * 4u;
    unsigned long long* const tape = (a.sel && *a.sel == 0) ? a.tape_alt : a.tape;
    const bool upstream_failed = (a.dev_count && (a.dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL))) ||
                                 (a.dev_strings && (a.dev_strings->flags & 0xFu));
    const bool string_errors = a.dev_strings && a.dev_strings->first_error_inv != 0;
    const int depth_limit = (a.max_depth < CW_LEVELS ? a.max_depth : CW_LEVELS) - 1;
    const uint32_t RUN = a.run_docs;  // (<= TS_RUN, the size of the per-run tables)
    const uint32_t n_runs = (a.n_docs + RUN - 1u) / RUN;
    for (uint32_t r = blockIdx.x * 4u + (uint32_t)wv; r < n_runs; r += nwaves) {
        // ================= the run's documents: lane = document =================
        const uint32_t k0 = r * RUN;
        const uint32_t nd = a.n_docs - k0 < RUN ? a.n_docs - k0 : RUN;
        DocMeta m = {};
        if ((uint32_t)lane <= nd) m = a.metas[k0 + (uint32_t)lane];  // (n_docs + 1 records: the last one carries the tapes' end)
        const bool isdoc = (uint32_t)lane < nd;
        const bool nonempty = isdoc && m.to != m.from;
        // (more than 2^30 structurals: tape positions keep a flag bit, the comma counters too -- such a document, one that failed
        //  stage 1 and any document behind a failed launch goes to the exact walker; so does a document without structurals)
        const bool unwalkable = nonempty && (upstream_failed || m.st != 0 || m.to - m.from >= (1u << 30) || m.to > 0xFFFFFF00u);
        const unsigned long long NE = cw_ballot(nonempty), UW = cw_ballot(unwalkable);
        if (isdoc && (!nonempty || UW != 0)) {  // (an unwalkable document in the run: rare enough to send the whole run along)
            if (a.tape_lens) a.tape_lens[k0 + (uint32_t)lane] = 0u;
            send_to_exact(k0 + (uint32_t)lane);
        }
        if (UW != 0 || NE == 0) continue;
        const uint32_t rn = (uint32_t)__popcll(NE);
        {
            const uint32_t nx_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m.tape_lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
            const uint32_t nx_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m.tape_hi, 0x130, 0xf, 0xf, false);
            const unsigned long long t_off = ((unsigned long long)m.tape_hi << 32) | m.tape_lo;
            const unsigned long long room64 = (((unsigned long long)nx_hi << 32) | nx_lo) - t_off;
            const uint32_t room = room64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)room64;
            const uint32_t slot = cw_below(NE);
            if (nonempty) {
                run.rec[slot] = make_uint4(m.tape_lo, m.tape_hi, room, k0 + (uint32_t)lane);
                run.dso[slot] = m.dso;
                run.from[slot] = m.from;
            }
        }
        const uint32_t I0 = (uint32_t)__builtin_amdgcn_readlane((int)m.from, __builtin_ctzll(NE));
        const uint32_t I1 = (uint32_t)__builtin_amdgcn_readlane((int)m.to, 63 - __builtin_clzll(NE));
        if (lane == 0) run.from[rn] = I1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ================= the stream =================
        const uint32_t n = I1 - I0, nchunks = (n + 63u) / 64u;
        uint32_t c = 0, head = 0, tail = 0;
        unsigned long long SEPp = 0, COLp = 0;
        uint32_t jc = 0, jn = 1;                       // the document of the chunk's first structural; the next one to start
        uint32_t nxt_from = run.from[1] - I0;          // ... and where (relative to I0)
        unsigned long long failed = 0;                 // bit j: document j of the run goes to the exact walker
        int A_d = 0;                                   // the scan's running totals in front of the step: depth field - tokens,
        uint32_t A_q = 0, A_w = 0;                     //   strings, tape words
        uint32_t c_token = (uint32_t)TK_NONE | 0xFF000000u, c_eo = 0;  // the previous step's last token (no document yet)
        int c_dbase = 0;
        bool seen = false;
        unsigned long long PQ = 0;
        unsigned long long* pq_addr = nullptr;
        uint32_t pq_off = 0;
        auto fail_lanes = [&](unsigned long long M, uint32_t dj_v) {  // (rare) the documents of the lanes in M
            for (; M; M &= M - 1) failed |= 1ull << ((uint32_t)__builtin_amdgcn_readlane((int)dj_v, __builtin_ctzll(M)) & 63u);
        };
        // positions are requested two chunks ahead, first bytes one chunk ahead
        auto pos_of = [&](uint32_t cc) -> uint32_t {
            const uint32_t i = cc * 64u + (uint32_t)lane;
#if SJMI_TS_NT_IDX
            return __builtin_nontemporal_load(&a.idx[I0 + (i < n ? i : n - 1u)]);
#else
            return a.idx[I0 + (i < n ? i : n - 1u)];
#endif
        };
        uint32_t Pa = pos_of(0), Pb = pos_of(1);
        uint32_t Ba = a.buf[Pa];
        auto ingest = [&]() {
            const uint32_t i0 = c * 64u;
            const uint32_t nvl = n - i0 < 64u ? n - i0 : 64u;
            const unsigned long long VL = cw_first(nvl);
            const uint32_t p = Pa, b0 = Ba;
            Pa = Pb;
            Ba = a.buf[Pa];
            Pb = pos_of(c + 2u);
            // the documents that start in this chunk (index space: no load)
            unsigned long long DSc = 0;
            while (nxt_from < i0 + nvl) {
                DSc |= 1ull << (nxt_from - i0);
                ++jn;
                nxt_from = run.from[jn <= rn ? jn : rn] - I0;
                if (jn > rn) nxt_from = 0xFFFFFFFFu;
            }
            const uint32_t dj = jc + cw_below(DSc) + (cw_lanes(DSc) ? 1u : 0u);
            jc += (uint32_t)__popcll(DSc);
            const uint32_t token = first_byte_token[b0];
            const unsigned long long COL = cw_ballot(b0 == ':') & VL;
            const unsigned long long SEP = (cw_ballot(b0 == ',') & VL) | COL;
            // what stands in front of a structural -- nothing, if it is a document's first
            const unsigned long long S1 = ((SEP << 1) | (SEPp >> 63)) & ~DSc, C1 = ((COL << 1) | (COLp >> 63)) & ~DSc;
            // never valid: a separator behind a separator, a separator as a document's last structural
            unsigned long long ends = DSc >> 1;
            if (nxt_from == i0 + 64u || c + 1u == nchunks) ends |= 1ull << (nvl - 1u);
            const unsigned long long bad_sep = (SEP & S1) | (SEP & ends);
            if (bad_sep) fail_lanes(bad_sep, dj);
            uint32_t pre = cw_lanes(S1) ? TOK_COMMA : 0u;
            pre = cw_lanes(C1) ? TOK_COLON : pre;
            const unsigned long long TOK = VL & ~SEP;
            const uint32_t slot = (tail + cw_below(TOK)) & (TS_RING - 1u);
            if (cw_lanes(TOK)) ring.e[slot] = make_uint2(p, token | pre | (dj << 24));
            tail += (uint32_t)__popcll(TOK);
            SEPp = SEP;
            COLp = COL;
            ++c;
        };
        for (;;) {
            while (c < nchunks && tail - head <= TS_RING - 64u) ingest();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t avail = tail - head;
            if (avail == 0u) break;
            // ---- one token step ----
            const uint32_t na = avail < 64u ? avail : 64u;
            const uint2 re = ring.e[(head + (uint32_t)lane) & (TS_RING - 1u)];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // a step ends in front of an opening bracket whose successor is not at hand (is it an empty container?)
            const bool more = c < nchunks || avail > 64u;
            const uint32_t nv = (more && ((uint32_t)__builtin_amdgcn_readlane((int)re.y, 63) & 7u) <= TK_OPEN_O) ? 63u : na;
            const unsigned long long V = cw_first(nv);
            const uint32_t p = re.x, token = cw_lanes(V) ? re.y : ((uint32_t)TK_NONE | 0xFF000000u);
            const uint32_t tk = token & 7u, dj = cw_lanes(V) ? token >> 24 : 0u;
            const unsigned long long OPEN = cw_ballot(tk <= TK_OPEN_O), CLOSE = cw_ballot(tk <= TK_CLOSE_O) & ~OPEN;
            const unsigned long long Q = cw_ballot(tk == TK_STRING), PRIM = cw_ballot(tk >= TK_ATOM);
            // my neighbours' tokens; a document's first token has none in front, its predecessor none behind
            const uint32_t prev_raw = (uint32_t)__builtin_amdgcn_update_dpp((int)c_token, (int)token, 0x138, 0xf, 0xf, false);   // wave_shr:1
            const uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp((int)TK_NONE, (int)token, 0x130, 0xf, 0xf, false);     // wave_shl:1
            const unsigned long long DS = cw_ballot(((prev_raw ^ token) >> 24) != 0u) & V;
            const uint32_t prev = cw_lanes(DS) ? (uint32_t)TK_NONE : prev_raw;
            // (1) empty containers: an opening bracket directly followed by its closing bracket is ONE value (TapeBuilder.java:205-208)
            const unsigned long long EO = cw_ballot(((next ^ (tk + 2u)) & (7u | TOK_COMMA | TOK_COLON)) == 0u) & OPEN & ~(DS >> 1);
            const unsigned long long EC = CLOSE & ((EO << 1) | ((unsigned long long)c_eo & ~DS));
            // (2) ONE scan, three fields: 1 + up - down | strings << 8 | tape words << 16
            const uint32_t inc = ((token & TOK_SCAN_FIELDS) >> 5) | (cw_lanes(Q) ? 0x100u : 0u);
            const uint32_t scan3 = cw_incl_scan(inc);
            const uint32_t tot3 = cw_last(scan3);
            const uint32_t excl = scan3 - inc;
            const int D = A_d + (int)(excl & 0xFFu) - lane;
            const uint32_t aq = A_q + ((excl >> 8) & 0xFFu), aw = A_w + (excl >> 16);
            // (3) my document: where its tape goes; its bases (a first token sets them)
            const uint4 rec = run.rec[dj];
            if (cw_lanes(DS)) run.dyn[dj] = make_uint4((uint32_t)D, aq - run.dso[dj], aw - 1u, 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 dyn = run.dyn[dj];
            const int h = D - (int)dyn.x;
            const uint32_t sord = aq - dyn.y, tpos = aw - dyn.z;
            const uint32_t rec_off = cw_lanes(Q) ? a.soff[sord] : 0u;  // (used one step later)
            // the depth in front of a document's first token, as the document in front of it counts: it must be back at 0
            const int dbase_prev = __builtin_amdgcn_update_dpp(c_dbase, (int)dyn.x, 0x138, 0xf, 0xf, false);  // wave_shr:1
            unsigned long long NC = DS & cw_ballot(D != dbase_prev);
            if (!seen) NC &= ~1ull;
            // only a document's first token stands at depth 0 (and the grammar wants an opening bracket there)
            const unsigned long long LOW = cw_ballot(h < 1) & V & ~DS;
            const unsigned long long DEEP = cw_ballot(h >= depth_limit);
            const unsigned long long NOROOM = cw_ballot(tpos + (inc >> 16) >= rec.z) & V;  // (+ the closing root word)
            // Tokens of a document that failed in an earlier step take no part in the LDS operations below; a token at depth < 1
            // (it fails its document in this step) adds to no counter -- the level its depth wraps to may be a real one of an
            // earlier document in the step -- but an opening bracket among them still REGISTERS: the tokens it contains stand at
            // depth >= 1 and must find it, not the last bracket of an earlier document.
            const unsigned long long FL0 = failed ? cw_ballot(((uint32_t)(failed >> dj) & 1u) != 0u) : 0ull;
            const unsigned long long ACT = V & ~LOW & ~FL0;
            const unsigned long long REG = OPEN & V & ~FL0;
            // (4) the container of every token: the step's opening brackets by level in LDS, the stack in LDS
            const uint32_t lvl = (uint32_t)(h - 1) & 63u;
            lv.open[lane] = 0ull;
            lv.cnt[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (cw_lanes(REG)) atomicOr(&lv.open[(uint32_t)h & 63u], lane_bit);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned long long om = lv.open[lvl];
            const uint32_t om_lo = (uint32_t)om & below_lo, om_hi = (uint32_t)(om >> 32) & below_hi;
            const uint32_t lz_hi = om_hi ? (uint32_t)__builtin_clz(om_hi) : 0xFFFFFFFFu;
            const uint32_t lz_lo = (om_lo ? (uint32_t)__builtin_clz(om_lo) : 0xFFFFFFFFu) | 32u;
            const uint32_t lz = lz_hi < lz_lo ? lz_hi : lz_lo;
            const unsigned long long IN_STEP = cw_ballot((int32_t)lz >= 0);
            const uint32_t par_lane = (63u - lz) & 63u;
            uint32_t* const my_counter = cw_lanes(IN_STEP) ? &lv.cnt[par_lane] : &lv.stk[lvl].y;
            uint32_t add = (token >> 3) & 1u;
            add = cw_lanes(CLOSE & ~EC & IN_STEP) ? add | 0x80000000u : add;
            if (cw_lanes(ACT)) atomicAdd(my_counter, add);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t se_x = lv.stk[lvl].x;
            const uint32_t pcnt = *my_counter;
            const uint32_t own = lv.cnt[lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t opener = tpos | ((tk - 1u) & 0x80000000u);  // is-array << 31 (TK_OPEN_A = 0)
            if (cw_lanes(REG & ~EO & ~cw_ballot((int32_t)own < 0))) lv.stk[(uint32_t)h & 63u] = make_uint2(opener, own);
            const uint32_t from_step = (uint32_t)__shfl((int)opener, (int)par_lane);
            const uint32_t par = cw_lanes(IN_STEP) ? from_step : se_x;
            const uint32_t par_tpos = par & 0x7FFFFFFFu;
            const uint32_t par_cnt = pcnt & 0x3FFFFFFFu;
            // (5) the token grammar: one table entry
            const uint32_t gi = (token & 0x1Fu) | ((prev & 0x17u) << 5) | ((par >> 21) & 0x400u);
            unsigned long long BAD = ((cw_ballot(grammar[gi] == 0) | NOROOM | (OPEN & ~EO & DEEP)) & V) | LOW;
            if (string_errors) {
                bool bad_string = false;
                if (cw_lanes(Q)) {
                    const uint8_t* hh = a.sb + rec_off;
                    bad_string = hh[0] == 0xFF && hh[1] == 0xFF && hh[2] == 0xFF;
                }
                BAD |= cw_ballot(bad_string);
            }
            unsigned long long FL = FL0;
            if (BAD | NC) {  // (rare) whose documents fail: a bad token's own, a first token's predecessor's
                fail_lanes(BAD, dj);
                const uint32_t dj_prev = prev_raw >> 24;
                fail_lanes(NC, dj_prev);
                FL = cw_ballot(((uint32_t)(failed >> dj) & 1u) != 0u);
            }
            const unsigned long long LIVE = V & ~FL;
            // (6) the tape words of this step
            unsigned long long* const T = tape + ((((unsigned long long)rec.y << 32) | rec.x) + tpos);
            if (cw_lanes(PQ)) *pq_addr = tape_word('"', a.string_base + pq_off);  // the previous step's strings
            PQ = Q & LIVE;
            pq_addr = T;
            pq_off = rec_off;
            if (PRIM & LIVE) {  // atoms and numbers: queued, parsed 64 at a time (flush_primitives)
                const unsigned long long PL = PRIM & LIVE;
                const uint32_t qs = (qtail + cw_below(PL)) & 127u;
                if (cw_lanes(PL)) {
                    const unsigned long long d = reinterpret_cast<unsigned long long>(T);
                    pq.e[qs] = make_uint4(p, rec.w, (uint32_t)d, (uint32_t)(d >> 32));
                }
                qtail += (uint32_t)__popcll(PL);
            }
            {   // brackets: an empty pair is two self-contained words (TapeBuilder.java:205-208); a closing bracket writes its own
                // word and its container's opening word (:197-203: element count = commas + 1, saturated)
                const uint32_t type_hi = __builtin_amdgcn_perm(token, 0u, 0x050C0C0Cu);  // the bracket itself << 24
                uint32_t pay1 = cw_lanes(EC) ? tpos : par_tpos;
                pay1 = cw_lanes(EO) ? tpos + 2u : pay1;
                if (cw_lanes((EO | CLOSE) & LIVE)) T[0] = ((unsigned long long)type_hi << 32) | pay1;
                uint32_t cnt = par_cnt + 1u;
                if (cnt > 0xFFFFFFu) cnt = 0xFFFFFFu;
                if (cw_lanes(CLOSE & ~EC & LIVE)) {
                    unsigned long long* const TP = T - tpos + par_tpos;
                    *TP = ((unsigned long long)((type_hi - 0x02000000u) | cnt) << 32) | (tpos + 1u);
                }
            }
            // the document in front of a first token has ended: both its root words, its tape length (visitDocumentEnd,
            // TapeBuilder.java:45-48); not for one that failed (NC included: it is in `failed` by now)
            unsigned long long DSW = DS;
            if (!seen) DSW &= ~1ull;
            if (DSW) {
                const uint32_t pj = (prev_raw >> 24) & (TS_RUN - 1u);
                if (cw_lanes(DSW) && !((uint32_t)(failed >> pj) & 1u)) {
                    const uint4 rp = run.rec[pj];
                    const uint32_t t0p = aw - run.dyn[pj].z;  // position of its closing root word
                    unsigned long long* const TP = tape + (((unsigned long long)rp.y << 32) | rp.x);
                    TP[t0p] = tape_word('r', 0);
                    TP[0] = tape_word('r', t0p + 1u);
                    if (a.tape_lens) a.tape_lens[rp.w] = t0p + 1u;
                }
            }
            // (7) carries
            const uint32_t lastv = nv - 1u;
            A_d += (int)(tot3 & 0xFFu) - (int)nv;
            A_q += (tot3 >> 8) & 0xFFu;
            A_w += tot3 >> 16;
            c_token = (uint32_t)__builtin_amdgcn_readlane((int)token, (int)lastv);
            c_eo = cw_bit(EO, lastv);
            c_dbase = __builtin_amdgcn_readlane((int)dyn.x, (int)lastv);
            seen = true;
            head += nv;
            if (qtail - qhead >= 64u) flush_primitives(64u);
        }
        if (cw_lanes(PQ)) *pq_addr = tape_word('"', a.string_base + pq_off);  // the last step's strings
        // ================= the run's end: its last document, the failed ones =================
        if (seen) {
            const uint32_t pj = (c_token >> 24) & (TS_RUN - 1u);
            if (A_d != c_dbase) failed |= 1ull << pj;  // the root container is never closed (JsonIterator.java:39-41,:51-53)
            if (!((failed >> pj) & 1ull) && lane < 2) {
                const uint4 rp = run.rec[pj];
                const uint32_t t0p = A_w - run.dyn[pj].z;
                unsigned long long* const TP = tape + (((unsigned long long)rp.y << 32) | rp.x);
                TP[lane == 0 ? t0p : 0u] = tape_word('r', lane == 0 ? 0u : t0p + 1u);
                if (a.tape_lens && lane == 0) a.tape_lens[rp.w] = t0p + 1u;
            }
        }
        // (a document all of whose structurals are separators never reached a step: the ingest failed it)
        if ((uint32_t)lane < rn && ((failed >> lane) & 1ull)) {
            const uint32_t g = run.rec[lane].w;
            if (a.tape_lens) a.tape_lens[g] = 0u;
            send_to_exact(g);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (qtail != qhead) flush_primitives(qtail - qhead);  // (fewer than 64 by construction)
}

// ---- the chunk passes around k_coop_walk<true> --------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_chunk_summary(const uint8_t* __restrict__ buf, const uint32_t* __restrict__ idx, const unsigned long long* __restrict__ index_offsets,
                ChunkWs cw) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    const unsigned long long from = index_offsets[0], to = index_offsets[1];
    if (to - from <= CW_SINGLE_N) {  // a short document after all (the host only knows a bound): the single-wave sweep
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(cw.fallback, 1u);
        return;
    }
    const uint32_t chunk = cw_chunk_of(to - from);
    const uint64_t nchunks = (to - from + chunk - 1) / chunk;
    for (uint64_t k = (uint64_t)blockIdx.x * 4 + wv; k < nchunks; k += nwaves) {
        const unsigned long long wfrom = from + k * chunk, wto = wfrom + chunk < to ? wfrom + chunk : to;
        const uint64_t nsteps = (wto - wfrom + 63) / 64;
        uint32_t st_tpos = 0, st_cnt = 0;  // LANE = relative level + CW_BIAS
        unsigned long long arr_mask = 0;
        int H = CW_BIAS, min_after = 0x7FFF;
        uint32_t T = 0, S = 0;
        bool out_of_range = false;
        uint32_t p_n = wfrom + lane < wto ? idx[wfrom + lane] : 0u;
        for (uint64_t s = 0; s < nsteps; ++s) {
            const uint64_t i = wfrom + s * 64 + lane;
            const bool valid = i < wto;
            const uint32_t c = buf[p_n];
            if (s + 1 < nsteps) p_n = i + 64 < wto ? idx[i + 64] : 0u;
            const uint32_t cls = valid ? class_of(c) : K_COMMA;
            const bool is_open = valid && cls <= K_OPEN_O, is_close = valid && (cls == K_CLOSE_A || cls == K_CLOSE_O);
            const uint32_t up = is_open ? 1u : 0u, down = is_close ? 1u : 0u;
            const uint32_t iu = cw_incl_scan(up), id = cw_incl_scan(down);
            const int h = H + (int)(iu - up) - (int)(id - down);
            const bool is_num = valid && cls == K_PRIM && (c == '-' || c - '0' <= 9u);
            const uint32_t words = !valid || cls == K_COMMA || cls == K_COLON ? 0u : (is_num ? 2u : 1u);
            const uint32_t iw = cw_incl_scan(words);
            const uint32_t tpos = T + iw - words;
            const uint32_t nstr = (uint32_t)__popcll(__ballot(valid && cls == K_QUOTE));  // strings of the step (S counts them)
            const int plevel = h - 1;
            const int after = cw_wave_minmax<false>(valid ? h + (int)up - (int)down : 0x7FFF);
            const int hmin = cw_wave_minmax<false>(valid ? plevel : 0x7FFF), hmax = cw_wave_minmax<true>(valid ? (is_open ? h : plevel) : -0x7FFF);
            min_after = min(min_after, after);
            if (hmin < 0 || hmax >= CW_LEVELS) {  // the depth swings out of the biased window: not a chunk for this path
                out_of_range = true;
                break;
            }
            for (int L = hmin; L <= hmax; ++L) {
                const unsigned long long O = __ballot(is_open && h == L);
                const unsigned long long C = __ballot(valid && cls == K_COMMA && plevel == L);
                const unsigned long long Z = __ballot(is_close && plevel == L);
                if (O) {
                    const int al = 63 - __builtin_clzll(O);
                    const unsigned long long above = al == 63 ? 0ull : ~((2ull << al) - 1ull);
                    if (!(Z & above)) {
                        const uint32_t tp = (uint32_t)__builtin_amdgcn_readlane((int)tpos, al);
                        const uint32_t kc = (uint32_t)__builtin_amdgcn_readlane((int)cls, al);
                        st_tpos = lane == L ? tp : st_tpos;
                        st_cnt = lane == L ? (uint32_t)__popcll(C & above) : st_cnt;
                        arr_mask = kc == K_OPEN_A ? (arr_mask | (1ull << L)) : (arr_mask & ~(1ull << L));
                    }
                } else if (!Z && C) {
                    const uint32_t sk = (uint32_t)__builtin_amdgcn_readlane((int)st_cnt, L);
                    st_cnt = lane == L ? sk + (uint32_t)__popcll(C) : st_cnt;
                } else if (Z && !O) {
                    // the container of this level (from before the chunk) closed: what the lane counted so far belonged to it;
                    // a container that opens at this level later starts from its own count (the O branch above)
                    st_cnt = lane == L ? 0u : st_cnt;
                }
            }
            H += (int)cw_last(iu) - (int)cw_last(id);
            T += cw_last(iw);
            S += nstr;
        }
        if (out_of_range && lane == 0) atomicOr(cw.fallback, 1u);
        if (lane == 0) {
            cw.sum.delta[k] = H - CW_BIAS;
            cw.sum.min_after[k] = (min_after == 0x7FFF ? CW_BIAS : min_after) - CW_BIAS;
            cw.sum.words[k] = T;
            cw.sum.ssz[k] = S;
            cw.sum.exp_arr[k] = arr_mask;
        }
        cw.sum.exp_tpos[k * 64 + lane] = st_tpos;
        cw.sum.exp_cnt[k * 64 + lane] = st_cnt;
    }
}

// ---- the entry states: a scan over the chunk summaries ------------------------------------------------------------------
// Applying a summary to a state is associative, so the scan has two levels: a wave per group of 64 chunks composes the
// group's summary in relative levels (k_group_summary), ONE wave walks the groups with absolute levels (k_top_scan), and
// a wave per group replays its chunks from the group's entry state (k_group_replay).  A 64 MiB document (35 k chunks)
// costs 64 + 550 + 64 sequential applications instead of 35 k.
struct ScanState {
    int H;
    uint32_t T;
    unsigned long long S, arr;
    uint32_t st_tpos, st_cnt;  // LANE = level
    bool rc, bad;
    int min_after;
    bool under;                // the depth went BELOW zero somewhere in front (a closing bracket without an opening one)
};
struct SumRow {
    int d, ma;
    uint32_t words, ssz, row_tpos, row_cnt;
    unsigned long long arr;
};
__device__ inline SumRow load_row(const ChunkSum& q, uint64_t k, int lane) {
    SumRow r;
    r.d = q.delta[k];
    r.ma = q.min_after[k];
    r.words = q.words[k];
    r.ssz = q.ssz[k];
    r.row_tpos = q.exp_tpos[k * 64 + lane];
    r.row_cnt = q.exp_cnt[k * 64 + lane];
    r.arr = q.exp_arr[k];
    return r;
}
template <bool RELATIVE>
__device__ inline void scan_apply(ScanState& s, const SumRow& r, int lane) {
    const int m = r.ma < 0 ? r.ma : 0;
    if (s.H + r.ma <= 0) s.rc = true;  // (absolute levels) the depth comes back to zero inside: the root value ends there
    if (s.H + r.ma < 0) s.under = true;
    s.min_after = min(s.min_after, s.H + r.ma);
    const int src = lane - s.H + CW_BIAS;  // my level's lane in the export
    const bool in_src = src >= 0 && src < CW_LEVELS;
    const uint32_t e_tpos = (uint32_t)__shfl((int)r.row_tpos, src & 63), e_cnt = (uint32_t)__shfl((int)r.row_cnt, src & 63);
    if (in_src && lane == s.H + m - 1) s.st_cnt += e_cnt;              // the innermost container left untouched: its commas
    if (in_src && lane >= s.H + m && lane < s.H + r.d) {               // opened inside and left open
        s.st_tpos = s.T + e_tpos;
        s.st_cnt = e_cnt;
    }
    const int lo = s.H + m < 0 ? 0 : s.H + m, hi = s.H + r.d > CW_LEVELS ? CW_LEVELS : s.H + r.d;
    if (hi > lo) {  // kinds of the levels [lo, hi): bit (level - H + BIAS) of the export
        const unsigned long long mask = (hi == 64 ? ~0ull : (1ull << hi) - 1ull) & ~((1ull << lo) - 1ull);
        const int sh = s.H - CW_BIAS;
        const unsigned long long moved = sh >= 0 ? (sh >= 64 ? 0ull : r.arr << sh) : (-sh >= 64 ? 0ull : r.arr >> -sh);
        s.arr = (s.arr & ~mask) | (moved & mask);
    }
    if (s.H + r.d >= CW_LEVELS) s.bad = true;           // deeper than the stack registers
    if (RELATIVE && s.H + m < 1) s.bad = true;          // the swing leaves the biased window
    s.H = s.H + r.d < 0 ? 0 : s.H + r.d;
    s.T += r.words;
    s.S += r.ssz;
}
__device__ inline void store_in(const ChunkIn& q, uint64_t k, const ScanState& s, int lane) {
    q.tpos[k * 64 + lane] = s.st_tpos;
    q.cnt[k * 64 + lane] = s.st_cnt;
    if (lane == 0) {
        q.H[k] = (uint32_t)s.H;
        q.T[k] = s.T;
        q.S[k] = s.S;
        q.arr[k] = s.arr;
        q.root_closed[k] = (s.rc ? 1u : 0u) | (s.under ? 2u : 0u);
    }
}

// applies the summaries [k0, k1) in order; the rows are requested four ahead (a lone wave pays a full memory latency
// per dependent load otherwise); pre(k, state) runs in front of summary k
template <bool RELATIVE, class Pre>
__device__ inline void scan_range(ScanState& s, const ChunkSum& q, uint64_t k0, uint64_t k1, int lane, Pre&& pre) {
    SumRow nx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (k0 + j < k1) nx[j] = load_row(q, k0 + j, lane);
    for (uint64_t k = k0; k < k1; k += 4) {
        SumRow cur[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nx[j];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + 4 + j < k1) nx[j] = load_row(q, k + 4 + j, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < k1) {
                pre(k + j, s);
                scan_apply<RELATIVE>(s, cur[j], lane);
            }
    }
}

__global__ void __launch_bounds__(256)
k_group_summary(const unsigned long long* __restrict__ index_offsets, ChunkWs cw) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    const uint64_t n_structurals = index_offsets[1] - index_offsets[0];
    if (n_structurals <= CW_SINGLE_N) return;
    const uint32_t chunk = cw_chunk_of(n_structurals);
    const uint64_t nchunks = (n_structurals + chunk - 1) / chunk;
    const uint32_t CW_GROUP = cw_group_of(nchunks);
    const uint64_t ngroups = (nchunks + CW_GROUP - 1) / CW_GROUP;
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + wv; g < ngroups; g += nwaves) {
        const uint64_t k0 = g * CW_GROUP, k1 = k0 + CW_GROUP < nchunks ? k0 + CW_GROUP : nchunks;
        ScanState s = {CW_BIAS, 0u, 0ull, 0ull, 0u, 0u, false, false, 0x7FFF, false};
        scan_range<true>(s, cw.sum, k0, k1, lane, [](uint64_t, const ScanState&) {});
        if (s.bad && lane == 0) atomicOr(cw.fallback, 1u);
        if (lane == 0) {
            cw.gsum.delta[g] = s.H - CW_BIAS;
            cw.gsum.min_after[g] = s.min_after - CW_BIAS;
            cw.gsum.words[g] = s.T;
            cw.gsum.ssz[g] = (uint32_t)s.S;
            cw.gsum.exp_arr[g] = s.arr;
        }
        cw.gsum.exp_tpos[g * 64 + lane] = s.st_tpos;
        cw.gsum.exp_cnt[g * 64 + lane] = s.st_cnt;
    }
}

__global__ void __launch_bounds__(64)
k_top_scan(const unsigned long long* __restrict__ index_offsets, const unsigned long long* __restrict__ doc_str_offsets, ChunkWs cw) {
    const int lane = threadIdx.x & 63;
    const uint64_t n_structurals = index_offsets[1] - index_offsets[0];
    if (n_structurals <= CW_SINGLE_N) return;
    const uint32_t chunk = cw_chunk_of(n_structurals);
    const uint64_t nchunks = (n_structurals + chunk - 1) / chunk;
    const uint32_t CW_GROUP = cw_group_of(nchunks);
    const uint64_t ngroups = (nchunks + CW_GROUP - 1) / CW_GROUP;
    if (*cw.fallback != 0 || ngroups == 0) return;
    ScanState s = {0, 1u, doc_str_offsets[0], 0ull, 0u, 0u, false, false, 0x7FFF, false};
    scan_range<false>(s, cw.gsum, 0, ngroups, lane, [&](uint64_t g, const ScanState& at) { store_in(cw.gin, g, at, lane); });
    if (lane == 0) {
        if (s.bad) atomicOr(cw.fallback, 1u);
        cw.fin[0] = (uint32_t)s.H;
        cw.fin[1] = s.T;
        cw.fin[2] = (s.H >= 1 && s.H <= CW_LEVELS && ((s.arr >> (s.H - 1)) & 1ull)) ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(256)
k_group_replay(const unsigned long long* __restrict__ index_offsets, ChunkWs cw) {
    if (*cw.fallback != 0) return;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    const uint64_t n_structurals = index_offsets[1] - index_offsets[0];
    if (n_structurals <= CW_SINGLE_N) return;
    const uint32_t chunk = cw_chunk_of(n_structurals);
    const uint64_t nchunks = (n_structurals + chunk - 1) / chunk;
    const uint32_t CW_GROUP = cw_group_of(nchunks);
    const uint64_t ngroups = (nchunks + CW_GROUP - 1) / CW_GROUP;
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + wv; g < ngroups; g += nwaves) {
        const uint64_t k0 = g * CW_GROUP, k1 = k0 + CW_GROUP < nchunks ? k0 + CW_GROUP : nchunks;
        ScanState s;
        s.H = (int)cw.gin.H[g];
        s.T = cw.gin.T[g];
        s.S = cw.gin.S[g];
        s.arr = cw.gin.arr[g];
        s.st_tpos = cw.gin.tpos[g * 64 + lane];
        s.st_cnt = cw.gin.cnt[g * 64 + lane];
        s.rc = (cw.gin.root_closed[g] & 1u) != 0;
        s.under = (cw.gin.root_closed[g] & 2u) != 0;
        s.bad = false;
        s.min_after = 0x7FFF;
        scan_range<false>(s, cw.sum, k0, k1, lane, [&](uint64_t k, const ScanState& at) { store_in(cw.in, k, at, lane); });
    }
}

__global__ void __launch_bounds__(64)
k_chunk_finish(const uint8_t* __restrict__ buf, const uint32_t* __restrict__ idx, const unsigned long long* __restrict__ index_offsets,
               const uint32_t* __restrict__ doc_status, unsigned long long* __restrict__ scratch_tape, uint32_t* __restrict__ tape_lens,
               int32_t* __restrict__ doc_errors, const Stage1Result* __restrict__ dev_count, const UnescapeResult* __restrict__ dev_strings,
               ChunkWs cw, SingleFinish fin, uint32_t with_tail) {
    __shared__ uint32_t wa[SJ_BIG_WORDS], wb[SJ_BIG_WORDS];
    if (*cw.fallback != 0) {  // the single-wave sweep writes the document's result
        if (with_tail && fin.pack && threadIdx.x == 0) fin.pack->fallback = 1;  // (optimistic tail: the caller queues that sweep)
        return;
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long from = index_offsets[0], to = index_offsets[1];
    const uint32_t chunk = cw_chunk_of(to - from);
    const uint64_t nchunks = (to - from + chunk - 1) / chunk;
    const uint32_t st = doc_status ? doc_status[0] : 0u;
    int code = 0;
    uint32_t tlen = 0;
    const bool upstream_failed = (dev_count && (dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL))) ||
                                 (dev_strings && (dev_strings->flags & 0xFu));
    // some string of the launch has a malformed escape (rare): then every string's record header is looked at
    const bool string_errors = dev_strings && dev_strings->first_error_inv != 0;
    if (upstream_failed) code = SJMI_E_CAPACITY;
    else if (st & SJMI_ST_UTF8) code = SJMI_E_UTF8;
    else if (st & SJMI_ST_UNCLOSED) code = SJMI_E_UNCLOSED_STRING;
    else if (st & SJMI_ST_UNESCAPED) code = SJMI_E_UNESCAPED_CHARS;
    else if (from == to) code = SJMI_E_NO_STRUCTURAL;
    if (code == 0) {
        // the first error by position over the chunks
        unsigned long long best = ~0ull;  // (position << 32) | chunk
        for (uint64_t k = lane; k < nchunks; k += 64) {
            const uint32_t ep = cw.err_pos[k];
            if (ep != 0xFFFFFFFFu) {
                const unsigned long long key = ((unsigned long long)ep << 32) | (unsigned long long)k;
                best = key < best ? key : best;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(best, d);
            best = o < best ? o : best;
        }
        const uint32_t root_c = buf[idx[from]], rk = class_of(root_c);
        const uint32_t last_c = buf[idx[to - 1]];
        if (rk <= K_OPEN_O && last_c != root_c + 2) code = rk == K_OPEN_O ? SJMI_E_UNCLOSED_OBJECT : SJMI_E_UNCLOSED_ARRAY;  // JsonIterator.java:39-41,:51-53
        else if (best != ~0ull) code = cw.err_code[(uint32_t)best];
        else if (cw.fin[0] != 0) code = cw.fin[2] ? SJMI_E_NO_COMMA_ARRAY : SJMI_E_NO_COMMA_OBJECT;  // the walker reads the sentinel (BitIndexes.java:82-96)
        if (code == 0) {
            unsigned long long* const T = scratch_tape + 2 * from;
            const uint32_t T0 = cw.fin[1];
            tlen = T0 + 1;
            if (lane == 0 && tlen <= fin.tape_capacity) {
                T[T0] = tape_word('r', 0);      // visitDocumentEnd, TapeBuilder.java:45-48
                T[0] = tape_word('r', tlen);
            }
        }
    }
    if (lane == 0) {
        tape_lens[0] = tlen;
        doc_errors[0] = code;
        if (with_tail) cw_single_finish(buf, fin, tlen, code, wa, wb);  // (nothing else is queued behind this launch)
    }
}

// single document: the delimiters the batch kernels expect, from the stage-1 record that is still on the device
__global__ void k_single_doc_setup(const Stage1Result* __restrict__ res, unsigned long long len, unsigned long long* doc_offsets,
                                   unsigned long long* index_offsets, uint32_t* doc_status, unsigned long long* doc_str_offsets,
                                   uint32_t* walk_result, uint32_t* slow_header) {
    // (also what two memsets would do on the latency path: the walk's result record and the header of its literal list / flags)
    if (walk_result && threadIdx.x < sizeof(WalkResult) / 4) walk_result[threadIdx.x] = 0;
    if (slow_header && threadIdx.x < 16) slow_header[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        doc_offsets[0] = 0;
        doc_offsets[1] = len;
        index_offsets[0] = 0;
        index_offsets[1] = (res->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL)) ? 0ull : res->count;
        doc_status[0] = res->status & 0xFFu;
        doc_str_offsets[0] = 0;
        doc_str_offsets[1] = 0;
    }
}

hipError_t single_doc_setup_launch(const Stage1Result* d_res, uint64_t len, unsigned long long* d_doc_offsets,
                                   unsigned long long* d_index_offsets, uint32_t* d_doc_status, unsigned long long* d_doc_str_offsets,
                                   hipStream_t stream, WalkResult* d_walk_result, void* d_slow_header) {
    hipLaunchKernelGGL(k_single_doc_setup, dim3(1), dim3(64), 0, stream, d_res, (unsigned long long)len, d_doc_offsets, d_index_offsets,
                       d_doc_status, d_doc_str_offsets, reinterpret_cast<uint32_t*>(d_walk_result), static_cast<uint32_t*>(d_slow_header));
    return hipGetLastError();
}

static size_t chunk_bound(uint64_t count_bound) {  // n <= CW_SMALL_N: chunks of 128, else of 512
    const uint64_t small = (count_bound < CW_SMALL_N ? count_bound : CW_SMALL_N) / 128, large = count_bound / 512;
    return (size_t)(small > large ? small : large) + 2;
}
static size_t group_bound(uint64_t count_bound) { return chunk_bound(count_bound) / 8 + 2; }
size_t coop_chunk_workspace_bytes(uint64_t count_bound) {
    return (chunk_bound(count_bound) + group_bound(count_bound)) * (4 * 64 * sizeof(uint32_t) + 24 * sizeof(unsigned long long)) + 4096 + 64;
}

static ChunkWs chunk_ws(void* ws, uint64_t count_bound, uint32_t* flags) {
    uint8_t* p = static_cast<uint8_t*>(ws);
    ChunkWs c;
    auto take = [&](size_t bytes) { uint8_t* r = p; p += (bytes + 63) / 64 * 64; return r; };
    c.fallback = flags;  // (in the header of the undecided-literal list: one memset per launch covers both)
    c.fin = c.fallback + 4;
    auto sums = [&](size_t n) {
        ChunkSum q;
        q.exp_tpos = reinterpret_cast<uint32_t*>(take(n * 64 * 4));
        q.exp_cnt = reinterpret_cast<uint32_t*>(take(n * 64 * 4));
        q.exp_arr = reinterpret_cast<unsigned long long*>(take(n * 8));
        q.delta = reinterpret_cast<int32_t*>(take(n * 4));
        q.min_after = reinterpret_cast<int32_t*>(take(n * 4));
        q.words = reinterpret_cast<uint32_t*>(take(n * 4));
        q.ssz = reinterpret_cast<uint32_t*>(take(n * 4));
        return q;
    };
    auto ins = [&](size_t n) {
        ChunkIn q;
        q.tpos = reinterpret_cast<uint32_t*>(take(n * 64 * 4));
        q.cnt = reinterpret_cast<uint32_t*>(take(n * 64 * 4));
        q.S = reinterpret_cast<unsigned long long*>(take(n * 8));
        q.arr = reinterpret_cast<unsigned long long*>(take(n * 8));
        q.H = reinterpret_cast<uint32_t*>(take(n * 4));
        q.T = reinterpret_cast<uint32_t*>(take(n * 4));
        q.root_closed = reinterpret_cast<uint32_t*>(take(n * 4));
        return q;
    };
    const size_t nc = chunk_bound(count_bound), ng = group_bound(count_bound);
    c.sum = sums(nc);
    c.in = ins(nc);
    c.gsum = sums(ng);
    c.gin = ins(ng);
    c.err_pos = reinterpret_cast<uint32_t*>(take(nc * 4));
    c.err_code = reinterpret_cast<int32_t*>(take(nc * 4));
    return c;
}

// d_chunk_ws != nullptr and one document of more than COOP_CHUNK_MIN structurals (by its bound): the chunk-parallel path,
// with the single-wave sweep queued behind it for the (flagged) cases it does not take
constexpr uint64_t COOP_CHUNK_MIN = 1536;      // (a bound in BYTES + 1: documents carry 4 .. 12 bytes per structural)
constexpr uint64_t COOP_OPTIMISTIC_MIN = 16384;  // below: the sweep for a document the chunk path declines is queued beforehand (no second round trip for a short document)
constexpr uint64_t COOP_WALK_MAX_GRID = 8192;  // workgroups of the wave-per-document kernel (grid-stride over the documents)
static size_t coop_slow_bytes() { return 64 + (size_t)CW_SLOW_CAP * 16; }
// the list of undecided literals + the deep levels of every wave of that kernel (see k_coop_walk)
size_t coop_deep_workspace_bytes(uint64_t n_docs) {
    const uint64_t want = (n_docs + 3) / 4;
    return coop_slow_bytes() + (size_t)(want < COOP_WALK_MAX_GRID ? want : COOP_WALK_MAX_GRID) * 4 * CW_OVF_LEVELS * sizeof(unsigned long long);
}
hipError_t coop_walk_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, const uint32_t* d_idx,
                            const unsigned long long* d_index_offsets, const uint32_t* d_doc_status, const uint32_t* d_soff,
                            const uint8_t* d_sb, const unsigned long long* d_doc_str_offsets, uint64_t string_base,
                            int max_depth, unsigned long long* d_scratch_tape, uint32_t* d_tape_lens, int32_t* d_doc_errors,
                            const Stage1Result* dev_count, const UnescapeResult* dev_strings, WalkResult* d_res,
                            hipStream_t stream, void* d_chunk_ws, uint64_t count_bound, void* d_deep_ws,
                            unsigned long long* d_single_tape_offsets, uint64_t tape_capacity, bool header_zeroed,
                            const SingleDocTail& tail) {
    if (!n_docs) return hipSuccess;
    // (the deep-level workspace begins with the list of undecided literals)
    SlowList slow;
    slow.count = static_cast<unsigned long long*>(d_deep_ws);
    slow.rec = slow.count + 8;
    slow.cap = CW_SLOW_CAP;
    d_deep_ws = static_cast<uint8_t*>(d_deep_ws) + coop_slow_bytes();
    if (!header_zeroed) {  // (64 bytes: the list's count and the chunk path's flags)
        hipError_t e0 = hipMemsetAsync(slow.count, 0, 64, stream);
        if (e0 != hipSuccess) return e0;
    }
    const uint32_t abl = (uint32_t)(getenv("SJMI_COOP_ABLATE") ? atoi(getenv("SJMI_COOP_ABLATE")) : 0);
    ChunkWs cw = {};
    static const bool no_chunks = getenv("SJMI_COOP_CHUNKS") && atoi(getenv("SJMI_COOP_CHUNKS")) == 0;
    const bool chunked = d_chunk_ws && n_docs == 1 && count_bound > COOP_CHUNK_MIN && !no_chunks && !tail.no_chunks;
    SingleFinish fin = {slow, d_tape_lens, d_doc_errors, tape_capacity, d_single_tape_offsets, d_res, tail.s1, tail.u, tail.pack};
    const bool optimistic = chunked && d_single_tape_offsets && tail.optimistic && tail.pack && count_bound > COOP_OPTIMISTIC_MIN;
    const uint32_t* only_if = nullptr;
    ExactMode in_place;
    in_place.cap = tail.in_place_cap;
    if (chunked) {
        cw = chunk_ws(d_chunk_ws, count_bound, reinterpret_cast<uint32_t*>(slow.count) + 8);  // (zeroed with the list's count above)
        const uint64_t nchunks = chunk_bound(count_bound);
        const uint64_t want = (nchunks + 3) / 4;
        const unsigned grid = (unsigned)(want < 8192 ? want : 8192);
        hipLaunchKernelGGL(k_chunk_summary, dim3(grid), dim3(256), 0, stream, d_buf, d_idx, d_index_offsets, cw);
        const uint64_t gwant = (group_bound(count_bound) + 3) / 4;
        const unsigned ggrid = (unsigned)(gwant < 4096 ? gwant : 4096);
        hipLaunchKernelGGL(k_group_summary, dim3(ggrid), dim3(256), 0, stream, d_index_offsets, cw);
        hipLaunchKernelGGL(k_top_scan, dim3(1), dim3(64), 0, stream, d_index_offsets, d_doc_str_offsets, cw);
        hipLaunchKernelGGL(k_group_replay, dim3(ggrid), dim3(256), 0, stream, d_index_offsets, cw);
        hipLaunchKernelGGL((k_coop_walk<true>), dim3(grid), dim3(256), 0, stream, d_buf, d_doc_offsets, n_docs, d_idx, d_index_offsets,
                           d_doc_status, d_soff, d_sb, d_doc_str_offsets, (unsigned long long)string_base, max_depth,
                           d_scratch_tape, d_tape_lens, d_doc_errors, dev_count, dev_strings, d_res, abl, cw, (const uint32_t*)nullptr,
                           (unsigned long long*)nullptr, slow, in_place);
        hipLaunchKernelGGL(k_chunk_finish, dim3(1), dim3(64), 0, stream, d_buf, d_idx, d_index_offsets, d_doc_status, d_scratch_tape,
                           d_tape_lens, d_doc_errors, dev_count, dev_strings, cw, fin, optimistic ? 1u : 0u);
        if (optimistic) return hipGetLastError();  // (three launches fewer on the single-document latency path)
        only_if = cw.fallback;
    }
    const uint64_t want = (n_docs + 3) / 4;  // four documents (waves) per workgroup and trip
    const unsigned grid = (unsigned)(want < COOP_WALK_MAX_GRID ? want : COOP_WALK_MAX_GRID);
    hipLaunchKernelGGL((k_coop_walk<false>), dim3(grid), dim3(256), 0, stream, d_buf, d_doc_offsets, n_docs, d_idx, d_index_offsets,
                       d_doc_status, d_soff, d_sb, d_doc_str_offsets, (unsigned long long)string_base, max_depth,
                       d_scratch_tape, d_tape_lens, d_doc_errors, dev_count, dev_strings, d_res, abl, cw, only_if,
                       static_cast<unsigned long long*>(d_deep_ws), slow, in_place);
    if (d_single_tape_offsets)
        hipLaunchKernelGGL(k_single_finish, dim3(1), dim3(64), 0, stream, d_buf, fin);
    else
        hipLaunchKernelGGL(k_slow_doubles, dim3(256), dim3(64), 0, stream, d_buf, slow);  // (nothing listed: 256 waves that leave at once)
    return hipGetLastError();
}

// A batch: the token walker over every document, then the exact walker over the documents it listed (none of a well-formed
// batch: its workgroups read the list's count and leave), then the literals the exact walker could not decide.
hipError_t tok_walk_launch(const TokLaunch& t, hipStream_t stream) {
    if (!t.n_docs) return hipSuccess;
    SlowList slow;
    slow.count = static_cast<unsigned long long*>(t.d_deep_ws);
    slow.rec = slow.count + 8;
    slow.cap = CW_SLOW_CAP;
    unsigned long long* const deep = reinterpret_cast<unsigned long long*>(static_cast<uint8_t*>(t.d_deep_ws) + coop_slow_bytes());
    if (!t.header_zeroed) {
        hipError_t e0 = hipMemsetAsync(slow.count, 0, 64, stream);
        if (e0 != hipSuccess) return e0;
    }
    TokArgs a;
    a.buf = t.d_buf;
    a.idx = t.d_idx;
    a.metas = t.d_metas;
    a.n_docs = (uint32_t)t.n_docs;
    a.max_depth = t.max_depth;
    a.soff = t.d_soff;
    a.sb = t.d_sb;
    a.string_base = t.string_base;
    a.tape = t.d_tape;
    a.tape_alt = t.d_scratch;
    a.sel = t.d_sel;
    a.doc_errors = t.d_doc_errors;
    a.tape_lens = t.d_tape_lens;
    a.list = t.d_list;
    a.dev_count = t.dev_count;
    a.dev_strings = t.dev_strings;
    a.run_docs = TS_RUN;
    const uint64_t want = (t.n_docs + 3) / 4;  // (the exact walker behind it: four documents -- waves -- per workgroup and trip)
    {
        // Documents per run, per launch: a wave walks one run at a time and W waves are resident, so a launch takes
        // j = ceil(runs / W) ROUNDS of runs -- with 16 documents per run a batch of 125,000 (one rank's share of a strong-scaled
        // million) is 1.27 rounds' worth of work done in 2.  The run is sized so that the rounds come out whole:
        // j = ceil(n / (W * TS_RUN)), run = ceil(n / (W * j)).  (125,000 documents: 11 per run, 188 -> 167 us.)
        static std::atomic<unsigned> resident_waves{0};
        unsigned W = resident_waves.load(std::memory_order_relaxed);
        if (!W) {
            int per_cu = 0, cus = 0, dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tok_stream, 256, 0) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1) {
                per_cu = 6;
                cus = 256;
            }
            W = (unsigned)per_cu * (unsigned)cus * 4u;
            resident_waves.store(W, std::memory_order_relaxed);
        }
        static const unsigned run_forced = getenv("SJMI_TS_RUN_DOCS") ? (unsigned)atoi(getenv("SJMI_TS_RUN_DOCS")) : 0u;  // (experiments)
        const uint64_t j = (t.n_docs + (uint64_t)W * TS_RUN - 1) / ((uint64_t)W * TS_RUN);
        uint64_t run = (t.n_docs + (uint64_t)W * j - 1) / ((uint64_t)W * j);
        if (run_forced >= 1 && run_forced <= TS_RUN) run = run_forced;
        if (run < 1) run = 1;
        if (run > TS_RUN) run = TS_RUN;
        a.run_docs = (uint32_t)run;
        const uint64_t runs = (t.n_docs + run - 1) / run, want_s = (runs + 3) / 4;  // four runs (waves) per workgroup and trip
        static const unsigned tok_grid_max = getenv("SJMI_TOK_GRID") ? (unsigned)atoi(getenv("SJMI_TOK_GRID")) : (unsigned)COOP_WALK_MAX_GRID;
        hipLaunchKernelGGL(k_tok_stream, dim3((unsigned)(want_s < tok_grid_max ? want_s : tok_grid_max)), dim3(256), 0, stream, a);
    }
    ExactMode ex;
    ex.list = t.d_list;
    ex.metas = t.d_metas;
    ex.tape = t.d_tape;
    ex.tape_alt = t.d_scratch;
    ex.sel = t.d_sel;
    const uint32_t abl = (uint32_t)(getenv("SJMI_COOP_ABLATE") ? atoi(getenv("SJMI_COOP_ABLATE")) : 0);
    const unsigned xgrid = (unsigned)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL((k_coop_walk<false>), dim3(xgrid), dim3(256), 0, stream, t.d_buf, t.d_doc_offsets, t.n_docs, t.d_idx,
                       t.d_index_offsets, t.d_doc_status, t.d_soff, t.d_sb, t.d_doc_str_ordinals, (unsigned long long)t.string_base,
                       t.max_depth, t.d_scratch, t.d_tape_lens, t.d_doc_errors, t.dev_count, t.dev_strings, t.d_res, abl, ChunkWs{},
                       (const uint32_t*)nullptr, deep, slow, ex);
    hipLaunchKernelGGL(k_slow_doubles, dim3(256), dim3(64), 0, stream, t.d_buf, slow);  // (nothing listed: 256 waves that leave at once)
    return hipGetLastError();
}

}  // namespace sjmi


