// unescape.hip -- batched string unescape on the GPU.
//
// Replaces the per-string calls StringParser.parseString(buf, idx, stringBuffer, stringBufferIdx)
// (/root/reference/src/main/java/org/simdjson/StringParser.java:18-68, helpers :112-161,
//  CharacterUtils.escape :52-83, hexToInt :241-247, IntegerUtils.toBytes :12-17) that the reference's
// sequential stage 2 issues once per '"' structural (TapeBuilder.visitString, TapeBuilder.java:174-177).
// In a valid document the strings are visited in structural order, so record k starts at
// sum_{j<k}(4 + len_j): that is an exclusive prefix sum, and every string is independent.
//
// Three launches over the structural indexes produced by stage 1:
//   k_str_measure : one workgroup per 256..4096 structurals.  A lane whose structural is '"' finds the closing quote
//                   (the last byte > 0x20 before the next structural -- stage 1 already proved it exists) and looks
//                   for a backslash.  No backslash (98 % of twitter.json's strings): length = raw length, done.
//                   Otherwise the WAVE unescapes the string, 64 bytes per step, into a scratch copy; the first
//                   failing string (lowest structural position) is kept with an atomicMax of the complement.
//                   Also the tile's byte total;
//   k_scan_sums   : exclusive scan of the tile totals;
//   k_str_write   : re-derives every record's offset (LDS scan + tile base) and produces the string buffer in
//                   aligned 16-byte chunks, one lane per chunk, gathering from the document or the scratch copy --
//                   [be32 length][bytes], bit-identical to the reference's stringBuffer.
// Output parity domain: string_buffer[0, total).  With an erroneous string the reference throws at that
// string; here every other string is still written, the failing one becomes a 4-byte record FF FF FF <code>
// (the host stage 2 throws when it reaches it) and the first one is also reported by position + code.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1.h"

namespace sjmi {

constexpr int UNESC_THREADS = 256;
// structurals per lane = tile / 256: 16 (4096 per workgroup) for big inputs; a single document of a few hundred KB has
// only tens of thousands of structurals, so small inputs use 1 (256 per workgroup) to have work for every CU
constexpr int UNESC_ITEMS_MAX = 16;

__device__ __forceinline__ bool is_json_ws(uint32_t c) { return c == 0x20 || c == 0x09 || c == 0x0A || c == 0x0D; }

// CharacterUtils.escape (CharacterUtils.java:52-83): 0 = "Escaped unexpected character"
__device__ __forceinline__ uint32_t escape_map(uint32_t e) {
    switch (e) {
    case '"': return 0x22;
    case '/': return 0x2f;
    case '\\': return 0x5c;
    case 'b': return 0x08;
    case 'f': return 0x0c;
    case 'n': return 0x0a;
    case 'r': return 0x0d;
    case 't': return 0x09;
    default: return 0;
    }
}

// position of the closing quote of the string opened at `open`: the last non-whitespace byte before
// the next structural (or before len for the last structural).  Returns 0 if that byte is not a quote
// after `open` (cannot happen when stage 1 reported status 0).
__device__ __forceinline__ uint32_t find_close(const uint8_t* __restrict__ buf, uint32_t open, uint32_t bound) {
    uint32_t p = bound;
    while (p > open + 1 && is_json_ws(buf[p - 1])) --p;
    if (p <= open + 1 || buf[p - 1] != '"') return 0;
    return p - 1;
}

// byte-granular wide accesses: gfx950 global memory runs in unaligned-access mode, hipcc turns these into
// global_load/store_dword[x2|x4] at any byte address
struct __attribute__((packed, aligned(1))) U16B { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) U8B { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) U4B { uint32_t a; };
struct __attribute__((packed, aligned(1))) U2B { uint16_t a; };

// ---- wave-cooperative unescape ------------------------------------------------------------------
// One string with escapes, all 64 lanes: lane j owns source byte (window start + j), 64 bytes per step.  The escape
// structure is bit algebra on ballot masks (the same odd/even backslash-run carry as stage 1,
// StructuralIndexer.java:211-229): every byte knows whether it is a backslash that starts an escape (emits nothing),
// an escaped character (emits its translation; a 'u' decodes its four hex digits, a high surrogate also the
// following \uXXXX), a hex digit consumed by a preceding \u (nothing), or plain (itself).  Output positions are a
// DPP prefix sum of the per-byte output lengths.  Errors are found by the byte that would raise them in
// StringParser.doParseString (:45-61, :112-129); the lowest position wins, which is the sequential parser's first
// error because everything in front of it parsed cleanly.
// (The lane-serial parser it replaced cost ~15 instructions per source byte with the whole wave waiting for one lane:
// on twitter.json more than half of all instructions of the unescape kernels.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t udpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v = udpp_add<0x111, 0xF>(v);  // row_shr:1
    v = udpp_add<0x112, 0xF>(v);  // row_shr:2
    v = udpp_add<0x114, 0xF>(v);  // row_shr:4
    v = udpp_add<0x118, 0xF>(v);  // row_shr:8
    v = udpp_add<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3
    v = udpp_add<0x143, 0xC>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}
// four hex digits packed little endian in w (first digit in the low byte): value, or negative (CharacterUtils.java:241-247)
__device__ __forceinline__ int32_t hex4_word(uint32_t w) {
    uint32_t v = 0, bad = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = (w >> (8 * i)) & 0xFFu;
        const uint32_t dec = c - '0', alpha = (c | 0x20u) - 'a';
        const uint32_t d = dec <= 9u ? dec : alpha + 10u;
        bad |= (dec > 9u) & (alpha > 5u);
        v = (v << 4) | (d & 15u);
    }
    return bad ? -1 : (int32_t)v;
}

// Must be called by all 64 lanes.  [s, e) = the string's bytes (e = its closing quote).  Returns the unescaped
// length or -(SJMI_E_* code); the bytes go to dst[0, length).
__device__ __forceinline__ int64_t unescape_wave(const uint8_t* __restrict__ buf, uint32_t s, uint32_t e,
                                                 uint8_t* __restrict__ dst, int lane) {
    const unsigned long long EVEN = 0x5555555555555555ull;
    unsigned long long prev_U = 0;  // \u escapes ('u' positions) of the previous window
    uint32_t carry = 0;             // the window's first byte is escaped
    uint32_t n = 0;
    for (uint32_t wb0 = s; wb0 < e; wb0 += 256) {
    // (the bytes of four windows are requested together: one memory round trip per 256 source bytes, not per 64)
    uint32_t cw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t pp = wb0 + 64u * u + (uint32_t)lane;
        cw[u] = pp < e ? (uint32_t)buf[pp] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t wb = wb0 + 64u * u;
        if (wb >= e) break;
        const uint32_t pos = wb + (uint32_t)lane;
        const bool valid = pos < e;
        const uint32_t c = cw[u];
        const unsigned long long B = __ballot(valid && c == '\\');
        const unsigned long long bs = B & ~(unsigned long long)carry;
        const unsigned long long follows = (bs << 1) | carry;
        const unsigned long long odd_starts = bs & ~EVEN & ~follows;
        const unsigned long long seq_even = odd_starts + bs;
        const uint32_t carry_out = seq_even < bs ? 1u : 0u;
        const unsigned long long escaped = (EVEN ^ (seq_even << 1)) & follows;
        const bool is_esc = valid && ((escaped >> lane) & 1ull);
        const bool is_start = ((B & ~escaped) >> lane) & 1ull;  // a backslash that starts an escape
        const unsigned long long U = __ballot(is_esc && c == 'u');
        const unsigned long long digits = (U << 1) | (U << 2) | (U << 3) | (U << 4) | (prev_U >> 63) | (prev_U >> 62) |
                                          (prev_U >> 61) | (prev_U >> 60);
        uint32_t outlen = 1, out = c, err = 0;  // out: up to 4 output bytes, first in the low byte
        if (!valid || is_start || ((digits >> lane) & 1ull)) {
            outlen = 0;
        } else if (is_esc) {
            if (c == 'u') {                                                      // StringParser.java:45-57
                int32_t cp = hex4_word(reinterpret_cast<const U4B*>(buf + pos + 1)->a);
                if (cp >= 0xD800 && cp <= 0xDBFF) {                              // parseLowSurrogate :112-124
                    const U8B t = *reinterpret_cast<const U8B*>(buf + pos + 5);  // \ u X X X X
                    if ((t.a & 0xFFFFu) != (uint32_t)('\\' | ('u' << 8))) {
                        err = SJMI_E_LOW_SURROGATE_NO_U;
                    } else {
                        const int32_t low = hex4_word((t.a >> 16) | (t.b << 16)) - 0xDC00;
                        if ((low >> 10) != 0) err = SJMI_E_LOW_SURROGATE_RANGE;
                        else cp = (((cp - 0xD800) << 10) | low) + 0x10000;
                    }
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                    // the second half of a pair (its first half emitted all four bytes), or a lone low surrogate (:53-55)
                    bool paired = false;
                    if (pos >= s + 6) {
                        const bool prev_is_u = lane >= 6 ? ((U >> (lane - 6)) & 1ull) : ((prev_U >> (58 + lane)) & 1ull);
                        if (prev_is_u) {
                            const int32_t hi = hex4_word(reinterpret_cast<const U4B*>(buf + pos - 5)->a);
                            paired = hi >= 0xD800 && hi <= 0xDBFF;
                        }
                    }
                    if (paired) cp = -2;  // emits nothing
                    else err = SJMI_E_LOW_SURROGATE_RESERVED;
                }
                if (!err) {
                    if (cp == -2) {
                        outlen = 0;
                    } else if (cp < 0) {
                        err = SJMI_E_INVALID_UNICODE_ESCAPE;                     // storeCodePointInStringBuffer :127-129
                    } else if (cp <= 0x7F) {
                        out = (uint32_t)cp;
                    } else if (cp <= 0x7FF) {
                        outlen = 2;
                        out = (uint32_t)((cp >> 6) + 192) | ((uint32_t)((cp & 63) + 128) << 8);
                    } else if (cp <= 0xFFFF) {
                        outlen = 3;
                        out = (uint32_t)((cp >> 12) + 224) | ((uint32_t)(((cp >> 6) & 63) + 128) << 8) |
                              ((uint32_t)((cp & 63) + 128) << 16);
                    } else {
                        outlen = 4;
                        out = (uint32_t)((cp >> 18) + 240) | ((uint32_t)(((cp >> 12) & 63) + 128) << 8) |
                              ((uint32_t)(((cp >> 6) & 63) + 128) << 16) | ((uint32_t)((cp & 63) + 128) << 24);
                    }
                }
            } else {                                                             // :58-61
                const uint32_t r = (c & 0x80u) ? 0u : escape_map(c);
                if (r == 0) err = SJMI_E_ESCAPE_UNEXPECTED;
                out = r;
            }
        }
        const unsigned long long errs = __ballot(err != 0);
        if (errs) return -(int64_t)(uint32_t)__builtin_amdgcn_readlane((int)err, __builtin_ctzll(errs));
        const uint32_t incl = wave_incl_scan_u32(outlen);
        uint8_t* q = dst + n + (incl - outlen);
        if (outlen >= 1) q[0] = (uint8_t)out;
        if (outlen >= 2) q[1] = (uint8_t)(out >> 8);
        if (outlen >= 3) q[2] = (uint8_t)(out >> 16);
        if (outlen >= 4) q[3] = (uint8_t)(out >> 24);
        n += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        prev_U = U;
        carry = carry_out;
    }
    }
    return (int64_t)n;
}

// ---- the escaped strings of 64 structurals as ONE packed byte stream -----------------------------------------------------
// unescape_wave gives a whole wave to one string: a 30-byte string uses 30 lanes and pays a memory round trip of its own,
// and where most strings carry an escape (synthetic configs: 5 % of the characters) the wave walks through ten of them one
// after the other.  Here the source bytes of all short escaped strings of the wave's 64 structurals form one virtual
// stream (string j at [voff_j, voff_j + len_j)), lane t of window w owns virtual byte 64 w + t, and the escape algebra of
// unescape_wave runs on the stream: 64 source bytes per trip whatever the lengths are.  What makes it legal: a string ends
// in front of an UNescaped quote, so the backslash run at its end is even (runs may touch across a boundary: harmless, see
// below) and no \uXXXX of a well-formed string crosses a boundary (the digit masks are cut at the boundaries for the
// malformed ones, whose hex test then fails on the closing quote).
// Per-string results through three 64-entry LDS arrays of the wave: first error (lowest position), unescaped length.
__device__ __forceinline__ void wave_fence_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t udpp_max(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t wave_incl_max_u32(uint32_t v) {
    v = udpp_max<0x111, 0xF>(v);
    v = udpp_max<0x112, 0xF>(v);
    v = udpp_max<0x114, 0xF>(v);
    v = udpp_max<0x118, 0xF>(v);
    v = udpp_max<0x142, 0xA>(v);
    v = udpp_max<0x143, 0xC>(v);
    return v;
}
#ifndef SJMI_PACK_MIN_STRINGS
#define SJMI_PACK_MIN_STRINGS 2
#endif
constexpr int PACK_MIN_STRINGS = SJMI_PACK_MIN_STRINGS;
constexpr uint32_t PACK_MAX_LEN = 256;  // longer strings keep the wave to themselves (four windows per round trip)

// Must be called by all 64 lanes.  take: this lane's string goes into the stream ([open + 1, close), 1..PACK_MAX_LEN
// bytes).  -> this lane's unescaped length or -(SJMI_E_* code) (0 if !take); the bytes go to scratch[open + 1 ...].
__device__ __forceinline__ int64_t unescape_packed(const uint8_t* __restrict__ buf, uint32_t open, uint32_t close, bool take,
                                                   uint8_t* __restrict__ scratch, int lane, uint32_t* __restrict__ lds) {
    uint32_t* const l_start = lds;        // [64] window position -> string + 1
    uint32_t* const l_len = lds + 64;     // [64] string -> bytes emitted so far
    uint32_t* const l_err = lds + 128;    // [64] string -> (lane << 8 | code) of its first error, ~0 = none
    const unsigned long long EVEN = 0x5555555555555555ull;
    const uint32_t slen = take ? close - open - 1u : 0u;
    const uint32_t voff_incl = wave_incl_scan_u32(slen);
    const uint32_t voff = voff_incl - slen;
    const uint32_t V = (uint32_t)__builtin_amdgcn_readlane((int)voff_incl, 63);
    l_len[lane] = 0;
    l_err[lane] = 0xFFFFFFFFu;
    unsigned long long pU = 0, pD1 = 0, pD2 = 0, pD3 = 0;  // bit 0: the previous window's last lane had U / D1 / D2 / D3
    unsigned long long prev_U = 0;
    uint32_t carry = 0, cur1 = 0;
    for (uint32_t w0 = 0; w0 < V; w0 += 64) {
        l_start[lane] = 0;
        wave_fence_lds();
        if (take && voff >= w0 && voff < w0 + 64u) l_start[voff - w0] = (uint32_t)lane + 1u;
        wave_fence_lds();
        const uint32_t sj = l_start[lane];
        uint32_t sidx = wave_incl_max_u32(sj);
        sidx = sidx > cur1 ? sidx : cur1;  // (no start in front of me in this window: the string that continues)
        const uint32_t v = w0 + (uint32_t)lane;
        const bool valid = v < V;
        const int j = (int)(sidx ? sidx - 1u : 0u);
        const uint32_t o_j = (uint32_t)__shfl((int)open, j), vo_j = (uint32_t)__shfl((int)voff, j);
        const uint32_t rel = v - vo_j, pos = o_j + 1u + rel;
        const uint32_t c = valid ? (uint32_t)buf[pos] : 0u;
        const uint32_t acc = l_len[j], e_prev = l_err[j];
        const unsigned long long S = __ballot(sj != 0);
        // (a backslash run that continues across a string boundary is left whole: the part in front of the boundary has
        //  even length -- its string ended at an unescaped quote -- so the part behind it is classified as if it stood alone)
        const unsigned long long B = __ballot(valid && c == '\\');
        const unsigned long long bs = B & ~(unsigned long long)carry;
        const unsigned long long follows = (bs << 1) | carry;
        const unsigned long long odd_starts = bs & ~EVEN & ~follows;
        const unsigned long long seq_even = odd_starts + bs;
        const uint32_t carry_out = seq_even < bs ? 1u : 0u;
        const unsigned long long escaped = (EVEN ^ (seq_even << 1)) & follows;
        const bool is_esc = valid && ((escaped >> lane) & 1ull);
        const bool is_start = ((B & ~escaped) >> lane) & 1ull;
        const unsigned long long U = __ballot(is_esc && c == 'u');
        const unsigned long long D1 = ((U << 1) | pU) & ~S, D2 = ((D1 << 1) | pD1) & ~S, D3 = ((D2 << 1) | pD2) & ~S,
                                 D4 = ((D3 << 1) | pD3) & ~S;
        const unsigned long long digits = D1 | D2 | D3 | D4;
        uint32_t outlen = 1, out = c, err = 0;
        if (!valid || is_start || ((digits >> lane) & 1ull)) {
            outlen = 0;
        } else if (is_esc) {
            if (c == 'u') {                                                      // StringParser.java:45-57
                int32_t cp = hex4_word(reinterpret_cast<const U4B*>(buf + pos + 1)->a);
                if (cp >= 0xD800 && cp <= 0xDBFF) {                              // parseLowSurrogate :112-124
                    const U8B t = *reinterpret_cast<const U8B*>(buf + pos + 5);
                    if ((t.a & 0xFFFFu) != (uint32_t)('\\' | ('u' << 8))) {
                        err = SJMI_E_LOW_SURROGATE_NO_U;
                    } else {
                        const int32_t low = hex4_word((t.a >> 16) | (t.b << 16)) - 0xDC00;
                        if ((low >> 10) != 0) err = SJMI_E_LOW_SURROGATE_RANGE;
                        else cp = (((cp - 0xD800) << 10) | low) + 0x10000;
                    }
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                    bool paired = false;
                    if (rel >= 6u) {  // (the six bytes in front are this string's: same stream, contiguous)
                        const bool prev_is_u = lane >= 6 ? ((U >> (lane - 6)) & 1ull) : ((prev_U >> (58 + lane)) & 1ull);
                        if (prev_is_u) {
                            const int32_t hi = hex4_word(reinterpret_cast<const U4B*>(buf + pos - 5)->a);
                            paired = hi >= 0xD800 && hi <= 0xDBFF;
                        }
                    }
                    if (paired) cp = -2;
                    else err = SJMI_E_LOW_SURROGATE_RESERVED;
                }
                if (!err) {
                    if (cp == -2) {
                        outlen = 0;
                    } else if (cp < 0) {
                        err = SJMI_E_INVALID_UNICODE_ESCAPE;                     // :127-129
                    } else if (cp <= 0x7F) {
                        out = (uint32_t)cp;
                    } else if (cp <= 0x7FF) {
                        outlen = 2;
                        out = (uint32_t)((cp >> 6) + 192) | ((uint32_t)((cp & 63) + 128) << 8);
                    } else if (cp <= 0xFFFF) {
                        outlen = 3;
                        out = (uint32_t)((cp >> 12) + 224) | ((uint32_t)(((cp >> 6) & 63) + 128) << 8) |
                              ((uint32_t)((cp & 63) + 128) << 16);
                    } else {
                        outlen = 4;
                        out = (uint32_t)((cp >> 18) + 240) | ((uint32_t)(((cp >> 12) & 63) + 128) << 8) |
                              ((uint32_t)(((cp >> 6) & 63) + 128) << 16) | ((uint32_t)((cp & 63) + 128) << 24);
                    }
                }
            } else {                                                             // :58-61
                const uint32_t r = (c & 0x80u) ? 0u : escape_map(c);
                if (r == 0) err = SJMI_E_ESCAPE_UNEXPECTED;
                out = r;
            }
        }
        // a string's first error = the lowest position: windows come in order, lanes by atomic min inside one
        if (err && e_prev == 0xFFFFFFFFu) atomicMin(&l_err[j], ((uint32_t)lane << 8) | err);
        const uint32_t incl = wave_incl_scan_u32(outlen), excl = incl - outlen;
        const uint32_t base = wave_incl_max_u32(sj ? excl : 0u);  // output position at which my string's part of the window begins
        uint8_t* q = scratch + o_j + 1u + acc + (excl - base);
        if (outlen >= 1) q[0] = (uint8_t)out;
        if (outlen >= 2) q[1] = (uint8_t)(out >> 8);
        if (outlen >= 3) q[2] = (uint8_t)(out >> 16);
        if (outlen >= 4) q[3] = (uint8_t)(out >> 24);
        uint32_t sidx_next = (uint32_t)__shfl_down((int)sidx, 1);
        if (lane == 63) sidx_next = 0;
        const bool last_of_string = valid && (v + 1u == V || sidx_next != sidx);
        if (last_of_string) l_len[j] = acc + (incl - base);
        cur1 = (uint32_t)__builtin_amdgcn_readlane((int)sidx, 63);
        pU = U >> 63;
        pD1 = D1 >> 63;
        pD2 = D2 >> 63;
        pD3 = D3 >> 63;
        prev_U = U;
        carry = carry_out;
        wave_fence_lds();
    }
    wave_fence_lds();
    const uint32_t e = l_err[lane], n = l_len[lane];
    if (!take) return 0;
    return e != 0xFFFFFFFFu ? -(int64_t)(e & 0xFFu) : (int64_t)n;
}

// any backslash in the aligned 16-byte chunks covering [from, to)?  (May look at up to 15 bytes on either side:
// a false positive only sends the string down the exact byte-wise path.)
__device__ __forceinline__ bool has_backslash(const uint8_t* __restrict__ buf, uint32_t from, uint32_t to) {
    uint32_t any = 0;
    for (uint32_t p = from & ~15u; p < to; p += 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(buf + p);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t z = w[i] ^ 0x5C5C5C5Cu;
            any |= ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;  // 0x80 where the byte is a backslash
        }
    }
    return any != 0;
}

constexpr uint32_t SIZE_SLOW = 0x80000000u;  // sizes[] flag: the string has escapes (lane-serial path)

// sizes[i] = 4 + unescaped length if structural i opens a string (| SIZE_SLOW if it has escapes), else 0;
// block_sums[tile] = the tile's bytes.
// A string with escapes is unescaped HERE, once: its bytes go to scratch[open + 1 ...] (a buffer as long as the
// document -- an unescaped string is never longer than its source), so that the write pass copies it like any other
// string, only from a different base.  A failed string has size 4 | SIZE_SLOW and its SJMI_E_* code in scratch[open].
//
// One workgroup per tile of 4096 structurals, 16 per lane, handled two at a time with all their loads in flight
// together: the first version (one structural per lane, one wave per 64) was bound by the workgroup dispatcher and by
// five dependent round trips per structural.  Now two: the structural and its successor, then the 16 bytes at the
// structural (is it a quote? backslashes in the head of the string) together with the 16 bytes in front of the
// successor (the closing quote is the last byte > 0x20 there: between it and the next structural there is only
// whitespace, or that byte would be a structural itself).
constexpr int MEAS_GROUP_MAX = 2;  // (4 costs a wave per SIMD in registers: 1.31 -> 1.16 ms with 2)
constexpr int SPAN_W = 2;  // 64-chunk words of the per-row backslash map: rows spanning up to SPAN_W KiB use it

struct MeasuredString {
    uint32_t close;  // position of the closing quote (0: not found)
    bool esc;        // may contain a backslash
};
// span_bs: one bit per aligned 16-byte chunk of the document from span_lo on (64 * SPAN_W chunks), set if the chunk
// holds a backslash; valid if span_ok (the wave's 64 structurals span no more than that)
__device__ __forceinline__ MeasuredString measure_string(const uint8_t* __restrict__ buf, uint32_t open, uint32_t bound,
                                                        uint32_t tpos, const U16B& hw, const U16B& tw, uint32_t span_lo,
                                                        bool span_ok, const unsigned long long span_bs[SPAN_W]) {
    MeasuredString m;
    const unsigned long long t0 = (unsigned long long)tw.a | ((unsigned long long)tw.b << 32);
    const unsigned long long t1 = (unsigned long long)tw.c | ((unsigned long long)tw.d << 32);
    const unsigned long long H = 0x8080808080808080ull, L = 0x7F7F7F7F7F7F7F7Full;
    const unsigned long long g0 = (((t0 & L) + 0x5F5F5F5F5F5F5F5Full) | t0) & H;  // 0x80 where the byte is > 0x20
    const unsigned long long g1 = (((t1 & L) + 0x5F5F5F5F5F5F5F5Full) | t1) & H;
    uint32_t last = 16;  // index of the last byte > 0x20 in the tail window (16 = none)
    if (g1) last = 15u - ((uint32_t)__builtin_clzll(g1) >> 3);
    else if (g0) last = 7u - ((uint32_t)__builtin_clzll(g0) >> 3);
    const uint32_t cpos = tpos + last;
    const uint32_t cbyte = last < 16 ? (uint32_t)((last < 8 ? t0 >> (8 * last) : t1 >> (8 * (last - 8))) & 0xFFu) : 0u;
    if (bound >= 16u && last < 16 && cbyte == '"' && cpos > open) {
        m.close = cpos;
        // backslashes: head and tail windows cover strings up to 30 bytes; the middle of longer ones is swept
        // with byte-granular 16-byte loads (false positives of neighbouring bytes only cost the exact path)
        const unsigned long long h0 = (unsigned long long)hw.a | ((unsigned long long)hw.b << 32);
        const unsigned long long h1 = (unsigned long long)hw.c | ((unsigned long long)hw.d << 32);
        unsigned long long any = 0;
        const unsigned long long z[4] = {h0 ^ 0x5C5C5C5C5C5C5C5Cull, h1 ^ 0x5C5C5C5C5C5C5C5Cull,
                                         t0 ^ 0x5C5C5C5C5C5C5C5Cull, t1 ^ 0x5C5C5C5C5C5C5C5Cull};
#pragma unroll
        for (int q = 0; q < 4; ++q) any |= ~(((z[q] & L) + L) | z[q]) & H;
        if (open + 16 < tpos) {  // the string has a middle: bytes [open + 16, tpos)
            if (span_ok) {
                // the wave swept the whole stretch of the document its 64 structurals cover (one coalesced load per
                // KiB, requested together with the head / tail windows): look the middle up in that map
                const uint32_t ca = (open + 16 - span_lo) >> 4, cb = (tpos - 1 - span_lo) >> 4;  // chunk range, inclusive
#pragma unroll
                for (int w = 0; w < SPAN_W; ++w) {
                    const uint32_t la = ca > 64u * w ? ca - 64u * w : 0u;
                    const uint32_t lb = cb < 64u * w + 63u ? cb - 64u * w : 63u;
                    if (ca <= 64u * w + 63u && cb >= 64u * w) any |= span_bs[w] & (~0ull >> (63u - lb)) & (~0ull << la);
                }
            } else {
                // (a wave whose structurals span more than that: four loads in flight per trip; a one-load loop costs a
                //  long string one memory round trip per 16 bytes, and the whole wave waits for its longest string)
                for (uint32_t p = open + 16; p < tpos; p += 64) {
                    U16B w[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const U16B*>(buf + (p + 16u * u < tpos ? p + 16u * u : tpos));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned long long m0 = ((unsigned long long)w[u].a | ((unsigned long long)w[u].b << 32)) ^ 0x5C5C5C5C5C5C5C5Cull;
                        const unsigned long long m1 = ((unsigned long long)w[u].c | ((unsigned long long)w[u].d << 32)) ^ 0x5C5C5C5C5C5C5C5Cull;
                        any |= (~(((m0 & L) + L) | m0) & H) | (~(((m1 & L) + L) | m1) & H);
                    }
                }
            }
        }
        m.esc = any != 0;
    } else {  // more than 15 bytes of whitespace after the string, or the very beginning of the document
        m.close = find_close(buf, open, bound);
        m.esc = m.close && has_backslash(buf, open + 1, m.close);
    }
    return m;
}

// Batches in isolated mode: a document that failed stage 1 contributes no structurals, so the text between the opening
// quote of a document's LAST string (a root-level string) and the next structural can hold whole dropped documents
// instead of white space only.  Such strings (marked by k_doc_mark_tails) end at their own document's end instead.
struct TailClip {
    const uint32_t* marks = nullptr;  // one bit per structural
    const unsigned long long* index_offsets = nullptr;
    const unsigned long long* doc_offsets = nullptr;
    uint64_t n_docs = 0;
};
__device__ __forceinline__ uint32_t tail_clip_bound(const TailClip& clip, uint64_t i) {
    uint64_t lo = 0, hi = clip.n_docs;  // the document k with index_offsets[k] <= i < index_offsets[k + 1]
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (clip.index_offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return (uint32_t)clip.doc_offsets[lo + 1];
}

// one thread per document: mark its last structural if that is a quote and the next document has no structurals
__global__ void __launch_bounds__(256)
k_doc_mark_tails(const uint8_t* __restrict__ buf, const uint32_t* __restrict__ idx,
                 const unsigned long long* __restrict__ index_offsets, uint64_t n_docs, uint32_t* __restrict__ marks,
                 const Stage1Result* __restrict__ dev_count) {
    if (dev_count && (dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL))) return;  // (incomplete index array)
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k + 1 >= n_docs) return;  // (the batch's last document ends where the batch ends)
    const unsigned long long from = index_offsets[k], to = index_offsets[k + 1];
    if (to == from || index_offsets[k + 2] != to) return;
    if (buf[idx[to - 1]] == '"') atomicOr(&marks[(to - 1) >> 5], 1u << ((to - 1) & 31));
}

template <int ITEMS>
// (7 waves per SIMD, 69 instead of 74 VGPRs with some SGPRs parked in VGPR lanes: -4 % on twitter, -6 % on the
// escape-heavy synthetic batches; half of the wave cycles here are waits, profiles/r1/batch_pmc.txt)
__global__ void __launch_bounds__(UNESC_THREADS) __attribute__((amdgpu_waves_per_eu(7, 7)))
k_str_measure(const uint8_t* __restrict__ buf, uint32_t len, const uint32_t* __restrict__ idx, uint64_t count,
              const Stage1Result* __restrict__ dev_count, uint32_t* __restrict__ sizes,
              unsigned long long* __restrict__ block_sums, uint32_t* __restrict__ group_sums, uint8_t* __restrict__ scratch,
              UnescapeResult* res, TailClip clip) {
    // dev_count != nullptr: the structural count is still on the device (stage 1 of the same document is queued right
    // in front); the grid was sized for an upper bound, surplus workgroups leave at once
    if (dev_count) {
        // (a stage 1 that ran out of index capacity or tripped a liveness bound left the index array incomplete:
        //  nothing may be dereferenced; the host reports SJMI_ERR_CAPACITY / SJMI_ERR_INTERNAL from the stage-1 record)
        if (dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL)) return;
        count = dev_count->count;
    }
    if ((uint64_t)blockIdx.x * (UNESC_THREADS * ITEMS) >= count) return;
    __shared__ unsigned long long s_part[UNESC_THREADS / 64];
    __shared__ uint32_t s_pack[UNESC_THREADS / 64][192];
    const int lane = threadIdx.x & 63;
    constexpr int MEAS_GROUP = ITEMS < MEAS_GROUP_MAX ? ITEMS : MEAS_GROUP_MAX;
    const uint32_t base32 = blockIdx.x * (uint32_t)(UNESC_THREADS * ITEMS);
    const uint32_t count32 = (uint32_t)count;
    unsigned long long sum = 0;
#pragma unroll 1
    for (int g = 0; g < ITEMS; g += MEAS_GROUP) {
        uint32_t open[MEAS_GROUP], bound[MEAS_GROUP], tpos[MEAS_GROUP];
        bool in_range[MEAS_GROUP];
#pragma unroll
        for (int q = 0; q < MEAS_GROUP; ++q) {
            const uint32_t i = base32 + (uint32_t)(g + q) * UNESC_THREADS + threadIdx.x;  // (< 2^32: one index per document byte at most)
            in_range[q] = i < count32;
            open[q] = in_range[q] ? idx[i] : 0u;
            bound[q] = (i + 1 < count32) ? idx[i + 1] : len;
            if (clip.marks && in_range[q] && ((clip.marks[i >> 5] >> (i & 31)) & 1u)) bound[q] = tail_clip_bound(clip, i);
        }
        U16B hw[MEAS_GROUP], tw[MEAS_GROUP];
        uint4 sp[MEAS_GROUP][SPAN_W];
        uint32_t span_lo[MEAS_GROUP];
        bool span_ok[MEAS_GROUP];
#pragma unroll
        for (int q = 0; q < MEAS_GROUP; ++q) {
            tpos[q] = bound[q] >= 16u ? bound[q] - 16u : 0u;  // (bound < 16: garbage below, fixed up by the slow path)
            hw[q] = *reinterpret_cast<const U16B*>(buf + open[q]);
            tw[q] = *reinterpret_cast<const U16B*>(buf + tpos[q]);
            // the stretch of the document this wave's 64 structurals cover (they are sorted): swept for backslashes
            // in aligned 16-byte chunks, lane j taking chunks j, j + 64, ... -- if it is no longer than SPAN_W KiB
            const unsigned long long rm = __ballot(in_range[q]);
            const int last_lane = rm ? 63 - __builtin_clzll(rm) : 0;
            span_lo[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)open[q]) & ~15u;
            const uint32_t span_hi = (uint32_t)__builtin_amdgcn_readlane((int)bound[q], last_lane);
            span_ok[q] = rm != 0 && span_hi - span_lo[q] <= 1024u * SPAN_W;
#pragma unroll
            for (int w = 0; w < SPAN_W; ++w) {
                const uint32_t p = span_lo[q] + 16u * (64u * w + (uint32_t)lane);
                sp[q][w] = (span_ok[q] && p < span_hi) ? *reinterpret_cast<const uint4*>(buf + p) : make_uint4(0, 0, 0, 0);
            }
        }
        unsigned long long span_bs[MEAS_GROUP][SPAN_W];
#pragma unroll
        for (int q = 0; q < MEAS_GROUP; ++q)
#pragma unroll
            for (int w = 0; w < SPAN_W; ++w) {
                const uint32_t x[4] = {sp[q][w].x ^ 0x5C5C5C5Cu, sp[q][w].y ^ 0x5C5C5C5Cu, sp[q][w].z ^ 0x5C5C5C5Cu,
                                       sp[q][w].w ^ 0x5C5C5C5Cu};
                uint32_t hit = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) hit |= ~(((x[t] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x[t]) & 0x80808080u;
                span_bs[q][w] = __ballot(hit != 0);
            }
        // (three passes over the group's items so that the window registers of the fast path are dead while the escaped
        //  strings are worked on: the kernel sits at its register budget)
        uint32_t close_[MEAS_GROUP];
        bool esc_[MEAS_GROUP], is_str_[MEAS_GROUP];
#pragma unroll
        for (int q = 0; q < MEAS_GROUP; ++q) {
            is_str_[q] = in_range[q] && (hw[q].a & 0xFFu) == '"';
            MeasuredString m = {0u, false};
            if (is_str_[q]) m = measure_string(buf, open[q], bound[q], tpos[q], hw[q], tw[q], span_lo[q], span_ok[q], span_bs[q]);
            close_[q] = m.close;
            esc_[q] = m.esc;
        }
        int64_t r_[MEAS_GROUP];
#pragma unroll
        for (int q = 0; q < MEAS_GROUP; ++q) {
            int64_t r = 0;
            if (is_str_[q]) {
                if (!close_[q]) r = -(int64_t)SJMI_E_INTERNAL;
                else if (!esc_[q]) r = (int64_t)(close_[q] - open[q] - 1);
            }
            // strings with escapes: the short ones together as one packed stream, the long ones one after the other
            bool packed = esc_[q] && close_[q] - open[q] - 1u <= PACK_MAX_LEN && close_[q] > open[q] + 1u;
            if (__popcll(__ballot(packed)) < PACK_MIN_STRINGS) packed = false;  // (few: a wave each costs less than the stream's bookkeeping)
            if (__ballot(packed)) {
                const int64_t rp = unescape_packed(buf, open[q], close_[q], packed, scratch, lane, s_pack[threadIdx.x >> 6]);
                if (packed) r = rp;
            }
            for (unsigned long long todo = __ballot(esc_[q] && !packed); todo; todo &= todo - 1) {
                const int j = __builtin_ctzll(todo);
                const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)open[q], j) + 1u;
                const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)close_[q], j);
                const int64_t rj = unescape_wave(buf, s0, e0, scratch + s0, lane);
                if (lane == j) r = rj;
            }
            r_[q] = r;
        }
#pragma unroll
        for (int q = 0; q < MEAS_GROUP; ++q) {
            const uint32_t i = base32 + (uint32_t)(g + q) * UNESC_THREADS + threadIdx.x;  // (< 2^32: one index per document byte at most)
            const bool is_str = is_str_[q];
            int64_t r = r_[q];
            uint32_t slow = (is_str && close_[q] && esc_[q]) ? SIZE_SLOW : 0u;
            if (is_str) {
                if (r == 0) slow = 0;  // (the backslash sweep may be a false positive of a neighbour: an empty string stays empty)
                if (r < 0) {
                    // first = lowest position: keep max of the complement so that the memset-to-zero state means "none"
                    atomicMax(reinterpret_cast<unsigned long long*>(&res->first_error_inv),
                              ~(((unsigned long long)i << 8) | (unsigned long long)(-r)));
                    scratch[open[q]] = (uint8_t)(-r);  // for the record FF FF FF <code> (the host stage 2 throws when it reaches it)
                    slow = SIZE_SLOW;
                    r = 0;
                }
                sizes[i] = (4u + (uint32_t)r) | slow;
                sum += 4u + (uint32_t)r;
            } else if (in_range[q]) {
                sizes[i] = 0;
            }
            if (group_sums) {  // batches: bytes per 64 structurals, for the per-document string-buffer offsets
                uint32_t gsum = is_str ? 4u + (uint32_t)r : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) gsum += __shfl_xor(gsum, d);
                if (lane == 0 && in_range[q]) group_sums[i >> 6] = gsum;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// exclusive scan of the workgroup sums by ONE workgroup (there are count/4096 of them); also the total
__global__ void __launch_bounds__(1024)
k_scan_sums(unsigned long long* __restrict__ block_sums, uint32_t nblocks, const Stage1Result* __restrict__ dev_count,
            uint32_t tile, UnescapeResult* res) {
    if (dev_count) {
        if (dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL)) return;  // (res stays zeroed: no records)
        nblocks = (uint32_t)((dev_count->count + tile - 1) / tile);
    }
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < nblocks ? block_sums[i] : 0ull;
        unsigned long long x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(x, d);
            if (lane >= d) x += t;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        unsigned long long off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (i < nblocks) block_sums[i] = off + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        res->total_bytes = s_carry;
        block_sums[nblocks] = s_carry;  // (one spare entry: the offset "behind the last tile")
    }
}

// ---- k_str_write ------------------------------------------------------------------------------
// OUTPUT-driven: the string buffer is produced in aligned 16-byte chunks, one lane per chunk, so every store is a
// full global_store_dwordx4 and the control flow does not depend on the string lengths.  A lane finds the record its
// chunk starts in (binary search over the pass's record table in LDS), then walks the records that overlap the
// chunk -- typically "rest of one string, next header, start of the next string" -- and shifts each piece (one
// byte-granular 16-byte load from the document, or from the scratch copy for a string that had escapes, or the
// big-endian length) into place.
// The lane-per-string versions before it were instruction-bound by divergence (8e8 VALU instructions and 5.7e7 store
// instructions per twitter x1024 launch: every length class, every 16-byte piece of a long string and every string
// with escapes was a separate pass of the whole wave).
constexpr int WSUB_ROWS_MAX = 8;  // records of up to 2048 structurals per pass
constexpr uint32_t REC_SCRATCH = 0x80000000u;    // rec_len flag: the bytes are in the scratch copy
constexpr uint32_t REC_FAILED = 0x40000000u;     // rec_len flag: failed string, low byte = SJMI_E_* code

// overwrite bytes [p, 16) of the 16-byte value (v0, v1) with (w0, w1) shifted up by p bytes; p in [-3, 15],
// negative p drops the first -p bytes of w
__device__ __forceinline__ void insert_at(unsigned long long& v0, unsigned long long& v1, unsigned long long w0,
                                          unsigned long long w1, int p) {
    if (p < 0) {  // (only the 4-byte header starts before the chunk: w1 == 0)
        w0 >>= 8 * (-p);
        p = 0;
    }
    const uint32_t s = 8u * ((uint32_t)p & 7u);
    const unsigned long long keep = (1ull << s) - 1ull;      // s == 0: nothing kept
    const unsigned long long spill = s ? (w0 >> (64u - s)) : 0ull;
    if (p < 8) {
        v0 = (v0 & keep) | (w0 << s);
        v1 = (w1 << s) | spill;
    } else {
        v1 = (v1 & keep) | (w0 << s);
    }
}

// (latency-bound: 6 waves per SIMD instead of 5 -- 78 instead of 85 VGPRs, no spills -- are worth 5 %)
template <int ITEMS>
__global__ void __launch_bounds__(UNESC_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_str_write(const uint8_t* __restrict__ buf, uint32_t len, const uint32_t* __restrict__ idx, uint64_t count,
            const Stage1Result* __restrict__ dev_count, const uint32_t* __restrict__ sizes, const unsigned long long* __restrict__ block_offsets,
            const uint8_t* __restrict__ scratch, uint8_t* __restrict__ sb, uint64_t sb_cap, UnescapeResult* res) {
    constexpr int WSUB_ROWS = ITEMS < WSUB_ROWS_MAX ? ITEMS : WSUB_ROWS_MAX;
    constexpr int WSUB = WSUB_ROWS * UNESC_THREADS;
    __shared__ uint32_t s_wave[ITEMS][UNESC_THREADS / 64];  // bytes per (row, wave)
    __shared__ uint32_t s_cnt[ITEMS][UNESC_THREADS / 64];   // strings per (row, wave)
    __shared__ uint32_t rec_d[WSUB], rec_src[WSUB], rec_len[WSUB];
    if (dev_count) {
        if (dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL)) return;
        count = dev_count->count;
    }
    const uint32_t nblocks_w = (uint32_t)((count + (uint64_t)UNESC_THREADS * ITEMS - 1) / ((uint64_t)UNESC_THREADS * ITEMS));
    if (blockIdx.x >= nblocks_w) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint64_t base = (uint64_t)blockIdx.x * (UNESC_THREADS * ITEMS);
    // bytes and strings per (row, wave) of the whole tile; the sizes themselves are read again pass by pass (they
    // would cost 32 VGPRs if kept)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint64_t i = base + (uint64_t)k * UNESC_THREADS + threadIdx.x;
        uint32_t x = (i < count ? sizes[i] : 0u) & ~SIZE_SLOW;
        const unsigned long long sm = __ballot(x != 0);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 63) {
            s_wave[k][wave] = x;
            s_cnt[k][wave] = (uint32_t)__popcll(sm);
        }
    }
    __syncthreads();
    unsigned long long row = block_offsets[blockIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == nblocks_w - 1) {  // (string buffer too small: reported once, by the last tile)
        unsigned long long end = row;
        for (int k = 0; k < ITEMS; ++k)
            for (int w = 0; w < UNESC_THREADS / 64; ++w) end += s_wave[k][w];
        if (end > sb_cap) atomicOr(&res->flags, 1u);
    }
#pragma unroll 1
    for (int k0 = 0; k0 < ITEMS; k0 += WSUB_ROWS) {
        // ---- the records of this pass, in order, as a table in LDS ----
        const unsigned long long D0 = row;  // where this pass's records start in the string buffer
        uint32_t nrec = 0;
        uint32_t opens[WSUB_ROWS], sz[WSUB_ROWS];  // (requested together: one memory round trip per pass, not per row)
#pragma unroll
        for (int kk = 0; kk < WSUB_ROWS; ++kk) {
            const uint64_t i = base + (uint64_t)(k0 + kk) * UNESC_THREADS + threadIdx.x;
            opens[kk] = i < count ? idx[i] : 0u;
            sz[kk] = i < count ? sizes[i] : 0u;
        }
#pragma unroll
        for (int kk = 0; kk < WSUB_ROWS; ++kk) {
            const int k = k0 + kk;
            unsigned long long off = row;
            uint32_t rowsum = 0, r = nrec;
#pragma unroll
            for (int w = 0; w < UNESC_THREADS / 64; ++w) {
                if (w < wave) {
                    off += s_wave[k][w];
                    r += s_cnt[k][w];
                }
                rowsum += s_wave[k][w];
                nrec += s_cnt[k][w];
            }
            row += rowsum;
            const uint32_t size = sz[kk] & ~SIZE_SLOW;
            const bool slow = (sz[kk] & SIZE_SLOW) != 0;
            {
                uint32_t x = size;  // inclusive scan inside the wave
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t = __shfl_up(x, d);
                    if (lane >= d) x += t;
                }
                off += x - size;
            }
            const bool isrec = size != 0;
            r += (uint32_t)__popcll(__ballot(isrec) & lt_mask);
            if (isrec) {
                const uint32_t open = opens[kk];
                const uint32_t n = size - 4u;
                rec_d[r] = (uint32_t)(off - D0);
                rec_src[r] = open + 1u;
                rec_len[r] = (slow && n == 0) ? (REC_FAILED | scratch[open]) : (n | (slow ? REC_SCRATCH : 0u));
            }
        }
        __syncthreads();
        // ---- the pass's part of the string buffer, [D0, D1), in aligned 16-byte chunks ----
        const unsigned long long D1 = row < sb_cap ? row : sb_cap;
        for (unsigned long long glo = (D0 & ~15ull) + 16ull * threadIdx.x; glo < D1 && nrec != 0; glo += 16ull * UNESC_THREADS) {
            const unsigned long long lo = glo > D0 ? glo : D0;
            const unsigned long long hi = glo + 16 < D1 ? glo + 16 : D1;
            const uint32_t rel = (uint32_t)(lo - D0);
            uint32_t r = 0, span = nrec;
            while (span > 1) {  // last record that starts at or before `lo` (uniform trip count)
                const uint32_t half = span >> 1;
                if (rec_d[r + half] <= rel) r += half;
                span -= half;
            }
            // A 16-byte chunk overlaps at most 5 records (a record has at least 4 bytes).  First the records are
            // looked up and their source bytes requested, then everything is shifted into place: a loop that loads and
            // inserts record by record pays one memory round trip per record.
            unsigned long long v0 = 0, v1 = 0;
            // (two rounds: up to 3 records -- all that most chunks touch -- and, only if some lane of the wave needs
            //  them, the remaining 2)
            for (uint32_t r0 = r, round = 0; round < 2; r0 += 3, ++round) {
                constexpr int NREC = 3;
                bool hdr_on[NREC], dat_on[NREC];
                int hdr_p[NREC], dat_p[NREC];
                uint32_t hdr_v[NREC];
                U16B w[NREC];
#pragma unroll
                for (int j = 0; j < NREC; ++j) {
                    const uint32_t rr = r0 + j < nrec ? r0 + j : nrec - 1;
                    const unsigned long long rd = D0 + rec_d[rr];
                    const bool on = r0 + j < nrec && rd < hi && (round == 0 || j < 2);
                    const uint32_t L = rec_len[rr];
                    const uint32_t n = (L & REC_FAILED) ? 0u : (L & ~REC_SCRATCH);
                    hdr_v[j] = __builtin_bswap32((L & REC_FAILED) ? (0xFFFFFF00u | (L & 0xFFu)) : n);  // IntegerUtils.toBytes :12-17: big endian
                    hdr_on[j] = on && rd + 4 > lo;
                    hdr_p[j] = (int)((long long)rd - (long long)glo);
                    const unsigned long long g0 = rd + 4 > lo ? rd + 4 : lo, g1 = rd + 4 + n < hi ? rd + 4 + n : hi;
                    dat_on[j] = on && g0 < g1;
                    dat_p[j] = (int)(g0 - glo);
                    const uint8_t* from = ((L & REC_SCRATCH) ? scratch : buf) + rec_src[rr] + (uint32_t)(g0 - (rd + 4));
                    if (dat_on[j]) w[j] = *reinterpret_cast<const U16B*>(from);
                }
#pragma unroll
                for (int j = 0; j < NREC; ++j) {
                    if (hdr_on[j]) insert_at(v0, v1, (unsigned long long)hdr_v[j], 0ull, hdr_p[j]);
                    if (dat_on[j])
                        insert_at(v0, v1, (unsigned long long)w[j].a | ((unsigned long long)w[j].b << 32),
                                  (unsigned long long)w[j].c | ((unsigned long long)w[j].d << 32), dat_p[j]);
                }
                // a fourth record can only overlap if the third one did
                if (!__ballot(hdr_on[NREC - 1] || dat_on[NREC - 1])) break;
            }
            if (lo == glo && hi == glo + 16) {
                *reinterpret_cast<uint4*>(sb + glo) = make_uint4((uint32_t)v0, (uint32_t)(v0 >> 32), (uint32_t)v1, (uint32_t)(v1 >> 32));
            } else {  // chunk shared with another pass / workgroup (or cut by the capacity): byte stores
                for (uint32_t j = (uint32_t)(lo - glo); j < (uint32_t)(hi - glo); ++j)
                    sb[glo + j] = (uint8_t)((j < 8 ? v0 >> (8 * j) : v1 >> (8 * (j - 8))));
            }
        }
        __syncthreads();  // the record table is rebuilt by the next pass
    }
}

// Batches: doc_str_offsets[k] = string-buffer offset of document k's first record = bytes of all records whose
// structural comes before index_offsets[k] (n_docs + 1 entries; the host stage 2 of every document can then start at
// its own offset, so documents can be walked in parallel).  One wave per boundary: tile offset + the tile's full
// groups of 64 structurals before it + the sizes of the partial group.
__global__ void __launch_bounds__(256)
k_doc_str_offsets(const unsigned long long* __restrict__ index_offsets, uint64_t n_docs, uint64_t count,
                  const uint32_t* __restrict__ sizes, const uint32_t* __restrict__ group_sums,
                  const unsigned long long* __restrict__ block_offsets, uint32_t tile,
                  unsigned long long* __restrict__ doc_str_offsets, const Stage1Result* __restrict__ dev_count) {
    const int lane = threadIdx.x & 63;
    const uint64_t k = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k > n_docs) return;
    if (dev_count) {
        if (dev_count->status & (SJMI_ST_CAPACITY | SJMI_ST_INTERNAL)) {  // (no records were made)
            if (lane == 0) doc_str_offsets[k] = 0;
            return;
        }
        count = dev_count->count;
    }
    unsigned long long s = index_offsets[k];
    if (s > count) s = count;
    const uint64_t t = s / tile, tstart = t * tile;
    unsigned long long acc = 0;
    const uint64_t g0 = tstart >> 6, g1 = s >> 6;  // full groups of the tile in front of s (at most tile / 64 <= 64)
    if (g0 + (uint64_t)lane < g1) acc += group_sums[g0 + lane];
    if ((uint64_t)lane < (s & 63)) acc += sizes[(s & ~63ull) + lane] & ~SIZE_SLOW;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) doc_str_offsets[k] = block_offsets[t] + acc;
}

// layout: sizes[count] | block sums (sized for the smallest tile) | group sums[count / 64] | tail marks[count / 32] |
// scratch[len + 128]
static int unescape_items(uint64_t count) { return count >= (1u << 20) ? UNESC_ITEMS_MAX : (count >= (1u << 17) ? 4 : 1); }
static size_t ws_sums_offset(uint64_t count) { return (((size_t)count * sizeof(uint32_t) + 63) / 64) * 64 + 64; }
static size_t ws_groups_offset(uint64_t count) {
    const uint64_t nblocks = (count + UNESC_THREADS - 1) / UNESC_THREADS;
    return ws_sums_offset(count) + (((size_t)(nblocks + 2) * sizeof(unsigned long long) + 63) / 64) * 64 + 64;
}
static size_t ws_marks_offset(uint64_t count) {
    return ws_groups_offset(count) + (((size_t)(count / 64 + 2) * sizeof(uint32_t) + 63) / 64) * 64 + 64;
}
static size_t ws_marks_bytes(uint64_t count) { return (((size_t)(count / 32 + 2) * sizeof(uint32_t) + 63) / 64) * 64; }
static size_t ws_scratch_offset(uint64_t count) { return ws_marks_offset(count) + ws_marks_bytes(count) + 64; }
size_t unescape_workspace_bytes(uint64_t count, uint64_t len) { return ws_scratch_offset(count) + (size_t)len + 128; }

void unescape_records(void* d_ws, uint64_t count_bound, const uint32_t** sizes, const uint8_t** scratch) {
    *sizes = reinterpret_cast<const uint32_t*>(d_ws);
    *scratch = static_cast<const uint8_t*>(d_ws) + ws_scratch_offset(count_bound);
}

template <int ITEMS>
static hipError_t unescape_launch_items(const uint8_t* d_buf, uint64_t len, const uint32_t* d_idx, uint64_t count,
                                        const Stage1Result* dev_count, uint8_t* d_sb, uint64_t sb_cap, uint32_t* sizes,
                                        unsigned long long* sums, uint32_t* groups, uint32_t* marks, uint8_t* scratch,
                                        UnescapeResult* d_res, hipStream_t stream, const UnescapeBatch& batch) {
    const uint64_t tile = (uint64_t)UNESC_THREADS * ITEMS;
    const uint64_t nblocks = (count + tile - 1) / tile;  // (an upper bound if the count is still on the device)
    TailClip clip;
    if (batch.d_index_offsets && batch.d_doc_offsets && batch.n_docs > 1) {
        hipError_t e = hipMemsetAsync(marks, 0, ws_marks_bytes(count), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_doc_mark_tails, dim3((unsigned)((batch.n_docs + 255) / 256)), dim3(256), 0, stream, d_buf, d_idx,
                           batch.d_index_offsets, batch.n_docs, marks, dev_count);
        clip.marks = marks;
        clip.index_offsets = batch.d_index_offsets;
        clip.doc_offsets = batch.d_doc_offsets;
        clip.n_docs = batch.n_docs;
    }
    hipLaunchKernelGGL((k_str_measure<ITEMS>), dim3((unsigned)nblocks), dim3(UNESC_THREADS), 0, stream, d_buf, (uint32_t)len,
                       d_idx, count, dev_count, sizes, sums, batch.d_doc_str_offsets ? groups : nullptr, scratch, d_res, clip);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, sums, (uint32_t)nblocks, dev_count, (uint32_t)tile, d_res);
    hipLaunchKernelGGL((k_str_write<ITEMS>), dim3((unsigned)nblocks), dim3(UNESC_THREADS), 0, stream, d_buf, (uint32_t)len,
                       d_idx, count, dev_count, sizes, sums, scratch, d_sb, sb_cap, d_res);
    if (batch.d_doc_str_offsets)
        hipLaunchKernelGGL(k_doc_str_offsets, dim3((unsigned)((batch.n_docs + 1 + 3) / 4)), dim3(256), 0, stream,
                           batch.d_index_offsets, batch.n_docs, count, sizes, groups, sums, (uint32_t)tile,
                           batch.d_doc_str_offsets, dev_count);
    return hipGetLastError();
}

// count_bound = the structural count, or -- with dev_count -- an upper bound of it (the workspace and the grids are
// sized for the bound; the kernels take the real count from *dev_count)
hipError_t unescape_launch(const uint8_t* d_buf, uint64_t len, const uint32_t* d_idx, uint64_t count_bound,
                           const Stage1Result* dev_count, uint8_t* d_sb, uint64_t sb_cap, void* d_ws, UnescapeResult* d_res,
                           hipStream_t stream, const UnescapeBatch& batch) {
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    uint32_t* sizes = reinterpret_cast<uint32_t*>(ws);
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(ws + ws_sums_offset(count_bound));
    uint32_t* groups = reinterpret_cast<uint32_t*>(ws + ws_groups_offset(count_bound));
    uint32_t* marks = reinterpret_cast<uint32_t*>(ws + ws_marks_offset(count_bound));
    uint8_t* scratch = ws + ws_scratch_offset(count_bound);
    hipError_t e = hipMemsetAsync(d_res, 0, sizeof(UnescapeResult), stream);
    if (e != hipSuccess) return e;
    if (count_bound == 0) {
        if (batch.d_doc_str_offsets)
            return hipMemsetAsync(batch.d_doc_str_offsets, 0, (batch.n_docs + 1) * sizeof(unsigned long long), stream);
        return hipSuccess;
    }
    // (with a bound, judge the size by the document: ~one structural per 8-11 bytes)
    switch (unescape_items(dev_count ? len / 8 : count_bound)) {
    case 1: return unescape_launch_items<1>(d_buf, len, d_idx, count_bound, dev_count, d_sb, sb_cap, sizes, sums, groups, marks, scratch, d_res, stream, batch);
    case 4: return unescape_launch_items<4>(d_buf, len, d_idx, count_bound, dev_count, d_sb, sb_cap, sizes, sums, groups, marks, scratch, d_res, stream, batch);
    default: return unescape_launch_items<UNESC_ITEMS_MAX>(d_buf, len, d_idx, count_bound, dev_count, d_sb, sb_cap, sizes, sums, groups, marks, scratch, d_res, stream, batch);
    }
}

}  // namespace sjmi
