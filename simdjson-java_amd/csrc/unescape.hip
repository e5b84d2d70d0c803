// unescape.hip -- batched string unescape on the GPU.
//
// Replaces the per-string calls StringParser.parseString(buf, idx, stringBuffer, stringBufferIdx)
// (/root/reference/src/main/java/org/simdjson/StringParser.java:18-68, helpers :112-161,
//  CharacterUtils.escape :52-83, hexToInt :241-247, IntegerUtils.toBytes :12-17) that the reference's
// sequential stage 2 issues once per '"' structural (TapeBuilder.visitString, TapeBuilder.java:174-177).
// In a valid document the strings are visited in structural order, so record k starts at
// sum_{j<k}(4 + len_j): that is an exclusive prefix sum, and every string is independent.
//
// Four launches over the structural indexes produced by stage 1:
//   k_str_measure : one lane per structural; a lane whose byte is '"' finds the closing quote (the
//                   last non-whitespace byte before the next structural -- stage 1 already proved it
//                   exists) and sweeps the string with aligned 16-byte loads for a backslash.  No
//                   backslash (98 % of twitter.json's strings): length = raw length, done.  Otherwise
//                   the lane parses the escapes byte by byte; the first failing string (lowest
//                   structural position) is kept with an atomicMax of the complement;
//   k_block_sums / k_scan_sums : reduce-then-scan of the 4+len sizes, 4096 structurals per workgroup;
//   k_str_write   : re-derives each lane's offset (LDS scan + workgroup base); escape-free strings are
//                   copied by the whole wave (coalesced byte lanes, one string after the other), strings
//                   with escapes by their own lane -- [be32 length][bytes], bit-identical to the
//                   reference's stringBuffer.
// Output parity domain: string_buffer[0, total).  With an erroneous string the reference throws at that
// string; here every other string is still written, the failing one becomes a 4-byte record FF FF FF <code>
// (the host stage 2 throws when it reaches it) and the first one is also reported by position + code.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1.h"

namespace sjmi {

constexpr int UNESC_THREADS = 256;
constexpr int UNESC_ITEMS = 16;                               // structurals per lane
constexpr int UNESC_TILE = UNESC_THREADS * UNESC_ITEMS;       // 4096 structurals per workgroup

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

// CharacterUtils.hexToInt (CharacterUtils.java:241-247): negative if any of the 4 digits is bad
__device__ __forceinline__ int32_t hex4(const uint8_t* p) {
    int32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = p[i];
        int32_t d;
        if (c - '0' <= 9u) d = (int32_t)(c - '0');
        else if ((c | 0x20u) - 'a' <= 5u) d = (int32_t)((c | 0x20u) - 'a' + 10);
        else return -1;
        v = (v << 4) | d;
    }
    return v;
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

// One pass of StringParser.doParseString (StringParser.java:29-68) over [open+1, close).
// WRITE = false: only count.  Returns the unescaped length or -(error code).
template <bool WRITE>
__device__ __forceinline__ int64_t unescape_one(const uint8_t* __restrict__ buf, uint32_t open, uint32_t close,
                                                uint8_t* __restrict__ dst) {
    uint32_t src = open + 1;
    uint32_t n = 0;
    while (src < close) {
        // plain run: one aligned 8-byte load per window instead of a dependent load per byte
        const uint32_t a = src & ~7u;
        const unsigned long long w = *reinterpret_cast<const unsigned long long*>(buf + a);
        const uint32_t lo = src - a;
        const uint32_t hi = (close - a) < 8u ? (close - a) : 8u;
        const unsigned long long z = w ^ 0x5C5C5C5C5C5C5C5Cull;
        unsigned long long f = ~(((z & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | z) & 0x8080808080808080ull;
        f &= ~0ull << (8 * lo);  // backslashes at or after src
        uint32_t stop = f ? (uint32_t)__builtin_ctzll(f) >> 3 : 8u;
        if (stop > hi) stop = hi;
        for (uint32_t b = lo; b < stop; ++b) {
            if (WRITE) dst[n] = (uint8_t)(w >> (8 * b));
            ++n;
        }
        src = a + stop;
        if (stop >= hi) continue;  // window (or string) exhausted without an escape
        const uint32_t e = buf[src + 1];
        if (e == 'u') {                                                   // :45-57
            int32_t cp = hex4(buf + src + 2);
            src += 6;
            if (cp >= 0xD800 && cp <= 0xDBFF) {                           // parseLowSurrogate :112-124
                if (!(buf[src] == '\\' && buf[src + 1] == 'u')) return -(int64_t)SJMI_E_LOW_SURROGATE_NO_U;
                const int32_t low = hex4(buf + src + 2) - 0xDC00;
                if ((low >> 10) != 0) return -(int64_t)SJMI_E_LOW_SURROGATE_RANGE;
                cp = (((cp - 0xD800) << 10) | low) + 0x10000;
                src += 6;
            } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                return -(int64_t)SJMI_E_LOW_SURROGATE_RESERVED;           // :53-55
            }
            if (cp < 0) return -(int64_t)SJMI_E_INVALID_UNICODE_ESCAPE;   // storeCodePointInStringBuffer :127-129
            if (cp <= 0x7F) {
                if (WRITE) dst[n] = (uint8_t)cp;
                n += 1;
            } else if (cp <= 0x7FF) {
                if (WRITE) {
                    dst[n] = (uint8_t)((cp >> 6) + 192);
                    dst[n + 1] = (uint8_t)((cp & 63) + 128);
                }
                n += 2;
            } else if (cp <= 0xFFFF) {
                if (WRITE) {
                    dst[n] = (uint8_t)((cp >> 12) + 224);
                    dst[n + 1] = (uint8_t)(((cp >> 6) & 63) + 128);
                    dst[n + 2] = (uint8_t)((cp & 63) + 128);
                }
                n += 3;
            } else {
                if (WRITE) {
                    dst[n] = (uint8_t)((cp >> 18) + 240);
                    dst[n + 1] = (uint8_t)(((cp >> 12) & 63) + 128);
                    dst[n + 2] = (uint8_t)(((cp >> 6) & 63) + 128);
                    dst[n + 3] = (uint8_t)((cp & 63) + 128);
                }
                n += 4;
            }
        } else {                                                          // :58-61
            const uint32_t r = (e & 0x80u) ? 0u : escape_map(e);
            if (r == 0) return -(int64_t)SJMI_E_ESCAPE_UNEXPECTED;
            if (WRITE) dst[n] = (uint8_t)r;
            ++n;
            src += 2;
        }
    }
    return (int64_t)n;
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

// sizes[i] = 4 + unescaped length if structural i opens a string (| SIZE_SLOW if it has escapes), else 0
__global__ void __launch_bounds__(UNESC_THREADS)
k_str_measure(const uint8_t* __restrict__ buf, uint32_t len, const uint32_t* __restrict__ idx, uint64_t count,
              uint32_t* __restrict__ sizes, UnescapeResult* res) {
    const uint64_t i = (uint64_t)blockIdx.x * UNESC_THREADS + threadIdx.x;
    if (i >= count) return;
    const uint32_t open = idx[i];
    uint32_t size = 0;
    if (buf[open] == '"') {
        const uint32_t bound = (i + 1 < count) ? idx[i + 1] : len;
        const uint32_t close = find_close(buf, open, bound);
        uint32_t slow = 0;
        int64_t r;
        if (!close) r = -(int64_t)SJMI_E_INTERNAL;
        else if (!has_backslash(buf, open + 1, close)) r = (int64_t)(close - open - 1);
        else {
            slow = SIZE_SLOW;
            r = unescape_one<false>(buf, open, close, nullptr);
        }
        if (r < 0) {
            // first = lowest position: keep max of the complement so that the memset-to-zero state means "none"
            atomicMax(reinterpret_cast<unsigned long long*>(&res->first_error_inv),
                      ~(((unsigned long long)i << 8) | (unsigned long long)(-r)));
            r = 0;
        }
        size = (4u + (uint32_t)r) | slow;
    }
    sizes[i] = size;
}

__global__ void __launch_bounds__(UNESC_THREADS)
k_block_sums(const uint32_t* __restrict__ sizes, uint64_t count, unsigned long long* __restrict__ block_sums) {
    __shared__ unsigned long long s_part[UNESC_THREADS / 64];
    const uint64_t base = (uint64_t)blockIdx.x * UNESC_TILE;
    unsigned long long sum = 0;
#pragma unroll
    for (int k = 0; k < UNESC_ITEMS; ++k) {
        const uint64_t i = base + (uint64_t)k * UNESC_THREADS + threadIdx.x;
        if (i < count) sum += sizes[i] & ~SIZE_SLOW;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// exclusive scan of the workgroup sums by ONE workgroup (there are count/4096 of them); also the total
__global__ void __launch_bounds__(1024)
k_scan_sums(unsigned long long* __restrict__ block_sums, uint32_t nblocks, UnescapeResult* res) {
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
    if (threadIdx.x == 0) res->total_bytes = s_carry;
}

__global__ void __launch_bounds__(UNESC_THREADS)
k_str_write(const uint8_t* __restrict__ buf, uint32_t len, const uint32_t* __restrict__ idx, uint64_t count,
            const uint32_t* __restrict__ sizes, const unsigned long long* __restrict__ block_offsets,
            uint8_t* __restrict__ sb, uint64_t sb_cap, UnescapeResult* res) {
    __shared__ uint32_t s_wave[UNESC_ITEMS][UNESC_THREADS / 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t base = (uint64_t)blockIdx.x * UNESC_TILE;
    uint32_t sz[UNESC_ITEMS], incl[UNESC_ITEMS];
#pragma unroll
    for (int k = 0; k < UNESC_ITEMS; ++k) {
        const uint64_t i = base + (uint64_t)k * UNESC_THREADS + threadIdx.x;
        sz[k] = i < count ? sizes[i] : 0u;
        uint32_t x = sz[k] & ~SIZE_SLOW;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(x, d);
            if (lane >= d) x += t;
        }
        incl[k] = x;
        if (lane == 63) s_wave[k][wave] = x;
    }
    __syncthreads();
    unsigned long long row = block_offsets[blockIdx.x];
#pragma unroll 1
    for (int k = 0; k < UNESC_ITEMS; ++k) {
        unsigned long long off = row;
        uint32_t rowsum = 0;
#pragma unroll
        for (int w = 0; w < UNESC_THREADS / 64; ++w) {
            if (w < wave) off += s_wave[k][w];
            rowsum += s_wave[k][w];
        }
        row += rowsum;
        const uint32_t size = sz[k] & ~SIZE_SLOW;
        const bool slow = (sz[k] & SIZE_SLOW) != 0;
        off += incl[k] - size;
        bool active = size != 0;
        if (active && off + size > sb_cap) {
            atomicOr(&res->flags, 1u);  // string buffer too small
            active = false;
        }
        const uint64_t i = base + (uint64_t)k * UNESC_THREADS + threadIdx.x;
        uint32_t open = 0, n = 0;
        if (active) {
            open = idx[i];
            n = size - 4u;
            uint8_t* dst = sb + off;
            uint32_t hdr = n;
            if (slow || n == 0) {  // escapes, empty or failed: the owning lane does everything
                const uint32_t bound = (i + 1 < count) ? idx[i + 1] : len;
                const uint32_t close = find_close(buf, open, bound);
                if (n == 0) {  // empty or failed string: a failed one is marked FF FF FF <code> for the host stage 2
                    const int64_t r = close ? unescape_one<false>(buf, open, close, nullptr) : -(int64_t)SJMI_E_INTERNAL;
                    if (r < 0) hdr = 0xFFFFFF00u | (uint32_t)(-r);
                } else if (close) {
                    unescape_one<true>(buf, open, close, dst + 4);
                }
            }
            dst[0] = (uint8_t)(hdr >> 24);  // IntegerUtils.toBytes :12-17
            dst[1] = (uint8_t)(hdr >> 16);
            dst[2] = (uint8_t)(hdr >> 8);
            dst[3] = (uint8_t)hdr;
        }
        // escape-free strings: the wave copies them, 64 consecutive bytes per instruction, EIGHT strings per
        // round so that eight independent loads are in flight before the first store (a one-string-at-a-time
        // loop pays a full memory round trip per string)
        unsigned long long m = __ballot(active && !slow && n != 0);
        const uint32_t off_lo = (uint32_t)off, off_hi = (uint32_t)(off >> 32);
        while (m) {
            uint32_t src[8], cnt[8];
            unsigned long long d[8];
            uint8_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                cnt[q] = 0;
                src[q] = 0;
                d[q] = 0;
                if (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    src[q] = (uint32_t)__builtin_amdgcn_readlane((int)open, j) + 1u;
                    cnt[q] = (uint32_t)__builtin_amdgcn_readlane((int)n, j);
                    d[q] = (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)off_hi, j) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)off_lo, j)) + 4ull;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (uint32_t)lane < cnt[q] ? buf[src[q] + lane] : (uint8_t)0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if ((uint32_t)lane < cnt[q]) sb[d[q] + lane] = v[q];
#pragma unroll
            for (int q = 0; q < 8; ++q)  // rare: strings longer than 64 bytes
                for (uint32_t t = 64 + lane; t < cnt[q]; t += 64) sb[d[q] + t] = buf[src[q] + t];
        }
    }
}

size_t unescape_workspace_bytes(uint64_t count) {
    const uint64_t nblocks = (count + UNESC_TILE - 1) / UNESC_TILE;
    return 64 + (size_t)count * sizeof(uint32_t) + 64 + (size_t)(nblocks + 1) * sizeof(unsigned long long);
}

hipError_t unescape_launch(const uint8_t* d_buf, uint64_t len, const uint32_t* d_idx, uint64_t count, uint8_t* d_sb,
                           uint64_t sb_cap, void* d_ws, UnescapeResult* d_res, hipStream_t stream) {
    const uint64_t nblocks = (count + UNESC_TILE - 1) / UNESC_TILE;
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    uint32_t* sizes = reinterpret_cast<uint32_t*>(ws);
    unsigned long long* sums =
        reinterpret_cast<unsigned long long*>(ws + (((size_t)count * sizeof(uint32_t) + 63) / 64) * 64 + 64);
    hipError_t e = hipMemsetAsync(d_res, 0, sizeof(UnescapeResult), stream);
    if (e != hipSuccess) return e;
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_str_measure, dim3((unsigned)((count + UNESC_THREADS - 1) / UNESC_THREADS)), dim3(UNESC_THREADS), 0,
                       stream, d_buf, (uint32_t)len, d_idx, count, sizes, d_res);
    hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nblocks), dim3(UNESC_THREADS), 0, stream, sizes, count, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, sums, (uint32_t)nblocks, d_res);
    hipLaunchKernelGGL(k_str_write, dim3((unsigned)nblocks), dim3(UNESC_THREADS), 0, stream, d_buf, (uint32_t)len, d_idx,
                       count, sizes, sums, d_sb, sb_cap, d_res);
    return hipGetLastError();
}

}  // namespace sjmi
