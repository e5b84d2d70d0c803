// SPDX-License-Identifier: Apache-2.0
// Stage 2 on the GPU for batches of small documents: JsonIterator.walkDocument
// (JsonIterator.java:26-200) driving TapeBuilder (TapeBuilder.java:41-217) -- the same state machine the host mirror
// runs (csrc/host/simdjson_parser.cpp, DocWalker::walkDocument), one LANE per document.
//
// Why a lane per document: the walk of one document is a sequential automaton over its structurals (container stack,
// element counts, "what may follow what"), but the documents of a batch are independent, and a batch of ~1 KB documents
// has a million of them.  Every lane reads its document's structurals in order, checks the grammar exactly as the
// reference does (so the FIRST error of a document is the reference's error), parses atoms and numbers
// (NumberParser.java:23-74, ExponentParser.java:14-69) and writes tape words (Tape.java:28-47) into its own slot of
// a scratch tape; k_tape_chunk_sums / k_tape_chunk_scan / k_tape_compact then pack the tapes back to back.  The container stack lives in
// the lane's private (scratch) memory: WALK_MAX_DEPTH levels.
//
// What stays on the host: a document nested deeper than WALK_MAX_DEPTH, and a document with a floating-point literal
// outside Clinger's exact range (more than 19 significant digits, a significand above 2^53 or |decimal exponent| > 22)
// -- there one IEEE multiplication or division of two exactly representable operands IS the correctly rounded result
// the reference's DoubleParser (DoubleParser.java:79-330) computes; outside it a correctly rounded conversion needs
// wide arithmetic.  Such documents get doc_errors[k] = SJMI_WALK_NEEDS_HOST and no tape; the host walker takes them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1.h"

namespace sjmi {

constexpr int WALK_MAX_DEPTH = 64;
constexpr int WALK_THREADS = 64;

namespace {

// byte-granular wide loads (gfx950 global memory is in unaligned-access mode)
struct __attribute__((packed, aligned(1))) W16B { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) W4B { uint32_t a; };

// A lane streams through its own document, so every access of a wave touches 64 different cache lines and nothing
// stays in the L1 between two accesses of the same lane: bytes and indexes are therefore fetched 16 bytes at a time
// into registers (a document has a structural every ~5 bytes, so a window serves about three of them; an index
// window serves four).
struct Lane {
    const uint8_t* buf;
    const uint32_t* ix;  // the batch's index array
    uint32_t ix_entries; // readable entries of it (count + sentinel)
    uint32_t iw_base;    // index window: entries [iw_base, iw_base + 4)
    uint32_t iw0, iw1, iw2, iw3;  // (scalars, not a uint4: hipcc 7.2's machine copy propagation crashes on the vector form)
    uint32_t bw_base;    // byte window: bytes [bw_base, bw_base + 16)
    uint32_t bw0, bw1, bw2, bw3;
    uint32_t from, to, rd;
    uint32_t doc_start, doc_end;
    unsigned long long* tape;
    uint32_t tl;
    const uint8_t* sb;
    unsigned long long sc;     // cursor in the string buffer
    unsigned long long sbase;  // what the caller adds to string offsets in tape payloads
    int code;                  // first error
};

__device__ __forceinline__ uint32_t pick4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t j) { return j == 0 ? a : (j == 1 ? b : (j == 2 ? c : d)); }
__device__ __forceinline__ uint32_t at(Lane& w, uint32_t i) {  // BitIndexes.java:82-96 (past the end: the sentinel)
    if (i >= w.to) return w.doc_start;
    if ((i & ~3u) != w.iw_base) {
        if ((i | 3u) >= w.ix_entries) return w.ix[i];
        w.iw_base = i & ~3u;
        const W16B v = *reinterpret_cast<const W16B*>(w.ix + w.iw_base);
        w.iw0 = v.a; w.iw1 = v.b; w.iw2 = v.c; w.iw3 = v.d;
    }
    return pick4(w.iw0, w.iw1, w.iw2, w.iw3, i & 3u);
}
__device__ __forceinline__ uint32_t byte_at(Lane& w, uint32_t p) {
    if (p - w.bw_base >= 16u) {
        w.bw_base = p;
        const W16B v = *reinterpret_cast<const W16B*>(w.buf + p);
        w.bw0 = v.a; w.bw1 = v.b; w.bw2 = v.c; w.bw3 = v.d;
    }
    const uint32_t o = p - w.bw_base;
    return (pick4(w.bw0, w.bw1, w.bw2, w.bw3, o >> 2) >> (8u * (o & 3u))) & 0xFFu;
}
__device__ __forceinline__ void append(Lane& w, unsigned long long v, char type) {                                // Tape.java:28-31
    w.tape[w.tl++] = v | ((unsigned long long)(uint8_t)type << 56);
}
__device__ __forceinline__ bool is_structural_or_ws(uint32_t b) {  // CharacterUtils.java:6-50
    return b == 0x20 || b == 0x0A || b == 0x0D || b == 0x09 || b == ',' || b == ':' || b == '[' || b == ']' || b == '{' || b == '}';
}

__device__ const double P10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// TapeBuilder.visitString (TapeBuilder.java:174-177): the record was written by the unescape kernels
__device__ __forceinline__ bool visit_string(Lane& w) {
    append(w, w.sbase + w.sc, '"');
    const uint32_t n = __builtin_bswap32(reinterpret_cast<const W4B*>(w.sb + w.sc)->a);  // be32 length
    if (n >= 0xFFFFFF00u) {  // a string StringParser would have thrown on: FF FF FF <code>
        w.code = (int)(n & 0xFFu);
        return false;
    }
    w.sc += 4 + (unsigned long long)n;
    return true;
}

// NumberParser.parseNumber (NumberParser.java:23-74) at p; bytes at or after `limit` read as spaces (the root number's
// padded copy, TapeBuilder.java:183-189)
__device__ bool parse_number(Lane& w, uint32_t p, uint32_t limit) {
    auto B = [&](uint32_t q) -> uint32_t { return q < limit ? byte_at(w, q) : 0x20u; };
    const bool negative = B(p) == '-';
    if (negative) ++p;
    const uint32_t digits_start = p;
    unsigned long long digits = 0;  // (wraps like the reference's long)
    // the significand for the floating-point case: up to 19 significant digits, zeros held back until a non-zero digit
    // follows them (trailing zeros of the fraction are dropped, those of the integer part become a power of ten)
    unsigned long long sig = 0;
    int nsig = 0, frac_used = 0, pend = 0, pend_int = 0;
    bool wide = false;
    auto push = [&](uint32_t d, bool frac) {
        if (d == 0) {
            if (sig == 0) frac_used += frac ? 1 : 0;  // a leading zero only moves the decimal point
            else { ++pend; pend_int += frac ? 0 : 1; }
            return;
        }
        frac_used += pend - pend_int;
        for (; pend; --pend) {
            if (nsig < 19) { sig *= 10; ++nsig; } else wide = true;
        }
        pend_int = 0;
        if (nsig < 19) { sig = sig * 10 + d; ++nsig; frac_used += frac ? 1 : 0; } else wide = true;
    };
    uint32_t c = B(p);
    while (c - '0' <= 9u) {
        const uint32_t d = c - '0';
        digits = 10 * digits + d;
        push(d, false);
        c = B(++p);
    }
    const uint32_t digit_count = p - digits_start;
    if (digit_count == 0) { w.code = SJMI_E_NUM_MINUS; return false; }
    if (B(digits_start) == '0' && digit_count > 1) { w.code = SJMI_E_NUM_LEADING_ZERO; return false; }
    bool floating = false;
    if (c == '.') {
        floating = true;
        c = B(++p);
        const uint32_t after = p;
        while (c - '0' <= 9u) {
            push(c - '0', true);
            c = B(++p);
        }
        if (p == after) { w.code = SJMI_E_NUM_DECIMAL_POINT; return false; }
    }
    int exp10 = 0;
    if (c == 'e' || c == 'E') {
        floating = true;
        c = B(++p);
        const bool eneg = c == '-';
        if (c == '-' || c == '+') c = B(++p);
        const uint32_t es = p;
        while (c - '0' <= 9u) {
            if (exp10 < 100000) exp10 = exp10 * 10 + (int)(c - '0');
            c = B(++p);
        }
        if (p == es) { w.code = SJMI_E_NUM_EXPONENT; return false; }
        if (eneg) exp10 = -exp10;
    }
    if (!is_structural_or_ws(c)) { w.code = SJMI_E_NUM_FOLLOWED; return false; }
    if (floating) {
        const int q = exp10 + pend_int - frac_used;
        if (wide || sig > (1ull << 53) || q < -22 || q > 22) { w.code = SJMI_WALK_NEEDS_HOST; return false; }
        double v = (double)sig;  // exact
        v = q < 0 ? v / P10[-q] : v * P10[q];
        if (negative) v = -v;
        append(w, 0, 'd');  // Tape.appendDouble :39-43
        w.tape[w.tl++] = (unsigned long long)__double_as_longlong(v);
    } else {
        bool out = false;  // isOutOfLongRange (NumberParser.java:313-328)
        if (digit_count > 19) out = true;
        else if (digit_count == 19) out = (negative && digits == 0x8000000000000000ull) ? false : ((long long)digits < 0);
        if (out) { w.code = SJMI_E_NUM_LONG_RANGE; return false; }
        append(w, 0, 'l');  // Tape.appendInt64 :33-37
        w.tape[w.tl++] = negative ? (~digits + 1) : digits;
    }
    return true;
}

// four bytes at p, little endian
__device__ __forceinline__ uint32_t word_at(Lane& w, uint32_t p) {
    return byte_at(w, p) | (byte_at(w, p + 1) << 8) | (byte_at(w, p + 2) << 16) | (byte_at(w, p + 3) << 24);
}
constexpr uint32_t W_TRUE = 0x65757274u, W_FALS = 0x736c6166u, W_NULL = 0x6c6c756eu;

// TapeBuilder.visitPrimitive (TapeBuilder.java:70-79) / visitRootPrimitive (:59-68): root = the document is this value
__device__ bool visit_primitive(Lane& w, uint32_t idx, bool root) {
    const uint32_t end = w.doc_end;
    switch (byte_at(w, idx)) {
    case '"': return visit_string(w);
    case 't':
        if (root ? !(idx + 4 <= end && word_at(w, idx) == W_TRUE && (idx + 4 == end || is_structural_or_ws(byte_at(w, idx + 4))))
                 : !(word_at(w, idx) == W_TRUE && is_structural_or_ws(byte_at(w, idx + 4)))) { w.code = SJMI_E_INVALID_TRUE; return false; }
        append(w, 0, 't');
        return true;
    case 'f':
        if (root ? !(idx + 5 <= end && word_at(w, idx) == W_FALS && byte_at(w, idx + 4) == 'e' &&
                     (idx + 5 == end || is_structural_or_ws(byte_at(w, idx + 5))))
                 : !(word_at(w, idx) == W_FALS && byte_at(w, idx + 4) == 'e' && is_structural_or_ws(byte_at(w, idx + 5)))) {
            w.code = SJMI_E_INVALID_FALSE;
            return false;
        }
        append(w, 0, 'f');
        return true;
    case 'n':
        if (root ? !(idx + 4 <= end && word_at(w, idx) == W_NULL && (idx + 4 == end || is_structural_or_ws(byte_at(w, idx + 4))))
                 : !(word_at(w, idx) == W_NULL && is_structural_or_ws(byte_at(w, idx + 4)))) { w.code = SJMI_E_INVALID_NULL; return false; }
        append(w, 0, 'n');
        return true;
    case '-': case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': case '8': case '9':
        return parse_number(w, idx, root ? end : 0xFFFFFFFFu);
    default: w.code = SJMI_E_UNRECOGNIZED_PRIMITIVE; return false;
    }
}

// JsonIterator.walkDocument (JsonIterator.java:26-200), state for state.  false: w.code holds the first error.
__device__ bool walk_document(Lane& w, int max_depth) {
    enum { OBJECT_BEGIN, ARRAY_BEGIN, DOCUMENT_END, OBJECT_FIELD, OBJECT_CONTINUE, SCOPE_END, ARRAY_CONTINUE, ARRAY_VALUE };
    uint32_t st_tape[WALK_MAX_DEPTH], st_count[WALK_MAX_DEPTH];  // TapeBuilder.OpenContainer (:210-213)
    unsigned long long is_array = 0;
    if (w.from == w.to) { w.code = SJMI_E_NO_STRUCTURAL; return false; }
#define SJ_FAIL(c) do { w.code = (c); return false; } while (0)
#define START_CONTAINER(d) do { st_tape[d] = w.tl; st_count[d] = 0; ++w.tl; } while (0)  /* TapeBuilder.java:191-195 */
#define END_CONTAINER(s, e, d) do { /* :197-203 */                                                               \
        const uint32_t st_ = st_tape[d];                                                                         \
        append(w, st_, e);                                                                                       \
        uint32_t cnt_ = st_count[d];                                                                             \
        if (cnt_ > 0xFFFFFFu) cnt_ = 0xFFFFFFu;                                                                  \
        w.tape[st_] = ((unsigned long long)w.tl | ((unsigned long long)cnt_ << 32)) | ((unsigned long long)(uint8_t)(s) << 56); \
    } while (0)
#define EMPTY_CONTAINER(s, e) do { append(w, w.tl + 2, s); append(w, w.tl, e); } while (0)  /* :205-208 */
    START_CONTAINER(0);  // visitDocumentStart :41-43
    int depth = 0, state;
    uint32_t idx = at(w, w.rd++);
    switch (byte_at(w, idx)) {
    case '{':
        if (w.buf[w.ix[w.to - 1]] != '}') SJ_FAIL(SJMI_E_UNCLOSED_OBJECT);
        if (byte_at(w, at(w, w.rd)) == '}') { ++w.rd; EMPTY_CONTAINER('{', '}'); state = DOCUMENT_END; }
        else state = OBJECT_BEGIN;
        break;
    case '[':
        if (w.buf[w.ix[w.to - 1]] != ']') SJ_FAIL(SJMI_E_UNCLOSED_ARRAY);
        if (byte_at(w, at(w, w.rd)) == ']') { ++w.rd; EMPTY_CONTAINER('[', ']'); state = DOCUMENT_END; }
        else state = ARRAY_BEGIN;
        break;
    default:
        if (!visit_primitive(w, idx, true)) return false;
        state = DOCUMENT_END;
    }
    while (state != DOCUMENT_END) {
        if (state == OBJECT_BEGIN) {
            ++depth;
            if (depth >= max_depth) SJ_FAIL(SJMI_E_DEPTH);
            if (depth >= WALK_MAX_DEPTH) SJ_FAIL(SJMI_WALK_NEEDS_HOST);
            is_array &= ~(1ull << depth);
            START_CONTAINER(depth);
            const uint32_t key = at(w, w.rd++);
            if (byte_at(w, key) != '"') SJ_FAIL(SJMI_E_OBJECT_NO_KEY);
            st_count[depth]++;
            if (!visit_string(w)) return false;
            state = OBJECT_FIELD;
        }
        if (state == OBJECT_FIELD) {
            if (byte_at(w, at(w, w.rd++)) != ':') SJ_FAIL(SJMI_E_MISSING_COLON);
            idx = at(w, w.rd++);
            switch (byte_at(w, idx)) {
            case '{':
                if (byte_at(w, at(w, w.rd)) == '}') { ++w.rd; EMPTY_CONTAINER('{', '}'); state = OBJECT_CONTINUE; }
                else state = OBJECT_BEGIN;
                break;
            case '[':
                if (byte_at(w, at(w, w.rd)) == ']') { ++w.rd; EMPTY_CONTAINER('[', ']'); state = OBJECT_CONTINUE; }
                else state = ARRAY_BEGIN;
                break;
            default:
                if (!visit_primitive(w, idx, false)) return false;
                state = OBJECT_CONTINUE;
            }
        }
        if (state == OBJECT_CONTINUE) {
            switch (byte_at(w, at(w, w.rd++))) {
            case ',': {
                st_count[depth]++;
                const uint32_t key = at(w, w.rd++);
                if (byte_at(w, key) != '"') SJ_FAIL(SJMI_E_KEY_MISSING);
                if (!visit_string(w)) return false;
                state = OBJECT_FIELD;
                break;
            }
            case '}':
                END_CONTAINER('{', '}', depth);
                state = SCOPE_END;
                break;
            default: SJ_FAIL(SJMI_E_NO_COMMA_OBJECT);
            }
        }
        if (state == SCOPE_END) {
            --depth;
            if (depth == 0) state = DOCUMENT_END;
            else if ((is_array >> depth) & 1ull) state = ARRAY_CONTINUE;
            else state = OBJECT_CONTINUE;
        }
        if (state == ARRAY_BEGIN) {
            ++depth;
            if (depth >= max_depth) SJ_FAIL(SJMI_E_DEPTH);
            if (depth >= WALK_MAX_DEPTH) SJ_FAIL(SJMI_WALK_NEEDS_HOST);
            is_array |= 1ull << depth;
            START_CONTAINER(depth);
            st_count[depth]++;
            state = ARRAY_VALUE;
        }
        if (state == ARRAY_VALUE) {
            idx = at(w, w.rd++);
            switch (byte_at(w, idx)) {
            case '{':
                if (byte_at(w, at(w, w.rd)) == '}') { ++w.rd; EMPTY_CONTAINER('{', '}'); state = ARRAY_CONTINUE; }
                else state = OBJECT_BEGIN;
                break;
            case '[':
                if (byte_at(w, at(w, w.rd)) == ']') { ++w.rd; EMPTY_CONTAINER('[', ']'); state = ARRAY_CONTINUE; }
                else state = ARRAY_BEGIN;
                break;
            default:
                if (!visit_primitive(w, idx, false)) return false;
                state = ARRAY_CONTINUE;
            }
        }
        if (state == ARRAY_CONTINUE) {
            switch (byte_at(w, at(w, w.rd++))) {
            case ',':
                st_count[depth]++;
                state = ARRAY_VALUE;
                break;
            case ']':
                END_CONTAINER('[', ']', depth);
                state = SCOPE_END;
                break;
            default: SJ_FAIL(SJMI_E_NO_COMMA_ARRAY);
            }
        }
    }
    append(w, 0, 'r');  // visitDocumentEnd :45-48
    w.tape[0] = (unsigned long long)w.tl | ((unsigned long long)(uint8_t)'r' << 56);
    if (w.rd != w.to) SJ_FAIL(SJMI_E_TRAILING_CONTENT);  // JsonIterator.java:196-198
    return true;
#undef SJ_FAIL
#undef START_CONTAINER
#undef END_CONTAINER
#undef EMPTY_CONTAINER
}

// document k's slot of the scratch tape: a structural makes at most two words, the root adds two
__device__ __forceinline__ unsigned long long scratch_slot(unsigned long long from, uint64_t k) { return 2 * from + 2 * k; }

}  // namespace

// (latency-bound: 64 % of the wave cycles wait for memory, profiles/r1/batch_pmc.txt -- a sixth wave per SIMD, 80
// instead of 82 VGPRs, is worth 8 %; a seventh needs SGPR spills and adds nothing)
__global__ void __launch_bounds__(WALK_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6)))
k_doc_walk(const uint8_t* __restrict__ buf, const unsigned long long* __restrict__ doc_offsets, uint64_t n_docs,
           const uint32_t* __restrict__ idx, uint32_t ix_entries, const unsigned long long* __restrict__ index_offsets,
           const uint32_t* __restrict__ doc_status, const uint8_t* __restrict__ sb,
           const unsigned long long* __restrict__ doc_str_offsets, unsigned long long string_base, int max_depth,
           unsigned long long* __restrict__ scratch_tape, uint32_t* __restrict__ tape_lens, int32_t* __restrict__ doc_errors) {
    const uint64_t k = (uint64_t)blockIdx.x * WALK_THREADS + threadIdx.x;
    if (k >= n_docs) return;
    const uint32_t st = doc_status[k];
    int code = 0;
    uint32_t len = 0;
    // SimdJsonParser.stage1 order: Utf8Validator.validate (:165-167), then StructuralIndexer.index (:297-302)
    if (st & SJMI_ST_UTF8) code = SJMI_E_UTF8;
    else if (st & SJMI_ST_UNCLOSED) code = SJMI_E_UNCLOSED_STRING;
    else if (st & SJMI_ST_UNESCAPED) code = SJMI_E_UNESCAPED_CHARS;
    else {
        Lane w;
        w.buf = buf;
        w.ix = idx;
        w.ix_entries = ix_entries;
        w.iw_base = 0xFFFFFFFFu;  // (never a multiple of four)
        w.bw_base = 0xFFFFFFF0u;  // p - base >= 16 for every p the lane can ask for
        w.from = (uint32_t)index_offsets[k];
        w.to = (uint32_t)index_offsets[k + 1];
        w.rd = w.from;
        w.doc_start = (uint32_t)doc_offsets[k];
        w.doc_end = (uint32_t)doc_offsets[k + 1];
        w.tape = scratch_tape + scratch_slot(index_offsets[k], k);
        w.tl = 0;
        w.sb = sb;
        w.sc = doc_str_offsets[k];
        w.sbase = string_base;
        w.code = 0;
        if (walk_document(w, max_depth)) len = w.tl;
        else code = w.code;
    }
    tape_lens[k] = len;
    doc_errors[k] = code;
}

// ---- packing the tapes ---------------------------------------------------------------------------
constexpr int PACK_DOCS = 1024;  // documents per workgroup

__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long* s_wave,
                                                              unsigned long long* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, t = 0;
    for (int i = 0; i < nw; ++i) {
        if (i < wave) base += s_wave[i];
        t += s_wave[i];
    }
    *total = t;
    return base + incl - v;
}

__global__ void __launch_bounds__(1024)
k_tape_chunk_sums(const uint32_t* __restrict__ tape_lens, const int32_t* __restrict__ doc_errors, uint64_t n_docs,
                  unsigned long long* __restrict__ chunk_sums, WalkResult* res) {
    __shared__ unsigned long long s_wave[16];
    const uint64_t k = (uint64_t)blockIdx.x * PACK_DOCS + threadIdx.x;
    unsigned long long total;
    (void)block_excl_scan(k < n_docs ? tape_lens[k] : 0u, s_wave, &total);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = total;
    const int e = k < n_docs ? doc_errors[k] : 0;
    const unsigned long long host = __ballot(e == SJMI_WALK_NEEDS_HOST), bad = __ballot(e > 0);
    if ((threadIdx.x & 63) == 0) {
        if (host) atomicAdd(&res->host_documents, (unsigned long long)__popcll(host));
        if (bad) atomicAdd(&res->failed_documents, (unsigned long long)__popcll(bad));
    }
}

__global__ void __launch_bounds__(1024)
k_tape_chunk_scan(unsigned long long* __restrict__ chunk_sums, uint64_t nchunks, uint64_t n_docs, uint64_t tape_capacity,
                  unsigned long long* __restrict__ tape_offsets, WalkResult* res) {
    __shared__ unsigned long long s_wave[16];
    unsigned long long carry = 0;
    for (uint64_t b = 0; b < nchunks; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const unsigned long long v = i < nchunks ? chunk_sums[i] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_scan(v, s_wave, &total);
        if (i < nchunks) chunk_sums[i] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tape_offsets[n_docs] = carry;
        res->tape_words = carry;
        if (carry > tape_capacity) res->flags |= 1u;
    }
}

// one workgroup per PACK_DOCS documents: their offsets, then every wave copies documents (coalesced 8-byte words)
__global__ void __launch_bounds__(1024)
k_tape_compact(const unsigned long long* __restrict__ scratch_tape, const uint32_t* __restrict__ tape_lens,
               const unsigned long long* __restrict__ index_offsets, uint64_t n_docs,
               const unsigned long long* __restrict__ chunk_base, unsigned long long* __restrict__ tape, uint64_t tape_capacity,
               unsigned long long* __restrict__ tape_offsets) {
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_off[PACK_DOCS];
    const uint64_t k0 = (uint64_t)blockIdx.x * PACK_DOCS, k = k0 + threadIdx.x;
    unsigned long long total;
    const unsigned long long off = chunk_base[blockIdx.x] + block_excl_scan(k < n_docs ? tape_lens[k] : 0u, s_wave, &total);
    s_off[threadIdx.x] = off;
    if (k < n_docs) tape_offsets[k] = off;
    __syncthreads();
    if (chunk_base[blockIdx.x] + total > tape_capacity) return;  // (reported by k_tape_chunk_scan)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < PACK_DOCS; j += 16) {
        const uint64_t d = k0 + (uint64_t)j;
        if (d >= n_docs) break;
        const uint32_t n = tape_lens[d];
        const unsigned long long* src = scratch_tape + scratch_slot(index_offsets[d], d);
        unsigned long long* dst = tape + s_off[j];
        for (uint32_t i = (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
    }
}

// workspace: scratch tape (2 count + 2 n words) | tape lengths [n] | chunk sums
static size_t walk_lens_offset(uint64_t count, uint64_t n_docs) { return ((2 * count + 2 * n_docs + 8) * sizeof(unsigned long long) + 63) / 64 * 64; }
static size_t walk_sums_offset(uint64_t count, uint64_t n_docs) { return walk_lens_offset(count, n_docs) + (n_docs * sizeof(uint32_t) + 63) / 64 * 64 + 64; }
size_t walk_workspace_bytes(uint64_t count, uint64_t n_docs) {
    return walk_sums_offset(count, n_docs) + ((n_docs + PACK_DOCS - 1) / PACK_DOCS + 2) * sizeof(unsigned long long) + 64;
}

hipError_t walk_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, const uint32_t* d_idx,
                       uint64_t count, const unsigned long long* d_index_offsets, const uint32_t* d_doc_status,
                       const uint8_t* d_sb, const unsigned long long* d_doc_str_offsets, uint64_t string_base, int max_depth,
                       unsigned long long* d_tape, uint64_t tape_capacity, unsigned long long* d_tape_offsets,
                       int32_t* d_doc_errors, void* d_ws, WalkResult* d_res, hipStream_t stream) {
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    unsigned long long* scratch = reinterpret_cast<unsigned long long*>(ws);
    uint32_t* lens = reinterpret_cast<uint32_t*>(ws + walk_lens_offset(count, n_docs));
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(ws + walk_sums_offset(count, n_docs));
    hipError_t e = hipMemsetAsync(d_res, 0, sizeof(WalkResult), stream);
    if (e != hipSuccess) return e;
    const uint64_t nchunks = (n_docs + PACK_DOCS - 1) / PACK_DOCS;
    if (n_docs) {
        hipLaunchKernelGGL(k_doc_walk, dim3((unsigned)((n_docs + WALK_THREADS - 1) / WALK_THREADS)), dim3(WALK_THREADS), 0, stream,
                           d_buf, d_doc_offsets, n_docs, d_idx, (uint32_t)(count + 1), d_index_offsets, d_doc_status, d_sb, d_doc_str_offsets,
                           (unsigned long long)string_base, max_depth, scratch, lens, d_doc_errors);
        hipLaunchKernelGGL(k_tape_chunk_sums, dim3((unsigned)nchunks), dim3(1024), 0, stream, lens, d_doc_errors, n_docs, sums, d_res);
    }
    hipLaunchKernelGGL(k_tape_chunk_scan, dim3(1), dim3(1024), 0, stream, sums, nchunks, n_docs, tape_capacity, d_tape_offsets, d_res);
    if (n_docs)
        hipLaunchKernelGGL(k_tape_compact, dim3((unsigned)nchunks), dim3(1024), 0, stream, scratch, lens, d_index_offsets, n_docs,
                           sums, d_tape, tape_capacity, d_tape_offsets);
    return hipGetLastError();
}

}  // namespace sjmi
