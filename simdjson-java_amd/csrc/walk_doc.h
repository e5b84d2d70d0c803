// SPDX-License-Identifier: Apache-2.0
// The per-document automaton of the GPU stage 2 (walk.hip): JsonIterator.walkDocument + TapeBuilder + the number
// grammar for ONE document, as plain code over pointers.  k_doc_walk runs it, one lane per document; the CPU test-suite
// compiles the same header with g++ (tests/host_sim/walkdev_sim.cpp) and fuzzes it against the oracle without a GPU --
// grammar, atoms, integers, the exact-range float conversion and the hand-back decisions are all in here.
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/sjmi.h"
#include "sj_number.h"

#if defined(__HIPCC__)
#define SJW_DEV __device__
#define SJW_INL __forceinline__
#define SJW_CONST __device__ const
#else
#define SJW_DEV static
#define SJW_INL inline
#define SJW_CONST static const
#endif

namespace sjmi {

constexpr int WALK_MAX_DEPTH = 64;

namespace {

SJW_DEV SJW_INL unsigned long long sjw_double_bits(double v) {
    unsigned long long u;
    memcpy(&u, &v, sizeof u);
    return u;
}


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

SJW_DEV SJW_INL uint32_t pick4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t j) { return j == 0 ? a : (j == 1 ? b : (j == 2 ? c : d)); }
SJW_DEV SJW_INL uint32_t at(Lane& w, uint32_t i) {  // BitIndexes.java:82-96 (past the end: the sentinel)
    if (i >= w.to) return w.doc_start;
    if ((i & ~3u) != w.iw_base) {
        if ((i | 3u) >= w.ix_entries) return w.ix[i];
        w.iw_base = i & ~3u;
        const W16B v = *reinterpret_cast<const W16B*>(w.ix + w.iw_base);
        w.iw0 = v.a; w.iw1 = v.b; w.iw2 = v.c; w.iw3 = v.d;
    }
    return pick4(w.iw0, w.iw1, w.iw2, w.iw3, i & 3u);
}
SJW_DEV SJW_INL uint32_t byte_at(Lane& w, uint32_t p) {
    if (p - w.bw_base >= 16u) {
        w.bw_base = p;
        const W16B v = *reinterpret_cast<const W16B*>(w.buf + p);
        w.bw0 = v.a; w.bw1 = v.b; w.bw2 = v.c; w.bw3 = v.d;
    }
    const uint32_t o = p - w.bw_base;
    return (pick4(w.bw0, w.bw1, w.bw2, w.bw3, o >> 2) >> (8u * (o & 3u))) & 0xFFu;
}
SJW_DEV SJW_INL void append(Lane& w, unsigned long long v, char type) {                                // Tape.java:28-31
    w.tape[w.tl++] = v | ((unsigned long long)(uint8_t)type << 56);
}
SJW_DEV SJW_INL bool is_structural_or_ws(uint32_t b) {  // CharacterUtils.java:6-50
    return b == 0x20 || b == 0x0A || b == 0x0D || b == 0x09 || b == ',' || b == ':' || b == '[' || b == ']' || b == '{' || b == '}';
}

// TapeBuilder.visitString (TapeBuilder.java:174-177): the record was written by the unescape kernels
SJW_DEV SJW_INL bool visit_string(Lane& w) {
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
// padded copy, TapeBuilder.java:183-189).  Grammar and conversion: sj_number.h (Clinger's exact range + Eisel-Lemire for
// every literal of at most 19 significant digits and for the longer ones whose two 19-digit neighbours round alike; the
// rest goes back to the host).
SJW_DEV bool parse_number(Lane& w, uint32_t p, uint32_t limit) {
    const SjNumber n = sj_scan_number([&](uint32_t q) -> uint32_t { return q < limit ? byte_at(w, q) : 0x20u; }, p);
    if (n.code) { w.code = n.code; return false; }
    if (n.floating) {
        unsigned long long bits;
        if (!sj_number_double_bits(n, &bits)) { w.code = SJMI_WALK_NEEDS_HOST; return false; }  // DoubleParser's slow path (:205-330)
        append(w, 0, 'd');  // Tape.appendDouble :39-43
        w.tape[w.tl++] = bits;
    } else {
        if (sj_out_of_long_range(n.negative, n.digits, n.digit_count)) { w.code = SJMI_E_NUM_LONG_RANGE; return false; }
        append(w, 0, 'l');  // Tape.appendInt64 :33-37
        w.tape[w.tl++] = n.negative ? (~n.digits + 1) : n.digits;
    }
    return true;
}

// four bytes at p, little endian
SJW_DEV SJW_INL uint32_t word_at(Lane& w, uint32_t p) {
    return byte_at(w, p) | (byte_at(w, p + 1) << 8) | (byte_at(w, p + 2) << 16) | (byte_at(w, p + 3) << 24);
}
constexpr uint32_t W_TRUE = 0x65757274u, W_FALS = 0x736c6166u, W_NULL = 0x6c6c756eu;

// TapeBuilder.visitPrimitive (TapeBuilder.java:70-79) / visitRootPrimitive (:59-68): root = the document is this value
SJW_DEV bool visit_primitive(Lane& w, uint32_t idx, bool root) {
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
SJW_DEV bool walk_document(Lane& w, int max_depth) {
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

}  // namespace

}  // namespace sjmi
