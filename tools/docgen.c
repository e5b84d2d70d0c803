// docgen.c -- workload generator for BASELINE.json configs[3] as SURVEY.md 8(d) specifies it: 1,000,000 UNIQUE documents,
// seed 20250825, length uniform in [768, 1280] B, each a flat-ish record (string / integer / atom fields, arrays of <= 8
// small integers, one-level nested objects; 5 % of the string characters escape sequences, 5 % non-ASCII), packed back
// to back with one '\n' behind each document and a u64 offsets table.  Document k depends on (seed, k) only, so any
// rank of a sharded run generates exactly its own range [first, first + n) and no rank ever holds the whole set.
// Workload infrastructure for bench.py / the full-scale tests (built by __graft_entry__.build() with gcc); not part
// of the product and not the oracle.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef struct { uint64_t s; } Rng;
static inline uint64_t splitmix(uint64_t* x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t rnd(Rng* r) {  // xorshift64*
    uint64_t x = r->s;
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    r->s = x;
    return x * 0x2545F4914F6CDD1Dull;
}
static inline uint32_t below(Rng* r, uint32_t n) { return (uint32_t)(((rnd(r) >> 32) * (uint64_t)n) >> 32); }
static inline uint32_t range(Rng* r, uint32_t lo, uint32_t hi) { return lo + below(r, hi - lo + 1); }  // inclusive

static const char ASCII[] = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 _-.,:;!?()[]{}@#$%&*+=<>|~'";
static const char* SIMPLE[8] = {"\\\"", "\\\\", "\\/", "\\b", "\\f", "\\n", "\\r", "\\t"};

static uint8_t* put_utf8(uint8_t* p, uint32_t cp) {
    if (cp < 0x80) *p++ = (uint8_t)cp;
    else if (cp < 0x800) { *p++ = 0xC0 | (cp >> 6); *p++ = 0x80 | (cp & 63); }
    else if (cp < 0x10000) { *p++ = 0xE0 | (cp >> 12); *p++ = 0x80 | ((cp >> 6) & 63); *p++ = 0x80 | (cp & 63); }
    else { *p++ = 0xF0 | (cp >> 18); *p++ = 0x80 | ((cp >> 12) & 63); *p++ = 0x80 | ((cp >> 6) & 63); *p++ = 0x80 | (cp & 63); }
    return p;
}
static uint8_t* put_str(uint8_t* p, const char* s) { while (*s) *p++ = (uint8_t)*s++; return p; }
static uint8_t* put_hex4(uint8_t* p, uint32_t v, int upper) {
    static const char lo[] = "0123456789abcdef", up[] = "0123456789ABCDEF";
    const char* d = upper ? up : lo;
    *p++ = '\\'; *p++ = 'u';
    *p++ = d[(v >> 12) & 15]; *p++ = d[(v >> 8) & 15]; *p++ = d[(v >> 4) & 15]; *p++ = d[v & 15];
    return p;
}
static uint8_t* put_int(uint8_t* p, int64_t v) {
    char tmp[24];
    int n = snprintf(tmp, sizeof tmp, "%lld", (long long)v);
    memcpy(p, tmp, (size_t)n);
    return p + n;
}
// a string literal of n_chars characters; esc_pm / na_pm = per-mille of escape sequences / non-ASCII characters
static uint8_t* put_string(uint8_t* p, Rng* r, uint32_t n_chars, uint32_t esc_pm, uint32_t na_pm) {
    *p++ = '"';
    for (uint32_t i = 0; i < n_chars; ++i) {
        const uint32_t t = below(r, 1000);
        if (t < esc_pm) {
            const uint32_t k = below(r, 100);
            if (k < 70) p = put_str(p, SIMPLE[below(r, 8)]);
            else if (k < 95) p = put_hex4(p, below(r, 5) < 4 ? range(r, 0x20, 0xD7FF) : range(r, 0xE000, 0xFFFF), 0);
            else {
                const uint32_t v = range(r, 0x10000, 0x10FFFF) - 0x10000;
                p = put_hex4(p, 0xD800 + (v >> 10), 1);
                p = put_hex4(p, 0xDC00 + (v & 0x3FF), 1);
            }
        } else if (t < esc_pm + na_pm) {
            const uint32_t k = below(r, 10);
            uint32_t cp;
            if (k < 6) cp = range(r, 0x80, 0x7FF);
            else if (k < 9) { cp = range(r, 0x800, 0xFFFF); if (cp >= 0xD800 && cp <= 0xDFFF) cp = 0x4E2D; }
            else cp = range(r, 0x10000, 0x10FFFF);
            p = put_utf8(p, cp);
        } else {
            *p++ = (uint8_t)ASCII[below(r, (uint32_t)(sizeof ASCII - 1))];
        }
    }
    *p++ = '"';
    return p;
}

// document k of the set: -> bytes written (without the separator); out needs 2048 bytes of room.  Fields are drawn until
// the next one would not fit; the last field "z" is a plain ASCII string that brings the document to exactly the drawn
// length (so the lengths ARE uniform in [768, 1280]).
#ifndef DOC_SCALE
#define DOC_SCALE 1  /* experiments only (tools/r5_doc_scale.sh): documents DOC_SCALE times as long; the configs[3] set is scale 1 */
#endif
static uint32_t one_doc(uint64_t seed, uint64_t k, uint8_t* out) {
    uint64_t sm = seed ^ (k * 0xD1342543DE82EF95ull);
    Rng r = {splitmix(&sm) | 1ull};
    const uint32_t target = range(&r, 768 * DOC_SCALE, 1280 * DOC_SCALE);
    uint8_t* p = out;
    *p++ = '{';
    for (uint32_t i = 0;; ++i) {
        uint8_t* const field = p;
        if (i) *p++ = ',';
        *p++ = '"'; *p++ = 'k';
        p = put_int(p, (int64_t)i);
        *p++ = '"'; *p++ = ':';
        const uint32_t kind = below(&r, 100);
        if (kind < 40) p = put_string(p, &r, range(&r, 8, 60), 50, 50);
        else if (kind < 70) p = put_int(p, (int64_t)below(&r, 1001000000u) - 1000000);
        else if (kind < 80) p = put_str(p, (const char*[]){"true", "false", "null"}[below(&r, 3)]);
        else if (kind < 90) {
            *p++ = '[';
            const uint32_t m = below(&r, 9);
            for (uint32_t j = 0; j < m; ++j) {
                if (j) *p++ = ',';
                p = put_int(p, (int64_t)below(&r, 100));
            }
            *p++ = ']';
        } else {
            p = put_str(p, "{\"x\":");
            p = put_int(p, (int64_t)below(&r, 100));
            p = put_str(p, ",\"y\":");
            p = put_string(p, &r, 6, 0, 0);
            *p++ = '}';
        }
        // room for this field, the filler field (8 bytes + its characters) and the closing brace?
        if ((uint32_t)(p - out) + 9 > target) {
            p = field;
            break;
        }
    }
    const uint32_t fill = target - (uint32_t)(p - out) - 8;  // ,"z":"<fill>"}
    p = put_str(p, ",\"z\":\"");
    for (uint32_t j = 0; j < fill; ++j) *p++ = (uint8_t)ASCII[below(&r, 62)];
    *p++ = '"';
    *p++ = '}';
    return (uint32_t)(p - out);
}

// lengths (incl. the '\n' separator) of documents [first, first + n)
void docgen_lengths(uint64_t seed, uint64_t first, uint64_t n, uint64_t* lens) {
    uint8_t tmp[2048 * DOC_SCALE];
    for (uint64_t i = 0; i < n; ++i) lens[i] = (uint64_t)one_doc(seed, first + i, tmp) + 1;
}
// documents [first, first + n) at out + offsets[i] (offsets relative to `out`; each followed by '\n')
void docgen_fill(uint64_t seed, uint64_t first, uint64_t n, uint8_t* out, const uint64_t* offsets) {
    uint8_t tmp[2048 * DOC_SCALE];
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t len = one_doc(seed, first + i, tmp);
        memcpy(out + offsets[i], tmp, len);
        out[offsets[i] + len] = '\n';
    }
}
