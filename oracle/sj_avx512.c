/*
 * sj_avx512.c -- AVX-512 restatement of the reference's stage 1 for the CPU TIMING baseline.
 *
 * TEST INFRASTRUCTURE (see sj_oracle.h): loaded only by bench.py's cpu_baseline leg and by the tests that check it
 * against the scalar restatement.  The reference's CPU path is Java's jdk.incubator.vector at 512 bits
 * (-Dorg.simdjson.species=512, README.md); no JVM exists in this image, so this is the same per-64-byte-block algorithm
 * written with the x86 instructions the JIT emits for it -- vpcmpb -> kmov for the character classes, vpshufb for the
 * low-nibble tables, 64-bit scalar mask arithmetic -- so that the GPU number stands beside a CPU number of the right
 * order of magnitude instead of the byte-at-a-time port (sj_oracle.c).  Two passes over the input like the reference
 * (SimdJsonParser.java:55-58): Utf8Validator.validate, then StructuralIndexer.index.
 *
 *   sjo_stage1_avx512   Utf8Validator.java:54-168 (512-bit species) + StructuralIndexer.java:196-303 (index512)
 *                       + BitIndexes.write / finish (BitIndexes.java:14-41,82-96)
 *
 * Built into liboracle_avx512.so with -mavx512f -mavx512bw -mbmi -mbmi2 -mlzcnt -mpopcnt; callers must check
 * sjo_avx512_supported() (sj_oracle.c, portable build) first.
 */
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "sj_oracle.h"
#include "sj_tables.h"

static inline __m512i table16(const uint8_t t[16]) { return _mm512_broadcast_i32x4(_mm_loadu_si128((const __m128i *)t)); }

/* bytes [-N, 64 - N) of the stream: the current chunk shifted forward by N bytes, the gap filled from the previous
 * chunk (Utf8Validator.java:73-80,95-104 do this with integer-lane shifts) */
#define PREV_BYTES(cur, lanes_back, N) _mm512_alignr_epi8((cur), (lanes_back), 16 - (N))

static inline uint64_t utf8_chunk512(__m512i cur, __m512i prev, uint64_t *incomplete, __m512i t1h, __m512i t1l, __m512i t2h,
                                     __m512i incomplete_check) {
    const __m512i nib = _mm512_set1_epi8(0x0F);
    /* 128-bit lane i of lanes_back = lane i - 1 of the stream (lane 3 of the previous chunk for i = 0) */
    const __m512i lanes_back = _mm512_alignr_epi32(cur, prev, 12);
    const __m512i prev1 = PREV_BYTES(cur, lanes_back, 1);
    const __m512i b1h = _mm512_shuffle_epi8(t1h, _mm512_and_si512(_mm512_srli_epi16(prev1, 4), nib));  /* :81-92 */
    const __m512i b1l = _mm512_shuffle_epi8(t1l, _mm512_and_si512(prev1, nib));
    const __m512i b2h = _mm512_shuffle_epi8(t2h, _mm512_and_si512(_mm512_srli_epi16(cur, 4), nib));
    const __m512i first = _mm512_and_si512(_mm512_and_si512(b1h, b1l), b2h);
    const __m512i prev2 = PREV_BYTES(cur, lanes_back, 2);
    const __m512i prev3 = PREV_BYTES(cur, lanes_back, 3);
    const __mmask64 lead3 = _mm512_cmpgt_epu8_mask(prev2, _mm512_set1_epi8((char)0xDF));  /* :100 */
    const __mmask64 lead4 = _mm512_cmpgt_epu8_mask(prev3, _mm512_set1_epi8((char)0xEF));  /* :106 */
    const __m512i second = _mm512_mask_add_epi8(first, lead3 | lead4, first, _mm512_set1_epi8((char)0x80));  /* :109 */
    *incomplete = _mm512_cmpge_epu8_mask(cur, incomplete_check);  /* :68 */
    return _mm512_test_epi8_mask(second, second);                 /* :110 */
}

static int utf8_validate512(const uint8_t *buf, uint64_t len) {
    const __m512i t1h = table16(BYTE_1_HIGH), t1l = table16(BYTE_1_LOW), t2h = table16(BYTE_2_HIGH);
    uint8_t chk[64];
    memset(chk, 0xFF, sizeof chk);  /* INCOMPLETE_CHECK :170-180 */
    chk[61] = 0xF0;
    chk[62] = 0xE0;
    chk[63] = 0xC0;
    const __m512i incomplete_check = _mm512_loadu_si512(chk);
    uint64_t previous_incomplete = 0, errors = 0;
    __m512i prev = _mm512_setzero_si512();
    const uint64_t loop_bound = len & ~(uint64_t)63;
    uint64_t off = 0;
    for (; off < loop_bound; off += 64) {
        const __m512i cur = _mm512_loadu_si512(buf + off);
        if (_mm512_movepi8_mask(cur) == 0) errors |= previous_incomplete;  /* :65-66 */
        else errors |= utf8_chunk512(cur, prev, &previous_incomplete, t1h, t1l, t2h, incomplete_check);
        prev = cur;  /* (the reference keeps the last four bytes, :112) */
    }
    const __mmask64 rem = len > off ? (~0ULL >> (64 - (len - off))) : 0;  /* :115-117 */
    const __m512i cur = _mm512_maskz_loadu_epi8(rem, buf + off);
    if (_mm512_movepi8_mask(cur) != 0) errors |= utf8_chunk512(cur, prev, &previous_incomplete, t1h, t1l, t2h, incomplete_check);
    return (errors | previous_incomplete) == 0;  /* :165 */
}

typedef struct {
    uint64_t prev_in_string, prev_escaped, prev_scalar, unescaped_error;
} idx512_state;

static inline uint64_t prefix_xor64(uint64_t m) {  /* StructuralIndexer.java:311-319 */
    m ^= m << 1;
    m ^= m << 2;
    m ^= m << 4;
    m ^= m << 8;
    m ^= m << 16;
    m ^= m << 32;
    return m;
}

/* BitIndexes.write (BitIndexes.java:14-41): 8 stores unconditionally, 8 more if needed, then one by one; the array has
 * 16 entries of slack behind the capacity for the speculative stores */
static inline uint64_t bits_write(uint32_t *out, uint64_t write_idx, uint32_t block_start, uint64_t bits) {
    if (bits == 0) return write_idx;
    const unsigned cnt = (unsigned)_mm_popcnt_u64(bits);
    uint32_t *q = out + write_idx;
    for (int i = 0; i < 8; i++) {
        q[i] = block_start + (uint32_t)_tzcnt_u64(bits);
        bits = _blsr_u64(bits);
    }
    if (cnt > 8) {
        for (int i = 8; i < 16; i++) {
            q[i] = block_start + (uint32_t)_tzcnt_u64(bits);
            bits = _blsr_u64(bits);
        }
        for (unsigned i = 16; i < cnt; i++) {
            q[i] = block_start + (uint32_t)_tzcnt_u64(bits);
            bits = _blsr_u64(bits);
        }
    }
    return write_idx + cnt;
}

static inline uint64_t index_block512(__m512i c, idx512_state *st, __m512i ws_table, __m512i op_table) {
    uint64_t backslash = _mm512_cmpeq_epi8_mask(c, _mm512_set1_epi8('\\'));  /* :210 */
    uint64_t escaped;
    if (backslash == 0) {  /* :213-215 */
        escaped = st->prev_escaped;
        st->prev_escaped = 0;
    } else {  /* :217-228 */
        backslash &= ~st->prev_escaped;
        const uint64_t follows_escape = backslash << 1 | st->prev_escaped;
        const uint64_t odd_starts = backslash & 0xAAAAAAAAAAAAAAAAULL & ~follows_escape;
        uint64_t seq_even;
        st->prev_escaped = (uint64_t)__builtin_add_overflow(odd_starts, backslash, &seq_even);
        escaped = (0x5555555555555555ULL ^ (seq_even << 1)) & follows_escape;
    }
    const uint64_t unescaped = _mm512_cmple_epu8_mask(c, _mm512_set1_epi8(0x1F));          /* :231 */
    const uint64_t quote = _mm512_cmpeq_epi8_mask(c, _mm512_set1_epi8('"')) & ~escaped;    /* :232 */
    const uint64_t in_string = prefix_xor64(quote) ^ st->prev_in_string;                  /* :233 */
    st->prev_in_string = (uint64_t)((int64_t)in_string >> 63);                            /* :234 */
    /* vpshufb looks the low nibble up and yields 0 for bytes >= 0x80, which never equals such a byte (:237-240) */
    const uint64_t whitespace = _mm512_cmpeq_epi8_mask(c, _mm512_shuffle_epi8(ws_table, c));
    const uint64_t op = _mm512_cmpeq_epi8_mask(_mm512_or_si512(c, _mm512_set1_epi8(0x20)), _mm512_shuffle_epi8(op_table, c));
    const uint64_t scalar = ~(op | whitespace);                          /* :243 */
    const uint64_t non_quote_scalar = scalar & ~quote;                   /* :244 */
    const uint64_t follows_nqs = non_quote_scalar << 1 | st->prev_scalar;
    st->prev_scalar = non_quote_scalar >> 63;
    const uint64_t potential = op | (scalar & ~follows_nqs);             /* :247-248 */
    st->unescaped_error |= unescaped & in_string;                        /* :252 */
    return potential & ~(in_string ^ quote);                             /* :251 */
}

/* indexes needs index_capacity + 16 entries of storage (speculative stores of BitIndexes.write) */
int sjo_stage1_avx512(const uint8_t *buf, uint64_t len, uint32_t *indexes, uint64_t index_capacity, uint64_t *count,
                      uint32_t *status) {
    uint32_t stt = utf8_validate512(buf, len) ? 0 : SJO_ST_UTF8;
    const __m512i ws_table = table16(WHITESPACE_TABLE), op_table = table16(OP_TABLE);
    idx512_state st = {0, 0, 0, 0};
    uint64_t n = 0;
    const uint64_t loop_bound = len & ~(uint64_t)63;
    uint64_t off = 0;
    for (; off < loop_bound; off += 64) {
        const uint64_t s = index_block512(_mm512_loadu_si512(buf + off), &st, ws_table, op_table);
        if (n + 64 > index_capacity) return -1;
        n = bits_write(indexes, n, (uint32_t)off, s);
    }
    /* tail: space-filled (StructuralIndexer.java:305-309), always processed */
    const __mmask64 rem = len > off ? (~0ULL >> (64 - (len - off))) : 0;
    const __m512i tail = _mm512_mask_loadu_epi8(_mm512_set1_epi8(' '), rem, buf + off);
    const uint64_t s = index_block512(tail, &st, ws_table, op_table);
    if (n + 64 >= index_capacity) return -1;
    n = bits_write(indexes, n, (uint32_t)off, s);
    indexes[n] = 0;  /* BitIndexes.finish :82-96 */
    *count = n;
    if (st.prev_in_string) stt |= SJO_ST_UNCLOSED;    /* :297-299 */
    if (st.unescaped_error) stt |= SJO_ST_UNESCAPED;  /* :300-302 */
    *status = stt;
    return 0;
}
