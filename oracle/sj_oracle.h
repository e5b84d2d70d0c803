/*
 * sj_oracle.h -- CPU ORACLE for the simdjson-java stage-1 / string / stage-2 path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library (libsjmi.so) never
 * links or calls anything in oracle/.
 *
 * It is a plain-C restatement of the reference algorithm (pure Java + Vector API, which
 * cannot run in this image: no JVM).  Every function cites the reference file:line it
 * follows (paths relative to /root/reference/src/main/java/org/simdjson/).  Parity is
 * pinned by the reference's own literal test vectors (tests/golden/*.json, transcribed
 * from src/test/java/org/simdjson/*Test.java) -- see tests/test_oracle_golden.py.
 */
#ifndef SJ_ORACLE_H
#define SJ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* stage-1 status bits (a replacement for the three JsonParsingExceptions thrown by
 * SimdJsonParser.stage1, SimdJsonParser.java:55-58; priority = lowest bit first). */
#define SJO_ST_UTF8 1u      /* "The input is not valid UTF-8"                 Utf8Validator.java:165-167 */
#define SJO_ST_UNCLOSED 2u  /* "Unclosed string. A string is opened, ..."      StructuralIndexer.java:297-299 */
#define SJO_ST_UNESCAPED 4u /* "Unescaped characters. Within strings, ..."     StructuralIndexer.java:300-302 */

/* error codes of the string parser / stage 2 (exact reference messages in sjo_error_message) */
enum sjo_error {
    SJO_OK = 0,
    SJO_E_UTF8 = 1,
    SJO_E_UNCLOSED_STRING = 2,
    SJO_E_UNESCAPED_CHARS = 3,
    SJO_E_ESCAPE_UNEXPECTED = 4,      /* CharacterUtils.java:74-83 */
    SJO_E_INVALID_UNICODE_ESCAPE = 5, /* StringParser.java:127-129 */
    SJO_E_LOW_SURROGATE_RESERVED = 6, /* StringParser.java:53-55 */
    SJO_E_LOW_SURROGATE_NO_U = 7,     /* StringParser.java:113-115 */
    SJO_E_LOW_SURROGATE_RANGE = 8,    /* StringParser.java:118-122 */
    SJO_E_NO_STRUCTURAL = 9,          /* JsonIterator.java:27-29 */
    SJO_E_UNCLOSED_OBJECT = 10,       /* JsonIterator.java:39-41 */
    SJO_E_UNCLOSED_ARRAY = 11,        /* JsonIterator.java:51-53 */
    SJO_E_OBJECT_NO_KEY = 12,         /* JsonIterator.java:75-77 */
    SJO_E_MISSING_COLON = 13,         /* JsonIterator.java:84-86 */
    SJO_E_KEY_MISSING = 14,           /* JsonIterator.java:121-123 */
    SJO_E_NO_COMMA_OBJECT = 15,       /* JsonIterator.java:131 */
    SJO_E_NO_COMMA_ARRAY = 16,        /* JsonIterator.java:189 */
    SJO_E_TRAILING_CONTENT = 17,      /* JsonIterator.java:196-198 */
    SJO_E_UNRECOGNIZED_PRIMITIVE = 18,/* TapeBuilder.java:66,77 */
    SJO_E_INVALID_TRUE = 19,          /* TapeBuilder.java:100-114 */
    SJO_E_INVALID_FALSE = 20,         /* TapeBuilder.java:123-137 */
    SJO_E_INVALID_NULL = 21,          /* TapeBuilder.java:147-161 */
    SJO_E_NUM_MINUS = 22,             /* NumberParser.java:34-36 */
    SJO_E_NUM_LEADING_ZERO = 23,      /* NumberParser.java:37-39 */
    SJO_E_NUM_DECIMAL_POINT = 24,     /* NumberParser.java:51-53 */
    SJO_E_NUM_EXPONENT = 25,          /* ExponentParser.java:27-29 */
    SJO_E_NUM_FOLLOWED = 26,          /* NumberParser.java:63-65 */
    SJO_E_NUM_LONG_RANGE = 27,        /* NumberParser.java:70-72 */
    SJO_E_DEPTH = 28,                 /* JsonIterator.java:69 (ArrayIndexOutOfBoundsException in the reference) */
    SJO_E_CAPACITY = 29               /* not a reference error: output arrays too small */
};

const char *sjo_error_message(int code);

/* ---- stage 1 ---------------------------------------------------------------------- */

/* Block-form restatement of StructuralIndexer.index512 (StructuralIndexer.java:196-303)
 * + BitIndexes.write/finish (BitIndexes.java:14-41,82-96).
 *  buf[0,len) is the only region read (the tail block is copied into a space-filled
 *  64-byte scratch exactly like StructuralIndexer.remainder :305-309).
 *  indexes[0..count) ascending byte offsets, indexes[count] = 0 (sentinel); needs
 *  index_capacity >= count+1, else returns -1.
 *  status: SJO_ST_UNCLOSED / SJO_ST_UNESCAPED bits (both are reported, the reference
 *  throws the first one).
 *  masks (optional, may be NULL): 6 x u64 per processed block (len/64+1 blocks):
 *  {escaped, quote, inString, op, whitespace, structurals}. */
int sjo_index_blocks(const uint8_t *buf, uint64_t len, uint32_t *indexes, uint64_t index_capacity,
                     uint64_t *count, uint32_t *status, uint64_t *masks);

/* Independent per-byte state machine with the same observable results (SURVEY.md 8(a) row a3').
 * Used to cross-check the block form; shares no code with it. */
int sjo_index_bytewise(const uint8_t *buf, uint64_t len, uint32_t *indexes, uint64_t index_capacity,
                       uint64_t *count, uint32_t *status);

/* Lookup-table UTF-8 validator, restating Utf8Validator.validate (Utf8Validator.java:54-168,
 * tables :170-249) with a chunk width of `species_bytes` (32 or 64).  Returns 1 if valid. */
int sjo_utf8_validate_lookup(const uint8_t *buf, uint64_t len, int species_bytes);

/* Independent strict RFC 3629 decoder-style validator.  Returns 1 if valid. */
int sjo_utf8_validate_strict(const uint8_t *buf, uint64_t len);

/* SimdJsonParser.stage1 (SimdJsonParser.java:55-58): validate then index.
 * status gets all applicable SJO_ST_* bits. Returns 0, or -1 on capacity. */
int sjo_stage1(const uint8_t *buf, uint64_t len, uint32_t *indexes, uint64_t index_capacity,
               uint64_t *count, uint32_t *status);

/* ---- strings ---------------------------------------------------------------------- */

/* StringParser.parseString (StringParser.java:18-23 -> doParseString :29-68):
 * unescape the string whose opening quote is at buf[idx]; write [be32 len][bytes] at
 * string_buffer[sb_idx]; return the next free index, or -(error code).
 * `buf` must have >= 64 readable bytes after the closing quote (reference padding). */
int64_t sjo_parse_string(const uint8_t *buf, uint64_t idx, uint8_t *string_buffer, uint64_t sb_idx,
                         uint64_t sb_capacity);

/* Batch form used as the oracle for the GPU unescape kernel: for every structural whose
 * byte is '"', in order, append a record; string_offsets[k] = offset of record k.
 * Stops at the first failing string: *first_error_ordinal = k, *first_error_code = code.
 * Returns total bytes written (records before the failing one). */
uint64_t sjo_unescape_all(const uint8_t *buf, const uint32_t *indexes, uint64_t count,
                          uint8_t *string_buffer, uint64_t sb_capacity, uint64_t *string_offsets,
                          uint64_t *n_strings, int64_t *first_error_ordinal, int *first_error_code);

/* ---- stage 2 (JsonIterator + TapeBuilder + Tape + number grammar) ------------------ */

typedef struct sjo_doc {
    uint64_t *tape;        /* tape words, Tape.java:28-43 */
    uint64_t tape_len;
    uint8_t *string_buffer;
    uint64_t string_len;
    int error;             /* enum sjo_error */
    uint64_t error_pos;    /* N of "Invalid value starting at N" (atoms), else 0 */
    uint32_t stage1_status;
    uint64_t n_structurals;
} sjo_doc;

/* Full SimdJsonParser.parse(byte[],int) (SimdJsonParser.java:35-40): pad, stage 1,
 * JsonIterator.walkDocument.  max_depth as in the reference ctor (default 1024).
 * Caller frees with sjo_doc_free. buf needs no padding (it is copied + padded). */
int sjo_parse(const uint8_t *buf, uint64_t len, int max_depth, sjo_doc *out);
void sjo_doc_free(sjo_doc *d);

/* timing loops for bench.py's CPU legs (see sj_oracle.c) */
typedef int (*sjo_stage1_fn)(const uint8_t *buf, uint64_t len, uint32_t *indexes, uint64_t index_capacity, uint64_t *count,
                             uint32_t *status);
uint64_t sjo_parse_many(const uint8_t *buf, const uint64_t *offsets, uint64_t n, int max_depth, int loops, sjo_stage1_fn stage1,
                        uint64_t *tape_words, uint64_t *string_bytes);
uint64_t sjo_unescape_loop(const uint8_t *padded, const uint32_t *indexes, uint64_t count, uint8_t *sb, uint64_t cap, int loops);

/* Stage 2 only, over given structural indexes (used to check the GPU stage-1 output
 * end to end).  padded_buf must have 64 bytes of padding after len. */
int sjo_stage2(const uint8_t *padded_buf, uint64_t len, const uint32_t *indexes, uint64_t count,
               int max_depth, sjo_doc *out);

/* FNV-1a 64 over a uint32 stream (cheap full-size parity digest) */
uint64_t sjo_fnv1a64_u32(const uint32_t *p, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
