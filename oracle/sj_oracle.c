/*
 * sj_oracle.c -- CPU ORACLE (test infrastructure only; see sj_oracle.h header comment).
 *
 * Plain-C restatement of the reference's stage 1 (StructuralIndexer + BitIndexes +
 * Utf8Validator), StringParser and the DOM stage 2 (JsonIterator + TapeBuilder + Tape +
 * number grammar).  Paths cited are relative to
 * /root/reference/src/main/java/org/simdjson/.
 *
 * Doubles: the reference's DoubleParser (Clinger / Eisel-Lemire / HPD slow path,
 * DoubleParser.java:79-330) is a correctly-rounded decimal->binary64 conversion that
 * saturates to +-inf / +-0 (DoubleParser.java:94-98).  glibc strtod is also correctly
 * rounded with the same saturation, so it stands in for it here.
 */
#include "sj_oracle.h"
#include "sj_tables.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* messages                                                                                   */
/* ------------------------------------------------------------------------------------------ */

const char *sjo_error_message(int code) {
    switch (code) {
    case SJO_OK: return "";
    case SJO_E_UTF8: return "The input is not valid UTF-8";
    case SJO_E_UNCLOSED_STRING: return "Unclosed string. A string is opened, but never closed.";
    case SJO_E_UNESCAPED_CHARS:
        return "Unescaped characters. Within strings, there are characters that should be escaped.";
    case SJO_E_ESCAPE_UNEXPECTED: return "Escaped unexpected character: ";
    case SJO_E_INVALID_UNICODE_ESCAPE: return "Invalid unicode escape sequence.";
    case SJO_E_LOW_SURROGATE_RESERVED:
        return "Invalid code point. The range U+DC00\xe2\x80\x93U+DFFF is reserved for low surrogate.";
    case SJO_E_LOW_SURROGATE_NO_U: return "Low surrogate should start with '\\u'";
    case SJO_E_LOW_SURROGATE_RANGE:
        return "Invalid code point. Low surrogate should be in the range U+DC00\xe2\x80\x93U+DFFF.";
    case SJO_E_NO_STRUCTURAL: return "No structural element found.";
    case SJO_E_UNCLOSED_OBJECT: return "Unclosed object. Missing '}' for starting '{'.";
    case SJO_E_UNCLOSED_ARRAY: return "Unclosed array. Missing ']' for starting '['.";
    case SJO_E_OBJECT_NO_KEY: return "Object does not start with a key";
    case SJO_E_MISSING_COLON: return "Missing colon after key in object";
    case SJO_E_KEY_MISSING: return "Key string missing at beginning of field in object";
    case SJO_E_NO_COMMA_OBJECT: return "No comma between object fields";
    case SJO_E_NO_COMMA_ARRAY: return "Missing comma between array values";
    case SJO_E_TRAILING_CONTENT:
        return "More than one JSON value at the root of the document, or extra characters at the end of the JSON!";
    case SJO_E_UNRECOGNIZED_PRIMITIVE:
        return "Unrecognized primitive. Expected: string, number, 'true', 'false' or 'null'.";
    case SJO_E_INVALID_TRUE: return "Invalid value starting at %d. Expected 'true'.";
    case SJO_E_INVALID_FALSE: return "Invalid value starting at %d. Expected 'false'.";
    case SJO_E_INVALID_NULL: return "Invalid value starting at %d. Expected 'null'.";
    case SJO_E_NUM_MINUS: return "Invalid number. Minus has to be followed by a digit.";
    case SJO_E_NUM_LEADING_ZERO: return "Invalid number. Leading zeroes are not allowed.";
    case SJO_E_NUM_DECIMAL_POINT: return "Invalid number. Decimal point has to be followed by a digit.";
    case SJO_E_NUM_EXPONENT: return "Invalid number. Exponent indicator has to be followed by a digit.";
    case SJO_E_NUM_FOLLOWED: return "Number has to be followed by a structural character or whitespace.";
    case SJO_E_NUM_LONG_RANGE:
        return "Number value is out of long range ([-9223372036854775808, 9223372036854775807]).";
    case SJO_E_DEPTH: return "ArrayIndexOutOfBoundsException (max depth exceeded)";
    case SJO_E_CAPACITY: return "capacity exceeded";
    default: return "unknown";
    }
}

/* ------------------------------------------------------------------------------------------ */
/* stage 1: block form                                                                        */
/* ------------------------------------------------------------------------------------------ */

#define EVEN_BITS 0x5555555555555555ULL /* StructuralIndexer.java:20 */
#define ODD_BITS (~EVEN_BITS)           /* :21 */

/* StructuralIndexer.java:311-319 */
static uint64_t prefix_xor(uint64_t m) {
    m ^= m << 1;
    m ^= m << 2;
    m ^= m << 4;
    m ^= m << 8;
    m ^= m << 16;
    m ^= m << 32;
    return m;
}

typedef struct {
    uint64_t prev_in_string, prev_escaped, prev_scalar, unescaped_error;
} idx_state;

typedef struct {
    uint32_t *indexes;
    uint64_t cap, write_idx;
    int overflow;
} bit_indexes;

/* BitIndexes.write (BitIndexes.java:14-41): flatten mask -> ascending offsets.  The
 * reference stores 8/16 entries speculatively; only [0,writeIdx) is meaningful, which is
 * what is reproduced here. `block_start` = blockIndex-64 of the reference call. */
static void bi_write(bit_indexes *bi, uint64_t block_start, uint64_t bits) {
    while (bits) {
        if (bi->write_idx >= bi->cap) {
            bi->overflow = 1;
            return;
        }
        bi->indexes[bi->write_idx++] = (uint32_t)(block_start + (uint64_t)__builtin_ctzll(bits));
        bits &= bits - 1;
    }
}

/* one 64-byte step of StructuralIndexer.index512 (StructuralIndexer.java:206-253) */
static uint64_t index_block(const uint8_t *c, idx_state *st, uint64_t *dbg) {
    uint64_t backslash = 0, rawquote = 0, unescaped = 0, whitespace = 0, op = 0;
    for (int i = 0; i < 64; i++) {
        uint8_t b = c[i];
        backslash |= (uint64_t)(b == '\\') << i;                         /* :210 */
        rawquote |= (uint64_t)(b == '"') << i;                           /* :232 */
        unescaped |= (uint64_t)(b <= 0x1F) << i;                         /* :231 */
        whitespace |= (uint64_t)(b == WHITESPACE_TABLE[b & 15]) << i;    /* :237-238 */
        op |= (uint64_t)((uint8_t)(b | 0x20) == OP_TABLE[b & 15]) << i;  /* :239-240 */
    }
    uint64_t escaped;
    if (backslash == 0) { /* :213-215 */
        escaped = st->prev_escaped;
        st->prev_escaped = 0;
    } else { /* :217-228 */
        backslash &= ~st->prev_escaped;
        uint64_t follows_escape = backslash << 1 | st->prev_escaped;
        uint64_t odd_sequence_starts = backslash & ODD_BITS & ~follows_escape;
        uint64_t sequences_starting_on_even_bits = odd_sequence_starts + backslash;
        st->prev_escaped =
            ((odd_sequence_starts >> 1) + (backslash >> 1) + ((odd_sequence_starts & backslash) & 1)) >> 63;
        uint64_t invert_mask = sequences_starting_on_even_bits << 1;
        escaped = (EVEN_BITS ^ invert_mask) & follows_escape;
    }
    uint64_t quote = rawquote & ~escaped;                              /* :232 */
    uint64_t in_string = prefix_xor(quote) ^ st->prev_in_string;       /* :233 */
    st->prev_in_string = (uint64_t)((int64_t)in_string >> 63);         /* :234 */
    uint64_t scalar = ~(op | whitespace);                              /* :243 */
    uint64_t non_quote_scalar = scalar & ~quote;                       /* :244 */
    uint64_t follows_nqs = non_quote_scalar << 1 | st->prev_scalar;    /* :245 */
    st->prev_scalar = non_quote_scalar >> 63;                          /* :246 */
    uint64_t potential_scalar_start = scalar & ~follows_nqs;           /* :247 */
    uint64_t potential_structural_start = op | potential_scalar_start; /* :248 */
    uint64_t structurals = potential_structural_start & ~(in_string ^ quote); /* :251 */
    st->unescaped_error |= unescaped & in_string;                      /* :252 */
    if (dbg) {
        dbg[0] = escaped;
        dbg[1] = quote;
        dbg[2] = in_string;
        dbg[3] = op;
        dbg[4] = whitespace;
        dbg[5] = structurals;
    }
    return structurals;
}

int sjo_index_blocks(const uint8_t *buf, uint64_t len, uint32_t *indexes, uint64_t index_capacity,
                     uint64_t *count, uint32_t *status, uint64_t *masks) {
    idx_state st = {0, 0, 0, 0};
    bit_indexes bi = {indexes, index_capacity, 0, 0};
    uint64_t loop_bound = len & ~(uint64_t)63; /* SPECIES_512.loopBound(length) :203 */
    uint64_t off = 0, blk = 0;
    for (; off < loop_bound; off += 64, blk++) {
        uint64_t s = index_block(buf + off, &st, masks ? masks + 6 * blk : NULL);
        bi_write(&bi, off, s); /* emission is delayed one block in the reference (:249-251); same output */
    }
    /* tail: StructuralIndexer.remainder :305-309 -- space-filled scratch, always processed */
    uint8_t last[64];
    memset(last, 0x20, 64);
    if (len > off) memcpy(last, buf + off, (size_t)(len - off));
    uint64_t s = index_block(last, &st, masks ? masks + 6 * blk : NULL);
    bi_write(&bi, off, s);
    if (bi.overflow || bi.write_idx >= bi.cap) return -1;
    indexes[bi.write_idx] = 0; /* BitIndexes.finish :82-96 */
    *count = bi.write_idx;
    uint32_t stt = 0;
    if (st.prev_in_string != 0) stt |= SJO_ST_UNCLOSED;    /* :297-299 */
    if (st.unescaped_error != 0) stt |= SJO_ST_UNESCAPED;  /* :300-302 */
    *status = stt;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* stage 1: independent per-byte state machine (SURVEY.md 8(a) a3')                            */
/* ------------------------------------------------------------------------------------------ */

int sjo_index_bytewise(const uint8_t *buf, uint64_t len, uint32_t *indexes, uint64_t index_capacity,
                       uint64_t *count, uint32_t *status) {
    int esc = 0, in_str = 0, prev_nqs = 0, unesc_err = 0;
    uint64_t n = 0;
    for (uint64_t i = 0; i < len; i++) {
        uint8_t c = buf[i];
        int escaped = esc;
        esc = (c == '\\' && !escaped);
        int q = (c == '"' && !escaped);
        if (q) in_str ^= 1;
        int ws = (c == 0x20 || c == 0x09 || c == 0x0A || c == 0x0D);
        int op = (c == ',' || c == ':' || c == '[' || c == ']' || c == '{' || c == '}' || c == 0x0C || c == 0x1A);
        int scalar = !(op || ws);
        int start = scalar && !prev_nqs;
        prev_nqs = scalar && !q;
        if ((op || start) && !(in_str ^ q)) {
            if (n >= index_capacity) return -1;
            indexes[n++] = (uint32_t)i;
        }
        if (c <= 0x1F && in_str) unesc_err = 1;
    }
    if (n >= index_capacity) return -1;
    indexes[n] = 0;
    *count = n;
    *status = (in_str ? SJO_ST_UNCLOSED : 0) | (unesc_err ? SJO_ST_UNESCAPED : 0);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* UTF-8: lookup-table form (Utf8Validator.java)                                              */
/* ------------------------------------------------------------------------------------------ */

/* one chunk of W bytes; prev4 = the previous chunk's last four bytes (prev4[3] is the
 * byte right before chunk[0]).  Mirrors Utf8Validator.java:68-110. Returns error!=0 and
 * sets *incomplete per :68 / :170-180. */
static int utf8_chunk(const uint8_t *chunk, int W, const uint8_t prev4[4], int *incomplete) {
    int err = 0, inc = 0;
    for (int i = 0; i < W; i++) {
        uint8_t cur = chunk[i];
        uint8_t p1 = i >= 1 ? chunk[i - 1] : prev4[3 + i];      /* previousOneByte   :77-80 */
        uint8_t p2 = i >= 2 ? chunk[i - 2] : prev4[2 + i];      /* previousTwoBytes  :95-98 */
        uint8_t p3 = i >= 3 ? chunk[i - 3] : prev4[1 + i];      /* previousThreeBytes:101-104 */
        uint8_t first = BYTE_1_HIGH[p1 >> 4] & BYTE_1_LOW[p1 & 15] & BYTE_2_HIGH[cur >> 4]; /* :81-92 */
        int must23 = (p2 > 0xDF) || (p3 > 0xEF);                /* :100,:106 */
        uint8_t second = (uint8_t)(first + (must23 ? 0x80 : 0)); /* :109 */
        if (second != 0) err = 1;                               /* :110 */
        uint8_t lim = 0xFF;                                     /* INCOMPLETE_CHECK :170-180 */
        if (i == W - 3) lim = 0xF0;
        if (i == W - 2) lim = 0xE0;
        if (i == W - 1) lim = 0xC0;
        if (cur >= lim) inc = 1;
    }
    *incomplete = inc;
    return err;
}

int sjo_utf8_validate_lookup(const uint8_t *buf, uint64_t len, int W) {
    int previous_incomplete = 0, errors = 0;
    uint8_t prev4[4] = {0, 0, 0, 0};
    uint64_t loop_bound = len - (len % (uint64_t)W);
    uint64_t off = 0;
    for (; off < loop_bound; off += (uint64_t)W) {
        const uint8_t *chunk = buf + off;
        int ascii = 1;
        for (int i = 0; i < W; i++) ascii &= !(chunk[i] & 0x80);
        if (ascii) {
            errors |= previous_incomplete; /* :65-66 */
        } else {
            errors |= utf8_chunk(chunk, W, prev4, &previous_incomplete);
        }
        memcpy(prev4, chunk + W - 4, 4); /* :112 */
    }
    uint8_t tail[64];
    memset(tail, 0, sizeof tail); /* zero-padded masked load :115-117 */
    if (len > off) memcpy(tail, buf + off, (size_t)(len - off));
    int ascii = 1;
    for (int i = 0; i < W; i++) ascii &= !(tail[i] & 0x80);
    if (!ascii) errors |= utf8_chunk(tail, W, prev4, &previous_incomplete); /* :118-163 */
    return !(errors | previous_incomplete);                                 /* :165 */
}

/* independent strict validator (RFC 3629 table 3-7 of the Unicode standard) */
int sjo_utf8_validate_strict(const uint8_t *s, uint64_t n) {
    uint64_t i = 0;
    while (i < n) {
        uint8_t c = s[i];
        if (c < 0x80) { i++; continue; }
        if (c >= 0xC2 && c <= 0xDF) {
            if (i + 1 >= n || (s[i + 1] & 0xC0) != 0x80) return 0;
            i += 2;
        } else if (c >= 0xE0 && c <= 0xEF) {
            if (i + 2 >= n) return 0;
            uint8_t lo = 0x80, hi = 0xBF;
            if (c == 0xE0) lo = 0xA0;
            if (c == 0xED) hi = 0x9F;
            if (s[i + 1] < lo || s[i + 1] > hi) return 0;
            if ((s[i + 2] & 0xC0) != 0x80) return 0;
            i += 3;
        } else if (c >= 0xF0 && c <= 0xF4) {
            if (i + 3 >= n) return 0;
            uint8_t lo = 0x80, hi = 0xBF;
            if (c == 0xF0) lo = 0x90;
            if (c == 0xF4) hi = 0x8F;
            if (s[i + 1] < lo || s[i + 1] > hi) return 0;
            if ((s[i + 2] & 0xC0) != 0x80 || (s[i + 3] & 0xC0) != 0x80) return 0;
            i += 4;
        } else {
            return 0;
        }
    }
    return 1;
}

int sjo_stage1(const uint8_t *buf, uint64_t len, uint32_t *indexes, uint64_t index_capacity,
               uint64_t *count, uint32_t *status) {
    uint32_t st = 0;
    int r = sjo_index_blocks(buf, len, indexes, index_capacity, count, &st, NULL);
    if (r) return r;
    if (!sjo_utf8_validate_lookup(buf, len, 64)) st |= SJO_ST_UTF8;
    *status = st;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* strings                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* CharacterUtils.escape (CharacterUtils.java:52-83): returns 0 for "unexpected" */
static uint8_t escape_map(uint8_t e) {
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

/* CharacterUtils.hexToInt (CharacterUtils.java:241-247): negative if any digit is bad */
static int32_t hex4(const uint8_t *p) {
    int32_t v = 0;
    for (int i = 0; i < 4; i++) {
        uint8_t c = p[i];
        int32_t d;
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else return -1;
        v = (v << 4) | d;
    }
    return v;
}

/* StringParser.doParseString (StringParser.java:29-68), byte-wise: the vector loop finds
 * the first '"' or '\\' in each chunk (:155-161), which is what this loop does one byte
 * at a time. The over-copy past the closing quote (:33-34) is outside the parity domain. */
int64_t sjo_parse_string(const uint8_t *buf, uint64_t idx, uint8_t *sb, uint64_t sb_idx, uint64_t cap) {
    uint64_t src = idx + 1;
    uint64_t dst = sb_idx + 4;
    for (;;) {
        uint8_t c = buf[src];
        if (c == '"') break; /* :38-41 */
        if (dst + 4 > cap) return -(int64_t)SJO_E_CAPACITY;
        if (c == '\\') {     /* :42-62 */
            uint8_t e = buf[src + 1];
            if (e == 'u') {
                int32_t cp = hex4(buf + src + 2); /* :48 */
                src += 6;
                if (cp >= 0xD800 && cp <= 0xDBFF) { /* :50-52 -> parseLowSurrogate :112-124 */
                    if (!(buf[src] == '\\' && buf[src + 1] == 'u')) return -(int64_t)SJO_E_LOW_SURROGATE_NO_U;
                    int32_t cp2 = hex4(buf + src + 2);
                    int32_t low = cp2 - 0xDC00;
                    if ((low >> 10) == 0) cp = (((cp - 0xD800) << 10) | low) + 0x10000;
                    else return -(int64_t)SJO_E_LOW_SURROGATE_RANGE;
                    src += 6;
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) { /* :53-55 */
                    return -(int64_t)SJO_E_LOW_SURROGATE_RESERVED;
                }
                /* storeCodePointInStringBuffer :126-153 */
                if (cp < 0) return -(int64_t)SJO_E_INVALID_UNICODE_ESCAPE;
                if (cp <= 0x7F) {
                    sb[dst++] = (uint8_t)cp;
                } else if (cp <= 0x7FF) {
                    sb[dst++] = (uint8_t)((cp >> 6) + 192);
                    sb[dst++] = (uint8_t)((cp & 63) + 128);
                } else if (cp <= 0xFFFF) {
                    sb[dst++] = (uint8_t)((cp >> 12) + 224);
                    sb[dst++] = (uint8_t)(((cp >> 6) & 63) + 128);
                    sb[dst++] = (uint8_t)((cp & 63) + 128);
                } else {
                    sb[dst++] = (uint8_t)((cp >> 18) + 240);
                    sb[dst++] = (uint8_t)(((cp >> 12) & 63) + 128);
                    sb[dst++] = (uint8_t)(((cp >> 6) & 63) + 128);
                    sb[dst++] = (uint8_t)((cp & 63) + 128);
                }
            } else {
                uint8_t r = (e & 0x80) ? 0 : escape_map(e); /* :58, CharacterUtils.java:74-83 */
                if (r == 0) return -(int64_t)SJO_E_ESCAPE_UNEXPECTED;
                sb[dst++] = r;
                src += 2;
            }
        } else {
            sb[dst++] = c;
            src++;
        }
    }
    uint32_t len = (uint32_t)(dst - sb_idx - 4); /* :20-21, IntegerUtils.toBytes :12-17 */
    sb[sb_idx] = (uint8_t)(len >> 24);
    sb[sb_idx + 1] = (uint8_t)(len >> 16);
    sb[sb_idx + 2] = (uint8_t)(len >> 8);
    sb[sb_idx + 3] = (uint8_t)len;
    return (int64_t)dst;
}

uint64_t sjo_unescape_all(const uint8_t *buf, const uint32_t *indexes, uint64_t count, uint8_t *sb,
                          uint64_t cap, uint64_t *string_offsets, uint64_t *n_strings,
                          int64_t *first_error_ordinal, int *first_error_code) {
    uint64_t k = 0, pos = 0;
    *first_error_ordinal = -1;
    *first_error_code = 0;
    for (uint64_t i = 0; i < count; i++) {
        if (buf[indexes[i]] != '"') continue;
        int64_t r = sjo_parse_string(buf, indexes[i], sb, pos, cap);
        if (r < 0) {
            *first_error_ordinal = (int64_t)k;
            *first_error_code = (int)(-r);
            break;
        }
        if (string_offsets) string_offsets[k] = pos;
        pos = (uint64_t)r;
        k++;
    }
    *n_strings = k;
    return pos;
}

/* ------------------------------------------------------------------------------------------ */
/* stage 2                                                                                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    uint64_t *w;
    uint64_t idx, cap;
} tape_t;

typedef struct {
    const uint8_t *buf;
    uint64_t len;
    const uint32_t *ix;
    uint64_t n, rd; /* BitIndexes cursor: n = writeIdx, rd = readIdx */
    tape_t tape;
    uint8_t *sb;
    uint64_t sb_idx, sb_cap;
    int err;
    uint64_t err_pos; /* the 'starting at N' of the atom messages */
} s2_t;

/* cursor (BitIndexes.java:47-80); reads past the sentinel are defined as 0 here */
static uint32_t ix_at(const s2_t *s, uint64_t i) { return i < s->n ? s->ix[i] : 0; }
static uint32_t ix_get_and_advance(s2_t *s) { return ix_at(s, s->rd++); }
static uint32_t ix_peek(const s2_t *s) { return ix_at(s, s->rd); }
static uint32_t ix_get_last(const s2_t *s) { return s->ix[s->n - 1]; }

static int tape_room(s2_t *s, uint64_t k) {
    if (s->tape.idx + k > s->tape.cap) {
        s->err = SJO_E_CAPACITY;
        return 0;
    }
    return 1;
}
/* Tape.append (Tape.java:28-31) */
static void tape_append(s2_t *s, uint64_t val, char type) {
    if (!tape_room(s, 1)) return;
    s->tape.w[s->tape.idx++] = val | ((uint64_t)(uint8_t)type << 56);
}
static void tape_write(s2_t *s, uint64_t i, uint64_t val, char type) { /* Tape.write :45-47 */
    s->tape.w[i] = val | ((uint64_t)(uint8_t)type << 56);
}

/* CharacterUtils.isStructuralOrWhitespace (CharacterUtils.java:6-50) */
static int is_structural_or_ws(uint8_t b) {
    switch (b) {
    case 0x09: case 0x0A: case 0x0D: case 0x20: case ',': case ':': case '[': case ']': case '{': case '}':
        return 1;
    default:
        return 0;
    }
}

/* NumberParser.parseNumber (NumberParser.java:23-74) + ExponentParser.parse (:14-69) +
 * isOutOfLongRange (:313-328).  p points at the first char of the number inside a buffer
 * that is readable (and terminated by a structural/whitespace/other byte) past the token. */
static void parse_number(s2_t *s, const uint8_t *p) {
    const uint8_t *start = p;
    int negative = (*p == '-');
    if (negative) p++;
    const uint8_t *digits_start = p;
    uint64_t digits = 0;
    while ((uint8_t)(*p - '0') <= 9) { /* parseDigits :330-338 */
        digits = 10 * digits + (uint64_t)(*p - '0');
        p++;
    }
    int64_t digit_count = p - digits_start;
    if (digit_count == 0) { s->err = SJO_E_NUM_MINUS; return; }
    if (*digits_start == '0' && digit_count > 1) { s->err = SJO_E_NUM_LEADING_ZERO; return; }
    int floating = 0;
    if (*p == '.') { /* :43-55 */
        floating = 1;
        p++;
        const uint8_t *after = p;
        while ((uint8_t)(*p - '0') <= 9) p++;
        if (p == after) { s->err = SJO_E_NUM_DECIMAL_POINT; return; }
    }
    if (*p == 'e' || *p == 'E') { /* :56-62 -> ExponentParser.parse */
        floating = 1;
        p++;
        if (*p == '-' || *p == '+') p++;
        const uint8_t *es = p;
        while ((uint8_t)(*p - '0') <= 9) p++;
        if (p == es) { s->err = SJO_E_NUM_EXPONENT; return; }
    }
    if (!is_structural_or_ws(*p)) { s->err = SJO_E_NUM_FOLLOWED; return; } /* :63-65 */
    if (floating) {
        /* DoubleParser.parse stand-in: correctly rounded, saturating (see file header) */
        size_t n = (size_t)(p - start);
        char tmp[64];
        char *t = n + 1 <= sizeof tmp ? tmp : (char *)malloc(n + 1);
        memcpy(t, start, n);
        t[n] = 0;
        double v = strtod(t, NULL);
        if (t != tmp) free(t);
        uint64_t bits;
        memcpy(&bits, &v, 8);
        if (!tape_room(s, 2)) return;
        tape_append(s, 0, 'd'); /* Tape.appendDouble :39-43 */
        s->tape.w[s->tape.idx++] = bits;
    } else {
        /* isOutOfLongRange :313-328 */
        int out = 0;
        if (digit_count > 19) out = 1;
        else if (digit_count == 19) {
            if (negative && digits == 0x8000000000000000ULL) out = 0;
            else out = ((int64_t)digits < 0);
        }
        if (out) { s->err = SJO_E_NUM_LONG_RANGE; return; }
        if (!tape_room(s, 2)) return;
        tape_append(s, 0, 'l'); /* Tape.appendInt64 :33-37 */
        s->tape.w[s->tape.idx++] = negative ? (~digits + 1) : digits;
    }
}

static void visit_string(s2_t *s, uint32_t idx) { /* TapeBuilder.visitString :174-177 */
    tape_append(s, s->sb_idx, '"');
    if (s->err) return;
    int64_t r = sjo_parse_string(s->buf, idx, s->sb, s->sb_idx, s->sb_cap);
    if (r < 0) { s->err = (int)(-r); return; }
    s->sb_idx = (uint64_t)r;
}

static int is_true(const uint8_t *b) { return b[0] == 't' && b[1] == 'r' && b[2] == 'u' && b[3] == 'e'; }
static int is_false(const uint8_t *b) { return b[0] == 'f' && b[1] == 'a' && b[2] == 'l' && b[3] == 's' && b[4] == 'e'; }
static int is_null(const uint8_t *b) { return b[0] == 'n' && b[1] == 'u' && b[2] == 'l' && b[3] == 'l'; }

/* TapeBuilder.visitPrimitive :70-79 */
static void visit_primitive(s2_t *s, uint32_t idx) {
    const uint8_t *b = s->buf + idx;
    switch (*b) {
    case '"': visit_string(s, idx); break;
    case 't': /* visitTrueAtom :100-106 */
        if (!(is_true(b) && is_structural_or_ws(b[4]))) { s->err = SJO_E_INVALID_TRUE; s->err_pos = idx; return; }
        tape_append(s, 0, 't');
        break;
    case 'f': /* visitFalseAtom :123-129 */
        if (!(is_false(b) && is_structural_or_ws(b[5]))) { s->err = SJO_E_INVALID_FALSE; s->err_pos = idx; return; }
        tape_append(s, 0, 'f');
        break;
    case 'n': /* visitNullAtom :147-153 */
        if (!(is_null(b) && is_structural_or_ws(b[4]))) { s->err = SJO_E_INVALID_NULL; s->err_pos = idx; return; }
        tape_append(s, 0, 'n');
        break;
    case '-': case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': case '8': case '9':
        parse_number(s, b); /* visitNumber :179-181 */
        break;
    default: s->err = SJO_E_UNRECOGNIZED_PRIMITIVE;
    }
}

/* TapeBuilder.visitRootPrimitive :59-68 */
static void visit_root_primitive(s2_t *s, uint32_t idx) {
    const uint8_t *b = s->buf + idx;
    uint64_t len = s->len;
    switch (*b) {
    case '"': visit_string(s, idx); break;
    case 't': /* visitRootTrueAtom :108-114 */
        if (!(idx + 4 <= len && is_true(b) && (idx + 4 == len || is_structural_or_ws(b[4])))) { s->err = SJO_E_INVALID_TRUE; s->err_pos = idx; return; }
        tape_append(s, 0, 't');
        break;
    case 'f': /* :131-137 */
        if (!(idx + 5 <= len && is_false(b) && (idx + 5 == len || is_structural_or_ws(b[5])))) { s->err = SJO_E_INVALID_FALSE; s->err_pos = idx; return; }
        tape_append(s, 0, 'f');
        break;
    case 'n': /* :155-161 */
        if (!(idx + 4 <= len && is_null(b) && (idx + 4 == len || is_structural_or_ws(b[4])))) { s->err = SJO_E_INVALID_NULL; s->err_pos = idx; return; }
        tape_append(s, 0, 'n');
        break;
    case '-': case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': case '8': case '9': {
        /* visitRootNumber :183-189: copy [idx,len) and pad with 64 spaces */
        uint64_t rem = len - idx;
        uint8_t *copy = (uint8_t *)malloc(rem + 64);
        memcpy(copy, b, rem);
        memset(copy + rem, 0x20, 64);
        parse_number(s, copy);
        free(copy);
        break;
    }
    default: s->err = SJO_E_UNRECOGNIZED_PRIMITIVE;
    }
}

typedef struct { uint64_t tape_index; uint32_t count; } open_container;

/* TapeBuilder.emptyContainer :205-208 */
static void empty_container(s2_t *s, char start, char end) {
    tape_append(s, s->tape.idx + 2, start);
    tape_append(s, s->tape.idx, end);
}

enum { OBJECT_BEGIN, ARRAY_BEGIN, DOCUMENT_END, OBJECT_FIELD, OBJECT_CONTINUE, SCOPE_END, ARRAY_CONTINUE, ARRAY_VALUE };

#define FAIL(code) do { s->err = (code); goto done; } while (0)
#define CHECK() do { if (s->err) goto done; } while (0)

/* JsonIterator.walkDocument (JsonIterator.java:26-200) */
static void walk_document(s2_t *s, int max_depth) {
    uint8_t *is_array = (uint8_t *)calloc((size_t)max_depth, 1);
    open_container *oc = (open_container *)calloc((size_t)max_depth, sizeof *oc);
    const uint8_t *buf = s->buf;
    if (s->rd == s->n) FAIL(SJO_E_NO_STRUCTURAL); /* :27-29 */
    /* visitDocumentStart -> startContainer(0) TapeBuilder.java:41-43,191-195 */
    oc[0].tape_index = s->tape.idx;
    oc[0].count = 0;
    if (!tape_room(s, 1)) goto done;
    s->tape.idx++;
    int depth = 0, state;
    uint32_t idx = ix_get_and_advance(s);
    switch (buf[idx]) { /* :36-66 */
    case '{':
        if (buf[ix_get_last(s)] != '}') FAIL(SJO_E_UNCLOSED_OBJECT);
        if (buf[ix_peek(s)] == '}') { s->rd++; empty_container(s, '{', '}'); state = DOCUMENT_END; }
        else state = OBJECT_BEGIN;
        break;
    case '[':
        if (buf[ix_get_last(s)] != ']') FAIL(SJO_E_UNCLOSED_ARRAY);
        if (buf[ix_peek(s)] == ']') { s->rd++; empty_container(s, '[', ']'); state = DOCUMENT_END; }
        else state = ARRAY_BEGIN;
        break;
    default:
        visit_root_primitive(s, idx);
        state = DOCUMENT_END;
    }
    CHECK();
    while (state != DOCUMENT_END) {
        if (state == OBJECT_BEGIN) { /* :69-81 */
            depth++;
            if (depth >= max_depth) FAIL(SJO_E_DEPTH);
            is_array[depth] = 0;
            oc[depth].tape_index = s->tape.idx; /* visitObjectStart -> startContainer */
            oc[depth].count = 0;
            if (!tape_room(s, 1)) goto done;
            s->tape.idx++;
            uint32_t key = ix_get_and_advance(s);
            if (buf[key] != '"') FAIL(SJO_E_OBJECT_NO_KEY);
            oc[depth].count++;
            visit_string(s, key);
            CHECK();
            state = OBJECT_FIELD;
        }
        if (state == OBJECT_FIELD) { /* :83-115 */
            if (buf[ix_get_and_advance(s)] != ':') FAIL(SJO_E_MISSING_COLON);
            idx = ix_get_and_advance(s);
            switch (buf[idx]) {
            case '{':
                if (buf[ix_peek(s)] == '}') { s->rd++; empty_container(s, '{', '}'); state = OBJECT_CONTINUE; }
                else state = OBJECT_BEGIN;
                break;
            case '[':
                if (buf[ix_peek(s)] == ']') { s->rd++; empty_container(s, '[', ']'); state = OBJECT_CONTINUE; }
                else state = ARRAY_BEGIN;
                break;
            default:
                visit_primitive(s, idx);
                state = OBJECT_CONTINUE;
            }
            CHECK();
        }
        if (state == OBJECT_CONTINUE) { /* :117-133 */
            switch (buf[ix_get_and_advance(s)]) {
            case ',': {
                oc[depth].count++;
                uint32_t key = ix_get_and_advance(s);
                if (buf[key] != '"') FAIL(SJO_E_KEY_MISSING);
                visit_string(s, key);
                CHECK();
                state = OBJECT_FIELD;
                break;
            }
            case '}': { /* visitObjectEnd -> endContainer TapeBuilder.java:197-203 */
                uint64_t st = oc[depth].tape_index;
                tape_append(s, st, '}');
                CHECK();
                uint32_t cnt = oc[depth].count > 0xFFFFFF ? 0xFFFFFF : oc[depth].count;
                tape_write(s, st, s->tape.idx | ((uint64_t)cnt << 32), '{');
                state = SCOPE_END;
                break;
            }
            default: FAIL(SJO_E_NO_COMMA_OBJECT);
            }
        }
        if (state == SCOPE_END) { /* :135-144 */
            depth--;
            if (depth == 0) state = DOCUMENT_END;
            else if (is_array[depth]) state = ARRAY_CONTINUE;
            else state = OBJECT_CONTINUE;
        }
        if (state == ARRAY_BEGIN) { /* :146-152 */
            depth++;
            if (depth >= max_depth) FAIL(SJO_E_DEPTH);
            is_array[depth] = 1;
            oc[depth].tape_index = s->tape.idx;
            oc[depth].count = 0;
            if (!tape_room(s, 1)) goto done;
            s->tape.idx++;
            oc[depth].count++;
            state = ARRAY_VALUE;
        }
        if (state == ARRAY_VALUE) { /* :154-181 */
            idx = ix_get_and_advance(s);
            switch (buf[idx]) {
            case '{':
                if (buf[ix_peek(s)] == '}') { s->rd++; empty_container(s, '{', '}'); state = ARRAY_CONTINUE; }
                else state = OBJECT_BEGIN;
                break;
            case '[':
                if (buf[ix_peek(s)] == ']') { s->rd++; empty_container(s, '[', ']'); state = ARRAY_CONTINUE; }
                else state = ARRAY_BEGIN;
                break;
            default:
                visit_primitive(s, idx);
                state = ARRAY_CONTINUE;
            }
            CHECK();
        }
        if (state == ARRAY_CONTINUE) { /* :183-191 */
            switch (buf[ix_get_and_advance(s)]) {
            case ',':
                oc[depth].count++;
                state = ARRAY_VALUE;
                break;
            case ']': {
                uint64_t st = oc[depth].tape_index;
                tape_append(s, st, ']');
                CHECK();
                uint32_t cnt = oc[depth].count > 0xFFFFFF ? 0xFFFFFF : oc[depth].count;
                tape_write(s, st, s->tape.idx | ((uint64_t)cnt << 32), '[');
                state = SCOPE_END;
                break;
            }
            default: FAIL(SJO_E_NO_COMMA_ARRAY);
            }
        }
    }
    /* visitDocumentEnd TapeBuilder.java:45-48 */
    tape_append(s, 0, 'r');
    CHECK();
    tape_write(s, 0, s->tape.idx, 'r');
    if (s->rd != s->n) FAIL(SJO_E_TRAILING_CONTENT); /* :196-198 */
done:
    free(is_array);
    free(oc);
}

int sjo_stage2(const uint8_t *padded_buf, uint64_t len, const uint32_t *indexes, uint64_t count,
               int max_depth, sjo_doc *out) {
    s2_t s;
    memset(&s, 0, sizeof s);
    s.buf = padded_buf;
    s.len = len;
    s.ix = indexes;
    s.n = count;
    s.tape.cap = 2 * count + 8;      /* every structural emits <= 2 tape words, + 2 root words */
    s.tape.w = (uint64_t *)malloc(s.tape.cap * 8);
    s.sb_cap = len + 4 * count + 64; /* sum(4+len_k) <= len + 4*#strings */
    s.sb = (uint8_t *)malloc(s.sb_cap);
    walk_document(&s, max_depth);
    out->tape = s.tape.w;
    out->tape_len = s.err ? 0 : s.tape.idx;
    out->string_buffer = s.sb;
    out->string_len = s.err ? 0 : s.sb_idx;
    out->error = s.err;
    out->error_pos = s.err_pos;
    out->n_structurals = count;
    return s.err;
}

int sjo_parse(const uint8_t *buf, uint64_t len, int max_depth, sjo_doc *out) {
    memset(out, 0, sizeof *out);
    /* padIfNeeded (SimdJsonParser.java:42-48): guarantee 64 readable bytes after len */
    uint8_t *padded = (uint8_t *)calloc(len + 64, 1);
    memcpy(padded, buf, len);
    uint64_t cap = len + 2, count = 0;
    uint32_t *ix = (uint32_t *)malloc(cap * 4);
    uint32_t st = 0;
    sjo_stage1(padded, len, ix, cap, &count, &st);
    out->stage1_status = st;
    out->n_structurals = count;
    if (st & SJO_ST_UTF8) out->error = SJO_E_UTF8;
    else if (st & SJO_ST_UNCLOSED) out->error = SJO_E_UNCLOSED_STRING;
    else if (st & SJO_ST_UNESCAPED) out->error = SJO_E_UNESCAPED_CHARS;
    else sjo_stage2(padded, len, ix, count, max_depth, out);
    out->stage1_status = st;
    free(ix);
    free(padded);
    return out->error;
}

/* bench.py's CPU legs (timing only): the documents buf[offsets[k], offsets[k + 1]) parsed one after the other, `loops`
 * times, with the scalar stage 1 above or with the one handed in (the AVX-512 restatement lives in its own library).
 * The buffer needs 64 readable bytes behind its end.  Returns the documents of the last loop that parsed without error;
 * *tape_words / *string_bytes = their totals. */
uint64_t sjo_parse_many(const uint8_t *buf, const uint64_t *offsets, uint64_t n, int max_depth, int loops, sjo_stage1_fn stage1,
                        uint64_t *tape_words, uint64_t *string_bytes) {
    uint64_t maxlen = 0, ok = 0;
    for (uint64_t k = 0; k < n; k++)
        if (offsets[k + 1] - offsets[k] > maxlen) maxlen = offsets[k + 1] - offsets[k];
    uint32_t *ix = (uint32_t *)malloc((maxlen + 2 + 64) * 4);
    uint8_t *padded = (uint8_t *)calloc(maxlen + 128, 1);
    for (int l = 0; l < loops; l++) {
        ok = 0;
        *tape_words = 0;
        *string_bytes = 0;
        for (uint64_t k = 0; k < n; k++) {
            const uint64_t len = offsets[k + 1] - offsets[k];
            /* padIfNeeded (SimdJsonParser.java:42-48): a document inside a batch has its neighbours, not padding, behind it */
            memcpy(padded, buf + offsets[k], len);
            memset(padded + len, 0, 64);
            uint64_t count = 0;
            uint32_t st = 0;
            sjo_doc d;
            memset(&d, 0, sizeof d);
            if (stage1) stage1(padded, len, ix, maxlen + 2 + 64, &count, &st);
            else sjo_stage1(padded, len, ix, maxlen + 2, &count, &st);
            if (st == 0 && sjo_stage2(padded, len, ix, count, max_depth, &d) == 0) {
                ok++;
                *tape_words += d.tape_len;
                *string_bytes += d.string_len;
            }
            sjo_doc_free(&d);
        }
    }
    free(ix);
    free(padded);
    return ok;
}

/* ... and StringParser.parseString for every string of one indexed document, `loops` times; returns the bytes of records */
uint64_t sjo_unescape_loop(const uint8_t *padded, const uint32_t *indexes, uint64_t count, uint8_t *sb, uint64_t cap, int loops) {
    uint64_t total = 0, ns = 0;
    int64_t feo = 0;
    int fec = 0;
    for (int l = 0; l < loops; l++) total = sjo_unescape_all(padded, indexes, count, sb, cap, NULL, &ns, &feo, &fec);
    return total;
}

void sjo_doc_free(sjo_doc *d) {
    free(d->tape);
    free(d->string_buffer);
    d->tape = NULL;
    d->string_buffer = NULL;
}

uint64_t sjo_fnv1a64_u32(const uint32_t *p, uint64_t n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}

/* does this host CPU run oracle/sj_avx512.c (AVX-512 F + BW, BMI1/2, POPCNT)?  The timing baseline asks before it
 * loads liboracle_avx512.so. */
int sjo_avx512_supported(void) {
#if defined(__x86_64__)
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("bmi") &&
           __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt");
#else
    return 0;
#endif
}

/* ---- per-document DIGESTS: every document of a large batch against the engine, not a sample (round 5) ------------------------
 * A tape is position-independent except for its STRING words, whose payload is an offset into whichever string buffer the
 * producer used (the oracle's is per document, the engine's one buffer per batch).  The digest walks the words linearly
 * (Tape.java:28-43): FNV-1a over every word as it is -- type, container payloads (positions inside the document's own tape),
 * element counts, the raw second word of a number -- and, for a STRING word, over its type byte, the record's be32 length and
 * its bytes instead of the offset.  Equal digests <=> equal trees in the sense of SURVEY.md 8(c) (type, int64, raw double bits,
 * UTF-8 bytes, order, sizes) and equal word-for-word tape layout. */
static uint64_t fnv_mix(uint64_t h, const uint8_t *p, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}
uint64_t sjo_tape_digest(const uint64_t *tape, uint64_t n, const uint8_t *strings, uint64_t strings_len) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t w = tape[i];
        const uint8_t t = (uint8_t)(w >> 56);
        if (t == '"') {
            const uint64_t off = w & 0x00FFFFFFFFFFFFFFull;
            if (off + 4 > strings_len) return ~0ull - 1;  /* (a record outside the buffer: never equal to a real digest) */
            const uint32_t len = ((uint32_t)strings[off] << 24) | ((uint32_t)strings[off + 1] << 16) | ((uint32_t)strings[off + 2] << 8) | strings[off + 3];
            if (off + 4 + (uint64_t)len > strings_len) return ~0ull - 1;
            h = fnv_mix(h, &t, 1);
            h = fnv_mix(h, strings + off, 4 + (uint64_t)len);
        } else {
            h = fnv_mix(h, (const uint8_t *)&w, 8);
            if ((t == 'l' || t == 'd') && i + 1 < n) {
                ++i;
                h = fnv_mix(h, (const uint8_t *)&tape[i], 8);
            }
        }
    }
    return h;
}
/* documents [0, n) of a packed batch through the restatement (scalar stage 1 + stage 2): digests[k] (0 for a failing document),
 * errors[k] (enum sjo_error; stage-1 verdicts included the way sjo_parse reports them), idx_hash[k] = FNV-1a of the document's
 * structural indexes relative to its first byte, counts[k] = their number */
void sjo_digest_many(const uint8_t *buf, const uint64_t *offsets, uint64_t n, int max_depth, uint64_t *digests, int32_t *errors,
                     uint64_t *idx_hash, uint32_t *counts) {
    uint64_t maxlen = 0;
    for (uint64_t k = 0; k < n; k++)
        if (offsets[k + 1] - offsets[k] > maxlen) maxlen = offsets[k + 1] - offsets[k];
    uint32_t *ix = (uint32_t *)malloc((maxlen + 2 + 64) * 4);
    uint8_t *padded = (uint8_t *)calloc(maxlen + 128, 1);
    for (uint64_t k = 0; k < n; k++) {
        const uint64_t len = offsets[k + 1] - offsets[k];
        memcpy(padded, buf + offsets[k], len);
        memset(padded + len, 0, 64);
        uint64_t count = 0;
        uint32_t st = 0;
        sjo_doc d;
        memset(&d, 0, sizeof d);
        sjo_stage1(padded, len, ix, maxlen + 2, &count, &st);
        counts[k] = (uint32_t)count;
        idx_hash[k] = fnv_mix(0xCBF29CE484222325ull, (const uint8_t *)ix, count * 4);
        digests[k] = 0;
        if (st != 0) {
            errors[k] = (st & 1) ? SJO_E_UTF8 : ((st & 2) ? SJO_E_UNCLOSED_STRING : SJO_E_UNESCAPED_CHARS);
        } else {
            errors[k] = sjo_stage2(padded, len, ix, count, max_depth, &d);
            if (errors[k] == 0) digests[k] = sjo_tape_digest(d.tape, d.tape_len, d.string_buffer, d.string_len);
        }
        sjo_doc_free(&d);
    }
    free(ix);
    free(padded);
}
/* the engine's outputs of a batch, the same way: digests of the tapes tape[tape_offsets[k], tape_offsets[k + 1]) over ONE string
 * buffer, hashes of the index ranges indexes[index_offsets[k], index_offsets[k + 1]) made relative to doc_offsets[k] */
void sjo_digest_outputs(const uint64_t *tape, const uint64_t *tape_offsets, const uint8_t *strings, uint64_t strings_len,
                        const uint32_t *indexes, const uint64_t *index_offsets, const uint64_t *doc_offsets, uint64_t n,
                        const int32_t *errors, uint64_t *digests, uint64_t *idx_hash, uint32_t *counts) {
    for (uint64_t k = 0; k < n; k++) {
        digests[k] = errors[k] == 0 ? sjo_tape_digest(tape + tape_offsets[k], tape_offsets[k + 1] - tape_offsets[k], strings, strings_len) : 0;
        uint64_t h = 0xCBF29CE484222325ull;
        const uint32_t base = (uint32_t)doc_offsets[k];
        for (uint64_t i = index_offsets[k]; i < index_offsets[k + 1]; i++) {
            const uint32_t rel = indexes[i] - base;
            h = fnv_mix(h, (const uint8_t *)&rel, 4);
        }
        idx_hash[k] = h;
        counts[k] = (uint32_t)(index_offsets[k + 1] - index_offsets[k]);
    }
}
