// ondemand.h -- C++ host-side mirror of the reference's on-demand cursor, the front end of its schema-based parser:
//   org.simdjson.OnDemandJsonIterator (/root/reference/src/main/java/org/simdjson/OnDemandJsonIterator.java:7-675),
//   driven by SchemaBasedJsonIterator.java:29-132 (one get* / startIterating* / skipChild call per field of the schema).
// Same method names, same depth bookkeeping, same exception messages.  It walks the GPU-made structural indexes
// (BitIndexes) and parses only the values it is asked for.  (Rounds 2-4 also had a GPU skip table -- up[] / match[] per
// structural, skipChild's scan as a lookup: it never paid at any size measured and was removed in round 5)
// Built: booleans, byte / short / int / long, float, double, String (and their Root / NonNull forms), null handling,
// arrays, objects, field names, skipChild, assertNoMoreJsonValues, char (a Java UTF-16 unit) -- every method of the class.
// Not built: the reflection-driven schema mapping itself (SchemaBasedJsonIterator, ClassResolver), which is Java-specific.
#pragma once
#include <locale.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../sj_number.h"
#include "../sj_bigdec.h"
#include "simdjson_parser.h"

namespace org_simdjson {

const char* errorMessage(int code);  // simdjson_parser.cpp (the messages shared with the full parser)

enum {  // numbering of include/sjmi.h SJMI_E_OD_*
    E_OD_NOT_ENOUGH_CLOSE = 40,    // "Not enough close braces."                                          :80
    E_OD_EXPECTED_CHAR = 41,       // "Expected 'x' but got: 'y'."                                        :662
    E_OD_EXPECTED_CHAR_END = 42,   // "Expected 'x' but reached end of buffer."                           :660
    E_OD_BOOLEAN = 43,             // "Unrecognized boolean value. Expected: 'true' or 'false'."          :88,:152
    E_OD_BOOLEAN_OR_NULL = 44,     // "Unrecognized boolean value. Expected: 'true', 'false' or 'null'."  :104,:167
    E_OD_STRING_OR_NULL = 45,      // "Invalid value starting at N. Expected either string or 'null'."    :455,:470
    E_OD_FLOAT_PART_MISSING = 46,  // "Invalid floating-point number. Fraction or exponent part is missing."  NumberParser.java:303
    E_OD_BYTE_RANGE = 47,          // "Number value is out of byte range ([-128, 127])."                  NumberParser.java:97
    E_OD_SHORT_RANGE = 48,         // "Number value is out of short range ([-32768, 32767])."             :136
    E_OD_INT_RANGE = 49,           // "Number value is out of int range ([-2147483648, 2147483647])."     :175
    E_OD_STRING_EXPECTED = 50,     // "Invalid value starting at N. Expected string."                      :480,:505
    E_OD_CHAR_CODE_POINT = 51,     // "Invalid code point. Should be within the range U+0000–U+D777 or U+E000–U+FFFF."  StringParser.java:78
    E_OD_CHAR_NOT_16BIT = 52,      // "String cannot be deserialized to a char. Expected a single 16-bit code unit character."  :104
    E_OD_CHAR_NOT_SINGLE = 53      // "String cannot be deserialized to a char. Expected a single-character string."  :107
};

class OnDemandJsonIterator {
public:
    enum IteratorResult { EMPTY = 0, NULL_VALUE = 1, NOT_EMPTY = 2 };  // :672-674

    explicit OnDemandJsonIterator(BitIndexes* indexer) : indexer_(indexer) {}

    // init(byte[] buffer, int len) :34-41.  buffer must be readable (and stay unchanged) up to len + 64.
    void init(const uint8_t* buffer, size_t len) {
        if (indexer_->isEnd()) throw error(9 /* E_NO_STRUCTURAL */);
        buffer_ = buffer;
        len_ = len;
        depth_ = 1;
    }
    int getDepth() const { return depth_; }                                       // :654-656
    uint8_t peekByte() const { return buffer_[indexer_->peek()]; }                // (for schema-less drivers: the next structural's byte)
    size_t readIdx() const { return indexer_->readIdx(); }
    void setDepthForTest(int d) { depth_ = d; }  // (tests/host_sim: skipChild from arbitrary positions)

    void skipChild() { skipChild(depth_ - 1); }                                   // :43-45
    void skipChild(int parentDepth) {                                             // :47-81
        if (depth_ <= parentDepth) return;
        uint32_t idx = indexer_->getAndAdvance();
        const uint8_t character = buffer_[idx];
        switch (character) {
        case '[': case '{': case ':': case ',':
            break;
        case '"':
            if (buffer_[indexer_->peek()] == ':') {
                indexer_->advance();  // skip ':'
                break;
            }
            // fall through
        default:
            depth_--;
            if (depth_ <= parentDepth) return;
        }
        while (indexer_->hasNext()) {
            idx = indexer_->getAndAdvance();
            const uint8_t c = buffer_[idx];
            const int delta = (c == '[' || c == '{') ? 1 : ((c == ']' || c == '}') ? -1 : 0);
            depth_ += delta;
            if (delta < 0 && depth_ <= parentDepth) return;
        }
        throw error(E_OD_NOT_ENOUGH_CLOSE);
    }

    // getRootNonNullBoolean :83-93, getRootBoolean :95-109, getNonNullBoolean :147-156, getBoolean :158-171
    // (*isNull = the reference returned null; only with nullable)
    bool getBoolean(bool root, bool nullable, bool* isNull) {
        const uint32_t idx = indexer_->getAndAdvance();
        bool result = false;
        *isNull = false;
        switch (buffer_[idx]) {
        case 't': visitTrueAtom(idx, root); result = true; break;
        case 'f': visitFalseAtom(idx, root); result = false; break;
        case 'n':
            if (nullable) {
                visitNullAtom(idx, root);
                *isNull = true;
                break;
            }
            // fall through
        default:
            throw error(nullable ? E_OD_BOOLEAN_OR_NULL : E_OD_BOOLEAN);
        }
        if (root) assertNoMoreJsonValues();
        depth_--;
        return result;
    }

    // getRootNonNullLong :321-328, getRootLong :330-342, getNonNullLong :344-348, getLong :350-358; bits = 8 / 16 / 32: the
    // Byte :204-241, Short :243-280 and Int :282-319 forms (the same cursor moves, narrower range check)
    int64_t getLong(bool root, bool nullable, bool* isNull, int bits = 64) {
        depth_--;
        const uint32_t idx = indexer_->getAndAdvance();
        *isNull = false;
        if (nullable && buffer_[idx] == 'n') {
            visitNullAtom(idx, true);  // (the reference uses the ROOT null check in getLong as well, :354)
            if (root) assertNoMoreJsonValues();
            *isNull = true;
            return 0;
        }
        const int64_t v = parseLong(idx, bits);
        if (root) assertNoMoreJsonValues();
        return v;
    }

    // getRootNonNullDouble :383-390, getRootDouble :392-404, getNonNullDouble :414-418, getDouble :420-428
    double getDouble(bool root, bool nullable, bool* isNull) {
        depth_--;
        const uint32_t idx = indexer_->getAndAdvance();
        *isNull = false;
        if (nullable && buffer_[idx] == 'n') {
            visitNullAtom(idx, true);  // (:424: the root form here too)
            if (root) assertNoMoreJsonValues();
            *isNull = true;
            return 0.0;
        }
        const double v = parseDouble(idx);
        if (root) assertNoMoreJsonValues();
        return v;
    }

    // getRootNonNullFloat :360-367, getRootFloat :369-381, getNonNullFloat :430-434, getFloat :436-444
    float getFloat(bool root, bool nullable, bool* isNull) {
        depth_--;
        const uint32_t idx = indexer_->getAndAdvance();
        *isNull = false;
        if (nullable && buffer_[idx] == 'n') {
            visitNullAtom(idx, true);  // (:440: the root form here too)
            if (root) assertNoMoreJsonValues();
            *isNull = true;
            return 0.0f;
        }
        const float v = parseFloat(idx);
        if (root) assertNoMoreJsonValues();
        return v;
    }

    // getNonNullChar :474-481, getChar :483-494, getRootNonNullChar :496-505, getRootChar :507-520 -> the UTF-16 unit
    uint16_t getChar(bool root, bool nullable, bool* isNull) {
        depth_--;
        const uint32_t idx = indexer_->getAndAdvance();
        *isNull = false;
        uint16_t ch = 0;
        if (buffer_[idx] == '"') {
            ch = parseChar(idx);
        } else if (nullable && buffer_[idx] == 'n') {
            visitNullAtom(idx, root);
            *isNull = true;
        } else {
            throw error(nullable ? E_OD_STRING_OR_NULL : E_OD_STRING_EXPECTED, idx);
        }
        if (root) assertNoMoreJsonValues();
        return ch;
    }

    // getRootString :446-459, getString :461-472: the unescaped bytes (valid until the next string call), or *isNull
    const std::vector<uint8_t>& getString(bool root, bool* isNull) {
        depth_--;
        const uint32_t idx = indexer_->getAndAdvance();
        *isNull = false;
        switch (buffer_[idx]) {
        case '"': parseString(idx); break;
        case 'n':
            visitNullAtom(idx, root);
            *isNull = true;
            string_.clear();
            break;
        default:
            throw error(E_OD_STRING_OR_NULL, idx);
        }
        if (root) assertNoMoreJsonValues();
        return string_;
    }

    // startIteratingArray :522-541, startIteratingRootArray :543-566
    IteratorResult startIteratingArray(bool root) { return startIterating(root, '[', ']', 11 /* E_UNCLOSED_ARRAY */, true); }
    // startIteratingObject :581-599, startIteratingRootObject :601-623
    IteratorResult startIteratingObject(bool root) { return startIterating(root, '{', '}', 10 /* E_UNCLOSED_OBJECT */, false); }

    bool nextArrayElement() {                                                     // :568-579
        const uint32_t idx = indexer_->getAndAdvance();
        if (buffer_[idx] == ']') {
            depth_--;
            return false;
        } else if (buffer_[idx] == ',') {
            depth_++;
            return true;
        }
        throw error(16 /* E_NO_COMMA_ARRAY: "Missing comma between array values" */);
    }
    bool nextObjectField() {                                                      // :625-636
        const uint32_t idx = indexer_->getAndAdvance();
        const uint8_t c = buffer_[idx];
        if (c == '}') {
            depth_--;
            return false;
        } else if (c == ',') {
            return true;
        }
        throw unexpectedChar(idx, ',');
    }
    void moveToFieldValue() {                                                     // :638-644
        const uint32_t idx = indexer_->getAndAdvance();
        if (buffer_[idx] != ':') throw unexpectedChar(idx, ':');
        depth_++;
    }
    const std::vector<uint8_t>& getFieldName() {                                  // :646-652
        const uint32_t idx = indexer_->getAndAdvance();
        if (buffer_[idx] != '"') throw unexpectedChar(idx, '"');
        parseString(idx);
        return string_;
    }
    void assertNoMoreJsonValues() {                                               // :666-670
        if (indexer_->hasNext()) throw error(17 /* E_TRAILING_CONTENT */);
    }

private:
    static bool structuralOrWs(uint8_t b) { return sjmi::sjn_is_structural_or_ws(b); }  // CharacterUtils.java:6-50
    bool isTrue(uint32_t i) const { return memcmp(buffer_ + i, "true", 4) == 0; }
    bool isFalse(uint32_t i) const { return memcmp(buffer_ + i, "false", 5) == 0; }
    bool isNull(uint32_t i) const { return memcmp(buffer_ + i, "null", 4) == 0; }

    static JsonParsingException error(int code, uint64_t pos = 0) {
        std::string m;
        switch (code) {
        case E_OD_NOT_ENOUGH_CLOSE: m = "Not enough close braces."; break;
        case E_OD_BOOLEAN: m = "Unrecognized boolean value. Expected: 'true' or 'false'."; break;
        case E_OD_BOOLEAN_OR_NULL: m = "Unrecognized boolean value. Expected: 'true', 'false' or 'null'."; break;
        case E_OD_STRING_OR_NULL: m = "Invalid value starting at " + std::to_string(pos) + ". Expected either string or 'null'."; break;
        case E_OD_FLOAT_PART_MISSING: m = "Invalid floating-point number. Fraction or exponent part is missing."; break;
        case E_OD_STRING_EXPECTED: m = "Invalid value starting at " + std::to_string(pos) + ". Expected string."; break;
        case E_OD_CHAR_CODE_POINT: m = "Invalid code point. Should be within the range U+0000\xe2\x80\x93U+D777 or U+E000\xe2\x80\x93U+FFFF."; break;
        case E_OD_CHAR_NOT_16BIT: m = "String cannot be deserialized to a char. Expected a single 16-bit code unit character."; break;
        case E_OD_CHAR_NOT_SINGLE: m = "String cannot be deserialized to a char. Expected a single-character string."; break;
        case E_OD_BYTE_RANGE: m = "Number value is out of byte range ([-128, 127])."; break;
        case E_OD_SHORT_RANGE: m = "Number value is out of short range ([-32768, 32767])."; break;
        case E_OD_INT_RANGE: m = "Number value is out of int range ([-2147483648, 2147483647])."; break;
        default: {
            m = errorMessage(code);
            const size_t at = m.find("%d");
            if (at != std::string::npos) m.replace(at, 2, std::to_string(pos));
        }
        }
        return JsonParsingException(code, m, pos);
    }
    JsonParsingException unexpectedChar(uint32_t idx, char expected) const {      // :658-664
        if (indexer_->isPastEnd())
            return JsonParsingException(E_OD_EXPECTED_CHAR_END, std::string("Expected '") + expected + "' but reached end of buffer.", idx);
        return JsonParsingException(E_OD_EXPECTED_CHAR, std::string("Expected '") + expected + "' but got: '" + javaChar(buffer_[idx]) + "'.", idx);
    }
    // Java's (char) cast of a byte sign-extends: a byte >= 0x80 becomes the UTF-16 unit 0xFF80..0xFFFF (here: its UTF-8)
    static std::string javaChar(uint8_t b) {
        std::string got;
        if (b < 0x80) {
            got.push_back((char)b);
        } else {
            const uint32_t u = 0xFF00u | b;
            got.push_back((char)(0xE0 | (u >> 12)));
            got.push_back((char)(0x80 | ((u >> 6) & 0x3F)));
            got.push_back((char)(0x80 | (u & 0x3F)));
        }
        return got;
    }

    // visitTrueAtom :173-179 / visitRootTrueAtom :111-117 (and False :188-194/:119-125, Null :134-138/:127-132)
    void visitTrueAtom(uint32_t idx, bool root) const {
        const bool valid = root ? (idx + 4 <= len_ && isTrue(idx) && (idx + 4 == len_ || structuralOrWs(buffer_[idx + 4])))
                                : (isTrue(idx) && structuralOrWs(buffer_[idx + 4]));
        if (!valid) throw error(19 /* E_INVALID_TRUE */, idx);
    }
    void visitFalseAtom(uint32_t idx, bool root) const {
        const bool valid = root ? (idx + 5 <= len_ && isFalse(idx) && (idx + 5 == len_ || structuralOrWs(buffer_[idx + 5])))
                                : (isFalse(idx) && structuralOrWs(buffer_[idx + 5]));
        if (!valid) throw error(20 /* E_INVALID_FALSE */, idx);
    }
    void visitNullAtom(uint32_t idx, bool root) const {
        const bool valid = root ? (idx + 4 <= len_ && isNull(idx) && (idx + 4 == len_ || structuralOrWs(buffer_[idx + 4])))
                                : isNull(idx);  // (:134-138: no test of the byte behind a non-root null)
        if (!valid) throw error(21 /* E_INVALID_NULL */, idx);
    }

    IteratorResult startIterating(bool root, char open, char close, int unclosedCode, bool isArray) {
        uint32_t idx = indexer_->peek();
        if (buffer_[idx] == 'n') {
            visitNullAtom(idx, root);
            indexer_->advance();
            depth_--;
            return NULL_VALUE;
        }
        if (buffer_[idx] != (uint8_t)open) throw unexpectedChar(idx, open);
        if (root && buffer_[indexer_->getLast()] != (uint8_t)close) throw error(unclosedCode);
        idx = indexer_->advanceAndGet();
        if (buffer_[idx] == (uint8_t)close) {
            indexer_->advance();
            depth_--;
            if (root) assertNoMoreJsonValues();
            return EMPTY;
        }
        if (isArray) depth_++;  // (:539,:564 -- an object's depth grows in moveToFieldValue instead)
        return NOT_EMPTY;
    }

    // bytes at / behind len read as spaces: the reference pads a root number with spaces (padRootNumber :406-412) and tests
    // `currentIdx < len` before the byte behind any other number (NumberParser.java:219,:299)
    uint32_t byteAt(uint32_t q) const { return q < len_ ? buffer_[q] : 0x20u; }

    int64_t parseLong(uint32_t offset, int bits) const {                          // NumberParser.parseByte / Short / Int / Long :76-224
        const bool negative = byteAt(offset) == '-';
        uint32_t cur = negative ? offset + 1 : offset;
        const uint32_t digitsStart = cur;
        unsigned long long digits = 0;
        while (byteAt(cur) - '0' <= 9u) digits = 10 * digits + (byteAt(cur++) - '0');
        const uint32_t digitCount = cur - digitsStart;
        if (digitCount == 0) throw error(22 /* E_NUM_MINUS */);
        if (byteAt(digitsStart) == '0' && digitCount > 1) throw error(23 /* E_NUM_LEADING_ZERO */);
        if (!structuralOrWs((uint8_t)byteAt(cur))) throw error(26 /* E_NUM_FOLLOWED */);
        if (bits == 64) {
            if (sjmi::sj_out_of_long_range(negative, digits, digitCount)) throw error(27 /* E_NUM_LONG_RANGE */);
        } else {  // isOutOfByteRange :102-113, isOutOfShortRange :141-152, isOutOfIntRange :180-191
            const uint32_t maxDigits = bits == 8 ? 3u : (bits == 16 ? 5u : 10u);
            const unsigned long long maxAbs = 1ull << (bits - 1);
            const bool out = digitCount > maxDigits || (digitCount == maxDigits && (negative ? digits > maxAbs : digits > maxAbs - 1));
            if (out) throw error(bits == 8 ? E_OD_BYTE_RANGE : (bits == 16 ? E_OD_SHORT_RANGE : E_OD_INT_RANGE));
        }
        return (int64_t)(negative ? (~digits + 1) : digits);
    }

    double parseDouble(uint32_t offset) const {                                   // NumberParser.parseDouble :268-310
        const sjmi::SjNumber n = sjmi::sj_scan_number([&](uint32_t q) -> uint32_t { return byteAt(q); }, offset);
        // the grammar errors in the reference's order; "fraction or exponent missing" is tested in front of the byte behind
        if (n.code && n.code != 26) throw error(n.code);
        if (!n.floating) throw error(E_OD_FLOAT_PART_MISSING);
        if (n.code) throw error(n.code);
        unsigned long long bits;
        double v;
        if (sjmi::sj_number_double_bits(n, &bits)) {
            memcpy(&v, &bits, 8);
        } else {
            // (within 10^-19 of a rounding boundary: DoubleParser's slow path :205-330 = the exact comparison of the literal with
            //  the midpoint of its two candidates, sj_bigdec.h -- the same routine as both walkers)
            std::vector<uint32_t> wa(sjmi::SJ_BIG_WORDS), wb(sjmi::SJ_BIG_WORDS);
            const uint32_t start = offset + (n.negative ? 1u : 0u);
            const unsigned long long mag =
                sjmi::sj_decide_double([&](uint32_t q) -> uint32_t { return byteAt(q); }, start, bits & ~(1ull << 63), wa.data(), wb.data());
            bits = mag | (n.negative ? 1ull << 63 : 0ull);
            memcpy(&v, &bits, 8);
        }
        return v;
    }

    float parseFloat(uint32_t offset) const {                                     // NumberParser.parseFloat :226-266
        const sjmi::SjNumber n = sjmi::sj_scan_number([&](uint32_t q) -> uint32_t { return byteAt(q); }, offset);
        if (n.code && n.code != 26) throw error(n.code);
        if (!n.floating) throw error(E_OD_FLOAT_PART_MISSING);
        if (n.code) throw error(n.code);
        uint32_t bits;
        float v;
        if (sjmi::sj_number_float_bits(n, &bits)) {
            memcpy(&v, &bits, 4);
        } else {  // (FloatParser's slow path :203-330 = a correctly rounded conversion)
            static const locale_t c_locale = newlocale(LC_ALL_MASK, "C", (locale_t)0);
            std::string lit;
            for (uint32_t q = offset; !structuralOrWs((uint8_t)byteAt(q)); ++q) lit.push_back((char)byteAt(q));
            v = strtof_l(lit.c_str(), nullptr, c_locale);
        }
        return v;
    }

    // StringParser.parseString(buffer, idx, stringBuffer) (StringParser.java:25-27 -> doParseString :29-68): the bytes of
    // the string whose opening quote is at idx, unescaped, into string_
    void parseString(uint32_t idx) {
        string_.clear();
        const uint8_t* src = buffer_ + idx + 1;
        for (;;) {
            const uint8_t c = *src;
            if (c == '"') return;
            if (c != '\\') {
                string_.push_back(c);
                ++src;
                continue;
            }
            const uint8_t e = src[1];
            if (e == 'u') {                                                       // :45-57
                int cp = hex4(src + 2);
                src += 6;
                if (cp >= 0xD800 && cp < 0xDC00) {                                // parseLowSurrogate :112-124
                    if (src[0] != '\\' || src[1] != 'u') throw error(7 /* E_LOW_SURROGATE_NO_U */);
                    const int low = hex4(src + 2) - 0xDC00;
                    if ((low >> 10) != 0) throw error(8 /* E_LOW_SURROGATE_RANGE */);
                    cp = (((cp - 0xD800) << 10) | low) + 0x10000;
                    src += 6;
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                    throw error(6 /* E_LOW_SURROGATE_RESERVED */);
                }
                storeCodePoint(cp);
            } else {                                                              // :58-61, CharacterUtils.escape :74-83
                uint8_t r = 0;
                switch (e) {
                case '"': r = '"'; break;
                case '\\': r = '\\'; break;
                case '/': r = '/'; break;
                case 'b': r = 0x08; break;
                case 'f': r = 0x0C; break;
                case 'n': r = 0x0A; break;
                case 'r': r = 0x0D; break;
                case 't': r = 0x09; break;
                default: {
                    throw JsonParsingException(4, std::string(errorMessage(4)) + javaChar(e), (uint64_t)(src - buffer_));
                }
                }
                string_.push_back(r);
                src += 2;
            }
        }
    }
    uint16_t parseChar(uint32_t startIdx) const {                                 // StringParser.parseChar :70-110
        const uint8_t* p = buffer_ + startIdx + 1;
        uint32_t character;
        if (p[0] == '\\') {
            const uint8_t e = p[1];
            if (e == 'u') {
                const int cp = hex4(p + 2);
                if (cp >= 0xD800 && cp <= 0xDFFF) throw error(E_OD_CHAR_CODE_POINT);
                if (cp < 0) throw error(5 /* E_INVALID_UNICODE_ESCAPE */);
                character = (uint32_t)cp;
                p += 6;
            } else {
                uint8_t r = 0;
                switch (e) {
                case '"': r = '"'; break;
                case '\\': r = '\\'; break;
                case '/': r = '/'; break;
                case 'b': r = 0x08; break;
                case 'f': r = 0x0C; break;
                case 'n': r = 0x0A; break;
                case 'r': r = 0x0D; break;
                case 't': r = 0x09; break;
                default: throw JsonParsingException(4, std::string(errorMessage(4)) + javaChar(e), (uint64_t)(p - buffer_));
                }
                character = r;
                p += 2;
            }
        } else if (p[0] < 0x80) {
            character = p[0];
            p += 1;
        } else if ((p[0] & 0xE0) == 0xC0) {
            character = (uint32_t)(p[0] & 0x1F) << 6 | (p[1] & 0x3F);
            p += 2;
        } else if ((p[0] & 0xF0) == 0xE0) {
            character = (uint32_t)(p[0] & 0x0F) << 12 | (uint32_t)(p[1] & 0x3F) << 6 | (p[2] & 0x3F);
            p += 3;
        } else {
            throw error(E_OD_CHAR_NOT_16BIT);
        }
        if (p[0] != '"') throw error(E_OD_CHAR_NOT_SINGLE);
        return (uint16_t)character;
    }
    static int hex4(const uint8_t* p) {                                           // CharacterUtils.hexToInt :241-247
        int v = 0;
        for (int i = 0; i < 4; ++i) {
            const uint8_t c = p[i];
            int d;
            if (c >= '0' && c <= '9') d = c - '0';
            else if ((c | 0x20) >= 'a' && (c | 0x20) <= 'f') d = (c | 0x20) - 'a' + 10;
            else return -1;
            v = (v << 4) | d;
        }
        return v;
    }
    void storeCodePoint(int cp) {                                                 // storeCodePointInStringBuffer :126-161
        if (cp < 0) throw error(5 /* E_INVALID_UNICODE_ESCAPE */);
        if (cp <= 0x7F) {
            string_.push_back((uint8_t)cp);
        } else if (cp <= 0x7FF) {
            string_.push_back((uint8_t)((cp >> 6) + 192));
            string_.push_back((uint8_t)((cp & 63) + 128));
        } else if (cp <= 0xFFFF) {
            string_.push_back((uint8_t)((cp >> 12) + 224));
            string_.push_back((uint8_t)(((cp >> 6) & 63) + 128));
            string_.push_back((uint8_t)((cp & 63) + 128));
        } else {
            string_.push_back((uint8_t)((cp >> 18) + 240));
            string_.push_back((uint8_t)(((cp >> 12) & 63) + 128));
            string_.push_back((uint8_t)(((cp >> 6) & 63) + 128));
            string_.push_back((uint8_t)((cp & 63) + 128));
        }
    }

    BitIndexes* indexer_;
    const uint8_t* buffer_ = nullptr;
    size_t len_ = 0;
    int depth_ = 0;
    std::vector<uint8_t> string_;
};

}  // namespace org_simdjson
