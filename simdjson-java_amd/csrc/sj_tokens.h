// The token walker's two tables (coop_walk.hip k_tok_stream), as plain functions shared verbatim with the CPU test
// (tests/host_sim/tok_sim.cpp, tests/test_host_tokens.py): what a structural's first byte makes of it, and the token grammar of
// JsonIterator.java:68-193 keyed by {token, separator in front of it, previous token, is my container an array}.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define SJT_HD __host__ __device__ inline
#else
#define SJT_HD inline
#endif

namespace sjmi {

// Token kinds of the batch walker (its own numbering: what a token step needs is a compare or a table index away)
enum : uint32_t { TK_OPEN_A = 0, TK_OPEN_O = 1, TK_CLOSE_A = 2, TK_CLOSE_O = 3, TK_STRING = 4, TK_NONE = 5, TK_ATOM = 6, TK_NUMBER = 7 };
// A token as the ring holds it: TK_* | ',' in front << 3 | ':' in front << 4 | depth field of the scan (1 + up - down) << 5 | first byte << 8 | its tape words << 21
constexpr uint32_t TOK_COMMA = 8u, TOK_COLON = 16u, TOK_SCAN_FIELDS = 0x00600060u;
constexpr uint32_t TOK_GRAMMAR_ENTRIES = 2048u;

// the token of a structural's first byte (separators: TK_NONE, they never reach the ring; anything that is not a bracket or a
// quote starts an atom or a number -- what it really is, the literal parser decides)
SJT_HD uint32_t tok_of_first_byte(uint32_t b) {
    uint32_t tk;
    switch (b) {
    case '[': tk = TK_OPEN_A; break;
    case '{': tk = TK_OPEN_O; break;
    case ']': tk = TK_CLOSE_A; break;
    case '}': tk = TK_CLOSE_O; break;
    case '"': tk = TK_STRING; break;
    case ',': case ':': return TK_NONE;
    default: tk = (b == '-' || b - '0' <= 9u) ? (uint32_t)TK_NUMBER : (uint32_t)TK_ATOM; break;
    }
    const uint32_t field = tk <= TK_OPEN_O ? 2u : (tk <= TK_CLOSE_O ? 0u : 1u), words = tk == TK_NUMBER ? 2u : 1u;
    return tk | (field << 5) | (b << 8) | (words << 21);
}

// The token grammar, JsonIterator.java:68-193 re-keyed for tokens: i = token (TK_* | TOK_COMMA | TOK_COLON) | the same five bits
// of the previous token << 5 | my container is an array << 10.  Every earlier token of the document was good (one bad token
// fails the document), so what came before is known from the previous token alone:
//     nothing (TK_NONE)            the root: an opening bracket, nothing in front (any other root: the exact walker)
//     '['                          no separator; a value, or ']' (the empty array, TapeBuilder.java:205-208)
//     '{'                          no separator; a key, or '}'
//     a key                        ':' and a value      (a string is a key: in an object, and no ':' in front of it)
//     a value                      ',' and a value (array) / a key (object), or no separator and the container's own closing bracket
SJT_HD uint32_t tok_grammar(uint32_t i) {
    const uint32_t tk = i & 7u, prev = (i >> 5) & 7u;
    const bool comma = (i & TOK_COMMA) != 0, colon = (i & TOK_COLON) != 0, prev_colon = ((i >> 5) & TOK_COLON) != 0, arr = ((i >> 10) & 1u) != 0;
    const bool close = tk == TK_CLOSE_A || tk == TK_CLOSE_O;
    if (tk == TK_NONE) return 1u;  // (a lane without a token)
    if (comma && colon) return 0u;
    if (prev == TK_NONE) return (tk <= TK_OPEN_O && !comma && !colon) ? 1u : 0u;
    if (prev == TK_OPEN_A) return (!comma && !colon && tk != TK_CLOSE_O) ? 1u : 0u;                     // :68-77
    if (prev == TK_OPEN_O) return (!comma && !colon && (tk == TK_STRING || tk == TK_CLOSE_O)) ? 1u : 0u;
    if (prev == TK_STRING && !arr && !prev_colon) return (colon && !close) ? 1u : 0u;                    // :84-86
    if (comma) return (arr ? !close : tk == TK_STRING) ? 1u : 0u;                                        // :121-123
    if (colon) return 0u;
    return tk == (arr ? (uint32_t)TK_CLOSE_A : (uint32_t)TK_CLOSE_O) ? 1u : 0u;                          // :131,:189
}

}  // namespace sjmi
