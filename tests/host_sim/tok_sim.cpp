// The token walker's decisions (coop_walk.hip k_tok_walk) without a GPU.  TEST ONLY.  The two tables come verbatim from
// csrc/sj_tokens.h; this file walks a document's structurals SEQUENTIALLY with exactly the kernel's rules around the tables --
// separators folded into the token behind them, "a separator behind a separator" and "behind the last token" fail the document,
// the container of a token from a stack of opening brackets, the root value's end, the depth limit -- and says whether the
// token walker would keep the document (1) or hand it to the exact walker (0), and how many tape words it would have written.
// tests/test_host_tokens.py compares that with the oracle's stage 2 over every short token sequence.
#include <stdint.h>
#include <vector>
#include "../../simdjson-java_amd/csrc/sj_tokens.h"

using namespace sjmi;

extern "C" {
// structurals: positions of the document's structurals (stage 1), n of them; max_depth as SimdJsonParser's; levels = the stack's
// depth in the kernel (CW_LEVELS = 64).  *words = tape words (incl. the two root words) when kept.
int sim_tok_walk(const uint8_t* doc, const uint32_t* structurals, uint32_t n, int max_depth, int levels, uint32_t* words) {
    static uint8_t grammar[TOK_GRAMMAR_ENTRIES];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < TOK_GRAMMAR_ENTRIES; ++i) grammar[i] = (uint8_t)tok_grammar(i);
        init = true;
    }
    if (n == 0) return 0;
    const int depth_limit = (max_depth < levels ? max_depth : levels) - 1;
    std::vector<uint32_t> stack;  // is-array per open container
    uint32_t prev_token = TK_NONE, pre = 0, t = 1;
    bool prev_sep = false, root_closed = false;
    int h = 0;  // depth in front of the token
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t b = doc[structurals[i]];
        if (b == ',' || b == ':') {
            if (prev_sep) return 0;             // a separator behind a separator
            if (i + 1 == n) return 0;           // a separator behind the last token
            prev_sep = true;
            pre = b == ',' ? TOK_COMMA : TOK_COLON;
            continue;
        }
        if (root_closed) return 0;              // something follows the root value
        const uint32_t token = tok_of_first_byte(b) | pre;
        const uint32_t tk = token & 7u;
        const uint32_t arr = !stack.empty() && stack.back() ? 1u : 0u;
        const uint32_t gi = (token & 0x1Fu) | ((prev_token & 0x17u) << 5) | (arr << 10);
        if (!grammar[gi]) return 0;
        if (tk <= TK_OPEN_O) {
            // an opening bracket directly followed by its closing bracket (nothing in front of that) is one value: not a level
            bool empty = false;
            if (i + 1 < n) {
                const uint32_t nb = doc[structurals[i + 1]];
                empty = (tk == TK_OPEN_A && nb == ']') || (tk == TK_OPEN_O && nb == '}');
            }
            if (!empty && h >= depth_limit) return 0;
            stack.push_back(tk == TK_OPEN_A ? 1u : 0u);
            ++h;
        } else if (tk <= TK_CLOSE_O) {
            if (stack.empty()) return 0;
            stack.pop_back();
            --h;
            if (h == 0) root_closed = true;
        }
        t += tk == TK_NUMBER ? 2u : 1u;
        prev_token = token;
        pre = 0;
        prev_sep = false;
    }
    if (!root_closed) return 0;
    *words = t + 1;
    return 1;
}
}

// the two tables themselves, for the lane-level model of the kernel (tools/tok_walk_model.py)
extern "C" uint32_t sim_tok_of_first_byte(uint32_t b) { return tok_of_first_byte(b); }
extern "C" uint32_t sim_tok_grammar(uint32_t i) { return tok_grammar(i); }
