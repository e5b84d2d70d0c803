// Host build of the GPU walker's per-document automaton (simdjson-java_amd/csrc/walk_doc.h, shared verbatim with
// walk.hip).  TEST ONLY: lets the CPU test-suite fuzz the device-side grammar, number conversion and hand-back rules
// against the oracle without a GPU; the GPU tests then cover the kernel around it (slots, packing, batches).
// Built by tests/test_host_walkdev.py with g++.
#include "../../simdjson-java_amd/csrc/walk_doc.h"

extern "C" int sim_walk_device(const uint8_t* buf, uint32_t doc_start, uint32_t doc_end, const uint32_t* ix, uint32_t from,
                               uint32_t to, uint32_t ix_entries, const uint8_t* sb, uint64_t sc, uint64_t sbase, int max_depth,
                               uint64_t* tape, uint32_t* tape_len) {
    sjmi::Lane w;  // (as k_doc_walk sets it up)
    w.buf = buf;
    w.ix = ix;
    w.ix_entries = ix_entries;
    w.iw_base = 0xFFFFFFFFu;
    w.bw_base = 0xFFFFFFF0u;
    w.from = from;
    w.to = to;
    w.rd = from;
    w.doc_start = doc_start;
    w.doc_end = doc_end;
    w.tape = reinterpret_cast<unsigned long long*>(tape);
    w.tl = 0;
    w.sb = sb;
    w.sc = sc;
    w.sbase = sbase;
    w.code = 0;
    if (sjmi::walk_document(w, max_depth)) {
        *tape_len = w.tl;
        return 0;
    }
    *tape_len = 0;
    return w.code;
}
