// Host stage 2 (DocWalker: JsonIterator.walkDocument + TapeBuilder, csrc/host/simdjson_parser.cpp) without a GPU.
// TEST ONLY: the engine ABI is stubbed out -- every engine call fails, so SimdJsonParser itself cannot be created
// here -- and a DocWalker is driven directly with indexes and string records the test took from the oracle.  This lets
// the CPU test-suite check the host mirror's grammar, number and tape logic; the GPU tests cover it end to end.
// Built by tests/test_host_walk.py with g++.
#include "../../simdjson-java_amd/csrc/host/simdjson_parser.cpp"

extern "C" {
int sjmi_create(sjmi_ctx** out, int, uint64_t) { if (out) *out = nullptr; return SJMI_ERR_NO_DEVICE; }
void sjmi_destroy(sjmi_ctx*) {}
const char* sjmi_last_error(const sjmi_ctx*) { return "host simulation: no engine"; }
int sjmi_host_register(sjmi_ctx*, void*, uint64_t) { return SJMI_ERR_NO_DEVICE; }
int sjmi_host_unregister(sjmi_ctx*, void*) { return SJMI_ERR_NO_DEVICE; }
int sjmi_stage1_unescape(sjmi_ctx*, const uint8_t*, uint64_t, uint32_t*, uint64_t, uint64_t*, uint32_t*, uint8_t*, uint64_t,
                         uint64_t*, uint64_t*, uint32_t*) { return SJMI_ERR_NO_DEVICE; }
int sjmi_stage1_batch_isolated(sjmi_ctx*, const uint8_t*, uint64_t, const uint64_t*, uint64_t, uint32_t*, uint64_t, uint64_t*,
                               uint32_t*, uint64_t*, uint32_t*) { return SJMI_ERR_NO_DEVICE; }
int sjmi_unescape_batch(sjmi_ctx*, uint8_t*, uint64_t, uint64_t*, uint64_t*, uint64_t*, uint32_t*) { return SJMI_ERR_NO_DEVICE; }
int sjmi_parse_document(sjmi_ctx*, const uint8_t*, uint64_t, int, uint64_t*, uint64_t, uint64_t*, uint8_t*, uint64_t, uint64_t*, int32_t*,
                        uint32_t*) { return SJMI_ERR_NO_DEVICE; }

// walk one document: padded = the document + 64 bytes, indexes[0, count) its structurals, sb its string records.
// Returns 0 and the tape, or the SJMI_E_* code of the JsonParsingException.
int sim_walk(const uint8_t* padded, uint64_t len, uint32_t* indexes, uint64_t count, const uint8_t* sb, int max_depth,
             uint64_t* tape_out, uint64_t tape_cap, uint64_t* tape_len) {
    org_simdjson::DocWalker w(padded, indexes, (size_t)count + 1, (size_t)tape_cap, max_depth);
    w.setStringBuffer(sb);
    w.bitIndexes().reset();
    w.bitIndexes().setWriteIdx((size_t)count);
    w.resetForDocument(0, 0);
    try {
        w.walkDocument((size_t)len);
    } catch (const org_simdjson::JsonParsingException& e) {
        return e.code();
    }
    *tape_len = w.tape().getCurrentIdx();
    for (size_t i = 0; i < w.tape().getCurrentIdx(); ++i) tape_out[i] = w.tape().data()[i];
    return 0;
}

// the engine is absent: creating a parser must fail, not fall back to anything
int sim_parser_create_fails() {
    sjmi_parser* p = nullptr;
    return sjmi_parser_create(&p, 1 << 20, 1024, 0) != 0 && p == nullptr;
}
}
