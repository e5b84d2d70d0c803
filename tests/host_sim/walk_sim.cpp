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
int sjmi_set_input_staging(sjmi_ctx*, void*, uint64_t) { return SJMI_ERR_NO_DEVICE; }
int sjmi_host_unregister(sjmi_ctx*, void*) { return SJMI_ERR_NO_DEVICE; }
int sjmi_stage1_unescape(sjmi_ctx*, const uint8_t*, uint64_t, uint32_t*, uint64_t, uint64_t*, uint32_t*, uint8_t*, uint64_t,
                         uint64_t*, uint64_t*, uint32_t*) { return SJMI_ERR_NO_DEVICE; }
int sjmi_stage1_batch_isolated(sjmi_ctx*, const uint8_t*, uint64_t, const uint64_t*, uint64_t, uint32_t*, uint64_t, uint64_t*,
                               uint32_t*, uint64_t*, uint32_t*) { return SJMI_ERR_NO_DEVICE; }
int sjmi_unescape_batch(sjmi_ctx*, uint8_t*, uint64_t, uint64_t*, uint64_t*, uint64_t*, uint32_t*) { return SJMI_ERR_NO_DEVICE; }
int sjmi_stage1(sjmi_ctx*, const uint8_t*, uint64_t, uint32_t*, uint64_t, uint64_t*, uint32_t*) { return SJMI_ERR_NO_DEVICE; }
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

// ---- the on-demand cursor (csrc/host/ondemand.h) over indexes the test supplies ----
struct SimOnDemand {
    org_simdjson::BitIndexes idx;
    org_simdjson::OnDemandJsonIterator it;
    std::string msg;
    SimOnDemand(uint32_t* indexes, size_t cap) : idx(indexes, cap), it(&idx) {}
};
void* sim_od_create(const uint8_t* padded, uint64_t len, uint32_t* indexes, uint64_t count, int* code) {
    SimOnDemand* s = new SimOnDemand(indexes, (size_t)count + 1);
    s->idx.reset();
    s->idx.setWriteIdx((size_t)count);
    *code = 0;
    try {
        s->it.init(padded, (size_t)len);
    } catch (const org_simdjson::JsonParsingException& e) {
        s->msg = e.what();
        *code = e.code();
    }
    return s;
}
void sim_od_destroy(void* h) { delete static_cast<SimOnDemand*>(h); }
const char* sim_od_message(void* h) { return static_cast<SimOnDemand*>(h)->msg.c_str(); }
// op: 0 skipChild(a) (a < 0: skipChild()), 1 getBoolean(root=a, nullable=b), 2 getLong, 3 getDouble, 4 getString(root=a),
// 5 getFieldName, 6 startIteratingArray(root=a), 7 nextArrayElement, 8 startIteratingObject(root=a), 9 nextObjectField,
// 10 moveToFieldValue, 11 assertNoMoreJsonValues, 12 getFloat(root=a, nullable=b), 13 getChar.  -> exception code or 0; results in out (int64) / dout / bytes
int sim_od_call(void* h, int op, int a, int b, int64_t* out, double* dout, const uint8_t** bytes, uint64_t* nbytes) {
    SimOnDemand* s = static_cast<SimOnDemand*>(h);
    s->msg.clear();
    *out = 0;
    *dout = 0;
    *bytes = nullptr;
    *nbytes = 0;
    try {
        bool isNull = false;
        switch (op) {
        case 0: if (a < 0) s->it.skipChild(); else s->it.skipChild(a); break;
        case 1: *out = s->it.getBoolean(a != 0, b != 0, &isNull) ? 1 : 0; if (isNull) *out = -1; break;
        case 2: *out = s->it.getLong((a & 1) != 0, b != 0, &isNull, (a >> 8) ? (a >> 8) : 64); if (isNull) *nbytes = 1; break;  // a = root | bits << 8
        case 3: *dout = s->it.getDouble(a != 0, b != 0, &isNull); if (isNull) *nbytes = 1; break;
        case 13: *out = s->it.getChar(a != 0, b != 0, &isNull); if (isNull) *out = -1; break;
        case 12: *dout = (double)s->it.getFloat(a != 0, b != 0, &isNull); if (isNull) *nbytes = 1; break;  // (float -> double: exact)
        case 4: { const std::vector<uint8_t>& v = s->it.getString(a != 0, &isNull); *bytes = v.data(); *nbytes = v.size(); if (isNull) *out = -1; break; }
        case 5: { const std::vector<uint8_t>& v = s->it.getFieldName(); *bytes = v.data(); *nbytes = v.size(); break; }
        case 6: *out = (int)s->it.startIteratingArray(a != 0); break;
        case 7: *out = s->it.nextArrayElement() ? 1 : 0; break;
        case 8: *out = (int)s->it.startIteratingObject(a != 0); break;
        case 9: *out = s->it.nextObjectField() ? 1 : 0; break;
        case 10: s->it.moveToFieldValue(); break;
        case 11: s->it.assertNoMoreJsonValues(); break;
        default: return -2;
        }
        return 0;
    } catch (const org_simdjson::JsonParsingException& e) {
        s->msg = e.what();
        return e.code();
    }
}
int sim_od_depth(void* h) { return static_cast<SimOnDemand*>(h)->it.getDepth(); }
// (test only) put the cursor at read position r with depth d
void sim_od_set(void* h, uint64_t r, int d) {
    SimOnDemand* s = static_cast<SimOnDemand*>(h);
    s->idx.setReadIdx((size_t)r);
    s->it.setDepthForTest(d);
}
int sim_od_peek(void* h) {
    SimOnDemand* s = static_cast<SimOnDemand*>(h);
    return s->idx.hasNext() ? (int)s->it.peekByte() : 256;
}
uint64_t sim_od_read_idx(void* h) { return static_cast<SimOnDemand*>(h)->it.readIdx(); }
}
