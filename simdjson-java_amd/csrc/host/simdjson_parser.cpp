// simdjson_parser.cpp -- host side of SimdJsonParser.parse: GPU stage 1 + GPU string unescape through the
// C ABI, then the reference's sequential stage 2 (JsonIterator + TapeBuilder + Tape + number grammar)
// re-implemented in C++.  Citations: /root/reference/src/main/java/org/simdjson/<file>:<lines>.
#include "simdjson_parser.h"
#include "ondemand.h"
#include "../sj_number.h"
#include "../sj_bigdec.h"

#include <stdio.h>
#include <stdlib.h>
#include <locale.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <exception>
#include <memory>
#include <thread>
#include <string.h>

namespace org_simdjson {

enum {  // same numbering as include/sjmi.h SJMI_E_*
    E_UTF8 = 1, E_UNCLOSED_STRING = 2, E_UNESCAPED_CHARS = 3, E_ESCAPE_UNEXPECTED = 4, E_INVALID_UNICODE_ESCAPE = 5,
    E_LOW_SURROGATE_RESERVED = 6, E_LOW_SURROGATE_NO_U = 7, E_LOW_SURROGATE_RANGE = 8, E_NO_STRUCTURAL = 9,
    E_UNCLOSED_OBJECT = 10, E_UNCLOSED_ARRAY = 11, E_OBJECT_NO_KEY = 12, E_MISSING_COLON = 13, E_KEY_MISSING = 14,
    E_NO_COMMA_OBJECT = 15, E_NO_COMMA_ARRAY = 16, E_TRAILING_CONTENT = 17, E_UNRECOGNIZED_PRIMITIVE = 18,
    E_INVALID_TRUE = 19, E_INVALID_FALSE = 20, E_INVALID_NULL = 21, E_NUM_MINUS = 22, E_NUM_LEADING_ZERO = 23,
    E_NUM_DECIMAL_POINT = 24, E_NUM_EXPONENT = 25, E_NUM_FOLLOWED = 26, E_NUM_LONG_RANGE = 27, E_DEPTH = 28,
    E_CAPACITY = 29
};

const char* errorMessage(int code) {
    switch (code) {
    case E_UTF8: return "The input is not valid UTF-8";                                   // Utf8Validator.java:166
    case E_UNCLOSED_STRING: return "Unclosed string. A string is opened, but never closed.";  // StructuralIndexer.java:298
    case E_UNESCAPED_CHARS:
        return "Unescaped characters. Within strings, there are characters that should be escaped.";  // :301
    case E_ESCAPE_UNEXPECTED: return "Escaped unexpected character: ";                     // CharacterUtils.java:76,80
    case E_INVALID_UNICODE_ESCAPE: return "Invalid unicode escape sequence.";              // StringParser.java:128
    case E_LOW_SURROGATE_RESERVED:
        return "Invalid code point. The range U+DC00\xe2\x80\x93U+DFFF is reserved for low surrogate.";  // :54
    case E_LOW_SURROGATE_NO_U: return "Low surrogate should start with '\\u'";              // :114
    case E_LOW_SURROGATE_RANGE:
        return "Invalid code point. Low surrogate should be in the range U+DC00\xe2\x80\x93U+DFFF.";  // :121
    case E_NO_STRUCTURAL: return "No structural element found.";                           // JsonIterator.java:28
    case E_UNCLOSED_OBJECT: return "Unclosed object. Missing '}' for starting '{'.";       // :40
    case E_UNCLOSED_ARRAY: return "Unclosed array. Missing ']' for starting '['.";         // :52
    case E_OBJECT_NO_KEY: return "Object does not start with a key";                       // :76
    case E_MISSING_COLON: return "Missing colon after key in object";                      // :85
    case E_KEY_MISSING: return "Key string missing at beginning of field in object";       // :122
    case E_NO_COMMA_OBJECT: return "No comma between object fields";                       // :131
    case E_NO_COMMA_ARRAY: return "Missing comma between array values";                    // :189
    case E_TRAILING_CONTENT:
        return "More than one JSON value at the root of the document, or extra characters at the end of the JSON!";  // :197
    case E_UNRECOGNIZED_PRIMITIVE:
        return "Unrecognized primitive. Expected: string, number, 'true', 'false' or 'null'.";  // TapeBuilder.java:66,77
    case E_INVALID_TRUE: return "Invalid value starting at %d. Expected 'true'.";          // :103,111
    case E_INVALID_FALSE: return "Invalid value starting at %d. Expected 'false'.";        // :126,134
    case E_INVALID_NULL: return "Invalid value starting at %d. Expected 'null'.";          // :150,158
    case E_NUM_MINUS: return "Invalid number. Minus has to be followed by a digit.";        // NumberParser.java:35
    case E_NUM_LEADING_ZERO: return "Invalid number. Leading zeroes are not allowed.";     // :38
    case E_NUM_DECIMAL_POINT: return "Invalid number. Decimal point has to be followed by a digit.";  // :52
    case E_NUM_EXPONENT: return "Invalid number. Exponent indicator has to be followed by a digit.";  // ExponentParser.java:28
    case E_NUM_FOLLOWED: return "Number has to be followed by a structural character or whitespace.";  // NumberParser.java:64
    case E_NUM_LONG_RANGE:
        return "Number value is out of long range ([-9223372036854775808, 9223372036854775807]).";  // :71
    case E_DEPTH: return "ArrayIndexOutOfBoundsException (max depth exceeded)";             // JsonIterator.java:69-70
    case E_CAPACITY: return "capacity exceeded";
    default: return "unknown";
    }
}

static JsonParsingException fail(int code, uint64_t pos = 0) {
    std::string m = errorMessage(code);
    const size_t at = m.find("%d");
    if (at != std::string::npos) m.replace(at, 2, std::to_string(pos));
    return JsonParsingException(code, m, pos);
}

void Tape::appendDouble(double v) {
    append(0, DOUBLE);
    uint64_t bits;
    memcpy(&bits, &v, 8);
    tape_[idx_++] = bits;
}
double Tape::getDouble(size_t i) const {
    double d;
    memcpy(&d, &tape_[i + 1], 8);
    return d;
}
size_t Tape::computeNextIndex(size_t i) const {  // Tape.java:86-98
    switch (getType(i)) {
    case START_ARRAY:
    case START_OBJECT: return getMatchingBraceIndex(i);
    case INT64:
    case DOUBLE: return i + 2;
    default: return i + 1;
    }
}

std::string JsonValue::asString() const {  // JsonValue.java:79-89, IntegerUtils.toInt :5-10
    const size_t off = (size_t)tape_->getValue(idx_);
    const uint32_t len = ((uint32_t)sb_[off] << 24) | ((uint32_t)sb_[off + 1] << 16) | ((uint32_t)sb_[off + 2] << 8) | sb_[off + 3];
    return std::string(reinterpret_cast<const char*>(sb_ + off + 4), len);
}

bool JsonValue::get(const std::string& name, JsonValue* out) const {  // JsonValue.java:91-107 (linear key scan)
    size_t i = idx_ + 1;
    const size_t end = tape_->getMatchingBraceIndex(idx_) - 1;
    while (i < end) {
        const size_t off = (size_t)tape_->getValue(i);
        const uint32_t len = ((uint32_t)sb_[off] << 24) | ((uint32_t)sb_[off + 1] << 16) | ((uint32_t)sb_[off + 2] << 8) | sb_[off + 3];
        const size_t val = tape_->computeNextIndex(i);
        i = tape_->computeNextIndex(val);
        if (len == name.size() && memcmp(sb_ + off + 4, name.data(), len) == 0) {
            *out = JsonValue(tape_, val, sb_);
            return true;
        }
    }
    return false;
}

// The host threads of parseBatch: n - 1 helpers that sleep between batches; run(k, fn) executes fn(0..k-1), fn(0) on
// the caller.  (Creating threads per batch cost ~1 ms per phase at 32 threads.)
class WorkerPool {
public:
    explicit WorkerPool(size_t n) {
        for (size_t t = 1; t < n; ++t) helpers_.emplace_back([this, t] { loop(t); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            ++generation_;
        }
        wake_.notify_all();
        for (std::thread& th : helpers_) th.join();
    }
    size_t size() const { return helpers_.size() + 1; }
    void run(size_t tasks, const std::function<void(size_t)>& fn) {
        if (tasks > size()) tasks = size();
        if (tasks > 1) {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            tasks_ = tasks;
            pending_ = tasks - 1;
            ++generation_;
        }
        if (tasks > 1) wake_.notify_all();
        if (tasks) fn(0);
        if (tasks > 1) {
            std::unique_lock<std::mutex> g(m_);
            done_.wait(g, [this] { return pending_ == 0; });
            fn_ = nullptr;
        }
    }

private:
    void loop(size_t t) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
                if (stop_) return;
                if (t < tasks_) fn = fn_;
            }
            if (!fn) continue;
            (*fn)(t);  // (fn never throws: parseBatch's tasks catch into their lane)
            bool last;
            {
                std::lock_guard<std::mutex> g(m_);
                last = --pending_ == 0;
            }
            if (last) done_.notify_one();
        }
    }
    std::vector<std::thread> helpers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t tasks_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

SimdJsonParser::SimdJsonParser(int capacity, int maxDepth, int device)
    : capacity_(capacity), maxDepth_(maxDepth), device_(device), stringBuffer_(3 * (size_t)capacity + 256),
      paddedBuffer_((size_t)capacity + PADDING),
      // (+ 2 per sub-batch of parseBatch: sub-batch j's count + sentinel may end at byteStart + byteLen + 2 j + 1)
      indexes_((size_t)capacity + 2 + 2 * 64),
      walker_(paddedBuffer_.data(), indexes_.data(), indexes_.size(), (size_t)capacity + 8, maxDepth) {
    const int rc = sjmi_create(&ctx_, device, (uint64_t)capacity);
    if (rc != SJMI_OK) throw std::runtime_error("SimdJsonParser: no usable MI355X device (sjmi_create rc=" + std::to_string(rc) + "); there is no CPU fallback");
    // parser-owned buffers that cross PCIe on every parse are page-locked (a hint: failures are ignored)
    pinned_[0] = sjmi_host_register(ctx_, paddedBuffer_.data(), paddedBuffer_.size()) == SJMI_OK ? paddedBuffer_.data() : nullptr;
    // ... and the padded copy doubles as the engine's input staging: padIfNeeded happens inside the upload, in chunks that
    // overlap their PCIe transfers for large documents (include/sjmi.h: sjmi_set_input_staging)
    stagedInput_ = pinned_[0] && sjmi_set_input_staging(ctx_, paddedBuffer_.data(), (uint64_t)capacity) == SJMI_OK;
    pinned_[1] = sjmi_host_register(ctx_, indexes_.data(), indexes_.size() * sizeof(uint32_t)) == SJMI_OK ? (void*)indexes_.data() : nullptr;
    pinned_[2] = sjmi_host_register(ctx_, stringBuffer_.data(), stringBuffer_.size()) == SJMI_OK ? stringBuffer_.data() : nullptr;
    // (the tape: the download of the GPU walker's result, sjmi_parse_document)
    pinned_[3] = sjmi_host_register(ctx_, walker_.tape().raw(), walker_.tape().capacity() * sizeof(uint64_t)) == SJMI_OK ? (void*)walker_.tape().raw() : nullptr;
    unsigned hw = std::thread::hardware_concurrency();
    batchThreads_ = (int)(hw ? (hw > 64 ? 64 : hw) : 1);  // (tools/batch_e2e.py: 32 -> 64 threads 7.8 -> 6.5 ms per 100k documents, worse beyond)
    if (const char* e = getenv("SJMI_PARSE_THREADS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 1024) batchThreads_ = v;
    }
    if (const char* e = getenv("SJMI_PARSE_PIPELINE")) {  // sub-batches per batch (default: by size, 1..4)
        const int v = atoi(e);
        if (v >= 1 && v <= 64) batchPipeline_ = v;
    }
}

SimdJsonParser::~SimdJsonParser() {
    pool_.reset();
    if (ctx2_) sjmi_destroy(ctx2_);
    for (void* p : pinned_)
        if (p) (void)sjmi_host_unregister(ctx_, p);
    sjmi_destroy(ctx_);
}

void SimdJsonParser::growStringBuffer(size_t need) {
    if (stringBuffer_.size() >= need) return;
    if (pinned_[2]) (void)sjmi_host_unregister(ctx_, pinned_[2]);
    pinned_[2] = nullptr;
    stringBuffer_.resize(need);
}

// SimdJsonParser.stage1 (SimdJsonParser.java:55-58) on the GPU + the string records stage 2 will need
void SimdJsonParser::stage1(const uint8_t* buffer, size_t len) {
    uint64_t count = 0, total = 0, fei = 0;
    uint32_t status = 0, fec = 0;
    growStringBuffer(len + 4 * (len / 2 + 2) + 64);  // every string takes >= 2 source bytes
    // stage 1 and the string records stage 2 will need, queued together (two synchronisations instead of four)
    int rc = sjmi_stage1_unescape(ctx_, buffer, len, indexes_.data(), indexes_.size(), &count, &status,
                                  stringBuffer_.data(), stringBuffer_.size(), &total, &fei, &fec);
    if (rc != SJMI_OK) throw std::runtime_error(std::string("sjmi_stage1_unescape: ") + sjmi_last_error(ctx_));
    walker_.bitIndexes().setWriteIdx((size_t)count);
    if (status & SJMI_ST_UTF8) throw fail(E_UTF8);                 // Utf8Validator.java:165-167 (checked first)
    if (status & SJMI_ST_UNCLOSED) throw fail(E_UNCLOSED_STRING);  // StructuralIndexer.java:297-299
    if (status & SJMI_ST_UNESCAPED) throw fail(E_UNESCAPED_CHARS); // :300-302
    stringBufferLen_ = (size_t)total;
}

JsonValue SimdJsonParser::parse(const uint8_t* buffer, size_t len) {
    if (len > (size_t)capacity_) throw fail(E_CAPACITY);
    // padIfNeeded (SimdJsonParser.java:42-48): the C++ caller's buffer has no known slack, so always copy -- inside the engine's
    // upload when the padded buffer is its staging buffer (the copy then overlaps the PCIe transfer), else here
    const uint8_t* src = paddedBuffer_.data();
    if (stagedInput_) src = buffer;
    else memcpy(paddedBuffer_.data(), buffer, len);
    memset(paddedBuffer_.data() + len, 0, PADDING);
    // reset (:50-53)
    walker_.bitIndexes().reset();
    walker_.resetForDocument(0, 0);
    if (gpuWalk_ > 0 || (gpuWalk_ < 0 && len >= GPU_WALK_AUTO_BYTES)) {
        // all three stages on the GPU: the tape arrives ready; a document that fails (or that the device hands back) takes
        // the host path below, which raises the reference's exception with its exact message and position
        uint64_t words = 0, sbLen = 0;
        int32_t err = 0;
        uint32_t status = 0;
        growStringBuffer(len + 4 * (len / 2 + 2) + 64);
        const int rc = sjmi_parse_document(ctx_, src, len, maxDepth_, walker_.tape().raw(), walker_.tape().capacity(),
                                           &words, stringBuffer_.data(), stringBuffer_.size(), &sbLen, &err, &status);
        if (rc != SJMI_OK) throw std::runtime_error(std::string("sjmi_parse_document: ") + sjmi_last_error(ctx_));
        if (err == 0) {
            walker_.tape().setCurrentIdx((size_t)words);
            stringBufferLen_ = (size_t)sbLen;
            return JsonValue(&walker_.tape(), 1, stringBuffer_.data());
        }
    }
    stage1(src, len);  // (after a declined all-device attempt the document is already in the padded buffer: src == it or re-staged)
    walker_.setStringBuffer(stringBuffer_.data());
    walker_.walkDocument(len);
    return JsonValue(&walker_.tape(), 1, stringBuffer_.data());  // TapeBuilder.createJsonValue :215-217
}

// The head of the schema-based parse (SimdJsonParser.java:31-33: padIfNeeded, reset, stage1; SchemaBasedJsonIterator.java
// :29-41: iterator.init): stage 1 alone on the GPU -- the on-demand cursor parses the strings it is asked for itself -- and,
// on request, the skip table of the document's brackets.
void SimdJsonParser::onDemandInit(const uint8_t* buffer, size_t len) {
    onDemandReady_ = false;
    if (len > (size_t)capacity_) throw fail(E_CAPACITY);
    memcpy(paddedBuffer_.data(), buffer, len);
    memset(paddedBuffer_.data() + len, 0, PADDING);
    walker_.bitIndexes().reset();
    uint64_t count = 0;
    uint32_t status = 0;
    int rc = sjmi_stage1(ctx_, paddedBuffer_.data(), len, indexes_.data(), indexes_.size(), &count, &status);
    if (rc != SJMI_OK) throw std::runtime_error(std::string("sjmi_stage1: ") + sjmi_last_error(ctx_));
    walker_.bitIndexes().setWriteIdx((size_t)count);
    if (status & SJMI_ST_UTF8) throw fail(E_UTF8);
    if (status & SJMI_ST_UNCLOSED) throw fail(E_UNCLOSED_STRING);
    if (status & SJMI_ST_UNESCAPED) throw fail(E_UNESCAPED_CHARS);
    if (!onDemand_) onDemand_.reset(new OnDemandJsonIterator(&walker_.bitIndexes()));
    onDemand_->init(paddedBuffer_.data(), len);
    onDemandReady_ = true;
}

// Batched parse.  The batch is cut into up to eight sub-batches of whole documents (about 8 MB or more each); two feeder threads, each with its
// own engine context (= its own stream), take them alternately through the GPU: isolated stage 1 (per-document
// verdicts and index ranges), the string records, and the string-buffer offset at which each document's records
// begin.  So the upload of one sub-batch overlaps the kernels and downloads of the other (PCIe is full duplex), and
// the host stage 2 of a finished sub-batch -- its documents spread over the worker pool by structural count, every
// thread with a DocWalker building tapes back to back in a slab, then the slabs copied to their place in batchTape() --
// overlaps the GPU part of the next.
// Sub-batch j's indexes / string records are placed at fixed bases of the shared arrays (sized for the worst case:
// one structural per byte, three string-buffer bytes per document byte), so with more than one sub-batch the string
// buffer has unused gaps between them; tape STRING payloads are offsets into that one buffer.
void SimdJsonParser::parseBatch(const uint8_t* buffer, size_t totalLen, const uint64_t* docOffsets, size_t nDocs) {
    if (totalLen > (size_t)capacity_) throw fail(E_CAPACITY);
    // the offsets come across the public ABI: a non-monotonic pair or an offset past the end must be an argument
    // error here, not a wrapped length on the device
    if (nDocs && docOffsets[0] != 0) throw std::invalid_argument("parseBatch: doc_offsets[0] != 0");
    for (size_t k = 0; k < nDocs; ++k)
        if (docOffsets[k + 1] < docOffsets[k]) throw std::invalid_argument("parseBatch: doc_offsets not monotonic");
    if (nDocs && docOffsets[nDocs] > totalLen) throw std::invalid_argument("parseBatch: doc_offsets[n] > total_len");
    // SJMI_PARSE_TIMING=1: phase times of every batch on stderr (tools/batch_e2e.py)
    static const bool timing = getenv("SJMI_PARSE_TIMING") != nullptr;
    using Clock = std::chrono::steady_clock;
    const Clock::time_point t0 = Clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); };
    if (!pool_) {
        pool_.reset(new WorkerPool((size_t)batchThreads_));
        lanes_.resize((size_t)batchThreads_);
    }
    const size_t T = (size_t)batchThreads_;

    // sub-batches: whole documents, about equal bytes
    size_t J = std::min<size_t>(8, totalLen / (8u << 20) + 1);
    if (batchPipeline_ > 0) J = (size_t)batchPipeline_;
    if (J > nDocs) J = nDocs ? nDocs : 1;
    struct SubBatch {
        size_t docLo = 0, docHi = 0, byteStart = 0, byteLen = 0, idxBase = 0, sbBase = 0, offBase = 0;
        uint64_t count = 0, total = 0;
        std::vector<uint64_t> rel;  // document offsets relative to byteStart
        std::vector<size_t> cut;    // document ranges of the walk
        size_t walkers = 0;
        bool ready = false;
        bool copied = false;        // its bytes are in paddedBuffer_ (padIfNeeded for the batch, sub-batch by sub-batch)
        std::exception_ptr error;
        double tGpu = 0, tWalk = 0;
    };
    std::vector<SubBatch> subs(J);
    for (size_t j = 0; j < J; ++j) {
        SubBatch& sb = subs[j];
        sb.docLo = j == 0 ? 0 : subs[j - 1].docHi;
        if (j + 1 == J) sb.docHi = nDocs;
        else {
            const uint64_t want = (uint64_t)totalLen * (j + 1) / J;
            sb.docHi = (size_t)(std::lower_bound(docOffsets, docOffsets + nDocs, want) - docOffsets);
            if (sb.docHi < sb.docLo) sb.docHi = sb.docLo;
        }
        sb.byteStart = nDocs ? (size_t)docOffsets[sb.docLo] : 0;
        sb.byteLen = nDocs ? (size_t)docOffsets[sb.docHi] - sb.byteStart : 0;
        sb.idxBase = sb.byteStart + 2 * j;       // count + sentinel <= byteLen + 1
        sb.sbBase = 3 * sb.byteStart + 8 * j;    // 4 + L <= 2 (L + 2) for every string: total <= 2 byteLen
        sb.offBase = sb.docLo + j;               // n + 1 offsets per sub-batch
    }
    growStringBuffer(3 * totalLen + 8 * J + 64);
    indexOffsets_.assign(nDocs + J + 1, 0);
    docStringOffsets_.assign(nDocs + J + 1, 0);
    docStatus_.assign(nDocs ? nDocs : 1, 0);
    batchTapeOffsets_.assign(nDocs + 1, 0);
    batchErrors_.assign(nDocs, 0);
    batchTapeLen_ = 0;
    if (!batchTape_) {
        batchTape_.reset(new uint64_t[1024]);
        batchTapeRoom_ = 1024;
    }
    if (J > 1 && !ctx2_) {
        const int rc = sjmi_create(&ctx2_, device_, (uint64_t)capacity_);
        if (rc != SJMI_OK) throw std::runtime_error("SimdJsonParser: second engine context: sjmi_create rc=" + std::to_string(rc));
    }
    if (pieces_.size() < J * T) pieces_.resize(J * T);

    std::mutex m;
    std::condition_variable readyCv, copiedCv;
    // the GPU part of sub-batches j = first, first + 2, ... on one context
    auto feed = [&](sjmi_ctx* ctx, size_t first) {
        for (size_t j = first; j < J; j += 2) {
            SubBatch& sb = subs[j];
            {
                std::unique_lock<std::mutex> g(m);
                copiedCv.wait(g, [&] { return sb.copied; });
            }
            try {
                const size_t n = sb.docHi - sb.docLo;
                sb.rel.resize(n + 1);
                for (size_t i = 0; i <= n; ++i) sb.rel[i] = docOffsets[sb.docLo + i] - sb.byteStart;
                uint32_t status = 0;
                // isolated mode: a document that fails stage 1 gets its own verdict and contributes no indexes, so
                // the strings and trees of all other documents are exactly what they would be alone
                int rc = sjmi_stage1_batch_isolated(ctx, paddedBuffer_.data() + sb.byteStart, sb.byteLen, sb.rel.data(), n,
                                                    indexes_.data() + sb.idxBase, sb.byteLen + 2,
                                                    indexOffsets_.data() + sb.offBase, docStatus_.data() + sb.docLo,
                                                    &sb.count, &status);
                if (rc != SJMI_OK) throw std::runtime_error(std::string("sjmi_stage1_batch_isolated: ") + sjmi_last_error(ctx));
                uint64_t fei = 0;
                uint32_t fec = 0;
                rc = sjmi_unescape_batch(ctx, stringBuffer_.data() + sb.sbBase, 3 * sb.byteLen + 8,
                                         docStringOffsets_.data() + sb.offBase, &sb.total, &fei, &fec);
                if (rc != SJMI_OK) throw std::runtime_error(std::string("sjmi_unescape_batch: ") + sjmi_last_error(ctx));
            } catch (...) {
                sb.error = std::current_exception();
            }
            sb.tGpu = since();
            {
                std::lock_guard<std::mutex> g(m);
                sb.ready = true;
            }
            readyCv.notify_all();
        }
    };
    std::thread feeders[2];
    // The feeders wait for their sub-batches' bytes (copiedCv).  Whatever is thrown between here and the join below -- the
    // pool, an allocation -- must not leave them waiting with joinable std::thread objects going out of scope (that is
    // std::terminate): on the way out every sub-batch is marked copied, so the feeders run dry, and they are joined.
    struct JoinFeeders {
        std::thread* th;
        std::vector<SubBatch>& subs;
        std::mutex& m;
        std::condition_variable& cv;
        ~JoinFeeders() {
            {
                std::lock_guard<std::mutex> g(m);
                for (SubBatch& sb : subs) sb.copied = true;
            }
            cv.notify_all();
            for (int i = 0; i < 2; ++i)
                if (th[i].joinable()) th[i].join();
        }
    } joinFeeders{feeders, subs, m, copiedCv};
    feeders[0] = std::thread(feed, ctx_, (size_t)0);
    if (J > 1) feeders[1] = std::thread(feed, ctx2_, (size_t)1);
    // padIfNeeded for the batch (SimdJsonParser.java:42-48), on the pool (one thread copies ~25 GB/s), sub-batch by sub-batch:
    // the GPU part of sub-batch 0 starts as soon as ITS bytes are in place instead of behind the whole batch's copy (647 MB:
    // 9 ms of a 31 ms call)
    memset(paddedBuffer_.data() + totalLen, 0, PADDING);
    for (size_t j = 0; j < J; ++j) {
        SubBatch& sb = subs[j];
        const size_t span = (j + 1 == J ? totalLen : sb.byteStart + sb.byteLen) - sb.byteStart;  // (the last one: up to the batch's end)
        const size_t parts = std::min<size_t>(T, std::min<size_t>(8, span / (1u << 20) + 1));
        const std::function<void(size_t)> copyPart = [&](size_t t) {
            const size_t lo = sb.byteStart + span * t / parts, hi = sb.byteStart + span * (t + 1) / parts;
            memcpy(paddedBuffer_.data() + lo, buffer + lo, hi - lo);
        };
        if (span) pool_->run(parts, copyPart);
        {
            std::lock_guard<std::mutex> g(m);
            sb.copied = true;
        }
        copiedCv.notify_all();
    }
    const double tCopy = since();

    const uint8_t* strings = stringBuffer_.data();
    std::exception_ptr failure;
    for (size_t j = 0; j < J; ++j) {
        SubBatch& sb = subs[j];
        {
            std::unique_lock<std::mutex> g(m);
            readyCv.wait(g, [&] { return sb.ready; });
        }
        if (sb.error && !failure) failure = sb.error;
        if (failure) continue;  // (still wait for the feeders' remaining sub-batches)
        // contiguous document ranges of about equal structural count, one per thread
        const size_t n = sb.docHi - sb.docLo;
        const uint64_t* io = indexOffsets_.data() + sb.offBase;
        const uint64_t* so = docStringOffsets_.data() + sb.offBase;
        size_t W = T;
        const size_t minPerThread = 4096;  // structurals: below this a thread costs more than it walks
        if (W > (size_t)sb.count / minPerThread + 1) W = (size_t)sb.count / minPerThread + 1;
        if (W > n) W = n ? n : 1;
        sb.walkers = W;
        sb.cut.assign(W + 1, n);
        sb.cut[0] = 0;
        for (size_t t = 1; t < W; ++t) {
            const uint64_t want = sb.count * t / W;
            sb.cut[t] = (size_t)(std::lower_bound(io, io + n, want) - io);
            if (sb.cut[t] < sb.cut[t - 1]) sb.cut[t] = sb.cut[t - 1];
        }
        const std::function<void(size_t)> walkRange = [&](size_t t) {
            BatchLane& lane = lanes_[t];
            TapeSlab& slab = pieces_[j * T + t];
            slab.used = 0;
            lane.error = nullptr;
            try {
                const size_t lo = sb.cut[t], hi = sb.cut[t + 1];
                // a structural makes at most two tape words (a number); the root adds two
                const size_t room = 2 * (size_t)(io[hi] - io[lo]) + 8 * (hi - lo) + 8;
                if (slab.room < room) {
                    slab.words.reset();
                    slab.words.reset(new uint64_t[room + room / 4]);
                    slab.room = room + room / 4;
                }
                if (!lane.walker) lane.walker.reset(new DocWalker(nullptr, nullptr, 0, 0, maxDepth_));
                DocWalker& w = *lane.walker;
                w.rebind(paddedBuffer_.data() + sb.byteStart, indexes_.data() + sb.idxBase, sb.byteLen + 2);
                w.setStringBuffer(strings);
                for (size_t k = lo; k < hi; ++k) {
                    const size_t doc = sb.docLo + k;
                    w.resetForDocument((size_t)sb.rel[k], sb.sbBase + (size_t)so[k]);
                    w.tape().rebase(slab.words.get() + slab.used, slab.room - slab.used);
                    w.bitIndexes().window((size_t)io[k], (size_t)io[k + 1], (uint32_t)sb.rel[k]);
                    try {
                        // SimdJsonParser.stage1 order: Utf8Validator.validate (:165-167), then StructuralIndexer.index (:297-302)
                        if (docStatus_[doc] & SJMI_ST_UTF8) throw fail(E_UTF8);
                        if (docStatus_[doc] & SJMI_ST_UNCLOSED) throw fail(E_UNCLOSED_STRING);
                        if (docStatus_[doc] & SJMI_ST_UNESCAPED) throw fail(E_UNESCAPED_CHARS);
                        w.walkDocument((size_t)sb.rel[k + 1]);
                        slab.used += w.tape().getCurrentIdx();
                    } catch (const JsonParsingException& e) {
                        batchErrors_[doc] = e.code();  // its partial tape is dropped
                    }
                    batchTapeOffsets_[doc + 1] = slab.used;  // slab-relative until the slabs are placed
                }
            } catch (...) {
                lane.error = std::current_exception();
            }
        };
        pool_->run(W, walkRange);
        for (size_t t = 0; t < W && !failure; ++t)
            if (lanes_[t].error) failure = lanes_[t].error;
        if (failure) continue;
        // slabs -> batchTape(), in document order (while the GPU works on the next sub-batch)
        std::vector<size_t> base(W + 1, batchTapeLen_);
        for (size_t t = 0; t < W; ++t) base[t + 1] = base[t] + pieces_[j * T + t].used;
        if (base[W] > batchTapeRoom_) {
            const size_t room = base[W] + base[W] / 2 + 1024;
            std::unique_ptr<uint64_t[]> grown(new uint64_t[room]);
            if (batchTapeLen_) memcpy(grown.get(), batchTape_.get(), batchTapeLen_ * sizeof(uint64_t));
            batchTape_ = std::move(grown);
            batchTapeRoom_ = room;
        }
        const std::function<void(size_t)> place = [&](size_t t) {
            const TapeSlab& slab = pieces_[j * T + t];
            if (slab.used) memcpy(batchTape_.get() + base[t], slab.words.get(), slab.used * sizeof(uint64_t));
            for (size_t k = sb.cut[t]; k < sb.cut[t + 1]; ++k) batchTapeOffsets_[sb.docLo + k + 1] += base[t];
        };
        pool_->run(W, place);
        batchTapeLen_ = base[W];
        sb.tWalk = since();
    }
    for (std::thread& th : feeders)
        if (th.joinable()) th.join();
    if (failure) std::rethrow_exception(failure);
    stringBufferLen_ = subs[J - 1].sbBase + (size_t)subs[J - 1].total;

    if (timing) {
        fprintf(stderr, "parseBatch %zu docs %zu B, %zu threads, %zu sub-batches: copy+pad done at %.2f ms;", nDocs, totalLen, T, J, tCopy);
        for (size_t j = 0; j < J; ++j) fprintf(stderr, " [%zu] gpu %.2f walked+placed %.2f;", j, subs[j].tGpu, subs[j].tWalk);
        fprintf(stderr, " done at %.2f ms\n", since());
    }
}

// TapeBuilder.visitString (TapeBuilder.java:174-177): the record [be32 len][bytes] was produced on the GPU at
// exactly the offset the sequential StringParser would have used; only the bookkeeping remains.
void DocWalker::visitString(uint32_t idx, size_t indexPos) {
    (void)idx;
    (void)indexPos;
    tape_.append(stringBufferIdx_, Tape::STRING);
    const uint8_t* r = stringBuffer_.data() + stringBufferIdx_;
    const uint32_t n = ((uint32_t)r[0] << 24) | ((uint32_t)r[1] << 16) | ((uint32_t)r[2] << 8) | r[3];
    // a string the reference's StringParser would have thrown on is marked FF FF FF <SJMI_E_* code> by the kernel
    if (n >= 0xFFFFFF00u) throw fail((int)(n & 0xFFu));
    stringBufferIdx_ += 4 + (size_t)n;
}

static inline bool isStructuralOrWhitespace(uint8_t b) {  // CharacterUtils.java:6-50
    switch (b) {
    case 0x09: case 0x0A: case 0x0D: case 0x20: case ',': case ':': case '[': case ']': case '{': case '}': return true;
    default: return false;
    }
}
static inline bool isTrue(const uint8_t* b) { return b[0] == 't' && b[1] == 'r' && b[2] == 'u' && b[3] == 'e'; }
static inline bool isFalse(const uint8_t* b) { return b[0] == 'f' && b[1] == 'a' && b[2] == 'l' && b[3] == 's' && b[4] == 'e'; }
static inline bool isNull(const uint8_t* b) { return b[0] == 'n' && b[1] == 'u' && b[2] == 'l' && b[3] == 'l'; }

// NumberParser.parseNumber (NumberParser.java:23-74), ExponentParser.parse (ExponentParser.java:14-69),
// isOutOfLongRange (NumberParser.java:313-328).  Doubles: the reference's DoubleParser is a correctly rounded,
// saturating decimal->binary64 conversion (DoubleParser.java:79-330).
void DocWalker::parseNumber(const uint8_t* p) {
    // grammar + Clinger / Eisel-Lemire conversion shared with the device walkers (csrc/sj_number.h): locale-independent,
    // no allocation, the literal is scanned in place (the buffer is padded and the literal ends at a structural or
    // whitespace byte)
    const sjmi::SjNumber n = sjmi::sj_scan_number([&](uint32_t q) -> uint32_t { return p[q]; }, 0);
    if (n.code) throw fail(n.code);
    if (n.floating) {
        unsigned long long bits;
        if (!sjmi::sj_number_double_bits(n, &bits)) {
            // more than 19 significant digits AND within 10^-19 of a rounding boundary: the reference's slow path
            // (DoubleParser.java:205-330) -- here the exact comparison with the midpoint of the two candidates (sj_bigdec.h:
            // big integers on the stack, no libc: a JVM host's locale cannot touch it)
            uint32_t wa[sjmi::SJ_BIG_WORDS], wb[sjmi::SJ_BIG_WORDS];
            const uint32_t start = n.negative ? 1u : 0u;
            const unsigned long long mag =
                sjmi::sj_decide_double([&](uint32_t q) -> uint32_t { return p[q]; }, start, bits & ~(1ull << 63), wa, wb);
            bits = mag | (n.negative ? 1ull << 63 : 0ull);
            double v;
            memcpy(&v, &bits, 8);
            tape_.appendDouble(v);
        } else {
            double v;
            memcpy(&v, &bits, 8);
            tape_.appendDouble(v);
        }
    } else {
        if (sjmi::sj_out_of_long_range(n.negative, n.digits, n.digit_count)) throw fail(E_NUM_LONG_RANGE);
        tape_.appendInt64((int64_t)(n.negative ? (~n.digits + 1) : n.digits));
    }
}

void DocWalker::visitPrimitive(uint32_t idx, size_t indexPos) {  // TapeBuilder.java:70-79
    const uint8_t* b = paddedBuffer_.data() + idx;
    switch (*b) {
    case '"': visitString(idx, indexPos); break;
    case 't':
        if (!(isTrue(b) && isStructuralOrWhitespace(b[4]))) throw fail(E_INVALID_TRUE, idx - docBase_);
        tape_.append(0, Tape::TRUE_VALUE);
        break;
    case 'f':
        if (!(isFalse(b) && isStructuralOrWhitespace(b[5]))) throw fail(E_INVALID_FALSE, idx - docBase_);
        tape_.append(0, Tape::FALSE_VALUE);
        break;
    case 'n':
        if (!(isNull(b) && isStructuralOrWhitespace(b[4]))) throw fail(E_INVALID_NULL, idx - docBase_);
        tape_.append(0, Tape::NULL_VALUE);
        break;
    case '-': case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': case '8': case '9':
        parseNumber(b);
        break;
    default: throw fail(E_UNRECOGNIZED_PRIMITIVE);
    }
}

void DocWalker::visitRootPrimitive(uint32_t idx, size_t indexPos, size_t len) {  // TapeBuilder.java:59-68 (len = end offset)
    const uint8_t* b = paddedBuffer_.data() + idx;
    switch (*b) {
    case '"': visitString(idx, indexPos); break;
    case 't':
        if (!(idx + 4 <= len && isTrue(b) && (idx + 4 == len || isStructuralOrWhitespace(b[4])))) throw fail(E_INVALID_TRUE, idx - docBase_);
        tape_.append(0, Tape::TRUE_VALUE);
        break;
    case 'f':
        if (!(idx + 5 <= len && isFalse(b) && (idx + 5 == len || isStructuralOrWhitespace(b[5])))) throw fail(E_INVALID_FALSE, idx - docBase_);
        tape_.append(0, Tape::FALSE_VALUE);
        break;
    case 'n':
        if (!(idx + 4 <= len && isNull(b) && (idx + 4 == len || isStructuralOrWhitespace(b[4])))) throw fail(E_INVALID_NULL, idx - docBase_);
        tape_.append(0, Tape::NULL_VALUE);
        break;
    case '-': case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': case '8': case '9': {
        // visitRootNumber :183-189: the number is re-parsed from a copy padded with 64 spaces
        std::vector<uint8_t> copy(len - idx + PADDING, 0x20);
        memcpy(copy.data(), b, len - idx);
        parseNumber(copy.data());
        break;
    }
    default: throw fail(E_UNRECOGNIZED_PRIMITIVE);
    }
}

void DocWalker::emptyContainer(char start, char end) {  // TapeBuilder.java:205-208
    tape_.append(tape_.getCurrentIdx() + 2, start);
    tape_.append(tape_.getCurrentIdx(), end);
}

// JsonIterator.walkDocument (JsonIterator.java:26-200), state for state
void DocWalker::walkDocument(size_t len) {
    enum { OBJECT_BEGIN, ARRAY_BEGIN, DOCUMENT_END, OBJECT_FIELD, OBJECT_CONTINUE, SCOPE_END, ARRAY_CONTINUE, ARRAY_VALUE };
    const uint8_t* buffer = paddedBuffer_.data();
    BitIndexes& indexer = bitIndexes_;
    if (indexer.isEnd()) throw fail(E_NO_STRUCTURAL);
    auto startContainer = [&](int depth) {  // TapeBuilder.java:191-195
        openContainers_[(size_t)depth].tapeIndex = tape_.getCurrentIdx();
        openContainers_[(size_t)depth].count = 0;
        tape_.skip();
    };
    auto endContainer = [&](char start, char end, int depth) {  // :197-203
        const size_t st = openContainers_[(size_t)depth].tapeIndex;
        tape_.append(st, end);
        uint32_t count = openContainers_[(size_t)depth].count;
        if (count > 0xFFFFFF) count = 0xFFFFFF;
        tape_.write(st, tape_.getCurrentIdx() | ((uint64_t)count << 32), start);
    };
    startContainer(0);  // visitDocumentStart :41-43
    int depth = 0, state;
    size_t pos = indexer.readIdx();
    uint32_t idx = indexer.getAndAdvance();
    switch (buffer[idx]) {
    case '{':
        if (buffer[indexer.getLast()] != '}') throw fail(E_UNCLOSED_OBJECT);
        if (buffer[indexer.peek()] == '}') { indexer.advance(); emptyContainer('{', '}'); state = DOCUMENT_END; }
        else state = OBJECT_BEGIN;
        break;
    case '[':
        if (buffer[indexer.getLast()] != ']') throw fail(E_UNCLOSED_ARRAY);
        if (buffer[indexer.peek()] == ']') { indexer.advance(); emptyContainer('[', ']'); state = DOCUMENT_END; }
        else state = ARRAY_BEGIN;
        break;
    default:
        visitRootPrimitive(idx, pos, len);
        state = DOCUMENT_END;
    }
    while (state != DOCUMENT_END) {
        if (state == OBJECT_BEGIN) {
            ++depth;
            if (depth >= maxDepth_) throw fail(E_DEPTH);
            isArray_[(size_t)depth] = 0;
            startContainer(depth);
            pos = indexer.readIdx();
            const uint32_t keyIdx = indexer.getAndAdvance();
            if (buffer[keyIdx] != '"') throw fail(E_OBJECT_NO_KEY);
            openContainers_[(size_t)depth].count++;
            visitString(keyIdx, pos);
            state = OBJECT_FIELD;
        }
        if (state == OBJECT_FIELD) {
            if (buffer[indexer.getAndAdvance()] != ':') throw fail(E_MISSING_COLON);
            pos = indexer.readIdx();
            idx = indexer.getAndAdvance();
            switch (buffer[idx]) {
            case '{':
                if (buffer[indexer.peek()] == '}') { indexer.advance(); emptyContainer('{', '}'); state = OBJECT_CONTINUE; }
                else state = OBJECT_BEGIN;
                break;
            case '[':
                if (buffer[indexer.peek()] == ']') { indexer.advance(); emptyContainer('[', ']'); state = OBJECT_CONTINUE; }
                else state = ARRAY_BEGIN;
                break;
            default:
                visitPrimitive(idx, pos);
                state = OBJECT_CONTINUE;
            }
        }
        if (state == OBJECT_CONTINUE) {
            switch (buffer[indexer.getAndAdvance()]) {
            case ',': {
                openContainers_[(size_t)depth].count++;
                pos = indexer.readIdx();
                const uint32_t keyIdx = indexer.getAndAdvance();
                if (buffer[keyIdx] != '"') throw fail(E_KEY_MISSING);
                visitString(keyIdx, pos);
                state = OBJECT_FIELD;
                break;
            }
            case '}':
                endContainer('{', '}', depth);
                state = SCOPE_END;
                break;
            default: throw fail(E_NO_COMMA_OBJECT);
            }
        }
        if (state == SCOPE_END) {
            --depth;
            if (depth == 0) state = DOCUMENT_END;
            else if (isArray_[(size_t)depth]) state = ARRAY_CONTINUE;
            else state = OBJECT_CONTINUE;
        }
        if (state == ARRAY_BEGIN) {
            ++depth;
            if (depth >= maxDepth_) throw fail(E_DEPTH);
            isArray_[(size_t)depth] = 1;
            startContainer(depth);
            openContainers_[(size_t)depth].count++;
            state = ARRAY_VALUE;
        }
        if (state == ARRAY_VALUE) {
            pos = indexer.readIdx();
            idx = indexer.getAndAdvance();
            switch (buffer[idx]) {
            case '{':
                if (buffer[indexer.peek()] == '}') { indexer.advance(); emptyContainer('{', '}'); state = ARRAY_CONTINUE; }
                else state = OBJECT_BEGIN;
                break;
            case '[':
                if (buffer[indexer.peek()] == ']') { indexer.advance(); emptyContainer('[', ']'); state = ARRAY_CONTINUE; }
                else state = ARRAY_BEGIN;
                break;
            default:
                visitPrimitive(idx, pos);
                state = ARRAY_CONTINUE;
            }
        }
        if (state == ARRAY_CONTINUE) {
            switch (buffer[indexer.getAndAdvance()]) {
            case ',':
                openContainers_[(size_t)depth].count++;
                state = ARRAY_VALUE;
                break;
            case ']':
                endContainer('[', ']', depth);
                state = SCOPE_END;
                break;
            default: throw fail(E_NO_COMMA_ARRAY);
            }
        }
    }
    tape_.append(0, Tape::ROOT);                       // visitDocumentEnd :45-48
    tape_.write(0, tape_.getCurrentIdx(), Tape::ROOT);
    if (!indexer.isEnd()) throw fail(E_TRAILING_CONTENT);  // JsonIterator.java:196-198
}

}  // namespace org_simdjson

// ---- C ABI of the parser (include/sjmi.h) ------------------------------------------------------
struct sjmi_parser {
    org_simdjson::SimdJsonParser* p;
    std::string msg;
};

// (the on-demand entry points below: one try / catch for all of them)
namespace {
template <class F>
int odCall(sjmi_parser* h, F&& f) {
    if (!h || !h->p->onDemandReady()) return SJMI_ERR_ARG;
    h->msg.clear();
    try {
        f(h->p->onDemand());
        return 0;
    } catch (const org_simdjson::JsonParsingException& e) {
        h->msg = e.what();
        return e.code();
    } catch (const std::exception& e) {
        h->msg = e.what();
        return SJMI_ERR_HIP;
    }
}
}  // namespace

extern "C" {

int sjmi_parser_create(sjmi_parser** out, int capacity, int max_depth, int device) {
    if (!out) return SJMI_ERR_ARG;
    *out = nullptr;
    try {
        sjmi_parser* h = new sjmi_parser();
        h->p = new org_simdjson::SimdJsonParser(capacity, max_depth, device);
        *out = h;
        return SJMI_OK;
    } catch (const std::exception&) {
        return SJMI_ERR_NO_DEVICE;
    }
}

void sjmi_parser_destroy(sjmi_parser* h) {
    if (!h) return;
    delete h->p;
    delete h;
}

const char* sjmi_parser_last_message(const sjmi_parser* h) { return h ? h->msg.c_str() : ""; }

int sjmi_parser_set_gpu_walk(sjmi_parser* h, int on) {
    if (!h) return SJMI_ERR_ARG;
    h->p->setGpuWalk(on);
    return SJMI_OK;
}

int sjmi_parser_parse_batch(sjmi_parser* h, const uint8_t* buf, uint64_t total_len, const uint64_t* doc_offsets,
                            uint64_t n_docs, const uint64_t** tape, const uint64_t** tape_offsets,
                            const uint8_t** strings, uint64_t* strings_len, const int32_t** errors) {
    if (!h || (!buf && total_len) || !doc_offsets || !tape || !tape_offsets || !strings || !strings_len || !errors)
        return SJMI_ERR_ARG;
    h->msg.clear();
    try {
        try {
            h->p->parseBatch(buf, (size_t)total_len, doc_offsets, (size_t)n_docs);
        } catch (const std::invalid_argument& e) {
            h->msg = e.what();
            return SJMI_ERR_ARG;
        }
        *tape = h->p->batchTape();
        *tape_offsets = h->p->batchTapeOffsets().data();
        *strings = h->p->stringBuffer().data();
        *strings_len = h->p->stringBufferLen();
        *errors = h->p->batchErrors().data();
        return 0;
    } catch (const org_simdjson::JsonParsingException& e) {
        h->msg = e.what();
        return e.code();
    } catch (const std::exception& e) {
        h->msg = e.what();
        return SJMI_ERR_HIP;
    }
}

int sjmi_parser_parse(sjmi_parser* h, const uint8_t* buf, uint64_t len, const uint64_t** tape, uint64_t* tape_len,
                      const uint8_t** strings, uint64_t* strings_len, uint64_t* error_pos) {
    if (!h || (!buf && len) || !tape || !tape_len || !strings || !strings_len) return SJMI_ERR_ARG;
    *tape = nullptr;
    *tape_len = 0;
    *strings = nullptr;
    *strings_len = 0;
    if (error_pos) *error_pos = 0;
    h->msg.clear();
    try {
        h->p->parse(buf, (size_t)len);
        *tape = h->p->tape().data();
        *tape_len = h->p->tape().getCurrentIdx();
        *strings = h->p->stringBuffer().data();
        *strings_len = h->p->stringBufferLen();
        return 0;
    } catch (const org_simdjson::JsonParsingException& e) {
        h->msg = e.what();
        if (error_pos) *error_pos = e.position();
        return e.code();  // > 0: a JSON error (SJMI_E_*)
    } catch (const std::exception& e) {
        h->msg = e.what();
        return SJMI_ERR_HIP;
    }
}

// ---- JsonValue over the C ABI (JsonValue.java:25-111): every accessor goes through org_simdjson::JsonValue ----
namespace {
// the tape a value handle refers to: the last parse() (doc == UINT64_MAX) or document `doc` of the last parse_batch()
struct ValueView {
    std::unique_ptr<org_simdjson::Tape> view;  // a read-only view of one document of the batch tape
    const org_simdjson::Tape* tape;
    const uint8_t* sb;
    ValueView(const sjmi_parser* h, uint64_t doc) : tape(nullptr), sb(h->p->stringBuffer().data()) {
        if (doc == UINT64_MAX) {
            tape = &h->p->tape();
        } else if (doc < h->p->batchErrors().size() && h->p->batchErrors()[doc] == 0) {
            const std::vector<uint64_t>& o = h->p->batchTapeOffsets();
            view.reset(new org_simdjson::Tape(h->p->batchTape() + o[doc], (size_t)(o[doc + 1] - o[doc])));
            tape = view.get();
        }
    }
    bool ok(const sjmi_value* v) const { return tape && v && v->tape_idx >= 1 && v->tape_idx < tape->getCurrentIdx(); }
    org_simdjson::JsonValue value(const sjmi_value* v) const { return org_simdjson::JsonValue(tape, (size_t)v->tape_idx, sb); }
};
}  // namespace

int sjmi_parser_root(const sjmi_parser* h, sjmi_value* out) {
    if (!h || !out || h->p->tape().getCurrentIdx() < 3) return SJMI_ERR_ARG;
    out->doc = UINT64_MAX;
    out->tape_idx = 1;  // TapeBuilder.createJsonValue, TapeBuilder.java:215-217
    return SJMI_OK;
}

int sjmi_parser_batch_root(const sjmi_parser* h, uint64_t doc, sjmi_value* out) {
    if (!h || !out || doc >= h->p->batchErrors().size()) return SJMI_ERR_ARG;
    if (h->p->batchErrors()[doc] != 0) return h->p->batchErrors()[doc];  // > 0: that document's JSON error
    out->doc = doc;
    out->tape_idx = 1;
    return SJMI_OK;
}

int sjmi_value_type(const sjmi_parser* h, const sjmi_value* v) {
    if (!h || !v) return SJMI_ERR_ARG;
    const ValueView vv(h, v->doc);
    if (!vv.ok(v)) return SJMI_ERR_ARG;
    const org_simdjson::JsonValue j = vv.value(v);
    return j.isArray() ? '[' : j.isObject() ? '{' : j.isString() ? '"' : j.isLong() ? 'l' : j.isDouble() ? 'd'
           : j.isNull() ? 'n' : j.isBoolean() ? (j.asBoolean() ? 't' : 'f') : SJMI_ERR_ARG;
}

int sjmi_value_as_long(const sjmi_parser* h, const sjmi_value* v, int64_t* out) {
    if (!h || !v || !out) return SJMI_ERR_ARG;
    const ValueView vv(h, v->doc);
    if (!vv.ok(v) || !vv.value(v).isLong()) return SJMI_ERR_ARG;
    *out = vv.value(v).asLong();
    return SJMI_OK;
}

int sjmi_value_as_double(const sjmi_parser* h, const sjmi_value* v, double* out) {
    if (!h || !v || !out) return SJMI_ERR_ARG;
    const ValueView vv(h, v->doc);
    if (!vv.ok(v) || !vv.value(v).isDouble()) return SJMI_ERR_ARG;
    *out = vv.value(v).asDouble();
    return SJMI_OK;
}

int sjmi_value_as_boolean(const sjmi_parser* h, const sjmi_value* v, int* out) {
    if (!h || !v || !out) return SJMI_ERR_ARG;
    const ValueView vv(h, v->doc);
    if (!vv.ok(v) || !vv.value(v).isBoolean()) return SJMI_ERR_ARG;
    *out = vv.value(v).asBoolean() ? 1 : 0;
    return SJMI_OK;
}

int sjmi_value_as_string(const sjmi_parser* h, const sjmi_value* v, uint8_t* dst, uint64_t dst_capacity, uint64_t* len) {
    if (!h || !v || !len || (!dst && dst_capacity)) return SJMI_ERR_ARG;
    const ValueView vv(h, v->doc);
    if (!vv.ok(v) || !vv.value(v).isString()) return SJMI_ERR_ARG;
    const std::string s = vv.value(v).asString();
    *len = s.size();
    if (s.size() > dst_capacity) return SJMI_ERR_CAPACITY;
    if (!s.empty()) memcpy(dst, s.data(), s.size());
    return SJMI_OK;
}

int sjmi_value_get(const sjmi_parser* h, const sjmi_value* v, const uint8_t* name, uint64_t name_len, sjmi_value* out) {
    if (!h || !v || !out || (!name && name_len)) return SJMI_ERR_ARG;
    const ValueView vv(h, v->doc);
    if (!vv.ok(v) || !vv.value(v).isObject()) return SJMI_ERR_ARG;
    org_simdjson::JsonValue found(nullptr, 0, nullptr);
    if (!vv.value(v).get(std::string(reinterpret_cast<const char*>(name), (size_t)name_len), &found)) return 1;  // Java: null
    out->doc = v->doc;
    out->tape_idx = found.tapeIdx();
    return SJMI_OK;
}

int sjmi_value_size(const sjmi_parser* h, const sjmi_value* v) {
    if (!h || !v) return SJMI_ERR_ARG;
    const ValueView vv(h, v->doc);
    if (!vv.ok(v) || !(vv.value(v).isArray() || vv.value(v).isObject())) return SJMI_ERR_ARG;
    return vv.value(v).getSize();
}

int sjmi_value_first(const sjmi_parser* h, const sjmi_value* v, sjmi_value* out) {
    if (!h || !v || !out) return SJMI_ERR_ARG;
    const ValueView vv(h, v->doc);
    if (!vv.ok(v) || !(vv.value(v).isArray() || vv.value(v).isObject())) return SJMI_ERR_ARG;
    const org_simdjson::JsonValue j = vv.value(v);
    if (j.firstChild() >= j.endChild()) return 1;  // hasNext() == false
    out->doc = v->doc;
    out->tape_idx = j.firstChild();
    return SJMI_OK;
}

int sjmi_value_next(const sjmi_parser* h, const sjmi_value* container, const sjmi_value* child, sjmi_value* out) {
    if (!h || !container || !child || !out) return SJMI_ERR_ARG;
    const ValueView vv(h, container->doc);
    if (!vv.ok(container) || !vv.ok(child) || child->doc != container->doc) return SJMI_ERR_ARG;
    const org_simdjson::JsonValue j = vv.value(container);
    if (!(j.isArray() || j.isObject()) || child->tape_idx < j.firstChild() || child->tape_idx >= j.endChild()) return SJMI_ERR_ARG;
    const size_t nx = j.next((size_t)child->tape_idx);
    if (nx >= j.endChild()) return 1;
    out->doc = container->doc;
    out->tape_idx = nx;
    return SJMI_OK;
}


// ---- the on-demand front end over the C ABI (ondemand.h) ----

int sjmi_parser_ondemand_init(sjmi_parser* h, const uint8_t* buf, uint64_t len, int reserved) {
    if (!h || (!buf && len)) return SJMI_ERR_ARG;
    h->msg.clear();
    try {
        h->p->onDemandInit(buf, (size_t)len);
        return 0;
    } catch (const org_simdjson::JsonParsingException& e) {
        h->msg = e.what();
        return e.code();
    } catch (const std::exception& e) {
        h->msg = e.what();
        return SJMI_ERR_HIP;
    }
}
int sjmi_od_skip_child(sjmi_parser* h, int parent_depth) {
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) { if (parent_depth < 0) it.skipChild(); else it.skipChild(parent_depth); });
}
int sjmi_od_get_boolean(sjmi_parser* h, int root, int nullable, int* is_null, int* value) {
    if (!is_null || !value) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) {
        bool n = false;
        *value = it.getBoolean(root != 0, nullable != 0, &n) ? 1 : 0;
        *is_null = n;
    });
}
int sjmi_od_get_long(sjmi_parser* h, int root, int nullable, int* is_null, int64_t* value) {
    if (!is_null || !value) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) {
        bool n = false;
        *value = it.getLong(root != 0, nullable != 0, &n);
        *is_null = n;
    });
}
int sjmi_od_get_integral(sjmi_parser* h, int bits, int root, int nullable, int* is_null, int64_t* value) {
    if (!is_null || !value || (bits != 8 && bits != 16 && bits != 32 && bits != 64)) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) {
        bool n = false;
        *value = it.getLong(root != 0, nullable != 0, &n, bits);
        *is_null = n;
    });
}
int sjmi_od_get_double(sjmi_parser* h, int root, int nullable, int* is_null, double* value) {
    if (!is_null || !value) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) {
        bool n = false;
        *value = it.getDouble(root != 0, nullable != 0, &n);
        *is_null = n;
    });
}
int sjmi_od_get_float(sjmi_parser* h, int root, int nullable, int* is_null, float* value) {
    if (!is_null || !value) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) {
        bool n = false;
        *value = it.getFloat(root != 0, nullable != 0, &n);
        *is_null = n;
    });
}
int sjmi_od_get_char(sjmi_parser* h, int root, int nullable, int* is_null, uint16_t* utf16_unit) {
    if (!is_null || !utf16_unit) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) {
        bool n = false;
        *utf16_unit = it.getChar(root != 0, nullable != 0, &n);
        *is_null = n;
    });
}
int sjmi_od_get_string(sjmi_parser* h, int root, int* is_null, const uint8_t** bytes, uint64_t* len) {
    if (!is_null || !bytes || !len) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) {
        bool n = false;
        const std::vector<uint8_t>& s = it.getString(root != 0, &n);
        *is_null = n;
        *bytes = s.data();
        *len = s.size();
    });
}
int sjmi_od_get_field_name(sjmi_parser* h, const uint8_t** bytes, uint64_t* len) {
    if (!bytes || !len) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) {
        const std::vector<uint8_t>& s = it.getFieldName();
        *bytes = s.data();
        *len = s.size();
    });
}
int sjmi_od_start_array(sjmi_parser* h, int root, int* result) {
    if (!result) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) { *result = (int)it.startIteratingArray(root != 0); });
}
int sjmi_od_next_array_element(sjmi_parser* h, int* has_next) {
    if (!has_next) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) { *has_next = it.nextArrayElement() ? 1 : 0; });
}
int sjmi_od_start_object(sjmi_parser* h, int root, int* result) {
    if (!result) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) { *result = (int)it.startIteratingObject(root != 0); });
}
int sjmi_od_next_object_field(sjmi_parser* h, int* has_next) {
    if (!has_next) return SJMI_ERR_ARG;
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) { *has_next = it.nextObjectField() ? 1 : 0; });
}
int sjmi_od_move_to_field_value(sjmi_parser* h) {
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) { it.moveToFieldValue(); });
}
int sjmi_od_assert_no_more_values(sjmi_parser* h) {
    return odCall(h, [&](org_simdjson::OnDemandJsonIterator& it) { it.assertNoMoreJsonValues(); });
}
int sjmi_od_depth(const sjmi_parser* h) { return (h && h->p->onDemandReady()) ? h->p->onDemand().getDepth() : SJMI_ERR_ARG; }
int sjmi_od_peek(const sjmi_parser* h) {
    if (!h || !h->p->onDemandReady()) return SJMI_ERR_ARG;
    return h->p->bitIndexes().isEnd() || h->p->bitIndexes().isPastEnd() ? SJMI_OD_END : (int)h->p->onDemand().peekByte();
}

}  // extern "C"
