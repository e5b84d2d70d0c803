// simdjson_parser.h -- C++ host-side mirror of the reference's public interface for the hot path:
//   org.simdjson.SimdJsonParser.parse(byte[], int) -> JsonValue
//   (/root/reference/src/main/java/org/simdjson/SimdJsonParser.java:15-58, JsonValue.java:25-111)
// Same class names, same argument meaning, same error messages.  Stage 1 (UTF-8 validation, structural
// indexing) and string unescaping run on the MI355X through the C ABI (include/sjmi.h); the sequential
// stage-2 tree builder (JsonIterator + TapeBuilder + Tape) stays on the host, as in the reference.
// There is no CPU implementation of stage 1 / string unescape here: without a GPU the constructor throws.
#pragma once
#include <stdint.h>

#include <exception>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/sjmi.h"

namespace org_simdjson {

// JsonParsingException.java:3-12 (unchecked; the message is part of the tested contract)
class JsonParsingException : public std::runtime_error {
public:
    JsonParsingException(int code, const std::string& msg, uint64_t pos = 0)
        : std::runtime_error(msg), code_(code), pos_(pos) {}
    int code() const { return code_; }
    uint64_t position() const { return pos_; }

private:
    int code_;
    uint64_t pos_;
};

// BitIndexes.java:5-101 -- stage-1 output container + stage-2 cursor
class BitIndexes {
public:
    explicit BitIndexes(size_t capacity) : own_(capacity), indexes_(own_.data()), capacity_(capacity) {}
    // a cursor over somebody else's index array (batch: one per worker thread over the shared array)
    BitIndexes(uint32_t* shared, size_t capacity) : indexes_(shared), capacity_(capacity) {}
    BitIndexes(const BitIndexes&) = delete;
    BitIndexes& operator=(const BitIndexes&) = delete;
    void rebind(uint32_t* shared, size_t capacity) { indexes_ = shared; capacity_ = capacity; reset(); }
    uint32_t* array() { return indexes_; }                  // filled by the engine (replaces write(), :14-41)
    size_t capacity() const { return capacity_; }
    void setWriteIdx(size_t n) { writeIdx_ = n; }            // replaces finish() (:82-96): sentinel written by the engine
    // batch: restrict the cursor to one document's slice [from, to) of the shared index array; the sentinel read
    // (index `to`) must land on the document's first structural, as BitIndexes.finish arranges (:82-96)
    void window(size_t from, size_t to, uint32_t docStart) { readIdx_ = from; base_ = from; writeIdx_ = to; sentinel_ = docStart; }
    void advance() { ++readIdx_; }                                                    // :47-49
    uint32_t getAndAdvance() { return at(readIdx_++); }                               // :51-54
    uint32_t getLast() const { return indexes_[writeIdx_ - 1]; }                      // :56-58
    uint32_t advanceAndGet() { return at(++readIdx_); }                               // :60-63
    uint32_t peek() const { return at(readIdx_); }                                    // :65-68
    bool hasNext() const { return writeIdx_ > readIdx_; }                             // :70-72
    bool isEnd() const { return writeIdx_ == readIdx_; }                              // :74-76
    bool isPastEnd() const { return readIdx_ > writeIdx_; }                           // :78-80
    void setReadIdx(size_t r) { readIdx_ = r; }   // the on-demand cursor's table-driven skipChild (ondemand.h)
    bool isEmpty() const { return writeIdx_ == base_; }
    size_t readIdx() const { return readIdx_; }
    size_t writeIdx() const { return writeIdx_; }
    void reset() { writeIdx_ = 0; readIdx_ = 0; base_ = 0; sentinel_ = 0; }                          // :98-101

private:
    // reads past the sentinel (never observed for any input: the sentinel read already fails the grammar) are defined as 0
    // at/after writeIdx the reference reads its sentinel 0 = the document's first byte (BitIndexes.java:82-96)
    uint32_t at(size_t i) const { return i < writeIdx_ ? indexes_[i] : sentinel_; }
    std::vector<uint32_t> own_;
    uint32_t* indexes_;
    size_t capacity_;
    size_t writeIdx_ = 0, readIdx_ = 0, base_ = 0;
    uint32_t sentinel_ = 0;
};

// Tape.java:5-98
class Tape {
public:
    static constexpr char ROOT = 'r', START_ARRAY = '[', START_OBJECT = '{', END_ARRAY = ']', END_OBJECT = '}',
                          STRING = '"', INT64 = 'l', DOUBLE = 'd', TRUE_VALUE = 't', FALSE_VALUE = 'f', NULL_VALUE = 'n';
    explicit Tape(size_t capacity) : own_(capacity), tape_(own_.data()), capacity_(capacity) {}
    // read-only view of a finished tape somewhere else (one document of batchTape()): JsonValue only reads
    Tape(const uint64_t* words, size_t n) : tape_(const_cast<uint64_t*>(words)), capacity_(n), idx_(n) {}
    Tape(const Tape&) = delete;
    Tape& operator=(const Tape&) = delete;
    // batch: build the next document's tape in place at `at` (room for `capacity` words) instead of in own storage
    void rebase(uint64_t* at, size_t capacity) { tape_ = at; capacity_ = capacity; idx_ = 0; }
    void append(uint64_t val, char type) { tape_[idx_++] = val | ((uint64_t)(uint8_t)type << 56); }  // :28-31
    void appendInt64(int64_t v) { append(0, INT64); tape_[idx_++] = (uint64_t)v; }                   // :33-37
    void appendDouble(double v);                                                                      // :39-43
    void write(size_t i, uint64_t val, char type) { tape_[i] = val | ((uint64_t)(uint8_t)type << 56); }  // :45-47
    void skip() { ++idx_; }
    void reset() { idx_ = 0; }
    size_t getCurrentIdx() const { return idx_; }
    char getType(size_t i) const { return (char)(tape_[i] >> 56); }
    uint64_t getValue(size_t i) const { return tape_[i] & 0x00FFFFFFFFFFFFFFull; }
    int64_t getInt64Value(size_t i) const { return (int64_t)tape_[i + 1]; }
    double getDouble(size_t i) const;
    size_t getMatchingBraceIndex(size_t i) const { return (size_t)(uint32_t)tape_[i]; }
    int getScopeCount(size_t i) const { return (int)((tape_[i] >> 32) & 0xFFFFFF); }
    size_t computeNextIndex(size_t i) const;                                                          // :86-98
    const uint64_t* data() const { return tape_; }
    uint64_t* raw() { return tape_; }                 // filled by the GPU walker (sjmi_parse_document)
    void setCurrentIdx(size_t n) { idx_ = n; }
    size_t capacity() const { return capacity_; }

private:
    std::vector<uint64_t> own_;
    uint64_t* tape_;
    size_t capacity_;
    size_t idx_ = 0;
};

class SimdJsonParser;

// JsonValue.java:18-221 -- read-only DOM view over (tape, tapeIdx, stringBuffer)
class JsonValue {
public:
    JsonValue(const Tape* tape, size_t tapeIdx, const uint8_t* stringBuffer) : tape_(tape), idx_(tapeIdx), sb_(stringBuffer) {}
    bool isArray() const { return tape_->getType(idx_) == Tape::START_ARRAY; }
    bool isObject() const { return tape_->getType(idx_) == Tape::START_OBJECT; }
    bool isLong() const { return tape_->getType(idx_) == Tape::INT64; }
    bool isDouble() const { return tape_->getType(idx_) == Tape::DOUBLE; }
    bool isBoolean() const { char t = tape_->getType(idx_); return t == Tape::TRUE_VALUE || t == Tape::FALSE_VALUE; }
    bool isNull() const { return tape_->getType(idx_) == Tape::NULL_VALUE; }
    bool isString() const { return tape_->getType(idx_) == Tape::STRING; }
    int64_t asLong() const { return tape_->getInt64Value(idx_); }
    double asDouble() const { return tape_->getDouble(idx_); }
    bool asBoolean() const { return tape_->getType(idx_) == Tape::TRUE_VALUE; }
    std::string asString() const;                    // :79-89
    bool get(const std::string& name, JsonValue* out) const;  // :91-107 (returns false where Java returns null)
    int getSize() const { return tape_->getScopeCount(idx_); }
    // iteration (arrayIterator / objectIterator, :143-194)
    size_t firstChild() const { return idx_ + 1; }
    size_t endChild() const { return tape_->getMatchingBraceIndex(idx_) - 1; }
    size_t next(size_t childIdx) const { return tape_->computeNextIndex(childIdx); }
    JsonValue at(size_t tapeIdx) const { return JsonValue(tape_, tapeIdx, sb_); }
    size_t tapeIdx() const { return idx_; }

private:
    const Tape* tape_;
    size_t idx_;
    const uint8_t* sb_;
};

// The sequential stage 2 of ONE document: JsonIterator.walkDocument (JsonIterator.java:26-200) driving TapeBuilder
// (TapeBuilder.java:41-217) over the GPU-made structural indexes and string records.  Everything it reads (document
// bytes, index array, string buffer) is shared and read-only; cursor, tape and container stacks are its own, so a
// batch can be walked by several of them in parallel threads.
class DocWalker {
public:
    DocWalker(const uint8_t* padded, uint32_t* indexes, size_t indexCapacity, size_t tapeCapacity, int maxDepth)
        : paddedBuffer_{padded}, bitIndexes_(indexes, indexCapacity), tape_(tapeCapacity), maxDepth_(maxDepth),
          openContainers_((size_t)maxDepth), isArray_((size_t)maxDepth) {}
    void setStringBuffer(const uint8_t* sb) { stringBuffer_.p = sb; }
    // point the walker at another document buffer / index array (batch: one per sub-batch)
    void rebind(const uint8_t* padded, uint32_t* indexes, size_t indexCapacity) {
        paddedBuffer_.p = padded;
        bitIndexes_.rebind(indexes, indexCapacity);
    }
    // walk the document whose structurals are the current window / contents of bitIndexes(); endOffset = its end
    void walkDocument(size_t endOffset);
    // reset (SimdJsonParser.java:50-53) for the document starting at byte docBase whose first string record (if any)
    // is at stringBufferIdx; the index cursor is set by the caller (bitIndexes().reset()/window())
    void resetForDocument(size_t docBase, size_t stringBufferIdx) {
        tape_.reset();
        docBase_ = docBase;
        stringBufferIdx_ = stringBufferIdx;
    }
    BitIndexes& bitIndexes() { return bitIndexes_; }
    const BitIndexes& bitIndexes() const { return bitIndexes_; }
    Tape& tape() { return tape_; }
    const Tape& tape() const { return tape_; }
    size_t stringBufferIdx() const { return stringBufferIdx_; }

private:
    struct Bytes {
        const uint8_t* p;
        const uint8_t* data() const { return p; }
    };
    static constexpr int PADDING = 64;
    void visitString(uint32_t idx, size_t indexPos);  // TapeBuilder.visitString :174-177 (record already on the GPU-made buffer)
    void visitPrimitive(uint32_t idx, size_t indexPos);
    void visitRootPrimitive(uint32_t idx, size_t indexPos, size_t endOffset);
    void parseNumber(const uint8_t* p);
    void emptyContainer(char start, char end);

    Bytes paddedBuffer_, stringBuffer_{nullptr};
    BitIndexes bitIndexes_;
    Tape tape_;
    int maxDepth_;
    size_t stringBufferIdx_ = 0;
    struct OpenContainer { size_t tapeIndex; uint32_t count; };
    std::vector<OpenContainer> openContainers_;
    std::vector<uint8_t> isArray_;
    size_t docBase_ = 0;  // batch: byte offset of the current document (error positions are document-relative)
};

class WorkerPool;  // simdjson_parser.cpp: the host threads of parseBatch
class OnDemandJsonIterator;  // ondemand.h

// SimdJsonParser.java:3-59
class SimdJsonParser {
public:
    static constexpr int PADDING = 64;                        // :5
    static constexpr int DEFAULT_CAPACITY = 34 * 1024 * 1024;  // :6
    static constexpr int DEFAULT_MAX_DEPTH = 1024;            // :7
    SimdJsonParser() : SimdJsonParser(DEFAULT_CAPACITY, DEFAULT_MAX_DEPTH) {}
    SimdJsonParser(int capacity, int maxDepth, int device = 0);
    ~SimdJsonParser();
    SimdJsonParser(const SimdJsonParser&) = delete;
    SimdJsonParser& operator=(const SimdJsonParser&) = delete;

    // parse(byte[] buffer, int len) :35-40.  Only buffer[0,len) is read.  The returned JsonValue aliases
    // parser-owned memory and is invalidated by the next parse(), exactly like the reference.
    JsonValue parse(const uint8_t* buffer, size_t len);
    // where stage 2 of parse() runs: 1 = on the GPU (the cooperative walker), 0 = the host walker, -1 (default) = by size --
    // the GPU from GPU_WALK_AUTO_BYTES on, where the call is the faster one (sjmi_parser_parse timed from C++,
    // tools/single_doc_sizes.py, end of round 3: 0.10 vs 0.05 ms at 1 KiB, 0.108 vs 0.072 ms at 14 KiB, 0.131-0.146 vs 0.148 ms
    // at 136 KiB, 0.148 vs 0.188 ms for twitter.json, 0.25 vs 0.67 ms at 1 MiB, 1.43 vs 9.9 ms at 16 MiB, 5.05 vs 41 ms at 64 MiB).
    // Round 2 switched at 1 MiB (a parse-and-select of twitter.json was then slower with the GPU-built tape: a tape the host
    // built is still in its caches), round 3 at 256 KiB and then, with the latency work on the single-document path, at 128 KiB:
    // the reference's own fixture is parsed with all three stages on the device by default.  Identical results either way.
    static constexpr size_t GPU_WALK_AUTO_BYTES = 128u << 10;
    void setGpuWalk(int mode) { gpuWalk_ = mode < 0 ? -1 : (mode ? 1 : 0); }

    // Batched parse: documents packed NDJSON-style at doc_offsets[k] (n+1 entries).  One GPU pass for the batch (isolated
    // stage 1 + string unescape + per-document string offsets), then the host stage 2 of the documents on several
    // threads.  Fills batchTape()/batchTapeOffsets()/batchErrors(); a broken document only affects its own entry.
    void parseBatch(const uint8_t* buffer, size_t totalLen, const uint64_t* docOffsets, size_t nDocs);
    const uint64_t* batchTape() const { return batchTape_.get(); }  // batchTapeLen() words; document k: [offsets[k], offsets[k+1])
    size_t batchTapeLen() const { return batchTapeLen_; }
    const std::vector<uint64_t>& batchTapeOffsets() const { return batchTapeOffsets_; }
    const std::vector<int32_t>& batchErrors() const { return batchErrors_; }

    // The on-demand front end (ondemand.h): pad + GPU stage 1 (+ the GPU skip table) + iterator.init, i.e. the head of
    // SimdJsonParser.parse(byte[], int, Class<T>) (SimdJsonParser.java:31-33 / SchemaBasedJsonIterator.java:29-41)
    void onDemandInit(const uint8_t* buffer, size_t len);
    OnDemandJsonIterator& onDemand() { return *onDemand_; }
    bool onDemandReady() const { return onDemandReady_; }

    const Tape& tape() const { return walker_.tape(); }
    const std::vector<uint8_t>& stringBuffer() const { return stringBuffer_; }
    size_t stringBufferLen() const { return stringBufferLen_; }
    const BitIndexes& bitIndexes() const { return walker_.bitIndexes(); }

private:
    void stage1(const uint8_t* buffer, size_t len);   // :55-58 -> GPU
    void growStringBuffer(size_t need);

    sjmi_ctx* ctx_ = nullptr;
    sjmi_ctx* ctx2_ = nullptr;  // parseBatch: the second stream of the sub-batch pipeline (created on first use)
    int capacity_, maxDepth_, device_;
    std::vector<uint8_t> stringBuffer_, paddedBuffer_;
    std::vector<uint32_t> indexes_;  // BitIndexes storage (filled by the engine)
    DocWalker walker_;
    size_t stringBufferLen_ = 0;
    std::unique_ptr<uint64_t[]> batchTape_;  // (not a vector: grown without zero-filling, kept between batches)
    size_t batchTapeLen_ = 0, batchTapeRoom_ = 0;
    std::vector<uint64_t> batchTapeOffsets_, indexOffsets_, docStringOffsets_;
    int gpuWalk_ = -1;
    int batchThreads_ = 1;  // host threads walking the documents of a batch (SJMI_PARSE_THREADS overrides)
    int batchPipeline_ = 0;  // sub-batches per batch, 0 = by size (SJMI_PARSE_PIPELINE overrides)
    // kept between batches: a walker per batch thread, and per (sub-batch, thread) the slab its tapes are built in
    struct BatchLane {
        std::unique_ptr<DocWalker> walker;
        std::exception_ptr error;
    };
    struct TapeSlab {
        std::unique_ptr<uint64_t[]> words;
        size_t room = 0, used = 0;
    };
    std::vector<BatchLane> lanes_;
    std::vector<TapeSlab> pieces_;
    std::unique_ptr<WorkerPool> pool_;  // created by the first parseBatch
    std::vector<uint32_t> docStatus_;
    void* pinned_[4] = {nullptr, nullptr, nullptr, nullptr};  // page-locked parser buffers (sjmi_host_register)
    bool stagedInput_ = false;  // paddedBuffer_ is the engine's input staging (sjmi_set_input_staging)
    std::vector<int32_t> batchErrors_;
    std::unique_ptr<OnDemandJsonIterator> onDemand_;
    bool onDemandReady_ = false;
};

}  // namespace org_simdjson
