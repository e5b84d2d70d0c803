// stage1.h -- internal declarations shared by the HIP kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sjmi.h"

namespace sjmi {

// written by the kernel into the workspace header; mirrors sjmi_stage1_result in include/sjmi.h
struct Stage1Result {
    unsigned long long count;  // number of structural indexes (BitIndexes.writeIdx)
    uint32_t status;           // SJMI_ST_* bits
    uint32_t reserved;
};
static_assert(sizeof(Stage1Result) == sizeof(sjmi_stage1_result), "ABI struct mismatch");

// device-side result of one unescape call; mirrors sjmi_unescape_result in include/sjmi.h
struct UnescapeResult {
    unsigned long long total_bytes;      // bytes of [be32 len][bytes] records
    unsigned long long first_error_inv;  // 0 = no error, else ~((byte position | structural position << 8) | SJMI_E_* code)
    uint32_t flags;                      // bit0: string buffer capacity exceeded; bits 2-3: engine fault
    uint32_t reserved;                   // strings.hip: number of string literals
};
static_assert(sizeof(UnescapeResult) == sizeof(sjmi_unescape_result), "ABI struct mismatch");

// sjmi_parse_document: the result records of the three stages in one place (one D2H), written by the walk's LAST launch
struct WalkResult;
struct SingleDocPack;
struct SingleDocTail {          // where that launch finds / leaves them; pack == nullptr: no such tail
    const Stage1Result* s1 = nullptr;
    const UnescapeResult* u = nullptr;
    SingleDocPack* pack = nullptr;
    bool optimistic = false;    // the chunk path ends in k_chunk_finish and nothing is queued behind it: a document it does not
                                // take (pack->fallback != 0) is the caller's to run again with no_chunks
    bool no_chunks = false;     // the single-wave sweep only
    uint64_t in_place_cap = 0;  // != 0: d_tape is the CALLER's tape (mapped host memory) with this many words: the walkers write
                                // it in place whatever the structural count's bound says, clamped to it
};

// device-side result of one batch walk; mirrors sjmi_walk_result in include/sjmi.h
struct WalkResult {
    unsigned long long tape_words, host_documents, failed_documents;
    uint32_t flags, reserved;
};
static_assert(sizeof(WalkResult) == sizeof(sjmi_walk_result), "ABI struct mismatch");
struct SingleDocPack {
    Stage1Result s1;
    UnescapeResult u;
    WalkResult w;
    unsigned long long to[2];
    int32_t err;
    uint32_t fallback;
};
static_assert(sizeof(SingleDocPack) <= 256, "d_single / h_single hold 256 bytes of results");

// ablation switches for performance experiments only (results are NOT valid with any of them set)
constexpr uint32_t DBG_NO_WRITE = 1, DBG_NO_LOOKBACK = 2;
// test hook: a fast-mode launch reports SJMI_ST_INTERNAL as if its look-back spin had tripped
constexpr uint32_t DBG_FAKE_TIMEOUT = 16;
// test hook: launch only 8 worker workgroups (as if the GPU were shared with other work)
constexpr uint32_t DBG_SMALL_GRID = 32;
// kernel flag (not an ablation): SAFE liveness mode -- no scanner workgroup, every worker looks back itself
constexpr uint32_t FLAG_SAFE = 0x100;
// shards of a longer document (sjmi_stage1_shard_device): the shard is entered inside a string; it is not the document's
// last shard (no tail block); bits 16..31 = readable 64-byte blocks in front of the buffer (left halo)
constexpr uint32_t FLAG_ENTRY_PARITY = 0x1000;
constexpr uint32_t FLAG_NO_TAIL = 0x4000;
// the shard's left halo begins at the document's first byte (a backslash run that fills it is complete)
constexpr uint32_t FLAG_HALO_FROM_START = 0x2000;
// kernel flag: FAST mode without static first granules (several contexts may be launching concurrently)
constexpr uint32_t FLAG_ALL_TICKETS = 0x800;

// workspace layout (zeroed by one hipMemsetAsync per launch)
constexpr size_t WS_RESULT_OFFSET = 0;        // Stage1Result (16 bytes)
constexpr uint32_t WS_SCANNER_CU_WORD = 8;    // u32 index from the workspace start: id of the scanner's CU
constexpr size_t WS_TICKET_OFFSET = 64;       // 8 u32 granule tickets, each alone in its 64-byte line
constexpr size_t WS_BATCH_FLAGS_OFFSET = 576;  // fused batch pipeline: u32 [0] = why the plain pass is rejected (0: it is not), [1] != 0:
                                               // accepted (written by k_batch_layout).  In the header of the plain pass's OWN workspace
                                               // half: zero when the launch starts, like the rest of it -- no memset of their own
constexpr size_t WS_TILE_STATE_OFFSET = 640;   // u64 aggregates[granules], then u64 prefixes[granules]

size_t stage1_workspace_bytes(uint64_t len, int steps);
struct Stage1Prefixes {
    const unsigned long long* pfx = nullptr;
    uint64_t ngran = 0;
    uint32_t granule_bytes = 0;
};
Stage1Prefixes stage1_prefixes(const void* d_ws, uint64_t len, int steps);
int stage1_pick_steps(uint64_t len);
size_t stage1_block_entries(uint64_t len);
// sjmi_parse_document: the delimiters of ONE document for the batch kernels behind stage 1 (doc / index / string offsets, status)
// and the two small records the walk wants zeroed -- written by the stage-1 scanner together with the result record
struct Stage1Single {
    unsigned long long* doc_offsets = nullptr;
    unsigned long long* index_offsets = nullptr;  // nullptr: nothing to do
    uint32_t* doc_status = nullptr;
    unsigned long long* doc_str_offsets = nullptr;
    uint32_t* walk_result = nullptr;   // sizeof(WalkResult) bytes
    uint32_t* slow_header = nullptr;   // 64 bytes
};
// optional work folded into the kernel (device-resident path): see zero_next_workspace / scanner_wave in stage1.hip
struct Stage1Extras {
    bool workspace_is_zero = false;  // skip the workspace memset (a previous launch zeroed this workspace)
    void* zero_next = nullptr;       // workspace the NEXT launch will use: zeroed by this one (16-byte aligned)
    size_t zero_bytes = 0;           //   ... this many bytes of it (multiple of 16)
    void* result_out = nullptr;      // device sjmi_stage1_result written by the scanner (FAST mode only)
    const uint32_t* skip = nullptr;  // device flag: != 0 -> the launch does nothing (fused batch pipeline)
    void* zero2 = nullptr;           // a second region zeroed by the workers on their way out (16-byte aligned, multiple of 16 bytes):
    size_t zero2_bytes = 0;          //   the string pass's workspace on the single-document latency path (one memset less)
    Stage1Single single;             // ... and what k_single_doc_setup would write, done by the scanner (FAST mode; one launch less)
    void* blkidx = nullptr;          // batch side outputs (both or neither): u32 per 64-byte block = index position of the block's
    void* blkw = nullptr;            //   first structural; u16 per block = its tape words for entry parity 0 | 1 << 8; stage1_block_entries(len) each
    void* blkpar = nullptr;          // side output for strings.hip: u64 per 4 KiB of input, bit l = block l is entered inside a
                                     // string (StructuralIndexer.java:233-234's prevInString, per block); len / 4096 + 4 words
};
// ev_start/ev_stop (optional) are attached to the kernel's dispatch (not the workspace memset)
hipError_t stage1_launch(const uint8_t* d_buf, uint64_t len, uint32_t* d_out, uint64_t out_cap, void* d_ws, int steps,
                         hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t dbg = 0,
                         const Stage1Extras& ex = Stage1Extras());
// strings.hip: the string buffer of a document in one streaming pass (from the bytes and stage 1's block parities)
size_t strings_workspace_bytes(uint64_t len);
size_t strings_parity_words(uint64_t len);
// the fused batch pipeline decides ON THE DEVICE whether the string pass runs over the batch itself (*d_sel != 0) or over its
// sanitized copy (d_buf / d_blkpar here)
struct StringsAlt {
    const uint32_t* d_sel = nullptr;
    const uint8_t* d_buf = nullptr;
    const unsigned long long* d_blkpar = nullptr;
};
hipError_t strings_launch(const uint8_t* d_buf, uint64_t len, const unsigned long long* d_blkpar, uint8_t* d_sb, uint64_t sb_cap,
                          uint32_t* d_soff, uint64_t soff_cap, uint32_t* d_blk_ord, void* d_ws, UnescapeResult* d_res,
                          hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                          const StringsAlt& alt = StringsAlt(), bool workspace_is_zero = false, const uint32_t* d_skip = nullptr);
// batches whose documents were indexed one by one: the copy the string pass runs on (failed documents blanked); d_skip != null
// and *d_skip != 0: nothing to do (the optimistic plain pass was accepted)
hipError_t strings_sanitize_launch(const uint8_t* d_buf, uint64_t total_len, const unsigned long long* d_doc_offsets,
                                   const unsigned long long* d_index_offsets, uint64_t n_docs, uint8_t* d_copy, const uint32_t* d_skip,
                                   hipStream_t stream, const uint32_t* d_doc_status = nullptr);
// (d_doc_status != nullptr: a document is blanked when its stage-1 verdict is not 0 -- the repair pass, which has no index offsets
//  yet; else when it has no structurals)
// per document: ordinal of its first string in the record table (and, optionally, that record's offset)
hipError_t strings_doc_ordinals_launch(const uint8_t* d_buf, const unsigned long long* d_blkpar, const StringsAlt& alt, uint64_t len,
                                       const unsigned long long* d_doc_offsets, uint64_t n_docs, const uint32_t* d_blk_ord,
                                       const uint32_t* d_soff, const UnescapeResult* d_res, unsigned long long* d_doc_ord,
                                       unsigned long long* d_doc_str_offsets, hipStream_t stream, const uint32_t* d_skip = nullptr);
size_t strings_parse_pack_bytes();
hipError_t strings_error_index_pack_launch(const uint32_t* d_idx, const Stage1Result* dev_count, const UnescapeResult* d_res, void* d_pack,
                                           hipStream_t stream);
UnescapeResult* strings_workspace_result(void* d_ws);
hipError_t strings_error_index_launch(const uint32_t* d_idx, uint64_t count, const Stage1Result* dev_count, const UnescapeResult* d_res,
                                      unsigned long long* d_out, hipStream_t stream);
// the string pass of a batch (sjmi_unescape_batch_device): document / index offsets on the device (n_docs + 1 entries each) and,
// optionally, where the string-buffer offset of every document's first record goes
struct UnescapeBatch {
    const unsigned long long* d_doc_offsets = nullptr;
    const unsigned long long* d_index_offsets = nullptr;
    uint64_t n_docs = 0;
    unsigned long long* d_doc_str_offsets = nullptr;
};
hipError_t split_docs_launch(const uint32_t* d_idx, const Stage1Result* d_res, const unsigned long long* d_doc_offsets,
                             uint64_t n_docs, unsigned long long* d_index_offsets, hipStream_t stream);
size_t batch_isolated_workspace_bytes(uint64_t n_docs);
// What the batch walker (coop_walk.hip k_tok_stream) reads per document, one 32-byte record instead of six arrays: n_docs + 1
// records, the last one carrying only `tape` (a document's room on the tape = the next record's `tape` - its own).
struct DocMeta {
    uint32_t from, to;            // its structurals: indexes[from, to)
    uint32_t dso;                 // ordinal of its first string in the string pass's record table
    uint32_t doc_start, doc_end;  // its bytes
    uint32_t st;                  // its stage-1 status (SJMI_ST_*)
    uint32_t tape_lo, tape_hi;    // where its tape goes (word offset)
};
static_assert(sizeof(DocMeta) == 32, "one s_load_dwordx8");
// The accepted plain pass of the fused batch pipeline, everything a document needs from it in ONE pass over the documents
// (batch.hip k_doc_prepare; replaces the binary search of k_split_docs_accept and k_doc_str_ordinals): index_offsets, doc_status
// = 0, the ordinal / record offset of its first string, its PREDICTED tape length (exact for a well-formed document) and the
// DocMeta record without `tape`; chunk_sums[j] = predicted words of documents [256 j, 256 j + 256).
struct DocPrepare {
    const uint8_t* buf;
    const uint32_t* idx;
    const unsigned long long* doc_offsets;
    uint64_t n_docs, total_len;
    const uint32_t* blkidx;
    const uint16_t* blkw;
    const unsigned long long* blkpar;
    const uint32_t* blk_ord;
    const uint32_t* soff;
    const UnescapeResult* strings;
    const Stage1Result* stage1;
    uint32_t* flags;        // [0]: reject bits -- this kernel adds "a separator is missing / the offsets do not cover the buffer"
    unsigned long long* index_offsets;
    uint32_t* doc_status;
    unsigned long long* doc_ord;
    unsigned long long* doc_str_offsets;
    uint32_t* lens;
    unsigned long long* chunk_sums;
    DocMeta* metas;
    // the REPAIR pass (round 6, sjmi_api.hip parse_batch_pipeline stage B): the same kernel over the sanitized copy of a batch
    // whose plain pass was rejected -- the documents that failed stage 1 (status_in[k] != 0, from the verdict pass
    // k_doc_pass<false>) are blank there, so every surviving document begins and ends outside a string whatever separates them
    const uint32_t* gate = nullptr;       // device flag: != 0 -> nothing to do
    const uint32_t* status_in = nullptr;  // per-document stage-1 verdicts to keep (nullptr: every document passed)
    const uint8_t* boundary_buf = nullptr;  // the ORIGINAL batch: a surviving document that ends in a backslash there fails the boundary
                                          // rule (the sanitized copy has an odd trailing backslash run shortened by one)
    bool relaxed = false;                 // boundary rule: the last byte of a document must not be a non-quote scalar character
                                          // (instead of: must be a control-character separator)
};
constexpr int PREP_DOCS = 256;  // documents per workgroup of k_doc_prepare = per chunk of the tape-offset scan
hipError_t batch_prepare_launch(const DocPrepare& a, hipStream_t stream);
// walk.hip k_batch_layout, queued behind k_doc_prepare: DECIDES whether the plain pass is accepted (flags[0] == 0 and a clean
// stage-1 verdict) -> flags[1]; zeroes the walk's result record, the literal list's header and the exact walker's list; hands the
// string pass's record (inside its workspace) to the caller's -- or zeroes that for the per-document string pass to fill;
// accepted: the scan of the predicted tape lengths (chunk bases, tape_offsets[n_docs], walk.tape_words).  optimistic_only: a
// rejected batch gets SJMI_ST_REJECTED in the caller's stage-1 record (nothing else is queued for it).
struct BatchLayout {
    uint32_t* flags;
    const Stage1Result* stage1;          // the plain pass's record as the kernel left it
    Stage1Result* stage1_out;            // the caller's (optimistic_only: |= SJMI_ST_REJECTED)
    const UnescapeResult* strings_ws;    // the string pass's own record
    UnescapeResult* strings_out;         // the caller's
    WalkResult* walk;
    void* slow_header;                   // 64 bytes
    uint32_t* list;
    unsigned long long* chunk_sums;
    uint64_t n_docs, tape_capacity;
    unsigned long long* tape_offsets;
    bool optimistic_only;
    // round 6: the pipeline's stage flags (sjmi_api.hip): pipe_flags[0] != 0 = the plain pass over the batch itself was accepted
    // (stage A), pipe_flags[1] != 0 = A or the repair pass over the sanitized copy (stage B) was -- the tapes are laid out
    uint32_t* pipe_flags = nullptr;
    int stage = 0;                         // 0: A (writes both flags), 1: B (writes [1])
    const uint32_t* gate = nullptr;        // device flag: != 0 -> nothing to do (stage B behind an accepted A)
    const uint32_t* status_or = nullptr;   // stage B: OR of the documents' verdicts, into the caller's stage-1 record
};
hipError_t batch_reject_launch(Stage1Result* d_stage1, hipStream_t stream);  // sjmi_parse_batch_device_optimistic on a batch it cannot even try
hipError_t batch_layout_launch(const BatchLayout& a, void* d_ws, uint64_t count, const uint32_t* lens, DocMeta* metas,
                               int32_t* d_doc_errors, hipStream_t stream);
// walk.hip: stage 2 of every document of a batch (the cooperative walker + packing of the tapes); d_doc_str_ordinals[k] =
// ordinal of document k's first string in the record table d_soff of the string pass (strings.hip)
size_t walk_workspace_bytes(uint64_t count, uint64_t n_docs);
hipError_t walk_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, const uint32_t* d_idx,
                       uint64_t count, const unsigned long long* d_index_offsets, const uint32_t* d_doc_status,
                       const uint8_t* d_sb, const unsigned long long* d_doc_str_ordinals, uint64_t string_base, int max_depth,
                       unsigned long long* d_tape, uint64_t tape_capacity, unsigned long long* d_tape_offsets,
                       int32_t* d_doc_errors, void* d_ws, WalkResult* d_res, hipStream_t stream,
                       const Stage1Result* dev_count, const UnescapeResult* dev_strings, const uint32_t* d_soff,
                       bool index_from_zero = false, bool results_zeroed = false, const SingleDocTail& tail = SingleDocTail(),
                       const uint32_t* d_prepared = nullptr, bool layout_done = false, bool optimistic_only = false);
// layout_done: batch_layout_launch ran (the tapes ARE laid out when *d_prepared != 0, the result record and list headers are
// zeroed): nothing of that is queued again.  optimistic_only: only the kernels of the accepted path are queued.
// d_prepared (the fused batch pipeline): device flag, != 0 = batch.hip k_doc_prepare ran (the accepted plain pass) and left
// every document's DocMeta / predicted tape length in the walk workspace (walk_prepared): the tapes are laid out before the
// walk and written at their final addresses -- there a document that fails keeps its (unused) slot: tape_offsets[k + 1] -
// tape_offsets[k] is its PREDICTED length, the words are unspecified, doc_errors[k] says so.
struct WalkPrepared {
    DocMeta* metas;
    uint32_t* lens;
    unsigned long long* chunk_sums;
};
WalkPrepared walk_prepared(void* d_ws, uint64_t count, uint64_t n_docs);
// coop_walk.hip: the batch walkers -- k_tok_stream over tokens, then k_coop_walk (list mode) for the documents it declined
struct TokLaunch {
    const uint8_t* d_buf;
    const uint32_t* d_idx;
    const DocMeta* d_metas;
    uint64_t n_docs;
    const unsigned long long* d_doc_offsets;
    const unsigned long long* d_index_offsets;
    const uint32_t* d_doc_status;
    const unsigned long long* d_doc_str_ordinals;
    const uint32_t* d_soff;
    const uint8_t* d_sb;
    uint64_t string_base;
    int max_depth;
    unsigned long long* d_tape;     // *d_sel != 0 (or d_sel == nullptr): where the DocMeta offsets point
    unsigned long long* d_scratch;  // otherwise
    const uint32_t* d_sel;
    uint32_t* d_tape_lens;
    int32_t* d_doc_errors;
    uint32_t* d_list;
    const Stage1Result* dev_count;
    const UnescapeResult* dev_strings;
    WalkResult* d_res;
    void* d_deep_ws;
    bool header_zeroed = false;     // the literal list's 64-byte header is zero already (k_batch_layout)
};
hipError_t tok_walk_launch(const TokLaunch& t, hipStream_t stream);
// coop_walk.hip: the cooperative walker (a wave per document); d_soff = offset of every string's record in d_sb, by ordinal
hipError_t coop_walk_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, const uint32_t* d_idx,
                            const unsigned long long* d_index_offsets, const uint32_t* d_doc_status, const uint32_t* d_soff,
                            const uint8_t* d_sb, const unsigned long long* d_doc_str_ordinals, uint64_t string_base,
                            int max_depth, unsigned long long* d_scratch_tape, uint32_t* d_tape_lens, int32_t* d_doc_errors,
                            const Stage1Result* dev_count, const UnescapeResult* dev_strings, WalkResult* d_res,
                            hipStream_t stream, void* d_chunk_ws = nullptr, uint64_t count_bound = 0, void* d_deep_ws = nullptr,
                            unsigned long long* d_single_tape_offsets = nullptr, uint64_t tape_capacity = 0, bool header_zeroed = false,
                            const SingleDocTail& tail = SingleDocTail());
// nesting levels 64 .. 1023 of the wave-per-document walker live in global memory: bytes for a batch of n_docs documents
size_t coop_deep_workspace_bytes(uint64_t n_docs);
// workspace of the chunk-parallel path for one large document (coop_walk.hip)
size_t coop_chunk_workspace_bytes(uint64_t count_bound);
hipError_t single_doc_setup_launch(const Stage1Result* d_res, uint64_t len, unsigned long long* d_doc_offsets,
                                   unsigned long long* d_index_offsets, uint32_t* d_doc_status, unsigned long long* d_doc_str_offsets,
                                   hipStream_t stream, WalkResult* d_walk_result = nullptr, void* d_slow_header = nullptr);
// the 64 bytes at the head of the walk workspace's literal list (count + the chunk path's flags): walk_launch zeroes them unless
// the caller did (results_zeroed: single_doc_setup_launch on the latency path, together with *d_res)
void* walk_slow_header(void* d_ws, uint64_t count, uint64_t n_docs);
hipError_t batch_isolated_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, uint32_t* d_out,
                                 uint64_t out_cap, unsigned long long* d_index_offsets, uint32_t* d_doc_status,
                                 uint32_t* d_counts, Stage1Result* d_res, hipStream_t stream, uint64_t total_len, const uint32_t* d_skip = nullptr);
// the same in two halves (round 6): the VERDICTS only (doc_status[k], counts[k], their OR in batch_status_or(d_counts, n_docs)) --
// all the repair pass needs -- and the rest (index offsets + the indexes of the passing documents)
hipError_t batch_verdicts_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, uint32_t* d_doc_status,
                                 uint32_t* d_counts, hipStream_t stream, uint64_t total_len, const uint32_t* d_skip, uint8_t* d_copy);
// (d_copy != nullptr: the sanitized copy for the repair pass made on the way -- the documents' own bytes, a failing document blank)
hipError_t batch_indexes_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, uint32_t* d_out,
                                uint64_t out_cap, unsigned long long* d_index_offsets, uint32_t* d_doc_status, uint32_t* d_counts,
                                Stage1Result* d_res, hipStream_t stream, uint64_t total_len, const uint32_t* d_skip);
const uint32_t* batch_status_or(const uint32_t* d_counts, uint64_t n_docs);
// the optimistic plain pass of the fused batch pipeline (batch.hip)
hipError_t batch_plain_check_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, uint64_t total_len,
                                    uint32_t* d_flags, hipStream_t stream);
hipError_t batch_plain_accept_launch(const uint32_t* d_idx, const Stage1Result* d_res, const unsigned long long* d_doc_offsets,
                                     uint64_t n_docs, unsigned long long* d_index_offsets, uint32_t* d_doc_status, uint32_t* d_flags,
                                     hipStream_t stream, const Stage1Prefixes& hint = Stage1Prefixes(), bool split = true);
// masks.hip: the reference's six per-block masks (6 x u64 per block, len / 64 + 1 blocks)
size_t masks_workspace_bytes(uint64_t len);
hipError_t masks_launch(const uint8_t* d_buf, uint64_t len, unsigned long long* d_masks, void* d_ws, hipStream_t stream);
hipError_t transpose_selftest_launch(const uint32_t* d_words, uint32_t nblocks, uint32_t* d_mismatches,
                                     hipStream_t stream);

}  // namespace sjmi
