/*
 * sjmi.h -- C ABI of libsjmi.so: the MI355X (gfx950) stage-1 engine behind
 * simdjson-java's SimdJsonParser.parse(byte[], int).
 *
 * The reference has no FFI: its narrowest seam is the private method
 *     SimdJsonParser.stage1(byte[] buffer, int length)
 *         /root/reference/src/main/java/org/simdjson/SimdJsonParser.java:55-58
 * whose only observable effects are (a) BitIndexes.indexes[0..writeIdx] (ascending byte
 * offsets + a 0 sentinel, BitIndexes.java:14-41,82-96) and (b) one of three
 * JsonParsingExceptions (Utf8Validator.java:165-167, StructuralIndexer.java:297-302).
 * Every entry point below is what a Panama/JNI binding of that seam calls; INTEGRATION.md
 * shows the Java side.  Plain pointers and sizes only; no callbacks; no exceptions.
 *
 * Return value: 0 = the call ran (JSON-level verdicts are in `status`), < 0 = infrastructure
 * error (HIP failure, bad argument, capacity) -- never conflated with a JSON error.
 */
#ifndef SJMI_H
#define SJMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SJMI_PADDING 64u /* readable bytes required after len; SimdJsonParser.java:5 (PADDING) */

/* status bits; the Java shim throws the lowest set bit first (reference order of checks) */
#define SJMI_ST_UTF8 1u       /* "The input is not valid UTF-8"                      Utf8Validator.java:165-167 */
#define SJMI_ST_UNCLOSED 2u   /* "Unclosed string. A string is opened, but never closed." StructuralIndexer.java:297-299 */
#define SJMI_ST_UNESCAPED 4u  /* "Unescaped characters. Within strings, ..."         StructuralIndexer.java:300-302 */
#define SJMI_ST_CAPACITY 0x100u /* index_capacity < count+1 (the reference throws AIOOBE here) */
#define SJMI_ST_INTERNAL 0x200u /* engine fault (look-back timeout); results invalid */
#define SJMI_ST_REJECTED 0x800u /* sjmi_parse_batch_device_optimistic only: the batch is not one the optimistic pipeline can take (a document
                                 * fails stage 1, a separator is missing, the offsets do not cover the buffer): NO output of the call is
                                 * valid -- call sjmi_parse_batch_device_rejected (or sjmi_parse_batch_device), which decide every document on its own */
#define SJMI_ST_HALO 0x400u     /* sjmi_stage1_shard_device / sjmi_stream_push: a backslash run fills the whole left halo, so whether the
                                   byte behind it is escaped cannot be told from what is readable; results invalid -- give more halo */

/* return codes */
#define SJMI_OK 0
#define SJMI_ERR_HIP (-1)
#define SJMI_ERR_ARG (-2)
#define SJMI_ERR_CAPACITY (-3)
#define SJMI_ERR_INTERNAL (-4)
#define SJMI_ERR_NO_DEVICE (-5)

/* string / stage-2 error codes (numbering follows oracle/sj_oracle.h so that parity tests compare ints) */
#define SJMI_E_ESCAPE_UNEXPECTED 4       /* "Escaped unexpected character: "     CharacterUtils.java:74-83 */
#define SJMI_E_INVALID_UNICODE_ESCAPE 5  /* "Invalid unicode escape sequence."   StringParser.java:127-129 */
#define SJMI_E_LOW_SURROGATE_RESERVED 6  /* StringParser.java:53-55 */
#define SJMI_E_LOW_SURROGATE_NO_U 7      /* StringParser.java:113-115 */
#define SJMI_E_LOW_SURROGATE_RANGE 8     /* StringParser.java:118-122 */
#define SJMI_E_UTF8 1
#define SJMI_E_UNCLOSED_STRING 2
#define SJMI_E_UNESCAPED_CHARS 3
#define SJMI_E_NO_STRUCTURAL 9           /* JsonIterator.java:27-29; 10..21: grammar errors JsonIterator.java / TapeBuilder.java */
#define SJMI_E_UNCLOSED_OBJECT 10
#define SJMI_E_UNCLOSED_ARRAY 11
#define SJMI_E_OBJECT_NO_KEY 12
#define SJMI_E_MISSING_COLON 13
#define SJMI_E_KEY_MISSING 14
#define SJMI_E_NO_COMMA_OBJECT 15
#define SJMI_E_NO_COMMA_ARRAY 16
#define SJMI_E_TRAILING_CONTENT 17
#define SJMI_E_UNRECOGNIZED_PRIMITIVE 18
#define SJMI_E_INVALID_TRUE 19
#define SJMI_E_INVALID_FALSE 20
#define SJMI_E_INVALID_NULL 21
#define SJMI_E_NUM_MINUS 22              /* 22..27: NumberParser.java:34-72, ExponentParser.java:27-29 */
#define SJMI_E_NUM_LEADING_ZERO 23
#define SJMI_E_NUM_DECIMAL_POINT 24
#define SJMI_E_NUM_EXPONENT 25
#define SJMI_E_NUM_FOLLOWED 26
#define SJMI_E_NUM_LONG_RANGE 27
#define SJMI_E_DEPTH 28                  /* the reference throws ArrayIndexOutOfBoundsException (JsonIterator.java:69-70) */
#define SJMI_E_CAPACITY 29
#define SJMI_E_INTERNAL 100              /* engine invariant violated (never for status == 0 input) */

typedef struct sjmi_ctx sjmi_ctx;

/* device-side result record of one stage-1 call */
typedef struct sjmi_stage1_result {
    uint64_t count;   /* BitIndexes.writeIdx */
    uint32_t status;  /* SJMI_ST_* */
    uint32_t reserved;
} sjmi_stage1_result;

/* One context per host thread (SimdJsonParser is not thread-safe either: SimdJsonParser.java:9-13
 * shares one BitIndexes).  capacity_bytes = largest document, as the reference's ctor argument
 * `capacity` (SimdJsonParser.java:19-26). Owns a HIP stream, device staging buffers and the
 * tile-state workspace. Fails with SJMI_ERR_NO_DEVICE when no gfx950 device is usable: there is
 * no CPU fallback. */
int sjmi_create(sjmi_ctx** out, int device, uint64_t capacity_bytes);
void sjmi_destroy(sjmi_ctx* ctx);
const char* sjmi_last_error(const sjmi_ctx* ctx);
const char* sjmi_version(void);

/* Replaces SimdJsonParser.stage1 (+ padIfNeeded) for HOST buffers: copies buf[0,len) to the
 * device (padding handled internally: bytes >= len are never read from `buf`), runs the fused
 * kernel, copies indexes[0..count] (sentinel included) and the verdict back.
 * indexes must hold index_capacity entries; needs index_capacity >= count+1.  Like BitIndexes.write's speculative
 * stores (BitIndexes.java:14-41: "garbage may follow the valid prefix"), the call may write ANY entry of
 * indexes[0, index_capacity) -- small documents are downloaded by their bound, not by their count -- so a caller
 * must not keep other data inside that range; only indexes[0..count] are meaningful.  The same holds for
 * string_buffer[0, string_capacity) of sjmi_stage1_unescape / sjmi_unescape (StringParser's over-copy, :33-34). */
int sjmi_stage1(sjmi_ctx* ctx, const uint8_t* buf, uint64_t len, uint32_t* indexes, uint64_t index_capacity,
                uint64_t* count, uint32_t* status);

/* Same path on DEVICE-resident data (roofline runs, pipelines that already hold the document in
 * HBM). d_buf must be 16-byte aligned with SJMI_PADDING readable bytes after len (contents
 * ignored); d_indexes (16-byte aligned) holds index_capacity u32; d_result is a device sjmi_stage1_result.
 * Asynchronous on `stream` (a hipStream_t, may be NULL = the context's stream). len < 2^32. */
int sjmi_stage1_device(sjmi_ctx* ctx, const void* d_buf, uint64_t len, void* d_indexes, uint64_t index_capacity,
                       void* d_result, void* stream);

/* Bit-mask parity entry point: the six 64-bit masks one iteration of the reference's stage-1 loop holds for every 64-byte
 * block (StructuralIndexer.java:210-252), masks[6*b + {0..5}] = {escaped, quote, inString, op, whitespace, structurals}
 * of block b, for all len / 64 + 1 blocks (the reference always processes one space-padded tail block, :255-294,305-309).
 * They are internal locals of the reference (never asserted by its tests); north_star asks for them bit-exact, so the
 * engine reconstructs them from its own per-block formulation + the resolved in-string parity (csrc/masks.hip).
 * Host form: copies buf[0,len) in, masks out; needs mask_capacity_blocks >= len / 64 + 1; *n_blocks = blocks written.
 * Device form: d_buf as for sjmi_stage1_device, d_masks holds 6 * (len / 64 + 1) uint64; asynchronous on `stream`. */
int sjmi_stage1_masks(sjmi_ctx* ctx, const uint8_t* buf, uint64_t len, uint64_t* masks, uint64_t mask_capacity_blocks,
                      uint64_t* n_blocks);
int sjmi_stage1_masks_device(sjmi_ctx* ctx, const void* d_buf, uint64_t len, void* d_masks, uint64_t mask_capacity_blocks,
                             void* stream);

/* ONE SHARD of a document that is split over several GPUs, or ONE CHUNK of a document stream (SURVEY.md 8(e) row 2, 8(f)
 * rank 4; the reference has no counterpart: one byte[] per parse).  d_buf points at the shard's first byte (16-byte
 * aligned), with halo_bytes (a multiple of 64, >= 64 unless the shard starts the document) of the document's preceding
 * bytes readable right in front of it: everything stage 1 carries from block to block except the in-string parity is
 * re-derived from those bytes (escape run, previous scalar, UTF-8 continuation -- a backslash run that fills the whole
 * halo is the only thing it cannot see through: the call then reports SJMI_ST_HALO and the caller repeats it with a larger
 * halo).  is_last = 0: the shard ends on a 64-byte boundary (len % 64 == 0) and has
 * no tail block -- what straddles the boundary is validated by the next shard.  entry_parity = 1: the shard starts
 * inside a string.  Indexes are relative to d_buf; result.status bit SJMI_ST_UNCLOSED = the parity AFTER the shard.
 * Protocol (sharding.py DocumentSplit): run every shard with entry_parity 0; exchange the parities (1 bit per shard: the
 * first collective); a shard whose true entry parity is 1 runs again with entry_parity 1; exchange the counts. */
int sjmi_stage1_shard_device(sjmi_ctx* ctx, const void* d_buf, uint64_t len, uint64_t halo_bytes, int is_last, int entry_parity,
                             void* d_indexes, uint64_t index_capacity, void* d_result, void* stream);
/* the same with one more fact: halo_from_document_start != 0 = the halo begins at the document's first byte (nothing is in
 * front of it, so a backslash run that fills it is complete and SJMI_ST_HALO is never reported) */
int sjmi_stage1_shard_device2(sjmi_ctx* ctx, const void* d_buf, uint64_t len, uint64_t halo_bytes, int halo_from_document_start,
                              int is_last, int entry_parity, void* d_indexes, uint64_t index_capacity, void* d_result, void* stream);

/* The same as a PRODUCT feature, with the protocol's state kept in C (no per-chunk allocation, nothing for a binding to
 * re-implement): ONE document as a stream of chunks on one GPU -- a document of any length (the 4 GiB - 1 of uint32 indexes
 * and the 2 GiB of a Java byte[] apply to a chunk, not to the stream).  sjmi_stream_open sizes the device buffers once
 * (max_chunk_bytes per push; halo_bytes, a multiple of 64, 0 = 64: how much of the stream in front of a chunk stage 1 looks
 * at; up to 4 KiB of the stream are kept on the device, and a chunk whose halo proves too short -- SJMI_ST_HALO -- is
 * repeated with more of them by the call itself).  sjmi_stream_push: chunk = host bytes, every chunk but the last a non-zero
 * multiple of 64 long; indexes[0..*count] = the chunk's structurals RELATIVE to the chunk (+ the sentinel), *base = the
 * chunk's offset in the stream; *status = the document's verdict so far (SJMI_ST_UNCLOSED only behind the last chunk).
 * The in-string parity is carried from chunk to chunk, so every chunk is scanned exactly once. */
typedef struct sjmi_stream sjmi_stream;
int sjmi_stream_open(sjmi_ctx* ctx, uint64_t max_chunk_bytes, uint64_t halo_bytes, sjmi_stream** out);
int sjmi_stream_push(sjmi_stream* s, const uint8_t* chunk, uint64_t len, int is_last, uint32_t* indexes, uint64_t index_capacity,
                     uint64_t* count, uint64_t* base, uint32_t* status);
void sjmi_stream_close(sjmi_stream* s);
/* ... and one rank's shard of a document split over several GPUs (arguments as sjmi_stage1_shard_device2; everything stays
 * on the device).  sjmi_split_scan: the shard as if it began outside a string; *flips_parity = does it flip the in-string
 * parity -- all_gather that bit (the first collective).  sjmi_split_resolve(entry_parity = XOR of the flips of the ranks in
 * front): scans again ONLY if the entry parity is 1; -> the shard's count / status bits / parity behind it -- all_gather
 * those (the second collective): global index offsets = prefix sums of the counts, the document is unclosed iff the LAST
 * shard's parity_after is 1.  (A shard entered inside a string costs a second scan: both polarities in one pass would double
 * the index stores of the hot kernel for everybody) */
typedef struct sjmi_split sjmi_split;
int sjmi_split_open(sjmi_ctx* ctx, const void* d_shard, uint64_t len, uint64_t halo_bytes, int halo_from_document_start, int is_last,
                    void* d_indexes, uint64_t index_capacity, sjmi_split** out);
int sjmi_split_scan(sjmi_split* s, void* stream, int* flips_parity, uint32_t* status);
int sjmi_split_resolve(sjmi_split* s, int entry_parity, void* stream, uint64_t* count, uint32_t* status, int* parity_after);
void sjmi_split_close(sjmi_split* s);

/* device-side result record of one unescape call */
typedef struct sjmi_unescape_result {
    uint64_t total_bytes;      /* bytes of [be32 length][unescaped bytes] records written */
    uint64_t first_error_inv;  /* 0 = every string is fine; else ~((p << 8) | SJMI_E_* code) of the first failing string, p = byte
                                  offset of the offending escape in the document (sjmi_unescape_device, sjmi_parse_*), or the
                                  string's position in indexes[] (sjmi_unescape_batch_device) */
    uint32_t flags;            /* bit 0: string_buffer capacity exceeded; bit 1: more strings than the record table holds
                                * (index_capacity - 1 + 64 entries: raise index_capacity; no tape is built); bits 2-3: engine fault (results invalid) */
    uint32_t n_strings;        /* string literals of the document (sjmi_unescape_device) */
} sjmi_unescape_result;

/* Batched replacement of the per-string StringParser.parseString calls (StringParser.java:18-68) that
 * the reference's stage 2 makes for every string (TapeBuilder.java:174-177): for every string literal of the
 * document, in order, appends [be32 length][unescaped UTF-8 bytes] to string_buffer, so that
 * record k starts at sum_{j<k}(4 + len_j) -- exactly the STRING tape payloads of a valid document, and byte for byte
 * the reference's stringBuffer.  One streaming pass over the document (csrc/strings.hip); the index array is not
 * read (it is part of the signature because the reference's loop is over the '"' structurals: every opening quote of a
 * document that passes stage 2 is one).  It starts from the in-string parity of every 64-byte block, which the last
 * sjmi_stage1*_device launch of this context over the same (d_buf, len) left on the device -- the bytes must not have
 * changed since; without one (another context indexed the document) the call derives them itself with one more pass.
 * A string StringParser would throw on gets the record header FF FF FF <SJMI_E_* code> (what follows it up to the next
 * record is unspecified), all other records are still exact and in place; the first such error is also reported in
 * d_result.  Requires a stage-1 status of 0 for this document (closed strings); string_capacity < 4 GiB.
 * Device-resident form, asynchronous on `stream`; d_buf 16-byte aligned; d_result is a device sjmi_unescape_result. */
int sjmi_unescape_device(sjmi_ctx* ctx, const void* d_buf, uint64_t len, const void* d_indexes, uint64_t count,
                         void* d_string_buffer, uint64_t string_capacity, void* d_result, void* stream);

/* Host form: unescapes the strings of the document given to the LAST sjmi_stage1 call on this context
 * (its bytes and indexes are still resident on the device) into string_buffer[0, *total_bytes).
 * *first_error_index = position in indexes[] of the first failing string or UINT64_MAX; *first_error_code = SJMI_E_*. */
int sjmi_unescape(sjmi_ctx* ctx, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* total_bytes,
                  uint64_t* first_error_index, uint32_t* first_error_code);

/* SimdJsonParser.parse's GPU part in ONE call: sjmi_stage1 followed by sjmi_unescape of the same document, with two
 * host synchronisations instead of four (the unescape kernels read the structural count from the stage-1 result on
 * the device).  Outputs as in the two separate calls; the string outputs are meaningful only if *status == 0. */
int sjmi_stage1_unescape(sjmi_ctx* ctx, const uint8_t* buf, uint64_t len, uint32_t* indexes, uint64_t index_capacity,
                         uint64_t* count, uint32_t* status, uint8_t* string_buffer, uint64_t string_capacity,
                         uint64_t* total_bytes, uint64_t* first_error_index, uint32_t* first_error_code);

/* ---- batched documents ---------------------------------------------------------------------------------
 * n_docs documents packed in one buffer, document k at [doc_offsets[k], doc_offsets[k+1]), each followed by at
 * least one JSON whitespace byte inside its range (NDJSON style; doc_offsets[n_docs] = total length).
 * One stage-1 launch indexes the whole buffer; index_offsets[k] (n_docs+1 entries) then delimits document k's
 * indexes, which stay ABSOLUTE byte offsets (subtract doc_offsets[k] for the reference's per-document values).
 * status is the verdict of the whole batch (exact whenever every document passes stage 1; see batch.hip).
 * Device form, asynchronous on `stream`. */
int sjmi_stage1_batch_device(sjmi_ctx* ctx, const void* d_buf, uint64_t total_len, const void* d_doc_offsets,
                             uint64_t n_docs, void* d_indexes, uint64_t index_capacity, void* d_index_offsets,
                             void* d_result, void* stream);
/* Host form: copies the batch in, returns indexes[0..count], index_offsets[0..n_docs], count and status. */
int sjmi_stage1_batch(sjmi_ctx* ctx, const uint8_t* buf, uint64_t total_len, const uint64_t* doc_offsets, uint64_t n_docs,
                      uint32_t* indexes, uint64_t index_capacity, uint64_t* index_offsets, uint64_t* count,
                      uint32_t* status);

/* ISOLATED batch: exact per document whatever the other documents contain.  One wave per document, every carry
 * (in-string parity, escape run, previous scalar, UTF-8 continuation) starts from zero at the document's first byte.
 * doc_status[k] = the SJMI_ST_* bits document k would get from sjmi_stage1 on its own; a document with a non-zero
 * status contributes NO indexes (index_offsets[k+1] == index_offsets[k], as the reference throws before its stage 2);
 * the others get exactly their own indexes (absolute byte offsets).  result.status = OR of the document statuses
 * (| SJMI_ST_CAPACITY), result.count = indexes written.  The plain batch above cannot attribute an error to a document,
 * and two documents with an unclosed string each cancel in its verdict.  This call tries it first all the same -- ONE
 * plain launch over the packed buffer -- and accepts it ON THE DEVICE when that is provably what the per-document passes
 * would give: every document ends in a control-character separator ('\n', '\r', '\t': an open string would turn it into
 * an unescaped character), doc_offsets[0] == 0 and doc_offsets[n_docs] == total_len, and the global verdict is clean.
 * The per-document passes are queued behind it and leave at once when it was accepted, so nothing comes back to the host.
 * Device form, asynchronous on `stream`; d_doc_status: n_docs uint32; d_buf and d_indexes 16-byte aligned for the
 * plain pass (else only the per-document passes run). */
int sjmi_stage1_batch_isolated_device(sjmi_ctx* ctx, const void* d_buf, uint64_t total_len, const void* d_doc_offsets,
                                      uint64_t n_docs, void* d_indexes, uint64_t index_capacity, void* d_index_offsets,
                                      void* d_doc_status, void* d_result, void* stream);
/* Host form. */
int sjmi_stage1_batch_isolated(sjmi_ctx* ctx, const uint8_t* buf, uint64_t total_len, const uint64_t* doc_offsets,
                               uint64_t n_docs, uint32_t* indexes, uint64_t index_capacity, uint64_t* index_offsets,
                               uint32_t* doc_status, uint64_t* count, uint32_t* status);

/* sjmi_unescape for the batch given to the LAST sjmi_stage1_batch / sjmi_stage1_batch_isolated call on this context:
 * one string buffer for the whole batch (records in structural order), plus doc_string_offsets[k] (n_docs + 1 entries)
 * = offset of document k's first record, so that the stage 2 of every document can start at its own offset (documents
 * can then be walked by parallel host threads). */
int sjmi_unescape_batch(sjmi_ctx* ctx, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* doc_string_offsets,
                        uint64_t* total_bytes, uint64_t* first_error_index, uint32_t* first_error_code);

/* Device-resident form of the same, asynchronous on `stream`: indexes / doc_offsets / index_offsets as given to and
 * produced by sjmi_stage1_batch[_isolated]_device (device pointers, n_docs + 1 uint64 entries each).  Unlike
 * sjmi_unescape_device it knows the document boundaries: in isolated mode a document that failed stage 1 has no
 * structurals, so the last string of the document in front of it must end at that document's end, not at the next
 * structural of the batch.  d_doc_string_offsets may be NULL. */
int sjmi_unescape_batch_device(sjmi_ctx* ctx, const void* d_buf, uint64_t total_len, const void* d_indexes, uint64_t count,
                               const void* d_doc_offsets, const void* d_index_offsets, uint64_t n_docs,
                               void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, void* d_result,
                               void* stream);

/* ---- stage 2 on the GPU for batches ---------------------------------------------------------
 * JsonIterator.walkDocument + TapeBuilder (JsonIterator.java:26-200, TapeBuilder.java:41-217) for every document of a
 * batch, one GPU wave per document (csrc/coop_walk.hip), from the outputs of sjmi_stage1_batch_isolated_device and
 * sjmi_unescape_batch_device (all device pointers; the string pass also leaves a table of record offsets by string
 * ordinal on the context, so both calls must come from this context, for the same d_indexes).  Produces the tapes back to
 * back in d_tape (Tape.java:5-47 word layout; document k: words [tape_offsets[k], tape_offsets[k+1]), its container words
 * relative to its own start, STRING payloads = string_base + offset of the record in d_string_buffer) and doc_errors[k]
 * (int32): 0, or the SJMI_E_* code of the document's first error (stage-1 verdicts and malformed escapes included) with an
 * empty tape, or SJMI_WALK_NEEDS_HOST with an empty tape.  The device decides everything the reference's defaults admit:
 * nesting up to 1024 open containers (SimdJsonParser.java:7) and every number literal, including floating-point literals
 * of more than 19 significant digits at a rounding boundary (exact big-integer comparison with the midpoint, DoubleParser.
 * java:205-330).  Handed back only: max_depth > 1024 and a document that uses it, more than 65,536 such boundary literals
 * in one launch, a single document of more than 2^31 - 256 structurals.  d_result: sjmi_walk_result, bit 0 of flags =
 * tape_capacity exceeded (offsets valid, tapes not written).  d_buf needs 64 readable bytes after the batch (the
 * reference's padding, SimdJsonParser.java:42-48).  Asynchronous on `stream`. */
#define SJMI_WALK_NEEDS_HOST (-1)
typedef struct sjmi_walk_result {
    uint64_t tape_words;       /* total tape words = tape_offsets[n_docs] */
    uint64_t host_documents;   /* documents left to the host walker */
    uint64_t failed_documents; /* documents with a JSON error */
    uint32_t flags;
    uint32_t reserved;
} sjmi_walk_result;
int sjmi_walk_batch_device(sjmi_ctx* ctx, const void* d_buf, const void* d_doc_offsets, uint64_t n_docs, const void* d_indexes,
                           uint64_t count, const void* d_index_offsets, const void* d_doc_status,
                           const void* d_string_buffer, const void* d_doc_string_offsets, uint64_t string_base,
                           int max_depth, void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors,
                           void* d_result, void* stream);

/* The whole batched parse, device-resident, in ONE call and without a host round trip between the stages:
 * sjmi_stage1_batch_isolated_device -> sjmi_unescape_batch_device -> sjmi_walk_batch_device (string_base 0), the
 * structural count handed from stage to stage on the device.  Outputs as documented for the three calls; index_capacity
 * bounds the structural count (+ sentinel) and sizes the workspaces.  d_result: a device sjmi_batch_result.  This is
 * what one rank of the sharded multi-GPU batch runs per step (sharding.py); asynchronous on `stream`.
 * Stage 1 is first tried as ONE plain launch over the packed batch and accepted on the device when every document ends in
 * a control-character separator ('\n', '\r', '\t'), the documents cover the buffer exactly (doc_offsets[0] == 0 and
 * doc_offsets[n_docs] == total_len: bytes outside the documents would feed state into them) and the global verdict is
 * clean -- then it is exactly what the per-document passes give; otherwise those run (queued behind it, they leave at once
 * when it was accepted).  (Round 5: the separators are checked by k_doc_prepare, which visits every boundary anyway, so the
 * verdict falls behind the string pass: a rejected batch pays one string pass over the raw batch before its own begins.)
 * Tapes: when the plain pass was accepted the tapes are laid out BEFORE the documents are walked (a document's tape length
 * is a function of its structurals' first bytes: TapeBuilder.java:41-48,191-208, Tape.java:33-43) and written at their final
 * addresses.  A document that then fails stage 2 (doc_errors[k] != 0) keeps its slot: tape_offsets[k + 1] - tape_offsets[k] is
 * its PREDICTED length and the words there are unspecified; walk.tape_words counts the slots.  Every well-formed document's
 * tape is where and what it always was, and a batch without failing documents is packed exactly as before.  (When the plain
 * pass was rejected -- some document fails stage 1, a separator is missing -- the REPAIR stage described at
 * sjmi_parse_batch_device_rejected takes the batch and reports like an accepted one; only behind a batch it cannot take either are
 * the tapes packed behind the walk and a failing document's range empty, as with sjmi_walk_batch_device.) */
typedef struct sjmi_batch_result {
    sjmi_stage1_result stage1;
    sjmi_unescape_result strings;
    sjmi_walk_result walk;
} sjmi_batch_result;
int sjmi_parse_batch_device(sjmi_ctx* ctx, const void* d_buf, uint64_t total_len, const void* d_doc_offsets, uint64_t n_docs,
                            void* d_indexes, uint64_t index_capacity, void* d_index_offsets, void* d_doc_status,
                            void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, int max_depth,
                            void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors, void* d_result,
                            void* stream);
/* The same call with ONLY the optimistic pipeline queued (round 5): k_stage1_batch -> k_strings -> k_doc_prepare -> tape layout ->
 * the token walker (+ the exact walker and the boundary-literal kernel for the documents it lists) -- eight queue entries instead
 * of thirty, which is what a small batch (one rank's share of a strong-scaled run) is bounded by.  Whether the batch qualified is
 * decided on the device like before and reported in result.stage1.status: SJMI_ST_REJECTED set = NOTHING this call wrote is
 * valid; the caller then makes the exact call above (same arguments), off its hot path.  Without that bit the outputs are
 * exactly those of sjmi_parse_batch_device.  NDJSON whose documents all pass stage 1 is never rejected. */
int sjmi_parse_batch_device_optimistic(sjmi_ctx* ctx, const void* d_buf, uint64_t total_len, const void* d_doc_offsets, uint64_t n_docs,
                                       void* d_indexes, uint64_t index_capacity, void* d_index_offsets, void* d_doc_status,
                                       void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, int max_depth,
                                       void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors, void* d_result,
                                       void* stream);

/* The call to make after sjmi_parse_batch_device_optimistic came back with SJMI_ST_REJECTED (round 6; same arguments, same outputs as
 * sjmi_parse_batch_device -- which is this call behind one more attempt at the plain pass).  The reference isolates a malformed
 * document for free -- one JsonParsingException per SimdJsonParser.parse, SimdJsonParser.java:35-40, the neighbours never see it --
 * so a batch with ONE bad document in a million must not cost a multiple of a clean one.  On the device, no host round trip:
 * every document gets its own stage-1 verdict (16 lanes per document, all carries from zero), the documents that fail are
 * blanked in a sanitized copy, and the optimistic pipeline runs over that copy (REPAIR) -- exact, because every surviving
 * document begins and ends outside a string; documents that are not separated at all are taken too as long as no scalar can run
 * on across a boundary (the last byte of every document is whitespace, an operator or a quote).  A batch that fails that rule
 * (or whose document offsets do not cover the buffer) comes back with SJMI_ST_REJECTED ONCE MORE: sjmi_parse_batch_device then
 * serves it with the per-document index passes -- three levels, each cheaper than the next, the last one takes anything.  A repaired batch reports like an accepted one: a document that failed stage 1
 * has doc_status[k] / doc_errors[k] set, no structurals and a two-word tape slot with unspecified contents; stage1.status is the
 * OR of the documents' verdicts.  (SJMI_BATCH_REPAIR=0 switches the repair stage off.) */
int sjmi_parse_batch_device_rejected(sjmi_ctx* ctx, const void* d_buf, uint64_t total_len, const void* d_doc_offsets, uint64_t n_docs,
                                     void* d_indexes, uint64_t index_capacity, void* d_index_offsets, void* d_doc_status,
                                     void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, int max_depth,
                                     void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors, void* d_result,
                                     void* stream);

/* One document, ALL stages on the GPU: stage 1, string records and the cooperative walker (csrc/coop_walk.hip: JsonIterator.
 * walkDocument + TapeBuilder as scans, JsonIterator.java:26-200, TapeBuilder.java:41-217); only the tape (Tape.java:5-47 word
 * layout, tape[0] = root) and the string buffer come back -- the structural indexes stay on the device.  *error = 0, or the
 * document's SJMI_E_* code (stage-1 verdicts included; no tape), or SJMI_WALK_NEEDS_HOST (see sjmi_walk_batch_device: not
 * for anything the reference's default limits admit; sjmi_parser_parse walks such a document on the host). */
int sjmi_parse_document(sjmi_ctx* ctx, const uint8_t* buf, uint64_t len, int max_depth, uint64_t* tape, uint64_t tape_capacity,
                        uint64_t* tape_len, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* strings_len,
                        int32_t* error, uint32_t* stage1_status);

/* ---- whole parse: SimdJsonParser.parse(byte[], int) (SimdJsonParser.java:35-40) ----------------------------
 * GPU stage 1 + GPU string unescape + the host stage-2 tree builder (C++ mirror of JsonIterator / TapeBuilder /
 * Tape: simdjson-java_amd/csrc/host/simdjson_parser.h).  The tape (Tape.java:5-47 word layout) and string
 * buffer views stay valid until the next parse on this parser (like the reference's JsonValue).
 * Returns 0, or > 0 = SJMI_E_* JSON error (message: sjmi_parser_last_message, exact reference text), or < 0. */
typedef struct sjmi_parser sjmi_parser;
int sjmi_parser_create(sjmi_parser** out, int capacity, int max_depth, int device);
void sjmi_parser_destroy(sjmi_parser* p);
int sjmi_parser_parse(sjmi_parser* p, const uint8_t* buf, uint64_t len, const uint64_t** tape, uint64_t* tape_len,
                      const uint8_t** strings, uint64_t* strings_len, uint64_t* error_pos);
const char* sjmi_parser_last_message(const sjmi_parser* p);
/* Where stage 2 of sjmi_parser_parse runs: 0 = the host walker over the GPU-made indexes and string records, 1 = the
 * cooperative GPU walker (sjmi_parse_document): the tape comes from the device, and only a document that fails or is
 * handed back is walked again on the host (for the exact exception); < 0 (the default) = by size: the GPU walker for
 * documents of 128 KiB and more (twitter.json 0.148 vs 0.188 ms, 2.4 x faster at 1 MiB, 6 x from 16 MiB on), the host
 * walker below.  Identical results. */
int sjmi_parser_set_gpu_walk(sjmi_parser* p, int on);
/* Batched parse (BASELINE.json configs[3]/[4]): the batch goes through the GPU (isolated stage 1 + string records)
 * as a pipeline of sub-batches on two streams, the host stage 2 of the documents runs on a pool of threads
 * (SJMI_PARSE_THREADS, default min(64, cores)) while the GPU works on the next sub-batch.  Document k's tape is
 * tape[tape_offsets[k] .. tape_offsets[k+1]) (container words relative to its own start, STRING payloads = offsets
 * into the shared `strings` buffer, which has unused gaps between sub-batches) and errors[k] is 0 or document k's
 * own SJMI_E_* error -- stage-1 errors included (isolated batch mode: one broken document never affects another).
 * The call itself is made from one thread; the parser is not thread-safe (like the reference's). */
int sjmi_parser_parse_batch(sjmi_parser* p, const uint8_t* buf, uint64_t total_len, const uint64_t* doc_offsets,
                            uint64_t n_docs, const uint64_t** tape, const uint64_t** tape_offsets,
                            const uint8_t** strings, uint64_t* strings_len, const int32_t** errors);

/* ---- the on-demand front end --------------------------------------------------------------
 * OnDemandJsonIterator (OnDemandJsonIterator.java:7-675), the cursor SchemaBasedJsonIterator.java:29-132 drives with one
 * call per field of the schema: it walks the structural indexes of stage 1 and parses only the values it is asked for.
 * sjmi_parser_ondemand_init = SimdJsonParser.parse(buffer, len, Class) up to and including iterator.init (:31-41): pad and
 * GPU stage 1.  (`reserved` must be 0.  Rounds 2-4 built a GPU skip table here -- up[] / match[] per structural, skipChild as
 * a lookup -- that never paid: 0.256 against 0.159 ms per parse-and-select of twitter.json, because the scan it replaced costs
 * ~1 ns per structural on the host and the table two kernels plus 8 bytes per structural over PCIe.  Removed in round 5.)
 * The calls below mirror
 * the iterator's methods one to one (root != 0: the Root form; nullable == 0: the NonNull form; *is_null: the method
 * returned null); each returns 0, or > 0 = the SJMI_E_* code of the JsonParsingException the reference throws there
 * (exact text: sjmi_parser_last_message), or < 0.  Every getter of the class is here. */
#define SJMI_E_OD_NOT_ENOUGH_CLOSE 40    /* "Not enough close braces."                                   :80 */
#define SJMI_E_OD_EXPECTED_CHAR 41       /* "Expected 'x' but got: 'y'."                                 :662 */
#define SJMI_E_OD_EXPECTED_CHAR_END 42   /* "Expected 'x' but reached end of buffer."                    :660 */
#define SJMI_E_OD_BOOLEAN 43             /* "Unrecognized boolean value. Expected: 'true' or 'false'."   :88,:152 */
#define SJMI_E_OD_BOOLEAN_OR_NULL 44     /* "... Expected: 'true', 'false' or 'null'."                   :104,:167 */
#define SJMI_E_OD_STRING_OR_NULL 45      /* "Invalid value starting at N. Expected either string or 'null'."  :455,:470 */
#define SJMI_E_OD_FLOAT_PART_MISSING 46  /* "Invalid floating-point number. Fraction or exponent part is missing."  NumberParser.java:303 */
#define SJMI_E_OD_BYTE_RANGE 47          /* "Number value is out of byte range ([-128, 127])."            NumberParser.java:97 */
#define SJMI_E_OD_SHORT_RANGE 48         /* "... out of short range ([-32768, 32767])."                   :136 */
#define SJMI_E_OD_INT_RANGE 49           /* "... out of int range ([-2147483648, 2147483647])."           :175 */
#define SJMI_E_OD_STRING_EXPECTED 50     /* "Invalid value starting at N. Expected string."               :480,:505 */
#define SJMI_E_OD_CHAR_CODE_POINT 51     /* "Invalid code point. Should be within the range U+0000–U+D777 or U+E000–U+FFFF."  StringParser.java:78 */
#define SJMI_E_OD_CHAR_NOT_16BIT 52      /* "String cannot be deserialized to a char. Expected a single 16-bit code unit character."  :104 */
#define SJMI_E_OD_CHAR_NOT_SINGLE 53     /* "... Expected a single-character string."                    :107 */
#define SJMI_OD_EMPTY 0                  /* IteratorResult :672-674 */
#define SJMI_OD_NULL 1
#define SJMI_OD_NOT_EMPTY 2
int sjmi_parser_ondemand_init(sjmi_parser* p, const uint8_t* buf, uint64_t len, int reserved);
int sjmi_od_skip_child(sjmi_parser* p, int parent_depth);             /* skipChild(parentDepth) :47-81; < 0: skipChild() :43-45 */
int sjmi_od_get_boolean(sjmi_parser* p, int root, int nullable, int* is_null, int* value);     /* :83-109,:147-171 */
int sjmi_od_get_long(sjmi_parser* p, int root, int nullable, int* is_null, int64_t* value);    /* :321-358 */
/* the Byte :204-241 / Short :243-280 / Int :282-319 getters: bits = 8, 16, 32 (64 = sjmi_od_get_long) */
int sjmi_od_get_integral(sjmi_parser* p, int bits, int root, int nullable, int* is_null, int64_t* value);
int sjmi_od_get_double(sjmi_parser* p, int root, int nullable, int* is_null, double* value);   /* :383-428 */
int sjmi_od_get_float(sjmi_parser* p, int root, int nullable, int* is_null, float* value);     /* :360-381,:430-444 */
int sjmi_od_get_char(sjmi_parser* p, int root, int nullable, int* is_null, uint16_t* utf16_unit);  /* :474-520 */
/* getRootString / getString :446-472, getFieldName :646-652: the unescaped bytes, valid until the next of these calls */
int sjmi_od_get_string(sjmi_parser* p, int root, int* is_null, const uint8_t** bytes, uint64_t* len);
int sjmi_od_get_field_name(sjmi_parser* p, const uint8_t** bytes, uint64_t* len);
int sjmi_od_start_array(sjmi_parser* p, int root, int* result);       /* startIterating[Root]Array :522-566 -> SJMI_OD_* */
int sjmi_od_next_array_element(sjmi_parser* p, int* has_next);        /* :568-579 */
int sjmi_od_start_object(sjmi_parser* p, int root, int* result);      /* startIterating[Root]Object :581-623 */
int sjmi_od_next_object_field(sjmi_parser* p, int* has_next);         /* :625-636 */
int sjmi_od_move_to_field_value(sjmi_parser* p);                      /* :638-644 */
int sjmi_od_assert_no_more_values(sjmi_parser* p);                    /* :666-670 */
int sjmi_od_depth(const sjmi_parser* p);                              /* getDepth :654-656 */
#define SJMI_OD_END 256
int sjmi_od_peek(const sjmi_parser* p);   /* (extension for schema-less drivers) the byte of the next structural, SJMI_OD_END behind the last */

/* ---- JsonValue (JsonValue.java:18-221) over the C ABI: the DOM view of the last parse -------------------------------
 * A value is (document, tape index), like the reference's (tape, tapeIdx, stringBuffer) tuple (JsonValue.java:20-30);
 * handles stay valid until the next parse / parse_batch on the parser.  Every accessor runs the C++ mirror class
 * org_simdjson::JsonValue (csrc/host/simdjson_parser.h).  Return 0 = ok, 1 = "null" / no more elements (where Java
 * returns null or hasNext() is false), < 0 = wrong type or bad handle (where Java would throw). */
typedef struct sjmi_value {
    uint64_t doc;      /* UINT64_MAX = the document of the last sjmi_parser_parse; else document index of the last batch */
    uint64_t tape_idx;
} sjmi_value;
int sjmi_parser_root(const sjmi_parser* p, sjmi_value* out);                       /* TapeBuilder.createJsonValue :215-217 */
int sjmi_parser_batch_root(const sjmi_parser* p, uint64_t doc, sjmi_value* out);   /* > 0: that document's SJMI_E_* error */
int sjmi_value_type(const sjmi_parser* p, const sjmi_value* v);  /* '[' '{' '"' 'l' 'd' 't' 'f' 'n' (isArray() ... isString(), :32-59) */
int sjmi_value_as_long(const sjmi_parser* p, const sjmi_value* v, int64_t* out);   /* asLong :69-71 */
int sjmi_value_as_double(const sjmi_parser* p, const sjmi_value* v, double* out);  /* asDouble :73-75 */
int sjmi_value_as_boolean(const sjmi_parser* p, const sjmi_value* v, int* out);    /* asBoolean :77-79 */
/* asString :81-89: copies the UTF-8 bytes (no terminator); *len = their number even when dst_capacity is too small */
int sjmi_value_as_string(const sjmi_parser* p, const sjmi_value* v, uint8_t* dst, uint64_t dst_capacity, uint64_t* len);
int sjmi_value_get(const sjmi_parser* p, const sjmi_value* v, const uint8_t* name, uint64_t name_len, sjmi_value* out); /* get :91-107 */
int sjmi_value_size(const sjmi_parser* p, const sjmi_value* v);                    /* getSize :109-111; < 0 on error */
/* arrayIterator / objectIterator (:61-67,143-194): first element (or key), then the following one.  In an object the
 * sequence alternates key (a string value), value, key, value ... exactly as they lie on the tape. */
int sjmi_value_first(const sjmi_parser* p, const sjmi_value* container, sjmi_value* out);
int sjmi_value_next(const sjmi_parser* p, const sjmi_value* container, const sjmi_value* child, sjmi_value* out);

/* Optional: page-lock caller-owned host memory that is passed to the host-buffer entry points again and again
 * (SimdJsonParser's padded input, index array and string buffer): H2D / D2H copies of pinned memory skip the
 * driver's staging copy (3-4x faster for the ~1 MB transfers of a single-document parse).  Purely a performance
 * hint: every entry point also works with pageable memory.  Unregister before freeing the memory.
 * ZERO-COPY OUTPUTS: when the output arrays of sjmi_stage1 (indexes), sjmi_stage1_unescape (indexes AND string_buffer) or
 * sjmi_parse_document (tape) are page-locked, device-visible and 16-byte aligned, the kernels write them directly over PCIe:
 * no download, one host synchronisation per call.  The results are the same bytes; as on the staged path, array elements
 * behind the returned counts (up to the capacities) may be overwritten.  A context remembers the device view of such a
 * buffer; sjmi_host_register / sjmi_host_unregister (on any context) make every context forget them, so memory that was
 * page-locked by other means (hipHostMalloc / hipHostRegister of the caller's own) must stay page-locked as long as it is
 * passed to a context, or be followed by one sjmi_host_unregister call (its failure for a foreign pointer is harmless).
 * SJMI_ZERO_COPY=0 in the environment switches the zero-copy paths off. */
int sjmi_host_register(sjmi_ctx* ctx, void* ptr, uint64_t bytes);
/* Optional: a page-locked staging buffer for the INPUT of the single-document host entry points (sjmi_stage1,
 * sjmi_stage1_unescape, sjmi_parse_document).  With one set (bytes >= the document's length), a document handed in from any
 * other address is copied through it -- the reference's padIfNeeded copy (SimdJsonParser.java:42-48), which a caller needs
 * anyway for its stage 2 -- and from 4 MiB on in chunks whose H2D copies run while the next chunk is being copied: the
 * memcpy and the PCIe transfer of a 64 MiB document overlap instead of adding up.  After the call staging[0, len) holds the
 * document.  pinned == NULL switches it off.  The buffer stays the caller's (register it with sjmi_host_register). */
int sjmi_set_input_staging(sjmi_ctx* ctx, void* pinned, uint64_t bytes);
int sjmi_host_unregister(sjmi_ctx* ctx, void* ptr);

/* Run the on-device self-test of the bit-plane transposition; *mismatches == 0 on success. */
int sjmi_selftest(sjmi_ctx* ctx, uint32_t* mismatches);

/* tuning knob for tests: force the chain granule = steps x 4 KiB per worker wave and iteration (1, 2 or 4;
 * 0 = automatic: 1 for documents up to 4 MiB, 2 up to 16 MiB, else 4) */
int sjmi_set_tile_steps(sjmi_ctx* ctx, int steps);

/* Liveness mode of the stage-1 kernel (results never depend on it): 0 (default) = FAST: persistent worker waves plus
 * one scanner workgroup that turns the workers' per-granule aggregates into prefixes; assumes that the whole grid
 * (sized with the occupancy API) is resident at the same time.  1 = SAFE: no scanner, granules by one atomic ticket,
 * every worker resolves its prefix by a decoupled look-back; no residency assumption at all, ~2.5x slower.
 * A FAST launch whose bounded spins trip reports SJMI_ST_INTERNAL: the host-buffer entry points then latch SAFE mode
 * and re-run the launch; the *_device entry points return the status bit to the caller.
 * Environment: SJMI_TILE_MODE=ticket selects SAFE mode at context creation. */
int sjmi_set_tile_mode(sjmi_ctx* ctx, int ticket);
/* Forward-progress assumption, stated once: a FAST launch spins (bounded: tens of milliseconds) on values only OTHER
 * workgroups of the same launch produce, so it needs its whole grid -- sized with the occupancy API to what fits the
 * device -- resident at the same time.  That holds when the kernel has the GPU to itself and, by giving up the static first
 * granule (done automatically while another context of this process still has a stage-1 launch running), when two such
 * kernels share it; it can fail when foreign kernels (another process, PyTorch on another stream) hold CUs for longer
 * than the spin bound.  Results never depend on it -- a launch either completes correctly or reports SJMI_ST_INTERNAL.
 * The host-buffer entry points then latch SAFE mode and re-run by themselves.  For the asynchronous *_device entry
 * point the caller chooses: read SJMI_ST_INTERNAL from its result record and call sjmi_set_tile_mode(ctx, 1), or opt in
 * here (on = 1): sjmi_stage1_device then synchronises its stream after every FAST launch, and on a tripped bound latches
 * SAFE mode and repeats the launch before returning (costs the asynchrony; SAFE mode itself has no residency assumption). */
int sjmi_set_auto_safe(sjmi_ctx* ctx, int on);

/* Measurement hooks for bench.py: when on, every sjmi_stage1_device launch is bracketed by HIP events
 * on the launch stream (kernel only); sjmi_kernel_time returns their summed duration and count. */
int sjmi_set_profiling(sjmi_ctx* ctx, int on);
/* performance-ablation switches for sjmi_stage1_device (1 = no index stores, 2 = no look-back): results are
 * INVALID while any is set (experiments only); test hooks with valid results: 16 = fast-mode launches report
 * SJMI_ST_INTERNAL, 32 = launch only 8 worker workgroups (as if most of the GPU were busy with other work). */
int sjmi_debug_set_flags(sjmi_ctx* ctx, uint32_t flags);
int sjmi_kernel_time(sjmi_ctx* ctx, double* sum_ms, uint32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* SJMI_H */
