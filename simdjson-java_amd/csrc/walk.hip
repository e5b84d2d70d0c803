// SPDX-License-Identifier: Apache-2.0
// Stage 2 on the GPU, host side of the launch: JsonIterator.walkDocument (JsonIterator.java:26-200)
// driving TapeBuilder (TapeBuilder.java:41-217) is the cooperative walker of coop_walk.hip (a wave per document, chunk-parallel
// for one large document); this file queues it and packs the documents' tapes back to back (k_tape_chunk_sums /
// k_tape_chunk_scan / k_tape_compact).  A document the device hands back gets doc_errors[k] = SJMI_WALK_NEEDS_HOST and no tape.
// (Rounds 1-2 also had a lane-per-document walker here, k_doc_walk + walk_doc.h: superseded by the cooperative walker
// and removed in round 3.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1.h"

namespace sjmi {

namespace {

// document k's slot of the scratch tape: a structural makes at most two words, the root adds two
__device__ __forceinline__ unsigned long long scratch_slot(unsigned long long from, uint64_t k) { return 2 * from + 2 * k; }

}  // namespace

// ---- packing the tapes ---------------------------------------------------------------------------
constexpr int PACK_DOCS = 1024;  // documents per workgroup

__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long* s_wave,
                                                              unsigned long long* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, t = 0;
    for (int i = 0; i < nw; ++i) {
        if (i < wave) base += s_wave[i];
        t += s_wave[i];
    }
    *total = t;
    return base + incl - v;
}

__global__ void __launch_bounds__(1024)
k_tape_chunk_sums(const uint32_t* __restrict__ tape_lens, const int32_t* __restrict__ doc_errors, uint64_t n_docs,
                  unsigned long long* __restrict__ chunk_sums, WalkResult* res, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;  // (the tapes were laid out before the walk: nothing to pack)
    __shared__ unsigned long long s_wave[16];
    const uint64_t k = (uint64_t)blockIdx.x * PACK_DOCS + threadIdx.x;
    unsigned long long total;
    (void)block_excl_scan(k < n_docs ? tape_lens[k] : 0u, s_wave, &total);
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = total;
    const int e = k < n_docs ? doc_errors[k] : 0;
    const unsigned long long host = __ballot(e == SJMI_WALK_NEEDS_HOST), bad = __ballot(e > 0);
    if ((threadIdx.x & 63) == 0) {
        if (host) atomicAdd(&res->host_documents, (unsigned long long)__popcll(host));
        if (bad) atomicAdd(&res->failed_documents, (unsigned long long)__popcll(bad));
    }
}

__global__ void __launch_bounds__(1024)
k_tape_chunk_scan(unsigned long long* __restrict__ chunk_sums, uint64_t nchunks, uint64_t n_docs, uint64_t tape_capacity,
                  unsigned long long* __restrict__ tape_offsets, WalkResult* res, const uint32_t* __restrict__ gate, uint32_t gate_want) {
    if (gate && (*gate != 0) != (gate_want != 0)) return;
    __shared__ unsigned long long s_wave[16];
    unsigned long long carry = 0;
    for (uint64_t b = 0; b < nchunks; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const unsigned long long v = i < nchunks ? chunk_sums[i] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_scan(v, s_wave, &total);
        if (i < nchunks) chunk_sums[i] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tape_offsets[0] = 0;  // (the others: k_tape_compact; a single document written in place has no other)
        tape_offsets[n_docs] = carry;
        res->tape_words = carry;
        // (assigned, not OR-ed: behind k_batch_layout -- which judged the PREDICTED lengths of an accepted batch -- this pass
        //  packs by the ACTUAL lengths of the cooperative walker, and a document that failed stage 2 has no tape at all)
        res->flags = (res->flags & ~1u) | (carry > tape_capacity ? 1u : 0u);
    }
}

// one workgroup per PACK_DOCS documents: their offsets, then every wave copies documents (coalesced 8-byte words)
__global__ void __launch_bounds__(1024)
k_tape_compact(const unsigned long long* __restrict__ scratch_tape, const uint32_t* __restrict__ tape_lens,
               const unsigned long long* __restrict__ index_offsets, uint64_t n_docs,
               const unsigned long long* __restrict__ chunk_base, unsigned long long* __restrict__ tape, uint64_t tape_capacity,
               unsigned long long* __restrict__ tape_offsets, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_off[PACK_DOCS];
    const uint64_t k0 = (uint64_t)blockIdx.x * PACK_DOCS, k = k0 + threadIdx.x;
    unsigned long long total;
    const unsigned long long off = chunk_base[blockIdx.x] + block_excl_scan(k < n_docs ? tape_lens[k] : 0u, s_wave, &total);
    s_off[threadIdx.x] = off;
    if (k < n_docs) tape_offsets[k] = off;
    __syncthreads();
    if (chunk_base[blockIdx.x] + total > tape_capacity) return;  // (reported by k_tape_chunk_scan)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < PACK_DOCS; j += 16) {
        const uint64_t d = k0 + (uint64_t)j;
        if (d >= n_docs) break;
        const uint32_t n = tape_lens[d];
        const unsigned long long* src = scratch_tape + scratch_slot(index_offsets[d], d);
        unsigned long long* dst = tape + s_off[j];
        for (uint32_t i = (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
    }
}

// ---- the fused batch pipeline's decision + layout (round 5; see BatchLayout in stage1.h): one workgroup behind k_doc_prepare ----
__global__ void __launch_bounds__(1024)
k_batch_layout(BatchLayout a, uint64_t nchunks) {
    __shared__ unsigned long long s_wave[16];
    __shared__ uint32_t s_acc;
    if (a.gate && *a.gate != 0) return;  // (the repair pass behind an accepted plain pass)
    if (threadIdx.x == 0) {
        const uint32_t st = a.stage1->status;
        const bool acc = a.flags[0] == 0 && st == 0;
        a.flags[1] = acc ? 1u : 0u;
        if (a.pipe_flags) {
            if (a.stage == 0) a.pipe_flags[0] = acc ? 1u : 0u;
            a.pipe_flags[1] = acc ? 1u : 0u;
        }
        s_acc = acc ? 1u : 0u;
        WalkResult zw = {};
        *a.walk = zw;
        a.list[0] = 0;
        UnescapeResult u = {};
        if (acc) u = *a.strings_ws;
        *a.strings_out = u;
        if (!acc && a.optimistic_only) a.stage1_out->status = st | SJMI_ST_REJECTED;
        // the repair pass was accepted: the batch's verdict is the OR of its documents' (the copy itself is clean by construction)
        if (acc && a.status_or) a.stage1_out->status = st | (*a.status_or & (0xFFu | SJMI_ST_INTERNAL));
    }
    if (threadIdx.x < 16) static_cast<uint32_t*>(a.slow_header)[threadIdx.x] = 0;
    __syncthreads();
    if (!s_acc) return;
    unsigned long long carry = 0;
    for (uint64_t b = 0; b < nchunks; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const unsigned long long v = i < nchunks ? a.chunk_sums[i] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_scan(v, s_wave, &total);
        if (i < nchunks) a.chunk_sums[i] = carry + ex;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.tape_offsets[0] = 0;
        a.tape_offsets[a.n_docs] = carry;
        a.walk->tape_words = carry;
        if (carry > a.tape_capacity) a.walk->flags |= 1u;
    }
}

// ---- tapes laid out BEFORE the walk (the accepted plain pass of the fused pipeline: batch.hip k_doc_prepare left every document's
// predicted length and the sums per PREP_DOCS documents; k_tape_chunk_scan turns those into chunk bases) ----------------------
__global__ void __launch_bounds__(PREP_DOCS)
k_tape_offsets(const uint32_t* __restrict__ lens, const unsigned long long* __restrict__ chunk_base, uint64_t n_docs,
               uint64_t tape_capacity, unsigned long long* __restrict__ tape_offsets, DocMeta* __restrict__ metas,
               uint32_t* __restrict__ list, const uint32_t* __restrict__ gate, int32_t* __restrict__ doc_errors) {
    if (gate && *gate == 0) return;
    __shared__ unsigned long long s_wave[16];
    const uint64_t k = (uint64_t)blockIdx.x * PREP_DOCS + threadIdx.x;
    unsigned long long total;
    unsigned long long off = chunk_base[blockIdx.x] + block_excl_scan(k < n_docs ? lens[k] : 0u, s_wave, &total);
    if (k < n_docs) doc_errors[k] = 0;  // (the token walker only writes to it when it declines a document)
    if (k <= n_docs) {
        if (k == n_docs) off = tape_offsets[n_docs];  // (the total: k_tape_chunk_scan)
        tape_offsets[k] = off;
        // a tape that does not fit is reported (WalkResult.flags) and never overrun: offsets beyond the capacity collapse onto
        // it, so the documents there have no room and the walkers write nothing
        const unsigned long long o = off > tape_capacity ? tape_capacity : off;
        metas[k].tape_lo = (uint32_t)o;
        metas[k].tape_hi = (uint32_t)(o >> 32);
    }
    if (k == 0) list[0] = 0;
}
// the same records from the arrays of the three separate calls; the tapes go to the scratch tape (two words per structural + 2
// per document: document k's slot begins at 2 * index_offsets[k] + 2 k) and are packed afterwards
__global__ void __launch_bounds__(256)
k_doc_meta(const unsigned long long* __restrict__ doc_offsets, const unsigned long long* __restrict__ index_offsets,
           const uint32_t* __restrict__ doc_status, const unsigned long long* __restrict__ doc_str_ordinals, uint64_t n_docs,
           DocMeta* __restrict__ metas, uint32_t* __restrict__ list, const uint32_t* __restrict__ skip, int32_t* __restrict__ doc_errors) {
    if (skip && *skip) return;
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k == 0) list[0] = 0;
    if (k > n_docs) return;
    if (k < n_docs) doc_errors[k] = 0;  // (the token walker only writes to it when it declines a document)
    DocMeta m = {};
    const unsigned long long from = index_offsets[k];
    const unsigned long long slot = scratch_slot(from, k);
    m.tape_lo = (uint32_t)slot;
    m.tape_hi = (uint32_t)(slot >> 32);
    if (k < n_docs) {
        m.from = (uint32_t)from;
        m.to = (uint32_t)index_offsets[k + 1];
        m.dso = (uint32_t)doc_str_ordinals[k];
        m.doc_start = (uint32_t)doc_offsets[k];
        m.doc_end = (uint32_t)doc_offsets[k + 1];
        m.st = doc_status ? doc_status[k] : 0u;
        // (a structural range the 32-bit fields cannot hold: the exact walker hands such a document back)
        if (from > 0xFFFFFF00ull || index_offsets[k + 1] > 0xFFFFFF00ull) m.st |= SJMI_ST_INTERNAL;
    }
    metas[k] = m;
}

// workspace: scratch tape (2 count + 2 n words) | tape lengths [n] | chunk sums
static size_t walk_lens_offset(uint64_t count, uint64_t n_docs) { return ((2 * count + 2 * n_docs + 8) * sizeof(unsigned long long) + 63) / 64 * 64; }
static size_t walk_sums_offset(uint64_t count, uint64_t n_docs) { return walk_lens_offset(count, n_docs) + (n_docs * sizeof(uint32_t) + 63) / 64 * 64 + 64; }
static size_t walk_chunks_offset(uint64_t count, uint64_t n_docs) {
    return (walk_sums_offset(count, n_docs) + ((n_docs + PACK_DOCS - 1) / PACK_DOCS + 2) * sizeof(unsigned long long) + 64 + 255) / 256 * 256;
}
static size_t walk_deep_offset(uint64_t count, uint64_t n_docs) {
    return (walk_chunks_offset(count, n_docs) + (n_docs == 1 ? coop_chunk_workspace_bytes(count) : 0) + 255) / 256 * 256;
}
// ... | DocMeta [n + 1] | the exact walker's list [16 + n] | predicted lengths [n] | their sums per PREP_DOCS documents
static size_t walk_metas_offset(uint64_t count, uint64_t n_docs) { return (walk_deep_offset(count, n_docs) + coop_deep_workspace_bytes(n_docs) + 255) / 256 * 256; }
static size_t walk_list_offset(uint64_t count, uint64_t n_docs) { return walk_metas_offset(count, n_docs) + (n_docs + 2) * sizeof(DocMeta); }
static size_t walk_plens_offset(uint64_t count, uint64_t n_docs) { return (walk_list_offset(count, n_docs) + (n_docs + 32) * sizeof(uint32_t) + 63) / 64 * 64; }
static size_t walk_psums_offset(uint64_t count, uint64_t n_docs) { return (walk_plens_offset(count, n_docs) + (n_docs + 16) * sizeof(uint32_t) + 63) / 64 * 64; }
size_t walk_workspace_bytes(uint64_t count, uint64_t n_docs) {
    // (+ the chunk states of the chunk-parallel path for one large document, + the deep nesting levels of every wave)
    return walk_psums_offset(count, n_docs) + ((n_docs + 1) / PREP_DOCS + 4) * sizeof(unsigned long long) + 64;
}
WalkPrepared walk_prepared(void* d_ws, uint64_t count, uint64_t n_docs) {
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    WalkPrepared w;
    w.metas = reinterpret_cast<DocMeta*>(ws + walk_metas_offset(count, n_docs));
    w.lens = reinterpret_cast<uint32_t*>(ws + walk_plens_offset(count, n_docs));
    w.chunk_sums = reinterpret_cast<unsigned long long*>(ws + walk_psums_offset(count, n_docs));
    return w;
}

void* walk_slow_header(void* d_ws, uint64_t count, uint64_t n_docs) { return static_cast<uint8_t*>(d_ws) + walk_deep_offset(count, n_docs); }

__global__ void k_batch_reject(Stage1Result* r) {
    r->count = 0;
    r->status = SJMI_ST_REJECTED;
    r->reserved = 0;
}
hipError_t batch_reject_launch(Stage1Result* d_stage1, hipStream_t stream) {
    hipLaunchKernelGGL(k_batch_reject, dim3(1), dim3(1), 0, stream, d_stage1);
    return hipGetLastError();
}

hipError_t batch_layout_launch(const BatchLayout& a0, void* d_ws, uint64_t count, const uint32_t* lens, DocMeta* metas,
                               int32_t* d_doc_errors, hipStream_t stream) {
    BatchLayout a = a0;
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    a.slow_header = walk_slow_header(d_ws, count, a.n_docs);
    a.list = reinterpret_cast<uint32_t*>(ws + walk_list_offset(count, a.n_docs));
    const uint64_t pchunks = (a.n_docs + PREP_DOCS - 1) / PREP_DOCS;
    hipLaunchKernelGGL(k_batch_layout, dim3(1), dim3(1024), 0, stream, a, pchunks);
    hipLaunchKernelGGL(k_tape_offsets, dim3((unsigned)((a.n_docs + 1 + PREP_DOCS - 1) / PREP_DOCS)), dim3(PREP_DOCS), 0, stream, lens,
                       (const unsigned long long*)a.chunk_sums, a.n_docs, a.tape_capacity, a.tape_offsets, metas, a.list,
                       (const uint32_t*)(a.flags + 1), d_doc_errors);
    return hipGetLastError();
}

hipError_t walk_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, const uint32_t* d_idx,
                       uint64_t count, const unsigned long long* d_index_offsets, const uint32_t* d_doc_status,
                       const uint8_t* d_sb, const unsigned long long* d_doc_str_ordinals, uint64_t string_base, int max_depth,
                       unsigned long long* d_tape, uint64_t tape_capacity, unsigned long long* d_tape_offsets,
                       int32_t* d_doc_errors, void* d_ws, WalkResult* d_res, hipStream_t stream, const Stage1Result* dev_count,
                       const UnescapeResult* dev_strings, const uint32_t* d_soff, bool index_from_zero, bool results_zeroed,
                       const SingleDocTail& tail, const uint32_t* d_prepared, bool layout_done, bool optimistic_only) {
    if (!d_soff) return hipErrorInvalidValue;  // (the record table of the string pass: strings.hip)
    if (optimistic_only && !(layout_done && d_prepared && n_docs > 1)) return hipErrorInvalidValue;
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    unsigned long long* scratch = reinterpret_cast<unsigned long long*>(ws);
    // one document whose index range starts at 0 (only sjmi_parse_document knows that: the walker's slot for document 0 is
    // T = tape + 2 * index_offsets[0]) and room for two words per structural: the walker writes the tape in place
    const bool direct = index_from_zero && n_docs == 1 && (tape_capacity >= 2 * count + 2 || tail.in_place_cap != 0);
    if (direct) scratch = d_tape;
    uint32_t* lens = reinterpret_cast<uint32_t*>(ws + walk_lens_offset(count, n_docs));
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(ws + walk_sums_offset(count, n_docs));
    hipError_t e = (results_zeroed || layout_done) ? hipSuccess : hipMemsetAsync(d_res, 0, sizeof(WalkResult), stream);
    if (e != hipSuccess) return e;
    const uint64_t nchunks = (n_docs + PACK_DOCS - 1) / PACK_DOCS;
    static const bool tokens_off = getenv("SJMI_TOKEN_WALK") && atoi(getenv("SJMI_TOKEN_WALK")) == 0;
    const uint32_t* packed_skip = nullptr;  // device flag != 0: the tapes were laid out before the walk, nothing is packed behind it
    if (n_docs > 1 && (!tokens_off || optimistic_only)) {
        // ---- a batch: the token walker (coop_walk.hip k_tok_stream) with the exact walker behind it for what it declines ----
        const WalkPrepared wp = walk_prepared(d_ws, count, n_docs);
        uint32_t* list = reinterpret_cast<uint32_t*>(ws + walk_list_offset(count, n_docs));
        if (d_prepared && !layout_done) {
            // the fused pipeline's accepted plain pass (*d_prepared != 0): batch.hip k_doc_prepare left the predicted lengths, so
            // the tapes are laid out NOW and the walkers store at the final addresses
            const uint64_t pchunks = (n_docs + PREP_DOCS - 1) / PREP_DOCS;
            hipLaunchKernelGGL(k_tape_chunk_scan, dim3(1), dim3(1024), 0, stream, wp.chunk_sums, pchunks, n_docs, tape_capacity,
                               d_tape_offsets, d_res, d_prepared, 1u);
            hipLaunchKernelGGL(k_tape_offsets, dim3((unsigned)((n_docs + 1 + PREP_DOCS - 1) / PREP_DOCS)), dim3(PREP_DOCS), 0, stream,
                               (const uint32_t*)wp.lens, (const unsigned long long*)wp.chunk_sums, n_docs, tape_capacity, d_tape_offsets,
                               wp.metas, list, d_prepared, d_doc_errors);
        }
        if (d_prepared) packed_skip = d_prepared;
        if (!optimistic_only)
            hipLaunchKernelGGL(k_doc_meta, dim3((unsigned)((n_docs + 1 + 255) / 256)), dim3(256), 0, stream, d_doc_offsets, d_index_offsets,
                           d_doc_status, d_doc_str_ordinals, n_docs, wp.metas, list, packed_skip, d_doc_errors);
        TokLaunch t;
        t.d_buf = d_buf;
        t.d_idx = d_idx;
        t.d_metas = wp.metas;
        t.n_docs = n_docs;
        t.d_doc_offsets = d_doc_offsets;
        t.d_index_offsets = d_index_offsets;
        t.d_doc_status = d_doc_status;
        t.d_doc_str_ordinals = d_doc_str_ordinals;
        t.d_soff = d_soff;
        t.d_sb = d_sb;
        t.string_base = string_base;
        t.max_depth = max_depth;
        t.d_tape = d_prepared ? d_tape : scratch;
        t.d_scratch = optimistic_only ? nullptr : scratch;
        t.header_zeroed = layout_done;
        t.d_sel = d_prepared;
        t.d_tape_lens = optimistic_only ? nullptr : lens;  // (read by the passes that pack scratch tapes: none of them behind tapes laid out in advance)
        t.d_doc_errors = d_doc_errors;
        t.d_list = list;
        t.dev_count = dev_count;
        t.dev_strings = dev_strings;
        t.d_res = d_res;
        t.d_deep_ws = ws + walk_deep_offset(count, n_docs);
        e = tok_walk_launch(t, stream);
        if (e != hipSuccess) return e;
        if (optimistic_only) return hipGetLastError();  // (nothing is packed behind tapes that were laid out before the walk)
        hipLaunchKernelGGL(k_tape_chunk_sums, dim3((unsigned)nchunks), dim3(1024), 0, stream, lens, d_doc_errors, n_docs, sums, d_res, packed_skip);
    } else if (n_docs) {
        // the cooperative walker (coop_walk.hip): a wave per document; STRING payloads from the string pass's record table
        // (direct = one document written in place: the walker's last launch also decides the listed literals and writes the
        //  tape offsets and counters -- three small launches fewer on the single-document latency path)
        e = coop_walk_launch(d_buf, d_doc_offsets, n_docs, d_idx, d_index_offsets, d_doc_status, d_soff, d_sb,
                             d_doc_str_ordinals, string_base, max_depth, scratch, lens, d_doc_errors, dev_count, dev_strings, d_res,
                             stream, n_docs == 1 ? ws + walk_chunks_offset(count, n_docs) : nullptr, count,
                             ws + walk_deep_offset(count, n_docs), direct ? d_tape_offsets : nullptr, tape_capacity, results_zeroed,
                             direct ? tail : SingleDocTail());
        if (e != hipSuccess) return e;
        if (direct) return hipGetLastError();
        hipLaunchKernelGGL(k_tape_chunk_sums, dim3((unsigned)nchunks), dim3(1024), 0, stream, lens, d_doc_errors, n_docs, sums, d_res,
                           (const uint32_t*)nullptr);
    }
    hipLaunchKernelGGL(k_tape_chunk_scan, dim3(1), dim3(1024), 0, stream, sums, nchunks, n_docs, tape_capacity, d_tape_offsets, d_res,
                       packed_skip, 0u);
    if (n_docs && !direct)
        hipLaunchKernelGGL(k_tape_compact, dim3((unsigned)nchunks), dim3(1024), 0, stream, scratch, lens, d_index_offsets, n_docs,
                           sums, d_tape, tape_capacity, d_tape_offsets, packed_skip);
    return hipGetLastError();
}

}  // namespace sjmi
