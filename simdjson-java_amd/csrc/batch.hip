// batch.hip -- batched documents: many JSON documents packed back to back in one buffer (each followed by at
// least one whitespace byte, i.e. NDJSON-style), indexed by ONE stage-1 launch over the whole buffer, then
// split per document.
//
// The reference has no batch mode (one document per SimdJsonParser.parse call,
// /root/reference/src/main/java/org/simdjson/SimdJsonParser.java:35-40); north_star adds it: "a batched mode
// shards many documents across the 8 GPUs of one node with RCCL over xGMI only as a gather of per-shard counts".
// Why one launch is exact for well-formed batches: stage 1 is alignment invariant (SURVEY.md 8(a) a3'), a closed
// document leaves the in-string parity at 0, and the whitespace separator clears the prevScalar / escape
// carries, so the structural indexes of the concatenation are the union of the documents' own indexes shifted by
// their offsets.  That launch cannot tell WHICH document is broken, though, and two documents with an unclosed
// string each even cancel in the batch verdict.  The ISOLATED mode below is exact per document whatever the
// others contain: 16 lanes per document, every carry starts from zero at the document's first byte, a document
// that fails stage 1 gets its own SJMI_ST_* bits and contributes no indexes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sj_block.h"
#include "sj_strings.h"
#include "stage1.h"

namespace sjmi {

// index_offsets[k] = number of structural indexes < doc_offsets[k]  (lower bound in the sorted index array)
__global__ void __launch_bounds__(256)
k_split_docs(const uint32_t* __restrict__ idx, const Stage1Result* __restrict__ res, const unsigned long long* __restrict__ doc_offsets,
             uint64_t n_docs, unsigned long long* __restrict__ index_offsets) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k > n_docs) return;
    const unsigned long long count = res->count;
    const unsigned long long target = doc_offsets[k];
    unsigned long long lo = 0, hi = count;
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) >> 1;
        if ((unsigned long long)idx[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    index_offsets[k] = lo;
}

// ---- the optimistic plain pass of the fused batch pipeline (sjmi_parse_batch_device) --------------------------------------
// A batch whose documents each end in a control-character separator (the '\n' of NDJSON; '\r', '\t') and ALL pass stage 1 is
// indexed exactly by ONE plain k_stage1 launch over the packed buffer: alignment invariance makes the indexes those of the
// per-document passes, the separators clear the scalar carry, and a string left open by one document would make the separator
// behind it an unescaped control character inside a string -- so "two unclosed strings cancelling" cannot hide behind a
// clean global verdict.  flags[0] = a separator is missing / an offset pair is unusable; flags[1] = accepted: the
// per-document passes queued behind leave at once.  Any global error bit (or a tripped liveness bound) rejects, and the
// per-document passes give every document its own verdict as before.
__global__ void __launch_bounds__(256)
k_batch_sep_check(const uint8_t* __restrict__ buf, const unsigned long long* __restrict__ doc_offsets, uint64_t n_docs, sj_u64 total_len,
                  uint32_t* __restrict__ flags) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_docs) return;
    const sj_u64 s = doc_offsets[k], e = doc_offsets[k + 1];
    bool bad = e < s || e > total_len;
    // the plain pass scans [0, total_len): bytes in front of the first or behind the last document would feed their in-string /
    // scalar / UTF-8 state into the documents (and their structurals into the count)
    if (k == 0 && s != 0) bad = true;
    if (k + 1 == n_docs && e != total_len) bad = true;
    if (!bad && k + 1 < n_docs) {  // (the last document ends where the batch ends)
        const uint8_t c = e > s ? buf[e - 1] : 0xFF;
        bad = !(c == 0x0A || c == 0x0D || c == 0x09);
    }
    if (bad) atomicOr(&flags[0], 1u);
}
__global__ void k_batch_plain_accept(const Stage1Result* __restrict__ res, uint32_t* __restrict__ flags) {
    if (threadIdx.x == 0) flags[1] = (flags[0] == 0 && res->status == 0) ? 1u : 0u;
}
// k_split_docs + "every document's status is 0", only if the plain pass was accepted
__global__ void __launch_bounds__(256)
k_split_docs_accept(const uint32_t* __restrict__ idx, const Stage1Result* __restrict__ res, const unsigned long long* __restrict__ doc_offsets,
                    uint64_t n_docs, unsigned long long* __restrict__ index_offsets, uint32_t* __restrict__ doc_status,
                    const uint32_t* __restrict__ flags, Stage1Prefixes hint) {
    if (flags[1] == 0) return;
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k > n_docs) return;
    const unsigned long long count = res->count;
    const unsigned long long target = doc_offsets[k];
    unsigned long long lo = 0, hi = count;
    // the stage-1 scanner's per-granule prefixes bracket the answer: 12 steps inside one 16 KiB granule instead of 27 over all
    // indexes (a structural at byte b lies in granule b / granule_bytes)
    if (hint.pfx && hint.granule_bytes) {
        const unsigned long long g = target / hint.granule_bytes;
        if (g < hint.ngran) {
            const unsigned long long above = hint.pfx[g], below = g ? hint.pfx[g - 1] : (2ull << 62);
            if ((above >> 62) == 2 && (below >> 62) == 2) {
                const unsigned long long a = below & 0xFFFFFFFFFFull, b = above & 0xFFFFFFFFFFull;
                if (a <= b && b <= count) {
                    lo = a;
                    hi = b;
                    // ... and inside the granule the structurals are spread evenly enough to guess: start at the interpolated
                    // position and widen the bracket geometrically (the search is a chain of dependent cache misses: ~3 instead
                    // of ~8 for the 200 lines a granule's indexes occupy)
                    if (b - a > 32) {
                        const unsigned long long gstart = g * hint.granule_bytes;
                        unsigned long long p = a + (b - a) * (target - gstart) / hint.granule_bytes;
                        if (p >= b) p = b - 1;
                        if ((unsigned long long)idx[p] < target) {  // the answer is above p
                            unsigned long long step = 16, q = p + 1;
                            lo = q;
                            while (q + step < b && (unsigned long long)idx[q + step] < target) {
                                q += step + 1;
                                lo = q;
                                step *= 4;
                            }
                            if (q + step < b) hi = q + step;
                        } else {                                     // at or below p
                            unsigned long long step = 16, q = p;
                            hi = q;
                            while (q >= a + step + 1 && (unsigned long long)idx[q - step - 1] >= target) {
                                q -= step + 1;
                                hi = q;
                                step *= 4;
                            }
                            if (q >= a + step + 1) lo = q - step;
                        }
                    }
                }
            }
        }
    }
    while (lo < hi) {
        const unsigned long long mid = (lo + hi) >> 1;
        if ((unsigned long long)idx[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    index_offsets[k] = lo;
    if (k < n_docs) doc_status[k] = 0;
}
hipError_t batch_plain_check_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, uint64_t total_len,
                                    uint32_t* d_flags, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_flags, 0, 8, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_batch_sep_check, dim3((unsigned)((n_docs + 255) / 256)), dim3(256), 0, stream, d_buf, d_doc_offsets, n_docs,
                       (sj_u64)total_len, d_flags);
    return hipGetLastError();
}
hipError_t batch_plain_accept_launch(const uint32_t* d_idx, const Stage1Result* d_res, const unsigned long long* d_doc_offsets,
                                     uint64_t n_docs, unsigned long long* d_index_offsets, uint32_t* d_doc_status, uint32_t* d_flags,
                                     hipStream_t stream, const Stage1Prefixes& hint, bool split) {
    hipLaunchKernelGGL(k_batch_plain_accept, dim3(1), dim3(64), 0, stream, d_res, d_flags);
    if (split)
        hipLaunchKernelGGL(k_split_docs_accept, dim3((unsigned)((n_docs + 1 + 255) / 256)), dim3(256), 0, stream, d_idx, d_res, d_doc_offsets,
                       n_docs, d_index_offsets, d_doc_status, (const uint32_t*)d_flags, hint);
    return hipGetLastError();
}

// ---- one pass over the documents of an ACCEPTED plain batch (see DocPrepare in stage1.h) -------------------------------------
// Thread k looks at BOUNDARY k = doc_offsets[k] (n_docs + 1 of them; a workgroup's last thread also at the one behind it):
//   * the structurals of the boundary's 64-byte block that lie in front of it, read from the index array at the position
//     k_stage1 recorded for the block (blkidx): how many (-> index_offsets[k], no search), how many tape words they make and
//     how many of them open a string (-> the document's first string ordinal, on top of blk_ord of the string pass);
//   * then document k = [boundary k, boundary k + 1): its predicted tape length = the per-block word counts of k_stage1
//     (blkw, by the block's entry parity) over its blocks, corrected at both ends, + the two root words.
// A batch is accepted only if every document passed stage 1, so every document begins outside a string.
__device__ __forceinline__ uint32_t prep_words_of(uint32_t c) {  // tape words of a structural by its first byte
    return (c == ',' || c == ':') ? 0u : ((c == '-' || c - '0' <= 9u) ? 2u : 1u);
}
__global__ void __launch_bounds__(PREP_DOCS)
k_doc_prepare(DocPrepare a) {
    // Round 5: this kernel runs BEFORE the acceptance of the plain pass is known -- it contributes to it: every thread looks at
    // the separator in front of its boundary (what k_batch_sep_check re-read 137 MB for) and ORs a missing one into flags[0];
    // k_batch_layout (walk.hip), queued behind, decides.  What it can know it checks: a plain pass with any verdict bit is
    // rejected already.  On a batch that ends up rejected everything written here is overwritten or unused, and every access
    // below is clamped to the buffer and to the index count whatever the offsets say.
    if (a.gate && *a.gate != 0) return;
    if (a.stage1->status != 0) return;
    __shared__ uint32_t s_io[PREP_DOCS + 1], s_pw[PREP_DOCS + 1];
    __shared__ unsigned long long s_sum[PREP_DOCS / 64];
    const uint64_t k0 = (uint64_t)blockIdx.x * PREP_DOCS, k = k0 + threadIdx.x;
    const unsigned long long count = a.stage1->count, nstr = a.strings->reserved;
    const bool strings_ok = !(a.strings->flags & 0xEu);
    auto boundary = [&](uint64_t kb, uint32_t* io, uint32_t* pw, unsigned long long* ord) {
        unsigned long long pos = a.doc_offsets[kb];
        if (pos > a.total_len) pos = a.total_len;
        // (the batch's end is a boundary like any other: the last document's structurals in the tail block are in front of it)
        const unsigned long long b = pos >> 6;
        unsigned long long j = a.blkidx[b];
        uint32_t w = 0, q = 0;
        // Round 5: the kernel is bound by L2 REQUESTS, not bytes -- every lane works on cache lines of its own, so each load
        // instruction of a wave is 64 requests (round 4: 16 index loads + 16 first-byte loads + ~13 others per boundary).  The
        // boundary's block is loaded ONCE (4 x 16 bytes) and everything about its bytes is SWAR algebra on those registers: which
        // bytes are quotes / backslashes (the strings opened in front of the boundary) and which are ',' ':' '-' digits (the
        // tape words of the structurals in front of it); the block's first 16 index entries come as 4 x 16 bytes.
        const unsigned long long start = b << 6;
        const uint32_t upto = (uint32_t)(pos & 63);
        if (upto) {
            struct alignas(16) Q4 { uint32_t a, b, c, d; };
            struct __attribute__((packed, aligned(4))) I4 { uint32_t a, b, c, d; };
            uint32_t w16[16], e[16];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const Q4 q4 = reinterpret_cast<const Q4*>(a.buf + start)[v];
                w16[4 * v] = q4.a;
                w16[4 * v + 1] = q4.b;
                w16[4 * v + 2] = q4.c;
                w16[4 * v + 3] = q4.d;
            }
            // (the array holds count + 1 entries, sentinel included: a group of four is read at most from its last in-bounds
            //  position and shifted; fewer than four entries in all: one by one)
            const unsigned long long last4 = count + 1 >= 4 ? count + 1 - 4 : 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const unsigned long long g = j + 4 * v, jj = g < last4 ? g : last4;
                uint32_t x[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                if (count + 1 >= 4) {
                    const I4 i4 = *reinterpret_cast<const I4*>(a.idx + jj);
                    x[0] = i4.a; x[1] = i4.b; x[2] = i4.c; x[3] = i4.d;
                } else {
                    for (int t = 0; t < 4; ++t) x[t] = (unsigned long long)t < count ? a.idx[t] : 0xFFFFFFFFu;
                }
                const uint32_t sh = (uint32_t)(g - jj);  // (only a clamped group is shifted; g >= count + 1: nothing of it is valid)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint32_t val = 0xFFFFFFFFu;
#pragma unroll
                    for (int u = 0; u < 4; ++u) val = (uint32_t)(t + (int)sh) == (uint32_t)u ? x[u] : val;
                    e[4 * v + t] = g + t < count ? val : 0xFFFFFFFFu;
                }
            }
            unsigned long long qm = 0, bm = 0, sepm = 0, numm = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t x = w16[i];
                auto eq = [&](uint32_t c4) -> uint32_t {  // 0x80 in every byte of x that equals the byte repeated in c4
                    const uint32_t z = x ^ c4;
                    return ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
                };
                auto nib = [](uint32_t f) -> unsigned long long { return (unsigned long long)((((f >> 7) * 0x00204081u) >> 21) & 0xFu); };
                const uint32_t ge30 = ((x | 0x80808080u) - 0x30303030u) & 0x80808080u;
                const uint32_t lt3a = ~((x & 0x7F7F7F7Fu) + 0x46464646u) & 0x80808080u;
                const uint32_t digit = ge30 & lt3a & ~x;
                qm |= nib(eq(0x22222222u)) << (4 * i);
                bm |= nib(eq(0x5C5C5C5Cu)) << (4 * i);
                sepm |= nib(eq(0x2C2C2C2Cu) | eq(0x3A3A3A3Au)) << (4 * i);
                numm |= nib(digit | eq(0x2D2D2D2Du)) << (4 * i);
            }
            // the structurals of the block in front of the boundary, as a mask of their positions
            unsigned long long M = 0;
            uint32_t nb = 0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const bool in = (unsigned long long)e[t] < pos;
                M |= in ? 1ull << (e[t] & 63u) : 0ull;
                nb += in ? 1u : 0u;
            }
            w = (uint32_t)__popcll(M & ~sepm) + (uint32_t)__popcll(M & numm);  // (bracket / string / atom 1 word, number 2, ',' ':' 0)
            j += nb;
            if (nb == 16u) {
                for (int t = 16; t < 64 && j < count; ++t, ++j) {  // (a block has at most 64 structurals)
                    const uint32_t p = a.idx[j];
                    if ((unsigned long long)p >= pos) break;
                    w += prep_words_of(a.buf[p]);
                }
            }
            // the strings opened in the block in front of the boundary: counted on the BYTES (the '"' structurals would miss a quote
            // directly behind a primitive -- 1"abc" -- which opens a string for the string pass all the same: a malformed document,
            // but the documents behind it in the same block need their ordinals right); sj_str_opens_before's algebra on the masks
            const uint32_t in_str = (uint32_t)(a.blkpar[b >> 6] >> (b & 63)) & 1u;
            const uint32_t e_in = b ? sj_backslash_run_parity(a.buf, 0, start) : 0u;
            const sj_u64 quote = qm & ~sj_escaped_mask<sj_u64>(bm, e_in);
            const sj_u64 in0 = sj_prefix_xor(quote);
            const sj_u64 opens = quote & (in_str ? ~in0 : in0);
            q = (uint32_t)__popcll(opens & ((1ull << upto) - 1ull));
        }
        *io = (uint32_t)j;
        *pw = w;
        unsigned long long o = strings_ok ? (unsigned long long)a.blk_ord[b] + q : nstr;
        *ord = o > nstr ? nstr : o;
        return pos;
    };
    uint32_t io = 0, pw = 0;
    unsigned long long ord = 0, s = 0;
    if (k <= a.n_docs) {
        s = boundary(k, &io, &pw, &ord);
        s_io[threadIdx.x] = io;
        s_pw[threadIdx.x] = pw;
        a.index_offsets[k] = io;
        a.doc_ord[k] = ord;
        if (a.doc_str_offsets) a.doc_str_offsets[k] = ord < nstr ? a.soff[ord] : a.strings->total_bytes;
        if (threadIdx.x == PREP_DOCS - 1 && k < a.n_docs) {  // the boundary behind this workgroup's last document
            uint32_t io2, pw2;
            unsigned long long ord2;
            (void)boundary(k + 1, &io2, &pw2, &ord2);
            s_io[PREP_DOCS] = io2;
            s_pw[PREP_DOCS] = pw2;
        }
    }
    __syncthreads();
    uint32_t len = 0;
    if (k < a.n_docs) {
        unsigned long long e = a.doc_offsets[k + 1];
        {   // the optimistic plain pass is exact only if the documents cover the buffer and each ends in a control-character separator
            // (see k_batch_sep_check above: the same test, on the boundary this thread reads anyway)
            const unsigned long long s_raw = a.doc_offsets[k];
            bool bad = e < s_raw || e > a.total_len || (k == 0 && s_raw != 0) || (k + 1 == a.n_docs && e != a.total_len);
            if (!bad && k + 1 < a.n_docs) {
                uint8_t c = e > s_raw ? a.buf[e - 1] : (a.relaxed ? 0x20 : 0xFF);
                // (the repair pass reads the COPY -- a failing document is blank there -- except for a surviving document's trailing
                //  backslash, which the copy has lost: the sanitizer shortens an odd run by one for the string pass)
                if (a.boundary_buf && e > s_raw && a.boundary_buf[e - 1] == 0x5C && !(a.status_in && a.status_in[k] != 0)) c = 0x5C;
                if (!a.relaxed) {
                    bad = !(c == 0x0A || c == 0x0D || c == 0x09);
                } else {
                    // the repair pass: every document has its own verdict and the failing ones are blank, so no string is open at
                    // a boundary; what is left to rule out is a scalar running ON across it -- StructuralIndexer.java:243-248: a
                    // scalar (or a quote) is a structural only behind a byte that is not a non-quote scalar character, i.e.
                    // behind whitespace, an operator or a quote
                    const bool ws = c == 0x20 || c == 0x0A || c == 0x0D || c == 0x09;
                    const bool op = c == '{' || c == '}' || c == '[' || c == ']' || c == ':' || c == ',';
                    bad = !(ws || op || c == '"');
                }
            }
            if (bad) atomicOr(&a.flags[0], 1u);
        }
        if (e > a.total_len) e = a.total_len;
        if (s > e) s = e;
        // whole blocks [bs, be): block bs counted from its start (the part in front of the document comes off again below),
        // the block that holds the end boundary only through the correction pw[k + 1]
        const unsigned long long bs = s >> 6, be = e >> 6;
        uint32_t sum = 0;
        {
            // eight blocks per trip: their word counts in one 16-byte load (the arrays are padded), their entry parities from
            // the one or two 64-block words the trip touches
            struct __attribute__((packed, aligned(2))) W8 { uint32_t a, b, c, d; };
            for (unsigned long long b0 = bs; b0 < be; b0 += 8) {
                const W8 v = *reinterpret_cast<const W8*>(a.blkw + b0);
                const unsigned long long p0 = a.blkpar[b0 >> 6], p1 = a.blkpar[(b0 + 7) >> 6];
                const uint32_t sh = (uint32_t)(b0 & 63);
                const uint32_t pb = (uint32_t)((p0 >> sh) | (sh > 56 ? p1 << (64 - sh) : 0ull)) & 0xFFu;  // bit t: block b0 + t enters inside a string
                const uint32_t ww[4] = {v.a, v.b, v.c, v.d};
                const uint32_t left = be - b0 < 8 ? (uint32_t)(be - b0) : 8u;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const uint32_t w2 = (ww[t >> 1] >> (16 * (t & 1))) & 0xFFFFu;
                    const uint32_t wv = ((pb >> t) & 1u) ? (w2 >> 8) : (w2 & 0xFFu);
                    sum += (uint32_t)t < left ? wv : 0u;
                }
            }
        }
        const uint32_t to = s_io[threadIdx.x + 1];
        len = sum - pw + s_pw[threadIdx.x + 1] + 2u;  // + the two root words (TapeBuilder.java:41-48)
        a.lens[k] = len;
        const uint32_t st_k = a.status_in ? a.status_in[k] : 0u;  // (status_in may BE doc_status: read before it is written)
        a.doc_status[k] = st_k;
        DocMeta m;
        m.from = io;
        m.to = to;
        m.dso = (uint32_t)ord;
        m.doc_start = (uint32_t)s;
        m.doc_end = (uint32_t)e;
        m.st = st_k;
        m.tape_lo = m.tape_hi = 0;  // (k_tape_offsets)
        a.metas[k] = m;
    }
    // the workgroup's predicted words -> chunk_sums (the scan of k_tape_chunk_scan / k_tape_offsets, walk.hip)
    unsigned long long v = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < PREP_DOCS / 64; ++w) t += s_sum[w];
        a.chunk_sums[blockIdx.x] = t;
    }
}
hipError_t batch_prepare_launch(const DocPrepare& a, hipStream_t stream) {
    hipLaunchKernelGGL(k_doc_prepare, dim3((unsigned)((a.n_docs + 1 + PREP_DOCS - 1) / PREP_DOCS)), dim3(PREP_DOCS), 0, stream, a);
    return hipGetLastError();
}

hipError_t split_docs_launch(const uint32_t* d_idx, const Stage1Result* d_res, const unsigned long long* d_doc_offsets,
                             uint64_t n_docs, unsigned long long* d_index_offsets, hipStream_t stream) {
    hipLaunchKernelGGL(k_split_docs, dim3((unsigned)((n_docs + 1 + 255) / 256)), dim3(256), 0, stream, d_idx, d_res,
                       d_doc_offsets, n_docs, d_index_offsets);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// isolated mode: one row of 16 lanes per document (documents of a batch are small; a long one is walked 1 KiB at a time)
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) DocU16 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) DocU8 { unsigned long long v; };

// the first n (0 .. 64) of a block's 64 bytes (w: its 16 dwords) to a byte-granular address: whole 16-byte chunks, then 8 / 4 / 2 / 1
__device__ __forceinline__ void doc_store_block(uint8_t* __restrict__ dst, const uint32_t (&w)[16], uint32_t n) {
    struct __attribute__((packed, aligned(1))) U4 { uint32_t v; };
    struct __attribute__((packed, aligned(1))) U2 { uint16_t v; };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (n >= 16u * (q + 1)) {
            const DocU16 v = {w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
            *reinterpret_cast<DocU16*>(dst + 16 * q) = v;
        } else if (n > 16u * q) {  // (one chunk per block at most)
            const uint32_t m = n - 16u * q;  // 1 .. 15
            uint8_t* p = dst + 16 * q;
            uint32_t a = w[4 * q], b = w[4 * q + 1], c2 = w[4 * q + 2], d = w[4 * q + 3];
            if (m & 8u) {
                DocU8 v;
                v.v = ((unsigned long long)b << 32) | a;
                *reinterpret_cast<DocU8*>(p) = v;
                p += 8;
                a = c2;
                b = d;
            }
            if (m & 4u) {
                U4 v = {a};
                *reinterpret_cast<U4*>(p) = v;
                p += 4;
                a = b;
            }
            if (m & 2u) {
                U2 v = {(uint16_t)a};
                *reinterpret_cast<U2*>(p) = v;
                p += 2;
                a >>= 16;
            }
            if (m & 1u) *p = (uint8_t)a;
        }
    }
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t bdpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
}
// WRITE = false: doc_status[k] and counts[k] (0 for a failing document); WRITE = true: the indexes of the passing
// documents at index_offsets[k].  Same per-block algebra as the single-document kernel (sj_block.h); blocks are counted
// from the document's first byte, so the loads are byte-granular.
// One document per ROW of 16 lanes (four documents per wave), 1 KiB per step: the documents of a batch are small (a
// whole wave per ~1 KB document left 3/4 of the lanes idle), the DPP row scans are exactly 16 lanes wide, and the four
// rows only share the trip count of the longest document.
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_doc_pass(const uint8_t* __restrict__ buf, const unsigned long long* __restrict__ doc_offsets, uint64_t n_docs,
           uint32_t* __restrict__ counts, uint32_t* __restrict__ doc_status,
           const unsigned long long* __restrict__ index_offsets, uint32_t* __restrict__ out, uint64_t out_cap,
           sj_u64 total_len, const uint32_t* __restrict__ skip, uint32_t* __restrict__ status_or, uint8_t* __restrict__ copy) {
    if (skip && *skip) return;  // (the optimistic plain pass of the fused pipeline was accepted: k_batch_plain_accept)
    __shared__ uint4 s_rows[WRITE ? 1 : 4][WRITE ? 1 : 64][4];  // (WRITE = false: a wave's 64 blocks on their way to the copy)
    uint32_t seen = 0;          // (WRITE = false: the OR of this thread's documents' verdicts, for status_or)
    // copy (WRITE = false, round 6): the SANITIZED COPY of the batch for the repair pass, made on the way -- the row stores the bytes
    // of its blocks as it classifies them, and the chunks of a document whose verdict is not 0 once more as spaces (every lane the
    // chunks it stored: the same lane to the same addresses, program order).  Exactly the documents' own bytes are written: a batch whose documents
    // do not cover the buffer is not one the repair pass takes (k_doc_prepare), and the per-document passes make their own copy.
    const int lane = threadIdx.x & 63;
    const int rl = lane & 15;         // lane inside the row
    const int rshift = lane & ~15;    // first lane of the row
    const uint32_t row_lt = (1u << rl) - 1u;
    const uint64_t nrows = (uint64_t)gridDim.x * 16;
    for (uint64_t k0 = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 6) * 4; k0 < n_docs; k0 += nrows) {
        const uint64_t k = k0 + (uint64_t)(lane >> 4);
        bool live = k < n_docs;
        bool bad_range = false;
        sj_u64 s = 0, len = 0, base = 0;
        if (live) {
            // (device-resident offsets cannot be validated by the host: a non-monotonic pair or an offset past the end
            //  becomes an empty / clipped document with SJMI_ST_INTERNAL instead of a wrapped length)
            s = doc_offsets[k];
            const sj_u64 e = doc_offsets[k + 1];
            bad_range = e < s || e > total_len;
            if (s > total_len) s = total_len;
            len = (e < s ? s : (e > total_len ? total_len : e)) - s;
            if (WRITE) {
                live = doc_status[k] == 0;
                base = index_offsets[k];
            }
        }
        const sj_u64 nblocks = live ? len / 64 + 1 : 0;  // the reference always processes one tail block (StructuralIndexer.java:255-294)
        uint32_t parity = 0, err = 0;
        sj_u64 cnt = 0;
        for (sj_u64 b0 = 0; __ballot(b0 < nblocks); b0 += 16) {
            const sj_u64 blk = b0 + rl;
            const bool active = blk < nblocks;
            const sj_u64 cblk = active ? blk : (nblocks ? nblocks - 1 : 0);
            const sj_u64 start = s + cblk * 64;
            uint32_t w[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const DocU16 v = *reinterpret_cast<const DocU16*>(buf + start + 16 * q);
                w[4 * q] = v.a;
                w[4 * q + 1] = v.b;
                w[4 * q + 2] = v.c;
                w[4 * q + 3] = v.d;
            }
            uint32_t e_in = 0, p_in = 0;
            SjUtf8Carry uc = {0, 0, 0, 0};
            if (active && blk > 0) {
                const sj_u64 halo = reinterpret_cast<const DocU8*>(buf + start - 8)->v;
                uc = sj_utf8_carry(halo);
                if (!sj_carry_from_halo(halo, &e_in, &p_in)) sj_carry_slow(buf, s, start, &e_in, &p_in);
            }
            sj_u64 p[8];
            const sj_u64 rem = len - cblk * 64;
            if (!WRITE && copy) {
                // The row's 16 blocks are 1 KiB of CONTIGUOUS copy: stored lane by lane -- a 64-byte block each -- they are four
                // instructions of 64 scattered 16-byte pieces each (measured: 0.80 ms for the pass).  Through LDS the row's lanes
                // store NEIGHBOURING 16-byte chunks, chunk c = rl + 16 j of the row in instruction j: 0.67 ms.  (The loads keep
                // the block-per-lane form: coalesced through LDS as well they cannot overlap the algebra any more, 0.91 ms.)
                const int wvi = threadIdx.x >> 6;
                uint4* const mine = &s_rows[wvi][lane][0];
#pragma unroll
                for (int q = 0; q < 4; ++q) mine[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const sj_u64 row_off = b0 * 64;  // (bytes of the document in front of the row's blocks of this trip)
                const uint32_t row_valid = (live && len > row_off) ? (len - row_off < 1024 ? (uint32_t)(len - row_off) : 1024u) : 0u;
                uint8_t* const row_dst = copy + s + row_off;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t c2 = (uint32_t)rl + 16u * j;
                    const uint4 v = s_rows[wvi][rshift + (c2 >> 2)][c2 & 3u];
                    if (16u * c2 + 16u <= row_valid) {
                        const DocU16 o = {v.x, v.y, v.z, v.w};
                        *reinterpret_cast<DocU16*>(row_dst + 16u * c2) = o;
                    } else if (16u * c2 < row_valid) {  // (the document's last chunk: one lane of the row, once per document)
                        const uint32_t wq[16] = {v.x, v.y, v.z, v.w, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        doc_store_block(row_dst + 16u * c2, wq, row_valid - 16u * c2);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            sj_transpose_butterfly(w, p);
            sj_mask_tail(p, rem < 64 ? (uint32_t)rem : 64u);
            SjBlockMasks bm = sj_block(p, e_in, p_in, uc);
            if (!active) {
                bm.pot = 0;
                bm.sm0 = 0;
                bm.qpar = bm.ue0 = bm.ue1 = bm.utf8 = 0;
            }
            const uint32_t bal = (uint32_t)(__ballot(bm.qpar != 0) >> rshift) & 0xFFFFu;  // this row's quote parities
            const uint32_t lp = ((uint32_t)__popc(bal & row_lt) & 1u) ^ parity;          // in-string parity entering the block
            parity ^= (uint32_t)__popc(bal) & 1u;
            const sj_u64 m = lp ? (bm.pot & bm.sm0) : (bm.pot & ~bm.sm0);  // StructuralIndexer.java:251
            if (lp ? bm.ue1 : bm.ue0) err |= SJMI_ST_UNESCAPED;            // :252
            if (bm.utf8) err |= SJMI_ST_UTF8;
            const uint32_t c = (uint32_t)__popcll(m);
            uint32_t incl = c;  // inclusive scan inside the row of 16 lanes
            incl = bdpp_add<0x111, 0xF>(incl);
            incl = bdpp_add<0x112, 0xF>(incl);
            incl = bdpp_add<0x114, 0xF>(incl);
            incl = bdpp_add<0x118, 0xF>(incl);
            if (WRITE) {
                sj_u64 pos = base + cnt + (incl - c);
                const uint32_t bstart = (uint32_t)start;
                for (sj_u64 bits = m; bits; bits &= bits - 1, ++pos)
                    if (pos < out_cap) out[pos] = bstart + (uint32_t)__builtin_ctzll(bits);  // BitIndexes.write :14-41
            }
            cnt += (uint32_t)__shfl((int)incl, rshift + 15);
        }
        if (!WRITE) {
            uint32_t all = err;
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) all |= __shfl_xor(all, d);  // (stays inside the row)
            if (parity) all |= SJMI_ST_UNCLOSED;  // :297-299
            if (bad_range) all |= SJMI_ST_INTERNAL;
            if (rl == 0 && k < n_docs) {
                doc_status[k] = all;
                counts[k] = all ? 0u : (uint32_t)cnt;
                seen |= all;
            }
            if (copy && all) {  // (rare) a failing document is blank in the copy
                // every lane blanks exactly the chunks IT stored above (chunk rl + 16 j of every 1 KiB trip): the same lane to the
                // same addresses, in program order
                uint32_t sp[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) sp[i] = 0x20202020u;
                for (sj_u64 row_off = 0; row_off < len; row_off += 1024) {
                    const uint32_t row_valid = len - row_off < 1024 ? (uint32_t)(len - row_off) : 1024u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t c2 = (uint32_t)rl + 16u * j;
                        if (16u * c2 < row_valid) doc_store_block(copy + s + row_off + 16u * c2, sp, row_valid - 16u * c2 < 16u ? row_valid - 16u * c2 : 16u);
                    }
                }
            }
        }
    }
    if (!WRITE && status_or && seen) atomicOr(status_or, seen);  // (rare: only the threads that saw a failing document)
}

// ---- index_offsets = exclusive scan of counts: chunk sums, scan of the chunk sums (one workgroup), chunk scans ----
constexpr int SCAN_CHUNK = 16384;  // documents per workgroup of 1024 lanes

__device__ __forceinline__ unsigned long long block_excl_scan_1024(unsigned long long v, unsigned long long* s_wave,
                                                                    unsigned long long* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl_up(x, d);
        if (lane >= d) x += t;
    }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    unsigned long long off = 0, all = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) off += s_wave[w];
        all += s_wave[w];
    }
    __syncthreads();
    *total = all;
    return off + x - v;
}

__global__ void __launch_bounds__(1024)
k_doc_chunk_sums(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ doc_status, uint64_t n_docs,
                 unsigned long long* __restrict__ chunk_sums, uint32_t* __restrict__ status_or, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    __shared__ unsigned long long s_wave[16];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK;
    unsigned long long sum = 0;
    uint32_t st = 0;
    for (int j = 0; j < SCAN_CHUNK / 1024; ++j) {
        const uint64_t i = base + (uint64_t)j * 1024 + threadIdx.x;
        if (i < n_docs) {
            sum += counts[i];
            st |= doc_status[i];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d);
        st |= __shfl_xor(st, d);
    }
    if ((threadIdx.x & 63) == 0) {
        s_wave[threadIdx.x >> 6] = sum;
        if (st) atomicOr(status_or, st);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += s_wave[w];
        chunk_sums[blockIdx.x] = t;
    }
}

// exclusive scan of the chunk sums in place (one workgroup); total, sentinel and the OR of the document statuses
__global__ void __launch_bounds__(1024)
k_doc_scan(unsigned long long* __restrict__ chunk_sums, uint64_t nchunks, const uint32_t* __restrict__ status_or,
           uint64_t n_docs, unsigned long long* __restrict__ index_offsets, uint32_t* __restrict__ out, uint64_t out_cap,
           Stage1Result* res, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    __shared__ unsigned long long s_wave[16];
    unsigned long long carry = 0;
    for (uint64_t b = 0; b < nchunks; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const unsigned long long v = i < nchunks ? chunk_sums[i] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_scan_1024(v, s_wave, &total);
        if (i < nchunks) chunk_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        index_offsets[n_docs] = carry;
        res->count = carry;
        uint32_t e = *status_or & (0xFFu | SJMI_ST_INTERNAL);
        if (carry < out_cap) out[carry] = 0;  // BitIndexes.finish :82-96
        else e |= SJMI_ST_CAPACITY;
        res->status = e;
    }
}

__global__ void __launch_bounds__(1024)
k_doc_offsets(const uint32_t* __restrict__ counts, uint64_t n_docs, const unsigned long long* __restrict__ chunk_base,
              unsigned long long* __restrict__ index_offsets, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    __shared__ unsigned long long s_wave[16];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK;
    unsigned long long carry = chunk_base[blockIdx.x];
    for (int j = 0; j < SCAN_CHUNK / 1024; ++j) {
        const uint64_t i = base + (uint64_t)j * 1024 + threadIdx.x;
        const unsigned long long v = i < n_docs ? counts[i] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_scan_1024(v, s_wave, &total);
        if (i < n_docs) index_offsets[i] = carry + ex;
        carry += total;
    }
}

// workspace: counts[n_docs] | status word | chunk sums
static size_t iso_status_offset(uint64_t n_docs) { return (((size_t)n_docs * sizeof(uint32_t) + 63) / 64) * 64; }
static size_t iso_chunks_offset(uint64_t n_docs) { return iso_status_offset(n_docs) + 64; }
size_t batch_isolated_workspace_bytes(uint64_t n_docs) {
    return iso_chunks_offset(n_docs) + ((size_t)(n_docs / SCAN_CHUNK) + 2) * sizeof(unsigned long long);
}

const uint32_t* batch_status_or(const uint32_t* d_counts, uint64_t n_docs) {
    return reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(d_counts) + iso_status_offset(n_docs));
}
static unsigned doc_pass_grid(uint64_t n_docs) {
    const uint64_t want = (n_docs + 15) / 16;  // 16 documents per workgroup and trip
    return (unsigned)(want < 8192 ? want : 8192);  // grid-stride over the documents
}
hipError_t batch_verdicts_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, uint32_t* d_doc_status,
                                 uint32_t* d_counts, hipStream_t stream, uint64_t total_len, const uint32_t* d_skip, uint8_t* d_copy) {
    uint8_t* ws = reinterpret_cast<uint8_t*>(d_counts);
    uint32_t* status_or = reinterpret_cast<uint32_t*>(ws + iso_status_offset(n_docs));
    hipError_t e = hipMemsetAsync(status_or, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if (n_docs)
        hipLaunchKernelGGL(k_doc_pass<false>, dim3(doc_pass_grid(n_docs)), dim3(256), 0, stream, d_buf, d_doc_offsets, n_docs, d_counts,
                           d_doc_status, (const unsigned long long*)nullptr, (uint32_t*)nullptr, (uint64_t)0, (sj_u64)total_len, d_skip,
                           status_or, d_copy);
    return hipGetLastError();
}
hipError_t batch_indexes_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs, uint32_t* d_out,
                                uint64_t out_cap, unsigned long long* d_index_offsets, uint32_t* d_doc_status, uint32_t* d_counts,
                                Stage1Result* d_res, hipStream_t stream, uint64_t total_len, const uint32_t* d_skip) {
    uint8_t* ws = reinterpret_cast<uint8_t*>(d_counts);
    uint32_t* status_or = reinterpret_cast<uint32_t*>(ws + iso_status_offset(n_docs));
    unsigned long long* chunk_sums = reinterpret_cast<unsigned long long*>(ws + iso_chunks_offset(n_docs));
    const uint64_t nchunks = (n_docs + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (n_docs)
        hipLaunchKernelGGL(k_doc_chunk_sums, dim3((unsigned)nchunks), dim3(1024), 0, stream, d_counts, d_doc_status, n_docs,
                           chunk_sums, status_or, d_skip);
    hipLaunchKernelGGL(k_doc_scan, dim3(1), dim3(1024), 0, stream, chunk_sums, nchunks, status_or, n_docs, d_index_offsets,
                       d_out, out_cap, d_res, d_skip);
    if (n_docs) {
        hipLaunchKernelGGL(k_doc_offsets, dim3((unsigned)nchunks), dim3(1024), 0, stream, d_counts, n_docs, chunk_sums,
                           d_index_offsets, d_skip);
        hipLaunchKernelGGL(k_doc_pass<true>, dim3(doc_pass_grid(n_docs)), dim3(256), 0, stream, d_buf, d_doc_offsets, n_docs, d_counts,
                           d_doc_status, (const unsigned long long*)d_index_offsets, d_out, out_cap, (sj_u64)total_len, d_skip,
                           (uint32_t*)nullptr, (uint8_t*)nullptr);
    }
    return hipGetLastError();
}
hipError_t batch_isolated_launch(const uint8_t* d_buf, const unsigned long long* d_doc_offsets, uint64_t n_docs,
                                 uint32_t* d_out, uint64_t out_cap, unsigned long long* d_index_offsets,
                                 uint32_t* d_doc_status, uint32_t* d_counts, Stage1Result* d_res, hipStream_t stream,
                                 uint64_t total_len, const uint32_t* d_skip) {
    const hipError_t e = batch_verdicts_launch(d_buf, d_doc_offsets, n_docs, d_doc_status, d_counts, stream, total_len, d_skip, nullptr);
    if (e != hipSuccess) return e;
    return batch_indexes_launch(d_buf, d_doc_offsets, n_docs, d_out, out_cap, d_index_offsets, d_doc_status, d_counts, d_res, stream,
                                total_len, d_skip);
}

}  // namespace sjmi
