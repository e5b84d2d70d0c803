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
// their offsets.  A document with an unclosed string or broken UTF-8 poisons the BATCH verdict (status is per
// launch); isolating it needs segmented carries -- listed as next in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

hipError_t split_docs_launch(const uint32_t* d_idx, const Stage1Result* d_res, const unsigned long long* d_doc_offsets,
                             uint64_t n_docs, unsigned long long* d_index_offsets, hipStream_t stream) {
    hipLaunchKernelGGL(k_split_docs, dim3((unsigned)((n_docs + 1 + 255) / 256)), dim3(256), 0, stream, d_idx, d_res,
                       d_doc_offsets, n_docs, d_index_offsets);
    return hipGetLastError();
}

}  // namespace sjmi
