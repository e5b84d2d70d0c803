// masks.hip -- sjmi_stage1_masks: the six per-block bit masks of the reference's stage-1 loop
// (/root/reference/src/main/java/org/simdjson/StructuralIndexer.java:210-252: escaped, quote, inString, op, whitespace,
// structurals), reconstructed from the engine's own formulation for the bit-mask parity tests (north_star: "bit-exact
// with the reference's own stage-1 bitmasks").
//
// The streaming kernel (stage1.hip) never holds these masks: per block it has `pot` and `sm0` for an incoming in-string
// parity of 0 and flips them once the parity prefix is known.  This diagnostic path runs the SAME per-block algebra
// (sj_block.h: plane transposition, halo carries, sj_block) and resolves the only global carry, the in-string parity,
// with a three-step XOR scan:
//   k_mask_parity  one lane per block: the block's quote parity; one ballot per 64 blocks -> a u64 word per wave-step
//   k_mask_scan    one workgroup: XOR-prefix over the words' parities -> parity entering each word
//   k_mask_write   one lane per block: recompute, apply the parity, store 6 x u64 (48 bytes per block)
// Not a hot path (the document is read twice, 0.75 B written per input byte); it exists so that the masks the reference
// computes can be compared bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sj_block.h"
#include "stage1.h"

namespace sjmi {

struct MaskU16 { uint32_t a, b, c, d; };

__device__ __forceinline__ SjBlockMasks mask_block(const uint8_t* __restrict__ buf, sj_u64 len, sj_u64 blk, SjBlockDetail* det) {
    const sj_u64 start = blk * 64;
    uint32_t w[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // (the buffer is 16-byte aligned with 64 readable bytes after len)
        const MaskU16 v = *reinterpret_cast<const MaskU16*>(buf + start + 16 * q);
        w[4 * q] = v.a;
        w[4 * q + 1] = v.b;
        w[4 * q + 2] = v.c;
        w[4 * q + 3] = v.d;
    }
    uint32_t e_in = 0, p_in = 0;
    SjUtf8Carry uc = {0, 0, 0, 0};
    if (blk > 0) {
        const sj_u64 halo = *reinterpret_cast<const sj_u64*>(buf + start - 8);
        uc = sj_utf8_carry(halo);
        if (!sj_carry_from_halo(halo, &e_in, &p_in)) sj_carry_slow(buf, 0, start, &e_in, &p_in);
    }
    sj_u64 p[8];
    sj_transpose_butterfly(w, p);
    const sj_u64 rem = len - start;
    sj_mask_tail(p, rem < 64 ? (uint32_t)rem : 64u);
    return sj_block(p, e_in, p_in, uc, true, det);
}

__global__ void __launch_bounds__(256)
k_mask_parity(const uint8_t* __restrict__ buf, sj_u64 len, sj_u64 nblocks, sj_u64* __restrict__ words) {
    const sj_u64 blk = (sj_u64)blockIdx.x * 256 + threadIdx.x;
    uint32_t qpar = 0;
    if (blk < nblocks) qpar = mask_block(buf, len, blk, nullptr).qpar;
    const sj_u64 bal = __ballot(qpar != 0);
    if ((threadIdx.x & 63) == 0 && (blk >> 6) < (nblocks + 63) / 64) words[blk >> 6] = bal;
}

// in place: words[w] keeps its 64 block parities; entering[w] = parity of all quotes in front of word w
__global__ void __launch_bounds__(1024)
k_mask_scan(const sj_u64* __restrict__ words, sj_u64 nwords, uint8_t* __restrict__ entering) {
    __shared__ uint32_t s_par[1024];
    const sj_u64 per = (nwords + 1023) / 1024;
    const sj_u64 b = (sj_u64)threadIdx.x * per, e = b + per < nwords ? b + per : nwords;
    uint32_t par = 0;
    for (sj_u64 i = b; i < e; ++i) par ^= (uint32_t)__popcll(words[i]) & 1u;
    s_par[threadIdx.x] = par;
    __syncthreads();
    uint32_t in = 0;
    for (unsigned t = 0; t < threadIdx.x; ++t) in ^= s_par[t];
    for (sj_u64 i = b; i < e; ++i) {
        entering[i] = (uint8_t)in;
        in ^= (uint32_t)__popcll(words[i]) & 1u;
    }
}

__global__ void __launch_bounds__(256)
k_mask_write(const uint8_t* __restrict__ buf, sj_u64 len, sj_u64 nblocks, const sj_u64* __restrict__ words,
             const uint8_t* __restrict__ entering, sj_u64* __restrict__ masks) {
    const sj_u64 blk = (sj_u64)blockIdx.x * 256 + threadIdx.x;
    if (blk >= nblocks) return;
    SjBlockDetail det;
    const SjBlockMasks bm = mask_block(buf, len, blk, &det);
    const sj_u64 wbits = words[blk >> 6];
    const uint32_t parity_in = ((uint32_t)entering[blk >> 6] ^ (uint32_t)__popcll(wbits & ((1ull << (blk & 63)) - 1ull))) & 1u;
    sj_u64 out[6];
    sj_reference_masks(bm, det, parity_in, out);
#pragma unroll
    for (int k = 0; k < 6; ++k) masks[blk * 6 + k] = out[k];
}

size_t masks_workspace_bytes(uint64_t len) {
    const uint64_t nwords = (len / 64 + 1 + 63) / 64;
    return (size_t)nwords * 8 + (((size_t)nwords + 63) & ~(size_t)63) + 64;
}

hipError_t masks_launch(const uint8_t* d_buf, uint64_t len, unsigned long long* d_masks, void* d_ws, hipStream_t stream) {
    const uint64_t nblocks = len / 64 + 1;  // the reference always processes one tail block (StructuralIndexer.java:255-294)
    const uint64_t nwords = (nblocks + 63) / 64;
    sj_u64* words = static_cast<sj_u64*>(d_ws);
    uint8_t* entering = reinterpret_cast<uint8_t*>(words + nwords);
    const unsigned grid = (unsigned)((nblocks + 255) / 256);
    hipLaunchKernelGGL(k_mask_parity, dim3(grid), dim3(256), 0, stream, d_buf, (sj_u64)len, (sj_u64)nblocks, words);
    hipLaunchKernelGGL(k_mask_scan, dim3(1), dim3(1024), 0, stream, (const sj_u64*)words, (sj_u64)nwords, entering);
    hipLaunchKernelGGL(k_mask_write, dim3(grid), dim3(256), 0, stream, d_buf, (sj_u64)len, (sj_u64)nblocks,
                       (const sj_u64*)words, (const uint8_t*)entering, (sj_u64*)d_masks);
    return hipGetLastError();
}

}  // namespace sjmi
