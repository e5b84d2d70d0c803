// stage1.hip -- MI355X (gfx950) stage-1 kernel: fused UTF-8 validation + structural indexing +
// index compaction in ONE pass over the document.
//
// Replaces SimdJsonParser.stage1 = Utf8Validator.validate + StructuralIndexer.index + BitIndexes.write
// (/root/reference/src/main/java/org/simdjson/SimdJsonParser.java:55-58, Utf8Validator.java:54-168,
//  StructuralIndexer.java:196-303, BitIndexes.java:14-41,82-96).
//
// Mapping to the hardware:
//   * one lane  = one 64-byte block (the reference's loop step), loaded as 4 x dwordx4;
//   * the block is transposed to 8 bit planes with v_and + v_msad_u8 (4 mask bits per op), and all
//     classification / escape / string / UTF-8 logic is 64-bit boolean algebra in VGPRs (sj_block.h);
//   * the three serial carries of the reference loop (prevEscaped, prevScalar, previous 4 UTF-8
//     bytes) are LOCAL: each lane re-derives them from the 8 bytes before its block;
//   * the two truly global carries -- in-string parity (XOR scan) and the output offset (+ scan of
//     popcounts) -- are resolved inside the wave by ballot/shuffle, inside the workgroup through
//     LDS, and across workgroups by a single-pass DECOUPLED LOOK-BACK over 8-byte {state,payload}
//     granules (agent-scope relaxed atomics, one granule per tile), so the input is read from HBM
//     exactly once.  Tiles are handed out by an atomic ticket so that a tile only ever waits for
//     tiles whose workgroups have already started (no dependence on dispatch order);
//   * structurals depend on the incoming parity only through a complement
//     (structurals(p) = p ? pot & sm : pot & ~sm), so each tile publishes counts for BOTH parities
//     and the look-back composes functions {0,1} -> (parity, count).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sj_block.h"
#include "stage1.h"

namespace sjmi {

// ---------------------------------------------------------------------------------------------
// bit-plane transposition: 16 dwords (64 bytes) -> 8 x u64
// v_msad_u8 D = S2 + sum_i (S1.byte[i] != 0 ? |S0.byte[i] - S1.byte[i]| : 0)
// With S1 = w & (0x01010101 << k) (each byte 0 or 2^k) and S0.byte[i] = 2^k +- weight_i the sum is
// the weighted popcount = 4 mask bits per instruction; two chained ops give one mask byte.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr uint32_t msad_const(int k, int base_weight_log2) {
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const int wgt = 1 << (base_weight_log2 + i);
        const int v = (1 << k) + wgt <= 255 ? (1 << k) + wgt : (1 << k) - wgt;
        r |= (uint32_t)v << (8 * i);
    }
    return r;
}

template <int K>
__device__ __forceinline__ sj_u64 plane_msad(const uint32_t w[16]) {
    constexpr uint32_t bit = 0x01010101u << K;
    constexpr uint32_t cA = msad_const(K, 0), cB = msad_const(K, 4);
    uint32_t half[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t v = __builtin_amdgcn_msad_u8(cA, w[8 * h + 2 * j] & bit, 0u);
            v = __builtin_amdgcn_msad_u8(cB, w[8 * h + 2 * j + 1] & bit, v);
            acc |= v << (8 * j);
        }
        half[h] = acc;
    }
    return (sj_u64)half[0] | ((sj_u64)half[1] << 32);
}

__device__ __forceinline__ void transpose_fast(const uint32_t w[16], sj_u64 p[8]) {
    p[0] = plane_msad<0>(w);
    p[1] = plane_msad<1>(w);
    p[2] = plane_msad<2>(w);
    p[3] = plane_msad<3>(w);
    p[4] = plane_msad<4>(w);
    p[5] = plane_msad<5>(w);
    p[6] = plane_msad<6>(w);
    p[7] = plane_msad<7>(w);
}

// self-test of the msad transposition against the portable loop (run once per context on the GPU)
__global__ void k_transpose_selftest(const uint32_t* __restrict__ words, uint32_t nblocks, uint32_t* mismatches) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = words[b * 16 + i];
    sj_u64 pf[8], pr[8];
    transpose_fast(w, pf);
    sj_transpose_ref(w, pr);
    uint32_t bad = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) bad += pf[k] != pr[k];
    if (bad) atomicAdd(mismatches, bad);
}

// ---------------------------------------------------------------------------------------------
// wave helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---------------------------------------------------------------------------------------------
// tile-state granules (one u64 per tile; zeroed by hipMemsetAsync before every launch)
//   bits 63..62 : 0 = nothing yet, 1 = AGGREGATE, 2 = INCLUSIVE PREFIX
//   AGGREGATE   : [19:0] structurals if the tile is entered with parity 0, [39:20] with parity 1,
//                 [40] quote parity of the tile
//   PREFIX      : [39:0] structurals in tiles 0..t, [40] in-string parity after tile t
// ---------------------------------------------------------------------------------------------
constexpr sj_u64 TS_AGG = 1ull << 62, TS_PFX = 2ull << 62;
constexpr uint32_t SPIN_LIMIT = 1u << 24;

__device__ __forceinline__ void ts_store(sj_u64* p, sj_u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ sj_u64 ts_load(const sj_u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Executed by all 64 lanes of wave 0.  Returns the parity / structural count entering `tile`.
__device__ __forceinline__ void tile_lookback(sj_u64* tile_state, uint32_t tile, int lane, uint32_t T0,
                                              uint32_t T1, uint32_t tpar, uint32_t* par_in, sj_u64* cnt_in,
                                              Stage1Result* res) {
    if (tile == 0) {
        *par_in = 0;
        *cnt_in = 0;
        if (lane == 0) ts_store(&tile_state[0], TS_PFX | ((sj_u64)tpar << 40) | (sj_u64)T0);
        return;
    }
    if (lane == 0) ts_store(&tile_state[tile], TS_AGG | ((sj_u64)tpar << 40) | ((sj_u64)T1 << 20) | (sj_u64)T0);

    sj_u64 g0 = 0, g1 = 0;  // structurals in the tiles already folded, if entered with parity 0 / 1
    uint32_t gpar = 0;      // their combined quote parity
    long long k = (long long)tile;  // lane i looks at tile k-1-i
    uint32_t P = 0;
    sj_u64 C = 0;
    for (;;) {
        const long long t = k - 1 - lane;
        sj_u64 v = 0;
        int J = 64;
        for (uint32_t spins = 0;; ++spins) {
            if (t >= 0) v = ts_load(&tile_state[t]);
            const bool is_pfx = (t < 0) || ((v >> 62) == 2);
            const bool ready = (t < 0) || (v != 0);
            const sj_u64 pm = __ballot(is_pfx);
            J = pm ? __builtin_ctzll(pm) : 64;
            const sj_u64 need = J >= 64 ? ~0ull : ((1ull << J) - 1ull);
            if ((__ballot(ready) & need) == need) break;
            if (spins > SPIN_LIMIT) {  // never expected: a predecessor tile did not publish
                if (lane == 0) atomicOr(&res->status, SJMI_ST_INTERNAL);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        const bool in_win = lane < J;  // AGGREGATE tiles newer than the nearest PREFIX
        const uint32_t apar = in_win ? (uint32_t)(v >> 40) & 1u : 0u;
        const sj_u64 pb = __ballot(apar);
        // parity accumulated by the window's tiles OLDER than this lane's tile (higher lanes)
        const uint32_t q = (lane < 63) ? (uint32_t)__popcll(pb >> (lane + 1)) & 1u : 0u;
        const uint32_t a0 = (uint32_t)v & 0xFFFFFu, a1 = (uint32_t)(v >> 20) & 0xFFFFFu;
        const uint32_t w0 = wave_sum(in_win ? (q ? a1 : a0) : 0u);  // window entered with parity 0
        const uint32_t w1 = wave_sum(in_win ? (q ? a0 : a1) : 0u);  // ... with parity 1
        const uint32_t wpar = (uint32_t)__popcll(pb) & 1u;
        if (J < 64) {
            const long long tj = k - 1 - J;
            const sj_u64 pv = __shfl(v, J);  // 64-bit shuffle of lane J's granule
            P = tj < 0 ? 0u : (uint32_t)(pv >> 40) & 1u;
            C = tj < 0 ? 0ull : (pv & ((1ull << 40) - 1ull));
            C += P ? w1 : w0;
            P ^= wpar;
            C += P ? g1 : g0;
            P ^= gpar;
            break;
        }
        // no prefix among these 64 tiles: fold the window in front of the suffix and keep walking
        const sj_u64 n0 = (sj_u64)w0 + (wpar ? g1 : g0);
        const sj_u64 n1 = (sj_u64)w1 + (wpar ? g0 : g1);
        g0 = n0;
        g1 = n1;
        gpar ^= wpar;
        k -= 64;
    }
    *par_in = P;
    *cnt_in = C;
    if (lane == 0)
        ts_store(&tile_state[tile], TS_PFX | ((sj_u64)(P ^ tpar) << 40) | (C + (P ? T1 : T0)));
}

// ---------------------------------------------------------------------------------------------
// the stage-1 kernel.  Tile = 256 lanes x S blocks = S * 16 KiB of input.
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(256)
k_stage1(const uint8_t* __restrict__ buf, sj_u64 len, uint32_t* __restrict__ out, sj_u64 out_cap,
         sj_u64* tile_state, uint32_t* ticket, Stage1Result* res) {
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_wpar[S * 4], s_wc0[S * 4], s_wcp[S * 4];
    __shared__ uint32_t s_par_in;
    __shared__ sj_u64 s_cnt_in;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const sj_u64 nblocks = len / 64 + 1;  // the reference always processes one tail block (:255-294)
    const sj_u64 blk0 = (sj_u64)tile * (256 * S);
    const sj_u64 lt_mask = (1ull << lane) - 1ull;

    sj_u64 pot[S], m0[S];
    uint32_t fl[S];  // bit0 quote parity, bit1 ue0, bit2 ue1, bit3 utf8 error

    // ---- phase 1: load, transpose, classify (everything that needs no cross-lane data) ----
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const sj_u64 blk = blk0 + (sj_u64)s * 256 + tid;
        pot[s] = 0;
        m0[s] = 0;
        fl[s] = 0;
        if (blk < nblocks) {
            const sj_u64 start = blk * 64;
            const uint4* src = reinterpret_cast<const uint4*>(buf + start);
            const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
            const uint32_t w[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                                    q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
            uint32_t e_in = 0, p_in = 0;
            SjUtf8Carry uc = {0, 0, 0, 0};
            if (blk > 0) {
                const sj_u64 halo = *reinterpret_cast<const sj_u64*>(buf + start - 8);
                uc = sj_utf8_carry(halo);
                if (!sj_carry_from_halo(halo, &e_in, &p_in)) sj_carry_slow(buf, 0, start, &e_in, &p_in);
            }
            sj_u64 p[8];
            transpose_fast(w, p);
            const sj_u64 rem = len - start;
            sj_mask_tail(p, rem < 64 ? (uint32_t)rem : 64u);
            const SjBlockMasks bm = sj_block(p, e_in, p_in, uc);
            pot[s] = bm.pot;
            m0[s] = bm.sm0;
            fl[s] = bm.qpar | (bm.ue0 << 1) | (bm.ue1 << 2) | (bm.utf8 << 3);
        }
    }

    // ---- phase 2: in-string parity prefix inside the tile (order: step, wave, lane) ----
    uint32_t lpar[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const sj_u64 bal = __ballot(fl[s] & 1u);
        lpar[s] = (uint32_t)__popcll(bal & lt_mask) & 1u;
        if (lane == 0) s_wpar[s * 4 + wave] = (uint32_t)__popcll(bal) & 1u;
    }
    __syncthreads();
    uint32_t tpar = 0;
    {
        uint32_t run = 0;
#pragma unroll
        for (int j = 0; j < S * 4; ++j) {
            const int s = j >> 2;
            if ((j & 3) == wave) lpar[s] ^= run;  // parity of everything before (s, wave)
            run ^= s_wpar[j];
        }
        tpar = run;
    }

    // ---- phase 3: structurals for tile-entry parity 0, counts for both parities, offsets ----
    uint32_t ex0[S], exp_[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        m0[s] = lpar[s] ? (pot[s] & m0[s]) : (pot[s] & ~m0[s]);  // StructuralIndexer.java:251
        const uint32_t c0 = (uint32_t)__popcll(m0[s]), cp = (uint32_t)__popcll(pot[s]);
        const uint32_t i0 = wave_incl_scan(c0, lane), ip = wave_incl_scan(cp, lane);
        ex0[s] = i0 - c0;
        exp_[s] = ip - cp;
        if (lane == 63) {
            s_wc0[s * 4 + wave] = i0;
            s_wcp[s * 4 + wave] = ip;
        }
    }
    __syncthreads();
    uint32_t T0 = 0, TP = 0;
    {
#pragma unroll
        for (int j = 0; j < S * 4; ++j) {
            const int s = j >> 2;
            if ((j & 3) == wave) {
                ex0[s] += T0;
                exp_[s] += TP;
            }
            T0 += s_wc0[j];
            TP += s_wcp[j];
        }
    }
    const uint32_t T1 = TP - T0;

    // ---- phase 4: decoupled look-back (wave 0), broadcast through LDS ----
    if (wave == 0) {
        uint32_t P;
        sj_u64 C;
        tile_lookback(tile_state, tile, lane, T0, T1, tpar, &P, &C, res);
        if (lane == 0) {
            s_par_in = P;
            s_cnt_in = C;
        }
    }
    __syncthreads();
    const uint32_t par_in = s_par_in;
    const sj_u64 cnt_in = s_cnt_in;

    // ---- phase 5: final masks, error flags, index emission (BitIndexes.write :14-41) ----
    uint32_t err = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const sj_u64 blk = blk0 + (sj_u64)s * 256 + tid;
        const sj_u64 mask = par_in ? (pot[s] ^ m0[s]) : m0[s];
        const sj_u64 off = cnt_in + (par_in ? (exp_[s] - ex0[s]) : ex0[s]);
        const uint32_t in_par = lpar[s] ^ par_in;  // parity entering this block
        if (in_par ? (fl[s] & 4u) : (fl[s] & 2u)) err |= SJMI_ST_UNESCAPED;  // :252,:300-302
        if (fl[s] & 8u) err |= SJMI_ST_UTF8;
        const uint32_t cnt = (uint32_t)__popcll(mask);
        if (off + cnt >= out_cap) {  // also keeps room for the sentinel
            if (cnt) err |= SJMI_ST_CAPACITY;
        } else {
            uint32_t* dst = out + off;
            uint32_t base = (uint32_t)(blk * 64);
            uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
            while (lo) {
                *dst++ = base + (uint32_t)__builtin_ctz(lo);
                lo &= lo - 1;
            }
            base += 32;
            while (hi) {
                *dst++ = base + (uint32_t)__builtin_ctz(hi);
                hi &= hi - 1;
            }
        }
    }
    const sj_u64 anyerr = __ballot(err != 0);
    if (anyerr) {
        uint32_t e = err;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) e |= __shfl_xor(e, d);
        if (lane == 0) atomicOr(&res->status, e);
    }
    // the tile that holds the final (tail) block finishes the job: count, sentinel, unclosed string
    if (tid == 0 && tile == (uint32_t)((nblocks - 1) / (256 * S))) {
        const sj_u64 total = cnt_in + (par_in ? T1 : T0);
        res->count = total;
        uint32_t e = 0;
        if (par_in ^ tpar) e |= SJMI_ST_UNCLOSED;  // StructuralIndexer.java:297-299
        if (total < out_cap) out[total] = 0;       // BitIndexes.finish :82-96
        else e |= SJMI_ST_CAPACITY;
        if (e) atomicOr(&res->status, e);
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
size_t stage1_workspace_bytes(uint64_t len, int steps) {
    const uint64_t nblocks = len / 64 + 1;
    const uint64_t tiles = (nblocks + 256ull * steps - 1) / (256ull * steps);
    return WS_TILE_STATE_OFFSET + (size_t)tiles * sizeof(sj_u64);
}

int stage1_pick_steps(uint64_t len) {
    // small documents: small tiles so that more CUs get work; large: 64 KiB tiles (fewer granules)
    return len <= (8u << 20) ? 1 : 4;
}

hipError_t stage1_launch(const uint8_t* d_buf, uint64_t len, uint32_t* d_out, uint64_t out_cap, void* d_ws,
                         int steps, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const uint64_t nblocks = len / 64 + 1;
    const uint64_t tiles = (nblocks + 256ull * steps - 1) / (256ull * steps);
    const size_t ws_bytes = WS_TILE_STATE_OFFSET + (size_t)tiles * sizeof(sj_u64);
    hipError_t e = hipMemsetAsync(d_ws, 0, ws_bytes, stream);
    if (e != hipSuccess) return e;
    uint8_t* ws = static_cast<uint8_t*>(d_ws);
    uint32_t* ticket = reinterpret_cast<uint32_t*>(ws + WS_TICKET_OFFSET);
    Stage1Result* res = reinterpret_cast<Stage1Result*>(ws + WS_RESULT_OFFSET);
    sj_u64* ts = reinterpret_cast<sj_u64*>(ws + WS_TILE_STATE_OFFSET);
    const dim3 grid((unsigned)tiles), block(256);
    if (ev_start && (e = hipEventRecord(ev_start, stream)) != hipSuccess) return e;
    switch (steps) {
    case 1: hipLaunchKernelGGL(k_stage1<1>, grid, block, 0, stream, d_buf, (sj_u64)len, d_out, (sj_u64)out_cap, ts, ticket, res); break;
    case 2: hipLaunchKernelGGL(k_stage1<2>, grid, block, 0, stream, d_buf, (sj_u64)len, d_out, (sj_u64)out_cap, ts, ticket, res); break;
    case 4: hipLaunchKernelGGL(k_stage1<4>, grid, block, 0, stream, d_buf, (sj_u64)len, d_out, (sj_u64)out_cap, ts, ticket, res); break;
    default: return hipErrorInvalidValue;
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_stop) return hipEventRecord(ev_stop, stream);
    return hipSuccess;
}

hipError_t transpose_selftest_launch(const uint32_t* d_words, uint32_t nblocks, uint32_t* d_mismatches,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(k_transpose_selftest, dim3((nblocks + 255) / 256), dim3(256), 0, stream, d_words, nblocks,
                       d_mismatches);
    return hipGetLastError();
}

}  // namespace sjmi
