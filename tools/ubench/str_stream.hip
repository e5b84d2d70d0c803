// Micro-benchmark (experiment, not product): can a ROW-STREAMING string copy -- lane = byte, 64-byte rows, ballots for the
// quote structure, byte-granular writes into an LDS staging buffer, aligned 16-byte stores out -- move twitter-like data
// fast enough to be worth building the real thing around it?  No escapes, no headers, fake tile bases: throughput only.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC str_stream.hip -o libstrstream.so
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int ROWS = 64;          // rows of 64 bytes per tile (4 KiB)
constexpr int PF = 8;             // rows in flight

__device__ __forceinline__ unsigned long long prefix_xor(unsigned long long m) {
    m ^= m << 1; m ^= m << 2; m ^= m << 4; m ^= m << 8; m ^= m << 16; m ^= m << 32;
    return m;
}

__global__ void __launch_bounds__(256)
k_stream(const uint8_t* __restrict__ buf, unsigned long long ntiles, uint8_t* __restrict__ out, unsigned long long* __restrict__ totals) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[4][ROWS * 64 + 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t* st = stage[wv];
    const unsigned long long nw = (unsigned long long)gridDim.x * 4;
    for (unsigned long long t = (unsigned long long)blockIdx.x * 4 + wv; t < ntiles; t += nw) {
        const uint8_t* src = buf + t * (ROWS * 64);
        uint32_t c[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) c[u] = src[u * 64 + lane];
        uint32_t pos = 0, parity = 0;
#pragma unroll 1
        for (int r0 = 0; r0 < ROWS; r0 += PF) {
            uint32_t cur[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) cur[u] = c[u];
            if (r0 + PF < ROWS) {
#pragma unroll
                for (int u = 0; u < PF; ++u) c[u] = src[(r0 + PF + u) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const unsigned long long q = __ballot(cur[u] == '"');
                const unsigned long long in = prefix_xor(q) ^ (parity ? ~0ull : 0ull);
                parity = (uint32_t)(in >> 63);
                const unsigned long long inside = in & ~q;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(inside >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)inside, 0));
                if ((inside >> lane) & 1ull) st[pos + rank] = (uint8_t)cur[u];
                pos += (uint32_t)__popcll(inside);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // flush: aligned 16-byte stores at a fake base
        uint8_t* dst = out + t * (ROWS * 64);
        for (uint32_t o = 16u * lane; o < pos; o += 1024u)
            *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(st + o);
        if (lane == 0) totals[t] = pos;
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int ub_stream(const void* d_buf, unsigned long long n, void* d_out, void* d_totals, void* stream, int grid) {
    const unsigned long long ntiles = n / (ROWS * 64);
    hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)d_buf, ntiles, (uint8_t*)d_out,
                       (unsigned long long*)d_totals);
    return (int)hipGetLastError();
}
