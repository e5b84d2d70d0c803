// Does the ACCESS PATTERN of stage 1's document loads matter to the memory system?  A read-only kernel over 4 GiB, persistent like
// k_stage1 (1024 workgroups x 256 threads, a wave per 4 KiB step), two patterns:
//   A  lane l loads the four 16-byte quarters of ITS 64-byte block (what k_stage1 does: an instruction touches a quarter of 32 lines)
//   B  lane l loads 16 bytes at l * 16 + i * 1024 (an instruction covers 8 whole lines)
// read only, with a contiguous write of 0.34 bytes per input byte beside it (the index array's share), and that write as streaming stores.  Prints TB/s of input.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/load_pattern tools/ubench/load_pattern.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
static double g_last_tbps = 0;
template <int PATTERN, int WRITE>
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ src, uint64_t nsteps, uint4* __restrict__ dst, uint32_t* sink) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (uint64_t s = wave; s < nsteps; s += nw) {
        const uint4* p = src + s * 256;  // 4 KiB = 256 x 16 bytes
        uint4 a, b, c, d;
        if (PATTERN == 0) { a = p[lane * 4]; b = p[lane * 4 + 1]; c = p[lane * 4 + 2]; d = p[lane * 4 + 3]; }
        else { a = p[lane]; b = p[lane + 64]; c = p[lane + 128]; d = p[lane + 192]; }
        acc += a.x ^ b.y ^ c.z ^ d.w;
        if (WRITE) {  // 88 x 16 bytes per 4 KiB step = 0.34 bytes written per byte read (twitter.json: 0.35), contiguous per wave
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t v = {acc, a.y, b.z, c.w};
            u32x4_t* o = reinterpret_cast<u32x4_t*>(dst + s * 88);
            if (WRITE == 2) { __builtin_nontemporal_store(v, o + lane); if (lane < 24) __builtin_nontemporal_store(v, o + 64 + lane); }
            else { o[lane] = v; if (lane < 24) o[64 + lane] = v; }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <int P, int W>
static void run(const char* name, const uint4* src, uint64_t nsteps, uint4* dst, uint32_t* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<P, W>), dim3(1024), dim3(256), 0, 0, src, nsteps, dst, sink);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<P, W>), dim3(1024), dim3(256), 0, 0, src, nsteps, dst, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-44s %.3f ms  %.2f TB/s of input\n", name, ms, nsteps * 4096.0 / ms / 1e9);
    g_last_tbps = nsteps * 4096.0 / ms / 1e9;
}

// ... and the SIZE of a wave's write burst: G steps read, then 88 * G x 16 bytes stored contiguously (k_stage1: G = 4)
template <int G, int NT, int Q = 88>
__global__ void __launch_bounds__(256) kb(const uint4* __restrict__ src, uint64_t ngran, uint4* __restrict__ dst, uint32_t* sink) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    for (uint64_t g = wave; g < ngran; g += nw) {
        for (int t = 0; t < G; ++t) {
            const uint4* p = src + (g * G + t) * 256;
            const uint4 a = p[lane * 4], b = p[lane * 4 + 1], c = p[lane * 4 + 2], d = p[lane * 4 + 3];
            acc += a.x ^ b.y ^ c.z ^ d.w;
        }
        u32x4_t v = {acc, 1, 2, 3};
        u32x4_t* o = reinterpret_cast<u32x4_t*>(dst + g * Q * G);
        for (int i = lane; i < Q * G; i += 64) {
            if (NT) __builtin_nontemporal_store(v, o + i);
            else o[i] = v;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <int G, int NT, int Q = 88>
static void runb(const char* name, const uint4* src, uint64_t nsteps, uint4* dst, uint32_t* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kb<G, NT, Q>), dim3(1024), dim3(256), 0, 0, src, nsteps / G, dst, sink);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kb<G, NT, Q>), dim3(1024), dim3(256), 0, 0, src, nsteps / G, dst, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-44s %.3f ms  %.2f TB/s of input\n", name, ms, nsteps * 4096.0 / ms / 1e9);
    g_last_tbps = nsteps * 4096.0 / ms / 1e9;
}
int main(int argc, char** argv) {
    const bool quick = argc > 1 && argv[1][0] == 'q';  // bench.py: three figures as one JSON object
    const uint64_t bytes = 4ull << 30, nsteps = bytes / 4096;
    uint4 *src, *dst; uint32_t* sink;
    hipMalloc(&src, bytes); hipMalloc(&dst, nsteps * 220 * 16 + 4096); hipMalloc(&sink, 4);
    hipMemset(src, 1, bytes);
    if (quick) {
        run<0, 0>("read only", src, nsteps, dst, sink);
        const double ro = g_last_tbps;
        runb<4, 0>("read + 0.34 B/B stored, 5.6 KB bursts", src, nsteps, dst, sink);
        const double rw = g_last_tbps;
        runb<4, 1>("... streaming stores", src, nsteps, dst, sink);
        const double rwnt = g_last_tbps;
        printf("{\"read_only_TBps\": %.3f, \"read_plus_index_stores_TBps_of_input\": %.3f, \"read_plus_streaming_index_stores_TBps_of_input\": %.3f}\n", ro, rw, rwnt);
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("A quarter-of-a-line loads, read only", src, nsteps, dst, sink);
        run<1, 0>("B coalesced loads, read only", src, nsteps, dst, sink);
        run<0, 1>("A quarter-of-a-line loads + index-like stores", src, nsteps, dst, sink);
        run<1, 1>("B coalesced loads + index-like stores", src, nsteps, dst, sink);
        run<0, 2>("A ... + streaming stores", src, nsteps, dst, sink);
        run<1, 2>("B ... + streaming stores", src, nsteps, dst, sink);
    }
    runb<1, 0>("bursts of 1.4 KB (1 step)", src, nsteps, dst, sink);
    runb<4, 0>("bursts of 5.6 KB (4 steps = k_stage1)", src, nsteps, dst, sink);
    runb<4, 1>("bursts of 5.6 KB, streaming stores", src, nsteps, dst, sink);
    runb<16, 0>("bursts of 22 KB (16 steps)", src, nsteps, dst, sink);
    runb<16, 1>("bursts of 22 KB, streaming stores", src, nsteps, dst, sink);
    runb<64, 1>("bursts of 90 KB, streaming stores", src, nsteps, dst, sink);
    // k_stage1_batch's mix on the configs[3] documents: 0.86 bytes stored per byte read (indexes 0.75 + side outputs), 1 GiB worth
    runb<4, 0, 220>("0.86 B/B stored, 14 KB bursts (1/4 of the buffer x4)", src, nsteps, dst, sink);
    runb<4, 1, 220>("0.86 B/B stored, streaming stores", src, nsteps, dst, sink);
    runb<2, 0, 220>("0.86 B/B stored, 7 KB bursts (two windows)", src, nsteps, dst, sink);
    return 0;
}
