// Micro-benchmark: issue rate of the VALU instructions stage 1 is made of (wave64, gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t a[8];
    unsigned long long b[4];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i * 77 + threadIdx.x; f[i] = (float)a[i]; }
    for (int i = 0; i < 4; ++i) b[i] = ((unsigned long long)a[i] << 32) | a[i + 4];
    const uint32_t c = seed * 3 + 1;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = (a[i] & c) + 0;                          // placeholder replaced below
        }
        if (OP == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_msad_u8 %0, %1, %0, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(1.0001f));
        } else if (OP == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(b[i]));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(b[i]));
        } else if (OP == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_or3_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(c));
        } else if (OP == 6) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (OP == 7) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 9) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 10) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_bfi_b32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 11) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 12) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        } else if (OP == 13) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_sad_u8 %0, %1, %0, %0" : "+v"(a[i]) : "v"(c));
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)f[i];
    for (int i = 0; i < 4; ++i) r ^= (uint32_t)b[i] ^ (uint32_t)(b[i] >> 32);
    if (r == 0x12345678) out[0] = r;
}
template <int OP> void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 blocks (32 waves) per CU
    k<OP><<<blocks, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 2); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0, instr = waves * ITERS * 8.0;
    const double cyc_per_inst_per_simd = (ms * 1e-3 * 2.4e9) / (instr / 1024.0);
    printf("%-16s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc_per_inst_per_simd);
}
int main() {
    uint32_t* d; hipMalloc(&d, 64);
    run<0>("v_and_b32", d); run<9>("v_xor_b32", d); run<11>("v_add_u32", d); run<1>("v_msad_u8", d); run<13>("v_sad_u8", d);
    run<2>("v_fma_f32", d); run<3>("v_lshlrev_b64", d); run<4>("v_or3_b32", d); run<5>("v_lshl_or_b32", d);
    run<10>("v_bfi_b32", d); run<6>("v_bcnt_u32_b32", d); run<7>("v_dot4_u32_u8", d); run<8>("v_perm_b32", d);
    run<12>("v_pk_add_u16", d);
    return 0;
}
