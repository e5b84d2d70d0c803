// valu_rates.hip -- issue cost of the VALU / SALU / LDS instructions the stage-1 and string kernels are made of, on gfx950.
// Every SIMD runs W waves (default 4: the occupancy of k_stage1) that each issue UNROLL x ITERS copies of one instruction on
// 8 independent register sets; the figure printed is shader-clock cycles per wave instruction PER SIMD (4.0 = full rate for a
// wave64 on a 16-lane SIMD).  Measurement tool, not part of the product.  build+run: tools/microbench/run.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>

#define ITERS 2048

// the body is 8 x 4 = 32 instructions per loop trip, on 8 independent destination registers
#define REP8(OP)                                                                                                       \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY(NAME, ASM)                                                                                                \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, unsigned seed) {                             \
        unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, \
                 a7 = a0 * 19;                                                                                         \
        unsigned b = seed * 29 + threadIdx.x, c = seed * 31 + 7;                                                       \
        unsigned long long q0 = a0, q1 = a1, q2 = a2, q3 = a3;                                                         \
        unsigned long long s0 = seed, s1 = seed * 3ull; const unsigned sc = seed * 77u;                                                               \
        const unsigned long long w0 = wall_clock64();                                                                  \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                    \
        for (int i = 0; i < ITERS; ++i) {                                                                              \
            asm volatile(ASM ASM ASM ASM                                                                               \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0), "+v"(q1), \
                           "+v"(q2), "+v"(q3), "+s"(s0), "+s"(s1)                                                     \
                         : "v"(b), "v"(c), "s"(sc)                                                                           \
                         : "vcc", "memory");                                                                           \
        }                                                                                                              \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                    \
        const unsigned long long w1 = wall_clock64();                                                                  \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = (t1 - t0) | ((w1 - w0) << 40);       \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + q0 + q1 + q2 + q3 + s0 + s1 == 0x12345) out[0] = 1;               \
    }

// operands: %0..%7 = a0..a7 (v32), %8..%11 = q0..q3 (v64), %12,%13 = s64, %14 = b, %15 = c (v32), %16 = s32
#define A8(INS, TAIL) INS " %0, " TAIL "\n" INS " %1, " TAIL "\n" INS " %2, " TAIL "\n" INS " %3, " TAIL "\n" INS " %4, " TAIL "\n" INS " %5, " TAIL "\n" INS " %6, " TAIL "\n" INS " %7, " TAIL "\n"
#define S8(I0, I1, I2, I3, I4, I5, I6, I7) I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"

BODY(k_and_vop2, S8("v_and_b32 %0, %14, %0", "v_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2", "v_and_b32 %3, %14, %3", "v_and_b32 %4, %14, %4", "v_and_b32 %5, %14, %5", "v_and_b32 %6, %14, %6", "v_and_b32 %7, %14, %7"))
BODY(k_and_literal, S8("v_and_b32 %0, 0x0f0f0f0f, %0", "v_and_b32 %1, 0x33333333, %1", "v_and_b32 %2, 0x55555555, %2", "v_and_b32 %3, 0x0f0f0f0f, %3", "v_and_b32 %4, 0x33333333, %4", "v_and_b32 %5, 0x55555555, %5", "v_and_b32 %6, 0x0f0f0f0f, %6", "v_and_b32 %7, 0x33333333, %7"))
BODY(k_and_sgpr, S8("v_and_b32 %0, %16, %0", "v_and_b32 %1, %16, %1", "v_and_b32 %2, %16, %2", "v_and_b32 %3, %16, %3", "v_and_b32 %4, %16, %4", "v_and_b32 %5, %16, %5", "v_and_b32 %6, %16, %6", "v_and_b32 %7, %16, %7"))
BODY(k_lshr_imm, S8("v_lshrrev_b32 %0, 4, %0", "v_lshrrev_b32 %1, 4, %1", "v_lshrrev_b32 %2, 4, %2", "v_lshrrev_b32 %3, 4, %3", "v_lshrrev_b32 %4, 4, %4", "v_lshrrev_b32 %5, 4, %5", "v_lshrrev_b32 %6, 4, %6", "v_lshrrev_b32 %7, 4, %7"))
BODY(k_bfi, S8("v_bfi_b32 %0, %14, %0, %15", "v_bfi_b32 %1, %14, %1, %15", "v_bfi_b32 %2, %14, %2, %15", "v_bfi_b32 %3, %14, %3, %15", "v_bfi_b32 %4, %14, %4, %15", "v_bfi_b32 %5, %14, %5, %15", "v_bfi_b32 %6, %14, %6, %15", "v_bfi_b32 %7, %14, %7, %15"))
BODY(k_bfi_sgpr_mask, S8("v_bfi_b32 %0, %16, %0, %15", "v_bfi_b32 %1, %16, %1, %15", "v_bfi_b32 %2, %16, %2, %15", "v_bfi_b32 %3, %16, %3, %15", "v_bfi_b32 %4, %16, %4, %15", "v_bfi_b32 %5, %16, %5, %15", "v_bfi_b32 %6, %16, %6, %15", "v_bfi_b32 %7, %16, %7, %15"))
BODY(k_bitop3, S8("v_bitop3_b32 %0, %14, %0, %15 bitop3:0x96", "v_bitop3_b32 %1, %14, %1, %15 bitop3:0x96", "v_bitop3_b32 %2, %14, %2, %15 bitop3:0x96", "v_bitop3_b32 %3, %14, %3, %15 bitop3:0x96", "v_bitop3_b32 %4, %14, %4, %15 bitop3:0x96", "v_bitop3_b32 %5, %14, %5, %15 bitop3:0x96", "v_bitop3_b32 %6, %14, %6, %15 bitop3:0x96", "v_bitop3_b32 %7, %14, %7, %15 bitop3:0x96"))
BODY(k_bitop3_two_distinct, S8("v_bitop3_b32 %0, %14, %0, %0 bitop3:0x96", "v_bitop3_b32 %1, %14, %1, %1 bitop3:0x96", "v_bitop3_b32 %2, %14, %2, %2 bitop3:0x96", "v_bitop3_b32 %3, %14, %3, %3 bitop3:0x96", "v_bitop3_b32 %4, %14, %4, %4 bitop3:0x96", "v_bitop3_b32 %5, %14, %5, %5 bitop3:0x96", "v_bitop3_b32 %6, %14, %6, %6 bitop3:0x96", "v_bitop3_b32 %7, %14, %7, %7 bitop3:0x96"))
BODY(k_perm, S8("v_perm_b32 %0, %14, %0, %15", "v_perm_b32 %1, %14, %1, %15", "v_perm_b32 %2, %14, %2, %15", "v_perm_b32 %3, %14, %3, %15", "v_perm_b32 %4, %14, %4, %15", "v_perm_b32 %5, %14, %5, %15", "v_perm_b32 %6, %14, %6, %15", "v_perm_b32 %7, %14, %7, %15"))
BODY(k_perm_sgpr_sel, S8("v_perm_b32 %0, %14, %0, %16", "v_perm_b32 %1, %14, %1, %16", "v_perm_b32 %2, %14, %2, %16", "v_perm_b32 %3, %14, %3, %16", "v_perm_b32 %4, %14, %4, %16", "v_perm_b32 %5, %14, %5, %16", "v_perm_b32 %6, %14, %6, %16", "v_perm_b32 %7, %14, %7, %16"))
BODY(k_lshl_or, S8("v_lshl_or_b32 %0, %0, 4, %14", "v_lshl_or_b32 %1, %1, 4, %14", "v_lshl_or_b32 %2, %2, 4, %14", "v_lshl_or_b32 %3, %3, 4, %14", "v_lshl_or_b32 %4, %4, 4, %14", "v_lshl_or_b32 %5, %5, 4, %14", "v_lshl_or_b32 %6, %6, 4, %14", "v_lshl_or_b32 %7, %7, 4, %14"))
BODY(k_and_or, S8("v_and_or_b32 %0, %0, %14, %15", "v_and_or_b32 %1, %1, %14, %15", "v_and_or_b32 %2, %2, %14, %15", "v_and_or_b32 %3, %3, %14, %15", "v_and_or_b32 %4, %4, %14, %15", "v_and_or_b32 %5, %5, %14, %15", "v_and_or_b32 %6, %6, %14, %15", "v_and_or_b32 %7, %7, %14, %15"))
BODY(k_or3, S8("v_or3_b32 %0, %0, %14, %15", "v_or3_b32 %1, %1, %14, %15", "v_or3_b32 %2, %2, %14, %15", "v_or3_b32 %3, %3, %14, %15", "v_or3_b32 %4, %4, %14, %15", "v_or3_b32 %5, %5, %14, %15", "v_or3_b32 %6, %6, %14, %15", "v_or3_b32 %7, %7, %14, %15"))
BODY(k_alignbit, S8("v_alignbit_b32 %0, %0, %14, 4", "v_alignbit_b32 %1, %1, %14, 4", "v_alignbit_b32 %2, %2, %14, 4", "v_alignbit_b32 %3, %3, %14, 4", "v_alignbit_b32 %4, %4, %14, 4", "v_alignbit_b32 %5, %5, %14, 4", "v_alignbit_b32 %6, %6, %14, 4", "v_alignbit_b32 %7, %7, %14, 4"))
BODY(k_bfe, S8("v_bfe_u32 %0, %0, 1, 30", "v_bfe_u32 %1, %1, 1, 30", "v_bfe_u32 %2, %2, 1, 30", "v_bfe_u32 %3, %3, 1, 30", "v_bfe_u32 %4, %4, 1, 30", "v_bfe_u32 %5, %5, 1, 30", "v_bfe_u32 %6, %6, 1, 30", "v_bfe_u32 %7, %7, 1, 30"))
BODY(k_lshl_b64, S8("v_lshlrev_b64 %8, 3, %8", "v_lshlrev_b64 %9, 3, %9", "v_lshlrev_b64 %10, 3, %10", "v_lshlrev_b64 %11, 3, %11", "v_lshlrev_b64 %8, 5, %8", "v_lshlrev_b64 %9, 5, %9", "v_lshlrev_b64 %10, 5, %10", "v_lshlrev_b64 %11, 5, %11"))
BODY(k_lshr_b64, S8("v_lshrrev_b64 %8, 3, %8", "v_lshrrev_b64 %9, 3, %9", "v_lshrrev_b64 %10, 3, %10", "v_lshrrev_b64 %11, 3, %11", "v_lshrrev_b64 %8, 5, %8", "v_lshrrev_b64 %9, 5, %9", "v_lshrrev_b64 %10, 5, %10", "v_lshrrev_b64 %11, 5, %11"))
BODY(k_lshl_add_u64, S8("v_lshl_add_u64 %8, %8, 1, %9", "v_lshl_add_u64 %9, %9, 1, %10", "v_lshl_add_u64 %10, %10, 1, %11", "v_lshl_add_u64 %11, %11, 1, %8", "v_lshl_add_u64 %8, %8, 1, %9", "v_lshl_add_u64 %9, %9, 1, %10", "v_lshl_add_u64 %10, %10, 1, %11", "v_lshl_add_u64 %11, %11, 1, %8"))
BODY(k_mov_b64, S8("v_mov_b64 %8, %9", "v_mov_b64 %9, %10", "v_mov_b64 %10, %11", "v_mov_b64 %11, %8", "v_mov_b64 %8, %9", "v_mov_b64 %9, %10", "v_mov_b64 %10, %11", "v_mov_b64 %11, %8"))
BODY(k_mov_b32, S8("v_mov_b32 %0, %1", "v_mov_b32 %1, %2", "v_mov_b32 %2, %3", "v_mov_b32 %3, %4", "v_mov_b32 %4, %5", "v_mov_b32 %5, %6", "v_mov_b32 %6, %7", "v_mov_b32 %7, %0"))
BODY(k_cndmask_vcc, S8("v_cndmask_b32 %0, %0, %14, vcc", "v_cndmask_b32 %1, %1, %14, vcc", "v_cndmask_b32 %2, %2, %14, vcc", "v_cndmask_b32 %3, %3, %14, vcc", "v_cndmask_b32 %4, %4, %14, vcc", "v_cndmask_b32 %5, %5, %14, vcc", "v_cndmask_b32 %6, %6, %14, vcc", "v_cndmask_b32 %7, %7, %14, vcc"))
BODY(k_cndmask_sgpr, S8("v_cndmask_b32 %0, %0, %14, %12", "v_cndmask_b32 %1, %1, %14, %12", "v_cndmask_b32 %2, %2, %14, %12", "v_cndmask_b32 %3, %3, %14, %12", "v_cndmask_b32 %4, %4, %14, %12", "v_cndmask_b32 %5, %5, %14, %12", "v_cndmask_b32 %6, %6, %14, %12", "v_cndmask_b32 %7, %7, %14, %12"))
BODY(k_add_dpp_row_shr, S8("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf"))
BODY(k_add_dpp_row_bcast, S8("v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf", "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf", "v_add_u32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf", "v_add_u32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf", "v_add_u32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf", "v_add_u32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf", "v_add_u32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf", "v_add_u32_dpp %7, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf"))
BODY(k_bcnt, S8("v_bcnt_u32_b32 %0, %14, %0", "v_bcnt_u32_b32 %1, %14, %1", "v_bcnt_u32_b32 %2, %14, %2", "v_bcnt_u32_b32 %3, %14, %3", "v_bcnt_u32_b32 %4, %14, %4", "v_bcnt_u32_b32 %5, %14, %5", "v_bcnt_u32_b32 %6, %14, %6", "v_bcnt_u32_b32 %7, %14, %7"))
BODY(k_ffbl, S8("v_ffbl_b32 %0, %0", "v_ffbl_b32 %1, %1", "v_ffbl_b32 %2, %2", "v_ffbl_b32 %3, %3", "v_ffbl_b32 %4, %4", "v_ffbl_b32 %5, %5", "v_ffbl_b32 %6, %6", "v_ffbl_b32 %7, %7"))
BODY(k_cmp_vcc, S8("v_cmp_eq_u32 vcc, %0, %14", "v_cmp_eq_u32 vcc, %1, %14", "v_cmp_eq_u32 vcc, %2, %14", "v_cmp_eq_u32 vcc, %3, %14", "v_cmp_eq_u32 vcc, %4, %14", "v_cmp_eq_u32 vcc, %5, %14", "v_cmp_eq_u32 vcc, %6, %14", "v_cmp_eq_u32 vcc, %7, %14"))
BODY(k_cmp_sdwa, S8("v_cmp_eq_u32_sdwa vcc, %0, %14 src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_eq_u32_sdwa vcc, %1, %14 src0_sel:BYTE_1 src1_sel:DWORD", "v_cmp_eq_u32_sdwa vcc, %2, %14 src0_sel:BYTE_2 src1_sel:DWORD", "v_cmp_eq_u32_sdwa vcc, %3, %14 src0_sel:BYTE_3 src1_sel:DWORD", "v_cmp_eq_u32_sdwa vcc, %4, %14 src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_eq_u32_sdwa vcc, %5, %14 src0_sel:BYTE_1 src1_sel:DWORD", "v_cmp_eq_u32_sdwa vcc, %6, %14 src0_sel:BYTE_2 src1_sel:DWORD", "v_cmp_eq_u32_sdwa vcc, %7, %14 src0_sel:BYTE_3 src1_sel:DWORD"))
BODY(k_mul_u24, S8("v_mul_u32_u24 %0, %0, %14", "v_mul_u32_u24 %1, %1, %14", "v_mul_u32_u24 %2, %2, %14", "v_mul_u32_u24 %3, %3, %14", "v_mul_u32_u24 %4, %4, %14", "v_mul_u32_u24 %5, %5, %14", "v_mul_u32_u24 %6, %6, %14", "v_mul_u32_u24 %7, %7, %14"))
BODY(k_mul_lo_u32, S8("v_mul_lo_u32 %0, %0, %14", "v_mul_lo_u32 %1, %1, %14", "v_mul_lo_u32 %2, %2, %14", "v_mul_lo_u32 %3, %3, %14", "v_mul_lo_u32 %4, %4, %14", "v_mul_lo_u32 %5, %5, %14", "v_mul_lo_u32 %6, %6, %14", "v_mul_lo_u32 %7, %7, %14"))
BODY(k_readlane, S8("v_readlane_b32 s20, %0, 3", "v_readlane_b32 s21, %1, 3", "v_readlane_b32 s22, %2, 3", "v_readlane_b32 s23, %3, 3", "v_readlane_b32 s20, %4, 3", "v_readlane_b32 s21, %5, 3", "v_readlane_b32 s22, %6, 3", "v_readlane_b32 s23, %7, 3"))
BODY(k_salu_and_b64, S8("s_and_b64 %12, %12, %13", "s_or_b64 %13, %12, %13", "s_and_b64 %12, %12, %13", "s_or_b64 %13, %12, %13", "s_and_b64 %12, %12, %13", "s_or_b64 %13, %12, %13", "s_and_b64 %12, %12, %13", "s_or_b64 %13, %12, %13"))
BODY(k_valu_salu_mix, S8("v_and_b32 %0, %14, %0", "s_and_b64 %12, %12, %13", "v_and_b32 %2, %14, %2", "s_or_b64 %13, %12, %13", "v_and_b32 %4, %14, %4", "s_and_b64 %12, %12, %13", "v_and_b32 %6, %14, %6", "s_or_b64 %13, %12, %13"))
BODY(k_bpermute, S8("ds_bpermute_b32 %0, %14, %0", "ds_bpermute_b32 %1, %14, %1", "ds_bpermute_b32 %2, %14, %2", "ds_bpermute_b32 %3, %14, %3", "ds_bpermute_b32 %4, %14, %4", "ds_bpermute_b32 %5, %14, %5", "ds_bpermute_b32 %6, %14, %6", "ds_bpermute_b32 %7, %14, %7\ns_waitcnt lgkmcnt(0)"))
BODY(k_pk_add_u16, S8("v_pk_add_u16 %0, %0, %14", "v_pk_add_u16 %1, %1, %14", "v_pk_add_u16 %2, %2, %14", "v_pk_add_u16 %3, %3, %14", "v_pk_add_u16 %4, %4, %14", "v_pk_add_u16 %5, %5, %14", "v_pk_add_u16 %6, %6, %14", "v_pk_add_u16 %7, %7, %14"))
BODY(k_xad, S8("v_xad_u32 %0, %0, %14, %15", "v_xad_u32 %1, %1, %14, %15", "v_xad_u32 %2, %2, %14, %15", "v_xad_u32 %3, %3, %14, %15", "v_xad_u32 %4, %4, %14, %15", "v_xad_u32 %5, %5, %14, %15", "v_xad_u32 %6, %6, %14, %15", "v_xad_u32 %7, %7, %14, %15"))
BODY(k_dependent_and, S8("v_and_b32 %0, %14, %0", "v_and_b32 %0, %14, %0", "v_and_b32 %0, %14, %0", "v_and_b32 %0, %14, %0", "v_and_b32 %0, %14, %0", "v_and_b32 %0, %14, %0", "v_and_b32 %0, %14, %0", "v_and_b32 %0, %14, %0"))

BODY(k_add_u32, S8("v_add_u32 %0, %14, %0", "v_add_u32 %1, %14, %1", "v_add_u32 %2, %14, %2", "v_add_u32 %3, %14, %3", "v_add_u32 %4, %14, %4", "v_add_u32 %5, %14, %5", "v_add_u32 %6, %14, %6", "v_add_u32 %7, %14, %7"))
BODY(k_sub_u32, S8("v_sub_u32 %0, %14, %0", "v_sub_u32 %1, %14, %1", "v_sub_u32 %2, %14, %2", "v_sub_u32 %3, %14, %3", "v_sub_u32 %4, %14, %4", "v_sub_u32 %5, %14, %5", "v_sub_u32 %6, %14, %6", "v_sub_u32 %7, %14, %7"))
BODY(k_xor, S8("v_xor_b32 %0, %14, %0", "v_xor_b32 %1, %14, %1", "v_xor_b32 %2, %14, %2", "v_xor_b32 %3, %14, %3", "v_xor_b32 %4, %14, %4", "v_xor_b32 %5, %14, %5", "v_xor_b32 %6, %14, %6", "v_xor_b32 %7, %14, %7"))
BODY(k_not, S8("v_not_b32 %0, %0", "v_not_b32 %1, %1", "v_not_b32 %2, %2", "v_not_b32 %3, %3", "v_not_b32 %4, %4", "v_not_b32 %5, %5", "v_not_b32 %6, %6", "v_not_b32 %7, %7"))
BODY(k_and_e64, S8("v_and_b32_e64 %0, %14, %0", "v_and_b32_e64 %1, %14, %1", "v_and_b32_e64 %2, %14, %2", "v_and_b32_e64 %3, %14, %3", "v_and_b32_e64 %4, %14, %4", "v_and_b32_e64 %5, %14, %5", "v_and_b32_e64 %6, %14, %6", "v_and_b32_e64 %7, %14, %7"))
BODY(k_lshl_vgpr, S8("v_lshlrev_b32 %0, %14, %0", "v_lshlrev_b32 %1, %14, %1", "v_lshlrev_b32 %2, %14, %2", "v_lshlrev_b32 %3, %14, %3", "v_lshlrev_b32 %4, %14, %4", "v_lshlrev_b32 %5, %14, %5", "v_lshlrev_b32 %6, %14, %6", "v_lshlrev_b32 %7, %14, %7"))
BODY(k_max_u32, S8("v_max_u32 %0, %14, %0", "v_max_u32 %1, %14, %1", "v_max_u32 %2, %14, %2", "v_max_u32 %3, %14, %3", "v_max_u32 %4, %14, %4", "v_max_u32 %5, %14, %5", "v_max_u32 %6, %14, %6", "v_max_u32 %7, %14, %7"))
BODY(k_add_co, S8("v_add_co_u32 %0, vcc, %14, %0", "v_add_co_u32 %1, vcc, %14, %1", "v_add_co_u32 %2, vcc, %14, %2", "v_add_co_u32 %3, vcc, %14, %3", "v_add_co_u32 %4, vcc, %14, %4", "v_add_co_u32 %5, vcc, %14, %5", "v_add_co_u32 %6, vcc, %14, %6", "v_add_co_u32 %7, vcc, %14, %7"))
BODY(k_bitop3_sgpr, S8("v_bitop3_b32 %0, %16, %0, %15 bitop3:0x96", "v_bitop3_b32 %1, %16, %1, %15 bitop3:0x96", "v_bitop3_b32 %2, %16, %2, %15 bitop3:0x96", "v_bitop3_b32 %3, %16, %3, %15 bitop3:0x96", "v_bitop3_b32 %4, %16, %4, %15 bitop3:0x96", "v_bitop3_b32 %5, %16, %5, %15 bitop3:0x96", "v_bitop3_b32 %6, %16, %6, %15 bitop3:0x96", "v_bitop3_b32 %7, %16, %7, %15 bitop3:0x96"))
BODY(k_bitop3_inline, S8("v_bitop3_b32 %0, -1, %0, %15 bitop3:0x96", "v_bitop3_b32 %1, -1, %1, %15 bitop3:0x96", "v_bitop3_b32 %2, -1, %2, %15 bitop3:0x96", "v_bitop3_b32 %3, -1, %3, %15 bitop3:0x96", "v_bitop3_b32 %4, -1, %4, %15 bitop3:0x96", "v_bitop3_b32 %5, -1, %5, %15 bitop3:0x96", "v_bitop3_b32 %6, -1, %6, %15 bitop3:0x96", "v_bitop3_b32 %7, -1, %7, %15 bitop3:0x96"))
BODY(k_bitop3_same_src, S8("v_bitop3_b32 %0, %0, %0, %0 bitop3:0x96", "v_bitop3_b32 %1, %1, %1, %1 bitop3:0x96", "v_bitop3_b32 %2, %2, %2, %2 bitop3:0x96", "v_bitop3_b32 %3, %3, %3, %3 bitop3:0x96", "v_bitop3_b32 %4, %4, %4, %4 bitop3:0x96", "v_bitop3_b32 %5, %5, %5, %5 bitop3:0x96", "v_bitop3_b32 %6, %6, %6, %6 bitop3:0x96", "v_bitop3_b32 %7, %7, %7, %7 bitop3:0x96"))
BODY(k_mov_dpp, S8("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf"))
BODY(k_add3, S8("v_add3_u32 %0, %0, %14, %15", "v_add3_u32 %1, %1, %14, %15", "v_add3_u32 %2, %2, %14, %15", "v_add3_u32 %3, %3, %14, %15", "v_add3_u32 %4, %4, %14, %15", "v_add3_u32 %5, %5, %14, %15", "v_add3_u32 %6, %6, %14, %15", "v_add3_u32 %7, %7, %14, %15"))
BODY(k_lshl_add_u32, S8("v_lshl_add_u32 %0, %0, 2, %14", "v_lshl_add_u32 %1, %1, 2, %14", "v_lshl_add_u32 %2, %2, 2, %14", "v_lshl_add_u32 %3, %3, 2, %14", "v_lshl_add_u32 %4, %4, 2, %14", "v_lshl_add_u32 %5, %5, 2, %14", "v_lshl_add_u32 %6, %6, 2, %14", "v_lshl_add_u32 %7, %7, 2, %14"))
BODY(k_mad_u24, S8("v_mad_u32_u24 %0, %0, %14, %15", "v_mad_u32_u24 %1, %1, %14, %15", "v_mad_u32_u24 %2, %2, %14, %15", "v_mad_u32_u24 %3, %3, %14, %15", "v_mad_u32_u24 %4, %4, %14, %15", "v_mad_u32_u24 %5, %5, %14, %15", "v_mad_u32_u24 %6, %6, %14, %15", "v_mad_u32_u24 %7, %7, %14, %15"))
BODY(k_sad_u8, S8("v_sad_u8 %0, %0, %14, %15", "v_sad_u8 %1, %1, %14, %15", "v_sad_u8 %2, %2, %14, %15", "v_sad_u8 %3, %3, %14, %15", "v_sad_u8 %4, %4, %14, %15", "v_sad_u8 %5, %5, %14, %15", "v_sad_u8 %6, %6, %14, %15", "v_sad_u8 %7, %7, %14, %15"))
BODY(k_readfirstlane, S8("v_readfirstlane_b32 s20, %0", "v_readfirstlane_b32 s20, %1", "v_readfirstlane_b32 s20, %2", "v_readfirstlane_b32 s20, %3", "v_readfirstlane_b32 s20, %4", "v_readfirstlane_b32 s20, %5", "v_readfirstlane_b32 s20, %6", "v_readfirstlane_b32 s20, %7"))
BODY(k_mbcnt, S8("v_mbcnt_lo_u32_b32 %0, %14, %0", "v_mbcnt_lo_u32_b32 %1, %14, %1", "v_mbcnt_lo_u32_b32 %2, %14, %2", "v_mbcnt_lo_u32_b32 %3, %14, %3", "v_mbcnt_lo_u32_b32 %4, %14, %4", "v_mbcnt_lo_u32_b32 %5, %14, %5", "v_mbcnt_lo_u32_b32 %6, %14, %6", "v_mbcnt_lo_u32_b32 %7, %14, %7"))
BODY(k_cmp_sgpr_dst, S8("v_cmp_eq_u32_e64 s[20:21], %0, %14", "v_cmp_eq_u32_e64 s[20:21], %1, %14", "v_cmp_eq_u32_e64 s[20:21], %2, %14", "v_cmp_eq_u32_e64 s[20:21], %3, %14", "v_cmp_eq_u32_e64 s[20:21], %4, %14", "v_cmp_eq_u32_e64 s[20:21], %5, %14", "v_cmp_eq_u32_e64 s[20:21], %6, %14", "v_cmp_eq_u32_e64 s[20:21], %7, %14"))
BODY(k_cndmask_e64_vcc, S8("v_cndmask_b32_e64 %0, %0, %14, vcc", "v_cndmask_b32_e64 %1, %1, %14, vcc", "v_cndmask_b32_e64 %2, %2, %14, vcc", "v_cndmask_b32_e64 %3, %3, %14, vcc", "v_cndmask_b32_e64 %4, %4, %14, vcc", "v_cndmask_b32_e64 %5, %5, %14, vcc", "v_cndmask_b32_e64 %6, %6, %14, vcc", "v_cndmask_b32_e64 %7, %7, %14, vcc"))
BODY(k_mix_and_perm, S8("v_and_b32 %0, %14, %0\nv_perm_b32 %1, %14, %1, %15", "v_and_b32 %1, %14, %1\nv_perm_b32 %2, %14, %2, %15", "v_and_b32 %2, %14, %2\nv_perm_b32 %3, %14, %3, %15", "v_and_b32 %3, %14, %3\nv_perm_b32 %4, %14, %4, %15", "v_and_b32 %4, %14, %4\nv_perm_b32 %5, %14, %5, %15", "v_and_b32 %5, %14, %5\nv_perm_b32 %6, %14, %6, %15", "v_and_b32 %6, %14, %6\nv_perm_b32 %7, %14, %7, %15", "v_and_b32 %7, %14, %7\nv_perm_b32 %0, %14, %0, %15"))
BODY(k_mix_and_bitop3, S8("v_and_b32 %0, %14, %0\nv_bitop3_b32 %1, %14, %1, %15 bitop3:0x96", "v_and_b32 %1, %14, %1\nv_bitop3_b32 %2, %14, %2, %15 bitop3:0x96", "v_and_b32 %2, %14, %2\nv_bitop3_b32 %3, %14, %3, %15 bitop3:0x96", "v_and_b32 %3, %14, %3\nv_bitop3_b32 %4, %14, %4, %15 bitop3:0x96", "v_and_b32 %4, %14, %4\nv_bitop3_b32 %5, %14, %5, %15 bitop3:0x96", "v_and_b32 %5, %14, %5\nv_bitop3_b32 %6, %14, %6, %15 bitop3:0x96", "v_and_b32 %6, %14, %6\nv_bitop3_b32 %7, %14, %7, %15 bitop3:0x96", "v_and_b32 %7, %14, %7\nv_bitop3_b32 %0, %14, %0, %15 bitop3:0x96"))
BODY(k_mix_perm_bitop3, S8("v_perm_b32 %0, %14, %0, %15\nv_bitop3_b32 %1, %14, %1, %15 bitop3:0x96", "v_perm_b32 %1, %14, %1, %15\nv_bitop3_b32 %2, %14, %2, %15 bitop3:0x96", "v_perm_b32 %2, %14, %2, %15\nv_bitop3_b32 %3, %14, %3, %15 bitop3:0x96", "v_perm_b32 %3, %14, %3, %15\nv_bitop3_b32 %4, %14, %4, %15 bitop3:0x96", "v_perm_b32 %4, %14, %4, %15\nv_bitop3_b32 %5, %14, %5, %15 bitop3:0x96", "v_perm_b32 %5, %14, %5, %15\nv_bitop3_b32 %6, %14, %6, %15 bitop3:0x96", "v_perm_b32 %6, %14, %6, %15\nv_bitop3_b32 %7, %14, %7, %15 bitop3:0x96", "v_perm_b32 %7, %14, %7, %15\nv_bitop3_b32 %0, %14, %0, %15 bitop3:0x96"))
BODY(k_mix_and_salu, S8("v_and_b32 %0, %14, %0\ns_and_b64 %12, %12, %13", "v_and_b32 %1, %14, %1\ns_and_b64 %12, %12, %13", "v_and_b32 %2, %14, %2\ns_and_b64 %12, %12, %13", "v_and_b32 %3, %14, %3\ns_and_b64 %12, %12, %13", "v_and_b32 %4, %14, %4\ns_and_b64 %12, %12, %13", "v_and_b32 %5, %14, %5\ns_and_b64 %12, %12, %13", "v_and_b32 %6, %14, %6\ns_and_b64 %12, %12, %13", "v_and_b32 %7, %14, %7\ns_and_b64 %12, %12, %13"))
BODY(k_mix_perm_salu, S8("v_perm_b32 %0, %14, %0, %15\ns_and_b64 %12, %12, %13", "v_perm_b32 %1, %14, %1, %15\ns_and_b64 %12, %12, %13", "v_perm_b32 %2, %14, %2, %15\ns_and_b64 %12, %12, %13", "v_perm_b32 %3, %14, %3, %15\ns_and_b64 %12, %12, %13", "v_perm_b32 %4, %14, %4, %15\ns_and_b64 %12, %12, %13", "v_perm_b32 %5, %14, %5, %15\ns_and_b64 %12, %12, %13", "v_perm_b32 %6, %14, %6, %15\ns_and_b64 %12, %12, %13", "v_perm_b32 %7, %14, %7, %15\ns_and_b64 %12, %12, %13"))
BODY(k_lshr_b64_vgpr, S8("v_lshrrev_b64 %8, %14, %8\n; %0", "v_lshrrev_b64 %8, %14, %8\n; %1", "v_lshrrev_b64 %8, %14, %8\n; %2", "v_lshrrev_b64 %8, %14, %8\n; %3", "v_lshrrev_b64 %8, %14, %8\n; %4", "v_lshrrev_b64 %8, %14, %8\n; %5", "v_lshrrev_b64 %8, %14, %8\n; %6", "v_lshrrev_b64 %8, %14, %8\n; %7"))
BODY(k_and_b32_x2_as_b64, S8("v_and_b32 %0, %14, %0", "v_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2", "v_and_b32 %3, %14, %3", "v_and_b32 %4, %14, %4", "v_and_b32 %5, %14, %5", "v_and_b32 %6, %14, %6", "v_and_b32 %7, %14, %7"))
BODY(k_pat_ABAB, S8("v_and_b32 %0, %14, %0", "v_perm_b32 %4, %14, %4, %15", "v_and_b32 %1, %14, %1", "v_perm_b32 %5, %14, %5, %15", "v_and_b32 %2, %14, %2", "v_perm_b32 %6, %14, %6, %15", "v_and_b32 %3, %14, %3", "v_perm_b32 %7, %14, %7, %15"))
BODY(k_pat_AABB, S8("v_and_b32 %0, %14, %0", "v_and_b32 %1, %14, %1", "v_perm_b32 %4, %14, %4, %15", "v_perm_b32 %5, %14, %5, %15", "v_and_b32 %2, %14, %2", "v_and_b32 %3, %14, %3", "v_perm_b32 %6, %14, %6, %15", "v_perm_b32 %7, %14, %7, %15"))
BODY(k_pat_AAAABBBB, S8("v_and_b32 %0, %14, %0", "v_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2", "v_and_b32 %3, %14, %3", "v_perm_b32 %4, %14, %4, %15", "v_perm_b32 %5, %14, %5, %15", "v_perm_b32 %6, %14, %6, %15", "v_perm_b32 %7, %14, %7, %15"))
BODY(k_pat_AAAB, S8("v_and_b32 %0, %14, %0", "v_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2", "v_perm_b32 %4, %14, %4, %15", "v_and_b32 %3, %14, %3", "v_and_b32 %0, %14, %0", "v_and_b32 %1, %14, %1", "v_perm_b32 %5, %14, %5, %15"))
BODY(k_pat_7A1B, S8("v_and_b32 %0, %14, %0", "v_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2", "v_and_b32 %3, %14, %3", "v_and_b32 %0, %14, %0", "v_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2", "v_perm_b32 %4, %14, %4, %15"))
BODY(k_pat_7B1A, S8("v_perm_b32 %4, %14, %4, %15", "v_perm_b32 %5, %14, %5, %15", "v_perm_b32 %6, %14, %6, %15", "v_perm_b32 %7, %14, %7, %15", "v_perm_b32 %4, %14, %4, %15", "v_perm_b32 %5, %14, %5, %15", "v_perm_b32 %6, %14, %6, %15", "v_and_b32 %0, %14, %0"))
BODY(k_pat_15A1B, S8("v_and_b32 %0, %14, %0\nv_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2\nv_and_b32 %3, %14, %3", "v_and_b32 %0, %14, %0\nv_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2\nv_and_b32 %3, %14, %3", "v_and_b32 %0, %14, %0\nv_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2\nv_and_b32 %3, %14, %3", "v_and_b32 %0, %14, %0\nv_and_b32 %1, %14, %1", "v_and_b32 %2, %14, %2\nv_perm_b32 %4, %14, %4, %15"))

__global__ __launch_bounds__(256) void k_split_waves(unsigned long long* out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned b = seed * 29 + threadIdx.x, c = seed * 31 + 7;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (blockIdx.x & 1) {
        for (int i = 0; i < ITERS; ++i)
            asm volatile("v_perm_b32 %0, %8, %0, %9\nv_perm_b32 %1, %8, %1, %9\nv_perm_b32 %2, %8, %2, %9\nv_perm_b32 %3, %8, %3, %9\nv_perm_b32 %4, %8, %4, %9\nv_perm_b32 %5, %8, %5, %9\nv_perm_b32 %6, %8, %6, %9\nv_perm_b32 %7, %8, %7, %9\n"
                         "v_perm_b32 %0, %8, %0, %9\nv_perm_b32 %1, %8, %1, %9\nv_perm_b32 %2, %8, %2, %9\nv_perm_b32 %3, %8, %3, %9\nv_perm_b32 %4, %8, %4, %9\nv_perm_b32 %5, %8, %5, %9\nv_perm_b32 %6, %8, %6, %9\nv_perm_b32 %7, %8, %7, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else {
        for (int i = 0; i < ITERS; ++i)
            asm volatile("v_and_b32 %0, %8, %0\nv_and_b32 %1, %8, %1\nv_and_b32 %2, %8, %2\nv_and_b32 %3, %8, %3\nv_and_b32 %4, %8, %4\nv_and_b32 %5, %8, %5\nv_and_b32 %6, %8, %6\nv_and_b32 %7, %8, %7\n"
                         "v_and_b32 %0, %8, %0\nv_and_b32 %1, %8, %1\nv_and_b32 %2, %8, %2\nv_and_b32 %3, %8, %3\nv_and_b32 %4, %8, %4\nv_and_b32 %5, %8, %5\nv_and_b32 %6, %8, %6\nv_and_b32 %7, %8, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = (t1 - t0) | ((w1 - w0) << 40);
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = 1;
}
typedef void (*Kern)(unsigned long long*, unsigned);
struct Entry { const char* name; Kern k; };
static double per_slot(const char* n) { return strstr(n, "k_mix_") == n || !strcmp(n, "k_pat_15A1B") ? 2 : !strcmp(n, "k_split_waves") ? 0.5 : 1; }
#define E(N) {#N, N}
static const Entry TABLE[] = {E(k_and_vop2), E(k_and_literal), E(k_and_sgpr), E(k_lshr_imm), E(k_bfi), E(k_bfi_sgpr_mask), E(k_bitop3),
    E(k_bitop3_two_distinct), E(k_perm), E(k_perm_sgpr_sel), E(k_lshl_or), E(k_and_or), E(k_or3), E(k_alignbit), E(k_bfe), E(k_lshl_b64), E(k_lshr_b64),
    E(k_lshl_add_u64), E(k_mov_b64), E(k_mov_b32), E(k_cndmask_vcc), E(k_cndmask_sgpr), E(k_add_dpp_row_shr), E(k_add_dpp_row_bcast), E(k_bcnt), E(k_ffbl),
    E(k_cmp_vcc), E(k_cmp_sdwa), E(k_mul_u24), E(k_mul_lo_u32), E(k_readlane), E(k_salu_and_b64), E(k_valu_salu_mix), E(k_bpermute), E(k_pk_add_u16),
    E(k_xad), E(k_dependent_and), E(k_add_u32), E(k_sub_u32), E(k_xor), E(k_not), E(k_and_e64), E(k_lshl_vgpr), E(k_max_u32), E(k_add_co), E(k_bitop3_sgpr), E(k_bitop3_inline), E(k_bitop3_same_src), E(k_mov_dpp), E(k_add3), E(k_lshl_add_u32), E(k_mad_u24), E(k_sad_u8), E(k_readfirstlane), E(k_mbcnt), E(k_cmp_sgpr_dst), E(k_cndmask_e64_vcc), E(k_mix_and_perm), E(k_mix_and_bitop3), E(k_mix_perm_bitop3), E(k_mix_and_salu), E(k_mix_perm_salu), E(k_lshr_b64_vgpr), E(k_and_b32_x2_as_b64), E(k_pat_ABAB), E(k_pat_AABB), E(k_pat_AAAABBBB), E(k_pat_AAAB), E(k_pat_7A1B), E(k_pat_7B1A), E(k_pat_15A1B), E(k_split_waves)};

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    unsigned long long* d;
    hipMalloc(&d, (size_t)blocks * 4 * 8);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": %d, \"instructions_per_wave\": %d, \"cycles_per_instruction_per_simd\": {", p.gcnArchName, cus, waves_per_simd, ITERS * 32);
    bool first = true;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Entry& e : TABLE) {
        if (argc > 2 && !strstr(e.name, argv[2])) continue;
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 12346u);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, wall;
        for (auto v : h) { cyc.push_back((double)(v & ((1ull << 40) - 1))); wall.push_back((double)(v >> 40)); }
        std::sort(cyc.begin(), cyc.end());
        std::sort(wall.begin(), wall.end());
        const double n_instr = (double)ITERS * 32 * per_slot(e.name);
        const double med = cyc[cyc.size() / 2], medw = wall[wall.size() / 2];
        // cycles per instruction per SIMD from the waves' own clocks; the same from the kernel's wall time at the measured clock
        const double ghz = med / (medw * 10.0);  // wall_clock64: 100 MHz
        printf("%s\"%s\": {\"cyc\": %.2f, \"wall_cyc\": %.2f, \"ghz\": %.2f}", first ? "" : ", ", e.name + 2, med / (n_instr * waves_per_simd),
               ms * 1e6 * ghz / (n_instr * waves_per_simd), ghz);
        first = false;
    }
    printf("}}\n");
    return 0;
}
