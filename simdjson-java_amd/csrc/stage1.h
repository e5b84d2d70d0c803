// stage1.h -- internal declarations shared by the HIP kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sjmi.h"

namespace sjmi {

// written by the kernel into the workspace header; mirrors sjmi_stage1_result in include/sjmi.h
struct Stage1Result {
    unsigned long long count;  // number of structural indexes (BitIndexes.writeIdx)
    uint32_t status;           // SJMI_ST_* bits
    uint32_t reserved;
};
static_assert(sizeof(Stage1Result) == sizeof(sjmi_stage1_result), "ABI struct mismatch");

// ablation switches for performance experiments only (results are NOT valid with any of them set)
constexpr uint32_t DBG_NO_WRITE = 1, DBG_NO_LOOKBACK = 2;

constexpr uint32_t STAGE_CAP = 1024;  // indexes staged in LDS per wave and round (4 KiB)

// workspace layout (zeroed by one hipMemsetAsync per launch)
constexpr size_t WS_RESULT_OFFSET = 0;        // Stage1Result
constexpr size_t WS_TICKET_OFFSET = 64;       // u32 tile ticket, alone in its 64-byte line
constexpr size_t WS_TILE_STATE_OFFSET = 128;  // u64 per tile

size_t stage1_workspace_bytes(uint64_t len, int steps);
int stage1_pick_steps(uint64_t len);
// ev_start/ev_stop (optional) bracket the kernel only (not the workspace memset)
hipError_t stage1_launch(const uint8_t* d_buf, uint64_t len, uint32_t* d_out, uint64_t out_cap, void* d_ws, int steps,
                         hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t dbg = 0);
hipError_t transpose_selftest_launch(const uint32_t* d_words, uint32_t nblocks, uint32_t* d_mismatches,
                                     hipStream_t stream);

}  // namespace sjmi
