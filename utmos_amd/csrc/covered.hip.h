// Covered-mask maintenance outside the scoring kernels.
#pragma once
#include "common.hip.h"

// covered |= pending winner column (used where the update is not fused into a scoring kernel)
__global__ __launch_bounds__(256) void k_apply_pending(u64 *__restrict__ covered, const u64 *__restrict__ cols, u64 wp,
                                                       const Pending pend,
                                                       const IterState *__restrict__ st, int in_loop)
{
    // inside the loop: like every loop kernel, nothing after the run has finished (launches enqueued ahead of the
    // stop must not read a peer's memory any more -- the owner may be tearing down); utm_get_covered passes 0
    if (in_loop && st->done) return;
    const u64 *wcol = pending_column(st, cols, wp, pend);
    if (!wcol) return;
    // the column may live in another process / on another GPU (hipIpc mapping): system-scope loads, so that
    // no cache of this GPU can answer with an older copy of those addresses
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256)
        covered[w] |= __hip_atomic_load(&wcol[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One-time copy of a peer shard's columns (hipIpc mapping, possibly across xGMI) into local memory.
__global__ __launch_bounds__(256) void k_copy_remote(const u64 *__restrict__ src, u64 *__restrict__ dst, u64 words)
{
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < words; w += (u64)gridDim.x * 256)
        dst[w] = __hip_atomic_load(&src[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// covered |= cols[col]  (utm_reset: samples that start out "used")
__global__ __launch_bounds__(256) void k_or_column(u64 *__restrict__ covered, const u64 *__restrict__ col, u64 wp)
{
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256) covered[w] |= col[w];
}

// ... the same for a peer's column read through its hipIpc mapping (system-scope loads, see k_apply_pending)
__global__ __launch_bounds__(256) void k_or_column_remote(u64 *__restrict__ covered, const u64 *__restrict__ col, u64 wp)
{
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256)
        covered[w] |= __hip_atomic_load(&col[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// RCCL exchange, root-free form: the winner's column on its owner, zeros everywhere else -- the sum over the ranks
// is then the column.  All chunks of the winner-column buffer in one launch (grid.y = chunk).
struct StageChunk {
    const u64 *cols;
    u64 wp, off;
};
__global__ __launch_bounds__(256) void k_stage_winner(u64 *__restrict__ wincol, const StageChunk *__restrict__ chunks,
                                                      const IterState *__restrict__ st, int rank, unsigned first)
{
    if (st->done || !st->prev_valid) return;
    const StageChunk ch = chunks[blockIdx.y];
    const bool mine = st->prev_rank == rank;
    const u64 *col = mine ? ch.cols + (u64)(st->prev_gidx - (i64)first) * ch.wp : nullptr;
    u64 *dst = wincol + ch.off;
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < ch.wp; w += (u64)gridDim.x * 256) dst[w] = mine ? col[w] : 0ull;
}
