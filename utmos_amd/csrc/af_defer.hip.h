// Deferred exact AF scores: the reported float64 score of an unambiguous winner, off the iteration's critical path.
#pragma once
#include "af_verify.hip.h"

// ------------------------------------------------------------------------------------------------
// On the only shard the verified-parallel AF loop needs the reference's sequential float64 sum of a winner for ONE
// thing when the intervals already single it out (k_cand: one candidate): the score it reports.  That sum runs over
// exactly the variants the winner newly covers (select.py:38-40: rows an earlier winner captured are skipped) --
// which is the mask the NEXT iteration's delta pass makes anyway while it stages its tiles (k_score_afs,
// covered_out form).  So that pass also logs the mask, one slot per iteration (UTM_DEFER_SLOTS of them, a whole
// column each), and after a batch of iterations four launches finish all of the batch's scores together:
//   k_defer_count   set bits per (row, segment of 4096 words)
//   k_defer_scan    exclusive prefix over (row, segment) -> where each run of addends starts in ONE pool (the masks
//                   of different iterations are disjoint, so the pool needs one slot per variant, whatever the batch)
//   k_defer_fill    the AF values of the set bits, ascending, into the pool
//   k_defer_chain   one workgroup per row: the parallel form of the reference's chain (chain_parallel) over the
//                   row's contiguous addends; score = sum (* weight, select.py:45-47) into the result row.
// Per iteration that is ~1 us instead of three dependent launches (k_cand -> k_chain_fill -> k_chain, ~35 us).
// Iterations whose intervals do NOT settle the winner still run the chains on the spot, as before.
// ------------------------------------------------------------------------------------------------
struct DeferArgs {
    const SeqChunk *chunks;  // af / w per chunk
    const ChainSeg *segs;
    int n_segs;
    const u64 *log;          // [slots][col_words]: row r's mask in slot r % slots
    int slots;
    u64 col_words;
    unsigned *counts;        // [n_rows][n_segs]
    u64 *offs;               // [n_rows * n_segs + 1]
    double *vals;            // the pool
    i64 row0;                // result rows [row0, row0 + n_rows)
    int n_rows;
};

__device__ __forceinline__ const u64 *defer_mask(const DeferArgs &d, int row_i, const ChainSeg &sg)
{
    return d.log + (u64)((d.row0 + row_i) % d.slots) * d.col_words + sg.off;
}

// The mask of the LAST row of a run: no later pass made it (utm_run's end).  Does not touch covered.
__global__ __launch_bounds__(256) void k_newly_log(const u64 *__restrict__ covered, const u64 *__restrict__ cols, u64 wp,
                                                   const Pending pend, const IterState *__restrict__ st, u64 *__restrict__ log_chunk,
                                                   u64 col_words, int slots)
{
    const u64 *wcol = pending_column(st, cols, wp, pend);
    if (!wcol || st->iter < 1) return;
    u64 *out = log_chunk + (u64)((st->iter - 1) % slots) * col_words;
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256) out[w] = wcol[w] & ~covered[w];
}

__global__ __launch_bounds__(256) void k_defer_count(DeferArgs d)
{
    __shared__ unsigned part[4];
    const ChainSeg sg = d.segs[blockIdx.x];
    const u64 w_end = d.chunks[sg.chunk].w;
    const u64 *mask = defer_mask(d, blockIdx.y, sg);
    unsigned n = 0;
#pragma unroll
    for (int k = 0; k < UTM_SEG_WORDS / 256; ++k) {
        const u64 w = (u64)k * 256 + threadIdx.x;
        n += sg.w0 + w < w_end ? __popcll(mask[w]) : 0;
    }
    n = wave_sum_u32(n);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) d.counts[(size_t)blockIdx.y * d.n_segs + blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// One workgroup; thread t owns a contiguous run of the (row, segment) counts.
__global__ __launch_bounds__(1024) void k_defer_scan(DeferArgs d)
{
    __shared__ u64 wtot[16];
    const size_t n = (size_t)d.n_rows * d.n_segs;
    const size_t per = (n + 1023) / 1024;
    const size_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    u64 mine = 0;
    for (size_t i = lo; i < hi; ++i) mine += d.counts[i];
    u64 incl = mine;  // inclusive scan over the wave, then over the 16 waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 up = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    u64 off = incl - mine;
    for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) off += wtot[w];
    for (size_t i = lo; i < hi; ++i) {
        d.offs[i] = off;
        off += d.counts[i];
    }
    if (threadIdx.x == 1023) d.offs[n] = off;  // (the last thread's run ends at n, possibly empty)
}

template <typename AF_T>
__global__ __launch_bounds__(1024) void k_defer_fill(DeferArgs d)
{
    __shared__ unsigned wtot[16];
    const size_t slot = (size_t)blockIdx.y * d.n_segs + blockIdx.x;
    if (d.counts[slot] == 0) return;
    const ChainSeg sg = d.segs[blockIdx.x];
    const SeqChunk ch = d.chunks[sg.chunk];
    const u64 *mask = defer_mask(d, blockIdx.y, sg);
    const AF_T *af = static_cast<const AF_T *>(ch.af);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u64 x[4];
    unsigned n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = sg.w0 + tid * 4 + k < ch.w ? mask[tid * 4 + k] : 0;
        n += __popcll(x[k]);
    }
    const unsigned incl = wave_scan_incl_u32(n);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int k = 0; k < wave; ++k) woff += wtot[k];
    double *out = d.vals + d.offs[slot] + (woff + incl - n);
    const u64 v0 = (sg.w0 + (u64)tid * 4) * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u64 y = x[k];
        while (y) {
            const int b = __builtin_ctzll(y);
            y &= y - 1;
            *out++ = (double)af[v0 + k * 64 + b];
        }
    }
}

__global__ __launch_bounds__(1024) void k_defer_chain(DeferArgs d, const i64 *__restrict__ res_idx, double *__restrict__ res_score,
                                                      const double *__restrict__ weights)
{
    __shared__ ParScratch sc;
    const u64 lo = d.offs[(size_t)blockIdx.x * d.n_segs], hi = d.offs[(size_t)(blockIdx.x + 1) * d.n_segs];
    const FlatAddends src{d.vals + lo, (unsigned)(hi - lo)};
    double sum = chain_parallel(src, sc);
    if (threadIdx.x == 0) {
        const i64 row = d.row0 + blockIdx.x;
        if (weights) sum *= weights[res_idx[row]];
        res_score[row] = sum;
    }
}
