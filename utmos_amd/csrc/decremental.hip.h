// Decremental scoring (k_newly / k_newly_mask / k_decr).
#pragma once
#include "score_af.hip.h"

// ------------------------------------------------------------------------------------------------
// Decremental scoring (optional, reported separately from the brute-force roofline).
// Coverage only grows, so count_{k+1}[s] = count_k[s] - popcount(col_s & newly_k) with
// newly_k = winner_k & ~covered_k, and only the words where newly_k != 0 have to be touched.
// k_newly applies the pending winner to `covered` and compacts those words into a list; k_decr lets
// one wave per selectable sample gather its own words at the listed positions and subtract.  Integer
// arithmetic on both sides: the counts (and the fixed-point AF sums) stay exactly what a full
// re-scoring would give.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_newly(u64 *__restrict__ covered, const u64 *__restrict__ cols, u64 wp,
                                               const Pending pend,
                                               const IterState *__restrict__ st, unsigned *__restrict__ list_idx,
                                               u64 *__restrict__ list_val, unsigned *__restrict__ list_n)
{
    if (st->done) return;
    const u64 *wcol = pending_column(st, cols, wp, pend);
    if (!wcol) return;
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256) {
        const u64 c = covered[w];
        const u64 x = __hip_atomic_load(&wcol[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & ~c;
        if (x) {
            const unsigned slot = atomicAdd(list_n, 1u);
            list_idx[slot] = (unsigned)w;
            list_val[slot] = x;
            covered[w] = c | x;
        }
    }
}

// Dense form for the streamed delta scoring: mask[w] = the bits of word w the pending winner newly covers
// (0 where none); covered is updated in the same pass.
__global__ __launch_bounds__(256) void k_newly_mask(u64 *__restrict__ covered, const u64 *__restrict__ cols, u64 wp,
                                                    const Pending pend, const IterState *__restrict__ st, u64 *__restrict__ mask)
{
    if (st->done) return;
    const u64 *wcol = pending_column(st, cols, wp, pend);
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256) {
        u64 x = 0;
        if (wcol) {
            const u64 c = covered[w];
            x = __hip_atomic_load(&wcol[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & ~c;
            if (x) covered[w] = c | x;
        }
        mask[w] = x;
    }
}

template <bool AF>
__global__ __launch_bounds__(256) void k_decr(const u64 *__restrict__ cols, u64 wp, const unsigned *__restrict__ afbits,
                                              int e_base, const IterState *__restrict__ st,
                                              const unsigned *__restrict__ act, const unsigned *__restrict__ list_idx,
                                              const u64 *__restrict__ list_val, const unsigned *__restrict__ list_n,
                                              u64 *__restrict__ cnt, i64 *__restrict__ afsum)
{
    if (st->done) return;
    const unsigned n = *list_n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned i = blockIdx.x * 4 + wave;
    if (i >= st->n_active || blockIdx.y * 64 >= n) return;
    const unsigned s = act[i];
    const u64 *col = cols + (u64)s * wp;
    unsigned dec = 0;
    u64 dsum = 0;
    for (unsigned e = blockIdx.y * 64 + lane; e < n; e += gridDim.y * 64) {
        const unsigned w = list_idx[e];
        u64 x = col[w] & list_val[e];
        dec += __popcll(x);
        if (AF) {
            const unsigned *a = afbits + (u64)w * 64;
            while (x) {
                dsum += af_fixed(a[__builtin_ctzll(x)], e_base);
                x &= x - 1;
            }
        }
    }
    const unsigned total = wave_sum_u32(dec);
    if (total) {  // wave uniform
        const i64 tsum = AF ? wave_sum_u63(dsum) : 0;
        if (lane == 0) {
            atomicAdd(&cnt[s], (u64)0 - (u64)total);
            if (AF) atomicAdd(reinterpret_cast<u64 *>(&afsum[s]), (u64)0 - (u64)tsum);
        }
    }
}
