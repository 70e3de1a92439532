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
// Two words per thread; the list slots of a workgroup are reserved with one atomic (ballot ranks inside the
// waves, LDS across them): per-lane -- or even per-wave -- atomics on the one counter are what this kernel's
// time would otherwise be.  The order of the list is irrelevant: integer sums.
__global__ __launch_bounds__(512) void k_newly(u64 *__restrict__ covered, const u64 *__restrict__ cols, u64 wp,
                                               const Pending pend,
                                               const IterState *__restrict__ st, unsigned *__restrict__ list_idx,
                                               u64 *__restrict__ list_val, unsigned *__restrict__ list_n)
{
    __shared__ unsigned wave_tot[8];
    __shared__ unsigned wg_base;
    if (st->done) return;
    const u64 *wcol = pending_column(st, cols, wp, pend);
    if (!wcol) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 below = (1ull << lane) - 1;
    const u64 pairs = wp / 2;  // wp is a multiple of 128
    for (u64 p0 = (u64)blockIdx.x * 512; p0 < pairs; p0 += (u64)gridDim.x * 512) {  // uniform
        const u64 w = (p0 + threadIdx.x) * 2;
        u64 x0 = 0, x1 = 0, c0 = 0, c1 = 0;
        if (w < wp) {
            const v2q c = *reinterpret_cast<const v2q *>(covered + w);
            c0 = c[0];
            c1 = c[1];
            x0 = __hip_atomic_load(&wcol[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & ~c0;
            x1 = __hip_atomic_load(&wcol[w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & ~c1;
        }
        const u64 b0 = __ballot(x0 != 0), b1 = __ballot(x1 != 0);
        const unsigned rank = __popcll(b0 & below) + __popcll(b1 & below);
        if (lane == 0) wave_tot[wave] = __popcll(b0) + __popcll(b1);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned t = 0;
            for (int i = 0; i < 8; ++i) t += wave_tot[i];
            wg_base = t ? atomicAdd(list_n, t) : 0;
        }
        __syncthreads();
        unsigned slot = wg_base + rank;
        for (int i = 0; i < wave; ++i) slot += wave_tot[i];
        if (x0) {
            list_idx[slot] = (unsigned)w;
            list_val[slot] = x0;
            ++slot;
        }
        if (x1) {
            list_idx[slot] = (unsigned)w + 1;
            list_val[slot] = x1;
        }
        if (x0 | x1) {
            v2q c;
            c[0] = c0 | x0;
            c[1] = c1 | x1;
            *reinterpret_cast<v2q *>(covered + w) = c;
        }
        __syncthreads();
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
                                              const IterState *__restrict__ st,
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
                dsum += af_fixed(a[__builtin_ctzll(x)]);
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

// ------------------------------------------------------------------------------------------------
// Word-interleaved copy of a chunk for the decremental iterations: rows_t[w * s_t + s] = cols[s * wp + w]
// (s_t = samples rounded up to 64, padding zero).  A listed word is then one contiguous run of s_t words --
// every byte fetched is used -- where the column layout pays a whole sector for each 8-byte gather.
// Costs a second copy of the chunk in HBM; built once per upload when there is room (host decides).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_interleave(const u64 *__restrict__ cols, u64 wp, unsigned n_local, u64 s_t,
                                                    u64 *__restrict__ rows_t)
{
    __shared__ u64 tile[64][65];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 w0 = (u64)blockIdx.x * 64;  // wp is a multiple of 64
    const u64 s0 = (u64)blockIdx.y * 64;
    for (int j = wave; j < 64; j += 4) tile[j][lane] = (s0 + j < n_local) ? cols[(s0 + j) * wp + w0 + lane] : 0;
    __syncthreads();
    for (int i = wave; i < 64; i += 4) rows_t[(w0 + i) * s_t + s0 + lane] = tile[lane][i];
}

// One thread per sample slot, a slice of the list per workgroup row: each listed word is a coalesced 512-byte
// read per wave; the running difference stays in a register and leaves with one atomic per (sample, slice).
// AF: the 64 fixed-point AF values of every listed word are staged in LDS first (one coalesced 256-byte read per
// word, shared by the 256 samples of the workgroup) -- per-bit gathers from global memory made the short lists
// of late iterations a chain of dependent loads (13 us floor against 4 us for the integer form).
#define UTM_DECR_STAGE 32  // listed words per LDS stage
template <bool AF>
__global__ __launch_bounds__(256) void k_decr_t(const u64 *__restrict__ rows_t, u64 s_t, const unsigned *__restrict__ afbits,
                                                const IterState *__restrict__ st,
                                                const unsigned char *__restrict__ state, unsigned n_local,
                                                const unsigned *__restrict__ list_idx, const u64 *__restrict__ list_val,
                                                const unsigned *__restrict__ list_n, u64 *__restrict__ cnt,
                                                i64 *__restrict__ afsum)
{
    __shared__ unsigned af_l[AF ? UTM_DECR_STAGE : 1][64];
    if (st->done) return;
    const unsigned n = *list_n;
    const unsigned per = (n + gridDim.y - 1) / gridDim.y;
    const unsigned e0 = blockIdx.y * per, e1 = min(n, e0 + per);
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (e0 >= e1) return;  // (workgroup-uniform)
    const bool live = s < s_t;
    const u64 *base = rows_t + (live ? s : 0);
    unsigned dec = 0;
    u64 dsum = 0;
    if (!AF) {
        if (!live) return;
#pragma unroll 4
        for (unsigned e = e0; e < e1; ++e) {
            const unsigned w = list_idx[e];  // uniform: scalar loads
            dec += __popcll(base[(u64)w * s_t] & list_val[e]);
        }
    } else {
        for (unsigned c0 = e0; c0 < e1; c0 += UTM_DECR_STAGE) {
            const unsigned m = min((unsigned)UTM_DECR_STAGE, e1 - c0);
            // the AF values of the stage and the first four row words are requested together: both hang off the list
            // only, and a short list is nothing but this chain of round trips
            unsigned v[UTM_DECR_STAGE * 64 / 256];
#pragma unroll
            for (int q = 0; q < UTM_DECR_STAGE * 64 / 256; ++q) {
                const unsigned i = threadIdx.x + q * 256;
                v[q] = i < m * 64 ? afbits[(u64)list_idx[c0 + (i >> 6)] * 64 + (i & 63)] : 0;
            }
            u64 x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = (live && (unsigned)k < m) ? base[(u64)list_idx[c0 + k] * s_t] & list_val[c0 + k] : 0;
            __syncthreads();  // the previous stage's readers are done
#pragma unroll
            for (int q = 0; q < UTM_DECR_STAGE * 64 / 256; ++q) (&af_l[0][0])[threadIdx.x + q * 256] = v[q];
            __syncthreads();
            for (unsigned j = 0; j < m; j += 4) {
                if (j) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        x[k] = (live && j + k < m) ? base[(u64)list_idx[c0 + j + k] * s_t] & list_val[c0 + j + k] : 0;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dec += __popcll(x[k]);
                    while (x[k]) {
                        dsum += af_fixed(af_l[j + k][__builtin_ctzll(x[k])]);
                        x[k] &= x[k] - 1;
                    }
                }
            }
        }
    }
    if (dec && s < n_local && state[s] == 1) {
        atomicAdd(&cnt[s], (u64)0 - (u64)dec);
        if (AF) atomicAdd(reinterpret_cast<u64 *>(&afsum[s]), (u64)0 - dsum);
    }
}
