// K1, integer scores: the bandwidth-bound bitset reduction (and the XCD-aware block -> tile map).
#pragma once
#include "common.hip.h"
#include "pick.hip.h"

// XCD-aware block -> (tile, group) map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and
// b + 8 share one; each XCD has its own L2).  The (tile, group) units are numbered tile-major and cut into
// 8 equal contiguous ranges, one per XCD: an XCD walks whole variant tiles (all sample groups of a tile one
// after the other), so a tile's ~covered words and the pending winner's words are fetched into ONE L2 and
// reused there, while every XCD still gets the same number of units.  Grid = 8 * ceil(units / 8); surplus
// blocks return.  Only speed depends on the placement, never results.
__device__ __forceinline__ bool tile_of_block(u64 wp, unsigned tile_words, unsigned n_groups, unsigned &tile, unsigned &grp)
{
    const unsigned n_tiles = (unsigned)((wp + tile_words - 1) / tile_words);
    const unsigned units = n_tiles * n_groups, per_xcd = (units + 7) / 8;
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const unsigned u = xcd * per_xcd + j;
    tile = u / n_groups;
    grp = u % n_groups;
    return j < per_xcd && u < units;
}

// ------------------------------------------------------------------------------------------------
// K1: integer scores.  count[s] += popcount(col_s & ~covered) over one tile of the variant axis,
// for one group of selectable samples (calculate_scores' row loop, select.py:37-41, as a bitset
// reduction).  Grid = tiles x groups.  The workgroup stages ~covered for its tile in LDS once
// (fusing the pending `covered |= winner` of the previous iteration, select.py:100), then each of
// its 4 waves streams whole samples through that tile: one global_load_dwordx4 (1 KiB per wave)
// + one ds_read_b128 + 4x(v_and, v_bcnt) per step, a wave reduction and ONE 64-bit atomic per
// (sample, tile).  Integer adds: exact and order independent.
// ------------------------------------------------------------------------------------------------
// FUSED (1: the only shard, 2: a shard exchanging through the device mailboxes): the pick of the iteration (mask /
// argmax / [exchange] / decide, pick.hip.h) runs inside the scoring launch,
// in one extra workgroup (the grid's last block) -- no k_pick launch, no kernel boundary between scoring and pick.
// Nobody signals and nobody waits on the scoring side: every (sample, tile) partial is added to the sample's
// count word as `count + 2^40`, so bits 40.. of a count word say how many tile partials it holds, and the picker
// simply reads the words (returning agent-scope atomics) until each shows all of the launch's tiles
// (fused_pick).  The scoring workgroups end with their fire-and-forget atomic, as in the plain form.
#define UTM_ARRIVAL_SHIFT 40
#ifndef UTM_SCORE_WAVES
#define UTM_SCORE_WAVES(STEPS) 1  // (forcing 8 waves/SIMD on the small tiles spilled the picker's registers: slower)
#endif
template <int STEPS, bool NT, int FUSED>
__global__ __launch_bounds__(256, UTM_SCORE_WAVES(STEPS)) void k_score_int(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp,
                                                   const Pending pend,
                                                   const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                   u64 *__restrict__ cnt, unsigned group_size, unsigned n_groups,
                                                   unsigned by_pos, const PickArgs pa)
{
    __shared__ v4u live[STEPS * 64];  // ~covered of this tile, STEPS KiB
    // Everything this workgroup needs from memory before it can request its columns is asked for in ONE go -- the loop
    // state (done flag, selectable count, where the pending winner's column is) and this wave's first two entries of
    // act[] (read ahead of the bounds that say whether they count: act[] is padded) -- instead of flag -> count ->
    // index -> column, four dependent round trips at the start of every workgroup and of every launch.
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned tile, grp;
    const bool has_unit = tile_of_block(wp, STEPS * UTM_STEP_WORDS, n_groups, tile, grp);
    const unsigned lo = grp * group_size;
    unsigned i = lo + wave;
    const bool in_act = has_unit && i + 4 < pa.n_local + UTM_PICK_PAD;  // (inside the padded array)
    unsigned s = in_act ? act[i] : 0;
    unsigned s_next = in_act ? act[i + 4] : 0;
    // (inline asm: left to itself the compiler fetches the flag first and sinks the other fields' loads below the branch
    // on it, one round trip each)
    typedef unsigned v8u __attribute__((ext_vector_type(8)));
    v8u raw;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(st) : "memory");
    const IterHead head = __builtin_bit_cast(IterHead, raw);
    const unsigned n_active = head.n_active;
    if (head.done) return;  // (uniform over the launch: only a launch's pick ever sets it)
    // the picker: one block behind the scoring grid (dispatched last: it starts polling when the launch is nearly over;
    // as the first block it polled all launch long and cost 2 %)
    if (FUSED && blockIdx.x == gridDim.x - 1) {
        const unsigned n_tiles = (unsigned)((wp + STEPS * UTM_STEP_WORDS - 1) / (STEPS * UTM_STEP_WORDS));
        fused_pick<FUSED>(pa, n_tiles, reinterpret_cast<IntCand *>(&live[0]));
        return;
    }
    if (!has_unit) return;
    const u64 w0 = (u64)tile * STEPS * UTM_STEP_WORDS;
    const u64 left = (wp - w0) / UTM_STEP_WORDS;
    const int nsteps = left < (u64)STEPS ? (int)left : STEPS;
    const bool full = nsteps == STEPS;
    const unsigned hi = lo + group_size < n_active ? lo + group_size : n_active;
    constexpr int U = STEPS < 8 ? STEPS : 8;  // loads in flight per wave: U KiB
#define UTM_COL_LOAD(ptr) (NT ? __builtin_nontemporal_load(ptr) : *(ptr))
    // One batch = U KiB of a column.  Full tiles use immediate offsets; the last tile of a column may be shorter
    // than STEPS KiB: there the missing steps re-read the tile's last KiB and count as zero (wave-uniform clamp and
    // select), so it is batched too -- on its own code path, which keeps the clamps out of the full tiles' registers.
    const v4u zero4 = {0, 0, 0, 0};
#define UTM_BATCH_LOAD(J0)                                                                 \
    if (full) {                                                                            \
        _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = UTM_COL_LOAD(p + ((J0) + u) * 64); \
    } else {                                                                               \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                      \
        {                                                                                  \
            const int step = (J0) + u;                                                     \
            x[u] = UTM_COL_LOAD(p + (step < nsteps ? step : nsteps - 1) * 64);             \
        }                                                                                  \
    }
#define UTM_BATCH_COUNT(J0)                                                                \
    if (full) {                                                                            \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                      \
        {                                                                                  \
            const v4u b = x[u] & live[((J0) + u) * 64 + lane];                             \
            acc += __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);                  \
        }                                                                                  \
    } else {                                                                               \
        _Pragma("unroll") for (int u = 0; u < U; ++u)                                      \
        {                                                                                  \
            const int step = (J0) + u;                                                     \
            const v4u b = (step < nsteps ? x[u] : zero4) & live[(step < nsteps ? step : 0) * 64 + lane]; \
            acc += __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);                  \
        }                                                                                  \
    }

    // Software pipeline: the first U KiB of this wave's first sample are requested BEFORE the tile is staged (the
    // two do not depend on each other: a short-lived workgroup would otherwise spend half its life waiting for
    // the tile, then again for its columns), and a sample's successor (index from act[], then its first U KiB) is
    // requested before the sample's reduction and atomic.
    const v4u *p = reinterpret_cast<const v4u *>(cols + (u64)s * wp + w0) + lane;
    v4u x[U];
    if (i < hi) { UTM_BATCH_LOAD(0) }

    v4u *cv = reinterpret_cast<v4u *>(covered + w0);
    const u64 *wcol = pend.fuse ? pending_column(&head, cols, wp, pend) : nullptr;
    const v4u *wc = wcol ? reinterpret_cast<const v4u *>(wcol + w0) : nullptr;
    for (int k = threadIdx.x; k < nsteps * 64; k += 256) {
        v4u c = cv[k];
        if (wc) {
            c |= wc[k];
            // every group of this tile computes the same words; group 0 stores them.  A racing reader
            // sees old or new words and ORs the winner in itself, so either is right.
            if (grp == 0) cv[k] = c;
        }
        live[k] = ~c;
    }
    __syncthreads();

    while (i < hi) {
        unsigned acc = 0;
#pragma unroll 1
        for (int j0 = 0; j0 < nsteps; j0 += U) {
            if (j0) { UTM_BATCH_LOAD(j0) }
            UTM_BATCH_COUNT(j0)
        }
        const unsigned done_i = i, done_s = s;
        i += 4;
        s = s_next;
        if (i < hi) {
            s_next = i + 4 < hi ? act[i + 4] : 0;
            p = reinterpret_cast<const v4u *>(cols + (u64)s * wp + w0) + lane;
            UTM_BATCH_LOAD(0)
        }
        acc = wave_sum_u32(acc);
        // counts are kept by position in act[] (by_pos) where the pick reads them by position too: one
        // dependent load less on its critical path.  FUSED: every partial arrives, a zero one too.
        if (FUSED) {
            const bool drop = pa.test_drop && blockIdx.x == 0 && done_i == lo;  // (test hook: see PickArgs::test_drop)
            if (lane == 0 && !drop) atomicAdd(&cnt[done_i], (u64)acc + (1ull << UTM_ARRIVAL_SHIFT));
        } else if (lane == 0 && acc) {
            atomicAdd(&cnt[by_pos ? done_i : done_s], (u64)acc);
        }
    }
#undef UTM_COL_LOAD
#undef UTM_BATCH_LOAD
#undef UTM_BATCH_COUNT
}
