// The integer greedy loop as ONE persistent launch per batch of iterations (DESIGN.md section 4
// "Persistent loop").  Replaces, for unweighted integer scores on one chunk, the per-iteration k_score_int<.., FUSED>
// launches: the per-iteration fixed costs of a launch (dispatch ramp, the tail after the last scoring wave, the kernel
// boundary: 4-5 us of a 30 us iteration at 1.1M x 2,504) are what keeps short scans off the HBM roofline.
//
// Roles.  Block 0 is the PICKER (mask / argmax / decide, select.py:43-53 and :93-112, as fused_pick does).  Every
// other block is a WORKER that owns ONE variant tile for the whole launch and one of Q slots of it; its four waves own
// the positions  i = 4*slot + wave + 4*Q*m  of act[] (strided, so the shares stay even while the selectable set
// shrinks).  A worker stages ~covered for its tile in LDS ONCE, at the start of the launch; from then on an iteration
// costs it one winner-tile read (`live &= ~winner`), never a covered read: the covered mask lives in the workers' LDS
// and goes back to memory when the launch ends.
//
// One iteration.  Workers stream their columns through the tile exactly like k_score_int (16-byte loads, 1 KiB per
// wave instruction, 8 in flight) and add every (position, tile) partial to the position's count word as
// `count + 2^40` (self-certifying words: the picker polls them until bits 40.. show all tiles; nobody signals).  The
// picker reduces, decides, and PUBLISHES the iteration in two tagged 8-byte words (sc1 stores):
//     W0 = epoch:24 | stop:1 | removed:1 | winner's local column:38      W1 = epoch:8 | best_pos:28 | moved sample:28
// (`moved` = the sample swap-removed into the winner's position of act[]).  Workers poll W0/W1 (one wave per
// workgroup), AND the winner's tile out of their LDS tile, patch their copy of act[] from the record and go on.
//
// What hides the hand-off: while a wave waits for the record it already holds the first 8 KiB of its next iteration in
// registers -- the sample at its first position is the one it had (unless the record says that very position changed,
// then it reloads) and column data never changes -- so the memory pipes stay full across the picker's critical path.
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility).  Everything that crosses workgroups inside the launch
// is an agent-scope atomic or an sc1 access on both sides: count words (atomic add / atomic load / atomic store),
// the record (atomic store / atomic load, tags in both words: a torn pair is re-read), act[] (the picker's atomic
// store; workers' atomic loads, patched with the newest record because that store may still be in flight), census
// counters.  Columns are immutable.  Count words alternate between two buffers by iteration parity, so the picker's
// clearing stores have a whole iteration to land before the words are added to again.
//
// Residency.  Waiting on another workgroup is only safe when it is running: the launch starts with a CENSUS (every
// block counts in on its XCD slot's counter; the picker waits -- bounded -- for all of them, then says go or abort).
// After an abort nothing has been touched; the host falls back to one launch per iteration and stops trying.
// Every later wait is bounded too and ends in st->xerror = 2 (reported as UTM_EHIP).
#pragma once
#include "common.hip.h"
#include "pick.hip.h"
#include "score_int.hip.h"

struct LoopSync {
    u64 pub[2];              // W0, W1
    u64 pad0[14];
    unsigned arrive[8 * 32];  // census counters, one per XCD slot (blockIdx & 7), 128 B apart
    unsigned go;              // the picker's census verdict: 1 go, 2 abort
    unsigned worker_timeout;  // a worker gave up waiting for a record (diagnostic)
    unsigned pad1[30];
};
static_assert(sizeof(LoopSync) % 16 == 0, "zeroed by one memset");

#define UTM_LOOP_EPOCH_MASK 0xFFFFFFull
#define UTM_LOOP_MAX_LOCAL (1u << 28)    // best_pos / moved are 28-bit fields
#define UTM_LOOP_CENSUS_SPINS (1u << 12) // x s_sleep(32): ~3.5 ms before a missing block aborts the launch
#define UTM_LOOP_WAIT_SPINS (1u << 25)   // x s_sleep(4): seconds before a worker gives up on a record

struct LoopRec {
    unsigned winner;    // local column of the winner
    unsigned best_pos;  // its position in act[] ...
    unsigned moved;     // ... and the sample that took it over
    int stop, removed, ok;
};

__device__ __forceinline__ LoopRec loop_read_record(const LoopSync *sync, unsigned epoch)
{
    const u64 w0 = __hip_atomic_load(&sync->pub[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 w1 = __hip_atomic_load(&sync->pub[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    LoopRec r;
    r.ok = (w0 >> 40) == (epoch & UTM_LOOP_EPOCH_MASK) && (w1 >> 56) == (epoch & 0xFFu);
    r.stop = (int)(w0 >> 39 & 1);
    r.removed = (int)(w0 >> 38 & 1);
    r.winner = (unsigned)(w0 & 0xFFFFFFFFull);
    r.best_pos = (unsigned)(w1 >> 28) & 0xFFFFFFFu;
    r.moved = (unsigned)w1 & 0xFFFFFFFu;
    return r;
}

// ------------------------------------------------------------------------------------------------ the picker
struct LoopPickLds {
    IntCand wbest[4];
    unsigned n_active;
    int stop, failed, pad;
};

__device__ __forceinline__ void loop_publish(LoopSync *sync, unsigned epoch, int stop, int removed, unsigned winner, unsigned best_pos,
                                             unsigned moved)
{
    const u64 w1 = ((u64)(epoch & 0xFFu) << 56) | ((u64)(best_pos & 0xFFFFFFFu) << 28) | (u64)(moved & 0xFFFFFFFu);
    const u64 w0 = ((u64)(epoch & UTM_LOOP_EPOCH_MASK) << 40) | ((u64)(stop ? 1 : 0) << 39) | ((u64)(removed ? 1 : 0) << 38) | (u64)winner;
    __hip_atomic_store(&sync->pub[1], w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sync->pub[0], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void loop_picker(const PickArgs &a, LoopSync *sync, u64 *cnt0, u64 *cnt1, unsigned n_tiles, unsigned n_blocks,
                                            int k_batch, LoopPickLds *L)
{
    IterState *st = a.st;
    const int lane = threadIdx.x & 63;
    // census: every block of the grid (this one included) has counted in => every block is resident
    if (threadIdx.x < 64) {
        if (lane == 0) __hip_atomic_fetch_add(&sync->arrive[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = false;
        for (unsigned spin = 0; spin < UTM_LOOP_CENSUS_SPINS; ++spin) {
            const unsigned v = lane < 8 ? __hip_atomic_load(&sync->arrive[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (wave_sum_u32(v) >= n_blocks) { ok = true; break; }
            __builtin_amdgcn_s_sleep(32);
        }
        if (lane == 0) {
            if (!ok) st->xerror = 3;  // not every block became resident: nothing was touched, the host takes the launch-per-iteration path
            __hip_atomic_store(&sync->go, ok ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L->stop = ok ? 0 : 1;
        }
    }
    __syncthreads();
    if (L->stop) return;

    unsigned n_active = st->n_active;  // (written before the launch)
    i64 iter = 0, tot = 0, n_active_total = 0;
    unsigned last_act = 0;
    if (threadIdx.x == 0) {
        iter = st->iter;
        tot = st->tot;
        n_active_total = st->n_active_total;
        last_act = n_active ? __hip_atomic_load(&a.act[n_active - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    const u64 count_mask = (1ull << UTM_ARRIVAL_SHIFT) - 1;
    u64 t_pub = wall_clock64(), t_len = 0;  // when the last record went out; how long the iteration before it took (100 MHz ticks)
    for (int k = 0;; ++k) {
        u64 *cnt = (k & 1) ? cnt1 : cnt0;
        // Stay off the memory system while the iteration is certainly still running: polling all launch long costs the
        // streaming waves bandwidth.  Iterations shrink slowly, so 3/4 of the last one's length is a safe nap.
        if (t_len > 40) {
            const u64 until = t_pub + t_len * 3 / 4;
            while ((u64)wall_clock64() < until) __builtin_amdgcn_s_sleep(16);
        }
        IntCand best{0, 0xFFFFFFFFu, 0};
        int failed = 0;
        for (unsigned base = 0; base < n_active && !failed; base += 256 * UTM_FUSED_E) {
            const unsigned i0 = base + threadIdx.x;
            unsigned s[UTM_FUSED_E];
            unsigned need = 0;
#pragma unroll
            for (int e = 0; e < UTM_FUSED_E; ++e) {  // (act[] and cnt[] carry UTM_PICK_PAD spare entries)
                s[e] = __hip_atomic_load(&a.act[i0 + e * 256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (i0 + e * 256 < n_active) need |= 1u << e;
            }
            u64 *words = cnt + i0;
            for (unsigned spin = 0; need; ++spin) {
                u64 v[UTM_FUSED_E];
#pragma unroll
                for (int e = 0; e < UTM_FUSED_E; ++e)
                    v[e] = __hip_atomic_fetch_add(words + e * 256, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int e = 0; e < UTM_FUSED_E; ++e) {
                    const bool fin = (need >> e & 1) && (v[e] >> UTM_ARRIVAL_SHIFT) == n_tiles;
                    if (fin) __hip_atomic_store(words + e * 256, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (next used two iterations on)
                    const IntCand cand{fin ? (v[e] & count_mask) : 0ull, fin ? s[e] : 0xFFFFFFFFu, i0 + e * 256};
                    if (better_int(cand, best)) best = cand;
                    need &= ~((fin ? 1u : 0u) << e);
                }
                if (need) {
                    if (spin > UTM_FUSED_SPINS) { failed = 1; break; }
                    __builtin_amdgcn_s_sleep(UTM_FUSED_SLEEP);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            IntCand other;
            other.cnt = __shfl_xor(best.cnt, o, 64);
            other.s = __shfl_xor(best.s, o, 64);
            other.pos = __shfl_xor(best.pos, o, 64);
            if (better_int(other, best)) best = other;
        }
        if (threadIdx.x == 0) L->failed = 0;
        __syncthreads();
        if (lane == 0) L->wbest[threadIdx.x >> 6] = best;
        if (failed) L->failed = 1;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned epoch = (unsigned)k + 1;
            int stop = 0;
            if (L->failed) {
                st->xerror = 2;  // a partial count never arrived (a logic error, not a data condition)
                st->done = 1;
                stop = 1;
                loop_publish(sync, epoch, 1, 0, 0, 0, 0);
            } else {
                for (int w4 = 1; w4 < 4; ++w4)
                    if (better_int(L->wbest[w4], best)) best = L->wbest[w4];
                // decide_single, on the loop state this thread carries in registers
                if (n_active == 0 || best.cnt == 0) {  // (unweighted integer scores are never negative)
                    st->done = 1;  // (None, None): no row (select.py:51-52, :93-96)
                    a.res_idx[iter] = -1;
                    stop = 1;
                    loop_publish(sync, epoch, 1, 0, 0, 0, 0);
                } else {
                    const unsigned moved = last_act;
                    const int finished = tot + (i64)best.cnt >= a.n_var_total;  // "Ran out of new variants" (select.py:110-112)
                    stop = finished || k + 1 >= k_batch;
                    // the record first: everything below is bookkeeping nobody inside the launch waits for
                    loop_publish(sync, epoch, stop, 1, best.s, best.pos, moved);
                    __hip_atomic_store(&a.act[best.pos], moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.res_idx[iter] = (i64)a.first + best.s;
                    a.res_new[iter] = (i64)best.cnt;
                    a.res_score[iter] = (double)best.cnt;
                    a.state[best.s] = 0;  // sample_mask[use_sample] = 0 (select.py:100)
                    iter += 1;
                    tot += (i64)best.cnt;
                    n_active_total -= 1;
                    n_active -= 1;
                    st->prev_valid = 1;
                    st->prev_rank = 0;
                    st->prev_gidx = (i64)a.first + best.s;
                    st->prev_local = (int)best.s;
                    st->best_pos = best.pos;
                    if (finished) st->done = 1;
                    Rec *rc = rec_of(a, a.rank);
                    rc->score = (double)best.cnt;
                    rc->idx = (i64)a.first + best.s;
                    rc->new_count = (i64)best.cnt;
                    if (!stop) {
                        // next iteration's swap-remove candidate (our own store to act[] may still be in flight)
                        last_act = n_active == 0            ? 0u
                                   : n_active - 1 == best.pos ? moved
                                                              : __hip_atomic_load(&a.act[n_active - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (stop) {
                st->iter = iter;
                st->tot = tot;
                st->n_active_total = n_active_total;
                st->n_active = n_active;
            }
            L->n_active = n_active;
            L->stop = stop;
        }
        // this thread's clearing stores (and thread 0's act[] store) have landed before anybody reads those words again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u64 now = wall_clock64();
        t_len = now - t_pub;
        t_pub = now;
        __syncthreads();
        if (L->stop) return;
        n_active = L->n_active;
    }
}

// ------------------------------------------------------------------------------------------------ the launch
// Grid = 1 + 8 * per_xcd blocks of 256 threads: block 0 picks; block b > 0 sits (placement observed, speed only) on XCD
// slot x = b & 7 and is that slot's j-th worker; unit u = x * per_xcd + j = (tile, slot) tile-major, so an XCD's workers
// share few tiles and a winner's tile is fetched into ONE L2.  drop_iter: test hook (0 = off) -- the first worker
// withholds one partial count in that iteration of the launch, so that the picker's bounded wait runs out.
template <int STEPS, bool NT>
__global__ __launch_bounds__(256) void k_loop_int(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp, const Pending pend,
                                                  IterState *__restrict__ st, unsigned *__restrict__ act, u64 *__restrict__ cnt0,
                                                  u64 *__restrict__ cnt1, unsigned q_slots, unsigned n_units, unsigned per_xcd, int k_batch,
                                                  LoopSync *__restrict__ sync, const PickArgs pa, int drop_iter)
{
    __shared__ v4u live[STEPS * 64];  // ~covered of this worker's tile, for the whole launch
    __shared__ LoopRec rec_lds;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    typedef unsigned v8u __attribute__((ext_vector_type(8)));
    v8u raw;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(st) : "memory");
    const IterHead head = __builtin_bit_cast(IterHead, raw);
    if (head.done) return;  // (uniform over the launch, the picker included: nobody waits for anybody)
    constexpr unsigned TILE_WORDS = STEPS * UTM_STEP_WORDS;
    const unsigned n_tiles = (unsigned)((wp + TILE_WORDS - 1) / TILE_WORDS);
    if (blockIdx.x == 0) {
        loop_picker(pa, sync, cnt0, cnt1, n_tiles, gridDim.x, k_batch, reinterpret_cast<LoopPickLds *>(&live[0]));
        return;
    }
    const unsigned xs = blockIdx.x & 7;
    const unsigned j = (blockIdx.x >> 3) - (xs == 0 ? 1u : 0u);
    const unsigned u = xs * per_xcd + j;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&sync->arrive[xs * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (u >= n_units) return;  // (padding of the grid to whole XCD rounds: counted in, nothing to do)
    const unsigned tile = u / q_slots, slot = u % q_slots;
    const u64 w0 = (u64)tile * TILE_WORDS;
    const u64 left = (wp - w0) / UTM_STEP_WORDS;
    const int nsteps = left < (u64)STEPS ? (int)left : STEPS;
    const bool full = nsteps == STEPS;
    constexpr int U = STEPS < 8 ? STEPS : 8;
#define UTM_COL_LOAD(ptr) (NT ? __builtin_nontemporal_load(ptr) : *(ptr))
    const v4u zero4 = {0, 0, 0, 0};
#define UTM_BATCH_LOAD(J0)                                                                 \
    if (full) {                                                                            \
        _Pragma("unroll") for (int q = 0; q < U; ++q) x[q] = UTM_COL_LOAD(p + ((J0) + q) * 64); \
    } else {                                                                               \
        _Pragma("unroll") for (int q = 0; q < U; ++q)                                      \
        {                                                                                  \
            const int step = (J0) + q;                                                     \
            x[q] = UTM_COL_LOAD(p + (step < nsteps ? step : nsteps - 1) * 64);             \
        }                                                                                  \
    }
#define UTM_BATCH_COUNT(J0)                                                                \
    if (full) {                                                                            \
        _Pragma("unroll") for (int q = 0; q < U; ++q)                                      \
        {                                                                                  \
            const v4u b = x[q] & live[((J0) + q) * 64 + lane];                             \
            acc += __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);                  \
        }                                                                                  \
    } else {                                                                               \
        _Pragma("unroll") for (int q = 0; q < U; ++q)                                      \
        {                                                                                  \
            const int step = (J0) + q;                                                     \
            const v4u b = (step < nsteps ? x[q] : zero4) & live[(step < nsteps ? step : 0) * 64 + lane]; \
            acc += __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);                  \
        }                                                                                  \
    }
    // this wave's positions of act[]: first, first + stride, ...
    const unsigned first = slot * 4 + wave, stride = q_slots * 4;
    unsigned n_act = head.n_active;
    const unsigned act_cap = pa.n_local + UTM_PICK_PAD;  // (entries that may be read ahead of the bounds that say whether they count)
    unsigned s_first = first < act_cap ? act[first] : 0u;  // (act[] is as the host / the last launch left it: plain loads)
    unsigned s_next = first + stride < act_cap ? act[first + stride] : 0u;
    const v4u *p = reinterpret_cast<const v4u *>(cols + (u64)s_first * wp + w0) + lane;
    v4u x[U];
    if (first < n_act) { UTM_BATCH_LOAD(0) }

    // stage ~(covered | pending winner) once; slot 0 brings covered itself up to date
    {
        v4u *cv = reinterpret_cast<v4u *>(covered + w0);
        const u64 *wcol = pend.fuse ? pending_column(&head, cols, wp, pend) : nullptr;
        const v4u *wc = wcol ? reinterpret_cast<const v4u *>(wcol + w0) : nullptr;
        for (int k = threadIdx.x; k < nsteps * 64; k += 256) {
            v4u c = cv[k];
            if (wc) c |= wc[k];
            live[k] = ~c;
        }
    }
    // the census verdict
    if (threadIdx.x == 0) {
        unsigned g = 0;
        for (unsigned spin = 0; spin < UTM_LOOP_WAIT_SPINS; ++spin) {
            g = __hip_atomic_load(&sync->go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g) break;
            __builtin_amdgcn_s_sleep(8);
        }
        rec_lds.ok = g == 1;
    }
    __syncthreads();
    if (!rec_lds.ok) return;  // abort (or no verdict): nothing has been written

    unsigned patch_pos = 0xFFFFFFFFu, patch_s = 0;  // the newest record's change to act[] (its store may still be in flight)
#define UTM_ACT_LOAD(i) ((i) == patch_pos ? patch_s : __hip_atomic_load(&act[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    for (int k = 0;; ++k) {
        u64 *cnt = (k & 1) ? cnt1 : cnt0;
        unsigned i = first, s = s_first;
        while (i < n_act) {
            unsigned acc = 0;
#pragma unroll 1
            for (int j0 = 0; j0 < nsteps; j0 += U) {
                if (j0) { UTM_BATCH_LOAD(j0) }
                UTM_BATCH_COUNT(j0)
            }
            const unsigned done_i = i;
            i += stride;
            s = s_next;
            if (i < n_act) {
                s_next = i + stride < n_act ? UTM_ACT_LOAD(i + stride) : 0u;
                p = reinterpret_cast<const v4u *>(cols + (u64)s * wp + w0) + lane;
                UTM_BATCH_LOAD(0)
            }
            acc = wave_sum_u32(acc);
            const bool drop = drop_iter && k + 1 == drop_iter && blockIdx.x == 1 && done_i == first;
            if (lane == 0 && !drop) atomicAdd(&cnt[done_i], (u64)acc + (1ull << UTM_ARRIVAL_SHIFT));
        }
        // ahead of the record: the first 8 KiB of the next iteration.  The sample at `first` stays where it is unless
        // the record names that very position (then it is loaded again below); a position that drops out costs one
        // wasted batch.
        p = reinterpret_cast<const v4u *>(cols + (u64)s_first * wp + w0) + lane;
        if (first < n_act) { UTM_BATCH_LOAD(0) }
        if (first + stride < n_act) s_next = UTM_ACT_LOAD(first + stride);

        const unsigned epoch = (unsigned)k + 1;
        if (wave == 0) {
            LoopRec r;
            r.ok = 0;
            for (unsigned spin = 0; spin < UTM_LOOP_WAIT_SPINS; ++spin) {
                r = loop_read_record(sync, epoch);
                if (r.ok) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (lane == 0) {
                if (!r.ok) {
                    r.stop = 1;
                    __hip_atomic_store(&sync->worker_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                rec_lds = r;
            }
        }
        __syncthreads();
        const LoopRec r = rec_lds;
        if (r.stop) break;  // (uniform) the launch ends here: the winner stays pending, exactly as after a k_score_int launch
        // covered |= winner, in LDS: live &= ~winner's tile
        {
            const v4u *wc = reinterpret_cast<const v4u *>(cols + (u64)r.winner * wp + w0);
            for (int kk = threadIdx.x; kk < nsteps * 64; kk += 256) live[kk] &= ~wc[kk];
        }
        if (r.removed) {
            n_act -= 1;
            patch_pos = r.best_pos;
            patch_s = r.moved;
            if (first == r.best_pos) {
                s_first = r.moved;
                p = reinterpret_cast<const v4u *>(cols + (u64)s_first * wp + w0) + lane;
                if (first < n_act) { UTM_BATCH_LOAD(0) }
            }
            if (first + stride == r.best_pos) s_next = r.moved;
        }
        __syncthreads();  // the tile is whole again (and rec_lds may be rewritten)
    }
    // the tile's covered words go back to memory (the pending winner is NOT in them: the next launch folds it in)
    if (slot == 0) {
        v4u *cv = reinterpret_cast<v4u *>(covered + w0);
        for (int k = threadIdx.x; k < nsteps * 64; k += 256) cv[k] = ~live[k];
    }
#undef UTM_ACT_LOAD
#undef UTM_COL_LOAD
#undef UTM_BATCH_LOAD
#undef UTM_BATCH_COUNT
}
