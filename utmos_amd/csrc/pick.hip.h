// K2: mask / weight / argmax, the decision of an iteration, and the exchange between shards.
#pragma once
#include "common.hip.h"

// ------------------------------------------------------------------------------------------------
// K2: mask / weight / argmax (select.py:43-53) over the selectable local samples and, when this is
// the only shard, the decision and bookkeeping of greedy_select (select.py:93-112).  One workgroup.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better(const Cand &a, const Cand &b)
{  // np.argmax: highest score, first (lowest) index on ties
    return a.val > b.val || (a.val == b.val && a.gidx < b.gidx);
}
__device__ __forceinline__ Cand shfl_cand(const Cand &c, int o)
{
    Cand r;
    r.val = __shfl_xor(c.val, o, 64);
    r.gidx = __shfl_xor(c.gidx, o, 64);
    r.cnt = __shfl_xor(c.cnt, o, 64);
    r.pos = __shfl_xor(c.pos, o, 64);
    return r;
}



// Runs in ONE thread.  Same inputs on every shard => same decision on every shard.
__device__ void decide(const PickArgs &a)
{
    IterState *st = a.st;
    Cand best{-__builtin_inf(), INT64_MAX, 0, 0};
    int best_rank = -1;
    for (int r = 0; r < a.n_ranks; ++r) {
        const Rec *rc = rec_of(a, r);
        if (rc->idx < 0) continue;
        Cand c{rc->score, rc->idx, rc->new_count, 0};
        if (best_rank < 0 || better(c, best)) { best = c; best_rank = r; }
    }
    const i64 k = st->iter;
    // argmax runs over ALL samples in the reference; non-selectable ones hold 0 (select.py:43), so a
    // negative best only wins when no such sample exists.
    const bool zero_elsewhere = st->n_active_total < (i64)a.n_total;
    if (best_rank < 0 || best.val == 0.0 || (best.val < 0.0 && zero_elsewhere)) {
        st->done = 1;  // (None, None): no row for this iteration (select.py:51-52, :93-96)
        a.res_idx[k] = -1;
        return;
    }
    a.res_idx[k] = best.gidx;
    a.res_new[k] = best.cnt;
    a.res_score[k] = best.val;
    st->iter = k + 1;
    st->tot += best.cnt;
    st->n_active_total -= 1;
    st->prev_valid = 1;
    st->prev_rank = best_rank;
    st->prev_gidx = best.gidx;
    if (best.gidx >= (i64)a.first && best.gidx < (i64)a.first + a.n_local) {
        const unsigned loc = (unsigned)(best.gidx - a.first);
        a.state[loc] = 0;  // sample_mask[use_sample] = 0 (select.py:100)
        const unsigned n = st->n_active;
        a.act[st->best_pos] = a.act[n - 1];
        st->n_active = n - 1;
        st->prev_local = a.remote_winner_test ? -1 : (int)loc;
    } else {
        st->prev_local = -1;
    }
    if (st->tot >= a.n_var_total) st->done = 1;  // "Ran out of new variants" (select.py:110-112)
}

// The same decision for the only shard (MODE 0), on values thread 0 loaded while the scan was in flight: nothing
// but stores are left on the critical path.
struct Preloaded {
    i64 iter, tot, n_active_total;
    unsigned last_act;  // act[n_active - 1]: moves into the winner's position
};
__device__ __forceinline__ void decide_single(const PickArgs &a, const Cand &best, unsigned n_active, const Preloaded &p)
{
    IterState *st = a.st;
    const i64 k = p.iter;
    const bool zero_elsewhere = p.n_active_total < (i64)a.n_total;
    if (n_active == 0 || best.val == 0.0 || (best.val < 0.0 && zero_elsewhere)) {
        st->done = 1;
        a.res_idx[k] = -1;
        return;
    }
    a.res_idx[k] = best.gidx;
    a.res_new[k] = best.cnt;
    a.res_score[k] = best.val;
    st->iter = k + 1;
    st->tot = p.tot + best.cnt;
    st->n_active_total = p.n_active_total - 1;
    st->prev_valid = 1;
    st->prev_rank = 0;
    st->prev_gidx = best.gidx;
    const unsigned loc = (unsigned)(best.gidx - a.first);
    a.state[loc] = 0;
    a.act[best.pos] = p.last_act;
    st->n_active = n_active - 1;
    st->prev_local = (int)loc;
    if (p.tot + best.cnt >= a.n_var_total) st->done = 1;
}

// Device-side exchange, receiving end: wait (bounded) until every shard's record of this exchange has
// landed in the local mailbox, copy them into the record slots, decide.  One lane per source shard.
#define UTM_MBOX_SPINS (1u << 24)  // x s_sleep(16): several seconds before a missing shard is declared lost (UTM_MBOX_SPINS_LOG2)
__device__ __forceinline__ bool mbox_wait(const Mailbox *slot, u64 expected, Rec *out, unsigned spins = UTM_MBOX_SPINS)
{
    for (unsigned spin = 0; spin < spins; ++spin) {
        if (__hip_atomic_load(&slot->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == expected) {
            out->score = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const u64 *>(&slot->score), __ATOMIC_RELAXED,
                                                                      __HIP_MEMORY_SCOPE_SYSTEM));
            out->idx = (i64)__hip_atomic_load(reinterpret_cast<const u64 *>(&slot->idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            out->new_count = (i64)__hip_atomic_load(reinterpret_cast<const u64 *>(&slot->new_count), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_SYSTEM);
            return true;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    return false;
}

// Device-side exchange, both ends, then the decision -- called by EVERY thread of a workgroup (>= n_ranks threads)
// once thread 0 has written this shard's record into recs[rank] and the caller has passed a barrier.  One lane per
// destination shard stores the record into that shard's mailbox slot [seq & 1][my rank] -- payload first, sequence
// number last (release, system scope) -- and collects that shard's record of the same exchange from the local mailbox.
__device__ __forceinline__ void mailbox_exchange_and_decide(const PickArgs &a, int *late /* LDS, set to 0 before the barrier */)
{
    IterState *st = a.st;
    if ((int)threadIdx.x < a.n_ranks) {
        const Rec mine = *rec_of(a, a.rank);
        const u64 seq = st->xseq + 1;
        Mailbox *dst = a.peer_mbox[threadIdx.x] + (seq & 1) * a.n_ranks + a.rank;
        if (!(a.test_mute && st->iter + 1 == (i64)a.test_mute)) {  // (test hook: a shard that posts nothing)
            __hip_atomic_store(reinterpret_cast<u64 *>(&dst->score), __builtin_bit_cast(u64, mine.score), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<u64 *>(&dst->idx), (u64)mine.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<u64 *>(&dst->new_count), (u64)mine.new_count, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&dst->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const Mailbox *slot = a.mbox + (seq & 1) * a.n_ranks + threadIdx.x;
        if (!mbox_wait(slot, seq, rec_of(a, threadIdx.x), a.mbox_spins)) *late = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (*late) {
            st->xerror = 1;  // a shard went away: end the loop, the host reports it
            st->done = 1;
        } else {
            st->xseq += 1;
            decide(a);
        }
    }
}

// The pick of an unweighted integer-score iteration, run INSIDE the scoring launch by its extra last block
// (k_score_int<.., FUSED>): FUSED = 1 on the only shard (same decision as k_pick<0>), FUSED = 2 on a shard whose
// records travel through the device mailboxes (same exchange and decision as k_pick<2>).  The count words (indexed by position in
// act[]) are being written by the scoring workgroups' agent-scope atomics while this runs; each partial carries
// 2^40 on top of its count, so a word is final once its upper bits equal the number of variant tiles.  Words are
// read with returning agent-scope atomics (add 0) -- the coherent read of a word other CUs update atomically --
// E consecutive words per thread in flight, re-read until complete, then cleared for the next iteration.  act[]
// and the loop state are static during the launch and are loaded before the wait.  Scores are the counts
// themselves here (no weights: the launcher keeps weighted runs on k_pick), compared as integers: count
// descending, global index ascending = np.argmax's first maximum.  The wait is bounded: a launch whose partials
// never arrive (a logic error, not a data condition) ends the loop with st->xerror = 2.
#ifndef UTM_FUSED_E
#define UTM_FUSED_E 4  // (5, 8, 10 in flight per thread measured 1-2 % slower at 2,504 samples: more registers, lower occupancy)
#endif
#ifndef UTM_FUSED_SLEEP
#define UTM_FUSED_SLEEP 4
#endif
#define UTM_PICK_PAD 4096  // spare entries behind act[] and cnt[] (>= 256 * UTM_FUSED_E)
#define UTM_FUSED_SPINS (1u << 22)
struct IntCand {
    u64 cnt;
    unsigned s, pos;  // local sample, position in act[]
};
__device__ __forceinline__ bool better_int(const IntCand &a, const IntCand &b)
{
    return a.cnt > b.cnt || (a.cnt == b.cnt && a.s < b.s);
}
template <int FUSED>
__device__ __forceinline__ void fused_pick(const PickArgs &a, unsigned n_tiles, IntCand *fbest /* LDS, 4 entries + flag */)
{
    IterState *st = a.st;
    const unsigned n_active = st->n_active;
    Preloaded pre{0, 0, 0, 0};
    if (threadIdx.x == 0) {
        pre.iter = st->iter;
        pre.tot = st->tot;
        pre.n_active_total = st->n_active_total;
        pre.last_act = n_active ? a.act[n_active - 1] : 0;
    }
    const u64 count_mask = (1ull << 40) - 1;
    IntCand best{0, 0xFFFFFFFFu, 0};  // (no selectable sample: stays like this, n_active == 0 decides)
    int failed = 0;
    for (unsigned base = 0; base < n_active && !failed; base += 256 * UTM_FUSED_E) {
        const unsigned i0 = base + threadIdx.x;  // a wave instruction reads 64 consecutive words (4 lines); E of them in flight
        unsigned s[UTM_FUSED_E];
        unsigned need = 0;
#pragma unroll
        for (int e = 0; e < UTM_FUSED_E; ++e) {  // (act[] and cnt[] are allocated with UTM_PICK_PAD spare entries: no bounds branches)
            s[e] = a.act[i0 + e * 256];
            if (i0 + e * 256 < n_active) need |= 1u << e;
        }
        u64 *words = a.cnt + i0;
        for (unsigned spin = 0; need; ++spin) {
            // (LLVM lowers the idempotent fetch_add to an agent-scope atomic 64-bit load: global_load_dwordx2 sc1)
            u64 v[UTM_FUSED_E];
#pragma unroll
            for (int e = 0; e < UTM_FUSED_E; ++e)
                v[e] = __hip_atomic_fetch_add(words + e * 256, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // a complete word is consumed at once: nothing but the `need` mask and the running best lives across rounds
#pragma unroll
            for (int e = 0; e < UTM_FUSED_E; ++e) {
                const bool fin = (need >> e & 1) && (v[e] >> 40) == n_tiles;
                if (fin) __hip_atomic_store(words + e * 256, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
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
    int *any_failed = reinterpret_cast<int *>(fbest + 4);
    if (threadIdx.x == 0) *any_failed = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) fbest[threadIdx.x >> 6] = best;
    if (failed) *any_failed = 1;
    __syncthreads();
    int *late = any_failed + 1;
    if (threadIdx.x == 0) {
        *late = 0;
        if (*any_failed) {
            st->xerror = 2;
            st->done = 1;
        } else {
            for (int w4 = 1; w4 < 4; ++w4)
                if (better_int(fbest[w4], best)) best = fbest[w4];
            const Cand win{(double)best.cnt, (i64)a.first + best.s, (i64)best.cnt, best.pos};
            Rec *rc = rec_of(a, a.rank);
            rc->score = n_active ? win.val : 0.0;
            rc->idx = n_active ? win.gidx : -1;
            rc->new_count = n_active ? win.cnt : 0;
            st->best_pos = win.pos;
            if (FUSED == 1) decide_single(a, win, n_active, pre);
        }
    }
    if (FUSED == 2) {
        __syncthreads();
        if (*any_failed) return;  // (uniform: an internal error ends the loop on this shard; the peers time out on its record)
        mailbox_exchange_and_decide(a, late);
    }
}

// MODE 0: single shard -- pick and decide.  1: write this shard's record into its exchange slot.
// 2: as 1, and post the record into every shard's mailbox (device-side exchange over P2P mappings).
// COHERENT_VALS: the candidates' chain results were written during THIS launch by other workgroups (the pick riding at
// the end of k_chain): they are read with agent-scope atomic loads instead of plain ones.
template <int MODE, bool COHERENT_VALS = false>
__device__ __forceinline__ void pick_body(const PickArgs &a)
{
    __shared__ Cand wbest[16];
    __shared__ int late;
    IterState *st = a.st;
    if (st->done) return;
    const unsigned n_active = st->n_active;
    Preloaded pre{0, 0, 0, 0};
    if (MODE == 0 && threadIdx.x == 0) {
        pre.iter = st->iter;
        pre.tot = st->tot;
        pre.n_active_total = st->n_active_total;
        pre.last_act = n_active ? a.act[n_active - 1] : 0;
    }
    // where this iteration's scores come from
    //   0 integer counts | 1 exact fixed-point AF sums | 2 sequential float64 scores of every sample
    //   3 the candidates' sequential float64 scores (k_chain)
    int src = a.afsum ? 1 : a.fscore ? 2 : 0;
    if (a.cand && st->need_chain) src = st->cand_overflow ? 2 : 3;
    Cand best{-__builtin_inf(), INT64_MAX, 0, 0};
    if (src == 3) {
        const unsigned n_cand = (unsigned)st->n_cand;
        for (unsigned b = threadIdx.x; b < n_cand; b += blockDim.x) {
            const unsigned s = a.cand->samp[b];
            double v = COHERENT_VALS ? __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const u64 *>(&a.cand->val[b]), __ATOMIC_RELAXED,
                                                                                    __HIP_MEMORY_SCOPE_AGENT))
                                     : a.cand->val[b];
            if (a.weights) v *= a.weights[a.first + s];
            const Cand cand{v, (i64)a.first + s, a.cand->cnt[b], a.cand->pos[b]};
            if (better(cand, best)) best = cand;
        }
    }
    // (candidates' chains with persistent accumulators: nothing to read, clear or mirror per sample)
    const bool scan_samples = !(src == 3 && !a.zero_after && !a.cnt_mirror && !a.afsum_mirror);
    for (unsigned i = threadIdx.x; scan_samples && i < n_active; i += blockDim.x) {
        const unsigned s = a.act[i];
        const unsigned ci = a.cnt_by_pos ? i : s;
        const u64 c = a.cnt[ci];
        if (a.zero_after) a.cnt[ci] = 0;  // ready for the next iteration's atomics
        if (a.cnt_mirror) a.cnt_mirror[s] = c;
        double v = (double)c;
        if (a.afsum) {
            const i64 q = a.afsum[s];
            if (a.zero_after) a.afsum[s] = 0;
            if (a.afsum_mirror) a.afsum_mirror[s] = q;
            v = (double)q * a.af_scale;  // exact: q < 2^53 whenever this value is used, and the scale is a power of two
            if (a.known_cnt && a.known_cnt[s] == c) v = a.known_val[s];  // ... or a chain's sum for this very count
        }
        if (src == 3) continue;
        if (src == 2) v = a.fscore[s];
        if (a.weights) v *= a.weights[a.first + s];
        const Cand cand{v, (i64)a.first + s, (i64)c, i};
        if (better(cand, best)) best = cand;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const Cand other = shfl_cand(best, o);
        if (better(other, best)) best = other;
    }
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (better(wbest[w], best)) best = wbest[w];
        Rec *rc = rec_of(a, a.rank);
        rc->score = n_active ? best.val : 0.0;
        rc->idx = n_active ? best.gidx : -1;
        rc->new_count = n_active ? best.cnt : 0;
        st->best_pos = best.pos;
        if (MODE == 2) late = 0;
        if (a.list_n) {
            u64 n_l = 0;
            for (int c = 0; c < a.n_chunks; ++c) {
                n_l += a.list_n[c];
                a.list_n[c] = 0;
            }
            st->decr_entries += n_l;
            st->decr_gathers += n_l * n_active;
        }
        if (MODE == 0) decide_single(a, best, n_active, pre);
    }
    if (MODE == 2) {
        __syncthreads();
        mailbox_exchange_and_decide(a, &late);
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_pick(PickArgs a)
{
    pick_body<MODE>(a);
}

__global__ void k_decide(PickArgs a)
{
    if (a.st->done) return;
    if (threadIdx.x == 0) decide(a);
}

// Mailbox self-test (utm_p2p_selftest): one full post + wait round with a recognisable payload.
__global__ __launch_bounds__(64) void k_mbox_ping(Mailbox *mbox, Mailbox *const *peer_mbox, int rank, int n_ranks, u64 seq, int *ok)
{
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if ((int)threadIdx.x < n_ranks) {
        Mailbox *dst = peer_mbox[threadIdx.x] + (seq & 1) * n_ranks + rank;
        __hip_atomic_store(reinterpret_cast<u64 *>(&dst->idx), (u64)(1000 * seq + rank), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&dst->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        Rec got;
        const Mailbox *slot = mbox + (seq & 1) * n_ranks + threadIdx.x;
        if (!mbox_wait(slot, seq, &got) || got.idx != (i64)(1000 * seq + threadIdx.x)) bad = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && bad) *ok = 0;
}

// Sum of the selectable samples' counts (AF byte accounting, see IterState::cnt_sum).  One workgroup.
__global__ __launch_bounds__(1024) void k_count_sum(IterState *st, const unsigned *__restrict__ act, const u64 *__restrict__ cnt, int base)
{
    __shared__ u64 part[16];
    u64 sum = 0;
    const unsigned n = st->n_active;
    for (unsigned i = threadIdx.x; i < n; i += 1024) sum += cnt[act[i]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 total = 0;
        for (int w = 0; w < 16; ++w) total += part[w];
        if (base) st->cnt_sum_base = total;
        else st->cnt_sum = total;
    }
}

// Final per-sample scores of the pending iteration (utm_peek_scores): mask, scale, weight.
__global__ __launch_bounds__(256) void k_final_scores(PickArgs a, i64 *__restrict__ counts_out, double *__restrict__ scores_out)
{
    const unsigned s = blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_local) return;
    const bool usable = a.state[s] == 1;
    const u64 c = usable ? a.cnt[s] : 0;
    double v = 0.0;
    if (usable) v = a.afsum ? (double)a.afsum[s] * a.af_scale : a.fscore ? a.fscore[s] : (double)c;
    if (a.weights) v *= a.weights[a.first + s];
    counts_out[s] = (i64)c;
    scores_out[s] = v;
}
