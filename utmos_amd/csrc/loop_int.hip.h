// The greedy loop as ONE persistent launch per batch of up to 256 iterations (DESIGN.md section 4 "The
// loop as ONE launch per batch of iterations").  Replaces, for integer scores (weighted or not) and the exact phase of
// float32 AF on one chunk / one shard, the per-iteration k_score_* launches: the per-iteration fixed costs of a launch
// (dispatch ramp, the tail after the last scoring wave, the kernel boundary: 4-5 us of a 30 us iteration at 1.1M x 2,504)
// are what keeps short scans off the HBM roofline.  (select.py:91-112 is the loop this runs; :24-53 one iteration.)
//
// Roles.  Block 0 is the PICKER (mask / weights / argmax / decide, select.py:43-53 and :93-112, as fused_pick does).  Every
// other block (512 threads) is a WORKER that owns ONE variant tile (1, 2, 4 or 8 batches of 8 KiB) for the whole launch and
// one of Q slots of it; a tile's workers are spread over all XCDs and dispatch ages (tile = worker % tiles).  A worker
// stages ~covered for its tile in LDS ONCE, at the start of the launch; from then on an iteration costs it one winner-tile
// read (`live &= ~winner`), never a covered read: the covered mask lives in the workers' LDS and goes back to memory when
// the launch ends.
//
// One iteration.  A wave's work is a stream of 8 KiB batches, two in flight (x0 / x1): the batches of its two static
// positions of act[] (first = 8*slot + wave, second = first + 8Q), then of positions it CLAIMS, one ticket at a time, from
// the tile's eight counters (three counter sets in turn, iteration % 3, so that an iteration's first two tickets can be
// taken while the one before still runs).  Every (position, tile) partial is added to the position's count word as
// `count + 2^40` (self-certifying words: the picker polls them until bits 40.. show all tiles; nobody signals).  The
// picker reduces, decides, and PUBLISHES the iteration in two tagged 8-byte words (sc1 stores):
//     W0 = epoch:24 | stop:1 | removed:1 | winner's local column:38      W1 = epoch:8 | best_pos:28 | moved sample:28
// (`moved` = the sample swap-removed into the winner's position of act[]).  Workers poll W0/W1 (one wave per workgroup),
// AND the winner's tile out of their LDS tile, patch their view of act[] from the record and go on.
//
// What hides the hand-off: while a wave waits for the record it holds the first two batches of its next iteration in
// registers -- the samples at its static positions are the ones it had (unless the record names those very positions,
// then they are loaded again) and column data never changes.  The first goes out as soon as the wave runs dry (it fills
// the iteration's tail), the second is held back until ~4 us before the record is due by the wave's own clock (iterations
// shrink smoothly), so that it is in flight while the picker reduces and the tile is updated.
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility).  Everything that crosses workgroups inside the launch
// is an agent-scope atomic or an sc1 access on both sides: count words (atomic add / atomic load / atomic store),
// the record (atomic store / atomic load, tags in both words: a torn pair is re-read), act[] (the picker's atomic
// store; workers' atomic loads, patched with the newest record because that store may still be in flight), claim and
// census counters.  Columns are immutable.  Count words alternate between two buffers by iteration parity, so the
// picker's clearing stores have a whole iteration to land before the words are added to again.
//
// AF form (template parameter AF, below): the exact float32-AF phase as delta iterations on per-sample accumulators -- a
// second LDS tile holds what the pending winner newly covers, the words carry decreases.
//
// Residency.  Waiting on another workgroup is only safe when it is running: the launch starts with a CENSUS (every
// block counts in on its XCD slot's counter; the picker waits -- bounded -- for all of them, then says go or abort).
// After an abort nothing has been touched; the host falls back to one launch per iteration and stops trying.
// Every later wait is bounded too and ends in st->xerror = 2 (reported as UTM_EHIP).
#pragma once
#include "common.hip.h"
#include "pick.hip.h"
#include "score_int.hip.h"

#define UTM_LOOP_MAX_CHAINERS 8
struct LoopSync {
    u64 pub[2];              // W0, W1
    u64 pad0[14];
    unsigned arrive[8 * 32];  // census counters, one per XCD slot (blockIdx & 7), 128 B apart
    unsigned go;              // the picker's census verdict: 1 go, 2 abort
    unsigned worker_timeout;  // a worker gave up waiting for a record (diagnostic)
    // interval form: the picker's request to the chainer blocks -- sums on record for these samples, please (chainer j
    // takes the samples s with s % n_chainers == j)
    unsigned req_s[2 * UTM_LOOP_MAX_CHAINERS];
    unsigned pad1[30 - 2 * UTM_LOOP_MAX_CHAINERS];
    u64 req_hdr;              // id << 8 | number of samples; id = epoch << 8 | round within the iteration (written last)
    u64 pad2[15];
    u64 req_done[UTM_LOOP_MAX_CHAINERS];  // the id chainer j has served
    u64 pad3[16 - UTM_LOOP_MAX_CHAINERS];
    // ... and every record's first word, by iteration of the launch: the chainer follows the records at its own pace (a
    // worker cannot miss one -- the picker waits for its counts -- the chainer can)
    u64 rec_log[256];
#ifdef UTM_DEBUG_STAMPS
    // per iteration of the launch (s_memrealtime, 10 ns ticks): 0 picker saw every count word complete, 1 record published,
    // 2 block 1 / wave 0 finished its positions, 3 ... saw the record, 4 ... tile updated, 5 unused, 6 block 1 / wave 0 first
    // batch of the next iteration counted
    u64 stamps[256][16];  // [0..7] the hand-off's stages; [8..15] the interval form's picker (candidates, chain stages)
    u64 wave_t[2][8192];  // iteration UTM_STAMP_ITER of the launch: every wave's [0] tile-ready and [1] last-partial times
#endif
};
#define UTM_STAMP_ITER 100
#ifdef UTM_DEBUG_STAMPS
#define UTM_LSTAMP(sync, k, slot) (sync)->stamps[(k) & 255][slot] = (u64)wall_clock64()
#else
#define UTM_LSTAMP(sync, k, slot) (void)0
#endif
static_assert(sizeof(LoopSync) % 16 == 0, "zeroed by one memset");

#define UTM_LOOP_THREADS 512  // 8 waves share a worker's tile; the picker's 512 threads cover 2,560 count words 5 apiece
#define UTM_LOOP_WAVES (UTM_LOOP_THREADS / 64)
#ifndef UTM_LOOP_E
#define UTM_LOOP_E 5  // count words a picker thread keeps in flight (x UTM_LOOP_THREADS = one chunk)
#endif
#define UTM_CLAIM_STRIDE 32  // claim counters are 128 B apart
// AF form: a count word = decrease:40 | partials arrived:12 | non-empty partials among them:12; a sum word = decrease:56 |
// non-empty partials arrived:8 (at most 64 tiles; a decrease stays below 2^53 units) -- the picker takes a position when
// the first shows all tiles and the second shows as many arrivals as the first announces
#ifndef UTM_AF_GATHER_ON
#define UTM_AF_GATHER_ON 1  // (0: a timing experiment only -- wrong sums -- that prices the AF form's gathers)
#endif
#define UTM_NONEMPTY_SHIFT 52
#define UTM_AFD_ARRIVAL_SHIFT 56
#define UTM_COLS_SLACK_BYTES (64u << 10)  // zeros behind a chunk's last column (whole-batch reads of a short last tile)
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
struct LoopCand {  // interval form: a sample whose score interval reaches the best lower bound
    double val;    // unweighted: the estimate, or the sequential float64 sum (on record / from a chain of this iteration)
    u64 cnt;
    unsigned s, pos;
    int exact, pad;
};
#ifndef UTM_IV_EXPERIMENT
#define UTM_IV_EXPERIMENT 0  // timing experiment only (wrong rows): 1 = no chains, ties go to the estimates
#endif
#ifndef UTM_LOOP_IV_BLOCKS
#define UTM_LOOP_IV_BLOCKS 4  // interval form: waves per SIMD the kernel is compiled for (a block is two waves per SIMD: 3 would leave ONE block per CU)
#endif
#define UTM_LOOP_CHAIN_CAP 512  // addends of one chain inside the launch (one per thread of the chainer); more: the host's launches take the iteration
#define UTM_LOOP_REQ_SPINS (1u << 16)     // x s_sleep(4): ~25 ms before the picker gives up on the chainer (the iteration then goes to the host); a chain takes tens of us
#define UTM_LOOP_CHAINER_IDLE (1u << 22)  // idle polls (~0.5 us each) before the chainer takes the launch for lost
struct LoopPickLds {
    IntCand wbest[UTM_LOOP_WAVES];
    double wval[UTM_LOOP_WAVES];  // (AF: the waves' best scores; count / sample / position travel in wbest)
    unsigned n_active;
    int stop, failed, pad;
    unsigned best_pos, moved;  // the decision's change to act[] (every thread keeps its own entries current)
    // interval form
    double wlo[UTM_LOOP_WAVES];  // the waves' best lower bounds
    unsigned n_c;                // candidates appended this iteration (may exceed UTM_MAX_CAND: overflow)
    int any_inexact, unresolved;
    int req_ok;
    LoopCand cand[UTM_MAX_CAND];
};

__device__ __forceinline__ void loop_publish(LoopSync *sync, unsigned epoch, int stop, int removed, unsigned winner, unsigned best_pos,
                                             unsigned moved, bool log = false)
{
    const u64 w1 = ((u64)(epoch & 0xFFu) << 56) | ((u64)(best_pos & 0xFFFFFFFu) << 28) | (u64)(moved & 0xFFFFFFFu);
    const u64 w0 = ((u64)(epoch & UTM_LOOP_EPOCH_MASK) << 40) | ((u64)(stop ? 1 : 0) << 39) | ((u64)(removed ? 1 : 0) << 38) | (u64)winner;
    __hip_atomic_store(&sync->pub[1], w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sync->pub[0], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (log) __hip_atomic_store(&sync->rec_log[(epoch - 1) & 255u], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct LoopAf {
    const unsigned *afbits;  // this chunk's fixed-point table (af_fixed), or nullptr
    u64 *afd0, *afd1;        // per position: this iteration's decrease of the fixed-point sum (by iteration parity)
    // interval form (AFM == 2)
    const void *af_raw;      // the AF values as the caller gave them: float or double per variant (PickArgs::af_is_f64)
    u64 *priv;               // the picker's own copy of `covered`, kept current (the chunk's second covered buffer)
    u64 *newly_log;          // deferred exact scores: the log of newly-covered masks (af_defer.hip.h), or nullptr
    u64 log_stride;          // words between two slots of the log
    int log_slots;           // slots of the log (row r: slot r % log_slots)
    int spec_min_ticks;      // the chainers work ahead while iterations take longer than this (10 ns ticks; < 0: never)
    int n_chainers;          // 1..UTM_LOOP_MAX_CHAINERS blocks at the end of the grid; priv holds one covered mask for each
};

// What the picker of the interval form needs beyond the words: the matrix itself and an up-to-date `covered` (chains).
struct LoopIv {
    const u64 *cols;
    u64 wp;
    const u64 *covered;  // as the launch found it
    Pending pend;
    const void *af_raw;
    u64 *priv;
    void *scratch;       // LDS: 2 x UTM_LOOP_CHAIN_CAP doubles
    int spec_min_ticks;
    unsigned chainer, n_chainers;  // (a chainer: which one of how many; the picker: how many)
};

// ------------------------------------------------------------------------------------------------ the chainer
// Interval form: the LAST block of the grid neither picks nor streams -- it keeps sequential float64 sums on record
// (PickArgs::known_*: the sum and the count it belongs to; a sample's uncovered set only shrinks, so the sum holds while
// the count does) for the samples that may have to be told apart soon, so that the picker finds them there when intervals
// overlap.  It follows the picker's records and keeps a covered mask of its own (priv: the chunk's second covered buffer);
// between two records it serves the picker's requests (samples whose sums are needed NOW) and otherwise works ahead: the
// samples with the best estimates that have no valid sum on record.
//
// A chain (select.py:38-40: the float64 sum, in ascending variant order, over the variants a sample would newly cover), for
// up to two samples at once, by the block's 512 threads: every set bit of column & ~priv is appended to its sample's list
// in LDS (unordered, one LDS atomic each -- at most UTM_LOOP_CHAIN_CAP of them, a longer list leaves no record), an
// entry's rank is the number of smaller entries, the values go to LDS in rank order and one wave per sample adds them one
// by one (ordered_sum64).  Word w of priv is only ever touched by thread w % 512.
struct LoopChainLds {
    LoopRec rec;
    u64 req, now;  // (thread 0's clock: every decision below is the same in all threads)
    unsigned chain_n[2];
    unsigned spec_s[2], spec_n;
    int go, bad;
    double wval[UTM_LOOP_WAVES];
    unsigned ws[UTM_LOOP_WAVES];
};

__device__ __forceinline__ void loop_chain2(const PickArgs &a, const LoopIv &iv, LoopChainLds *L, unsigned s0, unsigned s1, bool two)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *vals0 = reinterpret_cast<double *>(iv.scratch), *vals1 = vals0 + UTM_LOOP_CHAIN_CAP;
    unsigned *list0 = reinterpret_cast<unsigned *>(vals0), *list1 = reinterpret_cast<unsigned *>(vals1);
    if (threadIdx.x < 2) L->chain_n[threadIdx.x] = 0;
    __syncthreads();
    // 16 bytes per lane and load, 18 loads in flight: a chr22-sized column (17k words) is three rounds of memory latency
    const v2q *c0 = reinterpret_cast<const v2q *>(iv.cols + (u64)s0 * iv.wp), *c1 = reinterpret_cast<const v2q *>(iv.cols + (u64)s1 * iv.wp);
    const v2q *pr = reinterpret_cast<const v2q *>(iv.priv);
    const u64 pairs = iv.wp / 2;
    constexpr int LQ = 6;
    for (u64 base = 0; base < pairs; base += (u64)UTM_LOOP_THREADS * LQ) {
        v2q pv[LQ], x0[LQ], x1[LQ];
#pragma unroll
        for (int q = 0; q < LQ; ++q) {
            const u64 i = base + (u64)q * UTM_LOOP_THREADS + threadIdx.x;
            const bool in = i < pairs;
            const v2q ones = {~0ull, ~0ull}, zero = {0ull, 0ull};
            pv[q] = in ? pr[i] : ones;
            x0[q] = in ? c0[i] : zero;
            x1[q] = (in && two) ? c1[i] : zero;
        }
#pragma unroll
        for (int q = 0; q < LQ; ++q) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const u64 v0 = ((base + (u64)q * UTM_LOOP_THREADS + threadIdx.x) * 2 + d) * 64;
                u64 y = x0[q][d] & ~pv[q][d];
                while (y) {
                    const unsigned slot = atomicAdd(&L->chain_n[0], 1u);
                    if (slot < UTM_LOOP_CHAIN_CAP) list0[slot] = (unsigned)(v0 + (u64)__builtin_ctzll(y));
                    y &= y - 1;
                }
                y = x1[q][d] & ~pv[q][d];
                while (y) {
                    const unsigned slot = atomicAdd(&L->chain_n[1], 1u);
                    if (slot < UTM_LOOP_CHAIN_CAP) list1[slot] = (unsigned)(v0 + (u64)__builtin_ctzll(y));
                    y &= y - 1;
                }
            }
        }
    }
    __syncthreads();
    unsigned n0 = L->chain_n[0], n1 = two ? L->chain_n[1] : 0u;
    const bool ok0 = n0 <= UTM_LOOP_CHAIN_CAP, ok1 = two && n1 <= UTM_LOOP_CHAIN_CAP;
    if (!ok0) n0 = 0;
    if (!ok1) n1 = 0;
    unsigned r0 = 0, r1 = 0;
    double v0 = 0.0, v1 = 0.0;
    const u64 e_cap = iv.wp * 64;
    if (threadIdx.x < n0) {
        unsigned e = list0[threadIdx.x];
        if ((u64)e >= e_cap) { L->bad = 1; e = 0; }  // (cannot happen; no record then, and the picker hands the iteration to the host)
        v0 = a.af_is_f64 ? static_cast<const double *>(iv.af_raw)[e] : (double)static_cast<const float *>(iv.af_raw)[e];
        for (unsigned j = 0; j < n0; ++j) r0 += list0[j] < e ? 1u : 0u;
    }
    if (threadIdx.x < n1) {
        unsigned e = list1[threadIdx.x];
        if ((u64)e >= e_cap) { L->bad = 2; e = 0; }
        v1 = a.af_is_f64 ? static_cast<const double *>(iv.af_raw)[e] : (double)static_cast<const float *>(iv.af_raw)[e];
        for (unsigned j = 0; j < n1; ++j) r1 += list1[j] < e ? 1u : 0u;
    }
    __syncthreads();  // every rank is known: the lists make room for the values
    if (threadIdx.x < n0) vals0[r0] = v0;
    if (threadIdx.x < n1) vals1[r1] = v1;
    __syncthreads();
    if (((wave == 0 && ok0) || (wave == 1 && ok1)) && !L->bad) {
        const unsigned n = wave ? n1 : n0, sc = wave ? s1 : s0;
        const double *vals = wave ? vals1 : vals0;
        double acc = 0.0;
        for (unsigned t = 0; t < n; t += 64) acc = ordered_sum64(acc, t + lane < n ? vals[t + lane] : 0.0);
        if (lane == 0) {
            // the sum first, then the count that validates it (the picker reads them the other way round)
            __hip_atomic_store(reinterpret_cast<u64 *>(&a.known_val[sc]), __builtin_bit_cast(u64, acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&a.known_cnt[sc], (u64)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void loop_chainer(const PickArgs &a, LoopSync *sync, const LoopIv &iv, LoopChainLds *L, unsigned n_act)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // covered as the launch found it, with the pending winner folded in (as the workers' tiles)
    {
        const u64 *wcol = iv.pend.fuse ? pending_column(a.st, iv.cols, iv.wp, iv.pend) : nullptr;
        for (u64 i = threadIdx.x; i < iv.wp / 2; i += UTM_LOOP_THREADS) {  // (a 16-byte pair of words is only ever touched by thread pair % 512)
            iv.priv[2 * i] = iv.covered[2 * i] | (wcol ? wcol[2 * i] : 0ull);
            iv.priv[2 * i + 1] = iv.covered[2 * i + 1] | (wcol ? wcol[2 * i + 1] : 0ull);
        }
    }
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&sync->arrive[(blockIdx.x & 7) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned g = 0;
        for (unsigned spin = 0; spin < UTM_LOOP_WAIT_SPINS; ++spin) {
            g = __hip_atomic_load(&sync->go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g) break;
            __builtin_amdgcn_s_sleep(8);
        }
        L->go = g == 1;
    }
    if (threadIdx.x == 0) L->bad = 0;
    __syncthreads();
    if (!L->go) return;
    if (iv.spec_min_ticks == -2) return;  // (test hook: chainers that are gone -- the picker's requests run into their bounded wait)
    unsigned applied = 0;  // records followed so far: priv is covered as iteration `applied` of the launch sees it
    u64 served = 0, t_last = 0;
    unsigned idle = 0, spec_budget = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            const u64 w0 = __hip_atomic_load(&sync->rec_log[applied & 255u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            LoopRec lr;
            lr.ok = (w0 >> 40) == (u64)((applied + 1) & UTM_LOOP_EPOCH_MASK);
            lr.stop = (int)(w0 >> 39 & 1);
            lr.removed = (int)(w0 >> 38 & 1);
            lr.winner = (unsigned)(w0 & 0xFFFFFFFFull);
            lr.best_pos = lr.moved = 0;
            L->rec = lr;
            L->req = __hip_atomic_load(&sync->req_hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L->now = (u64)wall_clock64();
        }
        __syncthreads();
        const LoopRec r = L->rec;
        const u64 req = L->req, now = L->now;
        __syncthreads();
        if (r.ok) {
            if (r.stop) return;
            const v2q *wc = reinterpret_cast<const v2q *>(iv.cols + (u64)r.winner * iv.wp);
            v2q *pr = reinterpret_cast<v2q *>(iv.priv);
            const u64 pairs = iv.wp / 2;
            constexpr int LQ = 8;
            for (u64 base = 0; base < pairs; base += (u64)UTM_LOOP_THREADS * LQ) {
                v2q add[LQ], old[LQ];
#pragma unroll
                for (int q = 0; q < LQ; ++q) {
                    const u64 i = base + (u64)q * UTM_LOOP_THREADS + threadIdx.x;
                    const v2q zero = {0ull, 0ull};
                    add[q] = i < pairs ? wc[i] : zero;
                    old[q] = i < pairs ? pr[i] : zero;
                }
#pragma unroll
                for (int q = 0; q < LQ; ++q)
                    if ((add[q][0] & ~old[q][0]) | (add[q][1] & ~old[q][1])) pr[base + (u64)q * UTM_LOOP_THREADS + threadIdx.x] = old[q] | add[q];
            }
            // one chain ahead per record, started now -- and only while iterations are long enough for it to be over before
            // the picker may ask for something (a request waits for the chain in progress)
            spec_budget = (iv.spec_min_ticks >= 0 && t_last && now - t_last > (u64)iv.spec_min_ticks) ? 1u : 0u;
            t_last = now;
            applied += 1;
            n_act -= r.removed ? 1u : 0u;
            idle = 0;
            continue;
        }
        const u64 id = req >> 8;
        if (id != served && (id >> 8) == (u64)applied + 1) {
            // the picker is waiting: these samples' sums, two at a time
            // ... those of this chainer's residue class: a sample's record has ONE writer, whoever asks and whenever (two
            // chainers at different points of the record log writing the same sample's sum and count could interleave
            // into a pair that matches the picker's count with the other's sum)
            const unsigned n = (unsigned)(req & 0xFFu);
            unsigned mine[2 * UTM_LOOP_MAX_CHAINERS], n_mine = 0;
            for (unsigned i = 0; i < n && i < 2 * UTM_LOOP_MAX_CHAINERS; ++i) {
                const unsigned rs = __hip_atomic_load(&sync->req_s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rs % iv.n_chainers == iv.chainer) mine[n_mine++] = rs;
            }
            for (unsigned i = 0; i < n_mine; i += 2) loop_chain2(a, iv, L, mine[i], i + 1 < n_mine ? mine[i + 1] : mine[i], i + 1 < n_mine);
            if (threadIdx.x == 0) __hip_atomic_store(&sync->req_done[iv.chainer], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            served = id;
            idle = 0;
            continue;
        }
        // nothing asked for: work ahead -- the two best estimates (by wave) without a valid sum on record
        if (!spec_budget) {
            if (++idle > UTM_LOOP_CHAINER_IDLE) return;  // (no record for seconds: the launch is lost, its picker reports it)
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        spec_budget = 0;
        double bv = -__builtin_inf();
        unsigned bs = 0xFFFFFFFFu;
        for (unsigned i = threadIdx.x; i < n_act; i += UTM_LOOP_THREADS) {
            const unsigned s = __hip_atomic_load(&a.act[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s >= a.n_local || s % iv.n_chainers != iv.chainer) continue;  // (every chainer works ahead on its own residue class)
            const u64 c = __hip_atomic_load(&a.cnt[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 kc = __hip_atomic_load(&a.known_cnt[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c == 0 || c == kc || c > (u64)UTM_LOOP_CHAIN_CAP) continue;
            double v = (double)(i64)__hip_atomic_load(reinterpret_cast<u64 *>(&a.afsum[s]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * a.af_scale;
            if (a.weights) v *= a.weights[a.first + s];
            if (v > bv) {
                bv = v;
                bs = s;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(bv, o, 64);
            const unsigned os = __shfl_xor(bs, o, 64);
            if (ov > bv || (ov == bv && os < bs)) {
                bv = ov;
                bs = os;
            }
        }
        if (lane == 0) {
            L->wval[wave] = bv;
            L->ws[wave] = bs;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned n = 0;
            int w1 = -1;
            for (int pass = 0; pass < 2; ++pass) {
                int bw = -1;
                for (int w8 = 0; w8 < UTM_LOOP_WAVES; ++w8)
                    if (w8 != w1 && L->ws[w8] != 0xFFFFFFFFu && (bw < 0 || L->wval[w8] > L->wval[bw])) bw = w8;
                if (bw < 0) break;
                L->spec_s[n++] = L->ws[bw];
                w1 = bw;
            }
            L->spec_n = n;
        }
        __syncthreads();
        const unsigned n_spec = L->spec_n;
        if (n_spec) {
            loop_chain2(a, iv, L, L->spec_s[0], n_spec > 1 ? L->spec_s[1] : L->spec_s[0], n_spec > 1);
            idle = 0;
        }
    }
}

// AF = false: scores are the counts themselves (unweighted integer loop).  AF = true: the exact fixed-point AF phase
// (float32 values, every sum below 2^53 units: DESIGN.md section 4 "Exactness") and / or per-sample weights -- the count
// words carry this iteration's DECREASE of a position's count (what the pending winner newly covered of it), afd0 / afd1
// the decrease of its fixed-point sum (added, by a returning atomic, BEFORE the count word's arrival: the word certifies
// both); the picker keeps the per-sample accumulators (a.cnt / a.afsum, by sample) current and compares float64 scores.
struct AfCand {
    double val;
    u64 cnt;
    unsigned s, pos;
};
__device__ __forceinline__ bool better_af(const AfCand &a, const AfCand &b)
{
    return a.val > b.val || (a.val == b.val && a.s < b.s);
}
// MODE 0: unweighted integer scores.  1: the AF form.  2: integer counts times per-sample weights (full counts in the
// words as in MODE 0, float64 products compared as in MODE 1).  3: the interval form of the AF loop -- words and
// accumulators as in MODE 1, but the fixed-point sums only ESTIMATE the reference's float64 sums (float64 AF values, or
// float32 ones whose sums leave the exact range): a sample's score is an interval (af_interval_regs, as k_cand's), the
// samples whose intervals reach the best lower bound are the candidates; one candidate is the winner, several are told
// apart by their sequential float64 sums -- on record from an earlier chain, or chained here (loop_chain2).  What the
// picker cannot settle (more than UTM_MAX_CAND candidates, a chain longer than UTM_LOOP_CHAIN_CAP) ends the launch with
// the iteration scored but undecided (IterState::loop_unresolved): the host's verification launch decides it.
template <int MODE>
__device__ __forceinline__ void loop_picker(const PickArgs &a, LoopSync *sync, u64 *cnt0, u64 *cnt1, u64 *afd0, u64 *afd1, unsigned *claim,
                                            unsigned n_tiles, unsigned n_blocks, int k_batch, LoopPickLds *L, const LoopIv &iv)
{
    constexpr bool IV = MODE == 3;
    constexpr bool AF = MODE == 1 || IV;  // decrease words, per-sample accumulators
    constexpr bool DBL = MODE != 0;       // float64 scores: (score descending, sample ascending), the negative-best rule
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
    // all selectable samples within one chunk of count words (2,560): act[] is read once, then kept up to date from the
    // decisions themselves -- a picker iteration then starts polling without a round trip to memory
    const bool one_chunk = n_active <= UTM_LOOP_THREADS * UTM_LOOP_E;
    unsigned s[UTM_LOOP_E];
    u64 c_keep[UTM_LOOP_E], a_keep[UTM_LOOP_E];  // (AF: the accumulators of this thread's samples)
    i64 chain_events = 0;
#pragma unroll
    for (int e = 0; e < UTM_LOOP_E; ++e) {
        s[e] = 0;
        c_keep[e] = a_keep[e] = 0;
    }
    u64 t_pub = wall_clock64(), t_work = 0;  // when the last record went out; how long after ITS predecessor the last iteration's counts were complete (100 MHz ticks)
    for (int k = 0;; ++k) {
        u64 *cnt = (k & 1) ? cnt1 : cnt0;
        // Stay off the memory system while the iteration is certainly still running: polling all launch long costs the
        // streaming waves bandwidth.  Iterations shrink slowly (one column fewer each time), so the counts are not
        // complete before the last iteration's were, minus a margin; short iterations are polled from the start.
        if (t_work > 1200) {
            const u64 until = t_pub + t_work - 800;
            while ((u64)wall_clock64() < until) __builtin_amdgcn_s_sleep(16);
        }
        IntCand best{0, 0xFFFFFFFFu, 0};
        AfCand fbest{-__builtin_inf(), 0, 0xFFFFFFFFu, 0};
        double best_lo = -__builtin_inf();
        int failed = 0, any_inexact = 0;
        u64 *afd = (k & 1) ? afd1 : afd0;
        // UTM_LOOP_E words per thread in flight, EVERY incomplete word of the chunk re-read every round: once the last
        // partial has landed, the next round sees the chunk complete (2,504 samples are one chunk)
        for (unsigned base = 0; base < n_active && !failed; base += UTM_LOOP_THREADS * UTM_LOOP_E) {
            const unsigned i0 = base + threadIdx.x;
            unsigned need = 0;
#pragma unroll
            for (int e = 0; e < UTM_LOOP_E; ++e) {
                const unsigned i = i0 + e * UTM_LOOP_THREADS;
                const bool in = i < n_active;
                // (one chunk: this thread's entries of act[] -- and, AF, their accumulators -- stay in its registers from
                // iteration to iteration, see below)
                if (!one_chunk || k == 0) {
                    s[e] = in ? __hip_atomic_load(&a.act[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    if (AF && in) {
                        c_keep[e] = __hip_atomic_load(&a.cnt[s[e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a_keep[e] = __hip_atomic_load(reinterpret_cast<u64 *>(&a.afsum[s[e]]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (in) need |= 1u << e;
            }
            u64 *words = cnt + i0;
            for (unsigned spin = 0; need; ++spin) {
                u64 v[UTM_LOOP_E];
#pragma unroll
                for (int e = 0; e < UTM_LOOP_E; ++e)
                    v[e] = (need >> e & 1) ? __hip_atomic_fetch_add(words + e * UTM_LOOP_THREADS, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                for (int e = 0; e < UTM_LOOP_E; ++e) {
                    bool fin = (need >> e & 1) && ((v[e] >> UTM_ARRIVAL_SHIFT) & 0xFFFu) == n_tiles;
                    u64 d_af = 0;
                    if (AF && fin && (v[e] >> UTM_NONEMPTY_SHIFT)) {
                        // the sum word: complete when it shows as many arrivals as the count word announces (else next round)
                        u64 *dw = afd + i0 + e * UTM_LOOP_THREADS;
                        const u64 aw = __hip_atomic_load(dw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((aw >> UTM_AFD_ARRIVAL_SHIFT) == (v[e] >> UTM_NONEMPTY_SHIFT)) {
                            d_af = aw & ((1ull << UTM_AFD_ARRIVAL_SHIFT) - 1);
                            __hip_atomic_store(dw, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            fin = false;
                        }
                    }
                    if (fin) __hip_atomic_store(words + e * UTM_LOOP_THREADS, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (next used two iterations on)
                    if (!DBL) {
                        const IntCand cand{fin ? (v[e] & count_mask) : 0ull, fin ? s[e] : 0xFFFFFFFFu, i0 + e * UTM_LOOP_THREADS};
                        if (better_int(cand, best)) best = cand;
                    } else if (fin) {
                        const u64 d_cnt = v[e] & count_mask;
                        if (AF && d_cnt) {
                            c_keep[e] -= d_cnt;
                            a_keep[e] -= d_af;
                            __hip_atomic_store(&a.cnt[s[e]], c_keep[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(reinterpret_cast<u64 *>(&a.afsum[s[e]]), a_keep[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (IV) {
                            double lo, hi, est;
                            bool exact, est_exact;
                            af_interval_regs(a, s[e], c_keep[e], (i64)a_keep[e], ~0ull, 0.0, lo, hi, est, exact, est_exact);
                            best_lo = lo > best_lo ? lo : best_lo;
                            any_inexact |= est_exact ? 0 : 1;
                            __builtin_amdgcn_sched_barrier(0);  // (one sample's interval at a time: five interleaved ones do not fit the registers)
                        } else {
                            const u64 count = AF ? c_keep[e] : d_cnt;  // (MODE 2: the word IS the count)
                            double val = AF ? (double)(i64)a_keep[e] * a.af_scale : (double)count;  // exact: < 2^53 units, power-of-two scale
                            if (a.weights) val *= a.weights[a.first + s[e]];
                            const AfCand cand{val, count, s[e], i0 + e * UTM_LOOP_THREADS};
                            if (better_af(cand, fbest)) fbest = cand;
                        }
                    }
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
            if (IV) {
                const double other = __shfl_xor(best_lo, o, 64);
                best_lo = other > best_lo ? other : best_lo;
            } else if (!DBL) {
                IntCand other;
                other.cnt = __shfl_xor(best.cnt, o, 64);
                other.s = __shfl_xor(best.s, o, 64);
                other.pos = __shfl_xor(best.pos, o, 64);
                if (better_int(other, best)) best = other;
            } else {
                AfCand other;
                other.val = __shfl_xor(fbest.val, o, 64);
                other.cnt = __shfl_xor(fbest.cnt, o, 64);
                other.s = __shfl_xor(fbest.s, o, 64);
                other.pos = __shfl_xor(fbest.pos, o, 64);
                if (better_af(other, fbest)) fbest = other;
            }
        }
        t_work = (u64)wall_clock64() - t_pub;  // (this thread's words; all threads finish within a round of each other)
        if (threadIdx.x == 0) L->failed = 0;
        if (threadIdx.x == 0) UTM_LSTAMP(sync, k, 0);
        unsigned n_c = 0;
        if (IV) {
            if (lane == 0) L->wlo[threadIdx.x >> 6] = best_lo;
            if (threadIdx.x == 0) {
                L->n_c = 0;
                L->any_inexact = 0;
            }
        }
        __syncthreads();
        if (IV) {
            // the candidates: every sample whose interval reaches the best lower bound (this thread's samples are in its
            // registers: the intervals are made again rather than kept).  Sums on record are NOT used here -- an estimate's
            // interval contains the sum, so the candidates are a superset of k_cand's -- only looked up for the candidates.
            best_lo = L->wlo[0];
#pragma unroll
            for (int w8 = 1; w8 < UTM_LOOP_WAVES; ++w8) best_lo = L->wlo[w8] > best_lo ? L->wlo[w8] : best_lo;
#pragma unroll
            for (int e = 0; e < UTM_LOOP_E; ++e) {
                const unsigned i = threadIdx.x + e * UTM_LOOP_THREADS;
                if (i >= n_active || failed) continue;
                double lo, hi, est;
                bool exact, est_exact;
                af_interval_regs(a, s[e], c_keep[e], (i64)a_keep[e], ~0ull, 0.0, lo, hi, est, exact, est_exact);
                if (hi >= best_lo) {
                    const unsigned slot = atomicAdd(&L->n_c, 1u);
                    if (slot < UTM_MAX_CAND) L->cand[slot] = LoopCand{est, c_keep[e], s[e], i, exact ? 1 : 0, 0};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (any_inexact) L->any_inexact = 1;
        } else {
            if (DBL) best = IntCand{fbest.cnt, fbest.s, fbest.pos};
            if (lane == 0) {
                L->wbest[threadIdx.x >> 6] = best;
                if (DBL) L->wval[threadIdx.x >> 6] = fbest.val;
            }
        }
        if (failed) L->failed = 1;
        __syncthreads();
        if (IV) {
            // sequential float64 sums where the intervals do not settle the iteration (k_cand's rule): on record -- the
            // chainer block works ahead -- or asked for now
            n_c = L->n_c;
            if (threadIdx.x == 0) UTM_LSTAMP(sync, k, 8);  // the candidates are listed
            bool unresolved = n_c > UTM_MAX_CAND, asked = false;
            if (!unresolved && n_c > 0 && !L->failed) {
                bool inexact = (unsigned)lane < n_c && !L->cand[lane].exact;  // lane i of every wave: candidate i
                const bool zero_est = inexact && L->cand[lane].val == 0.0;  // (every addend floored away, yet the score is > 0: select.py:51 compares it with 0)
                const bool too_long = inexact && L->cand[lane].cnt > (u64)UTM_LOOP_CHAIN_CAP;
                const bool chain_needed = UTM_IV_EXPERIMENT == 0 && __ballot(inexact) != 0 && (__ballot(zero_est) != 0 || !(a.af_skip_single && n_c == 1));
                if (chain_needed && __ballot(too_long) != 0) {
                    unresolved = true;
                } else if (chain_needed) {
                    u64 todo_before = 0;
                    for (unsigned round = 0;; ++round) {
                        // what is on record by now (the sum belongs to the count stored AFTER it)
                        if (threadIdx.x < 64 && inexact) {
                            const unsigned cs = L->cand[lane].s;
                            if (__hip_atomic_load(&a.known_cnt[cs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == L->cand[lane].cnt) {
                                L->cand[lane].val = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<u64 *>(&a.known_val[cs]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                L->cand[lane].exact = 1;
                            }
                        }
                        __syncthreads();
                        if (threadIdx.x == 0 && round == 0) UTM_LSTAMP(sync, k, 9);  // the records have been looked up
                        inexact = (unsigned)lane < n_c && !L->cand[lane].exact;
                        u64 todo = __ballot(inexact);
                        if (threadIdx.x == 0 && !todo) UTM_LSTAMP(sync, k, 11);  // every candidate's sum is known
                        if (threadIdx.x == 0 && todo) UTM_LSTAMP(sync, k, 10);  // (the last request goes out)
                        if (!todo) break;
                        if (todo == todo_before) {  // (a served request that put nothing on record -- a chain longer than the chainers hold: the host decides)
                            unresolved = true;
                            break;
                        }
                        todo_before = todo;
                        asked = true;
                        // the first 16 still missing: to the chainers (each takes its residue class of samples)
                        const u64 id = ((u64)((unsigned)k + 1) << 8) | (u64)(round & 0xFFu);
                        if (threadIdx.x == 0) {
                            unsigned n = 0;
                            for (u64 t = todo; t && n < 2 * UTM_LOOP_MAX_CHAINERS; t &= t - 1, ++n)
                                __hip_atomic_store(&sync->req_s[n], L->cand[__builtin_ctzll(t)].s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __hip_atomic_store(&sync->req_hdr, (id << 8) | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            bool ok = true;
                            for (unsigned j = 0; j < iv.n_chainers && ok; ++j) {
                                ok = false;
                                for (unsigned spin = 0; spin < UTM_LOOP_REQ_SPINS; ++spin) {
                                    if (__hip_atomic_load(&sync->req_done[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == id) { ok = true; break; }
                                    __builtin_amdgcn_s_sleep(4);
                                }
                            }
                            L->req_ok = ok ? 1 : 0;
                        }
                        __syncthreads();
                        if (!L->req_ok) {  // (the chainer is gone, or a chain came out longer than it holds: the host decides)
                            unresolved = true;
                            break;
                        }
                    }
                }
            }
            if (threadIdx.x == 0) {
                L->unresolved = unresolved ? 1 : 0;
            }
            chain_events += asked ? 1 : 0;  // (IterState::chain_events: iterations whose pick had to wait for chains)
        }
        if (threadIdx.x == 0) {
            const unsigned epoch = (unsigned)k + 1;
            int stop = 0;
            unsigned last_act_used = 0;
            if (L->failed) {
                st->xerror = 2;  // a partial count never arrived (a logic error, not a data condition)
                st->done = 1;
                stop = 1;
                loop_publish(sync, epoch, 1, 0, 0, 0, 0, IV);
            } else if (IV && L->unresolved) {
                // scored, not decided: the accumulators and (once the workers have left) covered are current, nothing is
                // pending -- the host's verification launch makes this iteration's pick
                st->prev_valid = 0;
                st->loop_unresolved = 1;
                stop = 1;
                loop_publish(sync, epoch, 1, 0, 0, 0, 0, IV);
            } else {
                double best_val = DBL ? fbest.val : 0.0;
                if (IV) {
                    // every candidate's value is final (or it is the only one): mask / weight / argmax as pick_among_candidates
                    AfCand m{-__builtin_inf(), 0, 0xFFFFFFFFu, 0};
                    for (unsigned i = 0; i < n_c; ++i) {
                        const LoopCand cd = L->cand[i];
                        double v = cd.val;
                        if (a.weights) v *= a.weights[a.first + cd.s];
                        const AfCand o{v, cd.cnt, cd.s, cd.pos};
                        if (better_af(o, m)) m = o;
                    }
                    best = IntCand{m.cnt, m.s, m.pos};
                    best_val = m.val;
                }
                for (int w8 = 1; w8 < UTM_LOOP_WAVES && !IV; ++w8) {
                    if (!DBL) {
                        if (better_int(L->wbest[w8], best)) best = L->wbest[w8];
                    } else {
                        const AfCand o{L->wval[w8], L->wbest[w8].cnt, L->wbest[w8].s, L->wbest[w8].pos};
                        const AfCand m{best_val, best.cnt, best.s, best.pos};
                        if (better_af(o, m)) {
                            best = L->wbest[w8];
                            best_val = o.val;
                        }
                    }
                }
                if (!DBL) best_val = (double)best.cnt;
                // decide_single, on the loop state this thread carries in registers (a negative best score only wins when
                // no sample holds a masked 0: select.py:43-48)
                const bool zero_elsewhere = n_active_total < (i64)a.n_total;
                if (n_active == 0 || (DBL ? (best_val == 0.0 || (best_val < 0.0 && zero_elsewhere)) : best.cnt == 0)) {
                    st->done = 1;  // (None, None): no row (select.py:51-52, :93-96)
                    a.res_idx[iter] = -1;
                    stop = 1;
                    loop_publish(sync, epoch, 1, 0, 0, 0, 0, IV);
                } else {
                    const unsigned moved = last_act;
                    last_act_used = moved;
                    const int finished = tot + (i64)best.cnt >= a.n_var_total;  // "Ran out of new variants" (select.py:110-112)
                    stop = finished || k + 1 >= k_batch;
                    // the record first: everything below is bookkeeping nobody inside the launch waits for
                    loop_publish(sync, epoch, stop, 1, best.s, best.pos, moved, IV);
                    __hip_atomic_store(&a.act[best.pos], moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.res_idx[iter] = (i64)a.first + best.s;
                    a.res_new[iter] = (i64)best.cnt;
                    a.res_score[iter] = best_val;
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
                    rc->score = best_val;
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
                if (IV) {
                    st->chain_events += (u64)chain_events;
                    st->all_exact = L->any_inexact ? 0 : 1;
                }
            }
            L->n_active = n_active;
            L->stop = stop;
            L->best_pos = best.pos;
            L->moved = last_act_used;
            UTM_LSTAMP(sync, k, 1);
        }
        // every count word of iteration k is complete => every claim of iteration k has come back (a wave waits for its
        // claims before its last partial count): its counters can go back to 0.  THREE sets in turn (k % 3): a wave takes
        // its first ticket of iteration k + 1 while iteration k is still running, from the set that was cleared after
        // iteration k - 2 -- a whole iteration (and this block's vmcnt(0) below) earlier
        for (unsigned t = threadIdx.x; t < n_tiles * UTM_LOOP_WAVES; t += UTM_LOOP_THREADS)
            __hip_atomic_store(claim + ((size_t)(k % 3) * n_tiles * UTM_LOOP_WAVES + t) * UTM_CLAIM_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // this thread's clearing stores (and thread 0's act[] store) have landed before anybody reads those words again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_pub = wall_clock64();
        __syncthreads();
        if (L->stop) return;
        n_active = L->n_active;
        if (one_chunk) {
            const unsigned bp = L->best_pos, mv = L->moved;
#pragma unroll
            for (int e = 0; e < UTM_LOOP_E; ++e)
                if (threadIdx.x + e * UTM_LOOP_THREADS == bp) {
                    s[e] = mv;
                    if (AF) {  // (whoever held that sample stored every change; the barrier and vmcnt(0) above order it)
                        c_keep[e] = __hip_atomic_load(&a.cnt[mv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a_keep[e] = __hip_atomic_load(reinterpret_cast<u64 *>(&a.afsum[mv]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ the launch
// Grid = 1 + n_tiles * Q blocks of 512 threads: block 0 picks; worker w = blockIdx.x - 1 owns tile w % n_tiles, slot
// w / n_tiles -- a tile's workers are spread over all XCDs and over all dispatch ages (measured: a CU serves its oldest
// waves first and the XCDs differ by up to 18 % in speed, so statically equal shares finished up to a third of an
// iteration apart).  Positions: every wave has TWO static positions, first = 8 * slot + wave and second = first + 8Q (the
// ones whose first batches it holds across the hand-off); the positions from 16Q on are CLAIMED, one at a time, from the
// tile's eight counters (one per wave index: position = 16Q + 8 * ticket + wave), so fast waves take more and a tile's
// waves finish together.  A claim is issued ahead of the column loads of the batch in front of it, so its round trip
// hides behind them; a wave issues no claim after the one that came back out of range, and has every claim back before its
// last partial count goes out -- which is why the picker may reset an iteration's counters as soon as all its count
// words are complete.
// drop_iter: test hook (0 = off) -- the first worker withholds one partial count in that iteration of the launch, so
// that the picker's bounded wait runs out.
// AF: the delta form of the exact fixed-point AF phase.  The per-sample accumulators (count, fixed-point sum) are valid
// when the launch starts; an iteration subtracts what its pending winner newly covers: the worker keeps a second LDS tile,
// newly = live & winner (made for free while the tile is updated), counts column & newly instead of column & live, and
// gathers the fixed-point AF entries of the few surviving bits from the table in global memory (afbits; L2 / Infinity
// Cache resident).  Same bytes streamed as the integer loop, same hand-off.
// AFM: 0 integer scores | 1 the AF form, exact fixed-point phase | 2 the AF form with score intervals (loop_picker<3>)
template <int STEPS, bool NT, int AFM>
__global__ __launch_bounds__(UTM_LOOP_THREADS, AFM == 2 ? UTM_LOOP_IV_BLOCKS : 4) void k_loop_int(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp, const Pending pend,
                                                  IterState *__restrict__ st, unsigned *__restrict__ act, u64 *__restrict__ cnt0,
                                                  u64 *__restrict__ cnt1, unsigned q_slots, unsigned *__restrict__ claim, int k_batch,
                                                  LoopSync *__restrict__ sync, const PickArgs pa, int drop_iter, int use_claims, int ahead_ticks, int ahead0_ticks,
                                                  const LoopAf laf)
{
    static_assert(STEPS % 8 == 0 && STEPS <= 64, "a tile is 1..8 batches of 8 KiB");
    constexpr bool AF = AFM != 0;
    __shared__ v4u live[STEPS * 64];  // ~covered of this worker's tile (STEPS KiB), for the whole launch
    __shared__ v4u newly_lds[AF ? STEPS * 64 : 1];  // AF: what the pending winner newly covers of the tile
    __shared__ LoopRec rec_lds;
    __shared__ unsigned rec_epoch_lds;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    typedef unsigned v8u __attribute__((ext_vector_type(8)));
    v8u raw;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(raw) : "s"(st) : "memory");
    const IterHead head = __builtin_bit_cast(IterHead, raw);
    if (head.done) return;  // (uniform over the launch, the picker included: nobody waits for anybody)
    constexpr unsigned TILE_WORDS = STEPS * UTM_STEP_WORDS;
    const unsigned n_tiles = (unsigned)((wp + TILE_WORDS - 1) / TILE_WORDS);
    if (blockIdx.x == 0) {
        static_assert(sizeof(LoopPickLds) <= 8 * 64 * sizeof(v4u), "the picker's scratch lives in the tile of the smallest instantiation");
        LoopPickLds *lds = reinterpret_cast<LoopPickLds *>(&live[0]);
        const LoopIv iv{cols, wp, covered, pend, laf.af_raw, laf.priv, nullptr, 0, 0, (unsigned)laf.n_chainers};
        if (AFM == 2) loop_picker<3>(pa, sync, cnt0, cnt1, laf.afd0, laf.afd1, claim, n_tiles, gridDim.x, k_batch, lds, iv);
        else if (AFM == 1) loop_picker<1>(pa, sync, cnt0, cnt1, laf.afd0, laf.afd1, claim, n_tiles, gridDim.x, k_batch, lds, iv);
        else if (pa.weights) loop_picker<2>(pa, sync, cnt0, cnt1, nullptr, nullptr, claim, n_tiles, gridDim.x, k_batch, lds, iv);
        else loop_picker<0>(pa, sync, cnt0, cnt1, nullptr, nullptr, claim, n_tiles, gridDim.x, k_batch, lds, iv);
        return;
    }
    if (AFM == 2 && blockIdx.x >= gridDim.x - (unsigned)laf.n_chainers) {  // (grid = picker + workers + chainers)
        static_assert(sizeof(LoopChainLds) <= 8 * 64 * sizeof(v4u) && 2 * UTM_LOOP_CHAIN_CAP * sizeof(double) <= 8 * 64 * sizeof(v4u),
                      "the chainer's scratch lives in the two tiles of the smallest instantiation");
        const unsigned cj = blockIdx.x - (gridDim.x - (unsigned)laf.n_chainers);
        const LoopIv iv{cols, wp, covered, pend, laf.af_raw, laf.priv + (u64)cj * wp, &newly_lds[0], laf.spec_min_ticks, cj, (unsigned)laf.n_chainers};
        loop_chainer(pa, sync, iv, reinterpret_cast<LoopChainLds *>(&live[0]), head.n_active);
        return;
    }
    // (interval form with deferred exact scores: the row a launch's first iteration logs, see below)
    const i64 iter0 = (AFM == 2 && laf.newly_log) ? st->iter : 0;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&sync->arrive[(blockIdx.x & 7) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned w = blockIdx.x - 1;
    const unsigned tile = w % n_tiles, slot = w / n_tiles;
    const u64 w0 = (u64)tile * TILE_WORDS;
    const u64 left = (wp - w0) / UTM_STEP_WORDS;
    const int nsteps = left < (u64)STEPS ? (int)left : STEPS;  // KiB of this tile (the last tile of a column may be short)
    const int nb = STEPS == 8 ? 1 : (nsteps + 7) / 8;          // ... in batches of 8 KiB: what a wave has in flight per buffer
                                                               // (the 8 KiB tile: a compile-time 1, and every batch index below a 0)
    constexpr int U = 8;
#define UTM_COL_LOAD(ptr) (NT ? __builtin_nontemporal_load(ptr) : *(ptr))
    // Batch J (8 KiB) of a sample's tile-column into buffer X: one base address, immediate offsets.  A column's last tile
    // may end inside a batch: the loads then run on into the next column (or the slack behind the matrix,
    // UTM_COLS_SLACK_BYTES) and those words are counted against zero words of the LDS tile -- one code path, no clamps.
#define UTM_BATCH_LOAD(X, S, J)                                                                      \
    {                                                                                                \
        const v4u *p_ = reinterpret_cast<const v4u *>(cols + (u64)(S) * wp + w0) + (J) * 8 * 64 + lane; \
        _Pragma("unroll") for (int q = 0; q < U; ++q) X[q] = UTM_COL_LOAD(p_ + q * 64);              \
    }
#define UTM_BATCH_COUNT(X, J)                                                                        \
    {                                                                                                \
        const v4u *l_ = (AF ? &newly_lds[(J) * 8 * 64 + lane] : &live[(J) * 8 * 64 + lane]);         \
        if (!AF) {                                                                                   \
            _Pragma("unroll") for (int q = 0; q < U; ++q)                                            \
            {                                                                                        \
                const v4u b = X[q] & l_[q * 64];                                                     \
                acc += __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);                        \
            }                                                                                        \
        } else {                                                                                     \
            unsigned nb_ = 0;                                                                        \
            _Pragma("unroll") for (int q = 0; q < U; ++q)                                            \
            {                                                                                        \
                X[q] &= l_[q * 64];                                                                  \
                nb_ += __popc(X[q].x) + __popc(X[q].y) + __popc(X[q].z) + __popc(X[q].w);            \
            }                                                                                        \
            acc += nb_;                                                                              \
            if (UTM_AF_GATHER_ON && __ballot(nb_ != 0) != 0) { /* rare once coverage has grown: walk the surviving bits */ \
                const unsigned *af_ = laf.afbits + (w0 + (u64)(J) * 8 * UTM_STEP_WORDS + 2 * lane) * 64; \
                _Pragma("unroll") for (int q = 0; q < U; ++q)                                        \
                {                                                                                    \
                    _Pragma("unroll") for (int d = 0; d < 4; ++d)                                    \
                    {                                                                                \
                        unsigned bits = X[q][d];                                                     \
                        const unsigned *a_ = af_ + q * (UTM_STEP_WORDS * 64) + d * 32;               \
                        while (bits) {                                                               \
                            afacc += af_fixed(a_[__builtin_ctz(bits)]);                              \
                            bits &= bits - 1;                                                        \
                        }                                                                            \
                    }                                                                                \
                }                                                                                    \
            }                                                                                        \
        }                                                                                            \
    }
    // A wave's work in an iteration is a STREAM of batches: all nb batches of its first position, of its second, then of
    // the positions it claims.  first = 8 * slot + wave and second = first + 8Q are static; the positions from 16Q on are
    // claimed.  Two batches are in flight at any time (x0, x1 in turn) -- and across the hand-off: the first two batches
    // of the next iteration's stream are what a wave holds while the picker decides.
    const unsigned stride = q_slots * UTM_LOOP_WAVES;
    const unsigned first = slot * UTM_LOOP_WAVES + wave, second = first + stride, dyn0 = 2 * stride;
    unsigned n_act = head.n_active;
    const unsigned act_cap = pa.n_local + UTM_PICK_PAD;  // (entries that may be read ahead of the bounds that say whether they count)
    unsigned s_first = first < act_cap ? act[first] : 0u;  // (act[] is as the host / the last launch left it: plain loads)
    unsigned s_second = second < act_cap ? act[second] : 0u;
    v4u x0[U], x1[U];
    // the stream's first two batches: (first, 0) and then (first, 1), or (second, 0) where a tile is one batch
#define UTM_AHEAD0() if (first < n_act) { UTM_BATCH_LOAD(x0, s_first, 0) }
#define UTM_AHEAD1()                                                         \
    if (nb > 1) { if (first < n_act) { UTM_BATCH_LOAD(x1, s_first, 1) } }    \
    else if (second < n_act) { UTM_BATCH_LOAD(x1, s_second, 0) }
    UTM_AHEAD0()
    UTM_AHEAD1()

    // stage ~(covered | pending winner) once
    {
        const v4u *cv = reinterpret_cast<const v4u *>(covered + w0);
        const u64 *wcol = pend.fuse ? pending_column(&head, cols, wp, pend) : nullptr;
        const v4u *wc = wcol ? reinterpret_cast<const v4u *>(wcol + w0) : nullptr;
        const v4u zero4 = {0, 0, 0, 0};
        for (int k = threadIdx.x; k < nsteps * 64; k += UTM_LOOP_THREADS) {
            v4u c = cv[k];
            const v4u w = wc ? wc[k] : zero4;
            if (AF) newly_lds[k] = w & ~c;  // the first iteration's delta: the winner that is pending when the launch starts
            c |= w;
            live[k] = ~c;
        }
        for (int k = nsteps * 64 + threadIdx.x; k < STEPS * 64; k += UTM_LOOP_THREADS) {  // (a short tile's missing steps)
            live[k] = zero4;
            if (AF) newly_lds[k] = zero4;
        }
        // Deferred exact scores (af_defer.hip.h): what a row's winner newly covered is what its float64 score runs over --
        // the mask goes into the row's slot of the log, as from a delta pass of k_score_afs (row = the pending winner's)
        if (AFM == 2 && laf.newly_log && slot == 0 && wcol && iter0 >= 1) {
            v4u *lg = reinterpret_cast<v4u *>(laf.newly_log + (u64)((iter0 - 1) % laf.log_slots) * laf.log_stride + w0);
            for (int k = threadIdx.x; k < nsteps * 64; k += UTM_LOOP_THREADS) lg[k] = wc[k] & ~cv[k];
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
        rec_epoch_lds = 0;
    }
    __syncthreads();
    if (!rec_lds.ok) return;  // abort (or no verdict): nothing has been written

    bool have_spec = false;  // the first ticket of the coming iteration has been taken (and its sample read) ahead of the record
    unsigned spec_ticket = 0, spec_s = 0, spec_ticket2 = 0;
    bool spec2_out = false;  // ... and a second one is in flight
    u64 t_rec = 0, t_iter = 0;  // this wave's clock at the last record, and the interval between the last two (10 ns ticks)
    unsigned patch_pos = 0xFFFFFFFFu, patch_s = 0;  // the newest record's change to act[] (its store may still be in flight)
#define UTM_ACT_LOAD(i) ((i) == patch_pos ? patch_s : __hip_atomic_load(&act[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    // (wave-uniform values fetched by vector instructions: moved to scalar registers)
#define UTM_UNIFORM(v) ((unsigned)__builtin_amdgcn_readfirstlane((int)(v)))
    // one ticket of this wave's counter: ONE lane adds; the value stays in that lane's register until it is needed
    // (UTM_UNIFORM on it is the wait), so the column loads issued behind the claim are not held up by its round trip
    auto take_ticket = [&](unsigned *counter) -> unsigned {
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    };
#define UTM_PARTIAL(POS, ACC, DROP)                                                                                    \
    {                                                                                                                  \
        const unsigned sum_ = wave_sum_u32(ACC);                                                                       \
        u64 word_ = (u64)sum_ + (1ull << UTM_ARRIVAL_SHIFT);                                                           \
        if (AF && sum_) { /* a non-empty partial: the sum's decrease travels in a word of its own, self-certifying too */ \
            const u64 tot_ = (u64)wave_sum_u63(afacc);                                                                 \
            u64 *afd_ = (k & 1) ? laf.afd1 : laf.afd0;                                                                 \
            if (lane == 0) atomicAdd(&afd_[POS], tot_ + (1ull << UTM_AFD_ARRIVAL_SHIFT));                              \
            word_ += 1ull << UTM_NONEMPTY_SHIFT; /* ... and the count word says how many such partials to expect */    \
            afacc = 0;                                                                                                 \
        }                                                                                                              \
        if (lane == 0 && !(DROP)) atomicAdd(&cnt[POS], word_);                                                         \
    }
    const unsigned none = 0xFFFFFFFFu;
    for (int k = 0;; ++k) {
        u64 *cnt = (k & 1) ? cnt1 : cnt0;
        unsigned *my_claim = claim + ((size_t)((k % 3) * n_tiles + tile) * UTM_LOOP_WAVES + wave) * UTM_CLAIM_STRIDE;
        if (first < n_act) {  // (x0 / x1 hold the stream's first two batches)
            // Behind the static positions: a queue of ONE claimed position whose sample is being read from act[] (pos_q,
            // s_q) and ONE claim in flight (ticket_q).  Within a refill the queue's requests go out BEFORE the column
            // loads, so that the next step's wait for them (vmcnt is in order) leaves the other buffer's 8 KiB in flight.
            const bool dyn = use_claims && n_act > dyn0;  // (else every position is somebody's static one: no claims at all)
            unsigned pos_q = none, s_q = 0, ticket_q = 0;
            bool ticket_out = false;
            if (dyn) {
                // the first TWO tickets were taken ahead of the record (below) and the first one's sample read: the queue
                // starts full, and nothing that was requested a moment ago is waited for on the way from the record to
                // the first new column loads.  Without them (first iteration of the launch) the same, now.
                if (!have_spec) {
                    spec_ticket = take_ticket(my_claim);
                    spec_ticket2 = take_ticket(my_claim);
                    spec2_out = true;
                }
                pos_q = dyn0 + (unsigned)UTM_LOOP_WAVES * UTM_UNIFORM(spec_ticket) + (unsigned)wave;
                if (pos_q < n_act) {
                    s_q = have_spec ? (pos_q == patch_pos ? patch_s : spec_s) : UTM_ACT_LOAD(pos_q);
                    if (spec2_out) {
                        ticket_q = spec_ticket2;
                    } else {
                        ticket_q = take_ticket(my_claim);
                    }
                    ticket_out = true;
                }
                // (a second ticket behind a first one that came back out of range is simply dropped: both have returned --
                // UTM_UNIFORM above and below -- before this wave's partial counts of the iteration go out)
                else if (spec2_out) (void)UTM_UNIFORM(spec_ticket2);
            } else if (!use_claims) {  // (experiment switch: the positions behind the static two dealt statically too)
                pos_q = first + dyn0;
                if (pos_q < n_act) s_q = UTM_ACT_LOAD(pos_q);
            }
            bool second_pending = second < n_act;
            // the issue iterator: position it_pos (sample it_s) has had its batches below it_j requested.  The next batch of
            // the stream -> (POS, J, S): at a position's end the next position is taken THEN (not earlier: a claimed
            // position held back by a busy wave is one an idle wave could be reading) -- `second` once, then the queue's
            // head, whose successor is claimed, and its sample requested, on the spot.  POS = none: the stream has ended.
            unsigned it_pos = first, it_s = s_first;
            int it_j = 0;
#define UTM_IT_NEXT(POS, J, S)                                                                                             \
    {                                                                                                                      \
        if (it_j == nb && it_pos != none) {                                                                                \
            it_j = 0;                                                                                                      \
            if (second_pending) {                                                                                          \
                second_pending = false;                                                                                    \
                it_pos = second;                                                                                           \
                it_s = s_second;                                                                                           \
            } else if (pos_q < n_act) {                                                                                    \
                it_pos = pos_q;                                                                                            \
                it_s = UTM_UNIFORM(s_q);                                                                                   \
                if (dyn) {                                                                                                 \
                    pos_q = ticket_out ? dyn0 + (unsigned)UTM_LOOP_WAVES * UTM_UNIFORM(ticket_q) + (unsigned)wave : none;  \
                    ticket_out = false;                                                                                    \
                    if (pos_q < n_act) {                                                                                   \
                        s_q = UTM_ACT_LOAD(pos_q);                                                                         \
                        ticket_q = take_ticket(my_claim); /* (none is taken behind one that came back out of range) */    \
                        ticket_out = true;                                                                                 \
                    }                                                                                                      \
                } else {                                                                                                   \
                    pos_q += stride;                                                                                       \
                    if (pos_q < n_act) s_q = UTM_ACT_LOAD(pos_q);                                                          \
                }                                                                                                          \
            } else {                                                                                                       \
                it_pos = none;                                                                                             \
            }                                                                                                              \
        }                                                                                                                  \
        POS = it_pos;                                                                                                      \
        J = it_j;                                                                                                          \
        S = it_s;                                                                                                          \
        ++it_j;                                                                                                            \
    }
            // what the two buffers hold: the stream's batches 0 and 1 (requested ahead of the record)
            unsigned pos_a, pos_b, s_dummy;
            int j_a, j_b;
            UTM_IT_NEXT(pos_a, j_a, s_dummy)
            UTM_IT_NEXT(pos_b, j_b, s_dummy)
            (void)s_dummy;
            unsigned acc = 0;
            u64 afacc = 0;  // (AF: this lane's fixed-point sum over the position's newly covered bits)
            (void)afacc;
            // one step: count buffer X (batch J of position POS; a position's last batch sends its partial count), then
            // refill it with the stream's next batch (the queue's requests, if any, go out first)
#define UTM_LOOP_STEP(X, POS, J)                                                                                           \
    {                                                                                                                      \
        UTM_BATCH_COUNT(X, J)                                                                                              \
        if (J == nb - 1) {                                                                                                 \
            const bool drop = drop_iter && k + 1 == drop_iter && blockIdx.x == 1 && wave == 0 && POS == first;             \
            UTM_PARTIAL(POS, acc, drop)                                                                                    \
            acc = 0;                                                                                                       \
        }                                                                                                                  \
        unsigned s_now;                                                                                                    \
        UTM_IT_NEXT(POS, J, s_now)                                                                                         \
        if (POS != none) { UTM_BATCH_LOAD(X, s_now, J) }                                                                   \
    }
            for (;;) {
                if (pos_a == none && pos_b == none) break;
                if (pos_a != none) {
                    UTM_LOOP_STEP(x0, pos_a, j_a)
#ifdef UTM_DEBUG_STAMPS
                    if (lane == 0 && k > 0 && blockIdx.x == 1 && wave == 0 && sync->stamps[(k - 1) & 255][6] == 0) UTM_LSTAMP(sync, k - 1, 6);
#endif
                }
                if (pos_b != none) { UTM_LOOP_STEP(x1, pos_b, j_b) }
            }
#undef UTM_LOOP_STEP
#undef UTM_IT_NEXT
            // (ticket_out is false here: the last claim taken came back out of range and was waited for above -- every
            // claim of this wave has returned before its last partial count went out)
        }
#ifdef UTM_DEBUG_STAMPS
        if (lane == 0 && k == UTM_STAMP_ITER && blockIdx.x * UTM_LOOP_WAVES + wave < 8192) sync->wave_t[1][blockIdx.x * UTM_LOOP_WAVES + wave] = (u64)wall_clock64();
        if (blockIdx.x == 1 && threadIdx.x == 0) UTM_LSTAMP(sync, k, 2);
#endif
        // Ahead of the record: the first two batches of the next iteration's stream.  Their samples stay where they are
        // unless the record names one of those very positions (then they are loaded again below); a position that drops
        // out of range costs one wasted batch.  These 16 KiB per wave are what keeps the memory pipes busy while the
        // picker reduces, publishes, and the tile is brought up to date.
        // ... and ahead of everything: this wave's first ticket of the next iteration and that position's sample (both
        // round trips are then off the path from the record to the first new column loads)
        have_spec = use_claims && n_act > dyn0 + 1;
        spec2_out = false;
        if (have_spec) {
            unsigned *next_claim = claim + ((size_t)(((k + 1) % 3) * n_tiles + tile) * UTM_LOOP_WAVES + wave) * UTM_CLAIM_STRIDE;
            spec_ticket = take_ticket(next_claim);
            const unsigned pos_spec = dyn0 + (unsigned)UTM_LOOP_WAVES * UTM_UNIFORM(spec_ticket) + (unsigned)wave;
            spec_s = pos_spec < act_cap ? UTM_ACT_LOAD(pos_spec) : 0u;
            if (pos_spec + 1 < n_act) {  // (in range whatever the record says: a second ticket, in flight until the iteration starts)
                spec_ticket2 = take_ticket(next_claim);
                spec2_out = true;
            }
        }
        // The first batch goes out at once (it fills the tail of the iteration, while the slowest waves finish); the
        // second is held back until `ahead` before the record is due -- iterations shrink smoothly, so the last interval
        // between two records predicts this one -- so that it is still in flight when the record arrives.
        const u64 t_due0 = (ahead0_ticks && t_iter) ? t_rec + t_iter - (t_iter > (u64)ahead0_ticks ? (u64)ahead0_ticks : t_iter) : 0;
        if (t_due0)
            while ((u64)wall_clock64() < t_due0) __builtin_amdgcn_s_sleep(2);  // (experiment: the first batch held back too)
        UTM_AHEAD0()
        const u64 t_due = (ahead_ticks && t_iter) ? t_rec + t_iter - (t_iter > (u64)ahead_ticks ? (u64)ahead_ticks : t_iter) : 0;
        bool second_out = false;
        if (!t_due || (u64)wall_clock64() >= t_due) {
            UTM_AHEAD1()
            second_out = true;
        }

        const unsigned epoch = (unsigned)k + 1;
        if (wave == 0) {
            LoopRec r;
            r.ok = 0;
            for (unsigned spin = 0; spin < UTM_LOOP_WAIT_SPINS; ++spin) {
                r = loop_read_record(sync, epoch);
                if (r.ok) break;
                if (!second_out && (u64)wall_clock64() >= t_due) {
                    UTM_AHEAD1()
                    second_out = true;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if (lane == 0) {
                if (!r.ok) {
                    r.stop = 1;
                    __hip_atomic_store(&sync->worker_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                rec_lds = r;
                __hip_atomic_store(&rec_epoch_lds, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // wakes the other waves
                if (blockIdx.x == 1) UTM_LSTAMP(sync, k, 3);
            }
        } else if (!second_out) {
            // (a timed wait on the wave's own clock; the record, should it come first, ends it)
            while (__hip_atomic_load(&rec_epoch_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != epoch && (u64)wall_clock64() < t_due)
                __builtin_amdgcn_s_sleep(2);
        }
        if (!second_out) { UTM_AHEAD1() }
        __syncthreads();
        const LoopRec r = rec_lds;
        if (r.stop) break;  // (uniform) the launch ends here: the winner stays pending, exactly as after a k_score_int launch
        // covered |= winner, in LDS: live &= ~winner's tile
        {
            const v4u *wc = reinterpret_cast<const v4u *>(cols + (u64)r.winner * wp + w0);
            v4u *lg = (AFM == 2 && laf.newly_log && slot == 0)
                          ? reinterpret_cast<v4u *>(laf.newly_log + (u64)((iter0 + k) % laf.log_slots) * laf.log_stride + w0) : nullptr;
            for (int kk = threadIdx.x; kk < nsteps * 64; kk += UTM_LOOP_THREADS) {
                const v4u w = wc[kk], l = live[kk];
                if (AF) newly_lds[kk] = l & w;
                if (AFM == 2 && lg) lg[kk] = l & w;  // (the winner of this launch's iteration k made row iter0 + k)
                live[kk] = l & ~w;
            }
        }
        if (r.removed) {
            n_act -= 1;
            patch_pos = r.best_pos;
            patch_s = r.moved;
            if (first == r.best_pos) {
                s_first = r.moved;
                UTM_AHEAD0()
                if (nb > 1) { UTM_AHEAD1() }
            }
            if (second == r.best_pos) {
                s_second = r.moved;
                if (nb == 1) { UTM_AHEAD1() }
            }
        }
        __syncthreads();  // the tile is whole again (and rec_lds may be rewritten)
        {
            const u64 now = (u64)wall_clock64();
            t_iter = t_rec ? now - t_rec : 0;
            t_rec = now;
        }
        if (blockIdx.x == 1 && threadIdx.x == 0) UTM_LSTAMP(sync, k, 4);
#ifdef UTM_DEBUG_STAMPS
        if (lane == 0 && k + 1 == UTM_STAMP_ITER && blockIdx.x * UTM_LOOP_WAVES + wave < 8192) sync->wave_t[0][blockIdx.x * UTM_LOOP_WAVES + wave] = (u64)wall_clock64();
#endif
    }
    // the tile's covered words go back to memory (the pending winner is NOT in them: the next launch folds it in)
    if (slot == 0) {
        v4u *cv = reinterpret_cast<v4u *>(covered + w0);
        for (int k = threadIdx.x; k < nsteps * 64; k += UTM_LOOP_THREADS) cv[k] = ~live[k];
    }
#undef UTM_AHEAD0
#undef UTM_AHEAD1
#undef UTM_PARTIAL
#undef UTM_UNIFORM
#undef UTM_ACT_LOAD
#undef UTM_COL_LOAD
#undef UTM_BATCH_LOAD
#undef UTM_BATCH_COUNT
}
