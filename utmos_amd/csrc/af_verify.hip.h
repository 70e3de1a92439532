// Verification of the parallel AF sums: intervals, candidates, the reference's sequential chains.
#pragma once
#include "score_af.hip.h"

// ------------------------------------------------------------------------------------------------
// Verified-parallel AF scoring.  The reference's AF score of a sample is a float64 running sum in
// ascending variant order (select.py:40); float64 addition does not reassociate, so a parallel sum is
// only an *estimate* E with a rigorous bound B on |reference - E|:
//   float32 AF: E = exact integer sum of AF*2^q.  While E < 2^53 every partial sum of the reference
//               is exact, hence reference == E (B = 0).  Beyond: B = n * 2^-53 * E (n addends).
//   float64 AF: E sums the float32-rounded values exactly: B = (2^-24 + n * 2^-53) * E.
//   Coarse unit (af_trunc: the table's mass times 2^q would overflow int64 at the lossless q): every addend is
//               floored to the unit, so the exact sum of the float32 values lies in [E, E + n) units; the
//               bounds above are then taken on that interval's ends, and nothing counts as exact.
// k_cand keeps the samples whose weighted interval reaches the best lower bound -- only they can be
// the argmax -- and k_chain recomputes exactly those few with the reference's sequential chain.
// Result: bit-identical winner and score, with the bulk of the work order independent.
// ------------------------------------------------------------------------------------------------
// est_exact: the ESTIMATE is the reference's sum (what IterState::all_exact is about); exact: the value returned in
// `est` is -- the estimate, or a sum a chain produced earlier for this very count (PickArgs::known_*).
// (the accumulators and the record passed in: the persistent loop's picker keeps them in registers, loop_int.hip.h)
__device__ __forceinline__ void af_interval_regs(const PickArgs &a, unsigned s, u64 c, i64 e, u64 on_record, double rec_val, double &lo, double &hi,
                                                 double &est, bool &exact, bool &est_exact)
{
    est = (double)e * a.af_scale;
    const double top = a.af_trunc ? est + (double)c * a.af_scale : est;  // upper end of the exact float32 sum
    double rel;
    if (a.af_is_f64) {
        exact = c == 0;
        rel = 1.02 * (5.9604644775390625e-08 + (double)c * 1.1102230246251565e-16);
    } else {
        exact = (e < (1ll << 53) && !a.af_trunc) || c == 0;
        rel = 1.05 * ((double)c * 1.1102230246251565e-16 + 1.2e-16);
    }
    est_exact = exact;
    if (!exact && on_record == c) {
        est = rec_val;
        exact = true;
    }
    double l = exact ? est : est - rel * est, h = exact ? est : top + rel * top;
    if (l < 0.0) l = 0.0;
    if (a.weights) {
        const double w = a.weights[a.first + s];
        l *= w;  // rounding is monotone: fl(R*w) lies between fl(l*w) and fl(h*w)
        h *= w;
        if (w < 0.0) { const double t = l; l = h; h = t; }
    }
    lo = l;
    hi = h;
}
__device__ __forceinline__ void af_interval(const PickArgs &a, unsigned s, u64 c, double &lo, double &hi, double &est,
                                            bool &exact, bool &est_exact)
{
    const u64 on_record = a.known_cnt ? a.known_cnt[s] : ~0ull;  // (requested together with the sum, not behind it)
    const double rec_val = (a.known_cnt && on_record == c) ? a.known_val[s] : 0.0;
    af_interval_regs(a, s, c, a.afsum[s], on_record, rec_val, lo, hi, est, exact, est_exact);
}

// Candidates of the iteration (see above) by ONE workgroup of 256 or 1024 threads; returns (to every thread) whether
// chains are needed.  A thread keeps up to UTM_CAND_R samples' intervals in registers between the two passes (best
// lower bound, then the list), so 2,504 samples cost one round of loads; more samples than that are read twice.
// early_pick (the only shard): when no chain is needed -- the usual case once a lone candidate's exact sum may come
// later (af_defer.hip.h) or is not wanted -- this workgroup also makes the iteration's pick, from the candidate
// list alone: a sample outside it has its whole interval below the best lower bound, so it is not the argmax.
#define UTM_CAND_R 4
struct CandScratch {  // LDS
    double wmax[16];
    Cand wbest[16];
    unsigned n_c;
    int inexact, any_inexact, zero_est, chain_needed;
};
__device__ __forceinline__ bool cand_list(const PickArgs &a, CandScratch &sc, bool early_pick, Preloaded &pre)
{
    IterState *st = a.st;
    const unsigned n_active = st->n_active;
    if (early_pick && threadIdx.x == 0) {
        pre.iter = st->iter;
        pre.tot = st->tot;
        pre.n_active_total = st->n_active_total;
        pre.last_act = n_active ? a.act[n_active - 1] : 0;
    }
    if (threadIdx.x == 0) { sc.n_c = 0; sc.inexact = 0; sc.any_inexact = 0; sc.zero_est = 0; }
    const bool in_regs = n_active <= UTM_CAND_R * blockDim.x;
    unsigned rs[UTM_CAND_R];
    u64 rc[UTM_CAND_R];
    double rlo[UTM_CAND_R], rhi[UTM_CAND_R], rest[UTM_CAND_R];
    bool rex[UTM_CAND_R], rxx[UTM_CAND_R];  // value exact / estimate exact
    double best_lo = -__builtin_inf();
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < UTM_CAND_R; ++r) {
            const unsigned i = threadIdx.x + r * blockDim.x;
            rs[r] = i < n_active ? a.act[i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < UTM_CAND_R; ++r) {
            const unsigned i = threadIdx.x + r * blockDim.x;
            rc[r] = i < n_active ? a.cnt[rs[r]] : 0ull;
        }
#pragma unroll
        for (int r = 0; r < UTM_CAND_R; ++r) {
            const unsigned i = threadIdx.x + r * blockDim.x;
            rlo[r] = rhi[r] = -__builtin_inf();
            rest[r] = 0.0;
            rex[r] = rxx[r] = true;
            if (i < n_active) af_interval(a, rs[r], rc[r], rlo[r], rhi[r], rest[r], rex[r], rxx[r]);
            best_lo = rlo[r] > best_lo ? rlo[r] : best_lo;
        }
    } else {
        for (unsigned i = threadIdx.x; i < n_active; i += blockDim.x) {
            const unsigned s = a.act[i];
            double lo, hi, est;
            bool exact, est_exact;
            af_interval(a, s, a.cnt[s], lo, hi, est, exact, est_exact);
            best_lo = lo > best_lo ? lo : best_lo;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(best_lo, o, 64);
        best_lo = other > best_lo ? other : best_lo;
    }
    if ((threadIdx.x & 63) == 0) sc.wmax[threadIdx.x >> 6] = best_lo;
    __syncthreads();
    best_lo = sc.wmax[0];
    for (unsigned w = 1; w < (blockDim.x >> 6); ++w) best_lo = fmax(best_lo, sc.wmax[w]);
    auto consider = [&](unsigned i, unsigned s, u64 c, double hi, double est, bool exact, bool est_exact) {
        if (!est_exact) sc.any_inexact = 1;
        if (hi >= best_lo) {
            const unsigned slot = atomicAdd(&sc.n_c, 1u);
            if (slot < UTM_MAX_CAND) {
                a.cand->pos[slot] = i;
                a.cand->samp[slot] = s;
                a.cand->cnt[slot] = (i64)c;
                a.cand->val[slot] = est;
            }
            if (!exact) sc.inexact = 1;
            if (!exact && est == 0.0) sc.zero_est = 1;  // coarse unit: every addend floored away, yet the true score is > 0
        }
    };
    if (in_regs) {
#pragma unroll
        for (int r = 0; r < UTM_CAND_R; ++r) {
            const unsigned i = threadIdx.x + r * blockDim.x;
            if (i < n_active) consider(i, rs[r], rc[r], rhi[r], rest[r], rex[r], rxx[r]);
        }
    } else {
        for (unsigned i = threadIdx.x; i < n_active; i += blockDim.x) {
            const unsigned s = a.act[i];
            const u64 c = a.cnt[s];
            double lo, hi, est;
            bool exact, est_exact;
            af_interval(a, s, c, lo, hi, est, exact, est_exact);
            consider(i, s, c, hi, est, exact, est_exact);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n_c = sc.n_c;
        st->n_cand = n_c < UTM_MAX_CAND ? (int)n_c : UTM_MAX_CAND;
        st->cand_overflow = n_c > UTM_MAX_CAND;
        // one candidate only: the argmax is settled (its estimate is also the largest); its exact float64 sum is
        // needed just for the reported score, which the caller may not want
        // (... unless its estimate is 0 while it has addends: the stop rule compares the score with 0, select.py:51)
        sc.chain_needed = sc.inexact && (sc.zero_est || !(a.af_skip_single && n_c == 1));
        st->need_chain = sc.chain_needed;
        st->chain_events += sc.chain_needed ? 1 : 0;
        st->all_exact = !sc.any_inexact;
    }
    __syncthreads();
    return sc.chain_needed != 0;
}

// The pick of the only shard from the candidate list alone (1 <= n_c <= UTM_MAX_CAND entries, values final): mask /
// weight / argmax as in pick_body, one lane of wave 0 per candidate; thread 0 decides.  OTHERS_WROTE: the list's
// writers are other workgroups of the same launch (k_verify) -- it is read with agent-scope loads.
template <bool OTHERS_WROTE>
__device__ __forceinline__ void pick_among_candidates(const PickArgs &a, unsigned n_c, unsigned n_active, const Preloaded &pre)
{
    if (threadIdx.x >= 64) return;
    Cand best{-__builtin_inf(), INT64_MAX, 0, 0};
    if (threadIdx.x < n_c) {
        const unsigned b = threadIdx.x;
        unsigned s, pos;
        i64 c;
        double v;
        if (OTHERS_WROTE) {
            s = __hip_atomic_load(&a.cand->samp[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pos = __hip_atomic_load(&a.cand->pos[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = (i64)__hip_atomic_load(reinterpret_cast<const u64 *>(&a.cand->cnt[b]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const u64 *>(&a.cand->val[b]), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
        } else {
            s = a.cand->samp[b];
            pos = a.cand->pos[b];
            c = a.cand->cnt[b];
            v = a.cand->val[b];
        }
        if (a.weights) v *= a.weights[a.first + s];
        best = Cand{v, (i64)a.first + s, c, pos};
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const Cand other = shfl_cand(best, o);
        if (better(other, best)) best = other;
    }
    if (threadIdx.x == 0) {
        Rec *rc0 = rec_of(a, a.rank);
        rc0->score = best.val;
        rc0->idx = best.gidx;
        rc0->new_count = best.cnt;
        a.st->best_pos = best.pos;
        decide_single(a, best, n_active, pre);
    }
}

// ... and the pick when cand_list found that no chain is needed (early_pick).
__device__ __forceinline__ void cand_pick(const PickArgs &a, CandScratch &sc, const Preloaded &pre)
{
    IterState *st = a.st;
    const unsigned n_active = st->n_active;
    const unsigned n_c = sc.n_c;
    if (n_c == 0 || n_c > UTM_MAX_CAND || a.list_n) {  // (no list to pick from / decremental bookkeeping: the general pick)
        pick_body<0>(a);
        return;
    }
    pick_among_candidates<false>(a, n_c, n_active, pre);
}

__global__ __launch_bounds__(1024) void k_cand(PickArgs a)
{
    __shared__ CandScratch sc;
    if (a.st->done) return;
    Preloaded pre{0, 0, 0, 0};
    const bool chain_needed = cand_list(a, sc, a.early_pick != 0, pre);
    if (a.early_pick && !chain_needed) cand_pick(a, sc, pre);
}


// The reference's chain for ONE candidate per workgroup: 1024 lanes compact the AF values of the
// candidate's surviving bits, in ascending variant order, into LDS (popcount -> block prefix sum ->
// scatter); lane 0 then adds them one by one in float64.  Only the additions are serial.
// Strictly ordered float64 sum of the 64 values a wave holds (lane i = i-th addend): every lane reads the addends
// one after the other with v_readlane (no memory in the dependent chain) and all lanes keep the same running sum.
// Lanes past the end of a list must hold +0.0, which leaves the sum unchanged bit for bit.
__device__ __forceinline__ double ordered_sum64(double acc, double v)
{
    const int lo = (int)(__builtin_bit_cast(u64, v) & 0xFFFFFFFFu), hi = (int)(__builtin_bit_cast(u64, v) >> 32);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const u64 bits = ((u64)(unsigned)__builtin_amdgcn_readlane(hi, i) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, i);
        acc += __builtin_bit_cast(double, bits);
    }
    return acc;
}

// Fast path of the chains (sparse candidates, i.e. almost every iteration after the first few): the
// candidate's column is cut into segments of 4096 words; k_chain_fill lets one workgroup per
// (segment, candidate) compact the AF values of the surviving bits, in order, into a global buffer;
// k_chain's wave 0 then only walks the per-segment counts and adds the values in order.  A segment
// with more than UTM_SEG_CAP values, or more than UTM_FAST_CAND candidates, leaves the candidate to
// the one-workgroup chain below.
#define UTM_FAST_CAND 8
#define UTM_SEG_CAP 1024
#define UTM_SEG_WORDS 4096
#define UTM_SEG_FULL (UTM_SEG_WORDS * 64)  // every bit of a segment set
struct ChainSeg {
    int chunk;
    u64 w0;
    u64 off;  // the segment's first word inside a whole-column buffer (chunks back to back)
};
struct ChainFast {
    const ChainSeg *segs;
    int n_segs;
    unsigned *counts;  // [n_cand][n_segs]; 0xFFFFFFFF = more values than seg_cap in that segment
    double *vals;      // [n_cand][n_segs][seg_cap]
    unsigned seg_cap;  // values a segment's region holds: UTM_SEG_FULL (every segment fits) when the memory allows, else UTM_SEG_CAP
    int n_cand;        // candidates with buffers (<= UTM_FAST_CAND)
};

// (seg, ci): the segment and the candidate this workgroup of 1024 compacts; wtot: 16 words of LDS
template <typename AF_T>
__device__ __forceinline__ void chain_fill_block(const SeqChunk *__restrict__ chunks, const CandBuf *__restrict__ cand,
                                                 const ChainFast &f, unsigned seg, unsigned ci, unsigned *wtot)
{
    // (k_verify: the list was written, and the addends are read, by workgroups that may sit on another XCD -- the
    // sample index comes through an agent-scope load, counts and values leave through agent-scope stores)
    const ChainSeg sg = f.segs[seg];
    const SeqChunk ch = chunks[sg.chunk];
    const unsigned s = __hip_atomic_load(&cand->samp[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 *col = ch.cols + (u64)s * ch.wp;
    const AF_T *af = static_cast<const AF_T *>(ch.af);
    const u64 w = sg.w0 + (u64)tid * 4;
    u64 x[4];
    unsigned n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = w + k < ch.w ? (col[w + k] & ~ch.covered[w + k]) : 0;
        n += __popcll(x[k]);
    }
    const unsigned incl = wave_scan_incl_u32(n);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned woff = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned t = wtot[k];
        woff += k < wave ? t : 0;
        total += t;
    }
    const size_t slot = (size_t)ci * f.n_segs + seg;
    if (tid == 0) __hip_atomic_store(&f.counts[slot], total <= f.seg_cap ? total : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (total == 0 || total > f.seg_cap) return;
    double *out = f.vals + slot * f.seg_cap + (woff + incl - n);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u64 y = x[k];
        while (y) {
            const int b = __builtin_ctzll(y);
            y &= y - 1;
            __hip_atomic_store(reinterpret_cast<u64 *>(out++), __builtin_bit_cast(u64, (double)af[(w + k) * 64 + b]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename AF_T>
__global__ __launch_bounds__(1024) void k_chain_fill(const SeqChunk *__restrict__ chunks, const IterState *__restrict__ st,
                                                     const CandBuf *__restrict__ cand, ChainFast f)
{
    __shared__ unsigned wtot[16];
    if (st->done || !st->need_chain || st->cand_overflow || st->n_cand > f.n_cand || (int)blockIdx.y >= st->n_cand) return;
    chain_fill_block<AF_T>(chunks, cand, f, blockIdx.x, blockIdx.y, wtot);
}


// ------------------------------------------------------------------------------------------------
// The reference's chain, in parallel and still bit for bit.  acc_{i+1} = RN(acc_i + a_i) in float64, round to
// nearest even, ascending variant order (select.py:40), all a_i >= 0.  While the running sum stays inside one
// binade [2^e, 2^(e+1)) it is an integer M in [2^52, 2^53) times the unit U = 2^(e-52), and adding a_i adds the
// INTEGER q_i = a_i / U rounded to nearest -- which depends on a_i alone -- except for exact ties
// (a_i / U = k + 1/2), which go to whichever of k, k + 1 makes M even: they depend on the parity of M, and leave it
// even.  So the parity of M is a two-state machine driven by the addends (a tie resets it, anything else XORs it
// with q_i's low bit), its state in front of every addend comes out of a prefix scan over 2-bit maps, every q_i
// follows, and a second prefix scan (of the q_i) says where M would reach 2^53.  Up to there the window is
// finished in one go: acc = (M + sum q) * U, exactly the value the sequential additions produce.  The addend that
// crosses into the next binade is added with one real float64 addition, and the scan restarts behind it with the
// new unit.  1024 threads x 4 addends per round; a sum crosses ~log2(n) binades, so rounds ~ n / 4096 + log2(n)
// instead of n dependent additions.  Integer-valued doubles below 2^53 carry M, q and their partial sums exactly;
// a partial sum that reaches 2^53 may round, but never back below 2^53, which is all the crossing test needs.
// ------------------------------------------------------------------------------------------------
#define UTM_PAR_E 4
#ifndef UTM_PAR_HEAD
#define UTM_PAR_HEAD 256
#endif
#define UTM_PAR_MAX_SEGS 4095
__device__ __forceinline__ unsigned pm_then(unsigned first, unsigned second)  // maps on {0,1}: bit p = image of p
{
    return ((second >> (first & 1)) & 1) | (((second >> ((first >> 1) & 1)) & 1) << 1);
}
__device__ __forceinline__ int low_bit(double integer_valued) { return (int)(integer_valued - 2.0 * floor(0.5 * integer_valued)); }

// Inclusive wave scans on the DPP ladder of wave_scan_incl_u32 (row shifts, then row broadcasts): a lane without a
// source keeps the operator's identity, and the earlier lanes' value is always the first operand, so the ladder
// also serves the (associative, non-commutative) composition of parity maps.
#define UTM_DPP_U32(old, v, ctrl, rmask) (unsigned)__builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rmask, 0xf, false)
__device__ __forceinline__ unsigned wave_scan_maps(unsigned m)
{
    m = pm_then(UTM_DPP_U32(2u, m, 0x111, 0xf), m);  // row_shr:1
    m = pm_then(UTM_DPP_U32(2u, m, 0x112, 0xf), m);  // row_shr:2
    m = pm_then(UTM_DPP_U32(2u, m, 0x114, 0xf), m);  // row_shr:4
    m = pm_then(UTM_DPP_U32(2u, m, 0x118, 0xf), m);  // row_shr:8
    m = pm_then(UTM_DPP_U32(2u, m, 0x142, 0xa), m);  // row_bcast:15 -> rows 1, 3
    m = pm_then(UTM_DPP_U32(2u, m, 0x143, 0xc), m);  // row_bcast:31 -> rows 2, 3
    return m;
}
__device__ __forceinline__ double dpp_f64(double v, const int ctrl, const int rmask)
{
    const u64 b = __builtin_bit_cast(u64, v);
    unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    switch (ctrl) {  // (the control word must be a compile-time constant)
    case 0x111: lo = UTM_DPP_U32(0u, lo, 0x111, 0xf); hi = UTM_DPP_U32(0u, hi, 0x111, 0xf); break;
    case 0x112: lo = UTM_DPP_U32(0u, lo, 0x112, 0xf); hi = UTM_DPP_U32(0u, hi, 0x112, 0xf); break;
    case 0x114: lo = UTM_DPP_U32(0u, lo, 0x114, 0xf); hi = UTM_DPP_U32(0u, hi, 0x114, 0xf); break;
    case 0x118: lo = UTM_DPP_U32(0u, lo, 0x118, 0xf); hi = UTM_DPP_U32(0u, hi, 0x118, 0xf); break;
    case 0x142: lo = UTM_DPP_U32(0u, lo, 0x142, 0xa); hi = UTM_DPP_U32(0u, hi, 0x142, 0xa); break;
    default: lo = UTM_DPP_U32(0u, lo, 0x143, 0xc); hi = UTM_DPP_U32(0u, hi, 0x143, 0xc); break;
    }
    return __builtin_bit_cast(double, ((u64)hi << 32) | lo);
}
__device__ __forceinline__ double wave_scan_f64(double v)  // +0.0 is the identity
{
    v += dpp_f64(v, 0x111, 0xf);
    v += dpp_f64(v, 0x112, 0xf);
    v += dpp_f64(v, 0x114, 0xf);
    v += dpp_f64(v, 0x118, 0xf);
    v += dpp_f64(v, 0x142, 0xa);
    v += dpp_f64(v, 0x143, 0xc);
    return v;
}

struct ParScratch {       // LDS
    unsigned wmap[16];
    double wsum[16];
    unsigned wcross[16];
    double crossed;       // the running sum right after the crossing addend
};

// Where a chain's addends come from.  at(g, hint): addend g (g < total), `hint` a cursor the source may advance
// (callers walk g upwards); first(g): a cursor for addend g.
struct SegmentedAddends {  // k_chain_fill's layout: segment i's addends at vals + i * seg_cap; offs (LDS): exclusive
    const double *vals;    // prefix of the segments' counts, n_segs + 1 entries
    size_t seg_cap;
    const unsigned *offs;
    int n_segs;
    __device__ __forceinline__ unsigned total() const { return offs[n_segs]; }
    __device__ __forceinline__ int first(unsigned g) const
    {  // binary search: last segment whose offset is <= g
        int lo = 0, hi = n_segs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offs[mid] <= g) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    }
    __device__ __forceinline__ double at(unsigned g, int &seg) const
    {
        while (g >= offs[seg + 1]) ++seg;
        return vals[(size_t)seg * seg_cap + (g - offs[seg])];
    }
};
struct FlatAddends {  // one contiguous array
    const double *vals;
    unsigned n;
    __device__ __forceinline__ unsigned total() const { return n; }
    __device__ __forceinline__ int first(unsigned) const { return 0; }
    __device__ __forceinline__ double at(unsigned g, int &) const { return vals[g]; }
};

// Every thread of the 1024 calls this; every thread returns the sum.
template <class ADDENDS>
__device__ __forceinline__ double chain_parallel(const ADDENDS &src, ParScratch &sc)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned total = src.total();
    // a window's addends: UTM_PAR_E consecutive ones per thread
    auto load_window = [&](unsigned base, double *a) {
        const unsigned g0 = base + (unsigned)tid * UTM_PAR_E;
        int seg = g0 < total ? src.first(g0) : 0;
#pragma unroll
        for (int j = 0; j < UTM_PAR_E; ++j) a[j] = g0 + j < total ? src.at(g0 + j, seg) : 0.0;
    };
    double a_next[UTM_PAR_E];
    load_window(0, a_next);  // (in flight while the head is added up)
    // The first UTM_PAR_HEAD addends one by one (wave 0, registers only): a sum leaves a binade every few addends at
    // its start, and every crossing would cost the scan a round of its own.  All of them are requested first.
    const unsigned head = total < UTM_PAR_HEAD ? total : UTM_PAR_HEAD;
    if (wave == 0) {
        double hv[UTM_PAR_HEAD / 64];
        int seg = 0;
#pragma unroll
        for (int r = 0; r < UTM_PAR_HEAD / 64; ++r) hv[r] = r * 64 + lane < (int)head ? src.at(r * 64 + lane, seg) : 0.0;
        double first = 0.0;
#pragma unroll
        for (int r = 0; r < UTM_PAR_HEAD / 64; ++r)
            if (r * 64 < (int)head) first = ordered_sum64(first, hv[r]);  // (wave uniform)
        if (lane == 0) sc.crossed = first;
    }
    __syncthreads();
    double acc = sc.crossed;
    __syncthreads();
    for (unsigned base = 0; base < total; base += 1024u * UTM_PAR_E) {
        // this window's addends stay in registers until the window is used up: a round that ends at a binade crossing
        // only moves `done` (addends in front of it count as 0 from then on) and re-runs the scans with the new unit.
        // The next window's are requested meanwhile.
        const unsigned g0 = base + (unsigned)tid * UTM_PAR_E;
        double a[UTM_PAR_E];
#pragma unroll
        for (int j = 0; j < UTM_PAR_E; ++j) a[j] = a_next[j];
        if (base + 1024u * UTM_PAR_E < total) load_window(base + 1024u * UTM_PAR_E, a_next);
        const unsigned end = total - base > 1024u * UTM_PAR_E ? base + 1024u * UTM_PAR_E : total;
        unsigned done = base > head ? base : head;  // addends [base, done) are in acc already
        while (done < end) {
            const int e = (int)((__builtin_bit_cast(u64, acc) >> 52) & 0x7FF) - 1023;
            const int scale = 52 - e;
            const double m0 = ldexp(acc, scale);                 // M: integer in [2^52, 2^53)
            const double limit = 9007199254740992.0 - m0;        // the sum of q reaching this = leaving the binade
            double k[UTM_PAR_E];
            int tie[UTM_PAR_E];
            unsigned map = 2u;  // identity
#pragma unroll
            for (int j = 0; j < UTM_PAR_E; ++j) {
                const double x = g0 + j >= done ? ldexp(a[j], scale) : 0.0;  // a / U, exact (a power-of-two scaling)
                const double fl = floor(x);
                const double r = x - fl;                         // exact
                tie[j] = r == 0.5;
                k[j] = fl + (r > 0.5 ? 1.0 : 0.0);               // q unless a tie; a tie: k or k + 1
                map = pm_then(map, tie[j] ? 0u : (low_bit(k[j]) ? 1u : 2u));
            }
            const unsigned inc = wave_scan_maps(map);  // inclusive scan of the maps over the wave, then over the waves
            if (lane == 63) sc.wmap[wave] = inc;
            __syncthreads();
            unsigned before = 2u;
            for (int w = 0; w < wave; ++w) before = pm_then(before, sc.wmap[w]);
            unsigned excl = __shfl_up(inc, 1, 64);
            if (lane == 0) excl = 2u;
            int par = (pm_then(before, excl) >> low_bit(m0)) & 1;  // parity of M in front of this thread's first addend
            double q[UTM_PAR_E], mine = 0.0;
#pragma unroll
            for (int j = 0; j < UTM_PAR_E; ++j) {
                if (tie[j]) {
                    q[j] = k[j] + (double)((par + low_bit(k[j])) & 1);  // ... to the even neighbour
                    par = 0;
                } else {
                    q[j] = k[j];
                    par ^= low_bit(k[j]);
                }
                mine += q[j];
            }
            const double incs = wave_scan_f64(mine);
            if (lane == 63) sc.wsum[wave] = incs;
            __syncthreads();
            double in_front = 0.0, all = 0.0;
            for (int w = 0; w < 16; ++w) {
                const double t = sc.wsum[w];
                in_front += w < wave ? t : 0.0;
                all += t;
            }
            if (all < limit) {  // the rest of the window stays in the binade (uniform: every thread sees the same total)
                acc = ldexp(m0 + all, -scale);
                done = end;
                continue;
            }
            // the first addend whose q takes M to 2^53 or beyond is added for real.  (The sum in front of this thread's
            // addends comes from the previous lane's inclusive value, never from `incs - mine`: sums past 2^53 are no
            // longer exact, and this thread's own addends may be what takes them there.)
            double lanes_before = __shfl_up(incs, 1, 64);
            if (lane == 0) lanes_before = 0.0;
            double run = in_front + lanes_before, at = 0.0, addend = 0.0;
            unsigned first = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < UTM_PAR_E; ++j) {
                if (first == 0xFFFFFFFFu && run + q[j] >= limit) {
                    first = g0 + j;
                    at = run;
                    addend = a[j];
                }
                run += q[j];
            }
            unsigned wmin = first;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned other = __shfl_xor(wmin, o, 64);
                wmin = other < wmin ? other : wmin;
            }
            if (lane == 0) sc.wcross[wave] = wmin;
            __syncthreads();
            unsigned where = 0xFFFFFFFFu;
            for (int w = 0; w < 16; ++w) where = sc.wcross[w] < where ? sc.wcross[w] : where;
            if (first == where) sc.crossed = ldexp(m0 + at, -scale) + addend;  // (exactly one thread owns that addend)
            __syncthreads();
            acc = sc.crossed;
            done = where + 1;
        }
    }
    return acc;
}

#define UTM_CHAIN_CAP 2048
#define UTM_CHAIN_WPT 4  // consecutive words per lane and round: 4096 words (262,144 variants) per round
// Blocks [0, UTM_MAX_CAND): one candidate each.  Blocks behind them: only when more than UTM_MAX_CAND candidates tie
// within the error bound (cand_overflow) -- then every selectable sample is re-scored with the plain sequential
// chain, one lane per sample (what a k_score_seq launch of its own did, at a launch per iteration).
template <typename AF_T>
__device__ __forceinline__ void chain_block(const SeqChunk *__restrict__ chunks, int n_chunks, const IterState *__restrict__ st,
                                            CandBuf *__restrict__ cand, const ChainFast &f, const unsigned *__restrict__ act,
                                            u64 *__restrict__ cnt, double *__restrict__ fscore, unsigned bx, bool need_chain,
                                            bool overflow, int n_cand, u64 *__restrict__ known_cnt, double *__restrict__ known_val);

// PICK: the iteration's pick (k_pick<0>'s body) runs in whichever workgroup of this launch finishes last -- one launch
// less per iteration wherever candidates are verified (float64 AF: every iteration).  Arrival is one returning atomic
// per workgroup (67-odd of them); chain results are published with agent-scope stores and read back the same way.
template <typename AF_T, bool PICK>
__global__ __launch_bounds__(1024) void k_chain(const SeqChunk *__restrict__ chunks, int n_chunks,
                                                const IterState *__restrict__ st, CandBuf *__restrict__ cand, ChainFast f,
                                                const unsigned *__restrict__ act, u64 *__restrict__ cnt, double *__restrict__ fscore,
                                                const PickArgs pa, unsigned *__restrict__ arrivals)
{
    if (st->done) return;  // (uniform over the launch: nobody arrives, nobody picks)
    if (PICK && pa.early_pick && !st->need_chain) return;  // (... and so is this: k_cand has made the pick already)
    chain_block<AF_T>(chunks, n_chunks, st, cand, f, act, cnt, fscore, blockIdx.x, st->need_chain != 0, st->cand_overflow != 0, st->n_cand,
                      pa.known_cnt, pa.known_val);
    if (PICK) {
        __shared__ int last;
        __syncthreads();
        if (threadIdx.x == 0) {
            // this workgroup's results (cand->val by an agent-scope store; the overflow blocks' scores by plain stores
            // behind a fence) are out before it arrives
            if (blockIdx.x >= UTM_MAX_CAND) __threadfence();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == gridDim.x - 1;
            if (last) __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (last) {
            if (st->need_chain && st->cand_overflow) __threadfence();  // (rare: every sample was re-scored by other workgroups)
            pick_body<0, true>(pa);
        }
    }
}

template <typename AF_T>
__device__ __forceinline__ void chain_block(const SeqChunk *__restrict__ chunks, int n_chunks, const IterState *__restrict__ st,
                                            CandBuf *__restrict__ cand, const ChainFast &f, const unsigned *__restrict__ act,
                                            u64 *__restrict__ cnt, double *__restrict__ fscore, unsigned bx, bool need_chain,
                                            bool overflow, int n_cand, u64 *__restrict__ known_cnt, double *__restrict__ known_val)
{
    __shared__ double buf[UTM_CHAIN_CAP];
    __shared__ unsigned wtot[2][16];  // double buffered: one barrier per empty round
    __shared__ int dense;
    if (!need_chain) return;
    if (bx >= UTM_MAX_CAND) {
        const unsigned i = (bx - UTM_MAX_CAND) * 1024 + threadIdx.x;
        if (overflow && i < st->n_active) seq_score_sample<AF_T>(chunks, n_chunks, act[i], cnt, fscore);
        return;
    }
    if (overflow || (int)bx >= n_cand) return;
    const unsigned s = cand->samp[bx];
    const u64 s_cnt = (u64)cand->cnt[bx];
    // (a sum on record for this count: k_cand has put it into the list already)
    if (known_cnt && known_cnt[s] == s_cnt) return;
    auto put_on_record = [&](double sum) {  // thread 0
        __hip_atomic_store(reinterpret_cast<u64 *>(&cand->val[bx]), __builtin_bit_cast(u64, sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (known_cnt) {
            known_val[s] = sum;
            known_cnt[s] = s_cnt;
        }
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (f.counts && n_cand <= f.n_cand && f.n_segs <= UTM_PAR_MAX_SEGS) {
        // k_chain_fill compacted this candidate's addends per segment: if every segment fitted its region, all 1024
        // threads run the parallel form of the chain over them
        __shared__ ParScratch sc;
        const unsigned *cnts = f.counts + (size_t)bx * f.n_segs;
        unsigned *offs = reinterpret_cast<unsigned *>(buf);  // n_segs + 1 <= 4096 entries
        if (tid == 0) dense = 0;
        __syncthreads();
        // exclusive prefix of the segment counts: 4 consecutive segments per thread
        unsigned c4[4], mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = tid * 4 + j;
            c4[j] = g < f.n_segs ? cnts[g] : 0u;
            if (c4[j] == 0xFFFFFFFFu) dense = 1;
            mine += c4[j];
        }
        const unsigned incl = wave_scan_incl_u32(mine);
        if (lane == 63) wtot[0][wave] = incl;
        __syncthreads();
        if (!dense) {
            unsigned off = incl - mine;
            for (int w = 0; w < wave; ++w) off += wtot[0][w];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int g = tid * 4 + j;
                if (g <= f.n_segs) offs[g] = off;  // (g == n_segs: the total)
                off += c4[j];
            }
            __syncthreads();
            const SegmentedAddends src{f.vals + (size_t)bx * f.n_segs * f.seg_cap, f.seg_cap, offs, f.n_segs};
            const double sum = chain_parallel(src, sc);
            if (tid == 0) put_on_record(sum);
            return;
        }
        __syncthreads();  // (dense: the one-workgroup chain below reuses buf)
    }
    double acc = 0.0;
    unsigned round = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const SeqChunk ch = chunks[c];
        const u64 *col = ch.cols + (u64)s * ch.wp;
        const AF_T *af = static_cast<const AF_T *>(ch.af);
        // wp is a multiple of 128 words, so whole groups of UTM_CHAIN_WPT words never straddle its end
        auto fetch = [&](u64 w, u64 *x) {
#pragma unroll
            for (int k = 0; k < UTM_CHAIN_WPT; ++k) x[k] = w + k < ch.w ? (col[w + k] & ~ch.covered[w + k]) : 0;
        };
        u64 x_next[UTM_CHAIN_WPT];
        fetch((u64)tid * UTM_CHAIN_WPT, x_next);
        for (u64 w0 = 0; w0 < ch.w; w0 += 1024 * UTM_CHAIN_WPT, ++round) {
            const u64 w = w0 + (u64)tid * UTM_CHAIN_WPT;
            u64 x[UTM_CHAIN_WPT];
            unsigned n = 0;
#pragma unroll
            for (int k = 0; k < UTM_CHAIN_WPT; ++k) {
                x[k] = x_next[k];
                n += __popcll(x[k]);
            }
            fetch(w + 1024 * UTM_CHAIN_WPT, x_next);  // in flight during this round
            const unsigned incl = wave_scan_incl_u32(n);
            unsigned *wt = wtot[round & 1];
            if (lane == 63) wt[wave] = incl;
            __syncthreads();
            unsigned woff = 0, total = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned t = wt[k];
                woff += k < wave ? t : 0;
                total += t;
            }
            if (total == 0) continue;  // the other wtot buffer is written next round
            const unsigned off = woff + incl - n;
            for (unsigned base = 0; base < total; base += UTM_CHAIN_CAP) {
                unsigned p = off;
#pragma unroll
                for (int k = 0; k < UTM_CHAIN_WPT; ++k) {
                    u64 y = x[k];
                    while (y) {
                        const int b = __builtin_ctzll(y);
                        y &= y - 1;
                        if (p >= base && p < base + UTM_CHAIN_CAP) buf[p - base] = (double)af[(w + k) * 64 + b];
                        ++p;
                    }
                }
                __syncthreads();
                if (wave == 0) {
                    const unsigned m = total - base < UTM_CHAIN_CAP ? total - base : UTM_CHAIN_CAP;
                    for (unsigned t = 0; t < m; t += 64) acc = ordered_sum64(acc, t + lane < m ? buf[t + lane] : 0.0);
                }
                __syncthreads();
            }
        }
    }
    if (tid == 0) put_on_record(acc);
}

// ------------------------------------------------------------------------------------------------
// The whole verification of an iteration in ONE launch (the only shard): candidates -> their compacted addends ->
// their chains -> the pick, as stages of one grid instead of three dependent launches (a dependent launch costs
// ~4.4 us before its first instruction, needed or not).
//   workgroup 0                      cand_list; publishes (launch number, chains needed?, how many candidates) in ONE
//                                    64-bit word; when no chain is needed it goes on to make the pick itself
//   workgroups [1, 1 + n_fill)       wait for that word; nothing to do -> return; else chain_fill_block for their
//                                    (segment, candidate) and a count on the candidate's counter
//   the last 64 + overflow ones      wait for the word, then (fast path) for their candidate's counter to show every
//                                    segment; chain_block; the last one to arrive makes the pick (as k_chain<PICK>)
// A stage only ever waits for workgroups with LOWER indices, and workgroups start in index order on each XCD, so the
// ones waited for are running or done whatever else occupies the device; every wait is bounded all the same (a
// time-out ends the loop with IterState::xerror = 2, which the host reports).
// ------------------------------------------------------------------------------------------------
struct VerifySync {  // device memory, zeroed by utm_reset; self-resetting from launch to launch
    u64 stage1;      // [63:32] launch number, [17] chains needed, [16] candidate overflow, [15:0] candidates
    unsigned filled[UTM_FAST_CAND];
    unsigned arrivals;
    unsigned pad_;
#ifdef UTM_DEBUG_STAMPS
    u64 stamps[16];
#endif
};
#ifdef UTM_DEBUG_STAMPS
#define UTM_STAMP(i) do { if (threadIdx.x == 0) vs->stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define UTM_STAMP(i) do { } while (0)
#endif
#define UTM_VERIFY_SPINS (1u << 22)

// Memory order across workgroups (they may sit on different XCDs, each with its own L2): the stage words are read
// and written with agent-scope atomics only, and nothing else is exchanged while no chain is needed -- the common
// case costs no cache maintenance at all.  When chains are needed, a writer stage ends with: every wave's stores
// acknowledged (s_waitcnt), barrier, ONE agent-scope release fence (L2 write-back) by thread 0, then the counter /
// word; a reader stage starts with: the word seen, ONE agent-scope acquire fence (invalidate) by thread 0, barrier.
__device__ __forceinline__ void stage_release()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
__device__ __forceinline__ void stage_acquire()
{
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
}

template <typename AF_T>
__global__ __launch_bounds__(1024) void k_verify(const SeqChunk *__restrict__ chunks, int n_chunks, IterState *__restrict__ st,
                                                 CandBuf *__restrict__ cand, ChainFast f, const unsigned *__restrict__ act,
                                                 u64 *__restrict__ cnt, double *__restrict__ fscore, const PickArgs pa,
                                                 VerifySync *__restrict__ vs, unsigned launch_no, unsigned n_fill)
{
    __shared__ CandScratch sc;  // (workgroup 0; the fill workgroups use its first words)
    __shared__ u64 word;
    __shared__ int last;
    if (st->done) return;  // (set only by a pick: before this launch, or by this launch's own once nobody has work left)
    if (blockIdx.x == 0) {
        UTM_STAMP(0);
        Preloaded pre{0, 0, 0, 0};
        const bool chain_needed = cand_list(pa, sc, true, pre);
        UTM_STAMP(1);
        if (chain_needed) stage_release();
        UTM_STAMP(2);  // (the candidate list, IterState's fields: only chains read them)
        if (threadIdx.x == 0)
            __hip_atomic_store(&vs->stage1, ((u64)launch_no << 32) | (chain_needed ? 1u << 17 : 0u) | (st->cand_overflow ? 1u << 16 : 0u) |
                                                (unsigned)st->n_cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!chain_needed) cand_pick(pa, sc, pre);
        return;
    }
    // (what the pick will need from IterState: nothing of it changes before the pick itself, and here it loads while
    // the workgroup waits; only the chain workgroups can end up making the pick)
    Preloaded pre{0, 0, 0, 0};
    unsigned n_active_pre = 0;
    if (threadIdx.x == 0 && blockIdx.x > n_fill) {
        n_active_pre = st->n_active;
        pre.iter = st->iter;
        pre.tot = st->tot;
        pre.n_active_total = st->n_active_total;
        pre.last_act = n_active_pre ? act[n_active_pre - 1] : 0;
    }
    if (threadIdx.x == 0) {
        u64 w = 0;
        unsigned spin = 0;
        for (; spin < UTM_VERIFY_SPINS; ++spin) {
            w = __hip_atomic_load(&vs->stage1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == launch_no) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (spin == UTM_VERIFY_SPINS) {
            st->xerror = 2;
            st->done = 1;
            w = 0;  // (nothing to do)
        }
        word = w;
    }
    __syncthreads();
    const u64 w1 = word;
    const bool chain_needed = (w1 >> 17) & 1, overflow = (w1 >> 16) & 1;
    const int n_cand = (int)(w1 & 0xFFFF);
    if (!chain_needed) return;
    const bool fast = f.counts && !overflow && n_cand <= f.n_cand;
    if (blockIdx.x <= n_fill) {
        const unsigned j = blockIdx.x - 1, seg = j % (unsigned)f.n_segs, ci = j / (unsigned)f.n_segs;
        if (!fast || (int)ci >= n_cand) return;
        if (seg == 0 && ci == 0) UTM_STAMP(3);
        chain_fill_block<AF_T>(chunks, cand, f, seg, ci, reinterpret_cast<unsigned *>(&sc));  // (coherent loads / stores inside)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (seg == 0 && ci == 0) UTM_STAMP(4);
        // (test hook, PickArgs::test_drop: one segment's workgroup does not count in -- the chain's bounded wait runs out)
        if (threadIdx.x == 0 && !(pa.test_drop && seg == 0 && ci == 0))
            __hip_atomic_fetch_add(&vs->filled[ci], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const unsigned bx = blockIdx.x - 1 - n_fill, n_chain_blocks = gridDim.x - 1 - n_fill;
    const bool works = bx >= UTM_MAX_CAND ? overflow : (!overflow && (int)bx < n_cand);
    if (works && bx == 0) UTM_STAMP(5);
    if (works) {
        if (fast) {  // this candidate's addends: every segment's workgroup has counted in
            if (threadIdx.x == 0) {
                unsigned spin = 0;
                for (; spin < UTM_VERIFY_SPINS; ++spin) {
                    if (__hip_atomic_load(&vs->filled[bx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)f.n_segs) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (spin == UTM_VERIFY_SPINS) st->xerror = 2;  // (the pick below still ends the iteration; the host reports the error)
                __hip_atomic_store(&vs->filled[bx], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        if (bx == 0) UTM_STAMP(6);
        stage_acquire();  // the candidate list, IterState's fields, the compacted addends
        if (bx == 0) UTM_STAMP(7);
    }
    if (works) chain_block<AF_T>(chunks, n_chunks, st, cand, f, act, cnt, fscore, bx, true, overflow, n_cand, pa.known_cnt, pa.known_val);
    if (works && bx == 0) UTM_STAMP(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every wave's stores are acknowledged before thread 0 reports in)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (works && bx >= UTM_MAX_CAND) __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(&vs->arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == n_chain_blocks - 1;
        if (last) __hip_atomic_store(&vs->arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last) {
        UTM_STAMP(9);
        if (overflow || pa.list_n) {
            stage_acquire();  // (workgroup 0's IterState fields and list; the overflow workgroups' scores)
            pick_body<0, true>(pa);
        } else {
            pick_among_candidates<true>(pa, (unsigned)n_cand, n_active_pre, pre);  // (the chains' values: agent-scope stores)
        }
        UTM_STAMP(11);
    }
}
