// K1, allele-frequency weighted scores: dense LDS-tile kernel, streaming kernel (full and delta passes), sequential chain.
#pragma once
#include "score_int.hip.h"

// Fixed-point AF table entry (built by the host, host_options.hip.h): [31:24] shift, [23:0] mantissa -- the value
// floor(AF * 2^q) = mantissa << shift as int64.  At the lossless q (the usual case) the mantissa is the float32's 24
// bits and the shift its exponent above the unit; a table whose mass would overflow int64 there gets a coarser unit
// and its small values arrive already floored (shift 0) -- the interval arithmetic of af_verify.hip.h accounts for
// the < 1 unit each addend may lose.  Two VALU ops per set bit.
__device__ __forceinline__ u64 af_fixed(unsigned f)
{
    return (u64)(f & 0xFFFFFFu) << (f >> 24);
}

// ------------------------------------------------------------------------------------------------
// K1-AF (float32 AF as exact fixed point): besides the count, afsum[s] += the
// sum of AF[v] * 2^q as int64 over the set, uncovered bits.  Tile = 8192 variants = one KiB of every
// column: the float32 AF tile (32 KiB) and ~covered (1 KiB) sit in LDS; a wave keeps 4 samples' KiB
// in flight, skips samples whose KiB has no surviving bit (the common case once coverage has grown),
// otherwise walks the bits (ctz / clear-lowest / ds_read_b32 gather / mantissa << exponent / 64-bit add).
// A float32 a = m * 2^(e-150) (m = 24-bit mantissa with the hidden bit, e = biased exponent), so
// a * 2^q = af_fixed(table entry).
// ------------------------------------------------------------------------------------------------
#define UTM_AF_TILE_WORDS 128
#define UTM_DEFER_SLOTS 256  // most iterations whose newly-covered masks the log holds (af_defer.hip.h): a whole batch of the persistent loop;
#define UTM_DEFER_SLOTS_LAUNCHES 64  // ... and what a context gets whose matrix keeps the launches (utm_ctx::defer_slots)
// covered_out != nullptr: a *delta* pass as in k_score_afs below -- the tile's mask is what the pending winner newly
// covers, the shares are subtracted, group 0 writes covered | winner into the other buffer of the pair (and the mask
// into the log of af_defer.hip.h).  The form for the first iterations of a run, when a winner still newly covers
// percents of all variants: every surviving bit is a gather, and here the gathers hit LDS.
__global__ __launch_bounds__(256) void k_score_afq(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp,
                                                   const unsigned *__restrict__ af,
                                                   const Pending pend,
                                                   const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                   u64 *__restrict__ cnt, i64 *__restrict__ afsum, unsigned group_size,
                                                   unsigned n_groups, u64 *__restrict__ covered_out, u64 *__restrict__ newly_log)
{
    __shared__ unsigned aft[UTM_AF_TILE_WORDS * 64];  // fixed-point table entries
    __shared__ u64 live[UTM_AF_TILE_WORDS];
    if (st->done) return;
    unsigned tile, grp;
    if (!tile_of_block(wp, UTM_AF_TILE_WORDS, n_groups, tile, grp)) return;
    const u64 w0 = (u64)tile * UTM_AF_TILE_WORDS;
    const bool delta = covered_out != nullptr;
    const u64 *wcol = (delta || pend.fuse) ? pending_column(st, cols, wp, pend) : nullptr;
    if (threadIdx.x < UTM_AF_TILE_WORDS) {
        u64 c = covered[w0 + threadIdx.x];
        if (delta) {
            const u64 w = wcol ? wcol[w0 + threadIdx.x] : 0ull;
            if (grp == 0) {
                covered_out[w0 + threadIdx.x] = c | w;
                if (newly_log && wcol) newly_log[w0 + threadIdx.x] = w & ~c;
            }
            live[threadIdx.x] = w & ~c;
        } else {
            if (wcol) {
                c |= wcol[w0 + threadIdx.x];
                if (grp == 0) covered[w0 + threadIdx.x] = c;
            }
            live[threadIdx.x] = ~c;
        }
    }
    {
        const v4u *src = reinterpret_cast<const v4u *>(af + w0 * 64);
        v4u *dst = reinterpret_cast<v4u *>(aft);
#pragma unroll
        for (int i = 0; i < UTM_AF_TILE_WORDS * 16 / 256; ++i) dst[i * 256 + threadIdx.x] = src[i * 256 + threadIdx.x];
    }
    __syncthreads();

    const unsigned n_active = st->n_active;
    const unsigned lo = grp * group_size;
    const unsigned hi = lo + group_size < n_active ? lo + group_size : n_active;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 m0 = live[2 * lane], m1 = live[2 * lane + 1];
    const unsigned *a0 = aft + (2 * lane) * 64, *a1 = a0 + 64;
    constexpr int U = 4;
    // a batch = U samples' KiB of this tile per wave; the next batch is requested before this one is walked
    auto request = [&](unsigned i0, unsigned *s, v2q *x) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u < hi ? i0 + u : hi - 1;  // tail: re-read the last sample, ignored below
            s[u] = act[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = __builtin_nontemporal_load(reinterpret_cast<const v2q *>(cols + (u64)s[u] * wp + w0) + lane);
    };
    unsigned s_next[U];
    v2q x_next[U];
    if (lo + wave * U < hi) request(lo + wave * U, s_next, x_next);
    for (unsigned i0 = lo + wave * U; i0 < hi; i0 += 4 * U) {
        unsigned s[U];
        v2q x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s[u] = s_next[u];
            x[u] = x_next[u];
        }
        if (i0 + 4 * U < hi) request(i0 + 4 * U, s_next, x_next);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u64 b0 = x[u].x & m0, b1 = x[u].y & m1;
            const unsigned n_lane = __popcll(b0) + __popcll(b1);
            if (i0 + u >= hi || __ballot(n_lane != 0) == 0) continue;  // wave uniform
            u64 sum = 0;
            while (b0) {
                const unsigned bits = a0[__builtin_ctzll(b0)];
                b0 &= b0 - 1;
                sum += af_fixed(bits);
            }
            while (b1) {
                const unsigned bits = a1[__builtin_ctzll(b1)];
                b1 &= b1 - 1;
                sum += af_fixed(bits);
            }
            const unsigned n = wave_sum_u32(n_lane);
            const i64 total = wave_sum_u63(sum);  // per lane < 2^53 (the host's exactness precondition)
            if (lane == 0) {  // two's complement: adding the negated value subtracts
                atomicAdd(&cnt[s[u]], delta ? (u64)0 - (u64)n : (u64)n);
                atomicAdd(reinterpret_cast<u64 *>(&afsum[s[u]]), delta ? (u64)0 - (u64)total : (u64)total);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1-AF, full dense pass as table lookups (the first iteration of an --af run: nothing is covered yet, so every set bit
// of the matrix counts and walking them one by one is VALU work proportional to the density, a wave running as long
// as its fullest lane).  Here the work per column byte is fixed: the tile's variants are cut into groups of 4, every
// group gets a 16-entry table of the exact partial sums of its live members (a covered variant, or one with AF == 0,
// adds nothing), and a column nibble IS the index into its group's table -- one ds_read_b64 and one add per 4 variants,
// no bit loop, no divergence.
//   Tile = 4,096 variants = 512 B of every column = 32 lanes x 16 B: a wave instruction reads two samples' pieces,
// lane `sub` of a half wave owns the same 128 variants for every sample, i.e. the same 32 groups.  Table layout
// (128 KiB, one 1,024-thread workgroup per CU): entry (group j of lane sub, nibble b) sits at
//     j even:  65536 + (j/2) * 4096 + b * 256  + sub * 8        j odd:  (j/2) * 256 + b * 4096 + sub * 8
// so that ONE shifted copy of a column dword yields two addresses with an AND-OR each (the even nibble lands on bits
// 8..11, the odd one on bits 12..15, sub * 8 below them, the rest is the instruction's immediate offset), and the 32
// lanes a ds_read_b64 serves per LDS cycle (one sample's 32 subs) always fall on 32 different bank pairs, whatever
// their nibbles are: no bank conflicts (the first form, two samples per 32 lanes on one table copy, spent as many LDS
// cycles on conflicts as on reads).
//   An entry is kept as two 32-bit limbs (value = hi * 2^26 + lo, lo < 2^26): a lane adds its 32 entries with v_add3_u32
// and no carries; the host admits the kernel only when every table value is below 2^46 (af_table_ok), so that a
// lane's 128 variants stay below 2^53 and neither limb sum (nor, after one normalisation, their 32-lane DPP sums)
// can overflow.  Same integer sums as k_score_afq, same atomics.
// ------------------------------------------------------------------------------------------------
#define UTM_AFT_TILE_WORDS 64
#define UTM_AFT_THREADS 1024
#define UTM_AFT_LIMB 26
#define UTM_AFT_MAX_GROUP 1024
__device__ __forceinline__ u64 aft_limbs(u64 v)
{
    return (v & ((1ull << UTM_AFT_LIMB) - 1)) | ((v >> UTM_AFT_LIMB) << 32);
}
__device__ __forceinline__ unsigned half_sum32(unsigned v)  // DPP: lanes 31 and 63 end up with their half wave's sum
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1, 3
    return v;
}
struct AftSample {
    unsigned s;    // local sample
    unsigned kib;  // its column's offset inside the chunk, in KiB (columns are whole KiB: s * wp / 128)
};
__global__ __launch_bounds__(UTM_AFT_THREADS) void k_score_aft(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp,
                                                               const unsigned *__restrict__ af, const Pending pend,
                                                               const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                               u64 *__restrict__ cnt, i64 *__restrict__ afsum, unsigned group_size,
                                                               unsigned n_groups)
{
    __shared__ __attribute__((aligned(16))) u64 tab[16384];  // 128 KiB, layout above
    __shared__ __attribute__((aligned(16))) u64 live[UTM_AFT_TILE_WORDS];
    __shared__ AftSample sact[UTM_AFT_MAX_GROUP];  // the group's samples (a column request must not wait for a global act[] read)
    if (st->done) return;
    unsigned tile, grp;
    if (!tile_of_block(wp, UTM_AFT_TILE_WORDS, n_groups, tile, grp)) return;
    const u64 w0 = (u64)tile * UTM_AFT_TILE_WORDS;
    const unsigned n_active = st->n_active;
    const unsigned lo = grp * group_size;
    const unsigned hi = lo + group_size < n_active ? lo + group_size : n_active;
    {
        const unsigned kib_per_col = (unsigned)(wp / 128);
        for (unsigned i = threadIdx.x; lo + i < hi; i += UTM_AFT_THREADS) {
            const unsigned s = act[lo + i];
            sact[i] = AftSample{s, s * kib_per_col};
        }
    }
    if (threadIdx.x < UTM_AFT_TILE_WORDS) {
        const u64 *wcol = pend.fuse ? pending_column(st, cols, wp, pend) : nullptr;
        u64 c = covered[w0 + threadIdx.x];
        if (wcol) {
            c |= wcol[w0 + threadIdx.x];
            if (grp == 0) covered[w0 + threadIdx.x] = c;
        }
        live[threadIdx.x] = ~c;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned sub = lane & 31, half = lane >> 5;
    constexpr int U = 2, NW = UTM_AFT_THREADS / 64;
    // a unit = two samples' pieces of this tile (one wave instruction).  Two register sets of U units each: while one
    // is walked the other is in flight, and a set is re-requested only after its last use (a slot refilled while its
    // old contents are still live costs register copies behind a vmcnt(0) at the loop's end).  A request is
    // unconditional -- beyond the group's end it re-reads the tile's covered words (L2 hits, ignored).
    const unsigned n_units = (hi - lo + 1) / 2;
    const char *col_lane = reinterpret_cast<const char *>(cols + w0) + sub * 16;
    const char *dummy_lane = reinterpret_cast<const char *>(covered + w0) + sub * 16;
    auto request = [&](unsigned unit, unsigned &s, v4u &x) {
        const unsigned i = unit * 2 + half;  // position inside the group
        const bool ok = lo + i < hi;
        const AftSample a = sact[ok ? i : 0];
        s = a.s;
        const char *src = ok ? col_lane + ((u64)a.kib << 10) : dummy_lane;
        x = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src));
    };
    unsigned sa[U], sb[U];
    v4u xa[U], xb[U];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) request(wave + u * NW, sa[u], xa[u]);  // (on their way while the tables are built)
    {   // the tables: thread (j, sub) builds the 16 partial sums of variants sub * 128 + j * 4 .. + 3
        const unsigned bj = threadIdx.x >> 5, bsub = threadIdx.x & 31;
        const v4u e = reinterpret_cast<const v4u *>(af + w0 * 64)[bsub * 32 + bj];
        const unsigned lv = (unsigned)(live[bsub * 2 + (bj >> 4)] >> ((bj & 15) * 4)) & 15u;
        const u64 q0 = (lv & 1) ? af_fixed(e.x) : 0ull, q1 = (lv & 2) ? af_fixed(e.y) : 0ull;
        const u64 q2 = (lv & 4) ? af_fixed(e.z) : 0ull, q3 = (lv & 8) ? af_fixed(e.w) : 0ull;
        u64 t[16];
        t[0] = 0; t[1] = q0; t[2] = q1; t[3] = q0 + q1;
        t[4] = q2; t[5] = q2 + q0; t[6] = q2 + q1; t[7] = q2 + t[3];
#pragma unroll
        for (int b = 0; b < 8; ++b) t[8 + b] = q3 + t[b];
        char *base = reinterpret_cast<char *>(tab) + ((bj & 1) ? (bj >> 1) * 256 : 65536 + (bj >> 1) * 4096) + bsub * 8;
        const unsigned stride = (bj & 1) ? 4096 : 256;
#pragma unroll
        for (int b = 0; b < 16; ++b) *reinterpret_cast<u64 *>(base + b * stride) = aft_limbs(t[b]);
    }
    const v4u m = reinterpret_cast<const v4u *>(live)[sub];  // this lane's 128 live bits, for the counts
    __syncthreads();
    const char *tb = reinterpret_cast<const char *>(tab);
    unsigned sub_odd = sub * 8, sub_even = 65536u | (sub * 8);
    asm volatile("" : "+v"(sub_odd), "+v"(sub_even));  // (opaque: an OR the compiler can see through becomes v_and + v_add instead of one v_and_or)
    auto walk = [&](unsigned unit, unsigned s, const v4u &x) {
        if (unit >= n_units) return;  // wave uniform
        const bool valid = lo + unit * 2 + half < hi;
        const unsigned n_lane = valid ? __popc(x.x & m.x) + __popc(x.y & m.y) + __popc(x.z & m.z) + __popc(x.w & m.w) : 0u;
        if (__ballot(n_lane != 0) == 0) return;  // wave uniform
        unsigned alo = 0, ahi = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned y = k == 0 ? x[d] << 8 : x[d] >> (8 * k - 8);
                const int p = 4 * d + k;
                const u64 ev = *reinterpret_cast<const u64 *>(tb + ((y & 0xF00u) | sub_even) + p * 4096);
                const u64 od = *reinterpret_cast<const u64 *>(tb + ((y & 0xF000u) | sub_odd) + p * 256);
                alo += (unsigned)ev + (unsigned)od;
                ahi += (unsigned)(ev >> 32) + (unsigned)(od >> 32);
            }
        }
        ahi += alo >> UTM_AFT_LIMB;  // lane: lo < 2^26, hi < 2^27 + 32 -- the 32-lane sums below stay inside 32 bits
        alo &= (1u << UTM_AFT_LIMB) - 1;
        if (!valid) alo = ahi = 0;  // (a tail half read something else)
        const unsigned n = half_sum32(n_lane);
        const unsigned tlo = half_sum32(alo), thi = half_sum32(ahi);
        if (sub == 31 && n) {
            atomicAdd(&cnt[s], (u64)n);
            atomicAdd(reinterpret_cast<u64 *>(&afsum[s]), ((u64)thi << UTM_AFT_LIMB) + tlo);
        }
    };
    for (unsigned k = wave; k < n_units; k += 2 * U * NW) {
#pragma unroll
        for (int u = 0; u < U; ++u) request(k + (U + u) * NW, sb[u], xb[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) walk(k + u * NW, sa[u], xa[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) request(k + (2 * U + u) * NW, sa[u], xa[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) walk(k + (U + u) * NW, sb[u], xb[u]);
    }
}

// ------------------------------------------------------------------------------------------------
// K1-AF, sparse phase: once a good part of the variants is covered most loaded words are zero after
// the AND, so this kernel is k_score_int's streaming loop (LDS-staged ~covered tile, 8 KiB in flight
// per wave) plus a per-wave LDS queue: surviving bits only *enqueue* their variant index (prefix sum
// over the lanes, no memory wait); when the queue fills up, and at the end of the (sample, tile), all
// 64 lanes drain it together -- independent float32 gathers from the AF table in global memory (it
// stays in L2 / Infinity Cache), mantissa << exponent, 64-bit add -- then ONE reduction per (sample,
// tile).  The host switches from k_score_afq to this kernel when the captured fraction passes
// UTM_AF_SWITCH.  Same integer sums, same exactness argument.
// ------------------------------------------------------------------------------------------------
// queue entries per wave: STEPS KiB of ~covered + 4 queues must leave room for 4-5 workgroups per CU
// CAP = queue depth per LANE: every lane keeps its own little queue (slot-major in LDS, so a wave's
// pushes are conflict free) -- no cross-lane prefix sum is needed to place an entry.
// delta_mask == nullptr: full scoring against ~covered (adds to the accumulators, fuses the pending update).
// delta_mask != nullptr or covered_out != nullptr: *delta* scoring -- against the variants the last winner newly
// covered; their contribution is SUBTRACTED from the persistent accumulators.  Same bytes streamed, but only the few
// newly covered bits take the queue/gather path.  The mask is either made while the tile is staged (covered_out: the
// updated covered words go to the other buffer of a ping-pong pair) or was made beforehand by k_newly_mask (delta_mask:
// the form for winners read in place from another GPU, which one launch should read once, not once per workgroup).
template <int STEPS, int CAP>
__global__ __launch_bounds__(256) void k_score_afs(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp,
                                                   const unsigned *__restrict__ afbits, const Pending pend,
                                                   const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                   u64 *__restrict__ cnt, i64 *__restrict__ afsum, unsigned group_size,
                                                   unsigned n_groups, const u64 *__restrict__ delta_mask,
                                                   u64 *__restrict__ covered_out, u64 *__restrict__ newly_log)
{
    __shared__ v4u live[STEPS * 64];
    __shared__ unsigned queue[4][CAP][64];
    if (st->done) return;
    unsigned tile, grp;
    if (!tile_of_block(wp, STEPS * UTM_STEP_WORDS, n_groups, tile, grp)) return;
    const u64 w0 = (u64)tile * STEPS * UTM_STEP_WORDS;
    const u64 left = (wp - w0) / UTM_STEP_WORDS;
    const int nsteps = left < (u64)STEPS ? (int)left : STEPS;
    const bool delta = delta_mask || covered_out;
    if (covered_out) {
        // delta pass, mask made on the fly: newly covered = pending winner & ~covered.  Every group of the tile reads
        // the OLD covered words -- `covered` is not written by this launch -- and group 0 writes covered | winner
        // into the other buffer of the pair, which the host makes the current one for whatever it enqueues next
        // (a writer in place would hand later groups an already updated tile, i.e. an empty mask).
        const v4u *cv = reinterpret_cast<const v4u *>(covered + w0);
        v4u *co = reinterpret_cast<v4u *>(covered_out + w0);
        const u64 *wcol = pending_column(st, cols, wp, pend);
        const v4u *wc = wcol ? reinterpret_cast<const v4u *>(wcol + w0) : nullptr;
        const v4u zero4 = {0, 0, 0, 0};
        // newly_log: the mask is also what the last winner's exact float64 score runs over (af_defer.hip.h): group 0
        // leaves it in that row's slot of the log (the host passes this chunk's part of the slot)
        v4u *lg = (newly_log && wc && grp == 0) ? reinterpret_cast<v4u *>(newly_log + w0) : nullptr;
        for (int i = threadIdx.x; i < nsteps * 64; i += 256) {
            const v4u c = cv[i];
            const v4u w = wc ? wc[i] : zero4;
            if (grp == 0) co[i] = c | w;
            if (lg) lg[i] = w & ~c;
            live[i] = w & ~c;
        }
    } else if (delta_mask) {
        const v4u *mk = reinterpret_cast<const v4u *>(delta_mask + w0);
        for (int i = threadIdx.x; i < nsteps * 64; i += 256) live[i] = mk[i];
    } else {
        v4u *cv = reinterpret_cast<v4u *>(covered + w0);
        const u64 *wcol = pend.fuse ? pending_column(st, cols, wp, pend) : nullptr;
        const v4u *wc = wcol ? reinterpret_cast<const v4u *>(wcol + w0) : nullptr;
        for (int i = threadIdx.x; i < nsteps * 64; i += 256) {
            v4u c = cv[i];
            if (wc) {
                c |= wc[i];
                if (grp == 0) cv[i] = c;
            }
            live[i] = ~c;
        }
    }
    __syncthreads();

    const unsigned n_active = st->n_active;
    const unsigned lo = grp * group_size;
    const unsigned hi = lo + group_size < n_active ? lo + group_size : n_active;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int U = STEPS < 8 ? STEPS : 8;
    unsigned(*q)[64] = queue[wave];
    const unsigned *af_tile = afbits + w0 * 64;  // AF of the tile's first variant
    for (unsigned i = lo + wave; i < hi; i += 4) {
        const unsigned s = act[i];
        const v4u *p = reinterpret_cast<const v4u *>(cols + (u64)s * wp + w0) + lane;
        unsigned acc = 0, qc = 0;  // qc: entries in this lane's queue
        u64 sum = 0;
        auto drain = [&]() {
            for (unsigned j = 0; __ballot(j < qc) != 0; j += 2) {  // two independent gathers per round
                const unsigned f0 = j < qc ? af_tile[q[j][lane]] : 0u;
                const unsigned f1 = j + 1 < qc ? af_tile[q[j + 1][lane]] : 0u;
                if (j < qc) sum += af_fixed(f0);
                if (j + 1 < qc) sum += af_fixed(f1);
            }
            qc = 0;
        };
        for (int j0 = 0; j0 < nsteps; j0 += U) {
            v4u b[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                b[u] = j0 + u < nsteps ? __builtin_nontemporal_load(p + (j0 + u) * 64) : (v4u)(0);
            unsigned nb = 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j0 + u < nsteps) b[u] &= live[(j0 + u) * 64 + lane];
                nb += __popc(b[u].x) + __popc(b[u].y) + __popc(b[u].z) + __popc(b[u].w);
            }
            acc += nb;
            if (__ballot(nb != 0) == 0) continue;  // nothing survived in these 8 KiB
            if (__ballot(qc + nb > CAP) != 0) drain();
            const unsigned base = (unsigned)(j0 * UTM_STEP_WORDS + 2 * lane) * 64;  // variant offset inside the tile
            if (__ballot(nb > CAP) == 0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (__ballot((b[u].x | b[u].y | b[u].z | b[u].w) != 0) == 0) continue;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        unsigned bits = b[u][d];
                        const unsigned v0 = base + u * (UTM_STEP_WORDS * 64) + d * 32;
                        while (bits) {
                            q[qc++][lane] = v0 + __builtin_ctz(bits);
                            bits &= bits - 1;
                        }
                    }
                }
            } else {  // dense data: some lane has more bits in one batch than its queue holds -- gather directly
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        unsigned bits = b[u][d];
                        const unsigned v0 = base + u * (UTM_STEP_WORDS * 64) + d * 32;
                        while (bits) {
                            sum += af_fixed(af_tile[v0 + __builtin_ctz(bits)]);
                            bits &= bits - 1;
                        }
                    }
                }
            }
        }
        if (__ballot(qc != 0) != 0) drain();
        const unsigned n = wave_sum_u32(acc);
        if (n) {  // wave uniform
            const i64 total = wave_sum_u63(sum);
            if (lane == 0) {  // two's complement: adding the negated value subtracts
                atomicAdd(&cnt[s], delta ? (u64)0 - (u64)n : (u64)n);
                atomicAdd(reinterpret_cast<u64 *>(&afsum[s]), delta ? (u64)0 - (u64)total : (u64)total);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Sequential AF (float64 AF, or float32 AF that fails the fixed-point precondition): the reference
// adds row values into a float64 score in ascending variant order (`scores += row`, select.py:40);
// float64 addition does not reassociate, so each sample's chain is walked by ONE lane, chunk after
// chunk, word after word, bit after bit.  Latency bound by construction.
// ------------------------------------------------------------------------------------------------
template <typename AF_T>
__device__ __forceinline__ void seq_score_sample(const SeqChunk *__restrict__ chunks, int n_chunks, unsigned s, u64 *__restrict__ cnt,
                                                 double *__restrict__ fscore)
{
    double acc = 0.0;
    u64 n = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const SeqChunk ch = chunks[c];
        const ulonglong2 *col = reinterpret_cast<const ulonglong2 *>(ch.cols + (u64)s * ch.wp);
        const ulonglong2 *cov = reinterpret_cast<const ulonglong2 *>(ch.covered);
        const AF_T *af = static_cast<const AF_T *>(ch.af);
        for (u64 w2 = 0; w2 < (ch.w + 1) / 2; ++w2) {  // wp is even, padding words are zero
            const ulonglong2 x = col[w2];
            const ulonglong2 m = cov[w2];
            u64 b0 = x.x & ~m.x, b1 = x.y & ~m.y;
            n += __popcll(b0) + __popcll(b1);
            const AF_T *a = af + w2 * 128;
            while (b0) { acc += (double)a[__builtin_ctzll(b0)]; b0 &= b0 - 1; }
            a += 64;
            while (b1) { acc += (double)a[__builtin_ctzll(b1)]; b1 &= b1 - 1; }
        }
    }
    cnt[s] = n;
    fscore[s] = acc;
}

template <typename AF_T>
__global__ __launch_bounds__(64) void k_score_seq(const SeqChunk *__restrict__ chunks, int n_chunks,
                                                  const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                  u64 *__restrict__ cnt, double *__restrict__ fscore)
{
    if (st->done) return;
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    if (i >= st->n_active) return;
    seq_score_sample<AF_T>(chunks, n_chunks, act[i], cnt, fscore);
}
