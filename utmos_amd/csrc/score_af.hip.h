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
//   Tile = 4,096 variants = 512 B of every column = 32 lanes x 16 B: a wave instruction reads two samples' pieces (a
// "unit"), lane `sub` of a half wave owns the same 128 variants for every sample, i.e. the same 32 groups.  Table
// layout (128 KiB, one 1,024-thread workgroup per CU): entry (group j of lane sub, nibble b) sits at
//     j even:  65536 + (j/2) * 4096 + b * 256  + sub * 8        j odd:  (j/2) * 256 + b * 4096 + sub * 8
// so that ONE shifted copy of a column dword yields two addresses with an AND-OR each (the even nibble lands on bits
// 8..11, the odd one on bits 12..15, sub * 8 below them, the rest is the instruction's immediate offset), and the 32
// lanes a ds_read_b64 serves per LDS cycle (one sample's 32 subs) always fall on 32 different bank pairs, whatever
// their nibbles are: no bank conflicts.
//   An entry is kept as two 32-bit limbs, the limb-wise sums of its members' limbs (a value = hi * 2^25 + lo, lo < 2^25;
// an entry's lo limb is not carried into its hi limb, it stays below 2^27): a lane adds its 32 entries with v_add3_u32
// and no carries (32 * (2^27 - 4) < 2^32) and carries once per unit and tile.  The host admits the kernel only when
// every table value is below 2^45 (af_table_ok): a value's hi limb is then below 2^20, a lane's 128 variants add less
// than 2^27 to its hi sum per tile, and 16 tiles stay inside 32 bits.
//   A workgroup owns a group of samples and a run of consecutive tiles (as long as the grid allows: what a workgroup
// pays before its first lookup -- act[] -> column requests -> table, two to three memory round trips under load --
// and after its last measured 7 us, a tile 9): a wave keeps the SAME MAXU units for all of them, so a lane's partial
// sums (lo, hi, count per unit) stay in registers across the tiles and the cross-lane reduction and the two atomics
// per sample happen once per UTM_AFT_RUN tiles, not once per tile -- with them per tile the kernel spent a third of its
// VALU work outside the lookups.  Per tile: barrier, every thread
// builds one group's table from values it requested a tile earlier, barrier, the wave walks its units with D column
// requests in flight.  Same integer sums as k_score_afq.  The pending winner is applied to `covered` beforehand (host).
// ------------------------------------------------------------------------------------------------
#define UTM_AFT_TILE_WORDS 64
#define UTM_AFT_THREADS 1024
#define UTM_AFT_LIMB 25
#define UTM_AFT_RUN 16    // tiles between two flushes of the per-lane sums: a lane's hi limb grows by < 2^27 + 2^7 per tile
#define UTM_AFT_D 4       // column requests in flight per wave (1 KiB each); MAXU is a multiple
struct AftLimbs {
    unsigned lo, hi;
};
__device__ __forceinline__ AftLimbs aft_limbs(unsigned f, bool live)  // a table value (af_fixed) as limbs, 0 when its variant does not count
{
    const u64 v = live ? af_fixed(f) : 0ull;
    return AftLimbs{(unsigned)v & ((1u << UTM_AFT_LIMB) - 1), (unsigned)(v >> UTM_AFT_LIMB)};
}
__device__ __forceinline__ AftLimbs operator+(const AftLimbs &a, const AftLimbs &b)
{
    return AftLimbs{a.lo + b.lo, a.hi + b.hi};
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
// Grid: (runs of tiles) x (groups of samples), through tile_of_block with a run as its "tile".  group_size <= 32 * MAXU;
// MAXU = 4 or 8 (12 units' accumulators and addresses no longer fit 128 registers).
template <int MAXU>
__global__ __launch_bounds__(UTM_AFT_THREADS) void k_score_aft(const u64 *__restrict__ cols, const u64 *__restrict__ covered, u64 wp,
                                                               const unsigned *__restrict__ af, const IterState *__restrict__ st,
                                                               const unsigned *__restrict__ act, u64 *__restrict__ cnt,
                                                               i64 *__restrict__ afsum, unsigned group_size, unsigned n_groups,
                                                               unsigned run_tiles)
{
    static_assert(MAXU % UTM_AFT_D == 0, "the request ring wraps at a tile's end");
    __shared__ __attribute__((aligned(16))) u64 tab[16384];  // 128 KiB, layout above
    __shared__ unsigned s_of[UTM_AFT_THREADS / 64][MAXU][2];  // the samples behind a wave's units (for the flushes' atomics)
    if (st->done) return;
    unsigned run, grp;
    if (!tile_of_block(wp, UTM_AFT_TILE_WORDS * run_tiles, n_groups, run, grp)) return;
    const u64 w_first = (u64)run * run_tiles * UTM_AFT_TILE_WORDS;
    const unsigned n_t = (unsigned)((wp - w_first) / UTM_AFT_TILE_WORDS < run_tiles ? (wp - w_first) / UTM_AFT_TILE_WORDS : run_tiles);
    const unsigned n_active = st->n_active;
    const unsigned lo = grp * group_size;
    const unsigned hi = lo + group_size < n_active ? lo + group_size : n_active;
    if (lo >= hi) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned sub = lane & 31, half = lane >> 5;
    constexpr int NW = UTM_AFT_THREADS / 64, D = UTM_AFT_D;
    // this wave's units: unit u = positions lo + 2 * (wave + NW * u) + {0, 1} of act[]; a lane without a sample of
    // its own (the group's ragged end) reads the group's last sample instead and is cleared at the end
    const unsigned n_units = (hi - lo + 1) / 2;
    const unsigned n_my = (unsigned)wave < n_units ? (n_units - wave + NW - 1) / NW : 0;  // wave uniform
    // what the next tile's build needs, requested a tile ahead (the first tile's before the column
    // requests, which must wait for act[]: the first build then runs while those are in flight): this thread's group (j, sub) = 4 table values and the
    // covered word they fall into; and the lane's own 128 covered bits for the counts
    const unsigned bj = threadIdx.x >> 5, bsub = threadIdx.x & 31;
    const unsigned *af_thread = af + w_first * 64 + (bsub * 32 + bj) * 4;
    const u64 *cov_thread = covered + w_first + bsub * 2 + (bj >> 4);
    const u64 *cov_lane = covered + w_first + sub * 2;
    v4u e_next = *reinterpret_cast<const v4u *>(af_thread);
    u64 c_next = *cov_thread;
    v2q m_next = *reinterpret_cast<const v2q *>(cov_lane);
    const char *pa[MAXU];  // where this lane's 16 B of unit u start in the current tile
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        const unsigned i = lo + 2 * (wave + NW * u) + half;
        const unsigned s = act[i < hi ? i : hi - 1];
        if (sub == 31) s_of[wave][u][half] = s;  // (read back by the same lane)
        pa[u] = reinterpret_cast<const char *>(cols + (u64)s * wp + w_first) + sub * 16;
    }
    v4u ring[D];
#pragma unroll
    for (int u = 0; u < D; ++u) ring[u] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(pa[u]));
    unsigned alo[MAXU], ahi[MAXU], acn[MAXU];
#pragma unroll
    for (int u = 0; u < MAXU; ++u) alo[u] = ahi[u] = acn[u] = 0;
    char *tb_w = reinterpret_cast<char *>(tab) + ((bj & 1) ? (bj >> 1) * 256 : 65536 + (bj >> 1) * 4096) + bsub * 8;
    const unsigned w_stride = (bj & 1) ? 4096 : 256;
    const char *tb = reinterpret_cast<const char *>(tab);
    unsigned sub_odd = sub * 8, sub_even = 65536u | (sub * 8);
    asm volatile("" : "+v"(sub_odd), "+v"(sub_even));  // (opaque: an OR the compiler can see through becomes v_and + v_add instead of one v_and_or)
    for (unsigned t = 0; t < n_t; ++t) {
        const v4u e = e_next;
        const unsigned lv = (unsigned)(~c_next >> ((bj & 15) * 4)) & 15u;
        const v2q mq = ~m_next;
        const unsigned m0 = (unsigned)mq.x, m1 = (unsigned)(mq.x >> 32), m2 = (unsigned)mq.y, m3 = (unsigned)(mq.y >> 32);
        const u64 step = t + 1 < n_t ? 512 : 0;  // (the last tile re-reads itself: harmless, keeps every request unconditional)
        __syncthreads();  // the previous tile's lookups are done
        {
            const AftLimbs q0 = aft_limbs(e.x, lv & 1), q1 = aft_limbs(e.y, lv & 2), q2 = aft_limbs(e.z, lv & 4), q3 = aft_limbs(e.w, lv & 8);
            AftLimbs tv[16];
            tv[0] = AftLimbs{0, 0}; tv[1] = q0; tv[2] = q1; tv[3] = q0 + q1;
            tv[4] = q2; tv[5] = q2 + q0; tv[6] = q2 + q1; tv[7] = q2 + tv[3];
#pragma unroll
            for (int b = 0; b < 8; ++b) tv[8 + b] = q3 + tv[b];
#pragma unroll
            for (int b = 1; b < 16; ++b) *reinterpret_cast<u64 *>(tb_w + b * w_stride) = (u64)tv[b].lo | ((u64)tv[b].hi << 32);
            if (t == 0) *reinterpret_cast<u64 *>(tb_w) = 0ull;  // (entry 0 is zero in every tile)
        }
        af_thread += step / 8 * 64;  // (64 words of covered = 4,096 table values further on)
        cov_thread += step / 8;
        cov_lane += step / 8;
        e_next = *reinterpret_cast<const v4u *>(af_thread);
        c_next = *cov_thread;
        m_next = *reinterpret_cast<const v2q *>(cov_lane);
        __syncthreads();  // the tables stand
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            const v4u x = ring[u % D];
            if ((unsigned)u < n_my) {  // wave uniform
                const unsigned n_lane = __popc(x.x & m0) + __popc(x.y & m1) + __popc(x.z & m2) + __popc(x.w & m3);
                if (__ballot(n_lane != 0) != 0) {  // wave uniform
                    unsigned l = 0, h = ahi[u];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned y = k == 0 ? x[d] << 8 : x[d] >> (8 * k - 8);
                            const int p = 4 * d + k;
                            const u64 ev = *reinterpret_cast<const u64 *>(tb + ((y & 0xF00u) | sub_even) + p * 4096);
                            const u64 od = *reinterpret_cast<const u64 *>(tb + ((y & 0xF000u) | sub_odd) + p * 256);
                            l += (unsigned)ev + (unsigned)od;
                            h += (unsigned)(ev >> 32) + (unsigned)(od >> 32);
                        }
                    }
                    ahi[u] = h + (l >> UTM_AFT_LIMB);
                    alo[u] += l & ((1u << UTM_AFT_LIMB) - 1);  // (below 2^29 after 16 tiles)
                    acn[u] += n_lane;
                }
            }
            pa[u] += step;
            const int v = (u + D) % MAXU;  // the slot's next occupant: unit u + D of this tile, or unit u + D - MAXU of the next (pa[] already advanced)
            ring[u % D] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(pa[v]));
        }
        if ((t + 1) % UTM_AFT_RUN != 0 && t + 1 != n_t) continue;
        // every UTM_AFT_RUN tiles, and at the end: half-wave sums and the atomics.  hi < 2^31 + 2^12 per lane: its
        // 32-lane sum goes in two pieces.
#pragma unroll
        for (int u = 0; u < MAXU; ++u) {
            if ((unsigned)u >= n_my) break;  // wave uniform
            const bool mine = lo + 2 * (wave + NW * u) + half < hi;
            const unsigned l = mine ? alo[u] & ((1u << UTM_AFT_LIMB) - 1) : 0u, h = mine ? ahi[u] + (alo[u] >> UTM_AFT_LIMB) : 0u, c = mine ? acn[u] : 0u;
            alo[u] = ahi[u] = acn[u] = 0;
            const unsigned n = half_sum32(c);
            const unsigned tlo = half_sum32(l), th0 = half_sum32(h & 0xFFFFu), th1 = half_sum32(h >> 16);
            if (sub == 31 && n) {
                const unsigned s = s_of[wave][u][half];
                atomicAdd(&cnt[s], (u64)n);
                atomicAdd(reinterpret_cast<u64 *>(&afsum[s]), ((((u64)th1 << 16) + th0) << UTM_AFT_LIMB) + tlo);
            }
        }
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
