// Device code of libutmos_hip.so -- hand-written for gfx950 (CDNA4, wave64).  No portability layer.
//
// Data layout: inside a chunk every local sample is one column of `wp` uint64 words
// (wp = ceil(n_var/64) rounded up to 128 words = 1 KiB, zero padded); cols[s * wp + w].  One wave
// instruction reads 64 lanes x 16 B = 1 KiB of ONE column, so every HBM access of the scoring
// kernels is a full, aligned, contiguous KiB.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "synth_hash.h"

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));  // one global_load_dwordx4 / ds_read_b128
typedef unsigned long long v2q __attribute__((ext_vector_type(2)));

#define UTM_HDR_WORDS 8  // one utm_record (64 B) in front of every exchanged column
#define UTM_STEP_WORDS 128  // words one wave instruction covers (64 lanes x 2)

struct IterState {
    int done;        // loop finished: every later launch returns at once
    int prev_valid;  // a winner column still has to be OR-ed into `covered`
    int prev_local;  // its local column index, or -1: take it from exchange slot prev_rank
    int prev_rank;
    i64 prev_gidx;   // its global sample index
    unsigned n_active;  // selectable local samples = length of act[]
    unsigned best_pos;  // position in act[] of this shard's best of the current iteration
    i64 iter;           // rows produced so far
    i64 tot;            // tot_captured
    i64 n_active_total; // selectable samples over all shards
    // verified-parallel AF scoring (k_cand / k_chain)
    int n_cand;         // candidates whose score interval reaches the best lower bound
    int need_chain;     // some candidate's parallel sum is not provably the reference's float64 sum
    int cand_overflow;  // more candidates than UTM_MAX_CAND: every sample is re-scored sequentially
    int all_exact;      // every selectable sample's estimate is exact; scores only shrink, so it stays that way
    // decremental scoring: work actually done (for the byte accounting)
    u64 xseq;           // mailbox exchanges completed (identical on every shard)
    int xerror;         // a peer's record did not arrive in time
    int pad_;
    u64 decr_entries;   // sum over decremental iterations of the newly-covered word count
    u64 decr_gathers;   // ... of (selectable samples x newly-covered words)
};

#define UTM_MAX_CAND 64
struct CandBuf {
    unsigned pos[UTM_MAX_CAND];   // position in act[]
    unsigned samp[UTM_MAX_CAND];  // local sample
    i64 cnt[UTM_MAX_CAND];
    double val[UTM_MAX_CAND];     // unweighted score: exact estimate (k_cand) or sequential float64 sum (k_chain)
};

struct Rec {  // == utm_record
    double score;
    i64 idx;
    i64 new_count;
    i64 pad[5];
};

// Record mailbox for the device-side exchange between shards: every shard owns 2 x n_ranks slots in uncached
// device memory that all peers map (hipIpc); slot [seq & 1][r] receives rank r's record of exchange `seq`.
struct Mailbox {
    double score;
    i64 idx;
    i64 new_count;
    u64 seq;   // written last (release): the slot is complete when it equals the expected sequence number
    u64 pad[4];
};

struct SeqChunk {
    const u64 *cols;
    const u64 *covered;
    const void *af;
    u64 wp;
    u64 w;  // words holding variants
};

struct Cand {
    double val;
    i64 gidx;
    i64 cnt;
    unsigned pos;
};

struct PickArgs {
    IterState *st;
    unsigned *act;
    unsigned char *state;
    const double *weights;  // n_samp_total, or nullptr
    u64 *cnt;        // per-sample counts to read (accumulators, or the persistent copy in decremental mode)
    i64 *afsum;      // fixed-point AF sums, or nullptr
    u64 *cnt_mirror; // full mode: copy every count here (the persistent copy decremental iterations update), else nullptr
    i64 *afsum_mirror;
    int zero_after;  // full mode: clear the accumulators for the next iteration's atomics
    unsigned *list_n;  // decremental mode: per-chunk newly-covered word counts (read for the accounting, then cleared)
    int n_chunks;
    double *fscore;  // sequential AF scores, or nullptr
    double af_scale; // 2^-q
    Mailbox *mbox;             // this shard's mailbox slots [2][n_ranks], or nullptr
    Mailbox *const *peer_mbox; // every shard's mailbox base, as mapped here (index = rank)
    CandBuf *cand;   // verified-parallel AF: candidate list, else nullptr
    int af_is_f64;   // the estimate sums float32-rounded values of float64 AFs
    Rec *recs;       // exchange slot headers: recs[r] at xbuf + r*slot_words
    u64 slot_words;
    i64 *res_idx;
    i64 *res_new;
    double *res_score;
    i64 n_var_total;
    unsigned first, n_local, n_total;
    int rank, n_ranks;
};

__device__ __forceinline__ Rec *rec_of(const PickArgs &a, int r)
{
    return reinterpret_cast<Rec *>(reinterpret_cast<u64 *>(a.recs) + (u64)r * a.slot_words);
}

// Wave64 sum with DPP row shifts + row broadcasts (gfx9 family: row_bcast:15/31 exist); the total ends
// up in lane 63 and is returned wave-uniformly.  6 VALU ops, no LDS crossbar traffic.
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1,3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// Inclusive prefix sum over the 64 lanes (same DPP ladder, every lane keeps its partial).
__device__ __forceinline__ unsigned wave_scan_incl_u32(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);
    return v;
}
// 64-bit sums as three 32-bit reductions over 21-bit pieces (wave sums of a piece stay below 2^27);
// per-lane values must be below 2^63.
__device__ __forceinline__ i64 wave_sum_u63(u64 v)
{
    const unsigned p0 = wave_sum_u32((unsigned)(v & 0x1FFFFFu));
    const unsigned p1 = wave_sum_u32((unsigned)((v >> 21) & 0x1FFFFFu));
    const unsigned p2 = wave_sum_u32((unsigned)(v >> 42));
    return (i64)(((u64)p2 << 42) + ((u64)p1 << 21) + p0);
}

// Where the winner column of the previous iteration can be read from on this shard.
struct Pending {
    const u64 *xbuf;               // exchange slots {record, whole column} (column all-gather form)
    u64 slot_words;
    u64 chunk_off;
    const u64 *const *peer_cols;   // P2P form: this chunk's column base on every rank (IPC-mapped), or nullptr
    const unsigned *peer_first;    // first global sample of every rank
    int fuse;                      // scoring kernels: OR it into the covered tile while staging it
};

// Winner column of the previous iteration (base of the chunk's column), or nullptr.
__device__ __forceinline__ const u64 *pending_column(const IterState *st, const u64 *cols, u64 wp, const Pending &p)
{
    if (!st->prev_valid) return nullptr;
    if (st->prev_local >= 0) return cols + (u64)st->prev_local * wp;
    if (p.peer_cols) return p.peer_cols[st->prev_rank] + (u64)(st->prev_gidx - (i64)p.peer_first[st->prev_rank]) * wp;
    return p.xbuf + (u64)st->prev_rank * p.slot_words + UTM_HDR_WORDS + p.chunk_off;
}

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
template <int STEPS, bool NT>
__global__ __launch_bounds__(256) void k_score_int(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp,
                                                   const Pending pend,
                                                   const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                   u64 *__restrict__ cnt, unsigned group_size, unsigned n_groups)
{
    __shared__ v4u live[STEPS * 64];  // ~covered of this tile, STEPS KiB
    if (st->done) return;
    unsigned tile, grp;
    if (!tile_of_block(wp, STEPS * UTM_STEP_WORDS, n_groups, tile, grp)) return;
    const u64 w0 = (u64)tile * STEPS * UTM_STEP_WORDS;
    const u64 left = (wp - w0) / UTM_STEP_WORDS;
    const int nsteps = left < (u64)STEPS ? (int)left : STEPS;

    v4u *cv = reinterpret_cast<v4u *>(covered + w0);
    const u64 *wcol = pend.fuse ? pending_column(st, cols, wp, pend) : nullptr;
    const v4u *wc = wcol ? reinterpret_cast<const v4u *>(wcol + w0) : nullptr;
    for (int i = threadIdx.x; i < nsteps * 64; i += 256) {
        v4u c = cv[i];
        if (wc) {
            c |= wc[i];
            // every group of this tile computes the same words; group 0 stores them.  A racing reader
            // sees old or new words and ORs the winner in itself, so either is right.
            if (grp == 0) cv[i] = c;
        }
        live[i] = ~c;
    }
    __syncthreads();

    const unsigned n_active = st->n_active;
    const unsigned lo = grp * group_size;
    const unsigned hi = lo + group_size < n_active ? lo + group_size : n_active;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int U = STEPS < 8 ? STEPS : 8;  // loads in flight per wave: U KiB
    for (unsigned i = lo + wave; i < hi; i += 4) {
        const unsigned s = act[i];
        const v4u *p = reinterpret_cast<const v4u *>(cols + (u64)s * wp + w0) + lane;
        unsigned acc = 0;
        if (nsteps == STEPS) {
#pragma unroll 1
            for (int j0 = 0; j0 < STEPS; j0 += U) {
                v4u x[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    x[u] = NT ? __builtin_nontemporal_load(p + (j0 + u) * 64) : p[(j0 + u) * 64];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const v4u b = x[u] & live[(j0 + u) * 64 + lane];
                    acc += __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
                }
            }
        } else {
            for (int j = 0; j < nsteps; ++j) {
                const v4u b = p[j * 64] & live[j * 64 + lane];
                acc += __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
            }
        }
        acc = wave_sum_u32(acc);
        if (lane == 0 && acc) atomicAdd(&cnt[s], (u64)acc);
    }
}

// ------------------------------------------------------------------------------------------------
// K1-AF (float32 AF as exact fixed point): besides the count, afsum[s] += the
// sum of AF[v] * 2^q as int64 over the set, uncovered bits.  Tile = 8192 variants = one KiB of every
// column: the float32 AF tile (32 KiB) and ~covered (1 KiB) sit in LDS; a wave keeps 4 samples' KiB
// in flight, skips samples whose KiB has no surviving bit (the common case once coverage has grown),
// otherwise walks the bits (ctz / clear-lowest / ds_read_b32 gather / mantissa << exponent / 64-bit add).
// A float32 a = m * 2^(e-150) (m = 24-bit mantissa with the hidden bit, e = biased exponent), so
// a * 2^q = m << (e - e_base), e_base = 150 - q >= the smallest exponent present (host checks).
// ------------------------------------------------------------------------------------------------
#define UTM_AF_TILE_WORDS 128
__global__ __launch_bounds__(256) void k_score_afq(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp,
                                                   const float *__restrict__ af, int e_base,
                                                   const Pending pend,
                                                   const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                   u64 *__restrict__ cnt, i64 *__restrict__ afsum, unsigned group_size,
                                                   unsigned n_groups)
{
    __shared__ unsigned aft[UTM_AF_TILE_WORDS * 64];  // float32 bit patterns
    __shared__ u64 live[UTM_AF_TILE_WORDS];
    if (st->done) return;
    unsigned tile, grp;
    if (!tile_of_block(wp, UTM_AF_TILE_WORDS, n_groups, tile, grp)) return;
    const u64 w0 = (u64)tile * UTM_AF_TILE_WORDS;
    const u64 *wcol = pend.fuse ? pending_column(st, cols, wp, pend) : nullptr;
    if (threadIdx.x < UTM_AF_TILE_WORDS) {
        u64 c = covered[w0 + threadIdx.x];
        if (wcol) {
            c |= wcol[w0 + threadIdx.x];
            if (grp == 0) covered[w0 + threadIdx.x] = c;
        }
        live[threadIdx.x] = ~c;
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
    for (unsigned i0 = lo + wave * U; i0 < hi; i0 += 4 * U) {
        unsigned s[U];
        v2q x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u < hi ? i0 + u : hi - 1;  // tail: re-read the last sample, ignored below
            s[u] = act[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = __builtin_nontemporal_load(reinterpret_cast<const v2q *>(cols + (u64)s[u] * wp + w0) + lane);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            u64 b0 = x[u].x & m0, b1 = x[u].y & m1;
            const unsigned n_lane = __popcll(b0) + __popcll(b1);
            if (i0 + u >= hi || __ballot(n_lane != 0) == 0) continue;  // wave uniform
            u64 sum = 0;
            while (b0) {
                const unsigned bits = a0[__builtin_ctzll(b0)];
                b0 &= b0 - 1;
                sum += (u64)((bits & 0x7FFFFFu) | 0x800000u) << ((bits >> 23) - e_base);
            }
            while (b1) {
                const unsigned bits = a1[__builtin_ctzll(b1)];
                b1 &= b1 - 1;
                sum += (u64)((bits & 0x7FFFFFu) | 0x800000u) << ((bits >> 23) - e_base);
            }
            const unsigned n = wave_sum_u32(n_lane);
            const i64 total = wave_sum_u63(sum);  // per lane < 2^53 (the host's exactness precondition)
            if (lane == 0) {
                atomicAdd(&cnt[s[u]], (u64)n);
                atomicAdd(reinterpret_cast<u64 *>(&afsum[s[u]]), (u64)total);
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
__device__ __forceinline__ u64 af_fixed(unsigned f, int e_base)
{
    return (u64)((f & 0x7FFFFFu) | 0x800000u) << ((f >> 23) - e_base);
}

// CAP = queue depth per LANE: every lane keeps its own little queue (slot-major in LDS, so a wave's
// pushes are conflict free) -- no cross-lane prefix sum is needed to place an entry.
// delta_mask == nullptr: full scoring against ~covered (adds to the accumulators, fuses the pending update).
// delta_mask != nullptr: *delta* scoring -- the mask holds the variants the last winner newly covered
// (k_newly_mask made it and already updated covered); their contribution is SUBTRACTED from the persistent
// accumulators.  Same bytes streamed, but only the few newly covered bits take the queue/gather path.
template <int STEPS, int CAP>
__global__ __launch_bounds__(256) void k_score_afs(const u64 *__restrict__ cols, u64 *__restrict__ covered, u64 wp,
                                                   const unsigned *__restrict__ afbits, int e_base, const Pending pend,
                                                   const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                   u64 *__restrict__ cnt, i64 *__restrict__ afsum, unsigned group_size,
                                                   unsigned n_groups, const u64 *__restrict__ delta_mask)
{
    __shared__ v4u live[STEPS * 64];
    __shared__ unsigned queue[4][CAP][64];
    if (st->done) return;
    unsigned tile, grp;
    if (!tile_of_block(wp, STEPS * UTM_STEP_WORDS, n_groups, tile, grp)) return;
    const u64 w0 = (u64)tile * STEPS * UTM_STEP_WORDS;
    const u64 left = (wp - w0) / UTM_STEP_WORDS;
    const int nsteps = left < (u64)STEPS ? (int)left : STEPS;
    if (delta_mask) {
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
                if (j < qc) sum += af_fixed(f0, e_base);
                if (j + 1 < qc) sum += af_fixed(f1, e_base);
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
                            sum += af_fixed(af_tile[v0 + __builtin_ctz(bits)], e_base);
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
                atomicAdd(&cnt[s], delta_mask ? (u64)0 - (u64)n : (u64)n);
                atomicAdd(reinterpret_cast<u64 *>(&afsum[s]), delta_mask ? (u64)0 - (u64)total : (u64)total);
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
__global__ __launch_bounds__(64) void k_score_seq(const SeqChunk *__restrict__ chunks, int n_chunks,
                                                  const IterState *__restrict__ st, const unsigned *__restrict__ act,
                                                  u64 *__restrict__ cnt, double *__restrict__ fscore, int only_on_overflow)
{
    if (st->done) return;
    if (only_on_overflow && !(st->need_chain && st->cand_overflow)) return;
    const unsigned i = blockIdx.x * 64 + threadIdx.x;
    if (i >= st->n_active) return;
    const unsigned s = act[i];
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

// covered |= pending winner column (used where the update is not fused into a scoring kernel)
__global__ __launch_bounds__(256) void k_apply_pending(u64 *__restrict__ covered, const u64 *__restrict__ cols, u64 wp,
                                                       const Pending pend,
                                                       const IterState *__restrict__ st)
{
    const u64 *wcol = pending_column(st, cols, wp, pend);
    if (!wcol) return;
    // the column may live in another process / on another GPU (hipIpc mapping): system-scope loads, so that
    // no cache of this GPU can answer with an older copy of those addresses
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256)
        covered[w] |= __hip_atomic_load(&wcol[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// covered |= cols[col]  (utm_reset: samples that start out "used")
__global__ __launch_bounds__(256) void k_or_column(u64 *__restrict__ covered, const u64 *__restrict__ col, u64 wp)
{
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256) covered[w] |= col[w];
}

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

// ------------------------------------------------------------------------------------------------
// Verified-parallel AF scoring.  The reference's AF score of a sample is a float64 running sum in
// ascending variant order (select.py:40); float64 addition does not reassociate, so a parallel sum is
// only an *estimate* E with a rigorous bound B on |reference - E|:
//   float32 AF: E = exact integer sum of AF*2^q.  While E < 2^53 every partial sum of the reference
//               is exact, hence reference == E (B = 0).  Beyond: B = n * 2^-53 * E (n addends).
//   float64 AF: E sums the float32-rounded values exactly: B = (2^-24 + n * 2^-53) * E.
// k_cand keeps the samples whose weighted interval reaches the best lower bound -- only they can be
// the argmax -- and k_chain recomputes exactly those few with the reference's sequential chain.
// Result: bit-identical winner and score, with the bulk of the work order independent.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void af_interval(const PickArgs &a, unsigned s, u64 c, double &lo, double &hi, double &est,
                                            bool &exact)
{
    const i64 e = a.afsum[s];
    est = (double)e * a.af_scale;
    double bound;
    if (a.af_is_f64) {
        exact = c == 0;
        bound = exact ? 0.0 : 1.02 * (5.9604644775390625e-08 + (double)c * 1.1102230246251565e-16) * est;
    } else {
        exact = e < (1ll << 53);
        bound = exact ? 0.0 : 1.05 * ((double)c * 1.1102230246251565e-16 * est + 1.2e-16 * est);
    }
    double l = est - bound, h = est + bound;
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

__global__ __launch_bounds__(256) void k_cand(PickArgs a)
{
    __shared__ double wmax[4];
    __shared__ unsigned n_c;
    __shared__ int inexact, any_inexact;
    IterState *st = a.st;
    if (st->done) return;
    const unsigned n_active = st->n_active;
    if (threadIdx.x == 0) { n_c = 0; inexact = 0; any_inexact = 0; }
    double best_lo = -__builtin_inf();
    for (unsigned i = threadIdx.x; i < n_active; i += 256) {
        const unsigned s = a.act[i];
        double lo, hi, est;
        bool exact;
        af_interval(a, s, a.cnt[s], lo, hi, est, exact);
        best_lo = lo > best_lo ? lo : best_lo;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(best_lo, o, 64);
        best_lo = other > best_lo ? other : best_lo;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = best_lo;
    __syncthreads();
    best_lo = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    for (unsigned i = threadIdx.x; i < n_active; i += 256) {
        const unsigned s = a.act[i];
        const u64 c = a.cnt[s];
        double lo, hi, est;
        bool exact;
        af_interval(a, s, c, lo, hi, est, exact);
        if (!exact) any_inexact = 1;
        if (hi >= best_lo) {
            const unsigned slot = atomicAdd(&n_c, 1u);
            if (slot < UTM_MAX_CAND) {
                a.cand->pos[slot] = i;
                a.cand->samp[slot] = s;
                a.cand->cnt[slot] = (i64)c;
                a.cand->val[slot] = est;
            }
            if (!exact) inexact = 1;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st->n_cand = n_c < UTM_MAX_CAND ? (int)n_c : UTM_MAX_CAND;
        st->cand_overflow = n_c > UTM_MAX_CAND;
        st->need_chain = inexact;
        st->all_exact = !any_inexact;
    }
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
struct ChainSeg {
    int chunk;
    u64 w0;
};
struct ChainFast {
    const ChainSeg *segs;
    int n_segs;
    unsigned *counts;  // [UTM_FAST_CAND][n_segs]; 0xFFFFFFFF = segment too dense
    double *vals;      // [UTM_FAST_CAND][n_segs][UTM_SEG_CAP]
};

template <typename AF_T>
__global__ __launch_bounds__(1024) void k_chain_fill(const SeqChunk *__restrict__ chunks, const IterState *__restrict__ st,
                                                     const CandBuf *__restrict__ cand, ChainFast f)
{
    __shared__ unsigned wtot[16];
    if (st->done || !st->need_chain || st->cand_overflow || st->n_cand > UTM_FAST_CAND || (int)blockIdx.y >= st->n_cand) return;
    const ChainSeg sg = f.segs[blockIdx.x];
    const SeqChunk ch = chunks[sg.chunk];
    const unsigned s = cand->samp[blockIdx.y];
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
    const size_t slot = (size_t)blockIdx.y * f.n_segs + blockIdx.x;
    if (tid == 0) f.counts[slot] = total <= UTM_SEG_CAP ? total : 0xFFFFFFFFu;
    if (total == 0 || total > UTM_SEG_CAP) return;
    double *out = f.vals + slot * UTM_SEG_CAP + (woff + incl - n);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u64 y = x[k];
        while (y) {
            const int b = __builtin_ctzll(y);
            y &= y - 1;
            *out++ = (double)af[(w + k) * 64 + b];
        }
    }
}

#define UTM_CHAIN_CAP 2048
#define UTM_CHAIN_WPT 4  // consecutive words per lane and round: 4096 words (262,144 variants) per round
template <typename AF_T>
__global__ __launch_bounds__(1024) void k_chain(const SeqChunk *__restrict__ chunks, int n_chunks,
                                                const IterState *__restrict__ st, CandBuf *__restrict__ cand, ChainFast f)
{
    __shared__ double buf[UTM_CHAIN_CAP];
    __shared__ unsigned wtot[2][16];  // double buffered: one barrier per empty round
    __shared__ int dense;
    if (st->done || !st->need_chain || st->cand_overflow || (int)blockIdx.x >= st->n_cand) return;
    const unsigned s = cand->samp[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (f.counts && st->n_cand <= UTM_FAST_CAND) {
        // fast path: k_chain_fill compacted this candidate's values per segment; are all segments usable?
        const unsigned *cnts = f.counts + (size_t)blockIdx.x * f.n_segs;
        if (tid == 0) dense = 0;
        __syncthreads();
        for (int g = tid; g < f.n_segs; g += 1024)
            if (cnts[g] == 0xFFFFFFFFu) dense = 1;
        __syncthreads();
        if (!dense) {
            if (wave == 0) {  // one wave: coalesced loads of 64 values, then the ordered sum in registers
                double acc = 0.0;
                const double *vals = f.vals + (size_t)blockIdx.x * f.n_segs * UTM_SEG_CAP;
                for (int g = 0; g < f.n_segs; ++g) {
                    const unsigned m = cnts[g];
                    const double *v = vals + (size_t)g * UTM_SEG_CAP;
                    for (unsigned t = 0; t < m; t += 64) acc = ordered_sum64(acc, t + lane < m ? v[t + lane] : 0.0);
                }
                if (lane == 0) cand->val[blockIdx.x] = acc;
            }
            return;
        }
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
    if (tid == 0) cand->val[blockIdx.x] = acc;
}

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
        st->prev_local = (int)loc;
    } else {
        st->prev_local = -1;
    }
    if (st->tot >= a.n_var_total) st->done = 1;  // "Ran out of new variants" (select.py:110-112)
}

// Device-side exchange, receiving end: wait (bounded) until every shard's record of this exchange has
// landed in the local mailbox, copy them into the record slots, decide.  One lane per source shard.
#define UTM_MBOX_SPINS (1u << 24)  // x s_sleep(16): several seconds before a missing shard is declared lost
__device__ __forceinline__ bool mbox_wait(const Mailbox *slot, u64 expected, Rec *out)
{
    for (unsigned spin = 0; spin < UTM_MBOX_SPINS; ++spin) {
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

// MODE 0: single shard -- pick and decide.  1: write this shard's record into its exchange slot.
// 2: as 1, and post the record into every shard's mailbox (device-side exchange over P2P mappings).
template <int MODE>
__global__ __launch_bounds__(1024) void k_pick(PickArgs a)
{
    __shared__ Cand wbest[16];
    __shared__ Rec srec;
    __shared__ int late;
    IterState *st = a.st;
    if (st->done) return;
    const unsigned n_active = st->n_active;
    // where this iteration's scores come from
    //   0 integer counts | 1 exact fixed-point AF sums | 2 sequential float64 scores of every sample
    //   3 the candidates' sequential float64 scores (k_chain)
    int src = a.afsum ? 1 : a.fscore ? 2 : 0;
    if (a.cand && st->need_chain) src = st->cand_overflow ? 2 : 3;
    Cand best{-__builtin_inf(), INT64_MAX, 0, 0};
    if (src == 3) {
        const unsigned n_cand = (unsigned)st->n_cand;
        for (unsigned b = threadIdx.x; b < n_cand; b += 1024) {
            const unsigned s = a.cand->samp[b];
            double v = a.cand->val[b];
            if (a.weights) v *= a.weights[a.first + s];
            const Cand cand{v, (i64)a.first + s, a.cand->cnt[b], a.cand->pos[b]};
            if (better(cand, best)) best = cand;
        }
    }
    for (unsigned i = threadIdx.x; i < n_active; i += 1024) {
        const unsigned s = a.act[i];
        const u64 c = a.cnt[s];
        if (a.zero_after) a.cnt[s] = 0;  // ready for the next iteration's atomics
        if (a.cnt_mirror) a.cnt_mirror[s] = c;
        double v = (double)c;
        if (a.afsum) {
            const i64 q = a.afsum[s];
            if (a.zero_after) a.afsum[s] = 0;
            if (a.afsum_mirror) a.afsum_mirror[s] = q;
            v = (double)q * a.af_scale;  // exact: q < 2^53 whenever this value is used, and the scale is a power of two
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
        for (int w = 1; w < 16; ++w)
            if (better(wbest[w], best)) best = wbest[w];
        Rec *rc = rec_of(a, a.rank);
        rc->score = n_active ? best.val : 0.0;
        rc->idx = n_active ? best.gidx : -1;
        rc->new_count = n_active ? best.cnt : 0;
        st->best_pos = best.pos;
        if (MODE == 2) {
            late = 0;
            srec.score = rc->score;
            srec.idx = rc->idx;
            srec.new_count = rc->new_count;
        }
        if (a.list_n) {
            u64 n_l = 0;
            for (int c = 0; c < a.n_chunks; ++c) {
                n_l += a.list_n[c];
                a.list_n[c] = 0;
            }
            st->decr_entries += n_l;
            st->decr_gathers += n_l * n_active;
        }
        if (MODE == 0) decide(a);
    }
    if (MODE == 2) {
        __syncthreads();
        if ((int)threadIdx.x < a.n_ranks) {
            // one lane per destination shard; payload first, sequence number last (release, system scope)
            const u64 seq = st->xseq + 1;
            Mailbox *dst = a.peer_mbox[threadIdx.x] + (seq & 1) * a.n_ranks + a.rank;
            __hip_atomic_store(reinterpret_cast<u64 *>(&dst->score), __builtin_bit_cast(u64, srec.score), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<u64 *>(&dst->idx), (u64)srec.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<u64 *>(&dst->new_count), (u64)srec.new_count, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&dst->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            // ... and collect that shard's record of the same exchange from the local mailbox
            const Mailbox *slot = a.mbox + (seq & 1) * a.n_ranks + threadIdx.x;
            if (!mbox_wait(slot, seq, rec_of(a, threadIdx.x))) late = 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (late) {
                st->xerror = 1;  // a shard went away: end the loop, the host reports it
                st->done = 1;
            } else {
                st->xseq += 1;
                decide(a);
            }
        }
    }
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

// Exchange payload: this shard's best column (all chunks back to back) behind its record.
__global__ __launch_bounds__(256) void k_pack(u64 *__restrict__ slot_body, const u64 *__restrict__ cols, u64 wp,
                                              const IterState *__restrict__ st, const unsigned *__restrict__ act)
{
    if (st->done || st->n_active == 0) return;
    const u64 *col = cols + (u64)act[st->best_pos] * wp;
    for (u64 w = (u64)blockIdx.x * 256 + threadIdx.x; w < wp; w += (u64)gridDim.x * 256) slot_body[w] = col[w];
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

// ------------------------------------------------------------------------------------------------
// Ingest helpers
// ------------------------------------------------------------------------------------------------
// Bit transpose of the reference's packing (rows = variants, MSB-first bits along samples,
// convert.py:85) into columns.  One wave = 64 variants x 64 samples: lane l holds the 64 sample bits
// of variant v0+l; 64 ballots turn them into 64 column words; lane j stores sample j's word.
__global__ __launch_bounds__(64) void k_transpose_rows(const unsigned char *__restrict__ rows, u64 row_stride,
                                                       u64 n_rows, u64 first_word, u64 *__restrict__ cols, u64 wp,
                                                       unsigned first_sample, unsigned n_local, unsigned n_total)
{
    const u64 vw = blockIdx.x;     // word (64 variants) inside this upload
    const unsigned sb = blockIdx.y;  // block of 64 local samples
    const int lane = threadIdx.x;
    const u64 v = vw * 64 + lane;
    const unsigned sg0 = first_sample + sb * 64;  // first global sample of the block
    u64 window = 0;  // bit (63 - j) = sample sg0 + j
    if (v < n_rows) {
        const unsigned char *row = rows + v * row_stride;
        const unsigned byte0 = sg0 >> 3, sh = sg0 & 7;
        const unsigned n_bytes = (n_total + 7) >> 3;
        u64 hi = 0;
        for (int b = 0; b < 8; ++b) hi = (hi << 8) | (byte0 + b < n_bytes ? row[byte0 + b] : 0);
        const unsigned nxt = byte0 + 8 < n_bytes ? row[byte0 + 8] : 0;
        window = sh ? (hi << sh) | (nxt >> (8 - sh)) : hi;
    }
    u64 mine = 0;
    for (int j = 0; j < 64; ++j) {
        const u64 word = __ballot((window >> (63 - j)) & 1);
        if (lane == j) mine = word;
    }
    const unsigned s_local = sb * 64 + lane;
    if (s_local < n_local && sg0 + lane < n_total) cols[(u64)s_local * wp + first_word + vw] = mine;
}

// cols[s][w] &= keep[w]   (variants whose AF is exactly 0 are all-zero rows in the reference's matrix)
__global__ __launch_bounds__(256) void k_mask_rows(u64 *__restrict__ cols, u64 wp, const u64 *__restrict__ keep, u64 w_words,
                                                   unsigned n_local)
{
    const u64 total = (u64)n_local * w_words;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total; i += (u64)gridDim.x * 256) {
        const u64 s = i / w_words, w = i % w_words;
        cols[s * wp + w] &= keep[w];
    }
}

// var_count: out[s] += popcount(column s)
__global__ __launch_bounds__(256) void k_col_popcount(const u64 *__restrict__ cols, u64 wp, u64 *__restrict__ out)
{
    const unsigned s = blockIdx.x;
    const u64 *col = cols + (u64)s * wp;
    unsigned acc = 0;
    u64 total = 0;
    for (u64 w = threadIdx.x; w < wp; w += 256) {
        acc += __popcll(col[w]);
        if (acc > 0xF0000000u) { total += acc; acc = 0; }
    }
    total += acc;
    __shared__ u64 part[4];
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = total;
    __syncthreads();
    if (threadIdx.x == 0) out[s] += part[0] + part[1] + part[2] + part[3];
}

// Synthetic chunk contents: thread = one word (64 variants) of one local sample.  Grid = (word blocks,
// min(samples, 65535)); the samples are strided over grid.y -- a 1-D grid of words x samples would exceed
// HIP's 2^32 threads per grid dimension on large chunks and be silently truncated.
__global__ __launch_bounds__(256) void k_synth(u64 *__restrict__ cols, u64 wp, u64 n_var, u64 first_var_global,
                                               u64 seed, unsigned n_total, unsigned first_sample, unsigned octaves,
                                               u64 words_per_col, unsigned n_local)
{
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (w >= words_per_col) return;
    for (unsigned s = blockIdx.y; s < n_local; s += gridDim.y) {
        const unsigned sg = first_sample + s;
        const u64 skey = utm_sample_key(sg);
        u64 word = 0;
        for (int b = 0; b < 64; ++b) {
            const u64 v = w * 64 + b;
            if (v >= n_var) break;
            const u64 key = utm_var_key(seed, first_var_global + v);
            const unsigned thr = utm_var_threshold(key, octaves);
            const unsigned forced = utm_var_forced(key, n_total);
            word |= (u64)utm_cell(key, thr, forced, skey, sg) << b;
        }
        cols[(u64)s * wp + w] = word;
    }
}
