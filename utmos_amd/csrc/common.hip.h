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

#define UTM_HDR_WORDS 8  // one utm_record (64 B)
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
    int loop_unresolved;  // a persistent launch (interval form) ended with an iteration scored but not decided (loop_int.hip.h)
    u64 decr_entries;   // sum over decremental iterations of the newly-covered word count
    u64 decr_gathers;   // ... of (selectable samples x newly-covered words)
    // AF delta passes: sum of the selectable samples' counts (k_count_sum) -- its decrease between two readings, minus
    // the winners' own counts, is the number of AF values the delta passes in between gathered (byte accounting)
    u64 cnt_sum_base;   // right after the first full pass and its pick
    u64 cnt_sum;        // at the end of the last batch
    u64 chain_events;   // iterations so far whose pick needed chains on the spot (k_cand)
};

// The first 32 bytes of IterState: what a scoring workgroup needs at its start, fetched with one scalar load.
struct IterHead {
    int done, prev_valid, prev_local, prev_rank;
    i64 prev_gidx;
    unsigned n_active, best_pos;
};
static_assert(sizeof(IterHead) == 32, "IterHead mirrors the head of IterState");

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
    int cnt_by_pos;  // cnt[] is indexed by position in act[] (integer full passes), else by local sample
    unsigned *list_n;  // decremental mode: per-chunk newly-covered word counts (read for the accounting, then cleared)
    int n_chunks;
    double *fscore;  // sequential AF scores, or nullptr
    double af_scale; // 2^-q
    Mailbox *mbox;             // this shard's mailbox slots [2][n_ranks], or nullptr
    Mailbox *const *peer_mbox; // every shard's mailbox base, as mapped here (index = rank)
    CandBuf *cand;   // verified-parallel AF: candidate list, else nullptr
    int af_is_f64;   // the estimate sums float32-rounded values of float64 AFs
    int af_trunc;    // the fixed-point unit is coarser than the smallest AF's last bit: every addend may lose < 1 unit
    int af_skip_single;  // a single candidate is the winner whatever its exact sum is: skip its chain, report the estimate
    // float64 sums the chains have produced, by sample, with the count each belongs to: a sample's uncovered set only
    // shrinks, so while its count is the one on record its sequential sum is the one on record too -- no chain needed
    u64 *known_cnt;      // n_local (UINT64_MAX: nothing on record), or nullptr
    double *known_val;
    int early_pick;      // the only shard: k_cand also makes the pick when no chain is needed (the chain launches behind it
                         // then return at once; when one is needed, their last workgroup picks)
    Rec *recs;       // every shard's record of the current iteration, recs[rank]
    unsigned mbox_spins;     // polls of a mailbox slot before a peer's record is declared lost (x s_sleep(16))
    int test_mute;           // test hook (UTM_TEST_MUTE_EXCHANGE): in that iteration this shard posts no record (a peer that went away)
    int test_drop;           // test hook (UTM_TEST_DROP_ARRIVAL): this launch withholds one partial count, so that the pick's bounded wait runs out
    int remote_winner_test;  // test hook: treat a local winner's column as remote too (it is then read from the
                             // winner-column buffer the exchange filled), so that one rank can exercise that path
    i64 *res_idx;
    i64 *res_new;
    double *res_score;
    i64 n_var_total;
    unsigned first, n_local, n_total;
    int rank, n_ranks;
};

__device__ __forceinline__ Rec *rec_of(const PickArgs &a, int r)
{
    return a.recs + r;
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
    const u64 *wincol;             // the winner's whole column as the exchange delivered it (ncclBroadcast from its
                                   // owner, or utm_apply_records' winner_col), all chunks back to back
    u64 chunk_off;
    const u64 *const *peer_cols;   // P2P form: this chunk's column base on every rank (IPC-mapped), or nullptr
    const unsigned *peer_first;    // first global sample of every rank
    int fuse;                      // scoring kernels: OR it into the covered tile while staging it
};

// Winner column of the previous iteration (base of the chunk's column), or nullptr.
template <typename STATE>
__device__ __forceinline__ const u64 *pending_column(const STATE *st, const u64 *cols, u64 wp, const Pending &p)
{
    if (!st->prev_valid) return nullptr;
    if (st->prev_local >= 0) return cols + (u64)st->prev_local * wp;
    if (p.peer_cols) return p.peer_cols[st->prev_rank] + (u64)(st->prev_gidx - (i64)p.peer_first[st->prev_rank]) * wp;
    return p.wincol + p.chunk_off;
}
