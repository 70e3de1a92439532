// libutmos_hip.so -- host side of the C ABI declared in include/utmos_hip.h.
//
// Replaces the reference's hot path, utmos/select.py:24-53 (calculate_scores) and :69-112
// (greedy_select), with a device-resident loop: per iteration one scoring launch per chunk (K1, fuses
// the previous winner's `covered |= column`) and one single-workgroup pick (K2: mask, weights, argmax,
// bookkeeping).  Nothing is read back until the loop ends or a 64-iteration batch boundary.
#include "../../include/utmos_hip.h"

#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "kernels.hip.h"

// ---------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(e_ == hipErrorOutOfMemory ? UTM_ENOMEM : UTM_EHIP, "%s -> %s (%s:%d)", #expr,        \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                         \
    } while (0)
#define TRY(expr)                \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != UTM_OK) return rc_; \
    } while (0)

extern "C" const char *utm_last_error(void) { return g_err; }
extern "C" int utm_abi_version(void) { return UTM_ABI_VERSION; }
extern "C" int utm_device_count(int *n)
{
    if (!n) return fail(UTM_EINVAL, "n is NULL");
    HIP_TRY(hipGetDeviceCount(n));
    return UTM_OK;
}

extern "C" int utm_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes)
{
    if (!free_bytes || !total_bytes) return fail(UTM_EINVAL, "output is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(UTM_EINVAL, "device %d of %d", device, n);
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
    return UTM_OK;
}

// ---------------------------------------------------------------------------------------- RCCL (lazy)
// librccl is loaded on first use so that single-GPU runs do not depend on it.
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_rccl;
static int rccl_load()
{
    if (g_rccl.h) return UTM_OK;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(UTM_ECOMM, "cannot load librccl.so: %s", dlerror());
#define SYM(field, name)                                              \
    *(void **)(&g_rccl.field) = dlsym(h, name);                        \
    if (!g_rccl.field) return fail(UTM_ECOMM, "librccl.so lacks %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce")
    SYM(Broadcast, "ncclBroadcast")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(CommCount, "ncclCommCount")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.h = h;
    return UTM_OK;
}
#define NCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess)                                                                          \
            return fail(UTM_ECOMM, "%s -> %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)


// ---------------------------------------------------------------------------------------- environment knobs
// Every environment variable the library reads, in ONE table (include/utmos_hip.h "Environment" documents them).  They
// only move launch shapes, thresholds and test hooks -- never results.  A context reads them when it is created and
// again at every utm_reset (read_tune): nothing on the per-iteration path calls getenv, and a test that flips a knob
// takes effect at its next reset.  The first context of a process names the ones that are set, once, on stderr;
// utm_env_overrides() returns the same list (bench.py puts it into its JSON line).
struct Tune {
    int target_wgs, min_wgs, min_wgs_big, tile_steps, nt_loads, nt_min_mb;  // scoring grid shape
    int fuse_pick, pick_threads, batch;                                      // pick placement, host sync distance
    int af_steps, af_target_wgs, af_tables, af_table_run, af_table_wgs_per_cu, chain_pick, af_verify, af_record, af_defer; // AF kernels
    double af_switch, af_dense_delta;
    int decr_first_batch, decr_interleaved;                                  // decremental mode
    int p2p_replicate, test_remote_winner;                                   // shards
    int persistent, persist_max_mb, persist_wgs_per_cu, persist_claims, persist_ahead_ticks, persist_ahead0_ticks, persist_max_tiles, persist_tile_kib, persist_max_samples, persist_tall_max_samples, persist_af, persist_af_max_tiles, persist_af_interval, persist_spec_ticks, persist_chainers, test_drop_arrival;
    int mbox_spins_log2, test_mute_exchange;                                 // mailbox exchange: patience, test hook   // persistent loop kernel
};
struct KnobDef {
    const char *name;
    int is_double;
    size_t offset;
    double dflt;
};
#define UTM_KNOB_I(env, field, dflt) {env, 0, offsetof(Tune, field), (double)(dflt)}
#define UTM_KNOB_D(env, field, dflt) {env, 1, offsetof(Tune, field), (double)(dflt)}
static const KnobDef g_knobs[] = {
    UTM_KNOB_I("UTM_TARGET_WGS", target_wgs, 32768),
    UTM_KNOB_I("UTM_MIN_WGS", min_wgs, 128),          // (1024 before the pick moved into the launch: 8 KiB tiles now win down to the last iterations)
    UTM_KNOB_I("UTM_MIN_WGS_BIG", min_wgs_big, 8192), // the 32 KiB tile wants a deeper grid (chr22-sized: +17 % with 8 KiB)
    UTM_KNOB_I("UTM_TILE_STEPS", tile_steps, 0),
    UTM_KNOB_I("UTM_NT_LOADS", nt_loads, -1),
    UTM_KNOB_I("UTM_NT_MIN_MB", nt_min_mb, 512),
    UTM_KNOB_I("UTM_FUSE_PICK", fuse_pick, 1),
    UTM_KNOB_I("UTM_PICK_THREADS", pick_threads, 0),
    UTM_KNOB_I("UTM_BATCH", batch, 0),
    UTM_KNOB_I("UTM_AF_STEPS", af_steps, 16),
    UTM_KNOB_I("UTM_AF_TARGET_WGS", af_target_wgs, 16384),
    UTM_KNOB_I("UTM_AF_TABLE_RUN", af_table_run, 0),  // ... tiles per workgroup (0: by the grid, about UTM_AF_TABLE_WGS_PER_CU workgroups per CU)
    UTM_KNOB_I("UTM_AF_TABLE_WGS_PER_CU", af_table_wgs_per_cu, 8),  // (10M x 2,504, same box: 2 -> 1,121 us, 4 -> 951, 8 -> 842, 12 -> 867, 16 -> 921, 24 -> 974)
    UTM_KNOB_I("UTM_AF_TABLES", af_tables, 1),  // full dense AF passes as table lookups (k_score_aft) where the table admits it; 0: k_score_afq
    UTM_KNOB_I("UTM_CHAIN_PICK", chain_pick, 1),
    UTM_KNOB_I("UTM_AF_VERIFY", af_verify, 1),
    UTM_KNOB_I("UTM_AF_RECORD", af_record, 1),
    UTM_KNOB_I("UTM_AF_DEFER", af_defer, 1),
    UTM_KNOB_D("UTM_AF_SWITCH", af_switch, 0.2),
    UTM_KNOB_D("UTM_AF_DENSE_DELTA", af_dense_delta, 0.05),
    UTM_KNOB_I("UTM_DECR_FIRST_BATCH", decr_first_batch, 8),
    UTM_KNOB_I("UTM_DECR_INTERLEAVED", decr_interleaved, 1),
    UTM_KNOB_I("UTM_P2P_REPLICATE", p2p_replicate, 1),
    UTM_KNOB_I("UTM_TEST_REMOTE_WINNER", test_remote_winner, 0),
    UTM_KNOB_I("UTM_PERSISTENT", persistent, 1),
    UTM_KNOB_I("UTM_PERSIST_MAX_MB", persist_max_mb, 560),  // (2,504 samples, same box, against the launches with 8 KiB tiles: +5 % at 0.47 GB, +0.5 % at 0.63 GB, -3.4 ... +2 % at 0.94 GB, -2.6 % at 1.56 GB; with their 16 KiB tiles from 1.8M variants up the launches lead by 2 % at 0.63 GB = 599 MiB, so the limit sits below that)
    UTM_KNOB_I("UTM_PERSIST_WGS_PER_CU", persist_wgs_per_cu, 0),
    UTM_KNOB_I("UTM_PERSIST_MAX_TILES", persist_max_tiles, 32),
    UTM_KNOB_I("UTM_PERSIST_TILE_KIB", persist_tile_kib, 0),
    UTM_KNOB_I("UTM_PERSIST_TALL_MAX_SAMPLES", persist_tall_max_samples, 640),  // tiles of several batches (columns taller than 32 x 8 KiB) only up to this many samples
    UTM_KNOB_I("UTM_PERSIST_MAX_SAMPLES", persist_max_samples, 2560),  // (one chunk of count words for the picker: UTM_LOOP_THREADS x UTM_LOOP_E)
    UTM_KNOB_I("UTM_PERSIST_AF", persist_af, 1),  // the AF form (exact float32 phase) of the persistent loop
    UTM_KNOB_I("UTM_PERSIST_AF_MAX_TILES", persist_af_max_tiles, 26),  // ... both AF forms: 8 KiB tiles only, at most this many (taller matrices keep the launches)
    UTM_KNOB_I("UTM_PERSIST_SPEC_TICKS", persist_spec_ticks, 1000),  // interval form: the chainer works ahead (one chain per record) while iterations take longer than this many 10 ns ticks -- a request waits for a chain in progress; -1: never; -2 (test hook): the chainers leave at once, every request times out and the host decides
    UTM_KNOB_I("UTM_PERSIST_CHAINERS", persist_chainers, 4),  // interval form: blocks that keep sequential float64 sums on record (1..8; a request for 2 x this many samples is served in one go)
    UTM_KNOB_I("UTM_PERSIST_AF_INTERVAL", persist_af_interval, 1),  // ... and its interval form (float64 AF values; float32 sums outside the exact range): candidates and chains inside the picker
    UTM_KNOB_I("UTM_PERSIST_CLAIMS", persist_claims, 1),
    UTM_KNOB_I("UTM_PERSIST_AHEAD0_TICKS", persist_ahead0_ticks, 0),
    UTM_KNOB_I("UTM_PERSIST_AHEAD_TICKS", persist_ahead_ticks, 400),  // 10 ns ticks: the second run-ahead batch goes out this long before the record is due (0: at once)
    UTM_KNOB_I("UTM_TEST_DROP_ARRIVAL", test_drop_arrival, 0),
    UTM_KNOB_I("UTM_MBOX_SPINS_LOG2", mbox_spins_log2, 24),
    UTM_KNOB_I("UTM_TEST_MUTE_EXCHANGE", test_mute_exchange, 0),
};
static void read_tune(Tune *t)
{
    for (const KnobDef &k : g_knobs) {
        const char *v = getenv(k.name);
        const bool set = v && *v;
        char *at = reinterpret_cast<char *>(t) + k.offset;
        if (k.is_double) *reinterpret_cast<double *>(at) = set ? atof(v) : k.dflt;
        else *reinterpret_cast<int *>(at) = set ? atoi(v) : (int)k.dflt;
    }
}
// " NAME=value" for every knob that is set in the environment; returns the number of characters written.
static size_t format_env_overrides(char *buf, size_t cap)
{
    size_t used = 0;
    if (cap) buf[0] = 0;
    for (const KnobDef &k : g_knobs) {
        const char *v = getenv(k.name);
        if (v && *v && used + 64 < cap) used += (size_t)snprintf(buf + used, cap - used, "%s%s=%.24s", used ? " " : "", k.name, v);
    }
    return used;
}
extern "C" int utm_env_overrides(char *buf, uint64_t cap)
{
    if (!buf || cap < 2) return fail(UTM_EINVAL, "buffer too small");
    format_env_overrides(buf, (size_t)cap);
    return UTM_OK;
}

// ---------------------------------------------------------------------------------------- context
struct Chunk {
    u64 n_var = 0;
    u64 w = 0;   // words holding variants
    u64 wp = 0;  // words per column in memory (multiple of 128)
    u64 off = 0; // word offset of this chunk inside a whole-column buffer
    u64 *cols = nullptr;
    u64 *covered = nullptr;
    u64 *covered_alt = nullptr;  // AF delta passes write the updated mask here, then the two swap (host_loop.hip.h)
    void *af = nullptr;    // device: AF in its own type (float or double), wp*64 entries: chains read this
    float *af32 = nullptr; // device: float32 AF (== af when the AF is float32; float64 AF: not kept)
    unsigned *afx = nullptr;  // device: fixed-point table of the parallel estimate, wp*64 entries (af_fixed())
    int index = 0;
    const u64 **d_peer_cols = nullptr;  // device array [n_ranks]: this chunk's column base on every rank (P2P), or null
    std::vector<const u64 *> h_peer_cols;  // the same table on the host
    std::vector<void *> ipc_opened;     // mappings to close
    u64 *replica = nullptr;             // the other shards' columns of this chunk, copied once (when there is room)
    u64 *mask = nullptr;           // AF delta scoring: per word, the bits the last winner newly covered
    unsigned *list_idx = nullptr;  // decremental scoring: words newly covered by the last winner
    u64 *list_val = nullptr;
    u64 *rows_t = nullptr;         // decremental scoring: word-interleaved copy [wp][s_t] (optional: needs the room)
    bool rows_t_valid = false;
    std::vector<float> h_af32;
    std::vector<double> h_af64;
};

struct utm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t flags = 0;
    Tune tune;  // environment knobs as of the last utm_reset (read_tune)
    uint32_t n_total = 0, first = 0, n_local = 0;
    std::vector<Chunk> chunks;
    u64 n_var_total = 0;
    u64 col_words = 0;  // sum of wp over chunks

    unsigned char *d_state = nullptr;  // n_local
    double *d_weights = nullptr;       // n_total or null
    u64 *d_cnt = nullptr;              // n_local
    u64 *d_cnt_alt = nullptr;          // persistent loop: the count words of odd iterations (loop_int.hip.h)
    LoopSync *d_loop_sync = nullptr;   // ... its census counters and the picker's record
    u64 *d_loop_w[4] = {nullptr, nullptr, nullptr, nullptr};  // ... AF form: per-position count-decrease (2) and sum-decrease (2) words
    u64 *d_loop_priv = nullptr;        // ... interval form: the chainers' covered masks (one column each)
    size_t loop_priv_words = 0;
    unsigned *d_claim = nullptr;       // ... its position claim counters (sized for the tile grid at the first launch)
    size_t claim_bytes = 0;
    bool persist_off = false;          // ... a census failed on this context (not every block resident): launch per iteration from now on
    i64 persist_launches = 0, persist_iterations = 0;  // statistics since the last utm_reset
    i64 af_table_passes = 0;           // full dense AF passes taken by k_score_aft
    i64 persist_unresolved = 0;        // ... launches of the interval form that left their last iteration to the verification launch
    bool loop_unresolved = false;      // ... and that iteration is still to be decided (utm_run)
    i64 persist_backoff = 0, persist_backoff_len = 0;  // ... iterations to run as launches before the interval form is tried again
    i64 *d_afsum = nullptr;            // n_local
    double *d_fscore = nullptr;        // n_local
    unsigned *d_act = nullptr;         // n_local
    IterState *d_st = nullptr;
    IterState *h_st = nullptr;  // pinned
    i64 *d_res_idx = nullptr, *d_res_new = nullptr;
    double *d_res_score = nullptr;
    u64 *d_xbuf = nullptr;  // every shard's 64-byte record of the current iteration: xbuf_ranks x UTM_HDR_WORDS
    int xbuf_ranks = 0;
    u64 *d_wincol = nullptr;  // a remote winner's whole column (col_words) as the exchange delivered it: the
                              // ncclBroadcast from its owner, or utm_apply_records' winner_col
    u64 wincol_words = 0;
    StageChunk *d_stage = nullptr;  // chunk table of k_stage_winner (root-free RCCL column exchange)
    bool remote_winner_test = false;  // UTM_TEST_REMOTE_WINNER=1 (tests): read a local winner from d_wincol too
    SeqChunk *d_seq = nullptr;      // chunk table of the chain kernels, covered = the chunks' CURRENT buffers
    SeqChunk *d_seq_alt = nullptr;  // ... the same with the other buffer of every pair (swapped together)
    CandBuf *d_cand = nullptr;
    unsigned *d_arrivals = nullptr;  // workgroups of a k_chain launch that have finished (its last one runs the pick)
    u64 *d_known_cnt = nullptr;      // chains' sums on record, by sample (PickArgs::known_*)
    double *d_known_val = nullptr;
    VerifySync *d_vsync = nullptr;   // k_verify's stage words (one-launch verification on the only shard)
    unsigned verify_launches = 0;    // k_verify launches so far: the number the next one publishes (never reused)
    ChainFast chain_fast{nullptr, 0, nullptr, nullptr, 0, 0};  // device buffers of the chains' fast path
    ChainSeg *d_segs = nullptr;
    // deferred exact AF scores (af_defer.hip.h): the log of newly-covered masks and the finishing launches' buffers
    u64 *d_newly_log = nullptr;       // [defer_slots][col_words], or null (no room / not the only shard's AF loop)
    int defer_slots = UTM_DEFER_SLOTS_LAUNCHES;  // 256 where the persistent loop may take the matrix (a launch logs a whole batch), else 64
    unsigned *d_defer_counts = nullptr;
    u64 *d_defer_offs = nullptr;
    double *d_defer_vals = nullptr;
    i64 defer_lo = 0;                 // result rows below this one carry their final score
    i64 enq_iter = 0;                 // utm_run: the iteration whose launches are being enqueued
    i64 deferred_rows = 0;            // rows whose score came from the deferred launches (statistics)
    u64 *d_cnt_keep = nullptr;   // persistent per-sample counts (mirror of the last full scoring, then decremented)
    i64 *d_afsum_keep = nullptr;
    unsigned *d_listn = nullptr; // per chunk
    size_t listn_cap = 0;
    bool decr_enabled = false;
    bool decr_interleaved = false;  // the chunks carry a word-interleaved copy: decremental iterations stream it
    double decr_threshold = 0;  // <= 0: by layout (1.0 streaming the interleaved copy, 0.2 gathering)
    bool keep_valid = false;     // the persistent counts describe the state right before the pending winner
    i64 last_new = -1;           // new_count of the last row (host copy)
    i64 decr_iterations = 0;
    i64 brute_bytes = 0;
    u64 decr_entries_seen = 0, decr_gathers_seen = 0;
    i64 cov_swaps_enqueued = 0;  // folded AF delta passes enqueued so far (each swaps the chunks' covered pairs)
    u64 cnt_sum_prev = 0;        // AF byte accounting: sum of the selectable samples' counts at the last batch boundary
    u64 *d_varcount = nullptr;
    bool varcount_valid = false;

    std::vector<unsigned char> h_state;  // n_total, as last set (initial states)
    bool have_weights = false;
    int af_mode = UTM_AF_NONE;
    bool af_fixed = false;  // AF runs as the verified-parallel scheme (exact fixed-point estimate + chains)
    bool af_trunc = false;  // ... with a unit coarser than the smallest AF's last mantissa bit (addends lose < 1 unit each)
    bool af_table_ok = false;  // every fixed-point value is below 2^46: the full pass may take the table kernel (k_score_aft)
    int af_q = 0;
    bool prepared = false;  // device loop state matches h_state / AF tables
    bool dirty_tables = true;

    // host mirror of the loop
    i64 iter = 0;             // rows produced
    i64 scored = 0;           // scoring passes run (>= iter)
    unsigned active_ub = 0;   // upper bound of local selectable samples (exact while the loop is alive)
    i64 captured_seen = 0;    // tot_captured as of the last sync
    bool af_exact_scores = true; // chain a lone candidate too, so that the reported score is the reference's exact float64 sum
    bool af_all_exact = false; // latched from the device: AF estimates are exact from here on (no candidates needed)
    bool finished = false;

    // RCCL
    int rank = 0, n_ranks = 1;
    ncclComm_t comm = nullptr;
    bool column_by_allreduce = false;  // RCCL exchange: winner column by a root-free all-reduce instead of a broadcast
    std::vector<unsigned> rank_first, rank_local;  // every shard's sample range (utm_comm_init / utm_p2p_import)
    // P2P: every rank maps every other rank's columns (hipIpc); the winner's column is then read in place
    bool p2p = false;
    bool exported = false;               // peers map (and may have copied) the columns: they must not change any more
    bool replicated = false;             // the peers' columns were copied into this GPU's memory: pending columns are local reads
    u64 replica_bytes = 0;
    unsigned *d_peer_first = nullptr;  // [n_ranks]
    // record mailboxes (device-side exchange without a collective)
    Mailbox *d_mbox = nullptr;           // local slots [2][UTM_MAX_RANKS], uncached device memory, exported to the peers
    Mailbox **d_peer_mbox = nullptr;     // device array [n_ranks] of mapped mailbox bases
    std::vector<void *> mbox_opened;
    Mailbox *mbox_local = nullptr;       // the slots this shard polls (= d_mbox once the peers' mailboxes are mapped)
    bool mbox_ok = false;                // every shard passed the mailbox self-test: utm_run exchanges through them
    bool mbox_single = false;            // ... even with ONE shard (utm_p2p_use_mailboxes(ctx, 2)): the exchange's own cost, measurable on one GPU
    u64 xseq_host = 0;                   // exchanges completed so far (self-test rounds included)

    // stats
    i64 score_launches = 0;
    double loop_ms = 0.0;
    i64 algo_bytes = 0;
    std::vector<hipEvent_t> ev;  // pairs, UTM_FLAG_PROFILE_EVENTS
    size_t ev_used = 0;
    double score_ms = 0.0;
    hipEvent_t ev_loop0 = nullptr, ev_loop1 = nullptr;
};

static inline u64 round_up(u64 x, u64 m) { return (x + m - 1) / m * m; }


// How kernels find the previous winner's column.  With P2P the scoring kernels never fuse the update
// (every workgroup would pull the remote tile over xGMI): k_apply_pending reads the column once instead.
// The per-iteration exchange runs through RCCL (records all-gathered, winner column broadcast) rather than through
// the device mailboxes; a communicator next to working mailboxes is not used by the loop.
// The device mailboxes carry the loop: mapped and self-tested on every shard (and more than one shard, unless forced).
static bool mailbox_exchange(const utm_ctx *c) { return c->mbox_ok && (c->n_ranks > 1 || c->mbox_single); }
static bool rccl_exchange(const utm_ctx *c) { return c->comm && !mailbox_exchange(c); }
// ... in its broadcast form: the host has to learn the winner's rank after every iteration
static bool rccl_needs_root(const utm_ctx *c) { return rccl_exchange(c) && !c->column_by_allreduce; }
// Remote winners are read in place through the hipIpc mappings (not from a local copy / the broadcast buffer).
static bool remote_reads(const utm_ctx *c) { return c->p2p && !c->replicated && !rccl_exchange(c); }

// The AF loop leaves unambiguous winners' exact float64 scores to the deferred launches (af_defer.hip.h): the only
// shard's verified-parallel loop, exact scores wanted, buffers in place, estimates not (yet) all exact.
static bool defer_active(const utm_ctx *c)
{
    return c->d_newly_log && c->af_exact_scores && c->af_mode != UTM_AF_NONE && c->af_fixed && !c->af_all_exact && !c->decr_enabled &&
           c->n_ranks == 1 && c->n_local == c->n_total && !c->comm && !c->p2p;
}

static Pending pending_of(const utm_ctx *c, const Chunk &ch, bool scoring_kernel)
{
    Pending p;
    p.wincol = c->d_wincol;
    p.chunk_off = ch.off;
    p.peer_cols = (c->p2p && !rccl_exchange(c)) ? (const u64 *const *)ch.d_peer_cols : nullptr;
    p.peer_first = c->d_peer_first;
    p.fuse = scoring_kernel && !remote_reads(c);  // a remote column is read once, by k_apply_pending
    return p;
}

static void report_env_once()
{
    static bool said = false;
    if (said) return;
    said = true;
    char line[1024];
    if (format_env_overrides(line, sizeof line)) fprintf(stderr, "libutmos_hip: environment overrides in effect: %s\n", line);
}

extern "C" int utm_ctx_create(int device, uint32_t n_samp_total, uint32_t first_sample, uint32_t n_samp_local,
                              uint32_t flags, utm_ctx **out)
{
    report_env_once();
    if (!out) return fail(UTM_EINVAL, "out is NULL");
    if (n_samp_total == 0 || n_samp_local == 0 || (u64)first_sample + n_samp_local > n_samp_total)
        return fail(UTM_EINVAL, "bad sample range: total %u first %u local %u", n_samp_total, first_sample, n_samp_local);
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return fail(UTM_EINVAL, "device %d not in [0,%d)", device, n_dev);
    HIP_TRY(hipSetDevice(device));
    utm_ctx *c = new utm_ctx();
    c->device = device;
    c->flags = flags;
    read_tune(&c->tune);
    c->n_total = n_samp_total;
    c->first = first_sample;
    c->n_local = n_samp_local;
    c->h_state.assign(n_samp_total, 1);
    *out = c;  // so that a failing allocation below can still be destroyed by the caller
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&c->d_state, n_samp_local));
    HIP_TRY(hipMalloc(&c->d_cnt, ((size_t)n_samp_local + UTM_PICK_PAD) * 8));  // (the fused pick reads whole groups of words)
    HIP_TRY(hipMalloc(&c->d_afsum, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_fscore, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_act, ((size_t)n_samp_local + UTM_PICK_PAD) * 4));
    HIP_TRY(hipMemsetAsync(c->d_act, 0, ((size_t)n_samp_local + UTM_PICK_PAD) * 4, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, ((size_t)n_samp_local + UTM_PICK_PAD) * 8, c->stream));
    HIP_TRY(hipMalloc(&c->d_cnt_alt, ((size_t)n_samp_local + UTM_PICK_PAD) * 8));
    HIP_TRY(hipMemsetAsync(c->d_cnt_alt, 0, ((size_t)n_samp_local + UTM_PICK_PAD) * 8, c->stream));
    HIP_TRY(hipMalloc(&c->d_loop_sync, sizeof(LoopSync)));
    HIP_TRY(hipMalloc(&c->d_varcount, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_st, sizeof(IterState)));
    HIP_TRY(hipHostMalloc(&c->h_st, sizeof(IterState)));
    HIP_TRY(hipMalloc(&c->d_res_idx, ((size_t)n_samp_total + 1) * 8));
    HIP_TRY(hipMalloc(&c->d_res_new, ((size_t)n_samp_total + 1) * 8));
    HIP_TRY(hipMalloc(&c->d_res_score, ((size_t)n_samp_total + 1) * 8));
    HIP_TRY(hipMalloc(&c->d_xbuf, UTM_HDR_WORDS * 8));
    HIP_TRY(hipMalloc(&c->d_cand, sizeof(CandBuf)));
    HIP_TRY(hipMalloc(&c->d_arrivals, 128));
    HIP_TRY(hipMemsetAsync(c->d_arrivals, 0, 128, c->stream));
    HIP_TRY(hipMalloc(&c->d_known_cnt, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_known_val, (size_t)n_samp_local * 8));
    HIP_TRY(hipMemsetAsync(c->d_known_cnt, 0xFF, (size_t)n_samp_local * 8, c->stream));
    HIP_TRY(hipMalloc(&c->d_vsync, sizeof(VerifySync)));
    HIP_TRY(hipMemsetAsync(c->d_vsync, 0, sizeof(VerifySync), c->stream));
    HIP_TRY(hipMalloc(&c->d_cnt_keep, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_afsum_keep, (size_t)n_samp_local * 8));
    c->decr_enabled = flags & UTM_FLAG_DECREMENTAL;
    c->remote_winner_test = c->tune.test_remote_winner == 1;
    c->xbuf_ranks = 1;
    HIP_TRY(hipEventCreate(&c->ev_loop0));
    HIP_TRY(hipEventCreate(&c->ev_loop1));
    return UTM_OK;
}

static void p2p_close(utm_ctx *c);

extern "C" int utm_ctx_destroy(utm_ctx *c)
{
    if (!c) return UTM_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    p2p_close(c);
    (void)hipFree(c->d_mbox);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    for (auto &ch : c->chunks) {
        (void)hipFree(ch.cols);
        (void)hipFree(ch.covered);
        (void)hipFree(ch.covered_alt);
        if ((void *)ch.af32 != ch.af) (void)hipFree(ch.af32);
        (void)hipFree(ch.afx);
        (void)hipFree(ch.af);
        (void)hipFree(ch.list_idx);
        (void)hipFree(ch.rows_t);
        (void)hipFree(ch.list_val);
        (void)hipFree(ch.mask);
    }
    (void)hipFree(c->d_cand);
    (void)hipFree(c->d_arrivals);
    (void)hipFree(c->d_vsync);
    (void)hipFree(c->d_known_cnt); (void)hipFree(c->d_known_val);
    (void)hipFree(c->d_segs); (void)hipFree(c->chain_fast.counts); (void)hipFree(c->chain_fast.vals);
    (void)hipFree(c->d_newly_log); (void)hipFree(c->d_defer_counts); (void)hipFree(c->d_defer_offs); (void)hipFree(c->d_defer_vals);
    (void)hipFree(c->d_cnt_keep); (void)hipFree(c->d_afsum_keep); (void)hipFree(c->d_listn);
    (void)hipFree(c->d_cnt_alt); (void)hipFree(c->d_loop_sync); (void)hipFree(c->d_claim); (void)hipFree(c->d_loop_priv);
    for (auto *w : c->d_loop_w) (void)hipFree(w);
    (void)hipFree(c->d_state); (void)hipFree(c->d_weights); (void)hipFree(c->d_cnt); (void)hipFree(c->d_afsum); (void)hipFree(c->d_fscore);
    (void)hipFree(c->d_act); (void)hipFree(c->d_st); (void)hipFree(c->d_res_idx); (void)hipFree(c->d_res_new); (void)hipFree(c->d_res_score);
    (void)hipFree(c->d_xbuf); (void)hipFree(c->d_wincol); (void)hipFree(c->d_stage); (void)hipFree(c->d_seq); (void)hipFree(c->d_seq_alt); (void)hipFree(c->d_varcount);
    if (c->h_st) (void)hipHostFree(c->h_st);
    for (auto e : c->ev) (void)hipEventDestroy(e);
    if (c->ev_loop0) (void)hipEventDestroy(c->ev_loop0);
    if (c->ev_loop1) (void)hipEventDestroy(c->ev_loop1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return UTM_OK;
}

// Blocking copy ON THE CONTEXT'S STREAM.  The stream is non-blocking, so null-stream calls (hipMemcpy, hipMemset)
// are not ordered with the kernels launched on it; every transfer goes through here instead.
static hipError_t copy_sync(utm_ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
    return e == hipSuccess ? hipStreamSynchronize(c->stream) : e;
}

// A device allocation that lives for one call: released on every way out, error returns included.
template <typename T>
struct Scratch {
    T *p = nullptr;
    Scratch() = default;
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, count * sizeof(T)); }
    operator T *() const { return p; }
};

#define CTX(c)                                        \
    if (!(c)) return fail(UTM_EINVAL, "ctx is NULL"); \
    HIP_TRY(hipSetDevice((c)->device))

static int chunk_of(utm_ctx *c, int32_t chunk, Chunk **out)
{
    if (chunk < 0 || (size_t)chunk >= c->chunks.size()) return fail(UTM_EINVAL, "chunk %d not in [0,%zu)", chunk, c->chunks.size());
    *out = &c->chunks[chunk];
    return UTM_OK;
}

#include "host_matrix.hip.h"
#include "host_options.hip.h"
#include "host_reset.hip.h"
#include "host_loop.hip.h"
#include "host_shard.hip.h"
