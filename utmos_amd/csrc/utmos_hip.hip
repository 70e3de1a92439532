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

// ---------------------------------------------------------------------------------------- context
struct Chunk {
    u64 n_var = 0;
    u64 w = 0;   // words holding variants
    u64 wp = 0;  // words per column in memory (multiple of 128)
    u64 off = 0; // word offset of this chunk inside a whole-column buffer
    u64 *cols = nullptr;
    u64 *covered = nullptr;
    void *af = nullptr;    // device: AF in its own type (float or double), wp*64 entries: chains read this
    float *af32 = nullptr; // device: float32 AF for the parallel estimate (== af when the AF is float32)
    int index = 0;
    const u64 **d_peer_cols = nullptr;  // device array [n_ranks]: this chunk's column base on every rank (P2P), or null
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
    uint32_t n_total = 0, first = 0, n_local = 0;
    std::vector<Chunk> chunks;
    u64 n_var_total = 0;
    u64 col_words = 0;  // sum of wp over chunks

    unsigned char *d_state = nullptr;  // n_local
    double *d_weights = nullptr;       // n_total or null
    u64 *d_cnt = nullptr;              // n_local
    i64 *d_afsum = nullptr;            // n_local
    double *d_fscore = nullptr;        // n_local
    unsigned *d_act = nullptr;         // n_local
    IterState *d_st = nullptr;
    IterState *h_st = nullptr;  // pinned
    i64 *d_res_idx = nullptr, *d_res_new = nullptr;
    double *d_res_score = nullptr;
    u64 *d_xbuf = nullptr;  // n_ranks slots of slot_words
    u64 slot_words = UTM_HDR_WORDS;
    int xbuf_ranks = 0;
    u64 xbuf_slot_words = UTM_HDR_WORDS;  // slot size the buffer was allocated for
    SeqChunk *d_seq = nullptr;
    CandBuf *d_cand = nullptr;
    ChainFast chain_fast{nullptr, 0, nullptr, nullptr};  // device buffers of the chains' fast path
    ChainSeg *d_segs = nullptr;
    u64 *d_cnt_keep = nullptr;   // persistent per-sample counts (mirror of the last full scoring, then decremented)
    i64 *d_afsum_keep = nullptr;
    unsigned *d_listn = nullptr; // per chunk
    size_t listn_cap = 0;
    bool decr_enabled = false;
    bool decr_interleaved = false;  // the chunks carry a word-interleaved copy: decremental iterations stream it
    double decr_threshold = 0;  // <= 0: by layout (0.5 streaming the interleaved copy, 0.2 gathering)
    bool keep_valid = false;     // the persistent counts describe the state right before the pending winner
    i64 last_new = -1;           // new_count of the last row (host copy)
    i64 decr_iterations = 0;
    i64 brute_bytes = 0;
    u64 decr_entries_seen = 0, decr_gathers_seen = 0;
    u64 *d_varcount = nullptr;
    bool varcount_valid = false;

    std::vector<unsigned char> h_state;  // n_total, as last set (initial states)
    bool have_weights = false;
    int af_mode = UTM_AF_NONE;
    bool af_fixed = false;  // AF runs as the verified-parallel scheme (exact fixed-point estimate + chains)
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
    bool mbox_ok = false;                // every shard passed the mailbox self-test: utm_run exchanges through them
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
static Pending pending_of(const utm_ctx *c, const Chunk &ch, bool scoring_kernel)
{
    Pending p;
    p.xbuf = c->d_xbuf;
    p.slot_words = c->xbuf_slot_words;
    p.chunk_off = ch.off;
    p.peer_cols = c->p2p ? (const u64 *const *)ch.d_peer_cols : nullptr;
    p.peer_first = c->d_peer_first;
    p.fuse = scoring_kernel && (!c->p2p || c->replicated);  // a remote column is read once, by k_apply_pending
    return p;
}

extern "C" int utm_ctx_create(int device, uint32_t n_samp_total, uint32_t first_sample, uint32_t n_samp_local,
                              uint32_t flags, utm_ctx **out)
{
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
    c->n_total = n_samp_total;
    c->first = first_sample;
    c->n_local = n_samp_local;
    c->h_state.assign(n_samp_total, 1);
    *out = c;  // so that a failing allocation below can still be destroyed by the caller
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&c->d_state, n_samp_local));
    HIP_TRY(hipMalloc(&c->d_cnt, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_afsum, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_fscore, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_act, (size_t)n_samp_local * 4));
    HIP_TRY(hipMalloc(&c->d_varcount, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_st, sizeof(IterState)));
    HIP_TRY(hipHostMalloc(&c->h_st, sizeof(IterState)));
    HIP_TRY(hipMalloc(&c->d_res_idx, ((size_t)n_samp_total + 1) * 8));
    HIP_TRY(hipMalloc(&c->d_res_new, ((size_t)n_samp_total + 1) * 8));
    HIP_TRY(hipMalloc(&c->d_res_score, ((size_t)n_samp_total + 1) * 8));
    HIP_TRY(hipMalloc(&c->d_xbuf, UTM_HDR_WORDS * 8));
    HIP_TRY(hipMalloc(&c->d_cand, sizeof(CandBuf)));
    HIP_TRY(hipMalloc(&c->d_cnt_keep, (size_t)n_samp_local * 8));
    HIP_TRY(hipMalloc(&c->d_afsum_keep, (size_t)n_samp_local * 8));
    c->decr_enabled = flags & UTM_FLAG_DECREMENTAL;
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
        if ((void *)ch.af32 != ch.af) (void)hipFree(ch.af32);
        (void)hipFree(ch.af);
        (void)hipFree(ch.list_idx);
        (void)hipFree(ch.rows_t);
        (void)hipFree(ch.list_val);
        (void)hipFree(ch.mask);
    }
    (void)hipFree(c->d_cand);
    (void)hipFree(c->d_segs); (void)hipFree(c->chain_fast.counts); (void)hipFree(c->chain_fast.vals);
    (void)hipFree(c->d_cnt_keep); (void)hipFree(c->d_afsum_keep); (void)hipFree(c->d_listn);
    (void)hipFree(c->d_state); (void)hipFree(c->d_weights); (void)hipFree(c->d_cnt); (void)hipFree(c->d_afsum); (void)hipFree(c->d_fscore);
    (void)hipFree(c->d_act); (void)hipFree(c->d_st); (void)hipFree(c->d_res_idx); (void)hipFree(c->d_res_new); (void)hipFree(c->d_res_score);
    (void)hipFree(c->d_xbuf); (void)hipFree(c->d_seq); (void)hipFree(c->d_varcount);
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

#define CTX(c)                                        \
    if (!(c)) return fail(UTM_EINVAL, "ctx is NULL"); \
    HIP_TRY(hipSetDevice((c)->device))

static int chunk_of(utm_ctx *c, int32_t chunk, Chunk **out)
{
    if (chunk < 0 || (size_t)chunk >= c->chunks.size()) return fail(UTM_EINVAL, "chunk %d not in [0,%zu)", chunk, c->chunks.size());
    *out = &c->chunks[chunk];
    return UTM_OK;
}

// ---------------------------------------------------------------------------------------- matrix
extern "C" int utm_add_chunk(utm_ctx *c, uint64_t n_var, int32_t *chunk)
{
    CTX(c);
    if (n_var == 0) return fail(UTM_EINVAL, "empty chunk");
    if (c->comm) return fail(UTM_ESTATE, "chunks must be added before utm_comm_init");
    Chunk ch;
    ch.n_var = n_var;
    ch.w = (n_var + 63) / 64;
    ch.wp = round_up(ch.w, UTM_STEP_WORDS);
    ch.off = c->col_words;
    const size_t bytes = (size_t)c->n_local * ch.wp * 8;
    HIP_TRY(hipMalloc(&ch.cols, bytes));
    HIP_TRY(hipMemsetAsync(ch.cols, 0, bytes, c->stream));
    HIP_TRY(hipMalloc(&ch.covered, ch.wp * 8));
    HIP_TRY(hipMemsetAsync(ch.covered, 0, ch.wp * 8, c->stream));
    ch.index = (int)c->chunks.size();
    c->chunks.push_back(std::move(ch));
    c->n_var_total += n_var;
    c->col_words += c->chunks.back().wp;
    c->slot_words = UTM_HDR_WORDS + c->col_words;
    c->prepared = false;
    c->dirty_tables = true;
    c->varcount_valid = false;
    if (chunk) *chunk = (int32_t)c->chunks.size() - 1;
    return UTM_OK;
}

extern "C" int utm_upload_columns(utm_ctx *c, int32_t chunk, uint32_t first_col, uint32_t n_cols,
                                  const uint64_t *cols, uint64_t stride_words)
{
    CTX(c);
    if (c->exported) return fail(UTM_ESTATE, "the columns are exported to other shards (utm_p2p_export): fill them before exporting");
    Chunk *ch;
    TRY(chunk_of(c, chunk, &ch));
    if (!cols || (u64)first_col + n_cols > c->n_local || stride_words < ch->w)
        return fail(UTM_EINVAL, "bad column range/stride (first %u n %u stride %llu < %llu words)", first_col, n_cols,
                    (u64)stride_words, ch->w);
    HIP_TRY(hipMemcpy2DAsync(ch->cols + (u64)first_col * ch->wp, ch->wp * 8, cols, stride_words * 8, ch->w * 8, n_cols,
                             hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ch->rows_t_valid = false;
    c->prepared = false;
    c->varcount_valid = false;
    return UTM_OK;
}

extern "C" int utm_download_columns(utm_ctx *c, int32_t chunk, uint32_t first_col, uint32_t n_cols, uint64_t *cols,
                                    uint64_t stride_words)
{
    CTX(c);
    Chunk *ch;
    TRY(chunk_of(c, chunk, &ch));
    if (!cols || (u64)first_col + n_cols > c->n_local || stride_words < ch->w)
        return fail(UTM_EINVAL, "bad column range/stride");
    HIP_TRY(hipMemcpy2DAsync(cols, stride_words * 8, ch->cols + (u64)first_col * ch->wp, ch->wp * 8, ch->w * 8, n_cols,
                             hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UTM_OK;
}

extern "C" int utm_upload_rows_packed(utm_ctx *c, int32_t chunk, uint64_t first_var, uint64_t n_rows,
                                      const uint8_t *rows, uint64_t row_stride_bytes)
{
    CTX(c);
    if (c->exported) return fail(UTM_ESTATE, "the columns are exported to other shards (utm_p2p_export): fill them before exporting");
    Chunk *ch;
    TRY(chunk_of(c, chunk, &ch));
    if (!rows || n_rows == 0) return fail(UTM_EINVAL, "no rows");
    if (first_var % 64 || first_var + n_rows > ch->n_var) return fail(UTM_EINVAL, "first_var must be a multiple of 64 and rows must fit the chunk");
    if (row_stride_bytes < ((u64)c->n_total + 7) / 8) return fail(UTM_EINVAL, "row stride shorter than ceil(S/8)");
    // staged in slabs of at most 64 MiB
    const u64 slab_rows = std::max<u64>(64, ((64ull << 20) / row_stride_bytes) / 64 * 64);
    unsigned char *d_rows = nullptr;
    HIP_TRY(hipMalloc(&d_rows, std::min(slab_rows, round_up(n_rows, 64)) * row_stride_bytes));
    int rc = UTM_OK;
    for (u64 r0 = 0; r0 < n_rows && rc == UTM_OK; r0 += slab_rows) {
        const u64 nr = std::min(slab_rows, n_rows - r0);
        hipError_t e = hipMemcpyAsync(d_rows, rows + r0 * row_stride_bytes, nr * row_stride_bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = fail(UTM_EHIP, "row upload: %s", hipGetErrorString(e)); break; }
        dim3 grid((unsigned)((nr + 63) / 64), (c->n_local + 63) / 64);
        hipLaunchKernelGGL(k_transpose_rows, grid, dim3(64), 0, c->stream, d_rows, (u64)row_stride_bytes, nr,
                           (first_var + r0) / 64, ch->cols, ch->wp, c->first, c->n_local, c->n_total);
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(UTM_EHIP, "row transpose: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_rows);
    ch->rows_t_valid = false;
    c->prepared = false;
    c->varcount_valid = false;
    return rc;
}

static int ensure_var_count(utm_ctx *c)
{
    if (c->varcount_valid) return UTM_OK;
    HIP_TRY(hipMemsetAsync(c->d_varcount, 0, (size_t)c->n_local * 8, c->stream));
    for (auto &ch : c->chunks)
        hipLaunchKernelGGL(k_col_popcount, dim3(c->n_local), dim3(256), 0, c->stream, ch.cols, ch.wp, c->d_varcount);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->varcount_valid = true;
    return UTM_OK;
}

extern "C" int utm_var_count(utm_ctx *c, int64_t *out)
{
    CTX(c);
    if (!out) return fail(UTM_EINVAL, "out is NULL");
    TRY(ensure_var_count(c));
    HIP_TRY(copy_sync(c, out, c->d_varcount, (size_t)c->n_local * 8, hipMemcpyDeviceToHost));
    return UTM_OK;
}

extern "C" int utm_synth_fill(utm_ctx *c, int32_t chunk, uint64_t seed, uint64_t first_var_global)
{
    CTX(c);
    if (c->exported) return fail(UTM_ESTATE, "the columns are exported to other shards (utm_p2p_export): fill them before exporting");
    Chunk *ch;
    TRY(chunk_of(c, chunk, &ch));
    const u64 blocks_per_col = (ch->w + 255) / 256;
    if (blocks_per_col * 256 >= (1ull << 32)) return fail(UTM_EINVAL, "chunk too large for the generator");
    const dim3 grid((unsigned)blocks_per_col, std::min(c->n_local, 65535u));
    hipLaunchKernelGGL(k_synth, grid, dim3(256), 0, c->stream, ch->cols, ch->wp, ch->n_var, (u64)first_var_global, (u64)seed,
                       c->n_total, c->first, utm_octaves(c->n_total), ch->w, c->n_local);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    ch->rows_t_valid = false;
    c->prepared = false;
    c->varcount_valid = false;
    return UTM_OK;
}

extern "C" int utm_synth_host(uint64_t seed, uint64_t first_var_global, uint64_t n_var, uint32_t n_samp_total,
                              uint32_t first_sample, uint32_t n_samp, uint64_t *cols, uint64_t stride_words,
                              float *af_out)
{
    const uint32_t oct = utm_octaves(n_samp_total);
    const u64 w = (n_var + 63) / 64;
    if (cols && stride_words < w) return fail(UTM_EINVAL, "stride shorter than ceil(n_var/64)");
    std::vector<uint64_t> key(n_var);
    std::vector<uint32_t> thr(n_var), forced(n_var);
    for (u64 v = 0; v < n_var; ++v) {
        key[v] = utm_var_key(seed, first_var_global + v);
        thr[v] = utm_var_threshold(key[v], oct);
        forced[v] = utm_var_forced(key[v], n_samp_total);
        if (af_out) af_out[v] = utm_var_af(thr[v], n_samp_total);
    }
    if (cols)
        for (uint32_t s = 0; s < n_samp; ++s) {
            const uint32_t sg = first_sample + s;
            const uint64_t skey = utm_sample_key(sg);
            uint64_t *col = cols + (u64)s * stride_words;
            for (u64 wi = 0; wi < w; ++wi) {
                uint64_t word = 0;
                for (int b = 0; b < 64; ++b) {
                    const u64 v = wi * 64 + b;
                    if (v >= n_var) break;
                    word |= (uint64_t)utm_cell(key[v], thr[v], forced[v], skey, sg) << b;
                }
                col[wi] = word;
            }
        }
    return UTM_OK;
}

// ---------------------------------------------------------------------------------------- options
extern "C" int utm_set_sample_state(utm_ctx *c, const uint8_t *state)
{
    CTX(c);
    if (!state) return fail(UTM_EINVAL, "state is NULL");
    for (uint32_t s = 0; s < c->n_total; ++s)
        if (state[s] > 2) return fail(UTM_EINVAL, "state[%u] = %u not in {0,1,2}", s, state[s]);
    c->h_state.assign(state, state + c->n_total);
    c->prepared = false;
    return UTM_OK;
}

extern "C" int utm_set_weights(utm_ctx *c, const double *w)
{
    CTX(c);
    if (!w) {
        c->have_weights = false;
        return UTM_OK;
    }
    for (uint32_t s = 0; s < c->n_total; ++s)
        if (!isfinite(w[s])) return fail(UTM_EINVAL, "weights[%u] is not finite", s);
    if (!c->d_weights) HIP_TRY(hipMalloc(&c->d_weights, (size_t)c->n_total * 8));
    HIP_TRY(copy_sync(c, c->d_weights, w, (size_t)c->n_total * 8, hipMemcpyHostToDevice));
    c->have_weights = true;
    return UTM_OK;
}

extern "C" int utm_set_af(utm_ctx *c, int32_t chunk, int mode, const void *af)
{
    CTX(c);
    if (mode == UTM_AF_NONE) {
        if (af) return fail(UTM_EINVAL, "UTM_AF_NONE takes af == NULL");
        for (auto &ch : c->chunks) { ch.h_af32.clear(); ch.h_af64.clear(); }
        c->af_mode = UTM_AF_NONE;
        c->dirty_tables = true;
        c->prepared = false;
        return UTM_OK;
    }
    Chunk *ch;
    TRY(chunk_of(c, chunk, &ch));
    if (mode != UTM_AF_F32 && mode != UTM_AF_F64) return fail(UTM_EINVAL, "mode %d", mode);
    if (!af) return fail(UTM_EINVAL, "af is NULL");
    if (c->af_mode != UTM_AF_NONE && c->af_mode != mode) {
        for (auto &o : c->chunks)
            if (&o != ch && (!o.h_af32.empty() || !o.h_af64.empty()))
                return fail(UTM_ESTATE, "all chunks must use one AF mode");
    }
    // A variant whose AF is 0.0 is an all-zero row of the reference's float matrix (presence * AF):
    // never counted, never scored, never covered.  var_count is taken from the boolean matrix
    // (select.py:281-284), so it is latched before such rows are cleared.
    std::vector<u64> keep(ch->w, ~0ull);
    bool any_zero = false;
    for (u64 v = 0; v < ch->n_var; ++v) {
        const double a = mode == UTM_AF_F32 ? (double)((const float *)af)[v] : ((const double *)af)[v];
        if (!isfinite(a) || a < 0) return fail(UTM_EINVAL, "AF[%llu] = %g must be finite and >= 0", v, a);
        if (a == 0.0) { keep[v >> 6] &= ~(1ull << (v & 63)); any_zero = true; }
    }
    if (any_zero) {
        if (c->exported) return fail(UTM_ESTATE, "the columns are exported to other shards (utm_p2p_export): set the AF before exporting");
        TRY(ensure_var_count(c));
        ch->rows_t_valid = false;
        u64 *d_keep = nullptr;
        HIP_TRY(hipMalloc(&d_keep, ch->w * 8));
        HIP_TRY(copy_sync(c, d_keep, keep.data(), ch->w * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mask_rows, dim3(4096), dim3(256), 0, c->stream, ch->cols, ch->wp, d_keep, ch->w, c->n_local);
        hipError_t e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_keep);
        if (e != hipSuccess) return fail(UTM_EHIP, "mask rows: %s", hipGetErrorString(e));
    }
    ch->h_af32.clear();
    ch->h_af64.clear();
    if (mode == UTM_AF_F32) ch->h_af32.assign((const float *)af, (const float *)af + ch->n_var);
    else ch->h_af64.assign((const double *)af, (const double *)af + ch->n_var);
    c->af_mode = mode;
    c->dirty_tables = true;
    c->prepared = false;
    return UTM_OK;
}

// Decide the AF arithmetic and build the device tables.
static int build_af_tables(utm_ctx *c)
{
    if (!c->dirty_tables) return UTM_OK;
    for (auto &ch : c->chunks) {
        if ((void *)ch.af32 != ch.af) (void)hipFree(ch.af32);
        (void)hipFree(ch.af);
        ch.af = nullptr;
        ch.af32 = nullptr;
    }
    (void)hipFree(c->d_seq);
    c->d_seq = nullptr;
    c->af_fixed = false;
    c->af_q = 0;
    if (c->af_mode == UTM_AF_NONE) { c->dirty_tables = false; return UTM_OK; }
    for (auto &ch : c->chunks)
        if ((c->af_mode == UTM_AF_F32 ? ch.h_af32.size() : ch.h_af64.size()) != ch.n_var)
            return fail(UTM_ESTATE, "AF not set for every chunk");
    // float32 view of the AF values: the data itself (F32) or its rounding (F64, estimate only)
    std::vector<std::vector<float>> v32(c->chunks.size());
    bool representable = true;
    int e_min = 1000, e_max = -1000;
    long double mass = 0;
    for (size_t k = 0; k < c->chunks.size(); ++k) {
        Chunk &ch = c->chunks[k];
        v32[k].assign(ch.wp * 64, 0.0f);
        for (u64 v = 0; v < ch.n_var; ++v) {
            const float a = c->af_mode == UTM_AF_F32 ? ch.h_af32[v] : (float)ch.h_af64[v];
            v32[k][v] = a;
            if (c->af_mode == UTM_AF_F64 && ch.h_af64[v] != 0.0 && (!(a > 0.0f) || !isfinite(a))) representable = false;
            if (a == 0.0f) continue;
            if (fpclassify(a) == FP_SUBNORMAL) { representable = false; continue; }
            int e;
            frexpf(a, &e);  // a = f * 2^e, f in [0.5, 1): the leading bit has weight 2^(e-1)
            e_min = std::min(e_min, e - 1);
            e_max = std::max(e_max, e - 1);
            mass += a;
        }
    }
    if (e_min == 1000) e_min = e_max = 0;
    // Every float32 a > 0 is m * 2^(e-23), m < 2^24: a multiple of 2^-q for q = 23 - e_min.  The
    // estimate kernel sums a * 2^q = m << (e - e_min) as int64: needs q >= 0 and mass * 2^q < 2^62.
    const int q = 23 - e_min;
    if (!(c->flags & UTM_FLAG_AF_SEQUENTIAL) && representable && q >= 0 && q <= 149 &&
        e_max - e_min <= 38 && mass * ldexpl(1.0L, q - 62) < 1.0L) {
        c->af_fixed = true;
        c->af_q = q;
    }
    std::vector<SeqChunk> seq;
    for (size_t k = 0; k < c->chunks.size(); ++k) {
        Chunk &ch = c->chunks[k];
        const size_t n = ch.wp * 64;
        HIP_TRY(hipMalloc(&ch.af32, n * 4));
        HIP_TRY(copy_sync(c, ch.af32, v32[k].data(), n * 4, hipMemcpyHostToDevice));
        if (c->af_mode == UTM_AF_F32) {
            ch.af = ch.af32;
        } else {
            HIP_TRY(hipMalloc(&ch.af, n * 8));
            // (the context's stream is non-blocking: never mix in null-stream work, it would not be ordered with it)
            HIP_TRY(hipMemsetAsync(ch.af, 0, n * 8, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(copy_sync(c, ch.af, ch.h_af64.data(), ch.n_var * 8, hipMemcpyHostToDevice));
        }
        seq.push_back(SeqChunk{ch.cols, ch.covered, ch.af, ch.wp, ch.w});
    }
    HIP_TRY(hipMalloc(&c->d_seq, seq.size() * sizeof(SeqChunk)));
    HIP_TRY(copy_sync(c, c->d_seq, seq.data(), seq.size() * sizeof(SeqChunk), hipMemcpyHostToDevice));
    // segment table + buffers of the chains' fast path
    (void)hipFree(c->d_segs); (void)hipFree(c->chain_fast.counts); (void)hipFree(c->chain_fast.vals);
    c->d_segs = nullptr;
    c->chain_fast = ChainFast{nullptr, 0, nullptr, nullptr};
    if (c->af_fixed)
        for (auto &ch : c->chunks)
            if (!ch.mask) HIP_TRY(hipMalloc(&ch.mask, ch.wp * 8));
    if (c->af_fixed) {
        std::vector<ChainSeg> segs;
        for (size_t k = 0; k < c->chunks.size(); ++k)
            for (u64 w0 = 0; w0 < c->chunks[k].w; w0 += UTM_SEG_WORDS) segs.push_back(ChainSeg{(int)k, w0});
        const size_t n = segs.size();
        if (n * UTM_FAST_CAND * UTM_SEG_CAP * 8 <= (4ull << 30)) {  // keep the scratch within 4 GiB
            HIP_TRY(hipMalloc(&c->d_segs, n * sizeof(ChainSeg)));
            HIP_TRY(copy_sync(c, c->d_segs, segs.data(), n * sizeof(ChainSeg), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(&c->chain_fast.counts, n * UTM_FAST_CAND * 4));
            HIP_TRY(hipMalloc(&c->chain_fast.vals, n * UTM_FAST_CAND * UTM_SEG_CAP * 8));
            c->chain_fast.segs = c->d_segs;
            c->chain_fast.n_segs = (int)n;
        }
    }
    c->dirty_tables = false;
    return UTM_OK;
}

// ---------------------------------------------------------------------------------------- loop set-up
static int ensure_xbuf(utm_ctx *c, int n_ranks)
{
    // records only, unless whole columns travel through the slots (column all-gather / host-staged without P2P)
    const u64 slot = ((n_ranks == 1 && !c->comm) || c->p2p) ? UTM_HDR_WORDS : c->slot_words;
    if (c->d_xbuf && c->xbuf_ranks == n_ranks && c->xbuf_slot_words == slot) return UTM_OK;
    (void)hipFree(c->d_xbuf);
    c->d_xbuf = nullptr;
    HIP_TRY(hipMalloc(&c->d_xbuf, (size_t)n_ranks * slot * 8));
    HIP_TRY(hipMemsetAsync(c->d_xbuf, 0, (size_t)n_ranks * slot * 8, c->stream));  // same stream as every later use
    c->xbuf_ranks = n_ranks;
    c->xbuf_slot_words = slot;
    return UTM_OK;
}

// Decremental mode: the word-interleaved second copy of every chunk (decremental.hip.h), all chunks or none, only
// when it fits next to a reserve of free HBM.  UTM_DECR_INTERLEAVED=0 keeps the gather form (what a context
// without the room runs).
static u64 interleaved_stride(const utm_ctx *c) { return round_up((u64)c->n_local, 64); }

static int ensure_interleaved(utm_ctx *c)
{
    const char *env = getenv("UTM_DECR_INTERLEAVED");  // read per reset: tests flip it
    const bool wanted = !(env && *env == '0');
    const u64 s_t = interleaved_stride(c);
    bool have_all = true;
    u64 need = 0;
    for (auto &ch : c->chunks)
        if (!ch.rows_t) {
            have_all = false;
            need += ch.wp * s_t * 8;
        }
    if (!have_all || !wanted) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const bool fits = (u64)free_b > need + (4ull << 30);
        if (!wanted || !fits) {
            for (auto &ch : c->chunks) {
                (void)hipFree(ch.rows_t);
                ch.rows_t = nullptr;
                ch.rows_t_valid = false;
            }
            c->decr_interleaved = false;
            return UTM_OK;
        }
        for (auto &ch : c->chunks)
            if (!ch.rows_t) {
                if (hipMalloc(&ch.rows_t, ch.wp * s_t * 8) != hipSuccess) {
                    (void)hipGetLastError();
                    for (auto &o : c->chunks) {
                        (void)hipFree(o.rows_t);
                        o.rows_t = nullptr;
                        o.rows_t_valid = false;
                    }
                    c->decr_interleaved = false;
                    return UTM_OK;
                }
                ch.rows_t_valid = false;
            }
    }
    for (auto &ch : c->chunks)
        if (!ch.rows_t_valid) {
            hipLaunchKernelGGL(k_interleave, dim3((unsigned)(ch.wp / 64), (unsigned)(s_t / 64)), dim3(256), 0, c->stream, ch.cols,
                               ch.wp, c->n_local, s_t, ch.rows_t);
            ch.rows_t_valid = true;
        }
    HIP_TRY(hipGetLastError());
    c->decr_interleaved = true;
    return UTM_OK;
}

extern "C" int utm_reset(utm_ctx *c)
{
    CTX(c);
    if (c->chunks.empty()) return fail(UTM_ESTATE, "no chunks");
    TRY(build_af_tables(c));
    TRY(ensure_xbuf(c, std::max(c->xbuf_ranks, c->n_ranks)));
    // local state + active list
    std::vector<unsigned> act;
    i64 active_total = 0;
    for (uint32_t s = 0; s < c->n_total; ++s) active_total += c->h_state[s] == 1;
    for (uint32_t s = 0; s < c->n_local; ++s)
        if (c->h_state[c->first + s] == 1) act.push_back(s);
    HIP_TRY(hipMemcpyAsync(c->d_state, c->h_state.data() + c->first, c->n_local, hipMemcpyHostToDevice, c->stream));
    if (!act.empty()) HIP_TRY(hipMemcpyAsync(c->d_act, act.data(), act.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_cnt, 0, (size_t)c->n_local * 8, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_afsum, 0, (size_t)c->n_local * 8, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_fscore, 0, (size_t)c->n_local * 8, c->stream));
    for (auto &ch : c->chunks) {
        HIP_TRY(hipMemsetAsync(ch.covered, 0, ch.wp * 8, c->stream));
        // samples that start out used cover their variants from the first iteration (select.py:36-39)
        for (uint32_t s = 0; s < c->n_local; ++s)
            if (c->h_state[c->first + s] == 0)
                hipLaunchKernelGGL(k_or_column, dim3(256), dim3(256), 0, c->stream, ch.covered, ch.cols + (u64)s * ch.wp, ch.wp);
    }
    HIP_TRY(hipGetLastError());
    IterState st;
    memset(&st, 0, sizeof st);
    st.n_active = (unsigned)act.size();
    st.n_active_total = active_total;
    st.prev_local = -1;
    st.xseq = c->xseq_host;  // the exchange sequence keeps counting across resets (every shard resets alike)
    HIP_TRY(hipMemcpyAsync(c->d_st, &st, sizeof st, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->decr_enabled) {
        if (c->listn_cap < c->chunks.size()) {
            (void)hipFree(c->d_listn);
            c->d_listn = nullptr;
            HIP_TRY(hipMalloc(&c->d_listn, c->chunks.size() * 4));
            c->listn_cap = c->chunks.size();
        }
        for (auto &ch : c->chunks)
            if (!ch.list_idx) {
                HIP_TRY(hipMalloc(&ch.list_idx, ch.wp * 4));
                HIP_TRY(hipMalloc(&ch.list_val, ch.wp * 8));
            }
        HIP_TRY(hipMemsetAsync(c->d_listn, 0, c->chunks.size() * 4, c->stream));
        TRY(ensure_interleaved(c));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->keep_valid = false;
    c->last_new = -1;
    c->decr_iterations = 0;
    c->brute_bytes = 0;
    c->decr_entries_seen = c->decr_gathers_seen = 0;
    c->iter = 0;
    c->captured_seen = 0;
    c->af_all_exact = false;
    c->scored = 0;
    c->active_ub = (unsigned)act.size();
    c->finished = false;
    c->score_launches = 0;
    c->score_ms = 0;
    c->algo_bytes = 0;
    c->ev_used = 0;
    c->prepared = true;
    return UTM_OK;
}

static int ensure_prepared(utm_ctx *c)
{
    if (c->prepared && !c->dirty_tables) return UTM_OK;
    return utm_reset(c);
}

static PickArgs pick_args(utm_ctx *c, bool decr = false)
{
    PickArgs a;
    a.st = c->d_st;
    a.act = c->d_act;
    a.state = c->d_state;
    a.weights = c->have_weights ? c->d_weights : nullptr;
    const bool afs = c->af_mode != UTM_AF_NONE && c->af_fixed;
    if (afs) {
        // AF (verified-parallel): the accumulators are persistent -- a full pass fills them once, later passes
        // subtract what the last winner newly covered (streamed delta pass, or the gather form when that is tiny)
        a.cnt = c->d_cnt;
        a.afsum = c->d_afsum;
        a.cnt_mirror = nullptr;
        a.afsum_mirror = nullptr;
        a.zero_after = 0;
    } else {
        a.cnt = decr ? c->d_cnt_keep : c->d_cnt;
        a.afsum = nullptr;
        // full iterations leave a copy of every count behind: the state decremental iterations continue from
        a.cnt_mirror = (!decr && c->decr_enabled) ? c->d_cnt_keep : nullptr;
        a.afsum_mirror = nullptr;
        a.zero_after = decr ? 0 : 1;
    }
    a.list_n = decr ? c->d_listn : nullptr;  // read for the accounting, then cleared, by k_pick
    a.n_chunks = (int)c->chunks.size();
    a.fscore = c->af_mode != UTM_AF_NONE ? c->d_fscore : nullptr;  // sequential scores (fallback / overflow)
    a.af_scale = ldexp(1.0, -c->af_q);
    // float32 AF sums only shrink: once every estimate was exact (< 2^53 units) the plain exact pick suffices
    a.mbox = c->d_mbox;
    a.peer_mbox = c->d_peer_mbox;
    a.cand = (c->af_mode != UTM_AF_NONE && c->af_fixed && !c->af_all_exact) ? c->d_cand : nullptr;
    a.af_is_f64 = c->af_mode == UTM_AF_F64;
    // (a shard's record is compared with other shards' records: there the score has to be exact)
    a.af_skip_single = (!c->af_exact_scores && c->n_local == c->n_total) ? 1 : 0;
    a.recs = reinterpret_cast<Rec *>(c->d_xbuf);
    a.slot_words = c->xbuf_slot_words;
    a.res_idx = c->d_res_idx;
    a.res_new = c->d_res_new;
    a.res_score = c->d_res_score;
    a.n_var_total = (i64)c->n_var_total;
    a.first = c->first;
    a.n_local = c->n_local;
    a.n_total = c->n_total;
    a.rank = c->rank;
    a.n_ranks = c->xbuf_ranks;
    return a;
}

// ---------------------------------------------------------------------------------------- launches
static int tune_env(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// Start / stop events of ONE scoring dispatch (UTM_FLAG_PROFILE_EVENTS): handed to hipExtLaunchKernelGGL, which
// stamps them from the dispatch itself -- the kernel's own duration, as a kernel trace reports it (events recorded
// around the launch add ~5 us of bracket to every measurement).  Null events = a plain launch.
struct LaunchTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    explicit LaunchTimer(utm_ctx *c)
    {
        c->score_launches += 1;
        if (!(c->flags & UTM_FLAG_PROFILE_EVENTS)) return;
        if (c->ev_used + 2 > c->ev.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            c->ev.push_back(a);
            c->ev.push_back(b);
        }
        start = c->ev[c->ev_used];
        stop = c->ev[c->ev_used + 1];
        c->ev_used += 2;
    }
};
#define UTM_TIMED_LAUNCH(timer, kernel, grid, block, ...) \
    hipExtLaunchKernelGGL(kernel, grid, block, 0, c->stream, (timer).start, (timer).stop, 0, __VA_ARGS__)

template <int STEPS>
static void launch_score_int(utm_ctx *c, const LaunchTimer &t, const Chunk &ch, unsigned blocks, unsigned group, unsigned n_groups,
                             bool nt)
{
    const u64 *cols = ch.cols;
    if (nt)
        UTM_TIMED_LAUNCH(t, (k_score_int<STEPS, true>), dim3(blocks), dim3(256), cols, ch.covered, ch.wp, pending_of(c, ch, true),
                         (const IterState *)c->d_st, (const unsigned *)c->d_act, c->d_cnt, group, n_groups);
    else
        UTM_TIMED_LAUNCH(t, (k_score_int<STEPS, false>), dim3(blocks), dim3(256), cols, ch.covered, ch.wp, pending_of(c, ch, true),
                         (const IterState *)c->d_st, (const unsigned *)c->d_act, c->d_cnt, group, n_groups);
}

static void launch_apply_pending(utm_ctx *c)
{
    for (auto &ch : c->chunks)
        hipLaunchKernelGGL(k_apply_pending, dim3((unsigned)std::min<u64>(1024, (ch.wp + 255) / 256)), dim3(256), 0, c->stream,
                           ch.covered, ch.cols, ch.wp, pending_of(c, ch, false), c->d_st);
}

// Sequential AF scoring of every selectable sample: covered is brought up to date first, then one lane per
// sample walks all chunks in order.
static void launch_score_sequential(utm_ctx *c, unsigned a_ub)
{
    launch_apply_pending(c);
    LaunchTimer t(c);
    const unsigned blocks = (a_ub + 63) / 64;
    if (c->af_mode == UTM_AF_F32)
        hipExtLaunchKernelGGL(k_score_seq<float>, dim3(blocks), dim3(64), 0, c->stream, t.start, t.stop, 0, c->d_seq,
                              (int)c->chunks.size(), c->d_st, c->d_act, c->d_cnt, c->d_fscore, 0);
    else
        hipExtLaunchKernelGGL(k_score_seq<double>, dim3(blocks), dim3(64), 0, c->stream, t.start, t.stop, 0, c->d_seq,
                              (int)c->chunks.size(), c->d_st, c->d_act, c->d_cnt, c->d_fscore, 0);
}

// AF, dense phase: LDS AF tiles.  Every workgroup re-stages its 32 KiB AF tile (from L2 / Infinity Cache), so
// the groups hold >= 64 samples.
static void launch_score_af_dense(utm_ctx *c, const Chunk &ch, unsigned a_ub)
{
    static const int af_target = tune_env("UTM_AF_TARGET_WGS", 16384);
    const u64 tiles = ch.wp / UTM_AF_TILE_WORDS;
    unsigned n_groups = (unsigned)std::max<u64>(1, std::min<u64>((a_ub + 63) / 64, (u64)af_target / std::max<u64>(1, tiles)));
    const unsigned group = ((a_ub + n_groups - 1) / n_groups + 15) / 16 * 16;
    n_groups = (a_ub + group - 1) / group;
    LaunchTimer t(c);
    hipExtLaunchKernelGGL(k_score_afq, dim3((unsigned)round_up(tiles * n_groups, 8)), dim3(256), 0, c->stream, t.start, t.stop, 0, ch.cols,
                          ch.covered, ch.wp, ch.af32, 150 - c->af_q, pending_of(c, ch, true), c->d_st, c->d_act, c->d_cnt, c->d_afsum,
                          group, n_groups);
}

// The streaming kernels (k_score_int, k_score_afs): grid = variant tiles x groups of samples.  Tile = the largest
// of {32 (AF: 16), 8, 2} KiB that still yields >= UTM_MIN_WGS workgroups; group size such that the grid has about
// UTM_TARGET_WGS workgroups (>> 256 CUs, small enough units for an even tail), at least one sample per wave.
static void launch_score_streaming(utm_ctx *c, const Chunk &ch, unsigned a_ub, bool delta = false)
{
    static const int target_wgs = tune_env("UTM_TARGET_WGS", 32768);
    static const int min_wgs = tune_env("UTM_MIN_WGS", 1024);
    static const int min_wgs_big = tune_env("UTM_MIN_WGS_BIG", 8192);  // the 32 KiB tile wants a deeper grid (chr22-sized: +17 % with 8 KiB)
    static const int force_steps = tune_env("UTM_TILE_STEPS", 0);
    static const int nt_env = tune_env("UTM_NT_LOADS", -1);
    static const int nt_min_mb = tune_env("UTM_NT_MIN_MB", 512);
    // non-temporal column loads when the matrix is a stream far larger than the 256 MB Infinity Cache (+10 % at
    // 3 GB); a matrix that (nearly) fits is better left to the caches (chr22-sized 345 MB: +5 %).  By the matrix, not
    // by the columns still selectable: the tail of a 3 GB select-all run measured slower with cached loads.
    const bool use_nt = nt_env >= 0 ? nt_env != 0 : (u64)c->n_local * c->col_words * 8 > ((u64)nt_min_mb << 20);
    static const int af_big = tune_env("UTM_AF_STEPS", 16) == 32 ? 32 : 16;
    const bool af = c->af_mode != UTM_AF_NONE;
    const u64 steps_total = ch.wp / UTM_STEP_WORDS;
    const u64 waves_needed = (a_ub + 3) / 4;  // workgroups if every wave had one sample
    int steps = 2;
    for (int cand : {af ? af_big : 32, 8}) {  // (the AF kernel shares LDS with its bit queues)
        const u64 tiles = (steps_total + cand - 1) / cand;
        if (tiles * waves_needed >= (u64)(cand > 8 ? min_wgs_big : min_wgs)) { steps = cand; break; }
    }
    if (!af && (force_steps == 32 || force_steps == 16 || force_steps == 8 || force_steps == 2)) steps = force_steps;
    const u64 tiles = (steps_total + steps - 1) / steps;
    u64 group = ((u64)a_ub * tiles + target_wgs - 1) / target_wgs;
    group = std::max<u64>(4, (group + 3) / 4 * 4);
    const unsigned n_groups = (unsigned)((a_ub + group - 1) / group);
    const unsigned blocks = (unsigned)round_up(tiles * n_groups, 8);  // XCD-aware map: tile_of_block()
    LaunchTimer t(c);
    if (af) {
        const unsigned *afb = reinterpret_cast<const unsigned *>(ch.af32);
        const int eb = 150 - c->af_q;
#define UTM_LAUNCH_AFS(S, Q)                                                                                              \
    hipExtLaunchKernelGGL((k_score_afs<S, Q>), dim3(blocks), dim3(256), 0, c->stream, t.start, t.stop, 0, ch.cols,       \
                          ch.covered, ch.wp, afb, eb, pending_of(c, ch, true), c->d_st, c->d_act, c->d_cnt, c->d_afsum,   \
                          (unsigned)group, n_groups, delta ? ch.mask : nullptr)
        if (steps == 32) UTM_LAUNCH_AFS(32, 8);  // second argument: queue depth per lane
        else if (steps == 16) UTM_LAUNCH_AFS(16, 16);
        else if (steps == 8) UTM_LAUNCH_AFS(8, 16);
        else UTM_LAUNCH_AFS(2, 16);
#undef UTM_LAUNCH_AFS
    } else if (steps == 32) launch_score_int<32>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt);
    else if (steps == 16) launch_score_int<16>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt);
    else if (steps == 8) launch_score_int<8>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt);
    else launch_score_int<2>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt);
}

// Enqueue the scoring of one iteration for every chunk (and the pending covered update).
static int enqueue_score(utm_ctx *c, bool force_sequential = false)
{
    const unsigned a_ub = std::max(1u, c->active_ub);
    if (c->af_mode != UTM_AF_NONE && (!c->af_fixed || force_sequential)) {
        launch_score_sequential(c, a_ub);
    } else if (c->af_mode != UTM_AF_NONE) {
        // AF, verified-parallel: persistent accumulators.  Without valid accumulators: clear them and run a full
        // pass (dense phase -> LDS-tile kernel, else the streaming kernel).  Otherwise a *delta* pass: the mask of
        // variants the last winner newly covered is made once (k_newly_mask, which also updates covered, from a
        // local or a peer-mapped column) and the streaming kernel subtracts those variants' share -- same bytes
        // streamed, but only the newly covered bits take the queue / gather path.
        if (!c->keep_valid) {
            HIP_TRY(hipMemsetAsync(c->d_cnt, 0, (size_t)c->n_local * 8, c->stream));
            HIP_TRY(hipMemsetAsync(c->d_afsum, 0, (size_t)c->n_local * 8, c->stream));
            const char *sw = getenv("UTM_AF_SWITCH");  // read per call: tests flip it
            const double af_switch = sw && *sw ? atof(sw) : 0.2;
            const bool af_dense = (double)c->captured_seen < af_switch * (double)c->n_var_total;
            if (c->p2p && !c->replicated) launch_apply_pending(c);
            for (auto &ch : c->chunks) {
                if (af_dense) launch_score_af_dense(c, ch, a_ub);
                else launch_score_streaming(c, ch, a_ub);
            }
            c->keep_valid = true;
        } else {
            for (auto &ch : c->chunks)
                hipLaunchKernelGGL(k_newly_mask, dim3((unsigned)std::min<u64>(2048, (ch.wp + 255) / 256)), dim3(256), 0, c->stream,
                                   ch.covered, ch.cols, ch.wp, pending_of(c, ch, false), c->d_st, ch.mask);
            for (auto &ch : c->chunks) launch_score_streaming(c, ch, a_ub, /*delta=*/true);
        }
    } else {
        if (c->p2p && !c->replicated) launch_apply_pending(c);  // remote column: read it once, not once per workgroup
        for (auto &ch : c->chunks) launch_score_streaming(c, ch, a_ub);
    }
    HIP_TRY(hipGetLastError());
    return UTM_OK;
}

// Decremental scoring of one iteration: list the words the pending winner newly covers, subtract.
static int enqueue_score_decr(utm_ctx *c)
{
    const unsigned a_ub = std::max(1u, c->active_ub);
    const bool af = c->af_mode != UTM_AF_NONE;  // (k_pick clears the list counters after reading them)
    unsigned split = (2048 + a_ub - 1) / a_ub;
    split = std::min(16u, std::max(1u, split));
    const u64 s_t = interleaved_stride(c);
    // interleaved form: list slices sized from the last known gain (gains shrink over a run; any value is correct)
    const unsigned slices = (unsigned)std::min<i64>(256, std::max<i64>(1, c->last_new / 16));
    for (size_t k = 0; k < c->chunks.size(); ++k) {
        Chunk &ch = c->chunks[k];
        hipLaunchKernelGGL(k_newly, dim3((unsigned)std::min<u64>(1024, (ch.wp / 2 + 511) / 512)), dim3(512), 0, c->stream, ch.covered,
                           ch.cols, ch.wp, pending_of(c, ch, false), c->d_st, ch.list_idx, ch.list_val, c->d_listn + k);
        const unsigned *afbits = af ? reinterpret_cast<const unsigned *>(ch.af32) : nullptr;
        const int e_base = af ? 150 - c->af_q : 0;
        u64 *cnt = af ? c->d_cnt : c->d_cnt_keep;
        i64 *afsum = af ? c->d_afsum : c->d_afsum_keep;
        if (c->decr_interleaved) {
            const dim3 grid((unsigned)((s_t + 255) / 256), slices);
            if (af)
                hipLaunchKernelGGL(k_decr_t<true>, grid, dim3(256), 0, c->stream, ch.rows_t, s_t, afbits, e_base, c->d_st, c->d_state,
                                   c->n_local, ch.list_idx, ch.list_val, c->d_listn + k, cnt, afsum);
            else
                hipLaunchKernelGGL(k_decr_t<false>, grid, dim3(256), 0, c->stream, ch.rows_t, s_t, afbits, e_base, c->d_st, c->d_state,
                                   c->n_local, ch.list_idx, ch.list_val, c->d_listn + k, cnt, afsum);
            continue;
        }
        const dim3 grid((a_ub + 3) / 4, split);
        if (af)
            hipLaunchKernelGGL(k_decr<true>, grid, dim3(256), 0, c->stream, ch.cols, ch.wp, afbits, e_base, c->d_st, c->d_act,
                               ch.list_idx, ch.list_val, c->d_listn + k, cnt, afsum);
        else
            hipLaunchKernelGGL(k_decr<false>, grid, dim3(256), 0, c->stream, ch.cols, ch.wp, afbits, e_base, c->d_st, c->d_act,
                               ch.list_idx, ch.list_val, c->d_listn + k, cnt, afsum);
    }
    HIP_TRY(hipGetLastError());
    return UTM_OK;
}

// Algorithmic HBM bytes of one iteration with `a` selectable local samples:
// active columns + covered read + winner column re-read + covered write (+ AF values).
static i64 iteration_bytes(const utm_ctx *c, u64 a)
{
    i64 b = 0;
    for (auto &ch : c->chunks) {
        b += (i64)((a + 3) * ch.w * 8);
        if (c->af_mode == UTM_AF_F32) b += (i64)ch.n_var * 4;
        if (c->af_mode == UTM_AF_F64) b += (i64)ch.n_var * 8;
    }
    return b;
}

// Verified-parallel AF: candidates -> their sequential chains (-> everyone, if too many tie).
static void enqueue_candidates(utm_ctx *c, const PickArgs &a)
{
    if (!a.cand) return;
    hipLaunchKernelGGL(k_cand, dim3(1), dim3(256), 0, c->stream, a);
    const unsigned blocks = (std::max(1u, c->active_ub) + 63) / 64;
    const ChainFast &cf = c->chain_fast;
    if (c->af_mode == UTM_AF_F32) {
        if (cf.counts)
            hipLaunchKernelGGL(k_chain_fill<float>, dim3(cf.n_segs, UTM_FAST_CAND), dim3(1024), 0, c->stream, c->d_seq, c->d_st, c->d_cand, cf);
        hipLaunchKernelGGL(k_chain<float>, dim3(UTM_MAX_CAND), dim3(1024), 0, c->stream, c->d_seq, (int)c->chunks.size(), c->d_st, c->d_cand, cf);
        hipLaunchKernelGGL(k_score_seq<float>, dim3(blocks), dim3(64), 0, c->stream, c->d_seq, (int)c->chunks.size(), c->d_st,
                           c->d_act, c->d_cnt, c->d_fscore, 1);
    } else {
        if (cf.counts)
            hipLaunchKernelGGL(k_chain_fill<double>, dim3(cf.n_segs, UTM_FAST_CAND), dim3(1024), 0, c->stream, c->d_seq, c->d_st, c->d_cand, cf);
        hipLaunchKernelGGL(k_chain<double>, dim3(UTM_MAX_CAND), dim3(1024), 0, c->stream, c->d_seq, (int)c->chunks.size(), c->d_st, c->d_cand, cf);
        hipLaunchKernelGGL(k_score_seq<double>, dim3(blocks), dim3(64), 0, c->stream, c->d_seq, (int)c->chunks.size(), c->d_st,
                           c->d_act, c->d_cnt, c->d_fscore, 1);
    }
}

static int enqueue_pick_and_exchange(utm_ctx *c, bool decr = false)
{
    PickArgs a = pick_args(c, decr);
    enqueue_candidates(c, a);
    if (c->n_ranks > 1 && c->mbox_ok) {
        // device-side exchange: post this shard's record into every shard's mailbox, wait for theirs, decide
        hipLaunchKernelGGL(k_pick<2>, dim3(1), dim3(1024), 0, c->stream, a);  // pick, post, collect, decide
    } else if (c->comm) {
        hipLaunchKernelGGL(k_pick<1>, dim3(1), dim3(1024), 0, c->stream, a);
        u64 *slot = c->d_xbuf + (u64)c->rank * c->xbuf_slot_words;
        if (!c->p2p)
            for (auto &ch : c->chunks)
                hipLaunchKernelGGL(k_pack, dim3((unsigned)std::min<u64>(1024, (ch.wp + 255) / 256)), dim3(256), 0, c->stream,
                                   slot + UTM_HDR_WORDS + ch.off, ch.cols, ch.wp, c->d_st, c->d_act);
        HIP_TRY(hipGetLastError());
        // one collective per iteration: every shard's record (and, without P2P mappings, its candidate column), in place
        NCCL_TRY(g_rccl.AllGather(slot, c->d_xbuf, c->xbuf_slot_words, ncclUint64, c->comm, c->stream));
        hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, a);
    } else if (c->n_ranks == 1) {
        hipLaunchKernelGGL(k_pick<0>, dim3(1), dim3(1024), 0, c->stream, a);
    } else {
        return fail(UTM_ESTATE, "sharded context without a fused exchange: use utm_local_best / utm_apply_records, or enable the mailboxes / RCCL");
    }
    HIP_TRY(hipGetLastError());
    return UTM_OK;
}

// Bring the host mirror up to date with the device after a sync.
static int sync_state(utm_ctx *c)
{
    HIP_TRY(hipMemcpyAsync(c->h_st, c->d_st, sizeof(IterState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->iter = c->h_st->iter;
    c->captured_seen = c->h_st->tot;
    c->xseq_host = c->h_st->xseq;
    if (c->h_st->xerror) return fail(UTM_ECOMM, "a shard's record did not arrive through the mailboxes in time");
    if (c->h_st->all_exact) c->af_all_exact = true;
    c->active_ub = c->h_st->n_active;
    c->finished = c->h_st->done != 0;
    return UTM_OK;
}

static int collect_event_times(utm_ctx *c)
{
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        c->score_ms += ms;
    }
    c->ev_used = 0;
    return UTM_OK;
}

extern "C" int utm_run(utm_ctx *c, int64_t k_max, int64_t *idx_out, int64_t *new_out, double *score_out,
                       int64_t *n_done)
{
    CTX(c);
    if (k_max < 0 || !n_done || (k_max > 0 && (!idx_out || !new_out))) return fail(UTM_EINVAL, "bad outputs");
    TRY(ensure_prepared(c));
    *n_done = 0;
    const i64 iter0 = c->iter;
    const i64 room = (i64)c->n_total - iter0;
    if (k_max > room) k_max = room;
    HIP_TRY(hipEventRecord(c->ev_loop0, c->stream));
    // iterations enqueued between two host syncs: AF modes latch host-side decisions there (64); the integer loop
    // only needs the stop flag (256: a boundary costs an idle device for two round trips)
    static const int batch_env = tune_env("UTM_BATCH", 0);
    const int batch = batch_env > 0 ? batch_env : c->af_mode != UTM_AF_NONE ? 64 : 256;
    i64 enq = 0;
    while (enq < k_max && !c->finished) {
        // AF runs start with short batches: the dense -> sparse kernel switch is taken at a batch boundary
        // ... and so is the switch to decremental iterations
        const i64 this_batch = (c->af_mode != UTM_AF_NONE && c->af_fixed && c->iter < 64) ? std::min<i64>(batch, 8)
                               : (c->decr_enabled && c->iter < 64)                          ? std::min<i64>(batch, 16)
                                                                                            : batch;
        const i64 n = std::min<i64>(this_batch, k_max - enq);
        const unsigned a0 = c->active_ub;
        // Decremental batches: only when allowed, when the persistent counts are current, and when the last
        // winner newly covered few enough variants (gains shrink over a greedy run, so it stays that way).
        const bool decr = c->decr_enabled && c->keep_valid && c->last_new >= 0 && (c->af_mode == UTM_AF_NONE || c->af_fixed) &&
                          (double)c->last_new <= (c->decr_threshold > 0 ? c->decr_threshold : c->decr_interleaved ? 0.5 : 0.2) *
                                                     (double)c->col_words;
        for (i64 j = 0; j < n; ++j) {
            if (decr) TRY(enqueue_score_decr(c));
            else TRY(enqueue_score(c));
            TRY(enqueue_pick_and_exchange(c, decr));
            if (c->n_ranks == 1 && c->active_ub > 0) c->active_ub -= 1;  // exact while the loop is alive
        }
        enq += n;
        const i64 before = c->iter;
        TRY(sync_state(c));
        // bytes: iterations that were actually scored in this batch (rows + a terminating empty pass); the
        // local selectable count falls from a0 to a1 over the batch's rows (by one per row on a single shard)
        const i64 rows = c->iter - before;
        const i64 passes = std::min<i64>(n, rows + ((c->finished && c->h_st->tot < (i64)c->n_var_total && rows < n) ? 1 : 0));
        const unsigned a1 = c->active_ub;
        for (i64 j = 0; j < passes; ++j) {
            const u64 drop = rows > 0 ? (u64)(a0 - a1) * (u64)std::min(j, rows) / (u64)rows : 0;
            const i64 full = iteration_bytes(c, a0 - drop);
            c->brute_bytes += full;
            if (!decr) c->algo_bytes += full;
        }
        if (decr) {
            // what the decremental iterations had to touch: winner column + covered (read), the list (written
            // once, read once), covered words rewritten, and one word per (selectable sample, listed word)
            const u64 entries = c->h_st->decr_entries - c->decr_entries_seen;
            const u64 gathers = c->h_st->decr_gathers - c->decr_gathers_seen;
            // (interleaved copy: one word per (sample slot, listed word), selectable or not)
            const u64 touched = c->decr_interleaved ? entries * interleaved_stride(c) : gathers;
            c->algo_bytes += (i64)(passes * 2 * (i64)c->col_words * 8 + entries * 32 + touched * 8);
            c->decr_iterations += passes;
        }
        c->decr_entries_seen = c->h_st->decr_entries;
        c->decr_gathers_seen = c->h_st->decr_gathers;
        // a full pass mirrored the counts (integer mode with the decremental option) / the AF accumulators are persistent
        c->keep_valid = c->decr_enabled || (c->af_mode != UTM_AF_NONE && c->af_fixed);
        if (rows > 0) HIP_TRY(copy_sync(c, &c->last_new, c->d_res_new + c->iter - 1, 8, hipMemcpyDeviceToHost));
        c->scored += passes;
        if (c->flags & UTM_FLAG_PROFILE_EVENTS) TRY(collect_event_times(c));
    }
    HIP_TRY(hipEventRecord(c->ev_loop1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_loop1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_loop0, c->ev_loop1));
    c->loop_ms = ms;
    const i64 rows = c->iter - iter0;
    if (rows > 0) {
        HIP_TRY(copy_sync(c, idx_out, c->d_res_idx + iter0, rows * 8, hipMemcpyDeviceToHost));
        HIP_TRY(copy_sync(c, new_out, c->d_res_new + iter0, rows * 8, hipMemcpyDeviceToHost));
        if (score_out) HIP_TRY(copy_sync(c, score_out, c->d_res_score + iter0, rows * 8, hipMemcpyDeviceToHost));
    }
    *n_done = rows;
    return UTM_OK;
}

extern "C" int utm_step(utm_ctx *c, int64_t *idx, int64_t *new_count, double *score)
{
    int64_t i = -1, n = 0, done = 0;
    double s = 0;
    CTX(c);
    TRY(ensure_prepared(c));
    if (!c->finished && c->iter < (i64)c->n_total) TRY(utm_run(c, 1, &i, &n, &s, &done));
    if (done == 0) { i = -1; n = 0; s = 0; }
    if (idx) *idx = i;
    if (new_count) *new_count = n;
    if (score) *score = s;
    return UTM_OK;
}

extern "C" int utm_peek_scores(utm_ctx *c, int64_t *counts, double *scores)
{
    CTX(c);
    TRY(ensure_prepared(c));
    // with AF every sample's exact reference score is wanted, so all of them take the sequential chain
    c->keep_valid = false;  // the pending winner gets applied here: the next iteration must re-score in full
    TRY(enqueue_score(c, /*force_sequential=*/true));
    i64 *d_counts = nullptr;
    double *d_scores = nullptr;
    HIP_TRY(hipMalloc(&d_counts, (size_t)c->n_local * 8));
    HIP_TRY(hipMalloc(&d_scores, (size_t)c->n_local * 8));
    PickArgs pa = pick_args(c);
    pa.afsum = nullptr;
    hipLaunchKernelGGL(k_final_scores, dim3((c->n_local + 255) / 256), dim3(256), 0, c->stream, pa, d_counts, d_scores);
    (void)hipMemsetAsync(c->d_cnt, 0, (size_t)c->n_local * 8, c->stream);
    (void)hipMemsetAsync(c->d_afsum, 0, (size_t)c->n_local * 8, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && counts) e = copy_sync(c, counts, d_counts, (size_t)c->n_local * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && scores) e = copy_sync(c, scores, d_scores, (size_t)c->n_local * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_counts);
    (void)hipFree(d_scores);
    if (e != hipSuccess) return fail(UTM_EHIP, "peek: %s", hipGetErrorString(e));
    return UTM_OK;
}

static int flush_pending(utm_ctx *c)
{
    for (auto &ch : c->chunks)
        hipLaunchKernelGGL(k_apply_pending, dim3((unsigned)std::min<u64>(1024, (ch.wp + 255) / 256)), dim3(256), 0, c->stream,
                           ch.covered, ch.cols, ch.wp, pending_of(c, ch, false), c->d_st);
    HIP_TRY(hipGetLastError());
    return UTM_OK;
}

extern "C" int utm_get_covered(utm_ctx *c, int32_t chunk, uint64_t *out)
{
    CTX(c);
    Chunk *ch;
    TRY(chunk_of(c, chunk, &ch));
    if (!out) return fail(UTM_EINVAL, "out is NULL");
    TRY(ensure_prepared(c));
    c->keep_valid = false;
    TRY(flush_pending(c));
    HIP_TRY(hipMemcpyAsync(out, ch->covered, ch->w * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UTM_OK;
}

extern "C" int utm_get_stats(utm_ctx *c, utm_stats *out)
{
    CTX(c);
    if (!out) return fail(UTM_EINVAL, "out is NULL");
    memset(out, 0, sizeof *out);
    out->iterations = c->iter;
    out->tot_captured = c->prepared ? c->h_st->tot : 0;
    out->score_launches = c->score_launches;
    out->score_ms = c->score_ms;
    out->loop_ms = c->loop_ms;
    out->algo_bytes = c->algo_bytes;
    out->af_mode = c->af_mode;
    out->af_fixed_point = c->af_fixed;
    out->af_q = c->af_q;
    out->n_chunks = (int32_t)c->chunks.size();
    out->decr_iterations = c->decr_iterations;
    out->brute_force_bytes = c->brute_bytes;
    out->p2p_replica_bytes = (i64)c->replica_bytes;
    out->decr_interleaved_bytes = c->decr_interleaved ? (i64)(c->col_words * interleaved_stride(c) * 8) : 0;
    return UTM_OK;
}

extern "C" int utm_set_decremental(utm_ctx *c, int32_t on, double threshold)
{
    CTX(c);
    c->decr_enabled = on != 0;
    c->decr_threshold = threshold > 0 ? threshold : 0;
    c->prepared = false;  // buffers are allocated at the next reset
    return UTM_OK;
}

extern "C" int utm_set_af_exact_scores(utm_ctx *c, int32_t on)
{
    CTX(c);
    c->af_exact_scores = on != 0;
    return UTM_OK;
}

extern "C" int utm_set_profile(utm_ctx *c, int32_t on)
{
    CTX(c);
    if (on) c->flags |= UTM_FLAG_PROFILE_EVENTS;
    else c->flags &= ~UTM_FLAG_PROFILE_EVENTS;
    return UTM_OK;
}

// ---------------------------------------------------------------------------------------- sharded building blocks
extern "C" int utm_column_words(utm_ctx *c, uint64_t *n_words)
{
    CTX(c);
    if (!n_words) return fail(UTM_EINVAL, "n_words is NULL");
    *n_words = c->col_words;
    return UTM_OK;
}

extern "C" int utm_local_best(utm_ctx *c, utm_record *rec)
{
    CTX(c);
    if (!rec) return fail(UTM_EINVAL, "rec is NULL");
    TRY(ensure_prepared(c));
    memset(rec, 0, sizeof *rec);
    rec->idx = -1;
    if (c->finished) return UTM_OK;
    c->keep_valid = false;
    TRY(enqueue_score(c));
    PickArgs a = pick_args(c);
    enqueue_candidates(c, a);
    hipLaunchKernelGGL(k_pick<1>, dim3(1), dim3(1024), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    const u64 slot = c->xbuf_slot_words;
    HIP_TRY(hipMemcpyAsync(rec, c->d_xbuf + (u64)c->rank * slot, sizeof *rec, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->algo_bytes += iteration_bytes(c, c->active_ub);
    c->scored += 1;
    return UTM_OK;
}

extern "C" int utm_get_column(utm_ctx *c, int64_t global_idx, uint64_t *out)
{
    CTX(c);
    if (!out || global_idx < (i64)c->first || global_idx >= (i64)c->first + c->n_local)
        return fail(UTM_EINVAL, "sample %lld is not local", (long long)global_idx);
    const u64 s = (u64)(global_idx - c->first);
    for (auto &ch : c->chunks)
        HIP_TRY(hipMemcpyAsync(out + ch.off, ch.cols + s * ch.wp, ch.wp * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UTM_OK;
}

extern "C" int utm_apply_records(utm_ctx *c, const utm_record *recs, int32_t n_ranks, const uint64_t *winner_col,
                                 int64_t *idx, int64_t *new_count, double *score)
{
    CTX(c);
    if (!recs || n_ranks < 1) return fail(UTM_EINVAL, "bad records");
    if (c->comm) return fail(UTM_ESTATE, "context exchanges through RCCL; use utm_step/utm_run");
    TRY(ensure_prepared(c));
    TRY(ensure_xbuf(c, n_ranks));
    const u64 slot = c->xbuf_slot_words;
    // who wins (same rule as decide()) -- only needed to place the winner's column in its slot
    int win = -1;
    for (int r = 0; r < n_ranks; ++r) {
        if (recs[r].idx < 0) continue;
        if (win < 0 || recs[r].score > recs[win].score || (recs[r].score == recs[win].score && recs[r].idx < recs[win].idx)) win = r;
    }
    for (int r = 0; r < n_ranks; ++r)
        HIP_TRY(hipMemcpyAsync(c->d_xbuf + (u64)r * slot, &recs[r], sizeof(utm_record), hipMemcpyHostToDevice, c->stream));
    if (win >= 0 && winner_col && !c->p2p) {
        if (n_ranks == 1) return fail(UTM_EINVAL, "winner_col given for a single shard");
        HIP_TRY(hipMemcpyAsync(c->d_xbuf + (u64)win * slot + UTM_HDR_WORDS, winner_col, c->col_words * 8, hipMemcpyHostToDevice, c->stream));
    } else if (win >= 0 && !c->p2p) {
        const i64 g = recs[win].idx;
        if (g < (i64)c->first || g >= (i64)c->first + c->n_local) return fail(UTM_EINVAL, "winner %lld is remote but winner_col is NULL", (long long)g);
    }
    const i64 before = c->iter;
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, pick_args(c));
    HIP_TRY(hipGetLastError());
    TRY(sync_state(c));
    int64_t i = -1, n = 0;
    double s = 0;
    if (c->iter > before) {
        HIP_TRY(copy_sync(c, &i, c->d_res_idx + before, 8, hipMemcpyDeviceToHost));
        HIP_TRY(copy_sync(c, &n, c->d_res_new + before, 8, hipMemcpyDeviceToHost));
        HIP_TRY(copy_sync(c, &s, c->d_res_score + before, 8, hipMemcpyDeviceToHost));
    }
    if (idx) *idx = i;
    if (new_count) *new_count = n;
    if (score) *score = s;
    return UTM_OK;
}

// ---------------------------------------------------------------------------------------- P2P column access
#define UTM_MAX_RANKS 64
struct P2PHeader {
    uint32_t first, n_local, n_chunks, has_mbox;
};
// blob = header, n_chunks column handles, one mailbox handle

extern "C" int utm_p2p_blob_bytes(utm_ctx *c, uint64_t *n_bytes)
{
    CTX(c);
    if (!n_bytes) return fail(UTM_EINVAL, "n_bytes is NULL");
    *n_bytes = sizeof(P2PHeader) + (c->chunks.size() + 1) * sizeof(hipIpcMemHandle_t);
    return UTM_OK;
}

extern "C" int utm_p2p_export(utm_ctx *c, void *blob)
{
    CTX(c);
    if (!blob) return fail(UTM_EINVAL, "blob is NULL");
    if (c->chunks.empty()) return fail(UTM_ESTATE, "no chunks");
    if (!c->d_mbox) {
        // record mailboxes: uncached device memory so that neither side's caches sit between a peer's store and our poll
        const size_t bytes = 2 * UTM_MAX_RANKS * sizeof(Mailbox);
        void *p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess &&
            hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess)
            p = nullptr;
        (void)hipGetLastError();
        if (p) {
            c->d_mbox = static_cast<Mailbox *>(p);
            HIP_TRY(hipMemsetAsync(c->d_mbox, 0, bytes, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    c->exported = true;
    P2PHeader hd{c->first, c->n_local, (uint32_t)c->chunks.size(), c->d_mbox ? 1u : 0u};
    memcpy(blob, &hd, sizeof hd);
    hipIpcMemHandle_t *hs = reinterpret_cast<hipIpcMemHandle_t *>(static_cast<char *>(blob) + sizeof hd);
    for (size_t k = 0; k < c->chunks.size(); ++k) HIP_TRY(hipIpcGetMemHandle(&hs[k], c->chunks[k].cols));
    memset(&hs[c->chunks.size()], 0, sizeof(hipIpcMemHandle_t));
    if (c->d_mbox && hipIpcGetMemHandle(&hs[c->chunks.size()], c->d_mbox) != hipSuccess) {
        (void)hipGetLastError();
        hd.has_mbox = 0;
        memcpy(blob, &hd, sizeof hd);
    }
    return UTM_OK;
}

static void p2p_close(utm_ctx *c)
{
    for (auto &ch : c->chunks) {
        for (void *p : ch.ipc_opened) (void)hipIpcCloseMemHandle(p);
        ch.ipc_opened.clear();
        (void)hipFree(ch.d_peer_cols);
        ch.d_peer_cols = nullptr;
        (void)hipFree(ch.replica);
        ch.replica = nullptr;
    }
    c->replicated = false;
    c->replica_bytes = 0;
    (void)hipFree(c->d_peer_first);
    c->d_peer_first = nullptr;
    for (void *p : c->mbox_opened) (void)hipIpcCloseMemHandle(p);
    c->mbox_opened.clear();
    (void)hipFree(c->d_peer_mbox);
    c->d_peer_mbox = nullptr;
    c->mbox_ok = false;
    c->p2p = false;
}

extern "C" int utm_p2p_import(utm_ctx *c, int32_t rank, int32_t n_ranks, const void *blobs)
{
    CTX(c);
    if (!blobs || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(UTM_EINVAL, "bad rank %d of %d", rank, n_ranks);
    p2p_close(c);
    if (n_ranks > UTM_MAX_RANKS) return fail(UTM_EINVAL, "at most %d shards", UTM_MAX_RANKS);
    const size_t blob = sizeof(P2PHeader) + (c->chunks.size() + 1) * sizeof(hipIpcMemHandle_t);
    std::vector<Mailbox *> boxes(n_ranks, nullptr);
    bool all_boxes = c->d_mbox != nullptr;
    std::vector<unsigned> firsts(n_ranks), locals(n_ranks);
    std::vector<std::vector<const u64 *>> table(c->chunks.size(), std::vector<const u64 *>(n_ranks, nullptr));
    for (int r = 0; r < n_ranks; ++r) {
        const char *b = static_cast<const char *>(blobs) + (size_t)r * blob;
        P2PHeader hd;
        memcpy(&hd, b, sizeof hd);
        if (hd.n_chunks != c->chunks.size()) { p2p_close(c); return fail(UTM_EINVAL, "rank %d has %u chunks, this one %zu", r, hd.n_chunks, c->chunks.size()); }
        firsts[r] = hd.first;
        locals[r] = hd.n_local;
        const hipIpcMemHandle_t *hs = reinterpret_cast<const hipIpcMemHandle_t *>(b + sizeof hd);
        for (size_t k = 0; k < c->chunks.size(); ++k) {
            if (r == rank) { table[k][r] = c->chunks[k].cols; continue; }
            void *p = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&p, hs[k], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { p2p_close(c); return fail(UTM_EHIP, "hipIpcOpenMemHandle(rank %d, chunk %zu) -> %s", r, k, hipGetErrorString(e)); }
            c->chunks[k].ipc_opened.push_back(p);
            table[k][r] = static_cast<const u64 *>(p);
        }
        if (r == rank) {
            boxes[r] = c->d_mbox;
        } else if (hd.has_mbox && all_boxes) {
            void *p = nullptr;
            if (hipIpcOpenMemHandle(&p, hs[c->chunks.size()], hipIpcMemLazyEnablePeerAccess) == hipSuccess) {
                c->mbox_opened.push_back(p);
                boxes[r] = static_cast<Mailbox *>(p);
            } else {
                (void)hipGetLastError();
                all_boxes = false;
            }
        } else {
            all_boxes = false;
        }
    }
    // Room permitting, copy the peers' columns over once (xGMI, the same system-scope reads the loop would do per
    // iteration) and resolve pending columns in local memory from then on: at 10M variants a winner's column is
    // 1.25 MB, ~20 us over one link, every iteration -- against a 3.1 GB one-time copy.  Columns are static after
    // the export; matrices that do not fit (cfg4: 78 GB per shard) keep the in-place reads, where the scan dominates.
    {
        u64 need = 0;
        for (int r = 0; r < n_ranks; ++r)
            if (r != rank) need += (u64)locals[r] * c->col_words * 8;
        size_t free_b = 0, total_b = 0;
        const char *env = getenv("UTM_P2P_REPLICATE");
        const bool wanted = n_ranks > 1 && !(env && *env == '0');
        if (wanted && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (u64)free_b > need + (8ull << 30)) {
            bool ok = true;
            for (size_t k = 0; k < c->chunks.size() && ok; ++k) {
                Chunk &ch = c->chunks[k];
                ok = hipMalloc(&ch.replica, (size_t)(need / c->col_words * ch.wp)) == hipSuccess;
                u64 off = 0;
                for (int r = 0; r < n_ranks && ok; ++r) {
                    if (r == rank) continue;
                    const u64 words = (u64)locals[r] * ch.wp;
                    hipLaunchKernelGGL(k_copy_remote, dim3(2048), dim3(256), 0, c->stream, table[k][r], ch.replica + off, words);
                    table[k][r] = ch.replica + off;
                    off += words;
                }
            }
            if (ok) ok = hipStreamSynchronize(c->stream) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                p2p_close(c);
                return fail(UTM_EHIP, "copying the peers' columns failed");
            }
            c->replicated = true;
            c->replica_bytes = need;
        }
    }
    HIP_TRY(hipMalloc(&c->d_peer_first, (size_t)n_ranks * 4));
    HIP_TRY(copy_sync(c, c->d_peer_first, firsts.data(), (size_t)n_ranks * 4, hipMemcpyHostToDevice));
    for (size_t k = 0; k < c->chunks.size(); ++k) {
        HIP_TRY(hipMalloc(&c->chunks[k].d_peer_cols, (size_t)n_ranks * sizeof(u64 *)));
        HIP_TRY(copy_sync(c, c->chunks[k].d_peer_cols, table[k].data(), (size_t)n_ranks * sizeof(u64 *), hipMemcpyHostToDevice));
    }
    if (all_boxes) {
        HIP_TRY(hipMalloc(&c->d_peer_mbox, (size_t)n_ranks * sizeof(Mailbox *)));
        HIP_TRY(copy_sync(c, c->d_peer_mbox, boxes.data(), (size_t)n_ranks * sizeof(Mailbox *), hipMemcpyHostToDevice));
    }
    c->p2p = true;
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->prepared = false;  // exchange slots shrink to records
    return UTM_OK;
}

// One full post + wait round through the mailboxes, four times.  Collective: every shard calls it.  *ok = this
// shard received every peer's test record in time.  The caller combines the shards' answers and, if all are 1,
// switches the fused loop to the mailboxes with utm_p2p_use_mailboxes.
extern "C" int utm_p2p_selftest(utm_ctx *c, int32_t *ok)
{
    CTX(c);
    if (!ok) return fail(UTM_EINVAL, "ok is NULL");
    *ok = 0;
    if (!c->p2p || !c->d_peer_mbox) return UTM_OK;  // nothing to test: answer "no"
    int *d_ok = nullptr;
    HIP_TRY(hipMalloc(&d_ok, 4));
    int one = 1;
    HIP_TRY(copy_sync(c, d_ok, &one, 4, hipMemcpyHostToDevice));
    for (int round = 0; round < 4; ++round) {
        c->xseq_host += 1;
        hipLaunchKernelGGL(k_mbox_ping, dim3(1), dim3(64), 0, c->stream, c->d_mbox, c->d_peer_mbox, c->rank, c->n_ranks,
                           c->xseq_host, d_ok);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    int got = 0;
    if (e == hipSuccess) e = copy_sync(c, &got, d_ok, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_ok);
    if (e != hipSuccess) return fail(UTM_EHIP, "mailbox self-test: %s", hipGetErrorString(e));
    *ok = got;
    c->prepared = false;  // the loop state carries the exchange sequence number
    return UTM_OK;
}

extern "C" int utm_p2p_use_mailboxes(utm_ctx *c, int32_t on)
{
    CTX(c);
    if (on && (!c->p2p || !c->d_peer_mbox)) return fail(UTM_ESTATE, "mailboxes are not mapped");
    c->mbox_ok = on != 0;
    c->prepared = false;
    return UTM_OK;
}

// ---------------------------------------------------------------------------------------- RCCL
extern "C" int utm_comm_get_unique_id(void *id)
{
    if (!id) return fail(UTM_EINVAL, "id is NULL");
    TRY(rccl_load());
    static_assert(sizeof(ncclUniqueId) == UTM_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return UTM_OK;
}

extern "C" int utm_comm_init(utm_ctx *c, int32_t rank, int32_t n_ranks, const void *id)
{
    CTX(c);
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(UTM_EINVAL, "bad rank %d of %d", rank, n_ranks);
    if (c->comm) return fail(UTM_ESTATE, "communicator already initialised");
    if (c->chunks.empty()) return fail(UTM_ESTATE, "add the chunks before utm_comm_init (the exchange buffer is sized from them)");
    TRY(rccl_load());
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    NCCL_TRY(g_rccl.CommInitRank(&c->comm, n_ranks, u, rank));
    if (c->p2p && (c->rank != rank || c->n_ranks != n_ranks))
        return fail(UTM_EINVAL, "P2P mappings were imported as rank %d of %d", c->rank, c->n_ranks);
    c->rank = rank;
    c->n_ranks = n_ranks;
    // true only if `mine` is true on every rank (collective)
    auto everywhere = [&](bool mine, bool *all) -> int {
        double bad = mine ? 0.0 : 1.0;
        TRY(utm_comm_allreduce_max(c, &bad));
        *all = bad < 0.5;
        return UTM_OK;
    };
    // Unless the caller already did it (utm_p2p_import), map every rank's columns and record mailboxes (hipIpc):
    // a winner's column is then read in place over xGMI and the records travel through the mailboxes.  Every step
    // is agreed on by all ranks; whatever cannot be set up everywhere is left to RCCL.
    if (n_ranks > 1 && !c->p2p && !tune_env("UTM_NO_P2P", 0)) {
        uint64_t blob = 0;
        TRY(utm_p2p_blob_bytes(c, &blob));
        std::vector<char> mine(blob), all(blob * n_ranks);
        bool ok = utm_p2p_export(c, mine.data()) == UTM_OK;
        char *d_all = nullptr;
        HIP_TRY(hipMalloc(&d_all, blob * n_ranks));
        HIP_TRY(copy_sync(c, d_all + blob * rank, mine.data(), blob, hipMemcpyHostToDevice));
        NCCL_TRY(g_rccl.AllGather(d_all + blob * rank, d_all, blob, ncclChar, c->comm, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(copy_sync(c, all.data(), d_all, blob * n_ranks, hipMemcpyDeviceToHost));
        (void)hipFree(d_all);
        if (ok) ok = utm_p2p_import(c, rank, n_ranks, all.data()) == UTM_OK;
        bool mapped = false;
        TRY(everywhere(ok, &mapped));
        if (!mapped && c->p2p) p2p_close(c);
        if (mapped && !tune_env("UTM_NO_MAILBOX", 0)) {
            int32_t box_ok = 0;
            TRY(utm_p2p_selftest(c, &box_ok));  // can this rank see every peer's mailbox stores?
            bool boxes = false;
            TRY(everywhere(box_ok != 0, &boxes));
            if (boxes) TRY(utm_p2p_use_mailboxes(c, 1));
        }
    }
    TRY(ensure_xbuf(c, n_ranks));  // c->comm is set: slots carry whole columns unless P2P is on
    c->prepared = false;
    return UTM_OK;
}

extern "C" int utm_comm_allreduce_max(utm_ctx *c, double *value)
{
    CTX(c);
    if (!value) return fail(UTM_EINVAL, "value is NULL");
    if (!c->comm) return UTM_OK;  // single shard: identity
    double *d = nullptr;
    HIP_TRY(hipMalloc(&d, 8));
    HIP_TRY(hipMemcpyAsync(d, value, 8, hipMemcpyHostToDevice, c->stream));
    ncclResult_t r = g_rccl.AllReduce(d, d, 1, ncclDouble, ncclMax, c->comm, c->stream);
    hipError_t e = hipMemcpyAsync(value, d, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (r != ncclSuccess) return fail(UTM_ECOMM, "ncclAllReduce -> %s", g_rccl.GetErrorString(r));
    if (e != hipSuccess) return fail(UTM_EHIP, "allreduce copy: %s", hipGetErrorString(e));
    return UTM_OK;
}
