// Sharding over GPUs: host-staged building blocks, hipIpc column mappings + record mailboxes, RCCL.
// Part of the one translation unit utmos_hip.hip (included there, in order); not a stand-alone header.
#pragma once

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
    enqueue_candidates(c, a, false);
    hipLaunchKernelGGL(k_pick<1>, dim3(1), dim3(1024), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rec, c->d_xbuf + (u64)c->rank * UTM_HDR_WORDS, sizeof *rec, hipMemcpyDeviceToHost, c->stream));
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
    // who wins (same rule as decide()) -- only needed to know whether the winner's column has to come with the call
    int win = -1;
    for (int r = 0; r < n_ranks; ++r) {
        if (recs[r].idx < 0) continue;
        if (win < 0 || recs[r].score > recs[win].score || (recs[r].score == recs[win].score && recs[r].idx < recs[win].idx)) win = r;
    }
    HIP_TRY(hipMemcpyAsync(c->d_xbuf, recs, (size_t)n_ranks * sizeof(utm_record), hipMemcpyHostToDevice, c->stream));
    if (win >= 0 && !c->p2p) {
        const i64 g = recs[win].idx;
        const bool local = g >= (i64)c->first && g < (i64)c->first + c->n_local;
        if (winner_col && n_ranks == 1) return fail(UTM_EINVAL, "winner_col given for a single shard");
        if (!local && !winner_col) return fail(UTM_EINVAL, "winner %lld is remote but winner_col is NULL", (long long)g);
        if (!local) {
            TRY(ensure_wincol(c));
            HIP_TRY(hipMemcpyAsync(c->d_wincol, winner_col, c->col_words * 8, hipMemcpyHostToDevice, c->stream));
        }
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
        ch.h_peer_cols.clear();
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
            c->mbox_local = c->d_mbox;
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
        const bool wanted = n_ranks > 1 && c->tune.p2p_replicate != 0;
        if (n_ranks == 1 && c->tune.p2p_replicate != 0) c->replicated = true;  // (no peers: every column is local already)
        if (wanted && hipMemGetInfo(&free_b, &total_b) == hipSuccess && (u64)free_b > need + (8ull << 30)) {
            // allocate everything first: a refusal (another process took the room meanwhile) just keeps the in-place reads
            bool room = true;
            for (auto &ch : c->chunks)
                if (room && hipMalloc(&ch.replica, (size_t)(need / c->col_words * ch.wp)) != hipSuccess) room = false;
            if (!room) {
                (void)hipGetLastError();
                for (auto &ch : c->chunks) {
                    (void)hipFree(ch.replica);
                    ch.replica = nullptr;
                }
            } else {
                for (size_t k = 0; k < c->chunks.size(); ++k) {
                    Chunk &ch = c->chunks[k];
                    u64 off = 0;
                    for (int r = 0; r < n_ranks; ++r) {
                        if (r == rank) continue;
                        const u64 words = (u64)locals[r] * ch.wp;
                        hipLaunchKernelGGL(k_copy_remote, dim3(2048), dim3(256), 0, c->stream, table[k][r], ch.replica + off, words);
                        table[k][r] = ch.replica + off;
                        off += words;
                    }
                }
                if (hipStreamSynchronize(c->stream) != hipSuccess) {  // a fault reading a peer: no P2P at all on this shard
                    (void)hipGetLastError();
                    p2p_close(c);
                    return fail(UTM_EHIP, "copying the peers' columns failed");
                }
                c->replicated = true;
                c->replica_bytes = need;
            }
        }
    }
    HIP_TRY(hipMalloc(&c->d_peer_first, (size_t)n_ranks * 4));
    HIP_TRY(copy_sync(c, c->d_peer_first, firsts.data(), (size_t)n_ranks * 4, hipMemcpyHostToDevice));
    for (size_t k = 0; k < c->chunks.size(); ++k) {
        HIP_TRY(hipMalloc(&c->chunks[k].d_peer_cols, (size_t)n_ranks * sizeof(u64 *)));
        HIP_TRY(copy_sync(c, c->chunks[k].d_peer_cols, table[k].data(), (size_t)n_ranks * sizeof(u64 *), hipMemcpyHostToDevice));
        c->chunks[k].h_peer_cols = table[k];
    }
    c->rank_first = firsts;
    c->rank_local = locals;
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
    Scratch<int> d_ok;
    HIP_TRY(d_ok.alloc(1));
    int one = 1;
    HIP_TRY(copy_sync(c, d_ok.p, &one, 4, hipMemcpyHostToDevice));
    for (int round = 0; round < 4; ++round) {
        c->xseq_host += 1;
        hipLaunchKernelGGL(k_mbox_ping, dim3(1), dim3(64), 0, c->stream, c->mbox_local, c->d_peer_mbox, c->rank, c->n_ranks,
                           c->xseq_host, d_ok.p);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    int got = 0;
    if (e == hipSuccess) e = copy_sync(c, &got, d_ok.p, 4, hipMemcpyDeviceToHost);
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
    c->mbox_single = on == 2;  // (a single shard posts to and collects from itself: what the exchange costs, on one GPU)
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
    if (n_ranks > UTM_MAX_RANKS) return fail(UTM_EINVAL, "at most %d shards", UTM_MAX_RANKS);
    if (c->comm) return fail(UTM_ESTATE, "communicator already initialised");
    if (c->chunks.empty()) return fail(UTM_ESTATE, "add the chunks before utm_comm_init (the winner-column buffer is sized from them)");
    if (c->p2p && (c->rank != rank || c->n_ranks != n_ranks))
        return fail(UTM_EINVAL, "P2P mappings were imported as rank %d of %d", c->rank, c->n_ranks);
    TRY(rccl_load());
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    {
        ncclResult_t r = g_rccl.CommInitRank(&c->comm, n_ranks, u, rank);
        if (r != ncclSuccess) {
            c->comm = nullptr;
            return fail(UTM_ECOMM, "ncclCommInitRank(rank %d of %d) -> %s", rank, n_ranks, g_rccl.GetErrorString(r));
        }
    }
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->exported = true;  // the peers' covered replicas follow these columns: they must not change any more
    // every shard's sample range (who owns a global index: the root of a column broadcast)
    Scratch<unsigned> d_ranges;
    HIP_TRY(d_ranges.alloc(2 * (size_t)n_ranks));
    const unsigned mine[2] = {c->first, c->n_local};
    HIP_TRY(copy_sync(c, d_ranges.p + 2 * rank, mine, 8, hipMemcpyHostToDevice));
    NCCL_TRY(g_rccl.AllGather(d_ranges.p + 2 * rank, d_ranges.p, 2, ncclUint32, c->comm, c->stream));
    std::vector<unsigned> all(2 * (size_t)n_ranks);
    HIP_TRY(copy_sync(c, all.data(), d_ranges.p, (size_t)n_ranks * 8, hipMemcpyDeviceToHost));
    c->rank_first.assign(n_ranks, 0);
    c->rank_local.assign(n_ranks, 0);
    u64 covered_samples = 0;
    for (int k = 0; k < n_ranks; ++k) {
        c->rank_first[k] = all[2 * k];
        c->rank_local[k] = all[2 * k + 1];
        covered_samples += all[2 * k + 1];
    }
    if (covered_samples != c->n_total) return fail(UTM_EINVAL, "the shards' ranges do not add up to %u samples", c->n_total);
    TRY(ensure_xbuf(c, n_ranks));
    TRY(ensure_wincol(c));
    c->prepared = false;
    return UTM_OK;
}

extern "C" int utm_comm_column_by_allreduce(utm_ctx *c, int32_t on)
{
    CTX(c);
    if (!c->comm) return fail(UTM_ESTATE, "no communicator (utm_comm_init)");
    if (on && !c->d_stage) {
        std::vector<StageChunk> table;
        for (auto &ch : c->chunks) table.push_back(StageChunk{ch.cols, ch.wp, ch.off});
        HIP_TRY(hipMalloc(&c->d_stage, table.size() * sizeof(StageChunk)));
        HIP_TRY(copy_sync(c, c->d_stage, table.data(), table.size() * sizeof(StageChunk), hipMemcpyHostToDevice));
    }
    c->column_by_allreduce = on != 0;
    c->prepared = false;
    return UTM_OK;
}

extern "C" int utm_comm_allreduce_max(utm_ctx *c, double *value)
{
    CTX(c);
    if (!value) return fail(UTM_EINVAL, "value is NULL");
    if (!c->comm) return UTM_OK;  // single shard: identity
    Scratch<double> d;
    HIP_TRY(d.alloc(1));
    HIP_TRY(hipMemcpyAsync(d.p, value, 8, hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(g_rccl.AllReduce(d.p, d.p, 1, ncclDouble, ncclMax, c->comm, c->stream));
    HIP_TRY(copy_sync(c, value, d.p, 8, hipMemcpyDeviceToHost));
    return UTM_OK;
}
