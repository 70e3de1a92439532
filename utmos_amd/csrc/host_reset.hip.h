// Loop set-up: exchange buffers, AF device tables, the interleaved copy, utm_reset.
// Part of the one translation unit utmos_hip.hip (included there, in order); not a stand-alone header.
#pragma once

// ---------------------------------------------------------------------------------------- loop set-up
static int ensure_xbuf(utm_ctx *c, int n_ranks)
{
    if (c->d_xbuf && c->xbuf_ranks == n_ranks) return UTM_OK;
    (void)hipFree(c->d_xbuf);
    c->d_xbuf = nullptr;
    HIP_TRY(hipMalloc(&c->d_xbuf, (size_t)n_ranks * UTM_HDR_WORDS * 8));
    HIP_TRY(hipMemsetAsync(c->d_xbuf, 0, (size_t)n_ranks * UTM_HDR_WORDS * 8, c->stream));  // same stream as every later use
    c->xbuf_ranks = n_ranks;
    return UTM_OK;
}

// Room for one whole column delivered by the exchange (RCCL broadcast / utm_apply_records' winner_col).
static int ensure_wincol(utm_ctx *c)
{
    if (c->d_wincol && c->wincol_words == c->col_words) return UTM_OK;
    (void)hipFree(c->d_wincol);
    c->d_wincol = nullptr;
    c->wincol_words = 0;
    HIP_TRY(hipMalloc(&c->d_wincol, (size_t)c->col_words * 8));
    HIP_TRY(hipMemsetAsync(c->d_wincol, 0, (size_t)c->col_words * 8, c->stream));
    c->wincol_words = c->col_words;
    return UTM_OK;
}

// Decremental mode: the word-interleaved second copy of every chunk (decremental.hip.h), all chunks or none, only
// when it fits next to a reserve of free HBM.  UTM_DECR_INTERLEAVED=0 keeps the gather form (what a context
// without the room runs).
static u64 interleaved_stride(const utm_ctx *c) { return round_up((u64)c->n_local, 64); }

static int ensure_interleaved(utm_ctx *c)
{
    const bool wanted = c->tune.decr_interleaved != 0;
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

// One whole column from its owner to every shard's winner-column buffer: ncclBroadcast, one call per chunk in
// one group (the owner sends straight from its matrix).  local_col = the column's local index on the owner, else -1.
static int broadcast_column(utm_ctx *c, int owner, i64 local_col)
{
    TRY(ensure_wincol(c));
    NCCL_TRY(g_rccl.GroupStart());
    for (auto &ch : c->chunks) {
        const u64 *send = owner == c->rank ? ch.cols + (u64)local_col * ch.wp : c->d_wincol + ch.off;
        ncclResult_t r = g_rccl.Broadcast(send, c->d_wincol + ch.off, ch.wp, ncclUint64, owner, c->comm, c->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail(UTM_ECOMM, "ncclBroadcast(column from rank %d) -> %s", owner, g_rccl.GetErrorString(r));
        }
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return UTM_OK;
}

extern "C" int utm_reset(utm_ctx *c)
{
    CTX(c);
    if (c->chunks.empty()) return fail(UTM_ESTATE, "no chunks");
    read_tune(&c->tune);  // the one place a live context picks up changed environment knobs
    c->remote_winner_test = c->tune.test_remote_winner == 1;
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
    HIP_TRY(hipMemsetAsync(c->d_cnt_alt, 0, (size_t)c->n_local * 8, c->stream));  // (a launch that ended on an error may have left partial counts)
    for (auto *w : c->d_loop_w)
        if (w) HIP_TRY(hipMemsetAsync(w, 0, (size_t)c->n_local * 8, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_afsum, 0, (size_t)c->n_local * 8, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_fscore, 0, (size_t)c->n_local * 8, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_arrivals, 0, 128, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_vsync, 0, sizeof(VerifySync), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_known_cnt, 0xFF, (size_t)c->n_local * 8, c->stream));  // (a run that ended on an error may have left counts behind)
    for (auto &ch : c->chunks) HIP_TRY(hipMemsetAsync(ch.covered, 0, ch.wp * 8, c->stream));
    // Samples that start out used cover their variants from the first iteration (select.py:36-39) -- on every
    // shard's covered replica, whoever owns the column: local columns are OR-ed in place; a peer's column is read
    // through its mapping (P2P) or broadcast by its owner (RCCL; every shard walks the same list, so the calls
    // match); a context with neither cannot know a remote column and says so.
    for (uint32_t g = 0; g < c->n_total; ++g) {
        if (c->h_state[g] != 0) continue;
        if (g >= c->first && g < c->first + c->n_local) {
            for (auto &ch : c->chunks)
                hipLaunchKernelGGL(k_or_column, dim3(256), dim3(256), 0, c->stream, ch.covered, ch.cols + (u64)(g - c->first) * ch.wp, ch.wp);
            if (!c->comm || c->p2p || c->n_ranks == 1) continue;
        }
        if (c->n_local == c->n_total) continue;
        int owner = -1;
        for (size_t r = 0; r < c->rank_first.size(); ++r)
            if (g >= c->rank_first[r] && g < c->rank_first[r] + c->rank_local[r]) owner = (int)r;
        if (c->p2p && owner >= 0) {
            if (owner == c->rank) continue;
            for (auto &ch : c->chunks)
                hipLaunchKernelGGL(k_or_column_remote, dim3(256), dim3(256), 0, c->stream, ch.covered,
                                   ch.h_peer_cols[owner] + (u64)(g - c->rank_first[owner]) * ch.wp, ch.wp);
        } else if (c->comm && owner >= 0) {
            TRY(broadcast_column(c, owner, owner == c->rank ? (i64)(g - c->first) : -1));
            if (owner != c->rank)
                for (auto &ch : c->chunks)
                    hipLaunchKernelGGL(k_or_column, dim3(256), dim3(256), 0, c->stream, ch.covered, c->d_wincol + ch.off, ch.wp);
        } else {
            return fail(UTM_ESTATE, "sample %u starts out used but belongs to another shard: map the shards' columns (utm_p2p_import) or "
                                    "initialise RCCL (utm_comm_init) first, so that its column can reach this shard's covered mask", g);
        }
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
    c->defer_lo = 0;
    c->deferred_rows = 0;
    c->scored = 0;
    c->active_ub = (unsigned)act.size();
    c->finished = false;
    c->score_launches = 0;
    c->persist_launches = c->persist_iterations = c->persist_unresolved = 0;
    c->af_table_passes = 0;
    c->loop_unresolved = false;
    c->persist_backoff = c->persist_backoff_len = 0;
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
        a.cnt_by_pos = 0;
    } else {
        a.cnt = decr ? c->d_cnt_keep : c->d_cnt;
        a.afsum = nullptr;
        // full iterations leave a copy of every count behind: the state decremental iterations continue from
        a.cnt_mirror = (!decr && c->decr_enabled) ? c->d_cnt_keep : nullptr;
        a.afsum_mirror = nullptr;
        a.zero_after = decr ? 0 : 1;
        // integer full passes count by position in act[] (k_score_int's by_pos); the persistent copy is by sample
        a.cnt_by_pos = (!decr && c->af_mode == UTM_AF_NONE) ? 1 : 0;
    }
    a.list_n = decr ? c->d_listn : nullptr;  // read for the accounting, then cleared, by k_pick
    a.n_chunks = (int)c->chunks.size();
    a.fscore = c->af_mode != UTM_AF_NONE ? c->d_fscore : nullptr;  // sequential scores (fallback / overflow)
    a.af_scale = ldexp(1.0, -c->af_q);
    // float32 AF sums only shrink: once every estimate was exact (< 2^53 units) the plain exact pick suffices
    a.mbox = c->mbox_local;
    a.peer_mbox = c->d_peer_mbox;
    a.cand = (c->af_mode != UTM_AF_NONE && c->af_fixed && !c->af_all_exact) ? c->d_cand : nullptr;
    a.af_is_f64 = c->af_mode == UTM_AF_F64;
    a.af_trunc = c->af_trunc ? 1 : 0;
    // (a shard's record is compared with other shards' records: there the score has to be exact)
    // ... and so it is for the caller unless told otherwise -- but on the only shard a lone candidate's exact sum can
    // come later, from the deferred launches
    a.af_skip_single = ((!c->af_exact_scores || defer_active(c)) && c->n_local == c->n_total) ? 1 : 0;
    a.early_pick = 0;  // (enqueue_candidates decides)
    const int record_env = c->tune.af_record;
    a.known_cnt = (a.cand && record_env) ? c->d_known_cnt : nullptr;
    a.known_val = c->d_known_val;
    a.recs = reinterpret_cast<Rec *>(c->d_xbuf);
    a.remote_winner_test = c->remote_winner_test ? 1 : 0;
    a.mbox_spins = 1u << std::min(28, std::max(4, c->tune.mbox_spins_log2));
    a.test_mute = c->tune.test_mute_exchange;
    a.test_drop = (c->tune.test_drop_arrival > 0 && c->score_launches == c->tune.test_drop_arrival) ? 1 : 0;  // (the launch being enqueued)
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
