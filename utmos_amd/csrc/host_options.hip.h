// Per-run options: sample states, weights, AF tables (and their exactness analysis), decremental / profile switches.
// Part of the one translation unit utmos_hip.hip (included there, in order); not a stand-alone header.
#pragma once

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
        Scratch<u64> d_keep;
        HIP_TRY(d_keep.alloc(ch->w));
        HIP_TRY(copy_sync(c, d_keep.p, keep.data(), ch->w * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mask_rows, dim3(4096), dim3(256), 0, c->stream, ch->cols, ch->wp, d_keep.p, ch->w, c->n_local);
        hipError_t e = hipStreamSynchronize(c->stream);
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
        (void)hipFree(ch.afx);
        ch.af = nullptr;
        ch.af32 = nullptr;
        ch.afx = nullptr;
    }
    (void)hipFree(c->d_seq);
    (void)hipFree(c->d_seq_alt);
    c->d_seq = c->d_seq_alt = nullptr;
    c->af_fixed = false;
    c->af_table_ok = false;
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
    // Every float32 a > 0 is m * 2^(e-23), m < 2^24: a multiple of 2^-q for q = 23 - e_min.  The estimate kernels
    // sum floor(a * 2^q) as int64.  Lossless q when the whole table's mass fits (mass * 2^q < 2^62) and the largest
    // value's mantissa can be shifted into place (q <= 61 - e_max): true for real allele frequencies (>= 1/2S).
    // Otherwise the largest q that fits: small values lose their low bits (< 1 unit per addend), which the interval
    // arithmetic of the verification accounts for -- slower candidates, same rows, and no cliff to the sequential
    // kernel for an unusual table.
    c->af_trunc = false;
    int q = 23 - e_min;
    if (mass > 0) {
        int q_mass = 61;
        while (q_mass >= 0 && !(mass * ldexpl(1.0L, q_mass - 62) < 1.0L)) --q_mass;
        const int q_fit = std::min(q_mass, 61 - e_max);
        if (q > q_fit) {
            q = q_fit;
            c->af_trunc = true;
        }
    }
    if (!(c->flags & UTM_FLAG_AF_SEQUENTIAL) && representable && q >= 0 && q <= 149) {
        c->af_fixed = true;
        c->af_q = q;
    } else {
        c->af_trunc = false;
    }
    std::vector<SeqChunk> seq, seq_alt;
    for (size_t k = 0; k < c->chunks.size(); ++k) {
        Chunk &ch = c->chunks[k];
        const size_t n = ch.wp * 64;
        if (c->af_fixed && !ch.covered_alt) HIP_TRY(hipMalloc(&ch.covered_alt, ch.wp * 8));
        if (c->af_fixed) {
            // the estimate's table: floor(a * 2^q) as mantissa << shift (af_fixed(), score_af.hip.h)
            std::vector<unsigned> fx(n, 0u);
            const int e_base = 150 - c->af_q;
            if (k == 0) c->af_table_ok = true;
            for (u64 v = 0; v < ch.n_var; ++v) {
                unsigned bits;
                memcpy(&bits, &v32[k][v], 4);
                if ((bits & 0x7FFFFFFFu) == 0) continue;  // AF == 0: its row is cleared anyway
                const unsigned m = (bits & 0x7FFFFFu) | 0x800000u;
                const int sh = (int)(bits >> 23) - e_base;
                fx[v] = sh >= 0 ? ((unsigned)sh << 24) | m : (sh > -24 ? m >> -sh : 0u);
                if (sh > 21) c->af_table_ok = false;  // (k_score_aft's limbs: every value below 2^45)
            }
            HIP_TRY(hipMalloc(&ch.afx, n * 4));
            HIP_TRY(copy_sync(c, ch.afx, fx.data(), n * 4, hipMemcpyHostToDevice));
        }
        if (c->af_mode == UTM_AF_F32) {
            HIP_TRY(hipMalloc(&ch.af32, n * 4));
            HIP_TRY(copy_sync(c, ch.af32, v32[k].data(), n * 4, hipMemcpyHostToDevice));
            ch.af = ch.af32;
        } else {
            HIP_TRY(hipMalloc(&ch.af, n * 8));
            // (the context's stream is non-blocking: never mix in null-stream work, it would not be ordered with it)
            HIP_TRY(hipMemsetAsync(ch.af, 0, n * 8, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(copy_sync(c, ch.af, ch.h_af64.data(), ch.n_var * 8, hipMemcpyHostToDevice));
        }
        seq.push_back(SeqChunk{ch.cols, ch.covered, ch.af, ch.wp, ch.w});
        seq_alt.push_back(SeqChunk{ch.cols, ch.covered_alt ? ch.covered_alt : ch.covered, ch.af, ch.wp, ch.w});
    }
    HIP_TRY(hipMalloc(&c->d_seq, seq.size() * sizeof(SeqChunk)));
    HIP_TRY(copy_sync(c, c->d_seq, seq.data(), seq.size() * sizeof(SeqChunk), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&c->d_seq_alt, seq_alt.size() * sizeof(SeqChunk)));
    HIP_TRY(copy_sync(c, c->d_seq_alt, seq_alt.data(), seq_alt.size() * sizeof(SeqChunk), hipMemcpyHostToDevice));
    // segment table + buffers of the chains' fast path
    (void)hipFree(c->d_segs); (void)hipFree(c->chain_fast.counts); (void)hipFree(c->chain_fast.vals);
    c->d_segs = nullptr;
    c->chain_fast = ChainFast{nullptr, 0, nullptr, nullptr, 0, 0};
    if (c->af_fixed)
        for (auto &ch : c->chunks)
            if (!ch.mask) HIP_TRY(hipMalloc(&ch.mask, ch.wp * 8));
    if (c->af_fixed) {
        std::vector<ChainSeg> segs;
        for (size_t k = 0; k < c->chunks.size(); ++k)
            for (u64 w0 = 0; w0 < c->chunks[k].w; w0 += UTM_SEG_WORDS) segs.push_back(ChainSeg{(int)k, w0, c->chunks[k].off + w0});
        const size_t n = segs.size();
        // regions for the candidates' compacted addends: room for EVERY bit of a segment where the memory allows
        // (all candidates, then two), so that the parallel chain always applies; else 1024 values per segment and
        // denser segments leave their candidate to the one-workgroup chain
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const size_t budget = std::min<size_t>(free_b / 4, 16ull << 30);
        unsigned cap = 0;
        int n_cand = 0;
        if (n * UTM_FAST_CAND * (size_t)UTM_SEG_FULL * 8 <= budget) { cap = UTM_SEG_FULL; n_cand = UTM_FAST_CAND; }
        else if (n * 2 * (size_t)UTM_SEG_FULL * 8 <= budget) { cap = UTM_SEG_FULL; n_cand = 2; }
        else if (n * UTM_FAST_CAND * (size_t)UTM_SEG_CAP * 8 <= (4ull << 30)) { cap = UTM_SEG_CAP; n_cand = UTM_FAST_CAND; }
        if (cap) {
            HIP_TRY(hipMalloc(&c->d_segs, n * sizeof(ChainSeg)));
            HIP_TRY(copy_sync(c, c->d_segs, segs.data(), n * sizeof(ChainSeg), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(&c->chain_fast.counts, n * n_cand * 4));
            HIP_TRY(hipMalloc(&c->chain_fast.vals, n * n_cand * (size_t)cap * 8));
            c->chain_fast.segs = c->d_segs;
            c->chain_fast.n_segs = (int)n;
            c->chain_fast.seg_cap = cap;
            c->chain_fast.n_cand = n_cand;
        }
        // deferred exact scores (af_defer.hip.h): the mask log, one addend slot per variant, (row, segment) tables --
        // only where the loop can use them (the only shard) and the memory is to spare; without them every winner is
        // chained on the spot, as before
        (void)hipFree(c->d_newly_log); (void)hipFree(c->d_defer_counts); (void)hipFree(c->d_defer_offs); (void)hipFree(c->d_defer_vals);
        c->d_newly_log = nullptr; c->d_defer_counts = nullptr; c->d_defer_offs = nullptr; c->d_defer_vals = nullptr;
        const int defer_env = c->tune.af_defer;
        u64 slots = 0;  // addend slots: every word that holds variants
        for (auto &ch : c->chunks) slots += ch.w * 64;
        // (a whole persistent batch of rows where the matrix is one the persistent loop takes, else the launches' 64)
        const bool small = c->chunks.size() == 1 && c->tune.persist_max_mb > 0 && (u64)c->n_local * c->col_words * 8 <= ((u64)c->tune.persist_max_mb << 20);
        c->defer_slots = small ? UTM_DEFER_SLOTS : UTM_DEFER_SLOTS_LAUNCHES;
        const size_t need = (size_t)c->defer_slots * c->col_words * 8 + slots * 8 + (size_t)c->defer_slots * n * 12 + 8;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if (defer_env && cap && c->d_segs && c->n_local == c->n_total && slots < 0xFFFF0000ull && need <= free_b / 4) {
            const bool ok = hipMalloc(&c->d_newly_log, (size_t)c->defer_slots * c->col_words * 8) == hipSuccess &&
                            hipMalloc(&c->d_defer_vals, slots * 8) == hipSuccess &&
                            hipMalloc(&c->d_defer_counts, (size_t)c->defer_slots * n * 4) == hipSuccess &&
                            hipMalloc(&c->d_defer_offs, ((size_t)c->defer_slots * n + 1) * 8) == hipSuccess;
            if (!ok) {  // all four or none (defer_active looks at the log alone)
                (void)hipGetLastError();
                (void)hipFree(c->d_newly_log); (void)hipFree(c->d_defer_counts); (void)hipFree(c->d_defer_offs); (void)hipFree(c->d_defer_vals);
                c->d_newly_log = nullptr; c->d_defer_counts = nullptr; c->d_defer_offs = nullptr; c->d_defer_vals = nullptr;
            }
        }
    }
    c->dirty_tables = false;
    return UTM_OK;
}
