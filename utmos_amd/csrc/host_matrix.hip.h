// Matrix construction: chunks, uploads (column / packed-row), var_count, synthetic fill.
// Part of the one translation unit utmos_hip.hip (included there, in order); not a stand-alone header.
#pragma once

// ---------------------------------------------------------------------------------------- matrix
extern "C" int utm_add_chunk(utm_ctx *c, uint64_t n_var, int32_t *chunk)
{
    CTX(c);
    if (n_var == 0) return fail(UTM_EINVAL, "empty chunk");
    // a count word keeps bits 40.. for the tile-arrival count of a fused launch (k_score_int): counts stay below 2^40
    if (c->n_var_total + n_var >= (1ull << 40)) return fail(UTM_EINVAL, "more than 2^40 variants in one matrix");
    if (c->exported) return fail(UTM_ESTATE, "chunks must be added before the columns are exported to other shards");
    Chunk ch;
    ch.n_var = n_var;
    ch.w = (n_var + 63) / 64;
    ch.wp = round_up(ch.w, UTM_STEP_WORDS);
    ch.off = c->col_words;
    // (+ 64 KiB of zeros behind the last column: the persistent loop reads whole 8 KiB batches, and the last batch of a
    // column's last tile may run past the column's end -- into the next column, or here; those words are counted against
    // zero words of the covered tile)
    const size_t bytes = (size_t)c->n_local * ch.wp * 8 + UTM_COLS_SLACK_BYTES;
    HIP_TRY(hipMalloc(&ch.cols, bytes));
    HIP_TRY(hipMemsetAsync(ch.cols, 0, bytes, c->stream));
    HIP_TRY(hipMalloc(&ch.covered, ch.wp * 8));
    HIP_TRY(hipMemsetAsync(ch.covered, 0, ch.wp * 8, c->stream));
    ch.index = (int)c->chunks.size();
    c->chunks.push_back(std::move(ch));
    c->n_var_total += n_var;
    c->col_words += c->chunks.back().wp;
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
    Scratch<unsigned char> d_rows;
    HIP_TRY(d_rows.alloc(std::min(slab_rows, round_up(n_rows, 64)) * row_stride_bytes));
    int rc = UTM_OK;
    for (u64 r0 = 0; r0 < n_rows && rc == UTM_OK; r0 += slab_rows) {
        const u64 nr = std::min(slab_rows, n_rows - r0);
        hipError_t e = hipMemcpyAsync(d_rows.p, rows + r0 * row_stride_bytes, nr * row_stride_bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = fail(UTM_EHIP, "row upload: %s", hipGetErrorString(e)); break; }
        dim3 grid((unsigned)((nr + 63) / 64), (c->n_local + 63) / 64);
        hipLaunchKernelGGL(k_transpose_rows, grid, dim3(64), 0, c->stream, d_rows.p, (u64)row_stride_bytes, nr,
                           (first_var + r0) / 64, ch->cols, ch->wp, c->first, c->n_local, c->n_total);
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(UTM_EHIP, "row transpose: %s", hipGetErrorString(e));
    }
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
