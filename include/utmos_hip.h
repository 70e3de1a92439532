/* utmos_hip.h -- C ABI of libutmos_hip.so: greedy maximum-coverage selection on MI355X (gfx950).
 *
 * The reference (ACEnglish/utmos v2.2.0) has no FFI/plugin interface; its only seam for this
 * path is the Python call chain
 *
 *     run_selection   utmos/select.py:147-195
 *       greedy_select utmos/select.py:69-137   (generator; mutates sample_mask in place, :100)
 *         calculate_scores utmos/select.py:24-53  -> (idx, new_count) | (None, None)
 *
 * so the entry points below are what a ctypes binding inside utmos/select.py would call in place
 * of calculate_scores / greedy_select (INTEGRATION.md shows that binding).  Plain pointers and
 * sizes only; no Python, numpy or torch types.
 *
 * Conventions
 *   - every function returns 0 (UTM_OK) or a negative UTM_E* code; utm_last_error() gives the text
 *     of the last failure on the calling thread.
 *   - host buffers are borrowed for the duration of the call; device memory belongs to the context.
 *   - one context = one GPU = one contiguous shard [first_sample, first_sample + n_local) of the
 *     sample axis.  A context is not thread-safe; it owns one HIP stream.
 *   - the matrix is a list of *chunks* along the variant axis (the replacement for the reference's
 *     row-chunked hdf5 store, select.py:198-231).  Inside a chunk every sample is one column:
 *     W = ceil(n_var/64) little-endian uint64 words, variant v of the chunk = bit (v & 63) of word
 *     (v >> 6), zero padded.
 *   - sample states follow select.py:169-179: 1 selectable, 0 already used (covers), 2 excluded.  A used sample
 *     covers on EVERY shard: utm_reset fetches a remote one through the P2P mapping or an ncclBroadcast, and
 *     fails with UTM_ESTATE on a shard that has neither.
 *
 * Environment.  None of these changes a result; they move launch shapes and thresholds (tools/tune.py sweeps them)
 * or are test hooks.  ONE table in utmos_amd/csrc/utmos_hip.hip (g_knobs) names them all with their defaults; a context
 * reads them at utm_ctx_create and again at every utm_reset (read_tune) and nowhere else, so nothing on the
 * per-iteration path touches the environment.  The first context of a process lists on stderr the ones that are set;
 * utm_env_overrides() returns the same list (bench.py reports it as `env_overrides`).
 *   UTM_TARGET_WGS (32768)    workgroups a scoring launch aims for          UTM_MIN_WGS (128), UTM_MIN_WGS_BIG (8192)
 *   UTM_TILE_STEPS (auto)     force the covered tile to 32/16/8/4/2 KiB     smallest grids the 8 KiB / 32 KiB tiles are used for
 *   UTM_NT_LOADS (auto)       non-temporal column loads on/off              UTM_NT_MIN_MB (512) matrix size from which they are used
 *   UTM_FUSE_PICK (1)         pick inside the scoring launch (one shard or mailbox exchange, integer scores)
 *   UTM_CHAIN_PICK (1)        AF with candidate chains: pick inside the chain launch (one shard)
 *   UTM_AF_VERIFY (1)         ... and candidates, compaction, chains and pick as stages of ONE launch (one shard)
 *   UTM_AF_RECORD (1)         AF: a chain's float64 sum stays on record per sample for as long as the sample's count does not change
 *   UTM_AF_DEFER (1)          AF, one shard: exact float64 scores of unambiguous winners are finished per batch from a
 *                             log of newly-covered masks (needs 64 columns + 8 bytes per variant of HBM; 0: chained on the spot)
 *   UTM_PICK_THREADS (auto)   threads of the stand-alone k_pick             UTM_BATCH (256; AF 64) iterations between host syncs
 *   UTM_AF_STEPS (16), UTM_AF_SWITCH (0.2), UTM_AF_TARGET_WGS (16384)       AF kernels: tile, dense->streaming switch, grid
 *   UTM_AF_TABLES (1)         the full dense AF pass as table lookups (k_score_aft) where every fixed-point value is below 2^45; 0: k_score_afq
 *   UTM_AF_TABLE_WGS_PER_CU (8), UTM_AF_TABLE_RUN (0 = by the grid)         ... its grid: workgroups per CU, or tiles per workgroup outright
 *   UTM_AF_DENSE_DELTA (0.05) AF delta passes take the LDS-tile kernel while the last winner newly covered more than this share of all variants
 *   UTM_DECR_FIRST_BATCH (8), UTM_DECR_INTERLEAVED (1)                      decremental mode: first batch size, second copy on/off
 *   UTM_P2P_REPLICATE (1)     copy the peers' columns once (0: read winners in place over the mappings)
 *   UTM_TEST_REMOTE_WINNER (0) test hook: read local winners from the exchange's winner-column buffer too
 *   UTM_PERSISTENT (1)        integer scores (weighted or not) and the verified-parallel AF forms, one chunk, the only shard: a
 *                             batch of iterations as ONE persistent
 *                             launch (k_loop_int: workers keep their covered tile in LDS, the picker's record replaces the kernel
 *                             boundary, two batches per wave run ahead across the hand-off)
 *   UTM_PERSIST_MAX_MB (560), UTM_PERSIST_MAX_SAMPLES (2560), UTM_PERSIST_MAX_TILES (32)   largest matrix / sample count run that
 *                             way, and the tile count the tile size (8 / 16 / 32 / 64 KiB) is chosen for; UTM_PERSIST_TILE_KIB forces one;
 *                             tiles above 8 KiB (columns taller than 32 tiles) only up to UTM_PERSIST_TALL_MAX_SAMPLES (640) samples
 *   UTM_PERSIST_WGS_PER_CU (0 = what the occupancy query allows)   resident 512-thread blocks per CU the grid is sized for
 *   UTM_PERSIST_AHEAD_TICKS (400), UTM_PERSIST_AHEAD0_TICKS (0)    10 ns ticks before the next record is due at which a wave requests
 *                             its second / first run-ahead batch (0: as soon as it runs out of work)
 *   UTM_PERSIST_AF (1)        AF scores: the delta iterations run inside the persistent loop too -- float32 AF in its exact fixed-point
 *                             phase, and (UTM_PERSIST_AF_INTERVAL, 1) float64 AF / float32 sums outside the exact range with score intervals:
 *                             the picker lists the candidates, UTM_PERSIST_CHAINERS (4, 1..8) blocks at the end of the grid keep their
 *                             sequential float64 sums on record, working ahead while iterations take longer than UTM_PERSIST_SPEC_TICKS
 *                             (1000 x 10 ns; -1: only on request; -2: test hook, the chainers leave and every request times out).
 *                             Both AF forms: 8 KiB tiles only, at most UTM_PERSIST_AF_MAX_TILES (26)
 *   UTM_PERSIST_CLAIMS (1)    positions behind a wave's two static ones are claimed from per-tile counters (0: dealt statically)
 *   UTM_MBOX_SPINS_LOG2 (24)  mailbox exchange: polls (x ~0.5 us) before a peer's record is declared lost (UTM_ECOMM)
 *   UTM_TEST_MUTE_EXCHANGE (0) test hook: in that iteration since the reset this shard posts no record into the mailboxes
 *   UTM_TEST_DROP_ARRIVAL (0) test hook: in that scoring launch / persistent iteration since the reset one partial count is
 *                             withheld, so that the pick's bounded wait runs out (UTM_EHIP; the context works again after utm_reset)
 */
#ifndef UTMOS_HIP_H
#define UTMOS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UTM_OK 0
#define UTM_EINVAL (-1)   /* bad argument */
#define UTM_EHIP (-2)     /* HIP runtime error */
#define UTM_ENOMEM (-3)   /* allocation failed */
#define UTM_ESTATE (-4)   /* call not valid in the context's current state */
#define UTM_ECOMM (-5)    /* RCCL error / RCCL not available */

#define UTM_ABI_VERSION 3 /* 2: utm_stats.exchange / rccl_ranks; host-memory mailbox and replica entry points removed
                           * 3: utm_env_overrides; utm_stats.persist_launches / persist_iterations; utm_stream_calibration */

typedef struct utm_ctx utm_ctx;

/* AF value modes (utm_set_af).  F32 restates the reference's hdf5 store (float32 presence*AF,
 * select.py:218-223); F64 its in-memory matrix (float64, select.py:314-321). */
#define UTM_AF_NONE 0
#define UTM_AF_F32 1
#define UTM_AF_F64 2

/* utm_ctx_create flags */
#define UTM_FLAG_PROFILE_EVENTS 1u /* bracket every scoring launch with HIP events (utm_stats.score_ms) */
#define UTM_FLAG_AF_SEQUENTIAL 2u  /* never use the order-independent fixed-point form for F32 AF */
#define UTM_FLAG_DECREMENTAL 4u    /* allow decremental scoring (utm_set_decremental) */

/* One local-best record, as exchanged between shards (64 bytes). */
typedef struct utm_record {
    double score;      /* final (masked, weighted) score of the shard's best selectable sample */
    int64_t idx;       /* its GLOBAL sample index, or -1 when the shard has no selectable sample */
    int64_t new_count; /* variants it would newly capture (counts[use], select.py:49) */
    int64_t pad[5];
} utm_record;

typedef struct utm_stats {
    int64_t iterations;      /* greedy iterations completed since the last utm_reset */
    int64_t tot_captured;    /* running total of new_count */
    int64_t score_launches;  /* scoring-kernel launches since the last utm_reset */
    double score_ms;         /* summed HIP-event time of those launches (UTM_FLAG_PROFILE_EVENTS), else 0 */
    double loop_ms;          /* HIP-event time of the last utm_run (stream time, first launch to last) */
    int64_t algo_bytes;      /* algorithmic HBM bytes of the iterations since the last utm_reset */
    int32_t af_mode;         /* UTM_AF_* in effect */
    int32_t af_fixed_point;  /* AF estimate: 1 = int64 fixed point at a unit no AF value loses a bit to, 2 = coarser unit
                              * (mass would overflow: addends floored, intervals widened), 0 = sequential kernel */
    int32_t af_q;            /* fixed-point scale: scores = sum / 2^q */
    int32_t n_chunks;
    int64_t decr_iterations;   /* iterations scored decrementally (0 unless enabled) */
    int64_t brute_force_bytes; /* what full re-scoring of every iteration would have had to read */
    int64_t p2p_replica_bytes; /* HBM holding copies of the other shards' columns (utm_p2p_import; 0: read in place) */
    int64_t decr_interleaved_bytes; /* HBM held by the word-interleaved copy the decremental iterations stream (0: gather form) */
    int32_t exchange;        /* UTM_EXCHANGE_*: how utm_run's iterations meet the other shards */
    int32_t rccl_ranks;      /* ncclCommCount of the context's communicator, 0 without one */
    /* allele-frequency scoring, verified-parallel form, since the last utm_reset: */
    int64_t af_chained_iterations; /* iterations whose pick needed sequential float64 chains on the spot (near-ties) */
    int64_t af_deferred_rows;      /* rows whose exact float64 score was finished after their batch (one launch set per batch) */
    /* persistent loop (one chunk, the only shard; integer scores and the verified-parallel AF forms): batches of iterations run as ONE launch */
    int64_t persist_launches;      /* such launches since the last utm_reset (each counts once in score_launches) */
    int64_t persist_iterations;    /* rows they produced (0: every iteration was a launch of its own) */
    int64_t persist_unresolved;    /* launches (interval form of the AF loop) that left their last iteration to a verification launch */
    int64_t af_table_passes;       /* full dense AF passes (one per chunk) that ran as table lookups (k_score_aft) since the last utm_reset */
} utm_stats;

/* utm_stats.exchange */
#define UTM_EXCHANGE_NONE 0    /* the context holds every sample: nothing to exchange */
#define UTM_EXCHANGE_MAILBOX 1 /* 64-byte records through hipIpc-mapped device mailboxes; winner column read from a peer mapping / local copy */
#define UTM_EXCHANGE_RCCL 2    /* ncclAllGather of the records, then ncclBroadcast of the winner's column from its owner */
#define UTM_EXCHANGE_CALLER 3  /* a shard without a device-side exchange: the caller drives utm_local_best / utm_apply_records */
#define UTM_EXCHANGE_RCCL_SUM 4 /* RCCL, root-free: ncclAllGather of the records, then ncclAllReduce(sum) of a buffer that holds the winner's column on its owner and zeros elsewhere (utm_comm_column_by_allreduce) */

const char *utm_last_error(void);
int utm_abi_version(void);
int utm_device_count(int *n);
/* The UTM_* environment knobs that are set right now, as "NAME=value NAME=value" (empty string: none), into buf[cap].
 * A context reads the knobs when it is created and again at every utm_reset -- never inside the loop. */
int utm_env_overrides(char *buf, uint64_t cap);
/* Free and total HBM of a device in bytes (hipMemGetInfo): what the host's is_memsafe policy (select.py:56-63) asks. */
int utm_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- context ------------------------------------------------------------------------------ */
int utm_ctx_create(int device, uint32_t n_samp_total, uint32_t first_sample, uint32_t n_samp_local,
                   uint32_t flags, utm_ctx **out);
int utm_ctx_destroy(utm_ctx *ctx);

/* ---- matrix (select.py:275-321 builds the reference's; here it is built in HBM) -------------- */
/* Append an all-zero chunk of n_var variants; returns its index in *chunk. */
int utm_add_chunk(utm_ctx *ctx, uint64_t n_var, int32_t *chunk);
/* Upload n_cols whole columns of a chunk, local column first_col onwards.
 * cols[c * stride_words + w], stride_words >= ceil(n_var/64). */
int utm_upload_columns(utm_ctx *ctx, int32_t chunk, uint32_t first_col, uint32_t n_cols,
                       const uint64_t *cols, uint64_t stride_words);
/* Upload n_rows variants in the reference's own packing (numpy.packbits(axis=1), MSB first,
 * convert.py:85): rows[r * row_stride_bytes + (s >> 3)] bit (7 - (s & 7)) = GLOBAL sample s.
 * The bit transpose to column-major runs on the GPU.  first_var must be a multiple of 64. */
int utm_upload_rows_packed(utm_ctx *ctx, int32_t chunk, uint64_t first_var, uint64_t n_rows,
                           const uint8_t *rows, uint64_t row_stride_bytes);
int utm_download_columns(utm_ctx *ctx, int32_t chunk, uint32_t first_col, uint32_t n_cols,
                         uint64_t *cols, uint64_t stride_words);
/* Per-sample carrier totals over all chunks (var_count, select.py:281-284), n_samp_local values. */
int utm_var_count(utm_ctx *ctx, int64_t *out);
/* Seeded synthetic chunk contents generated on the GPU (bench/test input, DESIGN.md "Synthetic input"). */
int utm_synth_fill(utm_ctx *ctx, int32_t chunk, uint64_t seed, uint64_t first_var_global);
/* The same generator on the host (no GPU call): columns of samples [first, first+n) and/or per-variant AF. */
int utm_synth_host(uint64_t seed, uint64_t first_var_global, uint64_t n_var, uint32_t n_samp_total,
                   uint32_t first_sample, uint32_t n_samp, uint64_t *cols, uint64_t stride_words,
                   float *af_out);

/* ---- scoring options ------------------------------------------------------------------------ */
/* state[n_samp_total]: 1 selectable / 0 used / 2 excluded.  Implies utm_reset. */
int utm_set_sample_state(utm_ctx *ctx, const uint8_t *state);
/* weights[n_samp_total] (select.py:181-187) or NULL for none.  Must be finite. */
int utm_set_weights(utm_ctx *ctx, const double *weights);
/* Per-variant AF of one chunk: mode UTM_AF_F32 (const float*) or UTM_AF_F64 (const double*), n_var
 * values, finite and >= 0.  All chunks must use the same mode.  UTM_AF_NONE with af == NULL clears. */
int utm_set_af(utm_ctx *ctx, int32_t chunk, int mode, const void *af);

/* ---- the greedy loop ------------------------------------------------------------------------ */
/* Restart: covered := OR of the columns of all used (state 0) samples, local or not; tot_captured := 0. */
int utm_reset(utm_ctx *ctx);
/* One iteration = calculate_scores (select.py:24-53) + the winner update (select.py:99-100).
 * *idx = global sample index or -1 for the reference's (None, None). */
int utm_step(utm_ctx *ctx, int64_t *idx, int64_t *new_count, double *score);
/* Up to k_max iterations, device resident; stops like greedy_select (select.py:93-96, :110-112).
 * Outputs need k_max entries (score_out may be NULL); *n_done = rows produced. */
int utm_run(utm_ctx *ctx, int64_t k_max, int64_t *idx_out, int64_t *new_out, double *score_out,
            int64_t *n_done);
/* Scores of the NEXT iteration without selecting: counts[n_local] (new variants per local sample,
 * 0 for non-selectable) and final scores[n_local].  Either may be NULL.  Parity/debug aid. */
int utm_peek_scores(utm_ctx *ctx, int64_t *counts, double *scores);
/* AF modes in fixed point (utm_stats.af_fixed_point != 0): the same, from the PARALLEL full pass the loop's first iteration
 * runs (k_score_aft / k_score_afq / k_score_afs) instead of the sequential chains -- scores[s] = int64 sum * 2^-q (* weight),
 * which IS the reference's float64 sum wherever the sums are exact (float32 AF, lossless unit, below 2^53 units).
 * UTM_ESTATE otherwise.  Parity/debug aid: what the tests compare the first pass with. */
int utm_peek_estimates(utm_ctx *ctx, int64_t *counts, double *scores);
/* Current covered mask of a chunk (all pending updates applied), ceil(n_var/64) words. */
int utm_get_covered(utm_ctx *ctx, int32_t chunk, uint64_t *out);
int utm_get_stats(utm_ctx *ctx, utm_stats *out);
/* Decremental scoring: after a full scoring pass, later iterations only subtract the
 * contribution of the variants the last winner newly covered (counts and fixed-point AF sums stay exact;
 * same rows).  Reads far fewer bytes than the brute-force loop, so it is off by default and its numbers
 * are reported separately from the brute-force roofline.  When free HBM allows, the context keeps a
 * second, word-interleaved copy of the matrix (rows_t[word][sample]) for these iterations: a newly
 * covered word is then one contiguous read over all samples instead of one 8-byte gather per column
 * (utm_stats.decr_interleaved_bytes; UTM_DECR_INTERLEAVED=0 or lack of room keeps the gather form).
 * threshold = largest fraction of a column's words that may be newly covered for an iteration to go
 * decremental (<= 0: default -- 1.0 with the interleaved copy, 0.2 gathering). */
int utm_set_decremental(utm_ctx *ctx, int32_t on, double threshold);
/* AF modes: the winner of an iteration is always the reference's (sample order, new_count are exact).  The reported
 * *score* is, by default (on = 1), also the reference's float64 running sum bit for bit, which costs one sequential
 * chain per iteration whenever the parallel sum is not provably exact.  on = 0 skips that chain when the winner is
 * unambiguous and reports the parallel estimate (relative error <= 2^-24 + n 2^-53) -- what a caller that only
 * writes the reference's TSV columns needs.  Ambiguous iterations (ties within the error bound) are always chained,
 * and so is everything on a context that holds only a shard of the samples (its records meet other shards'). */
int utm_set_af_exact_scores(utm_ctx *ctx, int32_t on);
/* Streaming-read calibration: `launches` plain read-only passes over the context's resident columns with the scoring
 * kernel's access shape (16 B per lane non-temporal loads, 1 KiB per wave instruction, 8 in flight, ~32k workgroups) and
 * nothing else -- *gbps_out = bytes read / stream time.  What the box delivers today, beside the 8 TB/s spec peak: the bench
 * reports it as roofline.stream_calibration_gbps so that box-to-box variance can be told from kernel quality. */
int utm_stream_calibration(utm_ctx *ctx, int32_t launches, double *gbps_out);
/* Switch per-launch HIP-event timing of the scoring kernels on/off (same as UTM_FLAG_PROFILE_EVENTS). */
int utm_set_profile(utm_ctx *ctx, int32_t on);

/* ---- sharded operation: building blocks for any transport ------------------------------------- */
/* Score locally and report this shard's best candidate; nothing is selected. */
int utm_local_best(utm_ctx *ctx, utm_record *rec);
/* Words per sample summed over chunks (length of a whole-column buffer). */
int utm_column_words(utm_ctx *ctx, uint64_t *n_words);
/* Concatenated chunk columns of a LOCAL sample (global index), utm_column_words() words. */
int utm_get_column(utm_ctx *ctx, int64_t global_idx, uint64_t *out);
/* Apply one exchanged decision: recs[n_ranks] are all shards' records in rank order; the winner is
 * chosen exactly as the fused path does (score desc, index asc).  winner_col is the winner's whole
 * column, or NULL when the winner is local to this context.  Outputs as utm_step. */
int utm_apply_records(utm_ctx *ctx, const utm_record *recs, int32_t n_ranks, const uint64_t *winner_col,
                      int64_t *idx, int64_t *new_count, double *score);

/* ---- P2P column access (optional; utm_comm_init sets it up by itself) ---------------------------- */
/* Every shard can map the other shards' columns (hipIpc) and read a remote winner's column in place
 * instead of receiving it: export one blob per shard, hand all shards' blobs (rank order) to import.
 * Afterwards utm_apply_records accepts winner_col == NULL for remote winners. */
int utm_p2p_blob_bytes(utm_ctx *ctx, uint64_t *n_bytes);
int utm_p2p_export(utm_ctx *ctx, void *blob);
int utm_p2p_import(utm_ctx *ctx, int32_t rank, int32_t n_ranks, const void *blobs);
/* Record mailboxes: with the mappings in place the shards can also exchange their per-iteration records on
 * the device (remote stores into each other's uncached mailbox slots, bounded polling) -- no collective and no
 * host in the loop.  utm_p2p_selftest is collective (every shard calls it): *ok = 1 when this shard received
 * every peer's test record.  If ALL shards report 1, call utm_p2p_use_mailboxes(ctx, 1) on every shard; utm_run
 * is then collective over the shards exactly like after utm_comm_init. */
int utm_p2p_selftest(utm_ctx *ctx, int32_t *ok);
/* on = 1: as above.  on = 2: also on a context that is the ONLY shard (exported to and imported from itself): it then posts
 * to and collects from its own mailbox every iteration -- the exchange's own per-iteration cost, measurable on one GPU
 * (bench.py --force-mailboxes).  on = 0: off. */
int utm_p2p_use_mailboxes(utm_ctx *ctx, int32_t on);
/* ---- RCCL (one process per GPU; ids are exchanged by the caller) ------------------------------ */
#define UTM_UNIQUE_ID_BYTES 128
int utm_comm_get_unique_id(void *id);
/* One communicator per context, one rank per GPU.  After this utm_step / utm_run / utm_reset are collective over
 * the ranks and one greedy iteration is SURVEY.md 8e's protocol:
 *   local scoring + local best  ->  ncclAllGather of the 64-byte records (RCCL has no MAXLOC; every rank then takes
 *   the same maximum: score descending, global index ascending = np.argmax's first maximum, select.py:48)
 *   ->  ncclBroadcast of the winner's column (W x 8 bytes per chunk) from the rank that owns it  ->  every rank ORs
 *   it into its covered replica while it stages the next scoring pass.
 * The broadcast's root is known on the host only after the decision, so this mode synchronises the stream once per
 * iteration.  If the record mailboxes are also enabled (utm_p2p_use_mailboxes) they carry the loop and the
 * communicator is left to utm_comm_allreduce_max; utm_stats.exchange says which.  Samples that start out used
 * (state 0) on another rank are broadcast by their owner at utm_reset. */
int utm_comm_init(utm_ctx *ctx, int32_t rank, int32_t n_ranks, const void *id);
int utm_comm_allreduce_max(utm_ctx *ctx, double *value); /* in place; also a barrier */
/* How the winner's column travels in the RCCL exchange.  0 (default): ncclBroadcast from its owner -- the root is
 * data dependent, so the host reads it after every iteration (one stream sync per iteration).  1: the owner puts the
 * column into the winner-column buffer, everybody else zeros, and one ncclAllReduce(sum, uint64) leaves the column
 * on every rank -- about twice the bytes on the links, but no root and therefore no host in the loop (iterations are
 * enqueued in batches like on one GPU).  Same rows either way. */
int utm_comm_column_by_allreduce(utm_ctx *ctx, int32_t on);

#ifdef __cplusplus
}
#endif
#endif /* UTMOS_HIP_H */
