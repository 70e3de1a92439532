// One greedy iteration as kernel launches, and the entry points that run them: utm_run / utm_step / utm_peek_scores / stats.
// Part of the one translation unit utmos_hip.hip (included there, in order); not a stand-alone header.
#pragma once

// ---------------------------------------------------------------------------------------- launches
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

// How one integer scoring launch is to run: counts by position in act[] or by sample; the iteration's pick inside
// the launch (its last workgroup) or left to a k_pick launch.
struct IntLaunch {
    bool by_pos = true;
    int fused = 0;  // 0: a k_pick launch follows; 1: pick inside the launch (only shard); 2: pick + mailbox exchange inside
};

template <int STEPS>
static void launch_score_int(utm_ctx *c, const LaunchTimer &t, const Chunk &ch, unsigned blocks, unsigned group, unsigned n_groups,
                             bool nt, const IntLaunch &how)
{
    const u64 *cols = ch.cols;
    const PickArgs pa = pick_args(c);
#define UTM_LAUNCH_INT(NT, FUSED)                                                                                                  \
    UTM_TIMED_LAUNCH(t, (k_score_int<STEPS, NT, FUSED>), dim3(blocks + (FUSED ? 1u : 0u)), dim3(256), cols, ch.covered, ch.wp,       \
                     pending_of(c, ch, true), (const IterState *)c->d_st, (const unsigned *)c->d_act, c->d_cnt, group, n_groups,   \
                     how.by_pos ? 1u : 0u, pa)
    if (how.fused == 2) {
        if (nt) UTM_LAUNCH_INT(true, 2);
        else UTM_LAUNCH_INT(false, 2);
    } else if (how.fused == 1) {
        if (nt) UTM_LAUNCH_INT(true, 1);
        else UTM_LAUNCH_INT(false, 1);
    } else {
        if (nt) UTM_LAUNCH_INT(true, 0);
        else UTM_LAUNCH_INT(false, 0);
    }
#undef UTM_LAUNCH_INT
}

static void launch_apply_pending(utm_ctx *c)
{
    for (auto &ch : c->chunks)
        hipLaunchKernelGGL(k_apply_pending, dim3((unsigned)std::min<u64>(1024, (ch.wp + 255) / 256)), dim3(256), 0, c->stream,
                           ch.covered, ch.cols, ch.wp, pending_of(c, ch, false), c->d_st, 1);
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
                              (int)c->chunks.size(), c->d_st, c->d_act, c->d_cnt, c->d_fscore);
    else
        hipExtLaunchKernelGGL(k_score_seq<double>, dim3(blocks), dim3(64), 0, c->stream, t.start, t.stop, 0, c->d_seq,
                              (int)c->chunks.size(), c->d_st, c->d_act, c->d_cnt, c->d_fscore);
}

// AF, dense phase: LDS AF tiles.  Every workgroup re-stages its 32 KiB AF tile (from L2 / Infinity Cache), so
// the groups hold >= 64 samples.
// The full dense AF pass takes the table kernel (k_score_aft) when every fixed-point value fits its limbs.
static bool af_table_pass(const utm_ctx *c, unsigned a_ub)
{
    return c->af_table_ok && c->tune.af_tables && a_ub >= 64;
}

static void launch_score_af_dense(utm_ctx *c, const Chunk &ch, unsigned a_ub, bool delta_fold = false)
{
    if (!delta_fold && af_table_pass(c, a_ub)) {
        // full pass as table lookups (k_score_aft): 4,096-variant tiles, 1,024 threads; a workgroup = a run of up to
        // UTM_AFT_RUN tiles x a group of at most 32 * MAXU samples (MAXU = 8 or 4 units of two samples per wave, kept in registers
        // over the run).  The caller has applied the pending winner (enqueue_score).
        const u64 tiles = ch.wp / UTM_AFT_TILE_WORDS;
        int maxu = 8;
        unsigned n_groups = (a_ub + 32 * maxu - 1) / (32 * maxu);
        if ((u64)(a_ub + 127) / 128 * 128 + 16 < (u64)n_groups * 256) { maxu = 4; n_groups = (a_ub + 127) / 128; }  // fewer idle units
        const unsigned group = ((a_ub + n_groups - 1) / n_groups + 1) / 2 * 2;
        n_groups = (a_ub + group - 1) / group;
        // runs: as long as possible while the grid still has about UTM_AF_TABLE_WGS_PER_CU workgroups per CU (one is
        // resident per CU; the sums are flushed every UTM_AFT_RUN tiles inside the kernel)
        const u64 want = 256ull * (u64)std::max(1, c->tune.af_table_wgs_per_cu);
        unsigned run = (unsigned)std::max<u64>(1, (tiles * n_groups + want - 1) / want);  // (rounded up: the grid ends just below a multiple of the CUs)
        if (c->tune.af_table_run >= 1) run = (unsigned)c->tune.af_table_run;
        const unsigned n_runs = (unsigned)((tiles + run - 1) / run);
        LaunchTimer t(c);
#define UTM_LAUNCH_AFT(M)                                                                                                      \
    hipExtLaunchKernelGGL(k_score_aft<M>, dim3((unsigned)round_up((u64)n_runs * n_groups, 8)), dim3(UTM_AFT_THREADS), 0, c->stream, t.start, \
                          t.stop, 0, ch.cols, ch.covered, ch.wp, ch.afx, c->d_st, c->d_act, c->d_cnt, c->d_afsum, group, n_groups, run)
        if (maxu == 8) UTM_LAUNCH_AFT(8);
        else UTM_LAUNCH_AFT(4);
#undef UTM_LAUNCH_AFT
        c->af_table_passes += 1;
        return;
    }
    const int af_target = c->tune.af_target_wgs;
    const u64 tiles = ch.wp / UTM_AF_TILE_WORDS;
    unsigned n_groups = (unsigned)std::max<u64>(1, std::min<u64>((a_ub + 63) / 64, (u64)af_target / std::max<u64>(1, tiles)));
    const unsigned group = ((a_ub + n_groups - 1) / n_groups + 15) / 16 * 16;
    n_groups = (a_ub + group - 1) / group;
    LaunchTimer t(c);
    hipExtLaunchKernelGGL(k_score_afq, dim3((unsigned)round_up(tiles * n_groups, 8)), dim3(256), 0, c->stream, t.start, t.stop, 0, ch.cols,
                          ch.covered, ch.wp, ch.afx, pending_of(c, ch, true), c->d_st, c->d_act, c->d_cnt, c->d_afsum,
                          group, n_groups, delta_fold ? ch.covered_alt : nullptr,
                          (delta_fold && defer_active(c) && c->enq_iter > 0)
                              ? c->d_newly_log + (u64)((c->enq_iter - 1) % c->defer_slots) * c->col_words + ch.off : nullptr);
}

// The streaming kernels (k_score_int, k_score_afs): grid = variant tiles x groups of samples.  Tile = the largest
// of {32 (AF: 16), 8, 2} KiB that still yields >= UTM_MIN_WGS workgroups; group size such that the grid has about
// UTM_TARGET_WGS workgroups (>> 256 CUs, small enough units for an even tail), at least one sample per wave.
static void launch_score_streaming(utm_ctx *c, const Chunk &ch, unsigned a_ub, bool delta = false, const IntLaunch &how = IntLaunch(),
                                   bool fold = false)
{
    const Tune &tn = c->tune;  // (defaults and what each knob is for: g_knobs, utmos_hip.hip)
    const int target_wgs = tn.target_wgs, min_wgs = tn.min_wgs, min_wgs_big = tn.min_wgs_big, force_steps = tn.tile_steps;
    const int nt_env = tn.nt_loads, nt_min_mb = tn.nt_min_mb;
    // non-temporal column loads when the matrix is a stream far larger than the 256 MB Infinity Cache (+10 % at
    // 3 GB); a matrix that (nearly) fits is better left to the caches (chr22-sized 345 MB: +5 %).  By the matrix, not
    // by the columns still selectable: the tail of a 3 GB select-all run measured slower with cached loads.
    const bool use_nt = nt_env >= 0 ? nt_env != 0 : (u64)c->n_local * c->col_words * 8 > ((u64)nt_min_mb << 20);
    const int af_big = tn.af_steps == 32 ? 32 : 16;
    const bool af = c->af_mode != UTM_AF_NONE;
    const u64 steps_total = ch.wp / UTM_STEP_WORDS;
    const u64 waves_needed = (a_ub + 3) / 4;  // workgroups if every wave had one sample
    int steps = 2;
    if (af) {
        for (int cand : {af_big, 8}) {  // (the AF kernel shares LDS with its bit queues)
            const u64 tiles = (steps_total + cand - 1) / cand;
            if (tiles * waves_needed >= (u64)(cand > 8 ? min_wgs_big : min_wgs)) { steps = cand; break; }
        }
    } else {
        // integer scores: 32 KiB tiles only for a deep grid (10M x 2,504 down to ~1,700 selectable samples), else 16 KiB,
        // 8 KiB for the smallest grids.  Measured over heights 1.5M .. 10M x 150 .. 2,504 selectable samples, first 40
        // iterations each (tools/tile_grid.sh, profiles/r03_tile_grid.txt): 16 KiB is the best tile or within 1-2 % of it
        // in every cell but (10M, 2,504) -- where 32 KiB leads by 1 % -- and beats 8 KiB by 3 % (10M x 300), 4 % (2M x
        // 2,504) up to 20 % (10M x 150: few samples, many tiles -- the count words' atomics).
        // Columns shorter than 14 such tiles (1.8M variants) keep 8 KiB: at 1.1M x 2,504 the 16 KiB tile measured -2.4 %,
        // at 1.5M -1.7 % .. 0 (same box, ab/old.so against ab/new.so: profiles/r03_tile16_ab_same_box.txt).
        for (int cand : {32, 16, 8}) {
            const u64 tiles = (steps_total + cand - 1) / cand;
            if (cand == 16 && tiles < 14) continue;
            if (tiles * waves_needed >= (u64)(cand == 32 ? 2 * min_wgs_big : min_wgs)) { steps = cand; break; }
        }
    }
    if (!af && (force_steps == 32 || force_steps == 16 || force_steps == 8 || force_steps == 4 || force_steps == 2)) steps = force_steps;
    const u64 tiles = (steps_total + steps - 1) / steps;
    u64 group = ((u64)a_ub * tiles + target_wgs - 1) / target_wgs;
    group = std::max<u64>(4, (group + 3) / 4 * 4);
    const unsigned n_groups = (unsigned)((a_ub + group - 1) / group);
    const unsigned blocks = (unsigned)round_up(tiles * n_groups, 8);  // XCD-aware map: tile_of_block()
    LaunchTimer t(c);
    if (af) {
        const unsigned *afb = ch.afx;
#define UTM_LAUNCH_AFS(S, Q)                                                                                              \
    hipExtLaunchKernelGGL((k_score_afs<S, Q>), dim3(blocks), dim3(256), 0, c->stream, t.start, t.stop, 0, ch.cols,       \
                          ch.covered, ch.wp, afb, pending_of(c, ch, true), c->d_st, c->d_act, c->d_cnt, c->d_afsum,   \
                          (unsigned)group, n_groups, (delta && !fold) ? ch.mask : nullptr, (delta && fold) ? ch.covered_alt : nullptr, \
                          (delta && fold && defer_active(c) && c->enq_iter > 0)                                                       \
                              ? c->d_newly_log + (u64)((c->enq_iter - 1) % c->defer_slots) * c->col_words + ch.off : nullptr)
        if (steps == 32) UTM_LAUNCH_AFS(32, 8);  // second argument: queue depth per lane
        else if (steps == 16) UTM_LAUNCH_AFS(16, 16);
        else if (steps == 8) UTM_LAUNCH_AFS(8, 16);
        else UTM_LAUNCH_AFS(2, 16);
#undef UTM_LAUNCH_AFS
    } else if (steps == 32) launch_score_int<32>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt, how);
    else if (steps == 16) launch_score_int<16>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt, how);
    else if (steps == 8) launch_score_int<8>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt, how);
    else if (steps == 4) launch_score_int<4>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt, how);
    else launch_score_int<2>(c, t, ch, blocks, (unsigned)group, n_groups, use_nt, how);
}

// ---------------------------------------------------------------------------------------- the persistent loop
// Shape of a k_loop_int launch for this context, or ok = false when the loop has to run as one launch per iteration:
// unweighted integer scores, the only shard, one chunk, no decremental mode, and a tile grid that fits the blocks that
// can be resident together.  Tile: 8 KiB while every tile then gets at least two worker slots, else 16 KiB.
struct LoopShape {
    bool ok = false;
    int steps = 8;
    unsigned q_slots = 0, n_tiles = 0;
};
// af: the caller has established an AF form's preconditions (loop_af_form: 1 exact, 2 intervals); otherwise AF runs keep
// the launches.  (Weighted integer scores take the loop too: the picker compares float64 products.)
static int loop_chainers(const utm_ctx *c)
{
    return std::min(UTM_LOOP_MAX_CHAINERS, std::max(1, c->tune.persist_chainers));
}
static LoopShape loop_shape(utm_ctx *c, int af = 0)
{
    LoopShape sh;
    const Tune &tn = c->tune;
    // (interval form: the picker keeps every selectable sample's accumulators and record in registers -- one chunk of words)
    if (af == 2 && c->n_local > (unsigned)(UTM_LOOP_THREADS * UTM_LOOP_E)) return sh;
    if (!tn.persistent || c->persist_off || (!af && c->af_mode != UTM_AF_NONE) || c->decr_enabled || c->chunks.size() != 1 ||
        c->n_ranks != 1 || c->n_local != c->n_total || c->comm || c->p2p || c->n_local >= UTM_LOOP_MAX_LOCAL ||
        (tn.persist_max_samples > 0 && c->n_local > (unsigned)tn.persist_max_samples))
        return sh;
    const Chunk &ch = c->chunks[0];
    if (tn.persist_max_mb > 0 && (u64)c->n_local * ch.wp * 8 > ((u64)tn.persist_max_mb << 20)) return sh;
    static int cus = 0, occ_all[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};  // resident blocks per CU for the 8 / 16 / 32 / 64 KiB tile, integer / AF / AF-interval form
    if (!cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return sh;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[0][0], k_loop_int<8, true, 0>, UTM_LOOP_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[0][1], k_loop_int<16, true, 0>, UTM_LOOP_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[0][2], k_loop_int<32, true, 0>, UTM_LOOP_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[0][3], k_loop_int<64, true, 0>, UTM_LOOP_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[1][0], k_loop_int<8, true, 1>, UTM_LOOP_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[1][1], k_loop_int<16, true, 1>, UTM_LOOP_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[1][2], k_loop_int<32, true, 1>, UTM_LOOP_THREADS, 0);
        occ_all[1][3] = occ_all[2][3] = 0;  // (the AF forms are not built for 64 KiB tiles)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[2][0], k_loop_int<8, true, 2>, UTM_LOOP_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[2][1], k_loop_int<16, true, 2>, UTM_LOOP_THREADS, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_all[2][2], k_loop_int<32, true, 2>, UTM_LOOP_THREADS, 0);
        cus = prop.multiProcessorCount;
        if (getenv("UTM_VERBOSE"))
            fprintf(stderr, "libutmos_hip: k_loop_int occupancy query: %d / %d / %d / %d (AF form %d / %d / %d / %d, with intervals %d / %d / %d / %d) blocks of %d threads per CU (8 / 16 / 32 / 64 KiB tile), %d CUs\n",
                    occ_all[0][0], occ_all[0][1], occ_all[0][2], occ_all[0][3], occ_all[1][0], occ_all[1][1], occ_all[1][2], occ_all[1][3],
                    occ_all[2][0], occ_all[2][1], occ_all[2][2], occ_all[2][3], UTM_LOOP_THREADS, cus);
    }
    const int *occ = occ_all[af];
    // Tile: the smallest of 8 / 16 / 32 / 64 KiB that cuts a column into at most persist_max_tiles tiles -- every (position,
    // tile) pair costs one atomic on the position's count word, 16 count words share a cache line, and beyond ~30 tiles
    // those lines become the bottleneck (measured: 2,504 samples, 31 tiles +3.5 %, 39 tiles -2 %, 153 tiles -15 % against
    // one launch per iteration).  A matrix taller than that keeps the launch-per-iteration path.
    const u64 steps_total = ch.wp / UTM_STEP_WORDS;
    for (int t = 0; t < 4; ++t) {
        const int steps = 8 << t;
        if (tn.persist_tile_kib > 0 && steps != tn.persist_tile_kib) continue;
        const u64 tiles = (steps_total + steps - 1) / steps;
        if (tiles > (u64)std::max(1, tn.persist_max_tiles) && tn.persist_tile_kib <= 0) continue;
        // The AF forms: the one-batch tile only, and fewer tiles -- their workers keep two tiles in LDS and gather, their
        // words travel in pairs.  Measured against the launches (2,504 samples, float32 / float64 AF): 17 tiles +23 % /
        // +22 %, 24 tiles +7 % / +6 %, 28 tiles . / -1 %, 31 tiles -1.5 % / -5 %; 16 KiB tiles (3M x 640) +2 % / -13 %;
        // 64 KiB tiles (10M x 313) -50 % / -51 %.
        if (af && tn.persist_tile_kib <= 0 && (steps > 8 || tiles > (u64)std::max(1, tn.persist_af_max_tiles))) break;
        if (af && steps > 32) break;  // (64 KiB tiles with the AF forms' second tile: two waves per SIMD, half the launches' rate -- not even built)
        // tiles of several batches only pay where the launch per iteration is weak -- few samples, tall columns (10M x 313:
        // +12 %); with 2,504 samples they lose ~10 % to the one-batch tile at equal height (1.1M: 0.70 against 0.79) and
        // are level with the launches at best (3M: +2 % / -3.4 %)
        if (steps > 8 && tn.persist_tile_kib <= 0 && c->n_local > (unsigned)std::max(0, tn.persist_tall_max_samples)) break;
        int per_cu = std::min(occ[t], 2048 / UTM_LOOP_THREADS);
        if (tn.persist_wgs_per_cu > 0) per_cu = tn.persist_wgs_per_cu;  // (an override, also upwards: the census decides whether the grid is resident)
        const u64 max_workers = (u64)cus * (u64)std::max(per_cu, 0);
        const u64 others = af == 2 ? 1 + (u64)loop_chainers(c) : 1;  // the picker's block (and the interval form's chainers) come out of the same budget
        if (max_workers < 16 || tiles > max_workers - others) continue;
        const u64 q = (max_workers - others) / tiles;
        sh.ok = true;
        sh.steps = steps;
        sh.n_tiles = (unsigned)tiles;
        sh.q_slots = (unsigned)std::max<u64>(1, std::min<u64>(q, ((u64)c->n_local + UTM_LOOP_WAVES - 1) / UTM_LOOP_WAVES));  // (more wave slots than samples would only idle)
        return sh;
    }
    return sh;
}

// Up to k_batch iterations as ONE launch (k_loop_int): the picker's record replaces the kernel boundary.
// The AF form of the persistent loop applies in the exact fixed-point phase: float32 AF on a lossless unit, every
// selectable sample's sum below 2^53 units (latched from the device: af_all_exact -- the plain exact pick suffices, no
// candidates, no chains, no deferred scores), the per-sample accumulators valid, and the pending winner no longer covering
// percents of all variants (those first delta passes gather from LDS in k_score_afq; here a gather is a global load).
// The interval form (2) takes the other verified-parallel AF runs -- float64 AF values, or float32 sums that are not
// (yet) all exact: candidates and their chains inside the picker (loop_picker<3>), as long as a lone candidate needs no
// chain of its own (its exact score is deferred, or not wanted: PickArgs::af_skip_single).  -> 0 (the launches), 1, 2.
static int loop_af_form(const utm_ctx *c, bool first_is_full)
{
    if (c->persist_backoff > 0) return 0;  // (launches of the interval form kept ending undecided: some iterations as launches first)
    if (c->af_mode == UTM_AF_NONE || !c->af_fixed || !c->keep_valid || first_is_full || c->last_new < 0 ||
        (double)c->last_new > c->tune.af_dense_delta * (double)c->n_var_total)
        return 0;
    if (c->af_mode == UTM_AF_F32 && !c->af_trunc && c->af_all_exact) return c->tune.persist_af ? 1 : 0;
    if (c->af_all_exact) return 0;
    const bool skip_single = (!c->af_exact_scores || defer_active(c)) && c->n_local == c->n_total;
    return (c->tune.persist_af && c->tune.persist_af_interval && c->tune.af_record && skip_single) ? 2 : 0;  // (the sums on record are how the chainer answers)
}

static int enqueue_loop(utm_ctx *c, const LoopShape &sh, int k_batch, int af = 0)
{
    const Chunk &ch = c->chunks[0];
    if (af && !c->d_loop_w[0]) {  // the AF form's per-position words: two arrival / count-decrease words, two sum-decrease words
        const size_t bytes = ((size_t)c->n_local + UTM_PICK_PAD) * 8;
        for (int i = 0; i < 4; ++i) {
            HIP_TRY(hipMalloc(&c->d_loop_w[i], bytes));
            HIP_TRY(hipMemsetAsync(c->d_loop_w[i], 0, bytes, c->stream));
        }
    }
    const bool use_nt = c->tune.nt_loads >= 0 ? c->tune.nt_loads != 0 : (u64)c->n_local * c->col_words * 8 > ((u64)c->tune.nt_min_mb << 20);
    // position claim counters: [3 sets, iteration % 3][tiles][wave index], 128 B apart
    const size_t claim_bytes = (size_t)3 * sh.n_tiles * UTM_LOOP_WAVES * UTM_CLAIM_STRIDE * sizeof(unsigned);
    if (c->claim_bytes < claim_bytes) {
        (void)hipFree(c->d_claim);
        c->d_claim = nullptr;
        c->claim_bytes = 0;
        HIP_TRY(hipMalloc(&c->d_claim, claim_bytes));
        c->claim_bytes = claim_bytes;
    }
    HIP_TRY(hipMemsetAsync(c->d_claim, 0, claim_bytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_loop_sync, 0, sizeof(LoopSync), c->stream));
    const PickArgs pa = pick_args(c);
    const int n_chainers = af == 2 ? loop_chainers(c) : 0;
    if (af == 2 && c->loop_priv_words < (size_t)n_chainers * ch.wp) {  // the chainers' covered masks
        (void)hipFree(c->d_loop_priv);
        c->d_loop_priv = nullptr;
        c->loop_priv_words = 0;
        HIP_TRY(hipMalloc(&c->d_loop_priv, (size_t)n_chainers * ch.wp * 8));
        c->loop_priv_words = (size_t)n_chainers * ch.wp;
    }
    const dim3 grid(sh.n_tiles * sh.q_slots + 1 + n_chainers);  // picker + workers (+ chainers)
    const int drop = c->tune.test_drop_arrival;
    LaunchTimer t(c);
    const LoopAf laf{af ? ch.afx : nullptr, af ? c->d_loop_w[2] : nullptr, af ? c->d_loop_w[3] : nullptr,
                     af == 2 ? ch.af : nullptr, af == 2 ? c->d_loop_priv : nullptr,
                     (af == 2 && defer_active(c)) ? c->d_newly_log + ch.off : nullptr, c->col_words, c->defer_slots, c->tune.persist_spec_ticks, n_chainers};
    u64 *w0 = af ? c->d_loop_w[0] : c->d_cnt, *w1 = af ? c->d_loop_w[1] : c->d_cnt_alt;  // (AF: d_cnt holds the per-sample counts)
    if (getenv("UTM_VERBOSE") && c->persist_launches == 0)
        fprintf(stderr, "libutmos_hip: k_loop_int form %d: cols %p..%p covered %p priv %p (wp %llu) af %p afx %p log %p (stride %llu) known %p %p words %p %p %p %p act %p cnt %p afsum %p sync %p claim %p grid %u\n",
                af, (void *)ch.cols, (void *)(ch.cols + (size_t)c->n_local * ch.wp), (void *)ch.covered, (void *)c->d_loop_priv, (unsigned long long)ch.wp,
                (void *)ch.af, (void *)ch.afx, (void *)laf.newly_log, (unsigned long long)c->col_words, (void *)c->d_known_cnt, (void *)c->d_known_val,
                (void *)c->d_loop_w[0], (void *)c->d_loop_w[1], (void *)c->d_loop_w[2], (void *)c->d_loop_w[3], (void *)c->d_act, (void *)c->d_cnt,
                (void *)c->d_afsum, (void *)c->d_loop_sync, (void *)c->d_claim, grid.x);
#define UTM_LAUNCH_LOOP(S, NT, AFF)                                                                                                 \
    UTM_TIMED_LAUNCH(t, (k_loop_int<S, NT, AFF>), grid, dim3(UTM_LOOP_THREADS), (const u64 *)ch.cols, ch.covered, ch.wp, pending_of(c, ch, true), c->d_st, \
                     c->d_act, w0, w1, sh.q_slots, c->d_claim, k_batch, c->d_loop_sync, pa, drop, c->tune.persist_claims, c->tune.persist_ahead_ticks, c->tune.persist_ahead0_ticks, laf)
#define UTM_LAUNCH_LOOP2(S, NT) \
    if (af == 2) UTM_LAUNCH_LOOP(S, NT, 2); \
    else if (af) UTM_LAUNCH_LOOP(S, NT, 1); \
    else UTM_LAUNCH_LOOP(S, NT, 0)
    switch (sh.steps * 2 + (use_nt ? 1 : 0)) {
    case 17: UTM_LAUNCH_LOOP2(8, true); break;
    case 16: UTM_LAUNCH_LOOP2(8, false); break;
    case 33: UTM_LAUNCH_LOOP2(16, true); break;
    case 32: UTM_LAUNCH_LOOP2(16, false); break;
    case 65: UTM_LAUNCH_LOOP2(32, true); break;
    case 64: UTM_LAUNCH_LOOP2(32, false); break;
    case 129: UTM_LAUNCH_LOOP(64, true, 0); break;  // (integer only: loop_shape)
    default: UTM_LAUNCH_LOOP(64, false, 0); break;
    }
#undef UTM_LAUNCH_LOOP2
#undef UTM_LAUNCH_LOOP
    HIP_TRY(hipGetLastError());
    c->persist_launches += 1;
    return UTM_OK;
}

// Every chunk's (covered, covered_alt) pair and the chain kernels' chunk tables change roles together.
static void swap_covered(utm_ctx *c)
{
    for (auto &ch : c->chunks) std::swap(ch.covered, ch.covered_alt);
    std::swap(c->d_seq, c->d_seq_alt);
}

// Enqueue the scoring of one iteration for every chunk (and the pending covered update).
// `fuse_pick`: the caller wants the iteration's pick too and nothing else in between (utm_run on the only shard,
// integer scores): it then rides in the last chunk's scoring launch and *fused is set; `by_sample` keeps the counts
// indexed by sample for callers that read them back per sample (utm_peek_scores).
static int enqueue_score(utm_ctx *c, bool force_sequential = false, int fuse_pick = 0, bool *fused = nullptr,
                         bool by_sample = false)
{
    if (fused) *fused = false;
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
            const double af_switch = c->tune.af_switch;
            const bool af_dense = (double)c->captured_seen < af_switch * (double)c->n_var_total;
            if (remote_reads(c) || (af_dense && af_table_pass(c, a_ub))) launch_apply_pending(c);  // (k_score_aft reads `covered` as it is)
            for (auto &ch : c->chunks) {
                if (af_dense) launch_score_af_dense(c, ch, a_ub);
                else launch_score_streaming(c, ch, a_ub);
            }
            c->keep_valid = true;
        } else if (remote_reads(c)) {
            // the winner may sit on another GPU: one launch reads its column once and leaves the newly-covered mask
            for (auto &ch : c->chunks)
                hipLaunchKernelGGL(k_newly_mask, dim3((unsigned)std::min<u64>(2048, (ch.wp + 255) / 256)), dim3(256), 0, c->stream,
                                   ch.covered, ch.cols, ch.wp, pending_of(c, ch, false), c->d_st, ch.mask);
            for (auto &ch : c->chunks) launch_score_streaming(c, ch, a_ub, /*delta=*/true);
        } else {
            // the mask is made while the tiles are staged; the updated covered words land in the other buffer of each
            // chunk's pair, which becomes the current one for everything enqueued from here on (utm_run undoes the
            // swaps of launches that a finished loop skipped)
            // While a winner still newly covers percents of all variants (the first iterations of a run) every surviving
            // bit of a delta pass is a gather: those passes take the LDS-tile kernel, the rest the streaming one.  (By
            // the last gain the host has seen: batches are 4 iterations long while this matters.  At 10M x 2,504 the
            // first three delta passes take 1.8 / 1.3 / 1.1 ms there instead of 4.6 / 2.2 / 1.6 ms; from the fourth on
            // the streaming kernel is ahead.)
            const double dense_delta = c->tune.af_dense_delta;
            const bool in_lds = c->last_new < 0 || (double)c->last_new > dense_delta * (double)c->n_var_total;
            for (auto &ch : c->chunks) {
                if (in_lds) launch_score_af_dense(c, ch, a_ub, /*delta_fold=*/true);
                else launch_score_streaming(c, ch, a_ub, /*delta=*/true, IntLaunch(), /*fold=*/true);
            }
            swap_covered(c);
            c->cov_swaps_enqueued += 1;
        }
    } else {
        if (remote_reads(c)) launch_apply_pending(c);  // remote column: read it once, not once per workgroup
        const int fuse_env = c->tune.fuse_pick;
        IntLaunch how;
        how.by_pos = !by_sample;
        for (size_t k = 0; k < c->chunks.size(); ++k) {
            // (weighted scores are float64 products: they stay with k_pick, the fused pick compares integer counts)
            // ... and so do the full passes of a decremental run (they mirror their counts by sample)
            how.fused = (fuse_pick && fuse_env && !by_sample && !c->have_weights && !c->decr_enabled && k + 1 == c->chunks.size()) ? fuse_pick : 0;
            launch_score_streaming(c, c->chunks[k], a_ub, false, how);
            if (how.fused && fused) *fused = true;
        }
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
        const unsigned *afbits = af ? ch.afx : nullptr;
        u64 *cnt = af ? c->d_cnt : c->d_cnt_keep;
        i64 *afsum = af ? c->d_afsum : c->d_afsum_keep;
        if (c->decr_interleaved) {
            const dim3 grid((unsigned)((s_t + 255) / 256), slices);
            if (af)
                hipLaunchKernelGGL(k_decr_t<true>, grid, dim3(256), 0, c->stream, ch.rows_t, s_t, afbits, c->d_st, c->d_state,
                                   c->n_local, ch.list_idx, ch.list_val, c->d_listn + k, cnt, afsum);
            else
                hipLaunchKernelGGL(k_decr_t<false>, grid, dim3(256), 0, c->stream, ch.rows_t, s_t, afbits, c->d_st, c->d_state,
                                   c->n_local, ch.list_idx, ch.list_val, c->d_listn + k, cnt, afsum);
            continue;
        }
        const dim3 grid((a_ub + 3) / 4, split);
        if (af)
            hipLaunchKernelGGL(k_decr<true>, grid, dim3(256), 0, c->stream, ch.cols, ch.wp, afbits, c->d_st, c->d_act,
                               ch.list_idx, ch.list_val, c->d_listn + k, cnt, afsum);
        else
            hipLaunchKernelGGL(k_decr<false>, grid, dim3(256), 0, c->stream, ch.cols, ch.wp, afbits, c->d_st, c->d_act,
                               ch.list_idx, ch.list_val, c->d_listn + k, cnt, afsum);
    }
    HIP_TRY(hipGetLastError());
    return UTM_OK;
}

// Algorithmic HBM bytes of one scoring pass with `a` selectable local samples (SURVEY.md §8d "the
// bytes that variant actually has to read"): active columns + covered read + winner column re-read + covered write;
//   UTM_PASS_AF_FULL   + the per-variant AF table (a full AF pass, and every pass of the sequential AF kernel);
//   UTM_PASS_AF_DELTA  the AF values such a pass gathers are counted from the counts' decrease (utm_run) -- it never
//                        reads the table as a whole; + the newly-covered mask written once and read once where a
//                        separate launch makes it (winner columns read in place from another GPU).
enum { UTM_PASS_PLAIN = 0, UTM_PASS_AF_FULL = 1, UTM_PASS_AF_DELTA = 2 };
static i64 iteration_bytes(const utm_ctx *c, u64 a, int kind = -1)
{
    if (kind < 0) kind = c->af_mode != UTM_AF_NONE ? UTM_PASS_AF_FULL : UTM_PASS_PLAIN;
    i64 b = 0;
    for (auto &ch : c->chunks) {
        b += (i64)((a + 3) * ch.w * 8);
        if (kind == UTM_PASS_AF_FULL) b += (i64)ch.n_var * (c->af_mode == UTM_AF_F64 && !c->af_fixed ? 8 : 4);
        if (kind == UTM_PASS_AF_DELTA && remote_reads(c)) b += (i64)(2 * ch.w * 8);  // (a folded delta pass has no mask array)
    }
    return b;
}

// Verified-parallel AF: candidates -> their chains (-> everyone, if too many tie) [-> the pick, in the chain launch's
// last workgroup, when `pick_inside`].  Returns whether the pick was enqueued with it.
static void enqueue_chains(utm_ctx *c, const PickArgs &a, bool pick_inside)
{
    const unsigned seq_blocks = (std::max(1u, c->active_ub) + 1023) / 1024;  // (only busy when the candidate list overflowed)
    const ChainFast &cf = c->chain_fast;
    const dim3 grid(UTM_MAX_CAND + seq_blocks);
    const int n_chunks = (int)c->chunks.size();
#define UTM_LAUNCH_CHAIN(T, PICK)                                                                                                \
    hipLaunchKernelGGL((k_chain<T, PICK>), grid, dim3(1024), 0, c->stream, c->d_seq, n_chunks, c->d_st, c->d_cand, cf, c->d_act, \
                       c->d_cnt, c->d_fscore, a, c->d_arrivals)
    if (c->af_mode == UTM_AF_F32) {
        if (cf.counts)
            hipLaunchKernelGGL(k_chain_fill<float>, dim3(cf.n_segs, cf.n_cand), dim3(1024), 0, c->stream, c->d_seq, c->d_st, c->d_cand, cf);
        if (pick_inside) UTM_LAUNCH_CHAIN(float, true);
        else UTM_LAUNCH_CHAIN(float, false);
    } else {
        if (cf.counts)
            hipLaunchKernelGGL(k_chain_fill<double>, dim3(cf.n_segs, cf.n_cand), dim3(1024), 0, c->stream, c->d_seq, c->d_st, c->d_cand, cf);
        if (pick_inside) UTM_LAUNCH_CHAIN(double, true);
        else UTM_LAUNCH_CHAIN(double, false);
    }
#undef UTM_LAUNCH_CHAIN
}

static bool enqueue_candidates(utm_ctx *c, PickArgs a, bool pick_inside)
{
    if (!a.cand) return false;
    a.early_pick = pick_inside ? 1 : 0;
    const int verify_env = c->tune.af_verify;
    if (pick_inside && verify_env) {
        // the only shard: candidates, their addends, their chains and the pick as stages of ONE launch (k_verify)
        const ChainFast &cf = c->chain_fast;
        const unsigned n_fill = cf.counts ? (unsigned)cf.n_segs * (unsigned)cf.n_cand : 0u;
        const unsigned seq_blocks = (std::max(1u, c->active_ub) + 1023) / 1024;
        const dim3 grid(1 + n_fill + UTM_MAX_CAND + seq_blocks);
        const unsigned launch_no = ++c->verify_launches;
        const int n_chunks = (int)c->chunks.size();
        if (c->af_mode == UTM_AF_F32)
            hipLaunchKernelGGL(k_verify<float>, grid, dim3(1024), 0, c->stream, c->d_seq, n_chunks, c->d_st, c->d_cand, cf, c->d_act,
                               c->d_cnt, c->d_fscore, a, c->d_vsync, launch_no, n_fill);
        else
            hipLaunchKernelGGL(k_verify<double>, grid, dim3(1024), 0, c->stream, c->d_seq, n_chunks, c->d_st, c->d_cand, cf, c->d_act,
                               c->d_cnt, c->d_fscore, a, c->d_vsync, launch_no, n_fill);
        return true;
    }
    hipLaunchKernelGGL(k_cand, dim3(1), dim3(c->active_ub > 512 ? 1024 : 256), 0, c->stream, a);
    enqueue_chains(c, a, pick_inside);
    return pick_inside;
}

static int enqueue_pick_and_exchange(utm_ctx *c, bool decr = false)
{
    const int pick_env = c->tune.pick_threads;
    // single shard: 512 threads scan a few thousand counts as fast as 1024 and launch / join quicker (1024, 512, 256,
    // 128, 64 threads: 626.8, 623.6, 625.1, 629.6, 641.3 ms per cfg2 run)
    const unsigned pick_threads = pick_env ? (unsigned)pick_env : c->active_ub > 16384 ? 1024 : 512;
    PickArgs a = pick_args(c, decr);
    const int chain_pick = c->tune.chain_pick;
    const bool only_shard = c->n_ranks == 1 && c->n_local == c->n_total && !c->comm && !mailbox_exchange(c);
    if (enqueue_candidates(c, a, /*pick_inside=*/only_shard && chain_pick)) {
        HIP_TRY(hipGetLastError());
        return UTM_OK;  // the chain launch's last workgroup runs k_pick<0>'s body
    }
    if (mailbox_exchange(c)) {
        // device-side exchange: post this shard's record into every shard's mailbox, wait for theirs, decide
        hipLaunchKernelGGL(k_pick<2>, dim3(1), dim3(1024), 0, c->stream, a);  // pick, post, collect, decide
    } else if (c->comm) {
        // RCCL exchange, first half: every shard's 64-byte record to every shard (one ncclAllGather, in place), then the
        // same decision everywhere.  The second half -- the winner's column, ncclBroadcast from its owner -- needs the
        // owner's rank on the host: utm_run syncs after every iteration in this mode (rccl_exchange) and issues it.
        hipLaunchKernelGGL(k_pick<1>, dim3(1), dim3(1024), 0, c->stream, a);
        HIP_TRY(hipGetLastError());
        NCCL_TRY(g_rccl.AllGather(c->d_xbuf + (u64)c->rank * UTM_HDR_WORDS, c->d_xbuf, UTM_HDR_WORDS, ncclUint64, c->comm, c->stream));
        hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, a);
        if (c->column_by_allreduce) {
            // root-free second half: the owner stages its winner's column, everybody else zeros, the sum is the column
            u64 max_wp = 0;
            for (auto &ch : c->chunks) max_wp = std::max(max_wp, ch.wp);
            hipLaunchKernelGGL(k_stage_winner, dim3((unsigned)std::min<u64>(256, (max_wp + 255) / 256), (unsigned)c->chunks.size()), dim3(256), 0,
                               c->stream, c->d_wincol, c->d_stage, (const IterState *)c->d_st, c->rank, c->first);
            HIP_TRY(hipGetLastError());
            NCCL_TRY(g_rccl.AllReduce(c->d_wincol, c->d_wincol, c->col_words, ncclUint64, ncclSum, c->comm, c->stream));
        }
    } else if (c->n_ranks == 1 && c->n_local == c->n_total) {
        hipLaunchKernelGGL(k_pick<0>, dim3(1), dim3(pick_threads), 0, c->stream, a);
    } else {
        return fail(UTM_ESTATE, "this context holds a shard of the samples and has no exchange: enable the record mailboxes "
                                "(utm_p2p_import + utm_p2p_use_mailboxes) or RCCL (utm_comm_init), or drive it with utm_local_best / utm_apply_records");
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
    if (c->h_st->xerror == 3) { c->finished = false; return UTM_OK; }  // a persistent launch's census failed: utm_run falls back
    if (c->h_st->xerror == 2) return fail(UTM_EHIP, "a scoring launch's partial counts did not all arrive at its pick (internal error)");
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

// Deferred exact scores: finish result rows [lo, hi) -- their newly-covered masks are in the log.
static int defer_finish_rows(utm_ctx *c, i64 lo, i64 hi)
{
    if (hi <= lo) return UTM_OK;
    if (hi - lo > c->defer_slots) return fail(UTM_ESTATE, "deferred scores: %lld rows in one go (internal error)", (long long)(hi - lo));
    DeferArgs d;
    d.chunks = c->d_seq;
    d.segs = c->d_segs;
    d.n_segs = c->chain_fast.n_segs;
    d.log = c->d_newly_log;
    d.slots = c->defer_slots;
    d.col_words = c->col_words;
    d.counts = c->d_defer_counts;
    d.offs = c->d_defer_offs;
    d.vals = c->d_defer_vals;
    d.row0 = lo;
    d.n_rows = (int)(hi - lo);
    const dim3 grid((unsigned)d.n_segs, (unsigned)d.n_rows);
    hipLaunchKernelGGL(k_defer_count, grid, dim3(256), 0, c->stream, d);
    hipLaunchKernelGGL(k_defer_scan, dim3(1), dim3(1024), 0, c->stream, d);
    if (c->af_mode == UTM_AF_F32) hipLaunchKernelGGL(k_defer_fill<float>, grid, dim3(1024), 0, c->stream, d);
    else hipLaunchKernelGGL(k_defer_fill<double>, grid, dim3(1024), 0, c->stream, d);
    hipLaunchKernelGGL(k_defer_chain, dim3((unsigned)d.n_rows), dim3(1024), 0, c->stream, d, (const i64 *)c->d_res_idx, c->d_res_score,
                       c->have_weights ? (const double *)c->d_weights : nullptr);
    HIP_TRY(hipGetLastError());
    c->deferred_rows += hi - lo;
    c->defer_lo = hi;
    return UTM_OK;
}

// ... and the last row of a run, whose mask no later pass has made: pending winner & ~covered, then as above.
// Call with the device idle at a batch boundary (the host's covered roles are settled there).
static int defer_flush_last(utm_ctx *c)
{
    if (c->defer_lo >= c->iter) return UTM_OK;
    for (auto &ch : c->chunks)
        hipLaunchKernelGGL(k_newly_log, dim3((unsigned)std::min<u64>(1024, (ch.wp + 255) / 256)), dim3(256), 0, c->stream, ch.covered, ch.cols,
                           ch.wp, pending_of(c, ch, false), (const IterState *)c->d_st, c->d_newly_log + ch.off, c->col_words, c->defer_slots);
    return defer_finish_rows(c, c->defer_lo, c->iter);
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
    const int batch_env = c->tune.batch;
    const int decr_first = std::max(1, c->tune.decr_first_batch);
    // (RCCL exchange: one iteration per sync -- the column broadcast's root is only known on the host after it)
    int batch = rccl_needs_root(c) ? 1 : batch_env > 0 ? batch_env : c->af_mode != UTM_AF_NONE ? 64 : 256;
    // deferred exact AF scores log one newly-covered mask per row of a batch in UTM_DEFER_SLOTS slots (row % slots):
    // a longer batch would overwrite masks that are not finished yet
    if (c->af_mode != UTM_AF_NONE && c->af_fixed && defer_active(c)) batch = std::min(batch, c->defer_slots);
    if (c->af_mode == UTM_AF_NONE) batch = std::min(batch, 256);  // (the persistent loop's record carries an 8-bit iteration tag)
    i64 enq = 0;
    bool tail_deferred = false;  // the last batch left its last row's exact score to the end of the run
    while (enq < k_max && !c->finished) {
        // AF runs start with short batches: the dense -> sparse kernel switch is taken at a batch boundary
        // ... and so is the switch to decremental iterations
        // (the AF form of the persistent loop needs no host decision inside a batch: 256 iterations per launch, as for integers)
        const int form_hint = (batch_env <= 0 && !c->decr_enabled && !c->loop_unresolved) ? loop_af_form(c, false) : 0;
        const bool loop_af_batch = form_hint && loop_shape(c, form_hint).ok;
        // (interval form with deferred exact scores: a launch logs one newly-covered mask per row, UTM_DEFER_SLOTS slots)
        const i64 this_batch = c->loop_unresolved                                             ? 1
                               : c->persist_backoff > 0                                       ? std::min<i64>(batch, c->persist_backoff)
                               : loop_af_batch                                                ? ((form_hint == 2 && defer_active(c)) ? (i64)c->defer_slots : 256)
                               : (c->af_mode != UTM_AF_NONE && c->af_fixed && c->iter < 64) ? std::min<i64>(batch, c->iter < 8 ? 4 : 8)
                               : (c->decr_enabled && c->iter < 64)                          ? std::min<i64>(batch, decr_first)
                                                                                            : batch;
        i64 n = std::min<i64>(this_batch, k_max - enq);
        const unsigned a0 = c->active_ub;
        // Decremental batches: only when allowed, when the persistent counts are current, and when the last
        // winner newly covered few enough variants (gains shrink over a greedy run, so it stays that way).
        const bool decr = c->decr_enabled && c->keep_valid && c->last_new >= 0 && (c->af_mode == UTM_AF_NONE || c->af_fixed) &&
                          (double)c->last_new <= (c->decr_threshold > 0 ? c->decr_threshold : c->decr_interleaved ? 1.0 : 0.2) *
                                                     (double)c->col_words;
        // AF, verified-parallel: the first pass after anything invalidated the accumulators is a full one, the others
        // are delta passes (enqueue_score); the byte accounting below tells them apart
        const bool af_par = c->af_mode != UTM_AF_NONE && c->af_fixed && !decr;
        const bool first_is_full = af_par && !c->keep_valid;
        // deferred exact scores: this batch's delta passes log the masks of rows [iter - 1, ...) (a full pass logs
        // nothing: the row in front of it was finished when the run that made it ended)
        const bool defer = af_par && defer_active(c);
        if (defer) c->defer_lo = std::max(c->defer_lo, first_is_full ? c->iter : c->iter - 1);
        const i64 swaps0 = c->cov_swaps_enqueued;
        // where the pick runs: inside the scoring launch on the only shard (1) and on a shard of the mailbox exchange (2)
        const int fuse_mode = mailbox_exchange(c) ? 2 : (c->n_ranks == 1 && c->n_local == c->n_total && !c->comm) ? 1 : 0;
        // short scans (and any matrix whose tile grid fits the resident blocks): the whole batch as ONE persistent launch
        // An iteration that a persistent launch (interval form) scored but could not decide: the verification launch alone,
        // as a batch of its own -- the accumulators and covered are current, nothing is pending.  (Byte accounting: the
        // pass that scored it went uncounted when its launch ended; it is counted here.)
        const bool verify_only = c->loop_unresolved && af_par && !first_is_full;
        if (c->loop_unresolved) {
            if (!verify_only) return fail(UTM_ESTATE, "an undecided iteration of the persistent loop cannot be finished in this mode (internal error)");
            c->loop_unresolved = false;
            HIP_TRY(hipMemsetAsync(&c->d_st->loop_unresolved, 0, sizeof(int), c->stream));
        }
        const int loop_af = (af_par && !verify_only) ? loop_af_form(c, first_is_full) : 0;
        const LoopShape loop = (!decr && !verify_only && fuse_mode == 1 && c->tune.fuse_pick && (c->af_mode == UTM_AF_NONE || loop_af)) ? loop_shape(c, loop_af) : LoopShape();
        if (loop.ok) n = std::min<i64>(n, 256);  // (a launch's record carries an 8-bit iteration tag; UTM_BATCH may ask for more)
        if (loop.ok && loop_af == 2 && defer_active(c)) n = std::min<i64>(n, c->defer_slots);
        if (loop.ok) TRY(enqueue_loop(c, loop, (int)n, loop_af));
        for (i64 j = 0; j < n && !loop.ok; ++j) {
            bool picked = false;
            c->enq_iter = c->iter + j;  // (exact unless the loop ends first -- and then these launches do nothing)
            if (verify_only) picked = false;
            else if (decr) TRY(enqueue_score_decr(c));
            else TRY(enqueue_score(c, false, fuse_mode, &picked));
            if (!picked) TRY(enqueue_pick_and_exchange(c, decr));
            if (af_par && j == 0 && first_is_full) hipLaunchKernelGGL(k_count_sum, dim3(1), dim3(1024), 0, c->stream, c->d_st, c->d_act, c->d_cnt, 1);
            if (c->n_ranks == 1 && c->active_ub > 0) c->active_ub -= 1;  // exact while the loop is alive
        }
        if (af_par) hipLaunchKernelGGL(k_count_sum, dim3(1), dim3(1024), 0, c->stream, c->d_st, c->d_act, c->d_cnt, 0);
        const i64 before = c->iter;
        TRY(sync_state(c));
        if (loop.ok && c->h_st->xerror == 3) {
            // the census failed (not every block of the grid became resident, e.g. another process holds part of the GPU):
            // nothing was touched -- clear the mark, take the launch-per-iteration path from here on
            c->persist_off = true;
            c->persist_launches -= 1;
            c->h_st->xerror = 0;
            HIP_TRY(copy_sync(c, &c->d_st->xerror, &c->h_st->xerror, sizeof(int), hipMemcpyHostToDevice));
            if (c->flags & UTM_FLAG_PROFILE_EVENTS) c->ev_used = 0;
            continue;
        }
        if (loop.ok) c->persist_iterations += c->iter - before;
        if (loop.ok && c->h_st->loop_unresolved) {
            // the launch ended early, with an iteration scored but not decided: the next batch is its verification launch
            c->loop_unresolved = true;
            c->persist_unresolved += 1;
            n = c->iter - before;
            // ties the picker cannot settle come in runs (long twin columns): when launches keep ending after a few rows,
            // the next 4, 8, ... 64 iterations run as launches before the loop is tried again
            if (n < 4) {
                c->persist_backoff_len = std::min<i64>(64, std::max<i64>(4, c->persist_backoff_len * 2));
                c->persist_backoff = c->persist_backoff_len;
            } else if (n >= 16) {
                c->persist_backoff_len = 0;
            }
        } else if (!loop.ok && c->persist_backoff > 0 && !verify_only) {
            c->persist_backoff -= std::max<i64>(1, c->iter - before);
        }
        enq += n;
        if (rccl_needs_root(c) && c->iter > before && !c->finished) {
            // second half of the RCCL exchange: the winner's column from its owner into every shard's winner-column
            // buffer, where the next scoring pass ORs it into covered (select.py:100 on every replica)
            const int owner = c->h_st->prev_rank;
            TRY(broadcast_column(c, owner, owner == c->rank ? c->h_st->prev_gidx - (i64)c->first : -1));
        }
        // bytes: iterations that were actually scored in this batch (rows + a terminating empty pass); the
        // local selectable count falls from a0 to a1 over the batch's rows (by one per row on a single shard)
        const i64 rows = c->iter - before;
        const i64 passes = std::min<i64>(n, rows + ((c->finished && c->h_st->tot < (i64)c->n_var_total && rows < n) ? 1 : 0));
        const unsigned a1 = c->active_ub;
        {
            // folded delta passes swap the covered pair when they are ENQUEUED; the ones a finished loop skipped wrote
            // nothing, so an odd number of them leaves the roles one swap off
            const i64 swapped = c->cov_swaps_enqueued - swaps0;
            const i64 ran = std::max<i64>(0, passes - (first_is_full ? 1 : 0));
            if (swapped > ran && ((swapped - ran) & 1)) swap_covered(c);
        }
        for (i64 j = 0; j < passes; ++j) {
            const u64 drop = rows > 0 ? (u64)(a0 - a1) * (u64)std::min(j, rows) / (u64)rows : 0;
            const int kind = !af_par ? -1 : (j == 0 && first_is_full) ? UTM_PASS_AF_FULL : UTM_PASS_AF_DELTA;
            const i64 full = iteration_bytes(c, a0 - drop, kind);
            c->brute_bytes += full;
            if (!decr) c->algo_bytes += full;
        }
        if (af_par && rows > 0) {
            // AF values the delta passes of this batch gathered = how far the selectable samples' counts fell, less
            // what left with the winners themselves (4 bytes per fixed-point table entry)
            const i64 skip = first_is_full ? 1 : 0;
            std::vector<i64> gains((size_t)std::max<i64>(0, rows - skip));
            if (!gains.empty()) HIP_TRY(copy_sync(c, gains.data(), c->d_res_new + before + skip, gains.size() * 8, hipMemcpyDeviceToHost));
            i64 left_with_winners = 0;
            for (i64 g : gains) left_with_winners += g;
            const u64 prev = first_is_full ? c->h_st->cnt_sum_base : c->cnt_sum_prev;
            const i64 gathered = (i64)prev - (i64)c->h_st->cnt_sum - left_with_winners;
            if (gathered > 0) {
                c->algo_bytes += gathered * 4;
                c->brute_bytes += gathered * 4;
            }
        }
        if (af_par) c->cnt_sum_prev = c->h_st->cnt_sum;
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
        if (defer) {
            // pass j scored iteration before + j and logged row before + j - 1
            TRY(defer_finish_rows(c, c->defer_lo, before + passes - 1));
            if (!defer_active(c)) TRY(defer_flush_last(c));  // (the estimates just became exact: no later pass will log)
        }
        tail_deferred = defer;
        c->scored += passes;
        if (c->flags & UTM_FLAG_PROFILE_EVENTS) TRY(collect_event_times(c));
    }
    if (tail_deferred) TRY(defer_flush_last(c));
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
    TRY(enqueue_score(c, /*force_sequential=*/true, false, nullptr, /*by_sample=*/true));
    Scratch<i64> d_counts;
    Scratch<double> d_scores;
    HIP_TRY(d_counts.alloc(c->n_local));
    HIP_TRY(d_scores.alloc(c->n_local));
    PickArgs pa = pick_args(c);
    pa.afsum = nullptr;
    hipLaunchKernelGGL(k_final_scores, dim3((c->n_local + 255) / 256), dim3(256), 0, c->stream, pa, d_counts.p, d_scores.p);
    (void)hipMemsetAsync(c->d_cnt, 0, (size_t)c->n_local * 8, c->stream);
    (void)hipMemsetAsync(c->d_afsum, 0, (size_t)c->n_local * 8, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && counts) e = copy_sync(c, counts, d_counts.p, (size_t)c->n_local * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && scores) e = copy_sync(c, scores, d_scores.p, (size_t)c->n_local * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(UTM_EHIP, "peek: %s", hipGetErrorString(e));
    return UTM_OK;
}

extern "C" int utm_peek_estimates(utm_ctx *c, int64_t *counts, double *scores)
{
    CTX(c);
    TRY(ensure_prepared(c));
    if (c->af_mode == UTM_AF_NONE || !c->af_fixed) return fail(UTM_ESTATE, "estimates exist only for AF scores in fixed point");
    c->keep_valid = false;  // a full parallel pass, the pending winner applied
    TRY(enqueue_score(c, /*force_sequential=*/false, false, nullptr, /*by_sample=*/true));
    Scratch<i64> d_counts;
    Scratch<double> d_scores;
    HIP_TRY(d_counts.alloc(c->n_local));
    HIP_TRY(d_scores.alloc(c->n_local));
    PickArgs pa = pick_args(c);
    hipLaunchKernelGGL(k_final_scores, dim3((c->n_local + 255) / 256), dim3(256), 0, c->stream, pa, d_counts.p, d_scores.p);
    (void)hipMemsetAsync(c->d_cnt, 0, (size_t)c->n_local * 8, c->stream);
    (void)hipMemsetAsync(c->d_afsum, 0, (size_t)c->n_local * 8, c->stream);
    c->keep_valid = false;  // (the accumulators were borrowed: the next iteration re-scores in full)
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && counts) e = copy_sync(c, counts, d_counts.p, (size_t)c->n_local * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && scores) e = copy_sync(c, scores, d_scores.p, (size_t)c->n_local * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(UTM_EHIP, "peek: %s", hipGetErrorString(e));
    return UTM_OK;
}

static int flush_pending(utm_ctx *c)
{
    for (auto &ch : c->chunks)
        hipLaunchKernelGGL(k_apply_pending, dim3((unsigned)std::min<u64>(1024, (ch.wp + 255) / 256)), dim3(256), 0, c->stream,
                           ch.covered, ch.cols, ch.wp, pending_of(c, ch, false), c->d_st, 0);
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
    out->af_fixed_point = c->af_fixed ? (c->af_trunc ? 2 : 1) : 0;
    out->af_q = c->af_q;
    out->n_chunks = (int32_t)c->chunks.size();
    out->decr_iterations = c->decr_iterations;
    out->brute_force_bytes = c->brute_bytes;
    out->p2p_replica_bytes = (i64)c->replica_bytes;
    out->exchange = mailbox_exchange(c)                        ? UTM_EXCHANGE_MAILBOX
                    : c->n_local == c->n_total && !c->comm ? UTM_EXCHANGE_NONE
                    : c->comm                            ? (c->column_by_allreduce ? UTM_EXCHANGE_RCCL_SUM : UTM_EXCHANGE_RCCL)
                                                         : UTM_EXCHANGE_CALLER;
    out->af_chained_iterations = c->prepared ? (i64)c->h_st->chain_events : 0;
    out->af_deferred_rows = c->deferred_rows;
    out->persist_launches = c->persist_launches;
    out->persist_iterations = c->persist_iterations;
    out->persist_unresolved = c->persist_unresolved;
    out->af_table_passes = c->af_table_passes;
    out->rccl_ranks = 0;
    if (c->comm) {
        int n = 0;
        if (g_rccl.CommCount(c->comm, &n) == ncclSuccess) out->rccl_ranks = n;
    }
    out->decr_interleaved_bytes = c->decr_interleaved ? (i64)(c->col_words * interleaved_stride(c) * 8) : 0;
    return UTM_OK;
}

// What this GPU streams right now: `launches` plain read-only passes over the resident columns of every chunk, timed with
// the loop's own events on the context's stream.  The scoring kernels cannot go faster than this on this box today.
extern "C" int utm_stream_calibration(utm_ctx *c, int32_t launches, double *gbps_out)
{
    CTX(c);
    if (!gbps_out || launches < 1) return fail(UTM_EINVAL, "bad arguments");
    if (c->chunks.empty()) return fail(UTM_ESTATE, "no chunks");
    Scratch<u64> sink;
    HIP_TRY(sink.alloc(1));
    u64 bytes = 0;
    auto pass = [&]() {
        for (auto &ch : c->chunks) {
            const u64 n_kib = (u64)c->n_local * ch.wp * 8 / 1024;
            const unsigned grid = (unsigned)std::min<u64>(32768, std::max<u64>(1, (n_kib + 127) / 128));
            hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, c->stream, (const u64 *)ch.cols, n_kib, sink.p);
        }
    };
    pass();  // (warm-up: code object, page tables)
    HIP_TRY(hipEventRecord(c->ev_loop0, c->stream));
    for (int i = 0; i < launches; ++i) pass();
    HIP_TRY(hipEventRecord(c->ev_loop1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_loop1));
    HIP_TRY(hipGetLastError());
    for (auto &ch : c->chunks) bytes += (u64)c->n_local * ch.wp * 8;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev_loop0, c->ev_loop1));
    *gbps_out = ms > 0 ? (double)bytes * launches / (ms * 1e-3) / 1e9 : 0.0;
    return UTM_OK;
}

extern "C" int utm_set_decremental(utm_ctx *c, int32_t on, double threshold)
{
    CTX(c);
    c->decr_enabled = on != 0;
    c->decr_threshold = threshold > 0 ? threshold : 0;
    if (!on) {  // the interleaved copy doubles the matrix: hand it back with the mode
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (auto &ch : c->chunks) {
            (void)hipFree(ch.rows_t);
            ch.rows_t = nullptr;
            ch.rows_t_valid = false;
        }
        c->decr_interleaved = false;
    }
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

#ifdef UTM_DEBUG_STAMPS
extern "C" int utm_dbg_loop_stamps(utm_ctx *c, uint64_t *out /* 256 x 16 */, uint64_t *wave_t /* 2 x 8192 */)
{
    std::vector<LoopSync> h(1);
    HIP_TRY(copy_sync(c, h.data(), c->d_loop_sync, sizeof(LoopSync), hipMemcpyDeviceToHost));
    memcpy(out, h[0].stamps, sizeof h[0].stamps);
    if (wave_t) memcpy(wave_t, h[0].wave_t, sizeof h[0].wave_t);
    return UTM_OK;
}
extern "C" int utm_dbg_verify_stamps(utm_ctx *c, uint64_t *out)
{
    VerifySync h;
    HIP_TRY(copy_sync(c, &h, c->d_vsync, sizeof h, hipMemcpyDeviceToHost));
    memcpy(out, h.stamps, sizeof h.stamps);
    return UTM_OK;
}
#endif
