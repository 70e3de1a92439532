"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact bar: sample order, new_count and -- for AF modes -- the float64 score itself."""
import numpy as np
import pytest

import oracle_util as ou
from oracle_util import npo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from utmos_amd import _native as nat
    assert nat.device_count() >= 1, "no GPU visible"
    from utmos_amd import device
    return device


def make_matrix(dev, cols, n_var, **kw):
    m = dev.DeviceMatrix(cols.shape[0], **kw)
    c = m.add_chunk(n_var)
    m.upload_columns(c, cols)
    return m


def check_run(dev, dense, state=None, weights=None, af=None, k=None, chunks=None, decremental=None, estimate_scores=False,
              **kw):
    n_var, n_samp = dense.shape
    state = np.ones(n_samp, np.uint8) if state is None else state
    cols = npo.pack_columns(dense)
    exp = ou.c_greedy(cols, n_var, state, weights, af, k_max=k)
    m = dev.DeviceMatrix(n_samp, **kw)
    try:
        bounds = [0, n_var] if chunks is None else chunks
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            c = m.add_chunk(hi - lo)
            m.upload_columns(c, npo.pack_columns(dense[lo:hi]))
            if af is not None:
                m.set_af(c, af[lo:hi])
        m.set_state(state)
        m.set_weights(weights)
        if decremental is not None:
            m.set_decremental(True, decremental)
        if estimate_scores:
            m.set_af_exact_scores(False)
        got = m.run(n_samp if k is None else k)
        stats = m.stats()
    finally:
        m.close()
    assert got[0].tolist() == exp[0].tolist()
    assert got[1].tolist() == exp[1].tolist()
    if estimate_scores:
        assert np.allclose(got[2], exp[2], rtol=1e-6, atol=0)
    else:
        assert got[2].tolist() == exp[2].tolist()      # float64 scores, bit for bit
    return got, stats


def test_synth_device_equals_host(dev):
    n_var, n_samp = 5000, 77
    cols, _ = dev.synth_host(3, n_var, n_samp)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=3)
        assert (m.download_columns(c) == cols).all()
        bits = np.unpackbits(cols.view(np.uint8), axis=1, bitorder="little")[:, :n_var]
        assert (m.var_count() == bits.sum(axis=1)).all()


def test_upload_download_roundtrip_and_row_transpose(dev):
    rng = np.random.default_rng(5)
    n_var, n_samp = 1000, 203          # neither a multiple of 64 nor of 8
    dense = ou.random_dense(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    with make_matrix(dev, cols, n_var) as m:
        assert (m.download_columns(0) == cols).all()
    rows = np.packbits(dense, axis=1)   # the reference's own packing (convert.py:85)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.upload_rows_packed(c, rows)
        assert (m.download_columns(c) == cols).all()
    # a shard that starts at a sample that is not byte aligned
    with dev.DeviceMatrix(n_samp, first_sample=67, n_local=100) as m:
        c = m.add_chunk(n_var)
        m.upload_rows_packed(c, rows)
        assert (m.download_columns(c) == cols[67:167]).all()


@pytest.mark.parametrize("n_var,n_samp", [(1, 1), (63, 3), (64, 4), (65, 5), (1000, 130), (8192, 64), (8193, 257),
                                          (20000, 33)])
def test_int_mode_select_all(dev, n_var, n_samp):
    rng = np.random.default_rng(n_var + n_samp)
    check_run(dev, ou.random_dense(rng, n_var, n_samp))


def test_int_mode_large_tiles(dev):
    # enough variants for full 32-step tiles plus a ragged last tile
    rng = np.random.default_rng(11)
    n_var, n_samp = 64 * 128 * 70 + 777, 24
    dense = rng.random((n_var, n_samp)) < 0.02
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    check_run(dev, dense)


def test_exclude_used_and_k(dev):
    rng = np.random.default_rng(2)
    dense = ou.random_dense(rng, 3000, 90)
    state = np.ones(90, np.uint8)
    state[[3, 17, 40]] = 2          # excluded: never selected, never cover
    state[[5, 60]] = 0              # used from the start: cover from the first iteration
    check_run(dev, dense, state=state, k=25)


def test_weights_including_negative_and_zero(dev):
    rng = np.random.default_rng(4)
    dense = ou.random_dense(rng, 2500, 70)
    w = rng.choice([0.0, 0.25, 1.0, 3.0, 10.0], 70)
    check_run(dev, dense, weights=w)
    w2 = np.where(np.arange(70) % 2 == 0, -1.0, 1.0)
    state = np.ones(70, np.uint8)
    state[0] = 2
    check_run(dev, dense, weights=w2, state=state)
    check_run(dev, dense, weights=-np.ones(70))     # all negative, nothing masked: still selects


def test_ties_lowest_index(dev):
    dense = np.zeros((256, 8), dtype=bool)
    for s in range(8):
        dense[32 * s:32 * s + 32, s] = True        # eight identical scores every iteration
    got, _ = check_run(dev, dense)
    assert got[0].tolist() == list(range(8))


def test_zero_score_exhaustion(dev):
    # variants carried only by an excluded sample can never be captured: loop ends on score 0
    dense = np.zeros((10, 4), dtype=bool)
    dense[:4, 0] = True
    dense[2:6, 1] = True
    dense[6:, 3] = True
    dense[0, 2] = True
    state = np.array([1, 1, 1, 2], np.uint8)
    got, _ = check_run(dev, dense, state=state)
    assert got[0].tolist() == [0, 1]


@pytest.mark.parametrize("mode", ["f32", "f32_seq", "f64"])
def test_af_modes(dev, mode):
    rng = np.random.default_rng(9)
    n_var, n_samp = 9000, 140
    dense = ou.random_dense(rng, n_var, n_samp)
    af = dense.sum(axis=1) / (2.0 * n_samp)
    af[::97] = 0.0                                  # AF == 0 rows vanish from the reference's float matrix
    if mode != "f64":
        af = af.astype(np.float32)
    w = rng.choice([0.5, 1.0, 2.0], n_samp)
    _, stats = check_run(dev, dense, af=af, af_sequential=(mode == "f32_seq"))
    assert stats["af_fixed_point"] == (0 if mode == "f32_seq" else 1)     # 1 = verified-parallel scheme
    check_run(dev, dense, af=af, weights=w, af_sequential=(mode == "f32_seq"), k=40)


@pytest.mark.parametrize("switch", ["0", "2"])
def test_af_dense_and_sparse_kernels_agree(dev, switch, monkeypatch):
    """UTM_AF_SWITCH=0 runs every iteration on the sparse-phase kernel (global AF gathers), 2 keeps the
    LDS-tile kernel throughout; both must give the oracle's rows."""
    monkeypatch.setenv("UTM_AF_SWITCH", switch)
    rng = np.random.default_rng(18)
    n_var, n_samp = 64 * 128 * 9 + 321, 70
    dense = ou.random_dense(rng, n_var, n_samp)
    af = (dense.sum(axis=1) / (2.0 * n_samp)).astype(np.float32)
    check_run(dev, dense, af=af)
    check_run(dev, dense, af=af.astype(np.float64) / 3.0, chunks=[0, 8192 * 3, n_var])


@pytest.mark.parametrize("kind", ["f32", "f64"])
def test_af_table_too_wide_for_a_lossless_unit_gets_a_coarser_one(dev, kind):
    """An AF table whose mass times 2^q overflows int64 at the lossless q (a 1e-30 among ordinary values; uniformly
    random doubles over a million variants behave the same): the estimate floors the smallest values to a coarser
    unit and the verification widens its intervals by one unit per addend -- still the reference's rows and float64
    scores bit for bit, and still the parallel kernels."""
    rng = np.random.default_rng(10)
    dense = ou.random_dense(rng, 2000, 50)
    af = rng.random(2000)
    af[0] = 1e-30
    af[1:40] *= 1e-9                                  # values far below the unit, carried by many samples
    dense[:40, :25] = True
    _, stats = check_run(dev, dense, af=af.astype(np.float32) if kind == "f32" else af)
    assert stats["af_fixed_point"] == 2
    w = rng.choice([0.5, 1.0, -1.0, 3.0], 50)
    check_run(dev, dense, af=af.astype(np.float32) if kind == "f32" else af, weights=w, decremental=1.0)


@pytest.mark.parametrize("estimate", [False, True])
def test_af_coarse_unit_score_floored_to_zero_still_yields_its_row(dev, estimate):
    """A sample whose only new variant has an AF far below the (coarse) unit: its estimate is 0, its score is not --
    the loop must not take the estimate for the stop rule's `score == 0` (select.py:51), even when the caller asked
    for estimated scores."""
    dense = np.zeros((4, 2), bool)
    dense[0, 0] = True
    dense[1, 1] = True
    af = np.array([1e-30, 1.0, 0.5, 0.25])
    got, stats = check_run(dev, dense, af=af, estimate_scores=estimate)
    assert stats["af_fixed_point"] == 2 and got[0].tolist() == [1, 0] and got[2][1] == 1e-30


def test_af_sequential_kernel_on_request(dev):
    rng = np.random.default_rng(11)
    dense = ou.random_dense(rng, 3000, 40)
    af = rng.random(3000).astype(np.float32)
    _, stats = check_run(dev, dense, af=af, af_sequential=True)
    assert stats["af_fixed_point"] == 0             # one lane per sample, the reference's order: slow, bit exact


def test_af_f32_sums_beyond_exact_range_use_chains(dev):
    """float32 AFs whose running sums leave the range where float64 adds are exact (sum * 2^q >= 2^53):
    the reference's result depends on the order of its adds; candidates are re-summed sequentially."""
    rng = np.random.default_rng(16)
    n_var, n_samp = 12000, 80
    dense = rng.random((n_var, n_samp)) < 0.3
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    af = np.exp2(rng.uniform(-12, -8, n_var)).astype(np.float32)
    af[:40] = np.float32(2.0 ** -30) * (1 + rng.random(40).astype(np.float32))   # q = 53: exact only below 1.0
    got, stats = check_run(dev, dense, af=af)
    assert stats["af_fixed_point"] == 1 and got[2][0] > 1.0
    w = rng.choice([0.5, 1.0, -1.0, 3.0], n_samp)
    check_run(dev, dense, af=af, weights=w, k=30)


@pytest.mark.parametrize("kind", ["f32", "f64"])
def test_af_many_exact_ties_overflow_the_candidate_list(dev, kind):
    """100 identical columns tie in every iteration: more candidates than the chain kernel takes, so the
    sequential re-scoring of everyone decides (f64), or the exact estimate does (f32)."""
    rng = np.random.default_rng(17)
    n_var, n_samp = 5000, 100
    base = rng.random(n_var) < 0.2
    base[:10] = True
    dense = np.repeat(base[:, None], n_samp, axis=1)
    dense[rng.integers(0, n_var, 300), rng.integers(0, n_samp, 300)] ^= True     # a few differences
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    af = (rng.random(n_var) * 0.5 + 1e-3)
    check_run(dev, dense, af=af.astype(np.float32) if kind == "f32" else af, k=12)


def test_multi_chunk_equals_single_chunk(dev):
    rng = np.random.default_rng(12)
    n_var, n_samp = 30000, 60
    dense = ou.random_dense(rng, n_var, n_samp)
    af64 = dense.sum(axis=1) / (2.0 * n_samp)
    bounds = [0, 777, 9000, 9064, 30000]
    check_run(dev, dense, chunks=bounds)
    check_run(dev, dense, chunks=bounds, af=af64.astype(np.float32))
    check_run(dev, dense, chunks=bounds, af=af64)


@pytest.mark.parametrize("decremental", [True, False])
@pytest.mark.parametrize("af_kind", [None, "f32", "f64"])
def test_step_peek_and_covered(dev, af_kind, decremental):
    """step / peek_scores / covered interleaved: peeks and covered reads apply the pending winner outside the loop,
    so the persistent AF accumulators (and the decremental counts) must be rebuilt afterwards -- and, without the
    decremental mode, the deferred exact AF scores have to survive full passes turning up between delta passes."""
    rng = np.random.default_rng(13)
    n_var, n_samp = 5000, 40
    dense = ou.random_dense(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    af = None
    if af_kind:
        af = dense.sum(axis=1) / (2.0 * n_samp)
        af = af.astype(np.float32) if af_kind == "f32" else af / 3.0
    state = np.ones(n_samp, np.uint8)
    with make_matrix(dev, cols, n_var) as m:
        if af is not None:
            m.set_af(0, af)
        if decremental:
            m.set_decremental(True, 1.0)
        covered = np.zeros(cols.shape[1], np.uint64)
        for it in range(12):
            best, cnt, sc = ou.c_score(cols, n_var, state, af=af)
            if it % 2 == 0:
                counts, scores = m.peek_scores()
                assert counts.tolist() == cnt.tolist() and scores.tolist() == sc.tolist()
            if it in (5, 9):                                  # two rows in one call between the single steps
                idx2, new2, sc2 = m.run(2)
                for j in range(2):
                    best, cnt, sc = ou.c_score(cols, n_var, state, af=af)
                    assert (idx2[j], new2[j], sc2[j]) == (best, cnt[best], sc[best])
                    state[best] = 0
                    covered |= cols[best]
                continue
            got = m.step()
            assert got is not None and got[0] == best and got[1] == cnt[best] and got[2] == sc[best]
            state[best] = 0
            covered |= cols[best]
            if it % 3 == 0:
                assert (m.covered(0) == covered).all()


def test_golden_fixtures_through_the_device(dev):
    cases = ou.golden_cases()
    for name in ("select_multi", "select_exclude", "select_af", "select_af_h5", "select_weightsaf", "select_tiny"):
        case = cases[name]
        kw = ou.case_kwargs(case["args"])
        parts = [ou.load_part(n) for n in case["inputs"]]
        samples = parts[0]["samples"].astype(str)
        with dev.DeviceMatrix(len(samples)) as m:
            for p in parts:
                keep = p["GT"].any(axis=1)
                c = m.add_chunk(int(keep.sum()))
                m.upload_rows_packed(c, p["GT"][keep])
                if kw.get("af"):
                    af = p["AF"][keep]
                    m.set_af(c, af.astype(np.float32) if kw.get("af_dtype") == "f32" else af)
            var_count = m.var_count()
            m.set_state(npo.initial_state(samples, kw.get("subset"), kw.get("exclude")))
            m.set_weights(npo.weight_vector(samples, kw.get("weights")))
            k = npo.resolve_count(len(samples), kw.get("count", 0.02))
            idx, new, _ = m.run(k)
            n_var = m.shape[0]
        tot = np.cumsum(new)
        text = npo.HEADER + "".join(
            npo.format_row([samples[i], int(var_count[i]), int(n), int(t), round(t / n_var, 4)])
            for i, n, t in zip(idx, new, tot))
        assert text == ou.golden_text(case), name


def test_sharded_building_blocks_two_shards_one_gpu(dev):
    """Two contexts hold half the samples each; records + winner columns are exchanged by the test.
    Same decisions as one context holding everything."""
    rng = np.random.default_rng(14)
    n_var, n_samp = 6000, 50
    dense = ou.random_dense(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    w = rng.choice([1.0, 2.0], n_samp)
    state = np.ones(n_samp, np.uint8)
    state[7] = 2
    af = dense.sum(axis=1) / (2.0 * n_samp) / 3.0        # float64 AF: every local best needs its chain
    exp = ou.c_greedy(cols, n_var, state, w, af=af)
    shards = []
    for first, n in ((0, 21), (21, 29)):
        m = dev.DeviceMatrix(n_samp, first_sample=first, n_local=n)
        c = m.add_chunk(n_var)
        m.upload_columns(c, cols[first:first + n])
        m.set_af(c, af)
        m.set_state(state)
        m.set_weights(w)
        shards.append(m)
    got_idx, got_new, got_score = [], [], []
    for _ in range(n_samp):
        recs = [m.local_best() for m in shards]
        cand = [(-r[0], r[1], i) for i, r in enumerate(recs) if r[1] >= 0]
        if not cand:
            break
        owner = min(cand)[2]
        col = shards[owner].get_column(recs[owner][1])
        outs = [m.apply_records(recs, None if i == owner else col) for i, m in enumerate(shards)]
        assert outs[0] == outs[1]
        if outs[0] is None:
            break
        got_idx.append(outs[0][0])
        got_new.append(outs[0][1])
        got_score.append(outs[0][2])
    for m in shards:
        m.close()
    assert got_idx == exp[0].tolist() and got_new == exp[1].tolist() and got_score == exp[2].tolist()


@pytest.mark.parametrize("column", ["broadcast", "allreduce"])
@pytest.mark.parametrize("mode", ["int", "weights", "af32", "af64", "chunks", "used"])
def test_rccl_exchange_single_rank(dev, mode, column, monkeypatch):
    """north_star's RCCL protocol -- ncclAllGather of the records, k_decide, ncclBroadcast of the winner's column from
    its owner -- with a 1-rank communicator (RCCL refuses two ranks on this box's one GPU).  UTM_TEST_REMOTE_WINNER
    makes the context read every winner from the broadcast buffer, as a non-owner rank would, so the whole data path
    of a remote winner runs: collective, buffer, covered update fused into the next scoring pass."""
    monkeypatch.setenv("UTM_TEST_REMOTE_WINNER", "1")
    rng = np.random.default_rng(15)
    n_var, n_samp = 64 * 128 * 3 + 77, 70
    dense = ou.random_dense(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    state = np.ones(n_samp, np.uint8)
    state[[5, 40]] = 2
    if mode == "used":
        state[[7, 33]] = 0
    w = rng.choice([0.5, 1.0, 3.0], n_samp) if mode == "weights" else None
    af = None
    if mode.startswith("af"):
        af = (dense.sum(axis=1) / (2.0 * n_samp))
        af = af.astype(np.float32) if mode == "af32" else af / 3.0
    exp = ou.c_greedy(cols, n_var, state, w, af)
    m = dev.DeviceMatrix(n_samp)
    bounds = [(0, 10000), (10000, n_var)] if mode == "chunks" else [(0, n_var)]
    for lo, hi in bounds:
        c = m.add_chunk(hi - lo)
        sub = npo.pack_columns(dense[lo:hi])
        m.upload_columns(c, sub)
        if af is not None:
            m.set_af(c, af[lo:hi])
    with m:
        m.comm_init(0, 1, dev.DeviceMatrix.comm_unique_id())
        assert m.allreduce_max(3.5) == 3.5
        if column == "allreduce":       # root-free variant: the column by ncclAllReduce(sum) of owner's-column-else-zeros
            m.comm_column_by_allreduce(True)
        m.set_state(state)
        m.set_weights(w)
        got = m.run(n_samp)
        st = m.stats()
        assert m.exchange() == ("rccl" if column == "broadcast" else "rccl-allreduce") and st["rccl_ranks"] == 1
    assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist() and got[2].tolist() == exp[2].tolist()


def _gpu_p2p_worker(rank, world, port, q):
    from utmos_amd.sharded import SocketTransport, shard_bounds, sharded_greedy
    transport = SocketTransport(rank, world, port=port)
    try:
        from utmos_amd import device
        n_var, n_samp = 64 * 128 * 5 + 17, 64
        cols, af = device.synth_host(6, n_var, n_samp)
        first, n_local = shard_bounds(n_samp, rank, world)
        with device.DeviceMatrix(n_samp, device=0, first_sample=first, n_local=n_local) as m:
            for lo, hi in ((0, 20000), (20000, n_var)):          # two chunks: two mappings per peer
                c = m.add_chunk(hi - lo)
                m.synth_fill(c, seed=6, first_var_global=lo)
                m.set_af(c, af[lo:hi].astype(np.float64) / 3.0)
            m.p2p_import(rank, transport.allgather_bytes(m.p2p_export()))     # hipIpc mappings only: the host drives the loop
            on = m.p2p
            m.reset()
            got = list(sharded_greedy(m, transport, n_samp))
        exp = ou.c_greedy(cols, n_var, np.ones(n_samp, np.uint8), af=af.astype(np.float64) / 3.0)
        idx_ok = [g[0] for g in got] == exp[0].tolist()
        score_ok = [g[2] for g in got] == exp[2].tolist()
        first_bad = next((i for i, (g, e) in enumerate(zip(got, exp[0].tolist())) if g[0] != e), -1)
        q.put((rank, on and idx_ok and score_ok, len(got), f"p2p={on} idx_ok={idx_ok} score_ok={score_ok} first_bad={first_bad} exp_len={len(exp[0])}"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, False, 0, repr(e)))
    finally:
        transport.close()


def test_p2p_winner_columns_read_in_place_across_processes(dev):
    """Shards map each other's columns with hipIpc: the winner's column is never copied between them."""
    import multiprocessing as mp
    import os
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gpu_p2p_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res) and res[0][2] == res[1][2] == res[2][2] > 30, res


def _gpu_fused_worker(rank, world, port, q):
    import os
    from utmos_amd.sharded import SocketTransport, connect_shards, shard_bounds
    transport = SocketTransport(rank, world, port=port)
    try:
        from utmos_amd import device
        n_var, n_samp = 64 * 128 * 7 + 5, 90
        cols, af = device.synth_host(8, n_var, n_samp)
        state = np.ones(n_samp, np.uint8)
        state[[3, 60]] = 2
        w = np.where(np.arange(n_samp) % 5 == 0, 2.0, 1.0)
        used_state = state.copy()
        used_state[[1, 44, 89]] = 0                      # already used samples on every shard: they cover everywhere
        first, n_local = shard_bounds(n_samp, rank, world)
        out = {}
        with device.DeviceMatrix(n_samp, device=0, first_sample=first, n_local=n_local) as m:
            c = m.add_chunk(n_var)
            m.synth_fill(c, seed=8)
            how = connect_shards(m, transport, None, "mailboxes")
            fused = m.fused and how == "mailboxes" and m.exchange() == "mailboxes"
            replica = m.stats()["p2p_replica_bytes"]
            for mode in ("int", "af32", "af64", "decr", "used"):
                m.set_af(c, None if mode in ("int", "decr", "used") else (af if mode == "af32" else af.astype(np.float64) / 3.0))
                m.set_decremental(mode == "decr", 1.0)
                m.set_state(used_state if mode == "used" else state)
                m.set_weights(w)
                idx, new, score = m.run(n_samp)            # collective: every shard gets every row
                out[mode] = (idx.tolist(), new.tolist(), score.tolist())
            transport.allgather((0.0, 0, 0))               # nobody unmaps while a peer may still read
        ok = fused
        for mode, got in out.items():
            a = None if mode in ("int", "decr", "used") else (af if mode == "af32" else af.astype(np.float64) / 3.0)
            exp = ou.c_greedy(cols, n_var, used_state if mode == "used" else state, w, af=a)
            ok = ok and got[0] == exp[0].tolist() and got[1] == exp[1].tolist() and got[2] == exp[2].tolist()
        ok = ok and (replica > 0) == (os.environ.get("UTM_P2P_REPLICATE", "1") != "0")
        q.put((rank, ok, len(out["int"][0]), f"fused={fused} replica={replica}"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, False, 0, repr(e)))
    finally:
        transport.close()


@pytest.mark.parametrize("columns", ["replicated", "in-place"])
def test_fused_device_side_exchange_three_processes(dev, columns, monkeypatch):
    """The default multi-shard loop: records through hipIpc-mapped mailboxes, winner columns read from the one-time
    local copy of the peers' columns (or in place through the mappings, what a matrix too large to replicate runs),
    utm_run collective over three processes (sharing the box's one GPU); samples that start out used on another shard
    cover on every shard."""
    import multiprocessing as mp
    import os
    monkeypatch.setenv("UTM_P2P_REPLICATE", "0" if columns == "in-place" else "1")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 39500 + os.getpid() % 2000 + {"replicated": 0, "in-place": 2000}[columns]
    procs = [ctx.Process(target=_gpu_fused_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res) and res[0][2] == res[1][2] == res[2][2] > 40, res


def _gpu_shard_worker(rank, world, port, q):
    # no torch in a process that runs libutmos_hip.so: its wheel carries another HIP runtime
    from utmos_amd.sharded import SocketTransport
    transport = SocketTransport(rank, world, port=port)
    try:
        from utmos_amd import device
        from utmos_amd.sharded import shard_bounds, sharded_greedy
        n_var, n_samp = 40_000, 96
        cols, af = device.synth_host(5, n_var, n_samp)
        state = np.ones(n_samp, np.uint8)
        state[11] = 2
        first, n_local = shard_bounds(n_samp, rank, world)
        with device.DeviceMatrix(n_samp, device=0, first_sample=first, n_local=n_local) as m:
            c = m.add_chunk(n_var)
            m.synth_fill(c, seed=5)              # every rank generates its own columns of the same matrix
            m.set_af(c, af)
            m.set_state(state)
            m.reset()
            got = list(sharded_greedy(m, transport, n_samp))
        exp = ou.c_greedy(cols, n_var, state, af=af)
        ok = [g[0] for g in got] == exp[0].tolist() and [g[2] for g in got] == exp[2].tolist()
        q.put((rank, ok, len(got)))
    finally:
        transport.close()


def test_two_processes_share_the_gpu_host_staged_exchange(dev):
    """One process per shard (as on a multi-GPU node), here both on the single GPU of the box:
    HIP kernels for the local scoring, TCP sockets for the exchange (the gloo variant of the same
    protocol runs on CPU in tests/test_host_logic.py)."""
    import multiprocessing as mp
    import os
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gpu_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res) and res[0][2] == res[1][2] > 50


@pytest.mark.parametrize("layout", ["interleaved", "gather"])
@pytest.mark.parametrize("mode", ["int", "weights", "af32", "af64", "chunks"])
def test_decremental_scoring_gives_the_same_rows(dev, mode, layout, monkeypatch):
    """later iterations only subtract what the last winner newly covered; same rows, and the
    device really ran decremental iterations -- streaming the word-interleaved copy, or gathering from the
    columns (what a context without room for the copy runs)."""
    monkeypatch.setenv("UTM_DECR_INTERLEAVED", "0" if layout == "gather" else "1")
    rng = np.random.default_rng(30)
    n_var, n_samp = 64 * 128 * 6 + 99, 150
    dense = ou.random_dense(rng, n_var, n_samp)
    kw = {}
    if mode == "weights":
        kw["weights"] = rng.choice([0.5, 1.0, 2.0, -1.0], n_samp)
    if mode == "af32":
        kw["af"] = (dense.sum(axis=1) / (2.0 * n_samp)).astype(np.float32)
    if mode == "af64":
        kw["af"] = dense.sum(axis=1) / (2.0 * n_samp) / 3.0
    if mode == "chunks":
        kw["chunks"] = [0, 8192, 20000, n_var]
    state = np.ones(n_samp, np.uint8)
    state[[1, 50]] = 2
    _, stats = check_run(dev, dense, state=state, decremental=1.0, **kw)   # threshold 1.0: decremental from the 2nd batch on
    assert stats["decr_iterations"] > 0
    assert stats["algo_bytes"] < stats["brute_force_bytes"]
    assert (stats["decr_interleaved_bytes"] > 0) == (layout != "gather")


def test_decremental_survives_peek_and_covered_reads(dev):
    rng = np.random.default_rng(31)
    n_var, n_samp = 30000, 90
    dense = ou.random_dense(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    exp = ou.c_greedy(cols, n_var, np.ones(n_samp, np.uint8))
    with make_matrix(dev, cols, n_var) as m:
        m.set_decremental(True, 1.0)
        got_idx = []
        for chunk in (70, 3, 17):
            idx, _, _ = m.run(chunk)
            got_idx += idx.tolist()
            m.peek_scores()            # applies the pending winner outside the loop: next pass must be a full one
            m.covered(0)
        assert got_idx == exp[0][:len(got_idx)].tolist()


def test_parity_at_production_tile_sizes(dev):
    """2M variants x 2,504 samples (the BASELINE sample count, full 32 KiB tiles, ~24k workgroups per
    launch): first iterations of the integer, float32-AF and float64-AF loops against the OpenMP C oracle on the
    matrix downloaded from the device."""
    n_var, n_samp, k = 2_000_000, 2504, 24
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=9)
        cols = m.download_columns(c)
        _, af = dev.synth_host(9, n_var, n_samp, want_cols=False)
        state = np.ones(n_samp, np.uint8)
        state[100:110] = 2
        m.set_state(state)
        got = m.run(k)
        exp = ou.c_greedy(cols, n_var, state, k_max=k, omp=True)
        assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist()
        m.set_af(c, af)
        m.reset()
        got = m.run(k)
        exp = ou.c_greedy(cols, n_var, state, af=af, k_max=k, omp=True)
        assert got[0].tolist() == exp[0].tolist() and got[2].tolist() == exp[2].tolist()
        # ... and of the float64-AF loop (the reference's in-memory values): LDS-tile delta passes first, then the
        # streaming ones; verification in one launch; every reported score finished after its batch -- bit for bit
        af64 = af.astype(np.float64) / 3.0
        m.set_af(c, af64)
        m.reset()
        got = m.run(k)
        exp = ou.c_greedy(cols, n_var, state, af=af64, k_max=k, omp=True)
        assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist() and got[2].tolist() == exp[2].tolist()
        assert m.stats()["af_deferred_rows"] > 0


def test_parity_with_many_samples(dev):
    """100,000 samples (configs[3]'s sample count) x 150k variants: many groups per tile, a 100k-entry active list,
    25 pick rounds per thread -- first iterations of the integer, weighted and decremental loops against the OpenMP
    C oracle on the matrix downloaded from the device."""
    n_var, n_samp, k = 150_000, 100_000, 12
    rng = np.random.default_rng(77)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=3)
        cols = m.download_columns(c)
        state = np.ones(n_samp, np.uint8)
        state[rng.choice(n_samp, 500, replace=False)] = 2
        m.set_state(state)
        got = m.run(k)
        exp = ou.c_greedy(cols, n_var, state, k_max=k, omp=True)
        assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist()
        w = rng.choice([0.5, 1.0, 1.0, 2.0], n_samp)
        m.set_weights(w)
        m.set_decremental(True, 1.0)
        m.reset()
        got = m.run(k)
        exp = ou.c_greedy(cols, n_var, state, w, k_max=k, omp=True)
        assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist()
        assert m.stats()["decr_iterations"] > 0


@pytest.mark.parametrize("decremental", [False, True])
def test_full_size_select_all_invariants(dev, decremental):
    """BASELINE configs[1] in full (10M x 2,504, select all): size-independent properties of a greedy
    maximum-coverage run -- every sample exactly once, gains never grow, everything captured exactly at
    the end, per-sample gain bounded by its carrier total -- and brute force == decremental."""
    n_var, n_samp = 10_000_000, 2504
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=0)
        var_count = m.var_count()
        if decremental:
            m.set_decremental(True)
        idx, new, score = m.run(n_samp)
        assert len(idx) == n_samp and sorted(idx.tolist()) == list(range(n_samp))
        assert (np.diff(new) <= 0).all()                    # submodularity: marginal gains are non-increasing
        assert int(new.sum()) == n_var                      # every variant has a carrier, all get captured
        assert (new <= var_count[idx]).all() and new[0] == var_count.max()
        assert (score == new).all()
        test_full_size_select_all_invariants.rows = getattr(test_full_size_select_all_invariants, "rows", {})
        test_full_size_select_all_invariants.rows[decremental] = (idx.tolist(), new.tolist())
    rows = test_full_size_select_all_invariants.rows
    if len(rows) == 2:
        assert rows[False] == rows[True]


@pytest.mark.parametrize("af_dtype", ["f32", "f64"])
def test_full_size_af_select_all_invariants(dev, af_dtype):
    """BASELINE configs[2] in full (10M x 2,504 with AF weighting, select all): the winners' scores never grow
    (submodular), every sample exactly once, all variants captured, a winner's score is at most half its gain
    (AF <= 0.5) -- and the brute-force loop, the decremental loop and the estimate-score mode give the same rows."""
    n_var, n_samp = 10_000_000, 2504
    _, af = dev.synth_host(0, n_var, n_samp, want_cols=False)
    af = af if af_dtype == "f32" else af.astype(np.float64) / 3.0
    runs = {}
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=0)
        m.set_af(c, af)
        for mode in ("brute", "decremental", "estimate"):
            m.set_decremental(mode != "brute")
            m.set_af_exact_scores(mode != "estimate")
            m.reset()
            idx, new, score = m.run(n_samp)
            assert len(idx) == n_samp and sorted(idx.tolist()) == list(range(n_samp))
            assert int(new.sum()) == n_var
            assert (score <= 0.5 * new + 1e-9).all() and (score > 0).all()
            if mode != "estimate":
                assert (np.diff(score) <= 0).all()           # exact float64 scores: non-increasing, ties allowed
            else:
                assert np.allclose(score, runs["brute"][2], rtol=1e-6, atol=0)
            runs[mode] = (idx.tolist(), new.tolist(), score)
    assert runs["brute"][:2] == runs["decremental"][:2] == runs["estimate"][:2]
    assert (runs["brute"][2] == runs["decremental"][2]).all()   # the float64 scores too, bit for bit


def test_randomised_configurations_against_the_oracle(dev, monkeypatch):
    """Seeded sweep over shapes, chunkings, sample states, weights, AF modes and the decremental switch:
    every configuration must give the oracle's indices, counts and float64 scores."""
    import os
    rng = np.random.default_rng(int(os.environ.get("UTM_FUZZ_SEED", "2026")))
    for trial in range(int(os.environ.get("UTM_FUZZ_TRIALS", "70"))):
        big = os.environ.get("UTM_FUZZ_BIG") == "1"      # occasional campaign with multi-tile shapes
        n_var = int(rng.choice([20000, 70000, 150000, 300000] if big else [1, 5, 63, 64, 65, 127, 500, 1500, 4000, 9000]))
        n_samp = int(rng.choice([40, 130, 300] if big else [1, 2, 3, 7, 33, 64, 65, 130, 257]))
        density = float(rng.choice([0.01, 0.05, 0.3, 0.7]))
        dense = rng.random((n_var, n_samp)) < density
        if rng.random() < 0.7:
            dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True   # mostly informative rows, sometimes not
        if rng.random() < 0.3 and n_samp > 3:
            dense[:, 1] = dense[:, 0]                                           # exact ties
        state = np.ones(n_samp, np.uint8)
        if n_samp > 4:
            state[rng.choice(n_samp, rng.integers(0, 3), replace=False)] = 2
            state[rng.choice(n_samp, rng.integers(0, 2), replace=False)] = 0
        weights = None
        if rng.random() < 0.4:
            weights = rng.choice([-2.0, 0.0, 0.5, 1.0, 1.0, 3.0], n_samp)
        af = None
        mode = rng.choice(["none", "none", "f32", "f64"])
        if mode != "none":
            af = rng.random(n_var) * rng.choice([1.0, 1e-3])
            af[rng.random(n_var) < 0.05] = 0.0
            if rng.random() < 0.25:                                             # too wide for a lossless unit: coarse one
                af[rng.random(n_var) < 0.3] *= 1e-25
            af = af.astype(np.float32) if mode == "f32" else af
        chunks = None
        if n_var > 200 and rng.random() < 0.5:
            cuts = sorted(set(int(x) for x in rng.integers(1, n_var, rng.integers(1, 4))))
            chunks = [0] + cuts + [n_var]
        k = int(rng.integers(1, n_samp + 1))
        decr = 1.0 if rng.random() < 0.4 else None
        gather = bool(rng.random() < 0.4)
        monkeypatch.setenv("UTM_DECR_INTERLEAVED", "0" if gather else "1")
        est = mode != "none" and rng.random() < 0.3
        try:
            check_run(dev, dense, state=state, weights=weights, af=af, k=k, chunks=chunks, decremental=decr,
                      estimate_scores=est)
        except AssertionError as e:
            raise AssertionError(f"trial {trial}: n_var={n_var} n_samp={n_samp} density={density} mode={mode} "
                                 f"chunks={chunks} k={k} decr={decr} gather={gather} weights={'yes' if weights is not None else 'no'}") from e


def test_generator_and_scoring_do_not_depend_on_the_chunk_layout_at_large_sizes(dev):
    """20M variants x 15,000 samples (37.5 GB; words x samples x 256 exceeds 2^32, HIP's per-dimension thread limit,
    which once truncated the generator's launch): one chunk and three chunks must hold the same matrix and give
    the same rows."""
    n_var, n_samp = 20_000_000, 15_000
    out = []
    for bounds in ([0, n_var], [0, 6_400_000, 13_000_064, n_var]):
        with dev.DeviceMatrix(n_samp) as m:
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                m.synth_fill(m.add_chunk(hi - lo), seed=4, first_var_global=lo)
            vc = m.var_count()
            idx, new, _ = m.run(5)
            out.append((vc, idx.tolist(), new.tolist()))
    assert (out[0][0] == out[1][0]).all() and out[0][1:] == out[1][1:]
    assert out[0][0].min() > 0 and out[0][2][0] == out[0][0].max()      # every sample has data; first gain = largest column
    assert abs(out[0][0].sum() / (n_var * n_samp) - 0.11) < 0.03           # the generator's mean density (14 octaves)


def test_cfg5_full_size_chunked_matrix(dev):
    """BASELINE configs[4] at its real size: 500M variants x 2,504 samples, 156 GB of HBM in ten 50M-variant chunks.
    No oracle finishes at this size, so: the domain's invariants on the first 24 rows (distinct samples, gains never
    increase, the first gain is the largest column), brute force == decremental scoring, and the same rows when
    the same matrix is cut into four chunks instead of ten (the chunk layout is policy only, select.py:56-63)."""
    free, total = dev.nat.device_memory(0)
    if total < 200e9:
        pytest.skip("needs an MI355X-sized HBM (156 GB matrix)")
    n_var, n_samp, k = 500_000_000, 2504, 24

    def build(chunk_vars):
        m = dev.DeviceMatrix(n_samp)
        v0 = 0
        while v0 < n_var:
            nv = min(chunk_vars, n_var - v0)
            m.synth_fill(m.add_chunk(nv), seed=0, first_var_global=v0)
            v0 += nv
        return m

    with build(50_000_000) as m:
        assert m.stats()["n_chunks"] == 10
        vc = m.var_count()
        a = m.run(k)
        m.set_decremental(True, 1.0)
        m.reset()
        b = m.run(k)
        assert m.stats()["decr_iterations"] > 0
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    assert len(set(a[0].tolist())) == k and (np.diff(a[1]) <= 0).all()
    assert a[1][0] == vc.max() and a[0][0] == int(np.argmax(vc))
    with build(125_000_000) as m:
        c = m.run(k)
    assert (a[0] == c[0]).all() and (a[1] == c[1]).all()


@pytest.mark.parametrize("kind", ["f32_beyond_exact", "f64"])
def test_af_estimate_scores_mode_keeps_the_rows(dev, kind):
    """utm_set_af_exact_scores(0): unambiguous winners are not chained -- indices and counts stay the oracle's, the
    reported scores are estimates within the stated bound."""
    rng = np.random.default_rng(40)
    n_var, n_samp = 12000, 80
    dense = rng.random((n_var, n_samp)) < 0.3
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    dense[:, 1] = dense[:, 0]                                  # an exact tie: must still be resolved like the reference
    if kind == "f64":
        af = rng.random(n_var) * 0.5 + 1e-3
    else:
        af = np.exp2(rng.uniform(-12, -8, n_var)).astype(np.float32)
        af[:40] = np.float32(2.0 ** -30)
    cols = npo.pack_columns(dense)
    state = np.ones(n_samp, np.uint8)
    exp = ou.c_greedy(cols, n_var, state, af=af)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.upload_columns(c, cols)
        m.set_af(c, af)
        m.set_af_exact_scores(False)
        idx, new, score = m.run(n_samp)
    assert idx.tolist() == exp[0].tolist() and new.tolist() == exp[1].tolist()
    assert np.allclose(score, exp[2], rtol=1e-6, atol=0)


@pytest.mark.parametrize("kind", ["uniform", "tiny_mixed", "f32_grid", "coarse_grid", "long"])
def test_parallel_chain_reproduces_sequential_float64_sums(dev, kind):
    """k_chain's parallel form (parity-state scan, af_verify.hip.h) against plain sequential float64 addition in
    ascending variant order -- the reference's `scores += row` (select.py:40) -- on value patterns that stress it:
    full 53-bit mantissas, values far below the sum's last bit, float32 grids where every other addend is an exact
    tie, and sums that cross many binades."""
    rng = np.random.default_rng({"uniform": 1, "tiny_mixed": 2, "f32_grid": 3, "coarse_grid": 4, "long": 5}[kind])
    for trial in range(6 if kind == "long" else 40):
        # (short lists too: a sum crosses a binade every few addends at the start, and the addend that crosses is
        # where a window ends -- once a source of error, when the sum in front of it was taken from an inexact total)
        n_var = (int(rng.integers(200, 9000)) if trial % 2 else int(rng.integers(2, 80))) if kind != "long" else 600_000
        if kind in ("uniform", "long"):
            af = rng.random(n_var)
        elif kind == "tiny_mixed":
            af = rng.random(n_var) * 1e-3
            af[rng.random(n_var) < 0.3] *= 1e-25
        elif kind == "f32_grid":
            af = rng.random(n_var).astype(np.float32).astype(np.float64)
        else:
            af = rng.integers(1, 2 ** 24, n_var) * 2.0 ** -30
        af = np.maximum(af, 1e-300)
        dense = np.zeros((n_var, 3), dtype=bool)
        dense[:, 0] = rng.random(n_var) < (0.9 if kind != "long" else 0.97)
        dense[:, 1] = rng.random(n_var) < 0.2
        dense[~dense.any(axis=1), 2] = True
        with dev.DeviceMatrix(3) as m:
            c = m.add_chunk(n_var)
            m.upload_columns(c, npo.pack_columns(dense))
            m.set_af(c, af)
            got = m.step()
        assert got is not None, (kind, trial)
        want = 0.0
        for a in af[dense[:, got[0]]]:               # whichever sample won: its score is its own sequential sum
            want = want + float(a)
        assert float(got[2]).hex() == float(want).hex(), (kind, trial, n_var)


@pytest.mark.parametrize("kind", ["f64", "f64_weights", "f32_beyond_exact", "f64_chunks"])
def test_af_scores_finished_after_their_batch_are_the_reference_sums(dev, kind):
    """Exact AF scores of unambiguous winners come from the log of newly-covered masks, one set of launches per batch
    (af_defer.hip.h): however the run is cut into calls -- one call, single steps, uneven pieces that straddle the log's
    64 slots -- every row's float64 score is the oracle's sequential sum, bit for bit, and the rows are the same."""
    rng = np.random.default_rng(77)
    n_var, n_samp = 60_000, 300
    dense = ou.random_dense(rng, n_var, n_samp, density=0.02)
    # distinct frequencies: nearly every winner is unambiguous, so nearly every score takes the deferred path
    af = rng.uniform(1e-4, 0.5, size=n_var)
    weights = None
    bounds = [0, n_var]
    if kind == "f64_weights":
        weights = rng.uniform(0.5, 2.0, size=n_samp)
    if kind == "f32_beyond_exact":
        # a wide exponent range: the unit is that of the smallest value, and sums leave the range where float64 adds
        # are exact (>= 2^53 units) -- estimates are inexact until enough is covered
        af = np.exp2(rng.uniform(-12, -1, n_var)).astype(np.float32)
        af[:40] = np.float32(2.0 ** -30) * (1 + rng.random(40).astype(np.float32))
    if kind == "f64_chunks":
        bounds = [0, 4097, 30_000, n_var]
    cols = npo.pack_columns(dense)
    state = np.ones(n_samp, np.uint8)
    exp = ou.c_greedy(cols, n_var, state, weights, af)

    def fresh():
        m = dev.DeviceMatrix(n_samp)
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            c = m.add_chunk(hi - lo)
            m.upload_columns(c, npo.pack_columns(dense[lo:hi]))
            m.set_af(c, af[lo:hi])
        m.set_state(state)
        m.set_weights(weights)
        return m

    def same(idx, new, score, upto):
        assert list(idx) == exp[0][:upto].tolist() and list(new) == exp[1][:upto].tolist()
        assert list(score) == exp[2][:upto].tolist()

    with fresh() as m:                                   # one call
        got = m.run(n_samp)
        same(got[0], got[1], got[2], len(exp[0]))
        st = m.stats()
        assert st["af_fixed_point"] in (1, 2)
        if kind == "f32_beyond_exact":
            assert st["af_deferred_rows"] > 0, st
        else:
            assert st["af_deferred_rows"] > 0.5 * len(exp[0]), st
            assert st["af_chained_iterations"] < 0.5 * len(exp[0]), st
    with fresh() as m:                                   # uneven pieces: 1, 63, 64, 65, 7, the rest
        idx, new, score = [], [], []
        for piece in (1, 63, 64, 65, 7, n_samp):
            got = m.run(piece)
            idx += got[0].tolist(); new += got[1].tolist(); score += got[2].tolist()
            same(idx, new, score, len(idx))             # (every call returns final scores for its own rows)
        assert len(idx) == len(exp[0])
    with fresh() as m:                                   # single steps
        for it in range(40):
            i, n, s = m.step()
            assert (i, n, s) == (exp[0][it], exp[1][it], exp[2][it])


@pytest.mark.parametrize("batch", ["128", "7"])
def test_af_deferred_scores_with_a_batch_longer_than_the_mask_log(dev, batch, monkeypatch):
    """UTM_BATCH beyond the 64 slots of the newly-covered-mask log: the AF loop clamps its batch to the
    log, so every deferred float64 score is still the oracle's -- and a short odd batch works too."""
    monkeypatch.setenv("UTM_BATCH", batch)
    rng = np.random.default_rng(5)
    n_var, n_samp = 50_000, 400
    dense = ou.random_dense(rng, n_var, n_samp, density=0.02)
    af = rng.uniform(1e-4, 0.5, size=n_var)
    cols = npo.pack_columns(dense)
    state = np.ones(n_samp, np.uint8)
    exp = ou.c_greedy(cols, n_var, state, None, af)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.upload_columns(c, cols)
        m.set_af(c, af)
        got = m.run(n_samp)
        assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist() and got[2].tolist() == exp[2].tolist()
        assert m.stats()["af_deferred_rows"] > 0.5 * len(exp[0])


def _af_run_worker(seed, env, q):
    """One whole float64-AF selection in a process of its own (spawned): (seed, rows, scores as hex)."""
    import os
    os.environ.update(env)
    import numpy as np
    import oracle_util as ou
    from oracle_util import npo
    from utmos_amd import device
    rng = np.random.default_rng(seed)
    n_var, n_samp = 200_000, 260
    dense = ou.random_dense(rng, n_var, n_samp)
    af = np.round(rng.uniform(1, 40, size=n_var)) / (2.0 * n_samp)      # few distinct values: near-ties, chains on the spot
    cols = npo.pack_columns(dense)
    try:
        with device.DeviceMatrix(n_samp) as m:
            c = m.add_chunk(n_var)
            m.upload_columns(c, cols)
            m.set_af(c, af)
            got = m.run(n_samp)
            st = m.stats()
        exp = ou.c_greedy(cols, n_var, np.ones(n_samp, np.uint8), None, af)
        ok = got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist() and got[2].tolist() == exp[2].tolist()
        q.put((seed, ok, st["af_chained_iterations"], st["af_deferred_rows"], None))
    except Exception as e:      # noqa: BLE001 -- reported to the parent, which fails the test
        q.put((seed, False, -1, -1, repr(e)))


@pytest.mark.parametrize("env", [{}, {"UTM_AF_VERIFY": "0"}, {"UTM_AF_DEFER": "0"}, {"UTM_AF_VERIFY": "0", "UTM_AF_DEFER": "0"}],
                         ids=["default", "three-launch verification", "chained on the spot", "both off"])
def test_af_verification_forms_agree_while_processes_share_the_gpu(dev, env):
    """The one-launch verification (k_verify) has workgroups waiting for other workgroups of the same launch: three
    processes run such loops on the one GPU at the same time (their launches interleave on the CUs) and every one of
    them still reproduces the oracle bit for bit -- as do the separate-launch and chained-on-the-spot forms that
    shards of a multi-GPU run use (selected here through the library's environment switches)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_af_run_worker, args=(100 + r, env, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for seed, ok, chained, deferred, err in res:
        assert err is None and ok, (seed, err)
        assert chained > 0                                   # the data does force chains
        assert (deferred > 0) == (env.get("UTM_AF_DEFER") != "0")


def test_af_tie_groups_are_chained_once(dev):
    """Samples with equally many private variants of one and the same frequency tie at the top, one group after the
    other (what the tail of a real run looks like).  A chain's float64 sum stays on record for as long as the sample's
    count does not change, so each group costs ONE chained iteration, not one per member -- and the rows and scores
    are the oracle's all the same."""
    n_groups, per_group = 12, 5
    n_samp = n_groups * per_group
    sizes = [400 - 30 * (i // per_group) for i in range(n_samp)]          # group g: 400 - 30 g private variants each
    n_shared = 500
    n_var = n_shared + sum(sizes)
    rng = np.random.default_rng(5)
    dense = np.zeros((n_var, n_samp), bool)
    dense[:n_shared] = rng.random((n_shared, n_samp)) < 0.5               # shared variants: gone after a few picks
    at = n_shared
    for i, k in enumerate(sizes):
        dense[at:at + k, i] = True
        at += k
    order = rng.permutation(n_var)                                         # private blocks interleaved along the variant axis
    dense = dense[order]
    af = np.full(n_var, 1.0 / 5008.0)                                      # float64, not a float32 value: estimates are never exact
    af[order < n_shared] = rng.uniform(0.01, 0.4, size=int((order < n_shared).sum()))
    got, st = check_run(dev, dense, af=af)
    assert len(got[0]) == n_samp
    assert st["af_chained_iterations"] <= n_groups + 8, st               # one per group (+ the shared phase's near-ties)
