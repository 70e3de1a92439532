"""`utmos select` end to end on the GPU: every golden TSV of the reference's suite, byte for byte,
through the CLI (repo_utils/utmos_ssshtests.sh:81-235 with .npz re-encodings of the same inputs)."""
import os

import pytest

import oracle_util as ou

pytestmark = pytest.mark.gpu
CASES = ou.golden_cases()


def run_cli(argv):
    from utmos_amd.select import select_main
    select_main(argv)


def cli_args(case, tmp_path, extra=()):
    argv = []
    skip = False
    for a in case["args"]:
        if skip:
            skip = False
            continue
        if a == "--af-dtype":        # expressed through --lowmem in the CLI, like the reference's hdf5 path
            skip = True
            continue
        if a in ("weights.txt", "subset.txt", "exclude.txt"):
            a = os.path.join(ou.GOLD, a)
        argv.append(a)
    out = str(tmp_path / "out.txt")
    inputs = [os.path.join(ou.GOLD, n + ".npz") for n in case["inputs"]]
    return argv + list(extra) + ["-o", out] + inputs, out


@pytest.mark.parametrize("name", sorted(n for n in CASES if n != "select_af_h5"))
def test_cli_reproduces_golden(name, tmp_path):
    argv, out = cli_args(CASES[name], tmp_path)
    run_cli(argv)
    assert open(out).read() == ou.golden_text(CASES[name])


@pytest.mark.parametrize("name", ["select_first", "select_multi", "select_af", "select_weightsaf"])
def test_cli_forced_chunking_is_result_neutral(name, tmp_path):
    # --maxmem 0 is the reference's hook for its chunked/hdf5 branch (select.py:18-19); here it forces
    # many small HBM chunks
    argv, out = cli_args(CASES[name], tmp_path, extra=["--maxmem", "0", "--buffer", "256"])
    run_cli(argv)
    assert open(out).read() == ou.golden_text(CASES[name])


def test_input_files_are_merged_into_chunks_up_to_maxmem(monkeypatch):
    """Three input parts -> one chunk (every chunk costs launches in every iteration); a cap smaller than two parts
    -> one chunk per part; --maxmem 0 -> `buffer`-sized chunks.  The matrix (var_count, variant total) is the same."""
    from utmos_amd import select
    files = [os.path.join(ou.GOLD, n + ".npz") for n in ("chunk0", "chunk1", "chunk2")]
    shapes = {}
    for label, maxmem, buffer in (("merged", 2, 32768), ("forced", 0, 256)):
        monkeypatch.setattr(select, "MAXMEM", maxmem)
        data = select.load_files(files, None, buffer, True)
        shapes[label] = (list(data["data"].chunk_vars), data["var_count"].tolist(), data["data"].shape)
        data["data"].close()
    assert len(shapes["merged"][0]) == 1 and len(shapes["forced"][0]) > 6
    assert sum(shapes["merged"][0]) == sum(shapes["forced"][0])
    assert shapes["merged"][1:] == shapes["forced"][1:]
    one_part = max(ou.load_part(n)["GT"].shape[0] for n in ("chunk0", "chunk1", "chunk2"))
    monkeypatch.setattr(select, "is_memsafe", lambda shape, with_af=False: shape[0] <= one_part)     # cap: one part
    monkeypatch.setattr(select, "MAXMEM", 2)
    data = select.load_files(files, None, 32768, False)
    assert len(data["data"].chunk_vars) == 3 and sum(data["data"].chunk_vars) == sum(shapes["merged"][0])
    data["data"].close()
    # a part larger than --maxmem is cut into the largest memsafe pieces (not into `buffer`-sized ones)
    monkeypatch.undo()
    monkeypatch.setattr(select, "MAXMEM", 1e-4)                      # 100 kB: about 300 of a part's 1,000 variants
    data = select.load_files(files, None, 64, False)
    sizes = list(data["data"].chunk_vars)
    assert sum(sizes) == sum(shapes["merged"][0]) and data["var_count"].tolist() == shapes["merged"][1]
    assert max(sizes) > 64 and all(select.is_memsafe((n, len(data["samples"]))) for n in sizes)
    data["data"].close()


def test_cli_lowmem_store_create_and_reuse(tmp_path):
    # utmos_ssshtests.sh:197-216: create the store, then reuse it via --lowmem and as the only input
    store = str(tmp_path / "tiny.utm")
    argv, out = cli_args(CASES["select_first"], tmp_path, extra=["--maxmem", "0", "--lowmem", store])
    run_cli(argv)
    assert open(out).read() == ou.golden_text(CASES["select_first"])
    for reuse in (["--lowmem", store], [store]):
        out2 = str(tmp_path / "reuse.txt")
        run_cli(["--maxmem", "1", "-o", out2] + reuse)
        assert open(out2).read() == ou.golden_text(CASES["select_first"])


def test_cli_lowmem_af_uses_float32_values(tmp_path):
    # utmos_ssshtests.sh:218-235: the hdf5 path stores presence*AF as float32 and ranks differently
    store = str(tmp_path / "tiny.af.utm")
    case = CASES["select_af_h5"]
    argv, out = cli_args(case, tmp_path, extra=["--maxmem", "0", "--lowmem", store])
    run_cli(argv)
    assert open(out).read() == ou.golden_text(case)
    out2 = str(tmp_path / "reuse.txt")
    run_cli(["--maxmem", "1", "-c", "20", "--lowmem", store, "-o", out2])          # :225 (no --af: logs, still AF scores)
    assert open(out2).read() == ou.golden_text(case)
    run_cli(["--af", "--maxmem", "1", "-c", "20", store, "-o", out2])               # :231
    assert open(out2).read() == ou.golden_text(case)


def test_cli_af_on_store_without_af_exits_1(tmp_path):
    store = str(tmp_path / "plain.utm")
    argv, _ = cli_args(CASES["select_intcnt"], tmp_path, extra=["--lowmem", store])
    run_cli(argv)
    with pytest.raises(SystemExit) as e:
        run_cli(["--af", store])
    assert e.value.code == 1


def test_calculate_scores_drop_in(tmp_path):
    import numpy as np
    from utmos_amd import device
    from utmos_amd.select import calculate_scores
    rng = np.random.default_rng(3)
    dense = ou.random_dense(rng, 4000, 37)
    cols = ou.npo.pack_columns(dense)
    mask = np.ones(37, np.uint8)
    mask[[2, 9]] = 0
    mask[5] = 2
    w = rng.choice([1.0, 2.0], 37)
    with device.DeviceMatrix(37) as m:
        c = m.add_chunk(4000)
        m.upload_columns(c, cols)
        use, new = calculate_scores(m, mask, w)
        e_use, e_new = ou.npo.score_rowloop(dense, mask, w)
        assert (use, new) == (e_use, e_new)
        mask[:] = 0
        assert calculate_scores(m, mask) == (None, None)


def _cli_rank(rank, world, port, argv, q, extra_env=None):
    import os
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    os.environ.update(extra_env or {})
    try:
        from utmos_amd.select import select_main
        select_main(argv + ["--device", "0"])
        q.put((rank, "ok"))
    except SystemExit as e:
        q.put((rank, f"exit {e.code}"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def _run_two_ranks(argv, port, extra_env=None):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_rank, args=(r, 2, port, argv, q, extra_env)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


@pytest.mark.parametrize("name", ["select_multi", "select_af", "select_weights_subset", "select_first:store",
                                  "select_multi:in-place", "select_af:in-place"])
def test_cli_two_processes_sharded_over_samples(name, tmp_path):
    """`utmos select` as one process per shard (here both on the box's single GPU; socket start-up, then the
    device-side exchange through the mailboxes): rank 0 writes the golden TSV.  `:store` = the shards map their own
    columns of a packed .utm store.  `:in-place` = winner columns are read through the hipIpc mappings instead of a
    local copy, and the selection ends in the middle of a batch of enqueued iterations: the launches behind the stop
    must not touch the peer's columns any more, and nobody frees them before everybody is through (end barrier)."""
    import os
    name, _, variant = name.partition(":")
    argv, out = cli_args(CASES[name], tmp_path)
    if variant == "store":
        store = str(tmp_path / "m.utm")
        run_cli(argv + ["--lowmem", store])                   # single process writes the store
        argv = ["-o", out, "--lowmem", store]
    extra_env = {"UTM_P2P_REPLICATE": "0"} if variant == "in-place" else None
    res = _run_two_ranks(argv, 35500 + os.getpid() % 2000, extra_env)
    assert res == {0: "ok", 1: "ok"}
    assert open(out).read() == ou.golden_text(CASES[name])


def test_cli_exchange_that_cannot_be_set_up_is_an_error_not_a_fallback(tmp_path):
    """`--exchange rccl` with both ranks on one GPU: RCCL refuses two ranks on a device, there is no host-staged
    product path to fall back to, so every rank must leave with exit status 1 (and none may hang)."""
    import os
    argv, _ = cli_args(CASES["select_multi"], tmp_path)
    res = _run_two_ranks(argv + ["--exchange", "rccl"], 36500 + os.getpid() % 2000)
    assert res == {0: "exit 1", 1: "exit 1"}


def test_bench_multi_rank_launch_contract_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N > 1 (one process per rank, RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_* in the environment) -- here three ranks that all sit on the box's single GPU, small workload.
    Rank 0 prints the one JSON line: which exchange ran, the RCCL form timed beside it (or, on this one-GPU box,
    the reason RCCL would not take three ranks), and the sharded rows equal to its own single-GPU re-run."""
    import json
    import subprocess
    import sys
    port = 41500 + os.getpid() % 2000
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TMPDIR=str(tmp_path))
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ou.ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1",
             "--n-var", "400000", "--n-samp", "301", "--no-cpu-baseline"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], [o[1][-400:] for o in outs]
    assert outs[1][0].strip() == "" and outs[2][0].strip() == ""          # only rank 0 reports
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 3 and line["scaling"] == "strong" and line["value"] > 0
    assert line["sharded_rows_match_single_gpu"] is True
    assert line["exchange"] == "mailboxes" and line["p2p_replica_bytes"] > 0 and line["rccl_ranks"] is None
    rccl = line["also_exchange"]["rccl"]
    assert "error" in rccl or (rccl["rccl_ranks"] == 3 and rccl["rows_match_default_exchange"] is True
                               and line["also_exchange"]["rccl_allreduce"]["rows_match_default_exchange"] is True)
    assert line["config"]["iterations_per_step"] == 301 and line["also"] is None


def test_cli_input_without_any_carrier_writes_only_the_header(tmp_path):
    """Every row uninformative (dropped at ingest, select.py:276-279): zero variants, no selection, header only."""
    import numpy as np
    part = ou.load_part("tiny")
    empty = str(tmp_path / "empty.npz")
    np.savez(empty, GT=np.zeros_like(part["GT"]), AF=part["AF"], samples=part["samples"])
    out = str(tmp_path / "o.txt")
    run_cli(["-c", "5", "-o", out, empty])
    assert open(out).read() == "sample\tvar_count\tnew_count\ttot_captured\tpct_captured\n"
    # and mixed with a real part it changes nothing
    real = os.path.join(ou.GOLD, "tiny.npz")
    run_cli(["-c", "20", "-o", out, empty, real])
    assert open(out).read() == ou.golden_text(CASES["select_tiny"])


@pytest.mark.parametrize("extra", [[], ["--af"], ["--brute-force"]])
def test_cli_select_all_until_every_variant_is_captured(extra, tmp_path):
    """`-c -1` over the three fixture chunks: about a thousand iterations until tot_captured == N (the reference
    stops there, select.py:110-112) -- long enough for the CLI's default decremental iterations to take over.
    Expected text = what the reference itself wrote for this command (tests/golden/traces, tools/make_traces.py)."""
    expected = ou.load_trace("all_af64" if "--af" in extra else "all_int")["tsv"]
    out = str(tmp_path / "all.txt")
    run_cli(["-c", "-1", "-o", out] + extra + [os.path.join(ou.GOLD, n + ".npz") for n in ("chunk0", "chunk1", "chunk2")])
    got = open(out).read()
    assert got == expected
    assert got.count("\n") > 1000 and got.rstrip().endswith("\t1.0")


def test_bench_and_cli_under_torchrun_two_ranks(tmp_path):
    """The driver's launch line for N > 1 -- python -m torch.distributed.run --nproc-per-node N ... -- with both
    ranks on the box's one GPU (LOCAL_RANK wraps around the visible devices): bench.py prints its line and matches
    its own single-GPU re-run; `-m utmos_amd select` writes the golden TSV.  torch lives in the launcher only."""
    import json
    import subprocess
    import sys
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    env = dict(os.environ, TMPDIR=str(tmp_path))
    port = 43500 + os.getpid() % 2000
    out = subprocess.run(launch + ["--master-port", str(port), os.path.join(ou.ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                                   "--warmup", "1", "--n-var", "300000", "--n-samp", "200", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ou.ROOT)
    assert out.returncode == 0, out.stderr[-800:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["sharded_rows_match_single_gpu"] is True and line["config"]["iterations_per_step"] == 200
    # the additional RCCL measurements are behind a deadline: when they cannot finish (here: none is allowed to) the
    # headline already measured is still printed, and every rank leaves with status 4 ("measured, but the RCCL legs
    # hung" -- the launcher turns that into its own failure status; a driver can tell it from a clean run)
    out = subprocess.run(launch + ["--master-port", str(port + 2), os.path.join(ou.ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                                   "--warmup", "0", "--n-var", "300000", "--n-samp", "200", "--no-cpu-baseline",
                                   "--rccl-leg-timeout", "0.0001"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ou.ROOT)
    assert out.returncode != 0 and "exitcode  : 4" in out.stderr, out.stderr[-800:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    cut = json.loads(lines[0])
    assert cut["value"] > 0 and cut["exchange"] == "mailboxes" and "did not finish" in cut["also_exchange"]["error"]
    argv, tsv = cli_args(CASES["select_multi"], tmp_path)
    out = subprocess.run(launch + ["--master-port", str(port + 1), "-m", "utmos_amd", "select"] + argv,
                         env=env, capture_output=True, text=True, timeout=600, cwd=ou.ROOT)
    assert out.returncode == 0, out.stderr[-800:]
    assert open(tsv).read() == ou.golden_text(CASES["select_multi"])


def test_bench_line_contract_single_gpu(tmp_path):
    """One rank, small workload: the one JSON line carries the contract's keys, a roofline object whose live PMC
    traffic (two rocprofv3 --pmc child runs) is close to the algorithmic bytes, and a cpu_baseline timed on the oracle."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ou.ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--n-var", "2000000",
                          "--n-samp", "500", "--cpu-sample-vars", "20000"],
                         env=dict(os.environ, TMPDIR=str(tmp_path)), capture_output=True, text=True, timeout=900, cwd=ou.ROOT)
    assert out.returncode == 0, out.stderr[-800:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                             # exactly one line on stdout
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["vs_baseline"] is None and j["dtype"] == "u64"
    assert j["unit"] == "iterations/s" and j["value"] > 0 and j["config"]["iterations_per_step"] == 500
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is not None and r["traffic_source"].startswith("live")
    assert 0.9 < r["traffic"] / r["algo_bytes_per_launch"] < 1.3      # nothing re-read wholesale, nothing skipped
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "iterations/s"


def test_bench_default_line_carries_every_single_gpu_config(tmp_path):
    """The driver's plain `bench.py` invocation (no shape flags): the cfg2 headline plus a few steps each of cfg3, cfg1,
    float64 AF, one rank's share of cfg4 and cfg5 under `also`, each with its own bytes and roofline fraction."""
    import json
    import subprocess
    import sys
    from utmos_amd import _native as nat
    if nat.device_memory(0)[1] < 200e9:
        pytest.skip("cfg5 needs an MI355X-sized HBM")
    out = subprocess.run([sys.executable, os.path.join(ou.ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                          "--pmc-traffic", "off"], env=dict(os.environ, TMPDIR=str(tmp_path)), capture_output=True, text=True,
                         timeout=900, cwd=ou.ROOT)
    assert out.returncode == 0, out.stderr[-800:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["config"]["n_var"] == 10_000_000 and j["config"]["iterations_per_step"] == 2504 and j["exchange"] == "none"
    assert set(j["also"]) == {"cfg3", "af64", "cfg1", "cfg1af", "cfg1af64", "cfg4rank", "cfg5"}
    for name, want_iters in (("cfg3", 2504), ("af64", 2504), ("cfg1", 2504), ("cfg1af", 2504), ("cfg1af64", 2504), ("cfg4rank", 20), ("cfg5", 10)):
        e = j["also"][name]
        assert "error" not in e, e
        assert e["iterations_per_step"] == want_iters and e["value"] > 0 and e["algo_bytes_per_step"] > 0
        # (the whole-loop bound is a sanity check, not a performance gate: one stalled step of five must not turn the suite red)
        assert 0.3 < e["roofline"]["frac"] < 1.0 and 0.1 < e["hbm_frac_whole_loop"] < 1.0, (name, e)
    assert j["also"]["cfg5"]["chunks"] == 10
    # cfg3's bytes follow what the AF variant actually reads: a delta pass never re-reads the 40 MB AF table
    per_iter = j["also"]["cfg3"]["algo_bytes_per_step"] / 2504
    assert per_iter < 1.5694e9 * 1.012


@pytest.mark.parametrize("extra", [[], ["--af"], ["--af", "--lowmem", "STORE"]], ids=["int", "af", "af-store"])
def test_cli_reads_a_vcf(extra, tmp_path):
    """f3: `select x.vcf` -- text VCF -> packed rows -> device, against the oracle fed with the same parsed part.
    (What the reference's scikit-allel would make of half-missing / haploid calls is parity unpinned; the fixture
    is build-authored: tests/golden/vcf/build_tiny.vcf.)"""
    from oracle_util import npo
    from utmos_amd.vcfio import read_vcf
    vcf = os.path.join(ou.GOLD, "vcf", "build_tiny.vcf")
    part = read_vcf(vcf)
    f32 = "--lowmem" in extra
    expected = npo.select_tsv([{"GT": part["GT"], "AF": part["AF"].reshape(-1), "samples": part["samples"]}], count=-1,
                              af="--af" in extra, af_dtype="f32" if f32 else "f64")
    out = str(tmp_path / "o.tsv")
    argv = [a if a != "STORE" else str(tmp_path / "v.utm") for a in extra]
    run_cli(["-c", "-1", "-o", out] + argv + [vcf])
    assert open(out).read() == expected
    assert expected.count("\n") > 5
    # and gzip-compressed
    import gzip
    import shutil
    gz = str(tmp_path / "copy.vcf.gz")
    with open(vcf, "rb") as src, gzip.open(gz, "wb") as dst:
        shutil.copyfileobj(src, dst)
    argv = [a if a != "STORE" else str(tmp_path / "v2.utm") for a in extra]
    run_cli(["-c", "-1", "-o", out] + argv + [gz])
    assert open(out).read() == expected


def test_cli_reads_a_joblib_part(tmp_path):
    """f1: the `.jl` branch of the ingest (the reference's own container, utmos/convert.py:98) -- the file is written
    here with joblib.dump from a re-encoded fixture, mixed with an .npz part like the reference's multi-file test
    mixes .jl and .vcf (utmos_ssshtests.sh:105-121)."""
    import joblib
    p0 = ou.load_part("chunk0")
    jl = str(tmp_path / "chunk0.jl")
    joblib.dump({"GT": p0["GT"], "AF": p0["AF"].reshape(-1, 1), "samples": p0["samples"].astype("S"), "stats": {}}, jl, compress=5)
    out = str(tmp_path / "o.tsv")
    run_cli(["-o", out, jl, os.path.join(ou.GOLD, "chunk2.npz")])
    assert open(out).read() == ou.golden_text(CASES["select_multi"])
    run_cli(["-c", "20", "--af", "-o", out, jl, os.path.join(ou.GOLD, "chunk1.npz")])
    assert open(out).read() == ou.golden_text(CASES["select_af"])
