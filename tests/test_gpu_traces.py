"""The HIP path against traces of the REFERENCE's own code (tests/golden/traces; tools/make_traces.py) --
device == reference, not device == oracle: TSV bytes through the CLI, and through the C ABI the winner,
new_count and float64 score of every iteration, for micro-cases the whole score vector before every pick."""
import os

import numpy as np
import pytest

import oracle_util as ou
from oracle_util import npo

pytestmark = pytest.mark.gpu

CLI = [n for n in ou.trace_names() if ou.load_trace(n)["kind"] == "cli"]
ALL = ou.trace_names()


def write_parts(t, tmp_path):
    files = []
    for i, p in enumerate(t["part_list"]):
        path = str(tmp_path / f"part{i}.npz")
        np.savez(path, GT=p["GT"], AF=np.asarray(p["AF"]).reshape(-1), samples=np.asarray(p["samples"]).astype("U"))
        files.append(path)
    return files


def cli_argv(t, tmp_path):
    argv, src, i = [], t["argv"], 0
    while i < len(src):
        a = src[i]
        if a in ("--weights", "--subset") and (src[i + 1].endswith(".txt") or src[i + 1] == "<weights>"):
            v = src[i + 1]
            if v == "<weights>":
                v = str(tmp_path / "weights.tsv")
                with open(v, "w") as fh:
                    fh.write("".join(f"{k}\t{w!r}\n" for k, w in t["weights"]))
            else:
                v = os.path.join(ou.GOLD, v)
            argv += [a, v]
            i += 2
        else:
            argv.append(a)
            i += 1
    return argv


@pytest.mark.parametrize("extra", [[], ["--brute-force"], ["--maxmem", "0", "--buffer", "128"]], ids=["default", "brute", "chunked"])
@pytest.mark.parametrize("name", CLI)
def test_cli_writes_the_reference_tsv(name, extra, tmp_path):
    from utmos_amd.select import select_main
    t = ou.load_trace(name)
    out = str(tmp_path / "out.tsv")
    select_main(cli_argv(t, tmp_path) + extra + ["-o", out] + write_parts(t, tmp_path))
    assert open(out).read() == t["tsv"]


def device_setup(t, tmp_path):
    """load_files + the recorded call's state / weights; returns (data, state, weights, k)."""
    from utmos_amd import select
    kw = ou.trace_options(t)
    af = bool(kw.get("af"))
    f32 = kw.get("af_dtype") == "f32"
    # --lowmem is the reference's float32 path (hdf5 keeps presence*AF as float32, select.py:218-223)
    data = select.load_files(write_parts(t, tmp_path), str(tmp_path / "store.utm") if f32 else None, 32768, af)
    samples = np.asarray(data["samples"]).astype(str)
    state = np.array(t["state"], dtype=np.uint8) if "state" in t else npo.initial_state(samples, kw.get("subset"), kw.get("exclude"))
    w = np.array([float.fromhex(x) for x in t["weight_vector_hex"]]) if "weight_vector_hex" in t \
        else npo.weight_vector(samples, kw.get("weights"))
    return data, samples, state, w, npo.resolve_count(len(samples), kw.get("count", 0.02))


@pytest.mark.parametrize("name", ALL)
def test_greedy_select_yields_the_reference_rows(name, tmp_path):
    """The host mirror of the reference's seam, greedy_select(matrix, var_count, K, names, mask, weights)."""
    from utmos_amd import select
    t = ou.load_trace(name)
    data, samples, state, w, k = device_setup(t, tmp_path)
    rows = list(select.greedy_select(data["data"], np.asarray(data["var_count"]), k, samples, state, w))
    assert select.HEADER + "".join("\t".join(str(x) for x in r) + "\n" for r in rows) == t["tsv"]
    data["data"].close()


@pytest.mark.parametrize("decremental", [False, True], ids=["brute", "decremental"])
@pytest.mark.parametrize("name", [n for n in ALL if not n.startswith("compaction")])
def test_abi_run_reports_the_reference_winner_and_score(name, decremental, tmp_path):
    t = ou.load_trace(name)
    data, samples, state, w, k = device_setup(t, tmp_path)
    m = data["data"]
    m.set_decremental(decremental)
    m.set_state(state)
    m.set_weights(w)
    m.reset()
    idx, new, score = m.run(k)
    want = [ln.split("\t") for ln in t["tsv"].splitlines()[1:]]
    assert idx.tolist() == t["idx"]
    assert new.tolist() == [int(r[2]) for r in want]
    assert [float(s).hex() for s in score] == t["score_hex"]        # the float64 running sum, bit for bit
    m.close()


@pytest.mark.parametrize("name", [n for n in ALL if "scores_hex" in ou.load_trace(n)])
def test_every_samples_score_before_every_pick(name, tmp_path):
    """utm_peek_scores + utm_step: the vector the reference hands to np.argmax (select.py:43-48), all samples."""
    t = ou.load_trace(name)
    data, samples, state, w, k = device_setup(t, tmp_path)
    m = data["data"]
    m.set_state(state)
    m.set_weights(w)
    m.reset()
    vectors = t["scores_hex"]
    for it, vec in enumerate(vectors[:k]):
        _, scores = m.peek_scores()
        # (+ 0.0: a masked 0 times a negative weight is -0.0 in numpy; the sign of a zero never reaches the output)
        assert [float(x + 0.0).hex() for x in scores] == [float(float.fromhex(x) + 0.0).hex() for x in vec], it
        got = m.step()
        if it < len(t["idx"]):
            assert got is not None and got[0] == t["idx"][it]
        else:
            assert got is None
    m.close()
