"""Both CPU oracles against traces of the REFERENCE's own code (tests/golden/traces, written by
tools/make_traces.py in the build container from an import of /root/reference/utmos/select.py).

Beyond the 14 golden TSVs: the ~1,000-iteration select-all in every value mode, signed / zero weights
(select.py:45-48), initially used samples (:36-39), AF == 0 rows, exact ties (:48), the zero-score stop
(:51-52), everything excluded, the h5 compaction branch (:116-137) and the --count table (:157-159)."""
import json
import os

import numpy as np
import pytest

import oracle_util as ou
from oracle_util import npo


class Tap:
    """score_blocked, remembering every final score vector (what the reference hands to np.argmax)."""

    def __init__(self):
        self.vectors = []

    def __call__(self, matrix, state, weights=None):
        best, new, scores, _ = npo.score_blocked(matrix, state, weights, return_scores=True)
        self.vectors.append(scores.copy())
        return best, new


def oracle_run(t):
    kw = ou.trace_options(t)
    matrix, var_count, samples = npo.build_matrix(t["part_list"], af=kw.get("af", False), af_dtype=kw.get("af_dtype", "f64"))
    k = npo.resolve_count(len(samples), kw.get("count", 0.02))
    if "state" in t:
        state = np.array(t["state"], dtype=np.uint8)
    else:
        state = npo.initial_state(samples, kw.get("subset"), kw.get("exclude"))
    if "weight_vector_hex" in t:
        w = np.array([float.fromhex(x) for x in t["weight_vector_hex"]])
    else:
        w = npo.weight_vector(samples, kw.get("weights"))
    tap = Tap()
    rows = list(npo.greedy(matrix, var_count, k, samples, state, w, scorer=tap))
    tsv = npo.HEADER + "".join(npo.format_row(r) for r in rows)
    return tsv, tap, list(samples), (matrix, var_count, state, w, k)


@pytest.mark.parametrize("name", ou.trace_names())
def test_numpy_oracle_equals_reference_trace(name):
    t = ou.load_trace(name)
    tsv, tap, samples, _ = oracle_run(t)
    assert tsv == t["tsv"]
    assert len(tap.vectors) == t["argmax_calls"]            # including a final pass that found nothing
    if "idx" in t:
        idx = [samples.index(ln.split("\t")[0]) for ln in tsv.splitlines()[1:]]
        assert idx == t["idx"]
        assert [float(tap.vectors[k][i]).hex() for k, i in enumerate(idx)] == t["score_hex"]
    if "scores_hex" in t:                                   # every sample's score, every iteration, bit for bit
        assert [[float(x).hex() for x in v] for v in tap.vectors] == t["scores_hex"]


@pytest.mark.parametrize("name", [n for n in ou.trace_names() if not n.startswith("compaction")])
def test_c_bitset_oracle_equals_reference_trace(name):
    """The packed C oracle (what the GPU parity tests lean on at larger sizes) against the same traces."""
    t = ou.load_trace(name)
    kw = ou.trace_options(t)
    dense, var_count, samples = npo.build_matrix(t["part_list"])
    af = None
    if kw.get("af"):
        af = np.concatenate([np.asarray(p["AF"], dtype=np.float64).reshape(-1)[
            np.unpackbits(p["GT"], axis=1, count=len(samples)).any(axis=1)] for p in t["part_list"]])
        if kw.get("af_dtype") == "f32":
            af = af.astype(np.float32)
    state = np.array(t["state"], dtype=np.uint8) if "state" in t else npo.initial_state(samples, kw.get("subset"), kw.get("exclude"))
    w = np.array([float.fromhex(x) for x in t["weight_vector_hex"]]) if "weight_vector_hex" in t \
        else npo.weight_vector(samples, kw.get("weights"))
    k = npo.resolve_count(len(samples), kw.get("count", 0.02))
    idx, new, score = ou.c_greedy(npo.pack_columns(dense), dense.shape[0], state, w, af, k_max=k)
    want = [ln.split("\t") for ln in t["tsv"].splitlines()[1:]]
    assert [samples[i] for i in idx] == [r[0] for r in want]
    assert new.tolist() == [int(r[2]) for r in want]
    assert idx.tolist() == t["idx"]
    assert [float(s).hex() for s in score] == t["score_hex"]


def test_count_resolution_equals_reference_table():
    table = json.load(open(os.path.join(ou.TRACES, "count_table.json")))["n_samp,count,k"]
    assert len(table) >= 70
    for n_samp, count, k in table:
        assert npo.resolve_count(n_samp, count) == k, (n_samp, count)


def test_trace_set_covers_the_cases_the_goldens_do_not():
    names = set(ou.trace_names())
    assert {"all_int", "all_af64", "all_af32", "ties_int", "zero_score_stop", "all_excluded", "initial_used_int",
            "weights_signed_int", "af_zero_rows", "compaction_int"} <= names
    assert [ou.load_trace(n)["tsv"].count("\n") - 1 for n in ("all_int", "all_af64", "all_af32")] == [1052, 1098, 1101]
