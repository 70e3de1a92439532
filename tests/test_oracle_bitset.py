"""The packed C oracle against the golden-pinned numpy oracle (same inputs, every mode)."""
import numpy as np
import pytest

import oracle_util as ou
from oracle_util import npo


def run_np(dense, state, weights, af, af_dtype, k):
    n_var, n_samp = dense.shape
    m = dense
    if af is not None:
        m = dense * af.astype(np.float64).reshape(-1, 1)
        if af_dtype == "f32":
            m = m.astype(np.float32)
    names = np.arange(n_samp)
    st = state.copy()
    rows = list(npo.greedy(m, dense.sum(axis=0), k, names, st, weights))
    return [int(r[0]) for r in rows], [r[2] for r in rows]


@pytest.mark.parametrize("mode", ["int", "weights", "af32", "af64", "af64_weights"])
@pytest.mark.parametrize("seed", [0, 1])
def test_c_oracle_matches_numpy_oracle(mode, seed):
    rng = np.random.default_rng(seed)
    n_var, n_samp = 700 + 37 * seed, 130
    dense = ou.random_dense(rng, n_var, n_samp)
    state = np.ones(n_samp, np.uint8)
    state[rng.choice(n_samp, 7, replace=False)] = 2
    weights = rng.choice([0.5, 1.0, 2.0, 3.0], n_samp) if "weights" in mode else None
    af = None
    af_dtype = "f64"
    if mode.startswith("af"):
        ac = dense.sum(axis=1)
        af = ac / (2.0 * n_samp)
        if mode == "af32":
            af = af.astype(np.float32)
            af_dtype = "f32"
    exp_idx, exp_new = run_np(dense, state, weights, af, af_dtype, n_samp)
    cols = npo.pack_columns(dense)
    idx, new, _ = ou.c_greedy(cols, n_var, state, weights, af)
    assert idx.tolist() == exp_idx
    assert new.tolist() == exp_new


def test_c_oracle_on_golden_fixture_multi():
    parts = [ou.load_part("chunk0"), ou.load_part("chunk2")]
    dense, var_count, samples = npo.build_matrix(parts)
    cols = npo.pack_columns(dense)
    idx, new, _ = ou.c_greedy(cols, dense.shape[0], np.ones(len(samples), np.uint8), k_max=50)
    lines = ou.golden_text(ou.golden_cases()["select_multi"]).splitlines()[1:]
    assert [samples[i] for i in idx] == [ln.split("\t")[0] for ln in lines]
    assert new.tolist() == [int(ln.split("\t")[2]) for ln in lines]


def test_ties_pick_lowest_index_and_excluded_never_cover():
    dense = np.zeros((6, 4), dtype=bool)
    dense[0:3, 1] = True
    dense[0:3, 2] = True          # identical to column 1 -> tie, index 1 must win
    dense[3:6, 3] = True          # column 3 excluded: its variants stay uncaptured
    dense[3, 0] = True
    state = np.array([1, 1, 1, 2], np.uint8)
    idx, new, _ = ou.c_greedy(npo.pack_columns(dense), 6, state)
    assert idx.tolist() == [1, 0] and new.tolist() == [3, 1]


def test_fixture_provenance_static_jl_reader():
    """tests/golden/chunk*.npz are what tools/jl_static.py (opcode walk, nothing unpickled) reads out of the
    reference's own .jl fixtures -- checked whenever the reference tree is present (build container only)."""
    import os
    import sys
    ref = "/root/reference/repo_utils/test_files"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present (GPU box)")
    sys.path.insert(0, os.path.join(ou.ROOT, "tools"))
    from jl_static import read_jl
    for name in ("chunk0", "chunk1", "chunk2"):
        d = read_jl(os.path.join(ref, name + ".jl"))
        p = ou.load_part(name)
        assert (d["GT"] == p["GT"]).all() and (d["AF"].reshape(-1) == p["AF"]).all()
        assert (np.asarray(d["samples"], dtype=str) == p["samples"]).all()


def test_hypothesis_c_oracle_equals_numpy_oracle():
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 300), st.integers(1, 40), st.integers(0, 2 ** 31), st.sampled_from(["int", "w", "af32", "af64"]))
    def run(n_var, n_samp, seed, mode):
        rng = np.random.default_rng(seed)
        dense = rng.random((n_var, n_samp)) < rng.choice([0.02, 0.2, 0.6])
        state = rng.choice([1, 1, 1, 0, 2], n_samp).astype(np.uint8)
        weights = rng.choice([-1.0, 0.0, 0.5, 1.0, 2.0], n_samp) if mode == "w" else None
        af, af_dtype = None, "f64"
        if mode.startswith("af"):
            af = rng.random(n_var)
            af[rng.random(n_var) < 0.1] = 0.0
            if mode == "af32":
                af, af_dtype = af.astype(np.float32), "f32"
        k = int(rng.integers(1, n_samp + 1))
        exp_idx, exp_new = run_np(dense, state, weights, af, af_dtype, k)
        idx, new, _ = ou.c_greedy(npo.pack_columns(dense), n_var, state, weights, af, k_max=k)
        assert idx.tolist() == exp_idx and new.tolist() == exp_new

    run()
