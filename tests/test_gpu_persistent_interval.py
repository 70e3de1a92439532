"""The interval form of the persistent AF loop (k_loop_int<.., 2>, loop_picker<3>: utmos_amd/csrc/loop_int.hip.h): float64 AF
values -- and float32 ones outside the exact fixed-point range -- with the candidates found, and their sequential float64
chains run, inside the picker.  Same rows, counts and (where exact scores are asked for) float64 scores as the oracle, bit
for bit; what the picker cannot settle (a chain longer than it holds, more candidates than the list) goes to the host's
verification launch and still yields the oracle's rows."""
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


def quantized_af(rng, n_var, n_samp, dtype=np.float64):
    """ac / an quotients with few distinct numerators (then / 3: full 53-bit mantissas): different sets of variants give
    equal exact sums all the time -- the estimates tie, only the order of the float64 additions tells the samples apart."""
    af = rng.integers(1, 12, n_var) / (2.0 * n_samp) / 3.0
    return af.astype(dtype)


def run_af(dev, cols, n_var, n_samp, af, state=None, weights=None, pieces=None, exact=True, k=None, estimate_rtol=1e-6):
    state = np.ones(n_samp, np.uint8) if state is None else state
    exp = ou.c_greedy(cols, n_var, state, weights, af, k_max=k)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.upload_columns(c, cols)
        m.set_af(c, af)
        m.set_state(state)
        m.set_weights(weights)
        if not exact:
            m.set_af_exact_scores(False)
        idx, new, score = [], [], []
        for piece in (pieces or [n_samp if k is None else k]):
            got = m.run(piece)
            idx += got[0].tolist(); new += got[1].tolist(); score += got[2].tolist()
        st = m.stats()
    assert idx == exp[0][:len(idx)].tolist() and new == exp[1][:len(new)].tolist()
    if pieces is None:
        assert len(idx) == len(exp[0])
    if exact:
        assert score == exp[2][:len(score)].tolist()            # float64 scores, bit for bit
    elif estimate_rtol:
        assert np.allclose(score, exp[2][:len(score)], rtol=estimate_rtol, atol=0)
    return st


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("n_var,n_samp,weights", [(60_000, 300, False), (200_000, 120, True), (9_000, 1_200, False), (300_000, 64, False),
                                                  (70_001, 2_504, False)])
def test_float64_af_through_the_persistent_loop(dev, n_var, n_samp, weights, exact):
    rng = np.random.default_rng(n_var + n_samp)
    dense = rng.random((n_var, n_samp)) < (0.03 if n_samp < 2000 else 0.004)
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    af = quantized_af(rng, n_var, n_samp)
    w = rng.choice([0.5, 1.0, 1.0, 2.0, -1.0], n_samp) if weights else None
    state = np.ones(n_samp, np.uint8)
    state[rng.choice(n_samp, n_samp // 12, replace=False)] = 2
    cols = npo.pack_columns(dense)
    for pieces in (None, [5, 1, 70, 256, n_samp]):
        st = run_af(dev, cols, n_var, n_samp, af, state, w, pieces, exact)
        assert st["af_fixed_point"] == 1
        assert st["persist_iterations"] > 0.5 * st["iterations"], st      # most of the run went through k_loop_int<.., 2>
        assert st["af_chained_iterations"] > 0 or n_samp < 100, st        # ... and ties were told apart by chains
        assert st["persist_unresolved"] <= max(2, 0.02 * st["iterations"]), st   # ... (nearly) all of them inside the picker
        if exact:
            assert st["af_deferred_rows"] > 0.5 * st["iterations"], st    # the winners' exact scores: from the masks the launches logged


def test_float32_af_outside_the_exact_range_takes_the_interval_form(dev):
    """float32 AF values spread over many binades: the fixed-point unit is coarser than the smallest values' last bit, sums
    are estimates with an error bound -- candidates and float32 -> float64 chains as for float64 AF."""
    rng = np.random.default_rng(5)
    n_var, n_samp = 80_000, 260
    dense = rng.random((n_var, n_samp)) < 0.03
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    dense[:, 7] = dense[:, 3]                                   # exact ties all along
    af = np.exp2(rng.uniform(-40, -1, n_var)).astype(np.float32)
    cols = npo.pack_columns(dense)
    for exact in (True, False):   # (estimated scores on a coarse unit: each addend may lose up to one unit -- rows only)
        st = run_af(dev, cols, n_var, n_samp, af, exact=exact, estimate_rtol=None)
        assert st["persist_iterations"] > 0.5 * st["iterations"], st


def test_twin_columns_tie_in_every_iteration(dev):
    """Every sample has an identical twin: each iteration's top two intervals coincide, the chains give equal sums, the lower
    index wins (select.py:93 np.argmax).  Early ties run over more addends than a picker chain holds
    (UTM_LOOP_CHAIN_CAP): those iterations end their launch undecided and the verification launch decides them."""
    rng = np.random.default_rng(8)
    n_var, half = 150_000, 150
    base = rng.random((n_var, half)) < rng.uniform(0.0002, 0.008, half)[None, :]   # 30 .. 1,200 variants a sample
    dense = np.concatenate([base, base], axis=1)[:, rng.permutation(2 * half)]
    af = quantized_af(rng, n_var, 2 * half)
    cols = npo.pack_columns(dense)
    st = run_af(dev, cols, n_var, 2 * half, af)
    assert st["persist_iterations"] > 0 and st["af_chained_iterations"] > 10, st
    assert st["persist_unresolved"] > 0, st                     # (gains of thousands of variants early in the run)


def test_more_tied_candidates_than_the_list_holds(dev):
    """70 copies of one column among 200 samples: once they lead, 70 intervals reach the best lower bound -- more than the
    candidate list (UTM_MAX_CAND = 64): the launch ends undecided, the host's launches re-score everyone sequentially."""
    rng = np.random.default_rng(9)
    n_var, n_samp = 120_000, 200
    dense = rng.random((n_var, n_samp)) < 0.01
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    copies = rng.choice(n_samp, 70, replace=False)
    dense[:, copies] = (rng.random(n_var) < 0.004)[:, None]
    af = quantized_af(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    st = run_af(dev, cols, n_var, n_samp, af)
    assert st["persist_iterations"] > 0 and st["persist_unresolved"] > 0, st


def test_interval_form_can_be_switched_off(dev, monkeypatch):
    rng = np.random.default_rng(10)
    n_var, n_samp = 50_000, 200
    dense = rng.random((n_var, n_samp)) < 0.03
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    af = quantized_af(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    for env, expect in (("1", True), ("0", False)):
        monkeypatch.setenv("UTM_PERSIST_AF_INTERVAL", env)
        st = run_af(dev, cols, n_var, n_samp, af)
        assert (st["persist_iterations"] > 0) == expect, st


def test_weights_with_signs_and_zeros(dev):
    """Negative weights swap an interval's ends, a zero weight makes it [0, 0]; the negative-best rule (select.py:43-48)."""
    rng = np.random.default_rng(12)
    n_var, n_samp = 60_000, 240
    dense = rng.random((n_var, n_samp)) < 0.02
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    af = quantized_af(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    for kind in ("signed", "all_negative", "with_zero"):
        w = {"signed": rng.choice([-2.0, -0.5, 1.0, 2.0], n_samp),
             "all_negative": -rng.choice([0.5, 1.0, 2.0], n_samp),
             "with_zero": rng.choice([0.0, 1.0, 1.0, 2.0], n_samp)}[kind]
        state = np.ones(n_samp, np.uint8)
        if kind != "all_negative":
            state[rng.choice(n_samp, 20, replace=False)] = 2
        run_af(dev, cols, n_var, n_samp, af, state, w)


@pytest.mark.parametrize("env", [{"UTM_PERSIST_CHAINERS": "1"}, {"UTM_PERSIST_CHAINERS": "8"}, {"UTM_PERSIST_SPEC_TICKS": "-1"},
                                 {"UTM_PERSIST_SPEC_TICKS": "0", "UTM_PERSIST_CHAINERS": "2"}],
                         ids=["one-chainer", "eight-chainers", "no-work-ahead", "always-ahead"])
def test_chainer_knobs_keep_the_rows(dev, env, monkeypatch):
    """One to eight chainer blocks, working ahead or only on request: same rows, counts and float64 scores."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(14)
    n_var, n_samp = 90_000, 700
    dense = rng.random((n_var, n_samp)) < 0.01
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    af = quantized_af(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    st = run_af(dev, cols, n_var, n_samp, af)
    assert st["persist_iterations"] > 0.5 * st["iterations"], st


def test_requests_nobody_answers_run_into_their_bounded_wait(dev, monkeypatch):
    """Test hook UTM_PERSIST_SPEC_TICKS=-2: the chainers leave right after the census.  A pick that needs sums which are not
    on record asks, waits its bounded time (~0.1 s), ends the launch undecided, and the host's verification launch decides:
    the oracle's rows all the same (and launches that keep ending this way are spaced out)."""
    monkeypatch.setenv("UTM_PERSIST_SPEC_TICKS", "-2")
    rng = np.random.default_rng(15)
    n_var, n_samp = 40_000, 120
    dense = rng.random((n_var, n_samp)) < 0.02
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    dense[:, 11] = dense[:, 5]                                   # one exact tie that lasts until one of the two is picked
    af = quantized_af(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    st = run_af(dev, cols, n_var, n_samp, af)
    assert st["persist_unresolved"] > 0 and st["persist_iterations"] > 0, st


@pytest.mark.parametrize("tile_kib", ["16", "32", "64"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_af_forms_with_forced_larger_tiles(dev, tile_kib, dtype, monkeypatch):
    """By default the AF forms take the 8 KiB tile only (UTM_PERSIST_AF_MAX_TILES); UTM_PERSIST_TILE_KIB forces the
    instantiations with tiles of several batches: same rows and scores."""
    monkeypatch.setenv("UTM_PERSIST_TILE_KIB", tile_kib)
    rng = np.random.default_rng(int(tile_kib))
    n_var, n_samp = 400_000, 150
    dense = rng.random((n_var, n_samp)) < 0.02
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    af = quantized_af(rng, n_var, n_samp) if dtype == "f64" else (rng.integers(1, 2 * n_samp, n_var) / (2.0 * n_samp)).astype(np.float32)
    cols = npo.pack_columns(dense)
    st = run_af(dev, cols, n_var, n_samp, af)
    if tile_kib == "64":       # (not built for the AF forms: two waves per SIMD, half the launches' rate -- the launches take the run)
        assert st["persist_iterations"] == 0, st
    else:
        assert st["persist_iterations"] > 0.5 * st["iterations"], st


@pytest.mark.parametrize("seed", range(16))
def test_random_shapes_states_weights_and_cuts(dev, seed):
    """Random small cases through the interval form: shape, density, tie-prone or full-mantissa AF values, float32 values on
    a coarse unit, used / excluded samples, signed weights, estimated scores, runs cut at random rows."""
    rng = np.random.default_rng(1000 + seed)
    n_var = int(rng.integers(2_000, 160_000))
    n_samp = int(rng.integers(2, 700))
    dense = rng.random((n_var, n_samp)) < rng.uniform(0.002, 0.08)
    if rng.random() < 0.7:
        dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    if n_samp > 4 and rng.random() < 0.5:
        dense[:, 1] = dense[:, n_samp - 1]                      # twins
    kind = rng.integers(0, 3)
    if kind == 0:
        af = quantized_af(rng, n_var, n_samp)
    elif kind == 1:
        af = rng.random(n_var) * 0.5 + 1e-4                     # full 53-bit mantissas
    else:
        af = np.exp2(rng.uniform(-36, -1, n_var)).astype(np.float32)
    state = rng.choice([0, 1, 1, 1, 1, 2], n_samp).astype(np.uint8)
    w = rng.choice([-1.0, 0.0, 0.5, 1.0, 1.0, 2.0], n_samp) if rng.random() < 0.4 else None
    exact = bool(rng.random() < 0.6)
    pieces = None
    if rng.random() < 0.5:
        pieces = sorted(int(x) for x in rng.integers(1, max(2, n_samp), 3)) + [n_samp]
    cols = npo.pack_columns(dense)
    run_af(dev, cols, n_var, n_samp, af, state, w, pieces, exact, estimate_rtol=None if kind == 2 else 1e-6)


def test_chr22_sized_float64_af_first_rows_against_the_oracle(dev):
    """1,103,547 x 2,504 (BASELINE configs[0]'s size) with float64 AF values, the synthetic matrix of bench.py's cfg1af64: the
    first 300 rows -- the launch path's dense passes, then persistent launches with their first ties -- against the OpenMP
    C oracle on the matrix downloaded from the device; rows, counts and float64 scores bit for bit."""
    n_var, n_samp, k = 1_103_547, 2504, 300
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=0)
        cols = m.download_columns(c)
        _, af = dev.synth_host(0, n_var, n_samp, want_cols=False)
        af64 = af.astype(np.float64) / 3.0
        m.set_af(c, af64)
        got = m.run(k)
        st = m.stats()
    exp = ou.c_greedy(cols, n_var, np.ones(n_samp, np.uint8), af=af64, k_max=k, omp=True)
    assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist() and got[2].tolist() == exp[2].tolist()
    assert st["persist_iterations"] > 200 and st["af_deferred_rows"] > 0, st
