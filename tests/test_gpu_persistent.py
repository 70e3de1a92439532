"""The persistent loop kernel (k_loop_int, utmos_amd/csrc/loop_int.hip.h): batches of unweighted integer iterations as ONE
launch.  Same rows as the oracle (and therefore as one launch per iteration) over ragged shapes, used / excluded samples,
ties, zero-score stops, early full coverage, runs cut into calls, both tile sizes -- and the bounded waits' failure paths."""
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


def run_both(dev, dense, state=None, k=None, pieces=None, monkeypatch=None):
    n_var, n_samp = dense.shape
    state = np.ones(n_samp, np.uint8) if state is None else state
    cols = npo.pack_columns(dense)
    exp = ou.c_greedy(cols, n_var, state, k_max=k)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.upload_columns(c, cols)
        m.set_state(state)
        idx, new = [], []
        for piece in (pieces or [n_samp if k is None else k]):
            got = m.run(piece)
            idx += got[0].tolist()
            new += got[1].tolist()
            assert (got[2] == got[1]).all()
        st = m.stats()
        covered = m.covered(c)
    assert idx == exp[0][:len(idx)].tolist() and new == exp[1][:len(new)].tolist()
    if pieces is None:
        assert len(idx) == len(exp[0])
    # the covered mask that came back from the workers' LDS tiles = OR of the winners' columns (and of the used samples')
    want = np.zeros(cols.shape[1], np.uint64)
    for s in list(np.flatnonzero(state == 0)) + idx:
        want |= cols[s]
    assert (covered == want).all()
    return st


@pytest.mark.parametrize("n_var,n_samp", [(1, 1), (63, 3), (64, 4), (65, 5), (1000, 130), (8192, 64), (8193, 257), (70_000, 700),
                                          (200_000, 37), (3_000, 2_500)])
def test_persistent_loop_select_all(dev, n_var, n_samp):
    rng = np.random.default_rng(n_var * 31 + n_samp)
    st = run_both(dev, ou.random_dense(rng, n_var, n_samp))
    assert st["persist_iterations"] == st["iterations"] > 0 and st["persist_launches"] >= 1


def test_persistent_loop_is_what_runs_and_can_be_switched_off(dev, monkeypatch):
    rng = np.random.default_rng(3)
    dense = ou.random_dense(rng, 50_000, 300)
    on = run_both(dev, dense)
    assert on["persist_iterations"] == 300 and on["persist_launches"] == 2 and on["score_launches"] == 2      # 256 + 44
    monkeypatch.setenv("UTM_PERSISTENT", "0")
    off = run_both(dev, dense)
    assert off["persist_iterations"] == 0 and off["score_launches"] >= 300


def test_persistent_loop_states_ties_and_stops(dev):
    rng = np.random.default_rng(11)
    n_var, n_samp = 40_000, 500
    dense = ou.random_dense(rng, n_var, n_samp, density=0.02)
    dense[:, 7] = dense[:, 3]                          # exact ties: lowest index first (np.argmax)
    dense[:, 400] = dense[:, 3]
    state = np.ones(n_samp, np.uint8)
    state[rng.choice(n_samp, 60, replace=False)] = 2   # excluded: never selected, never cover
    state[[5, 99, 250]] = 0                            # start out used: cover from the first iteration
    state[3] = 1
    run_both(dev, dense, state)
    run_both(dev, dense, state, k=17)
    # cut into calls: 1, 2, 255, 256, 3 ... every call continues where the last one stopped (pending winner folded in)
    run_both(dev, dense, state, pieces=[1, 2, 255, 256, 3, 1000])
    # zero-score stop: half the samples carry nothing that is not covered by sample 0
    dense2 = np.zeros((5000, 64), bool)
    dense2[:, 0] = True
    dense2[rng.integers(0, 5000, 300), rng.integers(1, 32, 300)] = True
    st = run_both(dev, dense2)
    assert st["iterations"] == 1                       # sample 0 captures everything: "ran out of new variants"
    state2 = np.ones(64, np.uint8)
    state2[0] = 2                                      # now the rest is selected until the best score is 0
    st = run_both(dev, dense2, state2)
    assert 1 <= st["iterations"] < 32


@pytest.mark.parametrize("tile_kib", ["16", "32", "64"])
@pytest.mark.parametrize("n_var,n_samp", [(300_000, 300), (1_000_000, 90), (70_001, 1_500), (2_100_000, 40)])
def test_persistent_loop_tiles_of_several_batches(dev, tile_kib, n_var, n_samp, monkeypatch):
    """Tiles of 2 / 4 / 8 batches (taller matrices: fewer atomics per count word): a position's partial count goes out with
    its last batch; short last tiles, tiles shorter than one batch, columns that end inside a batch; claimed positions
    (few resident blocks)."""
    monkeypatch.setenv("UTM_PERSIST_TILE_KIB", tile_kib)
    monkeypatch.setenv("UTM_PERSIST_WGS_PER_CU", "1")
    rng = np.random.default_rng(int(tile_kib) + n_samp)
    dense = rng.random((n_var, n_samp)) < 0.03
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    st = run_both(dev, dense, k=min(n_samp, 40))
    assert st["persist_iterations"] == st["iterations"] > 0


def test_persistent_loop_tile_is_chosen_by_the_matrix_height_and_tall_or_wide_matrices_keep_the_launches(dev, monkeypatch):
    rng = np.random.default_rng(1)
    monkeypatch.setenv("UTM_PERSIST_MAX_TILES", "4")
    for n_var, expect in ((200_000, True), (1_900_000, True), (2_200_000, False)):      # 4 x 8 KiB ... 4 x 64 KiB tiles, then none
        dense = rng.random((n_var, 24)) < 0.05
        dense[np.arange(n_var), rng.integers(0, 24, n_var)] = True
        st = run_both(dev, dense, k=6)
        assert (st["persist_iterations"] > 0) == expect, (n_var, st)
    monkeypatch.delenv("UTM_PERSIST_MAX_TILES")
    st = run_both(dev, ou.random_dense(rng, 3_000, 3_000), k=20)                           # more samples than one picker chunk
    assert st["persist_iterations"] == 0


@pytest.mark.parametrize("n_var,n_samp,wgs", [(5_000, 2_000, "1"), (70_000, 2_500, "1"), (20_000, 2_300, "2"), (600_000, 1_100, "1")])
def test_persistent_loop_claims_positions_dynamically(dev, n_var, n_samp, wgs, monkeypatch):
    """More selectable samples than wave slots (few resident blocks): the positions behind every wave's static one are
    claimed from the tiles' counters -- every (position, tile) partial exactly once, every iteration."""
    monkeypatch.setenv("UTM_PERSIST_WGS_PER_CU", wgs)
    rng = np.random.default_rng(n_var + n_samp)
    dense = ou.random_dense(rng, n_var, n_samp, density=0.01)
    state = np.ones(n_samp, np.uint8)
    state[rng.choice(n_samp, n_samp // 10, replace=False)] = 2
    st = run_both(dev, dense, state, k=min(n_samp, 600))
    assert st["persist_iterations"] == st["iterations"] > 0


def test_withheld_partial_count_ends_in_an_error_and_the_context_recovers(dev, monkeypatch):
    """the pick's bounded wait.  UTM_TEST_DROP_ARRIVAL=n withholds one partial count in the n-th iteration
    of a persistent launch (or the n-th scoring launch): the picker gives up after its spin budget, the loop ends with
    UTM_EHIP, and after utm_reset the same context produces the oracle's rows."""
    rng = np.random.default_rng(8)
    n_var, n_samp = 30_000, 120
    dense = ou.random_dense(rng, n_var, n_samp)
    cols = npo.pack_columns(dense)
    exp = ou.c_greedy(cols, n_var, np.ones(n_samp, np.uint8))
    for persistent in ("1", "0"):
        monkeypatch.setenv("UTM_PERSISTENT", persistent)
        monkeypatch.setenv("UTM_TEST_DROP_ARRIVAL", "5")
        with dev.DeviceMatrix(n_samp) as m:
            c = m.add_chunk(n_var)
            m.upload_columns(c, cols)
            with pytest.raises(dev.nat.NativeError) as e:
                m.run(n_samp)
            assert e.value.code == -2 and "did not all arrive" in str(e.value)
            monkeypatch.setenv("UTM_TEST_DROP_ARRIVAL", "0")
            m.reset()                                   # (re-reads the knobs)
            got = m.run(n_samp)
            assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist()


def _af_case(rng, n_var, n_samp, density=0.03):
    dense = rng.random((n_var, n_samp)) < density
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    # float32 AFs on a narrow exponent range: every sum stays below 2^53 units, the exact fixed-point phase from the start
    af = (rng.integers(1, 2 * n_samp, n_var) / (2.0 * n_samp)).astype(np.float32)
    return dense, af


@pytest.mark.parametrize("n_var,n_samp,weights", [(60_000, 300, False), (200_000, 120, True), (9_000, 1_200, False), (300_000, 64, False)])
def test_persistent_af_loop_exact_phase(dev, n_var, n_samp, weights):
    """The AF form of the persistent loop (float32 AF, exact fixed-point phase, optionally weights): after the first full
    pass and the dense delta passes the per-sample accumulators are kept current from per-position decreases -- same rows,
    counts and float64 scores as the oracle, bit for bit, also when the run is cut into calls."""
    rng = np.random.default_rng(n_var + n_samp)
    dense, af = _af_case(rng, n_var, n_samp)
    w = rng.choice([0.5, 1.0, 1.0, 2.0, -1.0], n_samp) if weights else None
    state = np.ones(n_samp, np.uint8)
    state[rng.choice(n_samp, n_samp // 12, replace=False)] = 2
    cols = npo.pack_columns(dense)
    exp = ou.c_greedy(cols, n_var, state, w, af)
    for pieces in ([n_samp], [5, 1, 70, 256, n_samp]):
        with dev.DeviceMatrix(n_samp) as m:
            c = m.add_chunk(n_var)
            m.upload_columns(c, cols)
            m.set_af(c, af)
            m.set_state(state)
            m.set_weights(w)
            idx, new, score = [], [], []
            for piece in pieces:
                got = m.run(piece)
                idx += got[0].tolist(); new += got[1].tolist(); score += got[2].tolist()
            st = m.stats()
        assert idx == exp[0].tolist() and new == exp[1].tolist()
        assert score == exp[2].tolist()                          # float64 scores, bit for bit
        assert st["af_fixed_point"] == 1
        assert st["persist_iterations"] > 0.5 * len(idx), st    # most of the run went through k_loop_int<.., AF>


def test_persistent_af_loop_can_be_switched_off(dev, monkeypatch):
    """(UTM_PERSIST_AF=0 switches off both AF forms: float64 AF takes the interval form, tests/test_gpu_persistent_interval.py)"""
    rng = np.random.default_rng(77)
    dense, af = _af_case(rng, 50_000, 200)
    cols = npo.pack_columns(dense)
    state = np.ones(200, np.uint8)
    for af_values, env, expect in ((af, "1", True), (af, "0", False), (af.astype(np.float64) / 3.0, "1", True), (af.astype(np.float64) / 3.0, "0", False)):
        monkeypatch.setenv("UTM_PERSIST_AF", env)
        exp = ou.c_greedy(cols, 50_000, state, None, af_values)
        with dev.DeviceMatrix(200) as m:
            c = m.add_chunk(50_000)
            m.upload_columns(c, cols)
            m.set_af(c, af_values)
            got = m.run(200)
            st = m.stats()
        assert got[0].tolist() == exp[0].tolist() and got[2].tolist() == exp[2].tolist()
        assert (st["persist_iterations"] > 0) == expect, st


@pytest.mark.parametrize("kind", ["positive", "signed", "all_negative", "with_zero"])
def test_persistent_loop_with_weights(dev, kind):
    """Integer counts times per-sample weights inside the persistent loop (the picker compares float64 products): signed and
    zero weights, the negative-best rule (a negative product only wins when no sample holds a masked 0: select.py:43-48)."""
    rng = np.random.default_rng(len(kind))
    n_var, n_samp = 40_000, 260
    dense = ou.random_dense(rng, n_var, n_samp, density=0.02)
    w = {"positive": rng.choice([0.5, 1.0, 2.0, 3.5], n_samp),
         "signed": rng.choice([-2.0, -0.5, 1.0, 2.0], n_samp),
         "all_negative": -rng.choice([0.5, 1.0, 2.0], n_samp),
         "with_zero": rng.choice([0.0, 1.0, 1.0, 2.0], n_samp)}[kind]
    state = np.ones(n_samp, np.uint8)
    if kind != "all_negative":
        state[rng.choice(n_samp, 20, replace=False)] = 2
    cols = npo.pack_columns(dense)
    exp = ou.c_greedy(cols, n_var, state, w)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.upload_columns(c, cols)
        m.set_state(state)
        m.set_weights(w)
        got = m.run(n_samp)
        st = m.stats()
    assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist() and got[2].tolist() == exp[2].tolist()
    assert st["persist_iterations"] == st["iterations"] or len(exp[0]) == 0


def test_a_grid_that_cannot_be_resident_aborts_at_its_census_and_the_launches_take_over(dev, monkeypatch):
    """UTM_PERSIST_WGS_PER_CU far beyond what a CU holds: the blocks that do become resident count in, the picker's bounded
    census wait runs out, everybody leaves without having touched anything (xerror 3), and the context runs the same batch
    -- and everything after it -- as one launch per iteration.  Same rows; no persistent iteration on record."""
    monkeypatch.setenv("UTM_PERSIST_WGS_PER_CU", "12")
    rng = np.random.default_rng(21)
    dense = ou.random_dense(rng, 200_000, 2_000, density=0.01)
    st = run_both(dev, dense, k=40)
    assert st["persist_iterations"] == 0 and st["persist_launches"] == 0 and st["score_launches"] >= 40


def test_a_batch_longer_than_a_launch_holds_is_cut(dev, monkeypatch):
    monkeypatch.setenv("UTM_BATCH", "1000")
    rng = np.random.default_rng(22)
    dense = ou.random_dense(rng, 20_000, 700, density=0.01)
    st = run_both(dev, dense)
    assert st["persist_iterations"] == st["iterations"] and st["persist_launches"] >= 3


@pytest.mark.parametrize("af", [False, True])
def test_persistent_loop_with_more_samples_than_one_picker_chunk(dev, af, monkeypatch):
    """UTM_PERSIST_MAX_SAMPLES=0 lifts the default limit of 2,560 samples: the picker then walks the count words in
    several chunks (and re-reads act[] and the AF accumulators per chunk) -- slower, same rows."""
    monkeypatch.setenv("UTM_PERSIST_MAX_SAMPLES", "0")
    rng = np.random.default_rng(31 + af)
    n_var, n_samp = 30_000, 6_000
    dense = ou.random_dense(rng, n_var, n_samp, density=0.004)
    cols = npo.pack_columns(dense)
    state = np.ones(n_samp, np.uint8)
    state[rng.choice(n_samp, 300, replace=False)] = 2
    afv = (rng.integers(1, 2 * n_samp, n_var) / (2.0 * n_samp)).astype(np.float32) if af else None
    exp = ou.c_greedy(cols, n_var, state, None, afv, k_max=150)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.upload_columns(c, cols)
        if af:
            m.set_af(c, afv)
        m.set_state(state)
        got = m.run(150)
        st = m.stats()
    assert got[0].tolist() == exp[0].tolist() and got[1].tolist() == exp[1].tolist() and got[2].tolist() == exp[2].tolist()
    assert st["persist_iterations"] > 0
