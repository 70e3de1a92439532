"""The full dense AF pass as table lookups (k_score_aft, score_af.hip.h) against the CPU oracle: rows, counts and
float64 scores of whole runs, with the table kernel (UTM_AF_TABLES=1,
the default) and with the bit-walking kernel it replaces (k_score_afq, UTM_AF_TABLES=0).  Reference arithmetic:
utmos/select.py:36-48 (row value added to every carrier of an uncovered row; first maximum wins)."""
import numpy as np
import pytest

import oracle_util as ou
from test_gpu_parity import check_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from utmos_amd import _native as nat
    assert nat.device_count() >= 1, "no GPU visible"
    from utmos_amd import device
    return device


def _af_of(dense, n_samp, kind):
    af = np.maximum(dense.sum(axis=1), 1) / (2.0 * n_samp)
    return af.astype(np.float32) if kind == "f32" else af / 3.0


@pytest.mark.parametrize("tables", ["1", "0"])
@pytest.mark.parametrize("n_samp", [65, 66, 67, 131, 300])
def test_select_all_checks_every_samples_first_pass_sum(dev, tables, n_samp, monkeypatch):
    """float32 AF in the exact range: a winner's reported score IS its accumulator (first full pass minus the delta
    passes' exact decreases), so a select-all run compares every sample's first-pass sum with the oracle, bit for bit."""
    monkeypatch.setenv("UTM_AF_TABLES", tables)
    rng = np.random.default_rng(1000 + n_samp)
    n_var = 2048 * 5 + 777                      # ragged last tile of the 2,048-variant table tiles
    dense = ou.random_dense(rng, n_var, n_samp)
    af = _af_of(dense, n_samp, "f32")
    af[::53] = 0.0
    state = np.ones(n_samp, np.uint8)
    state[3] = 2                                # excluded
    state[n_samp - 1] = 0                       # already used: its variants start out covered
    got, st = check_run(dev, dense, state=state, af=af)
    assert len(got[0]) > n_samp // 2
    # (two samples are not selectable: below 64 selectable ones the pass stays with k_score_afq)
    assert st["af_table_passes"] == (1 if tables == "1" and n_samp - 2 >= 64 else 0), st


@pytest.mark.parametrize("tables", ["1", "0"])
@pytest.mark.parametrize("shape", [(2048 * 5 + 777, 131), (4096 * 9, 256), (4096 * 3 + 1, 513)])
def test_first_pass_of_every_sample_against_the_oracle(dev, tables, shape, monkeypatch):
    """utm_peek_estimates = the parallel full pass: counts and (float32 AF, exact range) float64 sums of ALL samples."""
    monkeypatch.setenv("UTM_AF_TABLES", tables)
    monkeypatch.setenv("UTM_AF_TABLE_RUN", "2")
    n_var, n_samp = shape
    rng = np.random.default_rng(n_samp)
    dense = ou.random_dense(rng, n_var, n_samp)
    af = _af_of(dense, n_samp, "f32")
    af[::41] = 0.0
    state = np.ones(n_samp, np.uint8)
    state[7] = 2
    state[n_samp // 2] = 0
    w = rng.choice([0.5, 1.0, 3.0], n_samp)
    cols = ou.npo.pack_columns(dense)
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.upload_columns(c, cols)
        m.set_af(c, af)
        m.set_state(state)
        m.set_weights(w)
        cnt, score = m.peek_estimates()
        st = m.stats()
        rows = m.run(5)                          # the loop still starts from scratch afterwards
    _, exp_cnt, exp_score = ou.c_score(cols, n_var, state, w, af)
    sel = state == 1
    assert cnt[sel].tolist() == exp_cnt[sel].tolist() and (cnt[~sel] == 0).all()
    assert score[sel].tolist() == exp_score[sel].tolist()
    assert st["af_table_passes"] == (1 if tables == "1" else 0), st
    exp = ou.c_greedy(cols, n_var, state, w, af, k_max=5)
    assert rows[0].tolist() == exp[0].tolist() and rows[2].tolist() == exp[2].tolist()


@pytest.mark.parametrize("tables", ["1", "0", "run3", "run40"])
@pytest.mark.parametrize("kind", ["f32", "f64", "f32_chunks", "f32_weights"])
def test_runs_with_the_table_pass(dev, tables, kind, monkeypatch):
    """`run3` / `run8`: a workgroup keeps its samples' partial sums in registers over 3 / all tiles (6 tiles here: a short
    last run, and one run shorter than asked for); by itself the host gives a matrix this small one tile per workgroup."""
    if tables.startswith("run"):
        monkeypatch.setenv("UTM_AF_TABLE_RUN", tables[3:])
        tables = "1"
    monkeypatch.setenv("UTM_AF_TABLES", tables)
    rng = np.random.default_rng(77)
    n_var, n_samp = 2048 * 11 + 5, 150
    dense = ou.random_dense(rng, n_var, n_samp)
    af = _af_of(dense, n_samp, "f64" if kind == "f64" else "f32")
    w = rng.choice([0.5, 1.0, 2.0], n_samp) if kind == "f32_weights" else None
    chunks = [0, 2048 * 3 + 100, 2048 * 7, n_var] if kind == "f32_chunks" else None
    _, st = check_run(dev, dense, af=af, weights=w, chunks=chunks)
    n_chunks = 3 if chunks else 1
    assert st["af_table_passes"] == (n_chunks if tables == "1" else 0), st


@pytest.mark.parametrize("run", ["0", "2"])
def test_more_samples_than_one_group_holds(dev, run, monkeypatch):
    """Above 256 samples the sample axis is cut into groups (a wave keeps at most 8 units of two samples in registers)."""
    monkeypatch.setenv("UTM_AF_TABLE_RUN", run)
    rng = np.random.default_rng(5)
    n_var, n_samp = 2048 * 2 + 64, 1100
    dense = rng.random((n_var, n_samp)) < 0.02
    dense[np.arange(n_var), rng.integers(0, n_samp, n_var)] = True
    af = _af_of(dense, n_samp, "f32")
    _, st = check_run(dev, dense, af=af)
    assert st["af_table_passes"] == 1, st


def test_values_too_wide_for_the_limbs_keep_the_bit_walking_kernel(dev):
    """Exponents spread over more than 21 bits: a table value could pass 2^45 units, k_score_aft is not admitted."""
    rng = np.random.default_rng(6)
    n_var, n_samp = 9000, 80
    dense = ou.random_dense(rng, n_var, n_samp)
    af = (2.0 ** -rng.integers(1, 30, n_var)).astype(np.float32)
    _, st = check_run(dev, dense, af=af, k=20)
    assert st["af_table_passes"] == 0, st


@pytest.mark.parametrize("n_samp", [96, 130, 200, 257])
def test_dense_columns(dev, n_samp, monkeypatch):
    """Every nibble value occurs: half of all cells set.  130 samples: groups of 4 units per wave instead of 8."""
    monkeypatch.setenv("UTM_AF_TABLE_RUN", "4")
    rng = np.random.default_rng(8)
    n_var = 4096 * 5 + 100
    dense = rng.random((n_var, n_samp)) < 0.5
    af = _af_of(dense, n_samp, "f32")
    _, st = check_run(dev, dense, af=af, k=10)
    assert st["af_table_passes"] == 1, st
