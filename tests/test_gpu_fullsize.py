"""Oracle comparisons at the BASELINE.json sizes (not invariants): the matrix the device generated is downloaded and the
packed C oracle (oracle/oracle_bitset.c, OpenMP over all host cores) re-computes

  * the first rows of the run from scratch (orc_greedy), and
  * whole iterations from states the GPU run reached late in the run (orc_score on `used` = the GPU's own winners so
    far): the winner, its new_count, its float64 score -- and every selectable sample's count / score through
    utm_peek_scores, bit for bit.

Reference being matched: utmos/select.py:36-53 (skip-if-covered, mask, weights, first argmax), :99-100.
cfg2 / cfg3 (10M x 2,504), cfg4's one-rank shape (50M x 12,500, 78 GB) and cfg5 (500M x 2,504 in ten chunks, per-chunk
oracle counts summed) -- no BASELINE configuration is left with invariants as its only full-size check.
"""
import time

import numpy as np
import pytest

import oracle_util as ou

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from utmos_amd import _native as nat
    assert nat.device_count() >= 1, "no GPU visible"
    from utmos_amd import device
    return device


def download(m, chunk, slice_cols=512):
    """A chunk's columns, fetched in slices of columns into one host array (a 78 GB matrix does not come in one copy)."""
    words = (m.chunk_vars[chunk] + 63) // 64
    out = np.empty((m.n_local, words), dtype=np.uint64)
    for lo in range(0, m.n_local, slice_cols):
        n = min(slice_cols, m.n_local - lo)
        m.download_columns(chunk, lo, n, out=out[lo:lo + n])
    return out


def state_after(n_samp, winners, base=None):
    st = np.ones(n_samp, np.uint8) if base is None else base.copy()
    st[np.asarray(winners, dtype=np.int64)] = 0
    return st


def check_late_state(m, cols, n_var, rows, k, af=None, base_state=None, follow=3):
    """The iteration that follows the GPU run's first k rows, recomputed by the oracle from nothing but those k winners:
    every selectable sample's count and final score (peek), the winner, and the next `follow` rows of a run resumed there."""
    idx, new, score = rows
    state = state_after(len(cols), idx[:k], base_state)
    best, cnt, sc = ou.c_score(cols, n_var, state, af=af, omp=True)
    assert best == idx[k], f"iteration {k}: oracle picks {best}, the GPU run picked {idx[k]}"
    assert cnt[best] == new[k]
    assert sc[best] == score[k]                        # float64, bit for bit (integer mode: the count itself)
    m.set_state(state)
    counts, scores = m.peek_scores()
    sel = state == 1
    assert (counts[sel] == cnt[sel]).all() and (counts[~sel] == 0).all()
    assert (scores == sc).all()
    resumed = m.run(follow)
    assert resumed[0].tolist() == idx[k:k + follow].tolist() and resumed[1].tolist() == new[k:k + follow].tolist()
    assert resumed[2].tolist() == score[k:k + follow].tolist()


def test_cfg2_full_size_rows_against_the_oracle(dev):
    """BASELINE configs[1], 10M x 2,504, select all: first 32 rows from scratch + the states after 1,252 and 2,440
    selections (all 2,504 counts each time)."""
    n_var, n_samp = 10_000_000, 2504
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=0)
        cols = download(m, c)
        rows = m.run(n_samp)
        assert len(rows[0]) == n_samp
        exp = ou.c_greedy(cols, n_var, np.ones(n_samp, np.uint8), k_max=32, omp=True)
        assert rows[0][:32].tolist() == exp[0].tolist() and rows[1][:32].tolist() == exp[1].tolist()
        for k in (1252, 2440):
            check_late_state(m, cols, n_var, rows, k)


@pytest.mark.parametrize("af_dtype", ["f32", "f64"])
def test_cfg3_full_size_rows_against_the_oracle(dev, af_dtype):
    """BASELINE configs[2], 10M x 2,504 with AF weighting (float32 = the reference's hdf5 values, float64 = its in-memory
    values): first 8 rows from scratch + the state after 1,252 selections, float64 scores bit for bit."""
    n_var, n_samp = 10_000_000, 2504
    _, af = dev.synth_host(0, n_var, n_samp, want_cols=False)
    af = af if af_dtype == "f32" else af.astype(np.float64) / 3.0
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=0)
        m.set_af(c, af)
        cols = download(m, c)
        if af_dtype == "f32":
            # the parallel first pass itself (k_score_aft: table lookups): every sample's count and exact sum
            counts, est = m.peek_estimates()
            _, cnt0, sc0 = ou.c_score(cols, n_var, np.ones(n_samp, np.uint8), af=af, omp=True)
            assert (counts == cnt0).all() and (est == sc0).all()
            assert m.stats()["af_table_passes"] == 1
        rows = m.run(n_samp)
        assert len(rows[0]) == n_samp
        exp = ou.c_greedy(cols, n_var, np.ones(n_samp, np.uint8), af=af, k_max=8, omp=True)
        assert rows[0][:8].tolist() == exp[0].tolist() and rows[1][:8].tolist() == exp[1].tolist()
        assert rows[2][:8].tolist() == exp[2].tolist()
        check_late_state(m, cols, n_var, rows, 1252, af=af)


def test_cfg4_one_rank_shape_against_the_oracle(dev):
    """One rank's share of BASELINE configs[3]: 50M variants x 12,500 samples (78 GB of HBM), some samples excluded:
    the first 3 rows and every sample's count of the first iteration against the OpenMP oracle on the downloaded
    matrix."""
    free, total = dev.nat.device_memory(0)
    if total < 120e9 or bench_host_mem_gb() < 140:
        pytest.skip("needs an MI355X-sized HBM and 140 GB of host memory")
    n_var, n_samp, k = 50_000_000, 12_500, 3
    with dev.DeviceMatrix(n_samp) as m:
        c = m.add_chunk(n_var)
        m.synth_fill(c, seed=0)
        t0 = time.perf_counter()
        cols = download(m, c, slice_cols=400)
        print(f"78 GB down in {time.perf_counter() - t0:.1f} s")
        state = np.ones(n_samp, np.uint8)
        state[::97] = 2
        m.set_state(state)
        counts, scores = m.peek_scores()
        best, cnt, sc = ou.c_score(cols, n_var, state, omp=True)
        assert (counts == cnt).all() and (scores == sc).all()
        rows = m.run(k)
        exp = ou.c_greedy(cols, n_var, state, k_max=k, omp=True)
        assert rows[0].tolist() == exp[0].tolist() and rows[1].tolist() == exp[1].tolist() and rows[0][0] == best


def bench_host_mem_gb():
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_cfg5_full_size_counts_against_the_oracle(dev):
    """BASELINE configs[4], 500M x 2,504 in ten HBM-resident chunks (156 GB): per-chunk oracle counts (one chunk on the
    host at a time), summed over the ten chunks, against utm_peek_scores for iteration 0 and for the state after 12
    selections; the oracle's argmax over the sums is the GPU run's row.  Then the first iteration of an `--af` run over
    the same 500M rows: the parallel first pass's counts and exact float32-AF sums of every eighth sample (utm_peek_estimates)."""
    free, total = dev.nat.device_memory(0)
    if total < 200e9:
        pytest.skip("needs an MI355X-sized HBM (156 GB matrix)")
    n_var, n_samp, chunk_vars, k = 500_000_000, 2504, 50_000_000, 16
    with dev.DeviceMatrix(n_samp) as m:
        v0 = 0
        while v0 < n_var:
            m.synth_fill(m.add_chunk(chunk_vars), seed=0, first_var_global=v0)
            v0 += chunk_vars
        rows = m.run(k)
        assert len(rows[0]) == k
        states = [np.ones(n_samp, np.uint8), state_after(n_samp, rows[0][:12])]
        sums = [np.zeros(n_samp, np.int64) for _ in states]
        afs, af_sum = [], np.zeros(n_samp, np.float64)
        af_state = np.full(n_samp, 2, np.uint8)          # the AF leg scores 313 samples (the oracle's float adds are the test's time)
        af_state[::8] = 1
        for c in range(10):
            cols = download(m, c)
            for st, acc in zip(states, sums):
                _, cnt, _ = ou.c_score(cols, chunk_vars, st, omp=True)     # covered = OR of the used samples' columns of THIS chunk
                acc += cnt
            # ... and the first iteration of an --af run over the same chunks (float32 AF: every partial sum is exact, so the
            # chunks' float64 sums add up to the reference's sum over all 500M rows)
            # (AF on a 2^-16 grid: a sample's sum over 500M rows then stays below 2^53 units, every float64 add is exact)
            af_c = dev.synth_host(0, chunk_vars, n_samp, first_var_global=c * chunk_vars, want_cols=False)[1]
            afs.append((np.maximum(np.rint(af_c.astype(np.float64) * 65536.0), 1.0) / 65536.0).astype(np.float32))
            _, af_cnt_c, af_sc_c = ou.c_score(cols, chunk_vars, af_state, af=afs[-1], omp=True)
            af_sum += af_sc_c
            af_cnt = af_cnt_c if c == 0 else af_cnt + af_cnt_c
            del cols
        for st, acc, at in zip(states, sums, (0, 12)):
            m.set_state(st)
            counts, scores = m.peek_scores()
            assert (counts == acc).all() and (scores == acc).all()
            assert int(np.argmax(acc)) == rows[0][at] and acc.max() == rows[1][at]      # first maximum = np.argmax (select.py:48)
        # the parallel first AF pass at this size (ten k_score_aft launches): the 313 selectable samples' counts and exact sums
        for c in range(10):
            m.set_af(c, afs[c])
        m.set_state(af_state)
        counts, est = m.peek_estimates()
        sel = af_state == 1
        assert (counts[sel] == af_cnt[sel]).all() and (counts[~sel] == 0).all() and (est == af_sum).all()
        assert m.stats()["af_table_passes"] == 10
