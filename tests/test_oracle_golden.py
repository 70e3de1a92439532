"""The oracle is pinned here: every golden TSV the reference's own suite uses for `utmos select`
(repo_utils/utmos_ssshtests.sh:81-235) must come out byte-for-byte from the re-encoded fixtures."""
import numpy as np
import pytest

import oracle_util as ou
from oracle_util import npo

CASES = ou.golden_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_numpy_oracle_reproduces_golden(name):
    case = CASES[name]
    parts = [ou.load_part(n) for n in case["inputs"]]
    assert npo.select_tsv(parts, **ou.case_kwargs(case["args"])) == ou.golden_text(case)


@pytest.mark.parametrize("name", ["select_intcnt", "select_weights_subset", "select_one_af", "select_tiny"])
def test_rowloop_scorer_equals_blocked(name):
    case = CASES[name]
    parts = [ou.load_part(n) for n in case["inputs"]]
    kw = ou.case_kwargs(case["args"])
    assert npo.select_tsv(parts, scorer=npo.score_rowloop, **kw) == ou.golden_text(case)


def test_f32_and_f64_af_paths_differ_as_in_reference():
    # answer_key/select_af.txt vs select_af_h5.txt: the reference's in-memory (float64) and hdf5
    # (float32) paths rank the last rows differently; the oracle must keep both behaviours apart.
    assert ou.golden_text(CASES["select_af"]) != ou.golden_text(CASES["select_af_h5"])


def test_count_resolution_table():
    # select.py:157-159
    assert npo.resolve_count(2504, -1) == 2504
    assert npo.resolve_count(2504, 0) == 1
    assert npo.resolve_count(2504, 0.02) == 50
    assert npo.resolve_count(2504, 0.005) == 12
    assert npo.resolve_count(2504, 1) == 1
    assert npo.resolve_count(2504, 10) == 10
    assert npo.resolve_count(2504, 0.0001) == 1


def test_pct_rounding_is_numpy_rounding():
    m = np.zeros((8, 2), dtype=bool)
    m[:5, 0] = True
    m[5:, 1] = True
    rows = list(npo.greedy(m, m.sum(axis=0), 2, np.array(["a", "b"]), np.ones(2, np.uint8)))
    assert [npo.format_row(r) for r in rows] == ["a\t5\t5\t5\t0.625\n", "b\t3\t3\t8\t1.0\n"]
