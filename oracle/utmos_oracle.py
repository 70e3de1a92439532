"""CPU oracle: a numpy restatement of utmos's greedy maximum-coverage selection.

TEST INFRASTRUCTURE ONLY.  Nothing in ``utmos_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.

What it restates (paths relative to /root/reference):

  build_matrix        utmos/select.py:275-284, 314-321 (in-memory) and :218-223 (hdf5, float32 values)
  resolve_count       utmos/select.py:157-159
  initial_state       utmos/select.py:169-179
  weight_vector       utmos/select.py:181-187
  score_rowloop       utmos/select.py:24-53   (literal per-row structure; this is what is timed as the CPU baseline)
  score_blocked       same arithmetic, row-blocked; sequential f64 accumulation kept via add.accumulate
  greedy              utmos/select.py:69-112  (the h5 compaction branch :116-137 is result-neutral and omitted)
  format_row          utmos/select.py:102-108, 445

Parity pins: (1) every golden TSV the reference's own suite uses for this path
(repo_utils/utmos_ssshtests.sh:81-235 -> tests/golden/answer_key/*.txt) is reproduced
byte-for-byte by tests/test_oracle_golden.py from the re-encoded fixtures
(tests/golden/*.npz, made by tools/make_golden.py); (2) outputs of the reference's own code,
imported in the build container by tools/make_traces.py (tests/golden/traces/*.json: TSVs, winners
and float64 scores of 24 cases the goldens do not reach, and the --count table), are reproduced by
tests/test_reference_traces.py -- by this file and by oracle_bitset.c.
"""
import numpy as np

HEADER = "sample\tvar_count\tnew_count\ttot_captured\tpct_captured\n"

USABLE, USED, EXCLUDED = 1, 0, 2


# --------------------------------------------------------------------------- ingest
def build_matrix(parts, af=False, af_dtype="f64"):
    """parts: list of dicts {GT uint8 (n, ceil(S/8)) MSB-first, AF float64 (n,), samples (S,)}.

    Returns (matrix, var_count, samples).  matrix is bool (N, S), or float64/float32
    presence*AF with ``af``.  Rows without any carrier are dropped per part
    (select.py:276-279); var_count sums the kept rows (select.py:281-284).
    af_dtype 'f32' restates the hdf5 store, which keeps presence*AF as float32
    (select.py:218-223) -- the reference's two paths differ numerically.
    """
    samples = np.asarray(parts[0]["samples"]).astype(str)
    kept, afs = [], []
    var_count = np.zeros(len(samples), dtype=np.int64)
    for p in parts:
        dense = np.unpackbits(p["GT"], axis=1, count=len(samples)).astype(bool)
        informative = dense.any(axis=1)
        dense = dense[informative]
        kept.append(dense)
        afs.append(np.asarray(p["AF"], dtype=np.float64).reshape(-1)[informative])
        var_count += dense.sum(axis=0)
    matrix = np.concatenate(kept) if len(kept) > 1 else kept[0]
    if af:
        col = np.concatenate(afs).reshape(-1, 1)
        matrix = matrix * col                       # bool * float64 (N,1) -> float64
        if af_dtype == "f32":
            matrix = matrix.astype(np.float32)
    return matrix, var_count, samples


def resolve_count(n_samples, count):
    """--count semantics: <0 all, <1 fraction (at least 1), else integer."""
    if count < 0:
        return n_samples
    return max(1, int(n_samples * count) if count < 1 else int(count))


def initial_state(samples, subset=None, exclude=None):
    """1 = selectable, 0 = already used (covers variants), 2 = excluded (never selected, never covers)."""
    state = np.ones(len(samples), dtype=np.uint8)
    if subset:
        state = np.where(np.isin(samples, subset), USABLE, EXCLUDED).astype(np.uint8)
    if exclude:
        state = np.where(np.isin(samples, exclude), EXCLUDED, state).astype(np.uint8)
    return state


def weight_vector(samples, weights):
    """weights: dict name -> float, or None.  Unlisted samples weigh 1.0."""
    if weights is None:
        return None
    w = np.ones(len(samples), dtype=np.float64)
    for i, name in enumerate(samples):
        if name in weights:
            w[i] = weights[name]
    return w


# --------------------------------------------------------------------------- scoring
def _finish(scores, counts, state, weights):
    scores[state != USABLE] = 0
    if weights is not None:
        scores *= weights
    best = int(np.argmax(scores))                   # first maximum
    if scores[best] == 0:
        return None, None, scores
    return best, int(counts[best]), scores


def score_rowloop(matrix, state, weights=None):
    """One greedy iteration, row at a time (the reference's loop shape, select.py:33-48)."""
    n_samp = matrix.shape[1]
    scores = np.zeros(n_samp)
    counts = np.zeros(n_samp, dtype=np.int64)
    used = np.flatnonzero(state == USED)
    for variant in matrix:
        if variant[used].any():
            continue                                # already captured by a selected sample
        scores += variant                           # float64 accumulate, in row order
        counts += variant != 0
    best, new, _ = _finish(scores, counts, state, weights)
    return best, new


def score_blocked(matrix, state, weights=None, block=4096, return_scores=False):
    """Same result as score_rowloop, processed in row blocks.

    add.accumulate along axis 0 is a strict left-to-right running sum per column, so the
    float64 rounding sequence equals the per-row ``scores += row`` of the reference
    (adding 0.0 for absent cells changes nothing).
    """
    n_samp = matrix.shape[1]
    scores = np.zeros(n_samp)
    counts = np.zeros(n_samp, dtype=np.int64)
    used = state == USED
    for lo in range(0, matrix.shape[0], block):
        blk = matrix[lo:lo + block]
        live = blk[~(blk[:, used] != 0).any(axis=1)] if used.any() else blk
        if live.shape[0] == 0:
            continue
        counts += (live != 0).sum(axis=0)
        if live.dtype == bool:
            scores += live.sum(axis=0)              # integers: exact in any order
        else:
            run = np.add.accumulate(np.vstack([scores[None, :], live.astype(np.float64)]), axis=0)
            scores = run[-1].copy()
    best, new, final = _finish(scores, counts, state, weights)
    if return_scores:
        return best, new, final, counts
    return best, new


# --------------------------------------------------------------------------- driver
def greedy(matrix, var_count, select_count, samples, state, weights=None, scorer=score_blocked):
    """Yield [name, var_count, new_count, tot_captured, pct_captured] per selected sample.

    Mutates ``state`` (winner -> USED), like the reference mutates sample_mask.
    Stops (a) after select_count rows, (b) silently when the best score is 0,
    (c) after the row that brings tot_captured to N.
    """
    n_var = matrix.shape[0]
    captured = 0
    for _ in range(select_count):
        best, new = scorer(matrix, state, weights)
        if best is None:
            break
        captured = captured + np.int64(new)
        state[best] = USED
        yield [samples[best], int(var_count[best]), int(new), int(captured),
               round(captured / n_var, 4)]          # np.float64.__round__ -> numpy rounding
        if captured >= n_var:
            return


def format_row(row):
    return "\t".join(str(x) for x in row) + "\n"


def select_tsv(parts, count=0.02, af=False, af_dtype="f64", subset=None, exclude=None, weights=None,
               scorer=score_blocked):
    """End to end: fixtures -> the TSV text ``utmos select`` would write."""
    matrix, var_count, samples = build_matrix(parts, af=af, af_dtype=af_dtype)
    k = resolve_count(len(samples), count)
    state = initial_state(samples, subset, exclude)
    w = weight_vector(samples, weights)
    out = [HEADER]
    for row in greedy(matrix, var_count, k, samples, state, w, scorer=scorer):
        out.append(format_row(row))
    return "".join(out)


# --------------------------------------------------------------------------- packed layout helpers
def pack_columns(dense_bool):
    """bool (N, S) -> uint64 (S, W) column-major bitsets, LSB-first within a word, zero padded."""
    n_var, n_samp = dense_bool.shape
    n_words = (n_var + 63) // 64
    padded = np.zeros((n_words * 64, n_samp), dtype=bool)
    padded[:n_var] = dense_bool
    bytes_ = np.packbits(padded.T.reshape(n_samp, n_words * 64), axis=1, bitorder="little")
    return np.ascontiguousarray(bytes_).view("<u8").reshape(n_samp, n_words)
