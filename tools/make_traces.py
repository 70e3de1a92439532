#!/usr/bin/env python3
"""Capture traces of the REFERENCE's own selection code (build container only; needs /root/reference).

What runs: `/root/reference/utmos/select.py`, imported unmodified from where it lies.  Its three absent
third-party imports are satisfied by empty in-memory modules (SURVEY.md §8c recipe): `h5py` (two marker
classes for the isinstance tests at select.py:116/:163/:191), `truvari` (a no-op `setup_logging`,
select.py:400) and `allel` (never touched on this path).  No reference source, bytecode or pickle is copied:
inputs are this repo's own re-encodings (tests/golden/*.npz) and seeded micro-matrices, written to a scratch
directory as `.jl` files by *this* script (joblib.dump of our own arrays) so that the reference's
`select_main` / `load_files` / `run_selection` / `greedy_select` / `calculate_scores` run end to end.

What is written (data only) to tests/golden/traces/<case>.json:
  argv / call description, the input parts (fixture names, or the packed GT/AF/samples of a micro-case),
  the TSV text the reference wrote, and per iteration the winner index and its float64 score as hex
  (captured by observing the array handed to `np.argmax` at select.py:48), for micro-cases the whole score
  vector.  `count_table.json` holds the `--count` resolution (select.py:157-159) observed through
  `run_selection`.

The hdf5 branch cannot run here (h5py absent): its numeric consequence -- presence*AF stored as float32,
select.py:218-223 -- is emulated by casting the in-memory `--af` matrix to float32 before `run_selection`,
and its compaction branch (select.py:116-137) is driven by an ndarray subclass registered as `h5py.Dataset`.
"""
import io
import json
import os
import sys
import tempfile
import types

import joblib
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")
OUT = os.path.join(GOLD, "traces")
REF = "/root/reference"


# ------------------------------------------------------------------ import the reference
class _Dataset(np.ndarray):
    """Marker type: `isinstance(matrix, h5py.Dataset)` (select.py:116) is true for views of this class."""


class _File(dict):
    """Marker type for `isinstance(data, h5py.File)` (select.py:163, :191)."""


def import_reference():
    h5 = types.ModuleType("h5py")
    h5.Dataset, h5.File = _Dataset, _File
    tru = types.ModuleType("truvari")
    tru.setup_logging = lambda *a, **k: None
    sys.modules.setdefault("h5py", h5)
    sys.modules.setdefault("truvari", tru)
    sys.modules.setdefault("allel", types.ModuleType("allel"))
    sys.path.insert(0, REF)
    import utmos.select as sel
    assert os.path.realpath(sel.__file__).startswith(REF), sel.__file__
    return sel


class ArgmaxTap:
    """Stands in for the name `np` inside the reference module: everything is numpy, `argmax` also records."""

    def __init__(self):
        self.seen = []

    def __getattr__(self, name):
        return getattr(np, name)

    def argmax(self, a, *args, **kw):
        self.seen.append(np.array(a, dtype=np.float64, copy=True))
        return np.argmax(a, *args, **kw)


# ------------------------------------------------------------------ helpers
def load_fixture(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {"GT": z["GT"], "AF": z["AF"].reshape(-1, 1), "samples": z["samples"].astype("S")}


def part_from_dense(dense, af, names):
    return {"GT": np.packbits(dense, axis=1), "AF": np.asarray(af, dtype=np.float64).reshape(-1, 1),
            "samples": np.asarray(names).astype("S")}


def part_json(part):
    return {"n_rows": int(part["GT"].shape[0]), "gt_hex": part["GT"].tobytes().hex(),
            "af_hex": [float(x).hex() for x in part["AF"].reshape(-1)],
            "samples": [s.decode() for s in part["samples"]]}


def write_jl(tmp, label, part):
    path = os.path.join(tmp, label + ".jl")
    joblib.dump(part, path)
    return path


def iter_trace(tap, tsv, names, full_vectors):
    """Winner index + score per emitted row; the last observed vector may belong to a row-less stop."""
    rows = [ln.split("\t") for ln in tsv.splitlines()[1:]]
    idx = [names.index(r[0]) for r in rows]
    out = {"idx": idx, "score_hex": [float(tap.seen[k][i]).hex() for k, i in enumerate(idx)],
           "argmax_calls": len(tap.seen)}
    if full_vectors:
        out["scores_hex"] = [[float(x).hex() for x in v] for v in tap.seen]
    return out


def run_cli(sel, tmp, parts, argv, full_vectors=False, maxmem=None):
    """The reference's `select_main` on .jl files written from `parts`."""
    files = [write_jl(tmp, f"p{i}", p) for i, p in enumerate(parts)]
    out = os.path.join(tmp, "out.tsv")
    tap = ArgmaxTap()
    sel.np = tap
    try:
        sel.select_main(list(argv) + (["--maxmem", str(maxmem)] if maxmem is not None else []) + ["-o", out] + files)
    finally:
        sel.np = np
        sel.MAXMEM = 2
    tsv = open(out).read()
    names = [s.decode() for s in parts[0]["samples"]]
    return tsv, iter_trace(tap, tsv, names, full_vectors)


def run_direct(sel, tmp, parts, af, cast_f32, count, state=None, weights=None, full_vectors=False,
               as_dataset=False, maxmem=None):
    """`load_files` then `greedy_select` (or `run_selection`) called directly: needed for an initial mask with
    used samples, the float32 emulation of the hdf5 store, and the compaction branch."""
    files = [write_jl(tmp, f"p{i}", p) for i, p in enumerate(parts)]
    data = sel.load_files(files, None, 32768, af)
    if cast_f32:
        data["data"] = data["data"].astype(np.float32)
    names = data["samples"].astype(str)
    n_samp = len(names)
    k = n_samp if count < 0 else max(1, int(n_samp * count) if count < 1 else int(count))
    mask = np.ones(n_samp, dtype="uint8") if state is None else np.array(state, dtype="uint8")
    matrix = data["data"].view(_Dataset) if as_dataset else data["data"]
    tap = ArgmaxTap()
    sel.np = tap
    if maxmem is not None:
        sel.MAXMEM = maxmem
    buf = io.StringIO()
    buf.write("sample\tvar_count\tnew_count\ttot_captured\tpct_captured\n")
    emitted = []
    try:
        w = None if weights is None else np.asarray(weights, dtype=np.float64)
        for row in sel.greedy_select(matrix, data["var_count"][:], k, names, mask, w):
            buf.write("\t".join(str(_) for _ in row) + "\n")
            emitted.append(row[0])
    finally:
        sel.np = np
        sel.MAXMEM = 2
    tsv = buf.getvalue()
    if as_dataset:      # after compaction the index space changes: keep names only
        return tsv, {"argmax_calls": len(tap.seen)}
    return tsv, iter_trace(tap, tsv, list(names), full_vectors)


def save(name, record):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name + ".json"), "w") as fh:
        json.dump(record, fh, indent=0, separators=(",", ":"))
    n_rows = record.get("tsv", "").count("\n") - 1 if "tsv" in record else None
    print(f"{name}: rows={n_rows}")


# ------------------------------------------------------------------ micro-matrices (seeded)
def micro(seed, n_var, n_samp, with_dups=False, zero_af_rows=0, blank_rows=0):
    rng = np.random.default_rng(seed)
    p = np.exp(rng.uniform(np.log(1.0 / n_samp), 0.0, size=n_var))[:, None]
    dense = rng.random((n_var, n_samp)) < p
    dense[np.arange(n_var), rng.integers(0, n_samp, size=n_var)] = True
    if with_dups:       # exact ties: identical columns at scattered positions
        dense[:, n_samp - 1] = dense[:, 2]
        dense[:, n_samp // 2] = dense[:, 2]
        dense[:, 1] = dense[:, 0]
    if blank_rows:      # uninformative rows the ingest must drop (select.py:276-279)
        dense[rng.choice(n_var, blank_rows, replace=False)] = False
    ac = dense.sum(axis=1) + rng.integers(0, 3, size=n_var)
    af = np.maximum(ac, 1) / (2.0 * n_samp)
    if zero_af_rows:    # informative rows whose AF is 0.0: all-zero rows of the float matrix
        af[rng.choice(n_var, zero_af_rows, replace=False)] = 0.0
    names = [f"S{i:03d}" for i in range(n_samp)] if n_samp <= 1000 else [f"S{i:04d}" for i in range(n_samp)]
    return part_from_dense(dense, af, names)


def main():
    sel = import_reference()
    fx = {n: load_fixture(n) for n in ("chunk0", "chunk1", "chunk2", "tiny")}
    three = ["chunk0", "chunk1", "chunk2"]
    wfile = os.path.join(GOLD, "weights.txt")
    with tempfile.TemporaryDirectory() as tmp:
        # ---- 1. select-all over the three fixture chunks, every value mode
        tsv, tr = run_cli(sel, tmp, [fx[n] for n in three], ["-c", "-1"])
        save("all_int", {"kind": "cli", "inputs": three, "argv": ["-c", "-1"], "tsv": tsv, **tr})
        tsv, tr = run_cli(sel, tmp, [fx[n] for n in three], ["-c", "-1", "--af"])
        save("all_af64", {"kind": "cli", "inputs": three, "argv": ["-c", "-1", "--af"], "tsv": tsv, **tr})
        tsv, tr = run_direct(sel, tmp, [fx[n] for n in three], True, True, -1)
        save("all_af32", {"kind": "direct", "inputs": three, "af": True, "af_dtype": "f32", "count": -1,
                          "tsv": tsv, **tr})
        tsv, tr = run_cli(sel, tmp, [fx[n] for n in three], ["-c", "-1", "--weights", wfile])
        save("all_weights", {"kind": "cli", "inputs": three, "argv": ["-c", "-1", "--weights", "weights.txt"],
                             "tsv": tsv, **tr})
        sub = os.path.join(GOLD, "subset.txt")
        argv = ["-c", "-1", "--subset", sub, "--exclude", "HG00096,NA21117,HG00280", "--af", "--weights", wfile]
        tsv, tr = run_cli(sel, tmp, [fx["chunk0"], fx["chunk2"]], argv)
        save("subset_exclude_af_weights",
             {"kind": "cli", "inputs": ["chunk0", "chunk2"],
              "argv": ["-c", "-1", "--subset", "subset.txt", "--exclude", "HG00096,NA21117,HG00280", "--af",
                       "--weights", "weights.txt"], "tsv": tsv, **tr})
        tsv, tr = run_cli(sel, tmp, [fx["tiny"]], ["-c", "-1"], full_vectors=True)
        save("tiny_all", {"kind": "cli", "inputs": ["tiny"], "argv": ["-c", "-1"], "tsv": tsv, **tr})
        tsv, tr = run_cli(sel, tmp, [fx["tiny"]], ["-c", "-1", "--af"], full_vectors=True)
        save("tiny_all_af", {"kind": "cli", "inputs": ["tiny"], "argv": ["-c", "-1", "--af"], "tsv": tsv, **tr})

        # ---- 2. the compaction branch (select.py:116-137) must not change rows
        for label, af in (("compaction_int", False), ("compaction_af32", True)):
            base, _ = run_direct(sel, tmp, [fx["chunk0"], fx["chunk1"]], af, af, 40)
            comp, tr = run_direct(sel, tmp, [fx["chunk0"], fx["chunk1"]], af, af, 40, as_dataset=True, maxmem=0)
            assert base == comp, label
            save(label, {"kind": "direct", "inputs": ["chunk0", "chunk1"], "af": af, "af_dtype": "f32" if af else None,
                         "count": 40, "tsv": comp, "note": "h5 compaction forced (MAXMEM=0); rows equal the "
                         "uncompacted run of the reference", **tr})

        # ---- 3. seeded micro-cases through the CLI
        m_ties = micro(11, 300, 24, with_dups=True)
        for label, argv in (("ties_int", ["-c", "-1"]), ("ties_af", ["-c", "-1", "--af"])):
            tsv, tr = run_cli(sel, tmp, [m_ties], argv, full_vectors=True)
            save(label, {"kind": "cli", "parts": [part_json(m_ties)], "argv": argv, "tsv": tsv, **tr})

        m_two = [micro(21, 257, 40, blank_rows=30), micro(22, 129, 40, blank_rows=5)]
        argv = ["-c", "-1", "--subset", "S003,S004,S005,S010,S011,S020,S039", "--exclude", "S004"]
        tsv, tr = run_cli(sel, tmp, m_two, argv, full_vectors=True)
        save("zero_score_stop", {"kind": "cli", "parts": [part_json(p) for p in m_two], "argv": argv, "tsv": tsv, **tr})

        argv = ["-c", "-1", "--exclude", ",".join(f"S{i:03d}" for i in range(40))]
        tsv, tr = run_cli(sel, tmp, m_two, argv, full_vectors=True)
        save("all_excluded", {"kind": "cli", "parts": [part_json(p) for p in m_two], "argv": argv, "tsv": tsv, **tr})

        m_af0 = micro(31, 400, 33, zero_af_rows=60, blank_rows=10)
        for label, argv in (("af_zero_rows", ["-c", "-1", "--af"]), ("af_zero_rows_int", ["-c", "-1"])):
            tsv, tr = run_cli(sel, tmp, [m_af0], argv, full_vectors=True)
            save(label, {"kind": "cli", "parts": [part_json(m_af0)], "argv": argv, "tsv": tsv, **tr})
        tsv, tr = run_direct(sel, tmp, [m_af0], True, True, -1, full_vectors=True)
        save("af_zero_rows_f32", {"kind": "direct", "parts": [part_json(m_af0)], "af": True, "af_dtype": "f32",
                                  "count": -1, "tsv": tsv, **tr})

        wtxt = os.path.join(tmp, "w.txt")
        wl = [("S000", -2.0), ("S001", 0.0), ("S002", 0.5), ("S007", 3.25), ("S008", -0.0), ("S030", 1e-3),
              ("NOPE", 9.0)]
        with open(wtxt, "w") as fh:
            fh.write("".join(f"{k}\t{v!r}\n" for k, v in wl))
        m_w = micro(41, 350, 32)
        for label, extra in (("weights_signed_int", []), ("weights_signed_af", ["--af"])):
            argv = ["-c", "-1", "--weights", wtxt] + extra
            tsv, tr = run_cli(sel, tmp, [m_w], argv, full_vectors=True)
            save(label, {"kind": "cli", "parts": [part_json(m_w)], "weights": wl,
                         "argv": ["-c", "-1", "--weights", "<weights>"] + extra, "tsv": tsv, **tr})
        # only negative weights among the selectable: np.argmax then lands on a masked 0 -> stop without a row
        wneg = [(f"S{i:03d}", -1.0 - i) for i in range(32)]
        with open(wtxt, "w") as fh:
            fh.write("".join(f"{k}\t{v!r}\n" for k, v in wneg))
        argv = ["-c", "-1", "--weights", wtxt, "--exclude", "S005"]
        tsv, tr = run_cli(sel, tmp, [m_w], argv, full_vectors=True)
        save("weights_all_negative_with_masked", {"kind": "cli", "parts": [part_json(m_w)], "weights": wneg,
             "argv": ["-c", "-1", "--weights", "<weights>", "--exclude", "S005"], "tsv": tsv, **tr})
        argv = ["-c", "-1", "--weights", wtxt]
        tsv, tr = run_cli(sel, tmp, [m_w], argv, full_vectors=True)
        save("weights_all_negative", {"kind": "cli", "parts": [part_json(m_w)], "weights": wneg,
             "argv": ["-c", "-1", "--weights", "<weights>"], "tsv": tsv, **tr})

        # ---- 4. initially used samples (mask == 0 covers, select.py:36-39): greedy_select called directly
        m_u = micro(51, 320, 28, blank_rows=8)
        state = np.ones(28, dtype=np.uint8)
        state[[3, 17]] = 0
        state[[5, 6]] = 2
        for label, af, f32 in (("initial_used_int", False, False), ("initial_used_af64", True, False),
                               ("initial_used_af32", True, True)):
            tsv, tr = run_direct(sel, tmp, [m_u], af, f32, -1, state=state, full_vectors=True)
            save(label, {"kind": "direct", "parts": [part_json(m_u)], "af": af,
                         "af_dtype": ("f32" if f32 else "f64") if af else None, "count": -1,
                         "state": state.tolist(), "tsv": tsv, **tr})
        wv = np.ones(28)
        wv[[0, 9, 20]] = [2.5, 0.0, -1.0]
        tsv, tr = run_direct(sel, tmp, [m_u], True, False, 12, state=state, weights=wv, full_vectors=True)
        save("initial_used_af64_weights", {"kind": "direct", "parts": [part_json(m_u)], "af": True, "af_dtype": "f64",
             "count": 12, "state": state.tolist(), "weight_vector_hex": [float(x).hex() for x in wv], "tsv": tsv, **tr})

        # ---- 4b. a matrix wide enough for several waves / picker rounds on the device (1,500 samples), select all
        m_mid = micro(61, 4000, 1500)    # (kept as a compressed fixture of its own, like the chunk re-encodings: tests/golden/mid.npz)
        np.savez_compressed(os.path.join(GOLD, "mid.npz"), GT=m_mid["GT"], AF=m_mid["AF"].reshape(-1),
                            samples=np.asarray([x.decode() for x in m_mid["samples"]], dtype="U"))
        for label, argv in (("mid_int", ["-c", "-1"]), ("mid_af", ["-c", "-1", "--af"])):
            tsv, tr = run_cli(sel, tmp, [m_mid], argv)
            save(label, {"kind": "cli", "inputs": ["mid"], "argv": argv, "tsv": tsv, **tr})
        # ... with AF, a weight on every third sample (positive, all different) and a few samples excluded: weighted
        # float64 scores at a width where the device's candidate lists, chains and deferred scores all come into play
        rng_w = np.random.default_rng(62)
        names_mid = [x.decode() for x in m_mid["samples"]]
        wl_mid = [(names_mid[i], float(np.round(rng_w.uniform(0.25, 4.0), 6))) for i in range(0, len(names_mid), 3)]
        with open(wtxt, "w") as fh:
            fh.write("".join(f"{k}\t{v!r}\n" for k, v in wl_mid))
        excl = ",".join(names_mid[i] for i in (5, 77, 700, 1499))
        argv = ["-c", "-1", "--af", "--weights", wtxt, "--exclude", excl]
        tsv, tr = run_cli(sel, tmp, [m_mid], argv)
        save("mid_af_weights_exclude", {"kind": "cli", "inputs": ["mid"], "weights": wl_mid,
                                        "argv": ["-c", "-1", "--af", "--weights", "<weights>", "--exclude", excl], "tsv": tsv, **tr})
        # ... and the hdf5 flavour of the AF values (float32) for a fixed number of picks
        tsv, tr = run_direct(sel, tmp, [m_mid], True, True, 300)
        save("mid_af32_c300", {"kind": "direct", "inputs": ["mid"], "af": True, "af_dtype": "f32", "count": 300, "tsv": tsv, **tr})

        # ---- 5. --count resolution as run_selection applies it (select.py:157-159)
        table = []
        seen = {}
        orig = sel.greedy_select
        sel.greedy_select = lambda m, vc, k, *a, **kw: seen.setdefault("k", k) and iter(())
        try:
            for n_samp in (1, 7, 36, 100, 2504):
                data = {"data": np.zeros((3, n_samp), dtype=bool), "samples": np.array([b"x"] * n_samp),
                        "var_count": np.zeros(n_samp, dtype=np.int64)}
                for c in (-1.0, -0.5, 0.0, 0.0001, 0.005, 0.02, 0.5, 0.999, 1.0, 1.5, 2.0, 10.0, 20.7, 5000.0):
                    seen.clear()
                    list(sel.run_selection(data, c) or ())
                    table.append([n_samp, c, int(seen["k"])])
        finally:
            sel.greedy_select = orig
        with open(os.path.join(OUT, "count_table.json"), "w") as fh:
            json.dump({"source": "run_selection (select.py:157-159) observed", "n_samp,count,k": table}, fh)
        print("count_table:", len(table))


if __name__ == "__main__":
    main()
