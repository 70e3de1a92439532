"""Test-side access to the CPU oracles (oracle/ is test infrastructure; the product never imports it)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import utmos_oracle as npo  # noqa: E402

_LIB = None


def oracle_lib(omp=False):
    global _LIB
    name = "liboracle_bitset_omp.so" if omp else "liboracle_bitset.so"
    path = os.path.join(ROOT, "oracle", name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(path)
    p = ctypes.c_void_p
    lib.orc_score.restype = ctypes.c_int64
    lib.orc_score.argtypes = [p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, p, p, ctypes.c_int, p, p, p]
    lib.orc_greedy.restype = ctypes.c_int64
    lib.orc_greedy.argtypes = [p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, p, p, ctypes.c_int, p,
                               ctypes.c_int64, p, p, p]
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _af_args(af):
    if af is None:
        return 0, None
    af = np.ascontiguousarray(af)
    if af.dtype == np.float32:
        return 1, af
    if af.dtype == np.float64:
        return 2, af
    raise TypeError(af.dtype)


def c_greedy(cols, n_var, state, weights=None, af=None, k_max=None, omp=False):
    """cols: uint64 (S, W) C-contiguous.  Returns (idx, new, score) arrays; state is copied."""
    lib = oracle_lib(omp)
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    n_samp, stride = cols.shape
    st = np.array(state, dtype=np.uint8, copy=True)
    k_max = n_samp if k_max is None else int(k_max)
    idx = np.zeros(max(k_max, 1), dtype=np.int64)
    new = np.zeros(max(k_max, 1), dtype=np.int64)
    sc = np.zeros(max(k_max, 1), dtype=np.float64)
    mode, afa = _af_args(af)
    if afa is not None:  # the C side reads whole 64-variant words
        pad = stride * 64 - len(afa)
        afa = np.concatenate([afa, np.zeros(pad, dtype=afa.dtype)])
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    n = lib.orc_greedy(_ptr(cols), stride, n_var, n_samp, _ptr(st), _ptr(w), mode, _ptr(afa),
                       k_max, _ptr(idx), _ptr(new), _ptr(sc))
    return idx[:n].copy(), new[:n].copy(), sc[:n].copy()


def c_score(cols, n_var, state, weights=None, af=None, omp=False):
    lib = oracle_lib(omp)
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    n_samp, stride = cols.shape
    st = np.ascontiguousarray(state, dtype=np.uint8)
    cnt = np.zeros(n_samp, dtype=np.int64)
    sc = np.zeros(n_samp, dtype=np.float64)
    mode, afa = _af_args(af)
    if afa is not None:
        afa = np.concatenate([afa, np.zeros(stride * 64 - len(afa), dtype=afa.dtype)])
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    best = lib.orc_score(_ptr(cols), stride, n_var, n_samp, _ptr(st), _ptr(w), mode, _ptr(afa), _ptr(cnt), _ptr(sc))
    return best, cnt, sc


# ----------------------------------------------------------------- golden cases
def load_part(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {"GT": z["GT"], "AF": z["AF"], "samples": z["samples"]}


def golden_cases():
    return json.load(open(os.path.join(GOLD, "cases.json")))


def read_list(path):
    return [ln.strip() for ln in open(path)]


def read_weights(path):
    out = {}
    for ln in open(path):
        if ln.strip():
            k, v = ln.rstrip("\n").split("\t")
            out[k] = float(v)
    return out


def case_kwargs(args):
    """Translate a case's CLI tail into oracle keyword arguments."""
    kw = {}
    i = 0
    while i < len(args):
        a = args[i]
        if a in ("-c", "--count"):
            kw["count"] = float(args[i + 1]); i += 2
        elif a == "--af":
            kw["af"] = True; i += 1
        elif a == "--af-dtype":
            kw["af_dtype"] = args[i + 1]; i += 2
        elif a == "--exclude":
            kw["exclude"] = args[i + 1].split(","); i += 2
        elif a == "--subset":
            kw["subset"] = read_list(os.path.join(GOLD, args[i + 1])); i += 2
        elif a == "--weights":
            kw["weights"] = read_weights(os.path.join(GOLD, args[i + 1])); i += 2
        else:
            raise ValueError(a)
    return kw


def golden_text(case):
    return open(os.path.join(GOLD, "answer_key", case["golden"])).read()


# ----------------------------------------------------------------- random matrices
def random_dense(rng, n_var, n_samp, density=0.05, sfs=True):
    """bool (N, S) with every row informative; 1/c-like carrier spectrum when sfs."""
    if sfs:
        p = np.exp(rng.uniform(np.log(1.0 / n_samp), 0.0, size=n_var))[:, None]
    else:
        p = density
    m = rng.random((n_var, n_samp)) < p
    forced = rng.integers(0, n_samp, size=n_var)
    m[np.arange(n_var), forced] = True
    return m


# ----------------------------------------------------------------- reference-generated traces
TRACES = os.path.join(GOLD, "traces")


def trace_names():
    return sorted(f[:-5] for f in os.listdir(TRACES) if f.endswith(".json") and f != "count_table.json")


def load_trace(name):
    """A record written by tools/make_traces.py: outputs of the REFERENCE's own code (data only)."""
    t = json.load(open(os.path.join(TRACES, name + ".json")))
    if "inputs" in t:
        t["part_list"] = [load_part(n) for n in t["inputs"]]
    else:
        t["part_list"] = []
        for p in t["parts"]:
            n_samp = len(p["samples"])
            gt = np.frombuffer(bytes.fromhex(p["gt_hex"]), dtype=np.uint8).reshape(p["n_rows"], (n_samp + 7) // 8)
            t["part_list"].append({"GT": gt, "AF": np.array([float.fromhex(x) for x in p["af_hex"]]),
                                   "samples": np.array(p["samples"])})
    return t


def trace_options(t, tmp_dir=None):
    """Oracle keyword arguments of a trace (its argv for CLI traces, its recorded call otherwise)."""
    if t["kind"] == "direct":
        return {"count": t["count"], "af": bool(t["af"]), "af_dtype": t.get("af_dtype") or "f64"}
    kw, argv, i = {}, t["argv"], 0
    while i < len(argv):
        a = argv[i]
        if a == "-c":
            kw["count"] = float(argv[i + 1]); i += 2
        elif a == "--af":
            kw["af"] = True; i += 1
        elif a == "--exclude":
            kw["exclude"] = argv[i + 1].split(","); i += 2
        elif a == "--subset":
            v = argv[i + 1]
            kw["subset"] = read_list(os.path.join(GOLD, v)) if v.endswith(".txt") else v.split(","); i += 2
        elif a == "--weights":
            v = argv[i + 1]
            kw["weights"] = read_weights(os.path.join(GOLD, v)) if v.endswith(".txt") else {k: w for k, w in t["weights"]}
            i += 2
        else:
            raise ValueError(a)
    return kw
