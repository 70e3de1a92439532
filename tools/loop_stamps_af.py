#!/usr/bin/env python3
"""Device timestamps of the interval form's picker (debug build only, as tools/loop_stamps.py):
    make -C utmos_amd/csrc -B EXTRA=-DUTM_DEBUG_STAMPS && python tools/loop_stamps_af.py 1103547 2504
per launch of a float64-AF select-all run: iteration length, and -- relative to the moment the picker saw all words -- when
the candidates were listed, the records looked up, the last request sent to the chainer, every sum known, and the record's
publication, for the iterations with and without overlapping intervals."""
import ctypes
import os
import sys
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
import numpy as np
import bench
from utmos_amd import device, _native as nat
lib = nat.lib()
lib.utm_dbg_loop_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
n_var, n_samp = int(sys.argv[1]), int(sys.argv[2])
spec = dict(n_var=n_var, n_samp=n_samp, select=-1, af=True, af_dtype="f64", chunk_vars=0, seed=0)
m, _ = bench.build_matrix(device, spec, 0)
m.reset()
out = (ctypes.c_uint64 * (256 * 16))()
done = 0
while done < n_samp:
    got = m.run(64)
    n = len(got[0])
    if n == 0:
        break
    st = m.stats()
    lib.utm_dbg_loop_stamps(m._h, out, None)
    t = np.array(list(out), dtype=np.float64).reshape(256, 16)[:n] * 0.01
    if n > 8 and st["persist_iterations"] > 0:
        seen, pub = t[:, 0], t[:, 1]
        chained = t[:, 11] > seen
        length = np.diff(pub)
        def rel(col, sel):
            return f"{np.mean((t[:, col] - seen)[sel]):.2f}" if sel.any() else "-"
        print(f"rows {done}..{done + n}: length {length.mean():.2f} us (chained {np.mean(length[chained[1:]]) if chained[1:].any() else 0:.2f}, "
              f"plain {np.mean(length[~chained[1:]]) if (~chained[1:]).any() else 0:.2f}) | {chained.sum()} chained | from all words seen: candidates "
              f"{rel(8, ~chained)} / {rel(8, chained)}, records looked up {rel(9, chained)}, last request {rel(10, chained & (t[:, 10] > 0))} ({(chained & (t[:, 10] > 0)).sum()} asked), sums known {rel(11, chained)}, "
              f"published {rel(1, ~chained)} / {rel(1, chained)}")
    done += n
