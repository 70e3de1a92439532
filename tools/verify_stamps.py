#!/usr/bin/env python3
"""Device timestamps of k_verify's stages on chained iterations (debug build only):
    make -C utmos_amd/csrc -B EXTRA=-DUTM_DEBUG_STAMPS && python tools/verify_stamps.py 10000000
prints, per point of the run, the mean time since workgroup 0 started at which each stage was reached (s_memrealtime, 10 ns
ticks).  Rebuild without the flag afterwards: the stamped library exports one symbol more than include/utmos_hip.h declares."""
import sys, os, ctypes
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
import numpy as np
from utmos_amd import device, _native as nat
lib = nat.lib()
lib.utm_dbg_verify_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
n_var, n_samp = int(sys.argv[1]), 2504
m = device.DeviceMatrix(n_samp)
c = m.add_chunk(n_var); m.synth_fill(c, seed=0)
_, af = device.synth_host(0, n_var, n_samp, want_cols=False)
m.set_af(c, af.astype(np.float64) / 3.0)
m.reset()
out = (ctypes.c_uint64 * 16)()
names = ["b0 start", "cand_list done", "released", "fill0 start", "fill0 done", "chain0 start", "chain0 filled", "chain0 acquired", "chain0 chained", "last arrived", "last acquired", "picked"]
for target in (300, 1500, 2200):
    while m.stats()["iterations"] < target:
        m.run(target - m.stats()["iterations"])
    rows = []
    for _ in range(40):
        ev0 = m.stats()["af_chained_iterations"]
        m.step()
        if m.stats()["af_chained_iterations"] == ev0:
            continue
        lib.utm_dbg_verify_stamps(m._h, out)
        t = np.array(list(out)[:12], dtype=np.float64)
        rows.append((t - t[0]) * 0.01)
    if rows:
        r = np.array(rows).mean(axis=0)
        print(f"iter ~{target}: {len(rows)} chained iterations; us since workgroup 0 started: " + ", ".join(f"{n} {v:.2f}" for n, v in zip(names, r)))
