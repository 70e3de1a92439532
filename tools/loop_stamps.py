#!/usr/bin/env python3
"""Device timestamps of the persistent loop's hand-off (debug build only):
    make -C utmos_amd/csrc -B EXTRA=-DUTM_DEBUG_STAMPS && python tools/loop_stamps.py 1103547 2504
prints, per launch of a select-all run, the mean over the launch's iterations of each stage of the hand-off relative to
the record's publication, and for iteration 100 of the launch how the waves' finishing times spread (s_memrealtime, 10 ns
ticks).  Rebuild without the flag afterwards: the stamped library exports symbols include/utmos_hip.h does not declare."""
import ctypes
import os
import sys
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
import numpy as np
from utmos_amd import device, _native as nat
lib = nat.lib()
lib.utm_dbg_loop_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
n_var, n_samp = int(sys.argv[1]), int(sys.argv[2])
m = device.DeviceMatrix(n_samp)
c = m.add_chunk(n_var)
m.synth_fill(c, seed=0)
m.reset()
out = (ctypes.c_uint64 * (256 * 16))()
wt = (ctypes.c_uint64 * (2 * 8192))()
done = 0
while done < n_samp:
    got = m.run(256)
    n = len(got[0])
    if n == 0:
        break
    lib.utm_dbg_loop_stamps(m._h, out, wt)
    t = np.array(list(out), dtype=np.float64).reshape(256, 16)[:n - 1] * 0.01       # us; the last iteration of a launch has no successor
    w = np.array(list(wt), dtype=np.float64).reshape(2, 8192) * 0.01
    if len(t) > 4:
        pub = t[:, 1]
        length = np.diff(pub)
        line = (f"iterations {done}..{done + n}: length {length.mean():.2f} us | relative to the record's publication: "
                f"picker saw all words {np.mean(t[:, 0] - pub):.2f}, block 1 saw the record {np.mean(t[:, 3] - pub):.2f}, "
                f"tile updated {np.mean(t[:, 4] - pub):.2f}, first batch counted {np.mean((t[:, 6] - pub)[:-1]):.2f}, "
                f"block 1 wave 0 had finished its positions {np.mean(t[:, 2] - pub):.2f}")
        if n > 101:
            ok = (w[0] > 0) & (w[1] > 0)
            start, fin = w[0][ok], w[1][ok]
            p100 = t[100, 1]
            d = fin - p100
            line += (f" | iteration 100: {ok.sum()} waves, last partial relative to its publication: min {d.min():.2f} p10 {np.percentile(d, 10):.2f} "
                     f"median {np.median(d):.2f} p90 {np.percentile(d, 90):.2f} max {d.max():.2f}; wave busy time median {np.median(fin - start):.2f}")
        print(line)
        if done == 0 and n > 101:
            np.save(os.path.join(root, "gpurun_out", f"loop_wave_t_{n_var}.npy"), np.array(list(wt), dtype=np.uint64).reshape(2, 8192))
            np.save(os.path.join(root, "gpurun_out", f"loop_stamps_{n_var}.npy"), np.array(list(out), dtype=np.uint64).reshape(256, 16))
    done += n
