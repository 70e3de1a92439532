#!/usr/bin/env python3
"""Where in a float64-AF run do the chained iterations (candidates the intervals cannot separate) fall?
Runs the synthetic workload in slices of 64 iterations and prints the chained-iteration count per slice.
usage: python tools/chain_histogram.py [n_var] [n_samp]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from utmos_amd import device  # noqa: E402

n_var = int(sys.argv[1]) if len(sys.argv) > 1 else 1103547
n_samp = int(sys.argv[2]) if len(sys.argv) > 2 else 2504
spec = dict(n_var=n_var, n_samp=n_samp, select=-1, af=True, af_dtype="f64", chunk_vars=0, seed=0)
m, _ = bench.build_matrix(device, spec, 0)
m.reset()
prev = 0
rows = 0
line = []
while rows < n_samp:
    idx, new, score = m.run(64)
    if len(idx) == 0:
        break
    rows += len(idx)
    st = m.stats()
    ch = st["af_chained_iterations"]
    line.append((rows, ch - prev, int(new[-1])))
    prev = ch
print("rows_done chained_in_slice last_gain")
for r, c, g in line:
    print(f"{r:6d} {c:3d} {g}")
print("total chained", prev, "of", rows)
