#!/usr/bin/env python3
"""End-to-end timing of `python -m utmos_amd select` on synthetic multi-file input (GPU box).
usage: tools/cli_e2e_bench.py [n_parts] [variants_per_part] [n_samples] [uniform|sfs]
Writes parts to $TMPDIR, runs the CLI.  uniform: every sample alike (the run ends after ~80 samples, AFs are random
doubles: a table too wide for a lossless fixed-point unit); sfs: the bench generator's site-frequency spectrum with
AF = carriers / 2S as float64 (what real inputs look like)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n_parts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n_var = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    n_samp = int(sys.argv[3]) if len(sys.argv) > 3 else 2504
    kind = sys.argv[4] if len(sys.argv) > 4 else "uniform"
    work = tempfile.mkdtemp(prefix="utm_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    rng = np.random.default_rng(0)
    samples = np.array([f"S{i:05d}" for i in range(n_samp)])
    files = []
    t = time.time()
    for p in range(n_parts):
        width = (n_samp + 7) // 8
        if kind == "sfs":
            sys.path.insert(0, ROOT)
            from utmos_amd import device
            cols, af32 = device.synth_host(p, n_var, n_samp, first_var_global=p * n_var)
            bits = np.unpackbits(cols.view(np.uint8), axis=1, bitorder="little")[:, :n_var]   # (S, n_var)
            gt = np.packbits(bits.T, axis=1)                                                   # rows, MSB first
            del bits, cols
            af = af32.astype(np.float64) / 3.0 * 3.0001                                        # full 53-bit mantissas
        else:
            gt = rng.integers(0, 256, (n_var, width), dtype=np.uint8)
            gt &= rng.integers(0, 256, (n_var, width), dtype=np.uint8)
            gt &= rng.integers(0, 256, (n_var, width), dtype=np.uint8)          # ~12 % carriers
            if n_samp % 8:
                gt[:, -1] &= np.uint8((0xFF << (8 - n_samp % 8)) & 0xFF)         # packbits padding stays zero
            af = rng.random(n_var)
        path = os.path.join(work, f"part{p}.npz")
        np.savez(path, GT=gt, AF=af, samples=samples)
        files.append(path)
    print(f"wrote {n_parts} parts x {n_var} variants x {n_samp} samples in {time.time() - t:.1f} s", flush=True)
    count = os.environ.get("UTM_E2E_COUNT", "200")          # samples to select (-1 = all)
    for extra in ([], ["--af"], ["--brute-force"]):
        out = os.path.join(work, "out.tsv")
        t = time.time()
        run = subprocess.run([sys.executable, "-m", "utmos_amd", "select", "-c", count, "-o", out] + extra + files,
                             cwd=ROOT, capture_output=True, text=True)
        dt = time.time() - t
        rows = sum(1 for _ in open(out)) - 1 if os.path.exists(out) else -1
        print(f"select -c {count} {' '.join(extra):14s}: {dt:6.2f} s wall, {rows} rows, rc {run.returncode}", flush=True)
        if run.returncode:
            print(run.stderr[-600:])
    subprocess.run(["rm", "-rf", work])


if __name__ == "__main__":
    main()
