#!/usr/bin/env python3
"""Sweep the launch-shape knobs of the scoring kernel (env vars read by libutmos_hip.so) on the GPU box.
usage: tools/tune.py [bench args...]   -- prints one line per combination."""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRID = {
    "UTM_TILE_STEPS": ["0", "8", "4", "2"],
    "UTM_TARGET_WGS": ["8192", "16384", "32768", "65536"],
    "UTM_MIN_WGS": ["1024"],
}
if os.environ.get("TUNE_GRID"):      # e.g. TUNE_GRID='{"UTM_MIN_WGS": ["256", "4096"]}'
    GRID = json.loads(os.environ["TUNE_GRID"])


def main():
    extra = sys.argv[1:]
    keys = list(GRID)
    for combo in itertools.product(*(GRID[k] for k in keys)):
        env = dict(os.environ, **dict(zip(keys, combo)))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                              "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            r = j["roofline"] or {}
            print(dict(zip(keys, combo)), f"it/s={j['value']:.1f} ms/step={j['ms_per_step']:.1f} "
                  f"loop_frac={j['hbm_frac_whole_loop']:.4f} kernel_frac={r.get('frac', 0):.4f}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(dict(zip(keys, combo)), "FAILED", e, out.stderr[-300:], flush=True)


if __name__ == "__main__":
    main()
