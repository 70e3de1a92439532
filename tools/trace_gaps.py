#!/usr/bin/env python3
"""Where a greedy run's stream time goes, from a rocprofv3 --kernel-trace CSV: per kernel the launch count, mean
duration and the mean idle gap in front of it, then the same per decile of the run (iterations get cheaper as the
selectable set shrinks, so the fixed per-iteration costs weigh most at the end).
usage: tools/trace_gaps.py <dir or *_kernel_trace.csv> [skip_launches]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    src = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if os.path.isdir(src):
        src = max(glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(src))))
    rows = rows[skip:]
    per = defaultdict(lambda: [0, 0, 0])
    prev_end = None
    for s, e, n in rows:
        p = per[n]
        p[0] += 1
        p[1] += e - s
        if prev_end is not None:
            p[2] += max(0, s - prev_end)
        prev_end = max(prev_end or 0, e)
    span = rows[-1][1] - rows[0][0]
    print(f"{src}: {len(rows)} launches, span {span / 1e6:.3f} ms")
    print(f"{'kernel':48s} {'n':>7s} {'mean us':>9s} {'gap us':>8s} {'% span':>7s}")
    for n, (c, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{n[:48]:48s} {c:7d} {d / c / 1e3:9.2f} {g / c / 1e3:8.2f} {100.0 * (d + g) / span:7.2f}")
    print("deciles of the run (busy us / gap us per launch, by kernel):")
    step = max(1, len(rows) // 10)
    for d in range(10):
        part = rows[d * step:(d + 1) * step]
        acc = defaultdict(lambda: [0, 0, 0])
        pe = rows[d * step - 1][1] if d else None
        for s, e, n in part:
            a = acc[n]
            a[0] += 1
            a[1] += e - s
            if pe is not None:
                a[2] += max(0, s - pe)
            pe = max(pe or 0, e)
        print(f"  {d}: " + "  ".join(f"{n[:22]} {a[1] / a[0] / 1e3:.2f}/{a[2] / a[0] / 1e3:.2f}" for n, a in sorted(acc.items())))


if __name__ == "__main__":
    main()
