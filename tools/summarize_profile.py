#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small files committed under profiles/.

  kernel-trace --stats dir  ->  <tag>_kernel_stats.csv   (rocprofv3's own per-kernel summary, verbatim)
  --pmc FETCH_SIZE dir, --pmc WRITE_SIZE dir -> <tag>_pmc_hbm.json: per-launch HBM bytes of the scoring kernel

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read,
so the read side is doubled; WRITE_SIZE is taken as is.  Counters are collected in their own passes.
"""
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    """Several runs may have written into one directory (PID-prefixed files): take the latest."""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def pmc_rows(d, prefix):
    f = newest(f"{d}/**/*_counter_collection.csv")
    return [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(tuple(prefix.split(",")))]


def main():
    tag, kt_dir, fetch_dir, write_dir, algo_note = sys.argv[1:6]
    prefix = sys.argv[6] if len(sys.argv) > 6 else "void k_score_int"
    stats = newest(f"{kt_dir}/**/*_kernel_stats.csv")
    shutil.copyfile(stats, f"profiles/{tag}_kernel_stats.csv")
    fr, wr = pmc_rows(fetch_dir, prefix), pmc_rows(write_dir, prefix)
    n = min(len(fr), len(wr))
    fetch_b = [float(r["Counter_Value"]) * 1024 * 2 for r in fr[:n]]
    write_b = [float(r["Counter_Value"]) * 1024 for r in wr[:n]]
    out = {
        "kernel": prefix, "launches": n,
        "correction": "read bytes = FETCH_SIZE KiB * 1024 * 2 (gfx950 wide-load undercount); write bytes = WRITE_SIZE KiB * 1024",
        "hbm_read_bytes_per_launch_mean": sum(fetch_b) / n, "hbm_write_bytes_per_launch_mean": sum(write_b) / n,
        "hbm_bytes_per_launch_mean": (sum(fetch_b) + sum(write_b)) / n,
        "first_launch": {"read": fetch_b[0], "write": write_b[0]}, "last_launch": {"read": fetch_b[-1], "write": write_b[-1]},
        "workload": algo_note,
    }
    json.dump(out, open(f"profiles/{tag}_pmc_hbm.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
