#!/usr/bin/env python3
"""SQ / TCC counters of the dominant scoring kernel, per launch (means), from separate `rocprofv3 --pmc` passes -- one pass
per counter group, nothing else enabled, the program itself behind `--`:

    python3 tools/pmc_sq.py <tag> <kernel name prefix> -- <bench.py flags>      ->  profiles/<tag>_pmc_sq.json

Run on the GPU box from the repo root (through gpurun).  Counter groups are sized for the PMC slots MI355X_MICROARCH.md
lists; a group the profiler rejects is reported under "failed" and skipped."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

GROUPS = {
    "sq1": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    "sq2": ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"],
    "tcc": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"],
    "grbm": ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
}


def main():
    tag, prefix = sys.argv[1], sys.argv[2]
    flags = sys.argv[sys.argv.index("--") + 1:]
    root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = {"kernel": prefix, "workload": "bench.py " + " ".join(flags) + " (per-launch means; one rocprofv3 --pmc pass per counter group)",
           "counters_per_launch": {}, "failed": []}
    for name, counters in GROUPS.items():
        d = tempfile.mkdtemp(prefix="utm_sq_", dir="/tmp")
        try:
            run = subprocess.run([exe, "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", sys.executable,
                                 os.path.join(root, "bench.py")] + flags, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                                 capture_output=True, text=True, timeout=900)
            files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if run.returncode != 0 or not files:
                out["failed"].append({name: (run.stderr or "")[-300:]})
                continue
            sums, launches = {}, {}
            with open(max(files, key=os.path.getmtime)) as fh:
                for r in csv.DictReader(fh):
                    if not r["Kernel_Name"].replace("void ", "").startswith(prefix):
                        continue
                    c = r["Counter_Name"]
                    sums[c] = sums.get(c, 0.0) + float(r["Counter_Value"])
                    launches[c] = launches.get(c, 0) + 1
            for c in sums:
                out["counters_per_launch"][c] = sums[c] / launches[c]
            out["counters_per_launch"]["launches_p_" + name] = max(launches.values()) if launches else 0
        finally:
            shutil.rmtree(d, ignore_errors=True)
    c = out["counters_per_launch"]
    reading = {}
    if c.get("SQ_WAVE_CYCLES"):
        reading["memory_bound"] = (f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f} of the wave time parked on s_waitcnt; "
                                   f"SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {c.get('SQ_ACTIVE_INST_VALU', 0) / c['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_LDS_BANK_CONFLICT" in c:
        reading["lds"] = f"SQ_LDS_BANK_CONFLICT = {c['SQ_LDS_BANK_CONFLICT']:.0f} per launch"
    if c.get("TCC_REQ_sum"):
        reading["l2"] = f"TCC_HIT / TCC_REQ = {c.get('TCC_HIT_sum', 0) / c['TCC_REQ_sum']:.3f}"
    if c.get("SQ_WAVES") and c.get("SQ_WAVE_CYCLES") and c.get("SQ_BUSY_CYCLES"):
        reading["occupancy"] = (f"SQ_WAVES = {c['SQ_WAVES']:.0f} per launch; mean resident waves = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES = "
                                f"{c['SQ_WAVE_CYCLES'] / c['SQ_BUSY_CYCLES']:.1f} per SQ-busy cycle (summed over the shader engines the counter covers)")
    out["reading"] = reading
    path = os.path.join(root, "profiles", f"{tag}_pmc_sq.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
