#!/bin/bash
# Per-kernel A/B on ONE box: tools/ab_kernels.sh "ab/a.so ab/b.so" workload "k_chain k_cand ..."
# Runs one bench step of `workload` under rocprofv3 --kernel-trace --stats per library variant and prints the average
# duration of the kernels whose names start with the given words.
R=${GRAFT_REPO_ROOT:-$(pwd)}
libs="$1"; w="$2"; names="$3"
cd /tmp && export TMPDIR=/tmp
for lib in $libs; do
  cp $R/$lib $R/utmos_amd/libutmos_hip.so
  d=$R/gpurun_out/abk_$(basename $lib .so)
  rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-roofline-pass --pmc-traffic off > /dev/null 2>&1 || exit 1
  f=$(find $d -name "*kernel_stats.csv")
  for n in $names; do
    python3 - "$f" "$n" "$lib" <<'PY'
import csv, sys
f, n, lib = sys.argv[1:]
for r in csv.DictReader(open(f)):
    if r["Name"].replace("void ", "").startswith(n):
        print(f"{lib} {r['Name'][:40]:40s} calls={r['Calls']} avg_us={float(r['AverageNs'])/1e3:.2f} total_ms={float(r['TotalDurationNs'])/1e6:.2f}")
PY
  done
  find $d -name "*.csv" -size +3M -delete
done
