#!/bin/bash
# The full dense AF pass on ONE box: table lookups (k_score_aft, UTM_AF_TABLES=1) against the bit-walking kernel
# (k_score_afq, UTM_AF_TABLES=0).  tools/ab_af_first_pass.sh [n_var] [n_samp] [extra bench flags]
# One short bench run per setting under rocprofv3 --kernel-trace --stats; prints both kernels' durations.
R=${GRAFT_REPO_ROOT:-$(pwd)}
nv=${1:-10000000}; ns=${2:-2504}; shift 2
cd /tmp && export TMPDIR=/tmp
for t in 1 0; do
  d=$R/gpurun_out/af_first_pass_t$t
  rm -rf $d
  UTM_AF_TABLES=$t rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --af --n-var $nv --n-samp $ns --select 6 --steps 3 --warmup 1 \
      --no-cpu-baseline --no-roofline-pass --no-calibration --no-also --pmc-traffic off "$@" > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  f=$(find $d -name "*kernel_stats.csv")
  python3 - "$f" "$t" "$nv" "$ns" "$@" <<'PY'
import csv, sys
f, t, nv, ns = sys.argv[1:5]
if "--chunk-vars" in sys.argv:  # a launch covers one chunk
    nv = min(int(nv), int(sys.argv[sys.argv.index("--chunk-vars") + 1]))
wp = (int(nv) + 8191) // 8192 * 128
gb = wp * 8 * int(ns) / 1e9
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("void ", "")
    if n.startswith("k_score_af"):
        avg = float(r["AverageNs"]) / 1e3
        print(f"UTM_AF_TABLES={t} {n[:34]:34s} calls={r['Calls']:>4s} avg_us={avg:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} max_us={float(r['MaxNs'])/1e3:9.2f}  one launch's columns {gb:.3f} GB -> {gb/avg*1e6:7.1f} GB/s at avg = {gb/avg*1e6/8000:.3f} of 8 TB/s")
PY
  find $d -name "*.csv" -size +3M -delete
done
