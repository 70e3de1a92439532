#!/bin/bash
# Same-box A/B of library builds:  tools/ab_libs.sh "ab/old.so ab/new.so" [reps] -- <bench.py flags>
# (ab/ is git-ignored scratch; the last library named stays installed as utmos_amd/libutmos_hip.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
libs=$1; reps=${2:-2}; shift 2; [ "$1" = "--" ] && shift
for r in $(seq 1 $reps); do
  for lib in $libs; do
    cp $R/$lib $R/utmos_amd/libutmos_hip.so
    python3 $R/bench.py "$@" --no-cpu-baseline --no-also --pmc-traffic off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); rf=d['roofline'] or {}
print('$lib rep $r: it/s=%.1f ms/step=%.2f whole_loop=%.4f of_stream=%s kernel_frac=%.4f launch_us=%.2f' % (d['value'], d['ms_per_step'], d.get('hbm_frac_whole_loop', 0), d.get('whole_loop_frac_of_stream'), rf.get('frac', 0), rf.get('avg_launch_us', 0)))"
  done
done
