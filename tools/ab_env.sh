#!/bin/bash
# Same-box A/B of one environment knob:  tools/ab_env.sh KNOB "v1 v2 ..." [reps] -- <bench.py flags>
# Runs bench.py once per value and repetition, interleaved, and prints it/s, whole-loop fraction and launch time.
R=${GRAFT_REPO_ROOT:-$(pwd)}
knob=$1; vals=$2; reps=${3:-2}; shift 3; [ "$1" = "--" ] && shift
for r in $(seq 1 $reps); do
  for v in $vals; do
    env $knob=$v python3 $R/bench.py "$@" --no-cpu-baseline --no-also --pmc-traffic off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); rf=d['roofline']
print('$knob=$v rep $r: it/s=%.1f ms/step=%.2f whole_loop=%.4f of_stream=%s kernel_frac=%.4f launch_us=%.2f' % (d['value'], d['ms_per_step'], d.get('hbm_frac_whole_loop', 0), d.get('whole_loop_frac_of_stream'), rf['frac'], rf.get('avg_launch_us', 0)))"
  done
done
