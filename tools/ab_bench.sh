#!/bin/bash
# A/B on ONE box (boxes differ by a few percent): tools/ab_bench.sh "ENV=a ENV=b ..." workload [workload ...]
# Each word of the first argument is one environment setting to try (use _ for "nothing").
envs="$1"; shift
for w in "$@"; do
  for rep in ${AB_REPS:-1 2}; do
  for e in $envs; do
    [ "$e" = "_" ] && e="UTM_NOP=1"
    case "$e" in LIB=*) cp "${e#LIB=}" utmos_amd/libutmos_hip.so;; esac   # a library variant built beforehand (ab/*.so)
    printf "%s %s: " "$w" "$e"
    env $e timeout -k 10 300 python bench.py --workload $w --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline --pmc-traffic off | \
      python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline'] or {}; print('it/s=%.1f ms/step=%.2f loop_frac=%.4f kernel_frac=%.4f launch_us=%.2f' % (j['value'], j['ms_per_step'], j['hbm_frac_whole_loop'], r.get('frac',0), r.get('avg_launch_us',0)))"
  done
  done
done
