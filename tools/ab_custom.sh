#!/bin/bash
# A/B on ONE box with free bench flags: tools/ab_custom.sh "LIB=ab/a.so LIB=ab/b.so" "--n-var 1103547 --af --af-dtype f64"
envs="$1"; flags="$2"
for rep in ${AB_REPS:-1 2}; do
for e in $envs; do
  [ "$e" = "_" ] && e="UTM_NOP=1"
  case "$e" in LIB=*) cp "${e#LIB=}" utmos_amd/libutmos_hip.so;; esac
  printf "%s | %s: " "$flags" "$e"
  env $e timeout -k 10 300 python bench.py $flags --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline --no-also --pmc-traffic off | \
    python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline'] or {}; print('it/s=%.1f ms/step=%.2f loop_frac=%.4f kernel_frac=%.4f launch_us=%.2f chained=%s' % (j['value'], j['ms_per_step'], j['hbm_frac_whole_loop'], r.get('frac',0), r.get('avg_launch_us',0), j['config'].get('af_chained_iterations')))"
done
done
