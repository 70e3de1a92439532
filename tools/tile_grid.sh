#!/bin/bash
# Which tile (8 / 16 / 32 KiB) the launch-per-iteration integer kernel wants, by column height and selectable samples:
# the first 40 iterations of a run (selectable samples ~ n_samp) per (n_var, n_samp, tile), one launch per iteration.
#   tools/tile_grid.sh "2000000 5000000 10000000" "2504 1250 600 300"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for nv in $1; do
  for ns in $2; do
    line="n_var=$nv n_samp=$ns:"
    for t in 8 16 32; do
      v=$(UTM_PERSISTENT=0 UTM_TILE_STEPS=$t python3 $R/bench.py --n-var $nv --n-samp $ns --select 40 --steps 4 --warmup 1 --no-calibration --no-cpu-baseline --no-also --no-roofline-pass --pmc-traffic off 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.4f' % d.get('hbm_frac_whole_loop', 0))")
      line="$line  tile$t=$v"
    done
    echo "$line"
  done
done
