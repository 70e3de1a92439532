#!/bin/bash
# The AF forms of the persistent loop vs one launch per iteration over shapes, on ONE box:
#   tools/ab_interval.sh "1103547x2504 2000000x2504 1103547x640 10000000x313" [f64|f32]
# f64: the interval form (UTM_PERSIST_AF_INTERVAL); f32: the exact fixed-point form (UTM_PERSIST_AF).
dt=${2:-f64}
knob=UTM_PERSIST_AF_INTERVAL; [ "$dt" = f32 ] && knob=UTM_PERSIST_AF
for shape in $1; do
  nv=${shape%x*}; ns=${shape#*x}
  AB_REPS="1" AB_STEPS=2 bash tools/ab_custom.sh "$knob=0 $knob=1" "--n-var $nv --n-samp $ns --af --af-dtype $dt --no-calibration"
done
