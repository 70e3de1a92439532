#!/bin/bash
# Persistent loop vs one launch per iteration over matrix heights, on ONE box: tools/ab_sizes.sh "200000 600000 ..." [n_samp]
for nv in $1; do
  AB_REPS="1" AB_STEPS=3 bash tools/ab_custom.sh "UTM_PERSISTENT=0 UTM_PERSISTENT=1" "--n-var $nv --n-samp ${2:-2504}"
done
