#!/bin/bash
# The full dense AF pass (k_score_aft against k_score_afq) and the table kernel's SQ counters; run from the repo root through
# gpurun, separately from tools/collect_profiles.sh (one gpurun call holds 20 minutes):  tools/collect_af_first_pass.sh r03
set -o pipefail
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/profiles_$tag
mkdir -p $out
echo "== the full dense AF pass: table lookups (k_score_aft) against the bit walk (k_score_afq), cfg3 and 500M x 2,504; SQ counters of k_score_aft"
(cd $R && { echo "10M x 2,504 (cfg3):"; bash tools/ab_af_first_pass.sh 10000000 2504 | grep "k_score_af[qt]"; echo "500M x 2,504 in 10 chunks of 50M (one launch per chunk):"; bash tools/ab_af_first_pass.sh 500000000 2504 --chunk-vars 50000000 | grep "k_score_af[qt]"; } > $out/${tag}_af_first_pass_tables_vs_bitwalk_raw.txt 2>&1)
(cd $R && python3 tools/pmc_sq.py ${tag}_cfg3_aft k_score_aft -- --af --select 2 --steps 2 --warmup 0 --no-cpu-baseline --no-roofline-pass --no-calibration --no-also --pmc-traffic off > /dev/null 2>&1)
cp $R/profiles/${tag}_cfg3_aft_pmc_sq.json $out/ 2>/dev/null
