#!/bin/bash
# Collect the round's evidence on a GPU box (run from the repo root through gpurun); writes gpurun_out/prof_<tag>/ and
# condensed files under gpurun_out/profiles_r02/ (copy those into profiles/).
#   tools/collect_profiles.sh r03
set -o pipefail
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
echo "== headline (unprofiled), default invocation"
python3 $R/bench.py --steps 5 --warmup 2 > $out/${tag}_cfg2_bench.json 2> $out/${tag}_cfg2_bench.err || exit 1
for cfg in cfg2 cfg3 cfg1; do
  extra="--workload $cfg"; [ $cfg = cfg2 ] && extra="--no-also"
  prefix="void k_score_int"; [ $cfg = cfg3 ] && prefix="void k_score_afs,void k_score_afq"; [ $cfg = cfg1 ] && prefix="void k_loop_int"
  echo "== $cfg kernel trace + stats"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_${cfg}_kt -- python3 $R/bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off > $out/${tag}_${cfg}_bench_under_rocprofv3.json 2> /dev/null || exit 1
  for counter in FETCH_SIZE WRITE_SIZE; do
    echo "== $cfg pmc $counter"
    rocprofv3 --pmc $counter --output-format csv -d $R/gpurun_out/prof_${tag}_${cfg}_$counter -- python3 $R/bench.py $extra --steps 1 --warmup 0 --no-cpu-baseline --no-roofline-pass --no-also --pmc-traffic off > /dev/null 2>&1 || exit 1
  done
  (cd $R && mkdir -p profiles_tmp && python3 tools/summarize_profile.py ${tag}_${cfg} gpurun_out/prof_${tag}_${cfg}_kt gpurun_out/prof_${tag}_${cfg}_FETCH_SIZE gpurun_out/prof_${tag}_${cfg}_WRITE_SIZE "bench.py $extra, one step" "$prefix" > /dev/null) || exit 1
  python3 $R/tools/trace_gaps.py $R/gpurun_out/prof_${tag}_${cfg}_kt > $out/${tag}_${cfg}_stream_time.txt 2>&1
  find $R/gpurun_out/prof_${tag}_${cfg}_kt $R/gpurun_out/prof_${tag}_${cfg}_FETCH_SIZE $R/gpurun_out/prof_${tag}_${cfg}_WRITE_SIZE -name "*.csv" -size +3M -delete
done
echo "== SQ / TCC counters of the integer scoring kernel (first 120 iterations of cfg2) and of the persistent loop (cfg1)"
(cd $R && python3 tools/pmc_sq.py ${tag}_cfg2 k_score_int -- --steps 1 --warmup 0 --select 120 --no-cpu-baseline --no-roofline-pass --no-also --no-calibration --pmc-traffic off > /dev/null 2>&1)
(cd $R && python3 tools/pmc_sq.py ${tag}_cfg1 k_loop_int -- --workload cfg1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline-pass --no-calibration --pmc-traffic off > /dev/null 2>&1)
cp $R/profiles/${tag}_cfg2_pmc_sq.json $R/profiles/${tag}_cfg1_pmc_sq.json $out/ 2>/dev/null
echo "== per-iteration cost of every exchange form, from one GPU (10M x 313 = one rank's share of cfg2 at 8 GPUs; cfg2)"
(cd $R && AB_REPS="1 2" bash tools/exchange_table.sh "--n-var 10000000 --n-samp 313" > $out/${tag}_exchange_cost_one_gpu_shard_shape_10Mx313.txt 2>&1)
(cd $R && AB_REPS=1 AB_STEPS=2 bash tools/exchange_table.sh "--n-var 10000000 --n-samp 2504" > $out/${tag}_exchange_cost_one_gpu_cfg2.txt 2>&1)
echo "== persistent loop vs one launch per iteration over matrix heights (same box)"
(cd $R && bash tools/ab_sizes.sh "100000 300000 600000 1103547 1500000 2000000 3000000 5000000" > $out/${tag}_persistent_vs_launches_by_height.txt 2>&1)
echo "== the AF forms of the persistent loop vs one launch per iteration over shapes (same box)"
(cd $R && bash tools/ab_interval.sh "300000x2504 1103547x2504 1500000x2504 2000000x2504" f32 > $out/${tag}_af_forms_vs_launches.txt 2>/dev/null)
(cd $R && bash tools/ab_interval.sh "300000x2504 1103547x2504 1500000x2504 2000000x2504" f64 >> $out/${tag}_af_forms_vs_launches.txt 2>/dev/null)
echo "== float64 AF (the reference's in-memory --af values)"
python3 $R/bench.py --af --af-dtype f64 --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off > $out/${tag}_af64_bench.json 2>/dev/null
python3 $R/bench.py --workload af64 --af-estimate-scores --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off > $out/${tag}_af64_cli_mode_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_af64_kt -- python3 $R/bench.py --workload af64 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline-pass --pmc-traffic off > /dev/null 2>&1 || exit 1
cut -c1-220 $(find $R/gpurun_out/prof_${tag}_af64_kt -name "*kernel_stats.csv") > $out/${tag}_af64_kernel_stats.csv
python3 $R/tools/kernel_dist.py $R/gpurun_out/prof_${tag}_af64_kt k_verify > $out/${tag}_af64_k_verify_durations.txt 2>&1
find $R/gpurun_out/prof_${tag}_af64_kt -name "*.csv" -size +3M -delete
echo "== float64 / float32 AF at chr22 size (1,103,547 x 2,504)"
python3 $R/bench.py --n-var 1103547 --af --af-dtype f64 --steps 3 --warmup 1 --no-cpu-baseline --no-also --pmc-traffic off > $out/${tag}_chr22size_af64_bench.json 2>/dev/null
python3 $R/bench.py --n-var 1103547 --af --af-dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-also --pmc-traffic off > $out/${tag}_chr22size_af32_bench.json 2>/dev/null
echo "== RCCL protocol overhead with a 1-rank communicator (collectives degenerate; launch and sync costs real)"
for x in rccl rccl-allreduce; do
  python3 $R/bench.py --force-comm --exchange $x --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off --no-sharded-check > $out/${tag}_cfg2_one_rank_${x}_bench.json 2>/dev/null
done
echo "== decremental (optional mode, reported separately)"
python3 $R/bench.py --decremental --steps 3 --warmup 1 --no-cpu-baseline --pmc-traffic off > $out/${tag}_cfg2_decremental_bench.json 2>/dev/null
cp $R/profiles/${tag}_*_kernel_stats.csv $R/profiles/${tag}_*_pmc_hbm.json $out/ 2>/dev/null   # (written there by summarize_profile.py just now)
ls -la $out
