#!/bin/bash
# Per-iteration cost of every exchange form, from ONE GPU: the same select-all run with no exchange,
# through the device mailboxes (the shard posts to itself), through RCCL with a 1-rank communicator in both column forms.
#   tools/exchange_table.sh "--n-var 10000000 --n-samp 313" [more bench flags]
flags="$1"
run() {  # label, env, extra flags
  printf "%-34s " "$1"
  env $2 timeout -k 10 400 python bench.py $flags $3 --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline --no-also --no-sharded-check --pmc-traffic off 2>/dev/null | \
    python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); n=j['config']['iterations_per_step']; print('exchange=%-15s it/s=%9.1f ms/step=%8.2f us/iteration=%7.2f whole-loop frac=%.4f' % (j['exchange'], j['value'], j['ms_per_step'], j['ms_per_step']*1e3/n, j['hbm_frac_whole_loop']))"
}
for rep in ${AB_REPS:-1 2}; do
run "none (persistent loop if it applies)" UTM_NOP=1 ""
run "none (one launch per iteration)" UTM_PERSISTENT=0 ""
run "mailboxes, one rank" UTM_NOP=1 "--force-mailboxes"
run "rccl all-reduce column, one rank" UTM_NOP=1 "--force-comm --exchange rccl-allreduce"
run "rccl broadcast column, one rank" UTM_NOP=1 "--force-comm --exchange rccl"
done
