#!/bin/bash
# Launch bench.py as N ranks that all use GPU 0 (exchange-overhead probe on a one-GPU box).  usage: N [bench args...]
N=$1; shift
PORT=$((42000 + RANDOM % 2000))
pids=()
for ((r=1; r<N; r++)); do
  RANK=$r WORLD_SIZE=$N LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT python3 bench.py --gpus $N "$@" > /dev/null 2>&1 &
  pids+=($!)
done
RANK=0 WORLD_SIZE=$N LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT python3 bench.py --gpus $N "$@"
for p in "${pids[@]}"; do wait $p; done
