#!/bin/bash
# A/B of bench_train_step.py variants inside one gpurun call.  Usage: bash tools/ab_train.sh "--torch-adam" ""  (each arg = one flag set)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for f in "$@"; do
  timeout -k 10 200 python $R/tools/bench_train_step.py --steps 10 --warmup 3 $f | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('flags [$f]', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'g', d['g_loss'], 'd', d['d_loss'])" || exit 1
done; done
