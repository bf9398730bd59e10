#!/bin/bash
# A/B of bench.py under environment settings inside one gpurun call.  Usage: bash tools/ab_bench.sh "VAR=0" "VAR=1"
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
for e in "$@"; do
  env $e timeout -k 10 100 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('env [$e]', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])" || exit 1
done; done
