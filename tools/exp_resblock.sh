#!/bin/bash
# Scheduling experiments on the fused ResBlock kernel (ablation build: `make -C super-resolution_amd/csrc ablate`),
# A/B inside ONE gpurun call (boxes differ by +-4 %).  Usage: bash tools/exp_resblock.sh "0 1 3 11" [DBG]
R=${GRAFT_REPO_ROOT:-/root/repo}
export SRMI_LIB=$R/super-resolution_amd/srmi/lib/libsrmi_ablate.so
for f in ${1:-0}; do
  SRMI_EXP=$f SRMI_DBG=${2:-0} timeout -k 10 120 python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $R/gpurun_out/exp.json && python -c "
import json;d=json.load(open('$R/gpurun_out/exp.json'));print('exp', $f, 'resblock us', d['roofline']['avg_launch_us'], 'ms/step', d['ms_per_step'])" || exit 1
done
