#!/bin/bash
# Ablations of the fused tail's LDS adds (ablation build; results wrong with these bits).  Usage: bash tools/exp_upfuse.sh "0 256 512 1024"
R=${GRAFT_REPO_ROOT:-/root/repo}
export SRMI_LIB=$R/super-resolution_amd/srmi/lib/libsrmi_ablate.so
for f in ${1:-0}; do
  SRMI_EXP=$f timeout -k 10 120 python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $R/gpurun_out/exp.json && python -c "
import json;d=json.load(open('$R/gpurun_out/exp.json'));print('exp', $f, 'upfuse us', [v['ms']*1e3 for k,v in d['roofline']['layers'].items() if 'upfuse' in k][0], 'ms/step', d['ms_per_step'])" || exit 1
done
