#!/bin/bash
# NOTE: the ablation build's run-time switches sit inside the persistent kernel's loops and cost it ~25 % (cfg4 8.8 ms
# against 6.9 for the product build): read these numbers against each other only; product-build A/Bs: tools/ab_cinp.sh.
R=${GRAFT_REPO_ROOT:-/root/repo}
export SRMI_LIB=$R/super-resolution_amd/srmi/lib/libsrmi_ablate.so
run() {
  echo "== SRMI_EXP=$1 ($2)"
  SRMI_EXP=$1 timeout -k 10 120 python $R/tools/bench_variants.py --only cfg4 --blocks 2 --layers --steps 5 --warmup 2 2>/dev/null | grep -o 'cin_bf16:[^"]*": {"us": [0-9.]*' | tail -4 || exit 1
}
S=7340032
run $S "skeleton: barriers, LDS reads, epilogue" && run $((S+8388608)) "skeleton without epilogue" && run $((S+16777216)) "skeleton without LDS reads" && run $((S+8388608+16777216)) "barriers only"
