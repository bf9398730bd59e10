#!/bin/bash
# NOTE: the ablation build's run-time switches sit inside the persistent kernel's loops and cost it ~25 % (cfg4 8.8 ms
# against 6.9 for the product build): read these numbers against each other only; product-build A/Bs: tools/ab_cinp.sh.
# Ablations of the persistent K-chunked conv (ablation build, SRMI_EXP bits 20-22; SRMI_DBG bit 17 = launch-per-tile kernel)
# on the four layers of an RRDB dense block at cfg4's size.  Through gpurun.
R=${GRAFT_REPO_ROOT:-/root/repo}
export SRMI_LIB=$R/super-resolution_amd/srmi/lib/libsrmi_ablate.so
run() {
  echo "== SRMI_EXP=$1 SRMI_DBG=$2 ($3)"
  SRMI_EXP=$1 SRMI_DBG=$2 timeout -k 10 120 python $R/tools/bench_variants.py --only cfg4 --blocks 2 --layers --steps 5 --warmup 2 2>/dev/null | grep -o 'cin_bf16:[^"]*": {"us": [0-9.]*' | tail -4 || exit 1
}
run 0 0 "persistent, as built" && run 0 131072 "launch-per-tile kernel" && run 1048576 0 "no MFMAs" && run 2097152 0 "no tile loads" && run 4194304 0 "no weight loads" && run 7340032 0 "barriers and LDS reads only"
