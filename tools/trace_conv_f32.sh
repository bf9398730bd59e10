#!/bin/bash
# Kernel-trace stats of tools/bench_conv_f32.py on one layer (run through gpurun): bash tools/trace_conv_f32.sh <tag> <filter> <only>
set -u
TAG=$1; F=$2; ONLY=$3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
RAW=/tmp/trace_$TAG
mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o t -- python3 $R/tools/bench_conv_f32.py --only $ONLY --filter "$F" > $OUT/${TAG}.log 2>&1 || { tail -5 $OUT/${TAG}.log; exit 1; }
for f in $(find $RAW -name '*kernel_stats.csv'); do cp $f $OUT/${TAG}_kernel_stats.csv; done
