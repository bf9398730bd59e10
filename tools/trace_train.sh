#!/bin/bash
# Kernel-trace stats of the SRGAN training step (run through gpurun).  Usage: bash tools/trace_train.sh <tag>
set -u
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
RAW=/tmp/trace_$TAG
mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/train -o train -- python3 $R/tools/bench_train_step.py --steps 5 --warmup 2 > $OUT/${TAG}_train_under_rocprof.json 2> $OUT/${TAG}_train_trace.err || { tail -5 $OUT/${TAG}_train_trace.err; exit 1; }
for f in $(find $RAW/train -name '*kernel_stats.csv'); do cp $f $OUT/${TAG}_train_kernel_stats.csv; done
echo done
