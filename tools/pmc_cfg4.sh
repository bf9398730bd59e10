#!/bin/bash
# cfg4 (RRDB-23, two blocks are enough: the layers repeat) under rocprofv3: kernel trace, then HBM traffic of the
# K-chunked bf16 conv layers in two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) and one SQ pass.
# Usage (through gpurun): bash tools/pmc_cfg4.sh <tag>   -> gpurun_out/<tag>_cfg4_*.csv
set -u
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
RAW=/tmp/pmc4_$TAG
mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/bench_variants.py --only cfg4 --blocks 2 --steps 2 --warmup 1"
keep() { for f in $(find $1 -name '*.csv'); do cp $f $OUT/${TAG}_cfg4_$2_$(basename $f); done; }
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $RAW/trace -o t -- $CMD > $OUT/${TAG}_cfg4_trace.out 2> $OUT/${TAG}_cfg4_trace.err || { tail -5 $OUT/${TAG}_cfg4_trace.err; exit 1; }
keep $RAW/trace trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $RAW/pmc_$c -- $CMD > /dev/null 2> $OUT/${TAG}_cfg4_pmc_$c.err || { tail -5 $OUT/${TAG}_cfg4_pmc_$c.err; exit 1; }
  keep $RAW/pmc_$c pmc_$c
done
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $RAW/pmc_sq -- $CMD > /dev/null 2> $OUT/${TAG}_cfg4_pmc_sq.err || { tail -5 $OUT/${TAG}_cfg4_pmc_sq.err; exit 1; }
keep $RAW/pmc_sq pmc_sq
ls -la $OUT | grep ${TAG}_cfg4
echo done
