#!/bin/bash
# Round profile batch (run through gpurun): kernel-trace stats of the headline bench, two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE) and one SQ pass on a short bench, kernel-trace stats of SRGAN training steps.
# Raw profiler output stays in /tmp on the box; only the CSV summaries come back.
# Usage: bash tools/profile_round.sh r2     -> gpurun_out/<tag>_*
set -u
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
RAW=/tmp/prof_$TAG
mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
keep() {   # keep <raw dir> <name>: copy the csv files of one run back, flattened
    for f in $(find $1 -name '*.csv'); do cp $f $OUT/${TAG}_$2_$(basename $f); done
}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-secondary > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_trace.err || { tail -5 $OUT/${TAG}_trace.err; exit 1; }
keep $RAW/trace trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $RAW/pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/${TAG}_pmc_$c.err || { tail -5 $OUT/${TAG}_pmc_$c.err; exit 1; }
  keep $RAW/pmc_$c pmc_$c
done
timeout -s KILL 240 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $RAW/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/${TAG}_pmc_sq.err || { tail -5 $OUT/${TAG}_pmc_sq.err; exit 1; }
keep $RAW/pmc_sq pmc_sq
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/train -o train -- python3 $R/tools/bench_train_step.py --steps 3 --warmup 2 > $OUT/${TAG}_train_under_rocprof.json 2> $OUT/${TAG}_train_trace.err || { tail -5 $OUT/${TAG}_train_trace.err; exit 1; }
keep $RAW/train train
rm -f $OUT/${TAG}_*_kernel_trace.csv      # per-launch rows: large; the stats files carry the averages
ls -la $OUT | grep ${TAG}_
echo done
