#!/bin/bash
# SQ counters of the f32 tiled conv on one VGG-size layer (run through gpurun): where the non-MFMA time goes.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
RAW=/tmp/pmc_f32t
mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
F=${1:-VGG 64->64 @96}
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $RAW/a -- python3 $R/tools/bench_conv_f32.py --only fwd --reps 3 --filter "$F" > $OUT/f32t_pmc_a.log 2>&1 || { tail -5 $OUT/f32t_pmc_a.log; exit 1; }
for f in $(find $RAW/a -name '*counter_collection.csv'); do cp $f $OUT/f32t_pmc_a.csv; done
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $RAW/b -- python3 $R/tools/bench_conv_f32.py --only fwd --reps 3 --filter "$F" > $OUT/f32t_pmc_b.log 2>&1 || { tail -5 $OUT/f32t_pmc_b.log; exit 1; }
for f in $(find $RAW/b -name '*counter_collection.csv'); do cp $f $OUT/f32t_pmc_b.csv; done
echo done
