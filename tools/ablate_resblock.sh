# phase / hazard ablation of the fused ResBlock kernel (timing only: results are wrong with these bits set)
# bits: 1 no staging, 2 no conv1 K loop, 16 no T write-back, 8 no conv2 K loop, 4 no epilogue, 512 no residual,
#       131072 no weight wait, 262144 no hand-off barrier, 524288 no weight stream
for f in 0 1 2 16 8 4 10 26 27 31 5 131072 262144 917504 0; do SRMI_DBG=$f timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/a.json && python -c "
import json;d=json.load(open('gpurun_out/a.json'));print($f, d['roofline']['avg_launch_us'], d['ms_per_step'])" || exit 1; done
