# phase / hazard ablation of the fused ResBlock kernel (timing only: results are wrong with these bits set)
for f in 0 131072 262144 393216 524288 917504 0; do SRMI_DBG=$f timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/a.json && python -c "
import json;d=json.load(open('gpurun_out/a.json'));print($f, d['roofline']['avg_launch_us'])" || exit 1; done
