#!/usr/bin/env python3
"""torch.profiler kernel table of the tiled 960x540 -> 4K path (BASELINE config 5): three frames through srmi.tiling.sr_tiled.
Round 3: 5 chained launches (67 %) + 5 fused tails (30 %) + 3 % crops / pastes per frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd"), os.path.join(ROOT, "tools")]
import torch
from torch.profiler import profile, ProfilerActivity
import bench_variants as V
from srmi.srgan_model import Generator
from srmi.tiling import sr_tiled
from bench import synthetic_state
gen = Generator(); gen.load_state_dict(synthetic_state(gen, 0)); gen = gen.to("cuda:0").eval().set_compute("bf16")
frame = (torch.rand(1, 3, 540, 960) * 2 - 1).to("cuda:0")
kw, how = V.cfg5_tiling()
with torch.no_grad():
    for _ in range(5): sr_tiled(gen, frame, halo=8, **kw)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3): sr_tiled(gen, frame, halo=8, **kw)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
