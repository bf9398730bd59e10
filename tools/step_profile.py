#!/usr/bin/env python3
"""Kernel table of ONE SRGAN iteration (tools/bench_train_step.py's step) through torch.profiler: launches, total and
average device time per kernel, and the aten ops that launch torch's own kernels (gradient accumulation, cat)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd"), os.path.join(ROOT, "tools")]
import torch
from torch.profiler import profile, ProfilerActivity
from bench_train_step import build_step

dev = torch.device("cuda:0")
step = build_step(dev)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=70))
