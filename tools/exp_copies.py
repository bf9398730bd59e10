#!/usr/bin/env python3
"""Which aten ops of the SRGAN iteration launch device copies / adds (torch.profiler, one step, Python stacks)."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60, max_src_column_width=110))
