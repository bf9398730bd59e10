#!/usr/bin/env python3
"""K-chunked bf16 conv (persistent form) on 8 x 256 x 256: channel slices of a 160-channel buffer (the RRDB dense block's
layout) against dense tensors - what the 320-byte pixel pitch costs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd")]
import torch
from srmi import _lib as L, functional as Fn
dev = "cuda:0"
N, H, W = 8, 256, 256
def run(Cin, Cout, in_cs, out_cs, out_co, res=False):
    d = Fn.make_desc(N, H, W, Cin, Cout, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16,
                     act=L.ACT_NONE if res else L.ACT_LRELU, act_param=0.2, in_cstride=in_cs, out_cstride=out_cs, out_coffset=out_co,
                     res_cstride=in_cs if res else 0)
    x = torch.randn(N, H, W, in_cs or Cin, device=dev).bfloat16()
    y = torch.zeros(N, H, W, out_cs or Cout, device=dev, dtype=torch.bfloat16)
    w = Fn.pack_weights(d, torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05)
    sc, sh, al = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev), torch.zeros(1, device=dev)
    r = x if res else None
    for _ in range(3): Fn.conv2d_fused(d, x, w, sc, sh, al, r, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): Fn.conv2d_fused(d, x, w, sc, sh, al, r, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    fl = 2.0 * N * H * W * Cin * Cout * 9
    print(f"{Cin:4d}->{Cout:3d} in_cs {in_cs or Cin:3d} out_cs {out_cs or Cout:3d} res {int(res)}: {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s", flush=True)
for Cin, Cout in [(64, 32), (96, 32), (128, 32)]:
    run(Cin, Cout, 160, 160, Cin)
    run(Cin, Cout, 0, 0, 0)
    run(Cin, Cout, 0, 160, Cin)
    run(Cin, Cout, 160, 0, 0)
run(160, 64, 160, 0, 0, res=True)
run(160, 64, 160, 0, 0, res=False)
run(128, 64, 0, 0, 0)
run(64, 64, 0, 0, 0)
