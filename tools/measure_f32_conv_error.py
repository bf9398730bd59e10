#!/usr/bin/env python3
"""Error of the f32 conv kernels against a float64 CPU convolution on one 3x3 64->64 layer (N=2, 48x48, inputs U(-1,1),
weights N(0, 1/K)): max and rms error relative to the output's rms.  The kernel is chosen by the environment:
default = halo-tile split-bf16, SRMI_DBG=512 = general split-bf16, SRMI_F32_EXACT=1 = exact-f32 MFMA.
    python tools/measure_f32_conv_error.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd")]
import numpy as np
import torch

from srmi import _lib as L
from srmi import functional as Fn

g = np.random.default_rng(0)
N, H, W, Cin, Cout, k = 2, 48, 48, 64, 64, 3
x = torch.from_numpy(g.uniform(-1, 1, (N, Cin, H, W)).astype(np.float32))
w = torch.from_numpy((g.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
ref32 = torch.nn.functional.conv2d(x, w, padding=1).double()
dev = torch.device("cuda:0")
desc = Fn.make_desc(N, H, W, Cin, Cout, k)
xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
wp = Fn.pack_weights(desc, w.to(dev))
yd = torch.empty(N, H, W, Cout, device=dev)
Fn.conv2d_fused(desc, xd, wp, None, None, None, None, out=yd)
torch.cuda.synchronize()
got = yd.cpu().permute(0, 3, 1, 2).double()
rms = float(ref.pow(2).mean().sqrt())
for name, t in (("kernel", got), ("torch CPU f32", ref32)):
    e = t - ref
    print(f"{name:14s}: max |err| / rms(out) = {float(e.abs().max()) / rms:.3e}   rms err / rms(out) = {float(e.pow(2).mean().sqrt()) / rms:.3e}")
