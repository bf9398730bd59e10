#!/usr/bin/env python3
"""Teacher-forced single-layer fixtures at BASELINE cfg3 shapes, from the REFERENCE's own modules (TEST INFRASTRUCTURE).

Run in the build container only (needs /root/reference, read-only; never on the GPU box):

    python oracle/make_golden_layers.py        ->  tests/golden/layers_cfg3.npz

One layer of each family of the SRGAN training step (P/mode.py:148-179, B=16, LR 24x24, HR / D patch 96):
  g_body   `conv(64, 64, 3, BN=True, act=PReLU)` of a ResBlock (P/ops.py:18-35,41) on 16x64x24x24, train-mode BN
  d_s2     `conv(128, 128, 3, BN=True, act=LeakyReLU, stride=2)` of a discrim_block (P/ops.py:86-92) on 16x128x48x48
  vgg_256  Conv2d(256, 256, 3, padding=1) + ReLU (torchvision's vgg19.features layer, P/vgg19.py:11) on 16x256x24x24
  d_linear Linear(18432, 1024) + LeakyReLU (P/srgan_model.py:53-59) on the flattened 16x512x6x6 map
x and the upstream gradient dy are seeded recipe tensors (NOT produced by neighbouring layers: "teacher forced"), the
weights come from oracle/weights.py-style seeded draws; the stored y / dx / dW / db / dgamma / dbeta / dalpha are what
the reference module computes with torch CPU fp32 autograd, plus the same layer in float64.  Large tensors are stored as
strided samples + fp64 sum / sum of squares / max.  Only data is stored (npz, no pickles, no reference source)."""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/pytorch srgan with lpips,dists/SRGAN_CustomDataset-main"
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

import ops as ref_ops  # noqa: E402  (the reference's P/ops.py, imported unmodified)

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


from oracle.layer_recipe import SPECS, layer_tensors  # noqa: E402


def build(name, dtype):
    spec, t = SPECS[name], layer_tensors(name)
    if spec.get("linear"):
        mod = nn.Sequential(nn.Linear(spec["cin"], spec["cout"]), nn.LeakyReLU(inplace=True))
        conv_m, bn_m, act_m = mod[0], None, mod[1]
    elif name == "vgg_256":
        mod = nn.Sequential(nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True))
        conv_m, bn_m, act_m = mod[0], None, mod[1]
    else:
        act = nn.PReLU() if spec["act"] == "prelu" else nn.LeakyReLU(inplace=True)
        mod = ref_ops.conv(spec["cin"], spec["cout"], spec["k"], BN=True, act=act, stride=spec["stride"])
        conv_m, bn_m, act_m = mod.body[0], mod.body[1], mod.body[2]
    with torch.no_grad():
        conv_m.weight.copy_(torch.from_numpy(t["w"]))
        conv_m.bias.copy_(torch.from_numpy(t["b"]))
        if bn_m is not None:
            bn_m.weight.copy_(torch.from_numpy(t["gamma"]))
            bn_m.bias.copy_(torch.from_numpy(t["beta"]))
    mod = mod.to(dtype).train()
    x = torch.from_numpy(t["x"]).to(dtype).requires_grad_(True)
    xin = x.view(x.size(0), -1) if spec.get("linear") else x
    y = mod(xin)
    y.backward(torch.from_numpy(t["dy"]).to(dtype))
    out = {"y": y.detach(), "dx": x.grad, "dw": conv_m.weight.grad, "db": conv_m.bias.grad}
    if bn_m is not None:
        out.update(dgamma=bn_m.weight.grad, dbeta=bn_m.bias.grad, running_mean=bn_m.running_mean, running_var=bn_m.running_var)
    if spec["act"] == "prelu":
        out["dalpha"] = act_m.weight.grad
    return out


def put(out, key, t):
    t = t.detach().double().reshape(-1)
    n = t.numel()
    out[key + ":absmax"] = float(t.abs().max())
    out[key + ":sum"] = float(t.sum())
    out[key + ":sumsq"] = float((t ** 2).sum())
    if n <= 8192:
        out[key] = t.float().numpy().copy()
    else:
        stride = max(1, n // 4096) | 1
        out[key + ":stride"] = stride
        out[key + ":sample"] = t[::stride].float().numpy().copy()


def main():
    out = {}
    for name in SPECS:
        f32, f64 = build(name, torch.float32), build(name, torch.float64)
        for k, v in f32.items():
            put(out, f"{name}/{k}", v)
            # float64 run of the same module: how far fp32 itself is from the truth (context for the test's tolerance)
            d = (v.double() - f64[k]).abs().max().item()
            out[f"{name}/{k}:f32_vs_f64"] = d
            print(f"{name:9s} {k:12s} shape {tuple(v.shape)!s:22s} max|.| {v.abs().max().item():.3e}  fp32 vs f64 {d:.2e}")
    np.savez_compressed(os.path.join(GOLD, "layers_cfg3.npz"), **out)
    print("wrote", os.path.join(GOLD, "layers_cfg3.npz"), os.path.getsize(os.path.join(GOLD, "layers_cfg3.npz")), "bytes")


if __name__ == "__main__":
    main()
