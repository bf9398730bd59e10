#!/usr/bin/env python3
"""Per-layer throughput of the exact-f32 training kernels (LDS-tiled forward / dgrad / wgrad) on the conv shapes of the
SRGAN step (BASELINE config 3: G at LR 24x24, D at 96x96, VGG19 at 96x96, batch 16), against the 157 TFLOP/s f32-MFMA peak.

    python tools/bench_conv_f32.py [--reps 20] [--only fwd|dgrad|wgrad]

Environment knobs of the library for experiments: SRMI_DBG bit 24 = old no-LDS forward/dgrad kernel, bit 25 = old wgrad.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd")]

import torch

from srmi import _lib as L
from srmi import functional as Fn

# (label, N, H, W, Cin, Cout, k, stride)
SHAPES = [
    ("G body 64->64 @24", 16, 24, 24, 64, 64, 3, 1),
    ("G up1 64->256 @24", 16, 24, 24, 64, 256, 3, 1),
    ("G up2 64->256 @48", 16, 48, 48, 64, 256, 3, 1),
    ("G last 64->3 @96", 16, 96, 96, 64, 3, 3, 1),
    ("G conv01 3->64 k9 @24", 16, 24, 24, 3, 64, 9, 1),
    ("D/VGG 3->64 @96", 16, 96, 96, 3, 64, 3, 1),
    ("VGG 64->64 @96", 16, 96, 96, 64, 64, 3, 1),
    ("D 64->64 s2 @96", 16, 96, 96, 64, 64, 3, 2),
    ("VGG 64->128 @48", 16, 48, 48, 64, 128, 3, 1),
    ("VGG 128->128 @48", 16, 48, 48, 128, 128, 3, 1),
    ("D 128->128 s2 @48", 16, 48, 48, 128, 128, 3, 2),
    ("VGG 128->256 @24", 16, 24, 24, 128, 256, 3, 1),
    ("VGG 256->256 @24", 16, 24, 24, 256, 256, 3, 1),
    ("D 256->256 s2 @24", 16, 24, 24, 256, 256, 3, 2),
    ("VGG 256->512 @12", 16, 12, 12, 256, 512, 3, 1),
    ("VGG 512->512 @12", 16, 12, 12, 512, 512, 3, 1),
    ("D 512->512 s2 @12", 16, 12, 12, 512, 512, 3, 2),
    ("VGG 512->512 @6", 16, 6, 6, 512, 512, 3, 1),
    ("D linear 18432->1024", 16, 6, 6, 512, 1024, 6, 1, 0),      # nn.Linear on the flattened 512x6x6 map: a 6x6 conv, pad 0
    ("D linear 1024->1", 16, 1, 1, 1024, 1, 1, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--filter", default="")
    ap.add_argument("--batch", type=int, default=0, help="override the batch size of every shape (default: 16, config 3)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    st = Fn.stream_ptr()
    ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps      # us

    print(f"{'layer':26s} {'GFLOP':>7s} | {'fwd us':>8s} {'TF/s':>6s} | {'dgrad us':>8s} {'TF/s':>6s} | {'wgrad us':>8s} {'TF/s':>6s}")
    tot = [0.0, 0.0, 0.0, 0.0]
    for label, N, H, W, Cin, Cout, k, s, *rest in SHAPES:
        N = args.batch or N
        if args.filter and args.filter not in label:
            continue
        pad = rest[0] if rest else k // 2
        desc = Fn.make_desc(N, H, W, Cin, Cout, k, stride=s, pad=pad, compute=L.COMPUTE_F32)
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
        bias = torch.zeros(Cout, device=dev)
        z = torch.empty(N, Ho, Wo, Cout, device=dev)
        dz = torch.randn(N, Ho, Wo, Cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        wp = Fn.pack_weights(desc, w)
        nb = C.c_size_t(0)
        L.check(lib.srmi_conv2d_dgrad_packed_weight_bytes(C.byref(desc), C.byref(nb)))
        wd = torch.empty(nb.value, dtype=torch.uint8, device=dev)
        L.check(lib.srmi_conv2d_dgrad_pack_weights(C.byref(desc), w.data_ptr(), wd.data_ptr(), st))
        a = L.ConvArgs()
        Fn.fill_args(a, desc, x, wp, None, bias, None, None, z)
        n1, n2, n3 = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        L.check(lib.srmi_conv2d_workspace_bytes(C.byref(desc), C.byref(n1)))
        L.check(lib.srmi_conv2d_dgrad_workspace_bytes(C.byref(desc), C.byref(n2)))
        L.check(lib.srmi_conv2d_wgrad_workspace_bytes(C.byref(desc), C.byref(n3)))
        assert max(n1.value, n2.value, n3.value) <= ws.numel()
        gf = 2.0 * N * Ho * Wo * Cout * k * k * Cin / 1e9
        t = [float("nan")] * 3
        if args.only in ("", "fwd"):
            t[0] = timeit(lambda: L.check(lib.srmi_conv2d_fwd_ws(C.byref(a), ws.data_ptr(), n1.value, st)))
        if args.only in ("", "dgrad"):
            t[1] = timeit(lambda: L.check(lib.srmi_conv2d_dgrad(C.byref(desc), dz.data_ptr(), wd.data_ptr(), dx.data_ptr(),
                                                                ws.data_ptr(), n2.value, st)))
        if args.only in ("", "wgrad"):
            t[2] = timeit(lambda: L.check(lib.srmi_conv2d_wgrad(C.byref(desc), x.data_ptr(), dz.data_ptr(), dw.data_ptr(),
                                                                ws.data_ptr(), n3.value, st)))
        if os.environ.get("SRMI_B3_STAMPS"):      # library built with -DSRMI_B3_STAMPS: phase stamps of the middle workgroup
            torch.cuda.synchronize()
            for bm in (64, 128):
                m0 = ((N * Ho * Wo + bm - 1) // bm // 2) * bm
                st_ = z.flatten()[m0 * Cout:m0 * Cout + 10].view(torch.int64).tolist()
                print(f"  stamps (tile {bm}), us since kernel entry (prologue, first chunk, K loop, epilogue):", [(v - st_[0]) / 100 for v in st_[1:]])
        print(f"{label:26s} {gf:7.2f} | " + " | ".join(f"{v:8.1f} {gf / v * 1e3:6.1f}" for v in t), flush=True)
        tot[0] += gf
        for i in range(3):
            tot[i + 1] += t[i]
    print(f"{'sum':26s} {tot[0]:7.2f} | " + " | ".join(f"{v:8.1f} {tot[0] / v * 1e3:6.1f}" for v in tot[1:]))


if __name__ == "__main__":
    main()
