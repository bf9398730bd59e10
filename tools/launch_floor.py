"""How long does an (almost) empty launch take when timed the way bench.py times kernels?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "super-resolution_amd"))
import torch
from srmi import _lib as L, functional as Fn
dev = "cuda:0"
t = torch.zeros(64, device=dev)
src = torch.zeros((1, 8, 8, 8), device=dev)
dst = torch.zeros((1, 8, 8, 8), device=dev)
def timed(fn, n=200):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    d = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
    return d[n // 2]
def untimed(fn, n=2000):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("torch add_  : event-median %.2f us, back-to-back %.2f us" % (timed(lambda: t.add_(1)), untimed(lambda: t.add_(1))))
f = lambda: Fn.layout_convert(src, 1, 8, 8, 8, L.NHWC, L.NCHW, torch.float32, out=dst)
print("srmi layout : event-median %.2f us, back-to-back %.2f us" % (timed(f), untimed(f)))
