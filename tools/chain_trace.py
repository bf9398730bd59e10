"""Phase time stamps inside the persistent ResBlock chain (run with SRMI_DBG=16384): per tile, 8 stamps of the middle
block: 0 block start, 1 neighbour flags seen, 2 x tile + weights staged, 3 conv1 done, 4 T written, 5 conv2 done,
6 halo rows stored + flag published, 7 epilogue done.  s_memrealtime ticks (100 MHz)."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
L = importlib.import_module("super-resolution_amd.srmi._lib")
Fn = importlib.import_module("super-resolution_amd.srmi.functional")
from test_chain_gpu import _blocks

lib = L.load()
lib.srmi_chain_enable(1)
arr, bufs, last, tmp, keep = _blocks(16, 96, 96, 16, seed=3)
for _ in range(3):
    L.check(lib.srmi_conv2d_fwd_seq(arr, 32, Fn.stream_ptr()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.check(lib.srmi_conv2d_fwd_seq(arr, 32, Fn.stream_ptr()))
e1.record(); torch.cuda.synchronize()
print("chain of 16 blocks: %.1f us (%.2f us per block)" % (e0.elapsed_time(e1) * 50, e0.elapsed_time(e1) * 50 / 16))
out = (C.c_ulonglong * (256 * 8))()
f = lib.srmi_debug_chain_trace
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]; f.restype = C.c_int
assert f(out, 256 * 8) == 0
t = np.array(out[:], dtype=np.int64).reshape(256, 8)
t0 = t[:, 0].min()
names = ["start", "flags seen", "staged", "conv1 done", "T written", "conv2 done", "flag out", "epilogue done"]
print("stamp (us after the earliest workgroup entered the block): median / min / max over 256 tiles")
for i in range(8):
    d = (t[:, i] - t0) / 100.0
    print("  %-14s %6.2f %6.2f %6.2f" % (names[i], np.median(d), d.min(), d.max()))
d = np.diff(t, axis=1) / 100.0
print("phase durations (median):", " ".join("%s->%s %.2f" % (names[i][:6], names[i + 1][:6], np.median(d[:, i])) for i in range(7)))
