#!/usr/bin/env python3
"""Phase time stamps of the fused ResBlock kernel (ablation build, SRMI_STAMPS): s_memrealtime (100 MHz) at kernel
entry / loads issued / own loads landed / tile barrier / conv1 done / T barrier / conv2 done / stores issued, waves 0 and 7
of every workgroup, over consecutive launches of one forward.  Prints per-phase means and the launch-to-launch gaps."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd")]
os.environ.setdefault("SRMI_LIB", os.path.join(ROOT, "super-resolution_amd/srmi/lib/libsrmi_ablate.so"))
os.environ.setdefault("SRMI_STAMPS", "64")
import torch
from oracle import srgan_oracle as O, weights as W
from srmi import _lib
from srmi.srgan_model import Generator

gen = Generator(num_block=16)
gen.load_state_dict(O.to_torch(W.generator_weights(seed=0, num_block=16)))
gen = gen.cuda().eval().set_compute("bf16")
x = torch.from_numpy(W.lr_input((16, 3, 96, 96), seed=1)).cuda()
lib = _lib.load()
fn = lib.srmi_dbg_resblock_stamps
fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_int]
with torch.no_grad():
    for _ in range(3):
        gen(x)
    torch.cuda.synchronize()
    n = C.c_int(0)
    fn(None, 0, C.byref(n), 1)
    gen(x)
    torch.cuda.synchronize()
buf = np.zeros((64, 1024, 2, 8), dtype=np.uint64)
fn(buf.ctypes.data, buf.nbytes, C.byref(n), 0)
L = n.value
st = buf[:L, :256].astype(np.int64)          # [launch][wg][wave][stamp]
t0 = st[0, :, 0, 0].min()
st = (st - t0) * 0.01                        # us
names = ["entry", "loads issued", "own loads landed", "tile barrier", "conv1 done", "T barrier", "conv2 done", "stores issued"]
print(f"{L} launches; per-phase durations (us), mean over workgroups of wave 0 [min..max], launches 2..{L-1}")
for i in range(1, 8):
    d = st[2:, :, 0, i] - st[2:, :, 0, i - 1]
    print(f"  {names[i-1]:>18s} -> {names[i]:<18s} {d.mean():6.2f} [{d.min():6.2f} .. {d.max():6.2f}]")
ck = buf[:L, :256, 1].astype(np.int64)
rt = buf[:L, :256, 0].astype(np.int64)
for a, b, nm in ((3, 4, "conv1 K loop"), (5, 6, "conv2 K loop"), (0, 7, "whole kernel")):
    f = (ck[2:, :, b] - ck[2:, :, a]) / ((rt[2:, :, b] - rt[2:, :, a]) * 10.0)       # cycles per ns = GHz
    print(f"  shader clock during {nm}: {f.mean():.3f} GHz [{f.min():.3f} .. {f.max():.3f}]")
tot = st[2:, :, 0, 7] - st[2:, :, 0, 0]
print(f"  entry -> stores issued {tot.mean():6.2f}")
for l in range(1, min(L, 8)):
    prev_end = st[l - 1, :, 0, 7].max(); first = st[l, :, 0, 0].min(); last_start = st[l, :, 0, 0].max()
    print(f"launch {l}: first entry {first:7.2f}  (prev last stores-issued {prev_end:7.2f}: gap {first-prev_end:5.2f})  entry spread {last_start-first:5.2f}  "
          f"last stores issued {st[l,:,0,7].max():7.2f}  period {st[l,:,0,0].min()-st[l-1,:,0,0].min():6.2f}")
