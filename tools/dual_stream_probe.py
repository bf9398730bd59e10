"""Probe: 16 fused ResBlocks over a batch of 16 (96x96) as ONE chain vs TWO half-batch chains on two streams.
Run with SRMI_DBG=4096 for the 6x48 / two-workgroups-per-CU kernel variant."""
import ctypes as C, importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("super-resolution_amd.srmi")
L = importlib.import_module("super-resolution_amd.srmi._lib")
Fn = importlib.import_module("super-resolution_amd.srmi.functional")
DEV = "cuda:0"
H = W = 96
NB = 16


def chain(N):
    g = np.random.default_rng(1)
    d1 = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_PRELU)
    d2 = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_NONE)
    w = torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)).to(DEV)
    wp1, wp2 = Fn.pack_weights(d1, w), Fn.pack_weights(d2, w)
    sc = torch.ones(64, device=DEV); sh = torch.zeros(64, device=DEV); al = torch.tensor([0.25], device=DEV)
    bufs = [torch.rand(N, H, W, 64, device=DEV).bfloat16() for _ in range(2)]
    tmp = torch.empty(N, H, W, 64, dtype=torch.bfloat16, device=DEV)
    pairs = []
    for i in range(2):
        a1, a2 = L.ConvArgs(), L.ConvArgs()
        Fn.fill_args(a1, d1, bufs[i], wp1, sc, sh, al, None, tmp)
        Fn.fill_args(a2, d2, tmp, wp2, sc, sh, None, bufs[i], bufs[1 - i])
        a1.flags = L.CONV_FUSE_NEXT
        pairs.append((a1, a2))
    arr = (L.ConvArgs * (2 * NB))()
    for b in range(NB):
        C.memmove(C.byref(arr[2 * b]), C.byref(pairs[b & 1][0]), C.sizeof(L.ConvArgs))
        C.memmove(C.byref(arr[2 * b + 1]), C.byref(pairs[b & 1][1]), C.sizeof(L.ConvArgs))
    keep = (wp1, wp2, sc, sh, al, bufs, tmp, pairs)
    return arr, keep


def run(arr, stream):
    L.check(L.load().srmi_conv2d_fwd_seq(arr, 2 * NB, stream.cuda_stream))


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


full, k0 = chain(16)
ha, k1 = chain(8)
hb, k2 = chain(8)
s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print("one chain, batch 16      : %.1f us per 16 blocks" % timeit(lambda: run(full, s0)))
print("one chain, batch 8       : %.1f us per 16 blocks" % timeit(lambda: run(ha, s0)))
for nb_ in ():
    c_, k_ = chain(nb_)
    print("one chain, batch %-2d      : %.1f us per 16 blocks" % (nb_, timeit(lambda: run(c_, s0))))


delay_buf = torch.empty(int(os.environ.get("DELAY_MB", "0")) * 262144 + 1, device=DEV)


def dual():
    ev = torch.cuda.Event(); ev.record(s0)
    s1.wait_event(ev); s2.wait_event(ev)
    if delay_buf.numel() > 1:
        with torch.cuda.stream(s2):
            delay_buf.fill_(1.0)          # phase offset between the chains
    run(ha, s1)
    run(hb, s2)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s1); e2.record(s2)
    s0.wait_event(e1); s0.wait_event(e2)


print("two chains of 8, 2 streams: %.1f us per 16 blocks" % timeit(dual))
