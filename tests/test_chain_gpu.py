"""Persistent ResBlock chain (srmi_conv2d_fwd_seq) == the same blocks launched one by one (srmi_resblock_fwd), bit for bit."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

L = importlib.import_module("super-resolution_amd.srmi._lib")
Fn = importlib.import_module("super-resolution_amd.srmi.functional")
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _blocks(N, H, W, nblk, seed):
    g = np.random.default_rng(seed)
    d1 = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_PRELU)
    d2 = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_NONE)
    keep = []
    x0 = torch.from_numpy(g.uniform(-1, 1, (N, H, W, 64)).astype(np.float32)).bfloat16().to(DEV)
    bufs = [x0, torch.zeros_like(x0), torch.zeros_like(x0)]
    tmp = torch.full_like(x0, 3.0)
    arr = (L.ConvArgs * (2 * nblk))()
    cur = 0
    for b in range(nblk):
        w1 = torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 30.0).astype(np.float32)).to(DEV)
        w2 = torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 30.0).astype(np.float32)).to(DEV)
        wp1, wp2 = Fn.pack_weights(d1, w1), Fn.pack_weights(d2, w2)
        sc1 = torch.from_numpy(g.uniform(0.5, 1.5, 64).astype(np.float32)).to(DEV)
        sh1 = torch.from_numpy(g.uniform(-0.3, 0.3, 64).astype(np.float32)).to(DEV)
        sc2 = torch.from_numpy(g.uniform(0.05, 0.15, 64).astype(np.float32)).to(DEV)
        sh2 = torch.from_numpy(g.uniform(-0.03, 0.03, 64).astype(np.float32)).to(DEV)
        al = torch.tensor([0.25], device=DEV)
        out = 1 + (b & 1)                     # ping-pong between bufs[1] and bufs[2]; block 0 reads bufs[0]
        Fn.fill_args(arr[2 * b], d1, bufs[cur], wp1, sc1, sh1, al, None, tmp)
        Fn.fill_args(arr[2 * b + 1], d2, tmp, wp2, sc2, sh2, None, bufs[cur], bufs[out])
        arr[2 * b].flags = L.CONV_FUSE_NEXT
        keep += [wp1, wp2, sc1, sh1, sc2, sh2, al]
        cur = out
    return arr, bufs, cur, tmp, keep


@pytest.mark.parametrize("case", [
    dict(N=16, H=96, W=96, nblk=16),          # cfg1: 256 tiles, one per CU
    dict(N=2, H=13, W=100, nblk=3),           # two tile columns, ragged both ways: 8 neighbours
    dict(N=1, H=24, W=192, nblk=5),
    dict(N=3, H=6, W=96, nblk=2),             # one tile per image: no neighbours at all
    dict(N=1, H=40, W=70, nblk=20),           # more blocks than one launch carries (16 + 4)
])
def test_chain_matches_per_block_launches(case):
    lib = L.load()
    prev = lib.srmi_chain_enable(1)
    N, H, W, nblk = case["N"], case["H"], case["W"], case["nblk"]
    t0 = C.c_uint32(0)
    L.check(lib.srmi_chain_status(C.byref(t0)))
    for rep in range(2):                     # twice: the flag words of the first run must not satisfy the second
        arr, bufs, last, tmp, keep = _blocks(N, H, W, nblk, seed=11 + rep)
        x0 = bufs[0].clone()
        L.check(lib.srmi_conv2d_fwd_seq(arr, 2 * nblk, Fn.stream_ptr()))
        torch.cuda.synchronize()
        got = bufs[last].clone()
        assert torch.equal(bufs[0], x0)
        assert torch.all(tmp == 3.0)
        bufs[1].zero_(); bufs[2].zero_()
        for b in range(nblk):
            L.check(lib.srmi_resblock_fwd(C.byref(arr[2 * b]), C.byref(arr[2 * b + 1]), Fn.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got, bufs[last])
    t1 = C.c_uint32(0)
    L.check(lib.srmi_chain_status(C.byref(t1)))
    lib.srmi_chain_enable(prev)
    assert t1.value == t0.value, "a neighbour wait timed out"
