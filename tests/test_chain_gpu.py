"""The chained trunk (SRMI_CONV_FUSE_CHAIN: consecutive fused ResBlocks as ONE persistent launch whose workgroups keep
their tile in LDS and exchange halos through flags, csrc/conv_chain_bf16.hip) against the launch-per-block form of the
same sequence: bit-identical, every block's output included."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

L = importlib.import_module("super-resolution_amd.srmi._lib")
Fn = importlib.import_module("super-resolution_amd.srmi.functional")
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(N, H, W, nblk, seed, alphas=None, act2=None):
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.uniform(-1, 1, (N, H, W, 64)).astype(np.float32)).bfloat16().to(DEV)
    keep = [x]
    bufs = [torch.full((N, H, W, 64), 7.0, dtype=torch.bfloat16, device=DEV) for _ in range(nblk)]   # one output per block
    tmp = torch.full((N, H, W, 64), 3.0, dtype=torch.bfloat16, device=DEV)
    arr = (L.ConvArgs * (2 * nblk))()
    cur = x
    for b in range(nblk):
        a = 0.25 if alphas is None else alphas[b % len(alphas)]
        d1 = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16,
                          act=L.ACT_PRELU)
        d2 = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16,
                          act=L.ACT_NONE if act2 is None else act2, act_param=0.1)
        ws = [torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 30.0).astype(np.float32)).to(DEV) for _ in range(2)]
        wp = [Fn.pack_weights(d1, ws[0]), Fn.pack_weights(d2, ws[1])]
        sc = [torch.from_numpy(g.uniform(0.5, 1.2, 64).astype(np.float32)).to(DEV) for _ in range(2)]
        sh = [torch.from_numpy(g.uniform(-0.2, 0.2, 64).astype(np.float32)).to(DEV) for _ in range(2)]
        al = torch.tensor([a], dtype=torch.float32, device=DEV)
        keep += wp + sc + sh + [al]
        Fn.fill_args(arr[2 * b], d1, cur, wp[0], sc[0], sh[0], al, None, tmp)
        Fn.fill_args(arr[2 * b + 1], d2, tmp, wp[1], sc[1], sh[1], None, cur, bufs[b])
        cur = bufs[b]
    return arr, bufs, tmp, keep


def _run(arr, n, flags):
    for i in range(0, n, 2):
        arr[i].flags = flags
    L.check(L.load().srmi_conv2d_fwd_seq(arr, n, Fn.stream_ptr()))
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [
    dict(N=16, H=96, W=96, nblk=16),                  # cfg1's trunk: 256 workgroups, one per CU
    dict(N=1, H=96, W=96, nblk=3),
    dict(N=2, H=50, W=70, nblk=4),                    # ragged tiles both ways
    dict(N=3, H=12, W=65, nblk=2),                    # one tile row, second tile column one pixel wide + its halo
    dict(N=1, H=13, W=100, nblk=5, alphas=[0.25, -0.3, 1.0, 0.0]),   # one-pixel second tile row; compare + select slopes
    dict(N=4, H=60, W=144, nblk=6, act2=L.ACT_LRELU),                # 60 workgroups; an activation on the second conv
    dict(N=1, H=180, W=192, nblk=22),                 # 60 workgroups, the longest run of pairs in one launch
])
def test_chain_equals_launch_per_block(case):
    lib = L.load()
    N, H, W, nblk = case["N"], case["H"], case["W"], case["nblk"]
    arr, bufs, tmp, keep = _build(N, H, W, nblk, seed=5, alphas=case.get("alphas"), act2=case.get("act2"))
    n = 2 * nblk
    for i in range(0, n, 2):
        arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN
    assert lib.srmi_conv2d_chain_length(arr, n) == n
    _run(arr, n, L.CONV_FUSE_NEXT)                    # one launch per block
    ref = [b.clone() for b in bufs]
    for b in bufs:
        b.fill_(-9.0)
    _run(arr, n, L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN)
    assert lib.srmi_chain_status() == 0, lib.srmi_last_error()
    assert torch.all(tmp == 3.0)                      # the intermediates are never materialised
    for i, (r, b) in enumerate(zip(ref, bufs)):
        assert torch.equal(r, b), "block %d: %d values differ" % (i, int((r != b).sum()))
    # a second run right behind the first (flags re-zeroed on the stream), outputs poisoned in between
    for b in bufs:
        b.fill_(4.0)
    _run(arr, n, L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN)
    assert torch.equal(ref[-1], bufs[-1]) and torch.equal(ref[0], bufs[0])
    assert lib.srmi_chain_status() == 0
    # SRMI_CONV_Y_TRANSIENT: intermediate outputs reach memory only as the two-pixel border of every 12 x 48 tile
    for b in bufs:
        b.fill_(4.0)
    _run(arr, n, L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN | L.CONV_Y_TRANSIENT)
    assert lib.srmi_chain_status() == 0
    assert torch.equal(ref[-1], bufs[-1])             # the chain's result: written in full
    yy = torch.arange(H, device=DEV) % 12
    xx = torch.arange(W, device=DEV) % 48
    border = ((yy < 2) | (yy >= 10))[:, None] | ((xx < 2) | (xx >= 46))[None, :]
    for r, b in zip(ref[:-1], bufs[:-1]):
        assert torch.equal(r[:, border], b[:, border])
        assert torch.all(b[:, ~border] == 4.0)


@pytest.mark.parametrize("case", [
    dict(N=16, H=96, W=96, nblk=16),
    dict(N=2, H=50, W=70, nblk=2),
    dict(N=1, H=25, W=130, nblk=3, act=L.ACT_PRELU),
])
def test_chain_with_tail_conv(case):
    """The single conv behind the last block (conv02 + long skip add, P/srgan_model.py:33-34) as the chain's last link."""
    lib = L.load()
    N, H, W, nblk = case["N"], case["H"], case["W"], case["nblk"]
    arr0, bufs, tmp, keep = _build(N, H, W, nblk, seed=21)
    n = 2 * nblk + 1
    arr = (L.ConvArgs * n)()
    for i in range(n - 1):
        C.memmove(C.byref(arr[i]), C.byref(arr0[i]), C.sizeof(L.ConvArgs))
    g = np.random.default_rng(4)
    d = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16,
                     act=case.get("act", L.ACT_NONE))
    w = torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 30.0).astype(np.float32)).to(DEV)
    wp = Fn.pack_weights(d, w)
    sc = torch.from_numpy(g.uniform(0.5, 1.2, 64).astype(np.float32)).to(DEV)
    sh = torch.from_numpy(g.uniform(-0.2, 0.2, 64).astype(np.float32)).to(DEV)
    al = torch.tensor([0.3], dtype=torch.float32, device=DEV)
    x0 = keep[0]
    out = torch.full((N, H, W, 64), 7.0, dtype=torch.bfloat16, device=DEV)
    Fn.fill_args(arr[n - 1], d, bufs[-1], wp, sc, sh, al if case.get("act") else None, x0, out)

    def run(chain):
        for i in range(0, n - 1, 2):
            arr[i].flags = L.CONV_FUSE_NEXT | ((L.CONV_FUSE_CHAIN | L.CONV_Y_TRANSIENT) if chain else 0)
        arr[n - 1].flags = L.CONV_FUSE_CHAIN if chain else 0
        L.check(lib.srmi_conv2d_fwd_seq(arr, n, Fn.stream_ptr()))
        torch.cuda.synchronize()

    run(False)
    ref = out.clone()
    out.fill_(-2.0)
    for i in range(0, n - 1, 2):
        arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN
    arr[n - 1].flags = L.CONV_FUSE_CHAIN
    assert lib.srmi_conv2d_chain_length(arr, n) == n
    run(True)
    assert lib.srmi_chain_status() == 0
    assert torch.equal(ref, out), int((ref != out).sum())
    # a tail whose residual is one of the chain's own outputs stays a launch of its own
    arr[n - 1].residual = bufs[0].data_ptr()
    assert lib.srmi_conv2d_chain_length(arr, n) == n - 1


@pytest.mark.parametrize("case", [
    dict(N=16, H=96, W=96, nblk=16, cout=256, r=2, xcopy=True, tail1=True),      # cfg1: trunk + conv02 + first Upsampler conv
    dict(N=2, H=50, W=72, nblk=2, cout=256, r=2, dcr=True, alpha_pc=True, xcopy=True, tail1=True),
    dict(N=1, H=25, W=130, nblk=3, cout=128, r=1, tail1=False),                  # wide link right behind the pairs, plain store
    dict(N=3, H=12, W=65, nblk=2, cout=64, r=1, tail1=True),
    dict(N=1, H=40, W=100, nblk=2, cout=192, r=1, xcopy=True, tail1=True),
])
def test_chain_with_wide_last_link(case):
    """... + a conv to 64..256 channels (PixelShuffle in the store addressing, optional NCHW f32 copy of its input) as the
    chain's last link, against the same sequence one launch per conv: every written tensor bit-identical."""
    lib = L.load()
    N, H, W, nblk = case["N"], case["H"], case["W"], case["nblk"]
    arr0, bufs, tmp, keep = _build(N, H, W, nblk, seed=31)
    ntail = 2 if case["tail1"] else 1
    n = 2 * nblk + ntail
    arr = (L.ConvArgs * n)()
    for i in range(2 * nblk):
        C.memmove(C.byref(arr[i]), C.byref(arr0[i]), C.sizeof(L.ConvArgs))
    g = np.random.default_rng(8)
    cur = bufs[-1]
    feat = None
    if case["tail1"]:
        d = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_NONE)
        w = torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 30.0).astype(np.float32)).to(DEV)
        wp = Fn.pack_weights(d, w)
        sc = torch.from_numpy(g.uniform(0.5, 1.2, 64).astype(np.float32)).to(DEV)
        sh = torch.from_numpy(g.uniform(-0.2, 0.2, 64).astype(np.float32)).to(DEV)
        feat = torch.full((N, H, W, 64), 7.0, dtype=torch.bfloat16, device=DEV)
        Fn.fill_args(arr[2 * nblk], d, cur, wp, sc, sh, None, keep[0], feat)
        keep += [wp, sc, sh]
        cur = feat
    cout, r = case["cout"], case["r"]
    apc = 1 if case.get("alpha_pc") else 0
    du = Fn.make_desc(N, H, W, 64, cout, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16,
                      act=L.ACT_PRELU, alpha_per_channel=apc, shuffle=r,
                      shuffle_order=L.SHUFFLE_DCR if case.get("dcr") else L.SHUFFLE_CRD)
    wu = torch.from_numpy((g.standard_normal((cout, 64, 3, 3)) / 30.0).astype(np.float32)).to(DEV)
    wpu = Fn.pack_weights(du, wu)
    scu = torch.from_numpy(g.uniform(0.5, 1.2, cout).astype(np.float32)).to(DEV)
    shu = torch.from_numpy(g.uniform(-0.2, 0.2, cout).astype(np.float32)).to(DEV)
    alu = torch.from_numpy(g.uniform(-0.2, 0.6, cout if apc else 1).astype(np.float32)).to(DEV)
    up = torch.full((N, H * r, W * r, cout // (r * r)), 5.0, dtype=torch.bfloat16, device=DEV)
    Fn.fill_args(arr[n - 1], du, cur, wpu, scu, shu, alu, None, up)
    xc = torch.full((N, 64, H, W), -3.0, dtype=torch.float32, device=DEV) if case.get("xcopy") else None
    if xc is not None:
        arr[n - 1].x_copy = xc.data_ptr()

    def run(chain):
        for i in range(0, 2 * nblk, 2):
            arr[i].flags = L.CONV_FUSE_NEXT | ((L.CONV_FUSE_CHAIN | L.CONV_Y_TRANSIENT) if chain else 0)
        for i in range(2 * nblk, n):
            arr[i].flags = (L.CONV_FUSE_CHAIN | L.CONV_Y_TRANSIENT) if chain else 0
        L.check(lib.srmi_conv2d_fwd_seq(arr, n, Fn.stream_ptr()))
        torch.cuda.synchronize()

    run(False)
    ref_up = up.clone()
    ref_xc = xc.clone() if xc is not None else None
    up.fill_(-2.0)
    if xc is not None:
        xc.fill_(9.0)
    if feat is not None:
        feat.fill_(4.0)
    for i in range(n):
        arr[i].flags = L.CONV_FUSE_CHAIN | (L.CONV_FUSE_NEXT if i < 2 * nblk and i % 2 == 0 else 0)
    assert lib.srmi_conv2d_chain_length(arr, n) == n
    run(True)
    assert lib.srmi_chain_status() == 0
    assert torch.equal(ref_up, up), int((ref_up != up).sum())
    if xc is not None:
        assert torch.equal(ref_xc, xc)
    if feat is not None:
        # conv02's output was transient: only the tile borders reached memory
        yy = torch.arange(H, device=DEV) % 12
        xx = torch.arange(W, device=DEV) % 48
        border = ((yy < 2) | (yy >= 10))[:, None] | ((xx < 2) | (xx >= 46))[None, :]
        assert torch.all(feat[:, ~border] == 4.0)


@pytest.mark.parametrize("case", [
    dict(N=16, H=96, W=96, nblk=16, nchw=True),       # cfg1's whole front: conv01 + trunk + conv02 (+ the up conv in the generator test)
    dict(N=2, H=50, W=70, nblk=2, nchw=True),
    dict(N=1, H=25, W=130, nblk=2, nchw=False, cin=2),
    dict(N=3, H=13, W=65, nblk=3, nchw=False, bf16=True, cin=3),
])
def test_chain_with_head_conv(case):
    """conv 9x9 from <= 3 image channels (conv01, P/srgan_model.py:29) as the chain's first link, conv + skip add as its
    last: the skip tensor is written AND read back inside the one launch."""
    lib = L.load()
    N, H, W, nblk = case["N"], case["H"], case["W"], case["nblk"]
    cin = case.get("cin", 3)
    arr0, bufs, tmp, keep = _build(N, H, W, nblk, seed=41)
    n = 2 * nblk + 2
    arr = (L.ConvArgs * n)()
    for i in range(2 * nblk):
        C.memmove(C.byref(arr[i + 1]), C.byref(arr0[i]), C.sizeof(L.ConvArgs))
    g = np.random.default_rng(14)
    skip = keep[0]                                  # the pairs' input buffer becomes conv01's output
    skip.fill_(1.5)
    if case["nchw"]:
        img = torch.from_numpy(g.uniform(0, 1, (N, cin, H, W)).astype(np.float32)).to(DEV)
        lay, dt = L.NCHW, L.F32
    else:
        img = torch.from_numpy(g.uniform(0, 1, (N, H, W, cin)).astype(np.float32)).to(DEV)
        lay, dt = L.NHWC, L.F32
        if case.get("bf16"):
            img, dt = img.bfloat16(), L.BF16
    d0 = Fn.make_desc(N, H, W, cin, 64, 9, in_dtype=dt, out_dtype=L.BF16, res_dtype=L.BF16, in_layout=lay, compute=L.COMPUTE_BF16,
                      act=L.ACT_PRELU)
    assert L.kernel_name(d0) == "k9c3_bf16"
    w0 = torch.from_numpy((g.standard_normal((64, cin, 9, 9)) / 9.0).astype(np.float32)).to(DEV)
    wp0 = Fn.pack_weights(d0, w0)
    sc0 = torch.ones(64, device=DEV)
    sh0 = torch.from_numpy(g.uniform(-0.2, 0.2, 64).astype(np.float32)).to(DEV)
    al0 = torch.tensor([0.25], dtype=torch.float32, device=DEV)
    Fn.fill_args(arr[0], d0, img, wp0, sc0, sh0, al0, None, skip)
    d = Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_NONE)
    w = torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 30.0).astype(np.float32)).to(DEV)
    wp = Fn.pack_weights(d, w)
    sc = torch.from_numpy(g.uniform(0.5, 1.2, 64).astype(np.float32)).to(DEV)
    sh = torch.from_numpy(g.uniform(-0.2, 0.2, 64).astype(np.float32)).to(DEV)
    out = torch.full((N, H, W, 64), 7.0, dtype=torch.bfloat16, device=DEV)
    Fn.fill_args(arr[n - 1], d, bufs[-1], wp, sc, sh, None, skip, out)

    def run(chain):
        arr[0].flags = L.CONV_FUSE_CHAIN if chain else 0
        for i in range(1, n - 1, 2):
            arr[i].flags = L.CONV_FUSE_NEXT | ((L.CONV_FUSE_CHAIN | L.CONV_Y_TRANSIENT) if chain else 0)
        arr[n - 1].flags = L.CONV_FUSE_CHAIN if chain else 0
        L.check(lib.srmi_conv2d_fwd_seq(arr, n, Fn.stream_ptr()))
        torch.cuda.synchronize()

    run(False)
    ref, ref_skip = out.clone(), skip.clone()
    out.fill_(-2.0)
    skip.fill_(0.5)
    arr[0].flags = L.CONV_FUSE_CHAIN
    for i in range(1, n - 1, 2):
        arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN
    arr[n - 1].flags = L.CONV_FUSE_CHAIN
    assert lib.srmi_conv2d_chain_length(arr, n) == n
    run(True)
    assert lib.srmi_chain_status() == 0
    assert torch.equal(ref_skip, skip), int((ref_skip != skip).sum())
    assert torch.equal(ref, out), int((ref != out).sum())


def test_chain_refused_shapes_fall_back():
    """More workgroups than CUs, narrow images, a foreign residual: no chain, same results through single launches."""
    lib = L.load()
    arr, bufs, tmp, keep = _build(8, 144, 576, 3, seed=9)     # 144 workgroups per image: one image per launch would fill 56 % of the chip
    for i in range(0, 6, 2):
        arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN
    assert lib.srmi_conv2d_chain_length(arr, 6) == 0
    del arr, bufs, tmp, keep
    arr, bufs, tmp, keep = _build(2, 40, 64, 3, seed=9)       # W <= 64: the small-tile kernel's shapes
    for i in range(0, 6, 2):
        arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN
    assert lib.srmi_conv2d_chain_length(arr, 6) == 0
    _run(arr, 6, L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN)
    got = bufs[-1].clone()
    _run(arr, 6, L.CONV_FUSE_NEXT)
    assert torch.equal(got, bufs[-1])
    # 24 blocks: 22 go out as one chain, the last two as a second one
    arr, bufs, tmp, keep = _build(1, 24, 96, 24, seed=12)
    for i in range(0, 48, 2):
        arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN
    assert lib.srmi_conv2d_chain_length(arr, 48) == 44
    assert lib.srmi_conv2d_chain_length(C.byref(arr[44]), 4) == 4
    _run(arr, 48, L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN | L.CONV_Y_TRANSIENT)
    got = bufs[-1].clone()
    _run(arr, 48, L.CONV_FUSE_NEXT)
    assert torch.equal(got, bufs[-1])
    # chain of two out of three: the middle block's residual is another tensor
    arr, bufs, tmp, keep = _build(1, 48, 96, 4, seed=10)
    other = torch.zeros_like(bufs[0])
    arr[3].residual = other.data_ptr()
    for i in range(0, 8, 2):
        arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN
    assert lib.srmi_conv2d_chain_length(arr, 8) == 0          # block 0 alone is no chain
    assert lib.srmi_conv2d_chain_length(C.byref(arr[4]), 4) == 4
    _run(arr, 8, L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN)
    got = [b.clone() for b in bufs]
    _run(arr, 8, L.CONV_FUSE_NEXT)
    for r, b in zip(got, bufs):
        assert torch.equal(r, b)


def test_chain_over_sub_batches_of_whole_images():
    """More tiles than CUs, but whole images fit: the chain runs as several launches of k images each (every per-image
    pointer advanced) - same bits as one launch per block.  8 x 96 x 200 = 8 x 40 workgroups: 6 + 2 images."""
    lib = L.load()
    arr, bufs, tmp, keep = _build(8, 96, 200, 3, seed=9)
    for i in range(0, 6, 2):
        arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN
    assert lib.srmi_conv2d_chain_length(arr, 6) == 6
    _run(arr, 6, L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN)
    got = [b.clone() for b in bufs]
    _run(arr, 6, L.CONV_FUSE_NEXT)
    assert lib.srmi_chain_status() == 0
    for r, b in zip(got, bufs):
        assert torch.equal(r, b)
    # the whole generator front (head conv, 5 blocks, conv02 + skip, wide up conv with its NCHW f32 copy): 20 images of 16
    # tiles = 16 + 4 per launch
    srmi = importlib.import_module("super-resolution_amd.srmi")
    engine = importlib.import_module("super-resolution_amd.srmi.engine")
    torch.manual_seed(5)
    gen = srmi.Generator(img_feat=3, n_feats=64, kernel_size=3, num_block=5, scale=4).to(DEV).eval().set_compute("bf16")
    x = torch.rand(20, 3, 96, 96, device=DEV)
    with torch.no_grad():
        sr1, f1 = gen(x)
        plan = gen._plans.get(gen, 20, 96, 96, 1)
        assert any(lbl == "resblock_chain_bf16" and cnt == 13 for (lbl, i0, cnt) in plan.launches), plan.launches
        engine.set_fuse_chain(False)
        try:
            engine.invalidate(gen)
            sr2, f2 = gen(x)
        finally:
            engine.set_fuse_chain(True)
            engine.invalidate(gen)
    torch.cuda.synchronize()
    assert torch.equal(sr1, sr2) and torch.equal(f1, f2)
    assert lib.srmi_chain_status() == 0


def test_generator_chain_switch():
    """`srmi.engine.set_fuse_chain(False)` gives the launch-per-block plan; both forms return the same bits."""
    srmi = importlib.import_module("super-resolution_amd.srmi")
    from importlib import import_module
    engine = import_module("super-resolution_amd.srmi.engine")
    torch.manual_seed(3)
    gen = srmi.Generator(img_feat=3, n_feats=64, kernel_size=3, num_block=5, scale=4).to(DEV).eval().set_compute("bf16")
    x = torch.rand(2, 3, 48, 96, device=DEV)
    with torch.no_grad():
        sr1, f1 = gen(x)
        plan = gen._plans.get(gen, 2, 48, 96, 1)
        assert any(lbl == "resblock_chain_bf16" and cnt == 13 for (lbl, i0, cnt) in plan.launches), plan.launches
        engine.set_fuse_chain(False)
        try:
            engine.invalidate(gen)
            sr2, f2 = gen(x)
            plan2 = gen._plans.get(gen, 2, 48, 96, 1)
            assert not any(lbl == "resblock_chain_bf16" for (lbl, i0, cnt) in plan2.launches)
        finally:
            engine.set_fuse_chain(True)
            engine.invalidate(gen)
    torch.cuda.synchronize()
    assert torch.equal(sr1, sr2) and torch.equal(f1, f2)
    assert L.load().srmi_chain_status() == 0


def test_chain_soak_and_uneven_load():
    """Hand-offs under repetition and under uneven load (MI355X_MICROARCH.md: 'test every hand-off under UNEVEN load,
    consumer L1-warm, checking every word'): 150 chained launches back to back over two alternating inputs, the second
    half of them while another stream keeps part of the chip busy with a memory-streaming kernel (so some of the chain's
    workgroups start late and their neighbours have to wait) - every launch's final tensor is compared bit for bit with
    the launch-per-block result of its input."""
    lib = L.load()
    N, H, W, nblk = 16, 96, 96, 6
    arr, bufs, tmp, keep = _build(N, H, W, nblk, seed=77)
    n = 2 * nblk
    xa = keep[0]
    xb = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, (N, H, W, 64)).astype(np.float32)).bfloat16().to(DEV)
    xs = [xa.clone(), xb]
    refs = []
    for xin in xs:
        xa.copy_(xin)
        _run(arr, n, L.CONV_FUSE_NEXT)
        refs.append(bufs[-1].clone())
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    outs = []
    for it in range(150):
        xa.copy_(xs[it & 1])
        if it >= 75:
            with torch.cuda.stream(side):
                big.add_(1.0)                         # 512 MB of traffic on the other stream, concurrently
        for i in range(0, n, 2):
            arr[i].flags = L.CONV_FUSE_NEXT | L.CONV_FUSE_CHAIN | L.CONV_Y_TRANSIENT
        L.check(lib.srmi_conv2d_fwd_seq(arr, n, Fn.stream_ptr()))
        outs.append(bufs[-1].clone())
    torch.cuda.synchronize()
    assert lib.srmi_chain_status() == 0, lib.srmi_last_error()
    bad = [it for it, o in enumerate(outs) if not torch.equal(o, refs[it & 1])]
    assert not bad, bad[:10]
