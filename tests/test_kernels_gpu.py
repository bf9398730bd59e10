"""GPU: single-kernel parity through the C ABI against a CPU torch restatement of the same fused op
(conv -> scale/shift -> act -> +residual -> sub-pixel store).  Integer/index work (layouts, shuffles,
slices) must be exact; f32 MFMA within summation-order error; bf16 within one output rounding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from srmi import _lib as L
from srmi import functional as Fn

pytestmark = pytest.mark.gpu
# the kernels of the f32 path, by the name srmi_conv2d_kernel_name reports (the one that runs): exact f32 MFMA without /
# with LDS tiling, split-bf16 general / halo-tile
F32_KERNELS = ("generic_f32", "f32t", "b3_f32", "b3h_f32")
DEV = "cuda:0"


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _ref_fused(x_nchw, w, scale, shift, act, act_param, alpha, residual_nchw, stride, pad, r, dcr):
    """CPU f64 restatement of one srmi_conv_desc."""
    z = F.conv2d(x_nchw.double(), w.double(), None, stride=stride, padding=pad)
    z = z * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if act == L.ACT_PRELU:
        a = alpha.double().view(1, -1, 1, 1) if alpha.numel() > 1 else alpha.double()
        z = torch.where(z >= 0, z, z * a)
    elif act == L.ACT_LRELU:
        z = F.leaky_relu(z, act_param)
    elif act == L.ACT_RELU:
        z = F.relu(z)
    elif act == L.ACT_TANH:
        z = torch.tanh(z)
    elif act == L.ACT_SIGMOID:
        z = torch.sigmoid(z)
    if r > 1:
        if dcr:   # depth_to_space: channel = (i*r+j)*C + c
            n, c, h, wd = z.shape
            cf = c // (r * r)
            z = z.view(n, r, r, cf, h, wd).permute(0, 3, 4, 1, 5, 2).reshape(n, cf, h * r, wd * r)
        else:
            z = F.pixel_shuffle(z, r)
    if residual_nchw is not None:
        z = z + residual_nchw.double()
    if act == L.ACT_ADD_RELU:   # the one activation that follows the residual
        z = F.relu(z)
    return z


def _run_case(N, H, W, Cin, Cout, k, *, stride=1, act=L.ACT_NONE, act_param=0.0, alpha_pc=0, alpha_val=None, residual=False, r=1,
              dcr=False, io=torch.float32, compute=L.COMPUTE_F32, in_layout=L.NHWC, out_layout=L.NHWC,
              in_cs=0, out_cs=0, out_co=0, res_cs=0, seed=0):
    g = _rng(seed)
    x = torch.from_numpy(g.uniform(-1, 1, (N, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((g.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32))
    scale = torch.from_numpy(g.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(g.uniform(-0.3, 0.3, Cout).astype(np.float32))
    alpha = torch.from_numpy(g.uniform(0.05, 0.4, Cout if alpha_pc else 1).astype(np.float32))
    if alpha_val is not None:            # a scalar slope outside [0, 1]: the compare + select form, not max(z, z*slope)
        alpha = torch.tensor([alpha_val], dtype=torch.float32)
    bf = io == torch.bfloat16
    if bf:                       # the kernel sees bf16-rounded activations / weights: so does the reference
        x = x.bfloat16().float()
        if compute == L.COMPUTE_BF16:
            w = w.bfloat16().float()
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    Cf = Cout // (r * r)
    res = None
    if residual:
        res = torch.from_numpy(g.uniform(-1, 1, (N, Cf, Ho * r, Wo * r)).astype(np.float32))
        if bf:
            res = res.bfloat16().float()
    ref = _ref_fused(x, w, scale, shift, act, act_param, alpha, res, stride, pad, r, dcr)

    dio = L.BF16 if bf else L.F32
    desc = Fn.make_desc(N, H, W, Cin, Cout, k, stride=stride, in_dtype=dio, out_dtype=dio, res_dtype=dio,
                        in_layout=in_layout, out_layout=out_layout, compute=compute, act=act, act_param=act_param,
                        alpha_per_channel=alpha_pc, shuffle=r, shuffle_order=L.SHUFFLE_DCR if dcr else L.SHUFFLE_CRD,
                        in_cstride=in_cs, out_cstride=out_cs, out_coffset=out_co, res_cstride=res_cs)
    # device buffers
    if in_layout == L.NHWC:
        cs = in_cs or Cin
        xd = torch.full((N, H, W, cs), 7.0, dtype=io)          # poison the unused slice channels
        xd[..., :Cin] = x.permute(0, 2, 3, 1).to(io)
    else:
        xd = x.to(io)
    xd = xd.to(DEV)
    wp = Fn.pack_weights(desc, w.to(DEV))
    if out_layout == L.NHWC:
        ocs = out_cs or Cf
        yd = torch.full((N, Ho * r, Wo * r, ocs), -5.0, dtype=io, device=DEV)
        rd = res.permute(0, 2, 3, 1).contiguous().to(io).to(DEV) if residual else None
        if residual and res_cs:                                  # residual = leading channels of a wider buffer
            wide = torch.full((N, Ho * r, Wo * r, res_cs), 9.0, dtype=io, device=DEV)
            wide[..., :Cf] = rd
            rd = wide
    else:
        yd = torch.full((N, Cf, Ho * r, Wo * r), -5.0, dtype=io, device=DEV)
        rd = res.to(io).to(DEV) if residual else None
    Fn.conv2d_fused(desc, xd, wp, scale.to(DEV), shift.to(DEV), alpha.to(DEV), rd, out=yd)
    torch.cuda.synchronize()
    y = yd.float().cpu()
    if out_layout == L.NHWC:
        got = y[..., out_co:out_co + Cf].permute(0, 3, 1, 2)
        # channels outside the written slice must be untouched
        if out_cs:
            mask = torch.ones(out_cs, dtype=torch.bool)
            mask[out_co:out_co + Cf] = False
            assert torch.all(y[..., mask] == -5.0)
    else:
        got = y
    err = (got.double() - ref).abs()
    tol = 3e-5 if not bf else 2.0 ** -8 * (ref.abs() + 1.0)     # bf16: one rounding of the output (+ of x)
    assert torch.all(err <= tol), f"max err {err.max().item():.3e} ({L.kernel_name(desc)})"
    return L.kernel_name(desc)


def test_layout_convert_exact():
    g = _rng(5)
    x = torch.from_numpy(g.uniform(-1, 1, (2, 5, 7, 9)).astype(np.float32)).to(DEV)
    nhwc = Fn.layout_convert(x, 2, 5, 7, 9, L.NCHW, L.NHWC, torch.float32)
    assert torch.equal(nhwc, x.permute(0, 2, 3, 1).contiguous())
    back = Fn.layout_convert(nhwc, 2, 5, 7, 9, L.NHWC, L.NCHW, torch.float32)
    assert torch.equal(back, x)
    b = Fn.layout_convert(x, 2, 5, 7, 9, L.NCHW, L.NHWC, torch.bfloat16)
    assert torch.equal(b, x.permute(0, 2, 3, 1).contiguous().bfloat16())     # RNE cast matches torch's
    # LDS-transposed NHWC -> NCHW path (C >= 8), ragged in pixels and channels
    # (H*W % 4 == 0 with bf16 input takes the vectorised kernel)
    for (n, c, h, w) in [(2, 64, 7, 9), (1, 72, 5, 13), (3, 8, 1, 1), (2, 64, 8, 10), (1, 72, 4, 13), (1, 128, 16, 17), (3, 8, 2, 2)]:
        t = torch.from_numpy(g.uniform(-1, 1, (n, h, w, c)).astype(np.float32)).to(DEV)
        assert torch.equal(Fn.layout_convert(t, n, c, h, w, L.NHWC, L.NCHW, torch.float32), t.permute(0, 3, 1, 2).contiguous())
        tb = t.bfloat16()
        assert torch.equal(Fn.layout_convert(tb, n, c, h, w, L.NHWC, L.NCHW, torch.float32),
                           tb.float().permute(0, 3, 1, 2).contiguous())


@pytest.mark.parametrize("case", [
    dict(N=1, H=9, W=7, Cin=3, Cout=64, k=9, act=L.ACT_PRELU, in_layout=L.NCHW),                # conv01
    dict(N=2, H=11, W=13, Cin=64, Cout=64, k=3, act=L.ACT_PRELU),                               # ResBlock conv 1
    dict(N=2, H=11, W=13, Cin=64, Cout=64, k=3, residual=True),                                 # ResBlock conv 2
    dict(N=1, H=6, W=5, Cin=64, Cout=256, k=3, act=L.ACT_PRELU, r=2),                           # Upsampler x2
    dict(N=1, H=6, W=5, Cin=64, Cout=576, k=3, act=L.ACT_PRELU, r=3),                           # Upsampler x3
    dict(N=1, H=6, W=5, Cin=64, Cout=256, k=3, act=L.ACT_PRELU, alpha_pc=1, r=2, dcr=True),     # Keras depth_to_space
    dict(N=2, H=10, W=12, Cin=64, Cout=3, k=3, act=L.ACT_TANH, out_layout=L.NCHW),              # last_conv
    dict(N=2, H=12, W=12, Cin=64, Cout=64, k=3, stride=2, act=L.ACT_LRELU, act_param=0.01),     # D stride-2
    dict(N=2, H=9, W=9, Cin=128, Cout=128, k=3, stride=2, act=L.ACT_LRELU, act_param=0.01),     # odd size stride 2
    dict(N=3, H=1, W=1, Cin=2048, Cout=1024, k=1, act=L.ACT_LRELU, act_param=0.01),             # Linear as 1x1
    dict(N=3, H=1, W=1, Cin=1024, Cout=1, k=1, act=L.ACT_SIGMOID),                              # Linear -> sigmoid
    dict(N=1, H=12, W=12, Cin=1, Cout=128, k=9, act=L.ACT_RELU),                                # SRCNN layer 1
    dict(N=1, H=12, W=12, Cin=64, Cout=1, k=5),                                                 # SRCNN layer 3
    dict(N=1, H=8, W=8, Cin=96, Cout=32, k=3, act=L.ACT_LRELU, act_param=0.2, in_cs=160, out_cs=160, out_co=96),  # RRDB dense
    dict(N=1, H=5, W=40, Cin=20, Cout=40, k=3, act=L.ACT_RELU),                                 # ragged channels
    dict(N=2, H=11, W=13, Cin=64, Cout=64, k=3, act=L.ACT_ADD_RELU, residual=True),             # res_conv_block: relu(conv + shortcut)
    dict(N=1, H=6, W=8, Cin=128, Cout=64, k=3, act=L.ACT_ADD_RELU, residual=True, out_cs=192, out_co=128),
    dict(N=1, H=6, W=8, Cin=64, Cout=64, k=3, act=L.ACT_ADD_RELU),                              # no residual: a plain ReLU
])
def test_generic_f32_kernel(case):
    assert _run_case(**case) in F32_KERNELS


@pytest.mark.parametrize("case", [
    dict(N=2, H=11, W=13, Cin=64, Cout=64, k=3, act=L.ACT_PRELU),
    dict(N=1, H=6, W=5, Cin=64, Cout=256, k=3, act=L.ACT_PRELU, r=2),
    dict(N=2, H=10, W=12, Cin=64, Cout=3, k=3, act=L.ACT_TANH, out_layout=L.NCHW),
    dict(N=2, H=11, W=13, Cin=64, Cout=64, k=3, act=L.ACT_ADD_RELU, residual=True),
    dict(N=2, H=11, W=13, Cin=64, Cout=64, k=1, act=L.ACT_ADD_RELU, residual=True, compute=L.COMPUTE_BF16),  # 1x1: no LDS-tiled kernel
])
def test_generic_kernel_bf16_storage(case):
    assert _run_case(io=torch.bfloat16, **case) == "generic_f32"          # bf16 storage: only the no-LDS kernel reads it


@pytest.mark.parametrize("case", [
    dict(N=2, H=11, W=13, Cout=64, act=L.ACT_PRELU),                          # small tile, ragged edges
    dict(N=2, H=11, W=13, Cout=64, residual=True),
    dict(N=1, H=4, W=32, Cout=64, act=L.ACT_LRELU, act_param=0.2),            # exactly one small tile
    dict(N=1, H=13, W=100, Cout=64, act=L.ACT_PRELU, residual=True),          # big 6x96 tile, ragged
    dict(N=2, H=6, W=96, Cout=64, act=L.ACT_RELU),                            # exactly one big tile per image
    dict(N=1, H=24, W=192, Cout=64, act=L.ACT_PRELU, alpha_pc=1),
    dict(N=1, H=7, W=9, Cout=256, act=L.ACT_PRELU, r=2),                      # PixelShuffle store (CRD)
    dict(N=1, H=12, W=97, Cout=256, act=L.ACT_PRELU, r=2),
    dict(N=1, H=7, W=9, Cout=256, act=L.ACT_PRELU, alpha_pc=1, r=2, dcr=True),  # depth_to_space store (DCR)
    dict(N=1, H=9, W=20, Cout=128, act=L.ACT_LRELU, act_param=0.01),          # D conv 64->128
    dict(N=1, H=8, W=8, Cout=64, act=L.ACT_LRELU, act_param=0.2, in_cs=160, out_cs=160, out_co=64),  # concat slices
    dict(N=1, H=15, W=120, Cout=64, act=L.ACT_PRELU, residual=True),          # 8 x 64 tile (fewer rounds), ragged
    dict(N=1, H=16, W=128, Cout=256, act=L.ACT_PRELU, r=2),                   # 8 x 64 tile + PixelShuffle store
    dict(N=1, H=9, W=20, Cout=64, act=L.ACT_PRELU, alpha_val=-0.5),           # learned slope outside [0, 1]
    dict(N=1, H=9, W=20, Cout=64, act=L.ACT_PRELU, alpha_val=1.5, residual=True),
    dict(N=1, H=9, W=20, Cout=64, act=L.ACT_LRELU, act_param=1.25),
    dict(N=2, H=13, W=70, Cout=64, act=L.ACT_NONE, residual=True, res_cs=160),   # residual = channel slice of a concat buffer (RRDB trunk skip)
])
def test_c64_bf16_kernel(case):
    name = _run_case(Cin=64, k=3, io=torch.bfloat16, compute=L.COMPUTE_BF16, **case)
    assert name == "c64_bf16"


@pytest.mark.parametrize("case", [
    # the ESRGAN-style dense block (ESRGAN/untitled0.py:18-35): inputs and outputs are slices of one 160-ch buffer
    dict(N=1, H=9, W=11, Cin=64, Cout=32, act=L.ACT_LRELU, act_param=0.2, in_cs=160, out_cs=160, out_co=64),
    dict(N=1, H=9, W=11, Cin=96, Cout=32, act=L.ACT_LRELU, act_param=0.2, in_cs=160, out_cs=160, out_co=96),
    dict(N=2, H=8, W=33, Cin=128, Cout=32, act=L.ACT_LRELU, act_param=0.2, in_cs=160, out_cs=160, out_co=128),
    dict(N=1, H=9, W=11, Cin=160, Cout=64, residual=True, res_cs=160),        # compress conv + x (slice residual)
    dict(N=1, H=13, W=100, Cin=160, Cout=64, residual=True, res_cs=160),      # big 6x96 tile, ragged
    dict(N=1, H=12, W=97, Cin=96, Cout=32, act=L.ACT_LRELU, act_param=0.2),   # dense (no slices), big tile
    dict(N=1, H=7, W=70, Cin=128, Cout=64, act=L.ACT_RELU),                   # SRCNN conv2 (SRCNN/srcnn.py:205)
    dict(N=1, H=6, W=10, Cin=256, Cout=128, act=L.ACT_PRELU, alpha_pc=1),     # 4 K-chunks, 2 channel groups
    dict(N=1, H=5, W=6, Cin=32, Cout=96, act=L.ACT_PRELU),                    # half chunk only, 3 groups of 32
    dict(N=2, H=6, W=96, Cin=512, Cout=64, act=L.ACT_RELU),                   # VGG-width contraction
    dict(N=1, H=15, W=120, Cin=160, Cout=64, residual=True, res_cs=160),      # 8 x 64 tile, ragged
    dict(N=1, H=16, W=128, Cin=96, Cout=32, act=L.ACT_LRELU, act_param=0.2, in_cs=160, out_cs=160, out_co=96),
    # res_conv_block's relu(conv + shortcut) (Res-U-SRGAN/untitled0.py:67-72): the activation after the residual
    dict(N=2, H=11, W=13, Cin=64, Cout=64, act=L.ACT_ADD_RELU, residual=True),
    dict(N=1, H=13, W=100, Cin=192, Cout=128, act=L.ACT_ADD_RELU, residual=True, out_cs=384, out_co=256),
    dict(N=1, H=6, W=10, Cin=128, Cout=128, act=L.ACT_ADD_RELU),
])
def test_cin_bf16_kernel(case):
    name = _run_case(k=3, io=torch.bfloat16, compute=L.COMPUTE_BF16, **case)
    assert name == "cin_bf16"


@pytest.mark.parametrize("case", [
    dict(Cin=64, Cout=32, act=L.ACT_LRELU, act_param=0.2, in_cs=160, out_cs=160, out_co=64),       # RRDB dense conv 1
    dict(Cin=96, Cout=32, act=L.ACT_LRELU, act_param=0.2, in_cs=160, out_cs=160, out_co=96),       # half last chunk
    dict(Cin=160, Cout=64, residual=True, res_cs=160),                                             # compress conv + x
    dict(Cin=128, Cout=128, act=L.ACT_ADD_RELU, residual=True),                                    # two channel groups
    dict(Cin=32, Cout=64, act=L.ACT_PRELU, alpha_pc=1),                                            # a half chunk only
])
def test_cin_bf16_persistent_form_equals_the_per_tile_kernel(case):
    """With >= 4 tiles of 8 x 32 per CU the K-chunked kernel runs as ONE persistent workgroup per CU whose tiles and
    weight taps are moved by dedicated waves (conv_cinp_kernel).  Same arithmetic in the same order: the batch in one
    call must equal, bit for bit, the images run one call each (few tiles: the launch-per-tile kernel, which
    test_cin_bf16_kernel pins to the f64 restatement); image 0 is checked against that restatement here as well."""
    _cinp_check(9, 70, 417, case)                         # 9 x 14 ragged tiles per image, 1134 in the batch


def _cinp_fuzz_cases(n, seed):
    g = _rng(seed)
    out = []
    for _ in range(n):
        H, W = int(g.integers(17, 130)), int(g.integers(65, 520))
        tpi = -(-H // 8) * -(-W // 32)
        N = max(2, -(-1024 // tpi) + int(g.integers(0, 3)))          # >= 1024 tiles in the batch, < 1024 per image
        Cin, Cout = int(g.choice([32, 64, 96, 128, 160, 192])), int(g.choice([32, 64, 96]))
        if Cin == 64 and Cout == 64:
            Cout = 96                                                # (64 -> 64 is the Cin = 64 kernel's)
        c = dict(N=N, H=H, W=W, Cin=Cin, Cout=Cout,
                 act=int(g.choice([L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU, L.ACT_PRELU, L.ACT_ADD_RELU])), act_param=0.2,
                 alpha_pc=int(g.integers(0, 2)), residual=bool(g.integers(0, 2)))
        if g.integers(0, 2):
            c.update(in_cs=Cin + 8 * int(g.integers(0, 9)), out_cs=Cout + 8 * int(g.integers(1, 9)))
            c["out_co"] = 8 * int(g.integers(0, (c["out_cs"] - Cout) // 8 + 1))
            if c["residual"]:
                c["res_cs"] = Cout + 8 * int(g.integers(0, 5))
        out.append(c)
    return out


@pytest.mark.parametrize("case", _cinp_fuzz_cases(8, seed=23))
def test_cin_bf16_persistent_form_fuzz(case):
    """Seeded shapes, channel counts, slices and epilogues for the persistent form: batch call == per-image calls, bit for bit."""
    assert case["N"] * -(-case["H"] // 8) * -(-case["W"] // 32) >= 1024 > -(-case["H"] // 8) * -(-case["W"] // 32)
    _cinp_check(case["N"], case["H"], case["W"], case, check_ref=False)


def test_cin_bf16_persistent_form_soak_two_streams():
    """The persistent form's hand-offs (tile buffers, weight ring: barriers and counted vmcnt only, no state outside the
    workgroup) under repetition and under contention: 40 launches back to back, the second half of them while a second
    stream runs the same kernel on other tensors (the two grids then share the CUs workgroup by workgroup) - every
    output equal to the first, bit for bit."""
    N, H, W, Cin, Cout = 8, 128, 256, 160, 64
    g = _rng(5)
    desc = Fn.make_desc(N, H, W, Cin, Cout, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16, compute=L.COMPUTE_BF16,
                        act=L.ACT_LRELU, act_param=0.2)
    assert L.kernel_name(desc) == "cin_bf16"

    def make(seed):
        gg = _rng(seed)
        x = torch.from_numpy(gg.uniform(-1, 1, (N, H, W, Cin)).astype(np.float32)).bfloat16().to(DEV)
        w = torch.from_numpy((gg.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32))
        return x, Fn.pack_weights(desc, w.to(DEV))

    (xa, wa), (xb, wb) = make(1), make(2)
    scale = torch.from_numpy(g.uniform(0.5, 1.5, Cout).astype(np.float32)).to(DEV)
    shift = torch.from_numpy(g.uniform(-0.3, 0.3, Cout).astype(np.float32)).to(DEV)
    alpha = torch.zeros(1, device=DEV)
    outs = [torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device=DEV) for _ in range(40)]
    yb = torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device=DEV)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for it in range(40):
        if it >= 20:
            with torch.cuda.stream(side):
                Fn.conv2d_fused(desc, xb, wb, scale, shift, alpha, None, out=yb)
        Fn.conv2d_fused(desc, xa, wa, scale, shift, alpha, None, out=outs[it])
    torch.cuda.synchronize()
    bad = [it for it in range(1, 40) if not torch.equal(outs[it], outs[0])]
    assert not bad, bad[:10]
    ref_b = torch.empty_like(yb)
    Fn.conv2d_fused(desc, xb, wb, scale, shift, alpha, None, out=ref_b)
    torch.cuda.synchronize()
    assert torch.equal(yb, ref_b)


def _cinp_check(N, H, W, case, check_ref=True):
    Cin, Cout = case["Cin"], case["Cout"]
    act, act_param, alpha_pc = case.get("act", L.ACT_NONE), case.get("act_param", 0.0), case.get("alpha_pc", 0)
    in_cs, out_cs, out_co, res_cs = case.get("in_cs", 0), case.get("out_cs", 0), case.get("out_co", 0), case.get("res_cs", 0)
    g = _rng(41)
    x = torch.from_numpy(g.uniform(-1, 1, (N, H, W, in_cs or Cin)).astype(np.float32)).bfloat16().to(DEV)
    w = torch.from_numpy((g.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)).bfloat16().float()
    scale = torch.from_numpy(g.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(g.uniform(-0.3, 0.3, Cout).astype(np.float32))
    alpha = torch.from_numpy(g.uniform(0.05, 0.4, Cout if alpha_pc else 1).astype(np.float32))
    res = None
    if case.get("residual"):
        res = torch.from_numpy(g.uniform(-1, 1, (N, H, W, res_cs or Cout)).astype(np.float32)).bfloat16().to(DEV)

    def run(n, xs, rs):
        desc = Fn.make_desc(n, H, W, Cin, Cout, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16,
                            compute=L.COMPUTE_BF16, act=act, act_param=act_param, alpha_per_channel=alpha_pc,
                            in_cstride=in_cs, out_cstride=out_cs, out_coffset=out_co, res_cstride=res_cs)
        assert L.kernel_name(desc) == "cin_bf16"
        y = torch.full((n, H, W, out_cs or Cout), -5.0, dtype=torch.bfloat16, device=DEV)
        Fn.conv2d_fused(desc, xs, Fn.pack_weights(desc, w.to(DEV)), scale.to(DEV), shift.to(DEV), alpha.to(DEV), rs, out=y)
        return y

    whole = run(N, x, res)
    torch.cuda.synchronize()
    for i in range(N):
        one = run(1, x[i:i + 1].contiguous(), None if res is None else res[i:i + 1].contiguous())
        assert torch.equal(whole[i:i + 1], one), f"image {i}"
    if not check_ref:
        return
    ref = _ref_fused(x[0:1, ..., :Cin].float().cpu().permute(0, 3, 1, 2), w, scale, shift, act, act_param, alpha,
                     None if res is None else res[0:1, ..., :Cout].float().cpu().permute(0, 3, 1, 2), 1, 1, 1, False)
    got = whole[0:1, ..., out_co:out_co + Cout].float().cpu().permute(0, 3, 1, 2).double()
    assert torch.all((got - ref).abs() <= 2.0 ** -8 * (ref.abs() + 1.0))


@pytest.mark.parametrize("case", [
    dict(N=2, H=9, W=13, Cin=64, Cout=32, act=L.ACT_RELU),                     # BASELINE config 2's middle layer (9-1-5)
    dict(N=1, H=4, W=4, Cin=32, Cout=16, act=L.ACT_NONE),                      # one pixel tile exactly
    dict(N=3, H=31, W=67, Cin=128, Cout=64, act=L.ACT_LRELU, act_param=0.2),   # many tiles, the 2-tiles-in-flight variant
    dict(N=1, H=5, W=7, Cin=64, Cout=64, act=L.ACT_PRELU, alpha_pc=1),
    dict(N=1, H=8, W=9, Cin=32, Cout=32, act=L.ACT_PRELU, in_cs=48, out_cs=64, out_co=16),   # channel slices both sides
    dict(N=2, H=6, W=5, Cin=128, Cout=16, act=L.ACT_RELU),
    dict(N=2, H=7, W=9, Cin=64, Cout=64, act=L.ACT_NONE, residual=True),                 # 1x1 skip projection + add
    dict(N=1, H=7, W=9, Cin=32, Cout=32, act=L.ACT_RELU, residual=True, res_cs=48),
])
def test_pw_bf16_kernel(case):
    """1x1 conv as a streaming bf16 MFMA GEMM (SRCNN-style 1x1 middle layer): one output rounding."""
    assert _run_case(k=1, io=torch.bfloat16, compute=L.COMPUTE_BF16, **case) == "pw_bf16"


@pytest.mark.parametrize("case", [
    dict(N=2, H=11, W=30, Cin=32, k=5, act=L.ACT_NONE),                        # BASELINE config 2's last layer (9-1-5)
    dict(N=1, H=40, W=200, Cin=64, k=5, act=L.ACT_NONE),                       # SRCNN/srcnn.py:209-210; 2 strips, several bands
    dict(N=1, H=1, W=1, Cin=32, k=5, act=L.ACT_TANH),
    dict(N=2, H=9, W=193, Cin=32, k=5, act=L.ACT_SIGMOID),                     # one column into the second strip
    dict(N=1, H=17, W=29, Cin=32, k=3, act=L.ACT_RELU),                        # 14 outputs per block: scalar stores
    dict(N=1, H=8, W=230, Cin=64, k=5, act=L.ACT_NONE, in_cs=96),              # channel-slice input
    dict(N=1, H=12, W=36, Cin=32, k=5, act=L.ACT_NONE, out_bf16=True),
])
def test_c1out_bf16_kernel(case):
    """k x k conv to ONE channel (banded MFMA operand, rows streamed once): bf16 in, f32 (or bf16) out, f32 sums of
    exact bf16 products - summation-order error only."""
    g = _rng(13)
    N, H, W, Cin, k = case["N"], case["H"], case["W"], case["Cin"], case["k"]
    act, in_cs = case["act"], case.get("in_cs", 0)
    obf = case.get("out_bf16", False)
    x = torch.from_numpy(g.uniform(-1, 1, (N, Cin, H, W)).astype(np.float32)).bfloat16().float()
    w = torch.from_numpy((g.standard_normal((1, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)).bfloat16().float()
    scale = torch.from_numpy(g.uniform(0.5, 1.5, 1).astype(np.float32))
    shift = torch.from_numpy(g.uniform(-0.3, 0.3, 1).astype(np.float32))
    ref = _ref_fused(x, w, scale, shift, act, 0.0, None, None, 1, k // 2, 1, False)
    desc = Fn.make_desc(N, H, W, Cin, 1, k, in_dtype=L.BF16, out_dtype=L.BF16 if obf else L.F32, out_layout=L.NCHW,
                        compute=L.COMPUTE_BF16, act=act, in_cstride=in_cs)
    # (3x3 from 64 channels is the older c64_thin kernel's shape: it keeps it)
    assert L.kernel_name(desc) == ("c64_thin_bf16" if (k, Cin) == (3, 64) else "c1out_bf16")
    xd = torch.full((N, H, W, in_cs or Cin), 7.0, dtype=torch.bfloat16)
    xd[..., :Cin] = x.permute(0, 2, 3, 1).bfloat16()
    wp = Fn.pack_weights(desc, w.to(DEV))
    y = Fn.conv2d_fused(desc, xd.to(DEV), wp, scale.to(DEV), shift.to(DEV))
    if (k, Cin) != (3, 64):
        import ctypes as C
        y2 = torch.empty_like(y)
        a, dv = L.ConvArgs(), [t.to(DEV) for t in (xd, scale, shift)]
        Fn.fill_args(a, desc, dv[0], wp, dv[1], dv[2], None, None, y2)
        L.check(L.load().srmi_srcnn_layer_c1out(C.byref(a), Fn.stream_ptr()))  # the by-name entry point: same launch
        assert torch.equal(y, y2)
    torch.cuda.synchronize()
    err = (y.float().cpu().double() - ref).abs()
    tol = 2.0 ** -8 * (ref.abs() + 1.0) if obf else torch.full_like(ref, 2e-5)
    assert torch.all(err <= tol), err.max().item()


@pytest.mark.parametrize("case", [
    dict(N=2, H=10, W=12, Cout=3, act=L.ACT_TANH, out_layout=L.NCHW),          # last_conv (ragged tile)
    dict(N=1, H=6, W=64, Cout=3, act=L.ACT_TANH, out_layout=L.NCHW),           # exactly one tile
    dict(N=1, H=13, W=130, Cout=3, act=L.ACT_TANH, out_layout=L.NCHW),         # several tiles, ragged both ways
    dict(N=1, H=7, W=20, Cout=1, act=L.ACT_NONE),                              # single channel, NHWC out
    dict(N=1, H=7, W=20, Cout=4, act=L.ACT_SIGMOID, in_cs=160),                # channel-slice input
])
def test_c64_thin_kernel(case):
    """64 -> <=4 channel conv (the HBM-bound last_conv): bf16 in, f32 out."""
    g = _rng(3)
    N, H, W, Cout = case["N"], case["H"], case["W"], case["Cout"]
    act, in_cs = case["act"], case.get("in_cs", 0)
    out_layout = case.get("out_layout", L.NHWC)
    x = torch.from_numpy(g.uniform(-1, 1, (N, 64, H, W)).astype(np.float32)).bfloat16().float()
    w = torch.from_numpy((g.standard_normal((Cout, 64, 3, 3)) / 24.0).astype(np.float32)).bfloat16().float()
    scale = torch.from_numpy(g.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(g.uniform(-0.3, 0.3, Cout).astype(np.float32))
    ref = _ref_fused(x, w, scale, shift, act, 0.0, None, None, 1, 1, 1, False)
    desc = Fn.make_desc(N, H, W, 64, Cout, 3, in_dtype=L.BF16, out_dtype=L.F32, out_layout=out_layout,
                        compute=L.COMPUTE_BF16, act=act, in_cstride=in_cs)
    assert L.kernel_name(desc) == "c64_thin_bf16"
    xd = torch.full((N, H, W, in_cs or 64), 7.0, dtype=torch.bfloat16)
    xd[..., :64] = x.permute(0, 2, 3, 1).bfloat16()
    wp = Fn.pack_weights(desc, w.to(DEV))
    y = Fn.conv2d_fused(desc, xd.to(DEV), wp, scale.to(DEV), shift.to(DEV))
    torch.cuda.synchronize()
    got = y.cpu() if out_layout == L.NCHW else y.cpu().permute(0, 3, 1, 2)
    err = (got.double() - ref).abs().max().item()
    assert err <= 2e-5, err          # f32 accumulate of exact bf16 products: summation-order error only


@pytest.mark.parametrize("case", [
    dict(N=2, H=9, W=7, in_layout=L.NCHW),                  # tiny image, smaller than the 9x9 window reach
    dict(N=1, H=13, W=100, in_layout=L.NCHW),               # big tile, ragged
    dict(N=2, H=6, W=96, in_layout=L.NCHW),                 # exactly one big tile per image
    dict(N=1, H=20, W=33, in_layout=L.NHWC),                # NHWC f32 input
    dict(N=1, H=8, W=40, in_layout=L.NHWC, Cin=1, alpha_pc=1),          # single-channel input (Y only), per-channel slopes
    dict(N=1, H=10, W=70, in_layout=L.NCHW, Cin=1, Cout=128, alpha_pc=1),   # SRCNN conv1 (SRCNN/srcnn.py:203): two 64-ch launches
    # 3x3 taps on the same kernel (KS = 3): the RRDB generator's head conv (ESRGAN/untitled0.py:41), D / VGG19 conv 1
    dict(N=2, H=9, W=7, in_layout=L.NCHW, k=3),
    dict(N=1, H=13, W=100, in_layout=L.NCHW, k=3),
    dict(N=1, H=20, W=33, in_layout=L.NHWC, k=3, Cin=2),
    dict(N=1, H=8, W=40, in_layout=L.NCHW, k=3, Cin=1, Cout=128, alpha_pc=1),
])
def test_k9c3_kernel(case):
    """conv01: 9x9 (or 3x3), Cin<=3 -> 64, PReLU, f32 NCHW/NHWC in -> NHWC bf16 out."""
    g = _rng(4)
    N, H, W, Cin = case["N"], case["H"], case["W"], case.get("Cin", 3)
    Cout, apc, k = case.get("Cout", 64), case.get("alpha_pc", 0), case.get("k", 9)
    x = torch.from_numpy(g.uniform(-1, 1, (N, Cin, H, W)).astype(np.float32))
    w = torch.from_numpy((g.standard_normal((Cout, Cin, k, k)) / np.sqrt(float(k * k) * Cin)).astype(np.float32))
    scale = torch.from_numpy(g.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(g.uniform(-0.3, 0.3, Cout).astype(np.float32))
    alpha = torch.from_numpy(g.uniform(0.05, 0.4, Cout).astype(np.float32)) if apc else torch.tensor([0.25])
    # the kernel rounds x and w to bf16 before the MFMA: so does the reference
    ref = _ref_fused(x.bfloat16().float(), w.bfloat16().float(), scale, shift, L.ACT_PRELU, 0.0, alpha, None, 1, k // 2, 1, False)
    desc = Fn.make_desc(N, H, W, Cin, Cout, k, in_dtype=L.F32, out_dtype=L.BF16, in_layout=case["in_layout"],
                        compute=L.COMPUTE_BF16, act=L.ACT_PRELU, alpha_per_channel=apc)
    # (one input channel with a scalar slope and 9x9 taps is the dense-tap kernel's shape)
    assert L.kernel_name(desc) == ("k9c1_bf16" if Cin == 1 and not apc and k == 9 else "k9c3_bf16")
    xd = (x if case["in_layout"] == L.NCHW else x.permute(0, 2, 3, 1).contiguous()).to(DEV)
    wp = Fn.pack_weights(desc, w.to(DEV))
    y = Fn.conv2d_fused(desc, xd, wp, scale.to(DEV), shift.to(DEV), alpha.to(DEV))
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got.double() - ref).abs()
    assert torch.all(err <= 2.0 ** -8 * (ref.abs() + 1.0)), err.max().item()


@pytest.mark.parametrize("case", [
    dict(N=2, H=9, W=7, Cout=64, act=L.ACT_RELU),                          # smaller than the window reach
    dict(N=1, H=16, W=64, Cout=64, act=L.ACT_RELU),                        # exactly one tile
    dict(N=1, H=37, W=150, Cout=128, act=L.ACT_RELU),                      # SRCNN/srcnn.py:205: two channel groups, ragged
    dict(N=2, H=17, W=65, Cout=64, act=L.ACT_PRELU, in_layout=L.NHWC),     # scalar PReLU, NHWC spelling of the same plane
    dict(N=1, H=5, W=130, Cout=64, act=L.ACT_NONE, out_cs=96, out_co=32),  # store into a channel slice
    dict(N=1, H=20, W=20, Cout=192, act=L.ACT_LRELU),
])
def test_k9c1_kernel(case):
    """9x9 conv of a one-channel image (SRCNN's first layer): 81 taps packed densely on the MFMA K axis, f32 in ->
    NHWC bf16 out; x and w are rounded to bf16 before the MFMA, so is the reference."""
    g = _rng(14)
    N, H, W, Cout, act = case["N"], case["H"], case["W"], case["Cout"], case["act"]
    out_cs, out_co = case.get("out_cs", 0), case.get("out_co", 0)
    x = torch.from_numpy(g.uniform(-1, 1, (N, 1, H, W)).astype(np.float32))
    w = torch.from_numpy((g.standard_normal((Cout, 1, 9, 9)) / 9.0).astype(np.float32))
    scale = torch.from_numpy(g.uniform(0.5, 1.5, Cout).astype(np.float32))
    shift = torch.from_numpy(g.uniform(-0.3, 0.3, Cout).astype(np.float32))
    alpha = torch.tensor([0.25])
    ref = _ref_fused(x.bfloat16().float(), w.bfloat16().float(), scale, shift, act, 0.1, alpha, None, 1, 4, 1, False)
    desc = Fn.make_desc(N, H, W, 1, Cout, 9, in_dtype=L.F32, out_dtype=L.BF16, in_layout=case.get("in_layout", L.NCHW),
                        compute=L.COMPUTE_BF16, act=act, act_param=0.1, out_cstride=out_cs, out_coffset=out_co)
    assert L.kernel_name(desc) == "k9c1_bf16"
    wp = Fn.pack_weights(desc, w.to(DEV))
    yd = torch.full((N, H, W, out_cs or Cout), -5.0, dtype=torch.bfloat16, device=DEV)
    import ctypes as C
    a, dv = L.ConvArgs(), [t.to(DEV) for t in (x, scale, shift, alpha)]
    Fn.fill_args(a, desc, dv[0], wp, dv[1], dv[2], dv[3], None, yd)
    L.check(L.load().srmi_srcnn_layer_c1in(C.byref(a), Fn.stream_ptr()))       # the by-name entry point
    assert L.load().srmi_srcnn_layer_c1out(C.byref(a), Fn.stream_ptr()) == -2   # ... which refuses other shapes
    torch.cuda.synchronize()
    y = yd.float().cpu()
    got = y[..., out_co:out_co + Cout].permute(0, 3, 1, 2)
    err = (got.double() - ref).abs()
    assert torch.all(err <= 2.0 ** -8 * (ref.abs() + 1.0)), err.max().item()
    if out_cs:
        mask = torch.ones(out_cs, dtype=torch.bool)
        mask[out_co:out_co + Cout] = False
        assert torch.all(y[..., mask] == -5.0)


@pytest.mark.parametrize("case", [
    dict(N=2, H=11, W=13),             # small tile config, ragged
    dict(N=1, H=4, W=32),              # exactly one small tile
    dict(N=1, H=13, W=100),            # big 6x96 tiles, ragged both ways
    dict(N=2, H=6, W=96),              # exactly one big tile per image
    dict(N=1, H=24, W=192),
    dict(N=1, H=13, W=100, other_res=True),     # the residual is NOT the block input: read from memory, not from the LDS tile
    dict(N=2, H=11, W=13, other_res=True),
    dict(N=1, H=13, W=100, alpha=-0.75),        # conv1 slope outside [0, 1]: compare + select form of the T write-back
    dict(N=1, H=13, W=100, alpha=1.0),
])
def test_fused_resblock_kernel(case):
    """ResBlock in one launch == conv-affine-PReLU -> (bf16 T) -> conv-affine -> +x, the intermediate rounded to
    bf16 exactly like the two-launch path; compared with the CPU restatement AND bit-for-bit with the unfused
    kernels (same MFMA order per output element)."""
    import ctypes as C
    g = _rng(6)
    N, H, W = case["N"], case["H"], case["W"]
    x = torch.from_numpy(g.uniform(-1, 1, (N, 64, H, W)).astype(np.float32)).bfloat16().float()
    ws, scs, shs = [], [], []
    for _ in range(2):
        ws.append(torch.from_numpy((g.standard_normal((64, 64, 3, 3)) / 24.0).astype(np.float32)).bfloat16().float())
        scs.append(torch.from_numpy(g.uniform(0.5, 1.5, 64).astype(np.float32)))
        shs.append(torch.from_numpy(g.uniform(-0.3, 0.3, 64).astype(np.float32)))
    alpha = torch.tensor([case.get("alpha", 0.25)])
    res = x
    if case.get("other_res"):
        res = torch.from_numpy(g.uniform(-1, 1, (N, 64, H, W)).astype(np.float32)).bfloat16().float()
    t_ref = _ref_fused(x, ws[0], scs[0], shs[0], L.ACT_PRELU, 0.0, alpha, None, 1, 1, 1, False).float().bfloat16().float()
    ref = _ref_fused(t_ref, ws[1], scs[1], shs[1], L.ACT_NONE, 0.0, None, res, 1, 1, 1, False)

    def desc(act):
        return Fn.make_desc(N, H, W, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, res_dtype=L.BF16,
                            compute=L.COMPUTE_BF16, act=act)
    d1, d2 = desc(L.ACT_PRELU), desc(L.ACT_NONE)
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    rd = xd if res is x else res.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    wp = [Fn.pack_weights(d1, ws[0].to(DEV)), Fn.pack_weights(d2, ws[1].to(DEV))]
    scd, shd, ad = [t.to(DEV) for t in scs], [t.to(DEV) for t in shs], alpha.to(DEV)
    tmp = torch.full((N, H, W, 64), 3.0, dtype=torch.bfloat16, device=DEV)
    y = torch.full((N, H, W, 64), -5.0, dtype=torch.bfloat16, device=DEV)
    a1, a2 = L.ConvArgs(), L.ConvArgs()
    Fn.fill_args(a1, d1, xd, wp[0], scd[0], shd[0], ad, None, tmp)
    Fn.fill_args(a2, d2, tmp, wp[1], scd[1], shd[1], None, rd, y)
    a1.flags = L.CONV_FUSE_NEXT
    assert L.load().srmi_conv2d_can_fuse(C.byref(a1), C.byref(a2)) == 1
    L.check(L.load().srmi_resblock_fwd(C.byref(a1), C.byref(a2), Fn.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.all(tmp == 3.0)                      # the intermediate is never materialised
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got.double() - ref).abs()
    assert torch.all(err <= 2.0 ** -7 * (ref.abs() + 1.0)), err.max().item()
    # unfused pair through the same ABI
    y2 = torch.empty_like(y)
    Fn.conv2d_fused(d1, xd, wp[0], scd[0], shd[0], ad, None, out=tmp)
    Fn.conv2d_fused(d2, tmp, wp[1], scd[1], shd[1], None, rd, out=y2)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    # aliasing the input is refused before any launch
    Fn.fill_args(a2, d2, tmp, wp[1], scd[1], shd[1], None, xd, xd)
    assert L.load().srmi_resblock_fwd(C.byref(a1), C.byref(a2), Fn.stream_ptr()) == -2


@pytest.mark.parametrize("case", [
    dict(N=1, H=24, W=24),                        # exactly one tile
    dict(N=1, H=48, W=72),                        # whole tiles only
    dict(N=2, H=5, W=7),                          # smaller than a tile
    dict(N=1, H=25, W=50),                        # ragged both ways, rings of neighbouring tiles overlap
    dict(N=2, H=9, W=97, act2=L.ACT_NONE),        # no tanh: the raw fixed-point sums
    dict(N=1, H=24, W=96, alpha=-0.75),           # slope outside [0, 1]: compare + select form
    dict(N=1, H=16, W=49, alpha=1.0),
    dict(N=1, H=1, W=1),
    dict(N=3, H=2, W=200),
    dict(N=1, H=25, W=30, dcr=True, alpha_pc=True),          # Keras depth_to_space order + per-channel PReLU (RRDBNet tail)
    dict(N=2, H=9, W=50, dcr=True),
    dict(N=1, H=12, W=26, alpha_pc=True),
])
def test_fused_up_tail_kernel(case):
    """conv3x3 64->256 -> PixelShuffle(2) -> PReLU -> conv3x3 64->3 -> tanh in one launch (srmi_conv2d_fwd_pair,
    SRMI_FUSE_UPTAIL) against the CPU restatement and the two-launch path.  The intermediate T is the same bf16 tensor in
    both paths (same MFMA order); the last conv differs by its int32 2^-18 fixed-point accumulation (nine adds:
    <= 9 * 2^-19 = 1.7e-5 before the activation) and f32 summation order.  Two fused runs are bit-identical."""
    import ctypes as C
    lib = L.load()
    g = _rng(11)
    N, H, W = case["N"], case["H"], case["W"]
    act2 = case.get("act2", L.ACT_TANH)
    x = torch.from_numpy(g.uniform(-1, 1, (N, 64, H, W)).astype(np.float32)).bfloat16().float()
    w1 = torch.from_numpy((g.standard_normal((256, 64, 3, 3)) / 24.0).astype(np.float32)).bfloat16().float()
    w2 = torch.from_numpy((g.standard_normal((3, 64, 3, 3)) / 24.0).astype(np.float32)).bfloat16().float()
    sc1 = torch.ones(256); sh1 = torch.from_numpy(g.uniform(-0.3, 0.3, 256).astype(np.float32))
    sc2 = torch.ones(3); sh2 = torch.from_numpy(g.uniform(-0.3, 0.3, 3).astype(np.float32))
    dcr, apc = case.get("dcr", False), case.get("alpha_pc", False)
    alpha = torch.tensor([case.get("alpha", 0.25)])
    if apc:        # one slope per conv channel (a post-shuffle PReLU(64) arrives expanded like this, engine._Seq.add)
        alpha = torch.from_numpy(g.uniform(-0.2, 0.6, 256).astype(np.float32))
    t_ref = _ref_fused(x, w1, sc1, sh1, L.ACT_PRELU, 0.0, alpha, None, 1, 1, 2, dcr).float().bfloat16().float()
    ref = _ref_fused(t_ref, w2, sc2, sh2, act2, 0.0, None, None, 1, 1, 1, False)

    d1 = Fn.make_desc(N, H, W, 64, 256, 3, in_dtype=L.BF16, out_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_PRELU, shuffle=2,
                      shuffle_order=L.SHUFFLE_DCR if dcr else L.SHUFFLE_CRD, alpha_per_channel=int(apc))
    d2 = Fn.make_desc(N, 2 * H, 2 * W, 64, 3, 3, in_dtype=L.BF16, out_dtype=L.F32, out_layout=L.NCHW, compute=L.COMPUTE_BF16, act=act2)
    assert L.kernel_name(d1) == "c64_bf16" and L.kernel_name(d2) == "c64_thin_bf16"
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    wp = [Fn.pack_weights(d1, w1.to(DEV)), Fn.pack_weights(d2, w2.to(DEV))]
    dv = [t.to(DEV) for t in (sc1, sh1, sc2, sh2, alpha)]
    tmp = torch.full((N, 2 * H, 2 * W, 64), 3.0, dtype=torch.bfloat16, device=DEV)
    ys = [torch.full((N, 3, 2 * H, 2 * W), -5.0, dtype=torch.float32, device=DEV) for _ in range(2)]
    a1, a2 = L.ConvArgs(), L.ConvArgs()
    for y in ys:
        Fn.fill_args(a1, d1, xd, wp[0], dv[0], dv[1], dv[4], None, tmp)
        Fn.fill_args(a2, d2, tmp, wp[1], dv[2], dv[3], None, None, y)
        a1.flags = L.CONV_FUSE_NEXT
        assert lib.srmi_conv2d_can_fuse(C.byref(a1), C.byref(a2)) == 2
        L.check(lib.srmi_conv2d_fwd_pair(C.byref(a1), C.byref(a2), Fn.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.all(tmp == 3.0)                      # the intermediate is never materialised
    assert torch.equal(ys[0], ys[1])
    got = ys[0].cpu()
    err = (got.double() - ref).abs()
    # vs the CPU restatement: a T element may round to the other bf16 neighbour (f32 vs f64 accumulation)
    assert torch.all(err <= 2e-3), err.max().item()
    # vs the two launches through the same ABI: same T bits, so only the last conv's accumulation differs
    y2 = torch.empty_like(ys[0])
    Fn.conv2d_fused(d1, xd, wp[0], dv[0], dv[1], dv[4], None, out=tmp)
    Fn.conv2d_fused(d2, tmp, wp[1], dv[2], dv[3], None, None, out=y2)
    torch.cuda.synchronize()
    d = (ys[0] - y2).abs().max().item()
    assert d <= 4e-5, d
    # a residual on the last conv, or an output that aliases the input, is refused before any launch
    Fn.fill_args(a2, d2, tmp, wp[1], dv[2], dv[3], None, ys[1], ys[0])
    assert lib.srmi_conv2d_can_fuse(C.byref(a1), C.byref(a2)) == 0
    assert lib.srmi_conv2d_fwd_pair(C.byref(a1), C.byref(a2), Fn.stream_ptr()) == -2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pool_and_upsample_addressing_ops(dtype):
    """srmi_maxpool2x2_nhwc (channel-slice input, floor sizes) and srmi_upsample2x_nhwc (store into a channel slice)."""
    lib = L.load()
    code = L.F32 if dtype == torch.float32 else L.BF16
    g = torch.Generator().manual_seed(17)
    N, H, W, Cc, CS, off = 2, 7, 9, 24, 40, 8
    buf = torch.randn(N, H, W, CS, generator=g).to(dtype).to(DEV)
    y = torch.empty(N, H // 2, W // 2, Cc, dtype=dtype, device=DEV)
    L.check(lib.srmi_maxpool2x2_nhwc(buf[..., off:].data_ptr(), y.data_ptr(), N, H, W, Cc, CS, code, Fn.stream_ptr()))
    ref = torch.nn.functional.max_pool2d(buf[..., off:off + Cc].float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.equal(y.float(), ref)
    x = torch.randn(N, 3, 5, Cc, generator=g).to(dtype).to(DEV)
    out = torch.full((N, 6, 10, CS), 7.0, dtype=dtype, device=DEV)
    L.check(lib.srmi_upsample2x_nhwc(x.data_ptr(), out.data_ptr(), N, 3, 5, Cc, CS, off, code, Fn.stream_ptr()))
    up = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    assert torch.equal(out[..., off:off + Cc], up)
    assert torch.all(out[..., :off] == 7.0) and torch.all(out[..., off + Cc:] == 7.0)
    assert lib.srmi_upsample2x_nhwc(x.data_ptr(), out.data_ptr(), N, 3, 5, Cc, CS, CS - 4, code, None) == -1     # slice overruns
    assert lib.srmi_maxpool2x2_nhwc(None, None, 0, 4, 4, 8, 0, code, None) == 0                                   # empty batch


@pytest.mark.parametrize("case", [
    dict(N=2, H=9, W=7, act1=L.ACT_RELU, act2=L.ACT_RELU),                     # smaller than the window reach
    dict(N=1, H=16, W=64, act1=L.ACT_RELU, act2=L.ACT_RELU),                   # exactly one tile
    dict(N=2, H=37, W=150, act1=L.ACT_RELU, act2=L.ACT_RELU),                  # ragged in both directions
    dict(N=1, H=21, W=67, act1=L.ACT_PRELU, act2=L.ACT_PRELU, apc2=1),         # scalar slope, then per-channel slopes
    dict(N=1, H=8, W=130, act1=L.ACT_LRELU, act2=L.ACT_NONE, out_cs=56, out_co=16),   # the 32 channels into a channel slice
])
def test_k9c1_pw_fused_pair(case):
    """SRCNN 9-1-5 (BASELINE config 2): conv 9x9 1 -> 64 + activation and conv 1x1 64 -> 32 + activation as ONE launch
    ("k9c1pw_bf16", SRMI_FUSE_K9PW): the 64-channel tensor stays in registers.  Bit-identical to the two launches."""
    import ctypes as C
    g = _rng(41)
    N, H, W = case["N"], case["H"], case["W"]
    act1, act2, apc2 = case["act1"], case["act2"], case.get("apc2", 0)
    out_cs, out_co = case.get("out_cs", 0), case.get("out_co", 0)
    x = torch.from_numpy(g.uniform(-1, 1, (N, 1, H, W)).astype(np.float32)).to(DEV)
    w1 = torch.from_numpy((g.standard_normal((64, 1, 9, 9)) / 9.0).astype(np.float32)).to(DEV)
    w2 = torch.from_numpy((g.standard_normal((32, 64, 1, 1)) / 8.0).astype(np.float32)).to(DEV)
    par = lambda n: (torch.from_numpy(g.uniform(0.5, 1.5, n).astype(np.float32)).to(DEV),
                     torch.from_numpy(g.uniform(-0.3, 0.3, n).astype(np.float32)).to(DEV))
    (sc1, sh1), (sc2, sh2) = par(64), par(32)
    al1 = torch.tensor([0.2], device=DEV)
    al2 = torch.from_numpy(g.uniform(0.05, 0.4, 32 if apc2 else 1).astype(np.float32)).to(DEV)
    d1 = Fn.make_desc(N, H, W, 1, 64, 9, in_dtype=L.F32, out_dtype=L.BF16, in_layout=L.NCHW, compute=L.COMPUTE_BF16,
                      act=act1, act_param=0.1)
    d2 = Fn.make_desc(N, H, W, 64, 32, 1, in_dtype=L.BF16, out_dtype=L.BF16, compute=L.COMPUTE_BF16, act=act2, act_param=0.3,
                      alpha_per_channel=apc2, out_cstride=out_cs, out_coffset=out_co)
    assert L.kernel_name(d1) == "k9c1_bf16" and L.kernel_name(d2) == "pw_bf16"
    wp1, wp2 = Fn.pack_weights(d1, w1), Fn.pack_weights(d2, w2)
    mid = torch.zeros((N, H, W, 64), dtype=torch.bfloat16, device=DEV)
    outs = []
    for fused in (False, True):
        y = torch.full((N, H, W, out_cs or 32), 3.0, dtype=torch.bfloat16, device=DEV)
        arr = (L.ConvArgs * 2)()
        Fn.fill_args(arr[0], d1, x, wp1, sc1, sh1, al1 if act1 == L.ACT_PRELU else None, None, mid)
        Fn.fill_args(arr[1], d2, mid, wp2, sc2, sh2, al2 if act2 == L.ACT_PRELU else None, None, y)
        arr[0].flags = L.CONV_FUSE_NEXT if fused else 0
        lib = L.load()
        assert lib.srmi_conv2d_can_fuse(C.byref(arr[0]), C.byref(arr[1])) == 3
        if fused:
            mid.fill_(-7.0)                 # the fused launch must neither read nor need it
        L.check(lib.srmi_conv2d_fwd_seq(arr, 2, Fn.stream_ptr()))
        torch.cuda.synchronize()
        if fused:
            assert float(mid.float().abs().min()) == 7.0 and float(mid.float().max()) == -7.0
        outs.append(y.clone())
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    if out_cs:                               # channels outside the slice untouched
        assert float(outs[1][..., :out_co].float().min()) == 3.0 and float(outs[1][..., out_co + 32:].float().max()) == 3.0
    # and the pair against the f64 restatement (bf16 inputs / weights / intermediate, one output rounding)
    t = _ref_fused(x.cpu().bfloat16().float(), w1.cpu().bfloat16().float(), sc1.cpu(), sh1.cpu(), act1, 0.1, al1.cpu(), None, 1, 4, 1, False)
    ref = _ref_fused(t.float().bfloat16().float(), w2.cpu().bfloat16().float(), sc2.cpu(), sh2.cpu(), act2, 0.3, al2.cpu(), None, 1, 0, 1, False)
    got = outs[1][..., out_co:out_co + 32].float().cpu().permute(0, 3, 1, 2)
    err = (got.double() - ref).abs()
    assert torch.all(err <= 2.0 ** -7 * (ref.abs() + 1.0)), err.max().item()
