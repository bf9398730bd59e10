"""GPU: whole-network parity of the drop-in `Generator` / `Discriminator` against the golden vectors
the REFERENCE produced (tests/golden, made by oracle/make_golden.py), and size-independent properties
at the BASELINE configuration.

Bars (BASELINE.json north_star): fp32 path max-abs <= 1e-4 on [0,1] pixels; bf16 path PSNR within
0.05 dB of the fp32 reference (operating point: the reference's own logged quality, 26.48 dB,
P/metrics.csv:5240)."""
import os

import numpy as np
import pytest
import torch

from oracle import srgan_oracle as O
from oracle import weights as W
from srmi import ops
from srmi.srgan_model import Discriminator, Generator

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FULL = ["gen_b1_1x3x12x16.npz", "gen_b2_2x3x8x8.npz", "gen_b16_1x3x16x16.npz",
        "gen_b1_scale2_1x3x9x7.npz", "gen_b1_scale3_1x3x6x5.npz"]
PIX_TOL = 1e-4      # on [0,1] pixels  ==  2e-4 on the tanh range [-1,1]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _make_gen(g, compute):
    nb, scale = int(g["num_block"]), int(g["scale"])
    gen = Generator(num_block=nb, scale=scale)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=int(g["wseed"]), alpha=float(g["alpha"]),
                                                       num_block=nb, scale=scale)), strict=True)
    return gen.to(DEV).eval().set_compute(compute)


def _psnr01(a, b):
    """PSNR on [0,1] pixels of two tanh-range images."""
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()) / 4.0
    return 10.0 * np.log10(1.0 / mse)


@pytest.fixture(autouse=True)
def _restore_shared_prelu():
    yield
    with torch.no_grad():
        Generator.__init__.__defaults__[4].weight.fill_(0.25)


@pytest.mark.parametrize("name", FULL)
def test_generator_fp32_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    gen = _make_gen(g, "fp32")
    with torch.no_grad():
        sr, feat = gen(torch.from_numpy(g["x"]).to(DEV))
    torch.cuda.synchronize()
    kernels = set(gen._plans.get(gen, *g["x"].shape[:1], *g["x"].shape[2:], 0).kernels)
    exact = os.environ.get("SRMI_F32_EXACT", "0") not in ("", "0")
    # conv01 reads NCHW (no-LDS kernel); the 64-channel convs: split-bf16 (halo-tile form where the epilogue is linear) or,
    # under SRMI_F32_EXACT=1, the exact f32 MFMA; last_conv (Cout = 3) always the exact f32 MFMA
    assert kernels <= {"generic_f32", "f32t", "b3_f32", "b3h_f32"} and "generic_f32" in kernels, kernels
    assert ("b3_f32" in kernels or "b3h_f32" in kernels) != exact, kernels
    assert sr.shape == g["sr"].shape and feat.shape == g["feat"].shape
    e_sr = np.abs(sr.cpu().numpy() - g["sr"]).max() / 2.0
    e_feat = (np.abs(feat.cpu().numpy() - g["feat"]) / np.maximum(1.0, np.abs(g["feat"]))).max()
    assert e_sr <= PIX_TOL, e_sr
    assert e_feat <= 1e-4, e_feat


@pytest.mark.parametrize("name", FULL[:3])
def test_generator_bf16_psnr(golden_dir, name):
    g = _load(golden_dir, name)
    gen = _make_gen(g, "bf16")
    with torch.no_grad():
        sr, feat = gen(torch.from_numpy(g["x"]).to(DEV))
    plan = gen._plans.get(gen, g["x"].shape[0], g["x"].shape[2], g["x"].shape[3], 1)
    assert plan.kernels.count("c64_bf16") == 2 * int(g["num_block"]) + 3          # body + conv02 + 2 upsamplers
    labels = [l for l, _, _ in plan.launches]
    assert labels.count("resblock_bf16") == int(g["num_block"])                  # every ResBlock is ONE launch
    # the second Upsampler and last_conv go out as ONE launch (the 4x-resolution tensor between them never reaches HBM)
    assert labels[0] == "k9c3_bf16" and labels[-1] == "upfuse_bf16" and len(labels) == int(g["num_block"]) + 4
    sr = sr.cpu().numpy()
    # synthetic ground truth at the reference's logged operating point (26.48 dB)
    rng = np.random.Generator(np.random.PCG64(99))
    sigma = 2.0 * 10.0 ** (-26.48 / 20.0)
    gt = g["sr"] + rng.standard_normal(g["sr"].shape) * sigma
    p_ref, p_bf = _psnr01(g["sr"], gt), _psnr01(sr, gt)
    direct = _psnr01(sr, g["sr"])
    print(f"{name}: PSNR(ref,gt)={p_ref:.3f} PSNR(bf16,gt)={p_bf:.3f} PSNR(bf16,ref)={direct:.2f} dB")
    assert abs(p_ref - p_bf) <= 0.05
    assert direct >= 40.0


def test_block_modules_standalone(golden_dir):
    """conv / ResBlock / Upsampler called on their own (NCHW in, NCHW out) against the reference taps."""
    g = _load(golden_dir, "gen_b1_1x3x12x16.npz")
    gen = _make_gen(g, "fp32")
    x = torch.from_numpy(g["x"]).to(DEV)
    with torch.no_grad():
        c1 = gen.conv01(x)
        b0 = gen.body[0](c1)
        feat = gen.conv02(b0, _residual=c1)
        u0 = gen.tail[0](feat)
    assert np.abs(c1.cpu().numpy() - g["tap_conv01"]).max() <= 1e-4
    assert np.abs(b0.cpu().numpy() - g["tap_block0"]).max() <= 1e-4
    assert np.abs(feat.cpu().numpy() - g["feat"]).max() <= 1e-4
    assert np.abs(u0.cpu().numpy() - g["tap_up0"]).max() <= 1e-4


def test_cfg1_fp32_vs_reference_samples(golden_dir):
    """BASELINE config 1 (16x3x96x96): strided samples and fp64 moments of the reference's output."""
    g = _load(golden_dir, "gen_b16_cfg1_16x3x96x96.npz")
    gen = _make_gen(g, "fp32")
    x = torch.from_numpy(W.lr_input((16, 3, 96, 96), seed=int(g["xseed"]))).to(DEV)
    with torch.no_grad():
        sr, feat = gen(x)
    st, fs = int(g["stride"]), int(g["feat_stride"])
    assert np.abs(sr[:, :, ::st, ::st].cpu().numpy() - g["sr_sample"]).max() / 2.0 <= PIX_TOL
    fe = np.abs(feat[:, :, ::fs, ::fs].cpu().numpy() - g["feat_sample"]) / np.maximum(1.0, np.abs(g["feat_sample"]))
    assert fe.max() <= 1e-4
    n = sr.numel()
    assert abs(sr.double().sum().item() - float(g["sr_sum"])) / n <= 1e-6
    assert abs((sr.double() ** 2).sum().item() - float(g["sr_sumsq"])) / n <= 1e-6
    assert abs(sr.abs().max().item() - float(g["sr_absmax"])) <= 2e-4


def test_cfg1_bf16_vs_reference_samples(golden_dir):
    g = _load(golden_dir, "gen_b16_cfg1_16x3x96x96.npz")
    gen = _make_gen(g, "bf16")
    x = torch.from_numpy(W.lr_input((16, 3, 96, 96), seed=int(g["xseed"]))).to(DEV)
    with torch.no_grad():
        sr, _ = gen(x)
    st = int(g["stride"])
    sample = sr[:, :, ::st, ::st].cpu().numpy()
    rng = np.random.Generator(np.random.PCG64(100))
    gt = g["sr_sample"] + rng.standard_normal(sample.shape) * 2.0 * 10.0 ** (-26.48 / 20.0)
    direct = _psnr01(sample, g["sr_sample"])
    print(f"cfg1 bf16: PSNR(bf16,ref)={direct:.2f} dB; dPSNR={_psnr01(sample, gt) - _psnr01(g['sr_sample'], gt):+.4f}")
    assert abs(_psnr01(sample, gt) - _psnr01(g["sr_sample"], gt)) <= 0.05
    n = sr.numel()
    assert abs(sr.double().sum().item() - float(g["sr_sum"])) / n <= 2e-3     # mean within bf16 noise


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_batch_items_are_independent(compute):
    """Inference shards by batch item with no collective: item i alone == item i in a batch."""
    gen = Generator(num_block=2)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=4, num_block=2)))
    gen = gen.to(DEV).eval().set_compute(compute)
    x = torch.from_numpy(W.lr_input((3, 3, 20, 28), seed=21)).to(DEV)
    with torch.no_grad():
        sr, feat = gen(x)
        sr1, feat1 = gen(x[1:2])
    assert torch.equal(sr[1:2], sr1) and torch.equal(feat[1:2], feat1)


def test_tile_with_halo_equals_whole_frame_interior():
    """Tiled inference (BASELINE config 5): far from the tile border a tile's output equals the whole
    frame's.  Receptive-field radius of a 2-block generator = 4 + 5 + 1 + 0.5 + 0.25 -> 11 LR pixels."""
    gen = Generator(num_block=2)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=4, num_block=2)))
    gen = gen.to(DEV).eval()
    x = torch.from_numpy(W.lr_input((1, 3, 48, 64), seed=22)).to(DEV)
    with torch.no_grad():
        whole, _ = gen(x)
        tile, _ = gen(x[:, :, 8:40, 16:56].contiguous())
    r = 11
    a = whole[:, :, 4 * (8 + r):4 * (40 - r), 4 * (16 + r):4 * (56 - r)]
    b = tile[:, :, 4 * r:4 * (32 - r), 4 * r:4 * (40 - r)]
    assert (a - b).abs().max().item() <= 1e-5


def test_outputs_do_not_alias_between_calls():
    gen = Generator(num_block=1).to(DEV).eval()
    x = torch.from_numpy(W.lr_input((1, 3, 8, 8), seed=2)).to(DEV)
    with torch.no_grad():
        a, fa = gen(x)
        a0 = a.clone()
        b, fb = gen(-x)
    assert a.data_ptr() != b.data_ptr() and fa.data_ptr() != fb.data_ptr()
    assert torch.equal(a, a0)


def test_plan_follows_weight_updates():
    gen = Generator(num_block=1).to(DEV).eval()
    x = torch.from_numpy(W.lr_input((1, 3, 8, 8), seed=2)).to(DEV)
    with torch.no_grad():
        a, _ = gen(x)
        gen.last_conv.body[0].weight.mul_(0.5)          # in-place update bumps the version counter
        b, _ = gen(x)
        sd = O.to_torch(W.generator_weights(seed=9, num_block=1))
        gen.load_state_dict(sd)
        c, _ = gen(x)
        ref, _ = O.generator_forward(sd, x.cpu(), num_block=1)
    assert not torch.equal(a, b)
    assert (c.cpu() - ref).abs().max().item() <= 2e-4


def test_guards():
    gen = Generator(num_block=1).to(DEV)
    x = torch.zeros(1, 3, 8, 8, device=DEV)
    gen.train()
    with torch.no_grad():                      # train-mode BN runs on the training kernels (tests/test_train_step_gpu.py)
        sr, _ = gen(x)
    assert sr.shape == (1, 3, 32, 32)
    gen.eval()
    with pytest.raises(NotImplementedError, match="eval-mode BatchNorm"):
        gen(x)                                 # autograd recording through eval-mode BN: not a path of the reference loops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gen.cpu()(x.cpu())
    gen.to(DEV)


def test_discriminator_eval_matches_reference(golden_dir):
    g = _load(golden_dir, "disc_p32_b3.npz")
    d = Discriminator(patch_size=int(g["patch"]))
    d.load_state_dict(O.to_torch(W.discriminator_weights(seed=int(g["wseed"]), patch_size=int(g["patch"]))))
    d = d.to(DEV).eval()
    with torch.no_grad():
        p = d(torch.from_numpy(g["x"]).to(DEV))
    assert p.shape == g["prob_eval"].shape
    assert np.abs(p.cpu().numpy() - g["prob_eval"]).max() <= 1e-5
    d.set_compute("bf16")
    with torch.no_grad():
        pb = d(torch.from_numpy(g["x"]).to(DEV))
    assert np.abs(pb.cpu().numpy() - g["prob_eval"]).max() <= 2e-2


def test_tiled_inference_equals_whole_frame_when_halo_covers_receptive_field():
    """BASELINE config 5 path (LR tiles + halo, stitched): with halo >= receptive-field radius (11 LR px for a
    2-block generator) the stitched frame is the whole-frame result; tiles are dealt to ranks with no exchange."""
    from srmi import tiling
    gen = Generator(num_block=2)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=4, num_block=2)))
    gen = gen.to(DEV).eval()
    x = torch.from_numpy(W.lr_input((1, 3, 54, 72), seed=23)).to(DEV)
    with torch.no_grad():
        whole, _ = gen(x)
        tiled, n = tiling.sr_tiled(gen, x, tile=(27, 24), halo=12)
        assert n == 6
        assert (whole - tiled).abs().max().item() <= 1e-5
        # two "ranks": disjoint tile sets whose union is the frame
        a, na = tiling.sr_tiled(gen, x, tile=(27, 24), halo=12, rank=0, world=2)
        b, nb = tiling.sr_tiled(gen, x, tile=(27, 24), halo=12, rank=1, world=2)
        assert na + nb == 6 and torch.equal(a + b, tiled)
        # BASELINE's 8-px halo: per-tile parity by construction, frame-level approximation reported
        approx, _ = tiling.sr_tiled(gen, x, tile=(27, 24), halo=8)
    print("8-px halo vs whole frame max|d| =", (whole - approx).abs().max().item())


def test_data_mutation_needs_invalidate():
    """a write through `.data` does not bump torch's version counter, so the cached plan keeps the old
    packed weights until `srmi.invalidate(module)`; a write under no_grad is seen automatically."""
    import srmi
    gen = Generator(num_block=2)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=0, num_block=2)))
    gen = gen.to(DEV).eval()
    x = torch.from_numpy(W.lr_input((1, 3, 12, 16), seed=1)).to(DEV)
    with torch.no_grad():
        a, _ = gen(x)
        gen.last_conv.body[0].weight.data.mul_(0.5)
        srmi.invalidate(gen)
        b, _ = gen(x)
        assert not torch.equal(a, b)
        gen.last_conv.body[0].weight.mul_(2.0)          # versioned write: detected without invalidate()
        c, _ = gen(x)
        assert torch.allclose(a, c, atol=1e-6)


def test_fused_tail_can_be_switched_off(golden_dir):
    """srmi.engine.set_fuse_tail(False): the second Upsampler + last_conv run as two launches (no int32 fixed-point
    accumulation, no +-900 clamp of its tap contributions: include/srmi.h SRMI_FUSE_UPTAIL); same result to 4e-5."""
    from srmi import engine
    g = _load(golden_dir, "gen_b2_2x3x8x8.npz")
    x = torch.from_numpy(g["x"]).to(DEV)
    try:
        gen = _make_gen(g, "bf16")
        with torch.no_grad():
            fused, _ = gen(x)
        labels = [u[0] for u in gen._plans.get(gen, x.shape[0], x.shape[2], x.shape[3], 1).launches]
        assert "upfuse_bf16" in labels
        engine.set_fuse_tail(False)
        engine.invalidate(gen)
        with torch.no_grad():
            plain, _ = gen(x)
        labels = [u[0] for u in gen._plans.get(gen, x.shape[0], x.shape[2], x.shape[3], 1).launches]
        assert "upfuse_bf16" not in labels and "c64_thin_bf16" in labels
        assert float((fused - plain).abs().max()) <= 4e-5
    finally:
        engine.set_fuse_tail(True)


def test_feat_copy_from_the_upsampler_launch_equals_the_relayout_pass(golden_dir):
    """bf16 path: `feat` (NCHW f32) is written by the first Upsampler's c64 launch from the tile it stages
    (srmi_conv_args.x_copy) - bit-identical to the stand-alone re-layout of the NHWC bf16 tensor; a non-c64 launch
    refuses an x_copy."""
    import ctypes as C
    from srmi import engine, functional as Fn, _lib as L
    g = _load(golden_dir, "gen_b2_2x3x8x8.npz")
    x = torch.from_numpy(g["x"]).to(DEV)
    for shape in [(2, 3, 8, 8), (1, 3, 20, 36), (3, 3, 13, 100)]:
        xin = x if shape == tuple(x.shape) else torch.from_numpy(W.lr_input(shape, seed=5)).to(DEV)
        gen = _make_gen(g, "bf16")
        with torch.no_grad():
            sr_a, feat_a = gen(xin)
        plan = gen._plans.get(gen, xin.shape[0], xin.shape[2], xin.shape[3], 1)
        assert plan.feat_idx is not None
        try:
            engine._X_COPY = False
            engine.invalidate(gen)
            with torch.no_grad():
                sr_b, feat_b = gen(xin)
            assert gen._plans.get(gen, xin.shape[0], xin.shape[2], xin.shape[3], 1).feat_idx is None
        finally:
            engine._X_COPY = True
        assert torch.equal(feat_a, feat_b) and torch.equal(sr_a, sr_b)
    # W % 4 != 0: the plan falls back to the re-layout pass
    gen = _make_gen(g, "bf16")
    with torch.no_grad():
        gen(torch.from_numpy(W.lr_input((1, 3, 9, 10), seed=6)).to(DEV))
    assert gen._plans.get(gen, 1, 9, 10, 1).feat_idx is None
    # a kernel that cannot write the copy says so
    d = Fn.make_desc(1, 8, 8, 64, 3, 3, in_dtype=L.BF16, out_dtype=L.F32, out_layout=L.NCHW, compute=L.COMPUTE_BF16, act=L.ACT_TANH)
    a = L.ConvArgs()
    a.d = d
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    for f in ("x", "w", "scale", "shift", "y", "x_copy"):
        setattr(a, f, buf.data_ptr())
    assert L.load().srmi_conv2d_fwd(C.byref(a), 0) == -2          # SRMI_EUNSUPPORTED
