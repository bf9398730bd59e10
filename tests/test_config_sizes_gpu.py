"""GPU: every BASELINE.json configuration at ITS size (VERDICT r1 "configs not exercised at their configuration").

  cfg1  SRResNet 16x3x96x96                      tests/test_generator_gpu.py (reference fixture gen_b16_cfg1_*)
  cfg2  SRCNN 32x1x384x384, 9-3-5 and 9-1-5      here, vs tests/golden/cfg2_* (oracle/make_golden_configs.py)
  cfg3  SRGAN step B=16, LR 24, 16 blocks        tests/test_srgan_step_gpu.py (reference fixture srgan_step_default_*)
  cfg4  RRDB x23, 8x3x256x256, bf16              here, vs tests/golden/cfg4_*
  cfg5  960x540 -> 3840x2160, 8 tiles, 8-px halo here: every tile-with-halo == Generator on the same crop (bit-exact),
        16-block generator; the whole-frame PSNR cost of the 8-px halo is measured and gated

cfg2 / cfg4 fixtures come from the Keras-semantics oracle (parity unpinned w.r.t. TensorFlow)."""
import os

import numpy as np
import pytest
import torch

from oracle import srgan_oracle as O
from oracle import weights as W
from srmi import tiling
from srmi.keras_models import SRCNN, RRDBNet
from srmi.srgan_model import Generator

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check_against_summary(y, g, tol_abs, psnr_min=None, moment_tol=1e-3):
    """y (device tensor) against a fixture of strided samples + float64 moments."""
    assert tuple(y.shape) == tuple(int(v) for v in g["shape"])
    flat = y.reshape(-1)
    got = flat[::int(g["stride"])].float().cpu()
    ref = torch.from_numpy(g["sample"])
    err = float((got - ref).abs().max())
    mse = float(((got.double() - ref.double()) ** 2).mean())
    assert err <= tol_abs, err
    ssq = float((flat.double() ** 2).sum())
    assert abs(ssq - float(g["sumsq"])) <= moment_tol * float(g["sumsq"])
    assert abs(float(flat.double().sum()) - float(g["sum"])) <= moment_tol * (float(g["sumsq"]) * flat.numel()) ** 0.5
    return err, mse


@pytest.mark.parametrize("tag,kw", [("935", dict()), ("915", dict(k=(9, 1, 5), ch=(64, 32)))])
def test_cfg2_srcnn_at_config_size(golden_dir, tag, kw):
    """BASELINE config 2: 32x1x384x384 Y planes, fp32 path <= 1e-4, bf16 path PSNR on the [0,1] output."""
    g = np.load(os.path.join(golden_dir, f"cfg2_srcnn_{tag}_32x1x384x384.npz"), allow_pickle=False)
    m = W.randomise_module(SRCNN(**kw), int(g["wseed"])).to(DEV).eval()
    x = torch.rand((32, 1, 384, 384), generator=torch.Generator().manual_seed(int(g["xseed"]))).to(DEV)
    with torch.no_grad():
        err, _ = _check_against_summary(m.set_compute("fp32")(x), g, 1e-4)
        errb, mse = _check_against_summary(m.set_compute("bf16")(x), g, 3e-2, moment_tol=2e-2)
    psnr = 10 * np.log10(1.0 / mse)
    print(f"cfg2 {tag}: fp32 max|d| {err:.2e}; bf16 max|d| {errb:.2e}, PSNR {psnr:.1f} dB")
    assert psnr >= 55.0          # measured 64.5 / 66.2 dB (fp32 path: max|d| 4e-7)


def test_cfg4_rrdb23_bf16_at_config_size(golden_dir):
    """BASELINE config 4: 23 blocks, 8x3x256x256 -> 8x3x1024x1024 in bf16: error accumulation through 23 x 4 + 5 bf16
    layers against the fp32 oracle (PSNR on [0,1] pixels >= 40 dB, moments within 2 %)."""
    g = np.load(os.path.join(golden_dir, "cfg4_rrdb23_8x3x256x256.npz"), allow_pickle=False)
    nb = int(g["num_blocks"])
    m = W.randomise_module(RRDBNet(num_blocks=nb), int(g["wseed"])).to(DEV).eval().set_compute("bf16")
    x = (torch.rand((8, 3, 256, 256), generator=torch.Generator().manual_seed(int(g["xseed"]))) * 2 - 1).to(DEV)
    with torch.no_grad():
        y = m(x)
    errb, mse = _check_against_summary(y, g, 0.1, moment_tol=2e-2)
    psnr = 10 * np.log10(1.0 / (mse / 4.0))            # tanh range [-1,1] -> [0,1] pixels
    print(f"cfg4 RRDB-23 bf16: max|d| {errb:.2e}, PSNR vs fp32 oracle {psnr:.1f} dB")
    assert psnr >= 55.0          # measured 66.6 dB, max|d| 4.3e-3


def test_cfg5_equal_extended_tiles_match_generator_on_crops():
    """The same property for the one-batch plan (`grid=(2, 4)`: eight 278 x 252 extended tiles through ONE launch
    sequence - what bench.py's cfg5 legs time): every tile's core region is bit-identical to `Generator` on that crop."""
    gen = Generator(num_block=16)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=0, num_block=16)))
    gen = gen.to(DEV).eval().set_compute("bf16")
    x = torch.from_numpy(W.lr_input((1, 3, 540, 960), seed=77)).to(DEV)
    plan = tiling.plan_tiles_equal(540, 960, 2, 4, 8)
    assert {(t[5] - t[4], t[7] - t[6]) for t in plan} == {(278, 252)}
    with torch.no_grad():
        stitched, n = tiling.sr_tiled(gen, x, halo=8, grid=(2, 4))
        assert n == 8 and stitched.shape == (1, 3, 2160, 3840)
        for (y0, y1, x0, x1, ey0, ey1, ex0, ex1) in plan:
            sr, _ = gen(x[:, :, ey0:ey1, ex0:ex1].contiguous())
            oy, ox = 4 * (y0 - ey0), 4 * (x0 - ex0)
            assert torch.equal(sr[:, :, oy:oy + 4 * (y1 - y0), ox:ox + 4 * (x1 - x0)],
                               stitched[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1]), (y0, x0)


def test_cfg5_chain_strips_match_launch_per_block_on_crops():
    """cfg5 as bench.py times it on a GPU whose CUs hold one strip's tiles: horizontal strips (tiling.strips_for_chain), all
    strips of the frame as ONE batch (the library runs the chained front strip by strip and the fused tail once), and one call
    per strip - both against `Generator` on the same crops with the chain switched off (one launch per ResBlock):
    bit-identical."""
    from importlib import import_module
    engine = import_module("super-resolution_amd.srmi.engine")
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    ny = tiling.strips_for_chain(540, 960, 8, cus)
    if ny is None or ny > 8:
        pytest.skip("no strip of this frame fits the GPU's CUs")
    gen = Generator(num_block=16)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=0, num_block=16)))
    gen = gen.to(DEV).eval().set_compute("bf16")
    x = torch.from_numpy(W.lr_input((1, 3, 540, 960), seed=77)).to(DEV)
    plan = tiling.plan_tiles_equal(540, 960, ny, 1, 8)
    with torch.no_grad():
        stitched, n = tiling.sr_tiled(gen, x, halo=8, grid=(ny, 1), max_batch=1)
        assert n == ny and stitched.shape == (1, 3, 2160, 3840)
        t = plan[0]
        p0 = gen._plans.get(gen, 1, t[5] - t[4], t[7] - t[6], 1)
        assert any(lbl == "resblock_chain_bf16" for (lbl, i0, cnt) in p0.launches), p0.launches
        batched, n = tiling.sr_tiled(gen, x, halo=8, grid=(ny, 1))
        pb = gen._plans.get(gen, ny, t[5] - t[4], t[7] - t[6], 1)
        assert any(lbl == "resblock_chain_bf16" for (lbl, i0, cnt) in pb.launches), pb.launches
        assert torch.equal(batched, stitched)
        engine.set_fuse_chain(False)
        try:
            engine.invalidate(gen)
            for (y0, y1, x0, x1, ey0, ey1, ex0, ex1) in plan:
                sr, _ = gen(x[:, :, ey0:ey1, ex0:ex1].contiguous())
                oy, ox = 4 * (y0 - ey0), 4 * (x0 - ex0)
                assert torch.equal(sr[:, :, oy:oy + 4 * (y1 - y0), ox:ox + 4 * (x1 - x0)],
                                   stitched[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1]), (y0, x0)
        finally:
            engine.set_fuse_chain(True)
            engine.invalidate(gen)


@pytest.mark.parametrize("compute", ["bf16", "fp32"])
def test_cfg5_tiles_match_generator_on_crops(compute):
    """BASELINE config 5: one 960x540 LR frame, 4 x 2 core tiles of 240 x 270 with an 8-px halo, 16-block generator.
    Parity is per tile: the core region each tile contributes to the stitched frame is bit-identical
    to what `Generator` returns for that tile's halo-extended crop run on its own."""
    gen = Generator(num_block=16)
    gen.load_state_dict(O.to_torch(W.generator_weights(seed=0, num_block=16)))
    gen = gen.to(DEV).eval().set_compute(compute)
    x = torch.from_numpy(W.lr_input((1, 3, 540, 960), seed=77)).to(DEV)
    with torch.no_grad():
        stitched, n = tiling.sr_tiled(gen, x, tile=(270, 240), halo=8)
        assert n == 8 and stitched.shape == (1, 3, 2160, 3840)
        for (y0, y1, x0, x1, ey0, ey1, ex0, ex1) in tiling.plan_tiles(540, 960, 270, 240, 8):
            sr, _ = gen(x[:, :, ey0:ey1, ex0:ex1].contiguous())
            oy, ox = 4 * (y0 - ey0), 4 * (x0 - ex0)
            core = sr[:, :, oy:oy + 4 * (y1 - y0), ox:ox + 4 * (x1 - x0)]
            assert torch.equal(core, stitched[:, :, 4 * y0:4 * y1, 4 * x0:4 * x1]), (y0, x0)
        if compute == "bf16":
            # what the 8-px halo (receptive-field radius is 39.75 LR px) costs against whole-frame inference
            whole, _ = gen(x)
            mse = float(((whole.double() - stitched.double()) ** 2).mean()) / 4.0
            psnr = 10 * np.log10(1.0 / mse)
            far = float((whole[:, :, 4 * 60:4 * 210, 4 * 60:4 * 180] - stitched[:, :, 4 * 60:4 * 210, 4 * 60:4 * 180]).abs().max())
            print(f"cfg5: stitched (8-px halo) vs whole frame PSNR {psnr:.1f} dB; tile interior max|d| {far:.2e}")
            assert psnr >= 45.0 and far == 0.0     # measured 56.5 dB; far from tile borders the frames are the same bits
