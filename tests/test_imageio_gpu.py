"""GPU: device-side image pre/post-processing (srmi.imageio) bit-exact against the reference's numpy expressions
(P/dataset.py:40-47, P/mode.py:360, :397-400), and `test_only` / `test` end to end on a folder of PNGs."""
import argparse
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import metrics_oracle as M
from oracle import srgan_oracle as O
from oracle import weights as W
from srmi import imageio, mode
from srmi.srgan_model import Generator

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_from_uint8_is_bit_exact():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 17, 23, 3), dtype=np.uint8)
    img[0, 0, 0] = (0, 255, 127)
    ref = np.stack([((im / 127.5) - 1.0).transpose(2, 0, 1).astype(np.float32) for im in img])   # P/dataset.py:40-47
    got = imageio.from_uint8(torch.from_numpy(img).to(DEV)).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, ref)
    assert imageio.from_uint8(torch.from_numpy(img[0]).to(DEV)).shape == (1, 3, 17, 23)


@pytest.mark.parametrize("clip", [False, True])
def test_to_uint8_is_bit_exact(clip):
    rng = np.random.default_rng(1)
    out = np.tanh(rng.standard_normal((2, 3, 19, 21)) * 2).astype(np.float32)
    out[0, :, 0, 0] = (-1.0, 1.0, 0.0)
    if clip:
        out[1, 0, 1, :5] = (1.3, -1.2, 1.0000001, -1.0000001, 0.999)
    ref = []
    for o in out:                                   # P/mode.py:359-364 / :397-400
        if clip:
            o = np.clip(o, -1.0, 1.0)
        o = (o + 1.0) / 2.0
        ref.append((o.transpose(1, 2, 0) * 255.0).astype(np.uint8))
    got = imageio.to_uint8(torch.from_numpy(out).to(DEV), clip=clip).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, np.stack(ref))


def test_test_only_and_test_write_the_reference_files(tmp_path):
    rng = np.random.default_rng(2)
    lr_dir, gt_dir, out_dir = tmp_path / "lr", tmp_path / "gt", tmp_path / "result"
    lr_dir.mkdir(); gt_dir.mkdir()
    lrs = []
    for i in range(2):
        lr = rng.integers(0, 256, (12, 16, 3), dtype=np.uint8)
        Image.fromarray(lr).save(lr_dir / f"{i}.png")
        Image.fromarray(rng.integers(0, 256, (50, 66, 3), dtype=np.uint8)).save(gt_dir / f"{i}.png")
        lrs.append(lr)
    np_sd = W.generator_weights(seed=0, num_block=1)
    ckpt = tmp_path / "g.pt"
    torch.save({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()}, ckpt)
    args = argparse.Namespace(LR_path=str(lr_dir), GT_path=str(gt_dir), res_num=1, scale=4, generator_path=str(ckpt))
    assert mode.test_only(args, out_dir=str(out_dir)) == 2
    sd = O.to_torch(np_sd)
    for i, lr in enumerate(lrs):                    # oracle: the reference pipeline on the host
        x = torch.from_numpy(((lr / 127.5) - 1.0).transpose(2, 0, 1).astype(np.float32))[None]
        with torch.no_grad():
            sr, _ = O.generator_forward(sd, x, num_block=1)
        ref = (((sr[0].numpy() + 1.0) / 2.0).transpose(1, 2, 0) * 255.0).astype(np.uint8)
        got = np.array(Image.open(out_dir / ("res_%04d.png" % i)))
        assert got.shape == ref.shape == (48, 64, 3)
        assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1          # <=1e-4 on pixels: at most one grey level
        assert (got != ref).mean() < 0.01
    psnrs = mode.test(args, out_dir=str(out_dir), result_txt=str(tmp_path / "result.txt"))
    txt = open(tmp_path / "result.txt").read()
    assert txt.count("\npsnr :") + txt.startswith("psnr :") == 2 and "avg psnr" in txt
    gt0 = torch.from_numpy(((np.array(Image.open(gt_dir / "0.png")) / 127.5) - 1.0).transpose(2, 0, 1).astype(np.float32))[None]
    x0 = torch.from_numpy(((lrs[0] / 127.5) - 1.0).transpose(2, 0, 1).astype(np.float32))[None]
    with torch.no_grad():
        sr0, _ = O.generator_forward(sd, x0, num_block=1)
    p_ref, _ = M.calculate_metrics(gt0.numpy(), sr0.numpy(), scale=4, clip=True)
    assert abs(psnrs[0] - p_ref) <= 1e-3
