"""`test_only(args)` and `test(args)` with the reference's names, arguments and output files (P/mode.py:335-401),
running the generator and the pre/post-processing on the MI355X.

Differences from the reference, all on the host side of the boundary: images cross PCIe as uint8 and are converted
on the device (srmi.imageio); the Y-PSNR of `test` is computed on the device (srmi.metrics); results are identical
(uint8 images bit-exact given the same generator output).  `train(args)` is not re-implemented here: the reference
loop runs unmodified on these modules.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from PIL import Image

from . import imageio, metrics
from .srgan_model import Generator


def _load_generator(args, device):
    generator = Generator(img_feat=3, n_feats=64, kernel_size=3, num_block=args.res_num)
    generator.load_state_dict(torch.load(args.generator_path, map_location="cpu", weights_only=True))
    generator = generator.to(device)
    generator.eval()
    if getattr(args, "compute", None):
        generator.set_compute(args.compute)
    return generator


def _read_rgb(path, convert):
    img = Image.open(path)
    if convert:
        img = img.convert("RGB")
    return np.array(img)


def test_only(args, out_dir="./result"):
    """P/mode.py:382-401: every image of args.LR_path -> out_dir/res_%04d.png."""
    device = torch.device("cuda")
    generator = _load_generator(args, device)
    os.makedirs(out_dir, exist_ok=True)
    names = sorted(os.listdir(args.LR_path))
    with torch.no_grad():
        for i, name in enumerate(names):
            lr = torch.from_numpy(_read_rgb(os.path.join(args.LR_path, name), False)).to(device)
            output, _ = generator(imageio.from_uint8(lr))
            result = imageio.to_uint8(output, clip=False)[0].cpu().numpy()
            Image.fromarray(result).save(os.path.join(out_dir, "res_%04d.png" % i))
    return len(names)


def test(args, out_dir="./result", result_txt="./result.txt"):
    """P/mode.py:335-379: SR every LR image, Y-PSNR against GT cropped to the SR size, result.txt + PNGs."""
    device = torch.device("cuda")
    generator = _load_generator(args, device)
    os.makedirs(out_dir, exist_ok=True)
    lr_names, gt_names = sorted(os.listdir(args.LR_path)), sorted(os.listdir(args.GT_path))
    psnr_list = []
    with open(result_txt, "w") as f, torch.no_grad():
        for i, (ln, gn) in enumerate(zip(lr_names, gt_names)):
            lr = imageio.from_uint8(torch.from_numpy(_read_rgb(os.path.join(args.LR_path, ln), True)).to(device))
            gt = imageio.from_uint8(torch.from_numpy(_read_rgb(os.path.join(args.GT_path, gn), True)).to(device))
            _, _, h, w = lr.shape
            gt = gt[:, :, :h * args.scale, :w * args.scale]
            output, _ = generator(lr)
            psnr = metrics.test_psnr(gt, output, args.scale)
            psnr_list.append(psnr)
            f.write("psnr : %04f \n" % psnr)
            Image.fromarray(imageio.to_uint8(output, clip=True)[0].cpu().numpy()).save(
                os.path.join(out_dir, "res_%04d.png" % i))
        f.write("avg psnr : %04f" % np.mean(psnr_list))
    return psnr_list
