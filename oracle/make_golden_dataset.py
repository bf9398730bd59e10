#!/usr/bin/env python3
"""Generate tests/golden/dataset_pipeline.npz by running the REFERENCE's data pipeline (TEST INFRASTRUCTURE).

    python oracle/make_golden_dataset.py          (build container only: needs /root/reference, read-only)

Seeded random uint8 images are written as PNGs into a temporary directory; the reference's `mydata` with
`crop` + `augmentation` (P/dataset.py, imported unmodified) is iterated by a real torch `DataLoader(shuffle=True,
num_workers=0)` exactly as P/mode.py:84-86 builds it (torchvision's `Compose` is absent here: a 3-line stand-in
that calls the transforms in order - it is torchvision code, not the reference's).  Stored: the images and every
batch of two epochs (in-memory = f32 arithmetic) and of one epoch of the from-disk path (f64 arithmetic).  Data only.
"""
import os
import random
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/pytorch srgan with lpips,dists/SRGAN_CustomDataset-main"
sys.path.insert(0, REF)

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader

import dataset as ref_ds  # noqa: E402  (the reference)


class Compose:      # stand-in for torchvision.transforms.Compose
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


SCALE, PATCH, BATCH, SEED = 4, 8, 4, 1234
g = np.random.default_rng(7)
sizes = [(20, 24), (9, 31), (16, 8), (8, 8), (27, 13), (12, 40), (10, 10)]       # LR (h, w); includes patch-sized edges
out = {"scale": SCALE, "patch": PATCH, "batch": BATCH, "seed": SEED, "n": len(sizes)}
with tempfile.TemporaryDirectory() as tmp:
    lr_dir, gt_dir = os.path.join(tmp, "LR"), os.path.join(tmp, "GT")
    os.makedirs(lr_dir); os.makedirs(gt_dir)
    for i, (h, w) in enumerate(sizes):
        lr = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        gt = g.integers(0, 256, (h * SCALE, w * SCALE, 3), dtype=np.uint8)
        Image.fromarray(lr).save(os.path.join(lr_dir, f"im{i:02d}.png"))
        Image.fromarray(gt).save(os.path.join(gt_dir, f"im{i:02d}.png"))
        out[f"lr{i}"], out[f"gt{i}"] = lr, gt
    for tag, in_memory, epochs in (("mem", True, 2), ("disk", False, 1)):
        tf = Compose([ref_ds.crop(SCALE, PATCH), ref_ds.augmentation()])
        ds = ref_ds.mydata(GT_path=gt_dir, LR_path=lr_dir, in_memory=in_memory, transform=tf)
        random.seed(SEED)
        torch.manual_seed(SEED)
        loader = DataLoader(ds, batch_size=BATCH, shuffle=True, num_workers=0)
        k = 0
        for _ in range(epochs):
            for batch in loader:
                out[f"{tag}_LR{k}"] = batch["LR"].numpy()
                out[f"{tag}_GT{k}"] = batch["GT"].numpy()
                k += 1
        out[f"{tag}_batches"] = k
        print(tag, "batches:", k, "LR", out[f"{tag}_LR0"].shape, out[f"{tag}_LR0"].dtype, "GT", out[f"{tag}_GT0"].shape)
path = os.path.join(ROOT, "tests", "golden", "dataset_pipeline.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
