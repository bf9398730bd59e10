"""Training data pipeline on the device: the reference's `mydata` / `crop` / `augmentation`
(P/dataset.py:9-132) with the same names and constructor signatures.

The reference keeps 93 uint8 images in host memory and, per sample, converts to float, crops, flips, transposes and
re-lays out with numpy, one image at a time under `DataLoader(num_workers=0)` (P/mode.py:84-86) - with a training
step of a few tens of milliseconds that loop is the bottleneck.  Here the uint8 images live in device memory and a whole
batch is cut, flipped, transposed, normalised and written as NCHW float32 by ONE kernel (`srmi_crop_augment_u8`).
The random draws stay on the host and are made by the same calls in the same order as the reference makes them
(`random.randrange` in `crop` then `augmentation`, sample by sample; torch's RandomSampler for the shuffle), so with the
same seeds the batches are bit-identical to `DataLoader(mydata(...), shuffle=True)` (tests/test_dataset_gpu.py).
"""
from __future__ import annotations

import ctypes as C
import os
import random

import numpy as np
import torch
from torch.utils.data import BatchSampler, Dataset, RandomSampler, SequentialSampler

from . import _lib as L
from .functional import stream_ptr


class crop(object):
    """P/dataset.py:81-102.  `draw` makes the reference's two random draws (ix, then iy)."""

    def __init__(self, scale, patch_size):
        self.scale = scale
        self.patch_size = patch_size

    def draw(self, ih, iw):
        ix = random.randrange(0, iw - self.patch_size + 1)
        iy = random.randrange(0, ih - self.patch_size + 1)
        return iy, ix

    def __call__(self, sample):
        """The reference's per-sample path on host arrays (HWC numpy), for callers that use the transform directly or
        inside a torchvision `Compose` (P/dataset.py:87-100).  `mydata` does not go through it (device batches)."""
        LR_img, GT_img = sample['LR'], sample['GT']
        ih, iw = LR_img.shape[:2]
        iy, ix = self.draw(ih, iw)
        p, s = self.patch_size, self.scale
        return {'LR': LR_img[iy:iy + p, ix:ix + p], 'GT': GT_img[iy * s:iy * s + s * p, ix * s:ix * s + s * p]}


class augmentation(object):
    """P/dataset.py:104-132.  `draw` makes the reference's three random draws; flips first, transpose last."""

    def draw(self):
        hor_flip = random.randrange(0, 2)
        ver_flip = random.randrange(0, 2)
        rot = random.randrange(0, 2)
        return (L.AUG_HFLIP if hor_flip else 0) | (L.AUG_VFLIP if ver_flip else 0) | (L.AUG_TRANSPOSE if rot else 0)

    def __call__(self, sample):
        """Host-array form (P/dataset.py:104-132): fliplr, flipud, then transpose(1, 0, 2), copies like the reference."""
        LR_img, GT_img = sample['LR'], sample['GT']
        flags = self.draw()
        if flags & L.AUG_HFLIP:
            LR_img, GT_img = np.fliplr(LR_img).copy(), np.fliplr(GT_img).copy()
        if flags & L.AUG_VFLIP:
            LR_img, GT_img = np.flipud(LR_img).copy(), np.flipud(GT_img).copy()
        if flags & L.AUG_TRANSPOSE:
            LR_img, GT_img = LR_img.transpose(1, 0, 2), GT_img.transpose(1, 0, 2)
        return {'LR': LR_img, 'GT': GT_img}


class Compose(object):
    """Stand-in for torchvision.transforms.Compose (any object with a `.transforms` list works as well)."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample


def _load_rgb(path):
    from PIL import Image
    return np.array(Image.open(path).convert("RGB")).astype(np.uint8)


class mydata(Dataset):
    """`mydata(LR_path, GT_path, in_memory=True, transform=None)` (P/dataset.py:9-49) with the images in device memory.

    `in_memory` keeps its one observable effect in the reference: the in-memory path normalises in float32
    (P/dataset.py:33-34), the from-disk path in float64 rounded once (:37-41).  `transform` must be `crop`,
    optionally followed by `augmentation` (the reference's `Compose([crop(...), augmentation()])`, P/mode.py:75):
    the batch kernel cuts first, then flips, then transposes.  Any other list (augmentation before crop, repeats,
    foreign transforms) raises instead of being silently re-ordered.
    """

    def __init__(self, LR_path=None, GT_path=None, in_memory=True, transform=None, device="cuda:0", images=None):
        self.LR_path, self.GT_path = LR_path, GT_path
        self.in_memory = in_memory
        self.transform = transform
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("srmi.dataset: the data pipeline runs on the GPU (no CPU fallback)")
        if images is not None:
            lr_imgs, gt_imgs = images
        else:
            lr_names, gt_names = sorted(os.listdir(LR_path)), sorted(os.listdir(GT_path))
            lr_imgs = [_load_rgb(os.path.join(LR_path, n)) for n in lr_names]
            gt_imgs = [_load_rgb(os.path.join(GT_path, n)) for n in gt_names]
        if len(lr_imgs) != len(gt_imgs):
            raise ValueError("mydata: LR and GT image counts differ")
        self.transforms = [] if transform is None else list(getattr(transform, "transforms", [transform]))
        kinds = [type(t) for t in self.transforms]
        if kinds not in ([], [crop], [crop, augmentation]):
            raise TypeError("mydata: transform must be crop, or Compose([crop, augmentation]) in that order (the device "
                            f"kernel cuts, then flips, then transposes); got {[k.__name__ for k in kinds]}")
        self._crop = next((t for t in self.transforms if isinstance(t, crop)), None)
        self.scale = self._crop.scale if self._crop is not None else None
        self.shapes, self.lr_off, self.gt_off = [], [], []
        lo = go = 0
        for lr, gt in zip(lr_imgs, gt_imgs):
            if lr.dtype != np.uint8 or gt.dtype != np.uint8 or lr.ndim != 3 or gt.ndim != 3 or lr.shape[2] != gt.shape[2]:
                raise TypeError("mydata: images must be uint8 HWC with equal channel counts")
            if self.scale is not None and (gt.shape[0] != lr.shape[0] * self.scale or gt.shape[1] != lr.shape[1] * self.scale):
                raise ValueError("mydata: every GT image must be scale x its LR image")
            self.shapes.append(lr.shape[:2])
            self.lr_off.append(lo); self.gt_off.append(go)
            lo += lr.size; go += gt.size
        self.channels = lr_imgs[0].shape[2] if lr_imgs else 3
        flat = lambda imgs: torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs]) if imgs else np.zeros(0, np.uint8))
        self.lr_pack = flat(lr_imgs).to(self.device)
        self.gt_pack = flat(gt_imgs).to(self.device)

    def __len__(self):
        return len(self.shapes)

    def _draw(self, i):
        """The random draws `self.transform(img_item)` makes for sample i (P/dataset.py:43-44), in the same order."""
        ih, iw = self.shapes[i]
        iy = ix = flags = 0
        for t in self.transforms:
            if isinstance(t, crop):
                iy, ix = t.draw(ih, iw)
            elif isinstance(t, augmentation):
                flags = t.draw()
            else:
                raise TypeError(f"mydata: unsupported transform {type(t).__name__} (crop / augmentation)")
        return iy, ix, flags

    def batch(self, indices):
        """{'LR': [B,C,p,p], 'GT': [B,C,s*p,s*p]} float32 device tensors for the given sample indices."""
        if self._crop is None:
            raise NotImplementedError("mydata: batches need a crop transform (one patch size per batch); for whole "
                                      "images use srmi.imageio.from_uint8 / testOnly-style evaluation")
        B, p, s_, Cc = len(indices), self._crop.patch_size, self._crop.scale, self.channels
        items = (L.CropItem * max(B, 1))()
        for b, i in enumerate(indices):
            iy, ix, flags = self._draw(i)
            it = items[b]
            it.lr_offset, it.gt_offset = self.lr_off[i], self.gt_off[i]
            it.lr_h, it.lr_w = self.shapes[i]
            it.iy, it.ix, it.flags, it.reserved = iy, ix, flags, 0
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)[: B * C.sizeof(L.CropItem)]
        items_d = raw.to(self.device)
        lr = torch.empty((B, Cc, p, p), dtype=torch.float32, device=self.device)
        gt = torch.empty((B, Cc, p * s_, p * s_), dtype=torch.float32, device=self.device)
        L.check(L.load().srmi_crop_augment_u8(self.lr_pack.data_ptr(), self.gt_pack.data_ptr(), items_d.data_ptr(), B, p, s_,
                                              Cc, 1 if self.in_memory else 0, lr.data_ptr(), gt.data_ptr(), stream_ptr()))
        self._keep = items_d        # alive until the launch has consumed it (same-stream reuse is ordered)
        return {"LR": lr, "GT": gt}

    def __getitem__(self, i):
        """One sample as the reference returns it: numpy CHW float32 (a device batch of one, copied back)."""
        out = self.batch([i])
        return {"GT": out["GT"][0].cpu().numpy(), "LR": out["LR"][0].cpu().numpy()}

    def loader(self, batch_size=1, shuffle=False, drop_last=False):
        """Iterable of device batches in the order `DataLoader(self, batch_size, shuffle, num_workers=0)` yields them."""
        return _DeviceLoader(self, batch_size, shuffle, drop_last)


class testOnly_data(Dataset):
    """`testOnly_data(LR_path, in_memory=True, transform=None)` (P/dataset.py:52-78): LR images only, whole frames,
    no `.convert("RGB")` (as the reference: greyscale / RGBA files keep their channel count).  `__getitem__` returns the
    reference's `{'LR': CHW float32 numpy}`; `device_item(i)` returns the same image as a [1, C, H, W] device tensor
    converted by `srmi_u8hwc_to_f32nchw` (bit-identical: `v / 127.5 - 1` in f64, rounded once), which is what
    `srmi.mode.test_only` feeds the generator."""

    def __init__(self, LR_path, in_memory=True, transform=None, device="cuda:0"):
        self.LR_path = LR_path
        self.names = sorted(os.listdir(LR_path))
        self.in_memory = in_memory
        self.device = torch.device(device)
        self.LR_img = [self._read(n) for n in self.names] if in_memory else None

    def _read(self, name):
        from PIL import Image
        return np.array(Image.open(os.path.join(self.LR_path, name)))

    def __len__(self):
        return len(self.names)

    def _u8(self, i):
        LR = self.LR_img[i] if self.in_memory else self._read(self.names[i])
        if LR.ndim != 3:
            raise ValueError(f"testOnly_data: {self.names[i]} is not an HWC image (the reference's transpose(2, 0, 1) "
                             "fails on it too, P/dataset.py:76)")
        return LR

    def __getitem__(self, i):
        LR = self._u8(i)
        return {'LR': ((LR / 127.5) - 1.0).transpose(2, 0, 1).astype(np.float32)}

    def device_item(self, i):
        from .imageio import from_uint8
        LR = self._u8(i)
        if LR.dtype != np.uint8:
            return {'LR': torch.from_numpy(self[i]['LR'])[None].to(self.device)}
        return {'LR': from_uint8(torch.from_numpy(np.ascontiguousarray(LR))[None].to(self.device))}


class _DeviceLoader:
    def __init__(self, ds, batch_size, shuffle, drop_last):
        self.ds = ds
        sampler = RandomSampler(ds) if shuffle else SequentialSampler(ds)
        self.batch_sampler = BatchSampler(sampler, batch_size, drop_last)

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        # a DataLoader iterator draws its base seed from the global torch generator before the sampler draws its own
        torch.empty((), dtype=torch.int64).random_()
        for indices in self.batch_sampler:
            yield self.ds.batch(indices)
