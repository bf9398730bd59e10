"""Deterministic NumPy weight / input recipe (TEST INFRASTRUCTURE - see oracle/__init__.py).

Independent of torch's RNG so the GPU box regenerates bit-identical weights without the
reference.  Each state_dict key gets its own PCG64 stream seeded by (seed, crc32(key)).

The key schemas restate the module trees of the reference:
  Generator      P/srgan_model.py:8-25  (+ P/ops.py:18-31,37-43,70-80), SURVEY App. B
  Discriminator  P/srgan_model.py:41-62 (+ P/ops.py:86-92)
(P/ = "pytorch srgan with lpips,dists/SRGAN_CustomDataset-main/").

The distributions are deliberately NOT the reference's init (N(0,0.02), BN identity):
that init makes eval-mode BN a no-op and keeps every output within +-0.04, which hides
errors.  Here BN statistics are non-trivial and conv gains keep activations O(1) through
16 residual blocks so tanh is not saturated.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

# kinds
CONV_W, CONV_B, BN_G, BN_B, BN_RM, BN_RV, BN_NBT, PRELU, LIN_W, LIN_B = (
    "conv_w", "conv_b", "bn_g", "bn_b", "bn_rm", "bn_rv", "bn_nbt", "prelu", "lin_w", "lin_b")


def _conv(s, prefix, cout, cin, k):
    s[prefix + ".weight"] = ((cout, cin, k, k), CONV_W)
    s[prefix + ".bias"] = ((cout,), CONV_B)


def _bn(s, prefix, c):
    s[prefix + ".weight"] = ((c,), BN_G)
    s[prefix + ".bias"] = ((c,), BN_B)
    s[prefix + ".running_mean"] = ((c,), BN_RM)
    s[prefix + ".running_var"] = ((c,), BN_RV)
    s[prefix + ".num_batches_tracked"] = ((), BN_NBT)


def generator_schema(img_feat=3, n_feats=64, kernel_size=3, num_block=16, scale=4):
    """Ordered key -> (shape, kind) for Generator.state_dict() (P/srgan_model.py:8-25).

    The single shared PReLU (default-argument module, P/srgan_model.py:8) shows up as
    19 alias keys; all of them carry kind PRELU and the same value.
    """
    s = OrderedDict()
    _conv(s, "conv01.body.0", n_feats, img_feat, 9)           # P/srgan_model.py:11
    s["conv01.body.1.weight"] = ((1,), PRELU)
    for i in range(num_block):                                   # P/srgan_model.py:13-14
        p = f"body.{i}.body"
        _conv(s, p + ".0.body.0", n_feats, n_feats, 3)           # P/ops.py:41
        _bn(s, p + ".0.body.1", n_feats)
        s[p + ".0.body.2.weight"] = ((1,), PRELU)
        _conv(s, p + ".1.body.0", n_feats, n_feats, 3)           # P/ops.py:42
        _bn(s, p + ".1.body.1", n_feats)
    _conv(s, "conv02.body.0", n_feats, n_feats, 3)            # P/srgan_model.py:16
    _bn(s, "conv02.body.1", n_feats)
    ups = [2, 2] if scale == 4 else [scale]                      # P/srgan_model.py:18-21
    for j, r in enumerate(ups):
        _conv(s, f"tail.{j}.body.0.body.0", n_feats * r * r, n_feats, 3)   # P/ops.py:74
        s[f"tail.{j}.body.2.weight"] = ((1,), PRELU)
    _conv(s, "last_conv.body.0", img_feat, n_feats, 3)        # P/srgan_model.py:25
    return s


def discriminator_schema(img_feat=3, n_feats=64, num_of_block=3, patch_size=96):
    """Ordered key -> (shape, kind) for Discriminator.state_dict() (P/srgan_model.py:41-62)."""
    s = OrderedDict()
    _conv(s, "conv01.body.0", n_feats, img_feat, 3)
    _conv(s, "conv02.body.0", n_feats, n_feats, 3)
    for i in range(num_of_block):
        cin, cout = n_feats * 2 ** i, n_feats * 2 ** (i + 1)
        _conv(s, f"body.{i}.body.0.body.0", cout, cin, 3)
        _bn(s, f"body.{i}.body.0.body.1", cout)
        _conv(s, f"body.{i}.body.1.body.0", cout, cout, 3)
        _bn(s, f"body.{i}.body.1.body.1", cout)
    lin = ((patch_size // (2 ** (num_of_block + 1))) ** 2) * (n_feats * 2 ** num_of_block)
    s["tail.0.weight"] = ((1024, lin), LIN_W)
    s["tail.0.bias"] = ((1024,), LIN_B)
    s["tail.2.weight"] = ((1, 1024), LIN_W)
    s["tail.2.bias"] = ((1,), LIN_B)
    return s


VGG19_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def vgg19_schema():
    """Ordered key -> (shape, kind) of torchvision's vgg19 `features` stack as wrapped by P/vgg19.py:11
    (`vgg_feature.<index>.weight/bias`; public VGG19 "E" configuration, conv3x3 pad 1 + ReLU, 2x2 max-pools)."""
    s = OrderedDict()
    idx, cin = 0, 3
    for v in VGG19_CFG_E:
        if v == "M":
            idx += 1
        else:
            _conv(s, f"vgg_feature.{idx}", v, cin, 3)
            cin = v
            idx += 2
    return s


def vgg19_weights(seed=0):
    """Seeded stand-in for the ImageNet weights (unavailable offline): He-scaled so that activations and
    gradients survive the 16 ReLU layers."""
    return fill(vgg19_schema(), seed=seed, gain=1.4142)


def _rng(seed, key):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(key.encode())]))


def fill(schema, seed=0, alpha=0.25, gain=0.75):
    """schema -> OrderedDict key -> np.ndarray (float32; num_batches_tracked int64)."""
    out = OrderedDict()
    for key, (shape, kind) in schema.items():
        r = _rng(seed, key)
        if kind == CONV_W:
            fan_in = shape[1] * shape[2] * shape[3]
            v = r.standard_normal(shape) * (gain / np.sqrt(fan_in))
        elif kind == LIN_W:
            v = r.standard_normal(shape) * (1.0 / np.sqrt(shape[1]))
        elif kind in (CONV_B, LIN_B):
            v = r.uniform(-0.1, 0.1, shape)
        elif kind == BN_G:
            v = r.uniform(0.5, 1.5, shape)
        elif kind == BN_B:
            v = r.uniform(-0.2, 0.2, shape)
        elif kind == BN_RM:
            v = r.standard_normal(shape) * 0.1
        elif kind == BN_RV:
            v = r.uniform(0.5, 1.5, shape)
        elif kind == BN_NBT:
            out[key] = np.zeros((), np.int64)
            continue
        elif kind == PRELU:
            v = np.full(shape, alpha)
        else:  # pragma: no cover
            raise ValueError(kind)
        out[key] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def generator_weights(seed=0, alpha=0.25, **kw):
    return fill(generator_schema(**kw), seed=seed, alpha=alpha)


def discriminator_weights(seed=0, **kw):
    return fill(discriminator_schema(**kw), seed=seed)


def lr_input(shape, seed=1, lo=-1.0, hi=1.0):
    """Synthetic LR batch U(lo,hi), NCHW float32 (dataset contract: P/dataset.py:40-47)."""
    r = np.random.Generator(np.random.PCG64([int(seed), 0xC0FFEE]))
    return np.ascontiguousarray(r.uniform(lo, hi, shape), dtype=np.float32)


def randomise_module(m, seed):
    """Seeded weights for the Keras-derived modules (srmi/keras_models.py), by parameter shape: conv W ~ N(0,
    0.8/sqrt(fan_in)), bias ~ U(-0.1, 0.1), per-channel PReLU slopes ~ U(0.05, 0.4).  torch's CPU generator, so the
    fixture script (oracle/make_golden_configs.py) and the GPU tests build identical weights."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 4:
                p.copy_(torch.randn(p.shape, generator=g) * (0.8 / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5))
            elif n.endswith("bias"):
                p.copy_(torch.empty(p.shape).uniform_(-0.1, 0.1, generator=g))
            else:
                p.copy_(torch.empty(p.shape).uniform_(0.05, 0.4, generator=g))
    return m
