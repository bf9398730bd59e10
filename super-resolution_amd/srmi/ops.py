"""Building blocks with the reference's names and constructor signatures (P/ops.py:6-96), backed by
the gfx950 HIP kernels of libsrmi.so.

The module TREE is kept identical to the reference (``body = nn.Sequential(_conv [, BatchNorm2d]
[, act])`` etc.) so ``state_dict()`` keys, ``parameters()`` order and the shared-PReLU aliasing are
the same and checkpoints round-trip.  What changes is ``forward``: instead of
running the Sequential through ATen, each block issues ONE fused launch (conv + bias + eval-BN +
activation [+ residual] [+ PixelShuffle store]).  Whole-network forwards go through the plans in
``engine.py``; the per-block forwards below serve users who call a block on its own.

Training (autograd recording, or a BatchNorm in train mode) routes every block through
``autograd.FusedConvFn`` - fp32 conv / train-BN / epilogue / dgrad / wgrad kernels of the same library.

No fallback: CPU tensors raise; nothing here calls an ATen convolution or batch-norm kernel.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from . import autograd as AG
from . import functional as Fn
from .engine import _sig, conv_parts


class _conv(nn.Conv2d):
    """`nn.Conv2d` with the reference's quirks (P/ops.py:6-15): padding is always k//2 and bias is always
    on whatever the caller passes; weight ~ N(0, 0.02), bias = 0."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, bias):
        super().__init__(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                         stride=stride, padding=kernel_size // 2, bias=True)
        with torch.no_grad():
            self.weight.normal_(0.0, 0.02)
            self.bias.zero_()
        for p in self.parameters():
            p.requires_grad = True


def _guard(x, *modules):
    """Device check; returns True when the call must take the training kernels (autograd recording or a
    BatchNorm in train mode), False for the inference plans."""
    Fn.require_cuda(x)
    return AG.wants_training_path(x, *modules)


class _FusedCache:
    """packed weights + folded scale/shift of one conv block, rebuilt when its tensors change."""

    def __init__(self):
        self.sig = None
        self.entries = {}

    def __deepcopy__(self, memo):
        return _FusedCache()

    def __getstate__(self):
        return {"sig": None, "entries": {}}

    def get(self, owner, key, build):
        sig = _sig(owner)
        if sig != self.sig:
            self.entries.clear()
            self.sig = sig
        if key not in self.entries:
            self.entries[key] = build()
        return self.entries[key]


def _run_block(owner, cache, c2d, bn, act, x, residual=None, shuffle=1):
    """NCHW f32 in -> NCHW f32 out, exact-f32 kernel, one launch."""
    x = x.contiguous().float()
    N, _, H, W = x.shape
    code, param, alpha, apc = Fn.act_of(act)
    desc = Fn.make_desc(N, H, W, c2d.in_channels, c2d.out_channels, c2d.kernel_size[0], stride=c2d.stride[0],
                        pad=c2d.padding[0], in_layout=L.NCHW, out_layout=L.NCHW, compute=L.COMPUTE_F32,
                        act=code, act_param=param, alpha_per_channel=apc, shuffle=shuffle)

    def build():
        wp = Fn.pack_weights(desc, c2d.weight)
        scale, shift = Fn.fold_bn(c2d.bias, bn, c2d.out_channels, x.device)
        return wp, scale, shift

    wp, scale, shift = cache.get(owner, ("w", shuffle), build)
    if residual is not None:
        residual = residual.contiguous().float()
    return Fn.conv2d_fused(desc, x, wp, scale, shift, alpha.detach() if alpha is not None else None, residual)


class conv(nn.Module):
    """`conv` (P/ops.py:18-35): _conv [+ BatchNorm2d] [+ act], one fused launch."""

    def __init__(self, in_channel, out_channel, kernel_size, BN=False, act=None, stride=1, bias=True):
        super().__init__()
        m = [_conv(in_channels=in_channel, out_channels=out_channel, kernel_size=kernel_size, stride=stride,
                   padding=kernel_size // 2, bias=True)]
        if BN:
            m.append(nn.BatchNorm2d(num_features=out_channel))
        if act is not None:
            m.append(act)
        self.body = nn.Sequential(*m)
        self._cache = _FusedCache()

    def forward(self, x, _residual=None):
        c2d, bn, act = conv_parts(self)
        if _guard(x, self):
            return AG.fused_conv(x, c2d, bn, act, residual=_residual)
        return _run_block(self, self._cache, c2d, bn, act, x, residual=_residual)


class ResBlock(nn.Module):
    """`ResBlock` (P/ops.py:37-48): conv-BN-act, conv-BN, += x; the add is fused into the 2nd launch."""

    def __init__(self, channels, kernel_size, act=nn.ReLU(inplace=True), bias=True):
        super().__init__()
        self.body = nn.Sequential(conv(channels, channels, kernel_size, BN=True, act=act),
                                  conv(channels, channels, kernel_size, BN=True, act=None))

    def forward(self, x):
        Fn.require_cuda(x)
        return self.body[1](self.body[0](x), _residual=x)


class BasicBlock(nn.Module):
    """`BasicBlock` (P/ops.py:50-68) - never instantiated by the reference; kept for API completeness."""

    def __init__(self, in_channels, out_channels, kernel_size, num_res_block, act=nn.ReLU(inplace=True)):
        super().__init__()
        self.conv = conv(in_channels, out_channels, kernel_size, BN=False, act=act)
        m = [ResBlock(out_channels, kernel_size, act) for _ in range(num_res_block)]
        m.append(conv(out_channels, out_channels, kernel_size, BN=True, act=None))
        self.body = nn.Sequential(*m)

    def forward(self, x):
        Fn.require_cuda(x)
        res = self.conv(x)
        out = res
        for blk in list(self.body)[:-1]:
            out = blk(out)
        return self.body[-1](out, _residual=res)


class Upsampler(nn.Module):
    """`Upsampler` (P/ops.py:70-84): conv(C -> C r^2) -> PixelShuffle(r) -> act.  The shuffle is the conv's
    store addressing; the activation commutes with it (element-wise) and runs in the epilogue."""

    def __init__(self, channel, kernel_size, scale, act=nn.ReLU(inplace=True)):
        super().__init__()
        m = [conv(channel, channel * scale * scale, kernel_size), nn.PixelShuffle(scale)]
        if act is not None:
            m.append(act)
        self.body = nn.Sequential(*m)
        self._cache = _FusedCache()

    def forward(self, x):
        c2d = self.body[0].body[0]
        act = self.body[2] if len(self.body) > 2 else None
        if _guard(x, self):
            return AG.fused_conv(x, c2d, None, act, shuffle=self.body[1].upscale_factor)
        if isinstance(act, nn.PReLU) and act.weight.numel() > 1:
            raise NotImplementedError("srmi: per-channel PReLU after PixelShuffle")
        return _run_block(self, self._cache, c2d, None, act, x, shuffle=self.body[1].upscale_factor)


class discrim_block(nn.Module):
    """`discrim_block` (P/ops.py:86-96): conv-BN-act, conv(stride 2)-BN-act."""

    def __init__(self, in_feats, out_feats, kernel_size, act=nn.LeakyReLU(inplace=True)):
        super().__init__()
        self.body = nn.Sequential(conv(in_feats, out_feats, kernel_size, BN=True, act=act),
                                  conv(out_feats, out_feats, kernel_size, BN=True, act=act, stride=2))

    def forward(self, x):
        Fn.require_cuda(x)
        return self.body[1](self.body[0](x))
