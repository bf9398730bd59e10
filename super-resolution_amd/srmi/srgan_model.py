"""`Generator` / `Discriminator` with the reference's constructor signatures, module tree and
state_dict keys (P/srgan_model.py:6-73), running on libsrmi's gfx950 kernels.

    from srmi.srgan_model import Generator
    g = Generator().cuda().eval(); g.load_state_dict(torch.load(path))
    with torch.no_grad():
        sr, feat = g(lr)            # NCHW f32 in [-1,1] -> (NCHW f32 tanh image, 64-ch feature)

`forward` returns the tuple the reference returns (P/srgan_model.py:39).  The compute path is chosen
per module: ``set_compute("fp32")`` (default; exact-f32 MFMA, <=1e-4 of the reference) or
``set_compute("bf16")`` (bf16 MFMA / bf16 NHWC activations: the throughput path).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib as L
from . import autograd as AG
from .engine import DiscriminatorPlan, GeneratorPlan, PlanCache
from .ops import Upsampler, ResBlock, _guard, conv, discrim_block

_COMPUTE = {"fp32": L.COMPUTE_F32, "f32": L.COMPUTE_F32, "bf16": L.COMPUTE_BF16}


def _default_compute():
    return os.environ.get("SRMI_COMPUTE", "fp32")


class _ComputeMixin:
    def set_compute(self, mode: str):
        if mode not in _COMPUTE:
            raise ValueError(f"compute mode must be one of {sorted(_COMPUTE)}, got {mode!r}")
        self._compute = mode
        return self

    @property
    def compute(self):
        return getattr(self, "_compute", None) or _default_compute()


class Generator(nn.Module, _ComputeMixin):
    # `act = nn.PReLU()` is evaluated once, at class-definition time, exactly like the reference
    # (P/srgan_model.py:8): ONE PReLU parameter shared by every layer and every Generator instance.
    def __init__(self, img_feat=3, n_feats=64, kernel_size=3, num_block=16, act=nn.PReLU(), scale=4):
        super().__init__()
        self.conv01 = conv(in_channel=img_feat, out_channel=n_feats, kernel_size=9, BN=False, act=act)
        self.body = nn.Sequential(*[ResBlock(channels=n_feats, kernel_size=3, act=act) for _ in range(num_block)])
        self.conv02 = conv(in_channel=n_feats, out_channel=n_feats, kernel_size=3, BN=True, act=None)
        if scale == 4:
            ups = [Upsampler(channel=n_feats, kernel_size=3, scale=2, act=act) for _ in range(2)]
        else:
            ups = [Upsampler(channel=n_feats, kernel_size=3, scale=scale, act=act)]
        self.tail = nn.Sequential(*ups)
        self.last_conv = conv(in_channel=n_feats, out_channel=img_feat, kernel_size=3, BN=False, act=nn.Tanh())
        self._plans = PlanCache(GeneratorPlan)

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError(f"Generator expects NCHW input, got shape {tuple(x.shape)}")
        if _guard(x, self):
            # training step (P/mode.py:101, :149, :164): the reference's forward, block by block, on the
            # fp32 training kernels; tensors are logical NCHW with channels_last (NHWC) storage
            x = self.conv01(x)
            _skip_connection = x
            x = self.body(x)
            feat = self.conv02(x, _residual=_skip_connection)
            x = self.tail(feat)
            x = self.last_conv(x)
            return x, feat
        N, _, H, W = x.shape
        plan = self._plans.get(self, N, H, W, _COMPUTE[self.compute])
        return plan.run(x)


class Discriminator(nn.Module, _ComputeMixin):
    def __init__(self, img_feat=3, n_feats=64, kernel_size=3, act=nn.LeakyReLU(inplace=True), num_of_block=3,
                 patch_size=96):
        super().__init__()
        self.act = act
        self.conv01 = conv(in_channel=img_feat, out_channel=n_feats, kernel_size=3, BN=False, act=self.act)
        self.conv02 = conv(in_channel=n_feats, out_channel=n_feats, kernel_size=3, BN=False, act=self.act, stride=2)
        self.body = nn.Sequential(*[discrim_block(in_feats=n_feats * (2 ** i), out_feats=n_feats * (2 ** (i + 1)),
                                                  kernel_size=3, act=self.act) for i in range(num_of_block)])
        self.linear_size = ((patch_size // (2 ** (num_of_block + 1))) ** 2) * (n_feats * (2 ** num_of_block))
        self.tail = nn.Sequential(nn.Linear(self.linear_size, 1024), self.act, nn.Linear(1024, 1), nn.Sigmoid())
        self._plans = PlanCache(DiscriminatorPlan)

    def forward(self, x):
        if _guard(x, self):
            x = self.conv01(x)
            x = self.conv02(x)
            x = self.body(x)
            x = AG.linear(x, self.tail[0], self.tail[1])       # flatten (NCHW order) + Linear + act
            x = AG.linear(x, self.tail[2], self.tail[3])       # Linear + Sigmoid
            return x.reshape(x.size(0), -1)
        N, _, H, W = x.shape
        plan = self._plans.get(self, N, H, W, _COMPUTE[self.compute])
        return plan.run(x)
