"""Loss reductions of the SRGAN step on the HIP kernels (csrc/metrics.hip).

Mirrors the reference's call sites: ``nn.MSELoss`` (P/mode.py:89, :169), ``nn.BCELoss``
(P/mode.py:136), ``TVLoss`` (P/losses.py:41-59).  Results stay on the device as 0-d float32 tensors, no
host synchronisation; MSE and BCE are differentiable (HIP backward kernels).  No CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from .functional import require_cuda, stream_ptr

_WS = {}


def _workspace(device):
    ws = _WS.get(device)
    if ws is None:
        ws = _WS[device] = torch.empty(L.REDUCE_WS_BYTES // 8, dtype=torch.float64, device=device)
    return ws


def _f32c(t, what):
    require_cuda(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f"srmi: {what} must be float32, got {t.dtype}")
    return t.contiguous()


def _same_storage_order(a, b):
    """Both operands as dense tensors in ONE common element order (a mean over elements does not care
    which): 4-D tensors go to NHWC storage (what the training path produces), others to row-major."""
    if a.dim() == 4:
        from .autograd import nhwc
        for t, w in ((a, "input"), (b, "target")):
            _f32c_check(t, w)
        return nhwc(a), nhwc(b)
    return _f32c(a, "input"), _f32c(b, "target")


def _f32c_check(t, what):
    require_cuda(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f"srmi: {what} must be float32, got {t.dtype}")


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _same_storage_order(a, b)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        L.check(L.load().srmi_mse_loss(a.data_ptr(), b.data_ptr(), a.numel(), out.data_ptr(),
                                       _workspace(a.device).data_ptr(), stream_ptr()))
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            L.check(L.load().srmi_mse_loss_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), a.numel(),
                                               da.data_ptr() if da is not None else None,
                                               db.data_ptr() if db is not None else None, stream_ptr()))
        return da, db


class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t):
        p, t = _f32c(p, "input"), _f32c(t, "target")
        out = torch.empty((), dtype=torch.float32, device=p.device)
        L.check(L.load().srmi_bce_loss(p.data_ptr(), t.data_ptr(), p.numel(), out.data_ptr(),
                                       _workspace(p.device).data_ptr(), stream_ptr()))
        ctx.save_for_backward(p, t)
        return out

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        dp = torch.empty_like(p)
        L.check(L.load().srmi_bce_loss_bwd(p.data_ptr(), t.data_ptr(), g.contiguous().data_ptr(), p.numel(),
                                           dp.data_ptr(), stream_ptr()))
        return dp, None


def mse_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``nn.MSELoss()(a, b)`` (mean reduction), differentiable w.r.t. both operands."""
    if a.shape != b.shape:
        raise ValueError(f"srmi.mse_loss: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    return _MSEFn.apply(a, b)


def bce_loss(p: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """``nn.BCELoss()(p, t)`` (mean reduction, logs clamped at -100 like torch), differentiable w.r.t. p."""
    if p.shape != t.shape:
        raise ValueError(f"srmi.bce_loss: shapes differ {tuple(p.shape)} vs {tuple(t.shape)}")
    return _BCEFn.apply(p, t)


class MSELoss(nn.Module):
    def forward(self, a, b):
        return mse_loss(a, b)


class BCELoss(nn.Module):
    def forward(self, p, t):
        return bce_loss(p, t)


class _TVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        x = _f32c(x, "input")           # the kernels index x as dense NCHW
        N, C_, H, W = x.shape
        out = torch.empty((), dtype=torch.float32, device=x.device)
        L.check(L.load().srmi_tv_loss(x.data_ptr(), N, C_, H, W, weight, out.data_ptr(),
                                      _workspace(x.device).data_ptr(), stream_ptr()))
        ctx.save_for_backward(x)
        ctx.weight = weight
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        N, C_, H, W = x.shape
        dx = torch.empty_like(x)
        L.check(L.load().srmi_tv_loss_bwd(x.data_ptr(), g.contiguous().data_ptr(), N, C_, H, W, ctx.weight,
                                          dx.data_ptr(), stream_ptr()))
        return dx, None


class TVLoss(nn.Module):
    """Same constructor and call as the reference's ``TVLoss`` (P/losses.py:41-59); differentiable."""

    def __init__(self, tv_loss_weight=1):
        super().__init__()
        self.tv_loss_weight = tv_loss_weight

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError("TVLoss expects an NCHW tensor")
        return _TVFn.apply(x, float(self.tv_loss_weight))

    @staticmethod
    def tensor_size(t):
        return t.size()[1] * t.size()[2] * t.size()[3]


class MeanShift(nn.Conv2d):
    """`MeanShift` (P/losses.py:6-19): frozen 1x1 conv, weight eye/std, bias sign*rgb_range*mean/std."""

    def __init__(self, rgb_range=1, norm_mean=(0.485, 0.456, 0.406), norm_std=(0.229, 0.224, 0.225), sign=-1):
        super().__init__(3, 3, kernel_size=1)
        std = torch.Tensor(norm_std)
        self.weight.data = torch.eye(3).view(3, 3, 1, 1) / std.view(3, 1, 1, 1)
        self.bias.data = sign * rgb_range * torch.Tensor(norm_mean) / std
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        from .autograd import fused_conv
        require_cuda(x)
        return fused_conv(x, self, None, None)


class perceptual_loss(nn.Module):
    """`perceptual_loss` (P/losses.py:21-39): MeanShift -> vgg -> MSE at `layer`; returns
    (loss, hr_feat, sr_feat).  HR and SR are expected in [0,1]."""

    def __init__(self, vgg):
        super().__init__()
        self.normalization_mean = [0.485, 0.456, 0.406]
        self.normalization_std = [0.229, 0.224, 0.225]
        self.transform = MeanShift(norm_mean=self.normalization_mean, norm_std=self.normalization_std)
        self.vgg = vgg
        self.criterion = MSELoss()

    def forward(self, HR, SR, layer="relu5_4"):
        dev = HR.device
        if self.transform.weight.device != dev:
            self.transform.to(dev)
        hr = self.transform(HR)
        sr = self.transform(SR)
        hr_feat = getattr(self.vgg(hr), layer)
        sr_feat = getattr(self.vgg(sr), layer)
        return self.criterion(hr_feat, sr_feat), hr_feat, sr_feat
