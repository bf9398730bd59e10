"""Thin tensor-level wrappers over the C ABI: descriptor construction, weight packing, one fused conv.

All tensors are CUDA (ROCm) tensors; nothing here computes on the host and nothing falls back.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L

_TORCH_DTYPE = {L.F32: torch.float32, L.BF16: torch.bfloat16}


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return L.F32
    if dt == torch.bfloat16:
        return L.BF16
    raise TypeError(f"srmi supports float32 and bfloat16 storage, got {dt}")


def torch_dtype(code: int) -> torch.dtype:
    return _TORCH_DTYPE[code]


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream on `device` (default: the current device).  The raw accessor costs
    ~0.3 us; torch.cuda.current_stream() builds a Stream object (~7 us, several per fused op of the training step)."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        idx = device.index if device is not None and device.index is not None else _GET_DEVICE()
        return _RAW_STREAM(idx)
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t: torch.Tensor, what="input"):
    if not t.is_cuda:
        raise RuntimeError(
            f"srmi: {what} is on {t.device}; the HIP kernels need a ROCm device tensor "
            "(there is deliberately no CPU fallback)")


class AddReLU(nn.Module):
    """Marker for the one epilogue whose activation follows the residual: relu(conv + residual), the
    `add([shortcut, conv])` -> ReLU of res_conv_block (Res-U-SRGAN/untitled0.py:67-72).  Only meaningful inside a plan."""


def act_of(module):
    """Map the activation module the reference passes around (P/srgan_model.py:8,43; P/ops.py:28-29)
    to (act code, param, alpha tensor or None, per_channel)."""
    if module is None:
        return L.ACT_NONE, 0.0, None, 0
    if isinstance(module, nn.PReLU):
        return L.ACT_PRELU, 0.0, module.weight, int(module.weight.numel() > 1)
    if isinstance(module, nn.LeakyReLU):
        return L.ACT_LRELU, float(module.negative_slope), None, 0
    if isinstance(module, nn.ReLU):
        return L.ACT_RELU, 0.0, None, 0
    if isinstance(module, nn.Tanh):
        return L.ACT_TANH, 0.0, None, 0
    if isinstance(module, nn.Sigmoid):
        return L.ACT_SIGMOID, 0.0, None, 0
    if isinstance(module, AddReLU):
        return L.ACT_ADD_RELU, 0.0, None, 0
    raise NotImplementedError(f"srmi: activation {type(module).__name__} has no fused epilogue")


def make_desc(N, H, W, Cin, Cout, k, stride=1, pad=None, in_dtype=L.F32, out_dtype=L.F32, res_dtype=L.F32,
              in_layout=L.NHWC, out_layout=L.NHWC, compute=L.COMPUTE_F32, act=L.ACT_NONE, act_param=0.0,
              alpha_per_channel=0, shuffle=1, shuffle_order=L.SHUFFLE_CRD, in_cstride=0, out_cstride=0,
              out_coffset=0, res_cstride=0) -> L.ConvDesc:
    kh, kw = (k, k) if isinstance(k, int) else k
    d = L.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout = N, H, W, Cin, Cout
    d.kh, d.kw, d.stride = kh, kw, stride
    d.pad = kh // 2 if pad is None else pad
    d.in_dtype, d.out_dtype, d.res_dtype = in_dtype, out_dtype, res_dtype
    d.in_layout, d.out_layout, d.compute = in_layout, out_layout, compute
    d.act, d.act_param, d.alpha_per_channel = act, act_param, alpha_per_channel
    d.shuffle, d.shuffle_order = shuffle, shuffle_order
    d.in_cstride, d.out_cstride, d.out_coffset = in_cstride, out_cstride, out_coffset
    d.res_cstride = res_cstride
    return d


def pack_weights(desc: L.ConvDesc, w_oihw: torch.Tensor) -> torch.Tensor:
    """OIHW f32 device tensor -> kernel-native packed buffer (uint8 device tensor)."""
    require_cuda(w_oihw, "weight")
    w = w_oihw.detach().to(torch.float32).contiguous()
    nbytes = L.packed_weight_bytes(desc)
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    L.check(L.load().srmi_conv2d_pack_weights(C.byref(desc), w.data_ptr(), out.data_ptr(), stream_ptr()))
    return out


def fold_bn(conv_bias, bn, cout, device):
    """bias + eval-mode BatchNorm -> per-channel (scale, shift), f32 on device.
    BN eval (P/ops.py:25-26): y = (z + b - mean) / sqrt(var + eps) * gamma + beta."""
    b = conv_bias.detach().float() if conv_bias is not None else torch.zeros(cout, device=device)
    if bn is None:
        return torch.ones(cout, device=device, dtype=torch.float32), b.contiguous()
    scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() + (b - bn.running_mean.detach().float()) * scale
    return scale.contiguous(), shift.contiguous()


def out_shape(desc: L.ConvDesc):
    Ho = (desc.H + 2 * desc.pad - desc.kh) // desc.stride + 1
    Wo = (desc.W + 2 * desc.pad - desc.kw) // desc.stride + 1
    r = desc.shuffle
    return desc.N, desc.Cout // (r * r), Ho * r, Wo * r


def fill_args(a: L.ConvArgs, desc, x, w, scale, shift, alpha, residual, y):
    a.d = desc
    a.x = x.data_ptr() if isinstance(x, torch.Tensor) else x
    a.w = w.data_ptr()
    a.scale = scale.data_ptr() if scale is not None else None
    a.shift = shift.data_ptr() if shift is not None else None
    a.alpha = alpha.data_ptr() if alpha is not None else None
    a.residual = residual.data_ptr() if residual is not None else None
    a.y = y.data_ptr()
    a.flags = 0
    a.reserved = 0
    return a


def conv2d_fused(desc, x, w_packed, scale, shift, alpha=None, residual=None, out=None):
    """One fused conv launch.  ``out`` is allocated (dense, in desc.out_layout) when not given."""
    require_cuda(x)
    n, c, h, w = out_shape(desc)
    if out is None:
        shape = (n, c, h, w) if desc.out_layout == L.NCHW else (n, h, w, desc.out_cstride or c)
        out = torch.empty(shape, dtype=torch_dtype(desc.out_dtype), device=x.device)
    a = L.ConvArgs()
    fill_args(a, desc, x, w_packed, scale, shift, alpha, residual, out)
    L.check(L.load().srmi_conv2d_fwd(C.byref(a), stream_ptr()))
    return out


def layout_convert(src, N, Cc, H, W, src_layout, dst_layout, dst_dtype: torch.dtype, out=None):
    require_cuda(src)
    src = src.contiguous()
    if out is None:
        shape = (N, Cc, H, W) if dst_layout == L.NCHW else (N, H, W, Cc)
        out = torch.empty(shape, dtype=dst_dtype, device=src.device)
    L.check(L.load().srmi_layout_convert(src.data_ptr(), out.data_ptr(), N, Cc, H, W, src_layout,
                                         dtype_code(src.dtype), dst_layout, dtype_code(dst_dtype), stream_ptr()))
    return out
