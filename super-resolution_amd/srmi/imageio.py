"""uint8 <-> generator-range conversions on the device, replacing the host numpy arithmetic on either
side of the generator: `(img / 127.5) - 1.0` + HWC->CHW (P/dataset.py:40-47, :75-76) and `(out + 1) / 2`, CHW->HWC,
`(. * 255).astype(np.uint8)` (P/mode.py:360, :397-400).  Moving uint8 instead of float32 over PCIe is 4x less
upload and 4x less download per frame.  Bit-exact with the numpy expressions (tests/test_imageio_gpu.py)."""
from __future__ import annotations

import torch

from . import _lib as L
from .functional import require_cuda, stream_ptr


def from_uint8(img: torch.Tensor) -> torch.Tensor:
    """uint8 [N,H,W,C] (or [H,W,C]) device tensor -> float32 [N,C,H,W] in [-1,1]."""
    require_cuda(img, "image")
    if img.dtype != torch.uint8:
        raise TypeError(f"from_uint8 expects uint8, got {img.dtype}")
    if img.dim() == 3:
        img = img.unsqueeze(0)
    img = img.contiguous()
    N, H, W, Cc = img.shape
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=img.device)
    L.check(L.load().srmi_u8hwc_to_f32nchw(img.data_ptr(), out.data_ptr(), N, H, W, Cc, stream_ptr()))
    return out


def to_uint8(x: torch.Tensor, clip: bool = False) -> torch.Tensor:
    """float32 [N,C,H,W] generator output -> uint8 [N,H,W,C]; clip=True is `test()`'s np.clip (P/mode.py:360)."""
    require_cuda(x, "output")
    if x.dtype != torch.float32 or x.dim() != 4:
        raise TypeError("to_uint8 expects a float32 NCHW tensor")
    x = x.contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty((N, H, W, Cc), dtype=torch.uint8, device=x.device)
    L.check(L.load().srmi_f32nchw_to_u8hwc(x.data_ptr(), out.data_ptr(), N, H, W, Cc, int(clip), stream_ptr()))
    return out
