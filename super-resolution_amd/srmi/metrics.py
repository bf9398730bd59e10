"""Y-channel PSNR / SSIM on the device (csrc/metrics.hip), mirroring ``calculate_metrics``
(P/mode.py:24-45) and the PSNR of ``test`` (P/mode.py:353-373).

The reference moves each image to the host and calls skimage; here the colour conversion, border
crop and reductions are one pass of HIP kernels over the tensors where they already live.  The LPIPS
and DISTS scores of ``calculate_metrics`` need third-party pretrained networks and are out of scope.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .functional import require_cuda, stream_ptr
from .losses import _workspace


def ssim_window(h, w, scale):
    """win_size rule of P/mode.py:40-43 on the cropped luma plane."""
    win = min(7, min(h - 2 * scale, w - 2 * scale))
    if win % 2 == 0:
        win -= 1
    return win


def y_psnr_ssim(gt: torch.Tensor, output: torch.Tensor, scale: int = 1, clip: bool = False, ssim: bool = True, index: int = 0):
    """(psnr, ssim) of image `index` of the batch (the reference scores image 0, P/mode.py:31-32) as a 3-element
    float64 DEVICE tensor [mse, psnr, ssim].

    gt may be larger than output; its top-left corner is used (P/mode.py:353)."""
    require_cuda(gt, "gt")
    require_cuda(output, "output")
    if output.dim() != 4 or gt.dim() != 4 or output.shape[1] != 3 or gt.shape[1] != 3:
        raise ValueError("y_psnr_ssim expects Nx3xHxW tensors")
    o = output[index].float().contiguous()
    g = gt[index].float().contiguous()
    H, W = o.shape[1:]
    gH, gW = g.shape[1:]
    if scale < 1:
        raise ValueError("scale must be >= 1 (the reference slices [scale:-scale])")
    win = ssim_window(H, W, scale) if ssim else 0
    if ssim and win < 3:
        raise ValueError(f"image {H}x{W} too small for SSIM after cropping {scale} px")
    res = torch.empty(3, dtype=torch.float64, device=o.device)
    L.check(L.load().srmi_y_psnr_ssim(o.data_ptr(), g.data_ptr(), H, W, gH, gW, scale, int(clip), win,
                                      res.data_ptr(), _workspace(o.device).data_ptr(), stream_ptr()))
    return res


def calculate_metrics(gt, output, device=None, scale=1):
    """Drop-in for ``calculate_metrics`` (P/mode.py:24-67): returns the reference's 4-tuple
    ``(psnr, ssim, lpips_score, dists_score)`` so that the loops' ``psnr, ssim, lpips_score, dists_score = ...``
    (P/mode.py:117, :191) unpack unchanged.  PSNR and SSIM are computed on the device (one device->host copy of
    24 bytes); LPIPS and DISTS need third-party pretrained networks that are out of scope: their columns are NaN
    ("columns optional/NaN")."""
    r = y_psnr_ssim(gt, output, scale=scale, clip=False, ssim=True).cpu()
    return float(r[1]), float(r[2]), float("nan"), float("nan")


def test_psnr(gt, output, scale):
    """PSNR of ``test`` (P/mode.py:353-373): output clipped to [-1,1], gt cropped to output's size."""
    return float(y_psnr_ssim(gt, output, scale=scale, clip=True, ssim=False)[1])


def y_psnr_ssim_batch(gt: torch.Tensor, output: torch.Tensor, scale: int = 1, clip: bool = False, ssim: bool = True):
    """[N, 3] float64 DEVICE tensor of [mse, psnr, ssim] for EVERY image of the batch: the reference scores
    one image per call on the host; here the whole batch is scored where it lives by three launches
    (``srmi_y_psnr_ssim_batch``), with no host synchronisation."""
    require_cuda(gt, "gt")
    require_cuda(output, "output")
    if output.dim() != 4 or gt.dim() != 4 or output.shape[1] != 3 or gt.shape[1] != 3:
        raise ValueError("y_psnr_ssim_batch expects Nx3xHxW tensors")
    n = min(gt.shape[0], output.shape[0])
    o = output[:n].float().contiguous()
    g = gt[:n].float().contiguous()
    H, W = o.shape[2:]
    gH, gW = g.shape[2:]
    if scale < 1:
        raise ValueError("scale must be >= 1 (the reference slices [scale:-scale])")
    win = ssim_window(H, W, scale) if ssim else 0
    if ssim and win < 3:
        raise ValueError(f"image {H}x{W} too small for SSIM after cropping {scale} px")
    res = torch.empty((n, 3), dtype=torch.float64, device=o.device)
    ws = _workspace(o.device).data_ptr()
    lib = L.load()
    for i0 in range(0, n, L.METRICS_MAX_BATCH):
        nb = min(L.METRICS_MAX_BATCH, n - i0)
        L.check(lib.srmi_y_psnr_ssim_batch(o[i0:].data_ptr(), g[i0:].data_ptr(), nb, H, W, gH, gW, scale, int(clip), win,
                                           res[i0:].data_ptr(), ws, stream_ptr()))
    return res


class MetricAverager:
    """Running mean of Y-PSNR / SSIM over a validation set, kept on the device: `update(gt, output)` per batch (any
    image sizes from batch to batch), ONE 16-byte device->host copy in `result()`."""

    def __init__(self, scale: int = 1, clip: bool = False):
        self.scale, self.clip = scale, clip
        self.sum = None
        self.count = 0

    def update(self, gt, output):
        r = y_psnr_ssim_batch(gt, output, scale=self.scale, clip=self.clip)[:, 1:].sum(dim=0)
        self.sum = r if self.sum is None else self.sum + r
        self.count += min(gt.shape[0], output.shape[0])

    def result(self):
        if self.count == 0:
            return float("nan"), float("nan")
        m = (self.sum / self.count).cpu()
        return float(m[0]), float(m[1])
