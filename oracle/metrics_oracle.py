"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the Y-channel PSNR / SSIM of P/mode.py:24-45 and
P/mode.py:353-373.

PARITY UNPINNED: the reference delegates to scikit-image (`rgb2ycbcr`, `peak_signal_noise_ratio`,
`structural_similarity`), which is not installed here and whose outputs the reference holds no
fixture for.  This file restates skimage's published formulas in float64 NumPy:
  * rgb2ycbcr: Y = 16 + 65.481 R + 128.553 G + 24.966 B for RGB in [0,1] (ITU-R BT.601);
  * PSNR = 10 log10(data_range^2 / mse);
  * SSIM (Wang et al. 2004 as implemented by skimage): uniform win x win filter
    (scipy.ndimage.uniform_filter, the primitive skimage itself calls), sample covariance
    (cov_norm = NP/(NP-1)), K1 = 0.01, K2 = 0.03, mean over the map cropped by (win-1)//2.
"""
import numpy as np
from scipy.ndimage import uniform_filter


def rgb2y(img_hwc):
    img = np.asarray(img_hwc, dtype=np.float64)
    return 16.0 + 65.481 * img[..., 0] + 128.553 * img[..., 1] + 24.966 * img[..., 2]


def psnr(a, b, data_range=1.0):
    mse = np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)
    return float(10.0 * np.log10(data_range ** 2 / mse)) if mse > 0 else float("inf")


def ssim(x, y, win_size=7, data_range=1.0):
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    NP = win_size ** 2
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
    uxx, uyy, uxy = (uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size),
                     uniform_filter(x * y, size=win_size))
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return float(S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean())


def calculate_metrics(gt_nchw, out_nchw, scale=1, clip=False):
    """(psnr, ssim) following P/mode.py:24-45 on image 0 (clip=True: the test() variant, :360)."""
    out = np.asarray(out_nchw[0], np.float64)
    gt = np.asarray(gt_nchw[0], np.float64)[:, :out.shape[1], :out.shape[2]]
    if clip:
        out = np.clip(out, -1.0, 1.0)
    out = ((out + 1.0) / 2.0).transpose(1, 2, 0)
    gt = ((gt + 1.0) / 2.0).transpose(1, 2, 0)
    yo = rgb2y(out)[scale:-scale, scale:-scale]
    yg = rgb2y(gt)[scale:-scale, scale:-scale]
    p = psnr(yo / 255.0, yg / 255.0)
    win = min(7, min(yo.shape))
    if win % 2 == 0:
        win -= 1
    return p, ssim(yo, yg, win_size=win, data_range=1.0)
