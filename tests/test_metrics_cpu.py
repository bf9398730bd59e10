"""CPU: the metrics oracle (PARITY UNPINNED w.r.t. skimage, see oracle/metrics_oracle.py) against an
independent brute-force restatement of the same published formulas, and known-answer cases."""
import numpy as np

from oracle import metrics_oracle as M


def _ssim_brute(x, y, win):
    H, W = x.shape
    NP = win * win
    cn = NP / (NP - 1.0)
    C1, C2 = 1e-4, 9e-4
    vals = []
    for i in range(H - win + 1):
        for j in range(W - win + 1):
            p, q = x[i:i + win, j:j + win].ravel(), y[i:i + win, j:j + win].ravel()
            ux, uy = p.mean(), q.mean()
            vx, vy, vxy = cn * ((p * p).mean() - ux * ux), cn * ((q * q).mean() - uy * uy), cn * ((p * q).mean() - ux * uy)
            vals.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
    return float(np.mean(vals))


def test_luma_known_answers():
    assert abs(M.rgb2y(np.zeros((1, 1, 3)))[0, 0] - 16.0) < 1e-12
    assert abs(M.rgb2y(np.ones((1, 1, 3)))[0, 0] - 235.0) < 1e-9           # 16 + 65.481 + 128.553 + 24.966


def test_psnr_known_answer():
    a = np.zeros((4, 4)); b = np.full((4, 4), 0.1)
    assert abs(M.psnr(a, b) - 20.0) < 1e-9
    assert M.psnr(a, a) == float("inf")


def test_ssim_matches_brute_force_and_identity():
    rng = np.random.default_rng(0)
    x = rng.uniform(16, 235, (19, 23))
    y = x + rng.normal(0, 9, x.shape)
    for win in (3, 5, 7):
        assert abs(M.ssim(x, y, win) - _ssim_brute(x, y, win)) < 1e-10
    assert abs(M.ssim(x, x, 7) - 1.0) < 1e-12


def test_calculate_metrics_window_rule():
    rng = np.random.default_rng(1)
    gt = rng.uniform(-1, 1, (1, 3, 12, 14))
    out = gt + rng.normal(0, 0.05, gt.shape)
    p, s = M.calculate_metrics(gt, out, scale=4)             # cropped to 4 x 6 -> win 3 (4 is even -> 3)
    assert np.isfinite(p) and 0 < s < 1
    p2, _ = M.calculate_metrics(gt, out * 3, scale=4, clip=True)
    assert p2 < p
