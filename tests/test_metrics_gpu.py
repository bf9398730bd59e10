"""GPU: loss / metric reduction kernels (csrc/metrics.hip) through the C ABI against the oracle.
TVLoss is pinned to the reference's own output (tests/golden/losses.npz); MSE/BCE against torch's CPU
implementation (the reference's nn.MSELoss / nn.BCELoss); PSNR/SSIM against oracle/metrics_oracle.py
(PARITY UNPINNED w.r.t. skimage)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as M
from oracle import srgan_oracle as O
from srmi import losses, metrics

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_tv_loss_matches_reference_vector(golden_dir):
    g = np.load(os.path.join(golden_dir, "losses.npz"), allow_pickle=False)
    got = losses.TVLoss()(torch.from_numpy(g["tv_x"]).to(DEV)).item()
    assert abs(got - float(g["tv"])) <= 1e-6 * max(1.0, abs(float(g["tv"])))
    assert losses.TVLoss.tensor_size(torch.zeros(2, 3, 4, 5)) == 60


@pytest.mark.parametrize("shape,weight", [((16, 512, 6, 6), 1), ((1, 3, 17, 33), 0.5), ((2, 1, 1, 9), 2.0)])
def test_tv_loss_shapes(shape, weight):
    x = torch.randn(shape, generator=torch.Generator().manual_seed(3))
    ref = (O.tv_loss(x.double()) * weight).item()
    got = losses.TVLoss(weight)(x.to(DEV)).item()
    if np.isnan(ref):            # H == 1: 0/0 in the reference
        assert np.isnan(got)
    else:
        assert abs(got - ref) <= 2e-6 * max(1.0, abs(ref))


@pytest.mark.parametrize("n", [1, 7, 1024, 16 * 3 * 96 * 96 + 3, 16 * 3 * 384 * 384])
def test_mse_loss(n):
    g = torch.Generator().manual_seed(n)
    a, b = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = torch.nn.functional.mse_loss(a.double(), b.double()).item()
    got = losses.mse_loss(a.to(DEV), b.to(DEV))
    assert got.shape == () and got.dtype == torch.float32
    assert abs(got.item() - ref) <= 1e-6 * abs(ref)
    # unaligned views take the scalar path
    if n > 8:
        got2 = losses.MSELoss()(a.to(DEV)[1:], b.to(DEV)[1:]).item()
        assert abs(got2 - torch.nn.functional.mse_loss(a.double()[1:], b.double()[1:]).item()) <= 1e-6 * abs(ref)
    # deterministic: same bits on a second run
    assert losses.mse_loss(a.to(DEV), b.to(DEV)).item() == got.item()


def test_bce_loss_including_saturation():
    g = torch.Generator().manual_seed(5)
    p = torch.rand(16, 1, generator=g)
    p[0], p[1] = 0.0, 1.0                       # log clamps at -100 (torch semantics)
    for t in (torch.ones_like(p), torch.zeros_like(p), torch.rand(16, 1, generator=g)):
        ref = torch.nn.functional.binary_cross_entropy(p, t).item()
        got = losses.BCELoss()(p.to(DEV), t.to(DEV)).item()
        assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref))


def test_loss_guards():
    with pytest.raises(RuntimeError):
        losses.mse_loss(torch.zeros(4), torch.zeros(4))
    with pytest.raises(ValueError):
        losses.mse_loss(torch.zeros(4, device=DEV), torch.zeros(5, device=DEV))
    with pytest.raises(TypeError):
        losses.mse_loss(torch.zeros(4, device=DEV, dtype=torch.float64), torch.zeros(4, device=DEV, dtype=torch.float64))


@pytest.mark.parametrize("H,W,scale,gt_extra", [(96, 96, 4, (0, 0)), (50, 71, 2, (3, 5)), (24, 20, 4, (0, 0)),
                                                (384, 384, 4, (0, 0)), (13, 13, 4, (0, 0))])
def test_y_psnr_ssim(H, W, scale, gt_extra):
    g = torch.Generator().manual_seed(H * 1000 + W)
    gt = torch.rand(2, 3, H + gt_extra[0], W + gt_extra[1], generator=g) * 2 - 1
    out = (gt[:, :, :H, :W] + 0.08 * torch.randn(2, 3, H, W, generator=g))
    p_ref, s_ref = M.calculate_metrics(gt.numpy(), out.numpy(), scale=scale)
    # the reference's 4-tuple (P/mode.py:67, unpacked at :117 and :191); LPIPS / DISTS columns are NaN
    p, s, lp, di = metrics.calculate_metrics(gt.to(DEV), out.to(DEV), None, scale=scale)
    assert np.isnan(lp) and np.isnan(di)
    assert abs(p - p_ref) <= 1e-4, (p, p_ref)          # float32 inputs, float64 arithmetic on both sides
    assert abs(s - s_ref) <= 1e-6, (s, s_ref)
    # test(): clipped output, PSNR only
    big = out * 1.5
    pc_ref, _ = M.calculate_metrics(gt.numpy(), big.numpy(), scale=scale, clip=True)
    assert abs(metrics.test_psnr(gt.to(DEV), big.to(DEV), scale) - pc_ref) <= 1e-4


def test_metric_properties_and_guards():
    g = torch.Generator().manual_seed(9)
    x = (torch.rand(1, 3, 64, 48, generator=g) * 2 - 1).to(DEV)
    r = metrics.y_psnr_ssim(x, x, scale=4)
    assert r[0].item() == 0.0 and np.isinf(r[1].item()) and abs(r[2].item() - 1.0) < 1e-12
    y = x + 0.1
    a, b = metrics.calculate_metrics(x, y, None, 4), metrics.calculate_metrics(y, x, None, 4)
    assert abs(a[0] - b[0]) < 1e-9 and abs(a[1] - b[1]) < 1e-9        # symmetric
    with pytest.raises(ValueError):
        metrics.y_psnr_ssim(x[:, :, :10, :10], x[:, :, :10, :10], scale=4)   # 2x2 after cropping: no SSIM window
    from srmi import _lib as L
    ws = torch.empty(2048, dtype=torch.float64, device=DEV)
    out = torch.empty(3, dtype=torch.float64, device=DEV)
    rc = L.load().srmi_y_psnr_ssim(x.data_ptr(), x.data_ptr(), 64, 48, 64, 48, 4, 0, 4, out.data_ptr(), ws.data_ptr(), 0)
    assert rc == -1 and b"window" in L.load().srmi_last_error()


def test_batched_metrics_and_averager():
    """every image of a batch scored on the device; running mean over batches of different sizes."""
    g = torch.Generator().manual_seed(31)
    sets = [(3, 40, 56), (2, 64, 33)]
    av = metrics.MetricAverager(scale=2)
    ps, ss = [], []
    for n, H, W in sets:
        gt = torch.rand(n, 3, H, W, generator=g) * 2 - 1
        out = gt + 0.05 * torch.randn(n, 3, H, W, generator=g)
        r = metrics.y_psnr_ssim_batch(gt.to(DEV), out.to(DEV), scale=2)
        assert r.shape == (n, 3) and r.is_cuda and r.dtype == torch.float64
        for i in range(n):
            p_ref, s_ref = M.calculate_metrics(gt[i:i + 1].numpy(), out[i:i + 1].numpy(), scale=2)
            assert abs(float(r[i, 1]) - p_ref) <= 1e-4 and abs(float(r[i, 2]) - s_ref) <= 1e-6
            ps.append(p_ref); ss.append(s_ref)
        # one batched call == per-image calls (same kernels, the workspace slots divided between the images)
        for i in range(n):
            one = metrics.y_psnr_ssim(gt.to(DEV), out.to(DEV), scale=2, index=i)
            assert torch.allclose(one, r[i], rtol=1e-12, atol=0)
        av.update(gt.to(DEV), out.to(DEV))
    p, s = av.result()
    assert abs(p - np.mean(ps)) <= 1e-4 and abs(s - np.mean(ss)) <= 1e-6


def test_batched_metrics_large_batch_and_guard():
    """More images than SRMI_METRICS_MAX_BATCH are split by the wrapper; the C entry point rejects them."""
    from srmi import _lib as L
    g = torch.Generator().manual_seed(5)
    n = L.METRICS_MAX_BATCH + 3
    gt = (torch.rand(n, 3, 12, 14, generator=g) * 2 - 1).to(DEV)
    out = gt + 0.05 * torch.randn(n, 3, 12, 14, generator=g).to(DEV)
    r = metrics.y_psnr_ssim_batch(gt, out, scale=1)
    for i in (0, 17, L.METRICS_MAX_BATCH - 1, L.METRICS_MAX_BATCH, n - 1):
        p_ref, s_ref = M.calculate_metrics(gt[i:i + 1].cpu().numpy(), out[i:i + 1].cpu().numpy(), scale=1)
        assert abs(float(r[i, 1]) - p_ref) <= 1e-4 and abs(float(r[i, 2]) - s_ref) <= 1e-6
    ws = torch.empty(2048, dtype=torch.float64, device=DEV)
    o = torch.empty(3 * n, dtype=torch.float64, device=DEV)
    rc = L.load().srmi_y_psnr_ssim_batch(out.data_ptr(), gt.data_ptr(), n, 12, 14, 12, 14, 1, 0, 7, o.data_ptr(),
                                         ws.data_ptr(), 0)
    assert rc == -1 and b"n_images" in L.load().srmi_last_error()
