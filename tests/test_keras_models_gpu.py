"""GPU: the Keras-derived generators (SRCNN, ESRGAN-style RRDB) against oracle/keras_oracle.py.
PARITY UNPINNED w.r.t. the reference (TensorFlow absent, no stored outputs - SURVEY 8c c5): the oracle is a
restatement of the Keras call sites; these tests pin the HIP path to it."""
import numpy as np
import pytest
import torch

from oracle import keras_oracle as K
from oracle import weights as W
from srmi.keras_models import SRCNN, AutoencoderSRGAN, KerasSRGAN, ResUNetSRGAN, RRDBNet, UNetSRGAN, from_keras_hwio

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _randomise(m, seed):
    W.randomise_module(m, seed)


def _psnr01(a, b):
    return 10 * np.log10(1.0 / (float(((a.double() - b.double()) ** 2).mean()) / 4.0))


def test_param_counts_match_reference_summary():
    assert sum(p.numel() for p in SRCNN().parameters()) == 10_496 + 73_792 + 1_601
    m = RRDBNet(num_blocks=16)
    per_block = (64 * 32 + 96 * 32 + 128 * 32) * 9 + 3 * 32 + 160 * 64 * 9 + 64
    assert sum(p.numel() for p in m.blocks[0].parameters()) == per_block
    assert from_keras_hwio(torch.zeros(3, 3, 64, 32)).shape == (32, 64, 3, 3)


@pytest.mark.parametrize("kw,shape", [(dict(), (2, 1, 20, 28)), (dict(k=(9, 1, 5), ch=(64, 32)), (1, 1, 33, 17))])
def test_srcnn_fp32_and_bf16(kw, shape):
    m = SRCNN(**kw)
    _randomise(m, 1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand(shape, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = K.srcnn_forward(sd, x)
        m = m.to(DEV).eval()
        y = m(x.to(DEV)).cpu()
        assert y.shape == ref.shape
        assert (y - ref).abs().max().item() <= 1e-4
        yb = m.set_compute("bf16")(x.to(DEV)).cpu()
    assert (yb - ref).abs().max().item() <= 3e-2


@pytest.mark.parametrize("nb,shape", [(2, (1, 3, 12, 16)), (3, (2, 3, 9, 7)), (0, (1, 3, 8, 8))])
def test_rrdbnet_fp32(nb, shape):
    m = RRDBNet(num_blocks=nb)
    _randomise(m, 3)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand(shape, generator=torch.Generator().manual_seed(4)) * 2 - 1
    with torch.no_grad():
        ref = K.rrdbnet_forward(sd, x, num_blocks=nb)
        y = m.to(DEV).eval()(x.to(DEV)).cpu()
    assert y.shape == ref.shape == (shape[0], 3, 4 * shape[2], 4 * shape[3])
    assert (y - ref).abs().max().item() / 2.0 <= 1e-4            # [0,1] pixel scale


def test_rrdbnet_bf16_psnr():
    m = RRDBNet(num_blocks=4)
    _randomise(m, 5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand((2, 3, 24, 40), generator=torch.Generator().manual_seed(6)) * 2 - 1
    with torch.no_grad():
        ref = K.rrdbnet_forward(sd, x, num_blocks=4)
        m = m.to(DEV).eval().set_compute("bf16")
        y = m(x.to(DEV)).cpu()
        plan = m._plans.get(m, 2, 24, 40, 1)
    assert plan.kernels.count("c64_bf16") == 3 and plan.kernels[-1] == "c64_thin_bf16"
    p = _psnr01(y, ref)
    print("RRDB bf16 PSNR vs fp32 oracle: %.1f dB" % p)
    assert p >= 40.0


def _randomise_bn(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(torch.empty(mod.weight.shape).uniform_(0.6, 1.4, generator=g))
                mod.bias.copy_(torch.empty(mod.bias.shape).uniform_(-0.2, 0.2, generator=g))
                mod.running_mean.copy_(torch.empty(mod.bias.shape).uniform_(-0.3, 0.3, generator=g))
                mod.running_var.copy_(torch.empty(mod.bias.shape).uniform_(0.5, 1.5, generator=g))


@pytest.mark.parametrize("nb,shape", [(2, (1, 3, 12, 16)), (3, (2, 3, 7, 100))])
def test_keras_srgan_fp32_and_bf16(nb, shape):
    """SURVEY 8f4 (first variant): the Keras-layout SRGAN generator (typical srgan .py:21-71).  Parity unpinned."""
    m = KerasSRGAN(num_res_blocks=nb)
    _randomise(m, 7)
    _randomise_bn(m, 8)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand(shape, generator=torch.Generator().manual_seed(9)) * 2 - 1
    with torch.no_grad():
        ref = K.keras_srgan_forward(sd, x, num_res_blocks=nb)
        m = m.to(DEV).eval()
        y = m(x.to(DEV)).cpu()
        assert y.shape == ref.shape == (shape[0], 3, 4 * shape[2], 4 * shape[3])
        assert (y - ref).abs().max().item() / 2.0 <= 1e-4
        yb = m.set_compute("bf16")(x.to(DEV)).cpu()
        plan = m._plans.get(m, shape[0], shape[2], shape[3], 1)
    assert plan.kernels[0] == "k9c3_bf16" and plan.kernels.count("c64_bf16") == 2 * nb + 3
    p = _psnr01(yb, ref)
    print("Keras SRGAN bf16 PSNR vs fp32 oracle: %.1f dB" % p)
    assert p >= 35.0


@pytest.mark.parametrize("kw,shape", [(dict(n_feats=64, depth=2, up_feats=64), (1, 3, 8, 12)), (dict(n_feats=64, depth=4), (1, 3, 16, 32))])
def test_unet_srgan_fp32_and_bf16(kw, shape):
    """the U-Net generator (U-SRGAN/srganunet.py:14-126) - pooling, nearest upsampling and concatenation as
    addressing.  Parity unpinned (no TensorFlow): compared with oracle/keras_oracle.py."""
    m = UNetSRGAN(**kw)
    _randomise(m, 11)
    _randomise_bn(m, 12)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand(shape, generator=torch.Generator().manual_seed(13)) * 2 - 1
    with torch.no_grad():
        ref = K.unet_srgan_forward(sd, x, depth=kw["depth"])
        m = m.to(DEV).eval()
        y = m(x.to(DEV)).cpu()
        assert y.shape == ref.shape == (shape[0], 3, 4 * shape[2], 4 * shape[3])
        assert (y - ref).abs().max().item() / 2.0 <= 1e-4
        yb = m.set_compute("bf16")(x.to(DEV)).cpu()
    p = _psnr01(yb, ref)
    print("U-Net SRGAN bf16 PSNR vs fp32 oracle: %.1f dB" % p)
    assert p >= 35.0
    with pytest.raises(ValueError, match="divisible"):
        m(torch.zeros(1, 3, 10, 12, device=DEV))


@pytest.mark.parametrize("kw,shape", [(dict(num_autoencoders=1), (1, 3, 8, 12)), (dict(num_autoencoders=3, num_filters=(64, 128, 256)), (2, 3, 12, 16))])
def test_autoencoder_srgan_fp32_and_bf16(kw, shape):
    """the stacked-autoencoder generator (A-SRGAN/srganautoencoders.py:55-144).  Parity unpinned."""
    m = AutoencoderSRGAN(**kw)
    _randomise(m, 21)
    _randomise_bn(m, 22)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand(shape, generator=torch.Generator().manual_seed(23)) * 2 - 1
    with torch.no_grad():
        ref = K.autoencoder_srgan_forward(sd, x, num_autoencoders=kw["num_autoencoders"])
        m = m.to(DEV).eval()
        y = m(x.to(DEV)).cpu()
        assert y.shape == ref.shape == (shape[0], 3, 4 * shape[2], 4 * shape[3])
        assert (y - ref).abs().max().item() / 2.0 <= 1e-4
        yb = m.set_compute("bf16")(x.to(DEV)).cpu()
    p = _psnr01(yb, ref)
    print("Autoencoder SRGAN bf16 PSNR vs fp32 oracle: %.1f dB" % p)
    assert p >= 35.0


@pytest.mark.parametrize("kw,shape", [(dict(n_feats=64, depth=2, up_feats=64), (2, 3, 8, 12)), (dict(n_feats=64, depth=4), (1, 3, 16, 32))])
def test_res_unet_srgan_fp32_and_bf16(kw, shape):
    """the residual U-Net generator (Res-U-SRGAN/untitled0.py:39-125, 190-226); its ReLU-after-add blocks use
    the SRMI_ACT_ADD_RELU epilogue.  Parity unpinned (no TensorFlow): compared with oracle/keras_oracle.py."""
    m = ResUNetSRGAN(**kw)
    _randomise(m, 31)
    _randomise_bn(m, 32)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand(shape, generator=torch.Generator().manual_seed(33)) * 2 - 1
    with torch.no_grad():
        ref = K.res_unet_srgan_forward(sd, x, depth=kw["depth"])
        m = m.to(DEV).eval()
        y = m(x.to(DEV)).cpu()
        assert y.shape == ref.shape == (shape[0], 3, 4 * shape[2], 4 * shape[3])
        assert (y - ref).abs().max().item() / 2.0 <= 1e-4
        yb = m.set_compute("bf16")(x.to(DEV)).cpu()
    p = _psnr01(yb, ref)
    print("Res-U-Net SRGAN bf16 PSNR vs fp32 oracle: %.1f dB" % p)
    assert p >= 35.0
