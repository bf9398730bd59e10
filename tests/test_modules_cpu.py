"""CPU: the drop-in module surface (names, signatures, state_dict schema, shared PReLU, loud failure
without a device).  No kernels run here."""
import copy
import inspect

import numpy as np
import pytest
import torch
import torch.nn as nn

import srmi
from oracle import srgan_oracle as O
from oracle import weights as W
from srmi import ops
from srmi.srgan_model import Discriminator, Generator


def test_signatures_match_reference():
    # P/srgan_model.py:8, :43 and P/ops.py:7,19,38,51,71,87
    assert list(inspect.signature(Generator.__init__).parameters)[1:] == [
        "img_feat", "n_feats", "kernel_size", "num_block", "act", "scale"]
    assert list(inspect.signature(Discriminator.__init__).parameters)[1:] == [
        "img_feat", "n_feats", "kernel_size", "act", "num_of_block", "patch_size"]
    assert list(inspect.signature(ops._conv.__init__).parameters)[1:] == [
        "in_channels", "out_channels", "kernel_size", "stride", "padding", "bias"]
    assert list(inspect.signature(ops.conv.__init__).parameters)[1:] == [
        "in_channel", "out_channel", "kernel_size", "BN", "act", "stride", "bias"]
    assert list(inspect.signature(ops.ResBlock.__init__).parameters)[1:] == ["channels", "kernel_size", "act", "bias"]
    assert list(inspect.signature(ops.Upsampler.__init__).parameters)[1:] == ["channel", "kernel_size", "scale", "act"]
    assert list(inspect.signature(ops.discrim_block.__init__).parameters)[1:] == [
        "in_feats", "out_feats", "kernel_size", "act"]
    assert list(inspect.signature(ops.BasicBlock.__init__).parameters)[1:] == [
        "in_channels", "out_channels", "kernel_size", "num_res_block", "act"]


@pytest.mark.parametrize("kw", [dict(), dict(num_block=2), dict(scale=2, num_block=1), dict(scale=3, num_block=1)])
def test_generator_state_dict_schema(kw):
    g = Generator(**kw)
    schema = W.generator_schema(**kw)
    sd = g.state_dict()
    assert list(sd.keys()) == list(schema.keys())
    for k, (shape, _) in schema.items():
        assert tuple(sd[k].shape) == tuple(shape), k


def test_generator_param_count_and_shared_prelu():
    g = Generator()
    assert len(list(g.parameters())) == 141
    assert sum(p.numel() for p in g.parameters()) == 1_535_620
    assert len({id(m) for m in g.modules() if isinstance(m, nn.PReLU)}) == 1
    g2 = Generator()
    assert g2.conv01.body[1] is g.conv01.body[1]        # shared across instances (P/srgan_model.py:8)
    assert float(g.conv01.body[1].weight) == pytest.approx(0.25)


def test_conv_quirks():
    c = ops._conv(3, 8, 5, stride=1, padding=0, bias=False)    # padding/bias args are ignored (P/ops.py:8-9)
    assert c.padding == (2, 2) and c.bias is not None
    assert float(c.bias.abs().max()) == 0.0
    assert 0.005 < float(c.weight.std()) < 0.04


def test_discriminator_schema_and_linear_size():
    d = Discriminator(patch_size=96)
    assert d.linear_size == 18432
    assert list(d.state_dict().keys()) == list(W.discriminator_schema(patch_size=96).keys())
    assert sum(p.numel() for p in d.parameters()) == 23_565_377
    assert isinstance(d.act, nn.LeakyReLU) and d.act.negative_slope == pytest.approx(0.01)


def test_load_recipe_weights_roundtrip():
    g = Generator(num_block=2)
    sd = O.to_torch(W.generator_weights(seed=3, num_block=2))
    g.load_state_dict(sd, strict=True)
    back = g.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k
    # restore the process-wide shared PReLU
    with torch.no_grad():
        g.conv01.body[1].weight.fill_(0.25)


def test_cpu_input_fails_loudly():
    g = Generator(num_block=1).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        g(torch.zeros(1, 3, 8, 8))
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        g.conv01(torch.zeros(1, 3, 8, 8))


def test_deepcopy_drops_plans():
    g = Generator(num_block=1)
    g2 = copy.deepcopy(g)
    assert g2._plans.plans == {}
    assert list(g2.state_dict().keys()) == list(g.state_dict().keys())


def test_compute_mode_switch():
    g = Generator(num_block=1)
    assert g.compute in ("fp32", "bf16")
    g.set_compute("bf16")
    assert g.compute == "bf16"
    with pytest.raises(ValueError):
        g.set_compute("fp8")


def test_dataset_transforms_callable_like_reference():
    """`crop` / `augmentation` / `Compose` applied to a host sample (P/dataset.py:87-132): same random draws in the same
    order as the reference's `__call__`s, same numpy result."""
    import random

    from srmi import dataset as D
    rng = np.random.default_rng(3)
    lr = rng.integers(0, 256, (20, 17, 3)).astype(np.float32)
    gt = rng.integers(0, 256, (80, 68, 3)).astype(np.float32)
    tf = D.Compose([D.crop(4, 8), D.augmentation()])
    random.seed(11)
    out = tf({"LR": lr, "GT": gt})
    # the reference's arithmetic, restated inline with the same draw order: ix, iy, hor_flip, ver_flip, rot
    random.seed(11)
    ix = random.randrange(0, 17 - 8 + 1); iy = random.randrange(0, 20 - 8 + 1)
    l, g = lr[iy:iy + 8, ix:ix + 8], gt[4 * iy:4 * iy + 32, 4 * ix:4 * ix + 32]
    hf, vf, rot = random.randrange(0, 2), random.randrange(0, 2), random.randrange(0, 2)
    if hf:
        l, g = np.fliplr(l), np.fliplr(g)
    if vf:
        l, g = np.flipud(l), np.flipud(g)
    if rot:
        l, g = l.transpose(1, 0, 2), g.transpose(1, 0, 2)
    assert np.array_equal(out["LR"], l) and np.array_equal(out["GT"], g)
    assert hasattr(D, "testOnly_data") and hasattr(D, "mydata")          # `from dataset import *` of P/mode.py:7


def test_invalidate_clears_caches():
    g = Generator(num_block=1)
    g._plans.plans[("x",)] = (None, None)
    g.conv01._cache.entries["k"] = 1
    assert srmi.invalidate(g) is g
    assert not g._plans.plans and not g.conv01._cache.entries
