#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (TEST INFRASTRUCTURE).

Run in the build container only (needs /root/reference, read-only; never on the GPU box):

    python oracle/make_golden.py

For every case it (1) builds the reference module (P/srgan_model.py, imported unmodified),
(2) loads the deterministic recipe weights (oracle/weights.py) through load_state_dict,
(3) runs the reference forward on CPU fp32, (4) checks the functional restatement in
oracle/srgan_oracle.py against it, and (5) stores inputs + the REFERENCE's outputs.
Only data is stored (npz, no pickles, no reference source or bytecode).
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference/pytorch srgan with lpips,dists/SRGAN_CustomDataset-main"
sys.path.insert(0, REF)

import types

import numpy as np
import torch

from oracle import srgan_oracle as O
from oracle import weights as W

sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))  # unused import in losses.py
sys.modules["torchvision"].transforms = types.ModuleType("torchvision.transforms")
from srgan_model import Discriminator, Generator  # noqa: E402  (the reference)
import losses as ref_losses  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)


def _check(name, a, b, tol=2e-6):
    d = (a - b).abs().max().item()
    print(f"  oracle-vs-reference {name:28s} max|d| = {d:.3e}")
    assert d <= tol, (name, d)


def gen_case(fname, num_block, shape, scale=4, alpha=0.25, wseed=0, xseed=1, taps=True,
             full=True, stride=16):
    print(fname)
    np_sd = W.generator_weights(seed=wseed, alpha=alpha, num_block=num_block, scale=scale)
    sd = O.to_torch(np_sd)
    g = Generator(num_block=num_block, scale=scale)
    assert list(g.state_dict().keys()) == list(np_sd.keys()), "schema drift vs reference"
    g.load_state_dict(sd, strict=True)
    g.eval()
    x = torch.from_numpy(W.lr_input(shape, seed=xseed))
    got = {}
    hooks = []
    if taps:
        hooks.append(g.conv01.register_forward_hook(lambda m, i, o: got.__setitem__("conv01", o.clone())))
        hooks.append(g.body[0].register_forward_hook(lambda m, i, o: got.__setitem__("block0", o.clone())))
        hooks.append(g.tail[0].register_forward_hook(lambda m, i, o: got.__setitem__("up0", o.clone())))
    with torch.no_grad():
        sr, feat = g(x)
        otaps = {}
        osr, ofeat = O.generator_forward(sd, x, num_block=num_block, scale=scale, taps=otaps)
    for h in hooks:
        h.remove()
    _check("sr", sr, osr)
    _check("feat", feat, ofeat, tol=2e-5)
    for k in got:
        _check(k, got[k], otaps[k], tol=2e-5)
    out = dict(num_block=num_block, scale=scale, alpha=np.float32(alpha), wseed=wseed, xseed=xseed,
               shape=np.array(shape))
    if full:
        out.update(x=x.numpy(), sr=sr.numpy(), feat=feat.numpy())
        for k, v in got.items():
            out["tap_" + k] = v.numpy()
    else:
        # config-shape case: strided samples + fp64 moments of the reference output
        out.update(sr_sample=sr[:, :, ::stride, ::stride].contiguous().numpy(),
                   feat_sample=feat[:, :, ::12, ::12].contiguous().numpy(), feat_stride=12,
                   sr_sum=sr.double().sum().item(), sr_sumsq=(sr.double() ** 2).sum().item(),
                   sr_absmax=sr.abs().max().item(),
                   feat_sum=feat.double().sum().item(), feat_sumsq=(feat.double() ** 2).sum().item(),
                   feat_absmax=feat.abs().max().item(), stride=stride)
    print(f"  sr range [{sr.min():.3f},{sr.max():.3f}] std {sr.std():.3f}; feat std {feat.std():.3f}")
    np.savez_compressed(os.path.join(GOLD, fname), **out)


def gen_train_case(fname, num_block, shape, wseed=0, xseed=3):
    """train()-mode forward: batch-stat BN + running-stat update (P/mode.py:99-101)."""
    print(fname)
    np_sd = W.generator_weights(seed=wseed, num_block=num_block)
    sd = O.to_torch(np_sd)
    g = Generator(num_block=num_block)
    g.load_state_dict(sd, strict=True)
    g.train()
    x = torch.from_numpy(W.lr_input(shape, seed=xseed))
    with torch.no_grad():
        sr, feat = g(x)
        stats = {}
        osr, ofeat = O.generator_forward(sd, x, num_block=num_block, training=True, stats_out=stats)
    _check("sr(train)", sr, osr, tol=1e-5)
    _check("feat(train)", feat, ofeat, tol=5e-5)
    after = g.state_dict()
    out = dict(num_block=num_block, wseed=wseed, xseed=xseed, shape=np.array(shape),
               x=x.numpy(), sr=sr.numpy(), feat=feat.numpy())
    for k, v in stats.items():
        _check(k, after[k].double(), v.double(), tol=1e-5)
        out["stat:" + k] = after[k].numpy()
    np.savez_compressed(os.path.join(GOLD, fname), **out)


def _summ(t, stride=37):
    """full tensor when small, else fp64 moments + a strided sample (same recipe on the test side)."""
    t = t.detach()
    flat = t.reshape(-1)
    return dict(sum=flat.double().sum().item(), sumsq=(flat.double() ** 2).sum().item(),
                absmax=flat.abs().max().item() if flat.numel() else 0.0, sample=flat[::stride].clone().numpy())


def pretrain_step_case(fname, n_feats, num_block, shape, full, wseed=0, xseed=21, steps=2):
    """`train()` phase 1 (P/mode.py:96-106): generator(lr) in train mode -> MSELoss(gt, output) ->
    g_optim.zero_grad(); loss.backward(); g_optim.step(), Adam(lr=1e-4) (P/mode.py:90), run by the REFERENCE
    Generator.  Stores the loss, every parameter gradient of step 1, and parameters / BN buffers after `steps`."""
    print(fname)
    np_sd = W.generator_weights(seed=wseed, num_block=num_block, n_feats=n_feats)
    sd = O.to_torch(np_sd)
    g = Generator(n_feats=n_feats, num_block=num_block)
    g.load_state_dict(sd, strict=True)
    g.train()
    lr = torch.from_numpy(W.lr_input(shape, seed=xseed))
    gt = torch.from_numpy(W.lr_input((shape[0], 3, shape[2] * 4, shape[3] * 4), seed=xseed + 1))
    l2 = torch.nn.MSELoss()
    opt = torch.optim.Adam(g.parameters(), lr=1e-4)
    out = dict(n_feats=n_feats, num_block=num_block, wseed=wseed, xseed=xseed, shape=np.array(shape), steps=steps,
               full=int(full))
    losses_ = []
    for it in range(steps):
        output, _ = g(lr)
        loss = l2(gt, output)
        opt.zero_grad()
        loss.backward()
        if it == 0:
            # oracle: same forward restated functionally, gradients by torch autograd
            leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
                      for k, v in sd.items()}
            shared = leaves["conv01.body.1.weight"]
            for k in leaves:
                if k.endswith(".weight") and leaves[k].shape == (1,) and k != "conv01.body.1.weight":
                    leaves[k] = shared
            osr, _ = O.generator_forward(leaves, lr, num_block=num_block, training=True, stats_out={})
            oloss = torch.nn.functional.mse_loss(osr, gt)
            oloss.backward()
            _check("loss", loss.detach(), oloss.detach(), tol=1e-6)
            for name, p in g.named_parameters():
                ref_g = leaves[name].grad
                tol = 1e-5 * max(1.0, p.grad.abs().max().item())
                _check("grad " + name[-26:], p.grad, ref_g, tol=tol)
                if full:
                    out["grad:" + name] = p.grad.numpy().copy()
                else:
                    for kk, vv in _summ(p.grad).items():
                        out[f"grad_{kk}:" + name] = vv
        opt.step()
        losses_.append(loss.item())
    out["loss"] = np.array(losses_, dtype=np.float64)
    for name, v in g.state_dict().items():
        if full:
            out["after:" + name] = v.numpy().copy()
        else:
            for kk, vv in _summ(v.float()).items():
                out[f"after_{kk}:" + name] = vv
    print(f"  loss {losses_}")
    np.savez_compressed(os.path.join(GOLD, fname), **out)


def disc_case(fname, patch, batch, wseed=0, xseed=5):
    print(fname)
    np_sd = W.discriminator_weights(seed=wseed, patch_size=patch)
    sd = O.to_torch(np_sd)
    d = Discriminator(patch_size=patch)
    assert list(d.state_dict().keys()) == list(np_sd.keys()), "schema drift vs reference"
    d.load_state_dict(sd, strict=True)
    x = torch.from_numpy(W.lr_input((batch, 3, patch, patch), seed=xseed))
    out = dict(patch=patch, wseed=wseed, xseed=xseed, x=x.numpy())
    with torch.no_grad():
        d.eval()
        p_eval = d(x)
        _check("prob(eval)", p_eval, O.discriminator_forward(sd, x))
        d.train()
        p_train = d(x)
        stats = {}
        _check("prob(train)", p_train, O.discriminator_forward(sd, x, training=True, stats_out=stats), tol=1e-5)
    after = d.state_dict()
    for k, v in stats.items():
        _check(k, after[k].double(), v.double(), tol=1e-5)
        out["stat:" + k] = after[k].numpy()
    out.update(prob_eval=p_eval.numpy(), prob_train=p_train.numpy())
    np.savez_compressed(os.path.join(GOLD, fname), **out)


def loss_case(fname):
    print(fname)
    x = torch.from_numpy(W.lr_input((2, 5, 6, 7), seed=7))
    tv = ref_losses.TVLoss()(x)
    _check("tv", tv, O.tv_loss(x))
    img = torch.from_numpy(W.lr_input((2, 3, 8, 8), seed=8, lo=0.0, hi=1.0))
    ms = ref_losses.MeanShift()
    with torch.no_grad():
        shifted = ms(img)
    _check("meanshift", shifted, O.mean_shift(img), tol=1e-6)
    np.savez_compressed(os.path.join(GOLD, fname), tv_x=x.numpy(), tv=tv.numpy(),
                        ms_x=img.numpy(), ms=shifted.numpy())


if __name__ == "__main__":
    gen_case("gen_b1_1x3x12x16.npz", 1, (1, 3, 12, 16))
    gen_case("gen_b2_2x3x8x8.npz", 2, (2, 3, 8, 8), alpha=0.1, wseed=2, xseed=11)
    gen_case("gen_b16_1x3x16x16.npz", 16, (1, 3, 16, 16), taps=False)
    gen_case("gen_b1_scale2_1x3x9x7.npz", 1, (1, 3, 9, 7), scale=2, taps=False)
    gen_case("gen_b1_scale3_1x3x6x5.npz", 1, (1, 3, 6, 5), scale=3, taps=False)
    gen_case("gen_b16_cfg1_16x3x96x96.npz", 16, (16, 3, 96, 96), taps=False, full=False)
    gen_train_case("gen_train_b2_4x3x8x8.npz", 2, (4, 3, 8, 8))
    disc_case("disc_p32_b3.npz", 32, 3)
    loss_case("losses.npz")
    pretrain_step_case("pretrain_step_f16_b2_4x3x8x8.npz", 16, 2, (4, 3, 8, 8), full=True)
    pretrain_step_case("pretrain_step_f64_b1_2x3x8x8.npz", 64, 1, (2, 3, 8, 8), full=False)
    # restore the process-wide shared PReLU to its default for anyone importing after us
    print("done ->", GOLD)
