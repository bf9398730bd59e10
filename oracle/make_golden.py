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


QUIET = False


def _check(name, a, b, tol=2e-6):
    d = (a - b).abs().max().item()
    if not QUIET:
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


class VggShaped(torch.nn.Module):
    """VGG19-shaped frozen stack with recipe weights standing in for torchvision's pretrained net (absent
    offline).  Built from plain torch.nn layers in torchvision's `features` order; forward returns the 37-entry
    namedtuple of P/vgg19.py:23-78 (in-place ReLUs, so conv entries alias the relu entries as in the reference)."""

    def __init__(self, np_sd):
        super().__init__()
        from collections import namedtuple
        layers, cin = [], 3
        for v in W.VGG19_CFG_E:
            if v == "M":
                layers.append(torch.nn.MaxPool2d(2, 2))
            else:
                layers += [torch.nn.Conv2d(cin, v, 3, padding=1), torch.nn.ReLU(inplace=True)]
                cin = v
        self.vgg_feature = torch.nn.Sequential(*layers)
        self.load_state_dict(O.to_torch(np_sd))
        for p_ in self.parameters():
            p_.requires_grad = False
        self._out = namedtuple("vgg_output", O.VGG_LAYERS)

    def forward(self, x):
        outs = []
        for m in self.vgg_feature:
            x = m(x)
            outs.append(x)
        return self._out(*outs)


def _put(out, prefix, name, t):
    t = t.detach().float()
    n = t.numel()
    if n <= 4096:
        out[f"{prefix}:{name}"] = t.numpy().copy()
        return
    stride = max(1, n // 2048) | 1
    flat = t.reshape(-1)
    out[f"{prefix}_stride:{name}"] = stride
    out[f"{prefix}_sample:{name}"] = flat[::stride].clone().numpy()
    out[f"{prefix}_sum:{name}"] = flat.double().sum().item()
    out[f"{prefix}_sumsq:{name}"] = (flat.double() ** 2).sum().item()
    out[f"{prefix}_absmax:{name}"] = flat.abs().max().item()


def _reference_step_grads(num_block, patch, tv_coeff, g_np, d_np, v_np, lr, gt):
    """D grads and G grads of one fine-tune iteration (P/mode.py:148-179) by the REFERENCE modules, fresh state."""
    import torch.nn as nn
    import torch.optim as optim
    batch = lr.shape[0]
    generator = Generator(num_block=num_block)
    generator.load_state_dict(O.to_torch(g_np), strict=True)
    generator.train()
    discriminator = Discriminator(patch_size=patch)
    discriminator.load_state_dict(O.to_torch(d_np), strict=True)
    discriminator.train()
    vgg_net = VggShaped(v_np).eval()
    l2_loss, cross_ent, tv_loss = nn.MSELoss(), nn.BCELoss(), ref_losses.TVLoss()
    g_optim = optim.Adam(generator.parameters(), lr=1e-4)
    d_optim = optim.Adam(discriminator.parameters(), lr=1e-4)
    VGG_loss = ref_losses.perceptual_loss(vgg_net)
    real_label, fake_label = torch.ones((batch, 1)), torch.zeros((batch, 1))
    output, _ = generator(lr)
    fake_prob = discriminator(output)
    real_prob = discriminator(gt)
    d_loss = cross_ent(real_prob, real_label) + cross_ent(fake_prob, fake_label)
    g_optim.zero_grad(); d_optim.zero_grad()
    d_loss.backward()
    dg = {n: p_.grad.clone() for n, p_ in discriminator.named_parameters()}
    d_optim.step()
    output, _ = generator(lr)
    fake_prob = discriminator(output)
    _percep_loss, hr_feat, sr_feat = VGG_loss((gt + 1.0) / 2.0, (output + 1.0) / 2.0, layer="relu5_4")
    g_loss = 0.006 * _percep_loss + 1e-3 * cross_ent(fake_prob, real_label) + \
        tv_coeff * tv_loss(0.006 * (hr_feat - sr_feat) ** 2) + l2_loss(output, gt)
    g_optim.zero_grad(); d_optim.zero_grad()
    g_loss.backward()
    gg = {n: p_.grad.clone() for n, p_ in generator.named_parameters()}
    return dg, gg


def _sensitivity(out, num_block, patch, tv_coeff, g_np, d_np, v_np, lr, gt, base_dg, base_gg, trials):
    """How far the REFERENCE's own fp32 gradients move when its inputs are perturbed at rounding level (every input
    value multiplied by 1 + 1e-7 * N(0,1): one or two ulps): the spread that activation-kink flips and Adam's
    sign(grad) first step cause in ANY fp32 implementation of this iteration.  Stored per gradient family as the worst
    and the median per-tensor relative L2 change and the worst max-abs change (fraction of max|grad|), maximised over
    `trials` perturbations.  The GPU tests gate srmi's distance from the reference at a small multiple of these."""
    keysd = set(d_np.keys())
    keysg = set(g_np.keys())

    def bn_fed_bias(name, keys):
        return name.endswith("body.0.bias") and name[:-len("0.bias")] + "1.running_mean" in keys

    stats = {"dgrad": [0.0, 0.0, 0.0], "ggrad": [0.0, 0.0, 0.0]}
    for t in range(trials):
        gen = torch.Generator().manual_seed(9000 + t)
        lr_p = lr * (1.0 + 1e-7 * torch.randn(lr.shape, generator=gen))
        gt_p = gt * (1.0 + 1e-7 * torch.randn(gt.shape, generator=gen))
        dg, gg = _reference_step_grads(num_block, patch, tv_coeff, g_np, d_np, v_np, lr_p, gt_p)
        for fam, got, base, keys in (("dgrad", dg, base_dg, keysd), ("ggrad", gg, base_gg, keysg)):
            l2, mx = [], []
            for n, b in base.items():
                if bn_fed_bias(n, keys):
                    continue
                d = (got[n] - b).double()
                l2.append(float(d.norm()) / max(float(b.double().norm()), 1e-30))
                mx.append(float(d.abs().max()) / max(float(b.abs().max()), 1e-30))
            st = stats[fam]
            st[0] = max(st[0], max(l2)); st[1] = max(st[1], sorted(l2)[len(l2) // 2]); st[2] = max(st[2], max(mx))
        print(f"  sensitivity trial {t}: " + ", ".join(f"{k} l2max {v[0]:.2e} l2med {v[1]:.2e} absmax {v[2]:.2e}" for k, v in stats.items()))
    for fam, st in stats.items():
        out[f"sens_{fam}_l2max"], out[f"sens_{fam}_l2med"], out[f"sens_{fam}_absmax"] = st
    out["sens_trials"] = trials


def srgan_step_case(fname, num_block, batch, lr_hw, tv_coeff, wseed=0, xseed=31, sens_trials=6):
    """One fine-tune iteration, P/mode.py:148-179, run with the REFERENCE Generator, Discriminator,
    perceptual_loss, TVLoss (imported) and a VGG-shaped recipe-weight stack; mode.py itself cannot be imported
    (skimage / lpips / DISTS absent), so its loop body is restated line by line below."""
    import torch.nn as nn
    import torch.optim as optim
    print(fname)
    T_ = __import__("oracle.train_oracle", fromlist=["x"])
    patch = lr_hw * 4
    g_np = W.generator_weights(seed=wseed, num_block=num_block)
    d_np = W.discriminator_weights(seed=wseed + 1, patch_size=patch)
    v_np = W.vgg19_weights(seed=wseed + 2)
    generator = Generator(num_block=num_block)
    generator.load_state_dict(O.to_torch(g_np), strict=True)
    generator.train()
    discriminator = Discriminator(patch_size=patch)
    discriminator.load_state_dict(O.to_torch(d_np), strict=True)
    discriminator.train()
    vgg_net = VggShaped(v_np).eval()
    lr = torch.from_numpy(W.lr_input((batch, 3, lr_hw, lr_hw), seed=xseed))
    gt = torch.from_numpy(W.lr_input((batch, 3, patch, patch), seed=xseed + 1))
    vgg_rescale_coeff, adv_coeff, feat_layer = 0.006, 1e-3, "relu5_4"       # P/main.py:16,18,19 defaults

    l2_loss = nn.MSELoss()
    g_optim = optim.Adam(generator.parameters(), lr=1e-4)
    d_optim = optim.Adam(discriminator.parameters(), lr=1e-4)
    VGG_loss = ref_losses.perceptual_loss(vgg_net)
    cross_ent = nn.BCELoss()
    tv_loss = ref_losses.TVLoss()
    real_label = torch.ones((batch, 1))
    fake_label = torch.zeros((batch, 1))

    # oracle twin
    gsd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in O.to_torch(g_np).items()}
    shared = gsd["conv01.body.1.weight"]
    for k in gsd:
        if k.endswith(".weight") and gsd[k].shape == (1,):
            gsd[k] = shared
    dsd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in O.to_torch(d_np).items()}
    vsd = O.to_torch(v_np)
    gstats, dstats = {}, {}
    d_part, g_part = T_.srgan_losses(gsd, dsd, vsd, lr, gt, num_block, gstats, dstats, vgg_rescale_coeff, adv_coeff,
                                     tv_coeff, feat_layer)

    out = dict(num_block=num_block, batch=batch, lr_hw=lr_hw, tv_coeff=tv_coeff, wseed=wseed, xseed=xseed)
    # ---- Training Discriminator (P/mode.py:148-161)
    output, _ = generator(lr)
    fake_prob = discriminator(output)
    real_prob = discriminator(gt)
    d_loss_real = cross_ent(real_prob, real_label)
    d_loss_fake = cross_ent(fake_prob, fake_label)
    d_loss = d_loss_real + d_loss_fake
    g_optim.zero_grad()
    d_optim.zero_grad()
    d_loss.backward()
    od_loss = d_part()
    od_loss.backward()
    _check("d_loss", d_loss.detach(), od_loss.detach(), tol=1e-6)
    base_dg = {}
    for name, p_ in discriminator.named_parameters():
        _check("dgrad " + name[-24:], p_.grad, dsd[name].grad, tol=1e-5 * max(1.0, p_.grad.abs().max().item()))
        _put(out, "dgrad", name, p_.grad)
        base_dg[name] = p_.grad.clone()
    d_optim.step()
    with torch.no_grad():                                   # move the oracle's D to the updated parameters
        for name, p_ in discriminator.named_parameters():
            dsd[name].copy_(p_)
            dsd[name].grad = None
    for k in gsd.values():
        if k.grad is not None:
            k.grad = None
    # ---- Training Generator (P/mode.py:163-179)
    output, _ = generator(lr)
    fake_prob = discriminator(output)
    _percep_loss, hr_feat, sr_feat = VGG_loss((gt + 1.0) / 2.0, (output + 1.0) / 2.0, layer=feat_layer)
    L2_loss = l2_loss(output, gt)
    percep_loss = vgg_rescale_coeff * _percep_loss
    adversarial_loss = adv_coeff * cross_ent(fake_prob, real_label)
    total_variance_loss = tv_coeff * tv_loss(vgg_rescale_coeff * (hr_feat - sr_feat) ** 2)
    g_loss = percep_loss + adversarial_loss + total_variance_loss + L2_loss
    g_optim.zero_grad()
    d_optim.zero_grad()
    g_loss.backward()
    og_loss, parts = g_part()
    og_loss.backward()
    _check("g_loss", g_loss.detach(), og_loss.detach(), tol=2e-6 * max(1.0, abs(g_loss.item())))
    _check("percep", percep_loss.detach(), parts["percep"].detach(), tol=2e-6 * max(1.0, abs(percep_loss.item())))
    base_gg = {}
    for name, p_ in generator.named_parameters():
        _check("ggrad " + name[-24:], p_.grad, gsd[name].grad, tol=2e-5 * max(1.0, p_.grad.abs().max().item()))
        _put(out, "ggrad", name, p_.grad)
        base_gg[name] = p_.grad.clone()
    g_optim.step()
    out.update(d_loss=d_loss.item(), g_loss=g_loss.item(), percep_loss=percep_loss.item(),
               adversarial_loss=adversarial_loss.item(), total_variance_loss=total_variance_loss.item(),
               L2_loss=L2_loss.item(), hr_feat_absmax=hr_feat.abs().max().item())
    for name, v in generator.state_dict().items():
        _put(out, "after_g", name, v)
    for name, v in discriminator.state_dict().items():
        _put(out, "after_d", name, v)
    # ---- the same iteration in float64 (oracle restatement, its own f64 Adam step for D): the yardstick that tells
    # rounding-order effects (activation-kink flips, Adam's first step = lr * sign(grad)) from arithmetic errors.
    # Tests require srmi's fp32 gradients to sit about as close to these as the reference's own fp32 gradients do.
    f64 = lambda sd: {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
                          else (v.double().clone() if v.is_floating_point() else v.clone()))
                      for k, v in O.to_torch(sd).items()}
    g64, d64, v64 = f64(g_np), f64(d_np), {k: v.double() for k, v in O.to_torch(v_np).items()}
    shared64 = g64["conv01.body.1.weight"]
    for k in g64:
        if k.endswith(".weight") and g64[k].shape == (1,):
            g64[k] = shared64
    d_part64, g_part64 = T_.srgan_losses(g64, d64, v64, lr.double(), gt.double(), num_block, {}, {}, vgg_rescale_coeff,
                                         adv_coeff, tv_coeff, feat_layer)
    d_part64().backward()
    dnames = [n for n, _ in discriminator.named_parameters()]
    for name in dnames:
        _put(out, "dgrad64", name, d64[name].grad)
    with torch.no_grad():
        ps = [d64[n] for n in dnames]
        gs = [d64[n].grad.clone() for n in dnames]
        T_.adam_step(ps, gs, dict(step=0, m=[torch.zeros_like(p_) for p_ in ps], v=[torch.zeros_like(p_) for p_ in ps]))
    for t in list(g64.values()) + list(d64.values()):
        if t.grad is not None:
            t.grad = None
    g_part64()[0].backward()
    for name, _ in generator.named_parameters():
        _put(out, "ggrad64", name, g64[name].grad)
    _sensitivity(out, num_block, patch, tv_coeff, g_np, d_np, v_np, lr, gt, base_dg, base_gg, sens_trials)
    for k in gstats:
        _check("G " + k[-28:], generator.state_dict()[k].double(), gstats[k].double(), tol=1e-5)
    for k in dstats:
        _check("D " + k[-28:], discriminator.state_dict()[k].double(), dstats[k].double(), tol=1e-5)
    print(f"  d_loss {d_loss.item():.6f} g_loss {g_loss.item():.6f} percep {percep_loss.item():.6f} "
          f"adv {adversarial_loss.item():.6f} tv {total_variance_loss.item():.3e} L2 {L2_loss.item():.6f}")
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
    # batch 8 x LR 16 (HR 64): the smallest size at which the discriminator's deepest train-mode BatchNorm
    # (N*4*4 = 128 elements per channel) is well conditioned - at batch 2 x LR 8 the REFERENCE's own fp32
    # gradients sit 0.5-3 % away from an fp64 run of the same graph, so nothing can be pinned to them.
    srgan_step_case("srgan_step_b8_lr16_tv0.npz", 2, 8, 16, 0.0)            # reference defaults (tv coeff 0, P/main.py:17)
    srgan_step_case("srgan_step_b8_lr16_tv.npz", 2, 8, 16, 2e-2, xseed=41)  # TV term switched on
    # the reference's DEFAULT configuration (P/main.py:12,14,21: res_num 16, batch_size 16, patch_size 24;
    # P/mode.py:128: Discriminator(patch_size=96)) = BASELINE config 3 per GPU
    srgan_step_case("srgan_step_default_b16_lr24.npz", 16, 16, 24, 0.0, xseed=51, sens_trials=4)
    # restore the process-wide shared PReLU to its default for anyone importing after us
    print("done ->", GOLD)
