"""GPU: the pre-training step of P/mode.py:96-106 (train-mode Generator -> MSELoss -> backward -> Adam) run by
srmi's modules + torch.optim.Adam exactly as the reference loop writes it, against fixtures produced by the
REFERENCE Generator on CPU (oracle/make_golden.py: loss, every .grad of step 1, parameters and BN buffers
after 2 steps)."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import weights as W
from srmi import losses
from srmi.srgan_model import Generator

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LR = 1e-4


def _run(golden_dir, fname, loss_module):
    g = np.load(os.path.join(golden_dir, fname), allow_pickle=False)
    n_feats, num_block, shape, steps = int(g["n_feats"]), int(g["num_block"]), tuple(int(v) for v in g["shape"]), int(g["steps"])
    np_sd = W.generator_weights(seed=int(g["wseed"]), num_block=num_block, n_feats=n_feats)
    gen = Generator(n_feats=n_feats, num_block=num_block)
    gen.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()})
    gen = gen.to(DEV)
    gen.train()
    lr_img = torch.from_numpy(W.lr_input(shape, seed=int(g["xseed"]))).to(DEV)
    gt = torch.from_numpy(W.lr_input((shape[0], 3, shape[2] * 4, shape[3] * 4), seed=int(g["xseed"]) + 1)).to(DEV)
    g_optim = torch.optim.Adam(gen.parameters(), lr=LR)
    grads, loss_vals = {}, []
    for it in range(steps):
        output, _ = gen(lr_img)                     # P/mode.py:101
        loss = loss_module(gt, output)              # :102
        g_optim.zero_grad()                         # :104
        loss.backward()                             # :105
        if it == 0:
            grads = {n: p.grad.detach().cpu().clone() for n, p in gen.named_parameters()}
        g_optim.step()                              # :106
        loss_vals.append(loss.item())
    return g, gen, grads, loss_vals


def _bn_fed_bias(name, sd_keys):
    """conv bias followed by train-mode BN: its true gradient is 0, the reference holds rounding noise."""
    return name.endswith("body.0.bias") and name[:-len("0.bias")] + "1.running_mean" in sd_keys


@pytest.mark.parametrize("loss_kind", ["srmi", "torch"])
def test_pretrain_step_full_fixture(golden_dir, loss_kind):
    loss_module = losses.MSELoss() if loss_kind == "srmi" else nn.MSELoss()
    g, gen, grads, loss_vals = _run(golden_dir, "pretrain_step_f16_b2_4x3x8x8.npz", loss_module)
    assert np.abs(np.array(loss_vals) - g["loss"]).max() <= 2e-6
    keys = set(gen.state_dict().keys())
    for name, got in grads.items():
        ref = torch.from_numpy(g["grad:" + name])
        tol = 2e-4 * float(ref.abs().max()) + 1e-7
        assert float((got - ref).abs().max()) <= tol, name
    for name, v in gen.state_dict().items():
        ref = torch.from_numpy(g["after:" + name])
        v = v.cpu()
        if name.endswith("num_batches_tracked"):
            assert int(v) == int(ref) == 2
        elif "running" in name:
            assert float((v - ref).abs().max()) <= 1e-5, name
        else:
            d = (v - ref).abs()
            assert float(d.max()) <= 2 * 2 * LR + 1e-7, name      # Adam moves each element <= ~lr per step
            pname = name
            if pname in grads and not _bn_fed_bias(name, keys):
                gr = torch.from_numpy(g["grad:" + pname]).abs()
                solid = gr > 1e-2 * gr.max()                       # elements whose gradient is not noise-level
                if solid.any():
                    assert float(d[solid].max()) <= 2e-6, name


def test_pretrain_step_width64_moments(golden_dir):
    g, gen, grads, loss_vals = _run(golden_dir, "pretrain_step_f64_b1_2x3x8x8.npz", losses.MSELoss())
    assert np.abs(np.array(loss_vals) - g["loss"]).max() <= 2e-6
    for name, got in grads.items():
        flat = got.reshape(-1)
        absmax = float(g["grad_absmax:" + name])
        ref_s = torch.from_numpy(g["grad_sample:" + name])
        assert float((flat[::37] - ref_s).abs().max()) <= 2e-4 * absmax + 1e-7, name
        n = flat.numel()
        assert abs(float(flat.double().sum()) - float(g["grad_sum:" + name])) <= 2e-4 * absmax * n ** 0.5 + 1e-6, name
        assert abs(float((flat.double() ** 2).sum()) - float(g["grad_sumsq:" + name])) <= 1e-3 * float(g["grad_sumsq:" + name]) + 1e-12, name
    for name, v in gen.state_dict().items():
        if "running" in name:
            ref_s = torch.from_numpy(g["after_sample:" + name])
            assert float((v.cpu().reshape(-1)[::37] - ref_s).abs().max()) <= 1e-5, name


def test_train_mode_forward_without_grad_updates_running_stats(golden_dir):
    """generator(lr) in .train() under no_grad (P/mode.py:99-101 semantics): batch-stat BN + running-stat update."""
    g = np.load(os.path.join(golden_dir, "gen_train_b2_4x3x8x8.npz"), allow_pickle=False)
    np_sd = W.generator_weights(seed=int(g["wseed"]), num_block=int(g["num_block"]))
    gen = Generator(num_block=int(g["num_block"]))
    gen.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()})
    gen = gen.to(DEV).train()
    with torch.no_grad():
        sr, feat = gen(torch.from_numpy(g["x"]).to(DEV))
    assert float((sr.cpu() - torch.from_numpy(g["sr"])).abs().max()) <= 1e-4
    assert float((feat.cpu() - torch.from_numpy(g["feat"])).abs().max()) <= 2e-4
    sd = gen.state_dict()
    for k in g.files:
        if k.startswith("stat:"):
            ref = torch.from_numpy(g[k])
            got = sd[k[5:]].cpu()
            if ref.dtype == torch.int64:
                assert int(got) == int(ref)
            else:
                assert float((got - ref).abs().max()) <= 1e-5, k
