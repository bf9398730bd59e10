"""CPU: the oracle restatement against the golden vectors produced by the reference itself
(oracle/make_golden.py), plus recipe determinism and schema facts."""
import os

import numpy as np
import pytest
import torch

from oracle import srgan_oracle as O
from oracle import weights as W

GEN_FULL = ["gen_b1_1x3x12x16.npz", "gen_b2_2x3x8x8.npz", "gen_b16_1x3x16x16.npz",
            "gen_b1_scale2_1x3x9x7.npz", "gen_b1_scale3_1x3x6x5.npz"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_schema_counts():
    s = W.generator_schema()
    assert len(s) == 258
    n_alias = sum(1 for _, (_, k) in s.items() if k == W.PRELU)
    assert n_alias == 19
    n_params = sum(int(np.prod(sh)) for _, (sh, k) in s.items()
                   if k in (W.CONV_W, W.CONV_B, W.BN_G, W.BN_B)) + 1
    assert n_params == 1_535_620
    d = W.discriminator_schema(patch_size=96)
    assert sum(int(np.prod(sh)) for _, (sh, k) in d.items() if k not in (W.BN_RM, W.BN_RV, W.BN_NBT)) == 23_565_377


def test_recipe_deterministic():
    a = W.generator_weights(seed=0, num_block=1)
    b = W.generator_weights(seed=0, num_block=1)
    c = W.generator_weights(seed=1, num_block=1)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert not np.array_equal(a["conv01.body.0.weight"], c["conv01.body.0.weight"])
    x1, x2 = W.lr_input((1, 3, 4, 4), seed=1), W.lr_input((1, 3, 4, 4), seed=1)
    assert np.array_equal(x1, x2) and x1.min() >= -1 and x1.max() <= 1


@pytest.mark.parametrize("name", GEN_FULL)
def test_generator_oracle_matches_reference_vectors(golden_dir, name):
    g = _load(golden_dir, name)
    nb, scale = int(g["num_block"]), int(g["scale"])
    sd = O.to_torch(W.generator_weights(seed=int(g["wseed"]), alpha=float(g["alpha"]), num_block=nb, scale=scale))
    x = torch.from_numpy(g["x"])
    assert np.array_equal(g["x"], W.lr_input(tuple(g["shape"]), seed=int(g["xseed"])))
    taps = {}
    with torch.no_grad():
        sr, feat = O.generator_forward(sd, x, num_block=nb, scale=scale, taps=taps)
    assert np.abs(sr.numpy() - g["sr"]).max() <= 2e-6
    assert np.abs(feat.numpy() - g["feat"]).max() <= 2e-5
    for k in ("conv01", "block0", "up0"):
        if "tap_" + k in g.files:
            assert np.abs(taps[k].numpy() - g["tap_" + k]).max() <= 2e-5


def test_generator_train_mode_oracle(golden_dir):
    g = _load(golden_dir, "gen_train_b2_4x3x8x8.npz")
    sd = O.to_torch(W.generator_weights(seed=int(g["wseed"]), num_block=2))
    stats = {}
    with torch.no_grad():
        sr, feat = O.generator_forward(sd, torch.from_numpy(g["x"]), num_block=2, training=True, stats_out=stats)
    assert np.abs(sr.numpy() - g["sr"]).max() <= 1e-5
    assert np.abs(feat.numpy() - g["feat"]).max() <= 5e-5
    keys = [k for k in g.files if k.startswith("stat:")]
    assert len(keys) == 5 * 3   # 5 BN layers x (mean, var, count)
    for k in keys:
        assert np.abs(stats[k[5:]].double().numpy() - g[k].astype(np.float64)).max() <= 1e-5


def test_discriminator_oracle(golden_dir):
    g = _load(golden_dir, "disc_p32_b3.npz")
    sd = O.to_torch(W.discriminator_weights(seed=int(g["wseed"]), patch_size=int(g["patch"])))
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        assert np.abs(O.discriminator_forward(sd, x).numpy() - g["prob_eval"]).max() <= 2e-6
        stats = {}
        pt = O.discriminator_forward(sd, x, training=True, stats_out=stats)
    assert np.abs(pt.numpy() - g["prob_train"]).max() <= 1e-5
    for k in [k for k in g.files if k.startswith("stat:")]:
        assert np.abs(stats[k[5:]].double().numpy() - g[k].astype(np.float64)).max() <= 1e-5


def test_losses_oracle(golden_dir):
    g = _load(golden_dir, "losses.npz")
    assert abs(O.tv_loss(torch.from_numpy(g["tv_x"])).item() - float(g["tv"])) <= 1e-6
    assert np.abs(O.mean_shift(torch.from_numpy(g["ms_x"])).numpy() - g["ms"]).max() <= 1e-6


def test_cfg1_fixture_is_consistent(golden_dir):
    g = _load(golden_dir, "gen_b16_cfg1_16x3x96x96.npz")
    assert tuple(g["shape"]) == (16, 3, 96, 96)
    assert g["sr_sample"].shape == (16, 3, 24, 24)
    assert float(g["sr_absmax"]) < 1.0      # tanh not saturated: the fixture can see errors
