#!/usr/bin/env python3
"""Fixtures for the BASELINE configurations whose oracle run is too long for a GPU test's CPU side:

  cfg4  ESRGAN-style RRDB generator, 23 blocks, 8x3x256x256   -> tests/golden/cfg4_rrdb23_8x3x256x256.npz
  cfg2  SRCNN on 32x1x384x384, both parameterisations         -> tests/golden/cfg2_srcnn_{935,915}_32x1x384x384.npz

TEST INFRASTRUCTURE ONLY.  The expected outputs come from oracle/keras_oracle.py (fp32, plain torch CPU ops), the
restatement of ESRGAN/untitled0.py:18-67 and SRCNN/srcnn.py:199-210.  PARITY UNPINNED w.r.t. the reference: TensorFlow is
absent, the Keras scripts cannot be imported and hold no stored outputs.  Each fixture keeps strided
samples of the output plus float64 moments (sum, sum of squares, abs-max) - a few hundred KB, no tensors of the
reference, no source.  Weights/inputs are regenerated from seeds by the tests (oracle/weights.py: randomise_module).

    python oracle/make_golden_configs.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd")]
from oracle import keras_oracle as K  # noqa: E402
from oracle import weights as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
NSAMPLE = 65536


def summarise(y, **meta):
    flat = y.reshape(-1)
    stride = max(1, flat.numel() // NSAMPLE) | 1
    out = dict(meta)
    out.update(shape=np.array(y.shape), stride=stride, sample=flat[::stride].clone().numpy(),
               sum=flat.double().sum().item(), sumsq=(flat.double() ** 2).sum().item(), absmax=flat.abs().max().item())
    return out


def cfg4():
    from srmi.keras_models import RRDBNet
    m = W.randomise_module(RRDBNet(num_blocks=23), 5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand((8, 3, 256, 256), generator=torch.Generator().manual_seed(6)) * 2 - 1
    t0 = time.time()
    with torch.no_grad():
        y = torch.cat([K.rrdbnet_forward(sd, x[i:i + 1], num_blocks=23) for i in range(8)])
    print("cfg4 oracle %.1f s, out %s absmax %.3f" % (time.time() - t0, tuple(y.shape), y.abs().max()))
    np.savez_compressed(os.path.join(GOLD, "cfg4_rrdb23_8x3x256x256.npz"), **summarise(y, wseed=5, xseed=6, num_blocks=23))


def cfg2(tag, kw):
    from srmi.keras_models import SRCNN
    m = W.randomise_module(SRCNN(**kw), 1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.rand((32, 1, 384, 384), generator=torch.Generator().manual_seed(2))
    t0 = time.time()
    with torch.no_grad():
        y = torch.cat([K.srcnn_forward(sd, x[i:i + 4]) for i in range(0, 32, 4)])
    print("cfg2 %s oracle %.1f s, out %s absmax %.3f" % (tag, time.time() - t0, tuple(y.shape), y.abs().max()))
    np.savez_compressed(os.path.join(GOLD, f"cfg2_srcnn_{tag}_32x1x384x384.npz"), **summarise(y, wseed=1, xseed=2))


if __name__ == "__main__":
    cfg2("935", dict())                              # the reference's topology, SRCNN/srcnn.py:205-210
    cfg2("915", dict(k=(9, 1, 5), ch=(64, 32)))      # BASELINE config 2's parameterisation
    cfg4()
