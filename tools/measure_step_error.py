#!/usr/bin/env python3
"""Measures what tests/test_srgan_step_gpu.py gates: per-tensor relative L2 / max error of the D and G gradients of one
SRGAN iteration against the reference-made fixtures, and run-to-run bit equality.  Prints the worst tensors.

    python tools/measure_step_error.py [fixture.npz ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution_amd"), os.path.join(ROOT, "tests")]

import numpy as np
import torch

import test_srgan_step_gpu as T


def main():
    for term in ("l2", "adv", "percep"):
        le, ge = T.loss_term_errors(term)
        print(f"loss term {term}: loss rel err {le:.2e}, worst grad rel-L2 {ge:.3e}")
    gold = os.path.join(ROOT, "tests", "golden")
    names = sys.argv[1:] or ["srgan_step_b8_lr16_tv0.npz", "srgan_step_b8_lr16_tv.npz", "srgan_step_default_b16_lr24.npz"]
    for fname in names:
        g = np.load(os.path.join(gold, fname), allow_pickle=False)
        r1 = T.run_step(g)
        r2 = T.run_step(g)
        same = all(torch.equal(r1[k][n], r2[k][n]) for k in ("dgrad", "ggrad") for n in r1[k])
        print(f"== {fname}: rerun bit-equal: {same}; d_loss err {abs(r1['d_loss'] - float(g['d_loss'])):.2e} "
              f"g_loss err {abs(r1['g_loss'] - float(g['g_loss'])):.2e}")
        keys = {"dgrad": set(r1["discriminator"].state_dict().keys()), "ggrad": set(r1["generator"].state_dict().keys())}
        for prefix in ("dgrad", "ggrad"):
            rows = []
            for name, got in r1[prefix].items():
                if T._bn_fed_bias(name, keys[prefix]):
                    continue                                   # exactly zero by construction
                e = T.grad_errors(g, prefix, name, got)
                e64 = T.grad_errors(g, prefix, name, got, truth=prefix + "64")
                r64 = T.grad_errors(g, prefix, name, None, truth=prefix + "64")
                rows.append((e[0], e[1], e64[0], r64[0], name))
            med = lambda i: sorted(r[i] for r in rows)[len(rows) // 2]
            mx = lambda i: max(r[i] for r in rows)
            print(f"  {prefix}: vs reference fp32: rel-L2 max {mx(0):.3e} median {med(0):.3e}, max/absmax max {mx(1):.3e}")
            print(f"  {prefix}: vs f64 yardstick:  ours max {mx(2):.3e} median {med(2):.3e} | reference fp32 max {mx(3):.3e} median {med(3):.3e}")
            rows.sort(reverse=True)
            for r in rows[:3]:
                print(f"     {r[4]:44s} vs ref32 {r[0]:.3e} (max {r[1]:.3e})  ours-f64 {r[2]:.3e}  ref32-f64 {r[3]:.3e}")


if __name__ == "__main__":
    main()
