"""Seeded shape fuzz of the LDS-tiled bf16 kernels: ragged tiles of every variant (6x96, 8x64, 4x32, the 6x48
two-workgroup tile), channel slices, all store modes - each case against the CPU f64 restatement (one output
rounding), and the fused ResBlock bit-for-bit against its two-launch form."""
import importlib

import numpy as np
import pytest
import torch

import test_kernels_gpu as K

L = importlib.import_module("super-resolution_amd.srmi._lib")
pytestmark = pytest.mark.gpu


def _c64_cases(n, seed):
    g = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = i % 5
        H = int(g.integers(1, 40)); W = int(g.integers(1, 200))
        N = int(g.integers(1, 4))
        case = dict(N=N, H=H, W=W)
        if kind == 0:
            case.update(Cout=64, act=L.ACT_PRELU, residual=bool(g.integers(0, 2)))
        elif kind == 1:
            case.update(Cout=256, act=L.ACT_PRELU, r=2, dcr=bool(g.integers(0, 2)), alpha_pc=int(g.integers(0, 2)))
        elif kind == 2:
            case.update(Cout=128, act=L.ACT_LRELU, act_param=float(g.uniform(0.0, 0.3)))
        elif kind == 3:
            case.update(Cout=64, act=L.ACT_NONE, residual=True, in_cs=160, out_cs=160, out_co=int(g.integers(0, 13)) * 8)
        else:
            case.update(Cout=64, act=L.ACT_RELU, in_cs=int(g.integers(8, 21)) * 8)
        out.append(case)
    # shapes that select the two-workgroup 6x48 tile (>= 1024 of them) and the 8x64 tile, ragged
    out.append(dict(N=2, H=190, W=187, Cout=256, act=L.ACT_PRELU, r=2))
    out.append(dict(N=1, H=250, W=250, Cout=64, act=L.ACT_PRELU, residual=True))
    return out


@pytest.mark.parametrize("case", _c64_cases(30, seed=2024))
def test_c64_fuzz(case):
    assert K._run_case(Cin=64, k=3, io=torch.bfloat16, compute=L.COMPUTE_BF16, **case) == "c64_bf16"


def _resblock_cases(n, seed):
    g = np.random.default_rng(seed)
    out = []
    for i in range(n):
        out.append(dict(N=int(g.integers(1, 4)), H=int(g.integers(1, 30)), W=int(g.integers(1, 210)),
                        other_res=bool(i % 4 == 3), alpha=float(g.choice([0.25, 0.0, 1.0, -0.3]))))
    return out


@pytest.mark.parametrize("case", _resblock_cases(16, seed=77))
def test_resblock_fuzz(case):
    K.test_fused_resblock_kernel(case)


def _up_tail_cases(n, seed):
    g = np.random.default_rng(seed)
    out = []
    for i in range(n):
        out.append(dict(N=int(g.integers(1, 4)), H=int(g.integers(1, 60)), W=int(g.integers(1, 110)),
                        alpha=float(g.choice([0.25, 0.0, 1.0, -0.3])), act2=int(g.choice([L.ACT_TANH, L.ACT_NONE, L.ACT_SIGMOID]))))
    return out


@pytest.mark.parametrize("case", _up_tail_cases(12, seed=91))
def test_up_tail_fuzz(case):
    K.test_fused_up_tail_kernel(case)


def _srcnn_cases(n, seed):
    g = np.random.default_rng(seed)
    pw, c1, k9 = [], [], []
    for i in range(n):
        Cin, Cout = int(g.choice([32, 64, 128])), int(g.choice([16, 32, 64]))
        case = dict(N=int(g.integers(1, 4)), H=int(g.integers(1, 40)), W=int(g.integers(1, 90)), Cin=Cin, Cout=Cout,
                    act=int(g.choice([L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU, L.ACT_PRELU])), act_param=0.2)
        if i % 3 == 0:
            case.update(in_cs=Cin + 8 * int(g.integers(1, 5)), out_cs=Cout + 32, out_co=4 * int(g.integers(0, 9)))
        if i % 4 == 1:
            case.update(residual=True)
        pw.append(case)
        c1.append(dict(N=int(g.integers(1, 3)), H=int(g.integers(1, 50)), W=int(g.integers(1, 420)), Cin=int(g.choice([32, 64])),
                       k=int(g.choice([3, 5])), act=int(g.choice([L.ACT_NONE, L.ACT_RELU, L.ACT_TANH])),
                       in_cs=0 if i % 3 else 72, out_bf16=bool(i % 5 == 4)))
        k9.append(dict(N=int(g.integers(1, 3)), H=int(g.integers(1, 50)), W=int(g.integers(1, 200)), Cout=int(g.choice([64, 128])),
                       act=int(g.choice([L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU, L.ACT_PRELU]))))
    return pw, c1, k9


_PW, _C1, _K9 = _srcnn_cases(10, seed=404)


@pytest.mark.parametrize("case", _PW)
def test_pw_fuzz(case):
    K.test_pw_bf16_kernel(case)


@pytest.mark.parametrize("case", _C1)
def test_c1out_fuzz(case):
    K.test_c1out_bf16_kernel(case)


@pytest.mark.parametrize("case", _K9)
def test_k9c1_fuzz(case):
    K.test_k9c1_kernel(case)


def _cin_cases(n, seed):
    g = np.random.default_rng(seed)
    out = []
    for i in range(n):
        Cin = int(g.choice([32, 96, 128, 160, 192, 256]))
        Cout = int(g.choice([32, 64, 96, 128]))
        case = dict(N=int(g.integers(1, 3)), H=int(g.integers(1, 30)), W=int(g.integers(1, 150)), Cin=Cin, Cout=Cout,
                    act=int(g.choice([L.ACT_NONE, L.ACT_RELU, L.ACT_LRELU, L.ACT_PRELU])), act_param=0.2)
        if i % 3 == 0:
            case.update(in_cs=Cin + 32, out_cs=Cout + 64, out_co=int(g.integers(0, 9)) * 8)
        if i % 4 == 1:
            case.update(residual=True)
        out.append(case)
    g2 = np.random.default_rng(seed + 1000)        # the activation that follows the residual (own stream: keeps the cases above)
    for i in range(4):
        Cin, Cout = int(g2.choice([64, 128, 192, 384])), int(g2.choice([64, 128]))
        case = dict(N=int(g2.integers(1, 3)), H=int(g2.integers(1, 30)), W=int(g2.integers(1, 150)), Cin=Cin, Cout=Cout,
                    act=L.ACT_ADD_RELU, residual=i != 3)
        if i == 1:
            case.update(out_cs=3 * Cout, out_co=2 * Cout)
        out.append(case)
    return out


@pytest.mark.parametrize("case", _cin_cases(16, seed=9))
def test_cin_fuzz(case):
    assert K._run_case(k=3, io=torch.bfloat16, compute=L.COMPUTE_BF16, **case) == "cin_bf16"


def _thin_cases(n, seed):
    g = np.random.default_rng(seed)
    return [dict(N=int(g.integers(1, 3)), H=int(g.integers(1, 60)), W=int(g.integers(1, 200)), Cout=int(g.integers(1, 5)),
                 act=int(g.choice([L.ACT_TANH, L.ACT_NONE, L.ACT_SIGMOID])), out_layout=int(g.choice([L.NCHW, L.NHWC])),
                 in_cs=int(g.choice([0, 96, 160]))) for _ in range(n)]


@pytest.mark.parametrize("case", _thin_cases(12, seed=10))
def test_thin_fuzz(case):
    K.test_c64_thin_kernel(case)


def _k9_cases(n, seed):
    g = np.random.default_rng(seed)
    return [dict(N=int(g.integers(1, 3)), H=int(g.integers(1, 40)), W=int(g.integers(1, 150)),
                 in_layout=int(g.choice([L.NCHW, L.NHWC])), Cin=int(g.choice([1, 3]))) for _ in range(n)]


@pytest.mark.parametrize("case", _k9_cases(10, seed=11))
def test_k9c3_fuzz(case):
    K.test_k9c3_kernel(case)


def _f32_cases(n, seed):
    """the f32 path: split-bf16 kernel where Cin, Cout are multiples of 32 (csrc/conv_b3_f32.hip), exact-f32 MFMA kernel elsewhere"""
    g = np.random.default_rng(seed)
    out = []
    for i in range(n):
        k = int(g.choice([1, 3, 3, 3, 5]))
        Cin = int(g.choice([32, 64, 96, 128, 160, 256])) if i % 4 else int(g.choice([3, 20, 40]))
        Cout = int(g.choice([32, 64, 96, 128, 256]))
        case = dict(N=int(g.integers(1, 4)), H=int(g.integers(1, 45)), W=int(g.integers(1, 70)), Cin=Cin, Cout=Cout, k=k,
                    stride=int(g.choice([1, 1, 2])))
        kind = i % 6
        if kind == 0:
            case.update(act=L.ACT_PRELU, residual=True)
        elif kind == 1 and Cout % 4 == 0:
            case.update(act=L.ACT_PRELU, r=2, dcr=bool(g.integers(0, 2)), alpha_pc=int(g.integers(0, 2)))
        elif kind == 2:
            case.update(act=L.ACT_LRELU, act_param=float(g.uniform(0.0, 0.3)), out_cs=Cout + 32, out_co=int(g.integers(0, 5)) * 8)
        elif kind == 3:
            case.update(act=L.ACT_TANH, out_layout=L.NCHW)
        elif kind == 4:
            case.update(act=L.ACT_ADD_RELU, residual=True, in_cs=Cin + 8 * int(g.integers(0, 5)))
        else:
            case.update(act=L.ACT_NONE)
        out.append(case)
    out.append(dict(N=2, H=96, W=96, Cin=64, Cout=64, k=3, act=L.ACT_RELU))           # several workgroups per CU, 128 x 64 tiles
    out.append(dict(N=1, H=40, W=40, Cin=64, Cout=256, k=3, act=L.ACT_PRELU, r=2))    # 128 x 128 tiles
    return out


@pytest.mark.parametrize("case", _f32_cases(28, seed=9))
def test_f32_path_fuzz(case):
    assert K._run_case(**case) in K.F32_KERNELS
