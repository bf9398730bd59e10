"""The one entry point that allocates nothing (srmi_mse_loss_noalloc, include/srmi.h): called through the C ABI and
checked bit-for-bit against srmi_mse_loss (same kernels, caller workspace) on two calls in a row, the second with a
smaller input; plus its size guard.  Collected last (file name) so a failure here hides no other test."""
import ctypes as C

import numpy as np
import pytest
import torch

from srmi import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(lib, a, b, n):
    ws = torch.empty(L.REDUCE_WS_BYTES // 8, dtype=torch.float64, device=DEV)
    ref = torch.full((1,), -1.0, dtype=torch.float32, device=DEV)
    got = torch.full((1,), -2.0, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.srmi_mse_loss(a.data_ptr(), b.data_ptr(), n, ref.data_ptr(), ws.data_ptr(), st))
    L.check(lib.srmi_mse_loss_noalloc(a.data_ptr(), b.data_ptr(), n, got.data_ptr(), st))
    torch.cuda.synchronize()
    return ref.cpu().numpy(), got.cpu().numpy()


def test_noalloc_matches_workspace_entry_point_twice():
    lib = L.load()
    g = torch.Generator().manual_seed(77)
    n1, n2 = 16 * 3 * 96 * 96 + 5, 3 * 37 * 11            # second call smaller: fewer partial slots than the first left
    a = torch.randn(n1, generator=g).to(DEV)
    b = torch.randn(n1, generator=g).to(DEV)
    ref1, got1 = _both(lib, a, b, n1)
    assert ref1.tobytes() == got1.tobytes()
    ref2, got2 = _both(lib, a, b, n2)
    assert ref2.tobytes() == got2.tobytes()
    want = float(((a[:n2].double() - b[:n2].double()) ** 2).mean())
    assert abs(float(got2[0]) - want) <= 1e-6 * want
    # and a third, larger again, to show nothing stale is read
    ref3, got3 = _both(lib, a, b, n1 - 1000)
    assert ref3.tobytes() == got3.tobytes()


def test_noalloc_rejects_oversize_and_null():
    lib = L.load()
    x = torch.zeros(8, device=DEV)
    out = torch.zeros(1, device=DEV)
    rc = lib.srmi_mse_loss_noalloc(x.data_ptr(), x.data_ptr(), L.NOALLOC_MAX_ELEMS + 1, out.data_ptr(), 0)
    assert rc == -1 and b"SRMI_NOALLOC_MAX_ELEMS" in lib.srmi_last_error()
    rc = lib.srmi_mse_loss_noalloc(None, x.data_ptr(), 8, out.data_ptr(), 0)
    assert rc == -1
