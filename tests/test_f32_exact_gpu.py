"""GPU: the f32 parity subset once more in ONE fresh child process with SRMI_F32_EXACT=1.

The switch is read once per process (csrc/conv_b3_f32.hip, conv_b3h_f32.hip), so the exact-f32 MFMA kernel
(`conv_f32t_kernel`) - the documented fallback arithmetic of every conv the split-bf16 kernels normally take - is only
reachable on those shapes from a process that starts with the variable set.  The child runs the f32 tests of the
generator fixtures, the f32 shape fuzz, the teacher-forced cfg3 layers and the single-op training tests; the kernel-name
assertions inside them (tests/test_abi.py, test_generator_gpu.py) check that no split kernel was selected."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_f32_parity_subset_on_the_exact_kernels():
    if os.environ.get("SRMI_F32_EXACT", "0") not in ("", "0"):
        pytest.skip("already running under SRMI_F32_EXACT=1")
    env = dict(os.environ, SRMI_F32_EXACT="1")
    cmd = [sys.executable, "-u", "-m", "pytest", "-x", "-q", "-m", "gpu or not gpu", "-p", "no:cacheprovider",
           "--timeout", "120", "--timeout-method", "thread",
           "tests/test_abi.py::test_kernel_selection",
           "tests/test_generator_gpu.py::test_generator_fp32_matches_reference",
           "tests/test_fuzz_gpu.py::test_f32_path_fuzz",
           "tests/test_layers_cfg3_gpu.py",
           "tests/test_train_ops_gpu.py::test_fused_conv_forward_backward"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
