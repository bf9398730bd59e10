"""CPU: the C-ABI library loads and exports every symbol include/srmi.h declares; descriptor structs
match the header; host-side validation rejects bad descriptors BEFORE any launch (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from srmi import _lib as L
from srmi import functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "srmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(srmi_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported():
    lib = L.load()
    syms = _header_symbols()
    assert len(syms) >= 10
    assert sorted(L.SYMBOLS) == syms, "ctypes table and header disagree"
    for s in syms:
        assert hasattr(lib, s), s


def test_version_and_struct_layout():
    lib = L.load()
    assert lib.srmi_version() == 119
    assert C.sizeof(L.ConvDesc) == 24 * 4
    assert C.sizeof(L.ConvArgs) == 96 + 7 * 8 + 8 + 8          # ... + flags, reserved + x_copy (ABI 117)
    assert L.ConvArgs.x_copy.offset == 160
    assert L.ConvArgs.flags.offset == 152
    assert L.ConvArgs.x.offset == 96


def test_no_hard_hip_dependency():
    # the .so must bind to the process's HIP runtime, not pull in a second one
    import subprocess
    out = subprocess.run(["readelf", "-d", L.lib_path()], capture_output=True, text=True).stdout
    assert "amdhip64" not in out


def test_kernel_selection():
    d = Fn.make_desc(16, 96, 96, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, compute=L.COMPUTE_BF16)
    assert L.kernel_name(d) == "c64_bf16"
    assert L.packed_weight_bytes(d) == 64 * 64 * 9 * 2
    d = Fn.make_desc(16, 96, 96, 64, 256, 3, in_dtype=L.BF16, out_dtype=L.BF16, compute=L.COMPUTE_BF16, shuffle=2)
    assert L.kernel_name(d) == "c64_bf16"
    d = Fn.make_desc(16, 96, 96, 64, 64, 3)                      # f32 compute -> f32 path
    # Cin, Cout multiples of 32: three bf16 planes of the split-operand kernel; [tap][Cin][Cout] f32 with SRMI_F32_EXACT=1
    exact = os.environ.get("SRMI_F32_EXACT", "0") not in ("", "0")
    # the name is the kernel that RUNS: 3x3 / 64 channels / slope activation or none -> the halo-tile split kernel, any
    # other activation -> the general split kernel (a different operand order in the same byte count: srmi.h)
    assert L.kernel_name(d) == ("f32t" if exact else "b3h_f32")
    d_relu = Fn.make_desc(16, 96, 96, 64, 64, 3, act=L.ACT_RELU)
    assert L.kernel_name(d_relu) == ("f32t" if exact else "b3h_f32")
    d_tanh = Fn.make_desc(16, 96, 96, 64, 64, 3, act=L.ACT_TANH)
    assert L.kernel_name(d_tanh) == ("f32t" if exact else "b3_f32")
    assert L.kernel_name(Fn.make_desc(1, 8, 8, 3, 64, 9, in_layout=L.NCHW)) == "generic_f32"    # NCHW input: no LDS-tiled kernel
    assert L.kernel_name(Fn.make_desc(2, 8, 8, 64, 24, 3)) == "f32t"                            # Cout % 32 != 0: exact f32 MFMA
    assert L.kernel_name(Fn.make_desc(2, 8, 8, 64, 3, 3)) == "thin_f32"                         # <= 4 channels on one side: vector ALU
    assert L.kernel_name(Fn.make_desc(2, 8, 8, 3, 64, 3, act=L.ACT_LRELU, act_param=0.2)) == "thin_f32"
    assert L.packed_weight_bytes(d) == (9 * 64 * 64 * 4 if exact else 3 * 9 * 64 * 64 * 2)
    d = Fn.make_desc(1, 8, 8, 3, 64, 9)                          # Cin padded to 8, Cout to 64
    assert L.packed_weight_bytes(d) == 81 * 8 * 64 * 4
    d = Fn.make_desc(2, 8, 8, 64, 64, 3, stride=2, in_dtype=L.BF16, out_dtype=L.BF16, compute=L.COMPUTE_BF16)
    assert L.kernel_name(d) == "generic_f32"                    # stride 2 is not a c64 shape; bf16 storage -> the no-LDS kernel


@pytest.mark.parametrize("field,value", [("Cin", 0), ("Cout", -1), ("stride", 0), ("kh", 0), ("in_dtype", 7),
                                         ("out_layout", 3), ("act", 9), ("compute", 2), ("shuffle", 0),
                                         ("shuffle_order", 2)])
def test_bad_descriptor_rejected(field, value):
    d = Fn.make_desc(1, 8, 8, 64, 64, 3)
    setattr(d, field, value)
    n = C.c_size_t(0)
    rc = L.load().srmi_conv2d_packed_weight_bytes(C.byref(d), C.byref(n))
    assert rc == -1
    assert L.load().srmi_last_error() != b""
    with pytest.raises(L.SrmiError):
        L.check(rc)


def test_shuffle_divisibility_and_slices():
    d = Fn.make_desc(1, 8, 8, 64, 250, 3, shuffle=2)
    with pytest.raises(L.SrmiError):
        L.packed_weight_bytes(d)
    d = Fn.make_desc(1, 8, 8, 64, 32, 3, out_cstride=40, out_coffset=16)   # 16+32 > 40
    with pytest.raises(L.SrmiError):
        L.packed_weight_bytes(d)
    d = Fn.make_desc(1, 8, 8, 64, 32, 3, out_cstride=160, out_coffset=64)  # RRDB-style concat slice
    assert L.packed_weight_bytes(d) > 0


def test_null_arguments_rejected_without_launch():
    lib = L.load()
    assert lib.srmi_conv2d_fwd(None, None) == -1
    a = L.ConvArgs()
    a.d = Fn.make_desc(1, 8, 8, 64, 64, 3)
    assert lib.srmi_conv2d_fwd(C.byref(a), None) == -1          # NULL buffers
    assert b"NULL" in lib.srmi_last_error()
    assert lib.srmi_conv2d_fwd_seq(None, 0, None) == 0           # empty sequence is a no-op
    a.d = Fn.make_desc(0, 8, 8, 64, 64, 3)
    assert lib.srmi_conv2d_fwd(C.byref(a), None) == 0            # empty batch: nothing to launch
    assert lib.srmi_layout_convert(None, None, 0, 3, 8, 8, 1, 0, 0, 1, None) == 0
    assert lib.srmi_layout_convert(None, None, 1, 3, 8, 8, 1, 0, 0, 1, None) == -1


def test_python_constants_match_header():
    """Every SRMI_* integer #define of include/srmi.h that srmi/_lib.py mirrors (name minus the prefix) has the same value."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "srmi.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+SRMI_(\w+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\b", hdr, re.M)}
    both = [k for k in defs if hasattr(L, k) and isinstance(getattr(L, k), int)]
    assert len(both) >= 15 and "ACT_ADD_RELU" in both
    for k in both:
        assert getattr(L, k) == defs[k], k


def test_add_relu_kernel_selection():
    # the activation that follows the residual (Res-U-SRGAN/untitled0.py:67-72) is served by the K-chunked bf16 kernel
    # for 3x3 convs and by the generic kernel otherwise; the 64-channel tiles and the thin kernel decline it
    d = Fn.make_desc(2, 16, 16, 64, 64, 3, in_dtype=L.BF16, out_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_ADD_RELU)
    assert L.kernel_name(d) == "cin_bf16"
    d = Fn.make_desc(2, 16, 16, 64, 64, 1, in_dtype=L.BF16, out_dtype=L.BF16, compute=L.COMPUTE_BF16, act=L.ACT_ADD_RELU)
    assert L.kernel_name(d) == "generic_f32"
    d = Fn.make_desc(2, 16, 16, 64, 3, 3, in_dtype=L.BF16, out_dtype=L.F32, compute=L.COMPUTE_BF16, act=L.ACT_ADD_RELU)
    assert L.kernel_name(d) == "generic_f32"
    d = Fn.make_desc(2, 16, 16, 64, 64, 3, act=L.ACT_ADD_RELU + 1)
    n = C.c_size_t(0)
    assert L.load().srmi_conv2d_packed_weight_bytes(C.byref(d), C.byref(n)) != 0


def test_graft_entry_build_runs():
    """__graft_entry__.build(): make (up to date after the first run), library loads, version = include/srmi.h, every symbol."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    importlib.import_module("__graft_entry__").build()
