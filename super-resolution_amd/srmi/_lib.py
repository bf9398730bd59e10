"""ctypes binding of libsrmi.so (C ABI: include/srmi.h).

The library is pure HIP (no torch types in its signatures).  PyTorch is only used here for
device memory and streams: tensors' ``data_ptr()`` and ``torch.cuda.current_stream().cuda_stream``
cross the boundary as plain pointers.

There is NO fallback: if the shared library is missing or fails to load, importing the product
path raises ``SrmiLibraryError``.
"""
from __future__ import annotations

import ctypes as C
import os

F32, BF16 = 0, 1
NHWC, NCHW = 0, 1
COMPUTE_F32, COMPUTE_BF16 = 0, 1
ACT_NONE, ACT_PRELU, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_ADD_RELU = 0, 1, 2, 3, 4, 5, 6
SHUFFLE_CRD, SHUFFLE_DCR = 0, 1


class SrmiLibraryError(RuntimeError):
    pass


class SrmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libsrmi error {code}: {msg}")
        self.code = code


class ConvDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("res_dtype", C.c_int32),
        ("in_layout", C.c_int32), ("out_layout", C.c_int32),
        ("compute", C.c_int32),
        ("act", C.c_int32), ("act_param", C.c_float),
        ("alpha_per_channel", C.c_int32),
        ("shuffle", C.c_int32), ("shuffle_order", C.c_int32),
        ("in_cstride", C.c_int32), ("out_cstride", C.c_int32), ("out_coffset", C.c_int32),
        ("res_cstride", C.c_int32),
    ]


class ConvArgs(C.Structure):
    _fields_ = [
        ("d", ConvDesc),
        ("x", C.c_void_p), ("w", C.c_void_p),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("alpha", C.c_void_p),
        ("residual", C.c_void_p), ("y", C.c_void_p),
        ("flags", C.c_int32), ("reserved", C.c_int32),
        ("x_copy", C.c_void_p),
    ]


CONV_FUSE_NEXT = 1
CONV_FUSE_CHAIN = 2
CONV_Y_TRANSIENT = 4
ETIMEOUT = -4


class EpiDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("act", C.c_int32), ("act_param", C.c_float), ("alpha_per_channel", C.c_int32),
        ("shuffle", C.c_int32), ("shuffle_order", C.c_int32), ("bn", C.c_int32),
    ]


class CropItem(C.Structure):
    _fields_ = [
        ("lr_offset", C.c_int64), ("gt_offset", C.c_int64),
        ("lr_h", C.c_int32), ("lr_w", C.c_int32), ("iy", C.c_int32), ("ix", C.c_int32),
        ("flags", C.c_int32), ("reserved", C.c_int32),
    ]


AUG_HFLIP, AUG_VFLIP, AUG_TRANSPOSE = 1, 2, 4

# every symbol include/srmi.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "srmi_version": (C.c_int, []),
    "srmi_last_error": (C.c_char_p, []),
    "srmi_conv2d_kernel_name": (C.c_int, [C.POINTER(ConvDesc), C.c_char_p, C.c_size_t]),
    "srmi_conv2d_packed_weight_bytes": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_size_t)]),
    "srmi_conv2d_pack_weights": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_conv2d_fwd": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "srmi_conv2d_workspace_bytes": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_size_t)]),
    "srmi_conv2d_fwd_ws": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p, C.c_size_t, C.c_void_p]),
    "srmi_conv2d_fwd_seq": (C.c_int, [C.POINTER(ConvArgs), C.c_int32, C.c_void_p]),
    "srmi_conv2d_can_fuse": (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs)]),
    "srmi_conv2d_chain_length": (C.c_int, [C.POINTER(ConvArgs), C.c_int32]),
    "srmi_chain_status": (C.c_int, []),
    "srmi_conv2d_fwd_pair": (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), C.c_void_p]),
    "srmi_srcnn_layer_c1in": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "srmi_srcnn_layer_c1out": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "srmi_resblock_fwd": (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), C.c_void_p]),
    "srmi_layout_convert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "srmi_maxpool2x2_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p]),
    "srmi_upsample2x_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p]),
    "srmi_crop_augment_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_u8hwc_to_f32nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "srmi_f32nchw_to_u8hwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p]),
    "srmi_mse_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_mse_loss_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "srmi_bce_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_tv_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    "srmi_y_psnr_ssim": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_y_psnr_ssim_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_train_workspace_bytes": (C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    "srmi_bn_train_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_conv2d_fwd_bn_stats": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "srmi_bn_train_stats_n": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "srmi_epilogue_fwd": (C.c_int, [C.POINTER(EpiDesc)] + [C.c_void_p] * 9),
    "srmi_epilogue_bwd": (C.c_int, [C.POINTER(EpiDesc)] + [C.c_void_p] * 14),
    "srmi_conv2d_dgrad_packed_weight_bytes": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_size_t)]),
    "srmi_conv2d_dgrad_pack_weights": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_conv2d_dgrad_workspace_bytes": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_size_t)]),
    "srmi_conv2d_dgrad": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "srmi_conv2d_wgrad_workspace_bytes": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_size_t)]),
    "srmi_conv2d_wgrad": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "srmi_maxpool2x2_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "srmi_maxpool2x2_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p]),
    "srmi_mse_loss_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srmi_bce_loss_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "srmi_tv_loss_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                   C.c_void_p, C.c_void_p]),
    "srmi_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double,
                                 C.c_double, C.c_double, C.c_int32, C.c_void_p]),
}
REDUCE_WS_BYTES = 16384
METRICS_MAX_BATCH = 256
NOALLOC_MAX_ELEMS = 1 << 32

_LIB = None


def lib_path():
    return os.environ.get("SRMI_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsrmi.so")


def _preload_hip_runtime():
    """libsrmi.so carries no DT_NEEDED for libamdhip64: it must bind to the ONE HIP runtime of the
    process.  PyTorch-ROCm ships its own copy, so promote that one to the global symbol scope."""
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)

    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    for path in (cand, "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return path
        except OSError:
            continue
    raise SrmiLibraryError("no HIP runtime (libamdhip64.so) could be loaded")


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise SrmiLibraryError(
            f"{path} not found: build it with `python __graft_entry__.py build` "
            "(or `make -C super-resolution_amd/csrc`); there is no fallback path")
    _preload_hip_runtime()
    try:
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise SrmiLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SrmiLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().srmi_last_error()
        raise SrmiError(rc, msg.decode() if msg else "")


def kernel_name(desc: ConvDesc) -> str:
    buf = C.create_string_buffer(64)
    check(load().srmi_conv2d_kernel_name(C.byref(desc), buf, 64))
    return buf.value.decode()


def packed_weight_bytes(desc: ConvDesc) -> int:
    n = C.c_size_t(0)
    check(load().srmi_conv2d_packed_weight_bytes(C.byref(desc), C.byref(n)))
    return n.value
