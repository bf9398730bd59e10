"""Training path: ``torch.autograd.Function``s over the C ABI (SURVEY 8 b1 "what calls the boundary").

One Function per fused op - conv (+bias) -> [train-mode BatchNorm] -> [activation] -> [sub-pixel
store] -> [+ residual] - so ``loss.backward()`` and ``optim.Adam(model.parameters())`` of the
reference loops (P/mode.py:96-106, :141-179) run unmodified.  Everything between the user's NCHW
tensors is fp32 NHWC on the device (tensors carry ``torch.channels_last`` strides, so their logical
shape stays NCHW); all arithmetic is in libsrmi's kernels:

  forward   z = conv(x)+b      srmi_conv2d_fwd (exact-f32 MFMA kernel)
            batch stats         srmi_bn_train_stats (also updates running_mean / running_var)
            y = act(bn(z)) (+r) srmi_epilogue_fwd
  backward  dz, dgamma, dbeta,  srmi_epilogue_bwd
            dalpha, dbias
            dW                  srmi_conv2d_wgrad
            dx                  srmi_conv2d_dgrad

There is no fallback to ATen convolution / batch-norm kernels anywhere in this file.
"""
from __future__ import annotations

import ctypes as C
import threading
import weakref

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import functional as Fn

_CL = torch.channels_last
_ONES = {}


def _ones(n, device):
    key = (device, n)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones(n, dtype=torch.float32, device=device)
    return t


def nhwc(x: torch.Tensor) -> torch.Tensor:
    """Logical NCHW tensor whose storage is dense NHWC f32 (converted by srmi_layout_convert if needed)."""
    Fn.require_cuda(x)
    if x.dtype != torch.float32:
        raise TypeError(f"srmi training path is fp32, got {x.dtype}")
    if x.is_contiguous(memory_format=_CL):
        return x
    x = x.contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty_like(x, memory_format=_CL)
    if x.numel():
        L.check(L.load().srmi_layout_convert(x.data_ptr(), out.data_ptr(), N, Cc, H, W, L.NCHW, L.F32, L.NHWC, L.F32,
                                             Fn.stream_ptr()))
    return out


def nchw(x: torch.Tensor) -> torch.Tensor:
    """Dense NCHW copy of a channels_last tensor (module edge, e.g. the SR image handed back to the user)."""
    if x.is_contiguous():
        return x
    x = nhwc(x)
    N, Cc, H, W = x.shape
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
    if x.numel():
        L.check(L.load().srmi_layout_convert(x.data_ptr(), out.data_ptr(), N, Cc, H, W, L.NHWC, L.F32, L.NCHW, L.F32,
                                             Fn.stream_ptr()))
    return out


def _empty_cl(shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device, memory_format=_CL)


def _ptr(t):
    return t.data_ptr() if t is not None else None


_POISON = bool(int(__import__("os").environ.get("SRMI_POISON_WS", "0")))


class _Workspace:
    """One growing device buffer per (device, stream) for the split-K slices of fwd / dgrad / wgrad.  Kernels that
    use it are enqueued on that stream and consume it before the next one starts, so sharing is ordered."""

    def __init__(self):
        self.bufs = {}
        self.lock = threading.Lock()     # forward (caller's thread) and backward (autograd engine threads) both come here

    def get(self, nbytes, device):
        if nbytes == 0:
            return None
        key = (device, Fn.stream_ptr(device))
        with self.lock:
            t = self.bufs.get(key)
            if t is None or t.numel() < nbytes:
                t = self.bufs[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        if _POISON:        # SRMI_POISON_WS=1 (tests): NaN-fill before every use - a kernel that reads a slot it did not write shows
            t.view(torch.float32).fill_(float("nan"))
        return t


_WS = _Workspace()
_WS_TRAIN = _Workspace()      # f64 partial slots of the per-channel reductions (BN statistics, epilogue backward)


_TRAIN_WS_BYTES = {}


def _train_ws(C_, device):
    nb = _TRAIN_WS_BYTES.get(C_)
    if nb is None:
        n = C.c_size_t(0)
        L.check(L.load().srmi_train_workspace_bytes(C_, C.byref(n)))
        nb = _TRAIN_WS_BYTES[C_] = n.value
    return _WS_TRAIN.get(nb, device)


class _ConvPlan:
    """Everything about one conv shape that does not depend on the tensors: the descriptor, its output shape and the three
    workspace sizes.  Cached per shape: the SRGAN step runs ~140 fused ops, each of which used to build its descriptor and
    ask the library for its workspace sizes on every call (4-5 ctypes round trips per op)."""

    __slots__ = ("desc", "Ho", "Wo", "ws_fwd", "ws_dgrad", "ws_wgrad", "dgrad_pack_bytes")

    def __init__(self, key):
        N, H, W, Cin, Cout, kh, kw, stride, pad, act, act_param = key
        lib = L.load()
        self.desc = Fn.make_desc(N, H, W, Cin, Cout, (kh, kw), stride=stride, pad=pad, compute=L.COMPUTE_F32)
        self.desc.act, self.desc.act_param = act, act_param      # (only the forward launch reads them)
        _, _, self.Ho, self.Wo = Fn.out_shape(self.desc)
        n = C.c_size_t(0)
        L.check(lib.srmi_conv2d_workspace_bytes(C.byref(self.desc), C.byref(n)))
        self.ws_fwd = n.value
        L.check(lib.srmi_conv2d_dgrad_workspace_bytes(C.byref(self.desc), C.byref(n)))
        self.ws_dgrad = n.value
        L.check(lib.srmi_conv2d_wgrad_workspace_bytes(C.byref(self.desc), C.byref(n)))
        self.ws_wgrad = n.value
        L.check(lib.srmi_conv2d_dgrad_packed_weight_bytes(C.byref(self.desc), C.byref(n)))
        self.dgrad_pack_bytes = n.value


_PLANS = {}
_PLANS_LOCK = threading.Lock()


def _conv_plan(N, H, W, Cin, Cout, kh, kw, stride, pad, act=L.ACT_NONE, act_param=0.0):
    key = (N, H, W, Cin, Cout, kh, kw, stride, pad, act, act_param)
    pl = _PLANS.get(key)
    if pl is None:
        with _PLANS_LOCK:
            pl = _PLANS.get(key)
            if pl is None:
                if len(_PLANS) > 4096:
                    _PLANS.clear()
                pl = _PLANS[key] = _ConvPlan(key)
    return pl


class _PackCache:
    """Packed forward / dgrad weights per parameter VERSION: re-packed only after the optimizer (or anything else
    torch versions) has written the weight - once per step, not once per call (the discriminator and the frozen VGG run
    several times per SRGAN iteration, P/mode.py:150-151, :165, :167).  Keyed by the parameter OBJECT (the base tensor of
    a `.view()`, so nn.Linear's weight viewed as OIHW hits its entry) and held through a weak reference: an entry dies
    with its parameter and can never be served to a new tensor that happens to reuse the address.  A write through
    `.data` is invisible to the version counter: call srmi.invalidate(module), which clears this cache too."""

    def __init__(self):
        self.entries = {}
        self.lock = threading.Lock()

    def get(self, kind, w, build):
        base = w._base if w._base is not None else w
        key = (kind, id(base), tuple(w.shape))
        hit = self.entries.get(key)
        if hit is not None and hit[0]() is base and hit[1] == (base._version, base.data_ptr()):
            return hit[2]
        packed = build()
        with self.lock:
            if len(self.entries) > 4096:           # dead entries of discarded models
                self.entries = {k: v for k, v in self.entries.items() if v[0]() is not None}
            self.entries[key] = (weakref.ref(base), (base._version, base.data_ptr()), packed)
        return packed

    def clear(self):
        self.entries.clear()


_PACKS = _PackCache()


_BN_UPDATES = threading.local()


class bn_running_updates:
    """`with bn_running_updates(2): out = model(x)` - this ONE forward stands for `n` forwards of the same modules over the
    same batch with unchanged weights: every train-mode BatchNorm moves its running statistics and `num_batches_tracked`
    n times (srmi_bn_train_stats_n), exactly as n forwards would leave them.  The SRGAN iteration of the reference runs the
    generator twice per step on the same LR batch (P/mode.py:149 for the discriminator step, :164 for its own step) with
    no generator update in between: the second forward's outputs are the first's, bit for bit, and only these buffers
    tell how often it ran."""

    def __init__(self, n):
        if not 0 <= int(n) <= 64:
            raise ValueError("bn_running_updates: 0 <= n <= 64")
        self.n = int(n)

    def __enter__(self):
        self.prev = getattr(_BN_UPDATES, "n", 1)
        _BN_UPDATES.n = self.n
        return self

    def __exit__(self, *exc):
        _BN_UPDATES.n = self.prev


class _Cfg:
    """static (non-tensor) configuration of one fused op"""

    __slots__ = ("stride", "pad", "act", "act_param", "apc", "shuffle", "order", "bn")

    def __init__(self, stride, pad, act, act_param, apc, shuffle, order, bn):
        self.stride, self.pad, self.act, self.act_param, self.apc = stride, pad, act, act_param, apc
        self.shuffle, self.order, self.bn = shuffle, order, bn


class FusedConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, alpha, residual, cfg: _Cfg):
        lib = L.load()
        st = Fn.stream_ptr()
        x = nhwc(x)
        dev = x.device
        N, Cin, H, W = x.shape
        Cout, _, kh, kw = weight.shape
        plan = _conv_plan(N, H, W, Cin, Cout, kh, kw, cfg.stride, cfg.pad)
        desc, Ho, Wo = plan.desc, plan.Ho, plan.Wo
        w = weight.detach().contiguous()
        # (the operand format belongs to the kernel the library selects, which may depend on H, W, stride and pad - never on
        # N: sub-pixel / full-map / halo-tile forms - so those are part of the cache key)
        wp = _PACKS.get(("fwd", H, W, cfg.stride, cfg.pad), weight, lambda: Fn.pack_weights(desc, w))
        shift = bias.detach().contiguous() if bias is not None else None
        bn = cfg.bn
        # ReLU / LeakyReLU of a layer without BN, residual or sub-pixel store (the VGG19 stack, D's first conv and Linear)
        # go into the conv launch when that launch does not split K (a split needs a linear epilogue): no second pass
        # over the output, and y alone is kept - its sign is the pre-activation's, all the backward needs.
        if (bn is None and residual is None and cfg.shuffle == 1 and plan.ws_fwd == 0 and N * Ho * Wo > 0 and
                (cfg.act == L.ACT_RELU or (cfg.act == L.ACT_LRELU and cfg.act_param > 0.0))):
            planf = _conv_plan(N, H, W, Cin, Cout, kh, kw, cfg.stride, cfg.pad, cfg.act, float(cfg.act_param))
            if planf.ws_fwd == 0:
                y = _empty_cl((N, Cout, Ho, Wo), dev)
                a = L.ConvArgs()
                Fn.fill_args(a, planf.desc, x, wp, None, shift, None, None, y)
                L.check(lib.srmi_conv2d_fwd_ws(C.byref(a), None, 0, st))
                ed = L.EpiDesc(N, Ho, Wo, Cout, cfg.act, cfg.act_param, cfg.apc, 1, cfg.order, 0)
                ctx.save_for_backward(x, w, y, None, None, None, None, None)
                ctx.cfg, ctx.desc, ctx.ed, ctx.plan = cfg, desc, ed, plan
                ctx.has_bias, ctx.has_res = bias is not None, False
                ctx.weight_obj = weight
                return y
        z = _empty_cl((N, Cout, Ho, Wo), dev)
        # scale NULL = 1: lets the kernel split K on small maps; the workspace makes that split deterministic
        a = L.ConvArgs()
        Fn.fill_args(a, desc, x, wp, None, shift, None, None, z)
        wsk = _WS.get(plan.ws_fwd, dev)
        mean = rstd = None
        M = N * Ho * Wo
        if bn is None:
            L.check(lib.srmi_conv2d_fwd_ws(C.byref(a), _ptr(wsk), plan.ws_fwd, st))
        else:
            if M == 0:
                raise ValueError("srmi: train-mode BatchNorm over an empty batch")
            mean = torch.empty(Cout, dtype=torch.float32, device=dev)
            rstd = torch.empty(Cout, dtype=torch.float32, device=dev)
            ws = _train_ws(Cout, dev)
            track = bn.track_running_stats and bn.running_mean is not None
            if bn.momentum is None:
                raise NotImplementedError("srmi: BatchNorm2d(momentum=None) (cumulative average) is not supported")
            nbt = bn.num_batches_tracked if track else None      # counted on the device by the finalize launch
            # conv + batch statistics in one native call; where the conv runs on the halo-tile kernel (the generator's body
            # convs) its epilogue leaves the per-workgroup sums behind and no statistics pass reads z again
            L.check(lib.srmi_conv2d_fwd_bn_stats(C.byref(a), _ptr(wsk), plan.ws_fwd, float(bn.eps), float(bn.momentum),
                                                 mean.data_ptr(), rstd.data_ptr(), _ptr(bn.running_mean) if track else None,
                                                 _ptr(bn.running_var) if track else None, _ptr(nbt),
                                                 getattr(_BN_UPDATES, "n", 1), ws.data_ptr(), st))
        r = cfg.shuffle
        ed = L.EpiDesc(N, Ho, Wo, Cout, cfg.act, cfg.act_param, cfg.apc, r, cfg.order, int(bn is not None))
        y = _empty_cl((N, Cout // (r * r), Ho * r, Wo * r), dev)
        res = nhwc(residual) if residual is not None else None
        if res is not None and res.shape != y.shape:
            raise ValueError(f"srmi: residual shape {tuple(res.shape)} != output shape {tuple(y.shape)}")
        a = alpha.detach().contiguous() if alpha is not None else None
        g_ = gamma.detach().contiguous() if gamma is not None else None
        b_ = beta.detach().contiguous() if beta is not None else None
        L.check(lib.srmi_epilogue_fwd(C.byref(ed), z.data_ptr(), _ptr(mean), _ptr(rstd), _ptr(g_), _ptr(b_), _ptr(a),
                                      _ptr(res), y.data_ptr(), st))
        ctx.save_for_backward(x, w, z, mean, rstd, g_, b_, a)
        ctx.cfg, ctx.desc, ctx.ed, ctx.plan = cfg, desc, ed, plan
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.weight_obj = weight        # cache key of the packed dgrad operand (an input: no reference cycle)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = L.load()
        st = Fn.stream_ptr()
        x, w, z, mean, rstd, gamma, beta, alpha = ctx.saved_tensors
        cfg, desc, ed, plan = ctx.cfg, ctx.desc, ctx.ed, ctx.plan
        dev = x.device
        dy = nhwc(dy)
        Cout = desc.Cout
        dz = torch.empty_like(z)
        ws = _train_ws(Cout, dev)
        need = ctx.needs_input_grad
        # per-channel sums only where somebody wants them: BN's backward, the bias gradient, the PReLU slope gradient
        # (none of them in the frozen VGG19 stack: its epilogue backward is then one streaming launch)
        want_bias = ctx.has_bias and need[2]
        dsum1 = torch.empty(Cout, dtype=torch.float32, device=dev) if (cfg.bn is not None or want_bias) else None
        dsum2 = torch.empty(Cout, dtype=torch.float32, device=dev) if cfg.bn is not None else None
        dalpha = torch.empty_like(alpha) if (alpha is not None and need[5]) else None
        # the bias of a conv feeding train-mode BN has an exactly-zero gradient (BN removes the mean): zero-filled by the
        # finalize launch of the epilogue backward, not by a fill launch of its own
        dzero = torch.empty(Cout, dtype=torch.float32, device=dev) if (cfg.bn is not None and ctx.has_bias) else None
        L.check(lib.srmi_epilogue_bwd(C.byref(ed), z.data_ptr(), dy.data_ptr(), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                      _ptr(beta), _ptr(alpha), dz.data_ptr(), _ptr(dsum1), _ptr(dsum2),
                                      _ptr(dalpha), _ptr(dzero), ws.data_ptr(), st))
        dx = dw = dbias = dgamma = dbeta = None
        if need[1]:
            dw = torch.empty_like(w)
            wsk = _WS.get(plan.ws_wgrad, dev)
            L.check(lib.srmi_conv2d_wgrad(C.byref(desc), x.data_ptr(), dz.data_ptr(), dw.data_ptr(), _ptr(wsk),
                                          plan.ws_wgrad, st))
        if need[0]:
            def pack_dgrad():
                wd_ = torch.empty(plan.dgrad_pack_bytes, dtype=torch.uint8, device=dev)
                L.check(lib.srmi_conv2d_dgrad_pack_weights(C.byref(desc), w.data_ptr(), wd_.data_ptr(), st))
                return wd_
            wd = _PACKS.get(("dgrad", desc.H, desc.W, desc.stride, desc.pad), ctx.weight_obj, pack_dgrad)
            dx = torch.empty_like(x)
            wsk = _WS.get(plan.ws_dgrad, dev)
            L.check(lib.srmi_conv2d_dgrad(C.byref(desc), dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), _ptr(wsk),
                                          plan.ws_dgrad, st))
        if cfg.bn is not None:
            dgamma, dbeta = dsum2, dsum1
            dbias = dzero
        else:
            dbias = dsum1 if want_bias else None
        return dx, dw, dbias, dgamma, dbeta, dalpha, dy if ctx.has_res else None, None


def fused_conv(x, c2d: nn.Conv2d, bn, act, residual=None, shuffle=1, shuffle_order=L.SHUFFLE_CRD, weight=None, pad=None):
    """conv -> [BN (train)] -> [act] -> [shuffle] -> [+residual] through FusedConvFn.
    `weight` overrides c2d.weight (a differentiable view, e.g. nn.Linear's weight as OIHW)."""
    code, param, alpha, apc = Fn.act_of(act)
    if bn is not None and not bn.training:
        raise NotImplementedError("srmi: autograd through eval-mode BatchNorm is not supported - call .train() "
                                  "(the reference trains with train-mode BN, P/mode.py:101) or use torch.no_grad()")
    cfg = _Cfg(c2d.stride[0] if c2d is not None else 1,
               pad if pad is not None else c2d.padding[0], code, param, apc, shuffle, shuffle_order, bn)
    w = weight if weight is not None else c2d.weight
    b = c2d.bias if c2d is not None else None
    return FusedConvFn.apply(x, w, b, bn.weight if bn is not None else None, bn.bias if bn is not None else None,
                             alpha, residual, cfg)


def linear(x, lin: nn.Linear, act=None):
    """nn.Linear on the NCHW-flattened feature map (P/srgan_model.py:69-71) = a conv whose kernel covers the
    whole map: weight [out, C*H*W] viewed as OIHW [out, C, H, W], no padding -> [N, out, 1, 1]."""
    N, Cc, H, W = x.shape
    if lin.in_features != Cc * H * W:
        raise ValueError(f"srmi: Linear expects {lin.in_features} features, got {Cc}x{H}x{W}")
    code, param, alpha, apc = Fn.act_of(act)
    cfg = _Cfg(1, 0, code, param, apc, 1, L.SHUFFLE_CRD, None)
    return FusedConvFn.apply(x, lin.weight.view(lin.out_features, Cc, H, W), lin.bias, None, None, alpha, None, cfg)


class MaxPool2x2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = nhwc(x)
        N, Cc, H, W = x.shape
        y = _empty_cl((N, Cc, H // 2, W // 2), x.device)
        L.check(L.load().srmi_maxpool2x2_fwd(x.data_ptr(), y.data_ptr(), N, H, W, Cc, Fn.stream_ptr()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = nhwc(dy)
        N, Cc, H, W = x.shape
        dx = torch.empty_like(x)
        L.check(L.load().srmi_maxpool2x2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, H, W, Cc, Fn.stream_ptr()))
        return dx


def maxpool2x2(x):
    return MaxPool2x2Fn.apply(x)


def wants_training_path(x, *modules):
    """True when the call needs the training kernels: autograd is recording, or a BatchNorm is in train mode."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for m in modules for p in m.parameters())):
        return True
    for m in modules:
        for sub in m.modules():
            if isinstance(sub, nn.BatchNorm2d) and sub.training:
                return True
    return False
