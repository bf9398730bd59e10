"""Inference plans: a whole network forward as ONE native call that enqueues every fused conv.

A plan owns (a) the packed weights and folded scale/shift vectors of every layer, (b) the NHWC
activation buffers sized for one input shape, and (c) a ctypes array of ``srmi_conv_args`` that
``srmi_conv2d_fwd_seq`` walks from C++ - one FFI crossing per forward instead of one per layer.
Plans are cached per (device, input shape, compute mode) and rebuilt when a parameter or buffer of the
module changes (``load_state_dict``, ``.to()``, an optimizer step): the signature is the tuple of
(data_ptr, version) of every tensor the plan folded.

A write through ``tensor.data`` (``p.data.copy_()``, ``p.data.mul_()``, hand-rolled EMA updates) does NOT bump
torch's version counter, so no signature can see it: after such a write call ``srmi.invalidate(module)`` (below),
or write through ``with torch.no_grad(): p.copy_(...)`` instead, which does bump it.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as Fn


def invalidate(module: nn.Module):
    """Drop every cached plan / packed weight of `module` and its sub-modules.  Needed only after parameters or
    buffers were modified through ``.data`` (which torch's version counter cannot see); every other update
    (optimizer steps, ``load_state_dict``, ``.to()``, in-place ops under ``no_grad``) is detected automatically."""
    for m in module.modules():
        plans = m.__dict__.get("_plans")
        if plans is not None:
            plans.clear()
        cache = m.__dict__.get("_cache")
        if cache is not None and hasattr(cache, "entries"):
            cache.entries.clear()
            cache.sig = None
    from . import autograd
    autograd._PACKS.clear()
    return module


def _sig(module: nn.Module):
    s = []
    for t in list(module.parameters()) + list(module.buffers()):
        s.append((t.data_ptr(), t._version))
    return tuple(s)


def conv_parts(block):
    """Split a reference-style `conv` block (Sequential(_conv [, BN] [, act]), P/ops.py:18-31) into
    (conv2d, bn or None, act module or None)."""
    mods = list(block.body)
    c2d = mods[0]
    bn = None
    act = None
    for m in mods[1:]:
        if isinstance(m, nn.BatchNorm2d):
            bn = m
        else:
            act = m
    return c2d, bn, act


class _Seq:
    """Builder for a conv sequence."""

    def __init__(self, device, compute):
        self.device = device
        self.compute = compute
        self.adt = L.BF16 if compute == L.COMPUTE_BF16 else L.F32   # activation storage dtype
        self.tdt = Fn.torch_dtype(self.adt)
        self.layers = []   # (desc, x, w, scale, shift, alpha, residual, y)
        self.keep = []     # tensors that must stay alive
        self.kernels = []

    def buf(self, *shape, dtype=None):
        t = torch.empty(shape, dtype=dtype or self.tdt, device=self.device)
        self.keep.append(t)
        return t

    def add(self, block_or_parts, x, y, N, H, W, *, residual=None, shuffle=1, in_layout=L.NHWC, out_layout=L.NHWC,
            in_dtype=None, out_dtype=None, weight=None, k=None, stride=None, cin=None, cout=None, fuse_next=False,
            shuffle_order=L.SHUFFLE_CRD, in_cstride=0, out_cstride=0, out_coffset=0, res_cstride=0, out_scale=None,
            chain=False, y_transient=False):
        c2d, bn, act = block_or_parts if isinstance(block_or_parts, tuple) else conv_parts(block_or_parts)
        if bn is not None and bn.training:
            raise NotImplementedError("srmi: train-mode BatchNorm is not part of the inference plan")
        w = c2d.weight if weight is None else weight
        cout_ = w.shape[0] if cout is None else cout
        cin_ = w.shape[1] if cin is None else cin
        k_ = w.shape[2] if k is None else k
        stride_ = (c2d.stride[0] if hasattr(c2d, "stride") else 1) if stride is None else stride
        pad_ = c2d.padding[0] if hasattr(c2d, "padding") else 0
        code, param, alpha, apc = Fn.act_of(act)
        desc = Fn.make_desc(N, H, W, cin_, cout_, k_, stride=stride_, pad=pad_,
                            in_dtype=self.adt if in_dtype is None else in_dtype,
                            out_dtype=self.adt if out_dtype is None else out_dtype,
                            res_dtype=self.adt, in_layout=in_layout, out_layout=out_layout,
                            compute=self.compute, act=code, act_param=param, alpha_per_channel=apc,
                            shuffle=shuffle, shuffle_order=shuffle_order, in_cstride=in_cstride,
                            out_cstride=out_cstride, out_coffset=out_coffset, res_cstride=res_cstride)
        wp = Fn.pack_weights(desc, w.reshape(cout_, cin_, k_, k_))
        scale, shift = Fn.fold_bn(c2d.bias, bn, cout_, self.device)
        if out_scale is not None:          # e.g. RRDB's `x * 0.2` (ESRGAN/untitled0.py:32): folded, not a pass
            scale, shift = scale * out_scale, shift * out_scale
        if alpha is not None:
            alpha = alpha.detach()
            if alpha.dtype != torch.float32:
                raise TypeError("PReLU weight must be float32")
            if apc and alpha.numel() != cout_:
                # per-channel PReLU applied AFTER a sub-pixel shuffle: expand to the conv's channel order
                rr = shuffle * shuffle
                if alpha.numel() * rr != cout_:
                    raise ValueError("per-channel PReLU size does not match the conv")
                alpha = alpha.repeat(rr) if shuffle_order == L.SHUFFLE_DCR else alpha.repeat_interleave(rr)
                alpha = alpha.contiguous()
                self.keep.append(alpha)
        self.keep += [wp, scale, shift]
        self.layers.append((desc, x, wp, scale, shift, alpha, residual, y,
                            (L.CONV_FUSE_NEXT if fuse_next else 0) | (L.CONV_FUSE_CHAIN if chain and _FUSE_CHAIN else 0)
                            | (L.CONV_Y_TRANSIENT if chain and y_transient and _FUSE_CHAIN else 0)))
        self.kernels.append(L.kernel_name(desc))
        return y

    def finish(self):
        n = len(self.layers)
        arr = (L.ConvArgs * n)()
        for i, (desc, x, wp, scale, shift, alpha, residual, y, flags) in enumerate(self.layers):
            Fn.fill_args(arr[i], desc, x if x is not None else 0, wp, scale, shift, alpha, residual, y)
            arr[i].flags = flags
        if not _FUSE_TAIL:
            # the fused tail clamps each tap contribution of the last conv to +-900 (int32 fixed-point accumulation,
            # include/srmi.h SRMI_FUSE_UPTAIL): a net whose last conv is not range-bounded runs the pair as two launches
            lib = L.load()
            for i in range(n - 1):
                if (arr[i].flags & L.CONV_FUSE_NEXT) and lib.srmi_conv2d_can_fuse(C.byref(arr[i]), C.byref(arr[i + 1])) == 2:
                    arr[i].flags = 0
        return arr, n

    @staticmethod
    def launches(arr, n, kernels):
        """[(label, first index, count)]: what srmi_conv2d_fwd_seq will actually launch."""
        lib = L.load()
        out, i = [], 0
        while i < n:
            kind = 0
            if arr[i].flags & L.CONV_FUSE_CHAIN:
                cnt = lib.srmi_conv2d_chain_length(C.byref(arr[i]), n - i)
                if cnt:
                    out.append(("resblock_chain_bf16", i, cnt))
                    i += cnt
                    continue
            if (arr[i].flags & L.CONV_FUSE_NEXT) and i + 1 < n:
                kind = lib.srmi_conv2d_can_fuse(C.byref(arr[i]), C.byref(arr[i + 1]))
            if kind:
                out.append(({1: "resblock_bf16", 2: "upfuse_bf16", 3: "k9c1pw_bf16"}[kind], i, 2))
                i += 2
            else:
                out.append((kernels[i], i, 1))
                i += 1
        return out


_FUSE_TAIL = True
_FUSE_CHAIN = True
_X_COPY = os.environ.get("SRMI_FEAT_RELAYOUT", "0") in ("", "0")   # (tests / A-B: False = `feat` always through the stand-alone re-layout pass)


def set_fuse_tail(enabled: bool):
    """False: plans built from now on run `Upsampler` conv + last conv as TWO launches instead of the fused tail
    (whose int32 fixed-point accumulation clamps each tap contribution of the last conv to +-900, include/srmi.h).
    Plans already cached keep their form: call srmi.engine.invalidate(module) to rebuild them."""
    global _FUSE_TAIL
    _FUSE_TAIL = bool(enabled)


def set_fuse_chain(enabled: bool):
    """False: plans built from now on run every ResBlock as its own launch instead of chaining the trunk into one
    persistent launch (SRMI_CONV_FUSE_CHAIN, include/srmi.h: bit-identical results; the chain needs the whole GPU - every
    workgroup resident at once - and allocates 4 KiB per device at its first launch).  Plans already cached keep their
    form: call srmi.engine.invalidate(module) to rebuild them."""
    global _FUSE_CHAIN
    _FUSE_CHAIN = bool(enabled)


class GeneratorPlan:
    """`Generator.forward` (P/srgan_model.py:27-39) for one (N, H, W) as 37 fused conv launches."""

    def __init__(self, gen, N, H, W, compute):
        dev = next(gen.parameters()).device
        s = _Seq(dev, compute)
        self.N, self.H, self.W = N, H, W
        nf = gen.conv01.body[0].out_channels
        img = gen.conv01.body[0].in_channels
        # conv01 reads the caller's NCHW f32 tensor directly (no separate re-layout pass)
        skip = s.buf(N, H, W, nf)
        s.add(gen.conv01, None, skip, N, H, W, in_layout=L.NCHW, in_dtype=L.F32, chain=True)     # (may be the chained trunk's head link)
        cur = skip
        tmp = s.buf(N, H, W, nf)
        ping = [s.buf(N, H, W, nf), s.buf(N, H, W, nf)]
        for i, blk in enumerate(gen.body):         # ResBlock, P/ops.py:45-48
            # the pair may run as ONE fused launch (tmp never written); ping-pong outputs because the
            # fused kernel's neighbours read the input's halo while this tile's output is stored
            out = ping[i & 1]
            # (every block's output is read by the next conv of the sequence only: inside a chained launch just its
            #  halo strips reach memory)
            s.add(blk.body[0], cur, tmp, N, H, W, fuse_next=True, chain=True, y_transient=True)
            s.add(blk.body[1], tmp, out, N, H, W, residual=cur)
            cur = out
        feat = s.buf(N, H, W, nf)
        # `feat` leaves as NCHW f32 (P/srgan_model.py:39).  The first Upsampler conv holds it in LDS anyway: when that conv
        # is a plain c64_bf16 launch, or the last link of the chained trunk, it writes the caller's copy on the side
        # (srmi_conv_args.x_copy) and the separate re-layout pass disappears - inside the chain the NHWC bf16 `feat` is then
        # read by the next link only (y_transient).  Otherwise (f32 compute, a single Upsampler fused into the tail, odd
        # widths) the stand-alone srmi_layout_convert runs.
        want_xcopy = compute == L.COMPUTE_BF16 and W % 4 == 0 and nf == 64 and _X_COPY
        s.add(gen.conv02, cur, feat, N, H, W, residual=skip, chain=True, y_transient=want_xcopy)   # P/srgan_model.py:33-34 (a tail link of the chain)
        x, h, w = feat, H, W
        for j, up in enumerate(gen.tail):          # Upsampler, P/ops.py:74-78
            r = up.body[1].upscale_factor
            c2d = up.body[0].body[0]
            act = up.body[2] if len(up.body) > 2 else None
            y = s.buf(N, h * r, w * r, c2d.out_channels // (r * r))
            # the last Upsampler and last_conv may run as ONE launch (its 4x-resolution output never reaches HBM); the first
            # one of two may be the chain's last link
            last = j == len(gen.tail) - 1
            s.add((c2d, None, act), x, y, N, h, w, shuffle=r, fuse_next=last, chain=(j == 0 and not last))
            x, h, w = y, h * r, w * r
        self.sr_shape = (N, img, h, w)
        sr0 = torch.empty(self.sr_shape, dtype=torch.float32, device=dev)   # placeholder; run() allocates fresh outputs
        s.add(gen.last_conv, x, sr0, N, h, w, out_layout=L.NCHW, out_dtype=L.F32)   # conv + tanh -> NCHW f32
        self.feat_nhwc = feat
        self.nf = nf
        self.dev = dev
        self.args, self.n = s.finish()
        self.kernels = s.kernels
        self._seq = s
        self.args[0].x = skip.data_ptr()          # placeholder so fusability can be queried before run()
        self.launches = _Seq.launches(self.args, self.n, self.kernels)
        self.feat_idx = None
        if want_xcopy:
            idx = self.n - 1 - len(gen.tail)                     # the conv that reads feat: first conv of the tail
            for (lbl, i0, cnt) in self.launches:
                if (lbl == "c64_bf16" and i0 == idx and cnt == 1) or (lbl == "resblock_chain_bf16" and i0 + cnt - 1 == idx):
                    self.feat_idx = idx
            if self.feat_idx is None:
                # nobody writes the NCHW copy: the NHWC tensor must be whole for srmi_layout_convert
                self.args[idx - 1].flags &= ~L.CONV_Y_TRANSIENT

    def run(self, x):
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        self._x = x
        # outputs are fresh tensors per call (a drop-in must not alias the previous call's result)
        sr = torch.empty(self.sr_shape, dtype=torch.float32, device=self.dev)
        self.args[0].x = x.data_ptr()
        self.args[self.n - 1].y = sr.data_ptr()
        if self.feat_idx is not None:
            feat = torch.empty((self.N, self.nf, self.H, self.W), dtype=torch.float32, device=self.dev)
            self.args[self.feat_idx].x_copy = feat.data_ptr()
            L.check(L.load().srmi_conv2d_fwd_seq(self.args, self.n, Fn.stream_ptr()))
            return sr, feat
        L.check(L.load().srmi_conv2d_fwd_seq(self.args, self.n, Fn.stream_ptr()))
        feat = Fn.layout_convert(self.feat_nhwc, self.N, self.nf, self.H, self.W, L.NHWC, L.NCHW, torch.float32)
        return sr, feat


class DiscriminatorPlan:
    """`Discriminator.forward` in eval mode (P/srgan_model.py:65-73): 8 convs + 2 Linear as 1x1 convs."""

    def __init__(self, dis, N, H, W, compute):
        dev = next(dis.parameters()).device
        s = _Seq(dev, compute)
        self.N = N
        act = dis.act
        blocks = [dis.conv01, dis.conv02]
        for b in dis.body:
            blocks += [b.body[0], b.body[1]]
        x, h, w, first = None, H, W, True
        for blk in blocks:
            c2d = blk.body[0]
            st = c2d.stride[0]
            ho, wo = (h + 2 * c2d.padding[0] - c2d.kernel_size[0]) // st + 1, (w + 2 * c2d.padding[0] - c2d.kernel_size[0]) // st + 1
            y = s.buf(N, ho, wo, c2d.out_channels)
            if first:
                s.add(blk, None, y, N, h, w, in_layout=L.NCHW, in_dtype=L.F32)
                first = False
            else:
                s.add(blk, x, y, N, h, w)
            x, h, w = y, ho, wo
        cl = x.shape[-1]
        lin1, lin2 = dis.tail[0], dis.tail[2]
        if lin1.in_features != cl * h * w:
            raise RuntimeError(f"Discriminator: input {H}x{W} does not match linear_size {lin1.in_features}")
        # flatten is x.view(-1, C*H*W) on NCHW (P/srgan_model.py:70); our buffer is NHWC, so permute the
        # Linear's columns (c,h,w) -> (h,w,c) once at pack time.
        w1 = lin1.weight.detach().view(lin1.out_features, cl, h, w).permute(0, 2, 3, 1).reshape(lin1.out_features, -1)
        y1 = s.buf(N, 1, 1, lin1.out_features)
        s.add((lin1, None, act), x, y1, N, 1, 1, weight=w1.contiguous(), k=1, stride=1, cin=lin1.in_features,
              cout=lin1.out_features)
        self.prob_shape = (N, lin2.out_features)
        self.dev = dev
        prob0 = torch.empty(self.prob_shape, dtype=torch.float32, device=dev)
        s.add((lin2, None, dis.tail[3]), y1, prob0, N, 1, 1, weight=lin2.weight.detach(), k=1, stride=1,
              cin=lin2.in_features, cout=lin2.out_features, out_dtype=L.F32)
        self.args, self.n = s.finish()
        self.kernels = s.kernels
        self._seq = s

    def run(self, x):
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        self._x = x
        prob = torch.empty(self.prob_shape, dtype=torch.float32, device=self.dev)
        self.args[0].x = x.data_ptr()
        self.args[self.n - 1].y = prob.data_ptr()
        L.check(L.load().srmi_conv2d_fwd_seq(self.args, self.n, Fn.stream_ptr()))
        return prob


class PlanCache:
    """Plans per (device, stream, input shape, compute mode).  A plan owns its activation buffers and its argument
    array, so two forwards that share a plan must not overlap: the device AND the stream are part of the key (forwards
    of one module on two streams get a plan each; nn.DataParallel replicas share this object through the __dict__
    copy and are told apart by the device), and the table itself is guarded by a lock.  Two host threads driving the
    SAME stream through the same module still serialise on the GPU, but their `run()` calls rewrite one argument
    array: give each thread its own stream."""

    def __init__(self, factory):
        self.factory = factory
        self.plans = {}
        self.lock = threading.Lock()

    def get(self, module, N, H, W, compute):
        dev = next(module.parameters()).device
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        key = (dev.index, stream, N, H, W, compute)
        sig = _sig(module)
        with self.lock:
            hit = self.plans.get(key)
            if hit is not None and hit[0] == sig:
                return hit[1]
            if hit is None and len(self.plans) >= 8:   # bound the buffers held for stale shapes / streams
                self.plans.pop(next(iter(self.plans)))
            plan = self.factory(module, N, H, W, compute)
            self.plans[key] = (sig, plan)
            return plan

    def clear(self):
        with self.lock:
            self.plans.clear()

    # plans hold device buffers and ctypes pointers: never copied or pickled with the module
    def __deepcopy__(self, memo):
        return PlanCache(self.factory)

    def __getstate__(self):
        return {"factory": self.factory, "plans": {}}

    def __setstate__(self, state):
        self.factory, self.plans = state["factory"], {}
        self.lock = threading.Lock()
