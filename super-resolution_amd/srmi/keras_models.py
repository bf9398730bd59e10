"""The reference's Keras generators as torch modules on libsrmi's kernels:

  * `SRCNN`   - SRCNN/srcnn.py:199-210: conv9x9(1->128)+ReLU, conv3x3(128->64)+ReLU, conv5x5(64->1); BASELINE config 2
                ("9-1-5, 64/32") is `SRCNN(k=(9, 1, 5), ch=(64, 32))`.
  * `RRDBNet` - ESRGAN/untitled0.py:18-67 (same in "25-ESRGAN lpips/esrgan.py:34-85"): conv3x3+PReLU(per channel) ·
                N x [3 x (conv3x3 -> 32, LeakyReLU 0.2, concat) -> conv3x3(160 -> 64) x 0.2 + x] · conv3x3 + first ·
                2 x [conv3x3(64 -> 256), depth_to_space(2), PReLU(per channel)] · conv3x3(64 -> 3) tanh.
                The reference builds 16 blocks; BASELINE config 4 asks 23: `RRDBNet(num_blocks=23)`.

  * `KerasSRGAN` - "Typical SRGAN with tensorflow/typical srgan .py":21-71: the SRResNet generator in its Keras layout:
                per-channel PReLU everywhere, BatchNormalization eps 1e-3, PReLU on the 256 conv channels BEFORE
                depth_to_space, and a 9x9 output conv (64 -> 3; no LDS-tiled kernel for that shape yet: it runs on
                the generic f32-MFMA kernel).

  * `UNetSRGAN` - U-SRGAN/srganunet.py:14-126: 9x9 head, a 4-level U-Net body (conv blocks 64..1024, MaxPooling2D,
                nearest UpSampling2D, channel concatenation), concat with the head output, two 1024-channel sub-pixel
                convs and a 9x9 output conv.  Concatenations are never copies (channel-slice stores); pooling and
                upsampling are the two addressing kernels `srmi_maxpool2x2_nhwc` / `srmi_upsample2x_nhwc`.

  * `AutoencoderSRGAN` - A-SRGAN/srganautoencoders.py:55-144: stacked encoder/decoder blocks (conv + per-channel PReLU,
                MaxPooling2D, UpSampling2D, additive skips fused into the conv epilogues) between the SRGAN head and tail.

  * `ResUNetSRGAN` - Res-U-SRGAN/untitled0.py:39-125, 190-226: the U-Net with residual conv blocks (conv-BN-ReLU-conv-BN plus
                a 1x1 conv-BN shortcut, ReLU after the add = the SRMI_ACT_ADD_RELU epilogue).

The Keras scripts are not importable (no TensorFlow here) and build their graph inline, so there is no module API
to mirror; parameters are ordinary `nn.Conv2d` (OIHW) so torch tooling works, and `from_keras_hwio` converts a Keras
kernel.  Tensors are NCHW like the rest of this package (Keras is NHWC).  The channel concat of the dense block is
never materialised as a copy: every dense conv writes its 32 channels into a slice of one 160-channel NHWC buffer
(`out_cstride/out_coffset`) and reads the leading channels of the same buffer (`in_cstride`).
Parity: unpinned (see oracle/keras_oracle.py); tests compare with that restatement.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as Fn
from .engine import PlanCache, _Seq
from .ops import _guard
from .srgan_model import _COMPUTE, _ComputeMixin


def from_keras_hwio(w):
    """Keras Conv2D kernel (kh, kw, in, out) -> torch OIHW."""
    return torch.as_tensor(w).permute(3, 2, 0, 1).contiguous()


def _c(cin, cout, k):
    return nn.Conv2d(cin, cout, k, padding=k // 2, bias=True)


class _Plan:
    def run(self, x):
        x = x.contiguous().float()
        self._x = x
        y = torch.empty(self.out_shape, dtype=torch.float32, device=self.dev)
        self.args[0].x = x.data_ptr()
        self.args[self.n - 1].y = y.data_ptr()
        L.check(L.load().srmi_conv2d_fwd_seq(self.args, self.n, Fn.stream_ptr()))
        return y


class SRCNNPlan(_Plan):
    def __init__(self, m, N, H, W, compute):
        dev = next(m.parameters()).device
        s = _Seq(dev, compute)
        relu = nn.ReLU()
        b1 = s.buf(N, H, W, m.conv1.out_channels)
        # (9-1-5 with 64 / 32 channels in bf16: conv1 + conv2 go out as ONE launch, "k9c1pw_bf16"; b1 is then never written)
        s.add((m.conv1, None, relu), None, b1, N, H, W, in_layout=L.NCHW, in_dtype=L.F32, fuse_next=True)
        b2 = s.buf(N, H, W, m.conv2.out_channels)
        s.add((m.conv2, None, relu), b1, b2, N, H, W)
        self.out_shape = (N, m.conv3.out_channels, H, W)
        y0 = torch.empty(self.out_shape, dtype=torch.float32, device=dev)
        s.add((m.conv3, None, None), b2, y0, N, H, W, out_layout=L.NCHW, out_dtype=L.F32)
        self.dev = dev
        self.args, self.n = s.finish()
        self.kernels, self._seq = s.kernels, s


class SRCNN(nn.Module, _ComputeMixin):
    def __init__(self, k=(9, 3, 5), ch=(128, 64), in_ch=1):
        super().__init__()
        self.conv1 = _c(in_ch, ch[0], k[0])
        self.conv2 = _c(ch[0], ch[1], k[1])
        self.conv3 = _c(ch[1], in_ch, k[2])
        for c in (self.conv1, self.conv2, self.conv3):     # Keras: glorot_uniform kernel, zero bias
            nn.init.xavier_uniform_(c.weight)
            nn.init.zeros_(c.bias)
        self._plans = PlanCache(SRCNNPlan)

    def forward(self, x):
        _guard(x, self)
        N, _, H, W = x.shape
        return self._plans.get(self, N, H, W, _COMPUTE[self.compute]).run(x)


class RRDBPlan(_Plan):
    def __init__(self, m, N, H, W, compute):
        dev = next(m.parameters()).device
        s = _Seq(dev, compute)
        nf, g, nd = m.n_feats, m.growth, m.num_dense_layers
        ctot = nf + nd * g
        lrelu = nn.LeakyReLU(0.2)
        # block b works in a concat buffer: x in channels [0, nf), dense outputs appended behind it.  Block 0 has a buffer
        # of its own whose first nf channels - the head output - are never overwritten: they are also the long skip
        # (a channel-slice residual of the trunk conv), so the head conv runs once; blocks 1.. alternate between two.
        nb = len(m.blocks)
        cat = [s.buf(N, H, W, ctot), s.buf(N, H, W, ctot)] if nb > 1 else []
        cat0 = s.buf(N, H, W, ctot) if nb else None
        first = cat0 if nb else s.buf(N, H, W, nf)
        first_cs = ctot if nb else 0
        s.add((m.head, None, m.head_act), None, first, N, H, W, in_layout=L.NCHW, in_dtype=L.F32,
              out_cstride=first_cs, out_coffset=0)
        for b, blk in enumerate(m.blocks):
            buf = cat0 if b == 0 else cat[b & 1]
            for i, dc in enumerate(blk.dense):
                s.add((dc, None, lrelu), buf, buf, N, H, W, in_cstride=ctot, out_cstride=ctot, out_coffset=nf + i * g)
            # compress: (conv + b) * 0.2 + x ; x = buf[..., :nf] (a channel slice -> res_cstride)
            last = b == nb - 1
            dst = s.buf(N, H, W, nf) if last else cat[(b + 1) & 1]
            s.add((blk.compress, None, None), buf, dst, N, H, W, in_cstride=ctot, residual=buf, res_cstride=ctot,
                  out_cstride=nf if last else ctot, out_coffset=0, out_scale=0.2)
            trunk_in = dst
        if not nb:
            trunk_in = first
        t = s.buf(N, H, W, nf)
        s.add((m.trunk, None, None), trunk_in, t, N, H, W, residual=first, res_cstride=first_cs)
        x, h, w = t, H, W
        for j in range(2):
            y = s.buf(N, 2 * h, 2 * w, nf)
            # the second up conv and the tail conv may run as ONE launch (srmi_conv2d_fwd_pair, SRMI_FUSE_UPTAIL)
            s.add((m.up[j], None, m.up_act[j]), x, y, N, h, w, shuffle=2, shuffle_order=L.SHUFFLE_DCR, fuse_next=(j == 1))
            x, h, w = y, 2 * h, 2 * w
        self.out_shape = (N, m.tail.out_channels, h, w)
        y0 = torch.empty(self.out_shape, dtype=torch.float32, device=dev)
        s.add((m.tail, None, nn.Tanh()), x, y0, N, h, w, out_layout=L.NCHW, out_dtype=L.F32)
        self.dev = dev
        self.args, self.n = s.finish()
        self.kernels, self._seq = s.kernels, s


class _RRDB(nn.Module):
    def __init__(self, nf, growth, nd):
        super().__init__()
        self.dense = nn.ModuleList([_c(nf + i * growth, growth, 3) for i in range(nd)])
        self.compress = _c(nf + nd * growth, nf, 3)


class RRDBNet(nn.Module, _ComputeMixin):
    def __init__(self, num_blocks=16, n_feats=64, growth=32, num_dense_layers=3, img_ch=3):
        super().__init__()
        self.n_feats, self.growth, self.num_dense_layers = n_feats, growth, num_dense_layers
        self.head = _c(img_ch, n_feats, 3)
        self.head_act = nn.PReLU(n_feats, init=0.0)            # Keras PReLU: alpha per channel, initialised to 0
        self.blocks = nn.ModuleList([_RRDB(n_feats, growth, num_dense_layers) for _ in range(num_blocks)])
        self.trunk = _c(n_feats, n_feats, 3)
        self.up = nn.ModuleList([_c(n_feats, 4 * n_feats, 3) for _ in range(2)])
        self.up_act = nn.ModuleList([nn.PReLU(n_feats, init=0.0) for _ in range(2)])
        self.tail = _c(n_feats, img_ch, 3)
        self._plans = PlanCache(RRDBPlan)

    def forward(self, x):
        _guard(x, self)
        N, _, H, W = x.shape
        return self._plans.get(self, N, H, W, _COMPUTE[self.compute]).run(x)


class _KResBlock(nn.Module):
    def __init__(self, nf):
        super().__init__()
        self.conv1 = _c(nf, nf, 3)
        self.bn1 = nn.BatchNorm2d(nf, eps=1e-3, momentum=0.01)       # Keras: epsilon 1e-3, momentum 0.99
        self.act = nn.PReLU(nf, init=0.0)
        self.conv2 = _c(nf, nf, 3)
        self.bn2 = nn.BatchNorm2d(nf, eps=1e-3, momentum=0.01)


class KerasSRGANPlan(_Plan):
    def __init__(self, m, N, H, W, compute):
        dev = next(m.parameters()).device
        s = _Seq(dev, compute)
        nf = m.n_feats
        first = s.buf(N, H, W, nf)
        s.add((m.head, None, m.head_act), None, first, N, H, W, in_layout=L.NCHW, in_dtype=L.F32, chain=True)
        cur, tmp, ping = first, s.buf(N, H, W, nf), [s.buf(N, H, W, nf), s.buf(N, H, W, nf)]
        for i, blk in enumerate(m.blocks):
            out = ping[i & 1]
            # one fused launch per block, or - where the whole grid fits the GPU - all blocks and `post` as one chained launch
            s.add((blk.conv1, blk.bn1, blk.act), cur, tmp, N, H, W, fuse_next=True, chain=True, y_transient=True)
            s.add((blk.conv2, blk.bn2, None), tmp, out, N, H, W, residual=cur)
            cur = out
        t = s.buf(N, H, W, nf)
        s.add((m.post, m.post_bn, None), cur, t, N, H, W, residual=first, chain=True)
        x, h, w = t, H, W
        for j in range(2):
            y = s.buf(N, 2 * h, 2 * w, nf)
            # the second up conv and the tail conv may run as ONE launch (srmi_conv2d_fwd_pair, SRMI_FUSE_UPTAIL)
            s.add((m.up[j], None, m.up_act[j]), x, y, N, h, w, shuffle=2, shuffle_order=L.SHUFFLE_DCR, fuse_next=(j == 1))
            x, h, w = y, 2 * h, 2 * w
        self.out_shape = (N, m.tail.out_channels, h, w)
        y0 = torch.empty(self.out_shape, dtype=torch.float32, device=dev)
        s.add((m.tail, None, nn.Tanh()), x, y0, N, h, w, out_layout=L.NCHW, out_dtype=L.F32)
        self.dev = dev
        self.args, self.n = s.finish()
        self.kernels, self._seq = s.kernels, s


class KerasSRGAN(nn.Module, _ComputeMixin):
    def __init__(self, num_res_blocks=16, n_feats=64, img_ch=3):
        super().__init__()
        self.n_feats = n_feats
        self.head = _c(img_ch, n_feats, 9)
        self.head_act = nn.PReLU(n_feats, init=0.0)
        self.blocks = nn.ModuleList([_KResBlock(n_feats) for _ in range(num_res_blocks)])
        self.post = _c(n_feats, n_feats, 3)
        self.post_bn = nn.BatchNorm2d(n_feats, eps=1e-3, momentum=0.01)
        self.up = nn.ModuleList([_c(n_feats, 4 * n_feats, 3) for _ in range(2)])
        self.up_act = nn.ModuleList([nn.PReLU(4 * n_feats, init=0.0) for _ in range(2)])     # slopes of the conv channels
        self.tail = _c(n_feats, img_ch, 9)
        self._plans = PlanCache(KerasSRGANPlan)

    def forward(self, x):
        _guard(x, self)
        N, _, H, W = x.shape
        return self._plans.get(self, N, H, W, _COMPUTE[self.compute]).run(x)


class _ConvBlock(nn.Module):
    """`conv_block` (U-SRGAN/srganunet.py:14-30)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = _c(cin, cout, 3)
        self.bn1 = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01)
        self.conv2 = _c(cout, cout, 3)
        self.bn2 = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01)


class _UNet(nn.Module):
    def __init__(self, nf=64, depth=4):
        super().__init__()
        w = [nf * (1 << d) for d in range(depth + 1)]                       # 64, 128, 256, 512, 1024
        self.down = nn.ModuleList([_ConvBlock(nf if d == 0 else w[d - 1], w[d]) for d in range(depth)])
        self.mid = _ConvBlock(w[depth - 1], w[depth])
        self.up = nn.ModuleList([_ConvBlock(w[depth - d] + w[depth - 1 - d], w[depth - 1 - d]) for d in range(depth)])
        self.final = _c(nf, nf, 1)


class UNetSRGANPlan:
    """Conv runs go out through `srmi_conv2d_fwd_seq`; between them the pooling / upsampling launches.  Every
    concatenation is a buffer whose producers write channel slices: level L of the U-Net owns
    cat_L = [ upsampled decoder input (2*cw) | encoder skip (cw) ] and the tail owns [ post conv (nf) | head output (nf) ]."""

    def __init__(self, m, N, H, W, compute):
        dev = next(m.parameters()).device
        depth = len(m.unet.down)
        if H % (1 << depth) or W % (1 << depth):
            raise ValueError(f"UNetSRGAN: input {H}x{W} must be divisible by {1 << depth} (Keras' concatenate needs equal sizes)")
        s = _Seq(dev, compute)
        relu = nn.ReLU()
        nf = m.n_feats
        self.steps = []          # ("conv", first layer index, count) | ("pool", src, dst, N, h, w, C, in_cs) | ("up", ...)
        mark = [0]

        def flush():
            n = len(s.layers) - mark[0]
            if n:
                self.steps.append(("conv", mark[0], n))
            mark[0] = len(s.layers)

        tailcat = s.buf(N, H, W, 2 * nf)
        s.add((m.head, None, m.head_act), None, tailcat, N, H, W, in_layout=L.NCHW, in_dtype=L.F32, out_cstride=2 * nf,
              out_coffset=nf)
        x, xcs, h, w = tailcat[..., nf:], 2 * nf, H, W           # a channel-slice view: data_ptr() is the slice start
        cats = []
        for blk in m.unet.down:
            cw = blk.conv1.out_channels
            cat = s.buf(N, h, w, 3 * cw)
            t = s.buf(N, h, w, cw)
            s.add((blk.conv1, blk.bn1, relu), x, t, N, h, w, in_cstride=xcs)
            s.add((blk.conv2, blk.bn2, relu), t, cat, N, h, w, out_cstride=3 * cw, out_coffset=2 * cw)
            flush()
            pooled = s.buf(N, h // 2, w // 2, cw)
            self.steps.append(("pool", cat[..., 2 * cw:], pooled, N, h, w, cw, 3 * cw))
            cats.append((cat, cw, h, w))
            x, xcs, h, w = pooled, 0, h // 2, w // 2
        t = s.buf(N, h, w, m.unet.mid.conv1.out_channels)
        y = s.buf(N, h, w, m.unet.mid.conv2.out_channels)
        s.add((m.unet.mid.conv1, m.unet.mid.bn1, relu), x, t, N, h, w)
        s.add((m.unet.mid.conv2, m.unet.mid.bn2, relu), t, y, N, h, w)
        flush()
        x = y
        for d, blk in enumerate(m.unet.up):
            cat, cw, ch, cwid = cats[depth - 1 - d]
            assert x.shape[-1] == 2 * cw and (ch, cwid) == (2 * h, 2 * w)
            flush()                                              # the convs that produce x run before it is upsampled
            self.steps.append(("up", x, cat, N, h, w, 2 * cw, 3 * cw, 0))
            h, w = ch, cwid
            t = s.buf(N, h, w, blk.conv1.out_channels)
            y = s.buf(N, h, w, blk.conv2.out_channels)
            s.add((blk.conv1, blk.bn1, relu), cat, t, N, h, w)
            s.add((blk.conv2, blk.bn2, relu), t, y, N, h, w)
            x = y
        u = s.buf(N, h, w, nf)
        s.add((m.unet.final, None, None), x, u, N, h, w)
        s.add((m.post, m.post_bn, None), u, tailcat, N, h, w, out_cstride=2 * nf, out_coffset=0)
        x = tailcat
        for j in range(2):
            co = m.up[j].out_channels
            y = s.buf(N, 2 * h, 2 * w, co // 4)
            # the second up conv and the tail conv may run as ONE launch (srmi_conv2d_fwd_pair, SRMI_FUSE_UPTAIL)
            s.add((m.up[j], None, m.up_act[j]), x, y, N, h, w, shuffle=2, shuffle_order=L.SHUFFLE_DCR, fuse_next=(j == 1))
            x, h, w = y, 2 * h, 2 * w
        self.out_shape = (N, m.tail.out_channels, h, w)
        y0 = torch.empty(self.out_shape, dtype=torch.float32, device=dev)
        s.add((m.tail, None, nn.Tanh()), x, y0, N, h, w, out_layout=L.NCHW, out_dtype=L.F32)
        flush()
        self.dev, self.adt = dev, s.adt
        self.args, self.n = s.finish()
        self.kernels, self._seq = s.kernels, s

    def run(self, x):
        import ctypes as C
        lib = L.load()
        x = x.contiguous().float()
        self._x = x
        y = torch.empty(self.out_shape, dtype=torch.float32, device=self.dev)
        self.args[0].x = x.data_ptr()
        self.args[self.n - 1].y = y.data_ptr()
        st = Fn.stream_ptr()
        for step in self.steps:
            if step[0] == "conv":
                _, i0, n = step
                sub = C.cast(C.byref(self.args, i0 * C.sizeof(L.ConvArgs)), C.POINTER(L.ConvArgs))
                L.check(lib.srmi_conv2d_fwd_seq(sub, n, st))
            elif step[0] == "pool":
                _, src, dst, N, h, w, c, cs = step
                L.check(lib.srmi_maxpool2x2_nhwc(src.data_ptr(), dst.data_ptr(), N, h, w, c, cs, self.adt, st))
            else:
                _, src, dst, N, h, w, c, cs, co = step
                L.check(lib.srmi_upsample2x_nhwc(src.data_ptr(), dst.data_ptr(), N, h, w, c, cs, co, self.adt, st))
        return y


class UNetSRGAN(nn.Module, _ComputeMixin):
    def __init__(self, n_feats=64, depth=4, img_ch=3, up_feats=256):
        super().__init__()
        self.n_feats = n_feats
        self.head = _c(img_ch, n_feats, 9)
        self.head_act = nn.PReLU(n_feats, init=0.0)
        self.unet = _UNet(n_feats, depth)
        self.post = _c(n_feats, n_feats, 3)
        self.post_bn = nn.BatchNorm2d(n_feats, eps=1e-3, momentum=0.01)
        self.up = nn.ModuleList([_c(2 * n_feats, 4 * up_feats, 3), _c(up_feats, 4 * up_feats, 3)])
        self.up_act = nn.ModuleList([nn.PReLU(4 * up_feats, init=0.0) for _ in range(2)])
        self.tail = _c(up_feats, img_ch, 9)
        self._plans = PlanCache(UNetSRGANPlan)

    def forward(self, x):
        _guard(x, self)
        N, _, H, W = x.shape
        return self._plans.get(self, N, H, W, _COMPUTE[self.compute]).run(x)


class _AEBlock(nn.Module):
    _LAYERS = (("e1", 0, 0), ("e2", 0, 0), ("e4", 0, 1), ("e5", 1, 1), ("b", 1, 2), ("d2", 2, 1), ("d3", 1, 1), ("d5", 1, 0), ("d6", 0, 0))

    def __init__(self, nf_in, f):
        super().__init__()
        for name, i, o in self._LAYERS:
            cin = nf_in if name == "e1" else f[i]
            setattr(self, name, _c(cin, f[o], 3))
            setattr(self, name + "_act", nn.PReLU(f[o], init=0.0))


class AutoencoderSRGANPlan(UNetSRGANPlan):
    def __init__(self, m, N, H, W, compute):
        dev = next(m.parameters()).device
        if H % 4 or W % 4:
            raise ValueError(f"AutoencoderSRGAN: input {H}x{W} must be divisible by 4 (the additive skips need equal sizes)")
        s = _Seq(dev, compute)
        f = m.num_filters
        self.steps = []
        mark = [0]

        def flush():
            n = len(s.layers) - mark[0]
            if n:
                self.steps.append(("conv", mark[0], n))
            mark[0] = len(s.layers)

        def conv(blk, name, x, h, w, residual=None):
            c2d = getattr(blk, name)
            y = s.buf(N, h, w, c2d.out_channels)
            s.add((c2d, None, getattr(blk, name + "_act")), x, y, N, h, w, residual=residual)
            return y

        def pool(x, h, w):
            flush()
            y = s.buf(N, h // 2, w // 2, x.shape[-1])
            self.steps.append(("pool", x, y, N, h, w, x.shape[-1], 0))
            return y

        def up(x, h, w):
            flush()
            y = s.buf(N, 2 * h, 2 * w, x.shape[-1])
            self.steps.append(("up", x, y, N, h, w, x.shape[-1], 0, 0))
            return y

        first = s.buf(N, H, W, f[0])
        s.add((m.head, None, m.head_act), None, first, N, H, W, in_layout=L.NCHW, in_dtype=L.F32)
        x = first
        for blk in m.blocks:
            e2 = conv(blk, "e2", conv(blk, "e1", x, H, W), H, W)
            e5 = conv(blk, "e5", conv(blk, "e4", pool(e2, H, W), H // 2, W // 2), H // 2, W // 2)
            b = conv(blk, "b", pool(e5, H // 2, W // 2), H // 4, W // 4)
            d3 = conv(blk, "d3", conv(blk, "d2", up(b, H // 4, W // 4), H // 2, W // 2), H // 2, W // 2, residual=e5)
            x = conv(blk, "d6", conv(blk, "d5", up(d3, H // 2, W // 2), H, W), H, W, residual=e2)
        t = s.buf(N, H, W, m.post.out_channels)
        s.add((m.post, m.post_bn, None), x, t, N, H, W, residual=first)
        x, h, w = t, H, W
        for j in range(2):
            co = m.up[j].out_channels
            y = s.buf(N, 2 * h, 2 * w, co // 4)
            # the second up conv and the tail conv may run as ONE launch (srmi_conv2d_fwd_pair, SRMI_FUSE_UPTAIL)
            s.add((m.up[j], None, m.up_act[j]), x, y, N, h, w, shuffle=2, shuffle_order=L.SHUFFLE_DCR, fuse_next=(j == 1))
            x, h, w = y, 2 * h, 2 * w
        self.out_shape = (N, m.tail.out_channels, h, w)
        y0 = torch.empty(self.out_shape, dtype=torch.float32, device=dev)
        s.add((m.tail, None, nn.Tanh()), x, y0, N, h, w, out_layout=L.NCHW, out_dtype=L.F32)
        flush()
        self.dev, self.adt = dev, s.adt
        self.args, self.n = s.finish()
        self.kernels, self._seq = s.kernels, s


class AutoencoderSRGAN(nn.Module, _ComputeMixin):
    def __init__(self, num_autoencoders=1, num_filters=(64, 128, 256), img_ch=3):
        super().__init__()
        f = tuple(num_filters)
        self.num_filters = f
        self.head = _c(img_ch, f[0], 9)
        self.head_act = nn.PReLU(f[0], init=0.0)
        self.blocks = nn.ModuleList([_AEBlock(f[0], f) for _ in range(num_autoencoders)])
        self.post = _c(f[0], 64, 3)
        self.post_bn = nn.BatchNorm2d(64, eps=1e-3, momentum=0.01)
        self.up = nn.ModuleList([_c(64, 4 * f[2], 3), _c(f[2], 4 * f[2], 3)])
        self.up_act = nn.ModuleList([nn.PReLU(4 * f[2], init=0.0) for _ in range(2)])
        self.tail = _c(f[2], img_ch, 9)
        self._plans = PlanCache(AutoencoderSRGANPlan)

    def forward(self, x):
        _guard(x, self)
        N, _, H, W = x.shape
        return self._plans.get(self, N, H, W, _COMPUTE[self.compute]).run(x)


class _ResConvBlock(nn.Module):
    """`res_conv_block` (Res-U-SRGAN/untitled0.py:39-73)."""

    def __init__(self, cin, cout):
        super().__init__()
        bn = lambda: nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01)
        self.conv1, self.bn1 = _c(cin, cout, 3), bn()
        self.conv2, self.bn2 = _c(cout, cout, 3), bn()
        self.shortcut, self.bn_s = _c(cin, cout, 1), bn()


class _ResUNet(nn.Module):
    def __init__(self, nf=64, depth=4):
        super().__init__()
        w = [nf * (1 << d) for d in range(depth + 1)]
        self.down = nn.ModuleList([_ResConvBlock(nf if d == 0 else w[d - 1], w[d]) for d in range(depth)])
        self.mid = _ResConvBlock(w[depth - 1], w[depth])
        self.up = nn.ModuleList([_ResConvBlock(w[depth - d] + w[depth - 1 - d], w[depth - 1 - d]) for d in range(depth)])
        self.final = _c(nf, nf, 1)


class ResUNetSRGANPlan(UNetSRGANPlan):
    def __init__(self, m, N, H, W, compute):
        dev = next(m.parameters()).device
        depth = len(m.unet.down)
        if H % (1 << depth) or W % (1 << depth):
            raise ValueError(f"ResUNetSRGAN: input {H}x{W} must be divisible by {1 << depth} (Keras' concatenate needs equal sizes)")
        s = _Seq(dev, compute)
        relu, add_relu = nn.ReLU(), Fn.AddReLU()
        nf = m.n_feats
        self.steps = []
        mark = [0]

        def flush():
            n = len(s.layers) - mark[0]
            if n:
                self.steps.append(("conv", mark[0], n))
            mark[0] = len(s.layers)

        def block(blk, x, h, w, y, in_cs=0, **out):
            cw = blk.conv1.out_channels
            sc, t = s.buf(N, h, w, cw), s.buf(N, h, w, cw)
            s.add((blk.shortcut, blk.bn_s, None), x, sc, N, h, w, in_cstride=in_cs)
            s.add((blk.conv1, blk.bn1, relu), x, t, N, h, w, in_cstride=in_cs)
            s.add((blk.conv2, blk.bn2, add_relu), t, y, N, h, w, residual=sc, **out)

        first = s.buf(N, H, W, nf)
        s.add((m.head, None, m.head_act), None, first, N, H, W, in_layout=L.NCHW, in_dtype=L.F32)
        x, h, w = first, H, W
        cats = []
        for blk in m.unet.down:
            cw = blk.conv1.out_channels
            cat = s.buf(N, h, w, 3 * cw)
            block(blk, x, h, w, cat, out_cstride=3 * cw, out_coffset=2 * cw)
            flush()
            pooled = s.buf(N, h // 2, w // 2, cw)
            self.steps.append(("pool", cat[..., 2 * cw:], pooled, N, h, w, cw, 3 * cw))
            cats.append((cat, cw, h, w))
            x, h, w = pooled, h // 2, w // 2
        y = s.buf(N, h, w, m.unet.mid.conv2.out_channels)
        block(m.unet.mid, x, h, w, y)
        x = y
        for d, blk in enumerate(m.unet.up):
            cat, cw, ch, cwid = cats[depth - 1 - d]
            assert x.shape[-1] == 2 * cw and (ch, cwid) == (2 * h, 2 * w)
            flush()
            self.steps.append(("up", x, cat, N, h, w, 2 * cw, 3 * cw, 0))
            h, w = ch, cwid
            y = s.buf(N, h, w, blk.conv2.out_channels)
            block(blk, cat, h, w, y)
            x = y
        u = s.buf(N, h, w, nf)
        s.add((m.unet.final, None, None), x, u, N, h, w, residual=first)
        x = u
        for j in range(2):
            co = m.up[j].out_channels
            y = s.buf(N, 2 * h, 2 * w, co // 4)
            # the second up conv and the tail conv may run as ONE launch (srmi_conv2d_fwd_pair, SRMI_FUSE_UPTAIL)
            s.add((m.up[j], None, m.up_act[j]), x, y, N, h, w, shuffle=2, shuffle_order=L.SHUFFLE_DCR, fuse_next=(j == 1))
            x, h, w = y, 2 * h, 2 * w
        self.out_shape = (N, m.tail.out_channels, h, w)
        y0 = torch.empty(self.out_shape, dtype=torch.float32, device=dev)
        s.add((m.tail, None, nn.Tanh()), x, y0, N, h, w, out_layout=L.NCHW, out_dtype=L.F32)
        flush()
        self.dev, self.adt = dev, s.adt
        self.args, self.n = s.finish()
        self.kernels, self._seq = s.kernels, s


class ResUNetSRGAN(nn.Module, _ComputeMixin):
    def __init__(self, n_feats=64, depth=4, img_ch=3, up_feats=256):
        super().__init__()
        self.n_feats = n_feats
        self.head = _c(img_ch, n_feats, 9)
        self.head_act = nn.PReLU(n_feats, init=0.0)
        self.unet = _ResUNet(n_feats, depth)
        self.up = nn.ModuleList([_c(n_feats, 4 * up_feats, 3), _c(up_feats, 4 * up_feats, 3)])
        self.up_act = nn.ModuleList([nn.PReLU(4 * up_feats, init=0.0) for _ in range(2)])
        self.tail = _c(up_feats, img_ch, 9)
        self._plans = PlanCache(ResUNetSRGANPlan)

    def forward(self, x):
        _guard(x, self)
        N, _, H, W = x.shape
        return self._plans.get(self, N, H, W, _COMPUTE[self.compute]).run(x)
