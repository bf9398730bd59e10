"""CPU restatement of the reference's two Keras generators (TEST INFRASTRUCTURE - see oracle/__init__.py).

PARITY UNPINNED: TensorFlow/Keras, h5py and cv2 are absent from the container, the scripts train at import
time and read Windows paths, and no outputs of these models are stored in the reference.
The functions below restate the published Keras semantics at the reference's call sites; they are checked
only structurally (parameter counts, shapes) and serve as the oracle of record for the HIP path.

Keras semantics used: Conv2D(padding='same', stride 1) = symmetric k//2 zero padding for odd k, bias on;
PReLU(shared_axes=[1,2]) = one alpha per channel; LeakyReLU(alpha=0.2); Concatenate() appends on the channel
axis; tf.nn.depth_to_space(x, 2) is DCR: out[b, 2h+i, 2w+j, c] = in[b, h, w, (2i+j)*C + c].
Tensors here are NCHW (torch); weights are OIHW (Keras stores HWIO: see srmi.keras_models.from_keras_hwio).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _conv(sd, name, x):
    w = sd[name + ".weight"]
    return F.conv2d(x, w, sd[name + ".bias"], padding=w.shape[-1] // 2)


def srcnn_forward(sd, x):
    """`model()`: conv9x9(1->128)+ReLU, conv3x3(128->64)+ReLU, conv5x5(64->1) linear  (SRCNN/srcnn.py:202-210)."""
    x = F.relu(_conv(sd, "conv1", x))
    x = F.relu(_conv(sd, "conv2", x))
    return _conv(sd, "conv3", x)


def depth_to_space_dcr(x, r):
    n, c, h, w = x.shape
    cf = c // (r * r)
    return x.view(n, r, r, cf, h, w).permute(0, 3, 4, 1, 5, 2).reshape(n, cf, h * r, w * r)


def _prelu_c(x, alpha):
    return torch.where(x >= 0, x, x * alpha.view(1, -1, 1, 1))


def rrdb_block(sd, prefix, x, num_dense_layers=3):
    """`rrdb_block`: dense block (conv3x3 -> growth, LeakyReLU 0.2, concat) x L, conv3x3 -> filters, x0.2, + x
    (ESRGAN/untitled0.py:18-35)."""
    cat = x
    for i in range(num_dense_layers):
        d = F.leaky_relu(_conv(sd, f"{prefix}.dense.{i}", cat), 0.2)
        cat = torch.cat([cat, d], dim=1)
    return x + 0.2 * _conv(sd, f"{prefix}.compress", cat)


def rrdbnet_forward(sd, x, num_blocks=16, num_dense_layers=3):
    """`build_generator_esrgan` (ESRGAN/untitled0.py:37-67)."""
    x = _prelu_c(_conv(sd, "head", x), sd["head_act.weight"])
    first = x
    for b in range(num_blocks):
        x = rrdb_block(sd, f"blocks.{b}", x, num_dense_layers)
    x = _conv(sd, "trunk", x) + first
    for j in range(2):
        x = depth_to_space_dcr(_conv(sd, f"up.{j}", x), 2)
        x = _prelu_c(x, sd[f"up_act.{j}.weight"])
    return torch.tanh(_conv(sd, "tail", x))


def _bn_eval(sd, name, x, eps=1e-3):
    """Keras BatchNormalization() at inference: (x - moving_mean) / sqrt(moving_var + 1e-3) * gamma + beta."""
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        training=False, eps=eps)


def keras_srgan_forward(sd, x, num_res_blocks=16):
    """`build_generator` of "Typical SRGAN with tensorflow/typical srgan .py":21-71: conv9x9 + PReLU(per channel) ·
    N x [conv3x3, BN, PReLU, conv3x3, BN, + skip] · conv3x3, BN, + first · 2 x [conv3x3(64 -> 256), PReLU(256 slopes,
    BEFORE the shuffle), depth_to_space(2)] · conv9x9(64 -> 3) tanh."""
    x = _prelu_c(_conv(sd, "head", x), sd["head_act.weight"])
    first = x
    for b in range(num_res_blocks):
        p = f"blocks.{b}"
        t = _prelu_c(_bn_eval(sd, p + ".bn1", _conv(sd, p + ".conv1", x)), sd[p + ".act.weight"])
        x = x + _bn_eval(sd, p + ".bn2", _conv(sd, p + ".conv2", t))
    x = _bn_eval(sd, "post_bn", _conv(sd, "post", x)) + first
    for j in range(2):
        x = depth_to_space_dcr(_prelu_c(_conv(sd, f"up.{j}", x), sd[f"up_act.{j}.weight"]), 2)
    return torch.tanh(_conv(sd, "tail", x))


def _conv_block(sd, p, x):
    """`conv_block` (U-SRGAN/srganunet.py:14-30): (conv3x3, BN, ReLU) x 2 (dropout 0)."""
    x = F.relu(_bn_eval(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    return F.relu(_bn_eval(sd, p + ".bn2", _conv(sd, p + ".conv2", x)))


def unet_forward(sd, x, depth=4):
    """`UNet` (U-SRGAN/srganunet.py:32-86): `depth` x (conv_block, MaxPooling2D(2)), bottleneck conv_block,
    `depth` x (UpSampling2D(2) nearest, concatenate([up, skip]), conv_block), conv1x1 -> 64."""
    skips = []
    for d in range(depth):
        x = _conv_block(sd, f"unet.down.{d}", x)
        skips.append(x)
        x = F.max_pool2d(x, 2)
    x = _conv_block(sd, "unet.mid", x)
    for d in range(depth):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = torch.cat([x, skips[depth - 1 - d]], dim=1)
        x = _conv_block(sd, f"unet.up.{d}", x)
    return _conv(sd, "unet.final", x)


def unet_srgan_forward(sd, x, depth=4):
    """`build_unet_srgan_generator` (U-SRGAN/srganunet.py:90-126): conv9x9 + PReLU · UNet · conv3x3, BN,
    Concatenate([., skip]) · 2 x [conv3x3(-> 1024), PReLU, depth_to_space(2)] · conv9x9(256 -> 3) tanh."""
    x = _prelu_c(_conv(sd, "head", x), sd["head_act.weight"])
    skip = x
    x = unet_forward(sd, x, depth)
    x = torch.cat([_bn_eval(sd, "post_bn", _conv(sd, "post", x)), skip], dim=1)
    for j in range(2):
        x = depth_to_space_dcr(_prelu_c(_conv(sd, f"up.{j}", x), sd[f"up_act.{j}.weight"]), 2)
    return torch.tanh(_conv(sd, "tail", x))


def autoencoder_block(sd, p, x):
    """`autoencoder_block` (A-SRGAN/srganautoencoders.py:55-104): two conv+PReLU, pool, two conv+PReLU, pool, bottleneck
    conv+PReLU, upsample, two conv+PReLU (+ e5), upsample, two conv+PReLU (+ e2)."""
    c = lambda name, t: _prelu_c(_conv(sd, f"{p}.{name}", t), sd[f"{p}.{name}_act.weight"])
    e2 = c("e2", c("e1", x))
    e5 = c("e5", c("e4", F.max_pool2d(e2, 2)))
    b = c("b", F.max_pool2d(e5, 2))
    d3 = c("d3", c("d2", F.interpolate(b, scale_factor=2, mode="nearest"))) + e5
    return c("d6", c("d5", F.interpolate(d3, scale_factor=2, mode="nearest"))) + e2


def autoencoder_srgan_forward(sd, x, num_autoencoders=1):
    """`build_autoencoder_srgan_generator` (A-SRGAN/srganautoencoders.py:106-144)."""
    x = _prelu_c(_conv(sd, "head", x), sd["head_act.weight"])
    skip = x
    for i in range(num_autoencoders):
        x = autoencoder_block(sd, f"blocks.{i}", x)
    x = _bn_eval(sd, "post_bn", _conv(sd, "post", x)) + skip
    for j in range(2):
        x = depth_to_space_dcr(_prelu_c(_conv(sd, f"up.{j}", x), sd[f"up_act.{j}.weight"]), 2)
    return torch.tanh(_conv(sd, "tail", x))


def _res_conv_block(sd, p, x):
    """`res_conv_block` (Res-U-SRGAN/untitled0.py:39-73): conv3x3, BN, ReLU, conv3x3, BN; shortcut conv1x1, BN;
    ReLU(shortcut + conv)."""
    c = F.relu(_bn_eval(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    c = _bn_eval(sd, p + ".bn2", _conv(sd, p + ".conv2", c))
    return F.relu(_bn_eval(sd, p + ".bn_s", _conv(sd, p + ".shortcut", x)) + c)


def res_unet_srgan_forward(sd, x, depth=4):
    """`build_unet_srgan_generator` of Res-U-SRGAN/untitled0.py:190-226 around `Res_UNet` (:76-125).  The generator's
    post conv + BN is dead in the reference graph (its output is overwritten by Add([skip, u_net_output]), :207-209), so
    the model is head -> Res_UNet -> + head output -> two sub-pixel stages -> conv9x9 tanh."""
    x = _prelu_c(_conv(sd, "head", x), sd["head_act.weight"])
    skip, skips = x, []
    for d in range(depth):
        x = _res_conv_block(sd, f"unet.down.{d}", x)
        skips.append(x)
        x = F.max_pool2d(x, 2)
    x = _res_conv_block(sd, "unet.mid", x)
    for d in range(depth):
        x = torch.cat([F.interpolate(x, scale_factor=2, mode="nearest"), skips[depth - 1 - d]], dim=1)
        x = _res_conv_block(sd, f"unet.up.{d}", x)
    x = _conv(sd, "unet.final", x) + skip
    for j in range(2):
        x = depth_to_space_dcr(_prelu_c(_conv(sd, f"up.{j}", x), sd[f"up_act.{j}.weight"]), 2)
    return torch.tanh(_conv(sd, "tail", x))
