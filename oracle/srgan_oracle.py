"""Functional CPU restatement of the SRGAN generator / discriminator forward
(TEST INFRASTRUCTURE - see oracle/__init__.py; never imported by the product).

Works on a plain ``dict[str, torch.Tensor]`` with the reference's state_dict keys and
uses only ``torch.nn.functional`` on CPU tensors (fp32 by default, fp64 if the dict and
input are fp64).  Each function cites the reference lines it follows
(P/ = "pytorch srgan with lpips,dists/SRGAN_CustomDataset-main/").

Pinned: ``oracle/make_golden.py`` runs the imported reference and this file on the same
recipe weights and stores the reference's outputs in ``tests/golden/``;
``tests/test_oracle.py`` re-checks this file against those vectors on every run.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default, P/ops.py:26
BN_MOMENTUM = 0.1    # nn.BatchNorm2d default
LRELU_SLOPE = 0.01   # nn.LeakyReLU() default slope, P/srgan_model.py:43


def _conv2d(sd, prefix, x, stride=1):
    """`_conv`: padding = k//2, bias always on (P/ops.py:6-9)."""
    w = sd[prefix + ".weight"]
    return F.conv2d(x, w, sd[prefix + ".bias"], stride=stride, padding=w.shape[-1] // 2)


def _bn(sd, prefix, x, training=False, stats_out=None):
    """nn.BatchNorm2d (P/ops.py:25-26).  eval: running stats.  train: biased batch var for
    normalisation, unbiased var into running_var, momentum 0.1; updates are written to
    ``stats_out`` (a dict) so that the input dict stays immutable."""
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, g, b, False, 0.0, BN_EPS)
    src = stats_out if (stats_out is not None and prefix + ".running_mean" in stats_out) else sd
    rm2 = src[prefix + ".running_mean"].clone()
    rv2 = src[prefix + ".running_var"].clone()
    y = F.batch_norm(x, rm2, rv2, g, b, True, BN_MOMENTUM, BN_EPS)
    if stats_out is not None:
        stats_out[prefix + ".running_mean"] = rm2
        stats_out[prefix + ".running_var"] = rv2
        stats_out[prefix + ".num_batches_tracked"] = src[prefix + ".num_batches_tracked"] + 1
    return y


def conv_block(sd, prefix, x, bn=False, act=None, alpha=None, stride=1, training=False,
               stats_out=None):
    """`conv` = Sequential(_conv [, BN] [, act])  (P/ops.py:18-35)."""
    x = _conv2d(sd, prefix + ".body.0", x, stride)
    if bn:
        x = _bn(sd, prefix + ".body.1", x, training, stats_out)
    if act == "prelu":
        x = F.prelu(x, alpha)
    elif act == "lrelu":
        x = F.leaky_relu(x, LRELU_SLOPE)
    elif act == "tanh":
        x = torch.tanh(x)
    elif act is not None:  # pragma: no cover
        raise ValueError(act)
    return x


def res_block(sd, prefix, x, alpha, training=False, stats_out=None):
    """`ResBlock`: conv-BN-act, conv-BN, res += x  (P/ops.py:37-48)."""
    r = conv_block(sd, prefix + ".body.0", x, bn=True, act="prelu", alpha=alpha,
                   training=training, stats_out=stats_out)
    r = conv_block(sd, prefix + ".body.1", r, bn=True, training=training, stats_out=stats_out)
    return r + x


def upsampler(sd, prefix, x, r, alpha):
    """`Upsampler`: conv(C -> C r^2) -> PixelShuffle(r) -> act  (P/ops.py:70-84)."""
    x = conv_block(sd, prefix + ".body.0", x)
    x = F.pixel_shuffle(x, r)
    return F.prelu(x, alpha)


def generator_forward(sd, x, num_block=16, scale=4, training=False, stats_out=None, taps=None):
    """`Generator.forward` -> (sr, feat)  (P/srgan_model.py:27-39).

    ``taps`` (optional dict) receives intermediate tensors for layer-level parity tests.
    """
    alpha = sd["conv01.body.1.weight"]           # the one shared PReLU, P/srgan_model.py:8
    x = conv_block(sd, "conv01", x, act="prelu", alpha=alpha)            # :29
    skip = x                                                              # :30
    if taps is not None:
        taps["conv01"] = x
    for i in range(num_block):                                            # :32
        x = res_block(sd, f"body.{i}", x, alpha, training, stats_out)
        if taps is not None and i == 0:
            taps["block0"] = x
    x = conv_block(sd, "conv02", x, bn=True, training=training, stats_out=stats_out)  # :33
    feat = x + skip                                                       # :34
    x = feat
    ups = [2, 2] if scale == 4 else [scale]                               # :18-21
    for j, r in enumerate(ups):                                           # :36
        x = upsampler(sd, f"tail.{j}", x, r, alpha)
        if taps is not None:
            taps[f"up{j}"] = x
    x = conv_block(sd, "last_conv", x, act="tanh")                        # :37
    return x, feat


def discriminator_forward(sd, x, num_of_block=3, training=False, stats_out=None):
    """`Discriminator.forward`  (P/srgan_model.py:65-73, blocks P/ops.py:86-96)."""
    x = conv_block(sd, "conv01", x, act="lrelu")
    x = conv_block(sd, "conv02", x, act="lrelu", stride=2)
    for i in range(num_of_block):
        x = conv_block(sd, f"body.{i}.body.0", x, bn=True, act="lrelu",
                       training=training, stats_out=stats_out)
        x = conv_block(sd, f"body.{i}.body.1", x, bn=True, act="lrelu", stride=2,
                       training=training, stats_out=stats_out)
    x = x.reshape(x.shape[0], -1)                       # x.view(-1, linear_size): NCHW order (:70)
    x = F.leaky_relu(F.linear(x, sd["tail.0.weight"], sd["tail.0.bias"]), LRELU_SLOPE)
    return torch.sigmoid(F.linear(x, sd["tail.2.weight"], sd["tail.2.bias"]))


def tv_loss(x, weight=1.0):
    """`TVLoss.forward`  (P/losses.py:46-59)."""
    b, c, h, w = x.shape
    count_h = c * (h - 1) * w
    count_w = c * h * (w - 1)
    h_tv = ((x[:, :, 1:, :] - x[:, :, :h - 1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :w - 1]) ** 2).sum()
    return weight * 2 * (h_tv / count_h + w_tv / count_w) / b


def mean_shift(x, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """`MeanShift`: 1x1 conv with weight eye/std and bias -mean/std  (P/losses.py:6-14)."""
    m = torch.tensor(mean, dtype=x.dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=x.dtype).view(1, 3, 1, 1)
    return x / s - m / s


VGG_LAYERS = ['conv1_1', 'relu1_1', 'conv1_2', 'relu1_2', 'pool1',
              'conv2_1', 'relu2_1', 'conv2_2', 'relu2_2', 'pool2',
              'conv3_1', 'relu3_1', 'conv3_2', 'relu3_2', 'conv3_3', 'relu3_3', 'conv3_4', 'relu3_4', 'pool3',
              'conv4_1', 'relu4_1', 'conv4_2', 'relu4_2', 'conv4_3', 'relu4_3', 'conv4_4', 'relu4_4', 'pool4',
              'conv5_1', 'relu5_1', 'conv5_2', 'relu5_2', 'conv5_3', 'relu5_3', 'conv5_4', 'relu5_4', 'pool5']


def vgg19_features(sd, x, upto="relu5_4"):
    """`vgg19.forward` (P/vgg19.py:23-78) over torchvision's VGG19-E `features` (conv3x3 pad 1, ReLU, 2x2
    max-pool).  torchvision's ReLUs are in-place, so each `convX_Y` entry the reference returns aliases the
    post-ReLU tensor; the dict returned here maps both names to the post-ReLU value.  Stops after `upto`."""
    out = {}
    idx = 0
    names = iter(VGG_LAYERS)
    for v in [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            out[next(names)] = x
            idx += 1
        else:
            x = F.relu(F.conv2d(x, sd[f"vgg_feature.{idx}.weight"], sd[f"vgg_feature.{idx}.bias"], padding=1))
            out[next(names)] = x
            out[next(names)] = x
            idx += 2
        if upto in out:
            break
    return out


def perceptual_loss(vgg_sd, HR, SR, layer="relu5_4"):
    """`perceptual_loss.forward` (P/losses.py:29-39) -> (mse, hr_feat, sr_feat); HR, SR in [0,1]."""
    hr_feat = vgg19_features(vgg_sd, mean_shift(HR), layer)[layer]
    sr_feat = vgg19_features(vgg_sd, mean_shift(SR), layer)[layer]
    return F.mse_loss(hr_feat, sr_feat), hr_feat, sr_feat


def to_torch(np_sd, dtype=torch.float32):
    """numpy recipe dict -> torch dict (floating entries cast to ``dtype``)."""
    out = {}
    for k, v in np_sd.items():
        t = torch.from_numpy(v.copy()) if hasattr(v, "dtype") else torch.as_tensor(v)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out
