"""`vgg19` feature extractor with the reference's class name, constructor and output tuple
(P/vgg19.py:7-78), on libsrmi's conv / max-pool kernels.

The reference wraps ``torchvision.models.vgg19(pretrained=True).features``; neither torchvision nor the
ImageNet weights exist offline, so this module builds the same 37-layer ``features`` stack itself (the
public VGG19 "E" configuration: 3x3 pad-1 convs + ReLU(inplace), 2x2 max-pools) under the same attribute
name, ``vgg_feature``, with torchvision's indices - a torchvision ``features`` state_dict loads as is:

    net = vgg19(pre_trained=False); net.vgg_feature.load_state_dict(torch.load("vgg19_features.pt"))

``pre_trained=True`` looks for that file at ``$SRMI_VGG19_WEIGHTS`` and fails loudly when it is absent.

Every conv+ReLU pair is ONE fused op.  Because torchvision's ReLUs are in-place, the reference's
``convX_Y`` entries alias the ``reluX_Y`` tensors (the conv output is overwritten before it is returned);
this module returns the same tensor for both names, which is that behaviour exactly.
"""
from __future__ import annotations

import os
from collections import namedtuple

import torch
import torch.nn as nn

from . import autograd as AG
from .functional import require_cuda

CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def make_features():
    layers, cin = [], 3
    for v in CFG_E:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


class vgg19(nn.Module):
    vgg_layer = ['conv1_1', 'relu1_1', 'conv1_2', 'relu1_2', 'pool1',
                 'conv2_1', 'relu2_1', 'conv2_2', 'relu2_2', 'pool2',
                 'conv3_1', 'relu3_1', 'conv3_2', 'relu3_2', 'conv3_3', 'relu3_3', 'conv3_4', 'relu3_4', 'pool3',
                 'conv4_1', 'relu4_1', 'conv4_2', 'relu4_2', 'conv4_3', 'relu4_3', 'conv4_4', 'relu4_4', 'pool4',
                 'conv5_1', 'relu5_1', 'conv5_2', 'relu5_2', 'conv5_3', 'relu5_3', 'conv5_4', 'relu5_4', 'pool5']
    _output = namedtuple("vgg_output", vgg_layer)

    def __init__(self, pre_trained=True, require_grad=False):
        super().__init__()
        self.vgg_feature = make_features()
        if pre_trained:
            path = os.environ.get("SRMI_VGG19_WEIGHTS")
            if not path or not os.path.exists(path):
                raise RuntimeError(
                    "srmi.vgg19(pre_trained=True): the ImageNet weights cannot be downloaded here; point "
                    "SRMI_VGG19_WEIGHTS at a torchvision vgg19 `features` state_dict, or pass pre_trained=False")
            sd = torch.load(path, map_location="cpu", weights_only=True)
            sd = {k[len("features."):] if k.startswith("features.") else k: v for k, v in sd.items()
                  if not k.startswith("classifier.")}
            self.vgg_feature.load_state_dict(sd)
        if not require_grad:
            for parameter in self.parameters():
                parameter.requires_grad = False

    def forward(self, x):
        require_cuda(x)
        outs = []
        mods = list(self.vgg_feature)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                x = AG.fused_conv(x, m, None, mods[i + 1])      # conv + in-place ReLU
                outs += [x, x]
                i += 2
            else:
                N, Cc, H, W = x.shape
                if H < 2 or W < 2:
                    raise RuntimeError(f"vgg19: max-pool over a {H}x{W} map (input too small)")
                x = AG.maxpool2x2(x)
                outs.append(x)
                i += 1
        return self._output(*outs)
