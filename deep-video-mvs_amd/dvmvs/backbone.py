"""torchvision-free MnasNet-1.0 trunk and feature pyramid, state-dict compatible with torchvision 0.6.1.

The reference builds its FeatureExtractor from ``torchvision.models.mnasnet1_0(pretrained=True).layers`` and its
FeatureShrinker from ``torchvision.ops.FeaturePyramidNetwork`` (/root/reference/dvmvs/fusionnet/model.py:122-164).
torchvision is not part of this stack, so the two architectures are restated here with the module nesting that
yields identical parameter names (``layers.{0,1,3,4,6,7}``, ``inner_blocks.N``, ``layer_blocks.N``), which lets
the published checkpoints load unchanged.  All dense convolutions are plain ``torch.nn.Conv2d`` and run on MIOpen; the depthwise layers are
``DepthwiseConv2d`` (an ``nn.Conv2d`` whose GPU forward / backward are HIP kernels).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn

# torchvision's MnasNet uses a TF-style BN momentum of 0.9997 (i.e. 1 - 0.9997 in torch convention)
MNASNET_BN_MOMENTUM = 1.0 - 0.9997

# (in, out, kernel, stride, expansion, repeats) of the six inverted-residual stacks of MnasNet-1.0
MNASNET_STACKS = (
    (16, 24, 3, 2, 3, 3),
    (24, 40, 5, 2, 3, 3),
    (40, 80, 5, 2, 6, 3),
    (80, 96, 3, 1, 6, 2),
    (96, 192, 5, 2, 6, 4),
    (192, 320, 3, 1, 6, 1),
)


class DepthwiseConv2d(nn.Conv2d):
    """nn.Conv2d for the depthwise layers of MnasNet (groups == channels, k in {3, 5}, padding k // 2, no bias): same parameters and
    state-dict keys; on the GPU in float32 the forward and both gradients are HIP kernels (dvmvs::depthwise_conv_train) -- MIOpen runs
    these layers through its naive reference kernels, ~10 ms of a training step.  (At inference the frame engine replaces the layer
    together with its BatchNorm + ReLU by dvmvs::depthwise_conv.)"""

    def forward(self, x):
        k = self.kernel_size
        if (x.is_cuda and x.dtype == torch.float32 and self.bias is None and self.groups == self.in_channels == self.out_channels and
                k[0] == k[1] and k[0] in (3, 5) and tuple(self.padding) == (k[0] // 2, k[0] // 2) and tuple(self.dilation) == (1, 1) and
                self.stride[0] == self.stride[1] and self.stride[0] in (1, 2) and self.padding_mode == "zeros"):
            from dvmvs.hip import ops as _ops
            return _ops.depthwise_conv_train(x, self.weight, self.stride[0])
        return super().forward(x)


def _bn(channels):
    return nn.BatchNorm2d(channels, momentum=MNASNET_BN_MOMENTUM)


class InvertedResidual(nn.Module):
    """1x1 expand -> kxk depthwise (stride) -> 1x1 project; identity shortcut iff shape-preserving."""

    def __init__(self, cin, cout, kernel, stride, expansion):
        super().__init__()
        mid = cin * expansion
        self.apply_residual = cin == cout and stride == 1
        self.layers = nn.Sequential(
            nn.Conv2d(cin, mid, 1, bias=False), _bn(mid), nn.ReLU(inplace=True),
            DepthwiseConv2d(mid, mid, kernel, padding=kernel // 2, stride=stride, groups=mid, bias=False), _bn(mid),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid, cout, 1, bias=False), _bn(cout))

    def forward(self, x):
        # dvmvs.engine.fuse_epilogues registers ``fused_head`` (layers[:6]) and ``fused_tail`` (the project conv whose
        # epilogue adds the shortcut) as sub-modules of shape-preserving blocks
        tail = self._modules.get("fused_tail")
        if tail is not None:
            return tail(self._modules["fused_head"](x), residual=x, residual_mode=1)
        y = self.layers(x)
        return y + x if self.apply_residual else y


def _stack(cin, cout, kernel, stride, expansion, repeats):
    blocks = [InvertedResidual(cin, cout, kernel, stride, expansion)]
    blocks += [InvertedResidual(cout, cout, kernel, 1, expansion) for _ in range(repeats - 1)]
    return nn.Sequential(*blocks)


def mnasnet1_0_trunk_layers():
    """The first 14 children of torchvision's ``mnasnet1_0().layers`` (stem + six stacks), freshly initialised.

    The 320->1280 head and the classifier are never used by the depth network
    (/root/reference/dvmvs/fusionnet/model.py:127-131 slices [0:14]) and are not built.
    """
    layers = [
        nn.Conv2d(3, 32, 3, padding=1, stride=2, bias=False), _bn(32), nn.ReLU(inplace=True),
        DepthwiseConv2d(32, 32, 3, padding=1, stride=1, groups=32, bias=False), _bn(32), nn.ReLU(inplace=True),
        nn.Conv2d(32, 16, 1, padding=0, stride=1, bias=False), _bn(16),
    ]
    layers += [_stack(*spec) for spec in MNASNET_STACKS]
    for m in layers:
        for sub in m.modules():
            if isinstance(sub, nn.Conv2d):
                nn.init.kaiming_normal_(sub.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(sub, nn.BatchNorm2d):
                nn.init.ones_(sub.weight)
                nn.init.zeros_(sub.bias)
    return layers


class FeaturePyramidNetwork(nn.Module):
    """Top-down feature pyramid: 1x1 lateral convs, nearest upsampling, 3x3 output convs (all with bias)."""

    def __init__(self, in_channels_list, out_channels, extra_blocks=None):
        super().__init__()
        if extra_blocks is not None:
            raise NotImplementedError("extra_blocks are not used by the depth networks")
        self.inner_blocks = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in in_channels_list)
        self.layer_blocks = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels_list)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    fused_top_down = False   # set by dvmvs.engine.fuse_epilogues: lateral conv epilogue adds the nearest-up-sampled coarser level

    def forward(self, x):
        names = list(x.keys())
        feats = list(x.values())
        top = self.inner_blocks[-1](feats[-1])
        outs = [self.layer_blocks[-1](top)]
        for level in range(len(feats) - 2, -1, -1):
            h, w = feats[level].shape[-2:]
            if self.fused_top_down and (h, w) == (2 * top.shape[-2], 2 * top.shape[-1]):
                top = self.inner_blocks[level](feats[level], residual=top, residual_mode=2)
            else:
                lateral = self.inner_blocks[level](feats[level])
                top = lateral + F.interpolate(top, size=lateral.shape[-2:], mode="nearest")
            outs.insert(0, self.layer_blocks[level](top))
        return OrderedDict(zip(names, outs))
