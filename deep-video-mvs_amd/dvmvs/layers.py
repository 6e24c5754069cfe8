"""Convolution building blocks shared by the pairnet / fusionnet encoder-decoder.

Same helper names and the same ``nn.Sequential`` child positions as /root/reference/dvmvs/layers.py:4-65, so
parameter names (``...0.weight`` for the conv, ``...1.*`` for the BatchNorm) match the reference checkpoints.
The convolutions themselves are ``torch.nn.Conv2d`` -> MIOpen; nothing here is on the hand-written HIP path.
"""
from torch import nn


def _same_pad(kernel_size):
    return (kernel_size - 1) // 2


def _conv(cin, cout, kernel_size, stride=1, bias=False):
    return nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=_same_pad(kernel_size), bias=bias)


def conv_layer(input_channels, output_channels, kernel_size, stride, apply_bn_relu):
    """conv (no bias) [+ BatchNorm + ReLU]."""
    mods = [_conv(input_channels, output_channels, kernel_size, stride)]
    if apply_bn_relu:
        mods += [nn.BatchNorm2d(output_channels), nn.ReLU(inplace=True)]
    return nn.Sequential(*mods)


def depth_layer_3x3(input_channels):
    """3x3 conv (with bias) to one channel, squashed to (0,1); scaled to inverse depth by the decoder."""
    return nn.Sequential(nn.Conv2d(input_channels, 1, 3, padding=1), nn.Sigmoid())


def down_conv_layer(input_channels, output_channels, kernel_size):
    """conv-BN-ReLU at stride 1 followed by conv-BN-ReLU at stride 2 (kept for surface compatibility)."""
    return nn.Sequential(
        _conv(input_channels, output_channels, kernel_size), nn.BatchNorm2d(output_channels), nn.ReLU(),
        _conv(output_channels, output_channels, kernel_size, stride=2), nn.BatchNorm2d(output_channels), nn.ReLU())


def up_conv_layer(input_channels, output_channels, kernel_size):
    """x2 bilinear (align_corners) upsample then conv-BN-ReLU (kept for surface compatibility)."""
    return nn.Sequential(
        nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
        _conv(input_channels, output_channels, kernel_size), nn.BatchNorm2d(output_channels), nn.ReLU())
