"""Module surface shared by ``dvmvs.pairnet.model`` and ``dvmvs.fusionnet.model``.

The reference keeps two near-identical copies of these classes (/root/reference/dvmvs/pairnet/model.py and
/root/reference/dvmvs/fusionnet/model.py:15-305 differ only in variable names); here they are defined once and
re-exported by both.  Constructors take no arguments, ``forward`` signatures and the attribute names that
determine the state-dict keys are the reference's, so its checkpoints (``0_feature_extractor`` ...
``4_decoder``) load positionally as before.  Every layer in this file is a dense convolution / BN / upsample and
is executed by MIOpen / rocBLAS through PyTorch-ROCm; the hand-written HIP kernels sit between these modules
(cost volume) and inside ``LSTMFusion`` (hidden-state warp, gate fusion).
"""
from collections import OrderedDict

import torch
from torch import nn
import torch.nn.functional as F

from dvmvs.backbone import FeaturePyramidNetwork, mnasnet1_0_trunk_layers
from dvmvs.config import Config
from dvmvs.layers import conv_layer, depth_layer_3x3

fpn_output_channels = 32
hyper_channels = 32


def _upsample2(x):
    """x2 bilinear up-sampling, align_corners=True.  On the GPU this is the HIP kernel (dvmvs_upsample2x_fwd, same interpolation
    formula; with autograd its adjoint dvmvs_upsample2x_bwd, a gather without atomics); on the CPU it is ATen's."""
    if x.is_cuda and x.dtype == torch.float32:
        from dvmvs.hip import ops as _ops
        return _ops.upsample2x(x)
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


class StandardLayer(nn.Module):
    def __init__(self, channels, kernel_size, apply_bn_relu):
        super().__init__()
        self.conv1 = conv_layer(channels, channels, kernel_size, 1, True)
        self.conv2 = conv_layer(channels, channels, kernel_size, 1, apply_bn_relu)

    def forward(self, x):
        return self.conv2(self.conv1(x))


class DownconvolutionLayer(nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size):
        super().__init__()
        self.down_conv = conv_layer(input_channels, output_channels, kernel_size, 2, True)

    def forward(self, x):
        return self.down_conv(x)


class UpconvolutionLayer(nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size):
        super().__init__()
        self.conv = conv_layer(input_channels, output_channels, kernel_size, 1, True)

    def forward(self, x):
        return self.conv(_upsample2(x))


class EncoderBlock(nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size):
        super().__init__()
        self.down_convolution = DownconvolutionLayer(input_channels, output_channels, kernel_size)
        self.standard_convolution = StandardLayer(output_channels, kernel_size, True)

    def forward(self, x):
        return self.standard_convolution(self.down_convolution(x))


class DecoderBlock(nn.Module):
    """Upsample the coarser level, concatenate the skip (and the upsampled coarser depth), two convs."""

    def __init__(self, input_channels, output_channels, kernel_size, apply_bn_relu, plus_one):
        super().__init__()
        self.up_convolution = UpconvolutionLayer(input_channels, output_channels, kernel_size)
        self.convolution1 = conv_layer(input_channels + (1 if plus_one else 0), output_channels, kernel_size, 1, True)
        self.convolution2 = conv_layer(output_channels, output_channels, kernel_size, 1, apply_bn_relu)

    def forward(self, x, skip, depth):
        parts = [self.up_convolution(x), skip]
        if depth is not None:
            parts.append(_upsample2(depth))
        return self.convolution2(self.convolution1(torch.cat(parts, dim=1)))


class FeatureExtractor(nn.Module):
    """MnasNet-1.0 trunk split at the five resolutions 1/2 ... 1/32 (16, 24, 40, 96, 320 channels).

    ``pretrained`` weights cannot be downloaded in this stack; the trunk is randomly initialised and the
    checkpoint ``0_feature_extractor`` is expected to be loaded on top (as the reference scripts do).
    """

    def __init__(self):
        super().__init__()
        trunk = mnasnet1_0_trunk_layers()
        for name, (lo, hi) in zip(("layer1", "layer2", "layer3", "layer4", "layer5"),
                                  ((0, 8), (8, 9), (9, 10), (10, 12), (12, 14))):
            setattr(self, name, nn.Sequential(*trunk[lo:hi]))

    def forward(self, image):
        outs = []
        x = image
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5):
            x = stage(x)
            outs.append(x)
        return tuple(outs)


class FeatureShrinker(nn.Module):
    """FPN to 32 channels; returns the 1/2, 1/4, 1/8, 1/16 maps (the 1/32 output is dropped)."""

    def __init__(self):
        super().__init__()
        self.fpn = FeaturePyramidNetwork([16, 24, 40, 96, 320], fpn_output_channels, extra_blocks=None)

    def forward(self, layer1, layer2, layer3, layer4, layer5):
        pyramid = self.fpn(OrderedDict(layer1=layer1, layer2=layer2, layer3=layer3, layer4=layer4, layer5=layer5))
        return pyramid["layer1"], pyramid["layer2"], pyramid["layer3"], pyramid["layer4"]


class CostVolumeEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        hc = hyper_channels
        # level: (aggregator in, aggregator/encoder in, encoder out, kernel)
        plan = ((Config.train_n_depth_levels + fpn_output_channels, hc, hc * 2, 5),
                (hc * 2 + fpn_output_channels, hc * 2, hc * 4, 3),
                (hc * 4 + fpn_output_channels, hc * 4, hc * 8, 3),
                (hc * 8 + fpn_output_channels, hc * 8, hc * 16, 3))
        for level, (agg_in, width, enc_out, k) in enumerate(plan):
            setattr(self, f"aggregator{level}", conv_layer(agg_in, width, k, 1, True))
            setattr(self, f"encoder_block{level}", EncoderBlock(width, enc_out, k))

    def forward(self, features_half, features_quarter, features_one_eight, features_one_sixteen, cost_volume):
        skips = []
        x = cost_volume
        for level, feat in enumerate((features_half, features_quarter, features_one_eight, features_one_sixteen)):
            skip = getattr(self, f"aggregator{level}")(torch.cat([feat, x], dim=1))
            x = getattr(self, f"encoder_block{level}")(skip)
            skips.append(skip)
        return skips[0], skips[1], skips[2], skips[3], x


class CostVolumeDecoder(nn.Module):
    def __init__(self):
        super().__init__()
        hc = hyper_channels
        self.inverse_depth_base = 1 / Config.train_max_depth
        self.inverse_depth_multiplier = 1 / Config.train_min_depth - 1 / Config.train_max_depth

        self.decoder_block1 = DecoderBlock(hc * 16, hc * 8, 3, True, plus_one=False)
        self.decoder_block2 = DecoderBlock(hc * 8, hc * 4, 3, True, plus_one=True)
        self.decoder_block3 = DecoderBlock(hc * 4, hc * 2, 3, True, plus_one=True)
        self.decoder_block4 = DecoderBlock(hc * 2, hc, 5, True, plus_one=True)
        self.refine = nn.Sequential(conv_layer(hc + 4, hc, 5, 1, True), conv_layer(hc, hc, 5, 1, True))

        self.depth_layer_one_sixteen = depth_layer_3x3(hc * 8)
        self.depth_layer_one_eight = depth_layer_3x3(hc * 4)
        self.depth_layer_quarter = depth_layer_3x3(hc * 2)
        self.depth_layer_half = depth_layer_3x3(hc)
        self.depth_layer_full = depth_layer_3x3(hc)

    def _to_depth(self, sigmoid_depth):
        return 1.0 / (self.inverse_depth_multiplier * sigmoid_depth + self.inverse_depth_base).squeeze(1)

    def forward(self, image, skip0, skip1, skip2, skip3, bottom, full_resolution_only=False):
        """Returns (full, half, quarter, 1/8, 1/16) depth maps like the reference; with ``full_resolution_only`` the four
        coarser maps (training-loss outputs that inference discards) are returned as None and never computed."""
        d1 = self.decoder_block1(bottom, skip3, None)
        s16 = self.depth_layer_one_sixteen(d1)
        d2 = self.decoder_block2(d1, skip2, s16)
        s8 = self.depth_layer_one_eight(d2)
        d3 = self.decoder_block3(d2, skip1, s8)
        s4 = self.depth_layer_quarter(d3)
        d4 = self.decoder_block4(d3, skip0, s4)
        s2 = self.depth_layer_half(d4)

        full_in = torch.cat([_upsample2(d4), _upsample2(s2), image], dim=1)
        s1 = self.depth_layer_full(self.refine(full_in))
        if full_resolution_only:
            return self._to_depth(s1), None, None, None, None
        return self._to_depth(s1), self._to_depth(s2), self._to_depth(s4), self._to_depth(s8), self._to_depth(s16)
