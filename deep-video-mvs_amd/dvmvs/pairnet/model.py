"""``dvmvs.pairnet.model`` -- per-frame network: features -> plane-sweep cost volume -> encoder -> decoder.

Same class names as /root/reference/dvmvs/pairnet/model.py; the definitions are shared with fusionnet
(``dvmvs.networks``).  Checkpoint order: 0_feature_extractor, 1_feature_pyramid, 2_encoder, 3_decoder.
"""
from dvmvs.networks import (CostVolumeDecoder, CostVolumeEncoder, DecoderBlock, DownconvolutionLayer, EncoderBlock,  # noqa: F401
                            FeatureExtractor, FeatureShrinker, StandardLayer, UpconvolutionLayer, fpn_output_channels,
                            hyper_channels)

MODULE_ORDER = ("feature_extractor", "feature_pyramid", "encoder", "decoder")
