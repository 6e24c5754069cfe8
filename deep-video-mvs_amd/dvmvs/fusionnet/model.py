"""``dvmvs.fusionnet.model`` -- pairnet plus a ConvLSTM at the 1/32 bottleneck.

Same class names as /root/reference/dvmvs/fusionnet/model.py; everything except ``LSTMFusion`` is shared with
pairnet (``dvmvs.networks``).  Checkpoint order: 0_feature_extractor, 1_feature_pyramid, 2_encoder, 3_lstm_fusion,
4_decoder.
"""
import torch
from torch import nn

from dvmvs.convlstm import MVSLayernormConvLSTMCell
from dvmvs.networks import (CostVolumeDecoder, CostVolumeEncoder, DecoderBlock, DownconvolutionLayer, EncoderBlock,  # noqa: F401
                            FeatureExtractor, FeatureShrinker, StandardLayer, UpconvolutionLayer, fpn_output_channels,
                            hyper_channels)

MODULE_ORDER = ("feature_extractor", "feature_pyramid", "encoder", "lstm_fusion", "decoder")


class LSTMFusion(nn.Module):
    """Owns the ConvLSTM cell (512 -> 512 channels, 3x3, CELU) and its lazily created zero state."""

    def __init__(self):
        super().__init__()
        width = hyper_channels * 16
        self.lstm_cell = MVSLayernormConvLSTMCell(input_dim=width, hidden_dim=width, kernel_size=(3, 3),
                                                  activation_function=torch.celu)

    def forward(self, current_encoding, current_state, previous_pose, current_pose, estimated_current_depth, camera_matrix,
                transformation=None):
        if current_state is None:
            batch, _, height, width = current_encoding.shape
            current_state = self.lstm_cell.init_hidden(batch_size=batch, image_size=(height, width))
        hidden_state, cell_state = current_state
        return self.lstm_cell(input_tensor=current_encoding, cur_state=[hidden_state, cell_state],
                              previous_pose=previous_pose, current_pose=current_pose,
                              estimated_current_depth=estimated_current_depth, camera_matrix=camera_matrix,
                              transformation=transformation)
