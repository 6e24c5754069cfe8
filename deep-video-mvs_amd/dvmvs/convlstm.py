"""LayerNorm ConvLSTM cell whose state is re-projected into the current view before every step.

Surface of /root/reference/dvmvs/convlstm.py:7-64 (same constructor, ``forward`` and ``init_hidden`` signatures, same
parameter name ``conv.weight``).  Per step: three HIP launches around one MIOpen convolution --

1. ``dvmvs.pose_algebra``    T = inverse(previous_pose) @ current_pose (the reference's fp32 expression by default; callers that
                             already hold it -- the frame engine -- pass it as ``transformation``)
2. ``dvmvs_hidden_warp_fwd`` depth-conditioned warp of h with the ``depth <= 0.01`` zeroing fused in
3. ``conv``                  3x3, 1024 -> 2048 channels, no bias (MIOpen / rocBLAS)
4. ``dvmvs_lstm_gates_fwd``  sigmoids, both spatial LayerNorms, CELUs and the cell update in one pass
"""
import torch
from torch import nn

from dvmvs.hip import ops as _ops
from dvmvs import pose_algebra as _pose_algebra


class MVSLayernormConvLSTMCell(nn.Module):

    def __init__(self, input_dim, hidden_dim, kernel_size, activation_function=None):
        super().__init__()
        if activation_function not in (None, torch.celu, torch.nn.functional.celu):
            raise NotImplementedError("the fused gate kernel implements the CELU cell used by the depth networks")
        self.activation_function = activation_function
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.kernel_size = kernel_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.conv = nn.Conv2d(input_dim + hidden_dim, 4 * hidden_dim, kernel_size, padding=self.padding, bias=False)

    def forward(self, input_tensor, cur_state, previous_pose, current_pose, estimated_current_depth, camera_matrix,
                transformation=None):
        """Reference signature plus the optional ``transformation`` [B,4,4] = inverse(previous_pose) @ current_pose for callers
        that have already evaluated it (``previous_pose`` must still be non-None to request the warp)."""
        h_cur, c_cur = cur_state
        if previous_pose is not None:
            if transformation is None:
                transformation = _pose_algebra.relative_pose(previous_pose, current_pose, h_cur.device)
            # The reference zeroes h where depth <= 0.01 by writing through .data, i.e. the forward value is masked
            # but the gradient is not; the op's backward reproduces exactly that.
            h_cur = _ops.hidden_warp(h_cur, estimated_current_depth, transformation, camera_matrix, True)
        combined_conv = self.conv(torch.cat([input_tensor, h_cur], dim=1))
        h_next, c_next = _ops.lstm_gates(combined_conv, c_cur)
        return h_next, c_next

    def init_hidden(self, batch_size, image_size):
        height, width = image_size
        device = self.conv.weight.device
        return (torch.zeros(batch_size, self.hidden_dim, height, width, device=device),
                torch.zeros(batch_size, self.hidden_dim, height, width, device=device))
