"""Training losses and running meters (surface of the reference's ``dvmvs.losses``: /root/reference/dvmvs/losses.py:7-82).

Host-side bookkeeping around the hot path: the multi-scale depth losses are a handful of masked reductions per decoder
output, evaluated by PyTorch-ROCm.  Differences in HOW (the values are pinned to the reference's by
tests/golden/losses.npz): the four sums are taken as masked reductions over the full map instead of over boolean-indexed
copies, so shapes are static and no gather kernels run; the number of valid pixels is still returned as a Python int
(one host synchronisation per call, as in the reference: the meters need host numbers anyway).
"""
import torch
from torch import nn

LOSS_TYPES = ("L1", "L1-inv", "L1-rel", "Huber")


class LossMeter(object):
    """Running sum / count with the last update's per-item mean; prints as ``last (running)`` (losses.py:7-24)."""

    def __init__(self):
        self.count = 0.0
        self.sum = 0.0
        self.avg = 0.0
        self.item_average = 0.0

    def update(self, loss, count):
        self.sum += loss
        self.count += count
        self.avg = self.sum / self.count
        self.item_average = loss / count

    def __repr__(self):
        return "{:.4f} ({:.4f})".format(self.item_average, self.avg)


def calculate_loss(groundtruth, prediction):
    """(sum |gt - p|, sum smooth_l1(p, gt), sum |1/gt - 1/p|, sum |gt - p| / gt, number of valid pixels) over the pixels
    whose ground truth, nearest-resized to the prediction's resolution, is non-zero (losses.py:56-82)."""
    batch, height, width = groundtruth.size()
    _, height_scaled, width_scaled = prediction.size()
    gt = nn.functional.interpolate(groundtruth.view(batch, 1, height, width), size=(height_scaled, width_scaled), mode="nearest")
    gt = gt.view(batch, height_scaled, width_scaled)
    valid = gt != 0
    valid_count = int(valid.sum().item())
    zero = torch.zeros((), dtype=prediction.dtype, device=prediction.device)
    safe_gt = torch.where(valid, gt, torch.ones_like(gt))             # keeps 1 / gt and the derivative finite on invalid pixels
    safe_pred = torch.where(valid, prediction, torch.ones_like(prediction))
    diff = (safe_gt - safe_pred).abs()
    l1 = torch.where(valid, diff, zero).sum()
    huber = torch.where(valid, nn.functional.smooth_l1_loss(safe_pred, safe_gt, reduction="none"), zero).sum()
    l1_inv = torch.where(valid, (1.0 / safe_gt - 1.0 / safe_pred).abs(), zero).sum()
    l1_rel = torch.where(valid, diff / safe_gt, zero).sum()
    return l1, huber, l1_inv, l1_rel, valid_count


def update_losses(predictions, weights, groundtruth, is_training, l1_meter, huber_meter, l1_inv_meter, l1_rel_meter, loss_type):
    """Training: returns sum_j weights[j] * loss_j / valid_j over the decoder outputs for ``loss_type`` and feeds the meters
    with the LAST output's sums (as the reference does: its loop variables survive the loop, losses.py:26-53).
    Evaluation: only ``predictions[-1]`` is scored and 0 is returned."""
    optimizer_loss = 0
    if is_training:
        if loss_type not in LOSS_TYPES:
            raise ValueError(f"loss_type must be one of {LOSS_TYPES}, got {loss_type!r}")
        pick = LOSS_TYPES.index(loss_type)
        for weight, prediction in zip(weights, predictions):
            l1, huber, l1_inv, l1_rel, valid_count = calculate_loss(groundtruth=groundtruth, prediction=prediction)
            optimizer_loss = optimizer_loss + weight * ((l1, l1_inv, l1_rel, huber)[pick] / valid_count)
    else:
        l1, huber, l1_inv, l1_rel, valid_count = calculate_loss(groundtruth=groundtruth, prediction=predictions[-1])
    l1_meter.update(l1.item(), valid_count)
    huber_meter.update(huber.item(), valid_count)
    l1_inv_meter.update(l1_inv.item(), valid_count)
    l1_rel_meter.update(l1_rel.item(), valid_count)
    return optimizer_loss
