"""Depth-map error metrics (numpy), as reported by the reference's evaluation (/root/reference/dvmvs/errors.py:4-28)."""
import numpy as np

METRIC_NAMES = ("abs_error", "abs_relative_error", "abs_inverse_error", "squared_relative_error", "rmse",
                "ratio_125", "ratio_125_2", "ratio_125_3")


def compute_errors(gt, pred, max_depth=np.inf):
    """Eight metrics over the pixels with 0.5 <= gt <= max_depth; all-NaN when no pixel qualifies."""
    keep = (gt >= 0.5) & (gt <= max_depth)
    gt, pred = gt[keep], pred[keep]
    if gt.size == 0:
        return (np.nan,) * len(METRIC_NAMES)
    n = np.float32(gt.size)
    diff = gt - pred
    ratio = np.maximum(gt / pred, pred / gt)
    thresholds = [np.count_nonzero(ratio < 1.25 ** k) / n for k in (1, 2, 3)]
    return (np.mean(np.abs(diff)),
            np.mean(np.abs(diff) / gt),
            np.mean(np.abs(1 / gt - 1 / pred)),
            np.mean(np.square(diff) / gt),
            np.sqrt(np.mean(np.square(diff))),
            *thresholds)
