"""Keyframe selection for online inference (host side, microseconds per frame).

Surface of /root/reference/dvmvs/keyframe_buffer.py: ``KeyframeBuffer.try_new_keyframe`` (response codes 0-5) and
``get_best_measurement_frames``; ``SimpleBuffer`` (codes 0-4) and ``get_measurement_frames``.  Replaying
``KeyframeBuffer(30, 0.1, 0.15, 0.0)`` over the sample scene's poses reproduces the reference's shipped index files
line for line (tests/test_keyframe_buffer.py), including the measurement-frame order, which comes from
``numpy.argpartition``.

Response codes of ``KeyframeBuffer.try_new_keyframe``:
  0 first keyframe stored (nothing to predict yet)      3 tracking lost: buffer cleared (reset the recurrent state)
  1 new keyframe stored -> predict a depth map          4 still lost
  2 pose valid, but too close to the last keyframe      5 pose invalid, not yet considered lost
"""
from collections import deque

import numpy as np

from dvmvs.utils import is_pose_available, pose_distance

LOST_AFTER_INVALID_POSES = 30


class _PoseBuffer:
    def __init__(self, maxlen, store_return_indices):
        self.buffer = deque([], maxlen=maxlen)
        self._invalid_streak = 0
        self._with_indices = store_return_indices

    def _entry(self, pose, image, index):
        if self._with_indices and index is None:
            raise ValueError("Storing and returning the frame indices is requested in the constructor, but index=None "
                             "is passed to the function")
        return (pose, image, index) if self._with_indices else (pose, image)

    def _invalid_pose(self, lost_code, still_lost_code, waiting_code):
        self._invalid_streak += 1
        if self._invalid_streak <= LOST_AFTER_INVALID_POSES:
            return waiting_code
        if len(self.buffer) > 0:
            self.buffer.clear()
            return lost_code
        return still_lost_code


class KeyframeBuffer(_PoseBuffer):
    def __init__(self, buffer_size, keyframe_pose_distance, optimal_t_score, optimal_R_score, store_return_indices):
        super().__init__(buffer_size, store_return_indices)
        self.keyframe_pose_distance = keyframe_pose_distance
        self.optimal_t_score = optimal_t_score
        self.optimal_R_score = optimal_R_score

    def calculate_penalty(self, t_score, R_score):
        """Quadratic distance to the preferred baseline / rotation; too-short baselines are penalised five-fold."""
        t_diff = t_score - self.optimal_t_score
        t_penalty = (5.0 if t_diff < 0.0 else 1.0) * np.abs(t_diff) ** 2.0
        return np.abs(R_score - self.optimal_R_score) ** 2.0 + t_penalty

    def try_new_keyframe(self, pose, image, index=None):
        entry = self._entry(pose, image, index)
        if not is_pose_available(pose):
            return self._invalid_pose(lost_code=3, still_lost_code=4, waiting_code=5)
        self._invalid_streak = 0
        if len(self.buffer) == 0:
            self.buffer.append(entry)
            return 0
        combined_measure, _, _ = pose_distance(pose, self.buffer[-1][0])
        if combined_measure >= self.keyframe_pose_distance:
            self.buffer.append(entry)
            return 1
        return 2

    def get_best_measurement_frames(self, n_requested_measurement_frames):
        frames = list(self.buffer)
        reference_pose = frames[-1][0]
        candidates = frames[:-1]
        n = min(n_requested_measurement_frames, len(candidates))
        penalties = []
        for candidate in candidates:
            _, R_measure, t_measure = pose_distance(reference_pose, candidate[0])
            penalties.append(self.calculate_penalty(t_measure, R_measure))
        chosen = np.argpartition(penalties, n - 1)[:n]
        return [candidates[i] for i in chosen]


class SimpleBuffer(_PoseBuffer):
    """Keeps the last ``buffer_size`` frames as measurement frames, no pose-distance test."""

    def __init__(self, buffer_size, store_return_indices):
        super().__init__(buffer_size + 1, store_return_indices)

    def try_new_keyframe(self, pose, image, index=None):
        entry = self._entry(pose, image, index)
        if not is_pose_available(pose):
            return self._invalid_pose(lost_code=2, still_lost_code=3, waiting_code=4)
        self._invalid_streak = 0
        first = len(self.buffer) == 0
        self.buffer.append(entry)
        return 0 if first else 1

    def get_measurement_frames(self):
        return list(self.buffer)[:-1]


def simulate_keyframe_index(poses, image_names, n_measurement_frames, buffer_size=30, keyframe_pose_distance=0.1,
                            optimal_t_measure=0.15, optimal_R_measure=0.0):
    """Offline keyframe selection over a whole pose list: the lines of a ``keyframe+<dataset>+<scene>+nmeas+<n>`` index file
    ("ref meas1 meas2 ..." per accepted keyframe, "TRACKING LOST" where the buffer was cleared), as produced by
    /root/reference/dvmvs/simulate_keyframe_buffer.py:7-51 with the same defaults."""
    buffer = KeyframeBuffer(buffer_size, keyframe_pose_distance, optimal_t_measure, optimal_R_measure, store_return_indices=True)
    lines = []
    for i, pose in enumerate(poses):
        response = buffer.try_new_keyframe(pose, None, index=i)
        if response == 3:
            lines.append("TRACKING LOST")
        elif response == 1:
            measurement_frames = buffer.get_best_measurement_frames(n_measurement_frames)
            lines.append(" ".join([image_names[i]] + [image_names[frame[2]] for frame in measurement_frames]))
    return lines


def write_keyframe_index(path, lines):
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
