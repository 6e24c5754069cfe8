"""CPU: keyframe selection replayed over the sample scene's poses must reproduce the reference's shipped index files
(tests/golden/indices/*, data fixtures of the reference) line for line, and the response-code state machine."""
import os

import numpy as np

import synthetic as syn


def replay(n_measurement_frames):
    from dvmvs.keyframe_buffer import KeyframeBuffer
    poses = syn.sample_poses()
    names = syn.sample_image_names()
    buf = KeyframeBuffer(buffer_size=30, keyframe_pose_distance=0.1, optimal_t_score=0.15, optimal_R_score=0.0, store_return_indices=True)
    lines = []
    for i in range(len(poses)):
        code = buf.try_new_keyframe(poses[i], None, index=i)
        if code == 3:
            lines.append("TRACKING LOST")
        elif code == 1:
            meas = buf.get_best_measurement_frames(n_measurement_frames)
            lines.append(" ".join([names[i]] + [names[m[2]] for m in meas]))
    return lines


def test_replay_reproduces_the_shipped_index_files(golden_dir):
    for n in (1, 2, 3):
        shipped = [l for l in open(os.path.join(golden_dir, "indices", f"keyframe+hololens-dataset+000+nmeas+{n}")).read().split("\n") if l]
        got = replay(n)
        assert len(got) == len(shipped) == 286
        assert got == shipped, next((a, b) for a, b in zip(got, shipped) if a != b)


def test_response_codes_and_tracking_loss():
    from dvmvs.keyframe_buffer import KeyframeBuffer, SimpleBuffer
    eye = np.eye(4)
    moved = np.eye(4)
    moved[0, 3] = 0.2
    bad = np.full((4, 4), np.nan)
    buf = KeyframeBuffer(30, 0.1, 0.15, 0.0, store_return_indices=False)
    assert buf.try_new_keyframe(eye, "a") == 0          # first frame
    assert buf.try_new_keyframe(eye, "b") == 2          # not enough motion
    assert buf.try_new_keyframe(moved, "c") == 1        # new keyframe
    assert [f[1] for f in buf.get_best_measurement_frames(2)] == ["a"]
    assert [buf.try_new_keyframe(bad, None) for _ in range(30)] == [5] * 30
    assert buf.try_new_keyframe(bad, None) == 3         # 31st invalid pose: lost, buffer cleared
    assert buf.try_new_keyframe(bad, None) == 4
    assert buf.try_new_keyframe(eye, "d") == 0
    import pytest
    with pytest.raises(ValueError):
        KeyframeBuffer(30, 0.1, 0.15, 0.0, store_return_indices=True).try_new_keyframe(eye, None)
    sb = SimpleBuffer(2, store_return_indices=False)
    assert [sb.try_new_keyframe(eye, k) for k in "abcd"] == [0, 1, 1, 1]
    assert [f[1] for f in sb.get_measurement_frames()] == ["b", "c"]
    assert [sb.try_new_keyframe(bad, None) for _ in range(31)] == [4] * 30 + [2]
    assert sb.try_new_keyframe(bad, None) == 3


def test_pose_distance_matches_definition():
    from dvmvs.utils import pose_distance
    a = syn.sample_poses()[9]
    b = syn.sample_poses()[6]
    combined, r, t = pose_distance(a, b)
    rel = np.linalg.inv(a) @ b
    assert abs(t - np.linalg.norm(rel[:3, 3])) < 1e-12
    assert abs(r - np.sqrt(2 * (1 - min(3.0, np.trace(rel[:3, :3])) / 3))) < 1e-12
    assert abs(combined - np.hypot(r, t)) < 1e-12
