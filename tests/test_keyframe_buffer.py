"""CPU: keyframe selection replayed over the sample scene's poses must reproduce the reference's shipped index files
(tests/golden/indices/*, data fixtures of the reference) line for line, and the response-code state machine."""
import os

import numpy as np

import synthetic as syn


def replay(n_measurement_frames):
    from dvmvs.keyframe_buffer import simulate_keyframe_index
    return simulate_keyframe_index(syn.sample_poses(), syn.sample_image_names(), n_measurement_frames)


def test_replay_reproduces_the_shipped_index_files(golden_dir):
    for n in (1, 2, 3):
        shipped = [l for l in open(os.path.join(golden_dir, "indices", f"keyframe+hololens-dataset+000+nmeas+{n}")).read().split("\n") if l]
        got = replay(n)
        assert len(got) == len(shipped) == 286
        assert got == shipped, next((a, b) for a, b in zip(got, shipped) if a != b)


def test_index_file_round_trip(tmp_path):
    from dvmvs.keyframe_buffer import simulate_keyframe_index, write_keyframe_index
    poses = syn.sample_poses().copy()
    poses[100:140] = np.nan                       # 40 frames without a pose: tracking is declared lost after 30
    lines = simulate_keyframe_index(poses, syn.sample_image_names(), 2)
    assert lines.count("TRACKING LOST") == 1
    lost = lines.index("TRACKING LOST")
    assert len(lines[lost + 1].split(" ")) == 2    # the buffer restarts: first keyframe after the loss has one measurement frame
    path = os.path.join(str(tmp_path), "keyframe+test+000+nmeas+2")
    write_keyframe_index(path, lines)
    assert [l for l in open(path).read().split("\n") if l] == lines


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
