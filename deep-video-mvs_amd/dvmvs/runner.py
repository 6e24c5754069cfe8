"""Scene runners: the loops of the reference's run-testing.py (pre-computed keyframe index) and run-testing-online.py
(KeyframeBuffer on the fly) around the MI355X frame engine.

/root/reference/dvmvs/fusionnet/run-testing.py:67-230 and run-testing-online.py:71-232, without cv2 / path / tqdm.  A scene
folder holds ``images/*.png``, ``poses.txt`` (one 4x4 camera-to-world per line), ``K.txt`` and optionally ``depth/*.png``
(uint16 millimetres).  Both runners return (predictions, reference_depths or None, InferenceTimer); ``save_results`` from
``dvmvs.utils`` writes the same ``.npz`` files as the reference.
"""
import os

import numpy as np
import torch

from dvmvs.config import Config
from dvmvs.dataset_loader import PreprocessImage, load_depth_png, load_image
from dvmvs.engine import DepthEngine
from dvmvs.keyframe_buffer import KeyframeBuffer
from dvmvs.utils import InferenceTimer

SCALE_RGB = 255.0
MEAN_RGB = [0.485, 0.456, 0.406]
STD_RGB = [0.229, 0.224, 0.225]


class Scene:
    def __init__(self, folder):
        self.folder = str(folder)
        self.K = np.loadtxt(os.path.join(self.folder, "K.txt")).astype(np.float32)
        self.poses = np.fromfile(os.path.join(self.folder, "poses.txt"), dtype=float, sep="\n ").reshape((-1, 4, 4))
        self.image_names = sorted(n for n in os.listdir(os.path.join(self.folder, "images")) if n.endswith(".png"))
        depth_dir = os.path.join(self.folder, "depth")
        self.depth_names = sorted(n for n in os.listdir(depth_dir) if n.endswith(".png")) if os.path.isdir(depth_dir) else None

    def image(self, i):
        return load_image(os.path.join(self.folder, "images", self.image_names[i]))

    def depth(self, i):
        return load_depth_png(os.path.join(self.folder, "depth", self.depth_names[i]))


def _to_device(image_hwc, device):
    return torch.from_numpy(np.ascontiguousarray(np.transpose(image_hwc, (2, 0, 1)))).float().unsqueeze(0).to(device)


def _preprocessor(scene, raw_image):
    return PreprocessImage(K=scene.K, old_width=raw_image.shape[1], old_height=raw_image.shape[0], new_width=Config.test_image_width,
                           new_height=Config.test_image_height, distortion_crop=Config.test_distortion_crop,
                           perform_crop=Config.test_perform_crop)


def _run_frame(engine, scene, timer, device, reference_index, measurement_indices, evaluate, images=None, next_reference_index=None,
               prepared=None, next_measurement_indices=None):
    """``next_reference_index``: the reference frame of the NEXT call when it is known (offline runs): its image is pre-processed now and
    handed to the engine as look-ahead (DepthEngine.step: its features are computed concurrently with this frame); ``prepared``
    (a dict) carries the pre-processed device image to that next call."""
    raw = images[reference_index] if images is not None and reference_index in images else scene.image(reference_index)
    pre = _preprocessor(scene, raw)
    ref_image = prepared.pop(reference_index, None) if prepared is not None else None
    if ref_image is None:
        ref_image = _to_device(pre.apply_rgb(raw, SCALE_RGB, MEAN_RGB, STD_RGB), device)
    next_image = None
    if next_reference_index is not None and prepared is not None:
        raw_next = images[next_reference_index] if images is not None and next_reference_index in images else scene.image(next_reference_index)
        next_image = _to_device(_preprocessor(scene, raw_next).apply_rgb(raw_next, SCALE_RGB, MEAN_RGB, STD_RGB), device)
        prepared.clear()
        prepared[next_reference_index] = next_image
    ref_pose = torch.from_numpy(scene.poses[reference_index]).float().unsqueeze(0)   # poses / K stay on the host (engine.step)
    full_K = torch.from_numpy(pre.get_updated_intrinsics()).float().unsqueeze(0)
    meas_images, meas_poses = [], []
    for m in measurement_indices:
        if engine.cache_features and m in engine._feature_cache:
            meas_images.append(None)     # features of this keyframe are cached: no need to load / pre-process the image again
        else:
            raw_m = images[m] if images is not None and m in images else scene.image(m)
            meas_images.append(_to_device(pre.apply_rgb(raw_m, SCALE_RGB, MEAN_RGB, STD_RGB), device))
        meas_poses.append(torch.from_numpy(scene.poses[m]).float().unsqueeze(0))
    timer.record_start_time()
    ahead = {}
    if next_image is not None:
        ahead = dict(next_reference_image=next_image, next_frame_id=next_reference_index)
        if next_measurement_indices is not None:      # ... and its poses: the engine then also runs its sweep + encoder a frame ahead
            ahead.update(next_reference_pose=torch.from_numpy(scene.poses[next_reference_index]).float().unsqueeze(0),
                         next_measurement_poses=[torch.from_numpy(scene.poses[m]).float().unsqueeze(0) for m in next_measurement_indices],
                         next_measurement_ids=list(next_measurement_indices))
    depth = engine.step(ref_image, ref_pose, meas_images, meas_poses, full_K, frame_id=reference_index,
                        measurement_ids=list(measurement_indices), **ahead)
    timer.record_end_time_and_elapsed_time()
    prediction = depth.cpu().numpy().squeeze()
    reference_depth = pre.apply_depth(scene.depth(reference_index)) if evaluate and scene.depth_names else None
    return prediction, reference_depth


def predict_offline(engine: DepthEngine, scene_folder, keyframe_index_file, evaluate=True, max_frames=None, frame_log=None):
    """Runs the lines of a keyframe index file ("ref meas1 meas2 ..." or "TRACKING LOST") through ``engine``.
    ``frame_log`` (a list) receives the line each prediction belongs to: "ref meas1 ..." file names, or "TRACKING LOST"."""
    scene = Scene(scene_folder)
    device = engine.device
    position = {name: i for i, name in enumerate(scene.image_names)}
    timer = InferenceTimer()
    predictions, reference_depths = [], []
    engine.new_sequence()
    lines = [l.strip() for l in open(keyframe_index_file) if l.strip()][:max_frames]
    prepared = {}      # the next keyframe's pre-processed image (the index file says which frame that is: feature look-ahead)
    for n, line in enumerate(lines):
        if frame_log is not None:
            frame_log.append(line)
        if line == "TRACKING LOST":
            engine.reset()
            continue
        indices = [position[name] for name in line.split(" ")]
        upcoming = next((l for l in lines[n + 1:] if l != "TRACKING LOST"), None)
        next_indices = [position[name] for name in upcoming.split(" ")] if upcoming is not None else None
        prediction, reference_depth = _run_frame(engine, scene, timer, device, indices[0], indices[1:], evaluate,
                                                 next_reference_index=next_indices[0] if next_indices else None, prepared=prepared,
                                                 next_measurement_indices=next_indices[1:] if next_indices else None)
        predictions.append(prediction)
        reference_depths.append(reference_depth)
    return predictions, (reference_depths if evaluate and scene.depth_names else None), timer


def predict_online(engine: DepthEngine, scene_folder, evaluate=False, max_frames=None, frame_log=None):
    """Feeds every frame of the scene to a KeyframeBuffer and predicts depth for the frames it accepts as keyframes.
    ``frame_log`` (a list) receives, in index-file syntax, what the buffer decided: one "ref meas1 ..." line per prediction and
    "TRACKING LOST" where it cleared itself -- the lines simulate_keyframe_index would write for the same poses."""
    scene = Scene(scene_folder)
    device = engine.device
    buffer = KeyframeBuffer(buffer_size=Config.test_keyframe_buffer_size, keyframe_pose_distance=Config.test_keyframe_pose_distance,
                            optimal_t_score=Config.test_optimal_t_measure, optimal_R_score=Config.test_optimal_R_measure,
                            store_return_indices=True)
    timer = InferenceTimer()
    predictions, reference_depths = [], []
    engine.new_sequence()
    n = len(scene.poses) if max_frames is None else min(max_frames, len(scene.poses))
    for i in range(n):
        response = buffer.try_new_keyframe(scene.poses[i], None, index=i)
        if response == 3:
            engine.reset()
            if frame_log is not None:
                frame_log.append("TRACKING LOST")
        if response != 1:
            continue
        measurement_indices = [frame[2] for frame in buffer.get_best_measurement_frames(Config.test_n_measurement_frames)]
        if frame_log is not None:
            frame_log.append(" ".join(scene.image_names[j] for j in [i] + measurement_indices))
        prediction, reference_depth = _run_frame(engine, scene, timer, device, i, measurement_indices, evaluate)
        predictions.append(prediction)
        reference_depths.append(reference_depth)
    return predictions, (reference_depths if evaluate and scene.depth_names else None), timer


def predict_sharded(make_engine, scene_folders, keyframe_index_files, evaluate=True, max_frames=None, rank=None, world=None):
    """BASELINE.json configs[3]: independent scenes sharded over the ranks of one node (scene ``s`` belongs to rank
    ``s % world``, dvmvs.sharding), each run through ``predict_offline`` on this rank's engine; no data-path collective.
    ``make_engine()`` builds the rank's DepthEngine lazily (a rank that owns no scene builds none).  Returns
    ({scene number: (predictions, reference depths or None, InferenceTimer)}, (frames, seconds, frames/s) of the whole job)."""
    import time

    from dvmvs.sharding import reduce_throughput, run_sharded
    if len(scene_folders) != len(keyframe_index_files):
        raise ValueError("one keyframe index file per scene folder")
    state = {"engine": None}

    def run_scene(s):
        if state["engine"] is None:
            state["engine"] = make_engine()
        return predict_offline(state["engine"], scene_folders[s], keyframe_index_files[s], evaluate=evaluate, max_frames=max_frames)

    t0 = time.perf_counter()
    results = run_sharded(len(scene_folders), run_scene, rank=rank, world=world)
    seconds = time.perf_counter() - t0
    frames = sum(len(r[0]) for r in results.values())
    # The reduction's device follows the BACKEND, not whether this rank built an engine: a rank that owns no scene (more ranks
    # than scenes) must still join an RCCL collective with a device tensor, or every other rank blocks in all_reduce.
    device = "cpu"
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    return results, reduce_throughput(frames, seconds, device=device)
