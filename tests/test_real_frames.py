"""Real pixels through the whole plumbing (BASELINE.json configs[0]: the reference's own CPU-runnable case, a 2-view plane
sweep on sample-data/hololens-dataset/000 at 320x256, 64 planes).

tests/golden/sample_scene holds four frames and two depth maps of that scene (data fixtures).  The check that needs no
reference run and no trained weights is geometric: sweep the RAW colours of frame 00012 against frame 00009 with the real
poses and intrinsics (SAD mode, the reference's baseline configuration: utils.py:83-84) and compare the winning plane with
the sensor's depth map.  It can only come out right if every convention on the way is right -- file order vs pose rows,
camera-to-world direction, the intrinsics update of PreprocessImage, the /2 scaling, plane order, millimetre depth.  With
the right conventions the median error is ~2.5 planes; with the poses swapped or equal it is ~10 planes.
CPU: the oracle.  GPU: the same through dvmvs.utils (HIP kernels), and the scene runner against the CPU pipeline.
"""
import os
import shutil

import numpy as np
import pytest
import torch

import dvmvs_oracle as orc
import synthetic as syn

SCENE = os.path.join(syn.GOLDEN_DIR, "sample_scene")
POSE_ROWS = {"00003.png": 0, "00009.png": 6, "00012.png": 9, "00013.png": 10}
MEAN_RGB, STD_RGB = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def load(name):
    from dvmvs.dataset_loader import PreprocessImage, load_image
    K = np.loadtxt(os.path.join(syn.GOLDEN_DIR, "hololens_000_K.txt")).astype(np.float32)
    raw = load_image(os.path.join(SCENE, "images", name))
    assert raw.shape == (360, 540, 3)
    pre = PreprocessImage(K, raw.shape[1], raw.shape[0], 320, 256, distortion_crop=0, perform_crop=False)
    image = pre.apply_rgb(raw, 255.0, MEAN_RGB, STD_RGB)
    return pre, torch.from_numpy(np.ascontiguousarray(np.transpose(image, (2, 0, 1)))).float()[None]


def sensor_planes(pre):
    from dvmvs.dataset_loader import load_depth_png
    gt = pre.apply_depth(load_depth_png(os.path.join(SCENE, "depth", "00012.png")))[::2, ::2]
    assert gt.shape == (128, 160) and 0.5 < np.median(gt[gt > 0]) < 5.0          # metres
    step = (1 / syn.MIN_DEPTH - 1 / syn.MAX_DEPTH) / (syn.N_PLANES - 1)
    return (1 / np.maximum(gt, 1e-3) - 1 / syn.MAX_DEPTH) / step, gt > 0.3


def plane_error(volume, planes, valid):
    """median |winning plane - sensor plane| of a SAD volume aggregated over 9x9 windows."""
    agg = torch.nn.functional.avg_pool2d(volume.cpu(), 9, stride=1, padding=4)
    return float(np.median(np.abs(agg[0].argmin(0).numpy() - planes)[valid]))


def inputs():
    pre, ref = load("00012.png")
    _, meas = load("00009.png")
    assert tuple(ref.shape) == (1, 3, 256, 320) and abs(float(ref.mean())) < 2.0 and 0.3 < float(ref.std()) < 3.0
    full_K = torch.from_numpy(pre.get_updated_intrinsics()).float()[None]
    np.testing.assert_allclose(full_K.numpy(), syn.full_K().numpy(), rtol=1e-6)
    half = lambda t: torch.nn.functional.avg_pool2d(t, 2)
    return pre, half(ref), half(meas), syn.scaled_K(full_K, 2.0)


def test_two_view_sweep_of_real_frames_agrees_with_the_depth_sensor():
    pre, ref, meas, half_K = inputs()
    planes, valid = sensor_planes(pre)
    p_ref, p_meas = syn.pose(POSE_ROWS["00012.png"]), syn.pose(POSE_ROWS["00009.png"])
    right = plane_error(orc.cost_volume(ref, meas, p_ref, p_meas, half_K, syn.MIN_DEPTH, syn.MAX_DEPTH, syn.N_PLANES, False), planes, valid)
    swapped = plane_error(orc.cost_volume(ref, meas, p_meas, p_ref, half_K, syn.MIN_DEPTH, syn.MAX_DEPTH, syn.N_PLANES, False), planes, valid)
    same = plane_error(orc.cost_volume(ref, meas, p_ref, p_ref, half_K, syn.MIN_DEPTH, syn.MAX_DEPTH, syn.N_PLANES, False), planes, valid)
    print(f"median plane error vs the depth sensor: {right:.2f} (poses swapped: {swapped:.2f}, no motion: {same:.2f})")
    assert right <= 4.0 and swapped >= 2.0 * right and same >= 2.0 * right


@pytest.mark.gpu
def test_two_view_sweep_of_real_frames_on_the_gpu(hip_device):
    from dvmvs import utils
    dev = hip_device
    pre, ref, meas, half_K = inputs()
    planes, valid = sensor_planes(pre)
    p_ref, p_meas = syn.pose(POSE_ROWS["00012.png"]), syn.pose(POSE_ROWS["00009.png"])
    grid = utils.get_warp_grid_for_cost_volume_calculation(160, 128, dev)
    with orc.exact_pose_algebra():
        exp = orc.cost_volume(ref, meas, p_ref, p_meas, half_K, syn.MIN_DEPTH, syn.MAX_DEPTH, syn.N_PLANES, False)
    got = utils.calculate_cost_volume_by_warping(ref.to(dev), meas.to(dev), p_ref.to(dev), p_meas.to(dev), half_K.to(dev), grid,
                                                 syn.MIN_DEPTH, syn.MAX_DEPTH, syn.N_PLANES, dev, False)
    assert float((got.cpu() - exp).abs().max()) <= 2e-4 * float(exp.abs().max())
    assert plane_error(got, planes, valid) <= 4.0


@pytest.mark.gpu
def test_scene_runner_on_real_frames_matches_the_cpu_pipeline(hip_device, tmp_path):
    """loader -> PreprocessImage -> engine (offline runner, evaluation on) on the fixture frames vs oracle/fusionnet_cpu.py."""
    from fusionnet_cpu import CpuDepthPipeline
    from dvmvs.engine import DepthEngine
    from dvmvs.errors import compute_errors
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    from dvmvs.runner import predict_offline
    folder = os.path.join(str(tmp_path), "000")
    shutil.copytree(SCENE, folder)
    for name in ("00003.png", "00009.png"):   # the runner expects one depth file per image; these two are never evaluated
        shutil.copy(os.path.join(SCENE, "depth", "00012.png"), os.path.join(folder, "depth", name))
    names = sorted(POSE_ROWS)
    np.savetxt(os.path.join(folder, "poses.txt"), syn.sample_poses()[[POSE_ROWS[n] for n in names]].reshape(len(names), 16))
    shutil.copy(os.path.join(syn.GOLDEN_DIR, "hololens_000_K.txt"), os.path.join(folder, "K.txt"))
    index = os.path.join(str(tmp_path), "index")
    with open(index, "w") as f:
        f.write("00012.png 00009.png 00003.png\n00013.png 00012.png 00009.png\n")
    ctors = (FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder)
    engine = DepthEngine(*syn.build_e2e_modules(ctors), device=hip_device)
    preds, gts, _ = predict_offline(engine, folder, index, evaluate=True)
    assert len(preds) == 2 and gts[0].shape == (256, 320) and float(np.median(gts[0][gts[0] > 0])) > 0.5
    assert len(compute_errors(gts[0], preds[0])) == 8          # the evaluation path runs on real ground truth
    cpu = CpuDepthPipeline(*syn.build_e2e_modules(ctors))
    _, ref = load("00012.png")
    d_cpu = cpu.step(ref, syn.pose(9), [load("00009.png")[1], load("00003.png")[1]], [syn.pose(6), syn.pose(0)], syn.full_K())[0].numpy()
    err = float(np.mean(np.abs(preds[0] - d_cpu) / d_cpu))
    print(f"real frames, first keyframe: engine vs CPU pipeline depth rel-L1 {err:.3e}")
    assert err <= 2.5e-4
