"""Host-side pieces of the scene runner: intrinsics update (pinned to the value measured from the reference), cv2-free
resampling semantics, and -- on the GPU -- the offline and online runners over a synthetic scene folder."""
import os

import numpy as np
import pytest
import torch

import synthetic as syn


def test_intrinsics_update_matches_the_reference_measurement():
    """K.txt through PreprocessImage(540x360 -> 320x256, no crop) = fx 293.66977 fy 352.34274
    cx 159.70543 cy 120.31581 (measured from the reference)."""
    from dvmvs.dataset_loader import PreprocessImage
    K = np.loadtxt(os.path.join(syn.GOLDEN_DIR, "hololens_000_K.txt")).astype(np.float32)
    p = PreprocessImage(K, 540, 360, 320, 256, distortion_crop=0, perform_crop=False)
    got = p.get_updated_intrinsics()
    np.testing.assert_allclose([got[0, 0], got[1, 1], got[0, 2], got[1, 2]], [293.66977, 352.34274, 159.70543, 120.31581], rtol=2e-7)
    assert np.allclose(got, syn.full_K()[0].numpy(), rtol=1e-6)
    # with cropping: 540x360 is wider than 320x256 (1.5 > 1.25): columns are cropped, principal point shifts
    pc = PreprocessImage(K, 540, 360, 320, 256, distortion_crop=0, perform_crop=True)
    assert pc.crop_y == 0 and pc.crop_x == int(np.floor((540 - 360 * 1.25) / 2.0)) == 45
    assert abs(pc.cx - (K[0, 2] - 45) * 320 / 450.0) < 1e-4 and abs(pc.fy - K[1, 1] * 256 / 360.0) < 1e-4


def test_resampling_semantics():
    from dvmvs.dataset_loader import resize_bilinear, resize_nearest
    ramp = np.tile(np.arange(8, dtype=np.float32), (4, 1))
    up = resize_bilinear(ramp, 16, 8)
    # half-pixel centres: dst x maps to (x + 0.5) / 2 - 0.5, clamped at the borders
    np.testing.assert_allclose(up[0, :4], [0.0, 0.25, 0.75, 1.25], atol=1e-6)
    np.testing.assert_allclose(up[0, -1], 7.0, atol=1e-6)
    same = resize_bilinear(ramp, 8, 4)
    np.testing.assert_allclose(same, ramp, atol=1e-6)
    down = resize_nearest(np.arange(12).reshape(1, 12), 4, 1)
    assert down.tolist() == [[0, 3, 6, 9]]
    rgb = np.random.RandomState(0).rand(6, 9, 3).astype(np.float32)
    ours = resize_bilinear(rgb, 18, 12)
    ref = torch.nn.functional.interpolate(torch.from_numpy(rgb).permute(2, 0, 1)[None], size=(12, 18), mode="bilinear",
                                          align_corners=False)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(ours, ref, atol=1e-6)     # cv2.INTER_LINEAR and torch's align_corners=False agree on up-sampling


def _write_scene(folder, n_frames):
    from PIL import Image
    os.makedirs(os.path.join(folder, "images"))
    os.makedirs(os.path.join(folder, "depth"))
    poses = syn.sample_poses()[:n_frames]
    np.savetxt(os.path.join(folder, "poses.txt"), poses.reshape(n_frames, 16))
    np.savetxt(os.path.join(folder, "K.txt"), np.loadtxt(os.path.join(syn.GOLDEN_DIR, "hololens_000_K.txt")))
    rng = np.random.RandomState(1)
    for i in range(n_frames):
        img = (syn.smooth_noise((3, 360, 540), seed=600 + i).permute(1, 2, 0).numpy() * 40 + 128).clip(0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(folder, "images", f"{i:05d}.png"))
        Image.fromarray((1500 + 200 * rng.rand(360, 540)).astype(np.uint16)).save(os.path.join(folder, "depth", f"{i:05d}.png"))


@pytest.mark.gpu
def test_offline_and_online_runners(hip_device, tmp_path):
    from dvmvs.engine import DepthEngine
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    from dvmvs.runner import predict_offline, predict_online
    from dvmvs.utils import save_results
    scene = os.path.join(str(tmp_path), "scene")
    _write_scene(scene, 24)
    mods = syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder))
    engine = DepthEngine(*mods, device=hip_device)
    index = os.path.join(str(tmp_path), "index")
    with open(index, "w") as f:
        f.write("00009.png 00006.png 00003.png\n00010.png 00009.png 00006.png\nTRACKING LOST\n00013.png 00010.png 00009.png\n")
    preds, gts, timer = predict_offline(engine, scene, index, evaluate=True)
    assert len(preds) == 3 and preds[0].shape == (256, 320) and gts[0].shape == (256, 320) and len(timer.times) == 3
    assert all(np.isfinite(p).all() and p.min() >= 0.25 - 1e-3 and p.max() <= 20.0 + 1e-2 for p in preds)
    preds_on, _, _ = predict_online(engine, scene, evaluate=False)
    assert len(preds_on) >= 5 and all(p.shape == (256, 320) for p in preds_on)
    save_results(preds, gts, "keyframe_test_320_256_2_dvmvs_fusionnet", "scene", str(tmp_path), max_depth=np.inf)
    assert os.path.exists(os.path.join(str(tmp_path), "keyframe_test_320_256_2_dvmvs_fusionnet_predictions_scene.npz"))
