"""Training data path on the CPU: the sub-sequence crawler against the reference's own output (tests/golden/crawler.json,
made by tests/golden/make_goldens.py --only-crawler from /root/reference/dvmvs/dataset_loader.py:112-219) and MVSDataset's
sample contract (dataset_loader.py:349-496)."""
import json
import os
import random

import numpy as np
import pytest
import torch

import synthetic as syn


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(syn.GOLDEN_DIR, "crawler.json")) as f:
        return json.load(f)


def scenes():
    return {"sample": syn.sample_poses(), "synthetic": syn.synthetic_trajectory(240, seed=77)}


def through_text(poses):
    """The reference crawls what np.loadtxt reads back from poses.txt."""
    import io
    buf = io.StringIO()
    np.savetxt(buf, np.reshape(poses, (-1, 16)))
    buf.seek(0)
    return np.reshape(np.loadtxt(buf), (-1, 4, 4))


@pytest.mark.parametrize("scene", ["sample", "synthetic"])
def test_pair_crawler_matches_the_reference(golden, scene):
    from dvmvs.dataset_loader import crawl_scene_pairs
    samples = crawl_scene_pairs(scene, through_text(scenes()[scene]))
    assert [s["indices"] for s in samples] == golden[f"{scene}_pairs"]
    assert all(s["scene"] == scene for s in samples)


@pytest.mark.parametrize("scene", ["sample", "synthetic"])
@pytest.mark.parametrize("length", [3, 8])
def test_subsequence_crawler_matches_the_reference(golden, scene, length):
    from dvmvs.dataset_loader import crawl_scene_subsequences
    samples = crawl_scene_subsequences(scene, through_text(scenes()[scene]), length)
    assert [s["indices"] for s in samples] == golden[f"{scene}_len{length}"]
    for s in samples:
        assert len(s["indices"]) == length and len(set(s["indices"])) == length


def test_is_valid_pair_measure(golden):
    from dvmvs.dataset_loader import is_valid_pair
    for scene, poses in scenes().items():
        for j, (valid, measure) in zip((1, 5, 20), golden[f"{scene}_is_valid_pair"]):
            got_valid, got_measure = is_valid_pair(poses[0], poses[j], 0.125, 0.325, return_measure=True)
            assert got_valid == valid and abs(got_measure - measure) <= 1e-12 * max(1.0, abs(measure))


def write_scene(folder, poses, n_frames, height=60, width=90, seed=0):
    rng = np.random.RandomState(seed)
    os.makedirs(folder)
    np.savetxt(os.path.join(folder, "poses.txt"), np.reshape(poses[:n_frames], (-1, 16)))
    np.savetxt(os.path.join(folder, "K.txt"), np.array([[80.0, 0, 45.0], [0, 80.0, 30.0], [0, 0, 1.0]]))
    for i in range(n_frames):
        image = (syn.smooth_noise((height, width, 3), seed=seed + i).numpy() * 40 + 120).clip(0, 255).astype(np.uint8)
        depth = (1500 + 800 * syn.smooth_noise((height, width), seed=1000 + seed + i).numpy()).clip(300, 9000).astype(np.uint16)
        depth[rng.rand(height, width) < 0.05] = 0
        np.savez(os.path.join(folder, "%06d.npz" % i), image=image, depth=depth)


@pytest.fixture()
def dataset_root(tmp_path):
    poses = syn.synthetic_trajectory(60, seed=5)
    write_scene(str(tmp_path / "scene_a"), poses, 40, seed=1)
    write_scene(str(tmp_path / "scene_b"), poses[10:], 36, seed=2)
    (tmp_path / "train.txt").write_text("scene_a\nscene_b\n")
    (tmp_path / "validation.txt").write_text("scene_b\n")
    return str(tmp_path)


def make_dataset(root, split, length, **kw):
    from dvmvs.config import Config
    from dvmvs.dataset_loader import MVSDataset
    old = Config.train_image_width, Config.train_image_height, Config.train_data_pipeline_workers
    Config.train_image_width, Config.train_image_height, Config.train_data_pipeline_workers = 64, 48, 1
    try:
        ds = MVSDataset(root=root, seed=3, split=split, subsequence_length=length, scale_rgb=255.0, mean_rgb=[0.485, 0.456, 0.406],
                        std_rgb=[0.229, 0.224, 0.225], **kw)
        items = [ds[i] for i in range(min(3, len(ds)))]
    finally:
        Config.train_image_width, Config.train_image_height, Config.train_data_pipeline_workers = old
    return ds, items


def test_dataset_sample_contract(dataset_root):
    ds, items = make_dataset(dataset_root, "VALIDATION", 4)
    assert len(ds) > 0 and {s["scene"] for s in ds.samples} == {"scene_b"}
    images, depths, poses, K = items[0]
    assert len(images) == len(depths) == len(poses) == 4
    assert images[0].shape == (3, 48, 64) and images[0].dtype == torch.float32
    assert depths[0].shape == (48, 64) and depths[0].dtype == torch.float32
    assert poses[0].shape == (4, 4) and K.shape == (3, 3)
    # 90x60 -> 64x48: the wider source is cropped horizontally to 4:3 (80 wide, 5 px each side), then scaled by 0.8
    np.testing.assert_allclose(K.numpy(), [[64.0, 0, 32.0], [0, 64.0, 24.0], [0, 0, 1]], rtol=1e-6)
    # validation: no jitter, no scale -> normalised colours of the resized frame, depth in metres, stored poses
    scene_poses = np.reshape(np.loadtxt(os.path.join(dataset_root, "scene_b", "poses.txt"), dtype=np.float32), (-1, 4, 4))
    idx = ds.samples[0]["indices"]
    for pose, i in zip(poses, idx):
        np.testing.assert_array_equal(pose.numpy(), scene_poses[i])
    raw = np.load(os.path.join(dataset_root, "scene_b", "%06d.npz" % idx[0]))
    d = depths[0].numpy()
    assert 0.3 <= d[d > 0].min() and d.max() <= 9.0 and (d == 0).any()
    assert set(np.unique(d)).issubset(set(np.unique(raw["depth"].astype(np.float32) / 1000.0)))
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[:, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[:, None, None]
    back = (images[0].numpy() * std + mean) * 255.0
    assert abs(float(back.mean()) - float(raw["image"][:, 5:85].mean())) < 1.0


def test_dataset_training_augmentations(dataset_root):
    ds, items = make_dataset(dataset_root, "TRAINING", 3, geometric_scale_augmentation=True)
    assert {s["scene"] for s in ds.samples} == {"scene_a", "scene_b"}
    from dvmvs.config import Config
    for (images, depths, poses, K), sample in zip(items, ds.samples):
        scene_poses = np.reshape(np.loadtxt(os.path.join(dataset_root, sample["scene"], "poses.txt"), dtype=np.float32), (-1, 4, 4))
        # one geometric scale for the whole sample: translations and depths scaled alike, rotations untouched
        scales = []
        for pose, i in zip(poses, sample["indices"]):
            np.testing.assert_array_equal(pose[:3, :3].numpy(), scene_poses[i][:3, :3])
            t, t0 = pose[:3, 3].numpy(), scene_poses[i][:3, 3]
            k = int(np.argmax(np.abs(t0)))
            scales.append(t[k] / t0[k])
        assert np.allclose(scales, scales[0], rtol=1e-5) and 0.6 <= scales[0] <= 1.55
        dmax = max(float(d.max()) for d in depths)
        dmin = min(float(d[d > 0].min()) for d in depths)
        assert dmin >= Config.train_min_depth * (1 - 1e-5) and dmax <= Config.train_max_depth * (1 + 1e-5)
        for image in images:
            assert torch.isfinite(image).all()


def test_dataset_batches_like_the_training_scripts_expect(dataset_root):
    """run-training.py:135-141 wraps the dataset in a DataLoader; forward_pass (:184) takes lists of batched tensors."""
    from dvmvs.config import Config
    old = Config.train_image_width, Config.train_image_height, Config.train_data_pipeline_workers
    Config.train_image_width, Config.train_image_height, Config.train_data_pipeline_workers = 64, 48, 1
    try:
        from dvmvs.dataset_loader import MVSDataset
        ds = MVSDataset(root=dataset_root, seed=3, split="VALIDATION", subsequence_length=3, scale_rgb=255.0,
                        mean_rgb=[0.485, 0.456, 0.406], std_rgb=[0.229, 0.224, 0.225])
        loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0, drop_last=True)
        images, depths, poses, K = next(iter(loader))
    finally:
        Config.train_image_width, Config.train_image_height, Config.train_data_pipeline_workers = old
    assert len(images) == len(depths) == len(poses) == 3
    assert images[0].shape == (2, 3, 48, 64) and depths[0].shape == (2, 48, 64) and poses[0].shape == (2, 4, 4) and K.shape == (2, 3, 3)


def test_colour_jitter_operators():
    from dvmvs.dataset_loader import adjust_brightness, adjust_contrast, adjust_gamma
    x = torch.tensor([0.0, 0.25, 0.5, 1.0])
    assert torch.allclose(adjust_gamma(x, 2.0), torch.tensor([0.0, 0.0625, 0.25, 1.0]))
    assert torch.allclose(adjust_contrast(x, 1.2), torch.tensor([0.0, 0.3, 0.6, 1.0]))
    assert torch.allclose(adjust_brightness(x, -0.03), torch.tensor([0.0, 0.22, 0.47, 0.97]))


def test_crawl_is_seeded_and_covers_every_scene(dataset_root):
    from dvmvs.dataset_loader import crawl
    random.seed(11)
    a = crawl(dataset_root, ["scene_a", "scene_b"], 3, num_workers=1)
    random.seed(11)
    b = crawl(dataset_root, ["scene_a", "scene_b"], 3, num_workers=2)
    assert a == b and {s["scene"] for s in a} == {"scene_a", "scene_b"}
    unshuffled = crawl(dataset_root, ["scene_a", "scene_b"], 3, shuffle=False)
    assert sorted(map(str, a)) == sorted(map(str, unshuffled)) and a != unshuffled
