"""cv2-free image / depth loading and pre-processing for inference (host side).

Interface of the inference half of /root/reference/dvmvs/dataset_loader.py:260-346: ``load_image``, ``PreprocessImage``
(``apply_rgb``, ``apply_depth``, ``get_updated_intrinsics``).  OpenCV is not part of this stack; the two resampling modes
the reference uses are restated with numpy: ``cv2.INTER_LINEAR`` on float32 = bilinear with half-pixel centres and edge
clamping, no anti-aliasing; ``cv2.INTER_NEAREST`` = source index floor(dst * scale).  cv2 itself is absent here, so image
resampling parity is UNPINNED (intrinsics arithmetic is pinned: tests/test_runner.py checks it against the values the
survey measured from the reference).  The second half of the file is the training-time sub-sequence crawler and
``MVSDataset`` (dataset_loader.py:18-252, :349-496); the crawler's sample lists are pinned to the reference's by
tests/golden/crawler.json.
"""
import os

import numpy as np
import torch
import torch.utils.data
from PIL import Image


def load_image(path):
    """RGB image as float32 [H,W,3] in 0..255 (the reference reads BGR with cv2 and converts to RGB)."""
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32)


def load_depth_png(path, scaling=1000.0):
    """16-bit depth PNG in millimetres -> metres (run-testing.py:112: cv2.imread(path, -1) / 1000)."""
    return np.asarray(Image.open(path)).astype(np.float64) / scaling


def resize_bilinear(image, new_width, new_height):
    """cv2.resize(..., interpolation=cv2.INTER_LINEAR) for float arrays [H,W] or [H,W,C]."""
    h, w = image.shape[:2]
    sx, sy = w / float(new_width), h / float(new_height)
    xs = np.clip((np.arange(new_width) + 0.5) * sx - 0.5, 0, None)
    ys = np.clip((np.arange(new_height) + 0.5) * sy - 0.5, 0, None)
    x0 = np.minimum(np.floor(xs).astype(int), w - 1)
    y0 = np.minimum(np.floor(ys).astype(int), h - 1)
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    fx = (xs - x0).astype(np.float32)
    fy = (ys - y0).astype(np.float32)
    if image.ndim == 3:
        fx, fy = fx[None, :, None], fy[:, None, None]
    else:
        fx, fy = fx[None, :], fy[:, None]
    top = image[y0][:, x0] * (1 - fx) + image[y0][:, x1] * fx
    bottom = image[y1][:, x0] * (1 - fx) + image[y1][:, x1] * fx
    return (top * (1 - fy) + bottom * fy).astype(image.dtype)


def resize_nearest(image, new_width, new_height):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): source index = floor(dst * scale)."""
    h, w = image.shape[:2]
    xs = np.minimum((np.arange(new_width) * (w / float(new_width))).astype(int), w - 1)
    ys = np.minimum((np.arange(new_height) * (h / float(new_height))).astype(int), h - 1)
    return image[ys][:, xs]


class PreprocessImage:
    """Optional centre crop to the target aspect ratio, resize to the network size, matching intrinsics update."""

    def __init__(self, K, old_width, old_height, new_width, new_height, distortion_crop=0, perform_crop=True):
        fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
        self.new_width, self.new_height = new_width, new_height
        self.perform_crop = perform_crop
        self.crop_x = self.crop_y = 0
        width, height = float(old_width), float(old_height)
        if perform_crop:
            inner_w, inner_h = old_width - 2 * distortion_crop, old_height - 2 * distortion_crop
            target_ratio = float(new_width) / float(new_height)
            if float(inner_w) / float(inner_h) > target_ratio:      # too wide: crop columns
                self.crop_x = int(np.floor((inner_w - inner_h * target_ratio) / 2.0)) + distortion_crop
                self.crop_y = distortion_crop
            else:                                                    # too tall: crop rows
                self.crop_x = distortion_crop
                self.crop_y = int(np.floor((inner_h - inner_w / target_ratio) / 2.0)) + distortion_crop
            cx -= self.crop_x
            cy -= self.crop_y
            width, height = old_width - 2 * self.crop_x, old_height - 2 * self.crop_y
        factor_x, factor_y = float(new_width) / width, float(new_height) / height
        self.fx, self.fy, self.cx, self.cy = fx * factor_x, fy * factor_y, cx * factor_x, cy * factor_y

    def _crop(self, array):
        h, w = array.shape[:2]
        return array[self.crop_y:h - self.crop_y, self.crop_x:w - self.crop_x]

    def apply_depth(self, depth):
        return resize_nearest(self._crop(depth), self.new_width, self.new_height)

    def apply_rgb(self, image, scale_rgb, mean_rgb, std_rgb, normalize_colors=True):
        out = resize_bilinear(self._crop(image), self.new_width, self.new_height)
        if normalize_colors:
            out = (out / scale_rgb - np.asarray(mean_rgb, dtype=out.dtype)) / np.asarray(std_rgb, dtype=out.dtype)
        return out

    def get_updated_intrinsics(self):
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]])


# ----------------------------------------------------------------------------------------------------------------------
# training data: sub-sequence crawler and dataset (/root/reference/dvmvs/dataset_loader.py:18-252,
# :349-496).  Host-side sampling logic; which frames form a training sample is pinned to the reference by
# tests/golden/crawler.json (generated by running the reference's crawl functions on the sample scene's poses).
# ----------------------------------------------------------------------------------------------------------------------
def is_valid_pair(reference_pose, measurement_pose, pose_dist_min, pose_dist_max, t_norm_threshold=0.05, return_measure=False):
    from dvmvs.utils import pose_distance
    combined, _, translation = pose_distance(reference_pose, measurement_pose)
    result = bool(pose_dist_min <= combined <= pose_dist_max and translation >= t_norm_threshold)
    return (result, combined) if return_measure else result


def gather_pairs_train(poses, used_pairs, is_backward, initial_pose_dist_min, initial_pose_dist_max):
    """Greedy (reference, measurement) pairing along a sequence (dataset_loader.py:33-109): walk the frames forward (or
    backward); for frame i look for the nearest unused partner on the far side first, then (``check_future``) on the near
    side, with a pose distance inside [min, max]; when neither side has one the window is widened by 10 % and, after two
    widenings, the frame is skipped."""
    n = len(poses)
    dist_min, dist_max = initial_pose_dist_min, initial_pose_dist_max
    used_measurements = set()
    pairs = []
    if is_backward:
        i, step, first_limit, second_limit = n - 1, -1, 5, n - 5
    else:
        i, step, first_limit, second_limit = 0, 1, n - 5, 5
    check_future, loosened = False, 0
    while 0 <= i < n:
        candidates = range(i + step, first_limit, step) if check_future else range(i - step, second_limit, -step)
        partner = -1
        for j in candidates:
            if j in used_measurements or (i, j) in used_pairs:
                continue
            if is_valid_pair(poses[i], poses[j], dist_min, dist_max):
                partner = j
                break
        if partner != -1:
            pairs.append((i, partner))
            used_pairs.add((i, partner))
            used_pairs.add((partner, i))
            used_measurements.add(partner)
            dist_min, dist_max = initial_pose_dist_min, initial_pose_dist_max
            i += step
            check_future, loosened = False, 0
        elif check_future:
            dist_min, dist_max = dist_min / 1.1, dist_max * 1.1
            check_future = False
            loosened += 1
            if loosened > 1:
                i += step
                loosened = 0
        else:
            check_future = True
    return pairs


_SHORT_PASSES = ((1.0, False), (0.666, True), (1.5, False))
_LONG_PASSES = ((0, 1.0, False), (1, 0.666, True), (2, 1.5, False), (3, 0.8, True), (4, 1.25, False),
                (5, 1.0, True), (6, 0.666, False), (7, 1.5, True), (8, 0.8, False), (9, 1.25, True))


def crawl_scene_pairs(scene, poses):
    """Two-frame samples of one scene: three pairing passes with scaled distance windows (dataset_loader.py:112-133)."""
    from dvmvs.config import Config
    samples, used_pairs = [], set()
    for multiplier, is_backward in _SHORT_PASSES:
        for i, j in gather_pairs_train(poses, used_pairs, is_backward, multiplier * Config.train_minimum_pose_distance,
                                       multiplier * Config.train_maximum_pose_distance):
            samples.append({"scene": scene, "indices": [i, j]})
    return samples


def crawl_scene_subsequences(scene, poses, subsequence_length):
    """Sub-sequences of ``subsequence_length`` frames of one scene (dataset_loader.py:136-219): ten passes (offset, distance
    multiplier, direction); from each start frame the next frames are appended greedily when they are not over-used, the
    pair was not used before and the pose distance to the previously accepted frame is inside the scaled window."""
    from dvmvs.config import Config
    n = len(poses)
    step_size = Config.train_crawl_step
    usage = [0] * n
    used_pairs = set()
    samples = []
    for offset, multiplier, is_backward in _LONG_PASSES:
        offset %= step_size
        starts = range(n - 1 - offset, subsequence_length, -step_size) if is_backward else range(offset, n - subsequence_length + 1, step_size)
        lo, hi = multiplier * Config.train_minimum_pose_distance, multiplier * Config.train_maximum_pose_distance
        for i in starts:
            if usage[i] > 1:
                continue
            indices, previous, hop, ran_out = [i], i, 1, False
            while len(indices) < subsequence_length:
                j = i - hop if is_backward else i + hop
                ran_out = j < 0 or j >= n
                if ran_out:
                    break
                if usage[j] <= 1 and (previous, j) not in used_pairs and is_valid_pair(poses[previous], poses[j], lo, hi, t_norm_threshold=lo * 0.5):
                    indices.append(j)
                    previous = j
                hop += 1
            if ran_out:
                continue
            for a, b in zip(indices, indices[1:]):
                used_pairs.add((a, b))
                used_pairs.add((b, a))
            for k in indices:
                usage[k] += 1
            samples.append({"scene": scene, "indices": indices})
    return samples


def _crawl_one(scene, dataset_path, subsequence_length):
    poses = np.reshape(np.loadtxt(os.path.join(str(dataset_path), scene, "poses.txt")), (-1, 4, 4))
    return crawl_scene_pairs(scene, poses) if subsequence_length == 2 else crawl_scene_subsequences(scene, poses, subsequence_length)


def crawl(dataset_path, scenes, subsequence_length, num_workers=1, shuffle=True):
    """All samples of ``scenes`` (dataset_loader.py:222-248), shuffled with the ``random`` module like the reference."""
    import random
    from functools import partial
    scenes = [str(s) for s in scenes]
    work = partial(_crawl_one, dataset_path=str(dataset_path), subsequence_length=subsequence_length)
    if num_workers > 1 and len(scenes) > 1:
        from multiprocessing.pool import Pool
        with Pool(num_workers) as pool:
            per_scene = pool.map(work, scenes)
    else:
        per_scene = [work(s) for s in scenes]
    samples = [sample for scene_samples in per_scene for sample in scene_samples]
    if shuffle:
        random.shuffle(samples)
    return samples


def read_split(path):
    """Scene names, one per line (dataset_loader.py:251-253 reads them with np.loadtxt(dtype=str, delimiter=newline), which
    numpy 2 rejects; same result, and a one-line split stays iterable)."""
    with open(str(path)) as f:
        return np.array([line.strip() for line in f if line.strip() and not line.lstrip().startswith("#")], dtype=str)


def adjust_gamma(image, gamma):        # kornia 0.3.2 colour ops on [0,1] images, restated (enhance/adjust.py): clamp to [0,1]
    return torch.clamp(torch.pow(image, gamma), 0.0, 1.0)


def adjust_contrast(image, factor):
    return torch.clamp(image * factor, 0.0, 1.0)


def adjust_brightness(image, delta):
    return torch.clamp(image + delta, 0.0, 1.0)


class MVSDataset(torch.utils.data.Dataset):
    """Training / validation samples: ``subsequence_length`` posed RGB-D frames of one scene, resized and cropped to the
    training resolution, with the reference's augmentations (dataset_loader.py:349-496): random reversal, geometric scale of
    depths and pose translations, and one colour jitter (gamma, contrast, brightness in random order) shared by the frames.
    A scene folder holds ``K.txt``, ``poses.txt`` and one ``*.npz`` per frame with ``image`` (uint8 RGB) and ``depth``
    (uint16 millimetres).  Returns (images, depths, poses, K) as lists of tensors, like the reference."""

    def __init__(self, root, seed, split, subsequence_length, scale_rgb, mean_rgb, std_rgb, geometric_scale_augmentation=False):
        import random
        from dvmvs.config import Config
        np.random.seed(seed)
        random.seed(seed)
        self.subsequence_length = subsequence_length
        self.geometric_scale_augmentation = geometric_scale_augmentation
        self.root = str(root)
        self.split = split
        if split not in ("TRAINING", "VALIDATION"):
            raise ValueError("split must be TRAINING or VALIDATION")
        self.scenes = read_split(os.path.join(self.root, "train.txt" if split == "TRAINING" else "validation.txt"))
        self.samples = crawl(dataset_path=self.root, scenes=self.scenes, subsequence_length=subsequence_length,
                             num_workers=Config.train_data_pipeline_workers)
        self.scale_rgb, self.mean_rgb, self.std_rgb = scale_rgb, mean_rgb, std_rgb

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, sample_index):
        import random
        from dvmvs.config import Config
        sample = self.samples[sample_index]
        scene_path = os.path.join(self.root, sample["scene"])
        indices = sample["indices"]
        K = np.loadtxt(os.path.join(scene_path, "K.txt"), dtype=np.float32)
        scene_poses = np.reshape(np.loadtxt(os.path.join(scene_path, "poses.txt"), dtype=np.float32), (-1, 4, 4))
        frame_files = sorted(n for n in os.listdir(scene_path) if n.endswith(".npz"))
        if self.split == "TRAINING" and np.random.random() > 0.5:
            indices.reverse()      # in place, as in the reference: the stored sample stays reversed
        frames = [np.load(os.path.join(scene_path, frame_files[i])) for i in indices]
        raw_images, raw_depths = [f["image"] for f in frames], [f["depth"] for f in frames]
        pre = PreprocessImage(K=K, old_width=raw_images[0].shape[1], old_height=raw_depths[0].shape[0],
                              new_width=Config.train_image_width, new_height=Config.train_image_height, distortion_crop=0)
        depths, images, rgb_sum = [], [], 0.0
        nearest, farthest = Config.train_max_depth, Config.train_min_depth
        for image, depth in zip(raw_images, raw_depths):
            depth = depth.astype(np.float32) / 1000.0
            depth[~np.isfinite(depth)] = 0
            depth = pre.apply_depth(depth)
            valid = depth[depth > 0]
            if len(valid) > 0:
                nearest, farthest = min(nearest, float(valid.min())), max(farthest, float(valid.max()))
            image = pre.apply_rgb(image=image, scale_rgb=1.0, mean_rgb=[0.0, 0.0, 0.0], std_rgb=[1.0, 1.0, 1.0], normalize_colors=False)
            rgb_sum += float(np.sum(image))
            depths.append(depth)
            images.append(image)
        rgb_average = rgb_sum / (len(images) * Config.train_image_height * Config.train_image_width * 3)

        scale = 1.0
        if self.geometric_scale_augmentation:
            lowest, highest = Config.train_min_depth / nearest, Config.train_max_depth / farthest
            if np.random.random() > 0.5:
                scale = np.random.uniform(low=max(lowest, 0.666), high=min(highest, 1.5))
            else:
                scale = np.random.uniform(low=max(lowest, 0.8), high=min(highest, 1.25))
        brightness, contrast, gamma = random.uniform(-0.03, 0.03), random.uniform(0.8, 1.2), random.uniform(0.8, 1.2)
        jitter = [(adjust_gamma, gamma), (adjust_contrast, contrast), (adjust_brightness, brightness)]
        random.shuffle(jitter)

        out_images, out_depths, out_poses = [], [], []
        for image, depth, i in zip(images, depths, indices):
            image = torch.from_numpy(np.transpose(image, (2, 0, 1)).astype(np.float32)) / 255.0
            if self.split == "TRAINING" and 55.0 < rgb_average < 200.0:
                for fn, value in jitter:
                    image = fn(image, value)
            image = (image * 255.0) / self.scale_rgb
            for c in range(3):
                image[c] = (image[c] - self.mean_rgb[c]) / self.std_rgb[c]
            pose = scene_poses[i].astype(np.float32).copy()
            pose[0:3, 3] *= scale
            out_images.append(image)
            out_depths.append(torch.from_numpy((depth * scale).astype(np.float32)))
            out_poses.append(torch.from_numpy(pose))
        return out_images, out_depths, out_poses, torch.from_numpy(pre.get_updated_intrinsics().astype(np.float32))
