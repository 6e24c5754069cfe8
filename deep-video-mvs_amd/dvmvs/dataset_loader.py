"""cv2-free image / depth loading and pre-processing for inference (host side).

Interface of the inference half of /root/reference/dvmvs/dataset_loader.py:260-346: ``load_image``, ``PreprocessImage``
(``apply_rgb``, ``apply_depth``, ``get_updated_intrinsics``).  OpenCV is not part of this stack; the two resampling modes
the reference uses are restated with numpy: ``cv2.INTER_LINEAR`` on float32 = bilinear with half-pixel centres and edge
clamping, no anti-aliasing; ``cv2.INTER_NEAREST`` = source index floor(dst * scale).  cv2 itself is absent here, so image
resampling parity is UNPINNED (intrinsics arithmetic is pinned: tests/test_runner.py checks it against the values the
survey measured from the reference).  The training-time dataset crawler / augmentation is out of scope.
"""
import numpy as np
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
