"""Volumetric TSDF fusion of posed RGB-D frames on an MI355X (surface of the ``TSDFVolume`` / ``TSDFFusion`` classes of the
reference's reconstruction script, /root/reference/sample-data/run-tsdf-reconstruction.py:30-330).

``TSDFVolume.integrate`` is the hot function: one HIP launch (``dvmvs_tsdf_integrate``) updates the whole voxel volume in
place in HBM; the volumes never leave the device until ``get_volume()``.  The reference compiles an equivalent CUDA kernel
with pycuda and launches it once per "gpu loop"; its numba CPU fall-back has no counterpart here (GPU only, like the rest of
the package).  Marching cubes (``get_mesh`` / ``get_point_cloud``) come from scikit-image in the reference and are out of
scope: use ``get_volume()`` with any iso-surface extractor.
"""
import numpy as np
import torch

from dvmvs.hip import _capi


def fold_color(color_im):
    """[H,W,3] RGB (0..255) -> float32 [H,W] holding b * 65536 + g * 256 + r (run-tsdf-reconstruction.py:236-238)."""
    c = np.asarray(color_im, dtype=np.float32)
    return np.floor(c[..., 2] * np.float32(65536.0) + c[..., 1] * np.float32(256.0) + c[..., 0]).astype(np.float32)


class TSDFVolume:
    """Voxel volume over ``vol_bnds`` ([[x0, x1], [y0, y1], [z0, z1]] in metres) with ``voxel_size`` edges.  tsdf starts at 1,
    weight and colour at 0; truncation = 5 voxels, as in the reference."""

    def __init__(self, vol_bnds, voxel_size, device="cuda", use_gpu=True):
        if not use_gpu:
            raise RuntimeError("TSDFVolume runs on an MI355X only: there is no CPU integration path in this package")
        vol_bnds = np.array(vol_bnds, dtype=np.float64)
        assert vol_bnds.shape == (3, 2), "[!] `vol_bnds` should be of shape (3, 2)."
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("TSDFVolume needs a HIP device")
        self._voxel_size = float(voxel_size)
        self._trunc_margin = 5 * self._voxel_size
        self._color_const = 256 * 256
        self._vol_dim = np.ceil((vol_bnds[:, 1] - vol_bnds[:, 0]) / self._voxel_size).astype(int)
        vol_bnds[:, 1] = vol_bnds[:, 0] + self._vol_dim * self._voxel_size
        self._vol_bnds = vol_bnds
        self._vol_origin = vol_bnds[:, 0].astype(np.float32)
        dims = tuple(int(d) for d in self._vol_dim)
        self._tsdf = torch.ones(dims, dtype=torch.float32, device=self.device)
        self._weight = torch.zeros(dims, dtype=torch.float32, device=self.device)
        self._color = torch.zeros(dims, dtype=torch.float32, device=self.device)

    @property
    def vol_dim(self):
        return self._vol_dim

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1.0):
        """Fuses one frame: ``color_im`` [H,W,3] RGB, ``depth_im`` [H,W] metres (0 = invalid), ``cam_intr`` [3,3],
        ``cam_pose`` [4,4] camera-to-world; numpy arrays or tensors (device tensors are used in place)."""
        dev = self.device
        depth = torch.as_tensor(np.asarray(depth_im, dtype=np.float32) if not torch.is_tensor(depth_im) else depth_im,
                                dtype=torch.float32, device=dev).contiguous()
        if torch.is_tensor(color_im) and color_im.dim() == 2:
            color = color_im.to(dev, torch.float32).contiguous()            # already folded
        else:
            color = torch.from_numpy(fold_color(color_im.cpu().numpy() if torch.is_tensor(color_im) else color_im)).to(dev)
        if color.shape != depth.shape:
            raise ValueError(f"colour {tuple(color.shape)} and depth {tuple(depth.shape)} images differ in size")
        K = torch.as_tensor(np.asarray(cam_intr, dtype=np.float32) if not torch.is_tensor(cam_intr) else cam_intr,
                            dtype=torch.float32, device=dev).reshape(3, 3).contiguous()
        P = torch.as_tensor(np.asarray(cam_pose, dtype=np.float32) if not torch.is_tensor(cam_pose) else cam_pose,
                            dtype=torch.float32, device=dev).reshape(4, 4).contiguous()
        im_h, im_w = depth.shape
        x, y, z = (int(d) for d in self._vol_dim)
        with torch.cuda.device(dev):
            rc = _capi.lib().dvmvs_tsdf_integrate(
                self._tsdf.data_ptr(), self._weight.data_ptr(), self._color.data_ptr(), x, y, z,
                float(self._vol_origin[0]), float(self._vol_origin[1]), float(self._vol_origin[2]), self._voxel_size,
                K.data_ptr(), P.data_ptr(), color.data_ptr(), depth.data_ptr(), im_h, im_w, self._trunc_margin, float(obs_weight),
                torch.cuda.current_stream(dev).cuda_stream)
        _capi.check(rc, "dvmvs_tsdf_integrate")

    def get_volume(self):
        """(tsdf, colour) as numpy arrays, like the reference; ``get_weight_volume()`` for the weights."""
        return self._tsdf.cpu().numpy(), self._color.cpu().numpy()

    def get_weight_volume(self):
        return self._weight.cpu().numpy()

    def get_mesh(self):
        raise NotImplementedError("marching cubes (scikit-image in the reference) is out of scope: extract the iso-surface from get_volume()")

    get_point_cloud = get_mesh


class TSDFFusion:
    """Host-side helpers of the reference class of the same name."""

    @staticmethod
    def rigid_transform(xyz, transform):
        xyz_h = np.hstack([xyz, np.ones((len(xyz), 1), dtype=np.float32)])
        return np.dot(transform, xyz_h.T).T[:, :3]

    @staticmethod
    def get_view_frustum(depth_im, cam_intr, cam_pose):
        """[3,5] world-space corners (camera centre + the four far corners) of the frame's view frustum."""
        im_h, im_w = depth_im.shape[0], depth_im.shape[1]
        far = np.max(depth_im)
        xs = (np.array([0, 0, 0, im_w, im_w]) - cam_intr[0, 2]) * np.array([0, far, far, far, far]) / cam_intr[0, 0]
        ys = (np.array([0, 0, im_h, 0, im_h]) - cam_intr[1, 2]) * np.array([0, far, far, far, far]) / cam_intr[1, 1]
        pts = np.array([xs, ys, np.array([0, far, far, far, far])])
        return TSDFFusion.rigid_transform(pts.T, cam_pose).T

    @staticmethod
    def volume_bounds(frames):
        """Axis-aligned bounds [[x0,x1],[y0,y1],[z0,z1]] enclosing the view frusta of ``frames`` = iterable of
        (depth_im, cam_intr, cam_pose), the way the reference's main loop accumulates them."""
        bounds = np.zeros((3, 2))
        bounds[:, 0], bounds[:, 1] = np.inf, -np.inf
        for depth_im, cam_intr, cam_pose in frames:
            frustum = TSDFFusion.get_view_frustum(depth_im, cam_intr, cam_pose)
            bounds[:, 0] = np.minimum(bounds[:, 0], np.amin(frustum, axis=1))
            bounds[:, 1] = np.maximum(bounds[:, 1], np.amax(frustum, axis=1))
        return bounds
