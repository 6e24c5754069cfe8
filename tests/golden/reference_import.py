"""Imports the REFERENCE implementation (``/root/reference/dvmvs``) for golden-vector generation.

Only usable in the build container (the reference tree does not exist on the GPU box) and only used by
``make_goldens.py``; nothing under ``tests/`` imports this at test time.  No reference source is copied: the
reference package is imported from where it lies, with stand-ins for the third-party imports it makes at module
level that are absent from this image:

* ``cv2``, ``path``, ``pytorch3d``, ``tensorboardX``: empty modules (never called on the paths we exercise);
* ``kornia`` (==0.3.2 in the reference's requirements): the four pin-hole helpers, supplied from
  ``oracle/dvmvs_oracle.py`` -- this is why parity of the warp / re-projection geometry is "unpinned at the kornia
  boundary" (see the oracle's header);
* ``torchvision``: ``models.mnasnet1_0`` / ``ops.FeaturePyramidNetwork`` supplied from this repo's
  ``dvmvs.backbone`` (key-compatible restatement; the real FPN checkpoint of the reference loads into it);
* ``torch.Tensor.cuda``: identity, because utils.py:141,149 hard-code ``.cuda()``.
"""
import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"
REPO_ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def _load_repo_module(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    """Returns a namespace with the reference modules: .utils, .convlstm, .fusionnet_model, .pairnet_model, .config."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "dvmvs")):
        raise RuntimeError("the reference tree is not available here; goldens can only be regenerated in the build container")
    for name in list(sys.modules):
        if name == "dvmvs" or name.startswith("dvmvs."):
            raise RuntimeError("a 'dvmvs' package is already imported in this process; run make_goldens.py on its own")

    oracle = _load_repo_module("dvmvs_oracle_for_goldens", "oracle/dvmvs_oracle.py")
    backbone = _load_repo_module("dvmvs_backbone_for_goldens", "deep-video-mvs_amd/dvmvs/backbone.py")

    for name in ("cv2", "path", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "tensorboardX"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["path"].Path = object
    sys.modules["pytorch3d"].structures = sys.modules["pytorch3d.structures"]
    sys.modules["pytorch3d"].renderer = sys.modules["pytorch3d.renderer"]

    kornia = types.ModuleType("kornia")

    def depth_to_3d(depth, camera_matrix, normalize_points=False):
        assert not normalize_points
        return oracle.depth_to_points(depth, camera_matrix).permute(0, 3, 1, 2)

    def transform_points(trans, points):
        # call sites pass trans[:, None] ([B,1,4,4]) and points [B,H,W,3]
        return oracle.rigid_transform(trans[:, 0], points)

    def project_points(points, camera_matrix):
        K = camera_matrix
        while K.dim() > 3:
            K = K[:, 0]
        return oracle.project(points, K)

    def normalize_pixel_coordinates(pixel_coordinates, height, width, eps=1e-8):
        hw = torch.tensor([width, height], dtype=pixel_coordinates.dtype)
        factor = torch.tensor(2.0, dtype=pixel_coordinates.dtype) / (hw - 1).clamp(eps)
        return factor * pixel_coordinates - 1

    kornia.depth_to_3d = depth_to_3d
    kornia.transform_points = transform_points
    kornia.project_points = project_points
    kornia.normalize_pixel_coordinates = normalize_pixel_coordinates
    kornia.adjust_brightness = kornia.adjust_gamma = kornia.adjust_contrast = None
    sys.modules["kornia"] = kornia

    tv = types.ModuleType("torchvision")
    tv_models = types.ModuleType("torchvision.models")
    tv_ops = types.ModuleType("torchvision.ops")

    class _MnasNetTrunk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.Sequential(*backbone.mnasnet1_0_trunk_layers())

    tv_models.mnasnet1_0 = lambda pretrained=False, **kw: _MnasNetTrunk()
    tv_ops.FeaturePyramidNetwork = backbone.FeaturePyramidNetwork
    tv.models, tv.ops = tv_models, tv_ops
    sys.modules.update({"torchvision": tv, "torchvision.models": tv_models, "torchvision.ops": tv_ops})

    torch.Tensor.cuda = lambda self, *a, **k: self

    sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    ns.utils = importlib.import_module("dvmvs.utils")
    ns.convlstm = importlib.import_module("dvmvs.convlstm")
    ns.config = importlib.import_module("dvmvs.config")
    ns.fusionnet_model = importlib.import_module("dvmvs.fusionnet.model")
    ns.pairnet_model = importlib.import_module("dvmvs.pairnet.model")
    ns.keyframe_buffer = importlib.import_module("dvmvs.keyframe_buffer")
    ns.errors = importlib.import_module("dvmvs.errors")
    ns.losses = importlib.import_module("dvmvs.losses")
    ns.dataset_loader = importlib.import_module("dvmvs.dataset_loader")
    ns.oracle = oracle
    assert ns.utils.__file__.startswith(REFERENCE_ROOT)
    return ns
