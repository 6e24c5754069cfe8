"""Writes tests/golden/preprocess_rows.npz: rows of the sample frame images/00012.png after the reference's pre-processing
(/root/reference/dvmvs/dataset_loader.py:325-341: centre crop, cv2.resize(INTER_LINEAR) to 320x256, /255, ImageNet
normalisation), computed WITHOUT this repo's loader: PIL decodes the PNG, torch.nn.functional.interpolate(mode="bilinear",
align_corners=False, antialias=False) resamples (the same half-pixel-centre, two-tap formula cv2 documents for INTER_LINEAR on
float32 images; cv2 itself is absent from this image, so this is as far as the resampling can be pinned), numpy normalises.

    python tests/golden/make_preprocess_fixture.py
"""
import os

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROWS = (0, 100, 255)


def main():
    image = np.asarray(Image.open(os.path.join(HERE, "sample_scene", "images", "00012.png")).convert("RGB"), dtype=np.float32)
    out = {}
    for tag, crop_x in (("crop", 45), ("nocrop", 0)):          # run-testing uses Config.test_perform_crop = False; training crops
        src = image[:, crop_x:image.shape[1] - crop_x]
        t = torch.from_numpy(np.ascontiguousarray(src)).permute(2, 0, 1)[None]
        res = torch.nn.functional.interpolate(t, size=(256, 320), mode="bilinear", align_corners=False, antialias=False)[0]
        res = res.permute(1, 2, 0).numpy() / 255.0
        res = (res - np.array([0.485, 0.456, 0.406], dtype=np.float32)) / np.array([0.229, 0.224, 0.225], dtype=np.float32)
        out[f"{tag}_rows"] = res[list(ROWS)].astype(np.float32)
    out["rows"] = np.array(ROWS)
    np.savez_compressed(os.path.join(HERE, "preprocess_rows"), **out)
    print("wrote preprocess_rows.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
