"""Footprint statistics of the plane sweep on the sample scene's real keyframe pairs (CPU, numpy).

For every line of tests/golden/indices/...nmeas+2 the sample positions of all (pixel, plane, measurement frame) are
evaluated in float64, and for a list of (tile width, tile height, planes per chunk) the bounding box of each
(tile, chunk, frame) in the measurement image is measured -- the quantity the LDS-staged sweep kernel has to stage.
Used to choose tile shapes / LDS budgets in csrc/cost_volume.hip; not part of the product.

    python tools/sweep_geometry.py [--lines 0,50,117,202] [--every 8]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synthetic as syn  # noqa: E402

H, W, D = 128, 160, 64


def index_lines(nmeas=2):
    names = {n: i for i, n in enumerate(syn.sample_image_names())}
    path = os.path.join(ROOT, "tests", "golden", "indices", f"keyframe+hololens-dataset+000+nmeas+{nmeas}")
    out = []
    for line in open(path):
        parts = line.split()
        if len(parts) == nmeas + 1 and all(p in names for p in parts):
            out.append([names[p] for p in parts])
    return out


def sample_positions(pose1, pose2, K):
    """[D,H,W] x and y sample positions in measurement pixels (align-corners convention of the reference), and Z."""
    E = np.linalg.inv(pose2) @ pose1
    R, t = E[:3, :3], E[:3, 3]
    Hm = K @ R @ np.linalg.inv(K)
    kt = K @ t
    inv = 1.0 / syn.MAX_DEPTH + np.arange(D) * (1.0 / syn.MIN_DEPTH - 1.0 / syn.MAX_DEPTH) / (D - 1)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    p = np.stack([xs, ys, np.ones_like(xs)], 0).reshape(3, -1)
    base = (Hm @ p)[:, None, :] + kt[:, None, None] * inv[None, :, None]      # [3,D,HW]
    Z = base[2] + 1e-8
    u, v = base[0] / Z, base[1] / Z
    sx = u * (W - 1) / W
    sy = v * (H - 1) / H
    return sx.reshape(D, H, W), sy.reshape(D, H, W), Z.reshape(D, H, W)


def box_areas(sx, sy, Z, tw, th, dp):
    """areas [tiles_y, tiles_x, D/dp] of the clamped bounding boxes (incl. +1 tap and apron), -1 where Z <= 0 occurs,
    0 where the whole box is outside the image."""
    ty, tx, nd = H // th, W // tw, D // dp   # (a partial last tile column is left out of the statistics)
    a = lambda v: v[:nd * dp, :ty * th, :tx * tw].reshape(nd, dp, ty, th, tx, tw)
    lo_x = np.floor(a(sx).min(axis=(1, 3, 5)))
    hi_x = np.floor(a(sx).max(axis=(1, 3, 5))) + 1
    lo_y = np.floor(a(sy).min(axis=(1, 3, 5)))
    hi_y = np.floor(a(sy).max(axis=(1, 3, 5))) + 1
    bad = a(Z).min(axis=(1, 3, 5)) <= 1e-6
    x0, x1 = np.maximum(lo_x, -1), np.minimum(hi_x, W)
    y0, y1 = np.maximum(lo_y, -1), np.minimum(hi_y, H)
    rw, rh = x1 - x0 + 1, y1 - y0 + 1
    area = np.where((rw <= 0) | (rh <= 0), 0, rw * rh)
    area = np.where(bad, -1, area)
    return np.moveaxis(area, 0, -1), np.moveaxis(np.where(area > 0, rw, 0), 0, -1), np.moveaxis(np.where(area > 0, rh, 0), 0, -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", default="")
    ap.add_argument("--every", type=int, default=8)
    ap.add_argument("--configs", default="32x8x8,32x8x4,16x16x8,64x4x8,16x8x8,16x8x16,32x8x16,32x4x8,32x4x16,16x16x16")
    args = ap.parse_args()
    lines = index_lines(2)
    sel = [int(v) for v in args.lines.split(",")] if args.lines else list(range(0, len(lines), args.every))
    poses = syn.sample_poses()
    K = syn.scaled_K(syn.full_K(), 2.0)[0].double().numpy()
    configs = [tuple(int(v) for v in c.split("x")) for c in args.configs.split(",")]
    stats = {c: [] for c in configs}
    for li in sel:
        ref, *meas = lines[li]
        for m in meas:
            sx, sy, Z = sample_positions(poses[ref], poses[m], K)
            step = np.hypot(sx[1:] - sx[:-1], sy[1:] - sy[:-1])
            for c in configs:
                area, rw, rh = box_areas(sx, sy, Z, *c)
                stats[c].append(area.reshape(-1))
        print(f"line {li:3d}: ref {ref} meas {meas}  median per-plane step {np.median(step):.2f} px, max {step.max():.1f}")
    print(f"\n{len(sel)} lines x 2 frames.  records per box (tile x planes), over all (tile, chunk, frame):")
    print(f"{'config':>10} {'px*pl':>6} {'median':>7} {'p90':>6} {'p99':>6} {'max':>6}  {'rec/(px*pl)':>11}  "
          f"{'<=384':>6} {'<=512':>6} {'<=640':>6} {'<=768':>6} {'<=1024':>6} {'Z<=0':>6} {'empty':>6}")
    for c in configs:
        a = np.concatenate(stats[c])
        ok = a[a > 0]
        work = c[0] * c[1] * c[2]
        frac = lambda cap: np.mean((a >= 0) & (a <= cap))
        print(f"{c[0]:>3}x{c[1]:<2}x{c[2]:<3} {work:6d} {np.median(ok):7.0f} {np.percentile(ok, 90):6.0f} {np.percentile(ok, 99):6.0f} {ok.max():6.0f}  "
              f"{np.mean(ok) / work:11.3f}  {frac(384):6.3f} {frac(512):6.3f} {frac(640):6.3f} {frac(768):6.3f} {frac(1024):6.3f} "
              f"{np.mean(a < 0):6.3f} {np.mean(a == 0):6.3f}")


if __name__ == "__main__":
    main()
