"""Work model of the correlate-then-interpolate sweep (csrc/sweep_mfma.hip) on the sample scene's keyframe pairs (CPU, numpy).

For a pixel-group shape (gw x gh = 16 pixels), planes per wave and table capacity it counts, per keyframe pair (all M frames):
the 16-cell MFMA tiles (8 v_mfma_f32_16x16x4_f32 each), the waves, the lookup rounds per sample (strips), and prints the MFMA time at
the fp32 matrix peak.  Used to choose the shipped configuration before touching the GPU; not part of the product.

    python tools/sweep_mfma_model.py [--every 8] [--configs 4x4x16x128,8x2x16x128]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_geometry import index_lines, sample_positions, syn, H, W, D  # noqa: E402


def chunk_boxes(sx, sy, gw, gh, planes):
    """cells per (chunk, group) box and the number of alive samples in it; boxes hold the in-image taps of alive samples only"""
    ix = np.clip(np.nan_to_num(sx, nan=-1.0), -1.0, W)
    iy = np.clip(np.nan_to_num(sy, nan=-1.0), -1.0, H)
    alive = (ix > -1) & (ix < W) & (iy > -1) & (iy < H)
    x0, y0 = np.floor(ix), np.floor(iy)
    nd, ty, tx = D // planes, H // gh, W // gw
    a = lambda v: v.reshape(nd, planes, ty, gh, tx, gw)
    big = 1e9
    lo_x = a(np.where(alive, np.maximum(x0, 0), big)).min(axis=(1, 3, 5))
    hi_x = a(np.where(alive, np.minimum(x0 + 1, W - 1), -big)).max(axis=(1, 3, 5))
    lo_y = a(np.where(alive, np.maximum(y0, 0), big)).min(axis=(1, 3, 5))
    hi_y = a(np.where(alive, np.minimum(y0 + 1, H - 1), -big)).max(axis=(1, 3, 5))
    n_alive = a(alive).sum(axis=(1, 3, 5))
    cells = np.where(n_alive > 0, (hi_x - lo_x + 1) * (hi_y - lo_y + 1), 0)
    return cells, n_alive


def model(sx, sy, gw, gh, pw, cap, levels, thr=1.0):
    """tiles, lookup rounds and passes for one frame.  Hierarchy: a box of `levels[0]` planes is split into boxes of the next level while
    it has more than thr * cap cells; a box that is not split is processed in strips of cap cells (one lookup round per strip)."""
    fine = None
    for lv in sorted(levels):
        cells, n_alive = chunk_boxes(sx, sy, gw, gh, lv)
        t = np.ceil(cells / 16.0)
        rounds = np.ceil(cells / float(cap)) * lv
        own_p = (cells > 0) * 1
        if fine is None:
            cost_t, cost_r, cost_p = t, rounds, own_p
        else:
            f_t, f_r, f_p, f_lv = fine
            r = lv // f_lv
            nd = cells.shape[0]
            child = lambda v: v.reshape(nd, r, *v.shape[1:]).sum(axis=1)
            keep = cells <= thr * cap
            cost_t = np.where(keep, t, child(f_t))
            cost_r = np.where(keep, rounds, child(f_r))
            cost_p = np.where(keep, own_p, child(f_p) + 1)
        fine = (cost_t, cost_r, cost_p, lv)
    return fine[0].sum(), fine[1].sum(), fine[2].sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--every", type=int, default=8)
    ap.add_argument("--lines", default="")
    ap.add_argument("--configs", default="4x4x16x128x16.4,4x4x16x256x16.4,8x2x16x128x16.4,2x8x16x128x16.4,4x4x32x256x32.8,4x4x16x128x16.8.4,4x4x64x512x64.16.4,16x1x16x128x16.4")
    args = ap.parse_args()
    lines = index_lines(2)
    sel = [int(v) for v in args.lines.split(",")] if args.lines else list(range(0, len(lines), args.every))
    poses = syn.sample_poses()
    K = syn.scaled_K(syn.full_K(), 2.0)[0].double().numpy()
    configs = []
    for c in args.configs.split(","):
        p = c.split("x")
        configs.append((int(p[0]), int(p[1]), int(p[2]), int(p[3]), tuple(int(v) for v in p[4].split(".")), float(p[5]) if len(p) > 5 else 1.0))
    res = {c: [] for c in configs}
    for li in sel:
        ref, *meas = lines[li]
        pos = [sample_positions(poses[ref], poses[m], K) for m in meas]
        for c in configs:
            gw, gh, pw, cap, levels, thr = c
            tt = gg = pp = 0
            for sx, sy, _ in pos:
                t, g, p = model(sx, sy, gw, gh, pw, cap, levels, thr)
                tt += t; gg += g; pp += p
            res[c].append((tt, gg, pp))
    total_samples = D * H * W * 2
    print(f"{len(sel)} pairs; per pair (M=2): tiles, MFMA time at 157.3 TF (8 MFMA x 2048 flop per tile), gathered samples")
    for c in configs:
        r = np.array(res[c], dtype=np.float64)
        tiles = r[:, 0]
        us = tiles * 8 * 2048 / 157.3e12 * 1e6
        print(f"{c[0]}x{c[1]} pw{c[2]:<3} cap{c[3]:<4} lv{'.'.join(map(str, c[4])):<8} thr{c[5]:<4} tiles mean {tiles.mean():9.0f} max {tiles.max():9.0f} | mfma us mean {us.mean():5.2f} "
              f"p90 {np.percentile(us, 90):5.2f} max {us.max():5.2f} | passes/wave-frame {r[:, 2].mean() / (H * W / 16 * D / c[2] * 2):.2f} | lookup rounds per sample mean {np.mean(r[:, 1]) * 16 / total_samples:.3f} max {r[:, 1].max() * 16 / total_samples:.3f}")


if __name__ == "__main__":
    main()
