"""Probe (MI355X): every dense (groups = 1) convolution of one fusionnet frame at batch 1, one at a time -- what MIOpen's immediate mode
gives each layer (the engine's setting: cudnn.benchmark off, the atomic NHWC implicit-GEMM kernels disabled, see dvmvs/engine.py) against
the layer's fp32 FLOPs at the 157 TFLOP/s MFMA / packed-FMA peak.  Shapes are read off the modules with forward hooks on the CPU.

    python tools/conv_layer_probe.py [--min-gflop 0.05]
"""
import argparse
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "deep-video-mvs_amd"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from dvmvs.engine import _graph_microseconds  # noqa: E402  (also sets the MIOpen environment switch of the engine)
from dvmvs.fusionnet import model as fm  # noqa: E402


def frame_layers():
    torch.manual_seed(0)
    fe, fs, ce, cd = fm.FeatureExtractor(), fm.FeatureShrinker(), fm.CostVolumeEncoder(), fm.CostVolumeDecoder()
    records = []

    def hook(name):
        def h(m, i, o):
            records.append((name, tuple(i[0].shape), tuple(o.shape), m.kernel_size[0], m.stride[0], m.padding[0], m.groups))
        return h

    for n, mod in dict(fe=fe, fs=fs, ce=ce, cd=cd).items():
        mod.eval()
        for k, m in mod.named_modules():
            if isinstance(m, torch.nn.Conv2d):
                m.register_forward_hook(hook(n + "." + k))
    with torch.no_grad():
        img = torch.randn(1, 3, 256, 320)
        skips = ce(*fs(*fe(img)), torch.randn(1, 64, 128, 160))
        cd(img, *skips[:-1], torch.zeros(1, 512, 8, 10))
    return records


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-gflop", type=float, default=0.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    total_us = total_flop = 0.0
    groups = {}
    for name, xi, xo, k, s, p, g in frame_layers():
        flop = 2.0 * xo[1] * xo[2] * xo[3] * (xi[1] // g) * k * k
        if flop < args.min_gflop * 1e9:
            continue
        x = torch.randn(*xi, device=dev)
        w = torch.randn(xo[1], xi[1] // g, k, k, device=dev) / (xi[1] * k * k) ** 0.5
        us = _graph_microseconds(lambda: F.conv2d(x, w, stride=s, padding=p, groups=g), reps=10, rounds=3)
        kind = "1x1" if k == 1 else ("depthwise" if g > 1 else f"dense {k}x{k}")
        print(f"{name:48s} {kind:10s} s{s} {xi[1]:4d}x{xi[2]:3d}x{xi[3]:3d} -> {xo[1]:4d}x{xo[2]:3d}x{xo[3]:3d}  {flop / 1e9:6.3f} GFLOP "
              f"{us:7.2f} us  {flop / us / 1e6:6.1f} TFLOP/s  ({flop / 157.3e6:6.2f} us at the fp32 peak)", flush=True)
        acc = groups.setdefault(kind, [0.0, 0.0, 0])
        acc[0] += us
        acc[1] += flop
        acc[2] += 1
        total_us += us
        total_flop += flop
    for kind, (us, flop, n) in groups.items():
        print(f"{kind:10s}: {n:3d} layers, {us:8.1f} us, {flop / 1e9:7.2f} GFLOP, {flop / us / 1e6:6.1f} TFLOP/s")
    print(f"all convolutions of a frame one after the other (plain F.conv2d, no epilogue): {total_us:.1f} us, {total_flop / 1e9:.2f} GFLOP")


if __name__ == "__main__":
    main()
