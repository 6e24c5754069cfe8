#!/usr/bin/env python3
"""Per-layer timing of the 1x1 convolutions of a 320x256 frame: the library convolution (MIOpen -> rocBLAS GEMM) + dvmvs_bias_act_fwd launch
pair against dvmvs_pointwise_conv_fwd (csrc/pointwise_conv.hip), the latter with its automatic split count and with every power of two.
Ten launches per hipGraph, best of three replays (the layer's weights stay in L2 between the launches: for both sides alike).

    python tools/pointwise_probe.py [--out profiles/rNN_pointwise_conv_layers.txt]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deep-video-mvs_amd"))

# (C_in, C_out, H, W, launches of this shape per frame, epilogue: 0 raw (the depthwise consumer applies it), 1 bias, 2 bias + residual, 3 bias + FPN top-down add)
LAYERS = [
    (32, 16, 128, 160, 1, 1), (16, 48, 128, 160, 1, 0), (48, 24, 64, 80, 1, 1), (24, 72, 64, 80, 3, 0), (72, 24, 64, 80, 2, 2), (72, 40, 32, 40, 1, 1),
    (40, 120, 32, 40, 2, 0), (120, 40, 32, 40, 2, 2), (40, 240, 32, 40, 1, 0), (240, 80, 16, 20, 1, 1), (80, 480, 16, 20, 3, 0), (480, 80, 16, 20, 2, 2),
    (480, 96, 16, 20, 1, 1), (96, 576, 16, 20, 2, 0), (576, 96, 16, 20, 1, 2), (576, 192, 8, 10, 1, 1), (192, 1152, 8, 10, 4, 0), (1152, 192, 8, 10, 3, 2),
    (1152, 320, 8, 10, 1, 1), (320, 32, 8, 10, 1, 1), (96, 32, 16, 20, 1, 3), (40, 32, 32, 40, 1, 3), (24, 32, 64, 80, 1, 3), (16, 32, 128, 160, 1, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    args = ap.parse_args()
    from dvmvs.engine import _graph_microseconds
    from dvmvs.hip import ops
    dev = torch.device("cuda:0")
    lines = ["# C_in -> C_out @ HxW  x launches/frame | library GEMM (+ epilogue launch) us | pointwise kernel us (auto) | splits 1 2 4 8 16"]
    total_lib = total_pw = 0.0
    for C_in, C_out, H, W, n, ep in LAYERS:
        x = torch.randn(1, C_in, H, W, device=dev)
        w = torch.randn(C_out, C_in, 1, 1, device=dev) / C_in ** 0.5
        bias = torch.randn(C_out, device=dev)
        res = torch.randn(1, C_out, H, W, device=dev) if ep == 2 else (torch.randn(1, C_out, H // 2, W // 2, device=dev) if ep == 3 else None)
        mode = {0: 0, 1: 0, 2: 1, 3: 2}[ep]
        dst = torch.empty(1, C_out, H, W, device=dev)
        packed = ops.pointwise_conv_pack(w)

        def library():
            y = F.conv2d(x, w)
            if ep:
                ops.bias_act_into(y, dst, bias, 0, res, mode)

        def pointwise(splits=0):
            ops.pointwise_conv_into(x, packed, bias if ep else None, dst, C_out, 0, res, mode, splits=splits)

        t_lib = _graph_microseconds(library)
        t_pw = _graph_microseconds(pointwise)
        sweep = [_graph_microseconds(lambda s=s: pointwise(s)) for s in (1, 2, 4, 8, 16)]
        total_lib += n * t_lib
        total_pw += n * t_pw
        lines.append(f"{C_in:5d} -> {C_out:4d} @ {H:3d}x{W:3d} x{n} ep{ep} | {t_lib:7.2f} | {t_pw:7.2f} | " + " ".join(f"{t:6.2f}" for t in sweep))
        print(lines[-1], flush=True)
    lines.append(f"# per frame: library {total_lib:.1f} us, pointwise kernel {total_pw:.1f} us")
    print(lines[-1])
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
