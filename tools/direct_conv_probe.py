"""Probe (MI355X): the direct convolution kernel (csrc/direct_conv.hip) on every dense layer of a fusionnet frame it takes, and the
depth-head kernel on the five one-output-channel layers -- max |error| against an fp64 convolution and time against MIOpen's
immediate-mode choice for the same layer (hipGraph of 10 calls, best of 3).

    python tools/direct_conv_probe.py [--batch 1]
"""
import argparse
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "deep-video-mvs_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from conv_layer_probe import frame_layers  # noqa: E402
from dvmvs.engine import _graph_microseconds  # noqa: E402
from dvmvs.hip import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    g = torch.Generator().manual_seed(3)
    tot_direct = tot_miopen = 0.0
    for name, xi, xo, k, s, p, groups in frame_layers():
        if groups != 1 or k == 1:
            continue
        B = args.batch
        C_in, H, W = xi[1:]
        C_out = xo[1]
        x = torch.randn(B, C_in, H, W, generator=g).to(dev)
        w = (torch.randn(C_out, C_in, k, k, generator=g) / (C_in * k * k) ** 0.5).to(dev)
        bias = torch.randn(C_out, generator=g).to(dev)
        exact = torch.relu(F.conv2d(x.double(), w.double(), bias.double(), stride=s, padding=p)).float()
        flop = 2.0 * B * xo[1] * xo[2] * xo[3] * C_in * k * k
        t_mi = _graph_microseconds(lambda: torch.relu_(F.conv2d(x, w, bias, stride=s, padding=p)), reps=10, rounds=3)
        err_mi = float((torch.relu(F.conv2d(x, w, bias, stride=s, padding=p)) - exact).abs().max())
        if C_out == 1 and k == 3 and s == 1:
            dst = torch.empty(B, 1, H, W, device=dev)
            ops.conv_head_into(x, w, bias, dst, ops.ACTIVATIONS["relu"])
            err = float((dst - exact).abs().max())
            t = _graph_microseconds(lambda: ops.conv_head_into(x, w, bias, dst, ops.ACTIVATIONS["relu"]), reps=10, rounds=3)
            kind = "head"
        else:
            n_tile = ops.direct_conv_tile(B, C_in, H, W, C_out, k, s)
            if n_tile == 0:
                print(f"{name:48s} k{k} s{s} {C_in:4d}x{H:3d}x{W:3d} -> {C_out:4d}: not taken (MIOpen {t_mi:7.2f} us)")
                continue
            packed = ops.direct_conv_pack(w, n_tile)
            dst = torch.empty(B, C_out, H // s, W // s, device=dev)
            ops.direct_conv_into(x, packed, n_tile, bias, dst, C_out, k, s, ops.ACTIVATIONS["relu"])
            err = float((dst - exact).abs().max())
            t = _graph_microseconds(lambda: ops.direct_conv_into(x, packed, n_tile, bias, dst, C_out, k, s, ops.ACTIVATIONS["relu"]), reps=10, rounds=3)
            kind = f"nt{n_tile}"
        tot_direct += t
        tot_miopen += t_mi
        print(f"{name:48s} k{k} s{s} {C_in:4d}x{H:3d}x{W:3d} -> {C_out:4d} {kind:5s} {t:7.2f} us ({flop / t / 1e6:6.1f} TFLOP/s)  err {err:.2e}   "
              f"MIOpen + relu {t_mi:7.2f} us  err {err_mi:.2e}", flush=True)
    print(f"layers taken: direct {tot_direct:.1f} us, MIOpen (+ separate ReLU launch) {tot_miopen:.1f} us")


if __name__ == "__main__":
    main()
