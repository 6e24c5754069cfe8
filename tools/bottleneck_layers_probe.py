#!/usr/bin/env python3
"""Per-layer timing of the bottleneck kernel (csrc/bottleneck_conv.hip) on the 3x3 layers of a 320x256 frame it takes, with the row-window staging
of round 6 and -- tools-only build, DVMVS_BC_WHOLE_MAP=1 -- the whole-map staging of rounds 4-5, + bit-identity of the two.

    DVMVS_HIP_LIB=deep-video-mvs_amd/lib/libdvmvs_hip_tuning.so python tools/bottleneck_layers_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deep-video-mvs_amd"))
import torch  # noqa: E402

LAYERS = [(128, 256, 32, 40, 2), (256, 256, 16, 20, 1), (288, 256, 16, 20, 1), (512, 256, 16, 20, 1), (256, 512, 16, 20, 2), (512, 512, 8, 10, 1),
          (1024, 2048, 8, 10, 1)]


def main():
    from dvmvs.engine import _graph_microseconds
    from dvmvs.hip import ops
    dev = torch.device("cuda:0")
    for C_in, C_out, H, W, stride in LAYERS:
        x = torch.randn(1, C_in, H, W, device=dev)
        w = torch.randn(C_out, C_in, 3, 3, device=dev) / (3 * C_in ** 0.5)
        packed = ops.bottleneck_conv_pack(w)
        S = ops.bottleneck_conv_splits(1, C_out, C_in, H, W, stride)
        P = (H // stride) * (W // stride)
        out = {}
        for mode in ("window", "whole"):
            if mode == "whole":
                os.environ["DVMVS_BC_WHOLE_MAP"] = "1"
            else:
                os.environ.pop("DVMVS_BC_WHOLE_MAP", None)
            partials = torch.zeros(S * C_out * P, device=dev)
            t = _graph_microseconds(lambda: ops.bottleneck_conv_into(x, packed, C_out, stride, partials), reps=10, rounds=5)
            out[mode] = (t, partials.clone())
        same = torch.equal(out["window"][1], out["whole"][1])
        print(f"{C_in:5d} -> {C_out:4d} @ {H}x{W} stride {stride}: {S:2d} splits | this build {out['window'][0]:6.2f} us | DVMVS_BC_WHOLE_MAP=1 {out['whole'][0]:6.2f} us | "
              f"bit-identical {same}", flush=True)


if __name__ == "__main__":
    main()
