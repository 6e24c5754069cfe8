"""Probe (MI355X): the ConvLSTM convolution of a fusionnet frame -- 1024 -> 2048 channels, 3x3, on the 8x10 bottleneck map, batch 1
(dvmvs/convlstm.py:43-44) -- as MIOpen convolutions and as GEMMs over an im2col operand (rocBLAS / hipBLASLt through torch.mm /
torch.bmm, with explicit split-K as a batched GEMM).  75.5 MB of weights: 9.4 us at 8 TB/s; 3.0 GFLOP: 19 us at the fp32 MFMA peak.

    python tools/lstm_conv_probe.py
"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "deep-video-mvs_amd"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from dvmvs.engine import _graph_microseconds  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 1024, 8, 10, generator=g).to(dev)
    w = (torch.randn(2048, 1024, 3, 3, generator=g) / 96).to(dev)
    ref = F.conv2d(x.double(), w.double(), padding=1).float()
    rows = []

    def report(name, fn, result):
        err = float((result().reshape(ref.shape) - ref).abs().max())
        rows.append((name, _graph_microseconds(fn, reps=10, rounds=5), err))
        print(f"{rows[-1][0]:58s} {rows[-1][1]:8.2f} us   max|err vs fp64| {rows[-1][2]:.2e}", flush=True)

    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        report(f"MIOpen conv NCHW (benchmark={bench})", lambda: F.conv2d(x, w, padding=1), lambda: F.conv2d(x, w, padding=1))
        xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        report(f"MIOpen conv NHWC (benchmark={bench})", lambda: F.conv2d(xc, wc, padding=1), lambda: F.conv2d(xc, wc, padding=1).contiguous())

    col = F.unfold(x, 3, padding=1)[0].contiguous()            # [9216, 80], row = c * 9 + tap
    w2 = w.view(2048, 9216)
    report("unfold (im2col through ATen)", lambda: F.unfold(x, 3, padding=1), lambda: w2 @ F.unfold(x, 3, padding=1)[0])
    for lib in ("default", "hipblaslt", "hipblas"):
        try:
            if lib != "default":
                torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(f"{lib}: not selectable ({e})")
            continue
        out = torch.empty(2048, 80, device=dev)
        report(f"mm [2048x9216]x[9216x80] ({lib})", lambda: torch.mm(w2, col, out=out), lambda: torch.mm(w2, col))
        colT, w2T = col.t().contiguous(), w2.t().contiguous()
        out2 = torch.empty(80, 2048, device=dev)
        report(f"mm [80x9216]x[9216x2048] ({lib})", lambda: torch.mm(colT, w2T, out=out2), lambda: torch.mm(colT, w2T).t())
        for S in (2, 4, 8, 16, 32):
            ks = 9216 // S
            wk = w2.view(2048, S, ks).permute(1, 0, 2).contiguous()       # [S, 2048, ks]
            ck = col.view(S, ks, 80)
            outk = torch.empty(S, 2048, 80, device=dev)
            report(f"bmm split-K {S:2d}: [{S}x2048x{ks}]x[{S}x{ks}x80] ({lib})", lambda: torch.bmm(wk, ck, out=outk), lambda: torch.bmm(wk, ck).sum(0))
            wkT = w2T.view(S, ks, 2048)
            ckT = colT.view(80, S, ks).permute(1, 0, 2).contiguous()      # [S, 80, ks]
            outkT = torch.empty(S, 80, 2048, device=dev)
            report(f"bmm split-K {S:2d}: [{S}x80x{ks}]x[{S}x{ks}x2048] ({lib})", lambda: torch.bmm(ckT, wkT, out=outkT), lambda: torch.bmm(ckT, wkT).sum(0).t())


if __name__ == "__main__":
    main()
