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

    # the product path: csrc/bottleneck_conv.hip (weight-streaming fp32 MFMA, deterministic split-K), LSTM layer and the other layers
    from dvmvs.hip import ops as _ops
    for (ci, co, h, wd, st) in ((1024, 2048, 8, 10, 1), (512, 512, 8, 10, 1), (256, 512, 16, 20, 2), (512, 256, 16, 20, 1), (256, 256, 16, 20, 1)):
        xx = torch.randn(1, ci, h, wd, generator=g).to(dev)
        ww = (torch.randn(co, ci, 3, 3, generator=g) / 96).to(dev)
        packed = _ops.bottleneck_conv_pack(ww)
        S = _ops.bottleneck_conv_splits(1, co, ci, h, wd, st)
        parts = torch.empty(S * co * (h // st) * (wd // st), device=dev)
        dst = torch.empty(1, co, h // st, wd // st, device=dev)
        bias = torch.zeros(co, device=dev)
        exact = F.conv2d(xx.double(), ww.double(), padding=1, stride=st).float()
        _ops.bottleneck_conv_into(xx, packed, co, st, parts)
        _ops.partial_sums_bias_act_into(parts, S, dst, bias, 0, tuple(dst.shape))
        err = float((dst - exact).abs().max())
        t_conv = _graph_microseconds(lambda: _ops.bottleneck_conv_into(xx, packed, co, st, parts), reps=10, rounds=5)
        t_both = _graph_microseconds(lambda: (_ops.bottleneck_conv_into(xx, packed, co, st, parts),
                                              _ops.partial_sums_bias_act_into(parts, S, dst, bias, 1, tuple(dst.shape))), reps=10, rounds=5)
        t_mi = _graph_microseconds(lambda: F.conv2d(xx, ww, padding=1, stride=st), reps=10, rounds=5)
        flop = 2.0 * co * ci * 9 * (h // st) * (wd // st)
        print(f"bottleneck kernel {ci:4d}->{co:4d} {h}x{wd} s{st}: {S:2d} splits  conv {t_conv:7.2f} us ({flop / t_conv / 1e6:6.1f} TFLOP/s, weights "
              f"{co * ci * 36 / t_conv / 1e3:6.0f} GB/s)  + epilogue {t_both:7.2f} us   MIOpen conv alone {t_mi:7.2f} us   max|err vs fp64| {err:.2e}", flush=True)
    cc_parts = torch.randn(16 * 2048 * 80, device=dev)
    c_state, h_state = torch.randn(1, 512, 8, 10, device=dev), torch.zeros(1, 512, 8, 10, device=dev)
    print(f"lstm gates on 16 partial sums: {_graph_microseconds(lambda: _ops.lstm_gates_partials_into(cc_parts, 16, c_state, h_state), reps=10, rounds=5):.2f} us; "
          f"on one: {_graph_microseconds(lambda: _ops.lstm_gates_into(cc_parts[:2048 * 80].view(1, 2048, 8, 10), c_state, h_state), reps=10, rounds=5):.2f} us", flush=True)
    if "--kernel-only" in sys.argv:
        return

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
