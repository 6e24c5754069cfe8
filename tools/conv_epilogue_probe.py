"""Probe (GPU box): can the frame's convolution epilogues ride inside MIOpen's own kernels?

The engine runs every dense convolution as MIOpen convolution + one dvmvs_bias_act_fwd launch (60 such launches per frame,
profiles/r03_bench_timed_region.csv).  ATen exposes MIOpen's fused convolution-bias-ReLU plan as
aten::miopen_convolution_relu; whether it is ONE kernel, which solver it picks and whether its output equals the two-launch
path depends on the problem.  This records every distinct dense ReLU convolution of one fusionnet frame (shapes from a real
engine step) and reports for each: launches and time of both forms (hipGraph of REPS calls, HIP events) and max |difference|.

    python tools/conv_epilogue_probe.py [--reps 20]
"""
import argparse
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (ROOT, os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402
from dvmvs import engine as eng  # noqa: E402
from dvmvs.hip import ops  # noqa: E402


def graph_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / reps)
    return best


def kernel_names(fn):
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return [e.key[:60] for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA or "cuda" in str(e.device_type).lower()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    modules = bench.build_modules()
    engine = eng.DepthEngine(*modules, device=dev, use_graphs=False)
    seen = {}
    original = eng.FusedConv2d.forward

    def spy(self, x, residual=None, residual_mode=0, out=None, activation=None, p0=0.0, p1=0.0, raw=False):
        act = self.activation if activation is None else activation
        if not self.depthwise and not self.defer_epilogue and not raw and act == ops.ACTIVATIONS["relu"] and residual is None:
            key = (tuple(x.shape), tuple(self.weight.shape), tuple(self.stride), tuple(self.padding), tuple(self.dilation), self.groups,
                   out is not None)
            seen.setdefault(key, [0, self])[0] += 1
        return original(self, x, residual, residual_mode, out, activation, p0, p1, raw)

    eng.FusedConv2d.forward = spy
    M = 2
    images, seq, full_K = bench.synthetic_sequence(0, 8, M + 3, M)
    images = [im.to(dev) for im in images]
    with torch.no_grad():
        for k in range(M):
            engine._half_features(k, images[k])
        engine.step(images[M], seq[M][0], None, seq[M][1], full_K, frame_id=M, measurement_ids=[M - 1 - i for i in range(M)])
        seen.clear()
        engine.step(images[M + 1], seq[M + 1][0], None, seq[M + 1][1], full_K, frame_id=M + 1, measurement_ids=[M - i for i in range(M)])
    eng.FusedConv2d.forward = original
    print(f"{sum(v[0] for v in seen.values())} dense ReLU convolution epilogues per frame in {len(seen)} distinct problems")
    total_two, total_fused = 0.0, 0.0
    with torch.no_grad():
        for key, (count, mod) in sorted(seen.items(), key=lambda kv: -kv[1][0]):
            xs, ws, stride, padding, dilation, groups, sliced = key
            x = torch.randn(*xs, device=dev)
            bias = mod.bias if mod.bias is not None else torch.zeros(ws[0], device=dev)

            def two():
                y = torch.nn.functional.conv2d(x, mod.weight, None, stride, padding, dilation, groups)
                return ops.bias_act_into(y, y, bias, ops.ACTIVATIONS["relu"], None, 0, 0.0, 0.0)

            def fused():
                return torch.ops.aten.miopen_convolution_relu(x, mod.weight, bias, list(stride), list(padding), list(dilation), groups)

            try:
                diff = (two() - fused()).abs().max().item()
                t2, tf = graph_time(two, args.reps), graph_time(fused, args.reps)
                names = kernel_names(fused)
            except Exception as e:  # noqa: BLE001 -- a probe: report and go on
                print(f"x{count} in {xs} w {ws} s{stride} p{padding} d{dilation}: fused form failed: {type(e).__name__}: {str(e)[:120]}")
                continue
            total_two += count * t2
            total_fused += count * tf
            print(f"x{count} in {xs} w {ws} s{stride[0]} p{padding[0]} d{dilation[0]} slice={int(sliced)}: conv+epilogue {t2:7.2f} us, "
                  f"miopen_convolution_relu {tf:7.2f} us, max|diff| {diff:.2e}, fused kernels: {names}", flush=True)
    print(f"per frame: {total_two:.1f} us as convolution + epilogue, {total_fused:.1f} us as miopen_convolution_relu")


if __name__ == "__main__":
    main()
