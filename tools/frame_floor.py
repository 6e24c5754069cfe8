"""Two measurements that bound what launch fusion can buy the frame (GPU box; information for the engine, not the product):

1. the floor of one kernel launch inside a replayed hipGraph (251 launches per frame today), with a trivial kernel;
2. a 3x3 convolution + bias + ReLU of the cost-volume encoder / decoder sizes as (a) MIOpen convolution followed by the
   in-place bias_act kernel (what the engine does) and (b) aten::miopen_convolution_relu (MIOpen's fused convolution-bias-
   activation), timed the same way, with the number of kernels each one launches.

    python tools/frame_floor.py
"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "deep-video-mvs_amd"))

import torch  # noqa: E402

from dvmvs.hip import ops as hip_ops  # noqa: E402


def time_graph(fn, reps):
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


def kernel_count(fn):
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    return len(names), sorted(set(n[:60] for n in names))


def main():
    dev = torch.device("cuda:0")
    tiny = torch.zeros(64, device=dev)
    bias1 = torch.zeros(1, device=dev)
    empty = torch.empty(0, device=dev)
    x4 = tiny.view(1, 1, 8, 8)
    print("launch floor: %.2f us per trivial kernel in a replayed graph (251 launches per frame today)"
          % time_graph(lambda: hip_ops.bias_act_(x4, bias1, 0, empty, 0), 251))
    for (cin, cout, h, w) in [(512, 512, 8, 10), (256, 256, 16, 20), (128, 128, 32, 40), (64, 64, 64, 80), (96, 64, 128, 160)]:
        x = torch.randn(1, cin, h, w, device=dev)
        wgt = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
        b = torch.randn(cout, device=dev)

        def separate():
            y = torch.nn.functional.conv2d(x, wgt, None, 1, 1)
            hip_ops.bias_act_(y, b, 1, empty, 0)
            return y

        def fused():
            return torch.ops.aten.miopen_convolution_relu(x, wgt, b, [1, 1], [1, 1], [1, 1], 1)

        try:
            diff = (separate() - fused()).abs().max().item()
            ns, names_s = kernel_count(separate)
            nf, names_f = kernel_count(fused)
            print("conv3x3 %4d->%4d @ %3dx%3d: conv + bias_act %6.1f us (%d kernels) | miopen_convolution_relu %6.1f us (%d kernels: %s)  max|diff| %.1e"
                  % (cin, cout, h, w, time_graph(separate, 20), ns, time_graph(fused, 20), nf, "; ".join(names_f), diff))
        except Exception as e:   # noqa: BLE001
            print("conv3x3 %d->%d @ %dx%d: %s" % (cin, cout, h, w, str(e)[:300]))


if __name__ == "__main__":
    main()
