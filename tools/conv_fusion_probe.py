"""Probe: does aten::miopen_convolution_relu fuse bias+ReLU into the convolution on this stack, and what do the
separate epilogue kernels cost?  (GPU box only.)"""
import torch, time
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
shapes = [(32, 32, 128, 160, 3), (96, 32, 128, 160, 5), (64, 64, 64, 80, 3), (256, 256, 16, 20, 3), (36, 32, 256, 320, 5), (1024, 2048, 8, 10, 3)]
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for cin, cout, H, W, k in shapes:
    x = torch.randn(1, cin, H, W, device=dev); w = torch.randn(cout, cin, k, k, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    p = k // 2
    with torch.no_grad():
        t_plain = timeit(lambda: torch.nn.functional.conv2d(x, w, None, 1, p))
        t_sep = timeit(lambda: torch.relu(torch.nn.functional.conv2d(x, w, b, 1, p)))
        t_fused = timeit(lambda: torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [p, p], [1, 1], 1))
        a = torch.relu(torch.nn.functional.conv2d(x, w, b, 1, p)); f = torch.ops.aten.miopen_convolution_relu(x, w, b, [1, 1], [p, p], [1, 1], 1)
        xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        t_cl = timeit(lambda: torch.relu(torch.nn.functional.conv2d(xc, wc, b, 1, p)))
    gflop = 2 * cin * cout * k * k * H * W / 1e9
    print(f"{cin:4d}->{cout:4d} {H}x{W} k{k}: conv {t_plain:7.1f} us ({gflop / t_plain * 1e3:6.1f} TF) | conv+bias+relu separate {t_sep:7.1f} | miopen_convolution_relu {t_fused:7.1f} "
          f"| channels_last {t_cl:7.1f} | maxdiff {float((a - f).abs().max()):.1e}")
x = torch.randn(1, 32, 128, 160, device=dev)
print("upsample 2x bilinear ac 32x128x160:", timeit(lambda: torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)))
x = torch.randn(1, 512, 8, 10, device=dev)
print("upsample 2x 512x8x10:", timeit(lambda: torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)))
x = torch.randn(1, 64, 128, 160, device=dev)
print("relu 64x128x160:", timeit(lambda: torch.relu(x)), " add bias:", timeit(lambda: x + torch.ones(1, 64, 1, 1, device=dev)))
