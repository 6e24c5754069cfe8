"""Where the drop-in route's time goes (bench.py: ModuleSurfaceLoop -- the reference's per-frame loop on the dvmvs module surface, eager): per stage
host + device time (a device synchronisation after every stage), the number of launches per frame (torch profiler), with MIOpen's solver search
(torch.backends.cudnn.benchmark) off and on.      python tools/module_surface_probe.py"""
import os
import sys
import time

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (ROOT, os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from dvmvs import pose_algebra
    pose_algebra.MODE = "auto"
    dev = torch.device("cuda:0")
    M = 2
    images, seq, full_K = bench.synthetic_sequence(0, 32, 40, M)
    images = [im.to(dev) for im in images]
    seq = [(r.to(dev), [p.to(dev) for p in ms]) for r, ms in seq]
    full_K = full_K.to(dev)
    for benchmark in (False, True):
        torch.backends.cudnn.benchmark = benchmark
        loop = bench.ModuleSurfaceLoop(bench.build_modules(), dev)
        with torch.no_grad():
            for k in range(M, M + 6):
                loop.step(images[k % 32], seq[k][0], [images[(k - 1 - i) % 32] for i in range(M)], seq[k][1], full_K)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(M + 6, M + 26):
                loop.step(images[k % 32], seq[k][0], [images[(k - 1 - i) % 32] for i in range(M)], seq[k][1], full_K)
            host = time.perf_counter() - t0
            torch.cuda.synchronize()
            total = time.perf_counter() - t0
            print(f"cudnn.benchmark={benchmark}: {20 / total:.1f} frames/s; host {host / 20 * 1e3:.2f} ms per frame, wall {total / 20 * 1e3:.2f} ms per frame")
            # stages, synchronised
            k = M + 30
            stages = {}

            def timed(name, fn):
                torch.cuda.synchronize()
                t = time.perf_counter()
                out = fn()
                h = time.perf_counter() - t
                torch.cuda.synchronize()
                stages[name] = (h * 1e3, (time.perf_counter() - t) * 1e3)
                return out

            fe, fs, enc, lstm, dec = loop.fe, loop.fs, loop.enc, loop.lstm, loop.dec
            taps = timed("feature extractor (MnasNet), 1 image", lambda: fe(images[k % 32]))
            feats = timed("feature shrinker (FPN), 1 image", lambda: fs(*taps))
            half_K = full_K.clone()
            half_K[:, 0:2, :] /= 2.0
            cv = timed("cost_volume_fusion (M = 2)", lambda: loop.utils.cost_volume_fusion(feats[0], [feats[0], feats[0]], seq[k][0], seq[k][1], half_K, loop.warp_grid,
                                                                                         0.25, 20.0, 64, dev, True))
            enc_out = timed("cost volume encoder", lambda: enc(features_half=feats[0], features_quarter=feats[1], features_one_eight=feats[2],
                                                               features_one_sixteen=feats[3], cost_volume=cv))
            timed("re-projection + interpolate", lambda: torch.nn.functional.interpolate(loop.utils.get_non_differentiable_rectangle_depth_estimation(
                seq[k][0], loop.previous_pose, loop.previous_depth, full_K, half_K, loop.W, loop.H), scale_factor=1.0 / 16.0, mode="nearest"))
            est = torch.rand(1, 1, 8, 10, device=dev) + 0.5
            lstm_K = full_K.clone()
            lstm_K[:, 0:2, :] /= 32.0
            state = timed("LSTM fusion", lambda: lstm(current_encoding=enc_out[4], current_state=loop.lstm_state, previous_pose=loop.previous_pose, current_pose=seq[k][0],
                                                      estimated_current_depth=est, camera_matrix=lstm_K))
            timed("decoder", lambda: dec(images[k % 32], *enc_out[:4], state[0]))
            for name, (h, w) in stages.items():
                print(f"    {name:42s} host {h:6.2f} ms, with device {w:6.2f} ms")
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                loop.step(images[k % 32], seq[k][0], [images[(k - 1 - i) % 32] for i in range(M)], seq[k][1], full_K)
                torch.cuda.synchronize()
            ev = prof.key_averages()
            kernels = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
            n_launch = sum(e.count for e in kernels)
            print(f"    launches per frame: {n_launch}; device kernel time {sum(e.device_time_total for e in kernels) / 1e3:.2f} ms")
            top = sorted(ev, key=lambda e: -e.self_cpu_time_total)[:12]
            for e in top:
                print(f"      cpu {e.self_cpu_time_total / 1e3:7.2f} ms  x{e.count:4d}  {e.key[:70]}")


if __name__ == "__main__":
    main()
