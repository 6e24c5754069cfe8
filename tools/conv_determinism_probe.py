"""Probe (MI355X): which convolution problems of a fusionnet frame does MIOpen solve with a kernel whose result varies from run to
run (split-K accumulated with atomics: the `..._gkgs` implicit-GEMM kernels)?  Every torch.nn.functional.conv2d call of one engine
frame is repeated on the same input and compared bit for bit; also the engine's own launches (depthwise, epilogues, sweep, gates).

    python tools/conv_determinism_probe.py
"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    import synthetic as syn
    from dvmvs.engine import DepthEngine
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    torch.backends.cudnn.benchmark = "--benchmark" in sys.argv
    dev = torch.device("cuda:0")
    mods = syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder))
    engine = DepthEngine(*mods, device=dev, use_graphs=False, conv_plans=False)
    real = F.conv2d
    seen = {}

    def twice(x, w, *a, **k):
        y = real(x, w, *a, **k)
        worst = 0.0
        for _ in range(3):
            worst = max(worst, float((real(x, w, *a, **k) - y).abs().max()))
        key = (tuple(x.shape), tuple(w.shape), tuple(x.stride()), str(a) + str(sorted(k.items())))
        seen[key] = max(seen.get(key, 0.0), worst)
        return y

    F.conv2d = twice
    torch.nn.functional.conv2d = twice
    fullK = syn.full_K()
    with torch.no_grad():
        for n, (r, ms) in enumerate(syn.E2E_FRAMES[:2]):
            engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK,
                        frame_id=r, measurement_ids=list(ms))
    F.conv2d = real
    bad = {k: v for k, v in seen.items() if v != 0.0}
    print(f"{len(seen)} distinct convolution problems, {len(bad)} with run-to-run differences (cudnn.benchmark={torch.backends.cudnn.benchmark}):")
    for (xs, ws, st, extra), v in sorted(bad.items(), key=lambda kv: -kv[1]):
        print(f"  x {xs} strides {st} w {ws} {extra}: max |run - run| {v:.3e}")
    # whole frames, same engine, repeated: everything else (HIP kernels, epilogues) on top
    outs = []
    with torch.no_grad():
        for rep in range(3):
            engine.reset()
            for n, (r, ms) in enumerate(syn.E2E_FRAMES[:2]):
                d = engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK,
                                frame_id=r, measurement_ids=list(ms)).clone()
            outs.append(d)
    print("frame 1 depth, repeated in-process: max |a - b| =", float((outs[0] - outs[1]).abs().max()), float((outs[0] - outs[2]).abs().max()))


if __name__ == "__main__":
    main()
