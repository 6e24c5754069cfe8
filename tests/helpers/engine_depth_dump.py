"""Child process of tests/test_determinism_gpu.py: builds the frame engine from the seeded weights in a FRESH process, runs the 3
golden frames twice (the second sweep replays the captured hipGraphs) and writes every depth map, the final hidden state and
the convolution-plan decisions to an .npz.  Two such processes must produce the same bits.

    python tests/helpers/engine_depth_dump.py OUT.npz [--benchmark]       (--benchmark: torch.backends.cudnn.benchmark = True)
"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    out = sys.argv[1]
    torch.backends.cudnn.benchmark = "--benchmark" in sys.argv
    import synthetic as syn
    from dvmvs.engine import DepthEngine
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    dev = torch.device("cuda:0")
    mods = syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder))
    engine = DepthEngine(*mods, device=dev)        # the benchmarked configuration: BN folded, feature cache, hipGraph replay, MIOpen plans
    fullK = syn.full_K()
    arrays = {}
    with torch.no_grad():
        for sweep in range(2):
            engine.reset()
            for n, (r, ms) in enumerate(syn.E2E_FRAMES):
                depth = engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms],
                                    fullK, frame_id=r, measurement_ids=list(ms))
                arrays[f"sweep{sweep}_frame{n}_depth"] = depth.cpu().numpy()
        arrays["h"] = engine._static["h"].cpu().numpy()
    report = sorted((str(shape), str(wshape), bool(use)) for shape, wshape, use, *_ in engine.conv_plan_report())
    arrays["plans"] = np.array([f"{a} {b} {c}" for a, b, c in report])
    np.savez(out, **arrays)


if __name__ == "__main__":
    main()
