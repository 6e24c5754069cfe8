"""Probe (MI355X): how long are the two halves of a frame -- reference-image feature extraction (MnasNet + FPN) and everything behind
it (sweep, encoder, re-projection, ConvLSTM, decoder) -- as separate hipGraphs, and how long do they take when the NEXT frame's feature
extraction runs on a second stream concurrently with the current frame's back half (batch 1 leaves most of the chip idle per kernel)?

    python tools/frame_stage_probe.py
"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def timed(graph, reps=30):
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    import synthetic as syn
    from dvmvs.engine import DepthEngine
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    dev = torch.device("cuda:0")
    mods = syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder))
    engine = DepthEngine(*mods, device=dev, use_graphs=False)
    fullK = syn.full_K()
    with torch.no_grad():
        for n, (r, ms) in enumerate(list(syn.E2E_FRAMES) * 2):
            engine.step(syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK,
                        frame_id=r, measurement_ids=list(ms))
        torch.cuda.synchronize()
        d = engine._direct_buffers
        cur, alt = d["sets"][0], d["sets"][1]
        alt["full_in"][:, 33:36].copy_(syn.e2e_image(12).to(dev))
        key = (2, True, 2)

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            return g

        side = torch.cuda.Stream()

        def both():
            main_stream = torch.cuda.current_stream()
            side.wait_stream(main_stream)
            with torch.cuda.stream(side):
                engine._reference_features_direct(alt)
            engine._after_features_direct(*key, cur)
            main_stream.wait_stream(side)

        g_feat = capture(lambda: engine._reference_features_direct(alt))
        g_rest = capture(lambda: engine._after_features_direct(*key, cur))
        g_whole = capture(lambda: engine._frame_body_direct(*key))
        g_both = capture(both)
        t_feat, t_rest, t_whole, t_both = timed(g_feat), timed(g_rest), timed(g_whole), timed(g_both)
        print(f"feature extraction (MnasNet + FPN): {t_feat:8.1f} us")
        print(f"sweep .. decoder:                   {t_rest:8.1f} us")
        print(f"whole frame, one stream:            {t_whole:8.1f} us   (sum of the halves {t_feat + t_rest:.1f})")
        print(f"next frame's features on a 2nd stream, concurrently with sweep .. decoder: {t_both:8.1f} us   "
              f"({100 * (1 - t_both / t_whole):.1f} % less than the one-stream frame)")
        # sanity: the concurrent features equal the serial ones
        g_feat.replay()
        torch.cuda.synchronize()
        serial = [c[:, :32].clone() for c in alt["enc_cat"]]
        for c in alt["enc_cat"]:
            c[:, :32].zero_()
        g_both.replay()
        torch.cuda.synchronize()
        print("concurrent features bit-identical to serial:", all(torch.equal(a[:, :32], b) for a, b in zip(alt["enc_cat"], serial)))


if __name__ == "__main__":
    main()
