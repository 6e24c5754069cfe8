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
        # the other buffer set gets a copy of this frame's sweep parameters and measurement features (timing only)
        params, off = engine._static["params"], engine._param_offsets
        for name in ("Hm", "kt", "sweep_items"):
            (o0, n), (o1, _) = off[name], off[name + "1"]
            params[o1:o1 + n].copy_(params[o0:o0 + n])
        for a, b in zip(alt["meas_feat"], cur["meas_feat"]):
            a.copy_(b)

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            return g

        side = torch.cuda.Stream()

        def overlapped(ahead, own):
            def body():
                main_stream = torch.cuda.current_stream()
                side.wait_stream(main_stream)
                with torch.cuda.stream(side):
                    ahead()
                own()
                main_stream.wait_stream(side)
            return body

        feat = lambda b: (lambda: engine._reference_features_direct(b))
        enc = lambda b: (lambda: engine._sweep_encoder_direct(b, 2, 2))
        dec = lambda b: (lambda: engine._lstm_decoder_direct(b, True))
        both = lambda f, g: (lambda: (f(), g()))
        rows = [
            ("A  feature extraction (MnasNet + FPN)", feat(alt)),
            ("B  plane sweep + encoder", enc(alt)),
            ("C  re-projection + ConvLSTM + decoder", dec(cur)),
            ("A + B + C on one stream (the frame of rounds 1-3)", both(both(feat(cur), enc(cur)), dec(cur))),
            ("look-ahead 1:  A(next) on a 2nd stream  ||  B + C", overlapped(feat(alt), both(enc(cur), dec(cur)))),
            ("look-ahead 2:  A(next) + B(next) on a 2nd stream  ||  C", overlapped(both(feat(alt), enc(alt)), dec(cur))),
        ]
        times = {}
        for name, fn in rows:
            times[name] = timed(capture(fn))
            print(f"{name:62s} {times[name]:8.1f} us")
        whole = times[rows[3][0]]
        for name in (rows[4][0], rows[5][0]):
            print(f"{name}: {100 * (1 - times[name] / whole):.1f} % less than the one-stream frame")


if __name__ == "__main__":
    main()
