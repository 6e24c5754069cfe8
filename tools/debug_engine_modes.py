"""Debug: which execution option makes graph replay diverge when eager work is interleaved between frames?"""
import os, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "deep-video-mvs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import synthetic as syn
from dvmvs.engine import DepthEngine
from dvmvs.hip import ops
from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion

dev = torch.device("cuda:0")
mods = syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder))
fullK = syn.full_K().to(dev)
halfK = syn.scaled_K(fullK, 2.0)

def run(kw, interleave):
    eng = DepthEngine(*mods, device=dev, **kw)
    outs = []
    prev = None
    for sweep in range(2):
        eng.reset()
        for n, (r, ms) in enumerate(syn.E2E_FRAMES):
            d = eng.step(syn.e2e_image(r).to(dev), syn.pose(r).to(dev), [syn.e2e_image(i).to(dev) for i in ms],
                         [syn.pose(i).to(dev) for i in ms], fullK, frame_id=r, measurement_ids=list(ms))
            outs.append((d.clone(), eng._static["h"].clone()))
            if interleave == "reproject" and prev is not None:
                ops.depth_reproject_lowres(syn.pose(r).to(dev), syn.pose(9).to(dev), prev, fullK, halfK, 16)
            elif interleave == "alloc":
                junk = [torch.randn(1, 64, 128, 160, device=dev) for _ in range(8)]
                del junk
            elif interleave == "sync":
                d.cpu()
            prev = d.clone().view(1, 1, 256, 320)
    return outs

base = run(dict(fold_bn=False, cache_features=False, use_graphs=False, fuse=False), None)
for name, kw in {"graphs only": dict(fold_bn=False, cache_features=False, use_graphs=True, fuse=False),
                 "graphs+cache": dict(fold_bn=False, cache_features=True, use_graphs=True, fuse=False),
                 "graphs+fuse": dict(fold_bn=True, cache_features=False, use_graphs=True, fuse=True),
                 "all": dict(fold_bn=True, cache_features=True, use_graphs=True, fuse=True),
                 "no graphs, rest": dict(fold_bn=True, cache_features=True, use_graphs=False, fuse=True)}.items():
    for inter in (None, "reproject", "alloc", "sync"):
        outs = run(kw, inter)
        line = " ".join(f"{((d - d0).abs() / d0).mean().item():.1e}" for (d, h), (d0, h0) in zip(outs, base))
        print(f"{name:16s} interleave={str(inter):9s} depth rel err per frame (2 sweeps x 3): {line}")
