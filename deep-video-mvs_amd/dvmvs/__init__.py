"""MI355X-native plane-sweep depth engine with the module surface of ardaduz/deep-video-mvs (``dvmvs``).

The geometric hot path (cost volume, hidden-state warp, depth re-projection, ConvLSTM gate fusion) runs as
hand-written gfx950 HIP kernels behind a C ABI (``include/dvmvs_hip.h``); dense convolutions run on MIOpen.
"""
__version__ = "0.1.0"
