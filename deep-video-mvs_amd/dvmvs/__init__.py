"""MI355X-native plane-sweep depth engine with the module surface of ardaduz/deep-video-mvs (``dvmvs``).

The geometric hot path (cost volume, hidden-state warp, depth re-projection, ConvLSTM gate fusion) runs as
hand-written gfx950 HIP kernels behind a C ABI (``include/dvmvs_hip.h``); dense convolutions run on MIOpen.
"""
__version__ = "0.1.0"

import os as _os

# MIOpen's fp32 `igemm_fwd_gtcx35_nhwc_*_gkgs` forward kernels split the reduction over workgroups and accumulate with float ATOMICS: their result
# differs from run to run (tools/conv_determinism_probe.py), which a depth pipeline whose recurrent state passes through a discrete z-buffer cannot
# tolerate.  MIOpen reads its MIOPEN_DEBUG_* switches ONCE, at its first convolution, so the switch is set here, when the package is imported --
# before any of its code can have run a convolution (round 5 set it when the first DepthEngine was built: too late whenever something else in the
# process had already convolved; round 6 found the same through `import dvmvs.engine` in the middle of a test session).  A caller that
# exported its own value keeps it; DVMVS_KEEP_MIOPEN_ATOMIC_KERNELS=1 leaves MIOpen's own choice (training speed experiments).
DETERMINISTIC_MIOPEN = ("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC", "0")
MIOPEN_SWITCH_WAS_PRESET = DETERMINISTIC_MIOPEN[0] in _os.environ
if _os.environ.get("DVMVS_KEEP_MIOPEN_ATOMIC_KERNELS", "0") != "1":
    _os.environ.setdefault(*DETERMINISTIC_MIOPEN)
