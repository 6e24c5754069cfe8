"""Per-keyframe forward orchestration for pairnet / fusionnet on one MI355X.

The reference keeps this logic inside its scripts (/root/reference/dvmvs/fusionnet/run-testing.py:151-204,
pairnet/run-testing.py:136-166); here it is a small class so that the same frame contract can be driven by the
test runner, the benchmark and the sequence-sharded multi-GPU runner:

    features(measurement frames) , features(reference) -> fused plane-sweep cost volume (HIP) -> encoder
      -> [fusionnet] previous depth splatted into the current view (HIP) -> ConvLSTM (HIP warp + MIOpen conv + HIP gates)
      -> decoder -> depth;   state = (h, c), previous depth, previous pose;   "TRACKING LOST" -> reset()

MI355X-specific execution choices (none of them changes results beyond fp32 round-off):
* **feature cache** -- a keyframe's half-resolution features are computed once and reused when the frame later
  serves as a measurement frame (the reference recomputes them, run-testing.py:153-156); eval-mode only;
* **BatchNorm folding** -- conv + eval-mode BN are folded into one biased convolution before inference;
* **epilogue fusion** -- bias add + ReLU / sigmoid after every MIOpen convolution run as one in-place HIP kernel, and
  the 2x bilinear up-sampling of the decoder is a HIP kernel (ATen's takes ~80 us on the 512 x 8 x 10 bottleneck map);
* **no concatenation or state copies** (one sequence per engine, the headline configuration) -- every producer writes straight
  into the buffer its consumer reads: FPN outputs, the cost volume and encoder outputs into channel slices of the encoder's
  concatenation buffers, skip connections and up-sampled maps into the decoder's, the warped hidden state into the ConvLSTM's
  input, the gates into the state buffers in place, the last convolution's epilogue (sigmoid -> depth) into the depth buffer that
  is also next frame's "previous depth"; the FPN's unused 1/32 output is not computed.  ~45 launches per frame less;
* **hipGraph replay** -- after a warm-up frame (MIOpen solver search) the whole frame is captured once per
  (number of measurement frames, has-previous-state) and replayed; the HIP ops are capture-safe (no host sync,
  no allocation inside the C ABI), images and the frame's small matrices live in static device buffers;
* **pose algebra on the host, one upload per frame** -- poses arrive from the host (a tracker, ``poses.txt``); the sweep
  constants and the two relative poses of the frame are evaluated there with the reference's own fp32 expressions
  (``dvmvs.pose_algebra``, default mode "reference": the kernels then sample where the reference samples, bit for bit), packed
  with the intrinsics into one pinned staging slot and sent to the device with a single asynchronous copy ahead of the graph
  launch.  ``pose_algebra="exact"`` evaluates them on the device in fp64 inside the captured frame instead.
"""
import contextlib
import copy
import time
import os
from collections import OrderedDict

import torch
from torch import nn

from dvmvs.config import Config
from dvmvs.hip import ops as _ops
from dvmvs import pose_algebra as _pose_algebra
from dvmvs import utils as _utils

# MIOpen's fp32 `igemm_fwd_gtcx35_nhwc_*_gkgs` kernels split the reduction over workgroups and accumulate with float ATOMICS: their
# result differs from run to run (measured: tools/conv_determinism_probe.py -- four layers of a frame, among them 64 -> 64 5x5 at quarter
# resolution), which a depth engine whose recurrent state passes through a discrete z-buffer cannot tolerate (one flipped pixel and the
# runs diverge).  MIOpen reads this switch when it looks for solvers; with the family off it solves those layers with its GEMM /
# Winograd / direct kernels, which are deterministic.  (The bottleneck layers do not reach MIOpen at all: csrc/bottleneck_conv.hip.)
# Set when the PACKAGE is imported (dvmvs/__init__.py: MIOpen reads its MIOPEN_DEBUG_* variables once, at its first convolution, and caches them).
# DepthEngine.__init__ warns when the variable reads differently from what was set there.
import dvmvs as _package
_DETERMINISTIC_MIOPEN = _package.DETERMINISTIC_MIOPEN
_MIOPEN_SWITCH_WAS_PRESET = _package.MIOPEN_SWITCH_WAS_PRESET or os.environ.get("DVMVS_KEEP_MIOPEN_ATOMIC_KERNELS", "0") == "1"

_MAX_MEAS = 8            # DVMVS_MAX_MEASUREMENTS of the C ABI
# Pinned staging ring = how many frames the host may run ahead of the device (it waits for the slot's previous upload to have executed).
# Round 3: with 8 the host (0.7 ms per step) raced up to 8 graph launches ahead of the device (1.3 ms per frame) and every 10-20 frames one
# frame took 2.5-5 ms instead of 1.3 on the device -- the deeper the queue of launched graphs, the more such stalls (2 / 3 / 8 / 64 slots:
# 1 / 2 / 4 / 6 of them in 70 frames, tools/step_times_probe.py) -- hence TWO.  Round 5: the one stall that was left (a 5.6-7.4 ms device
# gap four steps after every full device synchronisation, i.e. inside the timed steps of a short benchmark run: 830-910 instead of
# 1 160-1 180 frames/s on the driver's command) is the same effect -- a frame's graph launched again while its previous launch (two frames
# back: frames alternate between two graphs) is still queued or running.  With ONE slot the host is at most one frame ahead, a graph is
# never in flight twice, and the stall is gone at the same steady-state rate (3 / 2 / 1 slots: 2 / 1 / 0 stalls in 20 steps; 100 steps:
# 1 239 frames/s): the host's 0.75 ms per step still overlap the device's 0.8 ms frame completely.
_STAGING_SLOTS = int(os.environ.get("DVMVS_STAGING_SLOTS", "1"))
# hipGraphInstantiate gives a graph with two branches an internal stream for the second one, and the hardware queue behind that stream
# alternates from one instantiation to the next (two queues, ROCm 7.2).  Consecutive frames replay the graphs of the two buffer sets: when
# their second branches sit on DIFFERENT queues every frame is 50 us slower (0.855 against 0.805 ms at look-ahead 1, whichever sweep kernel
# runs) -- and which case a run gets used to depend on how many graphs were captured between the two: even (4 sweep configurations per
# buffer set) fast, odd (5, with the MFMA sweep) slow.  So after every two-branch capture this many two-kernel filler graphs are
# instantiated (kept, never launched): all frame graphs end up on the same queue.  DESIGN.md section 5; 0 = the runtime's own order.
_GRAPH_QUEUE_FILLERS = int(os.environ.get("DVMVS_GRAPH_QUEUE_FILLERS", "1"))
# experiments: "1" = a frame's sweep runs before the side-stream fork instead of next to the side stream's kernels (see _frame_body_direct)
_FORK_AFTER = int(os.environ.get("DVMVS_FORK_AFTER", "1"))      # see _frame_body_direct (measured: -1 / 0 / 1 / 2 / 3 / 4 = 1 355 / 1 371 / 1 382 / 1 377 / 1 319 / 1 239 frames/s)
_PAIRED_UPSAMPLING = os.environ.get("DVMVS_PAIRED_UPSAMPLING", "1") != "0"      # a decoder level's two up-samplings in one launch
_UP2X_IN_CONV = os.environ.get("DVMVS_UP2X_IN_CONV", "1") != "0"      # the decoder's first up-sampling inside its convolution's staging
_UPLOAD_IN_COPY_BATCH = os.environ.get("DVMVS_UPLOAD_IN_COPY_BATCH", "1") != "0"
_AUX_STREAM = os.environ.get("DVMVS_AUX_STREAM", "0")      # "0" off, "warp" / "heads" one of the two uses, "1" both (see DepthEngine._aux_stream)
# a step's input copies as one launch (see DepthEngine._copy); "0" = one runtime copy each
_BATCH_COPIES = os.environ.get("DVMVS_BATCH_COPIES", "1") != "0"


# ----------------------------------------------------------------------------------------------------------------------
# BatchNorm folding
# ----------------------------------------------------------------------------------------------------------------------
def _fold_pair(conv, bn):
    w = conv.weight
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    folded = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, stride=conv.stride, padding=conv.padding,
                       dilation=conv.dilation, groups=conv.groups, bias=True).to(w.device)
    with torch.no_grad():
        folded.weight.copy_(w * scale.reshape(-1, 1, 1, 1))
        bias = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
        folded.bias.copy_((bias - bn.running_mean) * scale + bn.bias)
    return folded


def fold_batchnorm(module):
    """Returns a deep copy of ``module`` (eval mode) in which every Conv2d directly followed by a BatchNorm2d inside
    an ``nn.Sequential`` is replaced by one biased convolution."""
    module = copy.deepcopy(module).eval()
    for parent in module.modules():
        if not isinstance(parent, nn.Sequential):
            continue
        names = list(parent._modules.keys())
        for a, b in zip(names, names[1:]):
            conv, bn = parent._modules[a], parent._modules[b]
            if isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d):
                parent._modules[a] = _fold_pair(conv, bn)
                parent._modules[b] = nn.Identity()
    return module


# ----------------------------------------------------------------------------------------------------------------------
# epilogue fusion: conv (MIOpen, no bias) + one in-place HIP kernel for bias and activation
# ----------------------------------------------------------------------------------------------------------------------
def _graph_microseconds(fn, reps=10, rounds=3):
    """GPU time of one ``fn()``: ``reps`` of them captured into a hipGraph, best of ``rounds`` replays (HIP events)."""
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(rounds):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        graph.replay()
        end.record()
        torch.cuda.synchronize()
        best = min(best, start.elapsed_time(end) * 1e3 / reps)
    return best


_PLAN_TIMINGS = {}     # (device, input shape, weight shape, stride, padding, activation) -> FusedConv2d._time_plan's tuple


class _ShapeOn:
    """A (shape, device) pair where code written for a tensor only asks for those two."""

    def __init__(self, shape, device):
        self.shape, self.device = shape, device


class FusedConv2d(nn.Module):
    """Convolution whose bias add and activation run as ONE HIP kernel (dvmvs_bias_act_fwd) instead of two ATen launches
    after the MIOpen convolution.  Same arithmetic (add, then max / sigmoid), so results are identical.  Three launch savers:
    * ``out=``: the epilogue writes into a caller-supplied channel slice of a concatenation buffer (no torch.cat copy later);
    * ``defer_epilogue``: a ReLU convolution directly followed by a depthwise one hands over its RAW output, and the depthwise
      kernel applies bias + ReLU to its input taps on the fly (``pre_bias``): one launch per MnasNet block less;
    * depthwise layers are a single HIP launch (convolution + bias + activation)."""

    def __init__(self, conv, activation):
        super().__init__()
        self.weight = conv.weight
        self.bias = conv.bias if conv.bias is not None else None
        self.stride, self.padding, self.dilation, self.groups = conv.stride, conv.padding, conv.dilation, conv.groups
        self.activation = _ops.ACTIVATIONS[activation]
        self.register_buffer("_no_bias", torch.empty(0, device=conv.weight.device), persistent=False)
        self.defer_epilogue = False      # set by fuse_epilogues: the next (depthwise) layer applies this layer's bias + ReLU
        self.pre_bias = None             # set by fuse_epilogues on that depthwise layer: the deferred bias
        # Epilogue inside MIOpen's kernel (dvmvs_conv_bias_act_fwd): decided per input shape the first time the shape is seen outside a
        # stream capture -- taken only when its output is BIT-IDENTICAL to convolution + epilogue (the Winograd case) and it is faster;
        # {input shape: (use the plan, plan us, two-launch us, max |difference|)}
        self.plan_epilogue = False       # set by DepthEngine(conv_plans=True)
        self.plans = {}
        # 3x3 layers on the bottleneck maps (8x10, 16x20): the deterministic weight-streaming MFMA kernel instead of MIOpen's
        # atomically accumulated split-K kernels (csrc/bottleneck_conv.hip); {input shape: (splits, partial-sum buffer) or None}
        self.bottleneck = False          # set by DepthEngine(bottleneck_convs=True)
        self._bottleneck_packed = None
        self._bottleneck_buffers = {}
        # dense 3x3 / 5x5 layers on the larger maps and the one-channel depth heads: the direct MFMA convolution with bias + ReLU in
        # its store path instead of MIOpen's Winograd / GEMM kernel + epilogue launch (csrc/direct_conv.hip);
        # {input shape: output-channel tiles per wave the problem needs (0: not taken)}, {tiles: packed weights}
        self.direct_conv = False         # set by DepthEngine(direct_convs=True)
        self._direct_tiles = {}
        self._direct_packed = {}
        # 1x1 layers (MnasNet expansion / projection layers, FPN lateral layers): the MFMA GEMM with bias + ReLU + residual in its store path
        # instead of the library GEMM + epilogue launch (csrc/pointwise_conv.hip); {(input shape, activation, residual mode): taken}
        self.pointwise_conv = False      # set by DepthEngine(pointwise_convs=True)
        self._pointwise_taken = {}
        self._pointwise_packed = None

        k = conv.kernel_size
        self.depthwise = (conv.groups == conv.in_channels == conv.out_channels and conv.groups > 1 and k[0] == k[1] and k[0] in (3, 5)
                          and tuple(conv.padding) == (k[0] // 2, k[0] // 2) and tuple(conv.dilation) == (1, 1)
                          and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2))

    def forward(self, x, residual=None, residual_mode=0, out=None, activation=None, p0=0.0, p1=0.0, raw=False, out_nhwc=None):
        """``residual`` (mode 1: same shape, mode 2: half resolution, nearest-up-sampled) is added in the same epilogue; ``out``
        is the destination (default: the convolution's own output buffer); ``activation`` overrides the layer's; ``raw`` returns
        the convolution output without the epilogue (the consumer applies it: up-sampler, depthwise kernel).  ``out_nhwc``: a second
        destination that receives the same values channels-last -- in the direct kernel's own epilogue, else by one transposing launch."""
        if out_nhwc is not None:
            y = self._forward(x, residual, residual_mode, out, activation, p0, p1, raw, out_nhwc)
            if y is not None:
                return y
            y = self._forward(x, residual, residual_mode, out, activation, p0, p1, raw, None)
            _ops.nchw_to_nhwc_into(y, out_nhwc)
            return y
        return self._forward(x, residual, residual_mode, out, activation, p0, p1, raw, None)

    def _forward(self, x, residual, residual_mode, out, activation, p0, p1, raw, out_nhwc):
        """``forward``; with ``out_nhwc``: the layer through the dual-destination direct kernel, or None when it does not take the problem."""
        if out_nhwc is not None:
            if self.direct_conv and residual is None and x.is_contiguous() and not (raw or self.defer_epilogue):
                act = self.activation if activation is None else activation
                return self._direct_forward(x, out, act, p0, p1, False, out_nhwc)
            return None
        act = self.activation if activation is None else activation
        if (self.defer_epilogue or raw) and (out is not None or residual is not None or (activation is not None and not raw)):
            raise RuntimeError("this layer hands over its RAW convolution output (its consumer applies bias + activation): "
                               "out=, residual= and an activation override would be silently ignored")
        if self.depthwise and residual is None and out is None:
            pre = self.pre_bias
            return _ops.depthwise_conv(x, self.weight, self.bias if self.bias is not None else self._no_bias, self.stride[0], act,
                                       pre if pre is not None else self._no_bias, pre is not None)
        if self.pre_bias is not None:
            raise RuntimeError("a depthwise layer with a deferred input epilogue must take the depthwise kernel")
        if self.bottleneck and residual is None and not raw and not self.defer_epilogue and x.is_contiguous() and \
                act in (_ops.ACTIVATIONS["none"], _ops.ACTIVATIONS["relu"]):
            buffers = self._bottleneck_for(x)
            if buffers is not None:
                B, _, H, W = x.shape
                shape = (B, self.weight.shape[0], H // self.stride[0], W // self.stride[0])
                splits = _ops.bottleneck_conv_into(x, self._bottleneck_packed, shape[1], self.stride[0], buffers)
                dst = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
                return _ops.partial_sums_bias_act_into(buffers, splits, dst, self.bias, act, shape)
        if self.direct_conv and residual is None and x.is_contiguous():
            y = self._direct_forward(x, out, act, p0, p1, raw or self.defer_epilogue)
            if y is not None:
                return y
        if self.pointwise_conv and x.is_contiguous() and (residual is None or residual.is_contiguous()):
            y = self._pointwise_forward(x, out, act, residual, residual_mode if residual is not None else 0, raw or self.defer_epilogue)
            if y is not None:
                return y
        if (self.plan_epilogue and residual is None and not raw and not self.defer_epilogue and self._plan_eligible(act)
                and x.is_contiguous()):
            key = tuple(x.shape)
            plan = self.plans.get(key)
            if plan is None and not torch.cuda.is_current_stream_capturing():
                # one timing per problem and process: engines built later (and this engine's other layers of the same shape) reuse it
                problem = (x.device.index, key, tuple(self.weight.shape), self.stride[0], self.padding[0], act)
                if problem not in _PLAN_TIMINGS:
                    _PLAN_TIMINGS[problem] = self._time_plan(x, act)
                plan = self.plans[key] = _PLAN_TIMINGS[problem]
            if plan is not None and plan[0]:
                y = _ops.conv_bias_act_into(x, self.weight, self.bias, out, self.stride[0], self.padding[0], act)
                if y is not None:
                    return y
        y = nn.functional.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups)
        if not y.is_contiguous():
            y = y.contiguous()
        if self.defer_epilogue or raw:
            return y
        return _ops.bias_act_into(y, y if out is None else out, self.bias, act, residual, residual_mode if residual is not None else 0, p0, p1)

    def _direct_forward(self, x, out, act, p0, p1, raw, out_nhwc=None):
        """The layer through csrc/direct_conv.hip, or None when that kernel does not take the problem (then MIOpen as before).  ``raw``:
        the convolution output without bias and activation (the consumer applies them)."""
        k = self.weight.shape
        if self.depthwise or self.groups != 1 or k[2] != k[3] or tuple(self.dilation) != (1, 1) or self.stride[0] != self.stride[1] or \
                tuple(self.padding) != (k[2] // 2, k[2] // 2):
            return None
        B, _, H, W = x.shape
        if x.data_ptr() % 16 != 0 or (out is not None and out.data_ptr() % 4 != 0):
            return None      # (the kernel stages aligned float4 rows; torch's own allocations always are)
        stride = self.stride[0]
        bias = None if raw or self.bias is None else self.bias
        if raw:
            act = _ops.ACTIVATIONS["none"]
        if k[0] == 1 and k[2] == 3 and stride == 1:      # depth head
            if out_nhwc is not None:
                return None
            dst = out if out is not None else torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
            return _ops.conv_head_into(x, self.weight, bias, dst, act, p0, p1)
        if act not in (_ops.ACTIVATIONS["none"], _ops.ACTIVATIONS["relu"]):
            return None
        key = tuple(x.shape)
        tile = self._direct_tiles.get(key)
        if tile is None:
            tile = self._direct_tiles[key] = _ops.direct_conv_tile(B, k[1], H, W, k[0], k[2], stride)
        if tile == 0:
            return None
        packed = self._direct_packed.get(tile)
        if packed is None:
            if torch.cuda.is_current_stream_capturing():
                return None      # (packed at the next eager call: the first frame of every kind runs eagerly)
            packed = self._direct_packed[tile] = _ops.direct_conv_pack(self.weight.detach(), tile)
        dst = out if out is not None else torch.empty((B, k[0], H // stride, W // stride), device=x.device, dtype=torch.float32)
        return _ops.direct_conv_into(x, packed, tile, bias, dst, k[0], k[2], stride, act, dst_nhwc=out_nhwc)

    def forward_upsampled(self, x, out=None):
        """``self(upsample2x(x), out=out)`` -- the decoder's up-convolutions.  Where the bottleneck kernel takes the layer on the up-sampled map it
        interpolates while it stages its input (one launch less; the same bits: csrc/bottleneck_conv.hip, UP2X); else the two launches."""
        if self.bottleneck and _UP2X_IN_CONV and not self.defer_epilogue and x.is_contiguous() and tuple(x.shape[2:]) == (8, 10) and self.stride[0] == 1 \
                and self.activation in (_ops.ACTIVATIONS["none"], _ops.ACTIVATIONS["relu"]):
            B, C = x.shape[0], x.shape[1]
            buffers = self._bottleneck_for(x, shape=(B, C, 16, 20))
            if buffers is not None:
                shape = (B, self.weight.shape[0], 16, 20)
                splits = _ops.bottleneck_conv_into(x, self._bottleneck_packed, shape[1], 1, buffers, upsample=True)
                dst = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
                return _ops.partial_sums_bias_act_into(buffers, splits, dst, self.bias, self.activation, shape)
        return self(_ops.upsample2x(x), out=out)

    def _pointwise_forward(self, x, out, act, residual, residual_mode, raw):
        """The layer through csrc/pointwise_conv.hip, or None when that kernel does not take the problem (then the library GEMM + epilogue launch
        as before).  ``raw``: the convolution output without bias and activation (the depthwise consumer applies them)."""
        k = self.weight.shape
        if k[2] != 1 or k[3] != 1 or self.groups != 1 or tuple(self.stride) != (1, 1) or tuple(self.padding) != (0, 0):
            return None
        if raw:
            act = _ops.ACTIVATIONS["none"]
        key = (tuple(x.shape), act, residual_mode)
        taken = self._pointwise_taken.get(key)
        if taken is None:
            B, _, H, W = x.shape
            taken = self._pointwise_taken[key] = _ops.pointwise_conv_supported(B, k[1], H, W, k[0], act, residual_mode)
        if not taken or x.data_ptr() % 16 != 0 or (out is not None and out.data_ptr() % 16 != 0):
            return None
        if self._pointwise_packed is None:
            if torch.cuda.is_current_stream_capturing():
                return None      # (packed at the next eager call: the first frame of every kind runs eagerly)
            self._pointwise_packed = _ops.pointwise_conv_pack(self.weight.detach())
        B, _, H, W = x.shape
        dst = out if out is not None else torch.empty((B, k[0], H, W), device=x.device, dtype=torch.float32)
        return _ops.pointwise_conv_into(x, self._pointwise_packed, None if raw else self.bias, dst, k[0], act, residual, residual_mode)

    def _bottleneck_for(self, x, shape=None):
        """The partial-sum buffer for this input shape if the bottleneck kernel takes the problem (else None); packs the weights the
        first time (outside a stream capture: the first frame of every kind runs eagerly).  ``shape``: the shape of the map the layer convolves
        when that is not ``x`` itself (``forward_upsampled``)."""
        key = tuple(x.shape) if shape is None else tuple(shape)
        if shape is not None:
            x = _ShapeOn(key, x.device)
        if key not in self._bottleneck_buffers:
            k = self.weight.shape
            ok = (not self.depthwise and self.groups == 1 and k[2] == 3 and k[3] == 3 and tuple(self.padding) == (1, 1) and
                  tuple(self.dilation) == (1, 1) and self.stride[0] == self.stride[1])
            splits = _ops.bottleneck_conv_splits(x.shape[0], k[0], k[1], x.shape[2], x.shape[3], self.stride[0]) if ok else 0
            if splits == 0:
                self._bottleneck_buffers[key] = None
            elif torch.cuda.is_current_stream_capturing():
                return None      # (not cached: decided at the next eager call)
            else:
                if self._bottleneck_packed is None:
                    self._bottleneck_packed = _ops.bottleneck_conv_pack(self.weight.detach())
                P = (x.shape[2] // self.stride[0]) * (x.shape[3] // self.stride[0])
                self._bottleneck_buffers[key] = torch.empty(splits * x.shape[0] * k[0] * P, device=x.device, dtype=torch.float32)
        return self._bottleneck_buffers[key]

    def _plan_eligible(self, act):
        k = self.weight.shape
        return (not self.depthwise and self.groups == 1 and self.bias is not None and k[2] == k[3] and self.stride[0] == self.stride[1]
                and self.padding[0] == self.padding[1] and tuple(self.dilation) == (1, 1)
                and act in (_ops.ACTIVATIONS["none"], _ops.ACTIVATIONS["relu"]))


    def _time_plan(self, x, act):
        """(use the MIOpen fusion plan for this input shape?, plan us, convolution + epilogue us, max |difference|).
        The plan is used only when its result is bit-identical to the two-launch form on this input AND faster: which of two
        differently-rounded fp32 results a layer produces must not depend on a timing (the timing then only decides
        between two ways of computing the same bits)."""
        with torch.no_grad():
            probe = _ops.conv_bias_act_into(x, self.weight, self.bias, None, self.stride[0], self.padding[0], act)
            if probe is None:
                return (False, float("nan"), float("nan"), float("nan"))

            def two_launches():
                y = nn.functional.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups)
                return _ops.bias_act_into(y, y, self.bias, act)

            difference = float((two_launches() - probe).abs().max())
            t_two = _graph_microseconds(two_launches)
            t_plan = _graph_microseconds(lambda: _ops.conv_bias_act_into(x, self.weight, self.bias, probe, self.stride[0], self.padding[0], act))
        return (difference == 0.0 and t_plan < t_two, t_plan, t_two, difference)


def fuse_epilogues(module):
    """In place on an eval-mode (BN-folded) copy: Conv2d [+ Identity] + ReLU / Sigmoid, and bare biased Conv2d, become
    FusedConv2d.  Convolutions without bias and without activation are left alone."""
    for parent in module.modules():
        names = list(parent._modules.keys())
        if isinstance(parent, nn.Sequential):
            i = 0
            while i < len(names):
                m = parent._modules[names[i]]
                if isinstance(m, nn.Conv2d):
                    j = i + 1
                    while j < len(names) and isinstance(parent._modules[names[j]], nn.Identity):
                        j += 1
                    nxt = parent._modules[names[j]] if j < len(names) else None
                    if isinstance(nxt, nn.ReLU):
                        parent._modules[names[i]] = FusedConv2d(m, "relu")
                        parent._modules[names[j]] = nn.Identity()
                    elif isinstance(nxt, nn.Sigmoid):
                        parent._modules[names[i]] = FusedConv2d(m, "sigmoid")
                        parent._modules[names[j]] = nn.Identity()
                    elif m.bias is not None:
                        parent._modules[names[i]] = FusedConv2d(m, "none")
                i += 1
        else:
            for name in names:
                m = parent._modules[name]
                if isinstance(m, nn.Conv2d) and m.bias is not None:
                    parent._modules[name] = FusedConv2d(m, "none")
    # a ReLU convolution directly followed by a depthwise layer: its epilogue moves into the depthwise kernel's input read
    for parent in module.modules():
        if not isinstance(parent, nn.Sequential):
            continue
        fused = [m for m in parent._modules.values() if not isinstance(m, nn.Identity)]
        for first, second in zip(fused, fused[1:]):
            if isinstance(first, FusedConv2d) and isinstance(second, FusedConv2d) and second.depthwise and not first.depthwise and \
                    first.activation == _ops.ACTIVATIONS["relu"] and first.bias is not None and first.groups == 1:
                first.defer_epilogue = True
                second.pre_bias = first.bias
    # shortcut sums folded into the producing convolution's epilogue
    from dvmvs.backbone import FeaturePyramidNetwork, InvertedResidual
    for m in module.modules():
        if isinstance(m, InvertedResidual) and m.apply_residual and isinstance(m.layers[6], FusedConv2d):
            m.fused_head, m.fused_tail = nn.Sequential(*list(m.layers)[:6]), m.layers[6]
        if isinstance(m, FeaturePyramidNetwork) and all(isinstance(b, FusedConv2d) for b in m.inner_blocks):
            m.fused_top_down = True
    return module


class GraphedModule(nn.Module):
    """An inference module whose forward is replayed as ONE hipGraph per input signature (shapes of its tensor arguments): the first call with a
    signature runs eagerly (MIOpen picks its solvers, the MFMA kernels pack their weights), the second captures, later ones copy the arguments into
    the graph's static input buffers (one batched launch: dvmvs_copy_batch) and replay.  Positional and keyword TENSOR arguments only (that is what the
    reference's loop passes to its modules).  Outputs are the graph's static output buffers -- valid until the next call with the same signature --
    or fresh copies of them (``clone_outputs``: for modules whose results of several calls are alive at once, e.g. the feature shrinker of a loop that
    computes the measurement frames' and the reference frame's features one after the other).  Same kernels on the same inputs as the wrapped
    module: bit-identical results."""

    def __init__(self, module, clone_outputs=False):
        super().__init__()
        self.module, self.clone_outputs = module, clone_outputs
        self._seen, self._graphs = set(), {}

    def forward(self, *args, **kwargs):
        names = sorted(kwargs)
        tensors = list(args) + [kwargs[k] for k in names]
        if torch.is_grad_enabled() or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in tensors) or torch.cuda.is_current_stream_capturing():
            return self.module(*args, **kwargs)
        key = (len(args), tuple(names)) + tuple((tuple(t.shape), t.dtype) for t in tensors)
        entry = self._graphs.get(key)
        if entry is None:
            if key not in self._seen:      # first sight: eager (warm-up)
                self._seen.add(key)
                return self.module(*args, **kwargs)
            static_in = [torch.empty_like(t, memory_format=torch.contiguous_format) for t in tensors]
            for d, t in zip(static_in, tensors):
                d.copy_(t)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.module(*static_in[:len(args)], **dict(zip(names, static_in[len(args):])))
            entry = self._graphs[key] = (graph, static_in, out)
        graph, static_in, out = entry
        pending = []
        for d, t in zip(static_in, tensors):
            if t.data_ptr() == d.data_ptr():
                continue
            if _ops.batchable(d, t):
                pending.append((d, t))
                if len(pending) == 8:
                    _ops.copy_batch(pending)
                    pending = []
            else:
                d.copy_(t)
        if len(pending) == 1:
            pending[0][0].copy_(pending[0][1])
        elif pending:
            _ops.copy_batch(pending)
        graph.replay()
        if not self.clone_outputs:
            return out
        return self._clone(out)

    @staticmethod
    def _clone(out):
        if isinstance(out, torch.Tensor):
            return out.clone()
        flat = [t for t in out if isinstance(t, torch.Tensor)]
        fresh = [torch.empty_like(t) for t in flat]
        pairs = [(d, t) for d, t in zip(fresh, flat)]
        if all(_ops.batchable(d, t) for d, t in pairs) and 1 < len(pairs) <= 8:
            _ops.copy_batch(pairs)
        else:
            for d, t in pairs:
                d.copy_(t)
        it = iter(fresh)
        return type(out)(next(it) if isinstance(t, torch.Tensor) else t for t in out)


def accelerate(*modules, direct_convs=True, bottleneck_convs=True, pointwise_convs=True, graphs=False):
    """Inference copies of the network modules for the reference's OWN per-frame loop (fusionnet/run-testing.py:151-204, pairnet alike) -- one added
    line in the script, ``feature_extractor, feature_shrinker, cost_volume_encoder, lstm_fusion, cost_volume_decoder = accelerate(...)`` after the
    checkpoints are loaded: same call signatures and return values as the modules, with eval-mode BatchNorm folded into the convolutions, bias +
    activation in the convolution's epilogue, and the dense 3x3 / 5x5 layers, the bottleneck layers and the 1x1 layers on the MFMA kernels of
    csrc/direct_conv.hip / csrc/bottleneck_conv.hip / csrc/pointwise_conv.hip -- what DepthEngine runs per layer, without its frame-level machinery (no graphs, no feature cache, no look-ahead, no
    destination passing).  The originals are left untouched; ``None`` entries (pairnet has no LSTM) pass through.  Eval-mode inference only.
    ``graphs=True``: every module except the LSTM fusion (a handful of launches, None-able arguments) is additionally wrapped in a ``GraphedModule``
    -- its forward becomes one hipGraph replay per call (the eager loop is host-bound: 728 launches per frame at ~12 us of host time each); the
    modules of which several results are alive at once in the reference's loop (feature extractor / shrinker: measurement frames, then the
    reference frame) return copies."""
    out = []
    for m in modules:
        if m is None:
            out.append(None)
            continue
        fast = fuse_epilogues(fold_batchnorm(m))
        for sub in fast.modules():
            if isinstance(sub, FusedConv2d):
                sub.direct_conv, sub.bottleneck, sub.pointwise_conv = bool(direct_convs), bool(bottleneck_convs), bool(pointwise_convs)
        out.append(fast)
    if graphs:
        from dvmvs.fusionnet.model import LSTMFusion
        wrapped = []
        for position, m in enumerate(out):
            if m is None or isinstance(m, LSTMFusion):
                wrapped.append(m)
            else:      # the first two modules of the scripts' lists are the feature extractor and the feature shrinker
                wrapped.append(GraphedModule(m, clone_outputs=position == 1))
        out = wrapped
    return out


# ----------------------------------------------------------------------------------------------------------------------
# frame engine
# ----------------------------------------------------------------------------------------------------------------------
class DepthEngine:
    """Sequential keyframe processor for one video sequence (batch 1: the reference's case) or ``sequences`` = S independent
    sequences advancing in lockstep (batch S: one cost-volume launch, one convolution per layer for all of them), pairnet
    (``lstm_fusion=None``) or fusionnet.

    ``step`` mirrors one iteration of the reference loop; ``reset`` is the "TRACKING LOST" rule (per sequence when S > 1).
    With S > 1 every frame takes the "has a previous frame" path: a sequence without one has zero state, zero previous
    depth and therefore an all-zero depth estimate, for which warp + mask return exactly the zeros the reference's
    first-frame path starts from (convlstm.py:29-41), so the per-sequence results do not depend on what the others do.
    """

    def __init__(self, feature_extractor, feature_shrinker, cost_volume_encoder, lstm_fusion, cost_volume_decoder,
                 device="cuda", min_depth=0.25, max_depth=20.0, n_depth_levels=64, fold_bn=True, cache_features=True,
                 use_graphs=True, cache_size=None, channels_last=False, fuse=True, lstm_channels_last=True, sequences=1,
                 pose_algebra=None, conv_plans=None, bottleneck_convs=None, direct_convs=None, max_lookahead=None, pointwise_convs=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DepthEngine runs on an MI355X; there is no CPU execution path in this package")
        if os.environ.get(_DETERMINISTIC_MIOPEN[0]) != _DETERMINISTIC_MIOPEN[1] and not _MIOPEN_SWITCH_WAS_PRESET:
            import warnings
            warnings.warn(f"{_DETERMINISTIC_MIOPEN[0]} was changed after dvmvs.engine set it: MIOpen may pick atomically accumulating kernels "
                          "and depth is then not bit-reproducible run to run", RuntimeWarning)
        prep = (lambda m: fold_batchnorm(m)) if fold_bn else (lambda m: copy.deepcopy(m).eval())
        mods = [feature_extractor, feature_shrinker, cost_volume_encoder, lstm_fusion, cost_volume_decoder]
        mods = [None if m is None else prep(m).to(self.device) for m in mods]
        if fuse and not channels_last:
            mods = [None if m is None else fuse_epilogues(m) for m in mods]
        if channels_last:
            mods = [None if m is None else m.to(memory_format=torch.channels_last) for m in mods]
        self.fe, self.fs, self.enc, self.lstm, self.dec = mods
        # epilogues inside MIOpen's Winograd kernels where that is measurably faster (FusedConv2d.plans; DVMVS_CONV_PLANS=0 disables)
        if conv_plans is None:
            conv_plans = os.environ.get("DVMVS_CONV_PLANS", "1") != "0"
        self.conv_plans = bool(conv_plans and fuse and not channels_last)
        for m in mods:
            for sub in ([] if m is None else m.modules()):
                if isinstance(sub, FusedConv2d):
                    sub.plan_epilogue = self.conv_plans
        # the 3x3 layers on the 8x10 / 16x20 maps and the ConvLSTM convolution through the deterministic MFMA kernel
        # (csrc/bottleneck_conv.hip; DVMVS_BOTTLENECK_CONVS=0: MIOpen)
        if bottleneck_convs is None:
            bottleneck_convs = os.environ.get("DVMVS_BOTTLENECK_CONVS", "1") != "0"
        self.bottleneck_convs = bool(bottleneck_convs and fuse and not channels_last)
        for m in mods:
            for sub in ([] if m is None else m.modules()):
                if isinstance(sub, FusedConv2d):
                    sub.bottleneck = self.bottleneck_convs
        # the dense 3x3 / 5x5 layers of the 1/8 ... full-resolution maps and the depth heads through the direct MFMA convolution
        # (csrc/direct_conv.hip; DVMVS_DIRECT_CONVS=0: MIOpen + epilogue launch as in rounds 1-3)
        if direct_convs is None:
            direct_convs = os.environ.get("DVMVS_DIRECT_CONVS", "1") != "0"
        self.direct_convs = bool(direct_convs and fuse and not channels_last)
        for m in mods:
            for sub in ([] if m is None else m.modules()):
                if isinstance(sub, FusedConv2d):
                    sub.direct_conv = self.direct_convs
        # the 1x1 layers (feature extractor, FPN lateral layers) through the MFMA GEMM with bias + ReLU + residual in its store path
        # (csrc/pointwise_conv.hip; DVMVS_POINTWISE_CONVS=0: library GEMM + epilogue launch as in rounds 1-5)
        if pointwise_convs is None:
            pointwise_convs = os.environ.get("DVMVS_POINTWISE_CONVS", "1") != "0"
        self.pointwise_convs = bool(pointwise_convs and fuse and not channels_last)
        for m in mods:
            for sub in ([] if m is None else m.modules()):
                if isinstance(sub, FusedConv2d):
                    sub.pointwise_conv = self.pointwise_convs
        self._lstm_packed, self._lstm_partials, self._lstm_combined = None, None, None
        if lstm_channels_last and self.lstm is not None and not channels_last:
            # the ConvLSTM convolution (1024 -> 2048 channels on an 8x10 map, 75 MB of weights) is weight-bandwidth bound;
            # MIOpen's NHWC kernel for it takes 56 us against 88 us for NCHW, which more than pays for the two small layout
            # copies around it (results are those of a different but equally valid fp32 summation order)
            conv = self.lstm.lstm_cell.conv
            conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
        self.channels_last = channels_last
        self.min_depth, self.max_depth, self.n_depth_levels = float(min_depth), float(max_depth), int(n_depth_levels)
        self.cache_features = cache_features
        self.cache_size = cache_size or (Config.test_keyframe_buffer_size + 2)
        self.use_graphs = use_graphs
        self.height, self.width = Config.test_image_height, Config.test_image_width
        self.sequences = int(sequences)
        if self.sequences < 1:
            raise ValueError("sequences must be >= 1")
        # destination-passing frame body (see the module docstring): needs the fused epilogues and contiguous channel slices
        self.direct = bool(fuse and fold_bn and not channels_last and self.sequences == 1)
        self.pose_algebra = _pose_algebra._mode(pose_algebra)
        if self.pose_algebra == "auto":      # (the engine takes poses and intrinsics from the host: that is "reference")
            self.pose_algebra = "reference"
        self._prev_pose_host = torch.eye(4).repeat(self.sequences, 1, 1)
        self._no_previous = torch.ones(self.sequences, dtype=torch.bool)   # sequences whose next frame has no previous frame
        self._ring, self._ring_pos = None, 0
        self._feature_cache = OrderedDict()      # frame id -> features (a buffer of the pool below), least recently used first
        self._feature_pool, self._feature_free, self._feature_slot = None, [], {}
        self._graphs = {}
        self._static = None
        self._direct_buffers = {}
        self._warm = set()
        self.step_clock = None
        self._parity, self._prefetched = 0, None      # buffer set of the next frame; (frame_id, buffer set) whose reference features are ready
        self._side_stream = torch.cuda.Stream(device=self.device)
        # a third stream for the frame's own independent small kernels (round 6): the re-projection + hidden-state warp next to the sweep and the
        # encoder, a decoder level's depth head + its up-sampling next to the level's up-convolution -- 3 - 5 us launches that only waited in line
        # behind kernels they do not depend on; inside the frame graph they are parallel branches (DVMVS_AUX_STREAM=0: in line, as in rounds 3-5)
        self._aux_stream = torch.cuda.Stream(device=self.device) if _AUX_STREAM != "0" else None
        self._planner, self._planned, self._param_host_ahead = None, None, None      # see plan_ahead
        self.plan_frames_ahead = os.environ.get("DVMVS_PLAN_AHEAD", "1") != "0"
        self.planned_frames_used = 0
        self.ring_wait_seconds = 0.0     # time step() spent waiting for the device at the staging ring (NOT host work: bench.py subtracts it)
        self.lookahead_rejected = 0      # announced frames whose prepared stages were not taken because the image came as another tensor
        self._switch_interval_before = None
        self._filler_buffers, self._filler_graphs = None, []
        self._copy_queue = None      # a list while step() collects its input copies (see _copy)
        self._staging_seen = {}      # staging slot address -> whether the device may read it in place (see _staging_visible)
        self.warm_captured_graphs = os.environ.get("DVMVS_WARM_GRAPHS", "1") != "0"
        # launches of every newly captured graph on throw-away results: the runtime finishes setting a graph up over its first launches (a
        # 5-6 ms device stall was still seen at a graph's third launch, i.e. a few steps into a short run's timed region, with one)
        self.warm_graph_launches = max(1, int(os.environ.get("DVMVS_WARM_GRAPH_LAUNCHES", "3")))
        # how much of the NEXT keyframe a step computes when the caller announces it (step's next_* arguments): 1 its feature extraction,
        # 2 also its sweep + encoder.  Default 1: with the direct convolution kernels the sweep and the encoder fill the chip on their
        # own, and running them next to the decoder only makes both slower (MI355X, 100 steps: level 0 / 1 / 2 = 827 / 1091 / 999
        # frames/s); the feature extractor's small kernels are what overlaps well.  DVMVS_LOOKAHEAD overrides.
        if max_lookahead is None:
            max_lookahead = int(os.environ.get("DVMVS_LOOKAHEAD", "1"))
        self.max_lookahead = max(0, min(2, int(max_lookahead)))
        # what warming up costs (bench.py reports it): wall seconds of the eager first frames, of graph capture, of the first launches of
        # graphs captured ahead; how many such graphs
        self.warmup_seconds = {"eager_first_frames": 0.0, "graph_capture": 0.0, "graph_first_launches": 0.0}
        self.warmup_graphs_launched = 0
        self.lstm_gates_on_partials = os.environ.get("DVMVS_LSTM_GATES_ON_PARTIALS", "1") != "0"      # (0: reduction launch + gates, round 4)
        self.sweep_variant_counts = {}     # frames per sweep configuration (dvmvs_cost_volume_fwd's variant) since construction
        # host-planned work list for the sweep: only where the matrices exist on the host, and one plan per launch (one sequence)
        self.sweep_work_list = bool(_utils.SWEEP_WORK_LIST and self.pose_algebra == "reference" and _utils.COST_VOLUME_VARIANT in (0, 2, 3, 4, 5))
        # Round 5: the correlate-then-interpolate sweep (variant 6, csrc/sweep_mfma.hip) takes the keyframe pairs on which its estimated work is
        # small (dvmvs_sweep_plan6: 176 of the sample scene's 285 pairs, 29 us against 33 on the easy ones, 36.5 against 42.0 us over all of them);
        # it reads one 128-byte line per measurement cell, so the engine then keeps its measurement maps -- the feature cache and the per-frame
        # buffers -- channels-last (one transposing launch per keyframe in place of the cache's contiguous copy).  DVMVS_SWEEP_MFMA=0: round 4's
        # all-tiled engine on NCHW maps.
        # (the tiled kernels read channels-last maps only from 64 x 64 cells on -- and so does dvmvs_sweep_plan6 take variant 6)
        self.sweep_mfma = bool(self.direct and self.sweep_work_list and _utils.COST_VOLUME_VARIANT == 0 and os.environ.get("DVMVS_SWEEP_MFMA", "1") != "0"
                               and (self.height // 2) * (self.width // 2) >= 64 * 64)
        # (round 6: dvmvs_sweep_plan6 returns 6 for every single-sequence frame -- one sweep configuration, one frame graph per buffer set and pattern)
        self._tiled_variants = (6,) if self.sweep_mfma else (2, 3, 4, 5)
        self.reset()

    def conv_plan_report(self):
        """[(input shape, weight shape, uses the MIOpen fusion plan, plan us, convolution + epilogue us, max |difference|)] of
        every dense convolution problem timed so far."""
        rows = []
        for m in (self.fe, self.fs, self.enc, self.lstm, self.dec):
            for sub in ([] if m is None else m.modules()):
                if isinstance(sub, FusedConv2d):
                    rows += [(shape, tuple(sub.weight.shape)) + tuple(plan) for shape, plan in sub.plans.items()]
        return rows

    # ---- state ------------------------------------------------------------------------------------------------------
    @property
    def is_fusionnet(self):
        return self.lstm is not None

    def reset(self, sequence=None):
        """Forget the recurrent state and the previous depth/pose (reference: "TRACKING LOST", run-testing.py:97-101), of all
        sequences or of one (S > 1)."""
        if sequence is None or self.sequences == 1:
            self.has_previous = False
            self._no_previous[:] = True
            if self._static is not None:
                for k in ("h", "c", "prev_depth"):
                    self._static[k].zero_()
                if self._direct_buffers:      # the splat's all-zero estimate invariant, also after a frame that failed half-way
                    for buffer_set in self._direct_buffers["sets"]:
                        buffer_set["estimate"].zero_()
        else:
            self._no_previous[sequence] = True
            if self._static is not None:
                for k in ("h", "c", "prev_depth"):
                    self._static[k][sequence].zero_()

    def load_state(self, hidden, cell, previous_depth, previous_pose):
        """Installs a recurrent state (h, c [S,512,H/32,W/32]), previous depth [S,1,H,W] or [S,H,W] and previous pose [S,4,4], e.g. a
        checkpointed one: the next ``step`` continues from it as if this engine had produced it (the state the reference's loop
        carries in lstm_state / previous_depth / previous_pose, fusionnet/run-testing.py:86-88,201-202)."""
        if not self.is_fusionnet:
            raise RuntimeError("pairnet keeps no state between frames")
        self._allocate_static(1)
        s, S = self._static, self.sequences
        s["h"].copy_(hidden.reshape(s["h"].shape))
        s["c"].copy_(cell.reshape(s["c"].shape))
        s["prev_depth"].copy_(previous_depth.reshape(S, 1, self.height, self.width))
        self._prev_pose_host = _pose_algebra.to_host(previous_pose).reshape(S, 4, 4).clone().float()
        self._no_previous[:] = False
        self.has_previous = True

    def state(self):
        """(h, c, previous depth, previous pose) as ``load_state`` takes them (clones; the pose lives on the host)."""
        s = self._static
        return s["h"].clone(), s["c"].clone(), s["prev_depth"].clone(), self._prev_pose_host.clone()

    def clear_feature_cache(self):
        self._feature_cache.clear()
        if self._feature_pool is not None:
            self._feature_free = list(range(self._feature_pool.shape[0] - 1, -1, -1))
        self._feature_slot = {}

    def new_sequence(self):
        """Start of another video sequence on the same engine: forget the recurrent state AND the cached keyframe features
        (the cache is keyed by the caller's frame ids, which restart with every sequence)."""
        self.reset()
        self.clear_feature_cache()
        self._prefetched = None      # (frame ids restart with the sequence)

    # ---- pieces -----------------------------------------------------------------------------------------------------
    def _features(self, image):
        if self.channels_last:
            image = image.contiguous(memory_format=torch.channels_last)
        return self.fs(*self.fe(image))

    def _half_features(self, frame_id, image):
        if self.cache_features and frame_id is not None and frame_id in self._feature_cache:
            self._feature_cache.move_to_end(frame_id)
            return self._feature_cache[frame_id]
        half = self._features(image)[0].contiguous()
        self._remember(frame_id, half)
        return half

    def _remember(self, frame_id, half):
        """Caches a keyframe's half-resolution features.  The entries live in a pool of ``cache_size`` buffers allocated once: a
        ``clone()`` per keyframe made the first ``cache_size`` frames of every run pay a device allocation each (hipMalloc synchronises:
        a 20-step run measured 1.7 ms per frame where 100 steps measured 1.39)."""
        if self.cache_features and frame_id is not None:
            if not half.is_contiguous() and not self.sweep_mfma:      # (channels-last engines keep their maps as they are: the sweep reads them as NHWC)
                self._feature_cache[frame_id] = half
                while len(self._feature_cache) > self.cache_size:
                    self._feature_slot.pop(self._feature_cache.popitem(last=False)[0], None)
                return
            if self._feature_pool is None or tuple(self._feature_pool.shape[1:]) != tuple(half.shape):
                if self.sweep_mfma and half.dim() == 4:      # channels-last entries (what the sweep reads): [n, S, H, W, C] storage, [n, S, C, H, W] view
                    S_, C_, H_, W_ = half.shape
                    self._feature_pool = torch.empty((self.cache_size + 1, S_, H_, W_, C_), device=self.device, dtype=torch.float32).permute(0, 1, 4, 2, 3)
                else:
                    self._feature_pool = torch.empty((self.cache_size + 1,) + tuple(half.shape), device=self.device, dtype=torch.float32)
                self._feature_free = list(range(self.cache_size, -1, -1))
                self._feature_cache.clear()
                self._feature_slot = {}
            if frame_id in self._feature_cache:
                slot = self._feature_cache.pop(frame_id)
            else:
                while len(self._feature_cache) >= self.cache_size:      # least recently used entry out, its buffer back to the pool
                    self._feature_free.append(self._feature_slot.pop(self._feature_cache.popitem(last=False)[0]))
                slot = self._feature_pool[self._feature_free[-1]]
                self._feature_slot[frame_id] = self._feature_free.pop()
            if slot.data_ptr() != half.data_ptr():
                self._copy(slot, half)
            self._feature_cache[frame_id] = slot

    @staticmethod
    def _store_measurement_map(dst, src):
        """``dst`` = ``src``; an NCHW map into a channels-last buffer through the transposing kernel (torch's strided copy writes 4 bytes per
        128-byte line)."""
        if src.stride() == dst.stride() or not (src.is_contiguous() and dst.is_contiguous(memory_format=torch.channels_last) and src.shape[1] % 4 == 0 and src.shape[1] <= 64):
            dst.copy_(src)      # (the frame path: channels-last into channels-last)
        else:
            _ops.nchw_to_nhwc_into(src, dst)

    def _copy(self, dst, src):
        """``dst`` = ``src`` (see _store_measurement_map).  Inside ``step`` the flat copies -- this keyframe's features into the feature cache,
        the measurement maps into the sweep's buffers, the images into their homes: four runtime copies of 0.6 - 1 MB in front of every frame graph,
        5 us each on the device and as much again on the host -- are collected and issued as ONE launch (dvmvs_copy_batch) before the
        frame's parameters go up.  A copy that touches a range a queued copy writes (or writes one a queued copy reads) flushes the queue first."""
        queue = self._copy_queue
        if queue is None or not _BATCH_COPIES or not _ops.batchable(dst, src):
            self._flush_copies()
            self._store_measurement_map(dst, src)
            return
        nbytes = 4 * dst.numel()
        d0, s0 = dst.data_ptr(), src.data_ptr()
        for qd, qs in queue:
            q0, r0, qn = qd.data_ptr(), qs.data_ptr(), 4 * qd.numel()
            if (d0 < q0 + qn and q0 < d0 + nbytes) or (s0 < q0 + qn and q0 < s0 + nbytes) or (d0 < r0 + qn and r0 < d0 + nbytes):
                self._flush_copies()
                break
        queue.append((dst, src))
        if len(queue) == 8:
            self._flush_copies()

    def _flush_copies(self):
        queue = self._copy_queue
        if queue:
            if len(queue) == 1:
                queue[0][0].copy_(queue[0][1])
            else:
                _ops.copy_batch(queue)
            queue.clear()

    def _allocate_static(self, n_meas):
        d, H, W, S = self.device, self.height, self.width, self.sequences
        z = lambda *s: torch.zeros(*s, device=d, dtype=torch.float32)
        if self._static is None:
            # the frame's small matrices: one device buffer (one upload per frame), fixed offsets so that captured graphs keep
            # pointing at the right place; Hm / kt are sized for the ABI's maximum number of measurement frames
            self._param_offsets, total = self._parameter_layout(S, H, W, self.n_depth_levels)
            params = z(total)
            view = lambda name, *shape: params[self._param_offsets[name][0]:self._param_offsets[name][0] + self._param_offsets[name][1]].view(*shape)
            direct = {}
            if self.direct:
                if self.n_depth_levels != 64:
                    raise ValueError("the destination-passing frame body is laid out for the network's 64 sweep planes")
                hc = 32
                direct = dict(enc_cat=[z(1, 32 + 64, H // 2, W // 2), z(1, 32 + 2 * hc, H // 4, W // 4), z(1, 32 + 4 * hc, H // 8, W // 8),
                                       z(1, 32 + 8 * hc, H // 16, W // 16)],
                              dec_cat=[z(1, 16 * hc, H // 16, W // 16), z(1, 8 * hc + 1, H // 8, W // 8), z(1, 4 * hc + 1, H // 4, W // 4),
                                       z(1, 2 * hc + 1, H // 2, W // 2)],
                              full_in=z(1, hc + 1 + 3, H, W), lstm_cat=z(1, 32 * hc, H // 32, W // 32),
                              estimate=z(1, 1, H // 32, W // 32), depth_store=z(1, H, W))
            if self.direct:
                # feature look-ahead (step(next_reference_image=...)): the NEXT frame's image and FPN outputs need a home while this
                # frame's encoder / decoder read their own -- a second set of the buffers the feature extraction writes and of the
                # buffer that holds the image; frames alternate between the two sets
                # and, for the deeper look-ahead (the next frame's sweep + encoder as well), of everything the encoder writes
                # ... and of the 8x10 depth estimate: a frame's splat zero-fills the OTHER set's buffer on the way (no clear launch)
                if self.sweep_mfma:      # a keyframe's half-resolution features once more, channels-last: what the feature cache keeps (written in-graph)
                    direct["ref_half_nhwc"] = z(1, 32, H // 2, W // 2).contiguous(memory_format=torch.channels_last)
                keys = ("enc_cat", "dec_cat", "full_in", "lstm_cat", "estimate") + (("ref_half_nhwc",) if self.sweep_mfma else ())
                clone = lambda v: [torch.zeros_like(t) for t in v] if isinstance(v, list) else torch.zeros_like(v)
                direct["sets"] = [dict({k: direct[k] for k in keys}, index=0, meas_feat=[]),
                                  dict({k: clone(direct[k]) for k in keys}, index=1, meas_feat=[])]
            self._direct_buffers = direct
            image = direct["full_in"][:, 33:36] if self.direct else z(S, 3, H, W)      # the decoder's last concatenation ends with the image
            depth = direct["depth_store"] if self.direct else z(S, H, W)
            ref_half = direct["enc_cat"][0][:, :32] if self.direct else z(S, 32, H // 2, W // 2)
            self._static = dict(image=image, params=params,
                                reproject_T=view("reproject_T", S, 4, 4), lstm_T=view("lstm_T", S, 4, 4),
                                full_K=view("full_K", S, 3, 3), half_K=view("half_K", S, 3, 3), lstm_K=view("lstm_K", S, 3, 3),
                                pose=view("pose", S, 4, 4), prev_pose=view("prev_pose", S, 4, 4),
                                meas_pose=[view("meas_pose", _MAX_MEAS, S, 4, 4)[i] for i in range(_MAX_MEAS)],
                                # direct: the depth buffer IS next frame's previous depth (written once, by the last epilogue)
                                prev_depth=depth.view(S, 1, H, W) if self.direct else z(S, 1, H, W), h=z(S, 512, H // 32, W // 32),
                                c=z(S, 512, H // 32, W // 32), meas_feat=[], ref_half=ref_half, depth=depth)
            self._ring = [(torch.zeros(total, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(_STAGING_SLOTS)]
            self._param_host = torch.zeros(total, dtype=torch.float32)      # host mirror of the block: a step rewrites only its own regions
        zm = (lambda: z(S, 32, H // 2, W // 2).contiguous(memory_format=torch.channels_last)) if self.sweep_mfma else (lambda: z(S, 32, H // 2, W // 2))
        while len(self._static["meas_feat"]) < n_meas:
            self._static["meas_feat"].append(zm())
        if self.direct:
            sets = self._direct_buffers["sets"]
            sets[0]["meas_feat"] = self._static["meas_feat"]
            while len(sets[1]["meas_feat"]) < n_meas:
                sets[1]["meas_feat"].append(zm())

    @staticmethod
    def _parameter_layout(S, H, W, n_depth_levels):
        """({name: (offset, floats)}, total floats) of the frame's parameter block: one device buffer, one upload per frame, fixed offsets so that
        captured graphs keep pointing at the right place; Hm / kt are sized for the ABI's maximum number of measurement frames."""
        items = _ops.sweep_work_list_words(S, H // 2, W // 2, n_depth_levels)
        sizes = [("Hm", S * _MAX_MEAS * 9), ("kt", S * _MAX_MEAS * 3), ("reproject_T", S * 16), ("lstm_T", S * 16),
                 ("full_K", S * 9), ("half_K", S * 9), ("lstm_K", S * 9), ("pose", S * 16), ("prev_pose", S * 16),
                 ("meas_pose", _MAX_MEAS * S * 16),
                 # the sweep's work list (32-bit words, planned on the host per frame: dvmvs_sweep_work_list) rides in the same upload
                 ("sweep_items", items),
                 # the sweep parameters of the OTHER buffer set (look-ahead: the next frame's sweep runs during this frame)
                 ("Hm1", S * _MAX_MEAS * 9), ("kt1", S * _MAX_MEAS * 3), ("sweep_items1", items)]
        offsets, total = {}, 0
        for name, n in sizes:
            offsets[name] = (total, n)
            total += n
        return offsets, (total + 3) // 4 * 4      # (whole float4: the upload can ride in the frame's copy batch)

    def _sweep_views(self, n_meas, index=0):
        """Hm [S,n_meas,9] and kt [S,n_meas,3] views of the parameter buffer (contiguous prefixes of their regions) of buffer set ``index``."""
        S, p = self.sequences, self._static["params"]
        suffix = "1" if index else ""
        o_h, o_k = self._param_offsets["Hm" + suffix][0], self._param_offsets["kt" + suffix][0]
        return p[o_h:o_h + S * n_meas * 9].view(S, n_meas, 9), p[o_k:o_k + S * n_meas * 3].view(S, n_meas, 3)

    def _sweep_items(self, index=0):
        """The device copy of a frame's work list (a fixed region of the parameter buffer per buffer set: captured graphs keep pointing at it)."""
        if not self.sweep_work_list:
            return None
        o, n = self._param_offsets["sweep_items1" if index else "sweep_items"]
        return self._static["params"].view(torch.int32)[o:o + n]

    def _upload_frame_parameters(self, n_meas, pose, measurement_poses, full_K, index=0, own_sweep=True, next_frame=None, pending_copies=None):
        """Evaluates the frame's small matrices on the host (reference mode) and sends them, the intrinsics and the poses to
        the device with ONE asynchronous copy out of a pinned staging slot.  The slot's previous copy (issued _STAGING_SLOTS
        frames ago) must have executed before it is overwritten: its event is waited for, which never blocks in practice.

        ``index``: the buffer set of this frame (its sweep parameters live in that set's region); ``own_sweep`` False: this frame's
        sweep already ran (look-ahead), its region is left as it is; ``next_frame`` = (pose, measurement poses) of the NEXT frame whose
        sweep runs during this one: its matrices / work list go into the other set's region.  Regions a step does not rewrite keep
        their contents through the host mirror of the block.
        Returns (the host pose that becomes "the previous pose" -- committed by step() only after the frame was launched, so that a
        frame that raises leaves (h, c, previous depth, previous pose) those of one frame --, this frame's sweep configuration or
        None, the next frame's or None)."""
        try:
            planned = self._take_planned(n_meas, pose, measurement_poses, full_K, index, own_sweep, next_frame)
            if planned is not None:
                result = planned
            else:
                result = self._evaluate_frame_parameters(self._param_host, n_meas, pose, measurement_poses, full_K, self._prev_pose_host,
                                                         self._no_previous, index, own_sweep, next_frame)
        except BaseException:
            if pending_copies:      # (they belong to cache entries that are already registered: written before the error leaves)
                self._copy_queue = pending_copies
                try:
                    self._flush_copies()
                finally:
                    self._copy_queue = None
            raise
        mirror = self._param_host
        staging, event = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        t_wait = time.perf_counter()
        event.synchronize()      # (one staging slot: the host runs at most one frame ahead of the device, so in steady state this is where a step waits for it)
        self.ring_wait_seconds += time.perf_counter() - t_wait
        staging.copy_(mirror)
        with torch.cuda.device(self.device):     # the ring guard must be recorded on the engine's device, whichever is current
            params = self._static["params"]
            if pending_copies and len(pending_copies) < 8 and _UPLOAD_IN_COPY_BATCH and self._staging_visible(staging):
                # the block goes up in the SAME launch as the frame's input copies (round 6): the copy kernel reads the pinned staging slot through
                # its device-visible address -- the runtime's own copy of a pinned buffer is a blit kernel too (__amd_rocclr_copyBuffer), one
                # launch more per frame
                _ops.copy_batch(pending_copies + [(params, staging)])
            else:
                if pending_copies:
                    self._copy_queue = pending_copies
                    try:
                        self._flush_copies()
                    finally:
                        self._copy_queue = None
                params.copy_(staging, non_blocking=True)
            event.record(torch.cuda.current_stream(self.device))
        return result

    def _staging_visible(self, staging):
        """Whether kernels on the engine's device may read this pinned staging slot directly (asked once per slot: dvmvs_host_pointer_device_visible)."""
        key = staging.data_ptr()
        seen = self._staging_seen.get(key)
        if seen is None:
            from dvmvs.hip import _capi
            seen = self._staging_seen[key] = bool(staging.is_pinned() and key % 16 == 0 and _capi.lib().dvmvs_host_pointer_device_visible(key))
        return seen

    def _evaluate_frame_parameters(self, mirror, n_meas, pose, measurement_poses, full_K, previous_pose, no_previous, index, own_sweep, next_frame):
        """The host half of ``_upload_frame_parameters``: evaluates the matrices, the sweep configuration and the work list and writes
        them into ``mirror`` (the host copy of the parameter block).  Touches no engine state -- it also runs on the planning thread, a
        frame ahead, into a second mirror (``plan_ahead``)."""
        S = self.sequences
        host = _pose_algebra.to_host
        pose, full_K = host(pose).reshape(S, 4, 4), host(full_K).reshape(S, 3, 3)
        measurement_poses = [host(p).reshape(S, 4, 4) for p in measurement_poses]
        half_K, lstm_K = full_K.clone(), full_K.clone()
        half_K[:, 0:2, :] = half_K[:, 0:2, :] / 2.0       # run-testing.py:139-140
        lstm_K[:, 0:2, :] = lstm_K[:, 0:2, :] / 32.0      # run-testing.py:142-143
        # A sequence without a previous frame (S > 1 runs every sequence through the previous-frame path) gets the identity as
        # relative pose: its previous depth is all zero, and under the identity every zero-depth point stays at z = 0, which the
        # splat never writes -- an exactly empty depth estimate, as on the reference's first-frame path.
        previous = torch.where(no_previous.view(S, 1, 1), pose, previous_pose)

        def put(name, tensor):
            o, n = self._param_offsets[name]
            flat = tensor.reshape(-1)
            mirror[o:o + flat.numel()].copy_(flat)

        def put_sweep(region, ref_pose, meas_poses):
            """Sweep matrices, configuration and work list of one frame into buffer set ``region``'s part of the block."""
            suffix = "1" if region else ""
            Hm, kt = _pose_algebra.sweep_matrices_host(ref_pose, meas_poses, half_K)
            put("Hm" + suffix, Hm)
            put("kt" + suffix, kt)
            # which sweep configuration suits this keyframe geometry and its work list: decided here, on the host copies, in one walk
            # over the (tile, chunk) pairs (one graph per configuration)
            variant = _utils.COST_VOLUME_VARIANT
            if self.sweep_work_list:
                o, n = self._param_offsets["sweep_items" + suffix]
                variant = _ops.sweep_plan_host(Hm, kt, self.height // 2, self.width // 2, self.n_depth_levels, self.min_depth, self.max_depth,
                                               variant if variant in (2, 3, 4, 5) else 0, mirror.view(torch.int32)[o:o + n], allow_mfma=self.sweep_mfma)
            elif variant == 0:
                variant = _utils.sweep_variant((Hm, kt), self.height // 2, self.width // 2, self.n_depth_levels, self.min_depth, self.max_depth)
            return variant

        sweep_variant = _utils.COST_VOLUME_VARIANT if own_sweep else None
        next_variant = None
        if self.pose_algebra == "reference":
            if own_sweep:
                sweep_variant = put_sweep(index, pose, measurement_poses)
            if next_frame is not None:
                next_variant = put_sweep(1 - index, host(next_frame[0]).reshape(S, 4, 4), [host(p).reshape(S, 4, 4) for p in next_frame[1]])
            if self.is_fusionnet:
                eye = torch.eye(4).expand(S, 4, 4)
                if bool(no_previous.all()):      # nothing to relate to: the identity, exactly (see above)
                    reproject_T = lstm_T = eye
                else:
                    fresh = no_previous.view(S, 1, 1)
                    reproject_T = torch.where(fresh, eye, _pose_algebra.relative_pose_host(pose, previous))   # utils.py:121
                    lstm_T = torch.where(fresh, eye, _pose_algebra.relative_pose_host(previous, pose))        # convlstm.py:30
                put("reproject_T", reproject_T)
                put("lstm_T", lstm_T)
        put("full_K", full_K)
        put("half_K", half_K)
        put("lstm_K", lstm_K)
        put("pose", pose)
        put("prev_pose", previous)
        put("meas_pose", torch.stack(measurement_poses))
        return pose.clone(), sweep_variant, next_variant

    # ---- planning a frame ahead on a second host thread -----------------------------------------------------------------
    def plan_ahead(self, n_meas, pose, measurement_poses, full_K, previous_pose, index):
        """Starts evaluating the NEXT frame's parameter block (pose algebra, sweep configuration, work list: ~0.3 ms of small host
        operations) on the planning thread, while this thread is inside hipGraphLaunch (0.4 ms, GIL released).  With the direct
        convolutions a frame takes the device 0.8 ms and took the host 0.9: the host had become the bound.  The next step takes the
        result if it is called with these very poses (else it evaluates its own, as before): same functions on the same inputs."""
        if self._planner is None:
            from concurrent.futures import ThreadPoolExecutor
            self._planner = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dvmvs-plan")
            # Two Python threads share the interpreter lock: a thread that wants it asks the holder to drop it only after the switch
            # interval -- 5 ms by default, six frames of device time.  A few steps into every run this thread came back from a graph launch
            # while the planning thread was mid-block and sat out the full interval (one 5.6-6.5 ms device gap in the first ten steps of every
            # short run; none with DVMVS_PLAN_AHEAD=0).  0.1 ms bounds the hand-over; DVMVS_SWITCH_INTERVAL overrides (seconds, 0 = leave it).
            # OPT-IN since round 6 (a library must not change the embedding program's interpreter behaviour): bench.py sets
            # DVMVS_SWITCH_INTERVAL=1e-4; unset = the interpreter's interval stays as it is.  close() restores it.
            import sys
            wanted = float(os.environ.get("DVMVS_SWITCH_INTERVAL", "0"))
            if wanted > 0.0 and sys.getswitchinterval() > wanted:
                self._switch_interval_before = sys.getswitchinterval()
                sys.setswitchinterval(wanted)
            self._param_host_ahead = torch.zeros_like(self._param_host)
        no_previous = torch.zeros(self.sequences, dtype=torch.bool)
        inputs = (pose, list(measurement_poses), full_K)
        future = self._planner.submit(self._evaluate_frame_parameters, self._param_host_ahead, n_meas, pose, measurement_poses, full_K,
                                      previous_pose, no_previous, index, True, None)
        self._planned = (inputs, previous_pose, index, future)

    def close(self):
        """Stops the planning thread and puts the interpreter's switch interval back (if DVMVS_SWITCH_INTERVAL made this engine change it)."""
        if self._planner is not None:
            self._planner.shutdown(wait=True)
            self._planner, self._planned = None, None
        if self._switch_interval_before is not None:
            import sys
            sys.setswitchinterval(self._switch_interval_before)
            self._switch_interval_before = None

    def _take_planned(self, n_meas, pose, measurement_poses, full_K, index, own_sweep, next_frame):
        """The block planned a frame ahead, copied into the mirror, if it was planned for exactly this call; else None."""
        planned, self._planned = self._planned, None
        if planned is None:
            return None
        (p_pose, p_meas, p_K), p_previous, p_index, future = planned
        try:
            result = future.result()          # (also when it is not used: the second mirror must be idle before the next plan)
        except Exception:
            return None                       # evaluated again by the caller, which raises in the caller's thread
        same = lambda a, b: a is b or (tuple(a.shape) == tuple(b.shape) and a.device == b.device and torch.equal(a, b))
        if not (own_sweep and next_frame is None and p_index == index and len(p_meas) == n_meas == len(measurement_poses)
                and not bool(self._no_previous.any()) and same(p_pose, pose) and same(p_K, full_K)
                and all(same(a, b) for a, b in zip(p_meas, measurement_poses)) and torch.equal(p_previous, self._prev_pose_host)):
            return None
        self._param_host.copy_(self._param_host_ahead)
        self.planned_frames_used += 1
        return result

    # ---- destination-passing frame body (one sequence) ------------------------------------------------------------------
    def _fpn_direct(self, taps, outs, half_nhwc=None):
        """FeaturePyramidNetwork.forward (dvmvs/backbone.py; torchvision's top-down pathway) with the four used outputs written
        into ``outs`` and the unused 1/32 output (fusionnet/model.py:159-164 drops it) not computed.  ``half_nhwc``: the half-resolution
        output once more, channels-last (what a later frame's MFMA sweep reads as measurement map): written by the smoothing layer's own
        epilogue (round 6; round 5: a transposing launch behind it, 5 - 16 us of every frame)."""
        fpn = self.fs.fpn
        top = fpn.inner_blocks[-1](taps[-1])
        for level in range(len(taps) - 2, -1, -1):
            top = fpn.inner_blocks[level](taps[level], residual=top, residual_mode=2)
            if level == 0 and half_nhwc is not None and isinstance(fpn.layer_blocks[0], FusedConv2d):
                fpn.layer_blocks[0](top, out=outs[0], out_nhwc=half_nhwc)
                half_nhwc = None
            else:
                fpn.layer_blocks[level](top, out=outs[level])
        if half_nhwc is not None:
            _ops.nchw_to_nhwc_into(outs[0], half_nhwc)

    def _decoder_block_direct(self, block, x, cat, depth_head, depth_input):
        """DecoderBlock.forward (dvmvs/networks.py) on the concatenation buffer ``cat`` = [up-convolution | skip | up(depth)]; the skip
        slice has already been written by the encoder's aggregator.  ``depth_head`` (the previous level's depth layer) runs as raw
        convolution and its bias + sigmoid are applied inside the up-sampling kernel."""
        up_channels = block.up_convolution.conv[0].weight.shape[0]
        up_conv = block.up_convolution.conv[0]
        if depth_head is not None and _PAIRED_UPSAMPLING and self._aux_stream is None:
            # the level's two up-samplings -- its feature map for the up-convolution, its depth head (raw convolution, bias + sigmoid on the taps) into the
            # last channel of ``cat`` -- in ONE launch (round 6; the same bits as the two)
            B, C, H, W = x.shape
            up = torch.empty((B, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
            _ops.upsample2x_pair_into(x, up, depth_head[0](depth_input, raw=True), cat[:, -1:], depth_head[0].bias, _ops.ACTIVATIONS["sigmoid"])
            up_conv(up, out=cat[:, :up_channels])
            return block.convolution2[0](block.convolution1[0](cat))
        if depth_head is not None:
            with self._beside("heads"):      # (reads the previous level's output, writes the last channel of ``cat``: nothing the up-convolution touches)
                self._upsampled_depth_head(depth_head, depth_input, cat[:, -1:])
        if isinstance(up_conv, FusedConv2d):
            up_conv.forward_upsampled(x, out=cat[:, :up_channels])
        else:
            up_conv(_ops.upsample2x(x), out=cat[:, :up_channels])
        self._join_beside()
        return block.convolution2[0](block.convolution1[0](cat))

    @contextlib.contextmanager
    def _beside(self, use):
        """The launches inside run on the auxiliary stream, forked from the current one -- concurrently with what the current stream is given
        next, until ``_join_beside``.  Without the auxiliary stream (or with it switched on for the other use only): in line."""
        aux = self._aux_stream if _AUX_STREAM in ("1", use) else None
        if aux is None:
            yield
            return
        aux.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(aux):
            yield
        self._beside_open = True

    def _join_beside(self):
        if self._aux_stream is not None and getattr(self, "_beside_open", False):
            torch.cuda.current_stream(self.device).wait_stream(self._aux_stream)
            self._beside_open = False

    @staticmethod
    def _upsampled_depth_head(head, x, dst):
        _ops.upsample2x_into(head[0](x, raw=True), dst, head[0].bias, _ops.ACTIVATIONS["sigmoid"])

    def _lstm_bottleneck(self, x):
        """Whether the ConvLSTM convolution of input ``x`` goes through the bottleneck kernel (packs its weights the first time)."""
        if not self.bottleneck_convs:
            return False
        if self._lstm_packed is None:
            conv = self.lstm.lstm_cell.conv
            k = conv.weight.shape
            ok = conv.bias is None and tuple(k[2:]) == (3, 3) and tuple(conv.padding) == (1, 1) and tuple(conv.stride) == (1, 1) and conv.groups == 1
            splits = _ops.bottleneck_conv_splits(x.shape[0], k[0], k[1], x.shape[2], x.shape[3], 1) if ok else 0
            if splits == 0 or torch.cuda.is_current_stream_capturing():
                return False
            self._lstm_packed = _ops.bottleneck_conv_pack(conv.weight.detach())
            self._lstm_partials = torch.empty(splits * x.shape[0] * k[0] * x.shape[2] * x.shape[3], device=x.device, dtype=torch.float32)
            self._lstm_combined = torch.empty(x.shape[0], k[0], x.shape[2], x.shape[3], device=x.device, dtype=torch.float32)
        return True

    def _frame_body_direct(self, n_meas, has_previous, sweep_variant=0, parity=0, have=0, give=0, n_meas_next=0, next_variant=0):
        """One frame on buffer set ``parity``.  ``have``: how much of it the previous step already computed (0 nothing, 1 its reference
        features, 2 also its sweep + encoder); ``give``: how much of the NEXT frame (other buffer set) this step computes (0 / 1 / 2).
        When this frame's own share does not include a stage the next frame's share includes, the next frame's work runs on a second
        stream, concurrently with this frame's remaining stages: at batch 1 most kernels of a frame leave most of the chip idle, and
        the next frame's feature extraction, sweep and encoder depend on nothing this frame computes -- only the ConvLSTM and the
        decoder carry state (measured: tools/frame_stage_probe.py).  Same kernels on the same inputs either way: results are
        bit-identical to the one-stream frame."""
        sets = self._direct_buffers["sets"]
        cur, nxt = sets[parity], sets[1 - parity]

        def own(sweep_done=False, after_level=None):
            if have < 1:
                self._reference_features_direct(cur)
            warped = False
            if have < 2:
                if self.is_fusionnet and self._aux_stream is not None and _AUX_STREAM in ("1", "warp"):
                    # two 5 us launches that depend on the PREVIOUS frame only: next to this frame's sweep + encoder instead of behind them
                    with self._beside("warp"):
                        self._state_warp_direct(cur, has_previous)
                    warped = True
                self._sweep_encoder_direct(cur, n_meas, sweep_variant, sweep_done=sweep_done, after_level=after_level)
            self._lstm_decoder_direct(cur, has_previous, state_warped=warped)

        def ahead():
            if give >= 1:
                self._reference_features_direct(nxt)
            if give >= 2:
                self._sweep_encoder_direct(nxt, n_meas_next, next_variant)

        if give == 0:
            own()
        elif have < give:
            # (one stream: a stage's modules and scratch buffers are not run concurrently with themselves)
            own()
            ahead()
        else:
            main = torch.cuda.current_stream(self.device)

            def fork():
                self._side_stream.wait_stream(main)
                with torch.cuda.stream(self._side_stream):
                    ahead()

            # WHERE the next frame's work is forked off (DVMVS_FORK_AFTER, round 6).  The frame's first kernels fill the chip on their own -- the persistent
            # sweep holds every CU's registers and LDS, the 5x5 layers of encoder level 0 run 256 workgroups -- so a side stream started next to them
            # only delays them (and waits itself); its small kernels belong next to the chain's launch-bound middle (1/8 ... 1/32 maps, ConvLSTM).
            # -1: at the start (rounds 4-5); 0: behind the sweep; k = 1 ... 4: behind encoder level k - 1.  Same kernels on the same inputs.
            level = _FORK_AFTER if have == 1 else -1
            if level < 0:
                fork()
                own()
            else:
                self._sweep_direct(cur, n_meas, sweep_variant)
                if level == 0:
                    fork()
                own(sweep_done=True, after_level=(level - 1, fork) if level > 0 else None)
            main.wait_stream(self._side_stream)

    def _reference_features_direct(self, buffers):
        """MnasNet taps -> FPN of the reference image of a buffer set, each used output into the front of its encoder concatenation buffer."""
        # (sweep_mfma: the copy a later frame's sweep reads as measurement map -- one 128-byte line per cell -- is made here, inside the frame graph, so
        # that the feature cache's per-keyframe copy stays a plain device copy on the host's side)
        self._fpn_direct(self.fe(buffers["full_in"][:, 33:36]), [c[:, :32] for c in buffers["enc_cat"]],
                         half_nhwc=buffers["ref_half_nhwc"] if self.sweep_mfma else None)

    def _after_features_direct(self, n_meas, has_previous, sweep_variant=0, buffers=None):
        """Everything of a frame behind the feature extraction: sweep, encoder, re-projection, ConvLSTM, decoder."""
        buffers = self._direct_buffers["sets"][0] if buffers is None else buffers
        self._sweep_encoder_direct(buffers, n_meas, sweep_variant)
        self._lstm_decoder_direct(buffers, has_previous)

    def _sweep_direct(self, buffers, n_meas, sweep_variant=0):
        """Plane sweep of the frame whose features are in ``buffers``, into the cost-volume slice of its first encoder concatenation buffer."""
        s = self._static
        enc_cat = buffers["enc_cat"]
        Hm, kt = self._sweep_views(n_meas, buffers["index"])
        if self.pose_algebra == "exact":
            Hm, kt = _ops.sweep_matrices(s["pose"], s["meas_pose"][:n_meas], s["half_K"])
        _ops.cost_volume_into(enc_cat[0][:, :32], buffers["meas_feat"][:n_meas], Hm, kt, self.min_depth, self.max_depth, enc_cat[0][:, 32:], sweep_variant,
                              self._sweep_items(buffers["index"]))

    def _sweep_encoder_direct(self, buffers, n_meas, sweep_variant=0, sweep_done=False, after_level=None):
        """Plane sweep + cost-volume encoder of the frame whose features are in ``buffers``: reads that set's measurement features and
        sweep parameters, writes its skip connections (into the decoder's concatenation buffers) and its bottleneck map.  Depends on
        nothing the PREVIOUS frame computes -- no recurrent state, no previous depth -- which is what lets it run a frame ahead."""
        enc_cat, dec_cat = buffers["enc_cat"], buffers["dec_cat"]
        if not sweep_done:
            self._sweep_direct(buffers, n_meas, sweep_variant)
        # encoder: aggregator output = skip connection, written where the decoder will read it
        enc, dec = self.enc, self.dec
        x = None
        for level in range(4):
            skip_channels = getattr(enc, f"aggregator{level}")[0].weight.shape[0]
            cat = dec_cat[3 - level]
            up_channels = cat.shape[1] - skip_channels - (1 if level < 3 else 0)
            skip = getattr(enc, f"aggregator{level}")[0](enc_cat[level], out=cat[:, up_channels:up_channels + skip_channels])
            block = getattr(enc, f"encoder_block{level}")
            x = block.standard_convolution.conv1[0](block.down_convolution.down_conv[0](skip))
            if level < 3:
                block.standard_convolution.conv2[0](x, out=enc_cat[level + 1][:, 32:])
            elif self.is_fusionnet:
                x = block.standard_convolution.conv2[0](x, out=buffers["lstm_cat"][:, :512])
            else:
                x = block.standard_convolution.conv2[0](x, out=buffers["lstm_cat"][:, :512])      # (pairnet: the buffer is just the bottleneck's home)
            if after_level is not None and after_level[0] == level:
                after_level[1]()      # (the look-ahead fork: _frame_body_direct)

    def _state_warp_direct(self, buffers, has_previous):
        """Re-projection of the previous depth into the frame's 8x10 estimate and the hidden state warped with it into the ConvLSTM's input
        (convlstm.py:27-41, utils.py:110-154).  Reads the previous frame's depth and state only -- nothing of this frame's sweep or encoder."""
        s, d = self._static, self._direct_buffers
        lstm_cat = buffers["lstm_cat"]
        if has_previous:
            exact = self.pose_algebra == "exact"
            reproject_T = _ops.relative_pose(s["pose"], s["prev_pose"]) if exact else s["reproject_T"]
            lstm_T = _ops.relative_pose(s["prev_pose"], s["pose"]) if exact else s["lstm_T"]
            # re-projection of the previous depth straight into this buffer set's 8x10 estimate (one launch: it also zero-fills the
            # other set's estimate for the next frame), then the hidden-state warp
            other = d["sets"][1 - buffers["index"]]["estimate"]
            _ops.depth_reproject_estimate_into(reproject_T, s["prev_depth"], s["full_K"], s["half_K"], buffers["estimate"], other, 16)
            _ops.hidden_warp_into(s["h"], buffers["estimate"], lstm_T, s["lstm_K"], True, lstm_cat[:, 512:])
        else:
            lstm_cat[:, 512:].copy_(s["h"])      # first frame of a sequence: the (zero) state as it is, no warp (convlstm.py:29)

    def _lstm_decoder_direct(self, buffers, has_previous, state_warped=False):
        """Re-projection of the previous depth, ConvLSTM and decoder of the frame whose encoder outputs are in ``buffers``: the part of a
        frame that carries the recurrent state.  ``state_warped``: the re-projection + hidden-state warp have been launched already (on the
        auxiliary stream, next to the sweep and the encoder: ``_frame_body_direct``) and are only waited for here."""
        s, d = self._static, self._direct_buffers
        dec, dec_cat, lstm_cat = self.dec, buffers["dec_cat"], buffers["lstm_cat"]
        bottom = lstm_cat[:, :512]
        if self.is_fusionnet:
            cell = self.lstm.lstm_cell
            if not state_warped:
                self._state_warp_direct(buffers, has_previous)
            self._join_beside()
            if self._lstm_bottleneck(lstm_cat):
                # the 1024 -> 2048-channel convolution as K-split partial sums (75 MB of weights streamed once through the MFMA
                # pipe), added up in a fixed order by a chip-wide reduction (the gates kernel can add them itself --
                # lstm_gates_partials_into -- but its 32 workgroups take 30 us over 16 splits; reduction + gates: 5.6 + 5.3 us)
                splits = _ops.bottleneck_conv_into(lstm_cat, self._lstm_packed, cell.conv.weight.shape[0], 1, self._lstm_partials)
                if self.lstm_gates_on_partials:
                    # the gates kernel adds the K-split partial sums itself, in ascending order (one wave per LayerNorm row): no reduction launch
                    _ops.lstm_gates_partials_into(self._lstm_partials, splits, s["c"], s["h"])
                else:
                    _ops.partial_sums_bias_act_into(self._lstm_partials, splits, self._lstm_combined, None, _ops.ACTIVATIONS["none"],
                                                    tuple(self._lstm_combined.shape))
                    _ops.lstm_gates_into(self._lstm_combined, s["c"], s["h"])
            else:
                combined = cell.conv(lstm_cat)
                if not combined.is_contiguous():       # channels-last convolution (lstm_channels_last): back to the gates' NCHW rows
                    combined = combined.contiguous()
                _ops.lstm_gates_into(combined, s["c"], s["h"])
            bottom = s["h"]
        d1 = self._decoder_block_direct(dec.decoder_block1, bottom, dec_cat[0], None, None)
        d2 = self._decoder_block_direct(dec.decoder_block2, d1, dec_cat[1], dec.depth_layer_one_sixteen, d1)
        d3 = self._decoder_block_direct(dec.decoder_block3, d2, dec_cat[2], dec.depth_layer_one_eight, d2)
        d4 = self._decoder_block_direct(dec.decoder_block4, d3, dec_cat[3], dec.depth_layer_quarter, d3)
        full_in = buffers["full_in"]
        if _PAIRED_UPSAMPLING and self._aux_stream is None:
            _ops.upsample2x_pair_into(d4, full_in[:, :32], dec.depth_layer_half[0](d4, raw=True), full_in[:, 32:33], dec.depth_layer_half[0].bias,
                                      _ops.ACTIVATIONS["sigmoid"])
        else:
            with self._beside("heads"):
                self._upsampled_depth_head(dec.depth_layer_half, d4, full_in[:, 32:33])
            _ops.upsample2x_into(d4, full_in[:, :32])
            self._join_beside()
        refined = dec.refine[1][0](dec.refine[0][0](full_in))
        # last convolution: bias + sigmoid + depth mapping (model.py:231-232) in one epilogue, into the depth / previous-depth buffer
        dec.depth_layer_full[0](refined, out=s["prev_depth"], activation=_ops.ACTIVATION_SIGMOID_TO_DEPTH,
                                p0=dec.inverse_depth_multiplier, p1=dec.inverse_depth_base)

    def _frame_body(self, n_meas, has_previous, sweep_variant=0, parity=0, have=0, give=0, n_meas_next=0, next_variant=0):
        """The per-frame computation on the static buffers (this is what gets captured into a hipGraph)."""
        if self.direct:
            return self._frame_body_direct(n_meas, has_previous, sweep_variant, parity, have, give, n_meas_next, next_variant)
        s = self._static
        feats = self._features(s["image"])
        ref_half = feats[0].contiguous()
        s["ref_half"].copy_(ref_half)
        Hm, kt = self._sweep_views(n_meas)
        exact = self.pose_algebra == "exact"
        if exact:
            Hm, kt = _ops.sweep_matrices(s["pose"], s["meas_pose"][:n_meas], s["half_K"])
        cost_volume = _ops.cost_volume(ref_half, s["meas_feat"][:n_meas], Hm, kt, self.min_depth, self.max_depth, self.n_depth_levels,
                                       True, sweep_variant, self._sweep_items())
        skip0, skip1, skip2, skip3, bottom = self.enc(ref_half, feats[1], feats[2], feats[3], cost_volume)
        if self.is_fusionnet:
            if has_previous:
                reproject_T = _ops.relative_pose(s["pose"], s["prev_pose"]) if exact else s["reproject_T"]
                lstm_T = _ops.relative_pose(s["prev_pose"], s["pose"]) if exact else s["lstm_T"]
                _, depth_estimation = _ops.depth_reproject_lowres(reproject_T, s["prev_depth"], s["full_K"], s["half_K"], 16)
                state = self.lstm(bottom, (s["h"], s["c"]), s["prev_pose"], s["pose"], depth_estimation, s["lstm_K"], transformation=lstm_T)
            else:
                depth_estimation = torch.zeros(self.sequences, 1, self.height // 32, self.width // 32, device=self.device)
                state = self.lstm(bottom, None, None, s["pose"], depth_estimation, s["lstm_K"])
            s["h"].copy_(state[0])
            s["c"].copy_(state[1])
            bottom = state[0]
        prediction = self.dec(s["image"], skip0, skip1, skip2, skip3, bottom, full_resolution_only=True)[0]
        s["depth"].copy_(prediction)
        if self.is_fusionnet:
            s["prev_depth"].copy_(prediction.view(self.sequences, 1, self.height, self.width))

    # ---- public -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, reference_image, reference_pose, measurement_images, measurement_poses, full_K, frame_id=None,
             measurement_ids=None, next_reference_image=None, next_frame_id=None, next_reference_pose=None,
             next_measurement_poses=None, next_measurement_ids=None):
        """One keyframe (of each of the S sequences).  Images [S,3,H,W] normalised, on the GPU; poses [S,4,4] cam-to-world and
        ``full_K`` [S,3,3] preferably as HOST tensors (that is where they come from, and where the frame's small matrices are
        evaluated; device tensors are copied back, which synchronises).  With S > 1 the sequences advance in lockstep:
        ``frame_id`` / ``measurement_ids`` name the step for all of them and a cached feature entry holds all S maps.

        ``measurement_images[i]`` may be ``None`` when ``measurement_ids[i]`` is in the feature cache.

        Look-ahead (one sequence per engine), for callers that already know the NEXT call's frame -- a pre-computed keyframe index, a
        camera that is a frame ahead.  The stages of a frame that carry no recurrent state are computed for the next frame during
        THIS call, on a second stream, concurrently with this frame's ConvLSTM and decoder:
        * ``next_reference_image`` + ``next_frame_id``: its MnasNet + FPN features;
        * additionally ``next_reference_pose``, ``next_measurement_poses`` and ``next_measurement_ids`` (every one of them this call's
          ``frame_id`` or a cached keyframe): its plane sweep and cost-volume encoder as well.
        The next call (same ``frame_id``; same poses / measurement ids for the second level) finds them ready; anything else is simply
        recomputed.  Same kernels on the same inputs: results are bit-identical to calls without look-ahead.
        Returns the full-resolution depth [S,H,W] (a static buffer that the next call overwrites: clone to keep).
        """
        n_meas = len(measurement_poses)
        if n_meas < 1 or n_meas > _MAX_MEAS:
            raise ValueError(f"need between 1 and {_MAX_MEAS} measurement frames")
        measurement_ids = measurement_ids or [None] * n_meas
        clock = self.step_clock      # None, or a list that receives this step's host checkpoints (tools/step_times_probe.py)
        if clock is not None:
            marks = [("enter", time.perf_counter())]
            clock.append(marks)
            mark = lambda name: marks.append((name, time.perf_counter()))
        else:
            mark = lambda name: None
        self._allocate_static(n_meas)
        s = self._static
        if tuple(reference_image.shape) != (self.sequences, 3, self.height, self.width):
            raise ValueError(f"image must be [{self.sequences},3,{self.height},{self.width}], got {tuple(reference_image.shape)}")
        parity = self._parity if self.direct else 0
        sets = self._direct_buffers.get("sets")
        cur = sets[parity] if self.direct else None
        self._copy_queue = []      # (closed at "inputs copied"; an exception on the way flushes what was queued: see the except clause there)
        try:
            if self.direct:
                self._direct_buffers["estimate"] = cur["estimate"]      # (diagnostics: the 8x10 depth estimate of the frame this call computes)

            # ---- what the previous call prepared for this frame: 0 nothing, 1 its reference features, 2 also its sweep + encoder ----
            have, ready = 0, self._prefetched
            if self.direct and ready is not None and frame_id is not None and ready["frame_id"] == frame_id and ready["parity"] == parity and not (
                    ready["image"].data_ptr() == reference_image.data_ptr() and ready["image_version"] == reference_image._version and
                    ready["image"].device == reference_image.device and tuple(ready["image"].stride()) == tuple(reference_image.stride())):
                # the announced frame came as ANOTHER tensor (re-materialised, modified): its prepared stages are not taken (say so)
                self.lookahead_rejected += 1
                if self.lookahead_rejected == 1:
                    import warnings
                    warnings.warn("DepthEngine.step: the frame announced by next_reference_image arrived as a different (or modified) tensor; its "
                                  "prepared features are recomputed.  Hand step() the very tensor that was announced to keep the look-ahead.", RuntimeWarning)
            # (a frame id alone does not identify an image: the prepared features are taken only for the very tensor that was announced --
            # same storage, unmodified since; the engine keeps the announced tensor alive, so the address cannot have been reused)
            if self.direct and frame_id is not None and ready is not None and ready["frame_id"] == frame_id and ready["parity"] == parity and \
                    ready["image"].data_ptr() == reference_image.data_ptr() and ready["image_version"] == reference_image._version and \
                    ready["image"].device == reference_image.device and tuple(ready["image"].stride()) == tuple(reference_image.stride()):
                have = 1
                if ready["level"] == 2 and ready["measurement_ids"] == list(measurement_ids) and len(ready["measurement_poses"]) == n_meas and \
                        torch.equal(ready["full_K"], _pose_algebra.to_host(full_K).reshape(-1, 3, 3)) and \
                        torch.equal(ready["pose"], _pose_algebra.to_host(reference_pose).reshape(-1, 4, 4)) and \
                        all(torch.equal(a, _pose_algebra.to_host(b).reshape(-1, 4, 4)) for a, b in zip(ready["measurement_poses"], measurement_poses)):
                    have = 2
            if have >= 1 and self.cache_features:
                self._remember(frame_id, cur["ref_half_nhwc"] if self.sweep_mfma else cur["enc_cat"][0][:, :32])     # (the next frame may use this one as a measurement frame)

            # ---- this frame's measurement features (not needed when its sweep already ran) ----
            if have < 2:
                # resolve every measurement frame BEFORE anything is inserted into the cache: an insertion may evict the least
                # recently used entry, which could be a frame this very call still needs
                fresh, target = [], (cur["meas_feat"] if self.direct else s["meas_feat"])
                for i in range(n_meas):
                    img = measurement_images[i] if measurement_images is not None else None
                    mid = measurement_ids[i]
                    if self.cache_features and mid is not None and mid in self._feature_cache:
                        self._feature_cache.move_to_end(mid)
                        half = self._feature_cache[mid]
                    elif img is None:
                        raise ValueError(f"measurement frame {mid} is not cached and no image was given")
                    else:
                        half = self._features(img)[0].contiguous()
                        fresh.append((mid, half))
                    self._copy(target[i], half)
                for mid, half in fresh:
                    self._remember(mid, half)

            # ---- how much of the next frame this call computes ----
            give, next_frame, n_meas_next = 0, None, 0
            if self.direct and next_reference_image is not None and self.max_lookahead >= 1:
                if tuple(next_reference_image.shape) != tuple(reference_image.shape):
                    raise ValueError("next_reference_image must have the reference image's shape")
                give = 1
                if self.max_lookahead >= 2 and next_reference_pose is not None and next_measurement_poses is not None and next_measurement_ids is not None and \
                        next_frame_id is not None and self.pose_algebra == "reference" and self.cache_features and \
                        1 <= len(next_measurement_poses) <= _MAX_MEAS and len(next_measurement_ids) == len(next_measurement_poses) and \
                        all(m is not None and (m in self._feature_cache or (m == frame_id and have >= 1)) for m in next_measurement_ids):
                    give, n_meas_next = 2, len(next_measurement_poses)
                    self._allocate_static(n_meas_next)
                    next_frame = (next_reference_pose, list(next_measurement_poses))
                    for i, mid in enumerate(next_measurement_ids):
                        self._copy(sets[1 - parity]["meas_feat"][i], self._feature_cache[mid])
            if self.direct:
                if have < 1:
                    self._copy(cur["full_in"][:, 33:36], reference_image)
                if give >= 1:
                    self._copy(sets[1 - parity]["full_in"][:, 33:36], next_reference_image)
                s["image"], s["ref_half"] = cur["full_in"][:, 33:36], cur["enc_cat"][0][:, :32]
            else:
                self._copy(s["image"], reference_image)
        except BaseException:
            try:      # (what was queued belongs to cache entries that are already registered: write it before the error leaves)
                self._flush_copies()
            finally:
                self._copy_queue = None
            raise
        pending = self._copy_queue      # (launched together with the parameter block: _upload_frame_parameters)
        self._copy_queue = None
        mark("inputs copied")
        committed_pose, sweep_variant, next_variant = self._upload_frame_parameters(n_meas, reference_pose, measurement_poses, full_K, index=parity,
                                                                                    own_sweep=have < 2, next_frame=next_frame, pending_copies=pending)
        mark("parameters planned + uploaded")
        if self.direct and give < 2 and self.pose_algebra == "reference" and self.plan_frames_ahead and next_reference_pose is not None and \
                next_measurement_poses is not None and 1 <= len(next_measurement_poses) <= _MAX_MEAS:
            # the announced next frame's parameter block, evaluated on the planning thread while this thread launches the frame
            self.plan_ahead(len(next_measurement_poses), next_reference_pose, list(next_measurement_poses), full_K, committed_pose, 1 - parity)
        if have == 2:
            sweep_variant = ready["sweep_variant"]
        self.sweep_variant_counts[sweep_variant] = self.sweep_variant_counts.get(sweep_variant, 0) + 1

        # S > 1: always the previous-state path (see the class docstring); S == 1: the reference's two frame kinds
        kind = (n_meas, (self.has_previous or self.sequences > 1) and self.is_fusionnet)

        def graph_key(par, hv, gv, variant, variant_next):
            # (stages that do not run in this graph do not tell graphs apart)
            return (n_meas if hv < 2 else 0, kind[1], variant if hv < 2 else 0, par, hv, gv, n_meas_next if gv == 2 else 0, variant_next if gv == 2 else 0)

        key = graph_key(parity, have, give, sweep_variant, next_variant)
        body = (n_meas, kind[1], sweep_variant, parity, have, give, n_meas_next, next_variant or 0)
        if not self.use_graphs:
            self._frame_body(*body)
        elif kind not in self._warm:
            # first occurrence of this kind of frame: run eagerly (lets MIOpen pick its solvers, times the fusion plans); state
            # buffers are updated by the body, so this is a real frame, not a throw-away
            t_eager = time.perf_counter()
            self._frame_body(*body)
            self._warm.add(kind)
            torch.cuda.synchronize(self.device)
            self.warmup_seconds["eager_first_frames"] += time.perf_counter() - t_eager
        else:
            known = set(self._graphs)
            if key not in self._graphs:
                # Capture records launches, it executes nothing -- so what this kind of frame will need later is captured now, while the
                # caller is still warming up: with look-ahead the steady-state pattern (this frame's share prefetched, the next frame's
                # being prefetched) for both buffer sets and both sweep configurations, without it both configurations.  A later frame
                # whose geometry asks for the other configuration finds its graph ready instead of paying ~0.1 s of capture mid-run.
                t_capture = time.perf_counter()
                self._graphs[key] = self._capture(body)
                if give:
                    tiled = self._tiled_variants      # the sweep's configurations x (two passes, one pass) [+ the MFMA sweep]: whichever a later geometry asks for
                    choices = tiled if next_variant in tiled or sweep_variant in tiled else (sweep_variant,)
                    for par in (0, 1):
                        for v in (choices if give < 2 else (0,)):
                            for vn in (choices if give == 2 else (0,)):
                                k = graph_key(par, give, give, v, vn)
                                if k not in self._graphs:
                                    self._graphs[k] = self._capture((n_meas, kind[1], v, par, give, give, n_meas_next, vn))
                elif sweep_variant in self._tiled_variants and have < 2:
                    for par in ((0, 1) if self.direct else (0,)):
                        for v in self._tiled_variants:
                            k = graph_key(par, have, 0, v, 0)
                            if k not in self._graphs:
                                self._graphs[k] = self._capture((n_meas, kind[1], v, par, have, 0, 0, 0))
                self.warmup_seconds["graph_capture"] += time.perf_counter() - t_capture
            # (the graph of THIS frame, when it is new, is warmed like the ones captured ahead: its throw-away launches are undone the same way)
            fresh = [k for k in self._graphs if k not in known]
            if fresh and self.warm_captured_graphs:
                # A graph's FIRST launch stalls the device for ~5 ms (its kernel arguments and code are set up then, not at capture):
                # once per pre-captured graph, i.e. a few steps into every run, whenever a geometry first asks for another sweep
                # configuration or buffer set.  So every graph captured ahead is launched once now, during warm-up, on throw-away
                # results: the recurrent state is put back afterwards, and everything else the launch writes is rewritten by the real
                # frame below (same image, same features) before anybody reads it.
                # One exception: a frame whose sweep + encoder ran a frame ahead (have == 2) does NOT recompute them, and a fresh graph of the
                # other parity runs ITS look-ahead stage -- other measurement count, other sweep configuration, the work list of another
                # geometry -- on this frame's buffer set.  Those buffers are saved and put back as well.
                t_warm = time.perf_counter()
                keep = [s[k].clone() for k in ("h", "c", "prev_depth")] if self.is_fusionnet else []
                keep_set = self._snapshot(cur) if (self.direct and have == 2) else None
                for k in fresh:      # back to back per graph: a graph's next launch is queued while its previous one still runs, as in the steady
                    for _ in range(self.warm_graph_launches):      # state, where the host is up to two frames ahead of the device
                        self._graphs[k].replay()
                for name, saved in zip(("h", "c", "prev_depth"), keep):
                    s[name].copy_(saved)
                if keep_set is not None:
                    self._restore(cur, keep_set)
                if self.direct:      # (the splat's invariant: a frame finds its estimate buffer all-zero; the throw-away launches wrote into both)
                    for buffer_set in sets:
                        buffer_set["estimate"].zero_()
                torch.cuda.synchronize(self.device)
                self.warmup_seconds["graph_first_launches"] += time.perf_counter() - t_warm
                self.warmup_graphs_launched += len(fresh)
            self._graphs[key].replay()
        mark("frame launched")
        self._prev_pose_host = committed_pose
        self._no_previous[:] = False
        self.has_previous = True
        if self.direct:
            self._prefetched = None
            if give >= 1 and next_frame_id is not None:
                self._prefetched = dict(frame_id=next_frame_id, parity=1 - parity, level=give, sweep_variant=next_variant,
                                        image=next_reference_image, image_version=next_reference_image._version)
                if give == 2:
                    to_host = _pose_algebra.to_host
                    self._prefetched.update(pose=to_host(next_reference_pose).reshape(-1, 4, 4).clone(), measurement_ids=list(next_measurement_ids),
                                            full_K=to_host(full_K).reshape(-1, 3, 3).clone(),      # (the sweep ahead used THIS call's intrinsics)
                                            measurement_poses=[to_host(p).reshape(-1, 4, 4).clone() for p in next_measurement_poses])
            self._parity = 1 - parity
        if self.cache_features and frame_id is not None and have < 1:
            self._remember(frame_id, cur["ref_half_nhwc"] if (self.direct and self.sweep_mfma) else s["ref_half"])
        mark("done")
        return s["depth"]

    @staticmethod
    def _snapshot(tree):
        """Clones every tensor of a (nested dict / list of) buffer set."""
        if isinstance(tree, torch.Tensor):
            return tree.clone()
        if isinstance(tree, dict):
            return {k: DepthEngine._snapshot(v) for k, v in tree.items()}
        if isinstance(tree, (list, tuple)):
            return [DepthEngine._snapshot(v) for v in tree]
        return None

    @staticmethod
    def _restore(tree, saved):
        if isinstance(tree, torch.Tensor):
            tree.copy_(saved)
        elif isinstance(tree, dict):
            for k, v in tree.items():
                DepthEngine._restore(v, saved[k])
        elif isinstance(tree, (list, tuple)):
            for v, w in zip(tree, saved):
                DepthEngine._restore(v, w)

    def graph_memory_report(self):
        """Captured frame graphs and the device memory of their private pools (bytes; None where the runtime does not say)."""
        pools = None
        try:
            stats = torch.cuda.memory_stats(self.device)
            pools = int(stats.get("reserved_bytes.all.current", 0)) - int(stats.get("allocated_bytes.all.current", 0))
        except Exception:
            pass
        return {"graphs": len(self._graphs), "graphs_launched_once_at_warmup": self.warmup_graphs_launched, "queue_filler_graphs": len(self._filler_graphs),
                "reserved_minus_allocated_bytes": pools, "warmup_seconds": {k: round(v, 4) for k, v in self.warmup_seconds.items()}}

    def refresh_weights(self):
        """Call after loading new parameters into this engine's modules (``engine.fe`` ... ``engine.dec``): the MFMA convolution kernels
        read re-packed copies of the weights, made at first use.  They are re-packed IN PLACE, so captured graphs stay valid."""
        for m in (self.fe, self.fs, self.enc, self.lstm, self.dec):
            for sub in ([] if m is None else m.modules()):
                if isinstance(sub, FusedConv2d):
                    if sub._bottleneck_packed is not None:
                        sub._bottleneck_packed.copy_(_ops.bottleneck_conv_pack(sub.weight.detach()))
                    for tile, packed in sub._direct_packed.items():
                        packed.copy_(_ops.direct_conv_pack(sub.weight.detach(), tile))
                    if sub._pointwise_packed is not None:
                        sub._pointwise_packed.copy_(_ops.pointwise_conv_pack(sub.weight.detach()))
        if self._lstm_packed is not None:
            self._lstm_packed.copy_(_ops.bottleneck_conv_pack(self.lstm.lstm_cell.conv.weight.detach()))

    def _capture(self, key):
        """Captures the frame body for ``key`` into a hipGraph.  Capture only records the launches (nothing executes, no
        state buffer changes); the caller replays the graph to actually run the frame."""
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        if getattr(self, "graph_debug", False):     # bench.py counts the kernel nodes of the frame graph from its dot dump
            graph.enable_debug_mode()
        with torch.cuda.graph(graph):
            self._frame_body(*key)
        if self.direct and key[5] and key[4] >= key[5]:      # a body with a second branch (see _frame_body_direct)
            for _ in range(_GRAPH_QUEUE_FILLERS):
                self._capture_queue_filler()
        return graph

    def _capture_queue_filler(self):
        """A two-branch graph of two tiny kernels, instantiated and never launched (see _GRAPH_QUEUE_FILLERS)."""
        if self._filler_buffers is None:
            self._filler_buffers = torch.zeros(2, 64, device=self.device)
        filler = torch.cuda.CUDAGraph()
        with torch.cuda.graph(filler):
            main = torch.cuda.current_stream(self.device)
            self._side_stream.wait_stream(main)
            with torch.cuda.stream(self._side_stream):
                self._filler_buffers[0].add_(1.0)
            self._filler_buffers[1].add_(1.0)
            main.wait_stream(self._side_stream)
        self._filler_graphs.append(filler)
