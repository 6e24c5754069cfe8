"""csrc/bottleneck_conv.hip on the GPU: the 3x3 layers of the 8x10 / 16x20 maps and the ConvLSTM convolution as a weight-streaming
fp32 MFMA GEMM with a deterministic split-K, against torch's convolution (float64 arbitration), bit-reproducibility, the consumers
that add the partial sums (epilogue, ConvLSTM gates), and the engine with / without it."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synthetic as syn

pytestmark = pytest.mark.gpu

SHAPES = [  # (C_in, C_out, H, W, stride): the layers of a 320x256 frame the kernel takes, + a ragged channel count
    (1024, 2048, 8, 10, 1), (512, 512, 8, 10, 1), (256, 512, 16, 20, 2), (512, 256, 16, 20, 1), (288, 256, 16, 20, 1),
    (256, 256, 16, 20, 1), (64, 40, 8, 10, 1), (32, 1, 16, 20, 1), (128, 256, 32, 40, 2), (48, 20, 32, 40, 2)]


@pytest.fixture(scope="module")
def ops():
    from dvmvs.hip import ops as _ops
    return _ops


def problem(C_in, C_out, H, W, B, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C_in, H, W, generator=g)
    w = torch.randn(C_out, C_in, 3, 3, generator=g) / (3.0 * C_in ** 0.5)       # asymmetric: a transposed operand cannot pass
    bias = torch.randn(C_out, generator=g)
    return x, w, bias


@pytest.mark.parametrize("channels", [(512, 256), (64, 40)])
@pytest.mark.parametrize("B", [1, 2])
def test_upsampling_inside_the_staging_gives_the_bits_of_the_two_launches(ops, hip_device, channels, B):
    """dvmvs_bottleneck_conv_up2x_fwd (round 6): the decoder's first up-convolution reads the 8x10 map and interpolates the 16x20 one while it
    stages its input -- the same partial sums, bit for bit, as dvmvs_upsample2x_fwd followed by dvmvs_bottleneck_conv_fwd."""
    C_in, C_out = channels
    x, w, _ = problem(C_in, C_out, 8, 10, B, seed=7 + C_in)
    x = x.to(hip_device)
    packed = ops.bottleneck_conv_pack(w.to(hip_device))
    S = ops.bottleneck_conv_splits(B, C_out, C_in, 16, 20, 1)
    two, one = (torch.full((S * B * C_out * 320,), float("nan"), device=hip_device) for _ in range(2))
    assert ops.bottleneck_conv_into(ops.upsample2x(x), packed, C_out, 1, two) == S
    assert ops.bottleneck_conv_into(x, packed, C_out, 1, one, upsample=True) == S
    assert not torch.isnan(one).any() and torch.equal(one, two)
    with pytest.raises(ValueError):
        ops.bottleneck_conv_into(torch.zeros(B, C_in, 16, 20, device=hip_device), packed, C_out, 1, one, upsample=True)      # only the 8x10 -> 16x20 layer


def test_split_count_respects_the_lds_budget(ops):
    """A large batch needs few splits for its wave count, but a split's slice of x must still fit the 64 KB a workgroup stages it in
    (round 4, first form: batch 8 of the ConvLSTM layer asked for 2 splits = 512 channels x 120 padded pixels = 240 KB and the launch
    was refused -- found by bench.py's 8-sequence leg)."""
    for B in (1, 2, 4, 8, 16):
        for (C_in, C_out, H, W, stride) in SHAPES:
            S = ops.bottleneck_conv_splits(B, C_out, C_in, H, W, stride)
            assert S >= 1 and (C_in // 16) % S == 0
            staged_rows = 9 if H == 32 else H + 2      # (the 32 x 40 stride-2 layer stages the nine rows a pixel group reads)
            assert (C_in // S) * staged_rows * (W + 2) * 4 <= 64 * 1024, (B, C_in, S)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("B", [1, 2, 8])
def test_partial_sums_add_up_to_the_convolution(ops, hip_device, shape, B):
    C_in, C_out, H, W, stride = shape
    dev = hip_device
    x, w, bias = problem(C_in, C_out, H, W, B, seed=C_in + C_out + B)
    exact = F.conv2d(x.double(), w.double(), padding=1, stride=stride)
    torch32 = F.conv2d(x, w, padding=1, stride=stride)
    Ho, Wo = H // stride, W // stride
    S = ops.bottleneck_conv_splits(B, C_out, C_in, H, W, stride)
    assert S >= 1 and C_in % S == 0
    packed = ops.bottleneck_conv_pack(w.to(dev))
    partials = torch.full((S * B * C_out * Ho * Wo,), float("nan"), device=dev)
    assert ops.bottleneck_conv_into(x.to(dev), packed, C_out, stride, partials) == S
    parts = partials.view(S, B, C_out, Ho, Wo)
    got = parts[0].clone()
    for s in range(1, S):
        got += parts[s]
    err = float((got.cpu().double() - exact).abs().max())
    ref_err = float((torch32.double() - exact).abs().max())
    print(f"{shape} B={B}: {S} splits, max |kernel - float64| {err:.2e} (torch fp32 convolution: {ref_err:.2e})")
    assert torch.isfinite(got).all()
    assert err <= 3.0 * ref_err + 2e-6            # as accurate as an fp32 convolution is
    # each split is the convolution over its own input channels
    cs = C_in // S
    for s in (0, S - 1):
        part = F.conv2d(x[:, s * cs:(s + 1) * cs].double(), w[:, s * cs:(s + 1) * cs].double(), padding=1, stride=stride)
        assert float((parts[s].cpu().double() - part).abs().max()) <= 3.0 * ref_err + 2e-6, s
    # bit-reproducible
    again = torch.empty_like(partials)
    for _ in range(3):
        again.fill_(float("nan"))
        ops.bottleneck_conv_into(x.to(dev), packed, C_out, stride, again)
        assert torch.equal(again, partials)
    # epilogue: partial sums + bias + ReLU into a channel slice of a larger buffer
    cat = torch.zeros(B, C_out + 5, Ho, Wo, device=dev)
    if B == 1:
        ops.partial_sums_bias_act_into(partials, S, cat[:, 3:3 + C_out], bias.to(dev), ops.ACTIVATIONS["relu"], (B, C_out, Ho, Wo))
        expect = torch.relu(got + bias.to(dev).view(1, -1, 1, 1))
        assert torch.equal(cat[:, 3:3 + C_out], expect)
        assert float(cat[:, :3].abs().max()) == 0.0 and float(cat[:, 3 + C_out:].abs().max()) == 0.0
    dense = torch.empty(B, C_out, Ho, Wo, device=dev)
    ops.partial_sums_bias_act_into(partials, S, dense, None, ops.ACTIVATIONS["none"], (B, C_out, Ho, Wo))
    assert torch.equal(dense, got)


def test_unsupported_shapes_are_refused(ops, hip_device):
    assert ops.bottleneck_conv_splits(1, 64, 64, 32, 40, 1) == 0          # larger maps stay on MIOpen
    assert ops.bottleneck_conv_splits(1, 64, 24, 8, 10, 1) == 0           # C_in % 16 != 0
    assert ops.bottleneck_conv_splits(1, 64, 64, 8, 10, 2) == 0
    with pytest.raises(ValueError):
        ops.bottleneck_conv_pack(torch.zeros(8, 24, 3, 3, device=hip_device))
    with pytest.raises(ValueError):
        ops.bottleneck_conv_pack(torch.zeros(8, 32, 5, 5, device=hip_device))
    with pytest.raises(RuntimeError):
        ops.bottleneck_conv_pack(torch.zeros(8, 32, 3, 3))                 # no CPU path


def test_gates_on_partial_sums_equal_gates_on_the_sum(ops, hip_device):
    dev = hip_device
    g = torch.Generator().manual_seed(5)
    S = 16
    parts = (torch.randn(S, 1, 2048, 8, 10, generator=g) * 0.3).to(dev)
    c0 = torch.randn(1, 512, 8, 10, generator=g).to(dev)
    total = parts[0].clone()
    for s in range(1, S):
        total += parts[s]                          # the kernel's order: ascending, one fp32 addition per split
    h_a, c_a = torch.zeros_like(c0), c0.clone()
    ops.lstm_gates_into(total, c_a, h_a)
    h_b, c_b = torch.zeros_like(c0), c0.clone()
    ops.lstm_gates_partials_into(parts.reshape(-1), S, c_b, h_b)
    assert torch.equal(h_a, h_b) and torch.equal(c_a, c_b)


def test_engine_with_bottleneck_kernels_equals_the_all_miopen_engine(hip_device):
    """Same frames through DepthEngine(bottleneck_convs=True) -- the default -- and (bottleneck_convs=False): eleven layers change from
    MIOpen to the MFMA kernel, i.e. to another fp32 summation order: depth within 2e-5 rel-L1, and the default engine repeats bit for bit."""
    from dvmvs.engine import DepthEngine
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    dev = hip_device
    mods = syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder))
    ours = DepthEngine(*mods, device=dev, use_graphs=True)
    again = DepthEngine(*mods, device=dev, use_graphs=True)
    miopen = DepthEngine(*mods, device=dev, use_graphs=True, bottleneck_convs=False)
    assert ours.bottleneck_convs and not miopen.bottleneck_convs
    fullK = syn.full_K()
    frames = list(syn.E2E_FRAMES) + [(12, (11, 9)), (13, (12, 10))]
    for n, (r, ms) in enumerate(frames):
        args = (syn.e2e_image(r).to(dev), syn.pose(r), [syn.e2e_image(i).to(dev) for i in ms], [syn.pose(i) for i in ms], fullK)
        if n > 0:      # every engine starts the frame from the same state, so that a flipped z-buffer pixel cannot blur the comparison
            state = ours.state()
            again.load_state(*state)
            miopen.load_state(*state)
        a = ours.step(*args, frame_id=r, measurement_ids=list(ms)).clone()
        b = miopen.step(*args, frame_id=r, measurement_ids=list(ms)).clone()
        c = again.step(*args, frame_id=r, measurement_ids=list(ms)).clone()
        err = float(((a - b).abs() / b).mean())
        print(f"frame {n}: bottleneck kernels vs all-MIOpen engine, depth rel-L1 {err:.3e}; repeat identical: {torch.equal(a, c)}")
        assert err <= 2e-5, (n, err)
        assert torch.equal(a, c), n
    used = [m for mod in (ours.enc, ours.dec) for m in mod.modules() if getattr(m, "_bottleneck_packed", None) is not None]
    print(f"{len(used)} encoder / decoder layers run through the bottleneck kernel, ConvLSTM convolution: {ours._lstm_packed is not None}")
    assert len(used) >= 6 and ours._lstm_packed is not None
