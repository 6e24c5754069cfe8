"""GPU parity of the 1x1 convolution kernel (csrc/pointwise_conv.hip) against an fp64 convolution + the epilogue arithmetic of
``dvmvs_bias_act_fwd`` (act(conv + bias) + residual): the 1x1 layer shapes of a 320x256 frame (MnasNet expansion / projection layers, FPN
lateral layers: fusionnet/model.py:20-124), ragged pixel and channel counts, batches, channel-slice destinations, every split count, both
residual modes, and that problems the kernel does not take are reported as such."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (C_in, C_out, H, W)
FRAME_SHAPES = [
    (32, 16, 128, 160), (16, 48, 128, 160), (48, 24, 64, 80), (24, 72, 64, 80), (72, 24, 64, 80), (72, 40, 32, 40), (40, 120, 32, 40), (120, 40, 32, 40),
    (40, 240, 32, 40), (240, 80, 16, 20), (80, 480, 16, 20), (480, 80, 16, 20), (480, 96, 16, 20), (96, 576, 16, 20), (576, 96, 16, 20), (576, 192, 8, 10),
    (192, 1152, 8, 10), (1152, 192, 8, 10), (1152, 320, 8, 10), (320, 32, 8, 10), (96, 32, 16, 20), (40, 32, 32, 40), (24, 32, 64, 80), (16, 32, 128, 160),
]
RAGGED_SHAPES = [(20, 19, 6, 10), (4, 1, 2, 2), (68, 33, 5, 12), (36, 100, 3, 4), (260, 17, 7, 8)]


def _ops():
    from dvmvs.hip import ops
    return ops


def _reference(x, w, bias, act, residual=None, mode=0):
    y = F.conv2d(x.double(), w.double(), None if bias is None else bias.double())
    if act:
        y = torch.relu(y)
    if mode == 1:
        y = y + residual.double()
    if mode == 2:
        y = y + F.interpolate(residual.double(), scale_factor=2, mode="nearest")
    return y.float()


def _problem(C_in, C_out, H, W, batch, seed):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, C_in, H, W, generator=g).to(dev)
    w = (torch.randn(C_out, C_in, 1, 1, generator=g) / C_in ** 0.5).to(dev)
    bias = torch.randn(C_out, generator=g).to(dev)
    return x, w, bias


def _tolerance(want):
    return 2e-5 * max(1.0, float(want.abs().max()))      # fp32 sums of <= 1 152 terms of O(1 / sqrt(n))


@pytest.mark.parametrize("shape", FRAME_SHAPES + RAGGED_SHAPES)
@pytest.mark.parametrize("batch", [1, 2])
def test_pointwise_conv_matches_fp64(shape, batch):
    ops = _ops()
    C_in, C_out, H, W = shape
    x, w, bias = _problem(C_in, C_out, H, W, batch, C_in * 13 + C_out)
    assert ops.pointwise_conv_supported(batch, C_in, H, W, C_out, 1, 0)
    packed = ops.pointwise_conv_pack(w)
    for act, b in ((1, bias), (0, None), (0, bias)):
        dst = torch.full((batch, C_out, H, W), float("nan"), device=x.device)
        ops.pointwise_conv_into(x, packed, b, dst, C_out, act)
        want = _reference(x, w, b, act)
        assert not torch.isnan(dst).any()
        assert float((dst - want).abs().max()) <= _tolerance(want)


@pytest.mark.parametrize("shape", [(72, 24, 64, 80), (1152, 192, 8, 10), (120, 40, 32, 40), (68, 33, 5, 12)])
def test_residual_of_the_same_shape_is_added_after_the_activation(shape):
    ops = _ops()
    C_in, C_out, H, W = shape
    x, w, bias = _problem(C_in, C_out, H, W, 2, 5)
    residual = torch.randn(2, C_out, H, W, device=x.device)
    packed = ops.pointwise_conv_pack(w)
    for act in (0, 1):
        dst = torch.full((2, C_out, H, W), float("nan"), device=x.device)
        ops.pointwise_conv_into(x, packed, bias, dst, C_out, act, residual, 1)
        want = _reference(x, w, bias, act, residual, 1)
        assert float((dst - want).abs().max()) <= _tolerance(want)
        # the launch pair it replaces: library convolution, then dvmvs_bias_act_fwd -- the same epilogue arithmetic on a sum in another order
        two = ops.bias_act_into(F.conv2d(x, w), torch.empty_like(dst), bias, act, residual, 1)
        assert float((dst - two).abs().max()) <= _tolerance(want)


@pytest.mark.parametrize("shape", [(96, 32, 16, 20), (24, 32, 64, 80), (16, 32, 128, 160), (36, 20, 6, 8)])
def test_half_resolution_residual_is_nearest_upsampled(shape):
    ops = _ops()
    C_in, C_out, H, W = shape
    x, w, bias = _problem(C_in, C_out, H, W, 2, 9)
    residual = torch.randn(2, C_out, H // 2, W // 2, device=x.device)
    packed = ops.pointwise_conv_pack(w)
    dst = torch.full((2, C_out, H, W), float("nan"), device=x.device)
    ops.pointwise_conv_into(x, packed, bias, dst, C_out, 0, residual, 2)
    want = _reference(x, w, bias, 0, residual, 2)
    assert float((dst - want).abs().max()) <= _tolerance(want)


def test_every_split_count_gives_the_sum_and_a_launch_is_deterministic():
    ops = _ops()
    C_in, C_out, H, W = 576, 96, 16, 20
    x, w, bias = _problem(C_in, C_out, H, W, 1, 3)
    packed = ops.pointwise_conv_pack(w)
    want = _reference(x, w, bias, 1)
    for splits in range(0, 17):
        dst = torch.full((1, C_out, H, W), float("nan"), device=x.device)
        ops.pointwise_conv_into(x, packed, bias, dst, C_out, 1, splits=splits)
        assert float((dst - want).abs().max()) <= _tolerance(want), splits
        again = torch.empty_like(dst)
        ops.pointwise_conv_into(x, packed, bias, again, C_out, 1, splits=splits)
        assert torch.equal(dst, again), splits


def test_writes_a_channel_slice_of_a_concatenation_buffer_only():
    ops = _ops()
    C_in, C_out, H, W = 40, 24, 32, 40
    x, w, bias = _problem(C_in, C_out, H, W, 2, 21)
    packed = ops.pointwise_conv_pack(w)
    cat = torch.full((2, 8 + C_out + 3, H, W), 7.0, device=x.device)
    ops.pointwise_conv_into(x, packed, bias, cat[:, 8:8 + C_out], C_out, 1)
    want = _reference(x, w, bias, 1)
    assert float((cat[:, 8:8 + C_out] - want).abs().max()) <= _tolerance(want)
    assert bool((cat[:, :8] == 7.0).all()) and bool((cat[:, 8 + C_out:] == 7.0).all())


def test_non_finite_inputs_stay_in_their_own_pixels():
    """A pixel tile that hangs over the end of a plane reads the next plane's first pixels into rows it never stores, and channels beyond C_in
    are read as zeros through the buffer descriptor: a NaN in one pixel must reach that pixel's outputs only."""
    ops = _ops()
    C_in, C_out, H, W = 20, 19, 6, 10
    x, w, bias = _problem(C_in, C_out, H, W, 1, 2)
    x[0, 3, 0, 0] = float("nan")      # first pixel of a plane: what the hanging tile of the plane before reads
    packed = ops.pointwise_conv_pack(w)
    dst = torch.empty(1, C_out, H, W, device=x.device)
    ops.pointwise_conv_into(x, packed, bias, dst, C_out, 0)
    nan = torch.isnan(dst)
    assert bool(nan[0, :, 0, 0].all()) and int(nan.sum()) == C_out


def test_problems_the_kernel_does_not_take_are_reported():
    ops = _ops()
    assert not ops.pointwise_conv_supported(1, 30, 8, 10, 16, 0, 0)       # C_in % 4
    assert not ops.pointwise_conv_supported(1, 32, 3, 5, 16, 0, 0)        # H * W % 4
    assert not ops.pointwise_conv_supported(1, 32, 8, 10, 16, 2, 0)       # sigmoid
    assert not ops.pointwise_conv_supported(1, 32, 8, 10, 16, 0, 2)       # mode 2 needs W % 4 == 0
    assert ops.pointwise_conv_supported(1, 32, 8, 12, 16, 0, 2)
    x, w, bias = _problem(30, 16, 8, 10, 1, 1)
    packed = ops.pointwise_conv_pack(w)
    with pytest.raises(RuntimeError):
        ops.pointwise_conv_into(x, packed, bias, torch.empty(1, 16, 8, 10, device=x.device), 16, 0)
