"""GPU parity of the direct convolution kernel and the depth-head kernel (csrc/direct_conv.hip) against an fp64 convolution: the
layer shapes of a 320x256 frame (all four workgroup shapes, both kernel sizes, both strides), ragged channel counts, batches,
channel-slice destinations, and that shapes the kernel does not take are reported as such."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (C_in, H, W, C_out, k, stride): one per workgroup shape and (k, stride) in use, plus ragged channel counts
SHAPES = [
    (36, 256, 320, 32, 5, 1), (32, 256, 320, 32, 5, 1), (3, 256, 320, 32, 3, 2),
    (96, 128, 160, 32, 5, 1), (65, 128, 160, 32, 5, 1), (32, 128, 160, 32, 3, 1), (32, 128, 160, 64, 5, 2),
    (64, 64, 80, 64, 5, 1), (129, 64, 80, 64, 3, 1), (96, 64, 80, 64, 3, 1), (64, 64, 80, 128, 3, 2),
    (128, 32, 40, 128, 3, 1), (257, 32, 40, 128, 3, 1), (5, 32, 40, 16, 5, 1), (7, 16, 80, 48, 3, 1),
]


def _ops():
    from dvmvs.hip import ops
    return ops


def _reference(x, w, bias, stride, act):
    y = F.conv2d(x.double(), w.double(), None if bias is None else bias.double(), stride=stride, padding=w.shape[-1] // 2)
    return (torch.relu(y) if act else y).float()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("batch", [1, 2])
def test_direct_conv_matches_fp64(shape, batch):
    ops = _ops()
    C_in, H, W, C_out, k, stride = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C_in * 7 + k)
    x = torch.randn(batch, C_in, H, W, generator=g).to(dev)
    w = (torch.randn(C_out, C_in, k, k, generator=g) / (C_in * k * k) ** 0.5).to(dev)
    bias = torch.randn(C_out, generator=g).to(dev)
    n_tile = ops.direct_conv_tile(batch, C_in, H, W, C_out, k, stride)
    assert n_tile in (1, 2)
    packed = ops.direct_conv_pack(w, n_tile)
    for act, b in ((1, bias), (0, None)):
        dst = torch.full((batch, C_out, H // stride, W // stride), float("nan"), device=dev)
        ops.direct_conv_into(x, packed, n_tile, b, dst, C_out, k, stride, act)
        want = _reference(x, w, b, stride, act)
        assert float((dst - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))     # fp32 sums of <= 6 425 terms of O(1/sqrt(n))


def test_direct_conv_writes_a_channel_slice_and_is_deterministic():
    ops = _ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 96, 64, 80, generator=g).to(dev)
    w = (torch.randn(64, 96, 3, 3, generator=g) / 30).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    n_tile = ops.direct_conv_tile(2, 96, 64, 80, 64, 3, 1)
    packed = ops.direct_conv_pack(w, n_tile)
    cat = torch.full((2, 64 + 64 + 1, 64, 80), 7.0, device=dev)
    ops.direct_conv_into(x, packed, n_tile, bias, cat[:, 64:128], 64, 3, 1, 1)
    want = _reference(x, w, bias, 1, 1)
    assert float((cat[:, 64:128] - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert bool((cat[:, :64] == 7.0).all()) and bool((cat[:, 128:] == 7.0).all())
    again = torch.empty(2, 64, 64, 80, device=dev)
    ops.direct_conv_into(x, packed, n_tile, bias, again, 64, 3, 1, 1)
    assert torch.equal(again, cat[:, 64:128])


@pytest.mark.parametrize("shape", [(32, 128, 160, 32, 3, 1), (96, 128, 160, 32, 5, 1), (64, 64, 80, 128, 3, 2), (36, 256, 320, 32, 5, 1), (7, 16, 80, 48, 3, 1)])
def test_second_channels_last_destination_holds_the_same_bits(shape):
    """dvmvs_direct_conv_dual_fwd (ABI 7): the epilogue writes its values a second time, channels-last -- how a keyframe's half-resolution FPN
    features reach the MFMA sweep without a transposing launch.  Equal bit for bit to the NCHW output (a channel slice of a concatenation
    buffer, as in the engine), batches included; nothing else of the slice's buffer is touched; a destination of the wrong layout is refused."""
    ops = _ops()
    C_in, H, W, C_out, k, stride = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C_in * 13 + k)
    for batch in (1, 2):
        x = torch.randn(batch, C_in, H, W, generator=g).to(dev)
        w = (torch.randn(C_out, C_in, k, k, generator=g) / (C_in * k * k) ** 0.5).to(dev)
        bias = torch.randn(C_out, generator=g).to(dev)
        n_tile = ops.direct_conv_tile(batch, C_in, H, W, C_out, k, stride)
        packed = ops.direct_conv_pack(w, n_tile)
        OH, OW = H // stride, W // stride
        cat = torch.full((batch, C_out + 5, OH, OW), 7.0, device=dev)
        plain = torch.empty(batch, C_out, OH, OW, device=dev)
        nhwc = torch.full((batch, C_out, OH, OW), float("nan"), device=dev).contiguous(memory_format=torch.channels_last)
        ops.direct_conv_into(x, packed, n_tile, bias, plain, C_out, k, stride, 1)
        if batch == 1:
            ops.direct_conv_into(x, packed, n_tile, bias, cat[:, :C_out], C_out, k, stride, 1, dst_nhwc=nhwc)
            assert torch.equal(cat[:, :C_out], plain) and float((cat[:, C_out:] - 7.0).abs().max()) == 0.0
        else:
            both = torch.empty_like(plain)
            ops.direct_conv_into(x, packed, n_tile, bias, both, C_out, k, stride, 1, dst_nhwc=nhwc)
            assert torch.equal(both, plain)
        assert torch.equal(nhwc, plain)                                                     # same values ...
        assert torch.equal(nhwc.permute(0, 2, 3, 1).contiguous(), plain.permute(0, 2, 3, 1).contiguous())      # ... in channels-last storage
    with pytest.raises(ValueError):
        ops.direct_conv_into(x, packed, n_tile, bias, plain, C_out, k, stride, 1, dst_nhwc=torch.empty_like(plain))


def test_problems_the_kernel_does_not_take():
    ops = _ops()
    assert ops.direct_conv_tile(1, 512, 16, 20, 256, 3, 1) == 0      # 20 columns: the bottleneck kernel's maps
    assert ops.direct_conv_tile(1, 32, 64, 80, 1, 3, 1) == 0         # one output channel: dvmvs_conv_head_fwd
    assert ops.direct_conv_tile(1, 32, 64, 80, 32, 7, 1) == 0
    assert ops.direct_conv_tile(1, 32, 63, 80, 32, 3, 2) == 0
    dev = torch.device("cuda:0")
    w = torch.randn(32, 8, 3, 3, device=dev)
    tile = ops.direct_conv_tile(1, 8, 32, 40, 32, 3, 1)
    assert tile in (1, 2)
    packed = ops.direct_conv_pack(w, 3 - tile)
    x, dst = torch.randn(1, 8, 32, 40, device=dev), torch.empty(1, 32, 32, 40, device=dev)
    with pytest.raises(RuntimeError):
        ops.direct_conv_into(x, packed, 3 - tile, None, dst, 32, 3, 1, 0)   # packed for the other tile count


@pytest.mark.parametrize("shape", [(256, 16, 20), (128, 32, 40), (64, 64, 80), (32, 128, 160), (32, 256, 320), (5, 7, 9)])
@pytest.mark.parametrize("batch", [1, 2])
def test_conv_head_matches_fp64(shape, batch):
    ops = _ops()
    C, H, W = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(batch, C, H, W, generator=g).to(dev)
    w = (torch.randn(1, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(dev)
    bias = torch.randn(1, generator=g).to(dev)
    raw = F.conv2d(x.double(), w.double(), padding=1)
    dst = torch.empty(batch, 1, H, W, device=dev)
    ops.conv_head_into(x, w, None, dst, 0)
    assert float((dst - raw.float()).abs().max()) <= 1e-5
    ops.conv_head_into(x, w, bias, dst, ops.ACTIVATIONS["sigmoid"])
    assert float((dst - torch.sigmoid(raw + bias.double()).float()).abs().max()) <= 1e-6
    p0, p1 = 3.95, 0.05
    ops.conv_head_into(x, w, bias, dst, ops.ACTIVATION_SIGMOID_TO_DEPTH, p0, p1)
    want = 1.0 / (p0 * torch.sigmoid(raw + bias.double()) + p1)
    assert float(((dst - want.float()).abs() / want.float()).max()) <= 1e-5


def test_engine_with_direct_convolutions_equals_the_miopen_engine():
    """Same frames through DepthEngine(direct_convs=True) -- the default -- and (direct_convs=False): ~30 layers change from MIOpen's
    Winograd / GEMM kernels + epilogue launch to the direct MFMA convolution, i.e. to another fp32 summation order (and away from the
    F(2,3) transforms' extra round-off): depth within 1e-4 rel-L1 frame by frame from the same recurrent state (measured 5e-5: the size
    of either engine's own distance to the reference, tests/test_e2e_gpu.py), and the default engine repeats bit for bit."""
    import synthetic as syn
    from dvmvs.engine import DepthEngine, FusedConv2d
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    dev = torch.device("cuda:0")
    mods = syn.build_e2e_modules((FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder))
    ours = DepthEngine(*mods, device=dev, use_graphs=True)
    again = DepthEngine(*mods, device=dev, use_graphs=True)
    miopen = DepthEngine(*mods, device=dev, use_graphs=True, direct_convs=False)
    assert ours.direct_convs and not miopen.direct_convs
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
        print(f"frame {n}: direct convolutions vs MIOpen engine, depth rel-L1 {err:.3e}; repeat identical: {torch.equal(a, c)}")
        assert err <= 1e-4, (n, err)
        assert torch.equal(a, c), n
    layers = [m for mod in (ours.fe, ours.fs, ours.enc, ours.dec) for m in mod.modules() if isinstance(m, FusedConv2d)]
    taken = [m for m in layers if m._direct_packed]
    heads = [m for m in layers if m.weight.shape[0] == 1]
    print(f"{len(taken)} layers run through the direct convolution kernel, {len(heads)} depth heads through the head kernel")
    assert len(taken) >= 24 and len(heads) == 5
