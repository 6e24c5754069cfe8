"""GPU tests specific to the correlate-then-interpolate sweep (variant 6, csrc/sweep_mfma.hip) beyond the per-variant cases of
tests/test_hip_parity.py: both layouts of the measurement maps produce the same bits, the strip path (footprints larger than the dot
table: strong magnification, forward motion onto near planes), ragged sizes with 32 channels, every keyframe geometry class of the
sample scene against the reference-order kernel and the float64 oracle.  Reference: /root/reference/dvmvs/utils.py:45-107."""
import pytest
import torch

import dvmvs_oracle as orc
import hipcall
import synthetic as syn
from test_hip_parity import as_accurate_as_reference, f64, maxerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(hip_device):
    from dvmvs.hip import _capi, ops as o
    _capi.lib()
    return o


def both_layouts(ops, dev, f1, f2s, p1, p2s, K, D=64, lo=0.25, hi=20.0):
    f1d, f2d = f1.to(dev), [t.to(dev) for t in f2s]
    nchw = hipcall.cost_volume(ops, f1d, f2d, p1, p2s, K, lo, hi, D, True, 6)
    cl = [t.contiguous(memory_format=torch.channels_last) for t in f2d]
    nhwc = hipcall.cost_volume(ops, f1d, cl, p1, p2s, K, lo, hi, D, True, 6)
    return nchw, nhwc


@pytest.mark.parametrize("line", [0, 40, 99, 117, 141, 180, 202, 250])
def test_keyframe_geometries_both_layouts(ops, hip_device, line):
    """Easy sideways pairs, the wide-baseline and forward-motion lines, a pair whose sweep crosses Z = 0: the volume equals the generic
    (reference-order) kernel's to summation-order round-off, is as close to float64 as the float32 oracle is, and does not depend on the
    layout of the measurement maps (same operands into the same MFMAs: bit-identical)."""
    dev = hip_device
    halfK = syn.scaled_K(syn.full_K(), 2.0)
    r, ms = (141, (135, 130)) if line == 141 else syn.keyframe_index_lines(2)[line]
    f = [syn.smooth_noise((1, 32, 128, 160), seed=500 + i) for i in range(3)]
    p1, p2s = syn.pose(r), [syn.pose(m) for m in ms]
    nchw, nhwc = both_layouts(ops, dev, f[0], f[1:], p1, p2s, halfK)
    assert torch.equal(nchw, nhwc), line
    again = hipcall.cost_volume(ops, f[0].to(dev), [t.to(dev) for t in f[1:]], p1, p2s, halfK, 0.25, 20.0, 64, True, 6)
    assert torch.equal(nchw, again), line
    generic = hipcall.cost_volume(ops, f[0].to(dev), [t.to(dev) for t in f[1:]], p1, p2s, halfK, 0.25, 20.0, 64, True, 1)
    assert maxerr(nchw, generic) < 3e-5, line
    exp = orc.cost_volume_fusion(f[0], f[1:], p1, p2s, halfK, 0.25, 20.0, 64, True)
    exp64 = orc.cost_volume_fusion(*f64(f[0], f[1:], p1, p2s, halfK), 0.25, 20.0, 64, True)
    as_accurate_as_reference(nchw, exp, exp64, floor=1e-5)


@pytest.mark.parametrize("forward", [0.12, 0.2, 0.24])
def test_strong_magnification_uses_strips(ops, hip_device, forward):
    """The measurement camera `forward` metres ahead of the reference camera: on the nearest planes (0.25 m) a 4 x 4 pixel group spreads
    over 2-25 x its size, the box of even 4 planes exceeds the dot table and is processed in strips.  Same volume as the generic kernel."""
    dev = hip_device
    halfK = syn.scaled_K(syn.full_K(), 2.0)
    p1 = syn.pose(50)
    step = torch.eye(4).unsqueeze(0)
    step[0, 2, 3] = forward                     # camera-frame +z
    step[0, 0, 3] = 0.03
    p2 = p1 @ step                              # cam -> world of the camera moved along its own axes
    f = [syn.smooth_noise((1, 32, 128, 160), seed=600 + i) for i in range(2)]
    nchw, nhwc = both_layouts(ops, dev, f[0], f[1:], p1, [p2], halfK)
    assert torch.equal(nchw, nhwc)
    generic = hipcall.cost_volume(ops, f[0].to(dev), [f[1].to(dev)], p1, [p2], halfK, 0.25, 20.0, 64, True, 1)
    assert maxerr(nchw, generic) < 3e-5
    exp = orc.cost_volume_fusion(f[0], f[1:], p1, [p2], halfK, 0.25, 20.0, 64, True)
    exp64 = orc.cost_volume_fusion(*f64(f[0], f[1:], p1, [p2], halfK), 0.25, 20.0, 64, True)
    as_accurate_as_reference(nchw, exp, exp64, floor=1e-5)


@pytest.mark.parametrize("shape", [(2, 32, 33, 47, 10, 3), (1, 32, 61, 83, 37, 2), (1, 32, 128, 160, 19, 1), (2, 24, 30, 42, 16, 2), (1, 32, 256, 320, 16, 1),
                                   (1, 32, 130, 162, 61, 3), (1, 28, 124, 157, 130, 2), (1, 32, 128, 160, 64, 8)])
def test_ragged_sizes(ops, hip_device, shape):
    """Image sizes that are not multiples of the 4 x 4 pixel group, plane counts that are not multiples of 16, batches with their own
    poses, a channel count below 32 (zero-padded operands), the full-resolution size (boxes of more than 2^16 cells possible).  The last four
    shapes run the persistent form (round 6: >= 4096 work items, one batch item): group rows that are no multiple of 16 (XCD row sets with slots
    beyond the image), ragged plane counts, 9 chunks per group, the maximum of 8 measurement frames."""
    dev = hip_device
    B, C, H, W, D, M = shape
    g = torch.Generator().manual_seed(B * 1000 + C * 100 + H)
    f1 = torch.randn(B, C, H, W, generator=g)
    f2s = [torch.randn(B, C, H, W, generator=g) for _ in range(M)]
    ids = [9, 10, 13, 16, 20, 141]
    p1 = torch.cat([syn.pose(ids[b % 6]) for b in range(B)])
    p2s = [torch.cat([syn.pose(ids[(b + 1 + m) % 6] - 3) for b in range(B)]) for m in range(M)]
    K = torch.cat([syn.scaled_K(syn.full_K(), 320.0 / W) for b in range(B)])
    nchw, nhwc = both_layouts(ops, dev, f1, f2s, p1, p2s, K, D=D)
    assert torch.equal(nchw, nhwc), shape
    exp = orc.cost_volume_fusion(f1, f2s, p1, p2s, K, 0.25, 20.0, D, True)
    exp64 = orc.cost_volume_fusion(*f64(f1, f2s, p1, p2s, K), 0.25, 20.0, D, True)
    assert maxerr(nchw, exp) < 5e-4 * max(1.0, exp.abs().max().item()), shape      # white-noise features (see test_hip_parity)
    as_accurate_as_reference(nchw, exp, exp64)


@pytest.mark.parametrize("case", [(0, 128, 160, 64, 2), (118, 128, 160, 64, 2), (170, 128, 160, 64, 2), (202, 128, 160, 64, 1), (40, 132, 172, 50, 3), (9, 256, 320, 64, 2)])
def test_persistent_form_is_bit_identical_to_one_item_per_workgroup(ops, hip_device, case):
    """Variant 6 on these shapes is ONE persistent 16-wave workgroup per CU: own items by enumeration, the tail as quarter items (4 of an item's 16
    planes) pulled through an LDS queue.  Which wave computes what must not change a bit: equal to variant 7 (one item per workgroup, no quarters)
    on easy, wide-baseline, forward-motion and ragged geometries, both layouts, every output element written (NaN-prefilled destination)."""
    dev = hip_device
    line, H, W, D, M = case
    K = syn.scaled_K(syn.full_K(), 320.0 / W)
    r, ms = syn.keyframe_index_lines(2)[line]
    ms = (list(ms) + [r - 5])[:M]
    f = [syn.smooth_noise((1, 32, H, W), seed=700 + i).to(dev) for i in range(M + 1)]
    p1, p2s = syn.pose(r), [syn.pose(m) for m in ms]
    from dvmvs import pose_algebra
    Hm, kt = pose_algebra.sweep_matrices(p1, p2s, K, dev, "reference")
    for meas in (f[1:], [t.contiguous(memory_format=torch.channels_last) for t in f[1:]]):
        outs = []
        for variant in (6, 7):
            dst = torch.full((1, D, H, W), float("nan"), device=dev)
            ops.cost_volume_into(f[0], meas, Hm, kt, 0.25, 20.0, dst, variant)
            assert not torch.isnan(dst).any(), (case, variant)
            outs.append(dst)
        assert torch.equal(outs[0], outs[1]), case
    generic = hipcall.cost_volume(ops, f[0], f[1:], p1, p2s, K, 0.25, 20.0, D, True, 1)
    assert maxerr(outs[0], generic) < 3e-5, case


def test_nchw_to_nhwc_and_copy_batch(ops, hip_device):
    """The two small data-movement entries of the frame path: dvmvs_nchw_to_nhwc (a keyframe's features into the channels-last feature cache)
    equals torch's channels-last conversion bit for bit, ragged pixel counts and batches included; dvmvs_copy_batch (a step's input copies as
    one launch) equals copy_ for contiguous and channels-last pairs of different sizes, and refuses what it cannot copy flat."""
    dev = hip_device
    g = torch.Generator().manual_seed(5)
    for shape in [(1, 32, 128, 160), (2, 32, 33, 47), (1, 64, 16, 20), (3, 4, 5, 7)]:
        src = torch.randn(*shape, generator=g).to(dev)
        dst = torch.zeros(shape, device=dev).contiguous(memory_format=torch.channels_last)
        ops.nchw_to_nhwc_into(src, dst)
        assert torch.equal(dst, src) and torch.equal(dst.permute(0, 2, 3, 1).contiguous(), src.permute(0, 2, 3, 1).contiguous()), shape
    with pytest.raises((ValueError, RuntimeError)):
        ops.nchw_to_nhwc_into(torch.zeros(1, 6, 4, 4, device=dev), torch.zeros(1, 6, 4, 4, device=dev).contiguous(memory_format=torch.channels_last))
    cat = torch.zeros(1, 36, 256, 320, device=dev)
    pairs = [(torch.zeros(1, 32, 128, 160, device=dev).contiguous(memory_format=torch.channels_last),
              torch.randn(1, 32, 128, 160, generator=g).to(dev).contiguous(memory_format=torch.channels_last)),
             (torch.zeros(1, 32, 128, 160, device=dev), torch.randn(1, 32, 128, 160, generator=g).to(dev)),
             (cat[:, 33:36], torch.randn(1, 3, 256, 320, generator=g).to(dev)),
             (torch.zeros(8, device=dev), torch.randn(8, generator=g).to(dev)),
             (torch.zeros(4, 1000, device=dev), torch.randn(4, 1000, generator=g).to(dev))]
    assert all(ops.batchable(d, s) for d, s in pairs)
    ops.copy_batch(pairs)
    for d, s in pairs:
        assert torch.equal(d, s)
    assert float(cat[:, :33].abs().max()) == 0.0                       # nothing beyond the slice was written
    assert not ops.batchable(torch.zeros(2, 36, 8, 8, device=dev)[:, 33:36], torch.zeros(2, 3, 8, 8, device=dev))      # strided batch slice
    assert not ops.batchable(torch.zeros(6, device=dev), torch.zeros(6, device=dev))                                        # not a multiple of 4
    assert not ops.batchable(torch.zeros(9, device=dev)[1:], torch.zeros(8, device=dev))                                    # misaligned
    assert not ops.batchable(torch.zeros(1, 32, 8, 8, device=dev), torch.zeros(1, 32, 8, 8, device=dev).contiguous(memory_format=torch.channels_last))
    with pytest.raises(RuntimeError):
        ops.copy_batch([(torch.zeros(9, device=dev)[1:], torch.zeros(8, device=dev))])
    # round 6: a source may be pinned host memory the device sees at the same address (the engine's parameter block rides in the frame's batch)
    from dvmvs.hip import _capi
    pinned, pageable = torch.randn(1024, generator=g).pin_memory(), torch.randn(1024, generator=g)
    with torch.cuda.device(dev):
        assert _capi.lib().dvmvs_host_pointer_device_visible(pinned.data_ptr()) == 1
        assert _capi.lib().dvmvs_host_pointer_device_visible(pageable.data_ptr()) == 0
        assert _capi.lib().dvmvs_host_pointer_device_visible(None) == 0
    up, other = torch.zeros(1024, device=dev), torch.zeros(8, device=dev)
    ops.copy_batch([(other, torch.ones(8, device=dev)), (up, pinned)])
    torch.cuda.synchronize()
    assert torch.equal(up.cpu(), pinned) and float(other.min()) == 1.0
