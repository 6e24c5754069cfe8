"""csrc/train_ops.hip on the GPU: gradients of the x2 bilinear up-sampler and of the MnasNet depthwise convolutions (gathers and a
fixed-order reduction, no atomics) against float64 autograd of the ATen ops on the CPU, and bit-reproducibility."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dvmvs.hip import ops as _ops
    return _ops


@pytest.mark.parametrize("shape", [(2, 3, 8, 10), (1, 5, 1, 7), (1, 2, 16, 1), (1, 4, 33, 47), (4, 64, 32, 32)])
def test_upsample2x_forward_and_gradient(ops, hip_device, shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    gout = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], generator=g)
    x64 = x.double().requires_grad_(True)
    y64 = F.interpolate(x64, scale_factor=2, mode="bilinear", align_corners=True)
    y64.backward(gout.double())
    xd = x.to(hip_device).requires_grad_(True)
    y = ops.upsample2x(xd)
    y.backward(gout.to(hip_device))
    # forward: ATen's own float32 arithmetic (the kernel restates it: the source position is the ROUNDED product scale * index, whose
    # integer part selects the taps) -- tight against float32 ATen, float32-position round-off (1e-5 on odd sizes) against float64
    y32 = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    assert float((y.detach().cpu() - y32).abs().max()) <= 1e-6
    assert float((y.detach().cpu().double() - y64.detach()).abs().max()) <= 3e-5
    err = float((xd.grad.cpu().double() - x64.grad).abs().max())
    print(f"upsample2x {shape}: max |grad - float64 autograd| {err:.2e}")
    assert err <= 5e-5 * max(1.0, float(x64.grad.abs().max()))
    again = ops.upsample2x_bwd(gout.to(hip_device))
    assert torch.equal(again, xd.grad)


@pytest.mark.parametrize("k,stride", [(3, 1), (3, 2), (5, 1), (5, 2)])
@pytest.mark.parametrize("shape", [(2, 8, 16, 20), (1, 3, 7, 9), (4, 48, 64, 64)])
def test_depthwise_forward_and_gradients(ops, hip_device, k, stride, shape):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(k * 10 + stride + C)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) / k
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, padding=k // 2, stride=stride, groups=C)
    gout = torch.randn(*y64.shape, generator=g)
    y64.backward(gout.double())
    xd, wd = x.to(hip_device).requires_grad_(True), w.to(hip_device).requires_grad_(True)
    y = ops.depthwise_conv_train(xd, wd, stride)
    assert tuple(y.shape) == tuple(y64.shape)
    y.backward(gout.to(hip_device))
    scale_w = max(1.0, float(w64.grad.abs().max()))
    e_y = float((y.detach().cpu().double() - y64.detach()).abs().max())
    e_x = float((xd.grad.cpu().double() - x64.grad).abs().max())
    e_w = float((wd.grad.cpu().double() - w64.grad).abs().max()) / scale_w
    print(f"depthwise k={k} s={stride} {shape}: forward {e_y:.2e}, grad_x {e_x:.2e}, grad_w (relative to its max) {e_w:.2e}")
    assert e_y <= 5e-6 and e_x <= 1e-5 and e_w <= 2e-5
    gx, gw = ops.depthwise_conv_bwd(gout.to(hip_device), xd.detach(), wd.detach(), stride, True, True)
    assert torch.equal(gx, xd.grad) and torch.equal(gw, wd.grad)          # bit-reproducible
    only_x, none_w = ops.depthwise_conv_bwd(gout.to(hip_device), xd.detach(), wd.detach(), stride, True, False)
    assert torch.equal(only_x, gx) and none_w.numel() == 0


def test_modules_use_the_hip_kernels_under_autograd(hip_device):
    """DepthwiseConv2d and the decoder's up-sampler run the HIP forward + backward when gradients are recorded, with the results of
    the plain ATen modules."""
    from dvmvs.backbone import DepthwiseConv2d
    from dvmvs.networks import _upsample2
    dev = hip_device
    torch.manual_seed(3)
    layer = DepthwiseConv2d(24, 24, 5, padding=2, stride=2, groups=24, bias=False).to(dev)
    x = torch.randn(2, 24, 32, 40, device=dev, requires_grad=True)
    y = layer(x)
    assert "DepthwiseConvTrain" in type(y.grad_fn).__name__ or "depthwise" in str(y.grad_fn).lower()
    ref = F.conv2d(x.detach().cpu().double(), layer.weight.detach().cpu().double(), padding=2, stride=2, groups=24)
    assert float((y.detach().cpu().double() - ref).abs().max()) <= 5e-6
    u = _upsample2(x)
    assert "upsample2x" in str(u.grad_fn).lower() or "Upsample2X" in type(u.grad_fn).__name__
    (y.sum() + u.sum()).backward()
    assert x.grad is not None and layer.weight.grad is not None and torch.isfinite(x.grad).all()
