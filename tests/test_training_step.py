"""Training path (BASELINE.json configs[4]): loss, bucketed gradient all-reduce (world_size 2 over gloo on CPU), and on
the GPU one full training step against the CPU oracle's autograd."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synthetic as syn


def test_inverse_depth_loss_definition():
    from dvmvs.training import inverse_depth_l1, multi_scale_loss
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(2, 32, 32, generator=g) * 4 + 0.5
    gt[:, :5, :7] = 0.0                                   # invalid pixels
    pred_full = torch.rand(2, 32, 32, generator=g) * 4 + 0.5
    pred_half = torch.rand(2, 16, 16, generator=g) * 4 + 0.5
    s, n = inverse_depth_l1(pred_half, gt)
    gt_half = gt[:, ::2, ::2]                              # nearest resize to half size picks even rows / columns
    valid = gt_half != 0
    assert int(n) == int(valid.sum())
    assert abs(float(s) - float((1 / gt_half[valid] - 1 / pred_half[valid]).abs().sum())) < 1e-4
    total = multi_scale_loss([pred_half, pred_full], gt)
    s2, n2 = inverse_depth_l1(pred_full, gt)
    assert abs(float(total) - float(s / n + s2 / n2)) < 1e-6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_net():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 1, 3, padding=1))


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(4, 3, 16, 16, generator=g), torch.randn(4, 1, 16, 16, generator=g)


def _ddp_worker(rank, world, port, out_dir):
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.path.insert(0, os.path.join(root, "deep-video-mvs_amd"))
    from dvmvs.training import BucketedGradientReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _make_net()
    reducer = BucketedGradientReducer(net.parameters(), bucket_bytes=1024)    # tiny buckets: several collectives
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x, y = _data(rank)
    for _ in range(2):
        reducer.zero_grad()
        loss = (net(x) - y).abs().mean()
        loss.backward()
        launched = reducer.launched_during_backward
        reducer.finish()
        grads = [p.grad.clone() for p in net.parameters()]
        opt.step()
    if rank == 0:
        torch.save({"grads": grads, "params": [p.detach().clone() for p in net.parameters()], "buckets": len(reducer.buckets),
                    "launched": launched}, os.path.join(out_dir, "rank0.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_all_reduce_equals_full_batch_gradient(tmp_path):
    world = 2
    mp.spawn(_ddp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=False)
    assert res["buckets"] >= 2 and res["launched"] == res["buckets"]          # every bucket was reduced from a backward hook
    # single-process replica: mean of the two ranks' losses = data-parallel average of gradients
    net = _make_net()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    for _ in range(2):
        opt.zero_grad()
        loss = sum((net(x) - y).abs().mean() for x, y in (_data(0), _data(1))) / world
        loss.backward()
        grads = [p.grad.clone() for p in net.parameters()]
        opt.step()
    for a, b in zip(res["grads"], grads):
        assert (a - b).abs().max().item() < 1e-6
    for a, b in zip(res["params"], net.parameters()):
        assert (a - b).abs().max().item() < 1e-6


@pytest.mark.gpu
def test_gpu_training_step_matches_cpu_oracle_autograd(hip_device):
    """3-frame sub-sequence at 64x64, B=2: loss and gradients of the HIP path (cost-volume / hidden-warp / gate backward
    kernels + MIOpen) against autograd through the CPU oracle."""
    from fusionnet_cpu import cpu_subsequence_loss
    from dvmvs.config import Config
    from dvmvs.fusionnet.model import CostVolumeDecoder, CostVolumeEncoder, FeatureExtractor, FeatureShrinker, LSTMFusion
    from dvmvs.training import BucketedGradientReducer, fusionnet_subsequence_loss, train_step
    dev = hip_device
    ctors = (FeatureExtractor, FeatureShrinker, CostVolumeEncoder, LSTMFusion, CostVolumeDecoder)
    # seeded weights with calibrated, frozen BN statistics (Config.train_freeze_batch_normalization): O(1) activations, and no
    # batch-norm noise from a batch of two in the comparison
    cpu_model = syn.build_e2e_modules(ctors)
    gpu_model = [m.to(dev) for m in syn.build_e2e_modules(ctors)]
    B, H, W = 2, 64, 64
    g = torch.Generator().manual_seed(5)
    images = [syn.smooth_noise((B, 3, H, W), seed=300 + i) for i in range(3)]
    depths = [torch.rand(B, H, W, generator=g) * 3 + 0.7 for _ in range(3)]
    depths[1][:, :8, :8] = 0.0
    poses = [torch.cat([syn.pose(9 + i), syn.pose(20 + i)]) for i in range(3)]
    K = torch.cat([syn.full_K(width=W, height=H)] * B)
    loss_cpu = cpu_subsequence_loss(cpu_model, images, depths, poses, K, Config.train_min_depth, Config.train_max_depth,
                                    Config.train_n_depth_levels)
    loss_cpu.backward()
    params = [p for m in gpu_model for p in m.parameters()]
    reducer = BucketedGradientReducer(params)
    reducer.zero_grad()
    loss_gpu, preds = fusionnet_subsequence_loss(gpu_model, [t.to(dev) for t in images], [t.to(dev) for t in depths],
                                                 [t.to(dev) for t in poses], K.to(dev))
    loss_gpu.backward()
    reducer.finish()
    assert abs(loss_gpu.item() - loss_cpu.item()) <= 2e-3 * abs(loss_cpu.item())
    checked = 0
    for mc, mg in zip(cpu_model, gpu_model):
        for (name, pc), (_, pg) in zip(mc.named_parameters(), mg.named_parameters()):
            if pc.grad is None:
                continue
            scale = pc.grad.abs().max().item()
            if scale < 1e-8:
                continue
            # |.| in the loss and ~50 ReLU layers make the gradient piecewise constant in the activations: fp32 differences
            # between MIOpen and oneDNN flip a few gates (isolated entries move by a few %), so the comparison is direction (cosine) + relative L2 error
            a, b = pg.grad.cpu().double().flatten(), pc.grad.double().flatten()
            cosine = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
            rel_l2 = float((a - b).norm() / (b.norm() + 1e-30))
            assert cosine >= 0.999 and rel_l2 <= 5e-2, (type(mc).__name__, name, cosine, rel_l2, scale)
            checked += 1
    assert checked > 100
    # and one optimiser step runs end to end
    opt = torch.optim.Adam(params, lr=1e-4)
    before = params[-1].detach().clone()
    out = train_step(gpu_model, opt, reducer, [t.to(dev) for t in images], [t.to(dev) for t in depths], [t.to(dev) for t in poses], K.to(dev))
    assert torch.isfinite(out) and not torch.equal(before, params[-1].detach())
