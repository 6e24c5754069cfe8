"""Epoch loop of the training scripts (surface of the reference's ``dvmvs.train``: /root/reference/dvmvs/train.py:10-148).

``train`` runs one epoch over ``train_loader`` with the script's ``forward_pass_function`` (for fusionnet:
``dvmvs.training.forward_pass``, whose hot-path ops are the HIP kernels), optionally validates and writes the staged
checkpoints the reference writes (one file per module + the optimiser state, named by step and the four validation
losses).  Differences in HOW: progress / TensorBoard / image grids are optional (``summary_writer`` may be ``None``;
tqdm is used when importable) and the debug image grid is assembled with plain torch ops instead of torchvision; an
optional ``gradient_reducer`` (``dvmvs.training.BucketedGradientReducer``) turns the step into the data-parallel step of
BASELINE.json configs[4] -- with one process per GPU every rank runs this same loop on its own shard of the loader.
"""
import torch

from dvmvs.config import Config
from dvmvs.losses import LossMeter
from dvmvs.utils import freeze_batchnorm, save_checkpoint, save_optimizer

METER_NAMES = ("L1", "Huber", "L1-inv", "L1-rel")


def switch_mode(model, mode):
    """'train': every module in train mode, batch-norm layers frozen again when Config asks for it; 'eval': eval mode."""
    if mode == "train":
        for module in model:
            module.train()
            if Config.train_freeze_batch_normalization:
                module.apply(freeze_batchnorm)
    elif mode == "eval":
        for module in model:
            module.eval()


def _progress(iterable):
    try:
        from tqdm import tqdm
        return tqdm(iterable)
    except ImportError:
        return iterable


def _debug_grid(images):
    """[3,H,W] images, each rescaled to [0,1], three per row (what torchvision.utils.make_grid(nrow=3, normalize=True,
    scale_each=True) is used for in the reference)."""
    tiles = []
    for image in images:
        lo, hi = image.min(), image.max()
        tiles.append((image - lo) / (hi - lo).clamp(min=1e-5))
    while len(tiles) % 3:
        tiles.append(torch.zeros_like(tiles[0]))
    rows = [torch.cat(tiles[i:i + 3], dim=2) for i in range(0, len(tiles), 3)]
    return torch.cat(rows, dim=1)


def _log_debug_images(summary_writer, images, depths, predictions, predictions_names, step):
    tiles = [images[-1][0].detach().cpu(), depths[-1][0].detach().cpu().repeat(3, 1, 1)]
    names = "input_image   ground_truth"
    for name, prediction in zip(predictions_names, predictions):
        names += "   " + name
        prediction = prediction[0].detach().cpu().repeat(3, 1, 1).unsqueeze(0)
        scale = Config.train_image_width / prediction.shape[-1]
        tiles.append(torch.nn.functional.interpolate(prediction, scale_factor=scale, mode="bilinear", align_corners=True).squeeze(0))
    summary_writer.add_image(names, _debug_grid(tiles), step)


def _run_epoch(loader, model, forward_pass_function, is_training, on_batch=None):
    meters = [LossMeter() for _ in METER_NAMES]
    for i, (images, depths, poses, K) in enumerate(_progress(loader)):
        l1, huber, l1_inv, l1_rel, optimizer_loss, predictions, names = forward_pass_function(
            images=images, depths=depths, poses=poses, K=K, model=model, is_training=is_training)
        for meter, batch in zip(meters, (l1, huber, l1_inv, l1_rel)):
            meter.update(loss=batch.sum, count=batch.count)
        if on_batch is not None:
            on_batch(i, images, depths, optimizer_loss, predictions, names, meters)
    return meters


def train(train_loader, val_loader, model, optimizer, summary_writer, epoch, best_loss, run_directory, forward_pass_function,
          gradient_reducer=None):
    switch_mode(model=model, mode="train")
    steps_per_epoch = len(train_loader)

    def step(i, images, depths, optimizer_loss, predictions, names, meters):
        global_step = epoch * steps_per_epoch + i
        if summary_writer is not None and i > 0 and i % Config.train_print_frequency == 0:
            _log_debug_images(summary_writer, images, depths, predictions, names, global_step)
        if gradient_reducer is not None:   # flat, view-backed gradient buckets: all-reduced while backward is still running
            gradient_reducer.zero_grad()
        else:
            optimizer.zero_grad()
        optimizer_loss.backward()
        if gradient_reducer is not None:
            gradient_reducer.finish()
        optimizer.step()
        if summary_writer is not None:
            for name, meter in zip(METER_NAMES, meters):
                summary_writer.add_scalar("Batch Loss/" + name, meter.item_average, global_step)

    training = _run_epoch(train_loader, model, forward_pass_function, True, on_batch=step)

    if Config.train_validate:
        # data-parallel run (one process per GPU, gradient_reducer): every rank validates its own shard, the meters are summed
        # over the ranks, so all ranks see the SAME validation losses, take the same "best so far" decision, and only rank 0
        # writes the checkpoint files into the shared run directory
        validation = validate(val_loader=val_loader, model=model, forward_pass_function=forward_pass_function,
                              reduce_over_ranks=gradient_reducer is not None)
        writer_rank = not (gradient_reducer is not None and _distributed() and torch.distributed.get_rank() != 0)
        end_step = (epoch + 1) * steps_per_epoch
        if summary_writer is not None:
            for name, meter, value in zip(METER_NAMES, training, validation):
                summary_writer.add_scalar(name + " Loss/Training", meter.avg, end_step)
                summary_writer.add_scalar(name + " Loss/Validation", value, end_step)
        if any(value < best for value, best in zip(validation, best_loss)):
            for k, value in enumerate(validation):
                best_loss[k] = min(value, best_loss[k])
            l1, huber, l1_inv, l1_rel = validation
            if writer_rank:
                named = [{"name": "module_" + str(k), "epoch": epoch + 1, "state_dict": module.state_dict()} for k, module in enumerate(model)]
                save_checkpoint(run_directory, named, step=end_step, loss=[l1, l1_inv, l1_rel, huber])
                save_optimizer(run_directory, optimizer=optimizer, step=end_step, loss=[l1, l1_inv, l1_rel, huber])
        switch_mode(model=model, mode="train")   # validation left the modules in eval mode
    return [meter.avg for meter in training]


def _distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def validate(val_loader, model, forward_pass_function, reduce_over_ranks=False):
    """Running means (L1, Huber, L1-inv, L1-rel) of the full-resolution prediction over ``val_loader``, modules in eval mode.
    With ``reduce_over_ranks`` (data-parallel training) sums and counts are all-reduced first: the means are those of the whole
    validation set on every rank."""
    switch_mode(model=model, mode="eval")
    with torch.no_grad():
        meters = _run_epoch(val_loader, model, forward_pass_function, False)
    if reduce_over_ranks and _distributed():
        device = next(model[0].parameters()).device
        if torch.distributed.get_backend() == "nccl" and device.type != "cuda":
            device = torch.device("cuda", torch.cuda.current_device())
        totals = torch.tensor([[float(m.sum), float(m.count)] for m in meters], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(totals, op=torch.distributed.ReduceOp.SUM)
        return tuple(float(s / c) if c > 0 else 0.0 for s, c in totals.tolist())
    return tuple(meter.avg for meter in meters)
