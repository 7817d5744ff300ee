"""The reference's per-epoch training drivers by name: `tri_train_one_epoch` (tri-modal, training/train.py:79-312) and
`train_dual_one_epoch` (modality <-> image-or-text, :315-560), so that a `main` written against the reference keeps
calling the same functions with the same arguments.

Both are the same loop around a different "forward + loss" of one batch, so they share one driver here:

    scheduler(step) -> zero_grad -> [accum_freq == 1: forward, loss, backward]
                                    [accum_freq  > 1: cache features of every micro-batch without a graph, then re-run each
                                     micro-batch WITH a graph against the cached features of the others and backward each time
                                     (train.py:154-210, 395-466)]
                    -> optional gradient clipping -> optimizer.step (through the GradScaler if one is given)
                    -> logit_scale clamp to [0, ln 100] -> log every `log_every_n_steps`

The model is the drop-in `TriCLIP` (differentiable towers on the HIP kernels) or anything with the same call surface;
precision is the towers' own business (bf16 GEMMs with fp32 accumulation), so `args.precision` selects no autocast
context here.  Distillation, label-mask losses, Horovod and the wandb / tensorboard sinks are outside the hot path and
raise / are ignored; the fastest way to run a recipe remains the fused step objects of `vitlens_hip.step`."""
import logging
import math
import time

import torch

from open_clip import get_input_dtype
from open_clip.utils import get_model


class AverageMeter(object):
    """Running value / mean of a logged quantity."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def unwrap_model(model):
    return model.module if hasattr(model, "module") else model


def backward(total_loss, scaler):
    (scaler.scale(total_loss) if scaler is not None else total_loss).backward()


def maybe_move_to_device(inp, device):
    return inp.to(device, non_blocking=True) if isinstance(inp, torch.Tensor) else inp


def _refuse(args):
    if getattr(args, "distill", False):
        raise NotImplementedError("distillation is outside the hot path (SURVEY section 2)")
    if getattr(args, "contra_loss_type", "general") != "general":
        raise NotImplementedError("label_mask / sim_mask losses are outside the hot path (SURVEY section 2)")
    if getattr(args, "horovod", False):
        raise NotImplementedError("Horovod is outside the hot path: one process per GPU on torch.distributed (RCCL)")


def _optimizer_step(model, optimizer, scaler, args):
    clip = getattr(args, "grad_clip_norm", None)
    if scaler is not None:
        if clip is not None:
            scaler.unscale_(optimizer)
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip, norm_type=2.0)
        scaler.step(optimizer)
        scaler.update()
    else:
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip, norm_type=2.0)
        optimizer.step()


def _run_epoch(model, data, epoch, optimizer, scaler, scheduler, args, fetch, features, loss_of):
    """fetch(batch) -> tuple of device inputs; features(inputs) -> ({name: feature tensor}, logit_scale);
    loss_of(feature dict, logit_scale) -> {name: loss tensor}."""
    _refuse(args)
    model.train()
    data["train"].set_epoch(epoch)
    dataloader = data["train"].dataloader
    accum = args.accum_freq
    per_epoch = dataloader.num_batches // accum
    digits = math.ceil(math.log(dataloader.num_samples + 1, 10))
    cached_inputs, cached = [], {}
    meters, batch_time, data_time = {}, AverageMeter(), AverageMeter()
    end = time.time()
    for i, batch in enumerate(dataloader):
        i_accum = i // accum
        step = per_epoch * epoch + i_accum
        if not args.skip_scheduler:
            scheduler(step)
        inputs = fetch(batch)
        data_time.update(time.time() - end)
        optimizer.zero_grad()
        if accum == 1:
            feats, logit_scale = features(inputs)
            losses = loss_of(feats, logit_scale)
            total = sum(losses.values())
            losses["loss"] = total
            backward(total, scaler)
        else:
            with torch.no_grad():
                feats, _ = features(inputs)
                for k, v in feats.items():
                    cached.setdefault(k, []).append(v)
            cached_inputs.append(inputs)
            if (i + 1) % accum:
                continue
            optimizer.zero_grad()
            for j in range(accum):
                feats, logit_scale = features(cached_inputs[j])
                merged = {k: torch.cat(v[:j] + [feats[k]] + v[j + 1:]) for k, v in cached.items()}
                losses = loss_of(merged, logit_scale)
                total = sum(losses.values())
                losses["loss"] = total
                backward(total, scaler)
            cached_inputs, cached = [], {}
        _optimizer_step(model, optimizer, scaler, args)
        with torch.no_grad():
            unwrap_model(model).logit_scale.clamp_(0, math.log(100))
        batch_time.update(time.time() - end)
        end = time.time()
        count = i_accum + 1
        if getattr(args, "rank", 0) == 0 and (i_accum % args.log_every_n_steps == 0 or count == per_epoch):
            bs = len(inputs[0])
            for k, v in losses.items():
                meters.setdefault(k, AverageMeter()).update(v.item(), bs)
            rate = accum * args.batch_size * args.world_size / batch_time.val
            logging.info(
                f"Train Epoch: {epoch} [{count * bs * accum * args.world_size:>{digits}}/{dataloader.num_samples} "
                f"({100.0 * count / per_epoch:.0f}%)] Data (t): {data_time.avg:.3f} Batch (t): {batch_time.avg:.3f}, {rate:#g}/s, "
                f"{rate / args.world_size:#g}/s/gpu LR: {optimizer.param_groups[0]['lr']:5f} Logit Scale: {logit_scale.item():.3f} "
                + " ".join(f"{k.capitalize()}: {m.val:#.5g} ({m.avg:#.5g})" for k, m in meters.items()))
            batch_time.reset(); data_time.reset()


def tri_train_one_epoch(model, data, loss, epoch, optimizer, scaler, scheduler, dist_model, args, tb_writer=None):
    device, dtype = torch.device(args.device), get_input_dtype(args.precision)

    def fetch(batch):
        return (batch["image"].to(device=device, dtype=dtype, non_blocking=True), batch["caption"].to(device=device, non_blocking=True),
                batch[args.v_key].to(device=device, non_blocking=True))

    def features(inputs):
        out = dict(model(*inputs))
        return out, out.pop("logit_scale")

    def loss_of(feats, logit_scale):
        return loss(**feats, logit_scale=logit_scale, output_dict=True)
    _run_epoch(model, data, epoch, optimizer, scaler, scheduler, args, fetch, features, loss_of)


def train_dual_one_epoch(model, data, loss, epoch, optimizer, scaler, scheduler, dist_model, args, tb_writer=None):
    device, dtype = torch.device(args.device), get_input_dtype(args.precision)
    to_image = args.align_to in ("image", "video")
    key = args.align_to if to_image else "caption"
    net = get_model(model)
    encode_anchor = net.encode_image if to_image else net.encode_text

    def fetch(batch):
        anchor = batch[key]
        anchor = anchor.to(device=device, dtype=dtype, non_blocking=True) if args.align_to in ("image", "text") else anchor.to(
            device=device, non_blocking=True)
        return anchor, batch[args.v_key].to(device=device, non_blocking=True)

    def features(inputs):
        feats = {"A_features": encode_anchor(inputs[0], normalize=True), "B_features": net.encode_visual(inputs[1], normalize=True)}
        return feats, net.logit_scale.exp()

    def loss_of(feats, logit_scale):
        return loss(feats["A_features"], feats["B_features"], logit_scale, output_dict=True, key=f"{args.align_to}-visual")
    _run_epoch(model, data, epoch, optimizer, scaler, scheduler, args, fetch, features, loss_of)
