"""InfoNCE losses + the feature all-gather, same call surface as the reference's open_clip/loss.py
(`gather_features` :20-78, `TriClipLoss` :81-165, `ClipLossGeneral` :234-308, `ClipLoss` :311-385).

What differs underneath:
  * the logits GEMM, row/column log-sum-exp, the loss reduction and dL/dlogits run in the HIP kernels
    of libvitlens_hip (vl_gemm_bf16 / vl_ce_*), wrapped in ONE autograd node per modality pair; the
    B x B logits matrix is written once and never transposed or re-materialised;
  * `gather_features` moves both feature sets with ONE collective (a packed [b, 2*D] all-gather over
    RCCL/xGMI on GPUs) instead of two, with identical semantics: rank-major concatenation, peers carry
    no gradient unless `gather_with_grad` (then the backward is a reduce-scatter), and the local slice is
    re-inserted so it stays differentiable when `local_loss` is off.
There is no CPU implementation of the loss math (GPU tensors required); `gather_features` itself is
device-agnostic host logic and is covered by gloo tests on CPU.
"""
import torch
import torch.nn as nn

try:
    import torch.distributed as dist
    has_distributed = dist.is_available()
except ImportError:  # pragma: no cover
    dist = None
    has_distributed = False


# ------------------------------------------------------------------------------------------------ gather
def _all_gather_cat(t: torch.Tensor, world_size: int) -> torch.Tensor:
    """Rank-major concatenation of `t` from every rank (no autograd)."""
    t = t.contiguous()
    out = torch.empty((world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if t.is_cuda:
        dist.all_gather_into_tensor(out, t)
    else:  # gloo
        dist.all_gather(list(out.chunk(world_size, dim=0)), t)
    return out


class _AllGatherWithGrad(torch.autograd.Function):
    """all_gather whose backward returns the SUM over ranks of the gradient of this rank's slice
    (= torch.distributed.nn.all_gather semantics, reference loss.py:55-61)."""

    @staticmethod
    def forward(ctx, t, world_size, rank):
        ctx.world_size, ctx.rank, ctx.rows = world_size, rank, t.shape[0]
        return _all_gather_cat(t, world_size)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if g.is_cuda:
            out = torch.empty((ctx.rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
        else:  # gloo has no reduce_scatter
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            out = g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].clone()
        return out, None, None


def gather_features(image_features, text_features, local_loss=False, gather_with_grad=False, rank=0,
                    world_size=1, use_horovod=False):
    """Same contract as the reference: returns (all_image_features, all_text_features), each
    [world_size*b, D] in rank-major order."""
    assert has_distributed, "torch.distributed did not import correctly"
    if use_horovod:
        raise NotImplementedError("horovod is not part of the MI355X path (RCCL via torch.distributed)")
    return tuple(gather_packed([image_features, text_features], local_loss, gather_with_grad, rank, world_size))


def gather_packed(feats, local_loss=False, gather_with_grad=False, rank=0, world_size=1):
    """All-gather any number of [b, D_i] feature sets with ONE collective on the packed [b, sum D_i]
    buffer (tri-modal step: image|text|visual = [b, 2304] -> a single latency-bound exchange)."""
    widths = [f.shape[1] for f in feats]
    packed = torch.cat(list(feats), dim=1)
    if gather_with_grad:
        allp = _AllGatherWithGrad.apply(packed, world_size, rank)
    else:
        with torch.no_grad():
            allp = _all_gather_cat(packed, world_size)
        if not local_loss:
            b = packed.shape[0]
            # re-insert the differentiable local slice (loss.py:71-74)
            allp = torch.cat([allp[:rank * b], packed, allp[(rank + 1) * b:]], dim=0)
    return list(allp.split(widths, dim=1))


# ------------------------------------------------------------------------------------------------ pair loss
class _ContrastivePair(torch.autograd.Function):
    """loss = w_row * CE(scale * X Y^T, r -> r+off) + w_col * CE(scale * Y X^T) as one autograd node (the same kernels
    and the same row-blocked mode as the fused training steps: vitlens_hip.step.pair_forward / pair_backward)."""

    @staticmethod
    def forward(ctx, x, y, logit_scale, label_off, w_row, w_col, chunk_rows):
        from vitlens_hip.step import pair_forward
        if not (x.is_cuda and y.is_cuda):
            raise RuntimeError("contrastive losses run on the GPU kernels only (no CPU fallback)")
        x = x.contiguous().float(); y = y.contiguous().float()
        loss, ctx.pair = pair_forward(x, y, float(logit_scale), label_off, w_row, w_col, chunk_rows=chunk_rows)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        from vitlens_hip.step import pair_backward
        dx, dy, dscale = pair_backward(ctx.pair, float(gout), need_dx=ctx.needs_input_grad[0], need_dy=ctx.needs_input_grad[1])
        ctx.pair = None
        return dx, dy, dscale.reshape(()), None, None, None, None


def contrastive_pair(x, y, logit_scale, label_off=0, w_row=0.5, w_col=0.5, chunk_rows=None):
    if not isinstance(logit_scale, torch.Tensor):
        logit_scale = torch.tensor(float(logit_scale), device=x.device)
    return _ContrastivePair.apply(x, y, logit_scale, int(label_off), float(w_row), float(w_col), chunk_rows)


class _LossBase(nn.Module):
    def __init__(self, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0, world_size=1,
                 use_horovod=False, chunk_rows=None):
        """chunk_rows (extension): rows per block of the logits (None = automatic: whole matrix up to 2^26 elements,
        2048-row blocks above; 0 = never block).  Results are the same up to fp32 summation order."""
        super().__init__()
        self.local_loss, self.gather_with_grad, self.cache_labels = local_loss, gather_with_grad, cache_labels
        self.rank, self.world_size, self.use_horovod = rank, world_size, use_horovod
        self.chunk_rows = chunk_rows

    def pair_loss(self, x, y, logit_scale, gathered=None):
        """(CE(logits_per_x) + CE(logits_per_y)) / 2 with the reference's gather / local_loss rules
        (get_logits + get_ground_truth, loss.py:103-138)."""
        if self.world_size > 1:
            all_x, all_y = gathered if gathered is not None else gather_features(
                x, y, self.local_loss, self.gather_with_grad, self.rank, self.world_size, self.use_horovod)
            if self.local_loss:
                off = x.shape[0] * self.rank
                return (contrastive_pair(x, all_y, logit_scale, off, 0.5, 0.0, self.chunk_rows)
                        + contrastive_pair(y, all_x, logit_scale, off, 0.5, 0.0, self.chunk_rows))
            return contrastive_pair(all_x, all_y, logit_scale, 0, 0.5, 0.5, self.chunk_rows)
        return contrastive_pair(x, y, logit_scale, 0, 0.5, 0.5, self.chunk_rows)


class ClipLoss(_LossBase):
    def forward(self, image_features, text_features, logit_scale, output_dict=False):
        total = self.pair_loss(image_features, text_features, logit_scale)
        return {"contrastive_loss": total} if output_dict else total


class ClipLossGeneral(_LossBase):
    def forward(self, x_features, y_features, logit_scale, output_dict=False, key="image-text"):
        total = self.pair_loss(x_features, y_features, logit_scale)
        return {key: total} if output_dict else total


class TriClipLoss(_LossBase):
    """(CE(IV) + CE(VI) + CE(TV) + CE(VT)) / 2   (reference loss.py:140-165)."""

    def forward(self, image_features, text_features, visual_features, logit_scale, output_dict=False):
        gi = gt = None
        if self.world_size > 1:   # one packed exchange instead of the reference's four all-gathers
            ai, at, av = gather_packed([image_features, text_features, visual_features], self.local_loss,
                                       self.gather_with_grad, self.rank, self.world_size)
            gi, gt = (ai, av), (at, av)
        total = (self.pair_loss(image_features, visual_features, logit_scale, gi)
                 + self.pair_loss(text_features, visual_features, logit_scale, gt))
        return {"contrastive_loss": total} if output_dict else total
