"""`CLIPBindWrap`: the `visual` tower of a TriCLIP model as a stand-alone point-cloud encoder (reference:
VitLens-OpenShape/src/models/clip_bind.py:9-101).  backbone = model.visual (point tokenizer -> Perceiver -> ViT);
when the tower's own projection does not end in `args.model.out_channel` it is dropped and a trainable
Linear(width, out_channel) takes its place (clip_bind.py:35-47); `lock` is the tower's lock recipe plus the class token
when layers are skipped or --unlock_cls is given (:58-97); `forward(x, **kwargs)` = proj_layer(backbone(x, **kwargs)),
called as `model(features, xyz=xyz)` by train.py:214."""
import math

import numpy as np
import torch
import torch.nn as nn


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the HIP GEMM, with its backward (the optional output projection of CLIPBindWrap)."""

    @staticmethod
    def forward(ctx, x, w, b):
        from vitlens_hip import ops
        xb, wb = ops.cast_bf16(x.contiguous().float()), ops.cast_bf16(w.contiguous().float())
        ctx.save_for_backward(xb, wb)
        return ops.gemm(xb, wb, b.float().contiguous(), epi=ops.EPI_F32)

    @staticmethod
    def backward(ctx, dy):
        from vitlens_hip import ops
        xb, wb = ctx.saved_tensors
        dyb = ops.cast_bf16(dy.contiguous().float())
        rows = (dyb.shape[0] + 63) // 64 * 64
        dx = ops.gemm(dyb, ops.transpose_to_bf16(wb, ldo=wb.shape[0]), None, epi=ops.EPI_F32)             # dy . W
        dw = torch.zeros(wb.shape, device=dy.device, dtype=torch.float32)
        ops.gemm_dw(ops.transpose_to_bf16(dyb, ldo=rows), ops.transpose_to_bf16(xb, ldo=rows), dw)      # dy^T x
        return dx, dw, dy.float().sum(0)


class _OutProj(nn.Module):
    """nn.Linear(width, out_channel) with nn.Linear's parameter names and initialisation."""

    def __init__(self, width, out_channel):
        super().__init__()
        bound = 1.0 / math.sqrt(width)
        self.weight = nn.Parameter((torch.rand(out_channel, width) * 2 - 1) * bound)
        self.bias = nn.Parameter((torch.rand(out_channel) * 2 - 1) * bound)

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias)


class CLIPBindWrap(nn.Module):
    def __init__(self, args, model=None):
        """args: the OpenShape config namespace (clip_model, pretrained, precision, force_image_size, cache_dir, model.out_channel,
        skip_trans_first_n_layers, unlock_cls + the Lens arguments).  `model`: an already built TriCLIP (tests)."""
        super().__init__()
        self.args = args
        if model is None:
            from open_clip import tri_create_model_and_transforms
            model, _, _ = tri_create_model_and_transforms(
                args.clip_model, args.pretrained, precision=args.precision, device="cpu", jit=False,
                force_quick_gelu=getattr(args, "force_quick_gelu", False), force_custom_text=getattr(args, "force_custom_text", False),
                force_patch_dropout=False, force_image_size=getattr(args, "force_image_size", None),
                pretrained_image=getattr(args, "pretrained_image", False), output_dict=True,
                cache_dir=getattr(args, "cache_dir", None), args=args)
        self.backbone = model.visual
        self.proj_layer = nn.Identity()
        out_channel = args.model.out_channel
        if self.backbone.embed_dim != out_channel:
            # the tower's own projection goes; features leave the trunk at transformer width
            self.backbone.drop_output_projection()
            self.proj_layer = _OutProj(self.backbone.cfg.width, out_channel)

    def lock(self, unlocked_groups=0, freeze_bn_stats=False, unlock_cls=False, unlock_trans_first_n_layers=None):
        self.backbone.lock(unlocked_groups=unlocked_groups, freeze_bn_stats=freeze_bn_stats, unlock_cls=unlock_cls,
                           unlock_trans_first_n_layers=unlock_trans_first_n_layers)
        skip = getattr(self.args, "skip_trans_first_n_layers", None)
        if (skip is not None and skip > 0) or getattr(self.args, "unlock_cls", False):
            self.backbone.class_embedding.requires_grad = True       # skipped layers: the class token is trained (clip_bind.py:69-80)

    def forward(self, x, **kwargs):
        return self.proj_layer(self.backbone(x, **kwargs))


class LogitScaleNetwork(nn.Module):
    """models/LogitScaleNetwork.py: a learnable log-temperature, returned exponentiated."""

    def __init__(self, init_scale=1 / 0.07):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(init_scale))

    def forward(self, x=None):
        return self.logit_scale.exp()
