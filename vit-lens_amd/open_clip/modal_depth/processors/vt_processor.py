"""Depth (disparity) preprocessing ON THE GPU (reference: open_clip/modal_depth/processors/
vt_processor.py:292-337 `DepthProcessorEval`, transforms_rgbd.py:366-411 `DepthNorm`).

The reference concatenates a throw-away random RGB image with the disparity map and sends the 4-channel tensor through
DepthNorm (clamp to [min_depth, max_depth], divide by max_depth) -> Resize(224, bicubic) -> CenterCrop(224) ->
Normalize, then keeps channel 3.  Only that channel is computed here: the clamp and the division are applied as the
source is read by the horizontal resampling pass, the vertical pass finishes with (x - depth_mean) / depth_std
(csrc/vl_preproc.hip, tables from vitlens_hip/preproc.py), and only the 224 x 224 crop window is produced.

torchvision's Resize of a TENSOR is torch.nn.functional.interpolate(mode="bicubic", align_corners=False,
antialias=...), whose default changed between torchvision releases (the reference does not pin one): `antialias=True`
(torchvision >= 0.17) is the default here, `antialias=False` reproduces the older behaviour.

Same class name, constructor arguments and call convention as the reference; the result is a [1, 224, 224] float32
tensor on the GPU, which is what the depth tokenizer takes."""
import numpy as np
import torch

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class BaseProcessor:
    def __init__(self):
        self.transform = lambda x: x

    def __call__(self, item):
        return self.transform(item)

    @classmethod
    def from_config(cls, cfg=None):
        return cls()

    def build(self, **kwargs):
        return self.from_config(dict(kwargs))


class DepthProcessorEval(BaseProcessor):
    def __init__(self, img_mean=None, img_std=None, depth_mean=0.0418, depth_std=0.0295, max_depth=75,
                 clamp_max_before_scale=True, min_depth=0.01, size=224, antialias=True, device="cuda"):
        img_mean = img_mean if img_mean is not None else OPENAI_CLIP_MEAN
        img_std = img_std if img_std is not None else OPENAI_CLIP_STD
        depth_mean = depth_mean if depth_mean is not None else 0.0
        depth_std = depth_std if depth_std is not None else 1.0
        if max_depth < 0.0:
            raise ValueError("max_depth must be > 0; got %.2f" % max_depth)
        self.mean = list(img_mean) + [depth_mean]
        self.std = list(img_std) + [depth_std]
        self.max_depth, self.min_depth, self.clamp_max_before_scale = float(max_depth), float(min_depth), clamp_max_before_scale
        self.size, self.antialias, self.device = size, antialias, torch.device(device)

    def __call__(self, depth, out=None):
        """depth: [H, W] or [1, H, W] disparity (tensor or numpy) -> [1, size, size] float32 on the GPU."""
        from vitlens_hip import preproc
        if isinstance(depth, np.ndarray):
            depth = torch.from_numpy(np.ascontiguousarray(depth))
        if depth.dim() == 3 and depth.shape[0] == 1:
            depth = depth[0]
        if depth.dim() != 2:
            raise ValueError(f"expected an [H, W] or [1, H, W] disparity map, got {tuple(depth.shape)}")
        d = depth.to(device=self.device, dtype=torch.float32, non_blocking=True)
        hi = self.max_depth if self.clamp_max_before_scale else float("inf")
        return preproc.depth_to_tensor(d, self.size, self.mean[3], self.std[3], clamp=(self.min_depth, hi, self.max_depth),
                                       antialias=self.antialias, out=out)

    def batch(self, depths):
        out = torch.empty(len(depths), 1, self.size, self.size, device=self.device, dtype=torch.float32)
        for i, d in enumerate(depths):
            self(d, out=out[i])
        return out

    @classmethod
    def from_config(cls, cfg=None):
        cfg = cfg or {}
        return cls(img_mean=cfg.get("img_mean", None), img_std=cfg.get("img_std", None), depth_mean=cfg.get("depth_mean", 0.0418),
                   depth_std=cfg.get("depth_std", 0.0295), max_depth=cfg.get("max_depth", 75),
                   clamp_max_before_scale=cfg.get("clamp_max_before_scale", True))
