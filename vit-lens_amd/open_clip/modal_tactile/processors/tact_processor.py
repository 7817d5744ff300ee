"""Tactile (GelSight RGB) preprocessing ON THE GPU (reference: open_clip/modal_tactile/processors/tact_processor.py:
281-300 `TactileRGBProcessorEval`): ToTensor -> Resize(256, bicubic) -> CenterCrop(224) -> Normalize.  Unlike the image
transform the resize runs on the FLOAT tensor (torchvision's tensor path = torch.nn.functional.interpolate, see
modal_depth/processors/vt_processor.py for the `antialias` default), so the three channels go through the float
resampler of csrc/vl_preproc.hip with ToTensor's /255 fused into the read.  Same class name and call convention as the
reference (a file path; a PIL image or an [H, W, 3] uint8 array is accepted too); result [3, 224, 224] float32 on the GPU."""
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


class TactileRGBProcessorEval(BaseProcessor):
    def __init__(self, img_mean=None, img_std=None, resize=256, size=224, antialias=True, device="cuda"):
        self.mean = img_mean if img_mean is not None else OPENAI_CLIP_MEAN
        self.std = img_std if img_std is not None else OPENAI_CLIP_STD
        self.resize, self.size, self.antialias, self.device = resize, size, antialias, torch.device(device)

    def __call__(self, img, out=None):
        from vitlens_hip import preproc
        if isinstance(img, (str, bytes)) or hasattr(img, "__fspath__"):
            from PIL import Image                                                 # decoding stays on the host, as in the reference
            img = Image.open(img)
        if hasattr(img, "convert"):
            img = np.array(img.convert("RGB"))
        if isinstance(img, np.ndarray):
            img = torch.from_numpy(np.ascontiguousarray(img))
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[-1] != 3:
            raise ValueError(f"expected an [H, W, 3] uint8 image, got {tuple(img.shape)} {img.dtype}")
        return preproc.float_image_to_tensor(img.to(self.device), self.resize, self.size, self.mean, self.std,
                                             antialias=self.antialias, out=out)
