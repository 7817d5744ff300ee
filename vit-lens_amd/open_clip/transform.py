"""Image preprocessing ON THE GPU behind the reference's `image_transform` (open_clip/transform.py:72-155).

The reference composes torchvision transforms over a PIL image on the data-loader workers:
    eval :  Resize(size, BICUBIC) -> CenterCrop(size) -> convert("RGB") -> ToTensor -> Normalize          (:138-155)
    train:  RandomResizedCrop(size, scale, BICUBIC)   -> convert("RGB") -> ToTensor -> Normalize          (:121-134)
Here the decoded bytes go to the GPU once and two kernels do the rest (csrc/vl_preproc.hip): Pillow's 8-bit bicubic
resampler restricted to the crop window, then (u8/255 - mean)/std.  The result is BIT-IDENTICAL to the reference's
float32 tensor for RGB and greyscale inputs (byte-exact resampling, IEEE divisions); it lives on the GPU as
[3, size, size], which is what `encode_image` takes.

`image_transform(...)` has the reference's signature and returns a callable taking a PIL image, an [H, W, 3|1] uint8
numpy array or a uint8 tensor.  The random crop box of the training transform is drawn on the host with the torch RNG
in torchvision's order of draws (RandomResizedCrop.get_params; torchvision itself is not installed in this image, so
that order is restated from its published source and not pinned by a test).  `resize_longest_max` (ResizeMaxSize +
pad, :33-66) and the timm augmentation branch are not used by the ViT-Lens recipes and raise."""
import math
from dataclasses import asdict, dataclass
from typing import Optional, Tuple, Union

import numpy as np
import torch

from .constants import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD


@dataclass
class AugmentationCfg:
    scale: Tuple[float, float] = (0.9, 1.0)
    ratio: Optional[Tuple[float, float]] = None
    color_jitter: Optional[Union[float, Tuple[float, float, float]]] = None
    interpolation: Optional[str] = None
    re_prob: Optional[float] = None
    re_count: Optional[int] = None
    use_timm: bool = False


def random_resized_crop_params(height, width, scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """RandomResizedCrop.get_params: (top, left, h, w), ten attempts with the global torch RNG, then the central
    crop at the nearest admissible aspect ratio."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            top = torch.randint(0, height - h + 1, size=(1,)).item()
            left = torch.randint(0, width - w + 1, size=(1,)).item()
            return top, left, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def _to_device_u8(img, device):
    """PIL image / numpy / tensor -> uint8 [H, W, C] on the device.  The reference converts to RGB after Resize; for
    greyscale that commutes (replicating a channel, then resampling each copy), other PIL modes are converted first."""
    return _to_host_or_device_u8(img).to(device, non_blocking=True)


def _to_host_or_device_u8(img):
    """As _to_device_u8, but host data stays on the host (the batched path packs it into one copy)."""
    if hasattr(img, "mode") and hasattr(img, "convert"):
        if img.mode not in ("RGB", "L"):
            img = img.convert("RGB")
        img = np.array(img)
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
    if img.dtype != torch.uint8:
        raise TypeError(f"image bytes expected (uint8), got {img.dtype}")
    if img.dim() == 2:
        img = img.unsqueeze(-1)
    if img.dim() != 3 or img.shape[-1] not in (1, 3):
        raise ValueError(f"expected an [H, W, 3] or [H, W(, 1)] image, got {tuple(img.shape)}")
    return img if img.is_cuda else img.contiguous()


class ImageTransform:
    def __init__(self, image_size, mean, std, is_train=False, scale=(0.9, 1.0), ratio=None, device="cuda"):
        self.image_size, self.mean, self.std, self.is_train = image_size, tuple(mean), tuple(std), is_train
        self.scale, self.ratio, self.device = tuple(scale), tuple(ratio) if ratio else (3.0 / 4.0, 4.0 / 3.0), torch.device(device)

    def __call__(self, img, out=None):
        from vitlens_hip import preproc
        x = _to_device_u8(img, self.device)
        if x.shape[-1] == 1:                                                      # convert("RGB") of a greyscale image
            x = x.expand(-1, -1, 3).contiguous()
        size = self.image_size
        if self.is_train:
            box = random_resized_crop_params(x.shape[0], x.shape[1], self.scale, self.ratio)
            hw = tuple(size) if isinstance(size, (tuple, list)) else (size, size)
            return preproc.image_to_tensor(x, hw, self.mean, self.std, out=out, box=box)
        return preproc.image_to_tensor(x, size, self.mean, self.std, out=out)

    def batch(self, images):
        """A list of images (any sizes) -> [B, 3, size, size] float32 on the GPU (what `encode_image` consumes) in two
        kernel launches: host images travel as ONE packed copy, each image is described by one descriptor row."""
        from vitlens_hip import preproc
        xs = []
        for im in images:
            x = _to_host_or_device_u8(im)
            xs.append(x.expand(-1, -1, 3).contiguous() if x.shape[-1] == 1 else x)
        size = self.image_size
        hw = tuple(size) if isinstance(size, (tuple, list)) else (size, size)
        out = torch.empty(len(xs), 3, hw[0], hw[1], device=self.device, dtype=torch.float32)
        boxes = [random_resized_crop_params(x.shape[0], x.shape[1], self.scale, self.ratio) for x in xs] if self.is_train else None
        return preproc.images_to_tensor(xs, hw if self.is_train else hw[0], self.mean, self.std, out=out, boxes=boxes)


def image_transform(image_size, is_train, mean=None, std=None, resize_longest_max=False, fill_color=0, aug_cfg=None,
                    device="cuda"):
    mean = mean or OPENAI_DATASET_MEAN
    if not isinstance(mean, (list, tuple)):
        mean = (mean,) * 3
    std = std or OPENAI_DATASET_STD
    if not isinstance(std, (list, tuple)):
        std = (std,) * 3
    if isinstance(image_size, (list, tuple)) and image_size[0] == image_size[1]:
        image_size = image_size[0]                                                # square: aspect-preserving shortest edge
    if isinstance(aug_cfg, dict):
        aug_cfg = AugmentationCfg(**aug_cfg)
    else:
        aug_cfg = aug_cfg or AugmentationCfg()
    if resize_longest_max:
        raise NotImplementedError("resize_longest_max (ResizeMaxSize + pad) is not part of the ViT-Lens recipes")
    if is_train:
        cfg = {k: v for k, v in asdict(aug_cfg).items() if v is not None}
        if cfg.pop("use_timm", False):
            raise NotImplementedError("the timm augmentation branch needs timm (host-side data augmentation, out of scope)")
        return ImageTransform(image_size, mean, std, is_train=True, scale=cfg.pop("scale"), ratio=cfg.pop("ratio", None),
                              device=device)
    if not isinstance(image_size, int):
        raise NotImplementedError("non-square evaluation sizes are not used by the ViT-Lens recipes")
    return ImageTransform(image_size, mean, std, device=device)
