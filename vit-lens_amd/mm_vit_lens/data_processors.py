"""Per-modality input processors of `ViTLens.encode` with the reference's class names and call convention
(mm_vit_lens/data_processors.py:15-323): `processor(paths_or_items, device=...)` -> the batched model-space tensor.

File decoding (JPEG/PNG via Pillow, `.npy`, torch-saved tensors) stays on the host as in the reference; everything after
the decode runs on the GPU through the on-GPU transforms (open_clip/transform.py, modal_depth / modal_3d /
modal_tactile processors).  Besides paths every processor accepts already-decoded items (PIL images,
arrays, tensors).  The audio processor takes spectrogram tensors only: the reference's waveform front end is torchaudio's
kaldi fbank (modal_audio/processors/at_processor.py:854-903), which is not available here."""
import re
from types import SimpleNamespace

import numpy as np
import torch

from open_clip import get_tokenizer


def wrap_list(data):
    if isinstance(data, list):
        return data
    return [data]


def _is_path(x):
    return isinstance(x, (str, bytes)) or hasattr(x, "__fspath__")


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


class TextProcessor(BaseProcessor):
    def __init__(self, prompt="", max_words=70, cfg=None):
        self.prompt, self.max_words = prompt, max_words
        self.cfg = cfg if cfg is not None else SimpleNamespace(model="ViT-L-14")
        self.tokenizer = get_tokenizer(self.cfg.model)

    def __call__(self, caption, device="cpu"):
        if caption is None:
            return None
        caption = [self.prompt + self.pre_caption(c) for c in wrap_list(caption)]
        return self.tokenizer(caption).to(device)

    @classmethod
    def from_config(cls, cfg=None):
        cfg = cfg if cfg is not None else {"model": "ViT-L-14"}
        get = cfg.get if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        return cls(prompt=get("prompt", ""), max_words=get("max_words", 70), cfg=SimpleNamespace(model=get("model", "ViT-L-14")))

    def pre_caption(self, caption):
        caption = re.sub(r"([.!\"()*#:;~])", " ", caption.lower())
        caption = re.sub(r"\s{2,}", " ", caption)
        caption = caption.rstrip("\n").strip(" ")
        words = caption.split(" ")
        if len(words) > self.max_words:
            caption = " ".join(words[:self.max_words])
        return caption


class ImageProcessor(BaseProcessor):
    def __init__(self, image_size=224, image_mean=None, image_std=None, transform=None):
        self.image_size, self.image_mean, self.image_std = image_size, image_mean, image_std
        if transform:
            self.transform = transform
        else:
            from open_clip.transform import image_transform
            self.transform = image_transform(image_size=image_size, is_train=False, mean=image_mean, std=image_std)

    def set_image_transform(self, transform):
        self.transform = transform

    def __call__(self, image_paths, device="cpu"):
        if image_paths is None:
            return None
        outs = []
        for item in wrap_list(image_paths):
            if _is_path(item):
                from PIL import Image
                with open(item, "rb") as f:
                    item = Image.open(f).convert("RGB")
            outs.append(self.transform(item))
        return torch.stack(outs, dim=0).to(device)


class PointCloudProcessor(BaseProcessor):
    def __init__(self, n_sample_points=8192, uniform=True, idendity=False):
        from open_clip.modal_3d.processors.pc_processor import PCProcessorEval
        self.n_sample_points, self.uniform = n_sample_points, uniform
        self.wrap_processor = PCProcessorEval(n_sample_points, uniform, idendity)

    def set_idendity(self, idendity_v):
        self.wrap_processor.idendity = idendity_v

    def __call__(self, pc_paths, device="cpu"):
        if pc_paths is None:
            return None
        outs = [self.wrap_processor(np.load(p) if _is_path(p) else p) for p in wrap_list(pc_paths)]
        return torch.stack(outs, dim=0).to(device)


class DepthProcessor(BaseProcessor):
    def __init__(self, depth_mean=0.0418, depth_std=0.0295, max_depth=75, clamp_max_before_scale=True):
        from open_clip.modal_depth.processors.vt_processor import DepthProcessorEval
        self.depth_mean, self.depth_std, self.max_depth, self.clamp_max_before_scale = depth_mean, depth_std, max_depth, clamp_max_before_scale
        self.wrap_processor = DepthProcessorEval(depth_mean=depth_mean, depth_std=depth_std, max_depth=max_depth,
                                                 clamp_max_before_scale=clamp_max_before_scale)

    def __call__(self, depth_paths, device="cpu"):
        if depth_paths is None:
            return None
        outs = []
        for p in wrap_list(depth_paths):
            d = torch.load(p, map_location="cpu", weights_only=False) if _is_path(p) else p
            outs.append(self.wrap_processor(d))
        return torch.stack(outs, dim=0).to(device)


class AudioProcessor(BaseProcessor):
    def __init__(self, sampling_rate=16000, clip_duration=5.0, n_clip=3, target_length=512, mel_bins=128, cfg=None):
        self.sampling_rate, self.clip_duration, self.n_clip = sampling_rate, clip_duration, n_clip
        self.target_length, self.mel_bins, self.cfg = target_length, mel_bins, cfg

    def setter(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __call__(self, audio_items, device="cpu"):
        """Items: spectrogram tensors [n_clip, target_length, mel_bins] (or [target_length, mel_bins]) in model space."""
        if audio_items is None:
            return None
        outs = []
        for a in wrap_list(audio_items):
            if _is_path(a) or (torch.is_tensor(a) and a.dim() <= 2 and a.shape[-1] != self.mel_bins):
                # a .wav path or a raw waveform [n] / [1, n]: clips + Kaldi log-mel filterbank on the GPU (at_processor.py:823-903)
                from open_clip.modal_audio.processors.at_processor import AudioASTProcessorEval
                proc = AudioASTProcessorEval(sampling_rate=self.sampling_rate, clip_duration=self.clip_duration, n_clip=self.n_clip,
                                             target_length=self.target_length, mel_bins=self.mel_bins,
                                             device=device if str(device) != "cpu" else "cuda")
                outs.append(proc(a))
                continue
            a = torch.as_tensor(a, dtype=torch.float32)
            if a.shape[-2:] != (self.target_length, self.mel_bins):
                raise ValueError(f"expected [.., {self.target_length}, {self.mel_bins}] spectrograms, got {tuple(a.shape)}")
            outs.append(a)
        return torch.stack([o.to(device) for o in outs], dim=0)


class TactileProcessor(BaseProcessor):
    def __init__(self, image_mean=None, image_std=None):
        from open_clip.modal_tactile.processors.tact_processor import TactileRGBProcessorEval
        self.image_mean, self.image_std = image_mean, image_std
        self.wrap_processor = TactileRGBProcessorEval(img_mean=image_mean, img_std=image_std)

    def __call__(self, tactile_flist, device="cpu"):
        if tactile_flist is None:
            return None
        return torch.stack([self.wrap_processor(t) for t in wrap_list(tactile_flist)], dim=0).to(device)


class EEGProcessor(BaseProcessor):
    def __init__(self, time_low=20, time_high=460, data_len=512):
        from open_clip.modal_eeg.processors.eeg_processor import EEGProcessorEval
        self.time_low, self.time_high, self.data_len = time_low, time_high, data_len
        self.wrap_processor = EEGProcessorEval(time_low=time_low, time_high=time_high, data_len=data_len)

    def __call__(self, eeg_paths, device="cpu"):
        if eeg_paths is None:
            return None
        return torch.stack([self.wrap_processor(e) for e in wrap_list(eeg_paths)], dim=0).to(device)


def vitlensL_processors():
    return dict(
        image=ImageProcessor(image_size=224, image_mean=None, image_std=None, transform=None),
        text=TextProcessor(cfg=SimpleNamespace(model="ViT-L-14")),
        pc=PointCloudProcessor(n_sample_points=8192, uniform=True),
        depth=DepthProcessor(),
        audio=AudioProcessor(),
        tactile=TactileProcessor(),
        eeg=EEGProcessor(),
    )


def vitlensB_processors():
    return None


def get_vitlens_processors_cls():
    return dict(vitlensL=vitlensL_processors, vitlensB=vitlensB_processors)
