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
    """One item or a list of items -> list (the reference's processors accept either)."""
    return data if isinstance(data, list) else [data]


def _is_path(x):
    return isinstance(x, (str, bytes)) or hasattr(x, "__fspath__")


def _batch(items, one, device):
    """The call convention every processor shares: None passes through; otherwise `one(item)` per item, stacked on a new
    leading axis and moved to `device`."""
    if items is None:
        return None
    return torch.stack([one(it) for it in wrap_list(items)], dim=0).to(device)


class BaseProcessor:
    """Identity processor; subclasses replace `transform` (or `__call__`).  `from_config` / `build` exist because callers of
    the reference construct processors from config dictionaries (mm_vit_lens/data_processors.py:25-43)."""
    transform = staticmethod(lambda item: item)

    def __call__(self, item):
        return self.transform(item)

    @classmethod
    def from_config(cls, cfg=None):
        return cls()

    def build(self, **kwargs):
        return self.from_config(dict(kwargs))


# caption clean-up of the reference's TextProcessor (data_processors.py:67-82), restated from its behaviour: lower-case,
# the ten punctuation marks . ! " ( ) * # : ; ~ become blanks, every run of two or more white-space characters becomes one
# blank (a lone tab / newline survives), trailing newlines and surrounding blanks go, at most `max_words` blank-separated
# words are kept.  tests/test_data_processors.py pins it against the imported reference on generated captions.
_PUNCT_TO_BLANK = str.maketrans({c: " " for c in '.!"()*#:;~'})
_WS_RUN = re.compile(r"\s\s+")


def clean_caption(caption: str, max_words: int) -> str:
    text = _WS_RUN.sub(" ", caption.lower().translate(_PUNCT_TO_BLANK))
    text = text.rstrip("\n").strip(" ")
    words = text.split(" ")
    return text if len(words) <= max_words else " ".join(words[:max_words])


class TextProcessor(BaseProcessor):
    def __init__(self, prompt="", max_words=70, cfg=None):
        self.prompt, self.max_words = prompt, max_words
        self.cfg = SimpleNamespace(model="ViT-L-14") if cfg is None else cfg
        self.tokenizer = get_tokenizer(self.cfg.model)

    def pre_caption(self, caption):
        return clean_caption(caption, self.max_words)

    def __call__(self, caption, device="cpu"):
        if caption is None:
            return None
        return self.tokenizer([self.prompt + self.pre_caption(c) for c in wrap_list(caption)]).to(device)

    @classmethod
    def from_config(cls, cfg=None):
        read = (lambda k, d: d) if cfg is None else (cfg.get if hasattr(cfg, "get") else (lambda k, d: getattr(cfg, k, d)))
        return cls(prompt=read("prompt", ""), max_words=read("max_words", 70), cfg=SimpleNamespace(model=read("model", "ViT-L-14")))


class ImageProcessor(BaseProcessor):
    def __init__(self, image_size=224, image_mean=None, image_std=None, transform=None):
        self.image_size, self.image_mean, self.image_std = image_size, image_mean, image_std
        if not transform:
            from open_clip.transform import image_transform
            transform = image_transform(image_size=image_size, is_train=False, mean=image_mean, std=image_std)
        self.transform = transform

    def set_image_transform(self, transform):
        self.transform = transform

    def _one(self, item):
        if _is_path(item):            # decode on the host (Pillow), everything after it on the GPU (open_clip/transform.py)
            from PIL import Image
            with open(item, "rb") as fh:
                item = Image.open(fh).convert("RGB")
        return self.transform(item)

    def __call__(self, image_paths, device="cpu"):
        return _batch(image_paths, self._one, device)


class PointCloudProcessor(BaseProcessor):
    def __init__(self, n_sample_points=8192, uniform=True, idendity=False):
        from open_clip.modal_3d.processors.pc_processor import PCProcessorEval
        self.n_sample_points, self.uniform = n_sample_points, uniform
        self.wrap_processor = PCProcessorEval(n_sample_points, uniform, idendity)

    def set_idendity(self, idendity_v):
        self.wrap_processor.idendity = idendity_v

    def __call__(self, pc_paths, device="cpu"):
        return _batch(pc_paths, lambda p: self.wrap_processor(np.load(p) if _is_path(p) else p), device)


class DepthProcessor(BaseProcessor):
    def __init__(self, depth_mean=0.0418, depth_std=0.0295, max_depth=75, clamp_max_before_scale=True):
        from open_clip.modal_depth.processors.vt_processor import DepthProcessorEval
        self.depth_mean, self.depth_std, self.max_depth, self.clamp_max_before_scale = depth_mean, depth_std, max_depth, clamp_max_before_scale
        self.wrap_processor = DepthProcessorEval(depth_mean=depth_mean, depth_std=depth_std, max_depth=max_depth,
                                                 clamp_max_before_scale=clamp_max_before_scale)

    def __call__(self, depth_paths, device="cpu"):
        load = lambda p: torch.load(p, map_location="cpu", weights_only=False) if _is_path(p) else p
        return _batch(depth_paths, lambda p: self.wrap_processor(load(p)), device)


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
        return _batch(tactile_flist, self.wrap_processor, device)


class EEGProcessor(BaseProcessor):
    def __init__(self, time_low=20, time_high=460, data_len=512):
        from open_clip.modal_eeg.processors.eeg_processor import EEGProcessorEval
        self.time_low, self.time_high, self.data_len = time_low, time_high, data_len
        self.wrap_processor = EEGProcessorEval(time_low=time_low, time_high=time_high, data_len=data_len)

    def __call__(self, eeg_paths, device="cpu"):
        return _batch(eeg_paths, self.wrap_processor, device)


# modality key of `ViTLens.encode` -> processor class; the release model's processors take their class defaults (ViT-L/14
# tokenizer, 224 px CLIP transform, 8192-point uniform sampling, SUN-RGBD depth statistics, 3 x 5 s clips of 512 x 128 log-mel)
_VITLENS_L = {"image": ImageProcessor, "text": TextProcessor, "pc": PointCloudProcessor, "depth": DepthProcessor,
              "audio": AudioProcessor, "tactile": TactileProcessor, "eeg": EEGProcessor}


def vitlensL_processors():
    return {modality: cls() for modality, cls in _VITLENS_L.items()}


def vitlensB_processors():
    return None


def get_vitlens_processors_cls():
    return dict(vitlensL=vitlensL_processors, vitlensB=vitlensB_processors)
