"""Model registry / factory with the reference's signatures (open_clip/factory.py:60-128,164-366,368-466,750-851).
Download / HF-hub / OpenAI-JIT loading are out of scope (no network): `pretrained` may be a
local checkpoint path, which is loaded with the reference's `visual.* -> image.*` duplication rule."""
import json
import logging
import os
import re
from copy import deepcopy
from pathlib import Path
from typing import Optional, Union

import torch

from .loss import ClipLoss, ClipLossGeneral, TriClipLoss
from .model import TriCLIP, resize_pos_embed
from .tokenizer import tokenize

_CONFIG_PATHS = [Path(__file__).parent / "model_configs"]
_CONFIGS = {}


def _rescan():
    global _CONFIGS
    found = {}
    for p in _CONFIG_PATHS:
        files = [p] if p.is_file() else sorted(p.glob("*.json"))
        for f in files:
            cfg = json.load(open(f))
            if all(k in cfg for k in ("embed_dim", "vision_cfg", "text_cfg")):
                found[f.stem] = cfg
    _CONFIGS = dict(sorted(found.items(), key=lambda kv: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", kv[0].lower())]))


_rescan()


def list_models():
    return list(_CONFIGS.keys())


def add_model_config(path):
    _CONFIG_PATHS.append(Path(path))
    _rescan()


def get_model_config(model_name):
    return deepcopy(_CONFIGS[model_name]) if model_name in _CONFIGS else None


def get_tokenizer(model_name):
    return tokenize


def _perceiver_cfg(a):
    keys = ["input_chan", "input_axis", "num_freq_bands", "max_freq", "depth", "num_latents", "latent_dim", "cross_heads",
            "latent_heads", "cross_dim_head", "latent_dim_head", "num_classes", "attn_dropout", "ff_dropout",
            "weight_tie_layers", "fourier_encode_data", "self_per_cross_attn"]
    d = {"use_perceiver": a.use_perceiver}
    d.update({k: getattr(a, "perceiver_" + k, None) for k in keys})
    return d


def load_state_dict(checkpoint_path: str, map_location="cpu"):
    ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=False)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    if next(iter(sd)).startswith("module"):
        sd = {k[7:]: v for k, v in sd.items()}
    return sd


def load_checkpoint(model, checkpoint_path, strict=True, args=None):
    """factory.py:130-160: open_clip checkpoints call the image encoder `visual`; copy it to `image.*` (and drop the
    original when the visual tower does not start from the pretrained ViT), resize the position embedding, load,
    log the incompatible keys."""
    sd = load_state_dict(checkpoint_path)
    do_pop = args is not None and (getattr(args, "visual_arch", "perceiver_vit") != "perceiver_vit"
                                   or getattr(args, "disable_pt_vit", False))
    if hasattr(model, "image") and hasattr(model, "visual"):
        for k in list(sd.keys()):
            if "visual." in k:
                sd[k.replace("visual.", "image.")] = sd[k]
                if do_pop:
                    sd.pop(k)
    resize_pos_embed(sd, model)
    incompatible = model.load_state_dict(sd, strict=strict)
    if len(incompatible.missing_keys) or len(incompatible.unexpected_keys):
        logging.info(msg=incompatible)
    return incompatible


def tri_create_model(model_name: str, pretrained: Optional[str] = None, precision: str = "fp32",
                     device: Union[str, torch.device] = "cpu", jit: bool = False, force_quick_gelu: bool = False,
                     force_custom_text: bool = False, force_patch_dropout: Optional[float] = None,
                     force_image_size=None, pretrained_image: bool = False, pretrained_hf: bool = True,
                     cache_dir: Optional[str] = None, output_dict: Optional[bool] = None,
                     require_pretrained: bool = False, strict: bool = False, args=None, text_arith: str = "f16"):
    """text_arith (appended to the reference's signature, factory.py:164): see TriCLIP.set_precision."""
    model_name = model_name.replace("/", "-")
    cfg = get_model_config(model_name)
    if cfg is None:
        raise RuntimeError(f"Model config for {model_name} not found; available models {list_models()}.")
    if jit or force_custom_text or force_quick_gelu or pretrained_image:
        raise NotImplementedError("jit / custom-text / quick-gelu / timm towers are outside the hot path")
    if precision not in ("fp32", "amp", "amp_bf16", "amp_bfloat16", "bf16"):
        raise NotImplementedError(f"precision {precision!r}: the MI355X path computes GEMMs in bf16 with fp32 accumulation")
    if force_image_size is not None:
        cfg["vision_cfg"]["image_size"] = force_image_size
    if args is not None:
        v = cfg["vision_cfg"]
        v["use_perceiver"] = args.use_perceiver
        v["visual_modality_type"] = args.visual_modality_type
        v["perceiver_cfg"] = _perceiver_cfg(args)
        v["use_visual_adapter"] = getattr(args, "use_visual_adapter", False)
        v["visual_arch"] = getattr(args, "visual_arch", "perceiver_vit")
        v["exp_args"] = args
    model = TriCLIP(**cfg)
    model.set_precision(precision, text_arith=text_arith)
    model.to(device=torch.device(device))
    if pretrained:
        if not os.path.exists(pretrained):
            raise RuntimeError(f"Pretrained weights ({pretrained}) not found for model {model_name} (no network access).")
        load_checkpoint(model, pretrained, strict, args)
    elif require_pretrained:
        raise RuntimeError(f"Pretrained weights were required for (model: {model_name}) but not loaded.")
    if output_dict:
        model.output_dict = True
    skip = getattr(args, "skip_trans_first_n_layers", None) if args is not None else None
    if skip is not None:
        # "Add skip-first-n-layers here to drop layers" (factory.py:347-360): after the weights are in place
        n_layers = model.visual.cfg.layers
        assert skip < n_layers
        logging.info("Using last %d out of %d transformer layers.", n_layers - skip, n_layers)
        model.visual.keep_last_layers(n_layers - skip)
    return model


def tri_create_model_and_transforms(model_name: str, pretrained: Optional[str] = None, precision: str = "fp32",
                                    device="cpu", jit=False, force_quick_gelu=False, force_custom_text=False,
                                    force_patch_dropout=None, force_image_size=None, pretrained_image=False,
                                    pretrained_hf=True, load_ckpt_strict=False, image_mean=None, image_std=None,
                                    aug_cfg=None, cache_dir=None, output_dict=None, args=None):
    """Returns (model, preprocess_train, preprocess_val) (factory.py:372-424); the two transforms are the on-GPU
    `image_transform`s of open_clip/transform.py (Pillow-exact bicubic resize + crop + normalise on the model's device)."""
    model = tri_create_model(model_name, pretrained, precision=precision, device=device, jit=jit,
                             force_quick_gelu=force_quick_gelu, force_custom_text=force_custom_text,
                             force_patch_dropout=force_patch_dropout, force_image_size=force_image_size,
                             pretrained_image=pretrained_image, pretrained_hf=pretrained_hf, cache_dir=cache_dir,
                             output_dict=output_dict, strict=load_ckpt_strict, args=args)
    from .transform import image_transform
    image_mean = image_mean or getattr(model.image, "image_mean", None)
    image_std = image_std or getattr(model.image, "image_std", None)
    size = model.image.cfg.image_size
    tdev = device if torch.device(device).type == "cuda" else "cuda"
    preprocess_train = image_transform(size, is_train=True, mean=image_mean, std=image_std, aug_cfg=aug_cfg, device=tdev)
    preprocess_val = image_transform(size, is_train=False, mean=image_mean, std=image_std, device=tdev)
    return model, preprocess_train, preprocess_val


def tri_create_model_from_pretrained(model_name: str, pretrained: Optional[str] = None, precision: str = "fp32",
                                     device: Union[str, torch.device] = "cpu", jit: bool = False, force_quick_gelu: bool = False,
                                     force_custom_text: bool = False, force_image_size=None, return_transform: bool = True,
                                     image_mean=None, image_std=None, cache_dir: Optional[str] = None):
    """factory.py:425-464: a model that MUST come with weights (`pretrained` = a local checkpoint path here), optionally
    with the evaluation transform."""
    model = tri_create_model(model_name, pretrained, precision=precision, device=device, jit=jit, force_quick_gelu=force_quick_gelu,
                             force_custom_text=force_custom_text, force_image_size=force_image_size, cache_dir=cache_dir,
                             require_pretrained=True)
    if not return_transform:
        return model
    from .transform import image_transform
    tdev = device if torch.device(device).type == "cuda" else "cuda"
    preprocess = image_transform(model.visual.cfg.image_size, is_train=False, mean=image_mean or getattr(model.visual, "image_mean", None),
                                 std=image_std or getattr(model.visual, "image_std", None), device=tdev)
    return model, preprocess


def create_loss(args):
    """factory.py:750-851 restricted to the hot-path losses (general contrastive, tri / dual / plain)."""
    kw = dict(local_loss=args.local_loss, gather_with_grad=args.gather_with_grad, cache_labels=True,
              rank=args.rank, world_size=args.world_size, use_horovod=getattr(args, "horovod", False))
    if getattr(args, "distill", False) or "coca" in getattr(args, "model", "").lower() or getattr(args, "vid_distill_tokens", False):
        raise NotImplementedError("distillation / CoCa / video-token losses are out of scope (SURVEY §2 A5)")
    if getattr(args, "n_tower", 2) == 3:
        if getattr(args, "contra_loss_type", "general") != "general":
            raise NotImplementedError("label_mask / sim_mask losses are out of scope (SURVEY §2 A5)")
        if getattr(args, "use_dual_loss", False):
            logging.info("[Loss class]: ClipLossGeneral")
            return ClipLossGeneral(**kw)
        logging.info("[Loss class]: TriClipLoss")
        return TriClipLoss(**kw)
    return ClipLoss(**kw)
