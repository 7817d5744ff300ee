"""`ViTLens.encode({ModalityType: inputs})` with the reference's call surface (mm_vit_lens/vitlens.py:21-189).

Inputs are either what the reference takes - file paths / captions, run through the per-modality processors of
`mm_vit_lens.data_processors` (decode on the host, everything else on the GPU) - or tensors already in model space,
which skip the processor:
   image [B,3,224,224] | text: list[str] or int64 [B,77] | depth [B,1,224,224] | audio [B,S,512,128] or [B,512,128]
   | pc [B,8192,3] | tactile [B,3,224,224] | eeg [B,128,512]
`encode` returns {modality: [B, 768]} (audio: mean over the S clips, vitlens.py:175-183), L2-normalised by default.
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from open_clip import ModalityType, tri_create_model
from .model_cfg import fetch_model_cfg



def _create(name, device, args):
    """tri_create_model with the reference wrapper's (default) precision="fp32" = the fp32 residual stream of this path.
    The factory's note that no fp32-ARITHMETIC mode exists is for callers who ask for it; this wrapper's arithmetic is
    documented on the class (bf16 operands, fp32 accumulation - the reference's example runs under autocast)."""
    import warnings
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", message="precision='fp32'")
        return tri_create_model(name, None, precision="fp32", device=device, args=args)


class ViTLens(nn.Module):
    def __init__(self, model_var: str = "vitlensL", modality_loaded: Optional[List[str]] = None,
                 load_from_ckpt: Optional[str] = None, device="cuda"):
        super().__init__()
        self.model_var = model_var
        # the reference's default (vitlens.py:22-25): every modality of the release
        self.modality_loaded = modality_loaded or ["image", "text", "pc", "depth", "audio", "tactile", "eeg"]
        self.vitlens = nn.ModuleDict()
        self._dev = torch.device(device)
        self.processors = {}
        base = None
        for m in self.modality_loaded:
            if m in (ModalityType.IMAGE, ModalityType.TEXT):
                if base is None:
                    cfg = fetch_model_cfg(modality="image", model_option=model_var)
                    base = _create(cfg.model, device=self._dev, args=cfg)
                self.vitlens[m] = base
            elif m in (ModalityType.DEPTH, ModalityType.AUDIO, ModalityType.PC, ModalityType.TACTILE, ModalityType.EEG):
                # only the modality's `visual` tower is kept, as in the reference (`self.vitlens.add_module(modality,
                # model.visual); del model`, vitlens.py:100-107): the model's own image / text towers would be 1.7 GB of unused,
                # randomly initialised fp32 parameters per modality
                cfg = fetch_model_cfg(modality=m, model_option=model_var)
                full = _create(cfg.model, device="cpu", args=cfg)
                self.vitlens[m] = full.visual.to(self._dev)
                del full
            else:
                raise NotImplementedError(f"modality {m!r} is outside the hot path (SURVEY §8)")
        if load_from_ckpt is not None:
            self.load_checkpoint(load_from_ckpt)

    @property
    def device(self):
        return self._dev

    # ---- checkpoint format of the reference release (vitlens.py:33,63-118,153-159) ----------------------------------
    # `vitlens.image.<VisionTransformer keys>`  = TriCLIP.image, `vitlens.text.{transformer.*, token_embedding.weight,
    # positional_embedding, ln_final.*, text_projection}` = the text tower (flattened into the TriCLIP root), and
    # `vitlens.<modality>.<VisionTransformer keys>` = TriCLIP.visual of that modality's model.
    _TEXT_KEYS = ("transformer.", "token_embedding.", "ln_final.")

    def _part(self, m):
        model = self.vitlens[m]
        if m == ModalityType.IMAGE:
            return model.image, None
        if m == ModalityType.TEXT:
            return model, lambda k: k.startswith(self._TEXT_KEYS) or k in ("positional_embedding", "text_projection")
        return model, None            # the modality's visual tower itself

    def state_dict(self, *args, **kwargs):
        out = {}
        for m in self.vitlens.keys():
            mod, keep = self._part(m)
            for k, v in mod.state_dict().items():
                if keep is None or keep(k):
                    out[f"vitlens.{m}.{k}"] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = False):
        """Loads a release-format dict; returns (missing, unexpected) and logs them (the reference logs its
        incompatible keys, vitlens.py:131-133)."""
        import logging
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        missing, unexpected, used = [], [], set()
        for m in self.vitlens.keys():
            pre = f"vitlens.{m}."
            sub = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
            used.update(pre + k for k in sub)
            mod, keep = self._part(m)
            own = {k for k in mod.state_dict().keys() if keep is None or keep(k)}
            missing += [pre + k for k in sorted(own - set(sub)) if not k.endswith("num_batches_tracked")]
            unexpected += [pre + k for k in sorted(set(sub) - own)]
            mod.load_state_dict({k: v for k, v in sub.items() if k in own}, strict=False)
        unexpected += sorted(k for k in sd if k.startswith("vitlens.") and k not in used)
        if missing or unexpected:
            logging.info("ViTLens.load_state_dict: missing %s, unexpected %s", missing, unexpected)
        if strict and (missing or unexpected):
            raise RuntimeError(f"ViTLens.load_state_dict: missing {missing}, unexpected {unexpected}")
        return missing, unexpected

    def load_checkpoint(self, path):
        """`path`: a release file, or - as the reference's `load_from_ckpt` (vitlens.py:24,118-133) - the directory that holds
        `<model_var>.pt` (there is no download: a missing file is an error)."""
        import os
        if os.path.isdir(path):
            path = os.path.join(path, f"{self.model_var}.pt")
        if not os.path.exists(path):
            raise FileNotFoundError(f"ViT-Lens weights not found at {path} (no network access: place the release file there)")
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        return self.load_state_dict(ckpt.get("state_dict", ckpt), strict=False)

    def load_modality_from_pt_ckpt(self, modality, pt_ckpt_path):
        """Load the `visual.*` tower of a TRAINING checkpoint into one modality (vitlens.py:135-151)."""
        ckpt = torch.load(pt_ckpt_path, map_location="cpu", weights_only=False)
        sd = ckpt.get("state_dict", ckpt)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        sd = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
        return self.vitlens[modality].load_state_dict(sd, strict=False)

    def export_checkpoint(self, save_path="model_release/vitlens.pt"):
        torch.save(dict(model_var=self.model_var, modality_loaded=self.modality_loaded, state_dict=self.state_dict()), save_path)

    def reduce_list(self, modality):
        """Modalities whose input is a LIST of clips per sample, encoded clip by clip and averaged (vitlens.py:165-168)."""
        return modality in [ModalityType.AUDIO]

    def processor(self, m):
        """The reference's per-modality input processor (built on first use: vitlens.py:108-116)."""
        if m not in self.processors:
            from . import data_processors as DP
            cls = {ModalityType.IMAGE: DP.ImageProcessor, ModalityType.TEXT: DP.TextProcessor, ModalityType.PC: DP.PointCloudProcessor,
                   ModalityType.DEPTH: DP.DepthProcessor, ModalityType.AUDIO: DP.AudioProcessor,
                   ModalityType.TACTILE: DP.TactileProcessor, ModalityType.EEG: DP.EEGProcessor}[m]
            self.processors[m] = cls()
        return self.processors[m]

    @torch.no_grad()
    def encode(self, input_dict: Dict[str, object], normalize: bool = True) -> Dict[str, torch.Tensor]:
        out = {}
        for m, x in input_dict.items():
            model = self.vitlens[m]
            if not isinstance(x, torch.Tensor):
                x = self.processor(m)(x, device=self._dev)
            if m == ModalityType.TEXT:
                f = model.encode_text(x.to(self._dev), normalize=False)
            elif m == ModalityType.IMAGE:
                f = model.encode_image(x.to(self._dev), normalize=False)
            elif self.reduce_list(m) and x.ndim == 4:
                B, S = x.shape[:2]
                f = model(x.reshape(B * S, *x.shape[2:]).to(self._dev))
                f = f.reshape(B, S, -1).mean(dim=1).contiguous()
            else:
                f = model(x.to(self._dev))
            if normalize:
                from vitlens_hip import ops
                f = ops.l2_normalize(f.contiguous().float())
            out[m] = f
        return out
