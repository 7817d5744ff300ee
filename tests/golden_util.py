"""Helpers shared by the tests: load committed golden fixtures, build oracle specs."""
import json
import os

import numpy as np
import torch

import vitlens_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def split(case):
    sd = {k[3:]: torch.from_numpy(v) for k, v in case.items() if k.startswith("sd/")}
    ins = {k[3:]: torch.from_numpy(v) for k, v in case.items() if k.startswith("in/")}
    outs = {k[4:]: torch.from_numpy(v) for k, v in case.items() if k.startswith("out/")}
    grads = {k[5:]: torch.from_numpy(v) for k, v in case.items() if k.startswith("grad/")}
    meta = json.loads(bytes(case["meta"]).decode()) if "meta" in case else {}
    return sd, ins, outs, grads, meta


def specs_from_meta(meta):
    cfg, a = meta["model_cfg"], meta["args"]
    v, t = cfg["vision_cfg"], cfg["text_cfg"]
    tower = O.TowerSpec(width=v["width"], layers=v["layers"], heads=v["width"] // v.get("head_width", 64),
                        patch=v["patch_size"], image_size=v["image_size"], embed_dim=cfg["embed_dim"])
    text = O.TextSpec(context_length=t["context_length"], vocab_size=t["vocab_size"], width=t["width"],
                      heads=t["heads"], layers=t["layers"], embed_dim=cfg["embed_dim"])
    mod = {"3dpc": "pc", "tactile": "image"}.get(a["visual_modality_type"], a["visual_modality_type"])
    lens = O.LensSpec(
        modality=mod, perceiver_identity=bool(a.get("perceiver_as_identity", False)),
        depth=a["perceiver_depth"], self_per_cross=a["perceiver_self_per_cross_attn"],
        num_latents=a["perceiver_num_latents"], latent_dim=a["perceiver_latent_dim"],
        input_chan=a["perceiver_input_chan"], cross_heads=a["perceiver_cross_heads"],
        cross_dim_head=a["perceiver_cross_dim_head"], latent_heads=a["perceiver_latent_heads"],
        latent_dim_head=a["perceiver_latent_dim_head"],
        audio_fstride=a.get("audio_fstride", 10), audio_tstride=a.get("audio_tstride", 10),
        audio_mel_bins=a.get("audio_mel_bins", 128), audio_target_length=a.get("audio_target_length", 512),
        eeg_chans=a.get("eeg_chans", 128), eeg_time_len=a.get("eeg_time_len", 512),
        eeg_window_size=a.get("eeg_window_size", 1), eeg_stride=a.get("eeg_stride", 1),
        pc_num_group=a.get("pc_num_group", 512), pc_group_size=a.get("pc_group_size", 32),
        pc_encoder_dims=a.get("pc_encoder_dims", 256), pc_trans_dim=a.get("pc_trans_dim", 384),
        use_orig_pos=not a.get("disable_orig_pos", False),
        disable_adapter_pos=bool(a.get("disable_visual_adapter_pos", False)),
        weight_tie_layers=bool(a.get("perceiver_weight_tie_layers", False)))
    return tower, text, lens
