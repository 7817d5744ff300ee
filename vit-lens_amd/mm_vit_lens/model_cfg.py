"""ViT-Lens-L modality configurations (values of the reference's mm_vit_lens/model_cfg.py:8-178: these are the
released model's hyper-parameters, i.e. data).  As in the reference every configuration carries the FULL default set
(an audio configuration also answers `cfg.pc_npoints`), overridden per modality; `unlock_from_head` and the `vid_*`
switches are added because `TriCLIP.forward` / `lock` read them and the reference's defaults omit them."""
from types import SimpleNamespace

_DEFAULT = dict(
    audio_clip_duration=5.0, audio_fstride=10, audio_mel_bins=128, audio_sampling_rate=16000, audio_target_length=512,
    audio_tstride=10, cache_dir="/PATH_TO/CACHE/DIR", dataset_type="image", device="cpu", disable_orig_pos=False,
    disable_pt_vit=False, disable_visual_adapter_pos=False, eeg_chans=128, eeg_stride=1, eeg_time_len=512,
    eeg_window_size=1, force_custom_text=False, force_image_size=None, force_patch_dropout=None, force_quick_gelu=False,
    image_mean=None, image_std=None, load_ckpt_strict=False, model="ViT-L-14", pc_encoder_dims=256, pc_group_size=32,
    pc_in_channel=3, pc_npoints=8192, pc_num_group=512, pc_radius=0.2, pc_tokenizer="pointbert", pc_trans_dim=384,
    perceiver_as_identity=False, perceiver_as_transformer=False, perceiver_attn_dropout=0.0, perceiver_cross_dim_head=64,
    perceiver_cross_heads=1, perceiver_depth=1, perceiver_ff_dropout=0.0, perceiver_fourier_encode_data=False,
    perceiver_input_axis=1, perceiver_input_chan=1024, perceiver_latent_dim=1024, perceiver_latent_dim_head=64,
    perceiver_latent_heads=16, perceiver_max_freq=10.0, perceiver_num_classes=1000, perceiver_num_freq_bands=32,
    perceiver_num_latents=256, perceiver_self_per_cross_attn=1, perceiver_weight_tie_layers=False, precision="fp32",
    pretrained="datacomp_xl_s13b_b90k", pretrained_image=False, skip_trans_first_n_layers=None, torchcompile=False,
    torchscript=False, trace=False, use_bn_sync=False, use_bnb_linear=None, use_eva_pt_lin=False,
    use_openclip_transform=False, use_perceiver=False, use_visual_adapter=False, v_key="image", visual_arch="perceiver_vit",
    visual_modality_type="image",
    unlock_from_head=False, vid_use_fpos=False, vid_use_ltpos=False, vid_distill_tokens=False)

_LENS = dict(use_perceiver=True, use_visual_adapter=True)
_MODALITY = {
    "image": dict(),
    "pc": dict(_LENS, visual_modality_type="3dpc", v_key="pc", perceiver_depth=4, perceiver_input_chan=384,
               ckpt_pth="/PATH_TO/vitlensL_pc.pt"),
    "depth": dict(_LENS, visual_modality_type="depth", v_key="depth", perceiver_as_identity=True,
                  ckpt_pth="/PATH_TO/vitlensL_depth.pt"),
    "audio": dict(_LENS, visual_modality_type="audio", v_key="audio", perceiver_depth=2, perceiver_self_per_cross_attn=3,
                  ckpt_pth="/PATH_TO/vitlensL_audio.pt"),
    "tactile": dict(visual_modality_type="tactile", v_key="tactile", ckpt_pth="/PATH_TO/vitlensL_tactile.pt"),
    "eeg": dict(_LENS, visual_modality_type="eeg", v_key="eeg", ckpt_pth="/PATH_TO/vitlensL_eeg.pt"),
}

MODEL = {"vitlensL": dict(model="ViT-L-14", pretrained="datacomp_xl_s13b_b90k")}


def fetch_model_cfg(model_keys=["model", "pretrained"], modality="pc", model_option="vitlensL"):
    """model_cfg.py:185-197: defaults, the listed keys of the model option, then the modality's overrides ("image",
    "video" and "text" have none)."""
    if model_option not in MODEL:
        raise NotImplementedError(model_option)
    d = dict(_DEFAULT)
    for k in model_keys:
        d[k] = MODEL[model_option][k]
    if modality not in ("image", "video", "text"):
        d.update(_MODALITY[modality])
    return SimpleNamespace(**d)
