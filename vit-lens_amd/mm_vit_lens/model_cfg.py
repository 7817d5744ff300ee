"""ViT-Lens-L modality configurations (values of the reference's mm_vit_lens/model_cfg.py:80-178: these are the
released model's hyper-parameters, i.e. data)."""
from types import SimpleNamespace

_COMMON = dict(use_perceiver=True, use_visual_adapter=True, visual_arch="perceiver_vit", disable_orig_pos=False,
               disable_visual_adapter_pos=False, perceiver_as_identity=False, perceiver_as_transformer=False,
               perceiver_input_axis=1, perceiver_num_freq_bands=32, perceiver_max_freq=10.0, perceiver_num_classes=1000,
               perceiver_attn_dropout=0.0, perceiver_ff_dropout=0.0, perceiver_weight_tie_layers=False,
               perceiver_fourier_encode_data=False, perceiver_cross_heads=1, perceiver_cross_dim_head=64,
               perceiver_latent_heads=16, perceiver_latent_dim_head=64, perceiver_latent_dim=1024,
               perceiver_num_latents=256, perceiver_depth=1, perceiver_self_per_cross_attn=1, perceiver_input_chan=1024,
               skip_trans_first_n_layers=None, unlock_from_head=False)

_MODALITY = {
    "pc": dict(visual_modality_type="3dpc", v_key="pc", pc_tokenizer="pointbert", pc_encoder_dims=256, pc_group_size=32,
               pc_npoints=8192, pc_num_group=512, pc_trans_dim=384, pc_in_channel=3, pc_radius=0.2, perceiver_depth=4,
               perceiver_input_chan=384, perceiver_self_per_cross_attn=1),
    "audio": dict(visual_modality_type="audio", v_key="audio", audio_clip_duration=5.0, audio_sampling_rate=16000,
                  audio_fstride=10, audio_tstride=10, audio_mel_bins=128, audio_target_length=512, perceiver_depth=2,
                  perceiver_input_chan=1024, perceiver_self_per_cross_attn=3),
    "depth": dict(visual_modality_type="depth", v_key="depth", perceiver_as_identity=True),
    "tactile": dict(visual_modality_type="tactile", v_key="tactile", use_perceiver=False, use_visual_adapter=False),
    "eeg": dict(visual_modality_type="eeg", v_key="eeg", eeg_chans=128, eeg_stride=1, eeg_time_len=512, eeg_window_size=1,
                perceiver_depth=1, perceiver_input_chan=1024, perceiver_self_per_cross_attn=1),
    "image": dict(visual_modality_type="image", v_key="image", use_perceiver=False, use_visual_adapter=False),
}

MODEL = {"vitlensL": dict(model="ViT-L-14", pretrained="datacomp_xl_s13b_b90k")}


def fetch_model_cfg(modality="pc", model_option="vitlensL"):
    if model_option not in MODEL:
        raise NotImplementedError(model_option)
    d = dict(_COMMON); d.update(MODEL[model_option]); d.update(_MODALITY[modality])
    return SimpleNamespace(**d)
