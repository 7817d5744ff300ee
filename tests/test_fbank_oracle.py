"""CPU: the numpy restatement of the Kaldi log-mel filterbank (oracle/fbank_oracle.py; reference call site
open_clip/modal_audio/processors/at_processor.py:854-873).  torchaudio is not installed, so the restatement cannot meet the
library itself; it is checked against the algorithm's own closed-form behaviour, against the product's host-built tables,
and (round 4) against the independent numpy implementation of the same call that ships with transformers."""
import math

import numpy as np
import pytest

import fbank_oracle as F


def test_framing_and_shapes():
    for n, frames in ((400, 1), (559, 1), (560, 2), (80000, 498), (16000, 98)):
        assert F.fbank(np.zeros(n, np.float32) + 1e-3).shape == (frames, 128)            # 1 + (n - 400) // 160, snip_edges
    out = F.ast_spectrogram(np.random.default_rng(0).standard_normal(16000).astype(np.float32) * 0.1)
    assert out.shape == (512, 128)
    pad = (0.0 - (-4.2677393)) / 4.5689974
    assert np.allclose(out[98:], pad, atol=1e-6)                                         # rows beyond the clip: Normalize(0)


def test_pure_tone_lands_in_its_mel_bin():
    sr = 16000
    t = np.arange(sr) / sr
    lo, hi = F.mel_scale(20.0), F.mel_scale(8000.0)
    for hz in (440.0, 1000.0, 3000.0, 6000.0):
        fb = F.fbank((0.5 * np.sin(2 * math.pi * hz * t)).astype(np.float32))
        want = (F.mel_scale(hz) - lo) / ((hi - lo) / 129) - 1.0                          # fractional index of the filter centred on hz
        assert abs(int(np.median(fb.argmax(axis=1))) - want) <= 1.0, (hz, want)


def test_constant_signal_is_removed_by_dc_offset_and_preemphasis():
    fb = F.fbank(np.full(4000, 0.7, np.float32))
    assert np.allclose(fb, math.log(float(F.EPS)), atol=1e-5)                            # every energy at the floor


def test_mel_banks_properties_and_host_tables():
    b = F.mel_banks()
    assert b.shape == (128, 257) and float(b[:, 256].max()) == 0.0 and float(b.min()) >= 0.0 and float(b.max()) <= 1.0
    # neighbouring triangles overlap so that interior FFT bins are covered with total weight 1
    cover = b.sum(axis=0)
    inner = slice(2, 250)
    assert np.allclose(cover[inner], 1.0, atol=1e-5)
    from vitlens_hip.audio import mel_filter_matrix
    assert np.array_equal(mel_filter_matrix(128, 512, 16000.0), b)
    assert np.allclose(mel_filter_matrix(64, 1024, 22050.0), F.mel_banks(64, 1024, 22050.0), atol=1e-7)


def test_energy_scales_quadratically():
    rng = np.random.default_rng(1)
    w = rng.standard_normal(8000).astype(np.float32) * 0.05
    a, b = F.fbank(w), F.fbank(2.0 * w)
    live = a > -10.0
    assert np.allclose((b - a)[live], math.log(4.0), atol=2e-4)


def test_oracle_equals_an_independent_kaldi_fbank_implementation():
    """torchaudio (the reference's dependency for this call, at_processor.py:854-873) is not in the image; Hugging Face
    transformers (5.15.0 here) ships its own numpy implementation of the SAME computation for exactly this case:
    ASTFeatureExtractor without torchaudio evaluates `spectrogram(..., frame_length=400, hop_length=160, fft_length=512,
    power=2.0, center=False, preemphasis=0.97, mel_filters=kaldi-scale triangles from 20 Hz, log_mel="log",
    mel_floor=1.192092955078125e-07, remove_dc_offset=True)` with a non-periodic Hann window - its authors' replacement for
    `ta_kaldi.fbank(waveform, sample_frequency=16000, window_type="hanning", num_mel_bins=128)`, the reference's call
    (htk_compat only matters with use_energy; dither 0 and frame_shift 10 are the defaults).  The restatement in
    oracle/fbank_oracle.py was written from the Kaldi / torchaudio source without looking at it: two independent readings
    that agree to 1e-4 in the log domain on noise, tones, a slowly modulated signal and near-silence."""
    tr = pytest.importorskip("transformers")
    from transformers.models.audio_spectrogram_transformer.feature_extraction_audio_spectrogram_transformer import ASTFeatureExtractor
    from transformers.utils import is_speech_available
    if is_speech_available():
        pytest.skip("torchaudio present: the extractor would call it instead of its own implementation")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fe = ASTFeatureExtractor(num_mel_bins=128, max_length=1024, do_normalize=False)
    rs = np.random.RandomState(0)
    cases = []
    for n in (16000, 48000, 20000 + 37):
        t = np.arange(n) / 16000.0
        cases += [(rs.randn(n) * 0.1).astype(np.float32),
                  (0.3 * np.sin(2 * math.pi * 440 * t) + 0.1 * np.sin(2 * math.pi * 3000 * t)).astype(np.float32),
                  (rs.randn(n).cumsum() * 1e-3 * np.sin(2 * math.pi * 3 * t)).astype(np.float32),
                  (rs.randn(n) * 1e-6).astype(np.float32)]
    worst = 0.0
    for x in cases:
        a = F.fbank(x)
        b = fe._extract_fbank_features(x, max_length=a.shape[0])
        assert a.shape == b.shape == (1 + (len(x) - 400) // 160, 128)
        worst = max(worst, float(np.abs(a - b).max()))
    assert worst < 2e-4, worst
