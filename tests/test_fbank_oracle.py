"""CPU: the numpy restatement of the Kaldi log-mel filterbank (oracle/fbank_oracle.py; reference call site
open_clip/modal_audio/processors/at_processor.py:854-873).  torchaudio is not installed (parity UNPINNED - see the oracle's
header), so what is checked here is the algorithm's own closed-form behaviour and the agreement of the product's
host-built tables with the oracle's independent formulation."""
import math

import numpy as np

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
