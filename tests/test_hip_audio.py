"""GPU: vl_kaldi_fbank and the audio evaluation processor (reference AudioASTProcessorEval,
open_clip/modal_audio/processors/at_processor.py:823-903) against the numpy oracle (oracle/fbank_oracle.py; parity
unpinned: torchaudio is not installed, the oracle restates the published Kaldi algorithm)."""
import math
import os
import wave

import numpy as np
import pytest
import torch

import fbank_oracle as F

pytestmark = pytest.mark.gpu


def _close(got, ref, atol=2e-3):
    """log-mel values; where the energy sits at the float32-epsilon floor both sides must sit there."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    floor = (math.log(float(F.EPS)) + 4.2677393) / 4.5689974
    live = ref > floor + 0.5
    assert np.abs(got - ref)[live].max() < atol, np.abs(got - ref)[live].max()
    assert (got[~live] < floor + 1.0).all()


def test_kaldi_fbank_vs_oracle():
    from vitlens_hip.audio import kaldi_fbank
    rng = np.random.default_rng(0)
    t = np.arange(80000) / 16000.0
    waves = [rng.standard_normal(80000).astype(np.float32) * 0.1,
             (0.4 * np.sin(2 * math.pi * 1000 * t) + 0.2 * np.sin(2 * math.pi * 3333 * t)).astype(np.float32),
             (rng.standard_normal(80000) * np.linspace(0, 1, 80000)).astype(np.float32) * 0.3]
    w = torch.tensor(np.stack(waves)).cuda()
    out = kaldi_fbank(w, mean=-4.2677393, std=4.5689974).cpu().numpy()
    assert out.shape == (3, 512, 128)
    for i, wv in enumerate(waves):
        _close(out[i], F.ast_spectrogram(wv))
    # a 1 s clip: 98 frames, the rest is the normalised zero row; and a raw (un-normalised) call
    short = kaldi_fbank(w[0, :16000], mean=-4.2677393, std=4.5689974).cpu().numpy()[0]
    _close(short, F.ast_spectrogram(waves[0][:16000]))
    raw = kaldi_fbank(w[0, :16000], target_length=98).cpu().numpy()[0]
    ref = F.fbank(waves[0][:16000])
    live = ref > -12
    assert np.abs(raw - ref)[live].max() < 5e-3


def test_audio_processor_on_waveform_and_wav_file(tmp_path):
    from open_clip.modal_audio.processors.at_processor import AudioASTProcessorEval, clip_timepoints
    rng = np.random.default_rng(3)
    sr = 16000
    long = (rng.standard_normal(sr * 12) * 0.1).astype(np.float32)          # 12 s: three clips spread over the recording
    proc = AudioASTProcessorEval()
    out = proc(torch.tensor(long)).cpu().numpy()
    assert out.shape == (3, 512, 128)
    pts = clip_timepoints(12.0, 5.0, 3)
    assert pts == [(0.0, 5.0), (3.5, 8.5), (7.0, 12.0)]
    for i, (s, e) in enumerate(pts):
        clip = long[int(s * sr):int(e * sr)]
        _close(out[i], F.ast_spectrogram(clip - clip.mean()))
    # a short recording is repeated to the clip length and used for every clip
    short = (rng.standard_normal(sr * 2) * 0.1).astype(np.float32)
    o2 = proc(torch.tensor(short)[None]).cpu().numpy()
    rep = np.concatenate([short] * 4)[:sr * 5] if False else None
    assert o2.shape == (3, 512, 128) and np.array_equal(o2[0], o2[1]) and np.array_equal(o2[1], o2[2])
    # 16-bit PCM file through the path entry
    path = os.path.join(tmp_path, "a.wav")
    pcm = np.clip(long[:sr * 5] * 32768.0, -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr); f.writeframes(pcm.tobytes())
    o3 = proc(path).cpu().numpy()
    wav = pcm.astype(np.float32) / 32768.0
    _close(o3[0], F.ast_spectrogram(wav - wav.mean()))
    from mm_vit_lens.data_processors import AudioProcessor
    batch = AudioProcessor()([path, torch.tensor(long)], device="cuda")
    assert batch.shape == (2, 3, 512, 128) and batch.is_cuda
