"""Audio front end on the GPU: Kaldi-compatible log-mel filterbank (vl_kaldi_fbank, csrc/vl_audio.hip) with the window and
the mel filter matrix built on the host from the published formulas (Kaldi feature-window / mel-computations, the ones
torchaudio.compliance.kaldi implements), cached per (device, geometry).  Reference call site:
AudioASTProcessorEval.convert2fbank, open_clip/modal_audio/processors/at_processor.py:854-873."""
import math

import numpy as np
import torch

from . import ops
from ._lib import check
from .ops import _lib, _p, _stream

_tables = {}


def _mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_filter_matrix(num_bins: int, nfft: int, sample_freq: float, low_freq: float = 20.0, high_freq: float = 0.0) -> np.ndarray:
    """Triangular filters, equally spaced on the mel scale between low_freq and high_freq (<= 0: relative to Nyquist), over
    the FFT bins 0 .. nfft/2 - 1; one zero column for the Nyquist bin.  [num_bins, nfft/2 + 1] float32."""
    half = nfft // 2
    hi = high_freq + 0.5 * sample_freq if high_freq <= 0.0 else high_freq
    lo_m, hi_m = _mel(low_freq), _mel(hi)
    step = (hi_m - lo_m) / (num_bins + 1)
    edges = lo_m + step * np.arange(num_bins + 2, dtype=np.float64)
    bin_mel = _mel(sample_freq / nfft * np.arange(half, dtype=np.float64))
    left, mid, right = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    tri = np.minimum((bin_mel[None, :] - left) / (mid - left), (right - bin_mel[None, :]) / (right - mid))
    out = np.zeros((num_bins, half + 1), dtype=np.float32)
    out[:, :half] = np.clip(tri, 0.0, None)
    return out


def _device_tables(device, win, nfft, nmel, sample_freq):
    key = (str(device), win, nfft, nmel, sample_freq)
    if key not in _tables:
        n = np.arange(win, dtype=np.float64)
        window = (0.5 - 0.5 * np.cos(2.0 * math.pi * n / (win - 1))).astype(np.float32)          # "hanning", symmetric
        _tables[key] = (torch.from_numpy(window).to(device), torch.from_numpy(mel_filter_matrix(nmel, nfft, sample_freq)).to(device))
    return _tables[key]


def kaldi_fbank(wave: torch.Tensor, target_length: int = 512, mel_bins: int = 128, sample_freq: float = 16000.0,
                frame_length_ms: float = 25.0, frame_shift_ms: float = 10.0, preemph: float = 0.97,
                mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    """wave [B, n] (or [n]) f32 on the GPU -> [B, target_length, mel_bins] f32: log-mel energies of the first
    target_length frames (zero rows beyond the clip's frames), normalised by (x - mean) / std."""
    if wave.device.type != "cuda":
        raise RuntimeError("kaldi_fbank runs on the MI355X kernels only")
    w = wave.reshape(1, -1) if wave.dim() == 1 else wave
    w = w.contiguous().float()
    win, shift = int(sample_freq * frame_length_ms * 0.001), int(sample_freq * frame_shift_ms * 0.001)
    nfft = 1 << (win - 1).bit_length()
    window, banks = _device_tables(w.device, win, nfft, mel_bins, float(sample_freq))
    out = torch.empty(w.shape[0], target_length, mel_bins, device=w.device, dtype=torch.float32)
    check(_lib.vl_kaldi_fbank(_p(w), w.stride(0), w.shape[0], w.shape[1], _p(window), _p(banks), _p(out), target_length, win, shift,
                              nfft, mel_bins, float(preemph), float(mean), float(std), _stream()))
    return out
