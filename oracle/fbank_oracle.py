"""TEST INFRASTRUCTURE (oracle): numpy restatement of the Kaldi log-mel filterbank as
`torchaudio.compliance.kaldi.fbank` computes it for the reference's call
(open_clip/modal_audio/processors/at_processor.py:854-873: htk_compat=True, sample_frequency=16000, use_energy=False,
window_type="hanning", num_mel_bins=128, dither=0.0, frame_shift=10; all other arguments at their defaults: 25 ms frames,
snip_edges, remove_dc_offset, preemphasis 0.97, round_to_power_of_two, low_freq 20 Hz, high_freq 0 = Nyquist, power
spectrum, natural log floored at float32 epsilon), plus the processor's pad / truncate to target_length rows and
transforms.Normalize(mean, std).

PARITY: torchaudio is a third-party dependency of the reference that is absent from /root/reference and from this image
(requirements: torchaudio matching torch >= 1.9, README pins 0.11 / 0.12), so this file restates the PUBLISHED algorithm
(Kaldi feature-fbank / torchaudio.compliance.kaldi source) and cannot be run against the library itself: by the letter of
the contract it stays "parity unpinned" against the reference's dependency.  Since round 4 it is CROSS-PINNED against an
independent implementation of the same call that IS in the image: Hugging Face transformers 5.15.0, whose
ASTFeatureExtractor falls back to its own numpy `spectrogram` / kaldi-scale `mel_filter_bank` when torchaudio is missing -
written by its authors to replace `ta_kaldi.fbank(waveform, sample_frequency=16000, window_type="hanning", num_mel_bins=128)`,
the reference's call.  tests/test_fbank_oracle.py: agreement within 1e-4 (log domain) on noise, tones, a modulated signal
and near-silence.  Also pinned: numpy's FFT as the spectrum, and the closed-form properties the tests check (a pure tone
peaks in the right mel bin, Parseval on the power spectrum, shift invariance of the framing)."""
import math

import numpy as np

EPS = np.float32(1.1920928955078125e-07)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_banks(num_bins=128, padded_window=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    """get_mel_banks (vtln_warp = 1): [num_bins, padded_window/2 + 1] float32, the last (Nyquist) column zero - Kaldi's
    filters cover FFT bins 0 .. N/2 - 1 and torchaudio pads one zero column."""
    num_fft_bins = padded_window // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded_window
    mel_low, mel_high = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    mel = mel_scale(fft_bin_width * np.arange(num_fft_bins, dtype=np.float64))[None, :]
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    banks = np.maximum(0.0, np.minimum(up, down))
    return np.pad(banks, ((0, 0), (0, 1))).astype(np.float32)


def hann_window(n=400):
    """torch.hann_window(n, periodic=False)."""
    return (0.5 - 0.5 * np.cos(2.0 * math.pi * np.arange(n, dtype=np.float64) / (n - 1))).astype(np.float32)


def fbank(wave, sample_freq=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0, num_mel_bins=128, preemph=0.97):
    """wave [n] float -> [frames, num_mel_bins] float32 log-mel energies."""
    wave = np.asarray(wave, dtype=np.float32)
    win, shift = int(sample_freq * frame_length_ms * 0.001), int(sample_freq * frame_shift_ms * 0.001)
    nfft = 1 << (win - 1).bit_length()
    m = 1 + (len(wave) - win) // shift                                  # snip_edges
    idx = np.arange(win)[None, :] + shift * np.arange(m)[:, None]
    fr = wave[idx].astype(np.float64)
    fr = fr - fr.mean(axis=1, keepdims=True)                            # remove_dc_offset
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)              # replicate-padded shift
    fr = (fr - preemph * prev) * hann_window(win).astype(np.float64)[None, :]
    spec = np.abs(np.fft.rfft(fr, n=nfft, axis=1)) ** 2                 # power spectrum
    mel = spec @ mel_banks(num_mel_bins, nfft, sample_freq).astype(np.float64).T
    return np.log(np.maximum(mel, float(EPS))).astype(np.float32)


def ast_spectrogram(wave, target_length=512, mean=-4.2677393, std=4.5689974, **kw):
    """convert2fbank + the eval transform of AudioASTProcessorEval (at_processor.py:839-873): zero-pad / truncate to
    target_length frames, then (x - mean) / std."""
    fb = fbank(wave, **kw)
    out = np.zeros((target_length, fb.shape[1]), dtype=np.float32)
    n = min(target_length, fb.shape[0])
    out[:n] = fb[:n]
    return ((out - np.float32(mean)) / np.float32(std)).astype(np.float32)
