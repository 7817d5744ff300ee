"""Evaluation-time audio preprocessing on the GPU (reference: AudioASTProcessorEval,
open_clip/modal_audio/processors/at_processor.py:823-903): a waveform -> `n_clip` clips of `clip_duration` seconds ->
Kaldi log-mel filterbank [512, 128] per clip -> (x - mean) / std.  The spectrogram runs in vl_kaldi_fbank; the clip
selection is host arithmetic on sample indices.

Inputs: a waveform tensor [channels, n] / [n] at `sampling_rate`, or the path of a PCM .wav file at that rate (read with the
standard library; torchaudio - the reference's loader and resampler - is not available, other rates are refused).
Clip placement for recordings longer than one clip: `clips_per_video` windows spread uniformly from the start to
(duration - clip_duration), which is what pytorchvideo's ConstantClipsPerVideoSampler computes for the reference."""
from fractions import Fraction

import torch

AST_AS_MEAN = (-4.2677393,)
AST_AS_STD = (4.5689974,)


def read_wav(path):
    """-> (waveform [channels, n] float32 in [-1, 1), sample rate): PCM 16 / 32-bit little-endian .wav, channel-major like
    torchaudio.load (the reference keeps every channel: `audio_get_clip` subtracts the mean over all of them and
    kaldi.fbank reads channel 0, at_processor.py:193-224,855-866)."""
    import wave

    import numpy as np
    if not str(path).lower().endswith((".wav", ".wave")):
        # the reference decodes through torchaudio's backends (at_processor.py:226-244); this image has no flac / mp3 decoder
        raise NotImplementedError(f"{path}: only PCM .wav is decoded here - convert the recording, or pass the waveform / "
                                  "the [clips, 512, 128] fbank tensor")
    with wave.open(str(path), "rb") as f:
        sr, ch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise NotImplementedError(f"{width * 8}-bit wav")
    return torch.from_numpy(np.ascontiguousarray(a.reshape(-1, ch).T)), sr


def clip_timepoints(duration: float, clip_duration: float, clips_per_video: int):
    """(start, end) seconds of every clip, uniformly spread over [0, duration - clip_duration], as exact Fractions - what
    pytorchvideo's ConstantClipsPerVideoSampler hands the reference (at_processor.py:55-65): end = start + clip_duration
    exactly, so `int(end * sr) - int(start * sr)` can never exceed the clip length by rounding start and end separately.
    (Round 3 returned floats: 4 % of the clips of long recordings came out one sample long and 0.4 % one short - a short
    clip is doubled and cropped at a RANDOM offset by audio_get_clip, i.e. non-deterministic eval input.)"""
    last = Fraction(max(duration - clip_duration, 0.0))
    step = last / max(clips_per_video - 1, 1)
    return [(step * i, step * i + Fraction(clip_duration)) for i in range(clips_per_video)]


def audio_get_clip(waveform, sampling_rate, target_duration, start=None, end=None, sub_mean=True, rng=None):
    """at_processor.py:193-224: cut [start, end), repeat short recordings up to 2^6 times, crop to the target length
    (random offset, `rng.randint`), subtract the clip mean."""
    import random
    wf = waveform
    dur = float(waveform.shape[1] / sampling_rate)
    if start is not None and end is not None and start < dur and end <= dur and end - start > 0.5:
        wf = wf[:, int(start * sampling_rate):int(end * sampling_rate)]
    target = int(sampling_rate * target_duration)
    rep = 0
    while wf.shape[1] < target and rep <= 5:
        wf = torch.cat([wf, wf], dim=1)
        rep += 1
    if rep > 5:
        raise ValueError(f"Original duration {dur} too short, please skip.")
    if wf.shape[1] > target:
        s = (rng or random).randint(0, (wf.shape[1] - 1) - target)
        wf = wf[:, s:s + target]
    return wf - wf.mean() if sub_mean else wf


class AudioASTProcessorEval:
    def __init__(self, mean=AST_AS_MEAN, std=AST_AS_STD, sampling_rate=16000, clip_duration=5.0, n_clip=3, target_length=512,
                 mel_bins=128, device="cuda"):
        self.mean = mean if mean is not None else AST_AS_MEAN
        self.std = std if std is not None else AST_AS_STD
        self.sampling_rate, self.clip_duration, self.n_clip = sampling_rate, clip_duration, n_clip
        self.target_length, self.mel_bins, self.device = target_length, mel_bins, device

    def convert2fbank(self, waveform):
        """[1, n] or [B, n] waveform -> normalised [B, target_length, mel_bins] on the GPU."""
        from vitlens_hip.audio import kaldi_fbank
        return kaldi_fbank(waveform.to(self.device), target_length=self.target_length, mel_bins=self.mel_bins,
                           sample_freq=float(self.sampling_rate), mean=float(self.mean[0]), std=float(self.std[0]))

    def __call__(self, item, **kwargs):
        if isinstance(item, str):
            wav, sr = read_wav(item)
        else:
            wav, sr = torch.as_tensor(item, dtype=torch.float32), self.sampling_rate
            wav = wav[None] if wav.dim() == 1 else wav
        if sr != self.sampling_rate:
            raise NotImplementedError(f"resampling {sr} -> {self.sampling_rate} Hz (torchaudio.functional.resample) is not available: "
                                      "provide audio at the model's sampling rate")
        dur = wav.shape[1] / self.sampling_rate
        if dur <= self.clip_duration:
            clips = [audio_get_clip(wav, self.sampling_rate, self.clip_duration)] * self.n_clip
        else:
            clips = [audio_get_clip(wav, self.sampling_rate, self.clip_duration, start=s, end=e)
                     for s, e in clip_timepoints(dur, self.clip_duration, self.n_clip)]
        # kaldi.fbank takes channel 0 of each clip (its default `channel=-1` -> 0); the clip mean above was over all channels
        return self.convert2fbank(torch.cat([c[:1] for c in clips], dim=0))            # [n_clip, target_length, mel_bins]
