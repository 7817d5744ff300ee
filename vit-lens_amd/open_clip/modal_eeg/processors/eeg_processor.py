"""EEG preprocessing (reference: open_clip/modal_eeg/processors/eeg_processor.py:229-247 `EEGProcessorEval`): a
[channels, time] recording is cut to the window [time_low, time_high) and linearly resampled to `data_len` points per
channel (scipy `interp1d` over a [0, 1] grid in the reference).  128 x 512 values per sample: host-side data plumbing,
done here with the same two-point formula in numpy (pinned against scipy in tests/test_preproc_host.py); the result is
the [channels, data_len] float32 tensor the EEG tokenizer (PatchEmbed1D) takes."""
import numpy as np
import torch


class BaseProcessor:
    def __init__(self):
        self.transform = lambda x: x

    def __call__(self, item):
        return self.transform(item)

    @classmethod
    def from_config(cls, cfg=None):
        return cls()

    def build(self, **kwargs):
        return self.from_config(dict(kwargs))


def linear_resample(y, n_out):
    """scipy.interpolate.interp1d(linspace(0, 1, T), y)(linspace(0, 1, n_out)) along the last axis (linear, float64)."""
    T = y.shape[-1]
    x, xn = np.linspace(0, 1, T), np.linspace(0, 1, n_out)
    hi = np.clip(np.searchsorted(x, xn), 1, T - 1)
    lo = hi - 1
    slope = (y[..., hi] - y[..., lo]) / (x[hi] - x[lo])
    return slope * (xn - x[lo]) + y[..., lo]


class EEGProcessorEval(BaseProcessor):
    def __init__(self, time_low=20, time_high=460, data_len=512):
        self.time_low, self.time_high, self.data_len = time_low, time_high, data_len

    def __call__(self, eeg):
        """eeg: path of a torch-saved [channels, time] tensor, or the tensor / array itself -> [channels, data_len]."""
        if isinstance(eeg, (str, bytes)) or hasattr(eeg, "__fspath__"):
            eeg = torch.load(eeg, map_location="cpu", weights_only=False)
        eeg = np.asarray(eeg.float().cpu() if isinstance(eeg, torch.Tensor) else eeg, dtype=np.float32)
        eeg = eeg[:, self.time_low:self.time_high]
        return torch.from_numpy(linear_resample(eeg, self.data_len)).float()
