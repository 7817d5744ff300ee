"""CPU: the per-modality input processors of `ViTLens.encode` (mm_vit_lens/data_processors.py, reference
mm_vit_lens/data_processors.py:15-323) - host logic only: caption cleaning, EEG resampling against scipy, argument
plumbing, loud failures where a front end is not available.  The GPU side is in tests/test_hip_preproc.py."""
import numpy as np
import pytest
import torch


def test_pre_caption_matches_the_reference_rules():
    from mm_vit_lens.data_processors import TextProcessor
    tp = TextProcessor()
    # data_processors.py:68-87: lower-case, the characters . ! " ( ) * # : ; ~ become blanks, runs of blanks collapse,
    # trailing newline / surrounding blanks go, at most max_words words survive
    assert tp.pre_caption('A Dog!  (running): "fast"; ~ok~ #1.\n') == "a dog running fast ok 1"
    assert tp.pre_caption("  sea   wave ") == "sea wave"
    assert tp.pre_caption("keep, commas & dashes - and? marks") == "keep, commas & dashes - and? marks"
    assert TextProcessor(max_words=3).pre_caption("one two three four five") == "one two three"
    ids = TextProcessor(prompt="a photo of ")(["A Bird.", "sea wave"])
    from open_clip import tokenize
    assert ids.shape == (2, 77) and torch.equal(ids, tokenize(["a photo of a bird", "a photo of sea wave"]))
    assert TextProcessor()(None) is None
    tp2 = TextProcessor.from_config({"prompt": "x ", "max_words": 5})
    assert tp2.prompt == "x " and tp2.max_words == 5


def test_eeg_processor_equals_scipy_interp1d(tmp_path):
    interp1d = pytest.importorskip("scipy.interpolate").interp1d
    from mm_vit_lens.data_processors import EEGProcessor
    g = torch.Generator().manual_seed(0)
    e = torch.randn(128, 500, generator=g)
    eeg = e.float().t()[20:460, :]                                                 # eeg_processor.py:236-246, verbatim order
    eeg = np.array(eeg.transpose(0, 1))
    want = torch.from_numpy(interp1d(np.linspace(0, 1, eeg.shape[-1]), eeg)(np.linspace(0, 1, 512))).float()
    path = tmp_path / "eeg.pth"
    torch.save(e, path)
    got = EEGProcessor()([str(path), e])
    assert got.shape == (2, 128, 512) and torch.equal(got[0], want) and torch.equal(got[1], want)


def test_audio_processor_takes_spectrograms_and_refuses_waveform_files():
    from mm_vit_lens.data_processors import AudioProcessor
    ap = AudioProcessor()
    x = ap([torch.zeros(3, 512, 128), np.ones((3, 512, 128), np.float32)])
    assert x.shape == (2, 3, 512, 128) and x.dtype == torch.float32
    with pytest.raises(NotImplementedError):
        ap(["clip.flac"])
    with pytest.raises(ValueError):
        ap([torch.zeros(3, 100, 128)])


def test_processor_table_and_vitlens_wiring():
    from mm_vit_lens import data_processors as DP
    from mm_vit_lens.vitlens import ViTLens
    from open_clip import ModalityType
    procs = DP.get_vitlens_processors_cls()["vitlensL"]()
    assert set(procs) == {"image", "text", "pc", "depth", "audio", "tactile", "eeg"}
    assert procs["pc"].wrap_processor.npoint == 8192 and procs["pc"].wrap_processor.uniform
    assert procs["depth"].wrap_processor.max_depth == 75.0 and procs["depth"].wrap_processor.mean[3] == 0.0418
    assert procs["tactile"].wrap_processor.resize == 256 and procs["tactile"].wrap_processor.size == 224
    assert procs["image"].transform.image_size == 224 and not procs["image"].transform.is_train
    assert DP.get_vitlens_processors_cls()["vitlensB"]() is None
    vl = ViTLens(modality_loaded=[ModalityType.TACTILE, ModalityType.EEG], device="cpu")
    assert isinstance(vl.processor(ModalityType.EEG), DP.EEGProcessor) and isinstance(vl.processor("tactile"), DP.TactileProcessor)
    assert vl.processor("eeg") is vl.processor("eeg")
    with pytest.raises(RuntimeError):                                             # no GPU here: the towers refuse CPU tensors loudly
        vl.encode({ModalityType.EEG: torch.zeros(1, 128, 512)})


_REF_PRE = r'''
import json, sys, types
sys.path.insert(0, sys.argv[1])
import ref_loader
ref_loader.load()
sys.modules.setdefault("omegaconf", types.SimpleNamespace(OmegaConf=types.SimpleNamespace(create=lambda *a, **k: {})))
from mm_vit_lens.data_processors import TextProcessor
from easydict import EasyDict
tp = TextProcessor(cfg=EasyDict(model="ViT-L-14"))
tp5 = TextProcessor(max_words=5, cfg=EasyDict(model="ViT-L-14"))
texts = json.load(open(sys.argv[2]))
print("JSON" + json.dumps({"pre": [tp.pre_caption(t) for t in texts], "pre5": [tp5.pre_caption(t) for t in texts],
                           "ids": tp(texts[:64]).tolist()}))
'''


def test_text_processor_against_the_reference(tmp_path):
    """TextProcessor.pre_caption / __call__ of the imported reference (build container only) on the tokenizer fuzz set."""
    import json
    import os
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/vitlens/src/mm_vit_lens"):
        pytest.skip("/root/reference not present")
    from test_tokenizer import _fuzz_texts
    from mm_vit_lens.data_processors import TextProcessor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    texts = _fuzz_texts(600, seed=1)
    p = tmp_path / "texts.json"
    json.dump(texts, open(p, "w"))
    r = subprocess.run([sys.executable, "-c", _REF_PRE, os.path.join(root, "oracle"), str(p)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    tp, tp5 = TextProcessor(), TextProcessor(max_words=5)
    assert [tp.pre_caption(t) for t in texts] == ref["pre"]
    assert [tp5.pre_caption(t) for t in texts] == ref["pre5"]
    assert tp(texts[:64]).tolist() == ref["ids"]


def test_audio_clips_have_exactly_the_target_length_and_keep_channels(tmp_path):
    """Host logic of AudioASTProcessorEval (reference at_processor.py:55-65,193-224,876-903): clip boundaries are exact
    Fractions (end = start + clip_duration), so every clip of a long recording has exactly sr * clip_duration samples -
    never one short (which would be doubled and cropped at a random offset) - and a multi-channel .wav keeps its channels:
    the clip mean is over all of them, the spectrogram reads channel 0."""
    import random
    import wave
    from fractions import Fraction

    import numpy as np
    from open_clip.modal_audio.processors import at_processor as AP
    sr, cd = 16000, 5.0
    rnd = random.Random(7)
    bad = 0
    for _ in range(3000):
        n = rnd.randint(int(sr * cd) + 1, sr * 600)
        dur = n / sr
        for s, e in AP.clip_timepoints(dur, cd, 3):
            assert isinstance(s, Fraction) and e - s == Fraction(cd)
            bad += (int(e * sr) - int(s * sr)) != int(sr * cd)
            assert 0 <= s and e <= dur + 1e-9
    assert bad == 0
    assert AP.clip_timepoints(12.0, 5.0, 3) == [(0, 5), (Fraction(7, 2), Fraction(17, 2)), (7, 12)]
    # audio_get_clip on those boundaries: exactly the slice, deterministic (no repetition, no random crop)
    wf = torch.arange(sr * 13, dtype=torch.float32)[None] / sr
    for s, e in AP.clip_timepoints(13.0, cd, 3):
        c = AP.audio_get_clip(wf, sr, cd, start=s, end=e, sub_mean=False)
        assert c.shape == (1, int(sr * cd)) and float(c[0, 0]) == float(wf[0, int(s * sr)])
    # stereo file -> [2, n]; the processor's clips keep both channels for the mean
    path = str(tmp_path / "st.wav")
    pcm = (np.stack([np.arange(1000), -2 * np.arange(1000)], 1) % 3000).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(sr); f.writeframes(pcm.tobytes())
    w, r = AP.read_wav(path)
    assert r == sr and w.shape == (2, 1000)
    assert torch.equal(w[0], torch.from_numpy(pcm[:, 0].astype(np.float32) / 32768.0))
    clip = AP.audio_get_clip(w.repeat(1, 100), sr, cd)
    assert clip.shape[0] == 2 and abs(float(clip.mean())) < 1e-6
