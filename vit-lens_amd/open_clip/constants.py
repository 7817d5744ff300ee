"""Names the drop-in boundary shares with the reference (open_clip/constants.py:4-16)."""
from types import SimpleNamespace

ModalityType = SimpleNamespace(IMAGE="image", VIDEO="video", TEXT="text", AUDIO="audio", DEPTH="depth",
                               EEG="eeg", TACTILE="tactile", PC="pc")
OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)
