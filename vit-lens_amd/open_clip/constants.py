"""Names the drop-in boundary shares with the reference (open_clip/constants.py:4-16)."""
from types import SimpleNamespace

ModalityType = SimpleNamespace(IMAGE="image", VIDEO="video", TEXT="text", AUDIO="audio", DEPTH="depth",
                               EEG="eeg", TACTILE="tactile", PC="pc")
OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)

# site paths of the reference (constants.py:18-47): placeholders there as well; the data / meta-data directories belong to
# the dataset code, which is out of scope - kept so that `from open_clip.constants import ...` lines keep working
PROJECT_DIR = "/PATH/TO/ViT-Lens"
CKPT_CACHE_DIR = "/PATH_TO/CACHE/DIR"
OBJAVERSE_DATA_DIR = "/PATH_TO/3d/ulip_batches"
PC_DATA_DIR = "/PATH_TO/3d"
PC_META_DATA_DIR = "/PATH_TO/vitlens/src/open_clip/modal_3d/data"
AUDIO_DATA_DIR = "/PATH_TO/audio_datasets"
AUDIO_META_DATA_DIR = "/PATH_TO/vitlens/src/open_clip/modal_audio/data"
DEPTH_DATA_DIR = "/PATH_TO/SUNRGBD"
DEPTH_META_DATA_DIR = "/PATH_TO/vitlens/src/open_clip/modal_depth/data"
TACTILE_DATA_DIR = "/PATH_TO/touch_and_go/dataset"
TACTILE_META_DATA_DIR = "/PATH_TO/vitlens/src/open_clip/modal_tactile/data"
EEG_DATA_DIR = "/PATH_TO/EEG"
EEG_META_DATA_DIR = "/PATH_TO/vitlens/src/open_clip/modal_eeg/data"
