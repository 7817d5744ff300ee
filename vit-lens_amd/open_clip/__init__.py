"""open_clip-compatible surface of the MI355X-native ViT-Lens hot path (drop-in for
`from open_clip import ModalityType, tokenize, tri_create_model, create_loss, ...`)."""
from .constants import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD, ModalityType
from .factory import (add_model_config, create_loss, get_model_config, get_tokenizer, list_models, load_checkpoint,
                      tri_create_model, tri_create_model_and_transforms, tri_create_model_from_pretrained)
from .loss import ClipLoss, ClipLossGeneral, TriClipLoss, gather_features
from .model import CLIPTextCfg, CLIPVisionCfg, TriCLIP, get_cast_dtype, get_input_dtype
from .tokenizer import SimpleTokenizer, decode, tokenize
from .transform import AugmentationCfg, image_transform
from .utils import all_gather, concat_all_gather, scaled_all_reduce
from .zero_shot_classifier import (acc, accuracy, build_zero_shot_classifier, build_zero_shot_classifier_legacy, cond_acc,
                                   zero_shot_logits)
