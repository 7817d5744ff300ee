"""The OpenShape flavour of ViT-Lens (reference tree VitLens-OpenShape/src): a point-cloud Lens + ViT tower
trained against PRECOMPUTED unit-norm image / text features.  `CLIPBindWrap` (models/clip_bind.py), the tri-modal loss with
its retrieval accuracies (loss.py:80-185), `LogitScaleNetwork` (models/LogitScaleNetwork.py) and the step body of
`Trainer.train_one_epoch_openclip` (train.py:784-843) on the drop-in modules; every FLOP runs in the HIP kernels."""
from .clip_bind import CLIPBindWrap, LogitScaleNetwork
from .loss import TriClipLoss
from .train import openclip_step

__all__ = ["CLIPBindWrap", "LogitScaleNetwork", "TriClipLoss", "openclip_step"]
