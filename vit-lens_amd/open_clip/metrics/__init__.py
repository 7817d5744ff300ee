"""Evaluation metric accumulators with the reference's names and protocol (open_clip/metrics/: initialize -> compute per
batch -> merge_results across ranks), used by the zero-shot / retrieval evaluation."""
from .accuracy import Accuracy
from .recall import Recall
from .map import MAP
