"""Mean average precision accumulator (reference: open_clip/metrics/map.py:12-53): logits and multi-hot targets are
collected per batch, gathered across ranks, squashed with a sigmoid and scored per class with scikit-learn's
`average_precision_score` on the host, exactly as the reference does (AudioSet-style evaluation)."""
import numpy as np
import torch
import torch.distributed as dist

from .base_metric import BaseMetric
from ..utils import all_gather


class MAP(BaseMetric):
    def __init__(self):
        super().__init__()

    def initialize(self):
        self.logits, self.targets, self.ids = [], [], []

    def compute(self, ids, logits, targets):
        self.ids.append(ids); self.logits.append(logits.float()); self.targets.append(targets.float())

    def merge_results(self, output_predict=False):
        from sklearn.metrics import average_precision_score
        ids, preds, targets = torch.cat(self.ids, 0), torch.cat(self.logits, 0), torch.cat(self.targets, 0)
        if dist.is_available() and dist.is_initialized():
            ids, preds, targets = all_gather(ids), all_gather(preds), all_gather(targets)
        preds = torch.sigmoid(preds).cpu().numpy()
        if targets.ndim != preds.ndim:
            if targets.size(0) == 1:
                targets = targets.squeeze(0)
            if targets.size(1) == 1:
                targets = targets.squeeze(1)
        targets = targets.cpu().numpy()
        predict_results = {}
        if output_predict:
            for idx, pred in zip(ids.cpu().tolist(), preds.tolist()):
                predict_results[idx] = pred
        return {"map": np.mean(average_precision_score(targets, preds, average=None)), "map_cnt": len(targets),
                "predict_results": predict_results}
