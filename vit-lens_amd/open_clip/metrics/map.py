"""Mean average precision accumulator (protocol and result keys of the reference's open_clip/metrics/map.py:12-53, the
AudioSet-style multi-label evaluation): scores and multi-hot targets of every batch are collected, gathered across ranks,
and each class is scored on the host by scikit-learn's `average_precision_score` over sigmoid(score)."""
import numpy as np
import torch

from .base_metric import BaseMetric


class MAP(BaseMetric):
    def initialize(self):
        self._reset()

    def compute(self, ids, logits, targets):
        self._push(ids=ids, score=logits.float(), target=targets.float())

    def merge_results(self, output_predict=False):
        from sklearn.metrics import average_precision_score
        ids, score, target = self._collected("ids"), self._collected("score"), self._collected("target")
        prob = torch.sigmoid(score).cpu().numpy()
        while target.dim() > prob.ndim:            # a stray singleton axis from the collate function
            target = target.squeeze(0) if target.size(0) == 1 else target.squeeze(1)
        truth = target.cpu().numpy()
        per_class = average_precision_score(truth, prob, average=None)
        return {"map": np.mean(per_class), "map_cnt": len(truth),
                "predict_results": self._prediction_table(ids, prob.tolist(), output_predict)}
