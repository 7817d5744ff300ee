"""Top-1 accuracy accumulator (reference: open_clip/metrics/accuracy.py:8-52).  The running sums live on the device of
the first batch (the reference hard-codes `.cuda()`); across ranks they are summed with one all-reduce each and the
per-sample ids / predictions gathered with the ragged `all_gather` of open_clip.utils."""
import torch
import torch.distributed as dist

from .base_metric import BaseMetric
from ..utils import all_gather


class Accuracy(BaseMetric):
    def __init__(self):
        super().__init__()

    def initialize(self, device=None):
        self.device = device
        self.score_sum = self.score_cnt = self.ids = self.hyps = None

    def _lazy(self, device):
        if self.score_sum is None:
            device = self.device or device
            self.score_sum = torch.zeros(1, device=device, dtype=torch.float32)
            self.score_cnt = torch.zeros(1, device=device, dtype=torch.int32)
            self.ids = torch.zeros(0, device=device, dtype=torch.long)
            self.hyps = torch.zeros(0, device=device, dtype=torch.long)

    def compute(self, ids, logits, targets):
        self._lazy(logits.device)
        predict_labels = logits.argmax(1)
        if targets.dim() == 2:                       # multi-hot targets: a prediction counts if its label is set
            n_correct = targets.gather(1, predict_labels.unsqueeze(1)).sum()
        else:
            n_correct = predict_labels.eq(targets).sum()
        self.score_sum += n_correct
        self.score_cnt += logits.size(0)
        self.ids = torch.cat([self.ids, ids], dim=0)
        self.hyps = torch.cat([self.hyps, predict_labels], dim=0)

    def merge_results(self, output_predict=False):
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(self.score_sum, op=dist.ReduceOp.SUM)
            dist.all_reduce(self.score_cnt, op=dist.ReduceOp.SUM)
            ids, hyps = all_gather(self.ids), all_gather(self.hyps)
        else:
            ids, hyps = self.ids, self.hyps
        predict_results = {}
        if output_predict:
            for i, h in zip(ids.cpu().tolist(), hyps.cpu().tolist()):
                predict_results[i] = h
        score_sum, score_cnt = self.score_sum.item(), self.score_cnt.item()
        return {"accuracy": score_sum / score_cnt, "score_sum": score_sum, "score_cnt": score_cnt, "predict_results": predict_results}
