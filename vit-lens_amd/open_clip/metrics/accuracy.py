"""Top-1 accuracy accumulator (protocol and result keys of the reference's open_clip/metrics/accuracy.py:8-52).
Per batch only the arg-max labels and the per-sample hit flags are kept; the totals are formed in `merge_results`, where a
multi-rank run adds the two counters with one all-reduce each and gathers the (ragged) id / prediction lists."""
import torch

from .base_metric import BaseMetric


class Accuracy(BaseMetric):
    def initialize(self, device=None):
        self.device = device
        self._reset()

    def compute(self, ids, logits, targets):
        pred = logits.argmax(dim=1)
        if targets.dim() == 2:                     # multi-hot targets: right if the predicted label is one of the set ones
            hit = targets[torch.arange(pred.numel(), device=pred.device), pred] != 0
        else:
            hit = pred == targets
        self._push(ids=ids, pred=pred, hit=hit.to(torch.int64))

    def merge_results(self, output_predict=False):
        local_hits = self.__dict__.get("_batches", {}).get("hit", [])
        dev = self.device or (local_hits[0].device if local_hits else torch.device("cpu"))
        dev = torch.device(dev)
        n_hit = self._global_sum(sum(int(h.sum()) for h in local_hits), dev)
        n_all = self._global_sum(sum(h.numel() for h in local_hits), dev)
        table = self._prediction_table(self._collected("ids"), self._collected("pred"), output_predict)
        return {"accuracy": n_hit / n_all, "score_sum": n_hit, "score_cnt": int(n_all), "predict_results": table}
