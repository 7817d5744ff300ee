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
        if targets.dim() == 2:
            # multi-hot (or soft) targets: the VALUE at the predicted label counts, as the reference's
            # `targets.gather(1, predict_labels.unsqueeze(1)).sum()` (metrics/accuracy.py:22-23)
            hit = targets.gather(1, pred.unsqueeze(1)).squeeze(1).to(torch.float64)
        else:
            hit = (pred == targets).to(torch.float64)
        self._push(ids=ids, pred=pred, hit=hit)

    def merge_results(self, output_predict=False):
        local_hits = self.__dict__.get("_batches", {}).get("hit", [])
        # the counters travel on the communicator's device (a rank without batches has no tensor to take one from)
        dev = torch.device(self.device) if self.device is not None else (self.comm_device() if self.multi_rank() else
                                                                          (local_hits[0].device if local_hits else torch.device("cpu")))
        n_hit = self._global_sum(sum(float(h.sum()) for h in local_hits), dev)
        n_all = self._global_sum(sum(h.numel() for h in local_hits), dev)
        table = self._prediction_table(self._collected("ids"), self._collected("pred"), output_predict)
        return {"accuracy": n_hit / n_all, "score_sum": n_hit, "score_cnt": int(n_all), "predict_results": table}
