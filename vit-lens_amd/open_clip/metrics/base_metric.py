"""Common ground of the evaluation accumulators (protocol of the reference's open_clip/metrics/base_metric.py:
`initialize` -> `compute` per batch -> `merge_results` once, on every rank).  Besides the three protocol methods this
base owns what all accumulators share: per-batch tensors are kept as a list and concatenated once, the cross-rank
exchange is the ragged `all_gather` / one all-reduce of open_clip.utils, and the optional id -> prediction dictionary of
`merge_results(output_predict=True)` is built in one place."""
import torch
import torch.distributed as dist

from ..utils import all_gather


class BaseMetric:
    def initialize(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}.initialize")

    def compute(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}.compute")

    def merge_results(self, output_predict=False):
        raise NotImplementedError(f"{type(self).__name__}.merge_results")

    # ------------------------------------------------------------------ shared machinery
    @staticmethod
    def multi_rank() -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _push(self, **batch):
        """Append this batch's tensors to the per-name lists (created on first use)."""
        store = self.__dict__.setdefault("_batches", {})
        for name, t in batch.items():
            store.setdefault(name, []).append(t)

    def _collected(self, name, like=None):
        """Everything pushed under `name`, concatenated along dim 0 and - in a multi-rank run - gathered rank-major."""
        parts = self.__dict__.get("_batches", {}).get(name, [])
        if parts:
            t = torch.cat(parts, dim=0)
        else:
            t = torch.zeros(0, dtype=torch.long) if like is None else like.new_zeros((0,) + tuple(like.shape[1:]))
        return all_gather(t) if self.multi_rank() else t

    def _reset(self):
        self._batches = {}

    def _global_sum(self, value: float, device) -> float:
        """Sum of a per-rank scalar over the ranks (one all-reduce), as a Python float."""
        if not self.multi_rank():
            return float(value)
        t = torch.tensor([float(value)], device=device, dtype=torch.float64 if device.type == "cpu" else torch.float32)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    @staticmethod
    def _prediction_table(ids, values, wanted: bool) -> dict:
        if not wanted:
            return {}
        return dict(zip(ids.cpu().tolist(), values.cpu().tolist() if torch.is_tensor(values) else list(values)))
