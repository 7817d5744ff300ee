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

    @staticmethod
    def comm_device() -> torch.device:
        """Where a collective's tensors must live: the current GPU under NCCL / RCCL, the host otherwise."""
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    _DTYPES = (torch.int64, torch.float32, torch.float64, torch.int32, torch.bool, torch.bfloat16, torch.float16)

    def _collected(self, name, like=None):
        """Everything pushed under `name`, concatenated along dim 0 and - in a multi-rank run - gathered rank-major.
        A rank that pushed nothing (fewer batches than ranks) does not know the dtype / trailing shape of `name`: the ranks
        first exchange a 10-word description of what they hold and the empty ones build their zero-row placeholder from a
        peer's, on the communicator's device - so the ragged gather sees the same dtype, rank and device everywhere."""
        parts = self.__dict__.get("_batches", {}).get(name, [])
        t = torch.cat(parts, dim=0) if parts else None
        if t is None and like is not None:
            t = like.new_zeros((0,) + tuple(like.shape[1:]))
        if not self.multi_rank():
            return t if t is not None else torch.zeros(0, dtype=torch.long)
        dev = self.comm_device()
        spec = torch.zeros(10, dtype=torch.int64)
        if t is not None:
            if t.dim() > 8 or t.dtype not in self._DTYPES:
                raise TypeError(f"metric tensor '{name}': unsupported dtype / rank for the cross-rank gather ({t.dtype}, {t.dim()}-d)")
            spec[0] = 1; spec[1] = self._DTYPES.index(t.dtype); spec[2] = t.dim()
            spec[3:3 + t.dim() - 1] = torch.tensor(t.shape[1:], dtype=torch.int64) if t.dim() > 1 else spec[3:3]
        world = dist.get_world_size()
        specs = torch.zeros(world * 10, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(specs, spec.to(dev))
        specs = specs.view(world, 10).cpu()
        have = [r for r in range(world) if int(specs[r, 0])]
        if not have:
            return torch.zeros(0, dtype=torch.long, device=dev)
        ref = specs[have[0]]
        dtype, nd = self._DTYPES[int(ref[1])], int(ref[2])
        trailing = tuple(int(v) for v in ref[3:3 + nd - 1])
        if t is None:
            t = torch.zeros((0,) + trailing, dtype=dtype, device=dev)
        elif t.dtype != dtype or tuple(t.shape[1:]) != trailing:
            raise ValueError(f"metric tensor '{name}': ranks disagree ({t.dtype}, {tuple(t.shape[1:])} here vs {dtype}, {trailing})")
        return all_gather(t.to(dev))

    def _reset(self):
        self._batches = {}

    def _global_sum(self, value: float, device) -> float:
        """Sum of a per-rank scalar over the ranks (one all-reduce), as a Python float."""
        if not self.multi_rank():
            return float(value)
        # float64 on every backend (RCCL reduces doubles): counts beyond 2^24 and soft-target sums stay exact, and the GPU
        # communicator reports the same accuracy as gloo
        t = torch.tensor([float(value)], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    @staticmethod
    def _prediction_table(ids, values, wanted: bool) -> dict:
        if not wanted:
            return {}
        return dict(zip(ids.cpu().tolist(), values.cpu().tolist() if torch.is_tensor(values) else list(values)))
