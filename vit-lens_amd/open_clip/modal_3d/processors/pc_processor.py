"""Point-cloud preprocessing of the 3D recipe ON THE GPU (reference: open_clip/modal_3d/processors/
pc_processor.py - numpy on the data-loader workers): `uniform` farthest-point sampling of a raw cloud down to `npoint`
points (:8-29), or a random subset (:41-45), then centring and scaling into the unit sphere (`pc_norm`, :32-38).

Same class name, constructor and call convention as the reference's `PCProcessorEval`; the result lives on the GPU
(a [npoint, C] f32 tensor), where the PointBERT tokenizer consumes it.  FPS runs in `vl_fps` (the kernel of the tokenizer,
bit-exact fp32 distances, lowest index on ties like np.argmax), the rest in `vl_pc_gather_normalize`.  The random choices
(FPS start, random subset) are drawn on the host with numpy exactly where the reference draws them, so a seeded
`np.random` reproduces the reference's sample."""
import numpy as np
import torch


class BaseProcessor:
    def __init__(self):
        self.transform = lambda x: x

    def __call__(self, item):
        return self.transform(item)

    @classmethod
    def from_config(cls, cfg=None):
        return cls()

    def build(self, **kwargs):
        return self.from_config(dict(kwargs))


def _as_device_f32(pc, device):
    t = torch.from_numpy(np.ascontiguousarray(pc)) if isinstance(pc, np.ndarray) else pc
    return t.to(device=device, dtype=torch.float32, non_blocking=True).contiguous()


class PCProcessorEval(BaseProcessor):
    def __init__(self, npoint, uniform, idendity=False, device="cuda"):
        self.npoint, self.uniform, self.idendity, self.device = npoint, uniform, idendity, torch.device(device)

    def process_batch(self, pcs, start=None, subset=None):
        """pcs [B, N, C] (numpy or tensor, C = 3..8, xyz first) -> [B, npoint, C] f32 on the GPU.
        start [B]: FPS start indices (default: np.random.randint per cloud, as farthest_point_sample draws it);
        subset [B, npoint]: indices of the random subset when not `uniform` (default: a numpy permutation per cloud)."""
        from vitlens_hip import ops
        x = _as_device_f32(pcs, self.device)
        B, N, C = x.shape
        if self.uniform and self.npoint < N:
            if start is None:
                start = np.array([np.random.randint(0, N) for _ in range(B)], dtype=np.int64)
            st = torch.as_tensor(np.asarray(start), dtype=torch.int64, device=self.device)
            idx, _ = ops.fps(x[:, :, :3].contiguous(), st, self.npoint, want_centers=False)
        else:
            if subset is None:
                subset = np.stack([np.random.permutation(N)[:self.npoint] for _ in range(B)])
            idx = torch.as_tensor(np.asarray(subset), dtype=torch.int64, device=self.device)
        return ops.pc_gather_normalize(x, idx)

    def __call__(self, pc):
        if self.idendity:
            return _as_device_f32(pc, self.device)
        return self.process_batch(pc[None] if isinstance(pc, np.ndarray) else pc.unsqueeze(0))[0]

    @classmethod
    def from_config(cls, cfg=None):
        cfg = cfg or {}
        return cls(npoint=cfg.get("npoint", 8192), uniform=cfg.get("uniform", True))
