"""PointBERT tokenizer of the 3D Lens on the HIP kernels (PointTokenizer.forward,
open_clip/modal_3d/models/pointbert/point_encoder.py:350-362): FPS -> kNN grouping -> mini-PointNet
(Encoder, dvae.py:179-212) -> reduce_dim, plus the centre MLP positional embedding; returns x + pos.

BatchNorm runs in EVAL mode (running statistics folded into the 1x1-conv weights at load time).  Train-mode
batch statistics (and SyncBN) for the PC recipe are not implemented yet and raise.
"""
import torch

from . import ops
from .engine import _dev

BF = torch.bfloat16


def _fold_bn(w, b, sd, p, eps=1e-5):
    s = sd[p + "weight"].float() / torch.sqrt(sd[p + "running_var"].float() + eps)
    return w * s[:, None], (b - sd[p + "running_mean"].float()) * s + sd[p + "bias"].float()


def _padk(w, K=64):
    out = torch.zeros(w.shape[0], K)
    out[:, :w.shape[1]] = w
    return out


class PointTokenizerEngine:
    def __init__(self, sd, a: str, lens, device, gemm_cfg=-1, bn_training=False):
        if bn_training:
            raise NotImplementedError("train-mode BatchNorm statistics for the point-cloud Lens (SURVEY §7 hard parts)")
        self.lens, self.device, self.cfg = lens, torch.device(device), gemm_cfg
        f = lambda k: sd[a + k].detach().float().cpu()
        w1, b1 = _fold_bn(f("encoder.first_conv.0.weight")[:, :, 0], f("encoder.first_conv.0.bias"), {k: v.cpu() for k, v in sd.items()}, a + "encoder.first_conv.1.")
        w3, b3 = _fold_bn(f("encoder.second_conv.0.weight")[:, :, 0], f("encoder.second_conv.0.bias"), {k: v.cpu() for k, v in sd.items()}, a + "encoder.second_conv.1.")
        half = w3.shape[1] // 2
        d = lambda t, dt=BF: t.to(device=device, dtype=dt).contiguous()
        self.w1, self.b1 = d(_padk(w1)), d(b1, torch.float32)
        self.w2, self.b2 = d(f("encoder.first_conv.3.weight")[:, :, 0]), d(f("encoder.first_conv.3.bias"), torch.float32)
        self.w3g, self.w3l, self.b3 = d(w3[:, :half]), d(w3[:, half:]), d(b3, torch.float32)
        self.w4, self.b4 = d(f("encoder.second_conv.3.weight")[:, :, 0]), d(f("encoder.second_conv.3.bias"), torch.float32)
        self.wr, self.br = d(f("reduce_dim.weight")), d(f("reduce_dim.bias"), torch.float32)
        self.wp0, self.bp0 = d(_padk(f("pos_embed.0.weight"))), d(f("pos_embed.0.bias"), torch.float32)
        self.wp2, self.bp2 = d(f("pos_embed.2.weight")), d(f("pos_embed.2.bias"), torch.float32)

    def group(self, pts: torch.Tensor, fps_start=None, want_idx=False):
        L = self.lens
        pts = pts.to(self.device).contiguous().float()
        if fps_start is None:   # misc.py:60 draws the first centre at random
            fps_start = torch.randint(0, pts.shape[1], (pts.shape[0],), device=self.device, dtype=torch.long)
        cidx, centers = ops.fps(pts, fps_start.to(self.device), L.pc_num_group)
        patches, nidx = ops.knn_group(pts, cidx, L.pc_group_size, Kp=64, want_idx=want_idx)
        return cidx, centers, patches, nidx

    def forward(self, pts: torch.Tensor, fps_start=None) -> torch.Tensor:
        """pts [B,N,3] -> tokens + pos, bf16 [B*G, trans_dim]."""
        M, c = self.lens.pc_group_size, self.cfg
        cidx, centers, patches, _ = self.group(pts, fps_start)
        h1 = ops.gemm(patches, self.w1, self.b1, act=ops.ACT_RELU, cfg=c)                    # conv 3->128 + BN + ReLU
        f = ops.gemm(h1, self.w2, self.b2, cfg=c)                                            # conv 128->256
        t = ops.gemm(ops.group_max(f, M), self.w3g, self.b3, cfg=c)                          # global half of conv 512->512
        h2 = torch.empty(f.shape[0], self.w3l.shape[0], device=self.device, dtype=BF)
        ops.gemm(f, self.w3l, None, out=h2, res=t, res_div=M, epi=ops.EPI_RES_BF16, act=ops.ACT_RELU, cfg=c)
        tok = ops.gemm(ops.group_max(ops.gemm(h2, self.w4, self.b4, cfg=c), M), self.wr, self.br, cfg=c)   # [B*G, trans]
        p1 = ops.gemm(ops.pad3(centers), self.wp0, self.bp0, act=ops.ACT_GELU, cfg=c)
        out = torch.empty_like(tok)
        ops.gemm(p1, self.wp2, self.bp2, out=out, res=tok, epi=ops.EPI_RES_BF16, cfg=c)      # pos + tokens
        return out
