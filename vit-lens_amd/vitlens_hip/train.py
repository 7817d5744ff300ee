"""Forward-with-saved-activations and backward of the trainable `visual.` tower on the HIP kernels.

Implements the autograd of VisionTransformer.forward (open_clip/transformer.py:723-792) for the lock
recipes of the reference (`VisionTransformer.lock`, transformer.py:553-627): dX flows through every
ResidualAttentionBlock, dW is produced only for the unlocked subset (depth recipe: visual_adapter +
the first n blocks; TRAIN_INFERENCE.md:229-242).  No torch autograd graph is built inside the tower:
`TowerTrainer.forward` stores exactly the tensors the backward kernels need and `backward` replays
the blocks in reverse with hand-written kernels (GEMM dX with fused dGELU, attention backward,
LayerNorm backward); weight gradients are NT GEMMs over transposed activation copies accumulated in
fp32 (`EPI_RES_F32` into the gradient buffer).
"""
from typing import Dict, Iterable, Optional

import torch

from . import ops
from .engine import VitEngine, _dev

BF = torch.bfloat16


class _Saved:
    """Per-(B, L) activation store: residual-stream snapshots and attention operands of every block."""

    def __init__(self, B, L, D, H, hidden, layers, device):
        dh = D // H
        T = B * L
        Lp = (L + 7) // 8 * 8
        f32 = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)
        bf = lambda *s: torch.empty(*s, device=device, dtype=BF)
        self.Lp = Lp
        self.X = [f32(T, D) for _ in range(2 * layers + 1)]           # X[2l]=block input, X[2l+1]=after attention
        self.stats = [[f32(T) for _ in range(4)] for _ in range(layers)]   # mean1, rstd1, mean2, rstd2
        self.q = [bf(B, H, L, dh) for _ in range(layers)]
        self.k = [bf(B, H, L, dh) for _ in range(layers)]
        self.v = [bf(B, H, L, dh) for _ in range(layers)]
        self.qt = [torch.zeros(B, H, dh, Lp, device=device, dtype=BF) for _ in range(layers)]
        self.kt = [torch.zeros(B, H, dh, Lp, device=device, dtype=BF) for _ in range(layers)]
        self.a = [bf(T, D) for _ in range(layers)]
        self.lse = [f32(B, H, L) for _ in range(layers)]
        self.u = [bf(T, hidden) for _ in range(layers)]
        # temporaries shared by all blocks
        self.h = bf(T, D); self.vt = torch.zeros(B, H, dh, Lp, device=device, dtype=BF); self.hid = bf(T, hidden)
        self.xpre = f32(T, D); self.pre_stats = [f32(T), f32(T)]
        self.post_stats = [f32(B), f32(B)]
        self.pooled = bf(B, D)
        # backward temporaries
        self.dx = f32(T, D); self.dxb = bf(T, D); self.du = bf(T, hidden); self.dh = bf(T, D)
        self.dO = bf(B, H, L, dh); self.dOt = torch.zeros(B, H, dh, Lp, device=device, dtype=BF)
        self.delta = f32(B, H, L); self.dqkv = bf(T, 3 * D)


class TowerTrainer:
    def __init__(self, eng: VitEngine, train_blocks: Iterable[int] = (), train_cls=False, train_pos=False,
                 param_prefix: str = "visual."):
        self.eng, self.prefix = eng, param_prefix
        self.train_blocks = sorted(set(train_blocks))
        self.train_cls, self.train_pos = train_cls, train_pos
        c = eng.cfg
        self.D, self.H, self.hidden, self.layers = c.width, c.heads, int(c.width * c.mlp_ratio), c.layers
        dev = eng.device
        # transposed bf16 weights for the dX GEMMs (C = dY . W  ==  NT GEMM against W^T)
        self.wT = [{k: w[k].t().contiguous() for k in ("in_w", "out_w", "fc_w", "proj_w")} for w in eng.blocks]
        self.proj = eng.projT.t().contiguous()            # [D, E]: the NT "W" operand of dpooled = dfeat . proj^T
        self._saved = {}
        self.grads: Dict[str, torch.Tensor] = {}
        self.ctx = None

    # ------------------------------------------------------------------------------------------ helpers
    def saved(self, B, L):
        key = (B, L)
        if key not in self._saved:
            self._saved[key] = _Saved(B, L, self.D, self.H, self.hidden, self.layers, self.eng.device)
        return self._saved[key]

    def grad_buffer(self, name, like):
        g = self.grads.get(name)
        if g is None:
            g = torch.zeros(like.shape, device=self.eng.device, dtype=torch.float32)
            self.grads[name] = g
        return g

    def zero_grads(self):
        for g in self.grads.values():
            g.zero_()

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, tokens: torch.Tensor, B: int, pos2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tokens [B*T, D] -> un-normalised features f32 [B, E]; keeps what backward() needs."""
        e, D, H = self.eng, self.D, self.H
        T = tokens.shape[0] // B
        L = T + 1
        dh = D // H
        S = self.saved(B, L)
        cfg = e.gemm_cfg
        ops.assemble_ln_pre(tokens, e.cls, e.pos, pos2, e.ln_pre[0], e.ln_pre[1], S.X[0], B, T, D,
                            xpre=S.xpre, mean=S.pre_stats[0], rstd=S.pre_stats[1])
        for l, w in enumerate(e.blocks):
            m1, r1, m2, r2 = S.stats[l]
            ops.layernorm(S.X[2 * l], w["ln1_w"], w["ln1_b"], S.h, B * L, D, mean=m1, rstd=r1)
            ops.gemm_qkv(S.h, w["in_w"], w["in_b"], S.q[l], S.k[l], S.vt, B, L, H, dh, cfg=cfg,
                         qt=S.qt[l], kt=S.kt[l], v=S.v[l])
            ops.attn_fwd(S.q[l], S.k[l], S.vt, S.a[l], lse=S.lse[l])
            ops.gemm(S.a[l], w["out_w"], w["out_b"], out=S.X[2 * l + 1], res=S.X[2 * l], epi=ops.EPI_RES_F32, cfg=cfg)
            ops.layernorm(S.X[2 * l + 1], w["ln2_w"], w["ln2_b"], S.h, B * L, D, mean=m2, rstd=r2)
            ops.gemm(S.h, w["fc_w"], w["fc_b"], out=S.hid, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg, out2=S.u[l])
            ops.gemm(S.hid, w["proj_w"], w["proj_b"], out=S.X[2 * l + 2], res=S.X[2 * l + 1], epi=ops.EPI_RES_F32, cfg=cfg)
        xl = S.X[2 * self.layers]
        ops.layernorm(xl, e.ln_post[0], e.ln_post[1], S.pooled, B, D, x_row_stride=L * D,
                      mean=S.post_stats[0], rstd=S.post_stats[1])
        feat = ops.gemm(S.pooled, e.projT, None, epi=ops.EPI_F32, cfg=cfg)
        self.ctx = (B, L, tokens, pos2 is not None)
        return feat

    # ------------------------------------------------------------------------------------------ backward
    def _dw(self, name, dy, x, rows):
        """grads[name] += dy^T x  (dy [rows, N], x [rows, K] -> [N, K]); operands transposed to bf16."""
        g = self.grad_buffer(name, torch.empty(dy.shape[1], x.shape[1]))
        rp = (rows + 63) // 64 * 64
        dyt = ops.transpose_to_bf16(dy, ldo=rp)
        xt = ops.transpose_to_bf16(x, ldo=rp)
        ops.gemm(dyt, xt, None, out=g, res=g, epi=ops.EPI_RES_F32, cfg=self.eng.gemm_cfg)

    def _db(self, name, dy):
        g = self.grad_buffer(name, torch.empty(dy.shape[1]))
        ops.colsum(dy, g)

    def backward(self, dfeat: torch.Tensor) -> torch.Tensor:
        """dfeat f32 [B, E] -> gradient w.r.t. the input tokens, f32 [B*T, D]; fills self.grads."""
        e, D, H = self.eng, self.D, self.H
        B, L, tokens, has_pos2 = self.ctx
        dh = D // H
        T = L - 1
        S = self.saved(B, L)
        rows = B * L
        cfg = e.gemm_cfg
        P = self.prefix
        # feat = pooled @ proj ; pooled = ln_post(x[:, 0])
        dpooled = ops.gemm(ops.cast_bf16(dfeat.contiguous()), self.proj, None, epi=ops.EPI_BF16, cfg=cfg)   # [B, D]
        S.dx.zero_()
        # only the cls rows (row b*L) of the final residual receive gradient: write them in place
        ops.layernorm_bwd(dpooled, S.X[2 * self.layers], S.post_stats[0], S.post_stats[1], e.ln_post[0], B, D,
                          dx=S.dx, x_row_stride=L * D, dx_row_stride=L * D)
        ops.cast_bf16(S.dx, out=S.dxb)
        for l in reversed(range(self.layers)):
            w, wT = e.blocks[l], self.wT[l]
            m1, r1, m2, r2 = S.stats[l]
            trainable = l in self.train_blocks
            bp = f"{P}transformer.resblocks.{l}."
            # ---- MLP branch: x2 = x1 + proj(gelu(fc(ln2(x1)))) ----
            ops.gemm(S.dxb, wT["proj_w"], None, out=S.du, res=S.u[l], epi=ops.EPI_DGELU, cfg=cfg)       # du = (dx W_proj) * gelu'(u)
            if trainable:
                ops.gelu_bf16(S.u[l], S.hid)
                self._dw(bp + "mlp.c_proj.weight", S.dx, S.hid, rows); self._db(bp + "mlp.c_proj.bias", S.dx)
                ops.layernorm(S.X[2 * l + 1], w["ln2_w"], w["ln2_b"], S.h, rows, D)
                self._dw(bp + "mlp.c_fc.weight", S.du, S.h, rows); self._db(bp + "mlp.c_fc.bias", S.du)
            ops.gemm(S.du, wT["fc_w"], None, out=S.dh, epi=ops.EPI_BF16, cfg=cfg)                           # dh2
            if trainable:
                ops.layernorm_bwd_params(S.dh, S.X[2 * l + 1], m2, r2, self.grad_buffer(bp + "ln_2.weight", w["ln2_w"]),
                                         self.grad_buffer(bp + "ln_2.bias", w["ln2_b"]), rows, D)
            ops.layernorm_bwd(S.dh, S.X[2 * l + 1], m2, r2, w["ln2_w"], rows, D, dres=S.dx, dx=S.dx, dx_bf16=S.dxb)
            # ---- attention branch: x1 = x0 + out(attn(qkv(ln1(x0)))) ----
            if trainable:
                self._dw(bp + "attn.out_proj.weight", S.dx, S.a[l], rows); self._db(bp + "attn.out_proj.bias", S.dx)
            ops.gemm_qkv(S.dxb, wT["out_w"], None, S.dO, None, None, B, L, H, dh, cfg=cfg, first=0, count=1,
                         qt=S.dOt, raw_scale=1.0)                                                           # dO (+ transposed)
            ops.attn_delta(S.dO, S.a[l], S.delta)
            ops.attn_bwd(S.q[l], S.k[l], S.v[l], S.qt[l], S.kt[l], S.dO, S.dOt, S.lse[l], S.delta,
                         S.dqkv, S.dqkv[:, D:], S.dqkv[:, 2 * D:], 3 * D, 3 * D)
            if trainable:
                ops.layernorm(S.X[2 * l], w["ln1_w"], w["ln1_b"], S.h, rows, D)
                self._dw(bp + "attn.in_proj_weight", S.dqkv, S.h, rows); self._db(bp + "attn.in_proj_bias", S.dqkv)
            ops.gemm(S.dqkv, wT["in_w"], None, out=S.dh, epi=ops.EPI_BF16, cfg=cfg)                         # dh1
            if trainable:
                ops.layernorm_bwd_params(S.dh, S.X[2 * l], m1, r1, self.grad_buffer(bp + "ln_1.weight", w["ln1_w"]),
                                         self.grad_buffer(bp + "ln_1.bias", w["ln1_b"]), rows, D)
            ops.layernorm_bwd(S.dh, S.X[2 * l], m1, r1, w["ln1_w"], rows, D, dres=S.dx, dx=S.dx, dx_bf16=S.dxb)
        # ---- ln_pre and the [cls; tokens] + pos assembly ----
        dxpre = torch.empty_like(S.dx)
        ops.layernorm_bwd(S.dx, S.xpre, S.pre_stats[0], S.pre_stats[1], e.ln_pre[0], rows, D, dx=dxpre)
        if self.train_cls:
            ops.batch_rowsum(dxpre, self.grad_buffer(P + "class_embedding", e.cls).view(1, D), B, 1, D, L, 0)
        if self.train_pos:
            ops.batch_rowsum(dxpre, self.grad_buffer(P + "positional_embedding", e.pos), B, L, D, L, 0)
        self.dxpre = dxpre
        return dxpre.view(B, L, D)[:, 1:, :].reshape(B * T, D)


class DepthLensTrainer:
    """`visual.` tower of the depth recipe: DepthTokenizer conv1 + pos_emb -> ViT trunk (Perceiver = Identity)."""

    def __init__(self, lens_engine, unlock_first_n: int = 4):
        self.le = lens_engine
        self.tower = TowerTrainer(lens_engine.vit, train_blocks=range(unlock_first_n), param_prefix="visual.")
        self.ctx = None

    @property
    def grads(self):
        return self.tower.grads

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        le = self.le
        p = le.tower.patch
        cols, gh, gw = ops.im2col(x.contiguous().float(), p, p, p, p, le.conv_w.shape[1])
        tok = ops.gemm(cols, le.conv_w, None, epi=ops.EPI_BF16, cfg=le.gemm_cfg)
        self.ctx = (cols, x.shape[0])
        return self.tower.forward(tok, x.shape[0], pos2=le.adapter_pos)

    def backward(self, dfeat: torch.Tensor):
        cols, B = self.ctx
        dtok = self.tower.backward(dfeat)                       # f32 [B*T, D]
        T, D = dtok.shape[0] // B, dtok.shape[1]
        t = self.tower
        g = t.grad_buffer("visual.visual_adapter.conv1.weight_gemm", torch.empty(D, cols.shape[1]))
        rp = (dtok.shape[0] + 63) // 64 * 64
        ops.gemm(ops.transpose_to_bf16(dtok, ldo=rp), ops.transpose_to_bf16(cols, ldo=rp), None, out=g, res=g,
                 epi=ops.EPI_RES_F32, cfg=self.le.gemm_cfg)
        ops.batch_rowsum(t.dxpre, t.grad_buffer("visual.visual_adapter.pos_emb", self.le.adapter_pos), B, T, D, T + 1, 1)


class AdamW:
    """torch.optim.AdamW semantics on f32 master tensors, one fused kernel launch per tensor
    (reference: depth_tri_main.py:394-419 -- two groups: no weight decay for ndim<2 / bn / ln / bias / logit_scale)."""

    def __init__(self, params: Dict[str, torch.Tensor], lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2):
        self.params, self.lr, self.betas, self.eps, self.wd = params, lr, betas, eps, weight_decay
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    @staticmethod
    def decays(name: str, p: torch.Tensor) -> bool:
        return not (p.ndim < 2 or "bn" in name or "ln" in name or "bias" in name or "logit_scale" in name)

    def step(self, grads: Dict[str, torch.Tensor], grad_scale: float = 1.0):
        self.t += 1
        for k, p in self.params.items():
            g = grads.get(k)
            if g is None:
                continue
            wd = self.wd if self.decays(k, p) else 0.0
            ops.adamw_step(p.view(-1), g.view(-1), self.m[k].view(-1), self.v[k].view(-1), self.lr, self.betas[0],
                           self.betas[1], self.eps, wd, self.t, grad_scale)
