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
from . import engine as _engine
from .engine import VitEngine

BF = torch.bfloat16


class _Saved:
    """Per-(B, L) activation store: residual-stream snapshots and attention operands of every block."""

    def __init__(self, B, L, D, H, hidden, layers, device, res_dtype=torch.float32, keep_blocks=(), checkpoint=False):
        """checkpoint=True (set_grad_checkpointing, transformer.py:366-368): only the block INPUTS are kept per layer; every
        other per-layer buffer is ONE tensor shared by all layers (the lists below repeat the same object), refilled by
        re-running a block's forward right before its backward.  ViT-L at 256 samples: 3.4 GB instead of 36 GB."""
        dh = D // H
        T = B * L
        Lp = (L + 7) // 8 * 8
        f32 = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)
        bf = lambda *s: torch.empty(*s, device=device, dtype=BF)
        self.Lp = Lp
        xres = f32 if res_dtype == torch.float32 else bf

        def per_layer(make):
            if not checkpoint:
                return [make() for _ in range(layers)]
            one = make()
            return [one] * layers
        mids = per_layer(lambda: xres(T, D))
        self.X = [None] * (2 * layers + 1)                            # X[2l]=block input, X[2l+1]=after attention
        for l in range(layers):
            self.X[2 * l] = xres(T, D); self.X[2 * l + 1] = mids[l]
        self.X[2 * layers] = xres(T, D)
        self.stats = per_layer(lambda: [f32(T) for _ in range(4)])       # mean1, rstd1, mean2, rstd2
        # the packed in-projection output of every block; the attention kernels read q / k / v out of it in place
        self.qkv = per_layer(lambda: bf(T, 3 * D))
        hv = lambda m, i=0: ops.heads_view(m, B, L, H, dh, i * D)
        self.q = [hv(m, 0) for m in self.qkv]; self.k = [hv(m, 1) for m in self.qkv]; self.v = [hv(m, 2) for m in self.qkv]
        self.a = per_layer(lambda: bf(T, D))
        self.av = [hv(m) for m in self.a]
        self.lse = per_layer(lambda: f32(B, H, L))
        self.u = per_layer(lambda: bf(T, hidden))
        # temporaries shared by all blocks
        self.h = bf(T, D); self.hid = bf(T, hidden)
        # trainable blocks keep their GEMM inputs (LN outputs, GELU output) for the weight gradients: 1.3 GB per block and
        # micro-batch at b = 256 instead of two LayerNorm passes and one GELU pass in the backward (HBM is 288 GB)
        if checkpoint and keep_blocks:
            k1, k2, k3 = bf(T, D), bf(T, D), bf(T, hidden)
            self.h1 = {l: k1 for l in keep_blocks}; self.h2 = {l: k2 for l in keep_blocks}; self.hidk = {l: k3 for l in keep_blocks}
        else:
            self.h1 = {l: bf(T, D) for l in keep_blocks}; self.h2 = {l: bf(T, D) for l in keep_blocks}
            self.hidk = {l: bf(T, hidden) for l in keep_blocks}
        # LayerNorm folding (frozen blocks, bf16 stream): partial row sums left by the GEMM that wrote `part_of`
        self.part = f32(T * (D // 64) * 2) if (D % 64 == 0 and res_dtype == BF) else None
        self.part_of, self.part_rows = None, 0
        self.xpre = f32(T, D); self.pre_stats = [f32(T), f32(T)]
        self.post_stats = [f32(B), f32(B)]
        self.pooled = bf(B, D)
        # backward temporaries
        # residual-gradient stream in the residual dtype (the reference's autocast: bf16 activations -> bf16 gradients);
        # with an f32 stream the GEMM operand is a separate bf16 copy
        self.dx = xres(T, D); self.dxb = bf(T, D) if res_dtype == torch.float32 else self.dx
        self.du = bf(T, hidden); self.dh = bf(T, D)
        self.dOm = bf(T, D); self.dO = hv(self.dOm)          # out-projection input gradient, read by heads in place
        self.delta = f32(B, H, L); self.dqkv = bf(T, 3 * D)


class TowerTrainer:
    def __init__(self, eng: VitEngine, train_blocks: Iterable[int] = (), train_cls=False, train_pos=False,
                 param_prefix: str = "visual.", train_ln_pre=False, train_ln_post=False, train_proj=False, checkpoint=False):
        self.eng, self.prefix = eng, param_prefix
        self.causal = False                         # (TextTowerTrainer: the text transformer's additive causal mask, transformer.py:870-876)
        self.checkpoint = bool(checkpoint)          # activation recompute per block (Transformer.forward, transformer.py:366-368)
        self.train_blocks = sorted(set(train_blocks))
        self.train_cls, self.train_pos = train_cls, train_pos
        # the stem / head pieces of the grouped (LiT) unlock, VisionTransformer.lock open_clip/transformer.py:564-597
        self.train_ln_pre, self.train_ln_post, self.train_proj = train_ln_pre, train_ln_post, train_proj
        c = eng.cfg
        self.D, self.H, self.hidden, self.layers = c.width, c.heads, int(c.width * c.mlp_ratio), c.layers
        dev = eng.device
        # transposed bf16 weights for the dX GEMMs (C = dY . W  ==  NT GEMM against W^T)
        self.wT = [{k: w[k].t().contiguous() for k in ("in_w", "out_w", "fc_w", "proj_w")} for w in eng.blocks]
        # [D, E]: the NT "W" operand of dpooled = dfeat . proj^T (None: a tower without output projection)
        self.proj = eng.projT.t().contiguous() if eng.projT is not None else None
        self._saved = {}
        self.grads: Dict[str, torch.Tensor] = {}
        self.ctx = None

    def refresh_derived(self, blocks=(), proj=False):
        """The engine's weights of `blocks` (and proj) were updated in place: redo their transposes."""
        for l in blocks:
            for k in ("in_w", "out_w", "fc_w", "proj_w"):
                self.wT[l][k].copy_(self.eng.blocks[l][k].t())
        if proj and self.proj is not None:
            self.proj.copy_(self.eng.projT.t())

    # ------------------------------------------------------------------------------------------ helpers
    def saved(self, B, L):
        key = (B, L)
        if key not in self._saved:
            self._saved[key] = _Saved(B, L, self.D, self.H, self.hidden, self.layers, self.eng.device, self.eng.res_dtype,
                                      keep_blocks=tuple(self.train_blocks), checkpoint=self.checkpoint)
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
        res_epi = ops.EPI_RES_F32 if e.res_dtype == torch.float32 else ops.EPI_RES_BF16
        ops.assemble_ln_pre(tokens, e.cls, e.pos, pos2, e.ln_pre[0], e.ln_pre[1], S.X[0], B, T, D,
                            xpre=S.xpre, mean=S.pre_stats[0], rstd=S.pre_stats[1])
        for l in range(self.layers):
            self._block_forward(S, l, B, L)
        xl = S.X[2 * self.layers]
        ops.layernorm(xl, e.ln_post[0], e.ln_post[1], S.pooled, B, D, x_row_stride=L * D,
                      mean=S.post_stats[0], rstd=S.post_stats[1])
        if e.projT is None:          # features = ln_post(cls) itself (bf16 activation, returned as f32)
            feat = S.pooled.float()
        else:
            feat = ops.gemm(S.pooled, e.projT, None, epi=ops.EPI_F32, cfg=cfg)
        self.ctx = (B, L, tokens, pos2 is not None)
        return feat

    def _block_forward(self, S, l, B, L, write_out=True):
        """One ResidualAttentionBlock forward (transformer.py:254-272) into the saved-activation slots of layer l.  With
        write_out=False (the recompute in front of a block's backward) the block output X[2l+2] - already there from the
        forward pass and not needed by the backward of block l - is not recomputed: the last GEMM is skipped."""
        e, D, H = self.eng, self.D, self.H
        dh = D // H
        cfg = e.gemm_cfg
        res_epi = ops.EPI_RES_F32 if e.res_dtype == torch.float32 else ops.EPI_RES_BF16
        w = e.blocks[l]
        m1, r1, m2, r2 = S.stats[l]
        h1, h2, hid = S.h1.get(l, S.h), S.h2.get(l, S.h), S.hidk.get(l, S.hid)
        # LayerNorm folding (engine.run_blocks; round 4): a FROZEN block never materialises ln_1 / ln_2 - its backward needs
        # only (mean, rstd), which ln_row_stats leaves in S.stats; a trainable block keeps its LayerNorm outputs (the operands
        # of its weight gradients).  Either kind leaves the partial row sums of its output when the next block is folded.
        fold_ok = _engine.LN_FOLD and S.part is not None
        folded = fold_ok and "in_f" in w and l not in self.train_blocks
        next_folded = (fold_ok and l + 1 < self.layers and "in_f" in e.blocks[l + 1] and (l + 1) not in self.train_blocks)
        if folded:
            x0, x1 = S.X[2 * l], S.X[2 * l + 1]
            mm0 = S.part_rows if S.part_of is x0 else 0      # (a recompute in front of the backward finds none: from the rows)
            S.part_of = None
            # (the row-statistics launch also leaves the LayerNorm output of the consuming GEMM's leftover rows in S.h)
            r_in, r_fc = ops.fold_rows(x0, S.qkv[l], 3 * D), ops.fold_rows(x1, hid, hid.shape[1])
            k_in = dict(ln_w=w["ln1_w"], ln_b=w["ln1_b"], h_left=S.h, h_row0=r_in) if mm0 <= r_in else {}
            ops.ln_row_stats(S.part, x0, mm0, m1, r1, **k_in)
            ops.gemm_lnfold(x0, w["in_f"], m1, r1, S.qkv[l], w["in_w"], w["in_b"], w["ln1_w"], w["ln1_b"], S.h, cfg=cfg,
                            h_ready=bool(k_in))
            ops.attn_fwd(S.q[l], S.k[l], S.v[l], S.a[l], lse=S.lse[l], causal=self.causal, qscale=dh ** -0.5 * ops.LOG2E)
            mm = ops.gemm_res_rowstats(S.a[l], w["out_w"], w["out_b"], x1, x0, S.part, cfg=cfg)
            k_fc = dict(ln_w=w["ln2_w"], ln_b=w["ln2_b"], h_left=S.h, h_row0=r_fc) if mm <= r_fc else {}
            ops.ln_row_stats(S.part, x1, mm, m2, r2, **k_fc)
            ops.gemm_lnfold(x1, w["fc_f"], m2, r2, hid, w["fc_w"], w["fc_b"], w["ln2_w"], w["ln2_b"], S.h,
                            act=ops.ACT_GELU_DSAVE, out2=S.u[l], cfg=cfg, h_ready=bool(k_fc))
        else:
            ops.layernorm(S.X[2 * l], w["ln1_w"], w["ln1_b"], h1, B * L, D, mean=m1, rstd=r1)
            ops.gemm(h1, w["in_w"], w["in_b"], out=S.qkv[l], epi=ops.EPI_BF16, cfg=cfg)
            ops.attn_fwd(S.q[l], S.k[l], S.v[l], S.a[l], lse=S.lse[l], causal=self.causal, qscale=dh ** -0.5 * ops.LOG2E)
            ops.gemm(S.a[l], w["out_w"], w["out_b"], out=S.X[2 * l + 1], res=S.X[2 * l], epi=res_epi, cfg=cfg)
            ops.layernorm(S.X[2 * l + 1], w["ln2_w"], w["ln2_b"], h2, B * L, D, mean=m2, rstd=r2)
            # S.u[l] = gelu'(fc output): all the backward needs of the pre-activation, evaluated next to gelu() from the same
            # exp / rational pieces (+3 VALU per element here) - the dX GEMM's epilogue is then one multiplication
            ops.gemm(h2, w["fc_w"], w["fc_b"], out=hid, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=cfg, out2=S.u[l])
        if write_out:
            S.part_of, S.part_rows = None, 0
            if next_folded:
                S.part_rows = ops.gemm_res_rowstats(hid, w["proj_w"], w["proj_b"], S.X[2 * l + 2], S.X[2 * l + 1], S.part, cfg=cfg)
                S.part_of = S.X[2 * l + 2]
            else:
                ops.gemm(hid, w["proj_w"], w["proj_b"], out=S.X[2 * l + 2], res=S.X[2 * l + 1], epi=res_epi, cfg=cfg)

    # ------------------------------------------------------------------------------------------ backward
    def _dw(self, name, dy, x, rows, bias_name=None):
        """grads[name] += dy^T x  (dy [rows, N], x [rows, K] -> [N, K]); operands transposed to bf16.  The bias gradient
        (column sums of dy) is produced by the transpose of dy, which reads all of dy anyway."""
        g = self.grad_buffer(name, torch.empty(dy.shape[1], x.shape[1]))
        rp = (rows + 63) // 64 * 64
        gb = self.grad_buffer(bias_name, torch.empty(dy.shape[1])) if bias_name else None
        if rows == dy.shape[0] == x.shape[0] and ops.gemm_dw_tn(dy, x, g):      # token-major operands: no transposed copies
            if gb is not None:
                ops.colsum(dy, gb)
            return
        dyt = ops.transpose_colsum(dy, rp, colsum_out=gb)
        xt = ops.transpose_colsum(x, rp)
        ops.gemm_dw(dyt, xt, g, cfg=self.eng.gemm_cfg)

    def backward(self, dfeat: torch.Tensor, on_block_done=None) -> torch.Tensor:
        """dfeat f32 [B, E] -> gradient w.r.t. the input tokens, f32 [B*T, D]; fills self.grads.
        on_block_done(l) is called as soon as every gradient of trainable block l has been enqueued (the multi-GPU step
        starts that block's gradient all-reduce there, under the backward of the blocks below it)."""
        e, D, H = self.eng, self.D, self.H
        B, L, tokens, has_pos2 = self.ctx
        dh = D // H
        T = L - 1
        S = self.saved(B, L)
        rows = B * L
        cfg = e.gemm_cfg
        P = self.prefix
        # feat = pooled @ proj ; pooled = ln_post(x[:, 0])
        dfb = ops.cast_bf16(dfeat.contiguous())
        dpooled = dfb if self.proj is None else ops.gemm(dfb, self.proj, None, epi=ops.EPI_BF16, cfg=cfg)   # [B, D]
        if self.train_proj and self.proj is not None:          # feat = pooled @ proj  ->  dproj[D, E] += pooled^T dfeat
            bp = (B + 63) // 64 * 64
            ops.gemm_dw(ops.transpose_to_bf16(S.pooled, ldo=bp), ops.transpose_to_bf16(dfb, ldo=bp),
                        self.grad_buffer(P + "proj", torch.empty(D, dfeat.shape[1])), cfg=cfg)
        if self.train_ln_post:
            ops.layernorm_bwd_params(dpooled, S.X[2 * self.layers], S.post_stats[0], S.post_stats[1],
                                     self.grad_buffer(P + "ln_post.weight", e.ln_post[0]),
                                     self.grad_buffer(P + "ln_post.bias", e.ln_post[1]), B, D, x_row_stride=L * D)
        S.dx.zero_()
        # only the cls rows (row b*L) of the final residual receive gradient: write them in place
        ops.layernorm_bwd(dpooled, S.X[2 * self.layers], S.post_stats[0], S.post_stats[1], e.ln_post[0], B, D,
                          dx=S.dx, x_row_stride=L * D, dx_row_stride=L * D)
        self._blocks_backward(S, B, L, on_block_done)
        # ---- ln_pre and the [cls; tokens] + pos assembly ----
        dxpre = torch.empty(S.dx.shape, device=S.dx.device, dtype=torch.float32)
        if self.train_ln_pre:
            ops.layernorm_bwd_params(S.dx, S.xpre, S.pre_stats[0], S.pre_stats[1], self.grad_buffer(P + "ln_pre.weight", e.ln_pre[0]),
                                     self.grad_buffer(P + "ln_pre.bias", e.ln_pre[1]), rows, D)
        ops.layernorm_bwd(S.dx, S.xpre, S.pre_stats[0], S.pre_stats[1], e.ln_pre[0], rows, D, dx=dxpre)
        if self.train_cls:
            ops.batch_rowsum(dxpre, self.grad_buffer(P + "class_embedding", e.cls).view(1, D), B, 1, D, L, 0)
        if self.train_pos:
            ops.batch_rowsum(dxpre, self.grad_buffer(P + "positional_embedding", e.pos), B, L, D, L, 0)
        self.dxpre = dxpre
        return dxpre.view(B, L, D)[:, 1:, :].reshape(B * T, D)

    def _blocks_backward(self, S, B, L, on_block_done=None):
        """S.dx (the gradient of the last block's output, residual-stream dtype) -> S.dx = the gradient of the first block's
        input, through every ResidualAttentionBlock in reverse; parameter gradients of the trainable blocks into self.grads."""
        e, D, H = self.eng, self.D, self.H
        rows = B * L
        cfg = e.gemm_cfg
        P = self.prefix
        f32_stream = S.dx.dtype == torch.float32
        dxb_out = S.dxb if f32_stream else None
        if f32_stream:
            ops.cast_bf16(S.dx, out=S.dxb)
        for l in reversed(range(self.layers)):
            if self.checkpoint:          # refill the shared per-layer buffers with block l's activations
                self._block_forward(S, l, B, L, write_out=False)
            w, wT = e.blocks[l], self.wT[l]
            m1, r1, m2, r2 = S.stats[l]
            trainable = l in self.train_blocks
            bp = f"{P}transformer.resblocks.{l}."
            # ---- MLP branch: x2 = x1 + proj(gelu(fc(ln2(x1)))) ----
            ops.gemm(S.dxb, wT["proj_w"], None, out=S.du, res=S.u[l], epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=cfg)   # du = (dx W_proj) * gelu'(u)
            if trainable:
                self._dw(bp + "mlp.c_proj.weight", S.dx, S.hidk[l], rows, bp + "mlp.c_proj.bias")
                self._dw(bp + "mlp.c_fc.weight", S.du, S.h2[l], rows, bp + "mlp.c_fc.bias")
            ops.gemm(S.du, wT["fc_w"], None, out=S.dh, epi=ops.EPI_BF16, cfg=cfg)                           # dh2
            if trainable:
                ops.layernorm_bwd_params(S.dh, S.X[2 * l + 1], m2, r2, self.grad_buffer(bp + "ln_2.weight", w["ln2_w"]),
                                         self.grad_buffer(bp + "ln_2.bias", w["ln2_b"]), rows, D)
            ops.layernorm_bwd(S.dh, S.X[2 * l + 1], m2, r2, w["ln2_w"], rows, D, dres=S.dx, dx=S.dx, dx_bf16=dxb_out)
            # ---- attention branch: x1 = x0 + out(attn(qkv(ln1(x0)))) ----
            if trainable:
                self._dw(bp + "attn.out_proj.weight", S.dx, S.a[l], rows, bp + "attn.out_proj.bias")
            ops.gemm(S.dxb, wT["out_w"], None, out=S.dOm, epi=ops.EPI_BF16, cfg=cfg)                        # dO
            ops.attn_bwd(S.q[l], S.k[l], S.v[l], S.dO, S.av[l], S.lse[l], S.delta,
                         S.dqkv, S.dqkv[:, D:], S.dqkv[:, 2 * D:], 3 * D, 3 * D, causal=self.causal)
            if trainable:
                self._dw(bp + "attn.in_proj_weight", S.dqkv, S.h1[l], rows, bp + "attn.in_proj_bias")
            ops.gemm(S.dqkv, wT["in_w"], None, out=S.dh, epi=ops.EPI_BF16, cfg=cfg)                         # dh1
            if trainable:
                ops.layernorm_bwd_params(S.dh, S.X[2 * l], m1, r1, self.grad_buffer(bp + "ln_1.weight", w["ln1_w"]),
                                         self.grad_buffer(bp + "ln_1.bias", w["ln1_b"]), rows, D)
            ops.layernorm_bwd(S.dh, S.X[2 * l], m1, r1, w["ln1_w"], rows, D, dres=S.dx, dx=S.dx, dx_bf16=dxb_out)
            if trainable and on_block_done is not None:
                on_block_done(l)


def conv_weight_grad(dtok: torch.Tensor, cols: torch.Tensor, g: torch.Tensor, gemm_cfg: int = -1):
    """g[D, Kp] += dtok^T cols: the weight gradient of a stride-patch convolution run as im2col + GEMM (dtok f32 [R, D] = the
    gradient of its output tokens, cols bf16 [R, Kp] = the im2col rows the forward kept).  Token-major operands for the dW
    kernel (the fp32 gradient cast to bf16 - the values a transposed copy would hold too); the transposing NT path where the
    shape does not fit."""
    if ops.gemm_dw_tn_any(ops.cast_bf16(dtok.contiguous()), cols, g):
        return
    rp = (dtok.shape[0] + 63) // 64 * 64
    ops.gemm_dw(ops.transpose_to_bf16(dtok, ldo=rp), ops.transpose_to_bf16(cols, ldo=rp), g, cfg=gemm_cfg)


class DepthLensTrainer:
    """`visual.` tower of the depth recipe: DepthTokenizer conv1 + pos_emb -> ViT trunk (Perceiver = Identity)."""

    def __init__(self, lens_engine, unlock_first_n: int = 4, tower_kw=None):
        self.le = lens_engine
        kw = dict(train_blocks=range(unlock_first_n)) if tower_kw is None else dict(tower_kw)
        self.tower = TowerTrainer(lens_engine.vit, param_prefix="visual.", **kw)
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

    def backward(self, dfeat: torch.Tensor, on_block_done=None):
        cols, B = self.ctx
        dtok = self.tower.backward(dfeat, on_block_done)         # f32 [B*T, D]
        T, D = dtok.shape[0] // B, dtok.shape[1]
        t = self.tower
        g = t.grad_buffer("visual.visual_adapter.conv1.weight_gemm", torch.empty(D, cols.shape[1]))
        conv_weight_grad(dtok, cols, g, self.le.gemm_cfg)
        ops.batch_rowsum(t.dxpre, t.grad_buffer("visual.visual_adapter.pos_emb", self.le.adapter_pos), B, T, D, T + 1, 1)


class ImageTowerTrainer:
    """An image-modality tower (conv1 patchify -> ViT trunk) under any lock recipe; the stem convolution's weight
    gradient is produced when `train_conv` (grouped unlock reaching the stem, transformer.py:566-573)."""

    def __init__(self, vit_engine, tower_kw=None, train_conv: bool = False):
        self.eng = vit_engine
        self.tower = TowerTrainer(vit_engine, param_prefix="visual.", **(tower_kw or {}))
        self.train_conv = train_conv
        self.ctx = None

    @property
    def grads(self):
        return self.tower.grads

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        e = self.eng
        p = e.cfg.patch
        cols, gh, gw = ops.im2col(x.contiguous().float(), p, p, p, p, e.conv_w.shape[1])
        tok = ops.gemm(cols, e.conv_w, None, epi=ops.EPI_BF16, cfg=e.gemm_cfg)
        self.ctx = cols
        return self.tower.forward(tok, x.shape[0])

    def backward(self, dfeat: torch.Tensor):
        dtok = self.tower.backward(dfeat)
        if self.train_conv:
            cols = self.ctx
            g = self.tower.grad_buffer("visual.conv1.weight_gemm", torch.empty(dtok.shape[1], cols.shape[1]))
            conv_weight_grad(dtok, cols, g, self.eng.gemm_cfg)


class _TextAsTower:
    """What TowerTrainer reads of an engine, for the text tower (a `TextEngine(arith="bf16")`: bf16 operands, fp32 residual stream)."""

    def __init__(self, te):
        from types import SimpleNamespace
        c = te.cfg
        self.cfg = SimpleNamespace(width=c.width, heads=c.heads, mlp_ratio=4.0, layers=c.layers)
        self.device, self.res_dtype, self.gemm_cfg = te.device, torch.float32, te.gemm_cfg
        self.blocks, self.projT, self.ln_post = te.blocks, te.projT, te.ln_final
        self.tok, self.pos = te.tok, te.pos


class TextTowerTrainer(TowerTrainer):
    """Forward with saved activations and backward of the TEXT tower (TriCLIP.encode_text, open_clip/model.py:528-540: token
    embedding + positional embedding -> 12 causal ResidualAttentionBlocks -> ln_final -> the EOT token's row @ text_projection)
    for runs that do NOT lock it (every ViT-Lens recipe does; `training/train.py:212-235` trains whatever requires grad).  The
    blocks are TowerTrainer's, with the additive causal mask (transformer.py:870-876) in the attention kernels; gradients come
    out under the reference's parameter names: token_embedding.weight, positional_embedding, transformer.resblocks.*, ln_final.*,
    text_projection.  bf16 operands on an fp32 residual stream (the frozen tower's fp16 operands are an inference choice)."""

    def __init__(self, text_engine):
        if getattr(text_engine, "arith", "bf16") != "bf16":
            raise ValueError("TextTowerTrainer needs a TextEngine(arith='bf16')")
        eng = _TextAsTower(text_engine)
        super().__init__(eng, train_blocks=range(eng.cfg.layers), param_prefix="", train_ln_post=True, train_proj=True)
        self.causal = True

    def forward(self, text: torch.Tensor) -> torch.Tensor:
        e, D = self.eng, self.D
        B, L = text.shape
        S = self.saved(B, L)
        text = text.to(e.device).contiguous()
        ops.text_embed(text, e.tok, e.pos, S.X[0])
        for l in range(self.layers):
            self._block_forward(S, l, B, L)
        eot = text.argmax(dim=-1).contiguous()            # index-exact EOT position (model.py:539)
        ops.layernorm(S.X[2 * self.layers], e.ln_post[0], e.ln_post[1], S.pooled, B, D, x_row_stride=D, row_index=eot, row_mul=L,
                      mean=S.post_stats[0], rstd=S.post_stats[1])
        feat = ops.gemm(S.pooled, e.projT, None, epi=ops.EPI_F32, cfg=e.gemm_cfg)
        self.ctx = (B, L, text, eot)
        return feat

    def backward(self, dfeat: torch.Tensor):
        """dfeat f32 [B, E]; fills self.grads (every text parameter)."""
        e, D = self.eng, self.D
        B, L, text, eot = self.ctx
        S = self.saved(B, L)
        cfg = e.gemm_cfg
        dfb = ops.cast_bf16(dfeat.contiguous())
        dpooled = ops.gemm(dfb, self.proj, None, epi=ops.EPI_BF16, cfg=cfg)                                  # [B, D]
        bp = (B + 63) // 64 * 64
        ops.gemm_dw(ops.transpose_to_bf16(S.pooled, ldo=bp), ops.transpose_to_bf16(dfb, ldo=bp),
                    self.grad_buffer("text_projection", torch.empty(D, dfeat.shape[1])), cfg=cfg)
        # ln_final acts on ONE row per caption, the EOT token's (an arbitrary position: gathered / scattered by index)
        rows = torch.arange(B, device=e.device) * L + eot
        x_eot = S.X[2 * self.layers].index_select(0, rows).contiguous()
        ops.layernorm_bwd_params(dpooled, x_eot, S.post_stats[0], S.post_stats[1], self.grad_buffer("ln_final.weight", e.ln_post[0]),
                                 self.grad_buffer("ln_final.bias", e.ln_post[1]), B, D)
        dx_eot = torch.empty(B, D, device=e.device, dtype=torch.float32)
        ops.layernorm_bwd(dpooled, x_eot, S.post_stats[0], S.post_stats[1], e.ln_post[0], B, D, dx=dx_eot)
        S.dx.zero_()
        S.dx.index_copy_(0, rows, dx_eot)
        self._blocks_backward(S, B, L)
        # x0 = token_embedding[text] + positional_embedding
        ops.batch_rowsum(S.dx, self.grad_buffer("positional_embedding", e.pos), B, L, D, L, 0)
        self.grad_buffer("token_embedding.weight", e.tok).index_add_(0, text.reshape(-1), S.dx)


class AdamW:
    """torch.optim.AdamW semantics on f32 master tensors, one fused kernel launch per tensor
    (reference: depth_tri_main.py:394-419 -- two groups: no weight decay for ndim<2 / bn / ln / bias / logit_scale)."""

    def __init__(self, params: Dict[str, torch.Tensor], lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2):
        # `param_groups[i]["lr"]` is what the reference's schedulers assign (training/scheduler.py:4-6); one group here,
        # weight decay is decided per tensor by `decays`
        self.param_groups = [{"lr": lr}]
        self.params, self.betas, self.eps, self.wd = params, betas, eps, weight_decay
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    @property
    def lr(self) -> float:
        return float(self.param_groups[0]["lr"])

    @lr.setter
    def lr(self, value):
        self.param_groups[0]["lr"] = float(value)

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


# ------------------------------------------------------------------------------------------------
# Perceiver ("Lens") training: autograd of Perceiver.forward (open_clip/perceiver.py:289-328)
# ------------------------------------------------------------------------------------------------
class _AttnSaved:
    """Saved operands of one Perceiver attention.  The projections' outputs are kept as token-major matrices (packed
    [q|k|v] for latent self-attention, q and [k|v] for cross-attention); the kernels read them by heads in place."""

    def __init__(self, B, H, Lq, Lk, dh, device, packed: bool):
        bf = lambda *s: torch.empty(*s, device=device, dtype=BF)
        inner = H * dh
        if packed:
            self.qkv2 = bf(B * Lq, 3 * inner)
            self.q2, self.kv2 = self.qkv2, None
            self.q, self.k, self.v = (ops.heads_view(self.qkv2, B, Lq, H, dh, i * inner) for i in range(3))
        else:
            self.q2, self.kv2 = bf(B * Lq, inner), bf(B * Lk, 2 * inner)
            self.q = ops.heads_view(self.q2, B, Lq, H, dh)
            self.k, self.v = ops.heads_view(self.kv2, B, Lk, H, dh), ops.heads_view(self.kv2, B, Lk, H, dh, inner)
        self.a = bf(B * Lq, inner)
        self.av = ops.heads_view(self.a, B, Lq, H, dh)
        self.lse = torch.empty(B, H, Lq, device=device, dtype=torch.float32)
        self.dOm = bf(B * Lq, inner)
        self.dO = ops.heads_view(self.dOm, B, Lq, H, dh)
        self.delta = torch.empty(B, H, Lq, device=device, dtype=torch.float32)


def deinterleave_geglu(g: torch.Tensor) -> torch.Tensor:
    """gradient / weight in the kernels' interleaved (a_j, gate_j) row order -> the reference's [a ; gate] order."""
    return torch.cat([g[0::2], g[1::2]], dim=0)


class PerceiverTrainer:
    def __init__(self, pe, param_prefix: str = "visual.perceiver."):
        self.pe, self.P = pe, param_prefix
        c = pe.cfg
        self.grads: Dict[str, torch.Tensor] = {}
        t = lambda w: w.t().contiguous()
        self.wT = []
        for li, lay in enumerate(pe.layers):
            if li >= 2 and lay is pe.layers[1]:           # tied layers (perceiver_weight_tie_layers): one set of transposes too
                self.wT.append(self.wT[1])
                continue
            d = {"x": {"q": t(lay["x_attn"]["q_w"]), "kv": t(lay["x_attn"]["kv_w"]), "out": t(lay["x_attn"]["to_out_w"])},
                 "xff": {"w0": t(lay["x_ff"]["w0"]), "w2": t(lay["x_ff"]["w2"])}, "selfs": []}
            for sl in lay["selfs"]:
                d["selfs"].append({"qkv": t(sl["attn"]["qkv_w"]), "out": t(sl["attn"]["to_out_w"]),
                                   "w0": t(sl["ff"]["w0"]), "w2": t(sl["ff"]["w2"])})
            self.wT.append(d)
        self._st = {}

    def refresh_derived(self):
        """The Perceiver engine's operands were updated in place: redo the transposes."""
        for li, lay in enumerate(self.pe.layers):
            if li >= 2 and lay is self.pe.layers[1]:
                continue
            d = self.wT[li]
            d["x"]["q"].copy_(lay["x_attn"]["q_w"].t()); d["x"]["kv"].copy_(lay["x_attn"]["kv_w"].t())
            d["x"]["out"].copy_(lay["x_attn"]["to_out_w"].t())
            d["xff"]["w0"].copy_(lay["x_ff"]["w0"].t()); d["xff"]["w2"].copy_(lay["x_ff"]["w2"].t())
            for ds, sl in zip(d["selfs"], lay["selfs"]):
                ds["qkv"].copy_(sl["attn"]["qkv_w"].t()); ds["out"].copy_(sl["attn"]["to_out_w"].t())
                ds["w0"].copy_(sl["ff"]["w0"].t()); ds["w2"].copy_(sl["ff"]["w2"].t())

    def grad_buffer(self, name, shape):
        g = self.grads.get(name)
        if g is None:
            g = torch.zeros(tuple(shape), device=self.pe.device, dtype=torch.float32)
            self.grads[name] = g
        return g

    def layer_name(self, li: int) -> int:
        """Index under which layer li's parameters are named: tied layers (>= 2) are layer 1's modules, so their
        gradients accumulate in layer 1's buffers (what autograd does for the reference's shared modules)."""
        return 1 if li >= 2 and self.pe.layers[li] is self.pe.layers[1] else li

    def _state(self, B, Tc):
        key = (B, Tc)
        if key in self._st:
            return self._st[key]
        c, dev = self.pe.cfg, self.pe.device
        n, D = c.num_latents, c.latent_dim
        R = B * n
        f32 = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        bf = lambda *s: torch.empty(*s, device=dev, dtype=BF)
        nres = c.depth * (2 + 2 * c.self_per_cross) + 1
        st = {"X": [f32(R, D) for _ in range(nres)],
              "dx": f32(R, D), "dxb": bf(R, D), "dh8": bf(R, 8 * D), "dhn": bf(R, D), "dqkv": bf(R, 3 * c.latent_heads * c.latent_dim_head),
              "dq": bf(R, c.cross_heads * c.cross_dim_head), "dkv": bf(B * Tc, 2 * c.cross_heads * c.cross_dim_head),
              "dctx": bf(B * Tc, c.input_chan), "ddata": f32(B * Tc, c.input_chan), "layers": []}
        for _ in range(c.depth):
            # (kept per sub-layer for the backward's weight gradients: every LayerNorm output `*_hn` / `c_n` and every GEGLU
            #  output `*_hid` - recomputing them was 2 % of the C4 step's kernel time for 2-7 GB of a 288 GB card)
            lay = {"x_stats": [f32(R), f32(R)], "c_stats": [f32(B * Tc), f32(B * Tc)], "x_hn": bf(R, D), "c_n": bf(B * Tc, c.input_chan),
                   "x_attn": _AttnSaved(B, c.cross_heads, n, Tc, c.cross_dim_head, dev, packed=False),
                   "xff_stats": [f32(R), f32(R)], "xff_h": bf(R, 8 * D), "xff_hn": bf(R, D), "xff_hid": bf(R, 4 * D), "selfs": []}
            for _ in range(c.self_per_cross):
                lay["selfs"].append({"stats": [f32(R), f32(R)], "attn": _AttnSaved(B, c.latent_heads, n, n, c.latent_dim_head, dev, packed=True),
                                     "hn": bf(R, D), "ff_stats": [f32(R), f32(R)], "ff_h": bf(R, 8 * D), "ff_hn": bf(R, D),
                                     "ff_hid": bf(R, 4 * D)})
            st["layers"].append(lay)
        self._st[key] = st
        return st

    # ---- forward ----------------------------------------------------------------------------------
    def _ff_fwd(self, st, xi, norm, ff, stats, hsave, hn, hid, rows, D):
        X, cfg = st["X"], self.pe.gemm_cfg
        ops.layernorm(X[xi], norm[0], norm[1], hn, rows, D, mean=stats[0], rstd=stats[1])
        ops.gemm(hn, ff["w0"], ff["b0"], out=hid, epi=ops.EPI_GEGLU, cfg=cfg, out2=hsave)
        ops.gemm(hid, ff["w2"], ff["b2"], out=X[xi + 1], res=X[xi], epi=ops.EPI_RES_F32, cfg=cfg)

    def forward(self, data: torch.Tensor, B: int) -> torch.Tensor:
        pe, c = self.pe, self.pe.cfg
        Tc, n, D = data.shape[0] // B, c.num_latents, c.latent_dim
        st = self._state(B, Tc)
        X, cfg, rows = st["X"], pe.gemm_cfg, B * n
        X[0].view(B, n, D).copy_(pe.latents)
        xi = 0
        for li, lay in enumerate(pe.layers):
            S = st["layers"][li]
            a, A = lay["x_attn"], S["x_attn"]
            ops.layernorm(X[xi], lay["x_norm"][0], lay["x_norm"][1], S["x_hn"], rows, D, mean=S["x_stats"][0], rstd=S["x_stats"][1])
            ops.layernorm(data, lay["x_norm_ctx"][0], lay["x_norm_ctx"][1], S["c_n"], B * Tc, c.input_chan,
                          mean=S["c_stats"][0], rstd=S["c_stats"][1])
            ops.gemm(S["x_hn"], a["q_w"], None, out=A.q2, epi=ops.EPI_BF16, cfg=cfg)
            ops.gemm(S["c_n"], a["kv_w"], None, out=A.kv2, epi=ops.EPI_BF16, cfg=cfg)
            ops.attn_fwd(A.q, A.k, A.v, A.a, lse=A.lse, qscale=c.cross_dim_head ** -0.5 * ops.LOG2E)
            ops.gemm(A.a, a["to_out_w"], a["to_out_b"], out=X[xi + 1], res=X[xi], epi=ops.EPI_RES_F32, cfg=cfg)
            xi += 1
            self._ff_fwd(st, xi, lay["x_ff_norm"], lay["x_ff"], S["xff_stats"], S["xff_h"], S["xff_hn"], S["xff_hid"], rows, D)
            xi += 1
            for sj, sl in enumerate(lay["selfs"]):
                T = S["selfs"][sj]
                a, A = sl["attn"], T["attn"]
                ops.layernorm(X[xi], sl["norm"][0], sl["norm"][1], T["hn"], rows, D, mean=T["stats"][0], rstd=T["stats"][1])
                ops.gemm(T["hn"], a["qkv_w"], None, out=A.qkv2, epi=ops.EPI_BF16, cfg=cfg)
                ops.attn_fwd(A.q, A.k, A.v, A.a, lse=A.lse, qscale=c.latent_dim_head ** -0.5 * ops.LOG2E)
                ops.gemm(A.a, a["to_out_w"], a["to_out_b"], out=X[xi + 1], res=X[xi], epi=ops.EPI_RES_F32, cfg=cfg)
                xi += 1
                self._ff_fwd(st, xi, sl["ff_norm"], sl["ff"], T["ff_stats"], T["ff_h"], T["ff_hn"], T["ff_hid"], rows, D)
                xi += 1
        self.ctx = (B, Tc, data)
        return X[xi]

    # ---- backward ---------------------------------------------------------------------------------
    def _dw(self, name, dy, x, rows):
        g = self.grad_buffer(name, (dy.shape[1], x.shape[1]))
        rp = (rows + 63) // 64 * 64
        # token-major operands, no transposed copies (narrow projections: the small operand zero-padded to whole tiles)
        if rows == dy.shape[0] == x.shape[0] and ops.gemm_dw_tn_any(dy, x, g):
            return
        ops.gemm_dw(ops.transpose_to_bf16(dy, ldo=rp), ops.transpose_to_bf16(x, ldo=rp), g, cfg=self.pe.gemm_cfg)

    def _ln_params(self, name, dy, x, stats, rows, D):
        ops.layernorm_bwd_params(dy, x, stats[0], stats[1], self.grad_buffer(name + ".weight", (D,)),
                                 self.grad_buffer(name + ".bias", (D,)), rows, D)

    def _ff_bwd(self, st, xi, norm, ff, wT, stats, hsave, hn, hid, rows, D, pname):
        """x_{xi+1} = x_xi + W2 geglu(W0 LN(x_xi) + b0) + b2 ; st['dx'] holds dL/dx_{xi+1} on entry, dL/dx_xi on exit.
        hn = LN(x_xi), hid = geglu(...) as the forward left them."""
        X, cfg = st["X"], self.pe.gemm_cfg
        # (dy = the bf16 copy of the residual gradient the dX GEMMs read: token-major operands for the dW kernel - the fp32
        #  stream would be rounded to the same bf16 values on its way through a transposed copy)
        self._dw(pname + "1.fn.net.2.weight", st["dxb"], hid, rows)
        ops.colsum(st["dx"], self.grad_buffer(pname + "1.fn.net.2.bias", (D,)))
        ops.gemm(st["dxb"], wT["w2"], None, out=st["dh8"], res=hsave, epi=ops.EPI_DGEGLU, cfg=cfg)     # d(pre-activation), interleaved
        self._dw(pname + "1.fn.net.0.weight_il", st["dh8"], hn, rows)
        ops.colsum(st["dh8"], self.grad_buffer(pname + "1.fn.net.0.bias_il", (8 * D,)))
        ops.gemm(st["dh8"], wT["w0"], None, out=st["dhn"], epi=ops.EPI_BF16, cfg=cfg)
        self._ln_params(pname + "1.norm", st["dhn"], X[xi], stats, rows, D)
        ops.layernorm_bwd(st["dhn"], X[xi], stats[0], stats[1], norm[0], rows, D, dres=st["dx"], dx=st["dx"], dx_bf16=st["dxb"])

    def backward(self, dlat: torch.Tensor) -> torch.Tensor:
        """dlat f32 [B*n, D] = dL/d(latent output) -> returns dL/d(data) f32 [B*Tc, C]; fills self.grads
        (GEGLU first-layer gradients are kept in the kernels' interleaved row order under '*_il' names)."""
        pe, c = self.pe, self.pe.cfg
        B, Tc, data = self.ctx
        n, D = c.num_latents, c.latent_dim
        st = self._state(B, Tc)
        X, cfg, rows, P = st["X"], pe.gemm_cfg, B * n, self.P
        st["dx"].copy_(dlat)
        ops.cast_bf16(st["dx"], out=st["dxb"])
        st["ddata"].zero_()
        xi = len(X) - 1
        for li in reversed(range(c.depth)):
            lay, S, wT = pe.layers[li], st["layers"][li], self.wT[li]
            for sj in reversed(range(c.self_per_cross)):
                sl, T, w = lay["selfs"][sj], S["selfs"][sj], wT["selfs"][sj]
                pn = f"{P}layers.{self.layer_name(li)}.2.{sj}."
                xi -= 1
                self._ff_bwd(st, xi, sl["ff_norm"], sl["ff"], w, T["ff_stats"], T["ff_h"], T["ff_hn"], T["ff_hid"], rows, D, pn)
                xi -= 1
                a, A = sl["attn"], T["attn"]
                H, dh = c.latent_heads, c.latent_dim_head
                inner = H * dh
                self._dw(pn + "0.fn.to_out.weight", st["dxb"], A.a, rows)
                ops.colsum(st["dx"], self.grad_buffer(pn + "0.fn.to_out.bias", (D,)))
                ops.gemm(st["dxb"], w["out"], None, out=A.dOm, epi=ops.EPI_BF16, cfg=cfg)
                dqkv = st["dqkv"]
                ops.attn_bwd(A.q, A.k, A.v, A.dO, A.av, A.lse, A.delta, dqkv, dqkv[:, inner:], dqkv[:, 2 * inner:],
                             3 * inner, 3 * inner)
                self._dw(pn + "0.fn.to_qkv.weight", dqkv, T["hn"], rows)            # rows [to_q ; to_kv]
                ops.gemm(dqkv, w["qkv"], None, out=st["dhn"], epi=ops.EPI_BF16, cfg=cfg)
                self._ln_params(pn + "0.norm", st["dhn"], X[xi], T["stats"], rows, D)
                ops.layernorm_bwd(st["dhn"], X[xi], T["stats"][0], T["stats"][1], sl["norm"][0], rows, D, dres=st["dx"],
                                  dx=st["dx"], dx_bf16=st["dxb"])
            pn = f"{P}layers.{self.layer_name(li)}."
            xi -= 1
            self._ff_bwd(st, xi, lay["x_ff_norm"], lay["x_ff"], wT["xff"], S["xff_stats"], S["xff_h"], S["xff_hn"], S["xff_hid"], rows, D, pn)
            xi -= 1
            a, A, w = lay["x_attn"], S["x_attn"], wT["x"]
            H, dh = c.cross_heads, c.cross_dim_head
            inner = H * dh
            self._dw(pn + "0.fn.to_out.weight", st["dxb"], A.a, rows)
            ops.colsum(st["dx"], self.grad_buffer(pn + "0.fn.to_out.bias", (D,)))
            ops.gemm(st["dxb"], w["out"], None, out=A.dOm, epi=ops.EPI_BF16, cfg=cfg)
            dkv = st["dkv"]
            ops.attn_bwd(A.q, A.k, A.v, A.dO, A.av, A.lse, A.delta, st["dq"], dkv, dkv[:, inner:], inner, 2 * inner)
            # query side: LN(x) -> to_q
            self._dw(pn + "0.fn.to_q.weight", st["dq"], S["x_hn"], rows)
            ops.gemm(st["dq"], w["q"], None, out=st["dhn"], epi=ops.EPI_BF16, cfg=cfg)
            self._ln_params(pn + "0.norm", st["dhn"], X[xi], S["x_stats"], rows, D)
            ops.layernorm_bwd(st["dhn"], X[xi], S["x_stats"][0], S["x_stats"][1], lay["x_norm"][0], rows, D, dres=st["dx"],
                              dx=st["dx"], dx_bf16=st["dxb"])
            # context side: LN_ctx(data) -> to_kv ; every cross layer reads the same data -> accumulate
            C = c.input_chan
            self._dw(pn + "0.fn.to_kv.weight", dkv, S["c_n"], B * Tc)
            ops.gemm(dkv, w["kv"], None, out=st["dctx"], epi=ops.EPI_BF16, cfg=cfg)
            self._ln_params(pn + "0.norm_context", st["dctx"], data, S["c_stats"], B * Tc, C)
            ops.layernorm_bwd(st["dctx"], data, S["c_stats"][0], S["c_stats"][1], lay["x_norm_ctx"][0], B * Tc, C,
                              dres=st["ddata"], dx=st["ddata"])
        ops.batch_rowsum(st["dx"], self.grad_buffer(P + "latents", (n, D)), B, n, D, n, 0)
        return st["ddata"]

    def reference_named_grads(self, grads: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Gradients under the reference's parameter names/layouts (de-interleaved GEGLU, split to_q/to_kv); `grads`: another
        dictionary in the kernels' layout (a fused step's merged buffer) instead of this trainer's own."""
        out = {}
        c = self.pe.cfg
        for k, g in (self.grads if grads is None else grads).items():
            if k.endswith("net.0.weight_il"):
                out[k[:-3]] = deinterleave_geglu(g)
            elif k.endswith("net.0.bias_il"):
                out[k[:-3]] = deinterleave_geglu(g)
            elif k.endswith("to_qkv.weight"):
                inner = c.latent_heads * c.latent_dim_head
                out[k.replace("to_qkv", "to_q")] = g[:inner]; out[k.replace("to_qkv", "to_kv")] = g[inner:]
            else:
                out[k] = g
        return out


class AudioLensTrainer:
    """`visual.` tower of the audio recipe (TRAIN_INFERENCE.md:283-299): AST tokenizer + Perceiver trainable,
    ViT blocks locked, class_embedding unlocked (--lock-visual --unlock-cls)."""

    def __init__(self, lens_engine, tower_kw=None):
        self.le = lens_engine
        kw = dict(train_blocks=(), train_cls=True) if tower_kw is None else dict(tower_kw)
        self.tower = TowerTrainer(lens_engine.vit, param_prefix="visual.", **kw)
        self.perc = PerceiverTrainer(lens_engine.perceiver, "visual.perceiver.")
        self.perc.grads = self.tower.grads            # one gradient dictionary
        self.ctx = None

    @property
    def grads(self):
        return self.tower.grads

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        le, L = self.le, self.le.lens
        p = le.tower.patch
        B = x.shape[0]
        cols, gh, gw = ops.im2col(x.contiguous().float().unsqueeze(1), p, p, L.audio_fstride, L.audio_tstride,
                                  le.conv_w.shape[1], transpose_hw=True)
        tok = ops.gemm(cols, le.conv_w, None, epi=ops.EPI_BF16, cfg=le.gemm_cfg)
        T = tok.shape[0] // B
        xin = torch.empty_like(tok)
        ops.add_rows(tok, le.adapter_pos, xin, tok.shape[0], T, tok.shape[1])
        lat = self.perc.forward(xin, B)
        self.ctx = (cols, B, T)
        return self.tower.forward(lat, B)

    def backward(self, dfeat: torch.Tensor):
        cols, B, T = self.ctx
        dlat = self.tower.backward(dfeat)
        ddata = self.perc.backward(dlat)                          # f32 [B*T, D]
        D = ddata.shape[1]
        g = self.tower.grad_buffer("visual.visual_adapter.conv1.weight_gemm", torch.empty(D, cols.shape[1]))
        conv_weight_grad(ddata, cols, g, self.le.gemm_cfg)
        ops.batch_rowsum(ddata, self.tower.grad_buffer("visual.visual_adapter.pos_emb", self.le.adapter_pos), B, T, D, T, 0)


class EEGLensTrainer(AudioLensTrainer):
    """`visual.` tower of the EEG recipe (mm_vit_lens/model_cfg.py:153-178): PatchEmbed1D (Conv1d with bias over the
    time axis) + pos_emb + Perceiver trainable in front of the locked ViT - the audio recipe with a 1-D tokenizer."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        le = self.le
        B = x.shape[0]
        cols = le.eeg_cols(x)
        tok = ops.gemm(cols, le.conv_w, le.conv_b, epi=ops.EPI_BF16, cfg=le.gemm_cfg)
        T = tok.shape[0] // B
        xin = torch.empty_like(tok)
        ops.add_rows(tok, le.adapter_pos, xin, tok.shape[0], T, tok.shape[1])
        lat = self.perc.forward(xin, B)
        self.ctx = (cols, B, T)
        return self.tower.forward(lat, B)

    def backward(self, dfeat: torch.Tensor):
        cols, B, T = self.ctx
        dlat = self.tower.backward(dfeat)
        ddata = self.perc.backward(dlat)                          # f32 [B*T, D]: gradient of tokens + pos
        D = ddata.shape[1]
        g = self.tower.grad_buffer("visual.visual_adapter.proj.weight_gemm", torch.empty(D, cols.shape[1]))
        conv_weight_grad(ddata, cols, g, self.le.gemm_cfg)
        ops.colsum(ddata, self.tower.grad_buffer("visual.visual_adapter.proj.bias", self.le.conv_b))
        ops.batch_rowsum(ddata, self.tower.grad_buffer("visual.visual_adapter.pos_emb", self.le.adapter_pos), B, T, D, T, 0)


class PCLensTrainer:
    """`visual.` tower of the point-cloud recipe: PointBERT tokenizer + Perceiver trainable, ViT blocks locked.
    `tok` is the shared PointTokenizerTrainer (masters, bf16 operands, running statistics); each micro-batch gets
    a shallow copy that only owns its saved activations."""

    def __init__(self, lens_engine, tok, train_cls: bool = False, tower_kw=None):
        import copy
        self.le = lens_engine
        kw = dict(train_blocks=(), train_cls=train_cls) if tower_kw is None else dict(tower_kw)
        self.tower = TowerTrainer(lens_engine.vit, param_prefix="visual.", **kw)
        self.perc = PerceiverTrainer(lens_engine.perceiver, "visual.perceiver.")
        self.perc.grads = self.tower.grads
        self.tok = copy.copy(tok)
        self.tok.ctx = None
        self.tok.grads = self.tower.grads

    @property
    def grads(self):
        return self.tower.grads

    def forward(self, pts: torch.Tensor, fps_start=None, **kw) -> torch.Tensor:
        """pts [B,N,3] (PointBERT tokenizer) or point features [B,N,in_dim] with xyz=[B,N,3] (pnsa tokenizer)."""
        B = pts.shape[0]
        ctx = self.tok.forward(pts, fps_start=fps_start, **kw)    # tokens (+ pos), bf16 [B*G, C]
        return self.tower.forward(self.perc.forward(ctx, B), B)

    def backward(self, dfeat: torch.Tensor):
        self.tok.backward(self.perc.backward(self.tower.backward(dfeat)))


def refresh_trainer(tr, sd, names):
    """After the engine under trainer `tr` was updated in place for the parameters `names` (relative to the tower prefix):
    refresh what the trainer derived from them - transposed weights, the point tokenizer's masters - keeping the trainer
    and its activation buffers (a rebuild re-allocates tens of GB at ViT-L size)."""
    tower = getattr(tr, "tower", None)
    if tower is not None:
        tower.refresh_derived({int(n.split(".")[2]) for n in names if n.startswith("transformer.resblocks.")}, "proj" in names)
    perc = getattr(tr, "perc", None)
    if perc is not None and any(n.startswith("perceiver.") for n in names):
        perc.refresh_derived()
    tok = getattr(tr, "tok", None)
    if tok is not None and any(n.startswith("visual_adapter.") for n in names):
        tok.load_params(sd)
