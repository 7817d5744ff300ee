"""Forward executors for the ViT / text towers on the HIP kernels.

`VitEngine` runs  tokens -> [cls;tokens]+pos -> ln_pre -> N x ResidualAttentionBlock -> ln_post(cls)
-> @proj  (open_clip/transformer.py:756-787) and `TextEngine` runs TriCLIP.encode_text
(open_clip/model.py:528-540), both as a fixed sequence of C-ABI calls on the current stream with
pre-allocated workspaces (no allocation inside the block loop -> hipGraph-capturable).

Precision contract (= the reference under torch autocast): GEMM operands bf16, accumulation fp32,
LayerNorm / softmax statistics / residual stream / final features fp32 (residual dtype selectable).
"""
from dataclasses import dataclass
from typing import Dict, Optional

import os

import torch

from . import ops


@dataclass
class TowerCfg:
    width: int = 1024
    layers: int = 24
    heads: int = 16
    mlp_ratio: float = 4.0
    patch: int = 14
    image_size: int = 224
    embed_dim: int = 768
    in_chans: int = 3


@dataclass
class TextCfg:
    context_length: int = 77
    vocab_size: int = 49408
    width: int = 768
    heads: int = 12
    layers: int = 12
    embed_dim: int = 768


def _pad64(n):
    return (n + 63) // 64 * 64


def _dev(t, device, dtype=torch.float32):
    """Device copy owned by the engine: never an alias of the caller's tensor (a fused step updates these in place through
    raw kernels, which would otherwise modify the model's Parameters behind autograd's back)."""
    out = t.detach().to(device=device, dtype=dtype).contiguous()
    return out.clone() if out.data_ptr() == t.data_ptr() else out


def prep_block(sd: Dict[str, torch.Tensor], p: str, device) -> Dict[str, torch.Tensor]:
    """Device copies of one ResidualAttentionBlock: GEMM weights bf16, the rest f32; with the LayerNorm folding switched on
    also the operands of the two LayerNorm -> Linear pairs with the LayerNorm folded in (ops.fold_ln_linear; used for frozen
    blocks on a bf16 stream)."""
    bf = torch.bfloat16
    dv = lambda k: sd[p + k].detach().to(device)
    blk = _prep_block_plain(sd, p, device)
    if LN_FOLD:          # (read at call time: engines built while the switch is on carry the folded operands)
        blk["in_f"] = ops.fold_ln_linear(dv("attn.in_proj_weight"), dv("attn.in_proj_bias"), dv("ln_1.weight"), dv("ln_1.bias"))
        blk["fc_f"] = ops.fold_ln_linear(dv("mlp.c_fc.weight"), dv("mlp.c_fc.bias"), dv("ln_2.weight"), dv("ln_2.bias"))
    return blk


def _prep_block_plain(sd: Dict[str, torch.Tensor], p: str, device) -> Dict[str, torch.Tensor]:
    bf = torch.bfloat16
    return {
        "ln1_w": _dev(sd[p + "ln_1.weight"], device), "ln1_b": _dev(sd[p + "ln_1.bias"], device),
        "in_w": _dev(sd[p + "attn.in_proj_weight"], device, bf), "in_b": _dev(sd[p + "attn.in_proj_bias"], device),
        "out_w": _dev(sd[p + "attn.out_proj.weight"], device, bf), "out_b": _dev(sd[p + "attn.out_proj.bias"], device),
        "ln2_w": _dev(sd[p + "ln_2.weight"], device), "ln2_b": _dev(sd[p + "ln_2.bias"], device),
        "fc_w": _dev(sd[p + "mlp.c_fc.weight"], device, bf), "fc_b": _dev(sd[p + "mlp.c_fc.bias"], device),
        "proj_w": _dev(sd[p + "mlp.c_proj.weight"], device, bf), "proj_b": _dev(sd[p + "mlp.c_proj.bias"], device),
    }


def conv_weight_as_gemm(w: torch.Tensor, device, dtype=torch.bfloat16) -> torch.Tensor:
    """Conv2d weight [O,C,kh,kw] -> [O, Kp] (K = C*kh*kw zero-padded to a multiple of 64), bf16 operand or f32 master.
    Built where the weight lives (no host round trip for a parameter that is already on the GPU)."""
    O = w.shape[0]
    K = w[0].numel()
    out = torch.zeros(O, _pad64(K), dtype=torch.float32, device=w.device)
    out[:, :K] = w.detach().reshape(O, K).float()
    return out.to(device=device, dtype=dtype).contiguous()


def copy_tree(dst, src):
    """dst <- src, in place, for every tensor of two identically shaped nests of dict / list / tuple.  The engines are
    updated through this after an optimizer step: object identities (and with them the trainers' activation buffers and
    every tensor another object holds a reference to) survive, only the trainable operands are re-derived."""
    if torch.is_tensor(dst):
        dst.copy_(src)
    elif isinstance(dst, dict):
        for k in dst:
            copy_tree(dst[k], src[k])
    else:
        for d, s_ in zip(dst, src):
            copy_tree(d, s_)


class _Workspace:
    def __init__(self, B, L, D, H, hidden, device, res_dtype):
        dh = D // H
        bf = torch.bfloat16
        self.Lp = (L + 7) // 8 * 8
        self.x = torch.empty(B * L, D, device=device, dtype=res_dtype)
        self.h = torch.empty(B * L, D, device=device, dtype=bf)
        self.qkv = torch.empty(B * L, 3 * D, device=device, dtype=bf)     # packed in-projection output, read in place
        hv = lambda i: ops.heads_view(self.qkv, B, L, H, dh, i * D)
        self.q, self.k, self.v = hv(0), hv(1), hv(2)
        self.a = torch.empty(B * L, D, device=device, dtype=bf)
        self.hid = torch.empty(B * L, hidden, device=device, dtype=bf)
        # LayerNorm folding: row statistics of the residual rows and the producing GEMMs' partial sums
        self.mean = torch.empty(B * L, device=device, dtype=torch.float32)
        self.rstd = torch.empty(B * L, device=device, dtype=torch.float32)
        self.part = torch.empty(B * L * (D // 64) * 2, device=device, dtype=torch.float32) if D % 64 == 0 else None


# LayerNorm folding is ON by default since round 5 (the whole GPU suite is green with it: profiles/r05_pytest_gpu_lnfold_on_*.log;
# it takes 88 of 96 LayerNorm passes and 270 MB of HBM traffic out of a frozen ViT-L micro-batch; +0.25 % on the C3 step - the
# chip is power-limited and gives most of the removed time back as lower GEMM clocks).  A module attribute,
# not an environment variable: `engine.LN_FOLD = False` (or `fold=False` on run_blocks, `bench.py --ln-fold off`) selects the
# separate LayerNorm passes; engines read it when they are built and when they run.
LN_FOLD = True


def run_blocks(blocks, ws: _Workspace, B, L, D, H, causal=False, cfg=-1, fold=None):
    """x (ws.x, residual stream) <- N pre-LN transformer blocks (transformer.py:254-272, 364-371).
    fold (default: engine.LN_FOLD, bf16 stream only): the LayerNorms folded into the GEMMs either side of them - ln_1 / ln_2 are never
    materialised, the in-projection and c_fc read the residual rows and apply (mean, rstd) in their epilogues, the
    out-projection and c_proj leave the partial row sums of what they store (ops.gemm_lnfold / gemm_res_rowstats)."""
    dh = D // H
    res_epi = ops.EPI_RES_F32 if ws.x.dtype == torch.float32 else ops.EPI_RES_BF16
    if fold is None:
        fold = LN_FOLD
    can_fold = bool(fold) and ws.x.dtype == torch.bfloat16 and ws.part is not None
    # decided PER BLOCK: a block folds when it carries the folded operands (prep_block) - a fused training step removes them
    # from the blocks it trains (their LayerNorm parameters move every step) and those run the LayerNorm passes
    folded = [can_fold and "in_f" in w for w in blocks]
    mm = 0                                            # rows of ws.x whose partial sums are in ws.part
    r_in = r_fc = 0
    if any(folded):
        r_in, r_fc = ops.fold_rows(ws.x, ws.qkv, 3 * D), ops.fold_rows(ws.x, ws.hid, ws.hid.shape[1])
    for i, w in enumerate(blocks):
        nxt = i + 1 < len(blocks) and folded[i + 1]   # the consumer of this block's output reads row statistics
        if folded[i]:
            # (the row-statistics launch also writes the LayerNorm output of the consuming GEMM's leftover rows into ws.h)
            k_in = dict(ln_w=w["ln1_w"], ln_b=w["ln1_b"], h_left=ws.h, h_row0=r_in) if mm <= r_in else {}
            ops.ln_row_stats(ws.part, ws.x, mm, ws.mean, ws.rstd, **k_in)
            ops.gemm_lnfold(ws.x, w["in_f"], ws.mean, ws.rstd, ws.qkv, w["in_w"], w["in_b"], w["ln1_w"], w["ln1_b"], ws.h, cfg=cfg,
                            h_ready=bool(k_in))
            ops.attn_fwd(ws.q, ws.k, ws.v, ws.a, causal=causal, qscale=dh ** -0.5 * ops.LOG2E)
            mm = ops.gemm_res_rowstats(ws.a, w["out_w"], w["out_b"], ws.x, ws.x, ws.part, cfg=cfg)
            k_fc = dict(ln_w=w["ln2_w"], ln_b=w["ln2_b"], h_left=ws.h, h_row0=r_fc) if mm <= r_fc else {}
            ops.ln_row_stats(ws.part, ws.x, mm, ws.mean, ws.rstd, **k_fc)
            ops.gemm_lnfold(ws.x, w["fc_f"], ws.mean, ws.rstd, ws.hid, w["fc_w"], w["fc_b"], w["ln2_w"], w["ln2_b"], ws.h,
                            act=ops.ACT_GELU, cfg=cfg, h_ready=bool(k_fc))
        else:
            ops.layernorm(ws.x, w["ln1_w"], w["ln1_b"], ws.h, B * L, D)
            ops.gemm(ws.h, w["in_w"], w["in_b"], out=ws.qkv, epi=ops.EPI_BF16, cfg=cfg)
            ops.attn_fwd(ws.q, ws.k, ws.v, ws.a, causal=causal, qscale=dh ** -0.5 * ops.LOG2E)
            ops.gemm(ws.a, w["out_w"], w["out_b"], out=ws.x, res=ws.x, epi=res_epi, cfg=cfg)
            ops.layernorm(ws.x, w["ln2_w"], w["ln2_b"], ws.h, B * L, D)
            ops.gemm(ws.h, w["fc_w"], w["fc_b"], out=ws.hid, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg)
        if nxt:
            mm = ops.gemm_res_rowstats(ws.hid, w["proj_w"], w["proj_b"], ws.x, ws.x, ws.part, cfg=cfg)
        else:
            ops.gemm(ws.hid, w["proj_w"], w["proj_b"], out=ws.x, res=ws.x, epi=res_epi, cfg=cfg)
            mm = 0


def _hi_lo(w: torch.Tensor, device):
    """fp32 weight -> (hi, lo) bf16 with hi + lo = w to ~2^-17 relative (hi = bf16(w), lo = bf16(w - hi))."""
    w = w.detach().float().to(device)
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    return hi.contiguous(), lo.contiguous()


def prep_block_wsplit(sd: Dict[str, torch.Tensor], p: str, device) -> Dict[str, torch.Tensor]:
    """One ResidualAttentionBlock with TWO-TERM bf16 weights (`TextEngine(wsplit=True)`): the projections that write a
    bf16 activation take the pair K-concatenated ([N, 2K] against an activation row [a, a]); the two residual projections
    run as two launches (hi, then lo) that accumulate into the fp32 residual stream."""
    blk = prep_block(sd, p, device)
    for key, name in (("in_w", "attn.in_proj_weight"), ("fc_w", "mlp.c_fc.weight")):
        hi, lo = _hi_lo(sd[p + name], device)
        blk[key + "2"] = torch.cat([hi, lo], dim=1).contiguous()
    for key, name in (("out_w", "attn.out_proj.weight"), ("proj_w", "mlp.c_proj.weight")):
        blk[key], blk[key + "_lo"] = _hi_lo(sd[p + name], device)
    return blk


def run_blocks_wsplit(blocks, ws: "_Workspace", B, L, D, H, causal=False, cfg=-1):
    """run_blocks with two-term weights (prep_block_wsplit); the residual stream must be fp32."""
    dh = D // H
    h2 = ws.h2
    for w in blocks:
        ops.layernorm(ws.x, w["ln1_w"], w["ln1_b"], h2[:, :D], B * L, D)
        ops.layernorm(ws.x, w["ln1_w"], w["ln1_b"], h2[:, D:], B * L, D)
        ops.gemm(h2, w["in_w2"], w["in_b"], out=ws.qkv, epi=ops.EPI_BF16, cfg=cfg)
        ops.attn_fwd(ws.q, ws.k, ws.v, ws.a, causal=causal, qscale=dh ** -0.5 * ops.LOG2E)
        ops.gemm(ws.a, w["out_w"], w["out_b"], out=ws.x, res=ws.x, epi=ops.EPI_RES_F32, cfg=cfg)
        ops.gemm(ws.a, w["out_w_lo"], None, out=ws.x, res=ws.x, epi=ops.EPI_RES_F32, cfg=cfg)
        ops.layernorm(ws.x, w["ln2_w"], w["ln2_b"], h2[:, :D], B * L, D)
        ops.layernorm(ws.x, w["ln2_w"], w["ln2_b"], h2[:, D:], B * L, D)
        ops.gemm(h2, w["fc_w2"], w["fc_b"], out=ws.hid, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg)
        ops.gemm(ws.hid, w["proj_w"], w["proj_b"], out=ws.x, res=ws.x, epi=ops.EPI_RES_F32, cfg=cfg)
        ops.gemm(ws.hid, w["proj_w_lo"], None, out=ws.x, res=ws.x, epi=ops.EPI_RES_F32, cfg=cfg)


class VitEngine:
    """One ViT tower (`image.` or `visual.` prefix of the TriCLIP state_dict) on the GPU."""

    def __init__(self, sd, prefix: str, cfg: TowerCfg, device, res_dtype=torch.float32, gemm_cfg: int = -1,
                 n_tokens: Optional[int] = None):
        self.cfg, self.device, self.res_dtype, self.gemm_cfg = cfg, torch.device(device), res_dtype, gemm_cfg
        self.prefix = prefix
        D = cfg.width
        self.cls = _dev(sd[prefix + "class_embedding"], device)
        self.pos = _dev(sd[prefix + "positional_embedding"], device)
        self.T = self.pos.shape[0] - 1
        self.ln_pre = (_dev(sd[prefix + "ln_pre.weight"], device), _dev(sd[prefix + "ln_pre.bias"], device))
        self.ln_post = (_dev(sd[prefix + "ln_post.weight"], device), _dev(sd[prefix + "ln_post.bias"], device))
        # [E, D]; None = the tower has no output projection (`if self.proj is not None`, transformer.py:783-784;
        # CLIPBindWrap drops it when the feature width differs, VitLens-OpenShape/src/models/clip_bind.py:35-47)
        self.projT = _dev(sd[prefix + "proj"].t(), device, torch.bfloat16) if prefix + "proj" in sd else None
        self.blocks = [prep_block(sd, f"{prefix}transformer.resblocks.{i}.", device) for i in range(cfg.layers)]
        self.conv_w = None
        if prefix + "conv1.weight" in sd:
            self.conv_w = conv_weight_as_gemm(sd[prefix + "conv1.weight"], device)
        self._ws = {}

    def update_params(self, sd, names):
        """Re-derive, in place, the device operands of the parameters `names` (relative to this tower's prefix) from `sd`."""
        p, dev = self.prefix, self.device
        top = {n for n in names if not n.startswith("transformer.resblocks.")}
        if "class_embedding" in top:
            self.cls.copy_(sd[p + "class_embedding"])
        if "positional_embedding" in top:
            self.pos.copy_(sd[p + "positional_embedding"])
        for nm, pair in (("ln_pre", self.ln_pre), ("ln_post", self.ln_post)):
            if nm + ".weight" in top or nm + ".bias" in top:
                pair[0].copy_(sd[p + nm + ".weight"]); pair[1].copy_(sd[p + nm + ".bias"])
        if "proj" in top and self.projT is not None:
            self.projT.copy_(sd[p + "proj"].t())
        if "conv1.weight" in top and self.conv_w is not None:
            self.conv_w.copy_(conv_weight_as_gemm(sd[p + "conv1.weight"], dev))
        for l in sorted({int(n.split(".")[2]) for n in names if n.startswith("transformer.resblocks.")}):
            copy_tree(self.blocks[l], prep_block(sd, f"{p}transformer.resblocks.{l}.", dev))

    def workspace(self, B, L):
        key = (B, L)
        if key not in self._ws:
            self._ws[key] = _Workspace(B, L, self.cfg.width, self.cfg.heads, int(self.cfg.width * self.cfg.mlp_ratio),
                                       self.device, self.res_dtype)
        return self._ws[key]

    # -- stages ---------------------------------------------------------------------------------
    def patch_tokens(self, image: torch.Tensor) -> torch.Tensor:
        """conv1 as im2col + GEMM (transformer.py:464-470,674-676): [N,C,H,W] f32 -> [N*T, D] bf16."""
        p = self.cfg.patch
        cols, gh, gw = ops.im2col(image.contiguous().float(), p, p, p, p, self.conv_w.shape[1])
        return ops.gemm(cols, self.conv_w, None, epi=ops.EPI_BF16, cfg=self.gemm_cfg)

    def trunk(self, tokens: torch.Tensor, B: int, pos2: Optional[torch.Tensor] = None, use_orig_pos=True):
        """tokens [B*T, D] (bf16|f32) -> un-normalised features f32 [B, E]."""
        cfg = self.cfg
        D, T = cfg.width, tokens.shape[0] // B
        L = T + 1
        ws = self.workspace(B, L)
        pos = self.pos if use_orig_pos else torch.zeros_like(self.pos)
        ops.assemble_ln_pre(tokens, self.cls, pos, pos2, self.ln_pre[0], self.ln_pre[1], ws.x, B, T, D)
        run_blocks(self.blocks, ws, B, L, D, cfg.heads, causal=False, cfg=self.gemm_cfg)
        pooled = torch.empty(B, D, device=self.device, dtype=torch.bfloat16)
        if self.projT is None:
            pooled = torch.empty(B, D, device=self.device, dtype=torch.float32)
        ops.layernorm(ws.x, self.ln_post[0], self.ln_post[1], pooled, B, D, x_row_stride=L * D)
        if self.projT is None:
            return pooled
        return ops.gemm(pooled, self.projT, None, epi=ops.EPI_F32, cfg=self.gemm_cfg)

    def encode_image(self, image: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        B = image.shape[0]
        f = self.trunk(self.patch_tokens(image), B)
        return ops.l2_normalize(f) if normalize else f


def prep_block_f16(sd: Dict[str, torch.Tensor], p: str, device) -> Dict[str, torch.Tensor]:
    """One ResidualAttentionBlock with IEEE-half GEMM weights (`TextEngine(arith="f16")`): LayerNorm parameters and biases f32."""
    hf = torch.float16
    return {
        "ln1_w": _dev(sd[p + "ln_1.weight"], device), "ln1_b": _dev(sd[p + "ln_1.bias"], device),
        "in_w": _dev(sd[p + "attn.in_proj_weight"], device, hf), "in_b": _dev(sd[p + "attn.in_proj_bias"], device),
        "out_w": _dev(sd[p + "attn.out_proj.weight"], device, hf), "out_b": _dev(sd[p + "attn.out_proj.bias"], device),
        "ln2_w": _dev(sd[p + "ln_2.weight"], device), "ln2_b": _dev(sd[p + "ln_2.bias"], device),
        "fc_w": _dev(sd[p + "mlp.c_fc.weight"], device, hf), "fc_b": _dev(sd[p + "mlp.c_fc.bias"], device),
        "proj_w": _dev(sd[p + "mlp.c_proj.weight"], device, hf), "proj_b": _dev(sd[p + "mlp.c_proj.bias"], device),
    }


class _WorkspaceF16:
    """Activations of the fp16 text tower.  Rows are padded to whole 256-row tiles (vl_gemm_f16 is the persistent kernel
    only); the padded rows are zero GEMM inputs, never normalised, never read by the attention or the pooling."""

    def __init__(self, B, L, D, H, device):
        hf = torch.float16
        self.rows = B * L
        self.Mp = (self.rows + 255) // 256 * 256
        self.Bp = (B + 255) // 256 * 256
        z = lambda *shape, dt=hf: torch.zeros(*shape, device=device, dtype=dt)
        self.x = z(self.Mp, D, dt=torch.float32)
        self.h, self.qkv, self.a, self.hid = z(self.Mp, D), z(self.Mp, 3 * D), z(self.Mp, D), z(self.Mp, 4 * D)
        dh = D // H
        hv = lambda i: ops.heads_view(self.qkv, B, L, H, dh, i * D)
        self.q, self.k, self.v = hv(0), hv(1), hv(2)
        self.pooled = z(self.Bp, D)


class TextEngine:
    """TriCLIP.encode_text (model.py:528-540): embedding + causal transformer + ln_final + EOT + proj.

    The tower is frozen in every recipe (forward only) and its cosine-similarity MATRIX amplifies operand rounding: random-init
    (and trained) text features share a mutual cosine of ~0.6, so with bf16 operands - the reference's amp_bf16 arithmetic -
    that matrix is 0.8-1.9e-3 from the fp32 CPU path (the reference's own amp_bf16 forward: 1.4e-3), above the 1e-3 of
    BASELINE.json's north_star.  CPU emulation per rounding point: weights 4.6e-4, GEMM input
    activations 5.0e-4, 16-bit stores 5.7e-4, attention internals 2.2e-4 taken alone.  `arith` selects the operands:
      "f16"    (default, round 5) every GEMM / attention operand IEEE half, fp32 residual stream, fp32 accumulation: three
               more mantissa bits on ALL four sources at the bf16 MFMA rate - 1.3-2.0e-4 emulated on four seeds.  (The
               reference converts CLIP to fp16 itself: convert_weights_to_fp16, model.py:393-419.)  Needs head dim 64 and a
               width that is a multiple of 256 and >= 512 (every CLIP text tower); otherwise falls back to "bf16x2"
      "bf16x2" (round 4) weights as the sum of TWO bf16 terms, fp32 residual stream: 6.1-8.1e-4 measured at twice the GEMM
               flops and twice the LayerNorm passes (+17 ms per C3 step)
      "bf16"   the reference's amp_bf16 arithmetic."""

    def __init__(self, sd, cfg: TextCfg, device, res_dtype=torch.float32, gemm_cfg: int = -1, wsplit: Optional[bool] = None,
                 arith: str = "f16"):
        self.cfg, self.device, self.gemm_cfg = cfg, torch.device(device), gemm_cfg
        if wsplit is not None:            # round-4 spelling
            arith = "bf16x2" if wsplit else "bf16"
        if arith not in ("f16", "bf16x2", "bf16"):
            raise ValueError(f"TextEngine: arith must be 'f16', 'bf16x2' or 'bf16', got {arith!r}")
        # (vl_gemm_f16 is the persistent kernel only: whole 256-wide tiles and K >= 512 - the in-projection's K is the width)
        if arith == "f16" and (cfg.width % 256 or cfg.width < 512 or cfg.width // cfg.heads != 64 or cfg.embed_dim % 256
                               or cfg.context_length > 288):
            arith = "bf16x2"
        self.arith = arith
        self.wsplit = arith == "bf16x2"
        self.res_dtype = res_dtype if arith == "bf16" else torch.float32
        self.tok = _dev(sd["token_embedding.weight"], device)
        self.pos = _dev(sd["positional_embedding"], device)
        self.ln_final = (_dev(sd["ln_final.weight"], device), _dev(sd["ln_final.bias"], device))
        if arith == "f16":
            self.projT = _dev(sd["text_projection"].t(), device, torch.float16)       # [E, D]
            self.blocks = [prep_block_f16(sd, f"transformer.resblocks.{i}.", device) for i in range(cfg.layers)]
        elif self.wsplit:
            hi, lo = _hi_lo(sd["text_projection"].t(), device)
            self.projT = torch.cat([hi, lo], dim=1).contiguous()                      # [E, 2D]
            self.blocks = [prep_block_wsplit(sd, f"transformer.resblocks.{i}.", device) for i in range(cfg.layers)]
        else:
            self.projT = _dev(sd["text_projection"].t(), device, torch.bfloat16)
            self.blocks = [_prep_block_plain(sd, f"transformer.resblocks.{i}.", device) for i in range(cfg.layers)]
        self._ws = {}

    def _encode_f16(self, text: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        B, L = text.shape
        D, H = cfg.width, cfg.heads
        key = ("f16", B, L)
        if key not in self._ws:
            self._ws[key] = _WorkspaceF16(B, L, D, H, self.device)
        ws = self._ws[key]
        rows = ws.rows
        ops.text_embed(text, self.tok, self.pos, ws.x[:rows])
        eot = text.argmax(dim=-1).contiguous()            # index-exact EOT position (model.py:539)
        qs = (D // H) ** -0.5 * ops.LOG2E
        for w in self.blocks:
            ops.layernorm(ws.x, w["ln1_w"], w["ln1_b"], ws.h, rows, D)
            ops.gemm_f16(ws.h, w["in_w"], w["in_b"], out=ws.qkv)
            ops.attn_fwd(ws.q, ws.k, ws.v, ws.a, causal=True, qscale=qs)
            ops.gemm_f16(ws.a, w["out_w"], w["out_b"], out=ws.x, res=ws.x, epi=ops.EPI_RES_F32)
            ops.layernorm(ws.x, w["ln2_w"], w["ln2_b"], ws.h, rows, D)
            ops.gemm_f16(ws.h, w["fc_w"], w["fc_b"], out=ws.hid, act=ops.ACT_GELU)
            ops.gemm_f16(ws.hid, w["proj_w"], w["proj_b"], out=ws.x, res=ws.x, epi=ops.EPI_RES_F32)
        ops.layernorm(ws.x, self.ln_final[0], self.ln_final[1], ws.pooled, B, D, x_row_stride=D, row_index=eot, row_mul=L)
        f = torch.zeros(ws.Bp, self.projT.shape[0], device=self.device, dtype=torch.float32)
        ops.gemm_f16(ws.pooled, self.projT, None, out=f, res=f, epi=ops.EPI_RES_F32)
        return f[:B]

    def encode_text(self, text: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        cfg = self.cfg
        B, L = text.shape
        D = cfg.width
        text = text.to(self.device).contiguous()
        if self.arith == "f16":
            f = self._encode_f16(text)
            return ops.l2_normalize(f) if normalize else f.contiguous()
        key = (B, L)
        if key not in self._ws:
            self._ws[key] = _Workspace(B, L, D, cfg.heads, 4 * D, self.device, self.res_dtype)
        ws = self._ws[key]
        ops.text_embed(text, self.tok, self.pos, ws.x)
        eot = text.argmax(dim=-1).contiguous()            # index-exact EOT position (model.py:539)
        if self.wsplit:
            if not hasattr(ws, "h2"):
                ws.h2 = torch.empty(B * L, 2 * D, device=self.device, dtype=torch.bfloat16)
            run_blocks_wsplit(self.blocks, ws, B, L, D, cfg.heads, causal=True, cfg=self.gemm_cfg)
            pooled = torch.empty(B, 2 * D, device=self.device, dtype=torch.bfloat16)
            for half in (pooled[:, :D], pooled[:, D:]):
                ops.layernorm(ws.x, self.ln_final[0], self.ln_final[1], half, B, D, x_row_stride=D, row_index=eot, row_mul=L)
        else:
            run_blocks(self.blocks, ws, B, L, D, cfg.heads, causal=True, cfg=self.gemm_cfg, fold=False)
            pooled = torch.empty(B, D, device=self.device, dtype=torch.bfloat16)
            ops.layernorm(ws.x, self.ln_final[0], self.ln_final[1], pooled, B, D, x_row_stride=D, row_index=eot, row_mul=L)
        f = ops.gemm(pooled, self.projT, None, epi=ops.EPI_F32, cfg=self.gemm_cfg)
        return ops.l2_normalize(f) if normalize else f


# ------------------------------------------------------------------------------------------------
# The "Lens": modality tokenizer + Perceiver resampler in front of the frozen ViT
# ------------------------------------------------------------------------------------------------
@dataclass
class LensCfg:
    """Mirror of the exp_args fields VisionTransformer.forward reads (module_cfg.py:37-92)."""
    modality: str = "depth"            # depth | audio | pc | image
    perceiver_identity: bool = True    # perceiver.py:370-371
    depth: int = 2
    self_per_cross: int = 3
    num_latents: int = 256
    latent_dim: int = 1024
    input_chan: int = 1024
    cross_heads: int = 1
    cross_dim_head: int = 64
    latent_heads: int = 16
    latent_dim_head: int = 64
    audio_fstride: int = 10
    audio_tstride: int = 10
    audio_mel_bins: int = 128
    audio_target_length: int = 512
    pc_num_group: int = 512
    pc_group_size: int = 32
    pc_encoder_dims: int = 256
    pc_trans_dim: int = 384
    pc_tokenizer: str = "pointbert"        # "pointbert" (FPS + kNN mini-PointNet) | "pnsa" (FPS + ball query set abstraction)
    pc_radius: float = 0.2                 # ball-query radius of the pnsa tokenizer
    pc_in_dim: int = 3                     # point feature channels of the pnsa tokenizer (its convs see 3 + pc_in_dim)
    use_orig_pos: bool = True
    disable_adapter_pos: bool = False
    eeg_chans: int = 128               # modal_eeg/models/EEG_tokenizer.py (PatchEmbed1D)
    eeg_time_len: int = 512
    eeg_window_size: int = 1
    eeg_stride: int = 1
    weight_tie_layers: bool = False    # perceiver.py:249-254: layers >= 1 share one set of modules


def _interleave_geglu(w: torch.Tensor, b: torch.Tensor):
    """Linear(D, 8D) rows [a(0..4D) ; gate(4D..8D)] -> interleaved (a_j, gate_j) so the GEGLU epilogue
    finds both halves of a pair in one lane (perceiver.py:85-89 `x, gates = x.chunk(2, dim=-1)`)."""
    half = w.shape[0] // 2
    wi = torch.stack([w[:half], w[half:]], dim=1).reshape(w.shape[0], w.shape[1])
    bi = torch.stack([b[:half], b[half:]], dim=1).reshape(-1)
    return wi, bi


def prep_lens_attn(sd, p, device, packed_self: bool):
    bf = torch.bfloat16
    d = {"to_out_w": _dev(sd[p + "to_out.weight"], device, bf), "to_out_b": _dev(sd[p + "to_out.bias"], device)}
    if packed_self:
        d["qkv_w"] = _dev(torch.cat([sd[p + "to_q.weight"], sd[p + "to_kv.weight"]], 0), device, bf)
    else:
        d["q_w"] = _dev(sd[p + "to_q.weight"], device, bf)
        d["kv_w"] = _dev(sd[p + "to_kv.weight"], device, bf)
    return d


def prep_lens_ff(sd, p, device):
    w0, b0 = _interleave_geglu(sd[p + "net.0.weight"].detach().float(), sd[p + "net.0.bias"].detach().float())
    return {"w0": _dev(w0, device, torch.bfloat16), "b0": _dev(b0, device),
            "w2": _dev(sd[p + "net.2.weight"], device, torch.bfloat16), "b2": _dev(sd[p + "net.2.bias"], device)}


class PerceiverEngine:
    """Perceiver.forward(return_embeddings=True) (open_clip/perceiver.py:289-328), fourier_encode_data=False."""

    def __init__(self, sd, prefix: str, cfg: LensCfg, device, gemm_cfg=-1):
        self.cfg, self.device, self.gemm_cfg = cfg, torch.device(device), gemm_cfg
        self.latents = _dev(sd[prefix + "latents"], device)
        ln = lambda q: (_dev(sd[q + ".weight"], device), _dev(sd[q + ".bias"], device))
        self.layers = []
        for i in range(cfg.depth):
            if cfg.weight_tie_layers and i >= 2:
                # perceiver.py:249-254 (`cache_fn`): layers 1 .. depth-1 ARE one set of modules; the state_dict repeats their
                # tensors under every layer index.  Sharing the objects here makes an update of layer 1 an update of all.
                self.layers.append(self.layers[1])
                continue
            q = f"{prefix}layers.{i}."
            lay = {"x_norm": ln(q + "0.norm"), "x_norm_ctx": ln(q + "0.norm_context"),
                   "x_attn": prep_lens_attn(sd, q + "0.fn.", device, False),
                   "x_ff_norm": ln(q + "1.norm"), "x_ff": prep_lens_ff(sd, q + "1.fn.", device), "selfs": []}
            for j in range(cfg.self_per_cross):
                r = f"{q}2.{j}."
                lay["selfs"].append({"norm": ln(r + "0.norm"), "attn": prep_lens_attn(sd, r + "0.fn.", device, True),
                                     "ff_norm": ln(r + "1.norm"), "ff": prep_lens_ff(sd, r + "1.fn.", device)})
            self.layers.append(lay)
        self._ws = {}

    def update_params(self, sd, prefix: str):
        """All Perceiver operands re-derived from `sd`, in place (the Perceiver is trainable as a whole in every recipe)."""
        fresh = PerceiverEngine(sd, prefix, self.cfg, self.device, self.gemm_cfg)
        self.latents.copy_(fresh.latents)
        copy_tree(self.layers, fresh.layers)

    def _workspace(self, B, Tc):
        key = (B, Tc)
        if key in self._ws:
            return self._ws[key]
        c, dev, bf = self.cfg, self.device, torch.bfloat16
        n, D = c.num_latents, c.latent_dim
        f = lambda *s, dt=bf: torch.empty(*s, device=dev, dtype=dt)
        ws = {
            "x": f(B * n, D, dt=torch.float32), "h": f(B * n, D), "ctx": f(B * Tc, c.input_chan),
            "xq2": f(B * n, c.cross_heads * c.cross_dim_head), "xkv2": f(B * Tc, 2 * c.cross_heads * c.cross_dim_head),
            "xa": f(B * n, c.cross_heads * c.cross_dim_head),
            "sqkv2": f(B * n, 3 * c.latent_heads * c.latent_dim_head),
            "sa": f(B * n, c.latent_heads * c.latent_dim_head), "hid": f(B * n, 4 * D),
        }
        xi, si = c.cross_heads * c.cross_dim_head, c.latent_heads * c.latent_dim_head
        ws["xq"] = ops.heads_view(ws["xq2"], B, n, c.cross_heads, c.cross_dim_head)
        ws["xk"] = ops.heads_view(ws["xkv2"], B, Tc, c.cross_heads, c.cross_dim_head)
        ws["xv"] = ops.heads_view(ws["xkv2"], B, Tc, c.cross_heads, c.cross_dim_head, xi)
        for i, nm in enumerate(("sq", "sk", "sv")):
            ws[nm] = ops.heads_view(ws["sqkv2"], B, n, c.latent_heads, c.latent_dim_head, i * si)
        self._ws[key] = ws
        return ws

    def _ff(self, ws, norm, ff, rows, D):
        ops.layernorm(ws["x"], norm[0], norm[1], ws["h"], rows, D)
        ops.gemm(ws["h"], ff["w0"], ff["b0"], out=ws["hid"], epi=ops.EPI_GEGLU, cfg=self.gemm_cfg)
        ops.gemm(ws["hid"], ff["w2"], ff["b2"], out=ws["x"], res=ws["x"], epi=ops.EPI_RES_F32, cfg=self.gemm_cfg)

    def forward(self, data: torch.Tensor, B: int) -> torch.Tensor:
        """data [B*Tc, C] (bf16|f32) -> latents [B*n, D] f32 (view of an internal workspace)."""
        c = self.cfg
        Tc, n, D = data.shape[0] // B, c.num_latents, c.latent_dim
        ws = self._workspace(B, Tc)
        ws["x"].view(B, n, D).copy_(self.latents)          # repeat(latents, 'n d -> b n d')
        rows = B * n
        for lay in self.layers:
            a = lay["x_attn"]
            ops.layernorm(ws["x"], lay["x_norm"][0], lay["x_norm"][1], ws["h"], rows, D)
            ops.layernorm(data, lay["x_norm_ctx"][0], lay["x_norm_ctx"][1], ws["ctx"], B * Tc, c.input_chan)
            ops.gemm(ws["h"], a["q_w"], None, out=ws["xq2"], epi=ops.EPI_BF16, cfg=self.gemm_cfg)
            ops.gemm(ws["ctx"], a["kv_w"], None, out=ws["xkv2"], epi=ops.EPI_BF16, cfg=self.gemm_cfg)
            ops.attn_fwd(ws["xq"], ws["xk"], ws["xv"], ws["xa"], qscale=c.cross_dim_head ** -0.5 * ops.LOG2E)
            ops.gemm(ws["xa"], a["to_out_w"], a["to_out_b"], out=ws["x"], res=ws["x"], epi=ops.EPI_RES_F32, cfg=self.gemm_cfg)
            self._ff(ws, lay["x_ff_norm"], lay["x_ff"], rows, D)
            for sl in lay["selfs"]:
                a = sl["attn"]
                ops.layernorm(ws["x"], sl["norm"][0], sl["norm"][1], ws["h"], rows, D)
                ops.gemm(ws["h"], a["qkv_w"], None, out=ws["sqkv2"], epi=ops.EPI_BF16, cfg=self.gemm_cfg)
                ops.attn_fwd(ws["sq"], ws["sk"], ws["sv"], ws["sa"], qscale=c.latent_dim_head ** -0.5 * ops.LOG2E)
                ops.gemm(ws["sa"], a["to_out_w"], a["to_out_b"], out=ws["x"], res=ws["x"], epi=ops.EPI_RES_F32, cfg=self.gemm_cfg)
                self._ff(ws, sl["ff_norm"], sl["ff"], rows, D)
        return ws["x"]


class LensEngine:
    """`visual.` tower of TriCLIP for a non-image modality: visual_adapter -> (+pos) -> Perceiver -> ViT trunk
    (VisionTransformer.forward, open_clip/transformer.py:723-792)."""

    def __init__(self, sd, prefix: str, tower: TowerCfg, lens: LensCfg, device, res_dtype=torch.float32, gemm_cfg=-1):
        self.tower, self.lens, self.device, self.gemm_cfg = tower, lens, torch.device(device), gemm_cfg
        self.vit = VitEngine(sd, prefix, tower, device, res_dtype=res_dtype, gemm_cfg=gemm_cfg)
        a = prefix + "visual_adapter."
        self.prefix_adapter = a
        self.adapter_pos = None
        if lens.modality in ("depth", "audio"):
            self.conv_w = conv_weight_as_gemm(sd[a + "conv1.weight"], device)
            pos = sd[a + "pos_emb"].detach().float()
            self.adapter_pos = _dev(pos * (0.0 if lens.disable_adapter_pos else 1.0), device)      # (the product is a new tensor)
        elif lens.modality == "eeg":
            # PatchEmbed1D: Conv1d(chans -> width, kernel = window, stride, bias) over time = a conv over [N, C, 1, T]
            self.conv_w = conv_weight_as_gemm(sd[a + "proj.weight"].unsqueeze(2), device)
            self.conv_b = _dev(sd[a + "proj.bias"], device)
            self.adapter_pos = _dev(sd[a + "pos_emb"].detach().float() * (0.0 if lens.disable_adapter_pos else 1.0), device)
        elif lens.modality == "pc":
            if lens.pc_tokenizer == "pnsa":       # inference = the trainer class with the running BatchNorm statistics
                from .points import PNSATokenizerTrainer
                self.points = PNSATokenizerTrainer(sd, a, lens, device, gemm_cfg=gemm_cfg, bn_training=False)
            else:
                from .points import PointTokenizerEngine
                self.points = PointTokenizerEngine(sd, a, lens, device, gemm_cfg=gemm_cfg)
        else:
            raise NotImplementedError(lens.modality)
        self.perceiver = None if lens.perceiver_identity else PerceiverEngine(sd, prefix + "perceiver.", lens, device, gemm_cfg)

    def update_params(self, sd, prefix: str, names):
        """In-place refresh after the parameters `names` (relative to `prefix`) changed, e.g. by an optimizer step."""
        L, a = self.lens, prefix + "visual_adapter."
        self.vit.update_params(sd, [n for n in names if not n.startswith(("visual_adapter.", "perceiver."))])
        if any(n.startswith("visual_adapter.") for n in names):
            scale = 0.0 if L.disable_adapter_pos else 1.0
            if L.modality in ("depth", "audio"):
                self.conv_w.copy_(conv_weight_as_gemm(sd[a + "conv1.weight"], self.device))
                self.adapter_pos.copy_(sd[a + "pos_emb"].detach().float() * scale)
            elif L.modality == "eeg":
                self.conv_w.copy_(conv_weight_as_gemm(sd[a + "proj.weight"].unsqueeze(2), self.device))
                self.conv_b.copy_(sd[a + "proj.bias"])
                self.adapter_pos.copy_(sd[a + "pos_emb"].detach().float() * scale)
            elif L.modality == "pc":
                if L.pc_tokenizer == "pnsa":
                    self.points.load_params(sd)
                else:
                    # the inference tokenizer (BatchNorm folded into the convolutions, a host round trip) is rebuilt at its next
                    # USE: while training every optimizer step lands here and the trainer runs its own tokenizer copy
                    self._points_pending = {k: v for k, v in sd.items() if k.startswith(a)}
        if self.perceiver is not None and any(n.startswith("perceiver.") for n in names):
            self.perceiver.update_params(sd, prefix + "perceiver.")

    def tokens(self, x: torch.Tensor):
        """-> (tokens [B*T, C] bf16, pos table [T, C] f32 or per-sample pos [B*T, C], B)."""
        L = self.lens
        p = self.tower.patch
        if L.modality == "depth":       # DepthTokenizer.py:35-60
            cols, gh, gw = ops.im2col(x.contiguous().float(), p, p, p, p, self.conv_w.shape[1])
            return ops.gemm(cols, self.conv_w, None, epi=ops.EPI_BF16, cfg=self.gemm_cfg), self.adapter_pos
        if L.modality == "audio":       # AST_tokenizer.py:44-57: [N,T,F] -> conv over [N,1,F,T]
            cols, gh, gw = ops.im2col(x.contiguous().float().unsqueeze(1), p, p, L.audio_fstride, L.audio_tstride,
                                      self.conv_w.shape[1], transpose_hw=True)
            return ops.gemm(cols, self.conv_w, None, epi=ops.EPI_BF16, cfg=self.gemm_cfg), self.adapter_pos
        if L.modality == "eeg":         # EEG_tokenizer.py:35-42: [N, chans, time] -> tokens [N*T', width]
            cols = self.eeg_cols(x)
            return ops.gemm(cols, self.conv_w, self.conv_b, epi=ops.EPI_BF16, cfg=self.gemm_cfg), self.adapter_pos
        raise NotImplementedError(L.modality)

    def eeg_cols(self, x: torch.Tensor) -> torch.Tensor:
        """windows of the time axis as GEMM rows: [N*T', pad64(chans*window)], column order (chan, tap) = Conv1d's weight"""
        L = self.lens
        cols, gh, gw = ops.im2col(x.contiguous().float().unsqueeze(2), 1, L.eeg_window_size, 1, L.eeg_stride, self.conv_w.shape[1])
        return cols

    def encode(self, x: torch.Tensor, normalize: bool = False, **kw) -> torch.Tensor:
        B = x.shape[0]
        if self.lens.modality == "pc":
            pend = getattr(self, "_points_pending", None)
            if pend is not None:
                from .points import PointTokenizerEngine
                self.points = PointTokenizerEngine(pend, self.prefix_adapter, self.lens, self.device, gemm_cfg=self.gemm_cfg)
                self._points_pending = None
            tok = self.points.forward(x, **kw)                 # already x + pos, [B*G, C] bf16
            lat = self.perceiver.forward(tok, B)
            f = self.vit.trunk(lat, B, use_orig_pos=self.lens.use_orig_pos)
        else:
            tok, pos = self.tokens(x)
            if self.perceiver is None:
                f = self.vit.trunk(tok, B, pos2=pos, use_orig_pos=self.lens.use_orig_pos)
            else:
                T = tok.shape[0] // B
                xin = torch.empty_like(tok)
                ops.add_rows(tok, pos, xin, tok.shape[0], T, tok.shape[1])
                lat = self.perceiver.forward(xin, B)
                f = self.vit.trunk(lat, B, use_orig_pos=self.lens.use_orig_pos)
        return ops.l2_normalize(f) if normalize else f
