"""True fp32 INFERENCE executors: what `precision="fp32"` of the reference's factory means (open_clip/factory.py:260-295,
training/precision.py:5-12 - no autocast, fp32 parameters, fp32 nn.Linear / attention).  Rounds 1-4 ran bf16 operands under
that precision and warned; these classes run every matrix product on gfx950's fp32-input MFMA (`vl_gemm_f32`, exact fmaf
chains, 157 TFLOP/s peak = 1/16 of the bf16 rate), attention in fp32 on the VALU (`vl_attn_fwd_f32`), LayerNorm / token
assembly / embedding on the existing f32 kernels.  Forward only - a tower whose parameters require grad under an enabled
autograd keeps the bf16-operand trainers with fp32 residual and gradient streams (open_clip/model.py says so in
`precision_effective`).  Covered: the image / tactile towers (conv stem + ViT), the depth Lens with an identity Perceiver
(DepthTokenizer -> ViT) and the text tower, with head dim 32 or 64; anything else stays on the 16-bit engines.

Reference ops: VisionTransformer.forward (open_clip/transformer.py:723-792), ResidualAttentionBlock (:254-272),
TriCLIP.encode_text (open_clip/model.py:528-540)."""
from typing import Dict, Optional

import torch

from . import ops
from .engine import TextCfg, TowerCfg, _dev, _pad64


def f32_supported(width: int, heads: int) -> bool:
    return width % heads == 0 and width // heads in (32, 64) and width % 4 == 0


def _block(sd: Dict[str, torch.Tensor], p: str, device) -> Dict[str, torch.Tensor]:
    names = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "in_w": "attn.in_proj_weight", "in_b": "attn.in_proj_bias",
             "out_w": "attn.out_proj.weight", "out_b": "attn.out_proj.bias", "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias",
             "fc_w": "mlp.c_fc.weight", "fc_b": "mlp.c_fc.bias", "proj_w": "mlp.c_proj.weight", "proj_b": "mlp.c_proj.bias"}
    return {k: _dev(sd[p + v], device) for k, v in names.items()}


class _Ws:
    def __init__(self, B, L, D, H, hidden, device):
        z = lambda *s: torch.empty(*s, device=device, dtype=torch.float32)
        self.x, self.h, self.qkv, self.a, self.hid = z(B * L, D), z(B * L, D), z(B * L, 3 * D), z(B * L, D), z(B * L, hidden)
        dh = D // H
        hv = lambda i: ops.heads_view(self.qkv, B, L, H, dh, i * D)
        self.q, self.k, self.v = hv(0), hv(1), hv(2)


def run_blocks_f32(blocks, ws: _Ws, B, L, D, H, causal=False):
    """x <- N pre-LN transformer blocks, fp32 throughout (transformer.py:254-272, 364-371)."""
    scale = (D // H) ** -0.5
    for w in blocks:
        ops.layernorm(ws.x, w["ln1_w"], w["ln1_b"], ws.h, B * L, D)
        ops.gemm_f32(ws.h, w["in_w"], w["in_b"], out=ws.qkv)
        ops.attn_fwd_f32(ws.q, ws.k, ws.v, ws.a, causal=causal, scale=scale)
        ops.gemm_f32(ws.a, w["out_w"], w["out_b"], out=ws.x, res=ws.x)
        ops.layernorm(ws.x, w["ln2_w"], w["ln2_b"], ws.h, B * L, D)
        ops.gemm_f32(ws.h, w["fc_w"], w["fc_b"], out=ws.hid, act=ops.ACT_GELU)
        ops.gemm_f32(ws.hid, w["proj_w"], w["proj_b"], out=ws.x, res=ws.x)


def _conv_as_gemm_f32(w: torch.Tensor, device) -> torch.Tensor:
    """Conv2d weight [O,C,kh,kw] -> f32 [O, Kp] (K zero-padded to a multiple of 64, as the 16-bit stem lays it out)."""
    O, K = w.shape[0], w[0].numel()
    out = torch.zeros(O, _pad64(K), dtype=torch.float32, device=device)
    out[:, :K] = w.detach().reshape(O, K).float().to(device)
    return out


class VitEngineF32:
    """One ViT tower in fp32: `image.` / `visual.` of TriCLIP for the image and tactile modalities, and - with `depth=True` - the
    depth Lens with an identity Perceiver (visual_adapter.conv1 + pos_emb in front of the same trunk)."""

    def __init__(self, sd, prefix: str, cfg: TowerCfg, device, depth: bool = False, use_orig_pos: bool = True,
                 disable_adapter_pos: bool = False):
        if not f32_supported(cfg.width, cfg.heads):
            raise NotImplementedError("fp32 inference: head dim must be 32 or 64")
        self.cfg, self.device, self.depth, self.use_orig_pos = cfg, torch.device(device), depth, use_orig_pos
        self.cls = _dev(sd[prefix + "class_embedding"], device)
        self.pos = _dev(sd[prefix + "positional_embedding"], device)
        ln = lambda n: (_dev(sd[prefix + n + ".weight"], device), _dev(sd[prefix + n + ".bias"], device))
        self.ln_pre, self.ln_post = ln("ln_pre"), ln("ln_post")
        self.projT = _dev(sd[prefix + "proj"].t(), device) if prefix + "proj" in sd else None       # [E, D]
        self.blocks = [_block(sd, f"{prefix}transformer.resblocks.{i}.", device) for i in range(cfg.layers)]
        self.pos2 = None
        if depth:
            a = prefix + "visual_adapter."
            self.conv_w = _conv_as_gemm_f32(sd[a + "conv1.weight"], device)
            self.pos2 = _dev(sd[a + "pos_emb"].detach().float() * (0.0 if disable_adapter_pos else 1.0), device)
        else:
            self.conv_w = _conv_as_gemm_f32(sd[prefix + "conv1.weight"], device)
        self._ws = {}

    def encode(self, x: torch.Tensor, normalize: bool = False, **kw) -> torch.Tensor:
        cfg = self.cfg
        B, D, p = x.shape[0], cfg.width, cfg.patch
        cols, gh, gw = ops.im2col_f32(x.to(self.device).contiguous().float(), p, p, p, p, self.conv_w.shape[1])
        tok = ops.gemm_f32(cols, self.conv_w)
        T = gh * gw
        L = T + 1
        key = (B, L)
        if key not in self._ws:
            self._ws[key] = _Ws(B, L, D, cfg.heads, int(D * cfg.mlp_ratio), self.device)
        ws = self._ws[key]
        pos = self.pos if (self.use_orig_pos or not self.depth) else torch.zeros_like(self.pos)
        ops.assemble_ln_pre(tok, self.cls, pos, self.pos2, self.ln_pre[0], self.ln_pre[1], ws.x, B, T, D)
        run_blocks_f32(self.blocks, ws, B, L, D, cfg.heads)
        pooled = torch.empty(B, D, device=self.device, dtype=torch.float32)
        ops.layernorm(ws.x, self.ln_post[0], self.ln_post[1], pooled, B, D, x_row_stride=L * D)
        f = pooled if self.projT is None else ops.gemm_f32(pooled, self.projT)
        return ops.l2_normalize(f) if normalize else f

    encode_image = encode


class TextEngineF32:
    """TriCLIP.encode_text in fp32 (model.py:528-540): embedding, causal transformer, ln_final at the EOT token, projection."""

    def __init__(self, sd, cfg: TextCfg, device):
        if not f32_supported(cfg.width, cfg.heads):
            raise NotImplementedError("fp32 inference: head dim must be 32 or 64")
        self.cfg, self.device = cfg, torch.device(device)
        self.tok = _dev(sd["token_embedding.weight"], device)
        self.pos = _dev(sd["positional_embedding"], device)
        self.ln_final = (_dev(sd["ln_final.weight"], device), _dev(sd["ln_final.bias"], device))
        self.projT = _dev(sd["text_projection"].t(), device)                                          # [E, D]
        self.blocks = [_block(sd, f"transformer.resblocks.{i}.", device) for i in range(cfg.layers)]
        self.arith = "f32"
        self._ws = {}

    def encode_text(self, text: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        cfg = self.cfg
        B, L = text.shape
        D = cfg.width
        key = (B, L)
        if key not in self._ws:
            self._ws[key] = _Ws(B, L, D, cfg.heads, 4 * D, self.device)
        ws = self._ws[key]
        text = text.to(self.device).contiguous()
        ops.text_embed(text, self.tok, self.pos, ws.x)
        eot = text.argmax(dim=-1).contiguous()            # index-exact EOT position (model.py:539)
        run_blocks_f32(self.blocks, ws, B, L, D, cfg.heads, causal=True)
        pooled = torch.empty(B, D, device=self.device, dtype=torch.float32)
        ops.layernorm(ws.x, self.ln_final[0], self.ln_final[1], pooled, B, D, x_row_stride=D, row_index=eot, row_mul=L)
        f = ops.gemm_f32(pooled, self.projT)
        return ops.l2_normalize(f) if normalize else f
