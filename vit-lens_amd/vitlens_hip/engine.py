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
    return t.detach().to(device=device, dtype=dtype).contiguous()


def prep_block(sd: Dict[str, torch.Tensor], p: str, device) -> Dict[str, torch.Tensor]:
    """Device copies of one ResidualAttentionBlock: GEMM weights bf16, the rest f32."""
    bf = torch.bfloat16
    return {
        "ln1_w": _dev(sd[p + "ln_1.weight"], device), "ln1_b": _dev(sd[p + "ln_1.bias"], device),
        "in_w": _dev(sd[p + "attn.in_proj_weight"], device, bf), "in_b": _dev(sd[p + "attn.in_proj_bias"], device),
        "out_w": _dev(sd[p + "attn.out_proj.weight"], device, bf), "out_b": _dev(sd[p + "attn.out_proj.bias"], device),
        "ln2_w": _dev(sd[p + "ln_2.weight"], device), "ln2_b": _dev(sd[p + "ln_2.bias"], device),
        "fc_w": _dev(sd[p + "mlp.c_fc.weight"], device, bf), "fc_b": _dev(sd[p + "mlp.c_fc.bias"], device),
        "proj_w": _dev(sd[p + "mlp.c_proj.weight"], device, bf), "proj_b": _dev(sd[p + "mlp.c_proj.bias"], device),
    }


def conv_weight_as_gemm(w: torch.Tensor, device) -> torch.Tensor:
    """Conv2d weight [O,C,kh,kw] -> bf16 [O, Kp] (K = C*kh*kw zero-padded to a multiple of 64)."""
    O = w.shape[0]
    K = w[0].numel()
    out = torch.zeros(O, _pad64(K), dtype=torch.float32)
    out[:, :K] = w.detach().reshape(O, K).float().cpu()
    return out.to(device=device, dtype=torch.bfloat16).contiguous()


class _Workspace:
    def __init__(self, B, L, D, H, hidden, device, res_dtype):
        dh = D // H
        bf = torch.bfloat16
        self.Lp = (L + 7) // 8 * 8
        self.x = torch.empty(B * L, D, device=device, dtype=res_dtype)
        self.h = torch.empty(B * L, D, device=device, dtype=bf)
        self.q = torch.empty(B, H, L, dh, device=device, dtype=bf)
        self.k = torch.empty(B, H, L, dh, device=device, dtype=bf)
        self.vt = torch.zeros(B, H, dh, self.Lp, device=device, dtype=bf)
        self.a = torch.empty(B * L, D, device=device, dtype=bf)
        self.hid = torch.empty(B * L, hidden, device=device, dtype=bf)


def run_blocks(blocks, ws: _Workspace, B, L, D, H, causal=False, cfg=-1):
    """x (ws.x, residual stream) <- N pre-LN transformer blocks (transformer.py:254-272, 364-371)."""
    dh = D // H
    res_epi = ops.EPI_RES_F32 if ws.x.dtype == torch.float32 else ops.EPI_RES_BF16
    for w in blocks:
        ops.layernorm(ws.x, w["ln1_w"], w["ln1_b"], ws.h, B * L, D)
        ops.gemm_qkv(ws.h, w["in_w"], w["in_b"], ws.q, ws.k, ws.vt, B, L, H, dh, cfg=cfg)
        ops.attn_fwd(ws.q, ws.k, ws.vt, ws.a, causal=causal)
        ops.gemm(ws.a, w["out_w"], w["out_b"], out=ws.x, res=ws.x, epi=res_epi, cfg=cfg)
        ops.layernorm(ws.x, w["ln2_w"], w["ln2_b"], ws.h, B * L, D)
        ops.gemm(ws.h, w["fc_w"], w["fc_b"], out=ws.hid, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg)
        ops.gemm(ws.hid, w["proj_w"], w["proj_b"], out=ws.x, res=ws.x, epi=res_epi, cfg=cfg)


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
        self.projT = _dev(sd[prefix + "proj"].t(), device, torch.bfloat16)          # [E, D]
        self.blocks = [prep_block(sd, f"{prefix}transformer.resblocks.{i}.", device) for i in range(cfg.layers)]
        self.conv_w = None
        if prefix + "conv1.weight" in sd:
            self.conv_w = conv_weight_as_gemm(sd[prefix + "conv1.weight"], device)
        self._ws = {}

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
        ops.layernorm(ws.x, self.ln_post[0], self.ln_post[1], pooled, B, D, x_row_stride=L * D)
        return ops.gemm(pooled, self.projT, None, epi=ops.EPI_F32, cfg=self.gemm_cfg)

    def encode_image(self, image: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        B = image.shape[0]
        f = self.trunk(self.patch_tokens(image), B)
        return ops.l2_normalize(f) if normalize else f


class TextEngine:
    """TriCLIP.encode_text (model.py:528-540): embedding + causal transformer + ln_final + EOT + proj."""

    def __init__(self, sd, cfg: TextCfg, device, res_dtype=torch.float32, gemm_cfg: int = -1):
        self.cfg, self.device, self.res_dtype, self.gemm_cfg = cfg, torch.device(device), res_dtype, gemm_cfg
        self.tok = _dev(sd["token_embedding.weight"], device)
        self.pos = _dev(sd["positional_embedding"], device)
        self.ln_final = (_dev(sd["ln_final.weight"], device), _dev(sd["ln_final.bias"], device))
        self.projT = _dev(sd["text_projection"].t(), device, torch.bfloat16)
        self.blocks = [prep_block(sd, f"transformer.resblocks.{i}.", device) for i in range(cfg.layers)]
        self._ws = {}

    def encode_text(self, text: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        cfg = self.cfg
        B, L = text.shape
        D = cfg.width
        key = (B, L)
        if key not in self._ws:
            self._ws[key] = _Workspace(B, L, D, cfg.heads, 4 * D, self.device, self.res_dtype)
        ws = self._ws[key]
        text = text.to(self.device).contiguous()
        ops.text_embed(text, self.tok, self.pos, ws.x)
        run_blocks(self.blocks, ws, B, L, D, cfg.heads, causal=True, cfg=self.gemm_cfg)
        eot = text.argmax(dim=-1).contiguous()            # index-exact EOT position (model.py:539)
        pooled = torch.empty(B, D, device=self.device, dtype=torch.bfloat16)
        ops.layernorm(ws.x, self.ln_final[0], self.ln_final[1], pooled, B, D, x_row_stride=D, row_index=eot, row_mul=L)
        f = ops.gemm(pooled, self.projT, None, epi=ops.EPI_F32, cfg=self.gemm_cfg)
        return ops.l2_normalize(f) if normalize else f
