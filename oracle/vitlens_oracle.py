"""CPU oracle for the ViT-Lens contrastive hot path.  TEST INFRASTRUCTURE — NOT PRODUCT.

A from-scratch, functional, fp32 PyTorch-CPU restatement of the arithmetic the reference
performs on the path named in BASELINE.json `north_star`.  Every function takes a plain
`state_dict` (the reference's own parameter names) plus inputs; there are
no nn.Modules and nothing is imported from the reference.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

Pinned (see oracle/gen_golden.py + tests/test_oracle_golden.py): the restatement is
checked here, in the build container, against the *imported* reference on seeded tiny
models and on full ViT-L/14; the resulting input/output vectors are committed under
tests/golden/.  The reference ships no tests or golden vectors of its own.

Reference citations are `vitlens/src/open_clip/<file>:<line>`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------- specs
@dataclass
class TowerSpec:
    """Geometry of one ViT tower (model_configs/*.json `vision_cfg`)."""
    width: int = 1024
    layers: int = 24
    heads: int = 16
    mlp_ratio: float = 4.0
    patch: int = 14
    image_size: int = 224
    embed_dim: int = 768


@dataclass
class TextSpec:
    context_length: int = 77
    vocab_size: int = 49408
    width: int = 768
    heads: int = 12
    layers: int = 12
    embed_dim: int = 768


@dataclass
class LensSpec:
    """What sits in front of the frozen ViT for a non-image modality
    (mm_vit_lens/model_cfg.py:80-178, module_cfg.py:37-92)."""
    modality: str = "depth"            # depth | audio | pc | eeg | image
    perceiver_identity: bool = True    # perceiver.py:370-371
    depth: int = 2                     # perceiver_depth
    self_per_cross: int = 3
    num_latents: int = 256
    latent_dim: int = 1024
    input_chan: int = 1024             # context dim
    cross_heads: int = 1
    cross_dim_head: int = 64
    latent_heads: int = 16
    latent_dim_head: int = 64
    # audio (AST_tokenizer.py)
    audio_fstride: int = 10
    audio_tstride: int = 10
    audio_mel_bins: int = 128
    audio_target_length: int = 512
    # EEG (modal_eeg/models/EEG_tokenizer.py: PatchEmbed1D)
    eeg_chans: int = 128
    eeg_time_len: int = 512
    eeg_window_size: int = 1
    eeg_stride: int = 1
    # point cloud (pointbert)
    pc_num_group: int = 512
    pc_group_size: int = 32
    pc_encoder_dims: int = 256
    pc_trans_dim: int = 384
    use_orig_pos: bool = True          # transformer.py:545-551
    disable_adapter_pos: bool = False  # transformer.py:738-745
    weight_tie_layers: bool = False    # perceiver.py:249-254 (see tie_perceiver_layers; the forward needs nothing special)


# ----------------------------------------------------------------------------- primitives
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """transformer.py:28-34 / nn.LayerNorm: biased variance over the last dim."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (transformer.py:230)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def sdpa(q: Tensor, k: Tensor, v: Tensor, scale: float, mask: Optional[Tensor] = None) -> Tensor:
    """softmax(q kᵀ·scale + mask) v on [..., L, d] operands."""
    s = (q @ k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask
    s = s - s.amax(dim=-1, keepdim=True)
    p = torch.exp(s)
    p = p / p.sum(dim=-1, keepdim=True)
    return p @ v


def causal_mask(n: int) -> Tensor:
    """TextTransformer.build_attention_mask, transformer.py:870-876."""
    m = torch.full((n, n), float("-inf"))
    return torch.triu(m, diagonal=1)


def mha_packed(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
               heads: int, mask: Optional[Tensor] = None) -> Tensor:
    """nn.MultiheadAttention self-attention with packed in_proj (transformer.py:215,241-252).
    x: [N, L, D] batch-first (the reference runs seq-first; the math is per-sample)."""
    N, L, D = x.shape
    d = D // heads
    qkv = linear(x, in_w, in_b)                      # [N, L, 3D]
    q, k, v = qkv.split(D, dim=-1)
    def heads_first(t):
        return t.reshape(N, L, heads, d).permute(0, 2, 1, 3)
    o = sdpa(heads_first(q), heads_first(k), heads_first(v), 1.0 / math.sqrt(d), mask)
    o = o.permute(0, 2, 1, 3).reshape(N, L, D)
    return linear(o, out_w, out_b)


def resblock(sd: SD, p: str, x: Tensor, heads: int, mask: Optional[Tensor] = None) -> Tensor:
    """ResidualAttentionBlock.forward, transformer.py:254-272 (ls_1/ls_2 = Identity)."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    x = x + mha_packed(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                       sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads, mask)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    h = gelu_erf(linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
    return x + linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])


def transformer(sd: SD, p: str, x: Tensor, layers: int, heads: int,
                mask: Optional[Tensor] = None) -> Tensor:
    """Transformer.forward, transformer.py:364-371."""
    for i in range(layers):
        x = resblock(sd, f"{p}resblocks.{i}.", x, heads, mask)
    return x


# ----------------------------------------------------------------------------- tokenizers
def conv_patchify(x: Tensor, w: Tensor, stride: Tuple[int, int]) -> Tensor:
    """Conv2d(bias=False) + reshape(N,C,-1).permute(0,2,1) written as unfold + GEMM.
    Token order is row-major over the conv output grid (transformer.py:674-676,
    DepthTokenizer.py:52-54, AST_tokenizer.py:47-50)."""
    kh, kw = w.shape[2], w.shape[3]
    cols = F.unfold(x, kernel_size=(kh, kw), stride=stride)       # [N, C*kh*kw, T]
    return cols.transpose(1, 2) @ w.reshape(w.shape[0], -1).t()   # [N, T, width]


def image_tokens(sd: SD, p: str, image: Tensor, spec: TowerSpec) -> Tensor:
    """VisionTransformer.img_adapter_forawrd, transformer.py:659-677 (no patchnorm)."""
    return conv_patchify(image, sd[p + "conv1.weight"], (spec.patch, spec.patch))


def depth_tokens(sd: SD, p: str, depth: Tensor, spec: TowerSpec) -> Tuple[Tensor, Tensor]:
    """DepthTokenizer.forward, modal_depth/models/DepthTokenizer.py:35-60."""
    x = conv_patchify(depth, sd[p + "visual_adapter.conv1.weight"], (spec.patch, spec.patch))
    return x, sd[p + "visual_adapter.pos_emb"]


def audio_tokens(sd: SD, p: str, spec_in: Tensor, lens: LensSpec) -> Tuple[Tensor, Tensor]:
    """AST_tokenizer.forward, modal_audio/models/AST_tokenizer.py:44-57:
    [N,T,F] -> unsqueeze(1).transpose(2,3) = [N,1,F,T] -> conv k14 stride (f,t)."""
    x = spec_in.unsqueeze(1).transpose(2, 3)
    x = conv_patchify(x, sd[p + "visual_adapter.conv1.weight"],
                      (lens.audio_fstride, lens.audio_tstride))
    return x, sd[p + "visual_adapter.pos_emb"]


def fps_indices(xyz: Tensor, npoint: int, start: Tensor) -> Tensor:
    """misc.fps, modal_3d/models/pointbert/misc.py:48-68, with the random start index
    (`torch.randint`, :60) passed in explicitly.  Returns int64 [B, npoint]."""
    B, N, _ = xyz.shape
    idx = torch.zeros(B, npoint, dtype=torch.long)
    dist = torch.full((B, N), 1e10, dtype=xyz.dtype)
    far = start.clone().long()
    ar = torch.arange(B)
    for i in range(npoint):
        idx[:, i] = far
        c = xyz[ar, far, :].view(B, 1, 3)
        d = ((xyz - c) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        far = dist.argmax(dim=-1)
    return idx


def knn_indices(xyz: Tensor, centers: Tensor, k: int) -> Tensor:
    """knn_point / square_distance, dvae.py:107-140 (same expanded form
    -2ab + |a|² + |b|²; neighbour ORDER is unspecified: sorted=False)."""
    d = -2.0 * centers @ xyz.transpose(1, 2)
    d = d + (centers ** 2).sum(-1)[:, :, None]
    d = d + (xyz ** 2).sum(-1)[:, None, :]
    return d.topk(k, dim=-1, largest=False, sorted=False).indices


def batch_norm_1d(x: Tensor, sd: SD, p: str, training: bool, eps: float = 1e-5,
                  running_out: Optional[Dict[str, Tensor]] = None, momentum: float = 0.1) -> Tensor:
    """nn.BatchNorm1d on [B, C, n] (dvae.py:183-194): batch stats (biased var) in train
    mode, running stats in eval mode.  In train mode the module also updates its running
    statistics (momentum 0.1, UNBIASED variance); they are returned through `running_out`."""
    if training:
        mu = x.mean(dim=(0, 2), keepdim=True)
        var = ((x - mu) ** 2).mean(dim=(0, 2), keepdim=True)
        if running_out is not None:
            n = x.shape[0] * x.shape[2]
            running_out[p + "running_mean"] = ((1 - momentum) * sd[p + "running_mean"] + momentum * mu.detach().flatten())
            running_out[p + "running_var"] = ((1 - momentum) * sd[p + "running_var"]
                                              + momentum * var.detach().flatten() * n / max(n - 1, 1))
    else:
        mu = sd[p + "running_mean"].view(1, -1, 1)
        var = sd[p + "running_var"].view(1, -1, 1)
    return (x - mu) / torch.sqrt(var + eps) * sd[p + "weight"].view(1, -1, 1) + sd[p + "bias"].view(1, -1, 1)


def point_tokens(sd: SD, p: str, pts: Tensor, lens: LensSpec, fps_start: Tensor,
                 training: bool = False, running_out: Optional[Dict[str, Tensor]] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """PointTokenizer.forward, point_encoder.py:350-362 (Group dvae.py:150-176,
    Encoder dvae.py:196-212).  Returns (x [B,G,trans], pos [B,G,trans], fps idx, knn idx)."""
    a = p + "visual_adapter."
    B, N, _ = pts.shape
    G, M = lens.pc_num_group, lens.pc_group_size
    cidx = fps_indices(pts, G, fps_start)
    center = torch.gather(pts, 1, cidx[:, :, None].expand(B, G, 3))
    nidx = knn_indices(pts, center, M)                                  # [B,G,M]
    nb = torch.gather(pts[:, None].expand(B, G, N, 3), 2, nidx[..., None].expand(B, G, M, 3))
    nb = nb - center[:, :, None, :]
    g = nb.reshape(B * G, M, 3).transpose(1, 2)                          # [BG,3,M]
    def conv1(x, name):
        return torch.einsum("oc,bcn->bon", sd[a + name + ".weight"][:, :, 0], x) + sd[a + name + ".bias"].view(1, -1, 1)
    f = conv1(g, "encoder.first_conv.0")
    f = torch.relu(batch_norm_1d(f, sd, a + "encoder.first_conv.1.", training, running_out=running_out))
    f = conv1(f, "encoder.first_conv.3")                                 # [BG,256,M]
    fg = f.max(dim=2, keepdim=True).values
    f = torch.cat([fg.expand(-1, -1, M), f], dim=1)                      # [BG,512,M]
    f = conv1(f, "encoder.second_conv.0")
    f = torch.relu(batch_norm_1d(f, sd, a + "encoder.second_conv.1.", training, running_out=running_out))
    f = conv1(f, "encoder.second_conv.3")
    tok = f.max(dim=2).values.reshape(B, G, lens.pc_encoder_dims)
    tok = linear(tok, sd[a + "reduce_dim.weight"], sd[a + "reduce_dim.bias"])
    pos = linear(gelu_erf(linear(center, sd[a + "pos_embed.0.weight"], sd[a + "pos_embed.0.bias"])),
                 sd[a + "pos_embed.2.weight"], sd[a + "pos_embed.2.bias"])
    return tok, pos, cidx, nidx


# ----------------------------------------------------------------------------- PointNet set-abstraction tokenizer ("pnsa")
def ball_query_indices(radius: float, nsample: int, xyz: Tensor, new_xyz: Tensor) -> Tensor:
    """query_ball_point, modal_3d/models/pointnet/pointnet_util.py:101-123 (+ square_distance :24-46): for every centre
    the FIRST `nsample` point indices (ascending) with squared distance <= radius^2, the distance taken as
    -2 c.p + |c|^2 + |p|^2 in that order of accumulation; short groups are filled with their first index."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    d = -2 * torch.matmul(new_xyz, xyz.permute(0, 2, 1))
    d = d + torch.sum(new_xyz ** 2, -1).view(B, S, 1)
    d = d + torch.sum(xyz ** 2, -1).view(B, 1, N)
    idx = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat(B, S, 1)
    idx[d > radius ** 2] = N
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[..., :1].repeat(1, 1, nsample)
    mask = idx == N
    idx[mask] = first[mask]
    return idx


def batch_norm_rows(x: Tensor, sd: SD, p: str, training: bool, eps: float = 1e-5) -> Tensor:
    """nn.BatchNorm2d of [B, C, n, S] expressed on [rows, C] (every position of every sample is one row)."""
    if training:
        mu = x.mean(0, keepdim=True)
        var = ((x - mu) ** 2).mean(0, keepdim=True)
    else:
        mu, var = sd[p + "running_mean"][None], sd[p + "running_var"][None]
    return (x - mu) / torch.sqrt(var + eps) * sd[p + "weight"][None] + sd[p + "bias"][None]


def pnsa_tokens(sd: SD, a: str, features: Tensor, xyz: Tensor, num_group: int, radius: float, group_size: int,
                fps_start: Tensor, training: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """PointNSATokenizer.forward, pointnet_util.py:345-368: PointNetSetAbstraction (:184-227 - sample_and_group :126-161:
    FPS centres with the random start passed in, ball query, centre-subtracted xyz ++ point features; three 1x1
    Conv2d + BatchNorm2d + ReLU; max over the group) then `lift`: Conv1d over [centre xyz ++ group feature], LayerNorm.
    features [B,N,D], xyz [B,N,3] -> (tokens [B,S,trans], fps idx [B,S], ball idx [B,S,nsample])."""
    B, N, _ = xyz.shape
    S, ns = num_group, group_size
    cidx = fps_indices(xyz, S, fps_start)
    new_xyz = torch.gather(xyz, 1, cidx[:, :, None].expand(B, S, 3))
    bidx = ball_query_indices(radius, ns, xyz, new_xyz)
    ar = torch.arange(B).view(B, 1, 1)
    g_xyz = xyz[ar, bidx] - new_xyz.view(B, S, 1, 3)
    x = torch.cat([g_xyz, features[ar, bidx]], dim=-1)                    # [B,S,ns,3+D]
    x = x.reshape(B * S * ns, -1)
    for i in range(3):
        w = sd[f"{a}sa.mlp_convs.{i}.weight"][:, :, 0, 0]
        x = torch.relu(batch_norm_rows(x @ w.t() + sd[f"{a}sa.mlp_convs.{i}.bias"], sd, f"{a}sa.mlp_bns.{i}.", training))
    feat = x.view(B, S, ns, -1).max(dim=2).values                         # [B,S,C']
    y = torch.cat([new_xyz, feat], dim=-1) @ sd[a + "lift.0.weight"][:, :, 0].t() + sd[a + "lift.0.bias"]
    return layer_norm(y, sd[a + "lift.2.weight"], sd[a + "lift.2.bias"]), cidx, bidx


# ----------------------------------------------------------------------------- Perceiver ("Lens")
def lens_attention(sd: SD, p: str, x: Tensor, ctx: Tensor, heads: int, dim_head: int) -> Tensor:
    """perceiver.Attention.forward, perceiver.py:121-154 (no mask, no xformers):
    to_q / to_kv have no bias, to_out has one; scale = dim_head**-0.5."""
    B, n, _ = x.shape
    m = ctx.shape[1]
    q = linear(x, sd[p + "to_q.weight"])
    k, v = linear(ctx, sd[p + "to_kv.weight"]).chunk(2, dim=-1)
    def hf(t, L):
        return t.reshape(B, L, heads, dim_head).permute(0, 2, 1, 3)
    o = sdpa(hf(q, n), hf(k, m), hf(v, m), dim_head ** -0.5)
    o = o.permute(0, 2, 1, 3).reshape(B, n, heads * dim_head)
    return linear(o, sd[p + "to_out.weight"], sd[p + "to_out.bias"])


def lens_ff(sd: SD, p: str, x: Tensor) -> Tensor:
    """FeedForward with GEGLU, perceiver.py:85-102: Linear(D,8D) -> a*gelu(gates) -> Linear(4D,D)."""
    h = linear(x, sd[p + "net.0.weight"], sd[p + "net.0.bias"])
    a, gates = h.chunk(2, dim=-1)
    return linear(a * gelu_erf(gates), sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def perceiver(sd: SD, p: str, data: Tensor, lens: LensSpec) -> Tensor:
    """Perceiver.forward(return_embeddings=True), perceiver.py:289-328, with
    fourier_encode_data=False (all ViT-Lens configs)."""
    B = data.shape[0]
    x = sd[p + "latents"].unsqueeze(0).expand(B, -1, -1)
    for i in range(lens.depth):
        q = f"{p}layers.{i}."
        h = layer_norm(x, sd[q + "0.norm.weight"], sd[q + "0.norm.bias"])
        c = layer_norm(data, sd[q + "0.norm_context.weight"], sd[q + "0.norm_context.bias"])
        x = lens_attention(sd, q + "0.fn.", h, c, lens.cross_heads, lens.cross_dim_head) + x
        h = layer_norm(x, sd[q + "1.norm.weight"], sd[q + "1.norm.bias"])
        x = lens_ff(sd, q + "1.fn.", h) + x
        for j in range(lens.self_per_cross):
            r = f"{q}2.{j}."
            h = layer_norm(x, sd[r + "0.norm.weight"], sd[r + "0.norm.bias"])
            x = lens_attention(sd, r + "0.fn.", h, h, lens.latent_heads, lens.latent_dim_head) + x
            h = layer_norm(x, sd[r + "1.norm.weight"], sd[r + "1.norm.bias"])
            x = lens_ff(sd, r + "1.fn.", h) + x
    return x


# ----------------------------------------------------------------------------- towers
def vit_trunk(sd: SD, p: str, tokens: Tensor, spec: TowerSpec, use_orig_pos: bool = True) -> Tensor:
    """VisionTransformer.forward from the cls concat on, transformer.py:756-787:
    [cls; tokens] + positional_embedding -> ln_pre -> blocks -> ln_post(x[:,0]) @ proj."""
    N = tokens.shape[0]
    cls = sd[p + "class_embedding"].view(1, 1, -1).expand(N, 1, -1)
    x = torch.cat([cls, tokens], dim=1)
    if use_orig_pos:
        x = x + sd[p + "positional_embedding"]
    x = layer_norm(x, sd[p + "ln_pre.weight"], sd[p + "ln_pre.bias"])
    x = transformer(sd, p + "transformer.", x, spec.layers, spec.heads)
    pooled = layer_norm(x[:, 0], sd[p + "ln_post.weight"], sd[p + "ln_post.bias"])
    return pooled @ sd[p + "proj"]


def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    """F.normalize(dim=-1), model.py:522."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def encode_image(sd: SD, image: Tensor, spec: TowerSpec, normalize: bool = False,
                 prefix: str = "image.") -> Tensor:
    """TriCLIP.encode_image, model.py:510-522 (4-D input)."""
    f = vit_trunk(sd, prefix, image_tokens(sd, prefix, image, spec), spec)
    return l2_normalize(f) if normalize else f


def eeg_tokens(sd: SD, prefix: str, x: Tensor, lens: LensSpec):
    """PatchEmbed1D.forward (modal_eeg/models/EEG_tokenizer.py:35-42): Conv1d(chans -> width, kernel = window, stride,
    WITH bias) over the time axis of x [N, chans, time], transposed to tokens [N, T', width]; pos_emb [T', width]."""
    a = prefix + "visual_adapter."
    t = torch.nn.functional.conv1d(x, sd[a + "proj.weight"], sd[a + "proj.bias"], stride=lens.eeg_stride)
    return t.transpose(1, 2).contiguous(), sd[a + "pos_emb"]


def tie_perceiver_layers(sd: SD, depth: int, prefix: str = "visual.perceiver.") -> SD:
    """perceiver_weight_tie_layers (perceiver.py:249-254): layers 1 .. depth-1 are the same modules.  Makes the entries of
    layers >= 2 the SAME tensor objects as layer 1's, so autograd on this dict sums their gradients as the reference does."""
    p1 = f"{prefix}layers.1."
    for k in [k for k in sd if k.startswith(p1)]:
        for li in range(2, depth):
            sd[f"{prefix}layers.{li}." + k[len(p1):]] = sd[k]
    return sd


def encode_visual(sd: SD, x: Tensor, spec: TowerSpec, lens: LensSpec, normalize: bool = False,
                  prefix: str = "visual.", fps_start: Optional[Tensor] = None,
                  training: bool = False, running_out: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """TriCLIP.encode_visual, model.py:524-526 -> VisionTransformer.forward 723-792."""
    if lens.modality == "image":
        tok = image_tokens(sd, prefix, x, spec)
    else:
        if lens.modality == "depth":
            t, pos = depth_tokens(sd, prefix, x, spec)
        elif lens.modality == "audio":
            t, pos = audio_tokens(sd, prefix, x, lens)
        elif lens.modality == "pc":
            t, pos, _, _ = point_tokens(sd, prefix, x, lens, fps_start, training, running_out)
        elif lens.modality == "eeg":
            t, pos = eeg_tokens(sd, prefix, x, lens)
        else:
            raise NotImplementedError(lens.modality)
        tok = t + (0 * pos if lens.disable_adapter_pos else pos)
        if not lens.perceiver_identity:
            tok = perceiver(sd, prefix + "perceiver.", tok, lens)
    f = vit_trunk(sd, prefix, tok, spec, lens.use_orig_pos)
    return l2_normalize(f) if normalize else f


def encode_text(sd: SD, text: Tensor, spec: TextSpec, normalize: bool = False) -> Tensor:
    """TriCLIP.encode_text, model.py:528-540 (text tower is flattened into the root)."""
    x = sd["token_embedding.weight"][text] + sd["positional_embedding"]
    x = transformer(sd, "transformer.", x, spec.layers, spec.heads, causal_mask(text.shape[1]))
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    eot = text.argmax(dim=-1)
    x = x[torch.arange(x.shape[0]), eot] @ sd["text_projection"]
    return l2_normalize(x) if normalize else x


# ----------------------------------------------------------------------------- InfoNCE
def cross_entropy_rows(logits: Tensor, labels: Tensor) -> Tensor:
    """F.cross_entropy(mean) restated: mean_i( logsumexp(row_i) - row_i[label_i] )."""
    m = logits.amax(dim=-1, keepdim=True)
    lse = (logits - m).exp().sum(-1).log() + m.squeeze(-1)
    return (lse - logits[torch.arange(logits.shape[0]), labels]).mean()


def clip_loss(x: Tensor, y: Tensor, logit_scale: Tensor) -> Tensor:
    """ClipLoss / ClipLossGeneral forward at world_size 1, loss.py:293-308,372-385."""
    lx = logit_scale * x @ y.t()
    ly = logit_scale * y @ x.t()
    lab = torch.arange(x.shape[0])
    return (cross_entropy_rows(lx, lab) + cross_entropy_rows(ly, lab)) / 2


def tri_clip_loss(image: Tensor, text: Tensor, visual: Tensor, logit_scale: Tensor) -> Tensor:
    """TriClipLoss.forward, loss.py:140-165: (CE(IV)+CE(VI)+CE(TV)+CE(VT)) / 2."""
    lab = torch.arange(image.shape[0])
    def pair(a, b):
        return cross_entropy_rows(logit_scale * a @ b.t(), lab) + cross_entropy_rows(logit_scale * b @ a.t(), lab)
    return (pair(image, visual) + pair(text, visual)) / 2


def gathered_clip_loss(xs: Sequence[Tensor], ys: Sequence[Tensor], logit_scale: Tensor,
                       rank: int, local_loss: bool = False) -> Tensor:
    """What rank `rank` computes in ClipLossGeneral at world_size = len(xs):
    gather_features loss.py:20-78 (rank-major concat; peers carry no grad unless
    gather_with_grad — only VALUES are modelled here) then get_logits loss.py:116-138."""
    W = len(xs)
    b = xs[0].shape[0]
    # loss.py:71-74: the own slice is re-inserted (differentiable) only when NOT local_loss
    own = (lambda r: r == rank and not local_loss)
    allx = torch.cat([xs[r] if own(r) else xs[r].detach() for r in range(W)], 0)
    ally = torch.cat([ys[r] if own(r) else ys[r].detach() for r in range(W)], 0)
    if local_loss:
        lx = logit_scale * xs[rank] @ ally.t()
        ly = logit_scale * ys[rank] @ allx.t()
        lab = torch.arange(b) + b * rank
    else:
        lx = logit_scale * allx @ ally.t()
        ly = lx.t()
        lab = torch.arange(W * b)
    return (cross_entropy_rows(lx, lab) + cross_entropy_rows(ly, lab)) / 2


# ----------------------------------------------------------------------------- seeded init
def init_tower(spec: TowerSpec, gen: torch.Generator, prefix: str, with_conv: bool = True,
               tokens: Optional[int] = None) -> SD:
    """Seeded random weights with the reference's init *distributions*
    (VisionTransformer.__init__ transformer.py:493-543: scale·randn for cls/pos/proj,
    PyTorch Linear defaults elsewhere).  Used where the ≈1.2 GB ViT-L state_dict cannot
    be committed."""
    D, Hd = spec.width, int(spec.width * spec.mlp_ratio)
    T = tokens if tokens is not None else (spec.image_size // spec.patch) ** 2
    s = D ** -0.5
    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=gen) * std
    def uni(*shape, bound):
        return (torch.rand(*shape, generator=gen) * 2 - 1) * bound
    sd: SD = {}
    sd[prefix + "class_embedding"] = rn(D, std=s)
    sd[prefix + "positional_embedding"] = rn(T + 1, D, std=s)
    sd[prefix + "proj"] = rn(D, spec.embed_dim, std=s)
    if with_conv:
        fan = 3 * spec.patch * spec.patch
        sd[prefix + "conv1.weight"] = uni(D, 3, spec.patch, spec.patch, bound=fan ** -0.5)
    for n in ("ln_pre", "ln_post"):
        sd[prefix + n + ".weight"] = 1.0 + rn(D, std=0.02)
        sd[prefix + n + ".bias"] = rn(D, std=0.02)
    for i in range(spec.layers):
        sd.update(init_block(f"{prefix}transformer.resblocks.{i}.", D, Hd, gen))
    return sd


def init_block(p: str, D: int, Hd: int, gen: torch.Generator) -> SD:
    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=gen) * std
    def uni(*shape, bound):
        return (torch.rand(*shape, generator=gen) * 2 - 1) * bound
    sd: SD = {}
    for n in ("ln_1", "ln_2"):
        sd[p + n + ".weight"] = 1.0 + rn(D, std=0.02)
        sd[p + n + ".bias"] = rn(D, std=0.02)
    xav = math.sqrt(6.0 / (D + 3 * D))                       # MHA in_proj: xavier_uniform
    sd[p + "attn.in_proj_weight"] = uni(3 * D, D, bound=xav)
    sd[p + "attn.in_proj_bias"] = rn(3 * D, std=0.02)
    sd[p + "attn.out_proj.weight"] = uni(D, D, bound=D ** -0.5)
    sd[p + "attn.out_proj.bias"] = rn(D, std=0.02)
    sd[p + "mlp.c_fc.weight"] = uni(Hd, D, bound=D ** -0.5)
    sd[p + "mlp.c_fc.bias"] = uni(Hd, bound=D ** -0.5)
    sd[p + "mlp.c_proj.weight"] = uni(D, Hd, bound=Hd ** -0.5)
    sd[p + "mlp.c_proj.bias"] = uni(D, bound=Hd ** -0.5)
    return sd


def init_text(spec: TextSpec, gen: torch.Generator) -> SD:
    """TextTransformer.init_parameters distributions, transformer.py:846-864."""
    D = spec.width
    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=gen) * std
    sd: SD = {
        "token_embedding.weight": rn(spec.vocab_size, D, std=0.02),
        "positional_embedding": rn(spec.context_length, D, std=0.01),
        "text_projection": rn(D, spec.embed_dim, std=D ** -0.5),
        "ln_final.weight": 1.0 + rn(D, std=0.02),
        "ln_final.bias": rn(D, std=0.02),
        "logit_scale": torch.tensor(math.log(1 / 0.07)),
    }
    proj_std = (D ** -0.5) * ((2 * spec.layers) ** -0.5)
    for i in range(spec.layers):
        p = f"transformer.resblocks.{i}."
        blk = init_block(p, D, 4 * D, gen)
        blk[p + "attn.in_proj_weight"] = rn(3 * D, D, std=D ** -0.5)
        blk[p + "attn.out_proj.weight"] = rn(D, D, std=proj_std)
        blk[p + "mlp.c_fc.weight"] = rn(4 * D, D, std=(2 * D) ** -0.5)
        blk[p + "mlp.c_proj.weight"] = rn(D, 4 * D, std=proj_std)
        sd.update(blk)
    return sd


def init_lens(spec: TowerSpec, lens: LensSpec, gen: torch.Generator, prefix: str = "visual.") -> SD:
    """Seeded weights for adapter + Perceiver with the reference's shapes."""
    D = spec.width
    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=gen) * std
    def uni(*shape, bound):
        return (torch.rand(*shape, generator=gen) * 2 - 1) * bound
    sd: SD = {}
    a = prefix + "visual_adapter."
    if lens.modality == "depth":
        g = (spec.image_size // spec.patch) ** 2
        sd[a + "conv1.weight"] = uni(D, 1, spec.patch, spec.patch, bound=1.0 / spec.patch)
        sd[a + "pos_emb"] = rn(g, D, std=D ** -0.5)
    elif lens.modality == "audio":
        fd = (lens.audio_mel_bins - spec.patch) // lens.audio_fstride + 1
        td = (lens.audio_target_length - spec.patch) // lens.audio_tstride + 1
        sd[a + "conv1.weight"] = uni(D, 1, spec.patch, spec.patch, bound=1.0 / spec.patch)
        sd[a + "pos_emb"] = rn(fd * td, D, std=D ** -0.5)
    elif lens.modality == "eeg":
        fan = lens.eeg_chans * lens.eeg_window_size
        sd[a + "proj.weight"] = uni(D, lens.eeg_chans, lens.eeg_window_size, bound=fan ** -0.5)
        sd[a + "proj.bias"] = uni(D, bound=fan ** -0.5)
        sd[a + "pos_emb"] = rn((lens.eeg_time_len - lens.eeg_window_size) // lens.eeg_stride + 1, D, std=D ** -0.5)
    elif lens.modality == "pc":
        E, Tr = lens.pc_encoder_dims, lens.pc_trans_dim
        def lin(name, o, i, conv=False):
            w = uni(o, i, bound=i ** -0.5)
            sd[a + name + ".weight"] = w[:, :, None] if conv else w
            sd[a + name + ".bias"] = uni(o, bound=i ** -0.5)
        def bn(name, c):
            sd[a + name + ".weight"] = 1.0 + rn(c, std=0.05)
            sd[a + name + ".bias"] = rn(c, std=0.05)
            sd[a + name + ".running_mean"] = rn(c, std=0.1)
            sd[a + name + ".running_var"] = 1.0 + 0.2 * torch.rand(c, generator=gen)
        lin("encoder.first_conv.0", 128, 3, True); bn("encoder.first_conv.1", 128)
        lin("encoder.first_conv.3", 256, 128, True)
        lin("encoder.second_conv.0", 512, 512, True); bn("encoder.second_conv.1", 512)
        lin("encoder.second_conv.3", E, 512, True)
        lin("reduce_dim", Tr, E)
        lin("pos_embed.0", 128, 3); lin("pos_embed.2", Tr, 128)
    if not lens.perceiver_identity:
        p = prefix + "perceiver."
        Ld, C = lens.latent_dim, lens.input_chan
        sd[p + "latents"] = rn(lens.num_latents, Ld)
        def ln(name, d):
            sd[name + ".weight"] = 1.0 + rn(d, std=0.02)
            sd[name + ".bias"] = rn(d, std=0.02)
        def attn(q, qd, cd, heads, dh):
            inner = heads * dh
            sd[q + "to_q.weight"] = uni(inner, qd, bound=qd ** -0.5)
            sd[q + "to_kv.weight"] = uni(2 * inner, cd, bound=cd ** -0.5)
            sd[q + "to_out.weight"] = uni(qd, inner, bound=inner ** -0.5)
            sd[q + "to_out.bias"] = uni(qd, bound=inner ** -0.5)
        def ff(q, d):
            sd[q + "net.0.weight"] = uni(8 * d, d, bound=d ** -0.5)
            sd[q + "net.0.bias"] = uni(8 * d, bound=d ** -0.5)
            sd[q + "net.2.weight"] = uni(d, 4 * d, bound=(4 * d) ** -0.5)
            sd[q + "net.2.bias"] = uni(d, bound=(4 * d) ** -0.5)
        for i in range(lens.depth):
            q = f"{p}layers.{i}."
            ln(q + "0.norm", Ld); ln(q + "0.norm_context", C)
            attn(q + "0.fn.", Ld, C, lens.cross_heads, lens.cross_dim_head)
            ln(q + "1.norm", Ld); ff(q + "1.fn.", Ld)
            for j in range(lens.self_per_cross):
                r = f"{q}2.{j}."
                ln(r + "0.norm", Ld)
                attn(r + "0.fn.", Ld, Ld, lens.latent_heads, lens.latent_dim_head)
                ln(r + "1.norm", Ld); ff(r + "1.fn.", Ld)
    return sd


# ----------------------------------------------------------------------------- synthetic inputs
def synth_text(n: int, gen: torch.Generator, ctx: int = 77, vocab: int = 49408) -> Tensor:
    """[SOT, k random ids, EOT, 0...] with k~U{4..20}; argmax == EOT position."""
    t = torch.zeros(n, ctx, dtype=torch.long)
    for i in range(n):
        k = int(torch.randint(4, 21, (1,), generator=gen))
        t[i, 0] = vocab - 2
        t[i, 1:1 + k] = torch.randint(1, vocab - 2, (k,), generator=gen)
        t[i, 1 + k] = vocab - 1
    return t


# ------------------------------------------------------------------------------------------------
# data-loader side of the 3D recipe: numpy, as the reference runs it
# ------------------------------------------------------------------------------------------------
def pc_farthest_point_sample(point, npoint: int, start: int):
    """farthest_point_sample, modal_3d/processors/pc_processor.py:8-29, with the random start made an argument.
    point [N, C] numpy (xyz first) -> (sampled points [npoint, C], indices [npoint])."""
    import numpy as np
    xyz = point[:, :3]
    N = xyz.shape[0]
    centroids = np.zeros((npoint,), dtype=np.int64)
    distance = np.ones((N,)) * 1e10
    farthest = int(start)
    for i in range(npoint):
        centroids[i] = farthest
        dist = np.sum((xyz - xyz[farthest, :]) ** 2, -1)
        mask = dist < distance
        distance[mask] = dist[mask]
        farthest = int(np.argmax(distance, -1))
    return point[centroids], centroids


def pc_norm(pc):
    """pc_norm, modal_3d/processors/pc_processor.py:32-38: centre on the centroid, scale the farthest point to radius 1."""
    import numpy as np
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
