"""`TriCLIP` with the reference's surface (open_clip/model.py:391-621) on the MI355X kernels.

The module tree owns ordinary fp32 `nn.Parameter`s under the reference's state_dict names,
so checkpoints and optimizers see the same thing; `encode_image / encode_text / encode_visual / forward`
run the towers through vitlens_hip engines (bf16 MFMA GEMMs, fp32 accumulation / statistics).  Engines
are (re)built from the parameters whenever their versions change.  There is no eager fallback: the
forward needs a GPU and libvitlens_hip.so.
"""
import logging
import math
from dataclasses import dataclass
from typing import Any, Optional

import numpy as np
import torch
import torch.nn as nn


@dataclass
class CLIPVisionCfg:
    layers: int = 12
    width: int = 768
    head_width: int = 64
    mlp_ratio: float = 4.0
    patch_size: int = 16
    image_size: int = 224
    visual_modality_type: str = "image"
    use_perceiver: bool = False
    perceiver_cfg: Optional[dict] = None
    use_visual_adapter: bool = False
    visual_adapter_cfg: Optional[dict] = None
    visual_arch: str = "perceiver_vit"
    exp_args: Optional[Any] = None


@dataclass
class CLIPTextCfg:
    context_length: int = 77
    vocab_size: int = 49408
    width: int = 512
    heads: int = 8
    layers: int = 12


def get_cast_dtype(precision: str):
    """model.py:100-106."""
    return {"bf16": torch.bfloat16, "fp16": torch.float16}.get(precision)


def get_input_dtype(precision: str):
    """model.py:109-115."""
    return {"bf16": torch.bfloat16, "pure_bf16": torch.bfloat16, "fp16": torch.float16, "pure_fp16": torch.float16}.get(precision)


def resize_pos_embed(state_dict, model, interpolation: str = "bicubic", antialias: bool = True):
    """Rescale `visual.positional_embedding` of a checkpoint to the model's token count (open_clip/model.py:1079-1150):
    bicubic resize of the square grid to grid_size - or to floor(sqrt(num_latents))^2 followed by a nearest resample to
    exactly num_latents when the tower resamples with a Perceiver (e.g. a ViT-B/16 checkpoint, grid 196, into 256 latents)."""
    import math
    import torch.nn.functional as F
    old = state_dict.get("visual.positional_embedding", None)
    vis = getattr(model, "visual", None)
    if old is None or vis is None:
        return
    g = vis.cfg.image_size // vis.cfg.patch_size
    new_len = g * g + 1
    n_lat = None
    if vis.use_perceiver:
        n_lat = getattr(vis.cfg.exp_args, "perceiver_num_latents", g * g)
        new_len = n_lat + 1
    if new_len == old.shape[0]:
        return
    tok, img = old[:1], old[1:]
    og = int(math.sqrt(len(img)))
    to = (int(math.sqrt(n_lat)),) * 2 if n_lat is not None else (g, g)
    logging.info("Resizing position embedding grid-size from %s to %s", (og, og), to)
    img = img.reshape(1, og, og, -1).permute(0, 3, 1, 2)
    img = F.interpolate(img.float(), size=to, mode=interpolation, antialias=antialias, align_corners=False)
    img = img.permute(0, 2, 3, 1).reshape(1, to[0] * to[1], -1)[0]
    if n_lat is not None and to[0] * to[1] != n_lat:
        img = F.interpolate(img.unsqueeze(0).transpose(1, 2), size=n_lat, mode="nearest").transpose(1, 2).squeeze(0)
    state_dict["visual.positional_embedding"] = torch.cat([tok.float(), img], dim=0).to(old.dtype)


class _Node(nn.Module):
    """Bare parameter container; children are attached by dotted name."""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool = False):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    if buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor))


def _g(obj, name, default=None):
    if obj is None:
        return default
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def _uni(shape, bound):
    return (torch.rand(*shape) * 2 - 1) * bound


def _block_params(prefix, D, hidden, text_init=None):
    """name -> tensor for one ResidualAttentionBlock with the reference's initial distributions
    (nn.MultiheadAttention: xavier_uniform in_proj, zero biases; nn.Linear: U(+-1/sqrt(fan_in)))."""
    out = {}
    for n in ("ln_1", "ln_2"):
        out[f"{prefix}{n}.weight"] = torch.ones(D); out[f"{prefix}{n}.bias"] = torch.zeros(D)
    out[prefix + "attn.in_proj_weight"] = _uni((3 * D, D), math.sqrt(6.0 / (4 * D)))
    out[prefix + "attn.in_proj_bias"] = torch.zeros(3 * D)
    out[prefix + "attn.out_proj.weight"] = _uni((D, D), D ** -0.5)
    out[prefix + "attn.out_proj.bias"] = torch.zeros(D)
    out[prefix + "mlp.c_fc.weight"] = _uni((hidden, D), D ** -0.5)
    out[prefix + "mlp.c_fc.bias"] = _uni((hidden,), D ** -0.5)
    out[prefix + "mlp.c_proj.weight"] = _uni((D, hidden), hidden ** -0.5)
    out[prefix + "mlp.c_proj.bias"] = _uni((D,), hidden ** -0.5)
    if text_init is not None:
        attn_std, proj_std, fc_std = text_init
        out[prefix + "attn.in_proj_weight"] = torch.randn(3 * D, D) * attn_std
        out[prefix + "attn.out_proj.weight"] = torch.randn(D, D) * proj_std
        out[prefix + "mlp.c_fc.weight"] = torch.randn(hidden, D) * fc_std
        out[prefix + "mlp.c_proj.weight"] = torch.randn(D, hidden) * proj_std
    return out


class VisionTransformer(nn.Module):
    """Parameter holder + HIP forward for one ViT tower (image tower, or Lens + ViT `visual` tower)."""

    def __init__(self, embed_dim: int, cfg: CLIPVisionCfg):
        super().__init__()
        self.cfg, self.embed_dim = cfg, embed_dim
        a = cfg.exp_args
        D, P = cfg.width, cfg.patch_size
        self.heads = D // cfg.head_width
        self.modality = {"3dpc": "pc", "pointcloud": "pc", "point_cloud": "pc", "point cloud": "pc"}.get(
            cfg.visual_modality_type, cfg.visual_modality_type)
        grid = (cfg.image_size // P) ** 2
        self.use_perceiver = bool(cfg.use_perceiver)
        self.perceiver_identity = bool(_g(a, "perceiver_as_identity", False)) or not self.use_perceiver
        if _g(a, "perceiver_as_transformer", False):
            raise NotImplementedError("perceiver_as_transformer is outside the hot path (SURVEY §8)")
        n_tok = _g(a, "perceiver_num_latents", grid) if self.use_perceiver else grid
        scale = D ** -0.5
        prm = {"class_embedding": scale * torch.randn(D), "positional_embedding": scale * torch.randn(n_tok + 1, D),
               "proj": scale * torch.randn(D, embed_dim)}
        for n in ("ln_pre", "ln_post"):
            prm[n + ".weight"] = torch.ones(D); prm[n + ".bias"] = torch.zeros(D)
        hidden = int(D * cfg.mlp_ratio)
        for i in range(cfg.layers):
            prm.update(_block_params(f"transformer.resblocks.{i}.", D, hidden))
        bufs = {}
        if self.modality in ("image", "tactile"):
            prm["conv1.weight"] = _uni((D, 3, P, P), (3 * P * P) ** -0.5)
        elif self.modality == "depth":
            prm["visual_adapter.conv1.weight"] = _uni((D, 1, P, P), (P * P) ** -0.5)
            prm["visual_adapter.pos_emb"] = scale * torch.randn(grid, D)
        elif self.modality == "audio":
            fd = (a.audio_mel_bins - P) // a.audio_fstride + 1
            td = (a.audio_target_length - P) // a.audio_tstride + 1
            prm["visual_adapter.conv1.weight"] = _uni((D, 1, P, P), (P * P) ** -0.5)
            prm["visual_adapter.pos_emb"] = scale * torch.randn(fd * td, D)
        elif self.modality == "eeg":
            fan = a.eeg_chans * a.eeg_window_size
            prm["visual_adapter.proj.weight"] = _uni((D, a.eeg_chans, a.eeg_window_size), fan ** -0.5)
            prm["visual_adapter.proj.bias"] = _uni((D,), fan ** -0.5)
            prm["visual_adapter.pos_emb"] = scale * torch.randn((a.eeg_time_len - a.eeg_window_size) // a.eeg_stride + 1, D)
        elif self.modality == "pc":
            E, Tr = a.pc_encoder_dims, a.pc_trans_dim
            def lin(name, o, i, conv=False):
                prm[f"visual_adapter.{name}.weight"] = _uni((o, i, 1) if conv else (o, i), i ** -0.5)
                prm[f"visual_adapter.{name}.bias"] = _uni((o,), i ** -0.5)
            def bn(name, c):
                prm[f"visual_adapter.{name}.weight"] = torch.ones(c); prm[f"visual_adapter.{name}.bias"] = torch.zeros(c)
                bufs[f"visual_adapter.{name}.running_mean"] = torch.zeros(c)
                bufs[f"visual_adapter.{name}.running_var"] = torch.ones(c)
                bufs[f"visual_adapter.{name}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
            if _g(a, "pc_tokenizer", "pointbert") == "pnsa":
                # PointNSATokenizer (modal_3d/models/pointnet/pointnet_util.py:345-360): set abstraction over
                # [xyz - centre ++ point features] with mlp [64, 64, encoder_dims], then lift = Conv1d + LayerNorm
                last = _g(a, "pc_in_channel", 3) + 3
                for i, oc in enumerate((64, 64, E)):
                    prm[f"visual_adapter.sa.mlp_convs.{i}.weight"] = _uni((oc, last, 1, 1), last ** -0.5)
                    prm[f"visual_adapter.sa.mlp_convs.{i}.bias"] = _uni((oc,), last ** -0.5)
                    last = oc
                for i, oc in enumerate((64, 64, E)):
                    bn(f"sa.mlp_bns.{i}", oc)
                lin("lift.0", Tr, E + 3, True)
                prm["visual_adapter.lift.2.weight"] = torch.ones(Tr); prm["visual_adapter.lift.2.bias"] = torch.zeros(Tr)
            elif _g(a, "pc_tokenizer", "pointbert") == "pointbert":
                lin("encoder.first_conv.0", 128, 3, True); bn("encoder.first_conv.1", 128); lin("encoder.first_conv.3", 256, 128, True)
                lin("encoder.second_conv.0", 512, 512, True); bn("encoder.second_conv.1", 512); lin("encoder.second_conv.3", E, 512, True)
                lin("reduce_dim", Tr, E); lin("pos_embed.0", 128, 3); lin("pos_embed.2", Tr, 128)
            else:
                raise NotImplementedError(f"pc_tokenizer {a.pc_tokenizer!r} (visual_adapter.py:11-22 knows pointbert and pnsa)")
        else:
            raise NotImplementedError(f"modality {self.modality!r}")
        if self.use_perceiver and not self.perceiver_identity:
            Ld, C = a.perceiver_latent_dim, a.perceiver_input_chan
            prm["perceiver.latents"] = torch.randn(a.perceiver_num_latents, Ld)
            def ln(name, d):
                prm[name + ".weight"] = torch.ones(d); prm[name + ".bias"] = torch.zeros(d)
            def attn(q, qd, cd, heads, dh):
                inner = heads * dh
                prm[q + "to_q.weight"] = _uni((inner, qd), qd ** -0.5)
                prm[q + "to_kv.weight"] = _uni((2 * inner, cd), cd ** -0.5)
                prm[q + "to_out.weight"] = _uni((qd, inner), inner ** -0.5); prm[q + "to_out.bias"] = _uni((qd,), inner ** -0.5)
            def ff(q, d):
                prm[q + "net.0.weight"] = _uni((8 * d, d), d ** -0.5); prm[q + "net.0.bias"] = _uni((8 * d,), d ** -0.5)
                prm[q + "net.2.weight"] = _uni((d, 4 * d), (4 * d) ** -0.5); prm[q + "net.2.bias"] = _uni((d,), (4 * d) ** -0.5)
            for i in range(a.perceiver_depth):
                q = f"perceiver.layers.{i}."
                ln(q + "0.norm", Ld); ln(q + "0.norm_context", C)
                attn(q + "0.fn.", Ld, C, a.perceiver_cross_heads, a.perceiver_cross_dim_head)
                ln(q + "1.norm", Ld); ff(q + "1.fn.", Ld)
                for j in range(a.perceiver_self_per_cross_attn):
                    r = f"{q}2.{j}."
                    ln(r + "0.norm", Ld); attn(r + "0.fn.", Ld, Ld, a.perceiver_latent_heads, a.perceiver_latent_dim_head)
                    ln(r + "1.norm", Ld); ff(r + "1.fn.", Ld)
        for k, v in prm.items():
            _attach(self, k, v)
        for k, v in bufs.items():
            _attach(self, k, v, buffer=True)
        if self.use_perceiver and not self.perceiver_identity and _g(a, "perceiver_weight_tie_layers", False):
            # perceiver.py:249-254 (`cache_fn`): layers 1 .. depth-1 are ONE set of modules - the state_dict lists them under
            # every index, named_parameters() (and so the optimizer and lock()) only once, under layer 1
            mods = self.perceiver.layers._modules
            for i in range(2, len(mods)):
                mods[str(i)] = mods["1"]
        self.image_mean = self.image_std = None
        self.res_dtype = torch.float32          # residual-stream dtype of the HIP towers; TriCLIP.set_precision maps `precision` to it
        self.arith_f32 = False                  # precision="fp32": eval-mode, graph-less forwards run true fp32 arithmetic (vitlens_hip/f32.py)
        self._f32_engine, self._f32_vers = None, None
        self._engine = None
        self._engine_key, self._engine_vers = None, None
        self._trainer_obj, self._trainer_key, self._gen = None, None, 0
        self._freeze_bn = False
        self._bn_sync = None          # (communicator, world size) once set_bn_sync(True) was called in a multi-rank job

    def set_bn_sync(self, enabled=True, comm=None, world_size=None):
        """Counterpart of `torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)` (--use-bn-sync, e.g.
        training/point_cloud/pc_tri_main.py:372-373) for this tower: its BatchNorm layers (the PointBERT mini-PointNet
        only) take their batch statistics and backward sums over all ranks of `torch.distributed` (or of `comm`, any
        object with all_gather / all_reduce_sum).  A no-op for towers without BatchNorm and in single-process runs."""
        if not enabled or self.modality != "pc":
            self._bn_sync = None
            return self
        if comm is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
                self._bn_sync = None
                return self
            from vitlens_hip.step import TorchComm
            comm, world_size = TorchComm(), dist.get_world_size()
        elif world_size is None:
            raise ValueError("set_bn_sync(comm=...) needs world_size: the number of ranks `comm` spans")
        self._bn_sync = (comm, int(world_size))
        return self

    # -------------------------------------------------------------------------------------- lock recipes
    def lock(self, unlocked_groups=0, freeze_bn_stats=False, unlock_cls=False, unlock_pos_emb=False,
             unlock_trans_first_n_layers=None):
        """VisionTransformer.lock, open_clip/transformer.py:553-627."""
        for p in self.parameters():
            p.requires_grad = False
        self._freeze_bn = bool(freeze_bn_stats)      # PointTokenizer BatchNorm: running statistics, no update (transformer.py:558-560)
        on = []
        if unlocked_groups != 0:
            # LiT-style grouped unlock (transformer.py:565-597): [stem] + one group per block but the last +
            # [last block, ln_post] + [proj]; the first k groups with exp_args.unlock_from_head, else the last k.
            L = self.cfg.layers
            stem = ("conv1.", "class_embedding", "positional_embedding", "ln_pre.")
            groups = [stem] + [(f"transformer.resblocks.{i}.",) for i in range(L - 1)]
            groups += [(f"transformer.resblocks.{L - 1}.", "ln_post."), ("proj",)]
            picked = groups[:unlocked_groups] if _g(self.cfg.exp_args, "unlock_from_head", False) else groups[-unlocked_groups:]
            prefixes = tuple(x for grp in picked for x in grp)
            for n, p in self.named_parameters():
                if n.startswith(prefixes):
                    on.append(p)
        for n, p in self.named_parameters():
            if n.startswith("perceiver.") or n.startswith("visual_adapter."):
                on.append(p)
        if unlock_cls:
            on.append(self.class_embedding)
        if unlock_pos_emb:
            on.append(self.positional_embedding)
        if unlock_trans_first_n_layers is not None:
            for i in range(unlock_trans_first_n_layers):
                on.extend(self.transformer.resblocks._modules[str(i)].parameters())
        for p in on:
            p.requires_grad = True

    def set_grad_checkpointing(self, enable=True):
        """Activation recompute for the transformer blocks (Transformer.forward, open_clip/transformer.py:366-368): the
        trainer keeps only each block's input and re-runs the block's forward right before its backward."""
        self.grad_checkpointing = bool(enable)

    def drop_output_projection(self):
        """`backbone.proj = None` (VitLens-OpenShape/src/models/clip_bind.py:38-41): the tower returns ln_post(cls) at
        transformer width (transformer.py:783-784 `if self.proj is not None`); a wrapper applies its own projection."""
        if "proj" in self._parameters:
            del self._parameters["proj"]
        self.proj = None
        self.embed_dim = self.cfg.width
        self._engine = self._engine_key = None
        self._trainer_obj = self._trainer_key = None

    def keep_last_layers(self, n_keep: int):
        """`model.visual.transformer.resblocks = resblocks[-n_keep:]` of --skip-trans-first-n-layers (factory.py:347-360):
        the first blocks are dropped AFTER the checkpoint is loaded and the kept ones are renumbered from 0, as slicing
        an nn.Sequential / ModuleList does (so a later state_dict() has n_keep blocks)."""
        import dataclasses
        mods = self.transformer.resblocks._modules
        L = len(mods)
        if not 0 < n_keep <= L:
            raise ValueError(f"cannot keep {n_keep} of {L} transformer layers")
        kept = [mods[str(i)] for i in range(L - n_keep, L)]
        mods.clear()
        for j, m in enumerate(kept):
            mods[str(j)] = m
        self.cfg = dataclasses.replace(self.cfg, layers=n_keep)
        self._engine = self._engine_key = None
        self._trainer_obj = self._trainer_key = None

    # -------------------------------------------------------------------------------------- engines
    def _cfgs(self):
        from vitlens_hip import engine as E
        c, a = self.cfg, self.cfg.exp_args
        tower = E.TowerCfg(width=c.width, layers=c.layers, heads=self.heads, mlp_ratio=c.mlp_ratio, patch=c.patch_size,
                           image_size=c.image_size, embed_dim=self.embed_dim)
        lens = None
        if self.modality not in ("image", "tactile"):
            lens = E.LensCfg(
                modality=self.modality, perceiver_identity=self.perceiver_identity,
                depth=_g(a, "perceiver_depth", 1), self_per_cross=_g(a, "perceiver_self_per_cross_attn", 1),
                num_latents=_g(a, "perceiver_num_latents", 256), latent_dim=_g(a, "perceiver_latent_dim", c.width),
                input_chan=_g(a, "perceiver_input_chan", c.width), cross_heads=_g(a, "perceiver_cross_heads", 1),
                cross_dim_head=_g(a, "perceiver_cross_dim_head", 64), latent_heads=_g(a, "perceiver_latent_heads", 16),
                latent_dim_head=_g(a, "perceiver_latent_dim_head", 64), audio_fstride=_g(a, "audio_fstride", 10),
                audio_tstride=_g(a, "audio_tstride", 10), audio_mel_bins=_g(a, "audio_mel_bins", 128),
                audio_target_length=_g(a, "audio_target_length", 512), eeg_chans=_g(a, "eeg_chans", 128),
                eeg_time_len=_g(a, "eeg_time_len", 512), eeg_window_size=_g(a, "eeg_window_size", 1),
                eeg_stride=_g(a, "eeg_stride", 1), pc_num_group=_g(a, "pc_num_group", 512),
                pc_group_size=_g(a, "pc_group_size", 32), pc_encoder_dims=_g(a, "pc_encoder_dims", 256),
                pc_trans_dim=_g(a, "pc_trans_dim", 384), pc_tokenizer=_g(a, "pc_tokenizer", "pointbert"),
                pc_radius=_g(a, "pc_radius", 0.2), pc_in_dim=_g(a, "pc_in_channel", 3),
                use_orig_pos=not _g(a, "disable_orig_pos", False),
                disable_adapter_pos=bool(_g(a, "disable_visual_adapter_pos", False)),
                weight_tie_layers=bool(_g(a, "perceiver_weight_tie_layers", False)))
        return tower, lens

    def engine(self):
        """The tower's executor on the HIP kernels.  Built ONCE per (device, train/eval mode); when parameters changed since
        the last call (an optimizer step, load_state_dict) only their device operands are re-derived, in place - the
        frozen ViT-L weights are cast to bf16 once, and the trainer bound to this engine keeps its activation buffers."""
        from vitlens_hip import engine as E
        dev = self.class_embedding.device
        if dev.type != "cuda":
            raise RuntimeError("the ViT-Lens towers run on the MI355X kernels only: move the model to a GPU")
        key = (str(dev), self.training, self.res_dtype)
        vers = {n: p._version for n, p in self.named_parameters()}
        if not self.training:
            # eval engines also follow the BUFFERS (BatchNorm running statistics edited in place); in train mode the trainer owns
            # and updates those every forward, which must not look like a parameter change
            vers.update({"buffer:" + n: b._version for n, b in self.named_buffers()})
        if self._engine is None or key != self._engine_key:
            sd = {("t." + k): v for k, v in self.state_dict().items()}
            tower, lens = self._cfgs()
            if lens is None:
                self._engine = E.VitEngine(sd, "t.", tower, dev, res_dtype=self.res_dtype)
            else:
                self._engine = E.LensEngine(sd, "t.", tower, lens, dev, res_dtype=self.res_dtype)
            self._engine_key, self._engine_vers = key, vers
        elif vers != self._engine_vers:
            changed = [n[7:] if n.startswith("buffer:") else n for n, v in vers.items() if self._engine_vers.get(n) != v]
            sd = {("t." + k): v for k, v in self.state_dict().items()}
            if isinstance(self._engine, E.VitEngine):
                self._engine.update_params(sd, changed)
            else:
                self._engine.update_params(sd, "t.", changed)
            if self._trainer_obj is not None and self._trainer_key is not None and self._trainer_key[0] == id(self._engine):
                from vitlens_hip.train import refresh_trainer
                refresh_trainer(self._trainer_obj, sd, changed)
            self._engine_vers = vers
        return self._engine

    # -------------------------------------------------------------------------------------- training (autograd)
    def _train_flags(self):
        req = {n for n, p in self.named_parameters() if p.requires_grad}
        blocks = tuple(sorted({int(n.split(".")[2]) for n in req if n.startswith("transformer.resblocks.")}))
        return (blocks, "class_embedding" in req, "positional_embedding" in req, any(n.startswith("ln_pre.") for n in req),
                any(n.startswith("ln_post.") for n in req), "proj" in req, "conv1.weight" in req,
                self.training and not self._freeze_bn, self._bn_sync is not None, bool(getattr(self, "grad_checkpointing", False)))

    def _trainer(self):
        """Forward-with-saved-activations / backward executor for the current lock recipe (vitlens_hip.train), bound to
        the up-to-date engine; rebuilt when the engine or the set of trainable parameters changes."""
        from vitlens_hip import train as T
        eng = self.engine()
        flags = self._train_flags()
        key = (id(eng), flags)
        if self._trainer_obj is None or key != self._trainer_key:
            blocks, cls, pos, lpre, lpost, proj, conv, bn_train, _, ckpt = flags
            kw = dict(train_blocks=blocks, train_cls=cls, train_pos=pos, train_ln_pre=lpre, train_ln_post=lpost, train_proj=proj,
                      checkpoint=ckpt)
            if self.modality in ("image", "tactile"):
                tr = T.ImageTowerTrainer(eng, kw, train_conv=conv)
            elif self.modality == "depth" and self.perceiver_identity:
                tr = T.DepthLensTrainer(eng, tower_kw=kw)
            elif self.modality == "audio" and not self.perceiver_identity:
                tr = T.AudioLensTrainer(eng, tower_kw=kw)
            elif self.modality == "eeg" and not self.perceiver_identity:
                tr = T.EEGLensTrainer(eng, tower_kw=kw)
            elif self.modality == "pc" and not self.perceiver_identity:
                from vitlens_hip.points import PNSATokenizerTrainer, PointTokenizerTrainer
                sd = {("t." + k): v for k, v in self.state_dict().items() if k.startswith("visual_adapter.")}
                sync = self._bn_sync or (None, 1)
                cls_tok = PNSATokenizerTrainer if eng.lens.pc_tokenizer == "pnsa" else PointTokenizerTrainer
                tok = cls_tok(sd, "t.visual_adapter.", eng.lens, eng.device, bn_training=bn_train, bn_sync=sync[0],
                              world_size=sync[1])
                tr = T.PCLensTrainer(eng, tok, tower_kw=kw)
            else:
                raise NotImplementedError(f"training recipe for modality {self.modality!r} "
                                          f"(perceiver_identity={self.perceiver_identity}) is not implemented")
            self._trainer_obj, self._trainer_key = tr, key
        return self._trainer_obj

    def _named_grads(self, tr):
        """Gradients of the last backward under THIS module's parameter names and shapes."""
        raw = tr.perc.reference_named_grads() if hasattr(tr, "perc") else dict(tr.grads)
        params = dict(self.named_parameters())
        out = {}
        for k, g in raw.items():
            for pre in ("visual.", "t."):
                if k.startswith(pre):
                    k = k[len(pre):]
                    break
            if k.endswith(".weight_gemm"):          # conv-as-GEMM layout [O, K padded to 64] -> [O, C, kh, kw]
                k = k[:-5]
                g = g[:, :params[k][0].numel()]
            if k in params:
                out[k] = g.reshape(params[k].shape)
        return out

    def forward(self, x: torch.Tensor, fwd_output_tokens: bool = False, **kwargs):
        if fwd_output_tokens:
            raise NotImplementedError("token outputs are only used by the video-distillation losses (out of scope)")
        x = x.to(self.class_embedding.device)
        trainable = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        if torch.is_grad_enabled() and trainable:
            names = tuple(n for n, _ in trainable)
            return _TowerFn.apply(self, x, kwargs, names, *[p for _, p in trainable])
        if self.modality == "pc" and self.training and not self._freeze_bn:
            # a no-grad forward in TRAIN mode (the feature-caching pass of the accumulation loop, training/train.py:154-178):
            # the reference's BatchNorm layers use the batch statistics there and update their running statistics, exactly
            # as in the pass with a graph - not the folded running statistics of the inference engine
            tr = self._trainer()
            self._gen += 1            # a pending backward of an earlier forward must not use the overwritten activations
            feat = tr.forward(x, **kwargs)
            self._sync_bn_buffers(tr)
            return feat.clone()
        f32 = self._engine_f32() if (self.arith_f32 and not self.training) else None
        if f32 is not None:
            return f32.encode(x)
        eng = self.engine()
        if self.modality in ("image", "tactile"):
            return eng.encode_image(x)
        return eng.encode(x, **kwargs)

    def _engine_f32(self):
        """The fp32-arithmetic executor of this tower (precision="fp32", inference), or None where it does not exist: the
        Perceiver / point-cloud / audio / EEG Lenses and head dims other than 32 / 64 stay on the 16-bit engines."""
        from vitlens_hip import f32 as F
        simple = self.modality in ("image", "tactile") or (self.modality == "depth" and self.perceiver_identity)
        if not simple or not F.f32_supported(self.cfg.width, self.heads) or self.class_embedding.device.type != "cuda":
            return None
        # (the device is part of the key: after `.to(other_gpu)` the cached engine's operands live on the old device)
        vers = (str(self.class_embedding.device), {n: p._version for n, p in self.named_parameters()})
        if self._f32_engine is None or vers != self._f32_vers:
            sd = {("t." + k): v for k, v in self.state_dict().items()}
            tower, lens = self._cfgs()
            self._f32_engine = F.VitEngineF32(sd, "t.", tower, self.class_embedding.device, depth=self.modality == "depth",
                                              use_orig_pos=True if lens is None else lens.use_orig_pos,
                                              disable_adapter_pos=False if lens is None else lens.disable_adapter_pos)
            self._f32_vers = vers
        return self._f32_engine

    def _sync_bn_buffers(self, tr):
        """Running statistics of the point tokenizer's BatchNorm layers after a train-mode forward -> this module's buffers."""
        bufs = dict(self.named_buffers())
        for k, (rm, rv) in tr.tok.running.items():
            for nm, src in ((".running_mean", rm), (".running_var", rv)):
                dst = bufs["visual_adapter." + k + nm]
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
            bufs["visual_adapter." + k + ".num_batches_tracked"] += 1


class _TowerFn(torch.autograd.Function):
    """One autograd node per tower call: forward = the HIP forward with saved activations, backward = the hand-written
    HIP backward of the tower (vitlens_hip.train); returns a gradient for every parameter that requires grad, so
    `loss.backward(); optimizer.step()` of the reference's loop (training/train.py:131-152, 212-235) trains the Lens."""

    @staticmethod
    def forward(ctx, module, x, kwargs, names, *params):
        tr = module._trainer()
        module._gen += 1
        ctx.module, ctx.names, ctx.gen = module, names, module._gen
        ctx.shapes = [tuple(p.shape) for p in params]
        feat = tr.forward(x.detach(), **kwargs)
        if module.modality == "pc" and module.training and not module._freeze_bn:      # BatchNorm running statistics
            module._sync_bn_buffers(tr)
        return feat.clone()        # the trainer's feature buffer is reused by the next forward

    @staticmethod
    def backward(ctx, dfeat):
        m = ctx.module
        if m._gen != ctx.gen:
            raise RuntimeError("this tower ran another forward before this backward: its saved activations were overwritten. "
                               "Run backward after each forward (the reference's accumulation loop does), or use vitlens_hip.step.")
        tr = m._trainer_obj
        for g in tr.grads.values():
            g.zero_()
        tr.backward(dfeat.contiguous().float())
        named = m._named_grads(tr)
        outs = []
        for n, shp in zip(ctx.names, ctx.shapes):
            g = named.get(n)
            outs.append(None if g is None else g.reshape(shp).clone())
        return (None, None, None, None, *outs)


class _TextFn(torch.autograd.Function):
    """The text tower as one autograd node (round 6): forward = the HIP forward with saved activations on bf16 operands,
    backward = vitlens_hip.train.TextTowerTrainer; a gradient for every text parameter that requires grad.  Used only when
    the text tower is NOT locked (`encode_text` under grad mode with trainable text parameters; training/train.py:212-235
    trains whatever requires grad - every ViT-Lens recipe locks it and takes the frozen fp16 engine instead)."""

    @staticmethod
    def forward(ctx, model, text, names, *params):
        tr = model._text_trainer()
        model._text_gen += 1
        ctx.model, ctx.names, ctx.gen = model, names, model._text_gen
        ctx.shapes = [tuple(p.shape) for p in params]
        return tr.forward(text).clone()

    @staticmethod
    def backward(ctx, dfeat):
        m = ctx.model
        if m._text_gen != ctx.gen:
            raise RuntimeError("the text tower ran another forward before this backward: its saved activations were overwritten")
        tr = m._text_trainer_obj
        for g in tr.grads.values():
            g.zero_()
        tr.backward(dfeat.contiguous().float())
        outs = []
        for n, shp in zip(ctx.names, ctx.shapes):
            g = tr.grads.get(n)
            outs.append(None if g is None else g.reshape(shp).clone())
        return (None, None, None, *outs)


class _NormalizeFn(torch.autograd.Function):
    """F.normalize(dim=-1) (model.py:522-540) with its backward, both on the HIP kernels."""

    @staticmethod
    def forward(ctx, x):
        from vitlens_hip import ops
        x = x.contiguous().float()
        norms = torch.empty(x.shape[0], device=x.device, dtype=torch.float32)
        f = ops.l2_normalize(x, norms=norms)
        ctx.save_for_backward(f, norms)
        return f

    @staticmethod
    def backward(ctx, df):
        from vitlens_hip import ops
        f, norms = ctx.saved_tensors
        return ops.l2_normalize_bwd(f, df.contiguous().float(), norms)


def _normalize(x):
    """F.normalize(dim=-1) on the HIP kernel (model.py:522-540); differentiable when the features carry a graph."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _NormalizeFn.apply(x)
    from vitlens_hip import ops
    return ops.l2_normalize(x.contiguous().float())


class TriCLIP(nn.Module):
    def __init__(self, embed_dim: int, vision_cfg, text_cfg, quick_gelu: bool = False, cast_dtype=None,
                 output_dict: bool = False):
        super().__init__()
        if quick_gelu:
            raise NotImplementedError("QuickGELU towers are not part of the named configs")
        vision_cfg = CLIPVisionCfg(**vision_cfg) if isinstance(vision_cfg, dict) else vision_cfg
        text_cfg = CLIPTextCfg(**text_cfg) if isinstance(text_cfg, dict) else text_cfg
        self.exp_args = vision_cfg.exp_args
        self.output_dict = output_dict
        self.visual_arch = vision_cfg.visual_arch
        if self.visual_arch != "perceiver_vit":
            raise NotImplementedError("only visual_arch='perceiver_vit' (Lens -> frozen ViT) is on the hot path")
        img_cfg = CLIPVisionCfg(layers=vision_cfg.layers, width=vision_cfg.width, head_width=vision_cfg.head_width,
                                mlp_ratio=vision_cfg.mlp_ratio, patch_size=vision_cfg.patch_size,
                                image_size=vision_cfg.image_size, exp_args=vision_cfg.exp_args)
        self.image = VisionTransformer(embed_dim, img_cfg)            # module_cfg.set_default_image_cfg
        self.visual = VisionTransformer(embed_dim, vision_cfg)
        # text tower flattened into the root (model.py:435-443)
        self.text_cfg = text_cfg
        self.context_length, self.vocab_size = text_cfg.context_length, text_cfg.vocab_size
        W, Ly = text_cfg.width, text_cfg.layers
        _attach(self, "token_embedding.weight", torch.randn(text_cfg.vocab_size, W) * 0.02)
        self.positional_embedding = nn.Parameter(torch.randn(text_cfg.context_length, W) * 0.01)
        self.text_projection = nn.Parameter(torch.randn(W, embed_dim) * W ** -0.5)
        _attach(self, "ln_final.weight", torch.ones(W)); _attach(self, "ln_final.bias", torch.zeros(W))
        stds = (W ** -0.5, (W ** -0.5) * ((2 * Ly) ** -0.5), (2 * W) ** -0.5)
        for i in range(Ly):
            for k, v in _block_params(f"transformer.resblocks.{i}.", W, 4 * W, text_init=stds).items():
                _attach(self, k, v)
        mask = torch.full((text_cfg.context_length, text_cfg.context_length), float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self._text_engine, self._text_key = None, None
        self._res_dtype = torch.float32
        self._arith_f32 = False
        self._text_arith = "f16"

    def set_precision(self, precision: str, text_arith: str = "f16"):
        """`precision` of tri_create_model (factory.py:164): the GEMMs always run on 16-bit operands with fp32 accumulation
        (what the reference's amp_bf16 autocast computes); "fp32" keeps the residual stream and its gradient in f32 (more
        precise than the reference's autocast), "amp" / "amp_bf16" / "bf16" carry them in bf16 exactly as autocast does.
        text_arith: operands of the frozen TEXT tower (vitlens_hip.engine.TextEngine): "f16" (default) IEEE half on an fp32
        residual stream - what holds its cosine matrix within 1e-3 of the fp32 CPU path (1-2e-4) at the bf16 rate; "bf16x2"
        two-term bf16 weights (round 4: 6-8e-4 at twice the flops); "bf16" the reference's amp_bf16 arithmetic (0.8-1.9e-3)."""
        import warnings
        dt = torch.float32 if precision == "fp32" else torch.bfloat16
        self._res_dtype = self.image.res_dtype = self.visual.res_dtype = dt
        self.image.arith_f32 = self.visual.arith_f32 = self._arith_f32 = precision == "fp32"
        if text_arith not in ("f16", "bf16x2", "bf16"):
            raise ValueError(f"text_arith must be 'f16', 'bf16x2' or 'bf16', got {text_arith!r}")
        self._text_arith = text_arith
        self._text_engine = None
        # what actually runs, stated where the caller asked for something else (reference: factory.py:260-295 converts the
        # model / sets up autocast; training/precision.py:5-12)
        self.precision_requested = precision
        text = {"f16": "fp16 x fp16 -> fp32, residual stream fp32 (falls back to two-term bf16 weights where the width is not a multiple of 256)",
                "bf16x2": "two-term bf16 weights x bf16 activations -> fp32, residual stream fp32",
                "bf16": "bf16 x bf16 -> fp32, residual stream " + ("fp32" if dt == torch.float32 else "bf16")}[text_arith]
        self.precision_effective = ("image / modality towers: bf16 x bf16 -> fp32 matrix products, residual stream "
                                    + ("fp32" if dt == torch.float32 else "bf16") + "; text tower: " + text
                                    + "; LayerNorm / softmax statistics, features, logits, loss in fp32")
        if precision == "fp32":
            self.precision_effective = ("eval mode, no autograd graph: true fp32 arithmetic (fp32-input MFMA, vitlens_hip/f32.py; 1/16 of "
                                        "the bf16 matrix rate: ViT-L/14 584 img/s against 5 425 with bf16 operands - pass "
                                        "precision='amp_bf16' for throughput) for the "
                                        "image / tactile / depth(identity Perceiver) towers and the text tower with head dim 32 or 64; "
                                        "otherwise (train mode, towers with trainable parameters, Perceiver / audio / point-cloud / EEG "
                                        "Lenses): " + self.precision_effective)
            warnings.warn("precision='fp32': fp32 arithmetic runs for eval-mode inference (model.eval(), no trainable tower in the "
                          "call); training keeps bf16 matrix operands with fp32 accumulation on fp32 residual / gradient streams - "
                          "the MI355X path has no fp32 backward.  See model.precision_effective.", UserWarning, stacklevel=3)
        elif precision == "amp":
            warnings.warn("precision='amp' (fp16 autocast in the reference) runs as amp_bf16 on the MI355X path: bf16 operands, "
                          "fp32 accumulation.  bf16 has fp32's range, so no loss scaling is needed; a GradScaler around the loop "
                          "body works as in the reference (scale / unscale_ / skipped step on overflow).", UserWarning, stacklevel=3)
        return self

    # ---- lock recipes (model.py:448-502) -------------------------------------------------------
    def lock_image_tower(self, unlocked_groups=0, freeze_bn_stats=False, unlock_cls=False, unlock_pos_emb=False):
        self.image.lock(unlocked_groups=unlocked_groups, freeze_bn_stats=freeze_bn_stats, unlock_cls=unlock_cls,
                        unlock_pos_emb=unlock_pos_emb)

    def lock_visual_tower(self, unlocked_groups=0, freeze_bn_stats=False, unlock_cls=False, unlock_pos_emb=False,
                          unlock_trans_first_n_layers=None):
        self.visual.lock(unlocked_groups=unlocked_groups, freeze_bn_stats=freeze_bn_stats, unlock_cls=unlock_cls,
                         unlock_pos_emb=unlock_pos_emb, unlock_trans_first_n_layers=unlock_trans_first_n_layers)

    def lock_text_tower(self, unlocked_layers: int = 0, freeze_layer_norm: bool = True):
        """As the reference: TriCLIP has no `.text` module, so its lock_text_tower falls through to Transformer.lock,
        which freezes every parameter whatever `unlocked_layers` says (model.py:479-502, transformer.py:373-375)."""
        for n, p in self.named_parameters():
            if n.startswith(("transformer.", "token_embedding.", "ln_final.")) or n in ("positional_embedding", "text_projection"):
                p.requires_grad = False

    def set_grad_checkpointing(self, enable=True):
        self.visual.set_grad_checkpointing(enable); self.image.set_grad_checkpointing(enable)

    # ---- encoders ---------------------------------------------------------------------------------
    def _text(self):
        from vitlens_hip import engine as E
        dev = self.positional_embedding.device
        if dev.type != "cuda":
            raise RuntimeError("the ViT-Lens towers run on the MI355X kernels only: move the model to a GPU")
        # (logit_scale is not a text-tower operand: it changes every step and must not invalidate the frozen tower's engine)
        prm = dict(self.named_parameters())
        names = [n for n in prm if not n.startswith(("image.", "visual.")) and n != "logit_scale"]
        from vitlens_hip import f32 as F32
        f32 = self._arith_f32 and not self.training and F32.f32_supported(self.text_cfg.width, self.text_cfg.heads)
        # two slots - the fp32-arithmetic engine (eval under precision="fp32") and the 16-bit one - so that a train() / eval()
        # flip under "fp32" re-uses both instead of rebuilding the tower's operands every time
        slot = "f32" if f32 else "b16"
        key = (str(dev), self._res_dtype, "f32" if f32 else self._text_arith, tuple(prm[n]._version for n in names))
        if not isinstance(self._text_engine, dict):
            self._text_engine, self._text_key = {}, {}
        if slot not in self._text_engine or key != self._text_key.get(slot):
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith(("image.", "visual."))}
            t = self.text_cfg
            cfg = E.TextCfg(context_length=t.context_length, vocab_size=t.vocab_size, width=t.width, heads=t.heads, layers=t.layers,
                            embed_dim=self.text_projection.shape[1])
            if f32:
                self._text_engine[slot] = F32.TextEngineF32(sd, cfg, dev)
            else:
                self._text_engine[slot] = E.TextEngine(sd, cfg, dev, res_dtype=self._res_dtype, arith=self._text_arith)
            self._text_key[slot] = key
        return self._text_engine[slot]

    def encode_image(self, image, normalize: bool = False):
        n_img = None
        if image.ndim == 5:                                           # [b, t, c, h, w]: mean of the frames' raw features (model.py:510-523)
            n_img = image.size(1)
            image = image.reshape(-1, *image.shape[2:])
        features = self.image(image)
        if n_img is not None:
            features = features.reshape(-1, n_img, features.shape[-1]).mean(1).contiguous()
        return _normalize(features) if normalize else features

    def encode_visual(self, visual_x, normalize: bool = False, **kwargs):
        features = self.visual(visual_x, **kwargs)
        return _normalize(features) if normalize else features

    def _text_trainer(self):
        """The trainable text tower (bf16 operands, saved activations), rebuilt when a text parameter changed."""
        from vitlens_hip import engine as E, train as TR
        dev = self.positional_embedding.device
        if dev.type != "cuda":
            raise RuntimeError("the ViT-Lens towers run on the MI355X kernels only: move the model to a GPU")
        prm = dict(self.named_parameters())
        names = [n for n in prm if not n.startswith(("image.", "visual.")) and n != "logit_scale"]
        key = (str(dev), tuple(prm[n]._version for n in names))
        if getattr(self, "_text_trainer_obj", None) is None or key != self._text_trainer_key:
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith(("image.", "visual."))}
            t = self.text_cfg
            cfg = E.TextCfg(context_length=t.context_length, vocab_size=t.vocab_size, width=t.width, heads=t.heads, layers=t.layers,
                            embed_dim=self.text_projection.shape[1])
            old = getattr(self, "_text_trainer_obj", None)
            eng = E.TextEngine(sd, cfg, dev, res_dtype=torch.float32, arith="bf16")
            tr = TR.TextTowerTrainer(eng)
            if old is not None:
                tr._saved = old._saved          # keep the activation buffers (only the operands changed)
            self._text_trainer_obj, self._text_trainer_key = tr, key
        return self._text_trainer_obj

    _text_gen = 0

    def encode_text(self, text, normalize: bool = False):
        if torch.is_grad_enabled():
            trainable = [(n, p) for n, p in self.named_parameters() if p.requires_grad and not n.startswith(("image.", "visual."))
                         and n != "logit_scale"]
            if trainable:
                # a text tower that is NOT locked trains (training/train.py:212-235); every ViT-Lens recipe locks it and runs the
                # frozen fp16 engine below
                features = _TextFn.apply(self, text.to(self.positional_embedding.device), tuple(n for n, _ in trainable),
                                         *[p for _, p in trainable])
                return _normalize(features) if normalize else features
        features = self._text().encode_text(text.to(self.positional_embedding.device))
        return _normalize(features) if normalize else features

    # The frozen towers' forwards on a second HIP stream beside the trainable tower's (round 6; what the fused steps do by
    # default, vitlens_hip/step.py `_side_by_side`): on when a tower of this call carries a graph and the others do not.
    overlap_frozen = True

    def _frozen_side_stream(self, image, text, visual_x):
        """-> the side stream when this call has independent frozen and trainable work on a GPU, else None."""
        if not self.overlap_frozen or visual_x is None or (image is None and text is None) or not torch.is_grad_enabled():
            return None
        if not torch.is_tensor(visual_x) or self.positional_embedding.device.type != "cuda":
            return None
        if any(p.requires_grad for p in self.image.parameters()) or not any(p.requires_grad for p in self.visual.parameters()):
            return None
        if any(p.requires_grad for n, p in self.named_parameters() if not n.startswith(("image.", "visual.")) and n != "logit_scale"):
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.positional_embedding.device)
        return self._side

    def forward(self, image=None, text=None, visual_x=None):
        def frozen():
            image_features = None
            if image is not None and image.ndim == 5:
                # [b, t, c, h, w]: normalise every frame's feature, mean over frames, normalise again (model.py:588-600)
                n_img = image.size(1)
                f = self.encode_image(image.reshape(-1, *image.shape[2:]), normalize=True)
                image_features = _normalize(f.reshape(-1, n_img, f.shape[-1]).mean(1).contiguous())
            elif image is not None:
                image_features = self.encode_image(image, normalize=True)
            text_features = self.encode_text(text, normalize=True) if text is not None else None
            return image_features, text_features
        side = self._frozen_side_stream(image, text, visual_x)
        if side is None:
            image_features, text_features = frozen()
            visual_features = self.encode_visual(visual_x, normalize=True) if visual_x is not None else None
        else:
            main = torch.cuda.current_stream()
            ready, done = torch.cuda.Event(), torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)                    # the inputs exist
            with torch.cuda.stream(side), torch.no_grad():
                image_features, text_features = frozen()
                done.record(side)
            visual_features = self.encode_visual(visual_x, normalize=True)
            main.wait_event(done)
            for t in (image_features, text_features):
                if t is not None:
                    t.record_stream(main)             # allocated on the side stream's pool, consumed by the loss on this one
        if self.output_dict:
            return {"image_features": image_features, "text_features": text_features,
                    "visual_features": visual_features, "logit_scale": self.logit_scale.exp()}
        return image_features, text_features, visual_features, self.logit_scale.exp()
