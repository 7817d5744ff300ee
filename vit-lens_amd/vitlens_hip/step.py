"""One tri-modal contrastive training step (the reference's `tri_train_one_epoch` loop body,
training/train.py:115-249) for the depth recipe, entirely on the HIP kernels:

    image tower fwd (frozen) | text tower fwd (frozen) | Lens(depth)->ViT fwd (saved activations)
      -> [one packed RCCL all-gather of img|txt|vis embeddings when world_size > 1]
      -> TriClipLoss (logits GEMM + fused row/col CE) and its gradient w.r.t. the visual embeddings
      -> backward through the visual tower (dX all blocks, dW for the unlocked first n + adapter)
      -> [one flat RCCL all-reduce of the fp32 gradient buffer = DDP mean]
      -> AdamW on the fp32 masters, bf16 operand refresh, logit_scale.clamp_(0, ln 100)

Micro-batching follows the reference's feature-cache scheme (train.py:154-210): every micro-batch sees the
full batch of negatives; here the activations of all micro-batches stay resident in the 288 GB of HBM so
nothing is recomputed.
"""
import math
from typing import Dict, Optional

import torch

from . import ops
from .engine import LensCfg, LensEngine, TextCfg, TextEngine, TowerCfg, VitEngine
from .train import AdamW, DepthLensTrainer


class TorchComm:
    """The two collectives of a step on `torch.distributed` (backend "nccl" = RCCL over xGMI, one process per GPU).
    A step takes any object with these two methods (the tests drive two ranks on one GPU through an in-process one)."""

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        import torch.distributed as dist
        if inp.is_cuda:
            dist.all_gather_into_tensor(out, inp)
        else:          # gloo
            dist.all_gather(list(out.chunk(dist.get_world_size(), dim=0)), inp.contiguous())

    def all_reduce_sum(self, t: torch.Tensor):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def reduce_scatter_sum(self, out: torch.Tensor, inp: torch.Tensor):
        """out [b, E] = this rank's slice of the sum over ranks of inp [W*b, E] (backward of a gather WITH grad)."""
        import torch.distributed as dist
        if inp.is_cuda:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)
        else:          # gloo (CPU tests) has no reduce_scatter
            t = inp.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            r = dist.get_rank()
            out.copy_(t[r * out.shape[0]:(r + 1) * out.shape[0]])

    def all_reduce_sum_async(self, t: torch.Tensor):
        """Start an all-reduce and return a handle with .wait(): on RCCL the collective runs on the communicator's own
        stream behind the kernels already enqueued, so a gradient bucket is reduced under the rest of the backward."""
        import torch.distributed as dist
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


class AbiComm:
    """The same collectives through the library's OWN RCCL entry points (include/vitlens_hip.h: vl_comm_create,
    vl_allgather_embed, vl_reducescatter_grad, vl_allreduce_grad; csrc/vl_comm.cpp) instead of torch.distributed - the calls a
    non-Python host of the C ABI makes, driven from Python so that they are tested with the steps.  fp32 payloads (what the steps
    exchange).  The 128-byte unique id is made on rank 0 and shipped by the caller: `AbiComm.over_torch_distributed()` uses an
    initialised process group (any backend) for that one broadcast; a one-rank communicator needs no shipping."""

    def __init__(self, rank: int = 0, world: int = 1, unique_id: Optional[bytes] = None, device=None):
        import ctypes as C
        from ._lib import check, load_library
        self._lib, self._check, self._C = load_library(), check, C
        self.rank, self.world = rank, world
        if device is not None:
            torch.cuda.set_device(device)
        if unique_id is None:
            if world != 1:
                raise ValueError("AbiComm: ranks > 0 need the unique id rank 0 made (AbiComm.make_unique_id / over_torch_distributed)")
            unique_id = self.make_unique_id()
        if len(unique_id) != 128:
            raise ValueError("AbiComm: the unique id is 128 bytes")
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(self._lib.vl_comm_create(C.byref(self._h), buf, rank, world))
        self._stream = None          # the bucket all-reduces run on their own stream, beside the backward

    @staticmethod
    def make_unique_id() -> bytes:
        import ctypes as C
        from ._lib import check, load_library
        buf = C.create_string_buffer(128)
        check(load_library().vl_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def over_torch_distributed(cls, device=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.make_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0], device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.vl_comm_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown: the library or ctypes may already be gone
            pass

    def _s(self):
        return self._C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _f32(t, name):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError(f"AbiComm.{name}: contiguous fp32 GPU tensors only")
        return t

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        self._f32(out, "all_gather"); inp = self._f32(inp.contiguous(), "all_gather")
        self._check(self._lib.vl_allgather_embed(self._h, inp.data_ptr(), out.data_ptr(), inp.numel(), self._s()))

    def all_reduce_sum(self, t: torch.Tensor):
        self._f32(t, "all_reduce_sum")
        self._check(self._lib.vl_allreduce_grad(self._h, t.data_ptr(), t.numel(), self._s()))

    def reduce_scatter_sum(self, out: torch.Tensor, inp: torch.Tensor):
        self._f32(out, "reduce_scatter_sum"); inp = self._f32(inp.contiguous(), "reduce_scatter_sum")
        self._check(self._lib.vl_reducescatter_grad(self._h, inp.data_ptr(), out.data_ptr(), out.numel(), self._s()))

    def all_reduce_sum_async(self, t: torch.Tensor):
        """The bucket's all-reduce on the communicator's own stream behind what the launch stream has enqueued so far;
        .wait() makes the launch stream wait for it."""
        self._f32(t, "all_reduce_sum_async")
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(main)
        self._stream.wait_event(ready)
        with torch.cuda.stream(self._stream):
            self.all_reduce_sum(t)
            done.record(self._stream)

        class _Handle:
            def wait(self_h):
                torch.cuda.current_stream().wait_event(done)
        return _Handle()


# ------------------------------------------------------------------------------------------------ checkpointing
def _master_from_sd(name: str, sd, like: torch.Tensor) -> torch.Tensor:
    """Value of master `name` (kernel layout) from a reference-layout state_dict."""
    from .engine import _interleave_geglu, conv_weight_as_gemm
    dev = like.device
    f32 = lambda k: sd[k].detach().float().to(dev).clone()
    if name.endswith("conv1.weight_gemm"):                       # conv-as-GEMM, K zero-padded to 64
        return conv_weight_as_gemm(sd[name[:-5]], dev, torch.float32)
    if name.endswith("net.0.weight_il") or name.endswith("net.0.bias_il"):      # GEGLU rows interleaved (a_j, gate_j)
        base = name[:name.rindex("net.0.")] + "net.0."
        w, b = _interleave_geglu(f32(base + "weight"), f32(base + "bias"))
        return (w if name.endswith("weight_il") else b).contiguous()
    if name.endswith("fn.to_qkv.weight"):                        # self-attention: [to_q ; to_kv]
        return torch.cat([f32(name.replace("to_qkv", "to_q")), f32(name.replace("to_qkv", "to_kv"))], 0).contiguous()
    return f32(name).reshape(like.shape).contiguous()


def _master_to_sd(name: str, m: torch.Tensor, base_sd, cfg) -> Dict[str, torch.Tensor]:
    """Reference-layout entries produced by master `name`."""
    from .train import deinterleave_geglu
    if name.endswith("conv1.weight_gemm"):
        ref = base_sd[name[:-5]]
        return {name[:-5]: m[:, :ref[0].numel()].reshape(ref.shape)}
    if name.endswith("_il"):
        return {name[:-3]: deinterleave_geglu(m)}
    if name.endswith("fn.to_qkv.weight"):
        inner = cfg.latent_heads * cfg.latent_dim_head
        return {name.replace("to_qkv", "to_q"): m[:inner], name.replace("to_qkv", "to_kv"): m[inner:]}
    return {name: m.reshape(base_sd[name].shape)}


class _StepState:
    """state_dict / load_state_dict of a fused training step in the REFERENCE's names and layouts (a TriCLIP state_dict:
    trained tensors from the fp32 masters, everything else as it was given), plus the AdamW moments - so training can be
    checkpointed, resumed, and handed back to `TriCLIP.load_state_dict` (training/train.py checkpoints `model.state_dict()`
    and `optimizer.state_dict()`)."""

    def _init_host(self, sd, device, micro_batch, rank, world_size, comm=None, local_loss=False, gather_with_grad=False,
                   force_comm=False, overlap_frozen=False, overlap_backward=True):
        """Everything of a step that is HOST state - flags, the communicator, the (still empty apart from logit_scale) master
        table, the gradient-bucket bookkeeping - and nothing that touches an engine or a kernel.  Every step's `__init__`
        runs this first and then its `_build()` (engines, masters of the trainable set); tests/test_step_gloo.py drives the
        REAL `__init__` of the product classes on CPU + gloo with `_build` / `_trainer` / `_refresh_operands` overridden by
        linear stand-in towers, so a field added here can never be missing from the object under test."""
        self.dev, self.mb, self.rank, self.world = torch.device(device), micro_batch, rank, world_size
        # the multi-rank path (packed all-gather, bucketed async all-reduce, optional reduce-scatter) runs when there are
        # peers - or when asked for on ONE rank, so that the RCCL calls execute on a single GPU (tests, bench --force-dist)
        self._force_comm = bool(force_comm)
        self.comm = comm or TorchComm()
        self.local_loss, self.gather_with_grad = local_loss, gather_with_grad
        # the frozen towers' forwards on a second HIP stream beside the trainable tower's forward: a request;
        # `_overlap_active` says whether this device can honour it (a CPU device - the gloo tests - runs the serial order,
        # the same arithmetic)
        self.overlap_frozen = bool(overlap_frozen)
        # with it: the two halves of a step's micro-batches run their BACKWARD on the two streams too (_backward_all)
        self.overlap_backward = bool(overlap_backward)
        self._side = None
        self._base_sd = {k: v.detach() for k, v in sd.items()}
        self.logit_scale = sd["logit_scale"].detach().float().reshape(1).to(device).clone()
        self.masters: Dict[str, torch.Tensor] = {"logit_scale": self.logit_scale}
        self.bf16_targets = {}   # depth step: master name -> (block, operand key)
        self.refresh = []        # Perceiver steps: (master name, forward bf16 tensor, key path of the transposed copy in trainer.perc.wT)
        self.trainers = []       # one activation store per micro-batch (created lazily)
        self.flat_grad, self.grads = None, {}
        self._pending, self._reduced_upto, self._reduce_done = [], None, False

    @property
    def dist(self) -> bool:
        """The multi-rank code path is taken: there are peers, or `force_comm` asked for it on one rank."""
        return self.world > 1 or self._force_comm

    @property
    def _overlap_active(self) -> bool:
        return self.overlap_frozen and self.dev.type == "cuda"

    def _side_by_side(self, beside, here):
        """Run two closures of independent work: `beside()` on the step's second HIP stream, `here()` on the launch stream,
        joined by events on both ends (one after the other on a CPU device or with `overlap_frozen` off).  Two callers: the
        forward (`beside` = the locked towers, forward only, results into buffers the caller allocated; `here` = the Lens
        tower with saved activations) and the backward (`_backward_all`: the two halves of the micro-batches, each into its
        own gradient buffer).  One closure's low-power phases (attention, LayerNorm, leftover rows) sit beside the other's
        GEMMs on a board that is power-limited in its GEMM phases (-1.35 % per C3 step for the forward, -3.2 % with the
        backward; bit-equal results: tests/test_hip_train.py)."""
        frozen, trainable = beside, here
        if not self._overlap_active:
            frozen(); trainable()
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        main = torch.cuda.current_stream(self.dev)
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(main)
        self._side.wait_event(ready)                  # inputs and output buffers exist
        with torch.cuda.stream(self._side):
            frozen()
            done.record(self._side)
        trainable()
        main.wait_event(done)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {k: v.detach().clone() for k, v in self._base_sd.items()}
        cfg = getattr(self.lens, "lens", None)
        for name, m in self.masters.items():
            for k, v in _master_to_sd(name, m.detach(), self._base_sd, cfg).items():
                out[k] = v.to(device=self._base_sd[k].device, dtype=self._base_sd[k].dtype).clone()
        if cfg is not None and getattr(cfg, "weight_tie_layers", False):
            # the reference's state_dict repeats the shared modules under every tied layer index (perceiver.py:249-254)
            P1 = "visual.perceiver.layers.1."
            for k in [k for k in out if k.startswith(P1)]:
                for li in range(2, cfg.depth):
                    out[f"visual.perceiver.layers.{li}." + k[len(P1):]] = out[k].clone()
        tok = getattr(self, "tok", None)
        if tok is not None:                                      # BatchNorm running statistics of the PointTokenizer
            for k, (rm, rv) in tok.running.items():
                out["visual.visual_adapter." + k + ".running_mean"] = rm.detach().cpu().clone()
                out["visual.visual_adapter." + k + ".running_var"] = rv.detach().cpu().clone()
        return out

    def optimizer_state_dict(self):
        return {"step": self.opt.t, "exp_avg": {k: v.detach().clone() for k, v in self.opt.m.items()},
                "exp_avg_sq": {k: v.detach().clone() for k, v in self.opt.v.items()}}

    def load_state_dict(self, sd):
        """Re-read every TRAINABLE tensor from a reference-layout state_dict into the masters (in place: the kernels keep
        reading the same buffers) and refresh the bf16 operands.  Frozen towers are fixed at construction."""
        for name, m in self.masters.items():
            m.copy_(_master_from_sd(name, sd, m))
        tok = getattr(self, "tok", None)
        if tok is not None:
            for k, (rm, rv) in tok.running.items():
                rm.copy_(sd["visual.visual_adapter." + k + ".running_mean"].float()); rv.copy_(sd["visual.visual_adapter." + k + ".running_var"].float())
        if not self.trainers:
            self._trainer(0)
        self._refresh_operands()
        self._base_sd = {k: v.detach() for k, v in sd.items()}

    def load_optimizer_state_dict(self, st):
        self.opt.t = int(st["step"])
        for k in self.opt.m:
            self.opt.m[k].copy_(st["exp_avg"][k]); self.opt.v[k].copy_(st["exp_avg_sq"][k])


# ------------------------------------------------------------------------------------------------ loss core
# logits of this many elements or more are never materialised whole: the pair loss runs in row blocks (268 MB of f32)
LOGITS_CHUNK_ELEMS = 1 << 26
LOGITS_CHUNK_ROWS = 2048


def _chunk_rows(R: int, Cn: int, chunk_rows: Optional[int]) -> int:
    """Rows per block of the pair loss: 0 = whole matrix at once."""
    if chunk_rows is None:
        chunk_rows = LOGITS_CHUNK_ROWS if R * Cn >= LOGITS_CHUNK_ELEMS else 0
    return 0 if chunk_rows <= 0 or chunk_rows >= R else int(chunk_rows)


def _is_dev_scale(scale) -> bool:
    return torch.is_tensor(scale)


def pair_forward(x, y, scale, label_off: int = 0, w_row: float = 0.5, w_col: float = 0.5,
                 chunk_rows: Optional[int] = None):
    """loss contribution w_row*CE(scale*x y^T) + w_col*CE(columns); returns (loss[1] tensor, ctx).

    scale: a Python float (the temperature itself), or - round 4 - the LEARNABLE LOG-temperature as a 1-element f32 tensor
    on the device (`logit_scale` as the model holds it, model.py:619): x is multiplied by exp(logit_scale) on the device
    (vl_scale_exp_f32) and the logits GEMM runs with alpha = 1, so the step never reads the scalar on the host and the loss
    section is capturable in a hipGraph.  In that mode pair_backward's third result is d/d(logit_scale) (the log-domain
    parameter) directly: sum(G * logits) - no division by the scale and re-multiplication.

    Row-blocked mode (chunk_rows, or automatically for B_glob^2 above LOGITS_CHUNK_ELEMS): the B_glob x B_glob logits
    are never held whole.  Forward = per block of rows: logits GEMM, exact row log-sum-exp, the block's column
    log-sum-exp; the column statistics of the blocks are merged by one logsumexp over [blocks, C].  Backward
    recomputes each block's logits (training/train.py:154-210 reaches batch 2048 by re-running the towers per
    accumulation step; here only one thin GEMM is re-run)."""
    R, Cn = x.shape[0], y.shape[0]
    rb = _chunk_rows(R, Cn, chunk_rows)
    dev_scale = _is_dev_scale(scale)
    alpha = 1.0 if dev_scale else scale
    xb = ops.split_bf16x3(ops.scale_exp(x.contiguous(), scale) if dev_scale else x, 0)
    yb = ops.split_bf16x3(y, 1)
    if rb == 0:
        logits = ops.logits_gemm(xb, yb, alpha)
        row_lse, col_lse, diag = ops.ce_stats(logits, label_off, want_cols=(w_col != 0.0))
        loss = torch.zeros(1, device=x.device, dtype=torch.float32)
        ops.ce_loss_accum(loss, row_lse if w_row != 0.0 else None, col_lse, diag, R, Cn, label_off, w_row, w_col)
        return loss, (x, y, logits, row_lse, col_lse, label_off, w_row, w_col, scale, 0, None, None)
    row_lse = torch.empty(R, device=x.device); diag = torch.empty(R, device=x.device)
    col_parts = []
    for r0 in range(0, R, rb):
        r1 = min(R, r0 + rb)
        lg = ops.logits_gemm(xb[r0:r1], yb, alpha)
        rl, cl, dg = ops.ce_stats(lg, label_off + r0, want_cols=(w_col != 0.0))
        row_lse[r0:r1] = rl; diag[r0:r1] = dg
        if cl is not None:
            col_parts.append(cl)
        del lg
    col_lse = torch.logsumexp(torch.stack(col_parts), dim=0) if col_parts else None
    loss = torch.zeros(1, device=x.device, dtype=torch.float32)
    ops.ce_loss_accum(loss, row_lse if w_row != 0.0 else None, col_lse, diag, R, Cn, label_off, w_row, w_col)
    return loss, (x, y, None, row_lse, col_lse, label_off, w_row, w_col, scale, rb, xb, yb)


def pair_backward(ctx, g: float = 1.0, need_dx=True, need_dy=True):
    """-> (dx, dy, dscale).  dscale = dL/d(scale) for a float scale, dL/d(logit_scale) (log domain) for a device scale."""
    x, y, logits, row_lse, col_lse, label_off, w_row, w_col, scale, rb, xb, yb = ctx
    dscale = torch.zeros(1, device=x.device, dtype=torch.float32)
    dev_scale = _is_dev_scale(scale)
    # vl_ce_grad accumulates sum(G * logits) / logit_scale: with 1.0 that IS d/d(log-scale)
    alpha, cscale = (1.0, 1.0) if dev_scale else (scale, scale)
    fin = (lambda t: ops.scale_exp(t, scale, out=t)) if dev_scale else (lambda t: t)
    if rb == 0:
        G, GT = ops.ce_grad(logits, row_lse if w_row != 0.0 else None, col_lse, label_off, w_row * g, w_col * g, cscale,
                            dscale, need_g=need_dx, need_gt=need_dy)
        dx = dy = None
        if need_dx:
            dx = fin(ops.gemm(G, ops.transpose_to_bf16(y, ldo=G.shape[1]), None, epi=ops.EPI_F32, alpha=alpha))
        if need_dy:
            dy = fin(ops.gemm(GT, ops.transpose_to_bf16(x, ldo=GT.shape[1]), None, epi=ops.EPI_F32, alpha=alpha))
        return dx, dy, dscale
    R, Cn = x.shape[0], y.shape[0]
    dx = torch.empty(R, x.shape[1], device=x.device) if need_dx else None
    dy = torch.zeros(Cn, y.shape[1], device=x.device) if need_dy else None
    yt = None
    for r0 in range(0, R, rb):
        r1 = min(R, r0 + rb)
        f = (r1 - r0) / R                      # the kernels average over the rows they are given: re-weight to the global mean
        lg = ops.logits_gemm(xb[r0:r1], yb, alpha)
        G, GT = ops.ce_grad(lg, row_lse[r0:r1] if w_row != 0.0 else None, col_lse, label_off + r0, w_row * g * f, w_col * g * f,
                            cscale, dscale, need_g=need_dx, need_gt=need_dy)
        if need_dx:
            if yt is None:
                yt = ops.transpose_to_bf16(y, ldo=G.shape[1])
            ops.gemm(G, yt, None, out=dx[r0:r1], epi=ops.EPI_F32, alpha=alpha)
        if need_dy:
            xt = ops.transpose_to_bf16(x[r0:r1].contiguous(), ldo=GT.shape[1])
            ops.gemm(GT, xt, None, out=dy, res=dy, epi=ops.EPI_RES_F32, alpha=alpha)       # dy += scale * G_b^T x_b
        del lg, G, GT
    return (fin(dx) if need_dx else None), (fin(dy) if need_dy else None), dscale


def pair_loss_and_grads(comm, rank: int, world: int, xl, yl, ax, ay, scale, local_loss: bool = False,
                        gather_with_grad: bool = False, need_x: bool = True, need_y: bool = True, dist: Optional[bool] = None):
    """One (x, y) pair of ClipLossGeneral / TriClipLoss over the global batch (loss.py:116-138, 293-308) as rank `rank`
    computes it, and the gradients that arrive at THIS rank's features.

    xl, yl: local unit features [b, E]; ax, ay: their rank-major gathers [W*b, E] (= xl, yl at world 1).
      local_loss=False: full B_glob x B_glob logits on every rank (rows and columns CE).
      local_loss=True : b x B_glob logits for the rank's own rows of both directions, labels offset by rank*b.
      gather_with_grad=False: peers are constants; the own slot of the gather is differentiable unless local_loss
                              (gather_features re-inserts the local tensor only then, loss.py:71-74).
      gather_with_grad=True : the gather is differentiable everywhere -> backward = reduce-scatter (sum over ranks).
    dist: run the multi-rank code path (default: world > 1).  True at world 1 = every collective of the path executes on a
    one-rank communicator (`force_comm` of the steps: the RCCL calls, their streams and their buffer contracts are exercised
    on one GPU; the values are those of the single-rank path).
    Returns (loss[1], dxl | None, dyl | None, dscale[1])."""
    b = xl.shape[0]
    dist = world > 1 if dist is None else dist
    if dist and local_loss:
        l1, c1 = pair_forward(xl, ay, scale, label_off=rank * b, w_row=0.5, w_col=0.0)
        l2, c2 = pair_forward(yl, ax, scale, label_off=rank * b, w_row=0.5, w_col=0.0)
        dxl, d_ay, ds1 = pair_backward(c1, need_dx=need_x, need_dy=need_y and gather_with_grad)
        dyl, d_ax, ds2 = pair_backward(c2, need_dx=need_y, need_dy=need_x and gather_with_grad)
        loss, ds = l1 + l2, ds1 + ds2
    else:
        loss, c = pair_forward(ax, ay, scale)
        d_ax, d_ay, ds = pair_backward(c, need_dx=need_x, need_dy=need_y)
        dxl = dyl = None

    def fold(dl, d_all):
        if d_all is None:
            return dl
        if not dist:
            g = d_all
        elif gather_with_grad:
            g = torch.empty(b, d_all.shape[1], device=d_all.device, dtype=d_all.dtype)
            comm.reduce_scatter_sum(g, d_all.contiguous())
        else:
            g = d_all[rank * b:(rank + 1) * b].contiguous()
        return g if dl is None else dl + g
    return loss, (fold(dxl, d_ax) if need_x else None), (fold(dyl, d_ay) if need_y else None), ds


class TriModalDepthStep(_StepState):
    def __init__(self, sd: Dict[str, torch.Tensor], tower: TowerCfg, text: TextCfg, device, micro_batch: int = 256,
                 unlock_first_n: int = 4, lr: float = 5e-4, betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.2,
                 rank: int = 0, world_size: int = 1, gemm_cfg: int = -1, comm=None, frozen_res_dtype=torch.float32,
                 local_loss: bool = False, gather_with_grad: bool = False, train_res_dtype=torch.float32,
                 grad_checkpointing: bool = False, force_comm: bool = False, text_wsplit: Optional[bool] = None, text_arith: str = "f16",
                 overlap_frozen: bool = True, overlap_backward: bool = True):
        """overlap_frozen (default ON since round 6): the image / text towers' forwards run on a second HIP stream beside
        the trainable tower's forward (`_side_by_side`); results are bit-identical to the serial order."""
        self._init_host(sd, device, micro_batch, rank, world_size, comm, local_loss, gather_with_grad, force_comm, overlap_frozen,
                        overlap_backward)
        self.grad_checkpointing = bool(grad_checkpointing)      # block recompute in the trainable tower (transformer.py:366-368)
        self.unlock_first_n = unlock_first_n
        self._build(sd, tower, text, gemm_cfg=gemm_cfg, frozen_res_dtype=frozen_res_dtype, train_res_dtype=train_res_dtype,
                    text_wsplit=text_wsplit, text_arith=text_arith)
        self.opt = AdamW(self.masters, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def _build(self, sd, tower, text, gemm_cfg=-1, frozen_res_dtype=torch.float32, train_res_dtype=torch.float32,
               text_wsplit=None, text_arith="f16"):
        """Engines and the fp32 masters of the trainable set (reference lock recipe: adapter + first n blocks + logit_scale)."""
        device, unlock_first_n = self.dev, self.unlock_first_n
        # frozen towers: forward only; their residual stream may be kept in bf16 (= the reference's autocast)
        self.image = VitEngine(sd, "image.", tower, device, gemm_cfg=gemm_cfg, res_dtype=frozen_res_dtype)
        self.text = TextEngine(sd, text, device, gemm_cfg=gemm_cfg, res_dtype=frozen_res_dtype, wsplit=text_wsplit, arith=text_arith)
        # trainable tower: residual stream AND residual-gradient stream in `train_res_dtype` (bf16 = the reference's amp_bf16)
        self.lens = LensEngine(sd, "visual.", tower, LensCfg(modality="depth", perceiver_identity=True), device, gemm_cfg=gemm_cfg,
                               res_dtype=train_res_dtype)
        eng = self.lens.vit
        for l in range(unlock_first_n):
            p = f"visual.transformer.resblocks.{l}."
            w = eng.blocks[l]
            # a block whose LayerNorm parameters and weights move every step carries no LayerNorm-folded operands: they would
            # go stale at the first AdamW step, and `run_blocks` / the trainers decide the folding per block by their presence
            w.pop("in_f", None); w.pop("fc_f", None)
            for nm, key in (("attn.in_proj_weight", "in_w"), ("attn.out_proj.weight", "out_w"), ("mlp.c_fc.weight", "fc_w"),
                            ("mlp.c_proj.weight", "proj_w")):
                self.masters[p + nm] = sd[p + nm].detach().float().to(device).contiguous().clone()      # never an alias of the caller's tensor
                self.bf16_targets[p + nm] = (l, key)
            for nm, key in (("ln_1.weight", "ln1_w"), ("ln_1.bias", "ln1_b"), ("ln_2.weight", "ln2_w"), ("ln_2.bias", "ln2_b"),
                            ("attn.in_proj_bias", "in_b"), ("attn.out_proj.bias", "out_b"), ("mlp.c_fc.bias", "fc_b"),
                            ("mlp.c_proj.bias", "proj_b")):
                self.masters[p + nm] = w[key]                   # f32 tensors the kernels read directly
        self.masters["visual.visual_adapter.pos_emb"] = self.lens.adapter_pos
        from .engine import conv_weight_as_gemm
        self.masters["visual.visual_adapter.conv1.weight_gemm"] = conv_weight_as_gemm(       # from the fp32 weight, not the bf16 operand
            sd["visual.visual_adapter.conv1.weight"], device, torch.float32)

    # -------------------------------------------------------------------------------------------
    def _trainer(self, i):
        while len(self.trainers) <= i:
            t = DepthLensTrainer(self.lens, tower_kw=dict(train_blocks=range(self.unlock_first_n), checkpoint=self.grad_checkpointing))
            if self.trainers:                      # share weight transposes + gradient buffers across micro-batches
                t.tower.wT = self.trainers[0].tower.wT
                t.tower.proj = self.trainers[0].tower.proj
                t.tower.grads = self.trainers[0].tower.grads
            self.trainers.append(t)
        return self.trainers[i]

    def _alloc_flat_grads(self):
        """All gradient buffers are views into ONE flat fp32 tensor -> a single all-reduce per step."""
        al = lambda n: (n + 3) // 4 * 4                       # every view starts 16-byte aligned
        self.flat_grad = torch.zeros(sum(al(v.numel()) for v in self.masters.values()), device=self.dev, dtype=torch.float32)
        off = 0
        for k, v in self.masters.items():
            self.grads[k] = self.flat_grad[off:off + v.numel()].view(v.shape)
            off += al(v.numel())
        # a second buffer of the same layout: the first half of a step's micro-batches accumulates here (forward_backward)
        self.flat_grad_b = torch.zeros_like(self.flat_grad)
        self.grads_b, off = {}, 0
        for k, v in self.masters.items():
            self.grads_b[k] = self.flat_grad_b[off:off + v.numel()].view(v.shape)
            off += al(v.numel())
        for t in self.trainers:
            t.tower.grads = self.grads

    def _refresh_operands(self):
        eng = self.lens.vit
        if not self.trainers:
            self._trainer(0)
        for name, (l, key) in self.bf16_targets.items():
            m = self.masters[name]
            ops.cast_bf16(m, out=eng.blocks[l][key])
            ops.transpose_to_bf16(m, ldo=m.shape[0], out=self.trainers[0].tower.wT[l][key])
        ops.cast_bf16(self.masters["visual.visual_adapter.conv1.weight_gemm"], out=self.lens.conv_w)

    # -------------------------------------------------------------------------------------------
    def step(self, images: torch.Tensor, texts: torch.Tensor, depths: torch.Tensor) -> torch.Tensor:
        loss = self.forward_backward(images, texts, depths)
        self.optimizer_step()
        return loss

    def finish_reduce(self):
        """Bring `flat_grad` / `grads` to ONE state: the sum over ranks of every gradient, all collectives complete.

        Between `forward_backward()` and this call the buffer is MIXED: the unlocked blocks' buckets were all-reduced
        during the last micro-batch's backward (possibly still in flight on the collective's stream) while logit_scale
        and the adapter - produced last - are still rank-local.  Anything that reads or rescales the gradients in
        between (clipping, norm logging, accumulation over several forward_backward calls) calls this first;
        `optimizer_step()` does.  Idempotent; a no-op on one rank.  The values are SUMS: the 1/world of DDP's mean is
        applied inside the optimizer step (`grad_scale`)."""
        if not self.dist or self._reduce_done:
            return
        for h in self._pending:
            h.wait()
        self._pending = []
        if self._reduced_upto is None:
            self.comm.all_reduce_sum(self.flat_grad)
        else:
            lo, hi = self._rest_ranges()
            for a, b in ((0, lo), (hi, self.flat_grad.numel())):
                if b > a:
                    self.comm.all_reduce_sum(self.flat_grad[a:b])
        self._reduced_upto = None
        self._reduce_done = True

    def reduced_grads(self):
        """The gradients by master name after `finish_reduce()` (sums over ranks)."""
        self.finish_reduce()
        return self.grads

    def optimizer_step(self):
        if self.dist:
            # DDP semantics: mean of per-rank gradients.  Block buckets were started during the last micro-batch's backward
            # (reverse layer order); what is left - logit_scale and the adapter, produced last - goes in one more call.
            self.finish_reduce()
            self.opt.step(self.grads, grad_scale=1.0 / self.world)
        else:
            self.opt.step(self.grads)
        self._refresh_operands()
        ops.clamp_scalar(self.logit_scale, 0.0, math.log(100.0))

    # ---- gradient buckets: the masters of one block are contiguous in the flat buffer ----
    def _block_range(self, l):
        p = f"visual.transformer.resblocks.{l}."
        names = [k for k in self.masters if k.startswith(p)]
        base = self.flat_grad.data_ptr()
        lo = min((self.grads[k].data_ptr() - base) // 4 for k in names)
        hi = max((self.grads[k].data_ptr() - base) // 4 + self.grads[k].numel() for k in names)
        return lo, min((hi + 3) // 4 * 4, self.flat_grad.numel())          # incl. the 16-byte alignment pad of the last view

    def _rest_ranges(self):
        rs = [self._block_range(l) for l in range(self.unlock_first_n)]
        return min(r[0] for r in rs), max(r[1] for r in rs)

    def _start_block_reduce(self, l):
        if self.dist and hasattr(self.comm, "all_reduce_sum_async"):
            lo, hi = self._block_range(l)
            self._pending.append(self.comm.all_reduce_sum_async(self.flat_grad[lo:hi]))
            self._reduced_upto = l

    def forward_backward(self, images: torch.Tensor, texts: torch.Tensor, depths: torch.Tensor) -> torch.Tensor:
        B = images.shape[0]
        mb = min(self.mb, B)
        assert B % mb == 0, "per-GPU batch must be a multiple of the micro-batch"
        nmb = B // mb
        if self.flat_grad is None:
            for i in range(nmb):
                self._trainer(i)
            self._alloc_flat_grads()
        for h in self._pending:          # (a forward_backward without optimizer_step: finish what was started)
            h.wait()
        self._pending, self._reduced_upto, self._reduce_done = [], None, False
        self.flat_grad.zero_()
        E = self.image.cfg.embed_dim
        fi = torch.empty(B, E, device=self.dev); ft = torch.empty(B, E, device=self.dev)
        fv = torch.empty(B, E, device=self.dev); vraw = torch.empty(B, E, device=self.dev)
        vnorm = torch.empty(B, device=self.dev)
        # the frozen text tower sees the whole per-GPU batch in one pass: 77-token sequences give a micro-batch only 77 row
        # tiles (one uneven round of the persistent GEMM, 600-900 TF/s); four times the rows run whole rounds
        def frozen():
            ops.l2_normalize(self.text.encode_text(texts), out=ft)
            for i in range(nmb):
                s = slice(i * mb, (i + 1) * mb)
                ops.l2_normalize(self.image.encode_image(images[s]), out=fi[s])

        def trainable():
            for i in range(nmb):
                s = slice(i * mb, (i + 1) * mb)
                vraw[s] = self._trainer(i).forward(depths[s])
        self._side_by_side(frozen, trainable)
        ops.l2_normalize(vraw, out=fv, norms=vnorm)
        scale = self.logit_scale          # the log-temperature, on the device: exp() is applied inside the loss section
        if self.dist:
            packed = torch.cat([fi, ft, fv], dim=1)
            allp = torch.empty(self.world * B, 3 * E, device=self.dev)
            self.comm.all_gather(allp, packed)                    # ONE exchange: [b, 3*768] per rank over xGMI
            ai, at, av = [t.contiguous() for t in allp.split(E, dim=1)]
        else:
            ai, at, av = fi, ft, fv
        kw = dict(local_loss=self.local_loss, gather_with_grad=self.gather_with_grad, need_x=False, dist=self.dist)
        l1, _, dv1, ds1 = pair_loss_and_grads(self.comm, self.rank, self.world, fi, fv, ai, av, scale, **kw)
        l2, _, dv2, ds2 = pair_loss_and_grads(self.comm, self.rank, self.world, ft, fv, at, av, scale, **kw)
        loss = l1 + l2
        dvraw = ops.l2_normalize_bwd(fv, dv1 + dv2, vnorm)
        self._backward_all(dvraw, nmb, mb)
        # logit_scale is exp()'d in forward (model.py:619); with the device-side scale pair_backward returns d/d(log-scale)
        self.grads["logit_scale"] += ds1 + ds2
        return loss

    def _backward_all(self, dvraw, nmb, mb):
        """Backward of every micro-batch; parameter gradients accumulate over them.

        With two or more micro-batches the FIRST half accumulates into a second gradient buffer (`flat_grad_b`) and the second
        half into `flat_grad`; a block's bucket is merged (and, across ranks, its all-reduce started) when the LAST micro-batch
        of the second half has passed the block, the rest (adapter, position table) at the end.  That makes the two halves
        independent streams of work: with `overlap_frozen` on a GPU the first half runs on the second HIP stream beside the
        second half - the backward is a chain of chip-filling GEMMs with latency-bound kernels between them (fused attention
        backward, LayerNorm backward, leftover rows, the drain of every persistent launch), exactly what the forward's two
        towers fill for each other.  The arithmetic - which products are added in which order - is the same on one stream
        and on two: results are bit-identical (tests/test_hip_train.py).  Reference: loss.backward() over the micro-batches
        of the accumulation loop, training/train.py:154-210 (gradients of all micro-batches summed into .grad)."""
        dist_cb = self.dist and self.unlock_first_n > 0
        dv = lambda i: dvraw[i * mb:(i + 1) * mb].contiguous()
        half = nmb // 2
        for i in range(nmb):
            self._trainer(i).tower.grads = self.grads_b if i < half else self.grads
        if half == 0:
            self._trainer(0).backward(dv(0), self._start_block_reduce if dist_cb else None)
            return
        self.flat_grad_b.zero_()
        two_streams = self._overlap_active and self.overlap_backward
        passed = {}          # block -> event on the first half's stream: its last micro-batch has passed the block

        def first_done(l):
            if two_streams:
                passed[l] = torch.cuda.Event()
                passed[l].record()

        def second_done(l):
            if two_streams:
                torch.cuda.current_stream(self.dev).wait_event(passed[l])
            lo, hi = self._block_range(l)
            ops.axpy(self.flat_grad[lo:hi], self.flat_grad_b[lo:hi])
            if dist_cb:
                self._start_block_reduce(l)

        def first_half():
            for i in range(half):
                self._trainer(i).backward(dv(i), first_done if i == half - 1 else None)

        def second_half():
            for i in range(half, nmb):
                self._trainer(i).backward(dv(i), second_done if i == nmb - 1 else None)
        if two_streams:
            self._side_by_side(first_half, second_half)      # (first closure on the second stream, joined at the end)
        else:
            first_half(); second_half()
        n = self.flat_grad.numel()
        lo, hi = self._rest_ranges() if self.unlock_first_n > 0 else (0, 0)
        for a, b in ((0, lo), (hi, n)):
            if b > a:
                ops.axpy(self.flat_grad[a:b], self.flat_grad_b[a:b])


class _PerceiverLensStep(_StepState):
    """Shared plumbing of the steps whose trainable part is a Lens (tokenizer + Perceiver) in front of a locked ViT:
    fp32 masters of the Perceiver under the reference's parameter names, one flat fp32 gradient buffer (a single
    all-reduce per step = DDP's mean of per-rank gradients), AdamW, bf16 operand refresh, logit-scale clamp."""

    def _collect_perceiver(self, sd):
        pe, P = self.lens.perceiver, "visual.perceiver."
        f32 = lambda k: sd[k].detach().float().to(self.dev).contiguous().clone()
        self.masters[P + "latents"] = pe.latents
        for li, lay in enumerate(pe.layers):
            if li >= 2 and lay is pe.layers[1]:
                continue      # tied to layer 1 (perceiver_weight_tie_layers): same tensors, same masters, summed gradients
            q = f"{P}layers.{li}."
            self._attn(q + "0.", lay["x_attn"], lay["x_norm"], (li, "x"), sd, f32, ctx_norm=lay["x_norm_ctx"])
            self._ff(q + "1.", lay["x_ff"], lay["x_ff_norm"], (li, "xff"), sd, f32)
            for sj, sl in enumerate(lay["selfs"]):
                r = f"{q}2.{sj}."
                self._attn(r + "0.", sl["attn"], sl["norm"], (li, "selfs", sj), sd, f32)
                self._ff(r + "1.", sl["ff"], sl["ff_norm"], (li, "selfs", sj), sd, f32)

    def _attn(self, p, a, norm, path, sd, f32, ctx_norm=None):
        self.masters[p + "norm.weight"], self.masters[p + "norm.bias"] = norm
        if ctx_norm is not None:
            self.masters[p + "norm_context.weight"], self.masters[p + "norm_context.bias"] = ctx_norm
            self.masters[p + "fn.to_q.weight"] = f32(p + "fn.to_q.weight"); self.refresh.append((p + "fn.to_q.weight", a["q_w"], path + ("q",)))
            self.masters[p + "fn.to_kv.weight"] = f32(p + "fn.to_kv.weight"); self.refresh.append((p + "fn.to_kv.weight", a["kv_w"], path + ("kv",)))
        else:
            self.masters[p + "fn.to_qkv.weight"] = torch.cat([f32(p + "fn.to_q.weight"), f32(p + "fn.to_kv.weight")], 0)
            self.refresh.append((p + "fn.to_qkv.weight", a["qkv_w"], path + ("qkv",)))
        self.masters[p + "fn.to_out.weight"] = f32(p + "fn.to_out.weight"); self.refresh.append((p + "fn.to_out.weight", a["to_out_w"], path + ("out",)))
        self.masters[p + "fn.to_out.bias"] = a["to_out_b"]

    def _ff(self, p, ff, norm, path, sd, f32):
        from .engine import _interleave_geglu
        self.masters[p + "norm.weight"], self.masters[p + "norm.bias"] = norm
        w0, _ = _interleave_geglu(f32(p + "fn.net.0.weight"), f32(p + "fn.net.0.bias"))
        self.masters[p + "fn.net.0.weight_il"] = w0.contiguous(); self.refresh.append((p + "fn.net.0.weight_il", ff["w0"], path + ("w0",)))
        self.masters[p + "fn.net.0.bias_il"] = ff["b0"]
        self.masters[p + "fn.net.2.weight"] = f32(p + "fn.net.2.weight"); self.refresh.append((p + "fn.net.2.weight", ff["w2"], path + ("w2",)))
        self.masters[p + "fn.net.2.bias"] = ff["b2"]

    def _trainer(self, i):
        while len(self.trainers) <= i:
            t = self._mk()
            if self.trainers:
                t.tower.wT, t.tower.proj = self.trainers[0].tower.wT, self.trainers[0].tower.proj
                t.perc.wT = self.trainers[0].perc.wT
            self.trainers.append(t)
        return self.trainers[i]

    def _bind_grads(self, t, grads=None):
        g = self.grads if grads is None else grads
        t.tower.grads = g; t.perc.grads = g

    def _alloc_flat_grads(self):
        al = lambda n: (n + 3) // 4 * 4                       # every view starts 16-byte aligned
        self.flat_grad = torch.zeros(sum(al(v.numel()) for v in self.masters.values()), device=self.dev, dtype=torch.float32)
        self.flat_grad_b = torch.zeros_like(self.flat_grad)  # the first half of a step's micro-batches accumulates here (_backward_all)
        self.grads_b, off = {}, 0
        for k, v in self.masters.items():
            self.grads[k] = self.flat_grad[off:off + v.numel()].view(v.shape)
            self.grads_b[k] = self.flat_grad_b[off:off + v.numel()].view(v.shape)
            off += al(v.numel())
        for t in self.trainers:
            self._bind_grads(t)

    def reference_named_grads(self):
        """The step's gradients (all micro-batches, after `forward_backward`) under the reference's parameter names / layouts."""
        return self.trainers[0].perc.reference_named_grads(self.grads)

    def _backward_all(self, dvraw, nmb, mb):
        """Backward of every micro-batch (TriModalDepthStep._backward_all): the first half of the micro-batches accumulates into
        `flat_grad_b`, the second into `flat_grad`, one merge at the end; with `overlap_frozen` on a GPU the two halves run on
        two HIP streams.  Same arithmetic on one stream and on two.  Not with SyncBatchNorm (its backward exchanges data on the
        communicator from inside the micro-batch: one stream)."""
        dv = lambda i: dvraw[i * mb:(i + 1) * mb].contiguous()
        half = nmb // 2
        for i in range(nmb):
            self._bind_grads(self._trainer(i), self.grads_b if i < half else self.grads)
        if half == 0:
            self._trainer(0).backward(dv(0))
            return
        self.flat_grad_b.zero_()

        def first_half():
            for i in range(half):
                self._trainer(i).backward(dv(i))

        def second_half():
            for i in range(half, nmb):
                self._trainer(i).backward(dv(i))
        if getattr(self, "_one_stream_backward", False) or not self.overlap_backward:
            first_half(); second_half()
        else:
            self._side_by_side(first_half, second_half)
        ops.axpy(self.flat_grad, self.flat_grad_b)

    def _prepare(self, B):
        mb = min(self.mb, B)
        assert B % mb == 0, "per-GPU batch must be a multiple of the micro-batch"
        nmb = B // mb
        if self.flat_grad is None:
            for i in range(nmb):
                self._trainer(i)
            self._alloc_flat_grads()
        self._reduce_done = False
        self.flat_grad.zero_()
        return mb, nmb

    def _refresh_perceiver(self):
        wT = self.trainers[0].perc.wT
        for name, fwd, path in self.refresh:
            m = self.masters[name]
            ops.cast_bf16(m, out=fwd)
            node = wT[path[0]]
            for k in path[1:-1]:
                node = node[k]
            ops.transpose_to_bf16(m, ldo=m.shape[0], out=node[path[-1]])

    def _refresh_operands(self):
        self._refresh_perceiver()

    def finish_reduce(self):
        """Sum `flat_grad` over ranks (one collective, once per forward_backward).  Until this has run the gradients are
        rank-local; clipping / logging / accumulation code calls it (or `reduced_grads()`) before touching them.  The 1/world
        of DDP's mean is applied in the optimizer step."""
        if self.dist and not self._reduce_done:
            self.comm.all_reduce_sum(self.flat_grad)
            self._reduce_done = True

    def reduced_grads(self):
        self.finish_reduce()
        return self.grads

    def optimizer_step(self):
        if self.dist:
            self.finish_reduce()
            self.opt.step(self.grads, grad_scale=1.0 / self.world)
        else:
            self.opt.step(self.grads)
        self._refresh_operands()
        ops.clamp_scalar(self.logit_scale, 0.0, math.log(100.0))


class DualAudioStep(_PerceiverLensStep):
    """Audio <-> text dual-tower step (reference `train_dual_one_epoch` + ClipLossGeneral, training/train.py:315-470,
    recipe TRAIN_INFERENCE.md:283-299 with --use_dual_loss --align_to text): text tower frozen, visual tower =
    AST tokenizer + Perceiver (trainable) -> locked ViT with unlocked class_embedding.  Multi-GPU semantics are the
    tri-modal step's (packed all-gather, flat gradient all-reduce = mean of per-rank gradients of the global loss)."""

    def __init__(self, sd, tower: TowerCfg, text: TextCfg, lens: LensCfg, device, micro_batch: int = 256, lr: float = 2e-4,
                 betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.2, rank: int = 0, world_size: int = 1,
                 gemm_cfg: int = -1, comm=None, frozen_res_dtype=torch.float32, local_loss: bool = False,
                 gather_with_grad: bool = False, train_res_dtype=torch.float32, force_comm: bool = False, text_wsplit: Optional[bool] = None, text_arith: str = "f16",
                 overlap_frozen: bool = True, overlap_backward: bool = True):
        self._init_host(sd, device, micro_batch, rank, world_size, comm, local_loss, gather_with_grad, force_comm, overlap_frozen,
                        overlap_backward)
        self._build(sd, tower, text, lens, gemm_cfg=gemm_cfg, frozen_res_dtype=frozen_res_dtype, train_res_dtype=train_res_dtype,
                    text_wsplit=text_wsplit, text_arith=text_arith)
        self.opt = AdamW(self.masters, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def _build(self, sd, tower, text, lens, gemm_cfg=-1, frozen_res_dtype=torch.float32, train_res_dtype=torch.float32,
               text_wsplit=None, text_arith="f16"):
        from .train import AudioLensTrainer
        device = self.dev
        self.text = TextEngine(sd, text, device, gemm_cfg=gemm_cfg, res_dtype=frozen_res_dtype, wsplit=text_wsplit, arith=text_arith)
        self.lens = LensEngine(sd, "visual.", tower, lens, device, gemm_cfg=gemm_cfg, res_dtype=train_res_dtype)
        self._mk = lambda: AudioLensTrainer(self.lens)
        self.masters["visual.class_embedding"] = self.lens.vit.cls
        self.masters["visual.visual_adapter.pos_emb"] = self.lens.adapter_pos
        from .engine import conv_weight_as_gemm
        self.masters["visual.visual_adapter.conv1.weight_gemm"] = conv_weight_as_gemm(
            sd["visual.visual_adapter.conv1.weight"], device, torch.float32)
        self._collect_perceiver(sd)

    def _refresh_operands(self):
        self._refresh_perceiver()
        ops.cast_bf16(self.masters["visual.visual_adapter.conv1.weight_gemm"], out=self.lens.conv_w)

    def forward_backward(self, audio: torch.Tensor, texts: torch.Tensor) -> torch.Tensor:
        B = audio.shape[0]
        mb, nmb = self._prepare(B)
        E = self.lens.tower.embed_dim
        ft = torch.empty(B, E, device=self.dev); fv = torch.empty(B, E, device=self.dev)
        vraw = torch.empty(B, E, device=self.dev); vnorm = torch.empty(B, device=self.dev)
        def frozen():
            ops.l2_normalize(self.text.encode_text(texts), out=ft)          # (whole batch at once: see TriModalDepthStep)

        def trainable():
            for i in range(nmb):
                s = slice(i * mb, (i + 1) * mb)
                vraw[s] = self._trainer(i).forward(audio[s])
        self._side_by_side(frozen, trainable)
        ops.l2_normalize(vraw, out=fv, norms=vnorm)
        scale = self.logit_scale          # device-side log-temperature (see TriModalDepthStep)
        if self.dist:
            allp = torch.empty(self.world * B, 2 * E, device=self.dev)
            self.comm.all_gather(allp, torch.cat([fv, ft], dim=1))
            av, at = [t.contiguous() for t in allp.split(E, dim=1)]
        else:
            av, at = fv, ft
        loss, dv, _, ds = pair_loss_and_grads(self.comm, self.rank, self.world, fv, ft, av, at, scale,    # ClipLossGeneral(x=visual, y=text)
                                              local_loss=self.local_loss, gather_with_grad=self.gather_with_grad, need_y=False, dist=self.dist)
        dvraw = ops.l2_normalize_bwd(fv, dv, vnorm)
        self._backward_all(dvraw, nmb, mb)
        self.grads["logit_scale"] += ds
        return loss

    def step(self, audio, texts):
        loss = self.forward_backward(audio, texts)
        self.optimizer_step()
        return loss


class TriModalPCStep(_PerceiverLensStep):
    """Point-cloud tri-modal step (reference pc_tri_main.py -> `tri_train_one_epoch` + TriClipLoss; model
    mm_vit_lens/model_cfg.py:85-110): image and text towers frozen, visual tower = PointBERT tokenizer (FPS -> kNN ->
    mini-PointNet with BatchNorm) + Perceiver, both trainable, in front of the locked ViT.  BatchNorm uses the batch
    statistics of each forward call (per micro-batch, per rank - SyncBN off, as the reference's default) and updates the
    running statistics; `bn_training=False` freezes it at the running statistics (--lock-visual-freeze-bn-stats);
    `bn_sync=True` (--use-bn-sync, pc_tri_main.py:372-373) makes the statistics and the backward sums global over the
    ranks: one all-gather of [2C+1] floats per BatchNorm forward, one all-reduce of [2C] per backward."""

    def __init__(self, sd, tower: TowerCfg, text: TextCfg, lens: LensCfg, device, micro_batch: int = 32, lr: float = 2e-4,
                 betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.2, rank: int = 0, world_size: int = 1,
                 gemm_cfg: int = -1, bn_training: bool = True, unlock_cls: bool = False, comm=None,
                 frozen_res_dtype=torch.float32, local_loss: bool = False, gather_with_grad: bool = False,
                 train_res_dtype=torch.float32, bn_sync: bool = False, force_comm: bool = False, text_wsplit: Optional[bool] = None, text_arith: str = "f16",
                 overlap_frozen: bool = True, overlap_backward: bool = True):
        self._init_host(sd, device, micro_batch, rank, world_size, comm, local_loss, gather_with_grad, force_comm, overlap_frozen,
                        overlap_backward)
        self._build(sd, tower, text, lens, gemm_cfg=gemm_cfg, frozen_res_dtype=frozen_res_dtype, train_res_dtype=train_res_dtype,
                    text_wsplit=text_wsplit, text_arith=text_arith, bn_training=bn_training, bn_sync=bn_sync, unlock_cls=unlock_cls)
        self.opt = AdamW(self.masters, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    def _build(self, sd, tower, text, lens, gemm_cfg=-1, frozen_res_dtype=torch.float32, train_res_dtype=torch.float32,
               text_wsplit=None, text_arith="f16", bn_training=True, bn_sync=False, unlock_cls=False):
        from .points import PointTokenizerTrainer
        from .train import PCLensTrainer
        device = self.dev
        self.image = VitEngine(sd, "image.", tower, device, gemm_cfg=gemm_cfg, res_dtype=frozen_res_dtype)
        self.text = TextEngine(sd, text, device, gemm_cfg=gemm_cfg, res_dtype=frozen_res_dtype, wsplit=text_wsplit, arith=text_arith)
        self.lens = LensEngine(sd, "visual.", tower, lens, device, gemm_cfg=gemm_cfg, res_dtype=train_res_dtype)
        self.tok = PointTokenizerTrainer(sd, "visual.visual_adapter.", lens, device, gemm_cfg=gemm_cfg, bn_training=bn_training,
                                         bn_sync=self.comm if bn_sync and self.dist else None, world_size=self.world)
        self._one_stream_backward = bool(bn_sync and self.dist)       # SyncBatchNorm's backward talks to the communicator
        self._mk = lambda: PCLensTrainer(self.lens, self.tok, train_cls=unlock_cls)
        if unlock_cls:
            self.masters["visual.class_embedding"] = self.lens.vit.cls
        self.masters.update(self.tok.masters)
        self._collect_perceiver(sd)

    def _bind_grads(self, t, grads=None):
        g = self.grads if grads is None else grads
        t.tower.grads = g; t.perc.grads = g; t.tok.grads = g
        self.tok.grads = self.grads

    def _refresh_operands(self):
        self._refresh_perceiver()
        self.tok.refresh_operands()
        for t in self.trainers:
            t.tok.op = self.tok.op

    def forward_backward(self, images, texts, points, fps_start=None) -> torch.Tensor:
        B = images.shape[0]
        mb, nmb = self._prepare(B)
        E = self.image.cfg.embed_dim
        fi = torch.empty(B, E, device=self.dev); ft = torch.empty(B, E, device=self.dev)
        fv = torch.empty(B, E, device=self.dev); vraw = torch.empty(B, E, device=self.dev)
        vnorm = torch.empty(B, device=self.dev)
        def frozen():
            ops.l2_normalize(self.text.encode_text(texts), out=ft)          # (whole batch at once: see TriModalDepthStep)
            for i in range(nmb):
                s = slice(i * mb, (i + 1) * mb)
                ops.l2_normalize(self.image.encode_image(images[s]), out=fi[s])

        def trainable():
            for i in range(nmb):
                s = slice(i * mb, (i + 1) * mb)
                vraw[s] = self._trainer(i).forward(points[s], None if fps_start is None else fps_start[s])
        self._side_by_side(frozen, trainable)
        ops.l2_normalize(vraw, out=fv, norms=vnorm)
        scale = self.logit_scale          # device-side log-temperature (see TriModalDepthStep)
        if self.dist:
            allp = torch.empty(self.world * B, 3 * E, device=self.dev)
            self.comm.all_gather(allp, torch.cat([fi, ft, fv], dim=1))
            ai, at, av = [t.contiguous() for t in allp.split(E, dim=1)]
        else:
            ai, at, av = fi, ft, fv
        kw = dict(local_loss=self.local_loss, gather_with_grad=self.gather_with_grad, need_x=False, dist=self.dist)
        l1, _, dv1, ds1 = pair_loss_and_grads(self.comm, self.rank, self.world, fi, fv, ai, av, scale, **kw)
        l2, _, dv2, ds2 = pair_loss_and_grads(self.comm, self.rank, self.world, ft, fv, at, av, scale, **kw)
        dvraw = ops.l2_normalize_bwd(fv, dv1 + dv2, vnorm)
        self._backward_all(dvraw, nmb, mb)
        self.grads["logit_scale"] += ds1 + ds2
        return l1 + l2

    def step(self, images, texts, points, fps_start=None):
        loss = self.forward_backward(images, texts, points, fps_start)
        self.optimizer_step()
        return loss
