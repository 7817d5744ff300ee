"""CPU, 2 processes on gloo: the WHOLE host logic of the fused training steps (`TriModalDepthStep`, `DualAudioStep`)
driven through the real `TorchComm` (torch.distributed), with the HIP `ops` replaced by a torch-CPU stand-in and the
towers by small linear "towers" whose exact gradients are known.  What is asserted, per rank:

  * the collective sequence of one step and its payloads: exactly ONE all-gather of the packed [b, k*E] embeddings
    (k = 3 tri-modal, 2 dual), reduce-scatter(s) only under gather_with_grad, and the gradient all-reduce(s) covering the
    flat gradient buffer exactly once (per-block async buckets + one call for the rest in the depth step);
  * the numbers: every rank reports the GLOBAL loss, and after the all-reduce each rank holds W x the mean = the sum of the
    per-rank gradients, equal to torch autograd of the global-batch loss on the concatenated data.

This is the N > 1 path of bench.py (`--gpus N`) minus the kernels; the kernels' side is covered on one GPU by
tests/test_hip_train.py (two ranks as threads through an in-process communicator)."""
import os
import sys
import types

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E_DIM, D_IN = 16, 24


# ------------------------------------------------------------------------------------------------ torch-CPU stand-in for ops
def _fake_ops():
    o = types.SimpleNamespace()
    o.EPI_F32 = 1

    def l2_normalize(x, out=None, out_bf16=None, norms=None, eps=1e-12):
        n = x.norm(dim=-1).clamp_min(eps)
        f = x / n[:, None]
        if norms is not None:
            norms.copy_(n)
        if out is not None:
            out.copy_(f)
            return out
        return f

    def l2_normalize_bwd(f, df, norms, eps=1e-12):
        s = (f * df).sum(-1, keepdim=True)
        return (df - f * s) / norms.clamp_min(eps)[:, None]
    o.l2_normalize, o.l2_normalize_bwd = l2_normalize, l2_normalize_bwd
    o.split_bf16x3 = lambda x, pattern: x

    def scale_exp(x, log_scale, out=None, mul=1.0):        # x * exp(logit_scale) with the scalar read from its tensor
        r = x * torch.exp(log_scale) * mul
        if out is not None:
            out.copy_(r)
            return out
        return r
    o.scale_exp = scale_exp
    o.logits_gemm = lambda xb, yb, scale: scale * xb @ yb.t()

    def ce_stats(logits, label_off=0, want_cols=True):
        R = logits.shape[0]
        diag = logits[torch.arange(R), torch.arange(R) + label_off].clone()
        return torch.logsumexp(logits, 1), (torch.logsumexp(logits, 0) if want_cols else None), diag

    def ce_loss_accum(loss, row_lse, col_lse, diag, R, Cc, label_off, w_row, w_col):
        if row_lse is not None:
            loss += w_row * (row_lse - diag).mean()
        if col_lse is not None and w_col != 0.0:
            loss += w_col * (col_lse[label_off:label_off + R] - diag).mean()

    def ce_grad(logits, row_lse, col_lse, label_off, w_row, w_col, logit_scale, dscale, need_g=True, need_gt=True):
        R, Cc = logits.shape
        G = torch.zeros_like(logits)
        eye = torch.zeros_like(logits); eye[torch.arange(R), torch.arange(R) + label_off] = 1.0
        if row_lse is not None and w_row != 0.0:
            G += w_row / R * (torch.exp(logits - row_lse[:, None]) - eye)
        if col_lse is not None and w_col != 0.0:
            G += w_col / R * (torch.exp(logits - col_lse[None, :]) - eye)
        dscale += (G * logits).sum() / logit_scale
        return (G if need_g else None), (G.t().contiguous() if need_gt else None)
    o.ce_stats, o.ce_loss_accum, o.ce_grad = ce_stats, ce_loss_accum, ce_grad
    o.transpose_to_bf16 = lambda y, ldo=None, out=None: y.t().contiguous()
    o.gemm = lambda a, w, bias=None, epi=1, alpha=1.0, **kw: alpha * a @ w.t()

    def adamw_step(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale=1.0):
        g = g * grad_scale
        p.mul_(1 - lr * wd)
        m.mul_(b1).add_(g, alpha=1 - b1); v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)
    o.adamw_step = adamw_step
    o.clamp_scalar = lambda t, lo, hi: t.clamp_(lo, hi)
    o.axpy = lambda y, x, alpha=1.0: y.add_(x, alpha=alpha)
    o.cast_bf16 = lambda x, out=None: x if out is None else out.copy_(x)
    return o


class _LinearTower:
    """features = flatten(x)[:, :D_IN] @ W; the 'trainer' interface of vitlens_hip.train (forward / backward / grads)."""

    def __init__(self, W, grads, name, blocks=0):
        self.W, self.grads, self.name, self.blocks = W, grads, name, blocks

    @property
    def tower(self):          # the steps bind the shared gradient dictionary through `.tower.grads`
        return self

    def forward(self, x):
        self.x = x.reshape(x.shape[0], -1)[:, :D_IN].float()
        return self.x @ self.W

    def backward(self, dfeat, on_block_done=None):
        self.grads[self.name] += self.x.t() @ dfeat
        for l in reversed(range(self.blocks)):        # block gradients become final in reverse layer order
            self.grads[f"visual.transformer.resblocks.{l}.mlp.c_fc.weight"] += (l + 1) * dfeat.sum() * torch.ones(3, 5)
            if on_block_done is not None:
                on_block_done(l)


class _Frozen:
    def __init__(self, W):
        self.W = W
        self.cfg = types.SimpleNamespace(embed_dim=E_DIM)

    def encode_image(self, x):
        return x.reshape(x.shape[0], -1)[:, :D_IN].float() @ self.W

    encode_text = encode_image


class _Rec:
    """TorchComm that logs (op, numel) of every call."""

    def __init__(self, inner):
        self.inner, self.log = inner, []

    def all_gather(self, out, inp):
        self.log.append(("all_gather", tuple(inp.shape))); self.inner.all_gather(out, inp)

    def all_reduce_sum(self, t):
        self.log.append(("all_reduce", t.numel())); self.inner.all_reduce_sum(t)

    def all_reduce_sum_async(self, t):
        self.log.append(("all_reduce_async", t.numel())); return self.inner.all_reduce_sum_async(t)

    def reduce_scatter_sum(self, out, inp):
        self.log.append(("reduce_scatter", tuple(inp.shape))); self.inner.reduce_scatter_sum(out, inp)


def _worker(rank, world, port, recipe, local_loss, gwg, ret, overlap=False):
    sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vitlens_hip import step as ST, train as TR
    ST.ops = _fake_ops(); TR.ops = ST.ops                      # the steps' host logic is what is under test
    g = torch.Generator().manual_seed(0)                        # same weights on every rank
    Wi, Wt, Wv = (torch.randn(D_IN, E_DIM, generator=g) * 0.3 for _ in range(3))
    b, nblk = 4, 2
    gd = torch.Generator().manual_seed(100)                     # same GLOBAL data on every rank; each takes its slice
    X = {k: torch.randn(world * b, D_IN + 3, generator=gd) for k in ("img", "txt", "vis")}
    mine = {k: v[rank * b:(rank + 1) * b] for k, v in X.items()}
    comm = _Rec(ST.TorchComm())
    errs = []
    sd0 = {"logit_scale": torch.tensor(2.0).log()}              # the parameter is log(scale) (model.py:446,619)

    # The objects under test are the PRODUCT classes, constructed by their own __init__ (-> _StepState._init_host: flags,
    # communicator, master table, bucket bookkeeping, optimizer); only the engine-building hook `_build`, the trainer factory
    # and the bf16 operand refresh are replaced by the linear stand-ins above.
    class _HostMixin:
        def _trainer(self, i):
            while len(self.trainers) <= i:
                self.trainers.append(self._mk())
            return self.trainers[i]

        def _refresh_operands(self):
            pass

    class DepthHost(_HostMixin, ST.TriModalDepthStep):
        def _build(self, sd, tower, text, **kw):
            self.image, self.text = _Frozen(Wi), _Frozen(Wt)
            for l in range(self.unlock_first_n):
                self.masters[f"visual.transformer.resblocks.{l}.mlp.c_fc.weight"] = torch.zeros(3, 5)
            self.masters["visual.W"] = Wv.clone()
            self._mk = lambda: _LinearTower(self.masters["visual.W"], self.grads, "visual.W", self.unlock_first_n)

    class AudioHost(_HostMixin, ST.DualAudioStep):
        def _build(self, sd, tower, text, lens, **kw):
            self.text = _Frozen(Wt)
            self.lens = types.SimpleNamespace(tower=types.SimpleNamespace(embed_dim=E_DIM))
            self.masters["visual.W"] = Wv.clone()
            self._mk = lambda: _LinearTower(self.masters["visual.W"], self.grads, "visual.W")

        def _bind_grads(self, t, grads=None):
            t.grads = self.grads if grads is None else grads

    kw = dict(micro_batch=2, lr=1e-2, rank=rank, world_size=world, comm=comm, local_loss=local_loss, gather_with_grad=gwg,
              overlap_frozen=overlap)
    if recipe == "depth":
        st = DepthHost(sd0, None, None, "cpu", unlock_first_n=nblk, **kw)
        # a CPU device cannot run a second HIP stream: the request is recorded, the schedule is the serial one
        if st.overlap_frozen != overlap or st._overlap_active:
            errs.append(f"overlap_frozen={st.overlap_frozen} active={st._overlap_active} on a CPU device")
        loss = st.forward_backward(mine["img"], mine["txt"], mine["vis"])
        k = 3
    else:
        st = AudioHost(sd0, None, None, None, "cpu", **kw)
        loss = st.forward_backward(mine["vis"], mine["txt"])
        k = 2
    # between forward_backward and optimizer_step the buffer is mixed (block buckets summed, the rest rank-local):
    # `finish_reduce()` / `reduced_grads()` is the accessor that makes it one state; idempotent (counted below: every
    # element is reduced exactly once although it is called twice and optimizer_step calls it again)
    st.finish_reduce()
    summed = {n: v.clone() for n, v in st.reduced_grads().items()}
    st.optimizer_step()
    for n, v in summed.items():
        if not torch.equal(v, st.grads[n]):
            errs.append(f"optimizer_step reduced {n} again after finish_reduce()")
    # ---- collective sequence ----
    ops_seen = [e[0] for e in comm.log]
    if ops_seen.count("all_gather") != 1 or comm.log[[e[0] for e in comm.log].index("all_gather")][1] != (b, k * E_DIM):
        errs.append(f"all_gather: {comm.log}")
    n_rs = ops_seen.count("reduce_scatter")
    want_rs = 0 if not gwg else (2 if recipe == "depth" else 1)       # one per contrastive pair: only the visual side is trained
    if n_rs != want_rs:
        errs.append(f"reduce_scatter count {n_rs} != {want_rs}: {comm.log}")
    red = sum(e[1] for e in comm.log if e[0].startswith("all_reduce"))
    if red != st.flat_grad.numel():
        errs.append(f"gradient all-reduce covers {red} of {st.flat_grad.numel()} elements: {comm.log}")
    if recipe == "depth":
        asyncs = [e for e in comm.log if e[0] == "all_reduce_async"]
        if len(asyncs) != nblk:
            errs.append(f"expected {nblk} per-block async all-reduces, got {comm.log}")
        if ops_seen.index("all_reduce_async") < ops_seen.index("all_gather"):
            errs.append("bucket all-reduce before the embedding exchange")
    # ---- numbers: global-batch autograd on the concatenated data ----
    Wr = Wv.clone().requires_grad_(True); ls = torch.tensor(2.0, requires_grad=True)
    nrm = lambda t: t / t.norm(dim=-1, keepdim=True)
    fv = nrm(X["vis"][:, :D_IN] @ Wr); ft = nrm(X["txt"][:, :D_IN] @ Wt); fi = nrm(X["img"][:, :D_IN] @ Wi)
    lab = torch.arange(world * b)
    ce = torch.nn.functional.cross_entropy

    def pair(x, y):
        lg = ls * x @ y.t()
        return (ce(lg, lab) + ce(lg.t(), lab)) / 2
    ref = pair(fi, fv) + pair(ft, fv) if recipe == "depth" else pair(fv, ft)
    ref.backward()
    # reported loss: the global loss on every rank, or (local_loss) this rank's rows, whose mean over ranks is the global loss
    lt = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(lt, loss.detach().reshape(1).float())
    if local_loss:
        if abs(float(sum(lt)) / world - float(ref)) > 1e-4:
            errs.append(f"mean of local losses {float(sum(lt)) / world} vs global {float(ref)}")
    elif any(abs(float(x) - float(ref)) > 1e-4 for x in lt):
        errs.append(f"loss {[float(x) for x in lt]} vs global {float(ref)}")
    # gradients after the all-reduce hold the SUM over ranks; the optimizer divides by W (DDP mean).  Reference semantics
    # (loss.py:55-78 + DDP): without gather_with_grad every rank differentiates only its own slice of the global loss, so the
    # mean over ranks is (1/W) dL/dW; with it (reduce-scatter of the feature gradients) the mean is dL/dW itself.
    got = st.grads["visual.W"] / world
    want = Wr.grad if gwg else Wr.grad / world
    if not torch.allclose(got, want, rtol=2e-4, atol=1e-6):
        errs.append(f"grad mismatch: {float((got - want).abs().max()):.3e} (scale {float(want.abs().max()):.3e})")
    want_ls = ls.grad * 2.0                             # the master is log(scale): chain-rule factor = scale; every rank sees dL/ds
    if not torch.allclose(st.grads["logit_scale"] / world, want_ls.reshape(1), rtol=1e-3, atol=1e-5):
        errs.append(f"logit_scale grad {float(st.grads['logit_scale'] / world)} vs {float(want_ls)}")
    if float((st.masters["visual.W"] - Wv).abs().max()) == 0.0:
        errs.append("optimizer step did not move the master")
    # replicas stay identical: compare a checksum across ranks
    cs = torch.tensor([float(st.masters["visual.W"].double().sum())])
    lst = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(lst, cs)
    if any(abs(float(x) - float(lst[0])) > 1e-9 for x in lst):
        errs.append(f"replicas diverged: {lst}")
    ret[rank] = errs
    dist.destroy_process_group()


def _run(world, recipe, local_loss, gwg, overlap=False):
    port = 29650 + (hash((world, recipe, local_loss, gwg, overlap)) % 300)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, recipe, local_loss, gwg, ret, overlap), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == [], (r, ret[r])


@pytest.mark.parametrize("recipe", ["depth", "audio"])
@pytest.mark.parametrize("local_loss,gwg", [(False, False), (False, True), (True, True)])
def test_step_host_logic_world2_gloo(recipe, local_loss, gwg):
    _run(2, recipe, local_loss, gwg)


def test_depth_step_world2_gloo_with_overlap_frozen_requested():
    """overlap_frozen=True (the product default) on a device without HIP streams: same collectives, same numbers."""
    _run(2, "depth", False, False, overlap=True)


@pytest.mark.parametrize("recipe,local_loss,gwg", [("depth", False, False), ("depth", True, True), ("audio", False, True)])
def test_step_host_logic_world8_gloo(recipe, local_loss, gwg):
    """The node size the north_star names (8 ranks): bucket order, ONE packed gather of [b, k*E] per rank, reduce-scatter
    under gather_with_grad, every gradient element reduced exactly once, replicas identical after AdamW."""
    _run(8, recipe, local_loss, gwg)
