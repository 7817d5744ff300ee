"""GPU: the open_clip-compatible loss modules (HIP kernels inside one autograd node per pair) against the
loss values and feature gradients the reference produced (tests/golden/tiny_*.npz, gather_w*.npz)."""
import pytest
import torch

from golden_util import load_npz, split

pytestmark = pytest.mark.gpu


def _loss_mod():
    import importlib, sys
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        f = getattr(sys.modules[k], "__file__", "") or ""
        if "vit-lens_amd" not in f:
            del sys.modules[k]
    return importlib.import_module("open_clip.loss")


@pytest.mark.parametrize("modality", ["depth", "audio", "pc"])
def test_tri_and_dual_losses_match_reference(modality):
    L = _loss_mod()
    sd, ins, outs, grads, meta = split(load_npz(f"tiny_{modality}.npz"))
    f = {k: outs[k + "_features"].cuda().requires_grad_(True) for k in ("image", "text", "visual")}
    ls = outs["logit_scale"].cuda().requires_grad_(True)
    loss = L.TriClipLoss()(f["image"], f["text"], f["visual"], ls)
    assert abs(float(loss) - float(outs["tri_loss"])) < 2e-3
    loss.backward()
    for k in f:
        ref = outs[f"tri_grad_{k}"]
        # dL/dfeatures = scale * G @ Y is a difference of nearly equal vectors (rows of G sum to ~0 and the
        # features are strongly correlated), which amplifies the bf16 rounding of G and Y ~5x: 4e-2 here.
        assert float((f[k].grad.cpu() - ref).norm() / ref.norm()) < 4e-2, k
    assert abs(float(ls.grad) - float(outs["tri_grad_logit_scale"])) < 2e-2 * max(1.0, abs(float(outs["tri_grad_logit_scale"])))
    x = outs["visual_features"].cuda().requires_grad_(True); y = outs["text_features"].cuda().requires_grad_(True)
    d = L.ClipLossGeneral()(x, y, outs["logit_scale"].cuda(), output_dict=True, key="v-t")
    assert abs(float(d["v-t"]) - float(outs["dual_loss"])) < 2e-3
    d["v-t"].backward()
    assert float((x.grad.cpu() - outs["dual_grad_x"]).norm() / outs["dual_grad_x"].norm()) < 4e-2


def test_global_loss_equals_reference_rank_value():
    """What every rank computes in the non-local multi-GPU loss: the loss over the rank-major concatenation."""
    L = _loss_mod()
    z = {k: torch.from_numpy(v) for k, v in load_npz("gather_w4.npz").items()}
    allx = torch.cat([z[f"in/x{r}"] for r in range(4)]).cuda()
    ally = torch.cat([z[f"in/y{r}"] for r in range(4)]).cuda()
    loss = L.ClipLossGeneral()(allx, ally, torch.tensor(14.285714).cuda())
    assert abs(float(loss) - float(z["rank0/dual_ll0_gg0_loss"])) < 2e-3


def test_loss_refuses_cpu_tensors():
    L = _loss_mod()
    x = torch.nn.functional.normalize(torch.randn(4, 64), dim=-1)
    with pytest.raises(RuntimeError):
        L.ClipLoss()(x, x, torch.tensor(10.0))


def _pair_case(R, Cn, D, seed, scale=14.285714, correlated=True):
    g = torch.Generator().manual_seed(seed)
    y = torch.nn.functional.normalize(torch.randn(Cn, D, generator=g), dim=-1)
    x = torch.randn(R, D, generator=g)
    if correlated:                                    # positives look like positives: x_r is close to y_r
        x = 0.6 * y[:R] + 0.8 * torch.nn.functional.normalize(x, dim=-1)
    x = torch.nn.functional.normalize(x, dim=-1)
    return x, y, scale


def _oracle_pair(x, y, scale, label_off=0, w_row=0.5, w_col=0.5):
    """(loss.py:129-136,158-163): fp32 autograd on the CPU."""
    x = x.clone().requires_grad_(True); y = y.clone().requires_grad_(True); s = torch.tensor(scale, requires_grad=True)
    logits = s * x @ y.t()
    labels = torch.arange(x.shape[0]) + label_off
    loss = w_row * torch.nn.functional.cross_entropy(logits, labels)
    if w_col:
        loss = loss + w_col * torch.nn.functional.cross_entropy(logits.t(), labels)
    loss.backward()
    return float(loss), x.grad, y.grad, float(s.grad)


@pytest.mark.parametrize("R,Cn,off,w_col,chunk", [(8192, 8192, 0, 0.5, 1024), (8192, 8192, 0, 0.5, None), (1000, 8000, 3000, 0.0, 384),
                                                  (2050, 2050, 0, 0.5, 512)])
def test_row_blocked_pair_loss_vs_oracle_and_whole_matrix(R, Cn, off, w_col, chunk):
    """N1: InfoNCE over a global batch of 8192 (8 ranks x 1024) without ever holding the 8192 x 8192 logits: row blocks
    with exact row LSE, merged column LSE, backward by recomputation.  Checked against fp32 autograd on the CPU AND
    against the whole-matrix path of the same kernels (same loss to fp32 summation order; gradients to bf16 rounding
    of dL/dlogits); (1000 x 8000, offset 3000, rows only) is the --local-loss geometry; chunk=None takes the automatic
    threshold (8192^2 elements > 2^26 -> 2048-row blocks)."""
    from vitlens_hip import step as ST
    x, y, scale = _pair_case(R, Cn, 768, seed=R + Cn)
    ref_loss, ref_dx, ref_dy, ref_ds = _oracle_pair(x, y, scale, off, 0.5, w_col)
    xc, yc = x.cuda(), y.cuda()
    loss_b, ctx_b = ST.pair_forward(xc, yc, scale, off, 0.5, w_col, chunk_rows=chunk)
    assert ctx_b[2] is None and ctx_b[9] > 0                                    # really blocked: no logits kept
    dx_b, dy_b, ds_b = ST.pair_backward(ctx_b)
    loss_w, ctx_w = ST.pair_forward(xc, yc, scale, off, 0.5, w_col, chunk_rows=0)
    dx_w, dy_w, ds_w = ST.pair_backward(ctx_w)
    rel = lambda a, b: float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
    assert abs(float(loss_b) - ref_loss) < 2e-4 * max(1.0, abs(ref_loss)), (float(loss_b), ref_loss)
    assert abs(float(loss_b) - float(loss_w)) < 2e-5 * max(1.0, abs(ref_loss))
    assert rel(dx_b, ref_dx) < 4e-2 and rel(dy_b, ref_dy) < 4e-2, (rel(dx_b, ref_dx), rel(dy_b, ref_dy))
    assert rel(dx_b, dx_w) < 1e-2 and rel(dy_b, dy_w) < 1e-2, (rel(dx_b, dx_w), rel(dy_b, dy_w))
    assert abs(float(ds_b) - ref_ds) < 2e-2 * max(1e-3, abs(ref_ds)) + 2e-5, (float(ds_b), ref_ds)
    assert abs(float(ds_b) - float(ds_w)) < 1e-2 * max(1e-3, abs(ref_ds)) + 2e-5


def test_dscale_is_deterministic():
    """d(loss)/d(logit_scale) is a two-stage fixed-order sum (round 1 used fp32 atomics): bit-identical run to run."""
    from vitlens_hip import step as ST
    x, y, scale = _pair_case(1024, 1024, 768, seed=5)
    xc, yc = x.cuda(), y.cuda()
    vals = set()
    for _ in range(5):
        _, c = ST.pair_forward(xc, yc, scale)
        vals.add(float(ST.pair_backward(c)[2]))
    assert len(vals) == 1, vals


def test_loss_module_takes_chunk_rows():
    L = _loss_mod()
    x, y, scale = _pair_case(1536, 1536, 768, seed=9)
    ls = torch.tensor(scale).cuda()
    a = L.ClipLoss()(x.cuda(), y.cuda(), ls)
    xb = x.cuda().requires_grad_(True)
    b = L.ClipLoss(chunk_rows=256)(xb, y.cuda(), ls)
    assert abs(float(a) - float(b)) < 1e-5 * max(1.0, abs(float(a)))
    b.backward()
    assert torch.isfinite(xb.grad).all()


@pytest.mark.parametrize("chunk_rows", [0, 128])
def test_device_side_temperature_equals_the_host_scalar_path(chunk_rows):
    """`logit_scale` as a 1-element device tensor (the log-temperature the model holds, model.py:619): exp() and every use
    of the scale happen on the device (vl_scale_exp_f32), nothing reads it on the host.  Same loss and feature gradients as
    the float-scale path; the third result is d/d(logit_scale) = scale * dL/d(scale)."""
    import math
    from vitlens_hip import step as ST
    g = torch.Generator().manual_seed(11)
    R, E = 384, 768
    x = torch.nn.functional.normalize(torch.randn(R, E, generator=g), dim=-1).cuda()
    y = torch.nn.functional.normalize(torch.randn(R, E, generator=g), dim=-1).cuda()
    log_s = torch.tensor([math.log(1 / 0.07)], device="cuda")
    s = float(log_s.exp())
    la, ca = ST.pair_forward(x, y, s, chunk_rows=chunk_rows)
    lb, cb = ST.pair_forward(x, y, log_s, chunk_rows=chunk_rows)
    assert abs(float(la) - float(lb)) < 2e-5 * max(1.0, abs(float(la)))
    dxa, dya, dsa = ST.pair_backward(ca)
    dxb, dyb, dsb = ST.pair_backward(cb)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(dxb, dxa) < 2e-3 and rel(dyb, dya) < 2e-3, (rel(dxb, dxa), rel(dyb, dya))
    assert abs(float(dsb) - float(dsa) * s) < 2e-3 * max(1.0, abs(float(dsa) * s)), (float(dsb), float(dsa) * s)
