"""GPU: backward / optimizer parity.  The HIP backward of the trainable tower is compared with torch
autograd through the ORACLE (fp32, CPU) and with the parameter gradients the reference itself produced
for the tiny golden model (tests/golden/tiny_depth.npz: `grad/visual.*`)."""
import math

import pytest
import torch

import vitlens_oracle as O
from golden_util import load_npz, split, specs_from_meta

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _tiny_depth():
    from vitlens_hip import engine as E
    sd, ins, outs, grads, meta = split(load_npz("tiny_depth.npz"))
    tower, text, lens = specs_from_meta(meta)
    tc = E.TowerCfg(width=tower.width, layers=tower.layers, heads=tower.heads, patch=tower.patch,
                    image_size=tower.image_size, embed_dim=tower.embed_dim)
    lc = E.LensCfg(modality="depth", perceiver_identity=True)
    return sd, ins, outs, grads, tc, lc


def test_depth_lens_forward_matches_golden():
    from vitlens_hip import engine as E
    sd, ins, outs, grads, tc, lc = _tiny_depth()
    le = E.LensEngine(sd, "visual.", tc, lc, "cuda")
    f = le.encode(ins["visual_x"].cuda())
    assert relerr(f, outs["visual_raw"]) < 2e-2, relerr(f, outs["visual_raw"])


def test_depth_tower_backward_vs_reference_grads():
    """d(tri-modal loss)/d(visual params): HIP backward vs the reference's own autograd (golden)."""
    from vitlens_hip import engine as E, train as TR
    sd, ins, outs, grads, tc, lc = _tiny_depth()
    le = E.LensEngine(sd, "visual.", tc, lc, "cuda")
    tr = TR.DepthLensTrainer(le, unlock_first_n=tc.layers)
    tr.tower.train_cls = tr.tower.train_pos = True
    feat = tr.forward(ins["visual_x"].cuda())
    assert relerr(feat, outs["visual_raw"]) < 2e-2
    # loss gradient w.r.t. the raw visual features, from the golden (reference) features, fp32 on CPU
    v = outs["visual_raw"].clone().requires_grad_(True)
    loss = O.tri_clip_loss(outs["image_features"], outs["text_features"], O.l2_normalize(v), outs["logit_scale"])
    loss.backward()
    tr.backward(v.grad.cuda())
    checked = 0
    for name, g in tr.grads.items():
        if name.endswith("conv1.weight_gemm"):
            ref = grads["visual.visual_adapter.conv1.weight"].reshape(g.shape[0], -1)
            got = g[:, :ref.shape[1]]
            assert g.shape[1] == ref.shape[1] or float(g[:, ref.shape[1]:].abs().max()) == 0.0
        else:
            ref, got = grads[name], g
        assert got.shape == ref.shape, name
        e = relerr(got, ref)
        assert e < 5e-2, (name, e)
        checked += 1
    assert checked >= 2 * 12 + 4, checked


def test_vitl_block_backward_vs_oracle_autograd():
    """ViT-L geometry (1024 wide, 16 heads, 257 tokens), 2 blocks, batch 2: token gradient and block-0
    weight gradients against torch autograd through the oracle."""
    from vitlens_hip import engine as E, train as TR
    spec = O.TowerSpec(layers=2)
    g = torch.Generator().manual_seed(3)
    sd = O.init_tower(spec, g, "visual.")
    tok = torch.randn(2, 256, 1024, generator=g) * 0.5
    dfeat = torch.randn(2, 768, generator=g)
    sdg = {k: v.clone().requires_grad_(k.startswith("visual.transformer.resblocks.0.")) for k, v in sd.items()}
    tk = tok.clone().requires_grad_(True)
    (O.vit_trunk(sdg, "visual.", tk, spec) * dfeat).sum().backward()
    eng = E.VitEngine(sd, "visual.", E.TowerCfg(layers=2), "cuda")
    tr = TR.TowerTrainer(eng, train_blocks=[0])
    feat = tr.forward(tok.reshape(-1, 1024).cuda().bfloat16(), 2)
    dtok = tr.backward(dfeat.cuda())
    assert relerr(dtok.reshape(2, 256, 1024), tk.grad) < 3e-2, relerr(dtok.reshape(2, 256, 1024), tk.grad)
    for name, gbuf in tr.grads.items():
        assert relerr(gbuf, sdg[name].grad) < 3e-2, (name, relerr(gbuf, sdg[name].grad))
    assert len(tr.grads) == 12


@pytest.mark.parametrize("B,L,H,dh,causal", [(2, 257, 4, 64, False), (2, 77, 3, 64, True), (1, 40, 2, 32, False),
                                             (2, 257, 2, 64, True), (1, 256, 2, 64, False), (1, 289, 2, 64, False),
                                             (2, 33, 2, 32, False), (1, 600, 1, 64, False),
                                             (2, 257, 2, 104, False), (1, 129, 2, 80, True), (1, 300, 1, 128, False)])
def test_attention_backward(B, L, H, dh, causal):
    """dq / dk / dv (and the in-kernel delta) against autograd through explicit softmax attention on the same bf16 q, k, v,
    all operands read in place from token-major matrices.  257 / 33 = shared last query AND key row, 289 = two
    workgroups per (b, h), 600 = several LDS chunks.  (The two-kernel path; the one-kernel backward: next test.)"""
    _attention_backward_case(B, L, H, dh, causal, fused=False)


@pytest.mark.parametrize("B,L,H,dh", [(2, 257, 4, 64), (1, 256, 2, 64), (2, 33, 2, 64), (3, 50, 12, 64), (2, 200, 2, 64), (1, 32, 1, 64),
                                      (2, 129, 3, 64), (1, 7, 2, 64), (5, 225, 2, 64)])
def test_attention_backward_fused(B, L, H, dh):
    """The ONE-kernel backward (vl_attn_bwd_fused_bf16: every score tile evaluated once, dS handed to the query-owning wave
    through LDS, transposed fragments by ds_read_b64_tr_b16, delta while staging) on every geometry it claims: the lone
    class-token row (257, 33, 129, 225 = 32 m + 1), whole tiles (256, 32), ragged last tiles (50, 200, 7)."""
    from vitlens_hip import ops
    assert ops._lib.vl_attn_bwd_fused_supported(L, L, dh, 0) == 1
    _attention_backward_case(B, L, H, dh, False, fused=True)


def test_attention_backward_fused_refuses_what_it_does_not_take():
    from vitlens_hip import ops
    sup = ops._lib.vl_attn_bwd_fused_supported
    assert sup(257, 257, 64, 0) == 1 and sup(256, 256, 64, 0) == 1 and sup(77, 77, 64, 0) == 1
    assert sup(257, 257, 64, 1) == 0 and sup(257, 256, 64, 0) == 0 and sup(257, 257, 32, 0) == 0 and sup(289, 289, 64, 0) == 0
    assert sup(258, 258, 64, 0) == 0 and sup(600, 600, 64, 0) == 0
    x = torch.zeros(4, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.attn_bwd(*[ops.heads_view(torch.zeros(2 * 300, 64, dtype=torch.bfloat16, device="cuda"), 2, 300, 1, 64)] * 5,
                     torch.zeros(2, 1, 300, device="cuda"), None, x, x, x, 64, 64, fused=True)


def _attention_backward_case(B, L, H, dh, causal, fused):
    from vitlens_hip import ops
    D = H * dh
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B * L, D, generator=g)).bfloat16().cuda()
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).bfloat16().cuda()
    mk = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device="cuda")
    qkv = mk(B * L, 3 * D)
    ops.gemm(x, w, None, out=qkv, epi=ops.EPI_BF16)
    q, k, v = (ops.heads_view(qkv, B, L, H, dh, i * D) for i in range(3))
    o = mk(B * L, D); lse = torch.empty(B, H, L, device="cuda")
    qscale = dh ** -0.5 * ops.LOG2E
    ops.attn_fwd(q, k, v, o, lse=lse, causal=causal, qscale=qscale)
    do_tok = torch.randn(B * L, D, generator=g).bfloat16().cuda()
    delta = torch.full((B, H, L), float("nan"), device="cuda")
    dqkv = torch.full((B * L, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.attn_bwd(q, k, v, ops.heads_view(do_tok, B, L, H, dh), ops.heads_view(o, B, L, H, dh), lse, delta,
                 dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, 3 * D, causal=causal, fused=fused)
    # reference: autograd through explicit softmax attention; the kernels use q2 = bf16(q * qscale)
    qr = q.float().cpu().requires_grad_(True)
    kr = k.float().cpu().requires_grad_(True); vr = v.float().cpu().requires_grad_(True)
    s = (qr @ kr.transpose(-1, -2)) * dh ** -0.5
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    out = torch.softmax(s, -1) @ vr
    dO = do_tok.float().cpu().reshape(B, L, H, dh).permute(0, 2, 1, 3)
    (out * dO).sum().backward()
    tok = lambda t: t.permute(0, 2, 1, 3).reshape(B * L, D)
    assert torch.isfinite(dqkv.float()).all()
    if not fused:          # (the one-kernel backward keeps delta in LDS)
        assert relerr(delta, (out.detach() * dO).sum(-1)) < 1e-2
    assert relerr(dqkv[:, :D], tok(qr.grad)) < 2e-2, relerr(dqkv[:, :D], tok(qr.grad))
    assert relerr(dqkv[:, D:2 * D], tok(kr.grad)) < 2e-2, relerr(dqkv[:, D:2 * D], tok(kr.grad))
    assert relerr(dqkv[:, 2 * D:], tok(vr.grad)) < 2e-2, relerr(dqkv[:, 2 * D:], tok(vr.grad))


@pytest.mark.parametrize("Lq,Lk", [(256, 600), (257, 64), (64, 257)])
def test_cross_attention_backward(Lq, Lk):
    """Perceiver cross-attention (Lq != Lk): q from one matrix, k | v packed in another."""
    from vitlens_hip import ops
    B, H, dh = 2, 2, 64
    inner = H * dh
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16().cuda()
    q2, kv2, do2 = mk(B * Lq, inner), mk(B * Lk, 2 * inner), mk(B * Lq, inner)
    q = ops.heads_view(q2, B, Lq, H, dh); k = ops.heads_view(kv2, B, Lk, H, dh); v = ops.heads_view(kv2, B, Lk, H, dh, inner)
    o = torch.empty(B * Lq, inner, dtype=torch.bfloat16, device="cuda"); lse = torch.empty(B, H, Lq, device="cuda")
    ops.attn_fwd(q, k, v, o, lse=lse, qscale=dh ** -0.5 * ops.LOG2E)
    delta = torch.empty(B, H, Lq, device="cuda")
    dq = torch.full((B * Lq, inner), float("nan"), dtype=torch.bfloat16, device="cuda")
    dkv = torch.full((B * Lk, 2 * inner), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.attn_bwd(q, k, v, ops.heads_view(do2, B, Lq, H, dh), ops.heads_view(o, B, Lq, H, dh), lse, delta, dq, dkv,
                 dkv[:, inner:], inner, 2 * inner)
    qr = q.float().cpu().requires_grad_(True); kr = k.float().cpu().requires_grad_(True); vr = v.float().cpu().requires_grad_(True)
    out = torch.softmax((qr @ kr.transpose(-1, -2)) * dh ** -0.5, -1) @ vr
    (out * do2.float().cpu().reshape(B, Lq, H, dh).permute(0, 2, 1, 3)).sum().backward()
    tok = lambda t, L: t.permute(0, 2, 1, 3).reshape(B * L, inner)
    assert relerr(o, tok(out.detach(), Lq)) < 1e-2
    assert relerr(dq, tok(qr.grad, Lq)) < 2e-2
    assert relerr(dkv[:, :inner], tok(kr.grad, Lk)) < 2e-2
    assert relerr(dkv[:, inner:], tok(vr.grad, Lk)) < 2e-2


def test_layernorm_backward_and_params():
    from vitlens_hip import ops
    rows, D = 50, 1024
    g = torch.Generator().manual_seed(6)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5)
    w = 1 + 0.1 * torch.randn(D, generator=g); b = 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(rows, D, generator=g).bfloat16()
    dres = torch.randn(rows, D, generator=g)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    (torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5) * dy.float()).sum().backward()
    xc = x.cuda(); y = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    ops.layernorm(xc, w.cuda(), b.cuda(), y, rows, D, mean=mean, rstd=rstd)
    dx = torch.empty(rows, D, device="cuda"); dxb = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_bwd(dy.cuda(), xc, mean, rstd, w.cuda(), rows, D, dres=dres.cuda(), dx=dx, dx_bf16=dxb)
    assert relerr(dx, xr.grad + dres) < 1e-5
    assert relerr(dxb, xr.grad + dres) < 4e-3
    dw = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    ops.layernorm_bwd_params(dy.cuda(), xc, mean, rstd, dw, db, rows, D)
    assert relerr(dw, wr.grad) < 1e-5 and relerr(db, br.grad) < 1e-5
    # deterministic reductions: a second call adds exactly the same numbers
    dw2 = torch.zeros(D, device="cuda"); db2 = torch.zeros(D, device="cuda")
    ops.layernorm_bwd_params(dy.cuda(), xc, mean, rstd, dw2, db2, rows, D)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("rows,D", [(50, 1024), (7, 768), (33, 512), (20, 384)])
def test_layernorm_backward_bf16_streams(rows, D):
    """bf16 residual x and bf16 residual-gradient stream (what the reference's amp_bf16 autocast carries): the kernel reads
    the bf16-rounded x / dres and rounds dx once -> compare with fp32 autograd on the rounded inputs at bf16 resolution."""
    from vitlens_hip import ops
    g = torch.Generator().manual_seed(16)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).bfloat16()
    w = 1 + 0.1 * torch.randn(D, generator=g); b = 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(rows, D, generator=g).bfloat16()
    dres = torch.randn(rows, D, generator=g).bfloat16()
    xr = x.float().requires_grad_(True)
    (torch.nn.functional.layer_norm(xr, (D,), w, b, 1e-5) * dy.float()).sum().backward()
    xc = x.cuda(); y = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    ops.layernorm(xc, w.cuda(), b.cuda(), y, rows, D, mean=mean, rstd=rstd)
    dx = dres.clone().cuda()
    ops.layernorm_bwd(dy.cuda(), xc, mean, rstd, w.cuda(), rows, D, dres=dx, dx=dx)        # in place, bf16 stream
    assert dx.dtype == torch.bfloat16 and relerr(dx, xr.grad + dres.float()) < 4e-3
    dxf = torch.empty(rows, D, device="cuda")
    ops.layernorm_bwd(dy.cuda(), xc, mean, rstd, w.cuda(), rows, D, dx=dxf)                 # bf16 x, f32 out, no upstream
    assert relerr(dxf, xr.grad) < 1e-5


@pytest.mark.parametrize("rows,D", [(50, 1024), (777, 1024), (1030, 768), (263, 72), (41, 100)])
def test_layernorm_param_gradients_bf16_streams(rows, D):
    """dgamma / dbeta from bf16 dy and bf16 x (the 16-byte kernel for D % 8 == 0, the per-column one otherwise), row counts that
    are no multiple of the row lanes or of the 256-row slabs, accumulating into existing values."""
    from vitlens_hip import ops
    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).bfloat16().cuda()
    dy = torch.randn(rows, D, generator=g).bfloat16().cuda()
    w = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    y = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    ops.layernorm(x, w, b, y, rows, D, mean=mean, rstd=rstd)
    dw = torch.full((D,), 2.0, device="cuda"); db = torch.full((D,), -1.0, device="cuda")
    ops.layernorm_bwd_params(dy, x, mean, rstd, dw, db, rows, D)
    xh = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
    assert relerr(dw, 2.0 + (dy.double() * xh).sum(0)) < 1e-5
    assert relerr(db, -1.0 + dy.double().sum(0)) < 1e-5


@pytest.mark.parametrize("rows,D", [(50, 1024), (777, 1024), (1030, 768), (263, 72), (41, 102)])
def test_layernorm_param_gradients_bf16_dy_fp32_x(rows, D):
    """dgamma / dbeta from a bf16 dy and an fp32 x (the Perceiver's LayerNorms on its fp32 residual stream): the 4-column kernel
    (D % 4 == 0) and the per-column one (D = 102), row counts that are no multiple of the row lanes or of the 256-row slabs,
    a strided x (a column window of a wider tensor), accumulation into existing values, deterministic."""
    from vitlens_hip import ops
    g = torch.Generator().manual_seed(rows * 3 + D)
    wide = (torch.randn(rows, D + 8, generator=g) * 2 + 0.5).cuda()
    x = wide[:, 4:4 + D] if D % 4 == 0 else wide[:, :D].contiguous()
    dy = torch.randn(rows, D, generator=g).bfloat16().cuda()
    xd = x.double()
    mean = xd.mean(1); rstd = (xd.var(1, unbiased=False) + 1e-5).rsqrt()
    dw = torch.full((D,), 2.0, device="cuda"); db = torch.full((D,), -1.0, device="cuda")
    ops.layernorm_bwd_params(dy, x, mean.float(), rstd.float(), dw, db, rows, D, x_row_stride=x.stride(0))
    xh = (xd - mean.float().double()[:, None]) * rstd.float().double()[:, None]
    assert relerr(dw, 2.0 + (dy.double() * xh).sum(0)) < 1e-5
    assert relerr(db, -1.0 + dy.double().sum(0)) < 1e-5
    dw2 = torch.full((D,), 2.0, device="cuda"); db2 = torch.full((D,), -1.0, device="cuda")
    ops.layernorm_bwd_params(dy, x, mean.float(), rstd.float(), dw2, db2, rows, D, x_row_stride=x.stride(0))
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


def test_adamw_matches_torch():
    from vitlens_hip import train as TR
    g = torch.Generator().manual_seed(7)
    p0 = {"w": torch.randn(37, 19, generator=g), "ln.bias": torch.randn(19, generator=g)}
    ref = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    opt = torch.optim.AdamW([{"params": [ref["w"]], "weight_decay": 0.2}, {"params": [ref["ln.bias"]], "weight_decay": 0.0}],
                            lr=1e-2, betas=(0.9, 0.98), eps=1e-6)
    mine = {k: v.clone().cuda() for k, v in p0.items()}
    mopt = TR.AdamW(mine, lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)
    for it in range(5):
        grads = {k: torch.randn(v.shape, generator=g) for k, v in p0.items()}
        for k in ref:
            ref[k].grad = grads[k].clone()
        opt.step()
        mopt.step({k: v.cuda() for k, v in grads.items()})
    for k in ref:
        assert relerr(mine[k], ref[k].detach()) < 1e-5, k


def test_gemm_dgelu_and_preact_save():
    from vitlens_hip import ops
    g = torch.Generator().manual_seed(8)
    M, N, K = 300, 256, 128
    a = torch.randn(M, K, generator=g).bfloat16().cuda(); w = (torch.randn(N, K, generator=g) * 0.2).bfloat16().cuda()
    bias = torch.randn(N, generator=g).cuda()
    u = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    h = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU, out2=u)
    acc = a.float().cpu() @ w.float().cpu().t() + bias.cpu()
    assert relerr(u, acc) < 4e-3 and relerr(h, torch.nn.functional.gelu(acc)) < 4e-3
    dy = torch.randn(M, K, generator=g).bfloat16().cuda()
    du = ops.gemm(dy, w, None, res=u, out=torch.empty_like(u), epi=ops.EPI_DGELU)
    ur = u.float().cpu().requires_grad_(True)
    torch.nn.functional.gelu(ur).sum().backward()
    ref = (dy.float().cpu() @ w.float().cpu().t()) * ur.grad
    assert relerr(du, ref) < 5e-3
    y = ops.gelu_bf16(u, torch.empty_like(u))
    assert relerr(y, torch.nn.functional.gelu(u.float().cpu())) < 4e-3


@pytest.mark.parametrize("res_dtype", [torch.float32, torch.bfloat16])
def test_tri_modal_step_matches_reference_step(res_dtype):
    """The whole tri-modal step (3 towers -> TriClipLoss -> backward) against the reference's own step on the
    tiny golden model: loss value and every gradient of the unlocked set (adapter + blocks + logit_scale).
    res_dtype = dtype of the trainable tower's residual stream and residual-gradient stream (bf16 = the reference's
    amp_bf16 autocast, what bench.py runs; the reference gradients here are its fp32 ones)."""
    from vitlens_hip import engine as E, step as ST
    sd, ins, outs, grads, tc, lc = _tiny_depth()
    _, _, _, _, meta = split(load_npz("tiny_depth.npz"))
    _, text, _ = specs_from_meta(meta)
    xc = E.TextCfg(context_length=text.context_length, vocab_size=text.vocab_size, width=text.width, heads=text.heads,
                   layers=text.layers, embed_dim=text.embed_dim)
    st = ST.TriModalDepthStep(sd, tc, xc, "cuda", micro_batch=2, unlock_first_n=tc.layers, lr=1e-3, train_res_dtype=res_dtype,
                              frozen_res_dtype=res_dtype)
    loss = st.forward_backward(ins["image"].cuda(), ins["text"].cuda(), ins["visual_x"].cuda())
    assert abs(float(loss) - float(outs["step_loss"])) < 2e-2, (float(loss), float(outs["step_loss"]))
    n = 0
    for name, g in st.grads.items():
        if name == "logit_scale":
            ref = grads["logit_scale"].reshape(1)
        elif name.endswith("conv1.weight_gemm"):
            ref = grads["visual.visual_adapter.conv1.weight"].reshape(g.shape[0], -1); g = g[:, :ref.shape[1]]
        else:
            ref = grads[name]
        assert relerr(g, ref) < 6e-2, (name, relerr(g, ref))
        n += 1
    assert n == 12 * tc.layers + 3
    # the update moves the masters and refreshes the bf16 operands; a second step still runs and the loss drops or stays
    before = st.masters["visual.transformer.resblocks.0.mlp.c_fc.weight"].clone()
    st.optimizer_step()
    assert float((st.masters["visual.transformer.resblocks.0.mlp.c_fc.weight"] - before).abs().max()) > 0
    loss2 = st.step(ins["image"].cuda(), ins["text"].cuda(), ins["visual_x"].cuda())
    assert torch.isfinite(loss2) and float(loss2) < float(loss) + 1e-3


@pytest.mark.parametrize("recipe", ["depth", "audio", "pc"])
def test_steps_with_the_frozen_towers_on_a_second_stream(recipe):
    """`overlap_frozen=True` (the steps' default since round 6): the frozen image / text towers' forwards run on a second HIP
    stream beside the trainable tower's forward.  Same kernels, same operands: loss, every gradient and the masters after
    the optimizer step are BIT-equal to the serial step, over several steps (a missing event would show as a stale or
    half-written feature) - for the depth (C3), audio (C4) and point-cloud (C5) steps."""
    import importlib, json, os, sys, tempfile
    from types import SimpleNamespace
    from vitlens_hip import engine as E, step as ST
    sd, ins, outs, grads, meta = split(load_npz(f"tiny_{recipe}.npz"))
    tower, text, lens = specs_from_meta(meta)
    tc = E.TowerCfg(width=tower.width, layers=tower.layers, heads=tower.heads, patch=tower.patch, image_size=tower.image_size,
                    embed_dim=tower.embed_dim)
    xc = E.TextCfg(context_length=text.context_length, vocab_size=text.vocab_size, width=text.width, heads=text.heads,
                   layers=text.layers, embed_dim=text.embed_dim)
    lc = E.LensCfg(**{k: getattr(lens, k) for k in E.LensCfg.__dataclass_fields__ if hasattr(lens, k)})
    img, txt, vis = ins["image"].cuda(), ins["text"].cuda(), ins["visual_x"].cuda()

    def make(overlap):
        if recipe == "depth":
            st = ST.TriModalDepthStep(sd, tc, xc, "cuda", micro_batch=2, unlock_first_n=1, lr=1e-3, overlap_frozen=overlap)
            return st, lambda: st.forward_backward(img, txt, vis)
        if recipe == "audio":
            st = ST.DualAudioStep(sd, tc, xc, lc, "cuda", micro_batch=2, lr=1e-3, overlap_frozen=overlap)
            return st, lambda: st.forward_backward(vis, txt)
        st = ST.TriModalPCStep(sd, tc, xc, lc, "cuda", micro_batch=4, lr=1e-3, bn_training=True, overlap_frozen=overlap)
        return st, lambda: st.forward_backward(img, txt, vis, ins["fps_start"].cuda())
    assert make(True)[0]._overlap_active and not make(False)[0]._overlap_active
    if recipe == "depth":
        assert ST.TriModalDepthStep(sd, tc, xc, "cuda", micro_batch=2, unlock_first_n=1).overlap_frozen, "overlap is the default"
    runs = {}
    for overlap in (False, True):
        st, fb = make(overlap)
        losses = []
        for it in range(4):
            losses.append(float(fb()))
            g = {k: v.clone() for k, v in st.grads.items()} if it == 0 else g
            st.optimizer_step()
        torch.cuda.synchronize()
        runs[overlap] = (losses, g, {k: v.clone() for k, v in st.masters.items()})
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    for k in runs[False][1]:
        assert torch.equal(runs[True][1][k], runs[False][1][k]), k
    for k in runs[False][2]:
        assert torch.equal(runs[True][2][k], runs[False][2][k]), k


def test_trained_blocks_carry_no_stale_layernorm_folds():
    """Advisor finding of round 5: with LayerNorm folding on, `prep_block` builds folded operands (bf16(W*gamma), b + W.beta,
    row sums) for EVERY block of the trainable tower - and an AdamW step moves W, gamma and beta of the unlocked blocks, so an
    inference through the step's own engine on a bf16 stream (`st.lens.encode`) used the construction-time folds for exactly
    the blocks that were trained.  Now a fused step strips the folded operands from the blocks it trains and `run_blocks`
    decides per block.  Width 512 on a bf16 stream = the folded path really runs for the frozen blocks (K >= 512); after
    three steps at a large learning rate the step's engine must agree with a FRESH engine built from `st.state_dict()`."""
    from vitlens_hip import engine as E, step as ST
    assert E.LN_FOLD
    g = torch.Generator().manual_seed(5)
    tspec = O.TowerSpec(width=512, layers=3, heads=8, patch=14, image_size=224, embed_dim=256)
    xspec = O.TextSpec(context_length=16, vocab_size=300, width=512, heads=8, layers=1, embed_dim=256)
    sd = O.init_tower(tspec, g, "image.")
    sd.update(O.init_tower(tspec, g, "visual."))
    sd.pop("visual.conv1.weight")
    sd.update(O.init_text(xspec, g))
    sd["visual.visual_adapter.conv1.weight"] = (torch.rand(512, 1, 14, 14, generator=g) * 2 - 1) / 14.0
    sd["visual.visual_adapter.pos_emb"] = torch.randn(256, 512, generator=g) * 512 ** -0.5
    sd["logit_scale"] = torch.tensor(2.659)
    tc = E.TowerCfg(width=512, layers=3, heads=8, embed_dim=256)
    xc = E.TextCfg(context_length=16, vocab_size=300, width=512, heads=8, layers=1, embed_dim=256)
    B = 4
    img = torch.randn(B, 3, 224, 224, generator=g).cuda(); dep = torch.randn(B, 1, 224, 224, generator=g).cuda()
    txt = torch.randint(1, 298, (B, 16), generator=g); txt[:, 7] = 299; txt = txt.cuda()
    st = ST.TriModalDepthStep(sd, tc, xc, "cuda", micro_batch=B, unlock_first_n=2, lr=2e-2, train_res_dtype=torch.bfloat16,
                              frozen_res_dtype=torch.bfloat16)
    assert all("in_f" not in st.lens.vit.blocks[l] for l in range(2)) and "in_f" in st.lens.vit.blocks[2]
    f0 = st.lens.encode(dep, normalize=True).clone()
    for _ in range(3):
        st.step(img, txt, dep)
    got = st.lens.encode(dep, normalize=True)
    assert relerr(got, f0) > 5e-2, "the steps did not move the tower: the test would prove nothing"
    fresh = E.LensEngine(st.state_dict(), "visual.", tc, E.LensCfg(modality="depth", perceiver_identity=True), "cuda",
                         res_dtype=torch.bfloat16)
    assert "in_f" in fresh.vit.blocks[0]                      # the fresh engine folds all three blocks (from the trained weights)
    want = fresh.encode(dep, normalize=True)
    assert relerr(got, want) < 2e-2, relerr(got, want)       # folded vs unfolded LayerNorm of the same weights: bf16 noise only


@pytest.mark.parametrize("recipe", ["depth", "audio", "pc"])
def test_step_checkpoint_export_reload_and_resume(recipe):
    """Train one step with a fused step object, export `state_dict()` (reference names / layouts: conv weight un-padded,
    GEGLU de-interleaved, to_qkv split) and load it into a fresh TriCLIP: the drop-in forward must reproduce the step's own
    tower.  Then resume a NEW step object from (state_dict, optimizer_state_dict): its next step equals the original's."""
    import importlib, json, os, sys, tempfile
    from types import SimpleNamespace
    from vitlens_hip import engine as E, step as ST
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        if "vit-lens_amd" not in (getattr(sys.modules[k], "__file__", "") or ""):
            del sys.modules[k]
    oc = importlib.import_module("open_clip")
    case = load_npz(f"tiny_{recipe}.npz")
    sd, ins, outs, grads, meta = split(case)
    tower, text, lens = specs_from_meta(meta)
    tc = E.TowerCfg(width=tower.width, layers=tower.layers, heads=tower.heads, patch=tower.patch, image_size=tower.image_size,
                    embed_dim=tower.embed_dim)
    xc = E.TextCfg(context_length=text.context_length, vocab_size=text.vocab_size, width=text.width, heads=text.heads,
                   layers=text.layers, embed_dim=text.embed_dim)
    sd = {k: v.cuda() if v.is_floating_point() else v for k, v in sd.items()}      # on-device fp32 input: the aliasing case
    before = {k: v.clone() for k, v in sd.items()}

    def make(state):
        if recipe == "depth":
            return ST.TriModalDepthStep(state, tc, xc, "cuda", micro_batch=2, unlock_first_n=1, lr=1e-3)
        with tempfile.TemporaryDirectory() as td:
            json.dump(meta["model_cfg"], open(os.path.join(td, "tiny-lens.json"), "w"))
            oc.add_model_config(td)
            m = oc.tri_create_model("tiny-lens", None, device="cpu", args=SimpleNamespace(**meta["args"]))
        _, lc = m.visual._cfgs()
        if recipe == "audio":
            return ST.DualAudioStep(state, tc, xc, lc, "cuda", micro_batch=2, lr=1e-3)
        return ST.TriModalPCStep(state, tc, xc, lc, "cuda", micro_batch=4, lr=1e-3, bn_training=True)

    def run(st):
        if recipe == "depth":
            return st.step(ins["image"].cuda(), ins["text"].cuda(), ins["visual_x"].cuda())
        if recipe == "audio":
            return st.step(ins["visual_x"].cuda(), ins["text"].cuda())
        return st.step(ins["image"].cuda(), ins["text"].cuda(), ins["visual_x"].cuda(), ins["fps_start"].cuda())
    st = make(sd)
    run(st)
    assert all(torch.equal(before[k], sd[k]) for k in sd), "the step modified the caller's state_dict in place"
    exported = st.state_dict()
    assert exported.keys() == sd.keys() and all(exported[k].shape == sd[k].shape for k in sd)
    changed = [k for k in sd if not torch.equal(exported[k].cpu(), before[k].cpu())]
    assert any(k.startswith("visual.visual_adapter.") for k in changed) and "logit_scale" in changed
    assert not [k for k in changed if k.startswith("image.") or k.startswith("transformer.")]
    # drop-in model on the exported weights == the step's own tower
    with tempfile.TemporaryDirectory() as td:
        json.dump(meta["model_cfg"], open(os.path.join(td, "tiny-lens.json"), "w"))
        oc.add_model_config(td)
        model = oc.tri_create_model("tiny-lens", None, device="cuda", args=SimpleNamespace(**meta["args"]))
    missing = model.load_state_dict(exported, strict=False)
    assert not missing.unexpected_keys and not [k for k in missing.missing_keys if not k.endswith("num_batches_tracked")]
    model.eval()
    model.visual.arith_f32 = False      # compare like with like: the step's tower runs bf16 operands (default precision "fp32" + eval()
    #                                     would route the depth tower through the fp32-arithmetic executor, tests/test_hip_f32.py)
    kw = {"fps_start": ins["fps_start"].cuda()} if recipe == "pc" else {}
    with torch.no_grad():
        got = model.encode_visual(ins["visual_x"].cuda(), normalize=True, **kw)
    if recipe == "pc":      # the step's engine-side tokenizer keeps construction-time weights; compare through a fresh step instead
        ref = make(exported).lens
        ref_f = E.LensEngine(exported, "visual.", tc, ref.lens, "cuda").encode(ins["visual_x"].cuda(), normalize=True, **kw)
    else:
        ref_f = st.lens.encode(ins["visual_x"].cuda(), normalize=True)
    assert relerr(got, ref_f) < 2e-3, relerr(got, ref_f)
    # resume
    st2 = make(exported)
    st2.load_optimizer_state_dict(st.optimizer_state_dict())
    l_a, l_b = run(st), run(st2)
    assert abs(float(l_a) - float(l_b)) < 1e-5
    for k in st.masters:
        assert relerr(st2.masters[k], st.masters[k]) < 1e-5, k
    # load_state_dict into an existing object restores the trainable set
    st2.load_state_dict(sd)
    for k, v in st2.state_dict().items():
        assert relerr(v.float(), before[k].float()) < 1e-6 or v.dtype == torch.long, k


@pytest.mark.parametrize("case", ["audio", "audio_tied"])
def test_audio_lens_backward_vs_reference_grads(case):
    """Audio recipe: AST tokenizer + Perceiver (cross + self attention, GEGLU FF) trainable, ViT locked, cls unlocked.
    Every gradient the HIP backward produces vs the reference's own autograd on the tiny golden model.  `audio_tied`:
    perceiver_weight_tie_layers with depth 3 - layers 1 and 2 are the same modules, their gradients arrive summed under
    layer 1's names (as named_parameters() of the reference lists them)."""
    from vitlens_hip import engine as E, train as TR
    sd, ins, outs, grads, meta = split(load_npz(f"tiny_{case}.npz"))
    tower, text, lens = specs_from_meta(meta)
    tc = E.TowerCfg(width=tower.width, layers=tower.layers, heads=tower.heads, patch=tower.patch,
                    image_size=tower.image_size, embed_dim=tower.embed_dim)
    lc = E.LensCfg(**{k: getattr(lens, k) for k in E.LensCfg.__dataclass_fields__ if hasattr(lens, k)})
    le = E.LensEngine(sd, "visual.", tc, lc, "cuda")
    tr = TR.AudioLensTrainer(le)
    feat = tr.forward(ins["visual_x"].cuda())
    assert relerr(feat, outs["visual_raw"]) < 3e-2, relerr(feat, outs["visual_raw"])
    v = outs["visual_raw"].clone().requires_grad_(True)
    loss = O.tri_clip_loss(outs["image_features"], outs["text_features"], O.l2_normalize(v), outs["logit_scale"])
    loss.backward()
    tr.backward(v.grad.cuda())
    got = tr.perc.reference_named_grads()
    n = 0
    for name, g in got.items():
        if name.endswith("conv1.weight_gemm"):
            ref = grads["visual.visual_adapter.conv1.weight"].reshape(g.shape[0], -1); g = g[:, :ref.shape[1]]
        else:
            assert name in grads, name
            ref = grads[name]
        assert g.shape == ref.shape, (name, g.shape, ref.shape)
        e = relerr(g, ref)
        assert e < 6e-2, (name, e)
        n += 1
    # latents, adapter (2), cls, per layer: cross attn 3 w + 1 b + 2 LN x2, ff 2w+2b+LN2, selfs ...
    assert n >= 40, n
    assert "visual.class_embedding" in got and "visual.perceiver.latents" in got
    if case == "audio_tied":
        assert not any(".layers.2." in k for k in got) and le.perceiver.layers[2] is le.perceiver.layers[1]


@pytest.mark.parametrize("case", ["audio", "audio_tied"])
def test_dual_audio_step_runs_and_matches_reference_loss(case):
    """Audio <-> text dual step (ClipLossGeneral): loss vs the reference's value on the tiny golden model, then two
    optimizer steps (masters move, bf16 operands + transposes refreshed, loss does not increase).  Tied Perceiver layers:
    one master per shared tensor, the exported state_dict repeats it under every tied layer index."""
    from vitlens_hip import engine as E, step as ST
    sd, ins, outs, grads, meta = split(load_npz(f"tiny_{case}.npz"))
    tower, text, lens = specs_from_meta(meta)
    tc = E.TowerCfg(width=tower.width, layers=tower.layers, heads=tower.heads, patch=tower.patch,
                    image_size=tower.image_size, embed_dim=tower.embed_dim)
    xc = E.TextCfg(context_length=text.context_length, vocab_size=text.vocab_size, width=text.width, heads=text.heads,
                   layers=text.layers, embed_dim=text.embed_dim)
    lc = E.LensCfg(**{k: getattr(lens, k) for k in E.LensCfg.__dataclass_fields__ if hasattr(lens, k)})
    st = ST.DualAudioStep(sd, tc, xc, lc, "cuda", micro_batch=2, lr=1e-3)
    loss = st.forward_backward(ins["visual_x"].cuda(), ins["text"].cuda())
    assert abs(float(loss) - float(outs["dual_loss"])) < 3e-2, (float(loss), float(outs["dual_loss"]))
    assert all(torch.isfinite(g).all() for g in st.grads.values())
    st.optimizer_step()
    l2 = st.step(ins["visual_x"].cuda(), ins["text"].cuda())
    l3 = st.step(ins["visual_x"].cuda(), ins["text"].cuda())
    assert torch.isfinite(l3) and float(l3) < float(loss) + 1e-3, (float(loss), float(l2), float(l3))
    if case == "audio_tied":
        assert not any(".layers.2." in k for k in st.masters)
        out = st.state_dict()
        assert set(out) == set(sd)
        k1 = "visual.perceiver.layers.1.0.fn.to_q.weight"
        assert torch.equal(out[k1], out[k1.replace(".layers.1.", ".layers.2.")]) and not torch.equal(out[k1].float().cpu(), sd[k1].float())


# ------------------------------------------------------------------------------------------------ point-cloud Lens
def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("R,C", [(512, 128), (4100, 512), (333, 64)])
def test_batchnorm_stats_apply_backward_vs_torch(R, C):
    """vl_bn_* against torch's fp32 F.batch_norm + autograd on the same bf16-rounded input: batch statistics,
    running-stat update (momentum 0.1, unbiased variance), fused ReLU, train- and eval-mode backward."""
    from vitlens_hip import ops
    F = torch.nn.functional
    x = (_rnd(R, C, seed=1) * 0.7 + _rnd(C, seed=2) * 3.0).bfloat16()          # columns with |mean| >> std
    gamma = 1 + 0.1 * _rnd(C, seed=3); beta = 0.1 * _rnd(C, seed=4)
    rm0 = 0.1 * _rnd(C, seed=5); rv0 = 1 + 0.2 * torch.rand(C, generator=torch.Generator().manual_seed(6))
    dy = _rnd(R, C, seed=7).bfloat16()
    for train in (True, False):
        xr = x.float().requires_grad_(True); g = gamma.clone().requires_grad_(True); b = beta.clone().requires_grad_(True)
        rm, rv = rm0.clone(), rv0.clone()
        y = torch.relu(F.batch_norm(xr, rm, rv, g, b, training=train, momentum=0.1, eps=1e-5))
        y.backward(dy.float())
        rmd, rvd = rm0.clone().cuda(), rv0.clone().cuda()
        xd = x.cuda()
        if train:
            mean, var = ops.bn_stats(xd, rmd, rvd, 0.1)
            assert relerr(mean, xr.detach().mean(0)) < 1e-5 and relerr(var, xr.detach().var(0, unbiased=False)) < 1e-4
            assert relerr(rmd, rm) < 1e-5 and relerr(rvd, rv) < 1e-4
        else:
            mean, var = rmd, rvd
        gd, bd = gamma.cuda(), beta.cuda()
        yd = ops.bn_apply(xd, mean, var, gd, bd, relu=True)
        assert relerr(yd, y.detach()) < 4e-3
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        dx = ops.bn_bwd(dy.cuda(), xd, mean, var, gd, bd, dg, db, relu=True, train=train)
        assert relerr(dg, g.grad) < 2e-3 and relerr(db, b.grad) < 2e-3, (train, relerr(dg, g.grad), relerr(db, b.grad))
        assert relerr(dx, xr.grad) < 6e-3, (train, relerr(dx, xr.grad))
        ops.bn_bwd(dy.cuda(), xd, mean, var, gd, bd, dg, db, relu=True, train=train, need_dx=False)     # accumulates
        assert relerr(dg, 2 * g.grad) < 2e-3


def test_group_max_backward_and_group_sum():
    from vitlens_hip import ops
    G, M, C = 37, 8, 96
    f = _rnd(G * M, C, seed=11).bfloat16(); dg = _rnd(G, C, seed=12).bfloat16(); base = _rnd(G * M, C, seed=13).bfloat16()
    fr = f.float().view(G, M, C).requires_grad_(True)
    fr.max(dim=1).values.backward(dg.float())
    got = ops.group_max_bwd(f.cuda(), dg.cuda(), M)
    assert torch.equal(got.float().cpu(), fr.grad.reshape(G * M, C))                 # one-hot scatter: exact
    got = ops.group_max_bwd(f.cuda(), dg.cuda(), M, base=base.cuda())
    assert relerr(got, fr.grad.reshape(G * M, C) + base.float()) < 4e-3
    s = ops.group_sum(f.cuda(), M)
    assert relerr(s, f.float().view(G, M, C).sum(1)) < 4e-3


# Parameters whose gradient is identically zero under a train-mode BatchNorm (a per-channel constant added in front of
# it is removed by the batch mean): both sides hold round-off only.
_PC_ZERO_GRAD = ("first_conv.0.bias", "first_conv.3.bias", "second_conv.0.bias")


def _pc_tol(name):
    """The mini-PointNet takes two max-pools over bf16 activations: where the top two candidates of a group differ by
    less than a bf16 ulp the arg-max - and with it the whole row the gradient is routed to - flips.  The reference's own
    `--precision amp_bf16` recipe shows the same effect: autocast-bf16 vs fp32 autograd of the reference tokenizer on this
    very case differ by 10-28 % (relative L2) on the encoder gradients, 0.4 % on pos_embed (measured with
    torch.autocast('cpu') on the oracle).  The fp32 restatement of the kernel sequence is exact to 1e-6, and its
    bf16-rounded emulation reproduces the error level measured here (9-15 %), so the bound below separates "bf16
    routing noise" from a wrong formula (a missing term or sign gives >= 1)."""
    return 6e-2 if ("pos_embed" in name or "reduce_dim" in name) else 0.30


def _pc_cfgs():
    from vitlens_hip import engine as E
    sd, ins, outs, grads, meta = split(load_npz("tiny_pc.npz"))
    tower, text, lens = specs_from_meta(meta)
    tc = E.TowerCfg(width=tower.width, layers=tower.layers, heads=tower.heads, patch=tower.patch,
                    image_size=tower.image_size, embed_dim=tower.embed_dim)
    xc = E.TextCfg(context_length=text.context_length, vocab_size=text.vocab_size, width=text.width, heads=text.heads,
                   layers=text.layers, embed_dim=text.embed_dim)
    lc = E.LensCfg(**{k: getattr(lens, k) for k in E.LensCfg.__dataclass_fields__ if hasattr(lens, k)})
    return sd, ins, outs, grads, tc, xc, lc


@pytest.mark.parametrize("bn_train", [False, True])
def test_point_tokenizer_backward_vs_oracle_autograd(bn_train):
    """PointTokenizerTrainer alone: tokens+pos and every parameter gradient for a random upstream gradient vs autograd
    through the oracle's point_tokens (pinned to the reference in tests/test_oracle_golden.py)."""
    from vitlens_hip.points import PointTokenizerTrainer
    sd, ins, outs, grads, tc, xc, lc = _pc_cfgs()
    a = "visual.visual_adapter."
    _, _, lens = specs_from_meta(split(load_npz("tiny_pc.npz"))[4])
    tr = PointTokenizerTrainer(sd, a, lc, "cuda", bn_training=bn_train)
    out = tr.forward(ins["visual_x"].cuda(), ins["fps_start"].cuda())
    sdr = {k: (v.clone().requires_grad_(True) if (k.startswith(a) and "running" not in k) else v) for k, v in sd.items()}
    tok, pos, _, _ = O.point_tokens(sdr, "visual.", ins["visual_x"], lens, ins["fps_start"], training=bn_train)
    ref = (tok + pos).reshape(-1, tok.shape[-1])
    assert relerr(out, ref.detach()) < 2e-2, relerr(out, ref.detach())
    dctx = _rnd(*ref.shape, seed=21)
    ref.backward(dctx)
    tr.backward(dctx.cuda())
    errs = {}
    for name, g in tr.grads.items():
        if bn_train and name.endswith(_PC_ZERO_GRAD):
            continue
        errs[name[len(a):]] = round(relerr(g, sdr[name].grad.reshape(g.shape)), 4)
    print(sorted(errs.items()))
    bad = {k: v for k, v in errs.items() if v >= _pc_tol(k)}
    assert len(errs) >= 15 and not bad, bad


def _point_tokens_routed(sd, a, g, centers, lens, idx1, idx2, training, gate1=None, gate2=None):
    """The oracle's mini-PointNet (vitlens_oracle.point_tokens = dvae.py:196-212) on given grouped patches, with both
    max-pools replaced by a gather at GIVEN arg-max indices (and, optionally, both ReLUs by GIVEN 0/1 gates): the gradient
    then follows exactly the routing the forward chose."""
    M = lens.pc_group_size
    def conv1(x, name):
        return torch.einsum("oc,bcn->bon", sd[a + name + ".weight"][:, :, 0], x) + sd[a + name + ".bias"].view(1, -1, 1)
    act = lambda x, gate: torch.relu(x) if gate is None else x * gate
    f = conv1(g, "encoder.first_conv.0")
    f = act(O.batch_norm_1d(f, sd, a + "encoder.first_conv.1.", training), gate1)
    f = conv1(f, "encoder.first_conv.3")
    fg = f.gather(2, idx1[:, :, None])
    f = torch.cat([fg.expand(-1, -1, M), f], dim=1)
    f = conv1(f, "encoder.second_conv.0")
    f = act(O.batch_norm_1d(f, sd, a + "encoder.second_conv.1.", training), gate2)
    f = conv1(f, "encoder.second_conv.3")
    tok = f.gather(2, idx2[:, :, None]).squeeze(2)
    tok = O.linear(tok, sd[a + "reduce_dim.weight"], sd[a + "reduce_dim.bias"])
    pos = O.linear(O.gelu_erf(O.linear(centers, sd[a + "pos_embed.0.weight"], sd[a + "pos_embed.0.bias"])),
                   sd[a + "pos_embed.2.weight"], sd[a + "pos_embed.2.bias"])
    return tok + pos


@pytest.mark.parametrize("bn_train", [False, True])
def test_point_tokenizer_backward_given_forward_routing(bn_train):
    """Round-1 finding: the 0.30 bound of the test above cannot tell a 25 % bug from arg-max routing noise.  Here the
    routing is taken out of the comparison: the arg-max indices of both max-pools are read from the HIP forward's own
    (bf16) activations and the oracle's autograd is evaluated WITH THOSE INDICES on the same grouped patches - every
    tokenizer gradient must then agree to 6e-2 (bf16 operands), the bound used for every other trainable tensor.
    With train-mode BatchNorm the batch statistics come from bf16 activations, which moves pre-activations across zero: the
    ReLU gates are a second discrete routing decision and are taken from the forward as well in that case."""
    from vitlens_hip.points import PointTokenizerTrainer
    sd, ins, outs, grads, tc, xc, lc = _pc_cfgs()
    a = "visual.visual_adapter."
    _, _, lens = specs_from_meta(split(load_npz("tiny_pc.npz"))[4])
    M = lens.pc_group_size
    tr = PointTokenizerTrainer(sd, a, lc, "cuda", bn_training=bn_train)
    out = tr.forward(ins["visual_x"].cuda(), ins["fps_start"].cuda())
    patches, h1, f, h2, f2, c3 = tr.ctx[0], tr.ctx[4], tr.ctx[5], tr.ctx[10], tr.ctx[11], tr.ctx[13]
    BG = f.shape[0] // M
    gates = (None, None)
    if bn_train:
        gates = tuple((h.float() > 0).float().cpu().view(BG, M, -1).transpose(1, 2).contiguous() for h in (h1, h2))
    idx1 = f.float().view(BG, M, -1).argmax(dim=1).cpu()           # first maximum, as group_max_bwd_kernel
    idx2 = f2.float().view(BG, M, -1).argmax(dim=1).cpu()
    g = patches[:, :3].float().cpu().view(BG, M, 3).transpose(1, 2).contiguous()      # the kernel's own (bf16) patch coordinates
    centers = c3[:, :3].float().cpu()
    sdr = {k: (v.clone().requires_grad_(True) if (k.startswith(a) and "running" not in k) else v) for k, v in sd.items()}
    ref = _point_tokens_routed(sdr, a, g, centers, lens, idx1, idx2, bn_train, *gates)
    assert relerr(out, ref.detach()) < 2e-2, relerr(out, ref.detach())
    dctx = _rnd(*ref.shape, seed=21)
    ref.backward(dctx)
    tr.backward(dctx.cuda())
    errs = {}
    for name, gr in tr.grads.items():
        if bn_train and name.endswith(_PC_ZERO_GRAD):
            continue
        errs[name[len(a):]] = round(relerr(gr, sdr[name].grad.reshape(gr.shape)), 4)
    print(sorted(errs.items()))
    bad = {k: v for k, v in errs.items() if v >= 6e-2}
    assert len(errs) >= 15 and not bad, bad


@pytest.mark.parametrize("bn_train", [False, True])
def test_pc_tri_modal_step_vs_reference_grads(bn_train):
    """Point-cloud recipe on the tiny golden model: loss and EVERY gradient of the trainable Lens (PointBERT
    tokenizer incl. BatchNorm affine, Perceiver, cls) vs the reference's own autograd - with BatchNorm frozen at the
    running statistics (tiny_pc.npz) and in train mode (tiny_pc_bntrain.npz, incl. the running-stat update)."""
    from vitlens_hip import step as ST
    sd, ins, outs, grads, tc, xc, lc = _pc_cfgs()
    if bn_train:
        z = load_npz("tiny_pc_bntrain.npz")
        _, _, outs, grads, _ = split(z)
    st = ST.TriModalPCStep(sd, tc, xc, lc, "cuda", micro_batch=4, lr=1e-3, bn_training=bn_train, unlock_cls=True)
    args = (ins["image"].cuda(), ins["text"].cuda(), ins["visual_x"].cuda(), ins["fps_start"].cuda())
    loss = st.forward_backward(*args)
    assert abs(float(loss) - float(outs["step_loss"])) < 3e-2, (float(loss), float(outs["step_loss"]))
    got = dict(st.grads)
    got.update(st.reference_named_grads())
    n = 0
    errs = {}
    for name, ref in grads.items():
        if name not in got:
            continue
        g = got[name]
        ref = ref.reshape(g.shape)
        if bn_train and name.endswith(_PC_ZERO_GRAD):
            continue
        errs[name] = round(relerr(g, ref), 4)
        n += 1
    print(sorted(errs.items()))
    # (train-mode BatchNorm over 4 samples amplifies every rounding difference upstream: the first cross-attention's latent
    #  LayerNorm gradients sit at 0.07-0.083 there - round 4 measured 0.0828 after the text tower moved to two-term weights,
    #  i.e. with MORE accurate text features - against 0.03-0.05 with frozen statistics)
    other = 1e-1 if bn_train else 8e-2
    bad = {k: v for k, v in errs.items() if v >= (_pc_tol(k) if "visual_adapter" in k else other)}
    assert not bad, bad
    assert n >= 60, n
    for k in ("encoder.first_conv.0.weight", "encoder.first_conv.1.weight", "encoder.second_conv.1.bias", "reduce_dim.weight",
              "pos_embed.0.weight", "pos_embed.2.bias"):
        assert "visual.visual_adapter." + k in got
    if bn_train:
        for k, v in z.items():
            if k.startswith("sd_after/"):
                rm = st.tok.running[k[9:].replace("visual.visual_adapter.", "").rsplit(".", 1)[0]][0 if k.endswith("mean") else 1]
                assert relerr(rm, torch.from_numpy(v)) < 5e-3, k
    # two optimizer steps: masters move, operands refreshed, loss goes down on the same batch
    st.optimizer_step()
    l2 = st.step(*args); l3 = st.step(*args)
    assert torch.isfinite(l3) and float(l3) < float(loss) + 1e-3, (float(loss), float(l2), float(l3))


# ------------------------------------------------------------------------------------------------ two ranks on one GPU
class _ThreadComm:
    """In-process stand-in for TorchComm: `world` threads (one per rank, sharing the GPU and its default stream) meet at
    a barrier; semantics of all_gather_into_tensor (rank-major) and all_reduce(SUM)."""

    def __init__(self, world):
        import threading
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)
        self.local = threading.local()

    def bind(self, rank):
        self.local.rank = rank

    def all_gather(self, out, inp):
        self.slots[self.local.rank] = inp
        self.barrier.wait()
        out.copy_(torch.cat(self.slots, dim=0))
        self.barrier.wait()

    def all_reduce_sum(self, t):
        self.slots[self.local.rank] = t
        self.barrier.wait()
        total = sum(s.clone() for s in self.slots)
        self.barrier.wait()
        t.copy_(total)
        self.barrier.wait()

    def reduce_scatter_sum(self, out, inp):
        r, b = self.local.rank, out.shape[0]
        self.slots[r] = inp
        self.barrier.wait()
        out.copy_(sum(s[r * b:(r + 1) * b] for s in self.slots))
        self.barrier.wait()


def _run_ranks(world, fn):
    """fn(rank, comm) on `world` threads sharing the GPU; returns the per-rank results, re-raises worker errors."""
    import threading
    comm = _ThreadComm(world)
    res, errs = [None] * world, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            comm.bind(r)
            res[r] = fn(r, comm)
        except Exception as e:
            errs.append(e); comm.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert not errs, errs
    return res


@pytest.mark.parametrize("world", [4])        # the world-2 fixture has 6 global rows; the logits GEMM wants N % 4 == 0
@pytest.mark.parametrize("local_loss,gather_with_grad", [(False, False), (False, True), (True, False), (True, True)])
def test_distributed_contrastive_pair_vs_reference_ranks(world, local_loss, gather_with_grad):
    """ClipLossGeneral across ranks, all four (local_loss, gather_with_grad) modes: the loss each rank reports and the
    gradients that arrive at its local features (incl. the reduce-scatter of a differentiable gather) against what the
    REFERENCE computed per rank on gloo (tests/golden/gather_w{2,4}.npz)."""
    from vitlens_hip import step as ST
    z = {k: torch.from_numpy(v) for k, v in load_npz(f"gather_w{world}.npz").items()}
    scale = 14.285714
    tag = f"dual_ll{int(local_loss)}_gg{int(gather_with_grad)}"

    def fn(r, comm):
        xl, yl = z[f"in/x{r}"].cuda(), z[f"in/y{r}"].cuda()
        b, E = xl.shape
        allp = torch.empty(world * b, 2 * E, device="cuda")
        comm.all_gather(allp, torch.cat([xl, yl], dim=1))
        ax, ay = [t.contiguous() for t in allp.split(E, dim=1)]
        loss, dx, dy, ds = ST.pair_loss_and_grads(comm, r, world, xl, yl, ax, ay, scale, local_loss=local_loss,
                                                  gather_with_grad=gather_with_grad)
        return float(loss), dx.cpu(), dy.cpu(), float(ds)
    for r, (loss, dx, dy, ds) in enumerate(_run_ranks(world, fn)):
        assert abs(loss - float(z[f"rank{r}/{tag}_loss"])) < 2e-3, (r, loss)
        for got, ref in ((dx, z[f"rank{r}/{tag}_gx"]), (dy, z[f"rank{r}/{tag}_gy"])):
            assert float((got - ref).norm() / ref.norm()) < 4e-2, (r, float((got - ref).norm() / ref.norm()))
        ref_s = float(z[f"rank{r}/{tag}_gls"])
        assert abs(ds - ref_s) < 2e-2 * max(1.0, abs(ref_s)), (r, ds, ref_s)


@pytest.mark.parametrize("recipe", ["depth_tri", "audio_dual"])
def test_two_rank_step_equals_global_batch_step(recipe):
    """The N>1 code path of the fused steps on real kernels: two ranks (threads, 2 samples each, packed all-gather +
    flat gradient all-reduce through an in-process communicator) against ONE rank on the 4-sample global batch.
    Every rank must see the global loss; the per-rank gradients must add up to the global-batch gradient (logit_scale:
    every rank holds the full derivative), which is what DDP's mean over ranks scales by 1/W in the reference."""
    import threading
    from vitlens_hip import engine as E, step as ST
    name = "tiny_depth.npz" if recipe == "depth_tri" else "tiny_audio.npz"
    sd, ins, outs, grads, meta = split(load_npz(name))
    tower, text, lens = specs_from_meta(meta)
    tc = E.TowerCfg(width=tower.width, layers=tower.layers, heads=tower.heads, patch=tower.patch,
                    image_size=tower.image_size, embed_dim=tower.embed_dim)
    xc = E.TextCfg(context_length=text.context_length, vocab_size=text.vocab_size, width=text.width, heads=text.heads,
                   layers=text.layers, embed_dim=text.embed_dim)
    lc = E.LensCfg(**{k: getattr(lens, k) for k in E.LensCfg.__dataclass_fields__ if hasattr(lens, k)})

    def make(rank, world, comm=None):
        if recipe == "depth_tri":
            return ST.TriModalDepthStep(sd, tc, xc, "cuda", micro_batch=2, unlock_first_n=1, rank=rank, world_size=world, comm=comm)
        return ST.DualAudioStep(sd, tc, xc, lc, "cuda", micro_batch=2, rank=rank, world_size=world, comm=comm)

    def batch(sl):
        if recipe == "depth_tri":
            return ins["image"][sl].cuda(), ins["text"][sl].cuda(), ins["visual_x"][sl].cuda()
        return ins["visual_x"][sl].cuda(), ins["text"][sl].cuda()

    one = make(0, 1)
    loss1 = float(one.forward_backward(*batch(slice(0, 4))))
    g1 = {k: v.clone() for k, v in one.grads.items()}
    W = 2
    comm = _ThreadComm(W)
    steps = [make(r, W, comm) for r in range(W)]
    res, errs = [None] * W, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            comm.bind(r)
            loss = float(steps[r].forward_backward(*batch(slice(2 * r, 2 * r + 2))))
            local = {k: v.clone() for k, v in steps[r].grads.items()}
            steps[r].optimizer_step()
            res[r] = (loss, local)
        except Exception as e:                      # surface worker failures instead of dead-locking the barrier
            errs.append(e); comm.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    [t.start() for t in th]; [t.join(timeout=300) for t in th]
    assert not errs, errs
    for r in range(W):
        assert abs(res[r][0] - loss1) < 2e-3, (res[r][0], loss1)
    n = 0
    for k, g in g1.items():
        tot = res[0][1][k] + res[1][1][k]
        ref = g * (W if k == "logit_scale" else 1)
        if float(ref.abs().max()) < 1e-6:
            continue
        assert relerr(tot, ref) < 3e-2, (k, relerr(tot, ref))
        n += 1
    assert n >= 10, n
    for k in steps[0].masters:                       # both replicas took the same optimizer step
        assert torch.equal(steps[0].masters[k], steps[1].masters[k]), k
