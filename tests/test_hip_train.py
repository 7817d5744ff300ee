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


@pytest.mark.parametrize("B,L,H,dh,causal", [(2, 257, 4, 64, False), (2, 77, 3, 64, True), (1, 40, 2, 32, False)])
def test_attention_backward(B, L, H, dh, causal):
    from vitlens_hip import ops
    D = H * dh
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B * L, D, generator=g)).bfloat16().cuda()
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).bfloat16().cuda()
    Lp = (L + 7) // 8 * 8
    mk = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device="cuda")
    q, k, v = mk(B, H, L, dh), mk(B, H, L, dh), mk(B, H, L, dh)
    vt, qt, kt = [torch.full((B, H, dh, Lp), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(3)]
    ops.gemm_qkv(x, w, None, q, k, vt, B, L, H, dh, qt=qt, kt=kt, v=v)
    o = mk(B * L, D); lse = torch.empty(B, H, L, device="cuda")
    ops.attn_fwd(q, k, vt, o, lse=lse, causal=causal)
    do_tok = torch.randn(B * L, D, generator=g).bfloat16()
    dO = do_tok.reshape(B, L, H, dh).permute(0, 2, 1, 3).contiguous().cuda()
    dOt = torch.zeros(B, H, dh, Lp, dtype=torch.bfloat16, device="cuda"); dOt[..., :L] = dO.transpose(-1, -2)
    delta = torch.empty(B, H, L, device="cuda")
    ops.attn_delta(dO, o, delta)
    dqkv = mk(B * L, 3 * D)
    ops.attn_bwd(q, k, v, qt, kt, dO, dOt, lse, delta, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, 3 * D, causal=causal)
    # reference: autograd through explicit softmax attention on the same bf16-rounded q,k,v (q un-scaled)
    qr = (q.float().cpu() / (dh ** -0.5 * ops.LOG2E)).requires_grad_(True)
    kr = k.float().cpu().requires_grad_(True); vr = v.float().cpu().requires_grad_(True)
    s = (qr @ kr.transpose(-1, -2)) * dh ** -0.5
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    out = torch.softmax(s, -1) @ vr
    (out * dO.float().cpu()).sum().backward()
    tok = lambda t: t.permute(0, 2, 1, 3).reshape(B * L, D)
    assert relerr(dqkv[:, :D], tok(qr.grad)) < 2e-2, relerr(dqkv[:, :D], tok(qr.grad))
    assert relerr(dqkv[:, D:2 * D], tok(kr.grad)) < 2e-2, relerr(dqkv[:, D:2 * D], tok(kr.grad))
    assert relerr(dqkv[:, 2 * D:], tok(vr.grad)) < 2e-2, relerr(dqkv[:, 2 * D:], tok(vr.grad))


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


def test_tri_modal_step_matches_reference_step():
    """The whole tri-modal step (3 towers -> TriClipLoss -> backward) against the reference's own step on the
    tiny golden model: loss value and every gradient of the unlocked set (adapter + blocks + logit_scale)."""
    from vitlens_hip import engine as E, step as ST
    sd, ins, outs, grads, tc, lc = _tiny_depth()
    _, _, _, _, meta = split(load_npz("tiny_depth.npz"))
    _, text, _ = specs_from_meta(meta)
    xc = E.TextCfg(context_length=text.context_length, vocab_size=text.vocab_size, width=text.width, heads=text.heads,
                   layers=text.layers, embed_dim=text.embed_dim)
    st = ST.TriModalDepthStep(sd, tc, xc, "cuda", micro_batch=2, unlock_first_n=tc.layers, lr=1e-3)
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


def test_audio_lens_backward_vs_reference_grads():
    """Audio recipe: AST tokenizer + Perceiver (cross + self attention, GEGLU FF) trainable, ViT locked, cls unlocked.
    Every gradient the HIP backward produces vs the reference's own autograd on the tiny golden model."""
    from vitlens_hip import engine as E, train as TR
    sd, ins, outs, grads, meta = split(load_npz("tiny_audio.npz"))
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


def test_dual_audio_step_runs_and_matches_reference_loss():
    """Audio <-> text dual step (ClipLossGeneral): loss vs the reference's value on the tiny golden model, then two
    optimizer steps (masters move, bf16 operands + transposes refreshed, loss does not increase)."""
    from vitlens_hip import engine as E, step as ST
    sd, ins, outs, grads, meta = split(load_npz("tiny_audio.npz"))
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
