"""GPU parity of the LayerNorm folding (round 4; include/vitlens_hip.h: vl_gemm_lnfold_bf16, vl_gemm_res_rowstats_bf16,
vl_ln_row_stats): for a frozen pre-LN block (transformer.py:254-272) the LayerNorm is never materialised - the consuming GEMM
reads the raw residual rows and applies (mean, rstd) in its epilogue, the GEMM that produced the rows leaves their partial sums.
Checked against fp32 torch on the same bf16 operands, against the unfolded product path (layernorm + gemm), and at tower level
(engine forward, trainer forward + backward) with the switch on and off."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _rows_with_structure(M, K, seed):
    """Residual-stream-like rows: per-row offsets and scales, a few large channels (bf16)."""
    x = rnd(M, K, seed=seed) * (0.5 + rnd(M, 1, seed=seed + 1).abs() * 2) + rnd(M, 1, seed=seed + 2) * 0.7
    x[:, 5] += 40.0
    x[:, K // 2] -= 25.0
    return x.bfloat16()


@pytest.mark.parametrize("M,N,K,act", [(256 * 65, 1024, 1024, "none"), (256 * 65, 3072, 1024, "none"), (256 * 17, 4096, 1024, "gelu"),
                                       (256 * 17, 4096, 1024, "dsave"), (256 * 64 + 40, 1024, 512, "gelu")])
def test_folded_gemm_equals_layernorm_then_gemm(M, N, K, act):
    """Main rows through the persistent kernel's folded epilogue, leftover rows through layernorm + gemm (ops.gemm_lnfold);
    reference: fp32 LayerNorm of the same bf16 rows, fp32 GEMM on the fp32 weight."""
    from vitlens_hip import ops
    x = _rows_with_structure(M, K, 3).cuda()
    w = rnd(N, K, seed=4, scale=K ** -0.5).cuda()
    b = rnd(N, seed=5, scale=0.2).cuda()
    gamma = (1.0 + 0.3 * rnd(K, seed=6)).cuda()
    beta = (0.2 * rnd(K, seed=7)).cuda()
    mm = ops._fold_rows(x, torch.empty(M, N, device="cuda", dtype=torch.bfloat16), N, K)
    assert 0 < mm <= M and mm % 256 == 0
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    ops.ln_row_stats(None, x, 0, mean, rstd)
    xf = x.float()
    mu = xf.mean(1); var = xf.var(1, unbiased=False)
    assert float((mean - mu).abs().max()) < 1e-5 * (1 + float(mu.abs().max()))
    assert float(((rstd - (var + 1e-5).rsqrt()) * (var + 1e-5).sqrt()).abs().max()) < 1e-5
    fold = ops.fold_ln_linear(w, b, gamma, beta)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    out2 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16) if act == "dsave" else None
    a = {"none": ops.ACT_NONE, "gelu": ops.ACT_GELU, "dsave": ops.ACT_GELU_DSAVE}[act]
    hws = torch.empty(M - mm if M > mm else 1, K, device="cuda", dtype=torch.bfloat16)
    ops.gemm_lnfold(x, fold, mean, rstd, out, w.bfloat16(), b, gamma, beta, hws, act=a, out2=out2)
    pre = torch.nn.functional.layer_norm(xf, (K,), gamma, beta, 1e-5) @ w.t() + b
    ref = pre if act == "none" else torch.nn.functional.gelu(pre)
    assert torch.isfinite(out.float()).all()
    e_fold = relerr(out[:mm], ref[:mm])
    # the unfolded product path on the same rows (bf16 LayerNorm output, bf16 weight)
    h = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    ops.layernorm(x, gamma, beta, h, M, K)
    plain = ops.gemm(h, w.bfloat16(), b, epi=ops.EPI_BF16, act=a if act != "dsave" else ops.ACT_GELU)
    e_plain = relerr(plain[:mm], ref[:mm])
    assert e_fold < 6e-3, (e_fold, e_plain)
    assert e_fold < 1.5 * e_plain + 1e-3, (e_fold, e_plain)      # no worse than the path it replaces
    if mm < M:
        assert relerr(out[mm:], ref[mm:]) < 6e-3
    if act == "dsave":
        t = pre.detach().clone().requires_grad_(True)
        torch.nn.functional.gelu(t).sum().backward()
        assert relerr(out2, t.grad) < 8e-3


@pytest.mark.parametrize("M,N,K,inplace", [(256 * 65, 1024, 1024, True), (256 * 49, 1024, 4096, False), (256 * 64 + 8, 1024, 1024, True)])
def test_residual_gemm_leaves_the_row_statistics_of_what_it_stores(M, N, K, inplace):
    from vitlens_hip import ops
    a = rnd(M, K, seed=1).bfloat16().cuda()
    w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().cuda()
    b = rnd(N, seed=3).cuda()
    res = _rows_with_structure(M, N, 9).cuda()
    want = ops.gemm(a, w, b, res=res, epi=ops.EPI_RES_BF16)                  # the plain residual epilogue
    out = res.clone() if inplace else torch.empty_like(res)
    part = torch.full((M * (N // 64) * 2,), float("nan"), device="cuda")
    mm = ops.gemm_res_rowstats(a, w, b, out, out if inplace else res, part)
    assert 0 < mm <= M
    # same bits as the epilogue without statistics on the rows of the persistent kernel (the leftover rows run on whichever
    # small-tile kernel the dispatcher picks for THAT row count: another summation order)
    assert torch.equal(out[:mm].view(torch.int16), want[:mm].view(torch.int16))
    assert relerr(out, want) < 1e-3
    mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    ops.ln_row_stats(part, out, mm, mean, rstd)
    of = out.float()
    mu = of.mean(1); var = of.var(1, unbiased=False)
    assert float((mean - mu).abs().max()) < 2e-5 * (1 + float(mu.abs().max()))
    assert float((rstd / (var + 1e-5).rsqrt() - 1).abs().max()) < 2e-4
    # deterministic: a second run leaves the same bits
    part2 = torch.empty_like(part)
    out2 = res.clone()
    ops.gemm_res_rowstats(a, w, b, out2, out2 if inplace else res, part2) if inplace else ops.gemm_res_rowstats(a, w, b, out2, res, part2)
    n = mm * (N // 64) * 2
    assert torch.equal(part[:n], part2[:n])


def _vitl_sd(layers, seed=0, width=1024, heads=16):
    import math
    g = torch.Generator().manual_seed(seed)
    D = width
    sd = {"class_embedding": torch.randn(D, generator=g) * D ** -0.5, "positional_embedding": torch.randn(257, D, generator=g) * D ** -0.5,
          "ln_pre.weight": 1 + 0.1 * torch.randn(D, generator=g), "ln_pre.bias": 0.1 * torch.randn(D, generator=g),
          "ln_post.weight": 1 + 0.1 * torch.randn(D, generator=g), "ln_post.bias": 0.1 * torch.randn(D, generator=g),
          "proj": torch.randn(D, 768, generator=g) * D ** -0.5,
          "conv1.weight": torch.randn(D, 3, 14, 14, generator=g) * 0.02}
    for l in range(layers):
        p = f"transformer.resblocks.{l}."
        sd[p + "ln_1.weight"] = 1 + 0.2 * torch.randn(D, generator=g); sd[p + "ln_1.bias"] = 0.1 * torch.randn(D, generator=g)
        sd[p + "ln_2.weight"] = 1 + 0.2 * torch.randn(D, generator=g); sd[p + "ln_2.bias"] = 0.1 * torch.randn(D, generator=g)
        sd[p + "attn.in_proj_weight"] = torch.randn(3 * D, D, generator=g) * D ** -0.5
        sd[p + "attn.in_proj_bias"] = 0.02 * torch.randn(3 * D, generator=g)
        sd[p + "attn.out_proj.weight"] = torch.randn(D, D, generator=g) * D ** -0.5 * (2 * layers) ** -0.5
        sd[p + "attn.out_proj.bias"] = 0.02 * torch.randn(D, generator=g)
        sd[p + "mlp.c_fc.weight"] = torch.randn(4 * D, D, generator=g) * (2 * D) ** -0.5
        sd[p + "mlp.c_fc.bias"] = 0.02 * torch.randn(4 * D, generator=g)
        sd[p + "mlp.c_proj.weight"] = torch.randn(D, 4 * D, generator=g) * D ** -0.5 * (2 * layers) ** -0.5
        sd[p + "mlp.c_proj.bias"] = 0.02 * torch.randn(D, generator=g)
    return {"visual." + k: v for k, v in sd.items()}


def test_tower_forward_and_backward_with_and_without_folding():
    """ViT-L width, 4 blocks (first one trainable), 64 images of 257 tokens (64 whole row tiles + leftover rows): engine forward
    and trainer forward + backward with the LayerNorms folded vs as their own passes - the features, the input gradient and the
    trainable block's gradients agree to bf16 accuracy, and the saved (mean, rstd) of the frozen blocks agree to 1e-4."""
    from vitlens_hip import engine as E, train as T
    cfg = E.TowerCfg(width=1024, layers=4, heads=16, patch=14, image_size=224, embed_dim=768)
    sd = _vitl_sd(4)
    B = 64
    tok = (rnd(B * 256, 1024, seed=11) * 0.5).bfloat16().cuda()
    dfeat = rnd(B, 768, seed=12).cuda()
    res = {}
    keep = E.LN_FOLD                       # (train.py reads engine.LN_FOLD through the module)
    try:
        for fold in (True, False):
            E.LN_FOLD = fold
            eng = E.VitEngine(sd, "visual.", cfg, "cuda", res_dtype=torch.bfloat16)
            f_inf = eng.trunk(tok, B).clone()
            tr = T.TowerTrainer(eng, train_blocks=[0])
            f_tr = tr.forward(tok, B).clone()
            dtok = tr.backward(dfeat).clone()
            S = tr.saved(B, 257)
            res[fold] = (f_inf, f_tr, dtok, {k: v.clone() for k, v in tr.grads.items()}, [[t.clone() for t in st] for st in S.stats])
    finally:
        E.LN_FOLD = keep
    (fi1, ft1, d1, g1, s1), (fi0, ft0, d0, g0, s0) = res[True], res[False]
    assert torch.isfinite(fi1).all() and torch.isfinite(d1).all()
    assert relerr(fi1, fi0) < 1e-2 and relerr(ft1, ft0) < 1e-2, (relerr(fi1, fi0), relerr(ft1, ft0))
    assert relerr(d1, d0) < 3e-2, relerr(d1, d0)
    assert set(g1) == set(g0)
    for k in g1:
        assert relerr(g1[k], g0[k]) < 3e-2, (k, relerr(g1[k], g0[k]))
    for l in (1, 2, 3):          # (mean1, rstd1, mean2, rstd2); the two runs' residual rows themselves differ by bf16 noise
        for m_a, r_a, m_b, r_b in ((s1[l][0], s1[l][1], s0[l][0], s0[l][1]), (s1[l][2], s1[l][3], s0[l][2], s0[l][3])):
            assert float(((m_a - m_b).abs() * r_b).max()) < 2e-3          # in units of the row's standard deviation
            assert float((r_a / r_b - 1).abs().max()) < 5e-3


def test_row_stats_launch_also_normalises_the_leftover_rows():
    """Round 5: `ln_row_stats(..., h_left=, h_row0=)` writes the LayerNorm output of the rows >= h_row0 (the consuming GEMM's
    leftover rows) from the same launch that finalises the statistics - against a separate layernorm of those rows, and the
    statistics unchanged by the extra output; gemm_lnfold(h_ready=True) then equals the path with its own layernorm launch."""
    from vitlens_hip import ops
    M, D, N = 65792, 1024, 3072
    x = (rnd(M, D, seed=21) * 1.5 + 0.2).bfloat16().cuda()
    gamma, beta = (1 + 0.2 * rnd(D, seed=22)).cuda(), (0.1 * rnd(D, seed=23)).cuda()
    w, b = rnd(N, D, seed=24, scale=D ** -0.5).cuda(), rnd(N, seed=25).cuda()
    out_a = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); out_b = torch.empty_like(out_a)
    r0 = ops.fold_rows(x, out_a, N)
    assert r0 == 65536
    mean0, rstd0 = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    mean1, rstd1 = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.ln_row_stats(None, x, 0, mean0, rstd0)
    h = torch.zeros(256, D, device="cuda", dtype=torch.bfloat16)
    ops.ln_row_stats(None, x, 0, mean1, rstd1, ln_w=gamma, ln_b=beta, h_left=h, h_row0=r0)
    assert torch.equal(mean0, mean1) and torch.equal(rstd0, rstd1)
    ref = torch.empty(256, D, device="cuda", dtype=torch.bfloat16)
    ops.layernorm(x[r0:], gamma, beta, ref, 256, D, x_row_stride=x.stride(0))
    assert relerr(h, ref) < 2e-3 and float((h.float() - ref.float()).abs().max()) <= 2.0 ** -6 * float(ref.float().abs().max())
    fold = ops.fold_ln_linear(w, b, gamma, beta)
    hws = torch.zeros(256, D, device="cuda", dtype=torch.bfloat16)
    ops.gemm_lnfold(x, fold, mean0, rstd0, out_a, w.bfloat16(), b, gamma, beta, hws)                     # own layernorm launch
    ops.gemm_lnfold(x, fold, mean1, rstd1, out_b, w.bfloat16(), b, gamma, beta, h, h_ready=True)        # operand left by ln_row_stats
    assert torch.equal(out_a[:r0], out_b[:r0])
    assert relerr(out_b[r0:], out_a[r0:]) < 4e-3
    with pytest.raises(ValueError):
        ops.ln_row_stats(None, x, 0, mean1, rstd1, ln_w=gamma, ln_b=beta, h_left=h[:100], h_row0=r0)
