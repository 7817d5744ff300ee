"""GPU parity of the persistent GEMMs with the store-hidden ("parked") epilogue (vl_gemm_park.hip: cfg=8 eight waves - what
cfg=-1 uses for whole rounds of tiles -, cfg=pcfg four waves) against fp32 PyTorch on
the same bf16 operands, and against the 8-wave kernel (cfg=5).  Every case has several tiles per workgroup or per wave
position so the tile hand-over (DMA prefetch across the tile boundary, epilogue operands, store burst) is exercised."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[8, 10])
def pcfg(request):
    return request.param


def _ops():
    from vitlens_hip import ops
    return ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_p4_identity_is_bit_exact(pcfg):
    """A = [I; 2I; ...] (K = 512): C rows are W^T scaled by powers of two -> exact in bf16; catches any row/column/chunk mix-up
    of the LDS transpose and of the parked-store addressing."""
    ops = _ops()
    K, N, reps = 512, 768, 6
    eye = torch.eye(K)
    a = torch.cat([eye * (2.0 ** r) for r in range(reps)], 0).bfloat16().cuda()              # [3072, 512]
    w = ((torch.arange(N * K).reshape(N, K) * 7 % 251) - 125).float().bfloat16().cuda()
    out = ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=pcfg)
    ref = torch.cat([w.float().cpu().t() * (2.0 ** r) for r in range(reps)], 0)
    assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("M,N,K", [(256 * 40, 4096, 512), (256 * 3, 512, 512), (256 * 257, 256, 768), (2048, 4096, 576)])
def test_p4_bf16_epilogues(M, N, K, pcfg):
    ops = _ops()
    a = rnd(M, K, seed=1).bfloat16().cuda(); w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=3).cuda()
    acc = a.float() @ w.float().t() + bias
    # plain, GELU, ReLU, no bias
    assert relerr(ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=pcfg), acc) < 4e-3
    assert relerr(ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=pcfg, alpha=0.5), 0.5 * (acc - bias)) < 4e-3
    out = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=pcfg)
    ref = torch.nn.functional.gelu(acc)
    assert relerr(out, ref) < 4e-3 and bool(((out.float() - ref).abs() <= ref.abs() * 2.0 ** -7 + 2e-3).all())
    assert relerr(ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_RELU, cfg=pcfg), torch.relu(acc)) < 4e-3
    # GELU with the saved pre-activation (training forward)
    u = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    out = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=pcfg, out2=u)
    assert relerr(u, acc) < 4e-3 and relerr(out, torch.nn.functional.gelu(u.float())) < 4e-3
    # same results as the 8-wave kernel up to the bf16 rounding point of the GELU (fp32 vs bf16-rounded pre-activation)
    old = ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=5)
    assert torch.equal(old, ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=pcfg))


@pytest.mark.parametrize("M,N,K", [(256 * 40, 1024, 1024), (256 * 5, 768, 3072), (256 * 257, 256, 512)])
def test_p4_residual_and_dgelu(M, N, K, pcfg):
    ops = _ops()
    a = rnd(M, K, seed=4).bfloat16().cuda(); w = rnd(N, K, seed=5, scale=K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=6).cuda()
    acc = a.float() @ w.float().t()
    # bf16 residual, out of place and in place (x += ...), bit-identical to the 8-wave kernel
    res = rnd(M, N, seed=7).bfloat16().cuda()
    o1 = ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_BF16, cfg=pcfg)
    assert relerr(o1, res.float() + acc + bias) < 4e-3
    assert torch.equal(o1, ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_BF16, cfg=5))
    x = res.clone()
    ops.gemm(a, w, bias, out=x, res=x, epi=ops.EPI_RES_BF16, cfg=pcfg)
    assert torch.equal(x, o1)
    # dX through GELU
    u = rnd(M, N, seed=9, scale=1.5).bfloat16().cuda()
    out = ops.gemm(a, w, None, res=u, epi=ops.EPI_DGELU, cfg=pcfg, out=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
    uf = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uf).sum().backward()
    assert relerr(out, acc * uf.grad) < 4e-3
    assert torch.equal(out, ops.gemm(a, w, None, res=u, epi=ops.EPI_DGELU, cfg=5, out=torch.empty_like(out)))


@pytest.mark.parametrize("cfg", [8, 10, 0, 1, 5, -1])
def test_gelu_forward_leaves_its_derivative_for_the_dx_gemm(cfg):
    """VL_ACT_GELU_DSAVE: the forward writes out = gelu(pre) and out2 = gelu'(pre) (both of the bf16-rounded pre-activation);
    VL_EPI_DGELU launched with the same act multiplies by that tensor.  Against fp32 torch, against the separate
    pre-activation + erf-in-the-backward pair, on every kernel family (persistent, ping-pong, plain tiles, tail rows)."""
    ops = _ops()
    M, N, K = (256 * 9 + (64 if cfg in (-1, 0, 1) else 0)), 1024, 512
    a = rnd(M, K, seed=31).bfloat16().cuda(); w = rnd(N, K, seed=32, scale=2 * K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=33).cuda()
    pre = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    y0 = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=cfg, out2=pre)
    d = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    y = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=cfg, out2=d)
    pf = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(pf).sum().backward()
    # bf16 of an fp32-accurate gelu' (|err| <= 4e-7 before rounding): at most one bf16 ulp from torch's
    assert bool(torch.isfinite(d).all())
    assert bool(((d.float() - pf.grad).abs() <= pf.grad.abs() * 2.0 ** -8 + 1e-6).all())
    if cfg in (8, 10, 5):                      # kernels whose GELU acts on the bf16-rounded pre-activation in both modes
        assert torch.equal(y, y0)
    else:                                      # plain-tile paths apply act=GELU to the fp32 value
        assert relerr(y, y0) < 4e-3
    # backward: (dy W) * gelu'
    dy = rnd(M, K, seed=34).bfloat16().cuda(); wt = rnd(N, K, seed=35, scale=K ** -0.5).bfloat16().cuda()
    acc = dy.float() @ wt.float().t()
    got = ops.gemm(dy, wt, None, res=d, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=cfg, out=torch.empty_like(d))
    assert relerr(got, acc * pf.grad) < 5e-3
    old = ops.gemm(dy, wt, None, res=pre, epi=ops.EPI_DGELU, cfg=cfg, out=torch.empty_like(d))
    assert relerr(got, old) < 4e-3            # one more bf16 rounding (of gelu') than the erf-in-the-backward form
    with pytest.raises(ValueError):
        ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=cfg)


def test_p4_is_what_auto_dispatch_uses_at_bench_geometry():
    """cfg=-1 at M = 257*256: whole rounds on p4 + tail kernel; every row must be written (NaN-poisoned outputs)."""
    ops = _ops()
    M, N, K = 257 * 256, 1024, 1024
    a = rnd(M, K, seed=21).bfloat16().cuda(); w = rnd(N, K, seed=22, scale=K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=23).cuda()
    res = rnd(M, N, seed=24).bfloat16().cuda()
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, bias, out=out, res=res, epi=ops.EPI_RES_BF16, cfg=-1)
    ref = res.float() + a.float() @ w.float().t() + bias
    assert bool(torch.isfinite(out).all()) and relerr(out, ref) < 4e-3
    assert relerr(out[-600:], ref[-600:]) < 4e-3 and relerr(out[:256], ref[:256]) < 4e-3


@pytest.mark.parametrize("M", [256 * 40, 256 * 6 + 77])
def test_auto_dispatch_width_1664(M):
    """ViT-bigG-14 geometry (width 1664 = 13 x 128, model_configs/ViT-bigG-14.json): N % 256 == 128 reaches the persistent
    256x128-tile (ping-pong) kernel through cfg=-1; the ragged rows go to the small-tile kernels."""
    ops = _ops()
    N = K = 1664
    a = rnd(M, K, seed=31).bfloat16().cuda(); w = rnd(N, K, seed=32, scale=K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=33).cuda()
    res = rnd(M, N, seed=34).bfloat16().cuda()
    acc = a.float() @ w.float().t() + bias
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, bias, out=out, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=-1)
    assert bool(torch.isfinite(out).all()) and relerr(out, torch.nn.functional.gelu(acc)) < 4e-3
    out2 = ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_BF16, cfg=-1)
    assert relerr(out2, res.float() + acc) < 4e-3
    if M % 256 == 0:
        assert torch.equal(out2, ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_BF16, cfg=10))


@pytest.mark.parametrize("M,N,K", [(256 * 24, 2048, 1024), (256 * 5, 512, 512), (257 * 64, 1024, 512)])
def test_geglu_and_dgeglu_epilogues_on_the_persistent_kernel(M, N, K):
    """Perceiver feed-forward epilogues (perceiver.py:85-102) on the 256x256 persistent kernel (round 3; round 2 ran them on
    the round-1 kernel at 600 TF/s): GEGLU = a * gelu(gate) over interleaved (a, gate) columns with the bf16 pre-activation
    saved, DGEGLU = its backward from the saved pre-activation.  Against fp32 torch and against the round-1 kernel."""
    ops = _ops()
    a = rnd(M, K, seed=51).bfloat16().cuda(); w = rnd(N, K, seed=52, scale=K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=53).cuda()
    acc = a.float() @ w.float().t() + bias
    cfg = 8 if M % 256 == 0 else -1
    hs = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    out = ops.gemm(a, w, bias, epi=ops.EPI_GEGLU, cfg=cfg, out2=hs)
    assert out.shape == (M, N // 2) and bool(torch.isfinite(out).all())
    assert relerr(hs, acc) < 4e-3
    hf = hs.float()                                                           # the reference's autocast multiplies the bf16 halves
    assert relerr(out, hf[:, 0::2] * torch.nn.functional.gelu(hf[:, 1::2])) < 4e-3
    assert relerr(out, acc[:, 0::2] * torch.nn.functional.gelu(acc[:, 1::2])) < 8e-3
    out_nosave = ops.gemm(a, w, bias, epi=ops.EPI_GEGLU, cfg=cfg)
    assert torch.equal(out, out_nosave)
    old = ops.gemm(a, w, bias, epi=ops.EPI_GEGLU, cfg=5)
    assert relerr(out, old) < 8e-3
    # backward: dy [M, N/2 ... here a fresh GEMM of width N] against h [M, 2N]
    h = rnd(M, 2 * N, seed=54, scale=1.2).bfloat16().cuda()
    dh = torch.full((M, 2 * N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, w, None, out=dh, res=h, epi=ops.EPI_DGEGLU, cfg=cfg)
    dy = (a.float() @ w.float().t())
    hf = h.float().requires_grad_(True)
    (hf[:, 0::2] * torch.nn.functional.gelu(hf[:, 1::2]) * dy).sum().backward()
    assert bool(torch.isfinite(dh).all()) and relerr(dh, hf.grad) < 6e-3
    old = torch.empty_like(dh)
    ops.gemm(a, w, None, out=old, res=h, epi=ops.EPI_DGEGLU, cfg=5)
    assert relerr(dh, old) < 6e-3


def test_many_tiles_no_sporadic_epilogue_faults(pcfg):
    """Round-3 finding (profiles/r03_pk_fma_fault.log): a compiler-chosen v_pk_fma_f32 form in the epilogue dropped the
    accumulator in the last 16 lanes of a wave a few times per launch - only with many tiles per workgroup, at different
    places in every run, invisible to a single small case.  16 tiles per workgroup, every element checked, four launches of
    each epilogue family; results must also be bit-identical from launch to launch."""
    ops = _ops()
    M, N, K = 65536, 4096, 1024
    a = rnd(M, K, seed=61).bfloat16().cuda(); w = rnd(N, K, seed=62, scale=K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=63).cuda()
    acc = a.float() @ w.float().t() + bias
    res = rnd(M, N, seed=64).bfloat16().cuda()

    def bad(out, ref):
        return int(((out.float() - ref).abs() > ref.abs() * 2.0 ** -6 + 2e-2).sum())
    first = {}
    for rep in range(4):
        outs = {"plain": ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=pcfg),
                "gelu": ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=pcfg),
                "res": ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_BF16, cfg=pcfg)}
        refs = {"plain": acc, "gelu": torch.nn.functional.gelu(acc), "res": res.float() + acc}
        for k, o in outs.items():
            assert bad(o, refs[k]) == 0, (k, rep, bad(o, refs[k]))
            if rep == 0:
                first[k] = o.clone()
            else:
                assert torch.equal(o, first[k]), (k, rep)


@pytest.mark.parametrize("M,N,K", [(256 * 40, 1024, 1024), (256 * 5, 768, 3072), (256 * 257, 256, 512), (256 * 77, 768, 768)])
def test_fp32_residual_epilogue_on_the_persistent_kernel(M, N, K):
    """EPI_RES_F32 (out f32 = res + acc * alpha + bias, in place allowed) on the persistent 256x256 kernel (round 4: the
    Perceiver's residual projections, the two-term text tower and f32-stream towers ran it on the round-1 kernel): against
    fp32 torch on the same bf16 operands and BIT-identical to the round-1 kernel (same products, same fp32 rounding points),
    out of place, in place, without bias, with alpha, into a strided residual view."""
    ops = _ops()
    a = rnd(M, K, seed=1).bfloat16().cuda(); w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=3).cuda()
    res = rnd(M, N, seed=4).cuda()
    acc = a.float() @ w.float().t()
    out8 = ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_F32, cfg=8)
    assert relerr(out8, acc + bias + res) < 1e-5
    out5 = ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_F32, cfg=5)
    assert float((out8 - out5).abs().max()) <= 2e-6 * float(out5.abs().max())
    x = res.clone()
    ops.gemm(a, w, None, out=x, res=x, epi=ops.EPI_RES_F32, cfg=8, alpha=0.25)        # in place, no bias, alpha
    assert relerr(x, 0.25 * acc + res) < 1e-5
    big = torch.zeros(M, N + 64, device="cuda"); view = big[:, 32:32 + N]
    view.copy_(res)
    ops.gemm(a, w, bias, out=view, res=view, epi=ops.EPI_RES_F32, cfg=8)
    assert relerr(view, acc + bias + res) < 1e-5 and float(big[:, :32].abs().max()) == 0.0 and float(big[:, 32 + N:].abs().max()) == 0.0
    # what the auto dispatch does with it (whole rounds persistent + leftover rows)
    auto = ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_F32)
    assert relerr(auto, acc + bias + res) < 1e-5


def test_many_tiles_every_epilogue_variant():
    """The first barrier of a tile behind an epilogue waits `vmcnt(NST)`, NST = the stores a wave issues per tile (derived from
    the epilogue's constants in vl_gemm_park.hip): an instantiation that issued fewer stores than its NST would let the barrier
    pass before the tile's second k-step has landed in LDS - silently wrong products, only with several tiles per workgroup.
    Every epilogue variant of the 8-wave kernel that the plain / GELU / residual gate above does not launch, 16 tiles per
    workgroup (8 for the half-width outputs), element-wise against fp32 torch and bit-identical over three launches."""
    ops = _ops()
    M, N, K = 65536, 4096, 1024
    a = rnd(M, K, seed=71).bfloat16().cuda(); w = rnd(N, K, seed=72, scale=K ** -0.5).bfloat16().cuda()
    bias = rnd(N, seed=73).cuda()
    acc = a.float() @ w.float().t()
    pre = (acc + bias).bfloat16().float()                    # the two-output GELU variants act on the bf16-rounded pre-activation
    g = rnd(M, N, seed=74, scale=0.5).bfloat16().cuda()
    resf = rnd(M, N, seed=75).cuda()
    h2 = rnd(M, 2 * N, seed=76, scale=1.2).bfloat16().cuda()

    def close(out, ref, tol=2.0 ** -6):
        return int(((out.float() - ref).abs() > ref.abs() * tol + 2e-2).sum())

    def gelu_grad(x):
        x = x.clone().requires_grad_(True)
        torch.nn.functional.gelu(x).sum().backward()
        return x.grad

    def launch():
        o = {}
        u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        o["gelu+save"] = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=8, out2=u); o["gelu+save:pre"] = u
        d = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        o["gelu+dsave"] = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=8, out2=d); o["gelu+dsave:d"] = d
        o["relu"] = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_RELU, cfg=8)
        o["dgelu"] = ops.gemm(a, w, None, out=torch.empty_like(g), res=g, epi=ops.EPI_DGELU, cfg=8)
        o["dgelu_saved"] = ops.gemm(a, w, None, out=torch.empty_like(g), res=g, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=8)
        o["res_f32"] = ops.gemm(a, w, bias, res=resf, epi=ops.EPI_RES_F32, cfg=8)
        hs = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        o["geglu"] = ops.gemm(a, w, bias, epi=ops.EPI_GEGLU, cfg=8, out2=hs); o["geglu:pre"] = hs
        o["geglu_nosave"] = ops.gemm(a, w, bias, epi=ops.EPI_GEGLU, cfg=8)
        dh = torch.empty(M, 2 * N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, w, None, out=dh, res=h2, epi=ops.EPI_DGEGLU, cfg=8); o["dgeglu"] = dh
        return o
    first = launch()
    hf = first["geglu:pre"].float()
    refs = {"gelu+save": torch.nn.functional.gelu(pre), "gelu+save:pre": acc + bias,
            "gelu+dsave": torch.nn.functional.gelu(pre), "gelu+dsave:d": gelu_grad(pre), "relu": torch.relu(acc + bias),
            "dgelu": acc * gelu_grad(g.float()), "dgelu_saved": acc * g.float(), "res_f32": acc + bias + resf,
            "geglu:pre": acc + bias, "geglu": hf[:, 0::2] * torch.nn.functional.gelu(hf[:, 1::2])}
    refs["geglu_nosave"] = refs["geglu"]
    hq = h2.float().requires_grad_(True)
    (hq[:, 0::2] * torch.nn.functional.gelu(hq[:, 1::2]) * acc).sum().backward()
    refs["dgeglu"] = hq.grad
    for k, ref in refs.items():
        assert close(first[k], ref) == 0, (k, close(first[k], ref))
    del refs, hq, hf
    for rep in range(2):
        again = launch()
        for k, v in again.items():
            assert torch.equal(v, first[k]), (k, rep)


def test_many_tiles_lnfold_variants():
    """The same gate for the LayerNorm-folding instantiations (consumer ACT 10 / 11 / 14, producer ACT 20 with its extra
    partial-sum stores): bit-identical over three launches and against the un-folded kernel on the same operands."""
    ops = _ops()
    M, N, K = 65536, 4096, 1024
    x = rnd(M, K, seed=81).bfloat16().cuda(); w = rnd(N, K, seed=82, scale=K ** -0.5).cuda()
    b = rnd(N, seed=83).cuda()
    gam, bet = (1 + 0.2 * rnd(K, seed=84)).cuda(), (0.1 * rnd(K, seed=85)).cuda()
    fold = ops.fold_ln_linear(w, b, gam, bet)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.ln_row_stats(None, x, 0, mean, rstd)
    hws = torch.empty(1, K, device="cuda", dtype=torch.bfloat16)
    ln = torch.nn.functional.layer_norm(x.float(), (K,), gam, bet)
    ref = ln @ w.t() + b

    def launch():
        o = {}
        for name, act in (("ln", ops.ACT_NONE), ("ln_gelu", ops.ACT_GELU), ("ln_dsave", ops.ACT_GELU_DSAVE)):
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            d = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if act == ops.ACT_GELU_DSAVE else None
            ops.gemm_lnfold(x, fold, mean, rstd, out, w.bfloat16(), b, gam, bet, hws, act=act, out2=d)
            o[name] = out
            if d is not None:
                o[name + ":d"] = d
        return o
    first = launch()
    assert relerr(first["ln"], ref) < 6e-3
    assert relerr(first["ln_gelu"], torch.nn.functional.gelu(ref)) < 8e-3
    assert relerr(first["ln_dsave"], torch.nn.functional.gelu(ref)) < 8e-3
    assert bool(torch.isfinite(first["ln_dsave:d"]).all())
    for rep in range(2):
        for k, v in launch().items():
            assert torch.equal(v, first[k]), (k, rep)
    # producer: residual + partial row sums (two stores per chunk)
    a2 = rnd(M, K, seed=86).bfloat16().cuda(); w2 = rnd(1024, K, seed=87, scale=K ** -0.5).bfloat16().cuda()
    b2 = rnd(1024, seed=88).cuda()
    res = rnd(M, 1024, seed=89).bfloat16().cuda()
    plain = ops.gemm(a2, w2, b2, res=res, epi=ops.EPI_RES_BF16, cfg=8)
    parts = []
    for rep in range(3):
        part = torch.full((M * 16 * 2,), float("nan"), device="cuda")
        out = torch.empty_like(res)
        mm = ops.gemm_res_rowstats(a2, w2, b2, out, res, part)
        assert mm == M and torch.equal(out, plain)
        parts.append(part)
    assert torch.equal(parts[0], parts[1]) and torch.equal(parts[0], parts[2])
    s = parts[0].view(M, 16, 2)
    of = plain.float()
    assert float((s[:, :, 0].sum(1) - of.sum(1)).abs().max()) < 2e-2 * float(of.abs().sum(1).max())
    assert relerr(s[:, :, 1].sum(1), (of * of).sum(1)) < 1e-4
