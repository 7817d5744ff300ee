"""GPU parity of the one-wave-per-SIMD persistent GEMM (vl_gemm_pk4.hip, cfg = 14: four waves of 128x128, accumulators in
AGPRs, hand-ordered k-step, hand-counted vmcnt / lgkmcnt): every epilogue it has - plain, GELU, GELU + gelu', bf16 residual
in place, dGELU from the saved gelu' - against an fp32 torch product of the same bf16 operands and against the shipped 8-wave
kernel (cfg = 8), at the shapes of a ViT-L block (open_clip/transformer.py:226-234,254-272) with whole rounds (M = 65 536)
and an uneven last round (M = 65 792 = 257 row tiles), bit-identical from launch to launch, plus the many-tiles sporadic
fault gate of the 8-wave kernel.  The kernel was written in round 4 without a GPU in reach; this file is what decides
whether it may be dispatched at all."""
import pytest
import torch

pytestmark = pytest.mark.gpu

PK4 = 14


def _ops():
    from vitlens_hip import ops
    return ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _ref_rows(a, w, rows):
    """fp32 product of a row sample (the full 65 792 x 4096 fp32 reference would be 1 GB per tensor and seconds of fp32 GEMM)."""
    return a[rows].float() @ w.float().t()


def _rows(M):
    idx = torch.cat([torch.arange(0, 512), torch.arange(M // 2 - 128, M // 2 + 128), torch.arange(M - 768, M)])
    return idx.cuda()


def test_pk4_identity_is_bit_exact():
    """A = [I; 2I; ...]: C rows are W^T scaled by powers of two -> exact in bf16; catches any row / column / chunk mix-up of the
    DMA swizzle, the fragment addressing and the slab round trip of the epilogue."""
    ops = _ops()
    K, N, reps = 512, 768, 6
    eye = torch.eye(K)
    a = torch.cat([eye * (2.0 ** r) for r in range(reps)], 0).bfloat16().cuda()
    w = ((torch.arange(N * K).reshape(N, K) * 7 % 251) - 125).float().bfloat16().cuda()
    out = ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=PK4)
    ref = torch.cat([w.float().t() * (2.0 ** r) for r in range(reps)], 0)
    assert torch.equal(out.float(), ref)


@pytest.mark.parametrize("M", [65536, 65792, 256 * 3])
@pytest.mark.parametrize("N,K", [(1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)])
def test_pk4_all_epilogues_vs_fp32(M, N, K):
    ops = _ops()
    a = rnd(M, K, seed=1).bfloat16(); w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16()
    bias = rnd(N, seed=3)
    rows = _rows(M) if M > 2048 else torch.arange(M).cuda()
    acc = _ref_rows(a, w, rows)
    nan = lambda: torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)

    # plain (+ alpha, no bias)
    o = ops.gemm(a, w, bias, out=nan(), epi=ops.EPI_BF16, cfg=PK4)
    assert bool(torch.isfinite(o).all())
    assert relerr(o[rows], acc + bias) < 4e-3
    o8 = ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=8)
    assert torch.equal(o, o8)                       # same products in the same order per accumulator, same rounding points
    o = ops.gemm(a, w, None, out=nan(), epi=ops.EPI_BF16, cfg=PK4, alpha=0.5)
    assert relerr(o[rows], 0.5 * acc) < 4e-3
    # GELU
    o = ops.gemm(a, w, bias, out=nan(), epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=PK4)
    assert bool(torch.isfinite(o).all()) and relerr(o[rows], torch.nn.functional.gelu(acc + bias)) < 4e-3
    # GELU + gelu' saved for the dX GEMM
    d = nan()
    o = ops.gemm(a, w, bias, out=nan(), epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=PK4, out2=d)
    pre = (acc + bias).bfloat16().float().requires_grad_(True)          # the kernel applies both to the bf16-rounded pre-activation
    torch.nn.functional.gelu(pre).sum().backward()
    assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(d).all())
    assert relerr(o[rows], torch.nn.functional.gelu(pre.detach())) < 4e-3
    assert relerr(d[rows], pre.grad) < 6e-3
    d8 = nan()
    o8 = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=8, out2=d8)
    assert torch.equal(o, o8) and torch.equal(d, d8)
    # bf16 residual, out of place and in place
    res = rnd(M, N, seed=7).bfloat16()
    o = ops.gemm(a, w, bias, out=nan(), res=res, epi=ops.EPI_RES_BF16, cfg=PK4)
    assert bool(torch.isfinite(o).all()) and relerr(o[rows], res[rows].float() + acc + bias) < 4e-3
    x = res.clone()
    ops.gemm(a, w, bias, out=x, res=x, epi=ops.EPI_RES_BF16, cfg=PK4)
    assert torch.equal(x, o)
    assert torch.equal(o, ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_BF16, cfg=8))
    # dGELU from the saved gelu'
    g = rnd(M, N, seed=9, scale=0.5).bfloat16()
    o = ops.gemm(a, w, None, out=nan(), res=g, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=PK4)
    assert bool(torch.isfinite(o).all()) and relerr(o[rows], acc * g[rows].float()) < 6e-3
    assert torch.equal(o, ops.gemm(a, w, None, out=nan(), res=g, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=8))


def test_pk4_refuses_what_it_does_not_take():
    ops = _ops()
    a = rnd(300, 512).bfloat16(); w = rnd(256, 512).bfloat16()
    with pytest.raises(Exception):
        ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=PK4)                 # ragged M
    a = rnd(512, 256).bfloat16(); w = rnd(256, 256).bfloat16()
    with pytest.raises(Exception):
        ops.gemm(a, w, None, epi=ops.EPI_BF16, cfg=PK4)                 # K < 512
    a = rnd(512, 512).bfloat16(); w = rnd(256, 512).bfloat16()
    with pytest.raises(Exception):
        ops.gemm(a, w, None, epi=ops.EPI_BF16, act=ops.ACT_RELU, cfg=PK4)


def test_pk4_many_tiles_no_sporadic_epilogue_faults():
    """The gate of tests/test_hip_gemm_park.py::test_many_tiles_no_sporadic_epilogue_faults on this kernel: 16 tiles per
    workgroup, every element checked, four launches of every epilogue, bit-identical from launch to launch."""
    ops = _ops()
    M, N, K = 65536, 4096, 1024
    a = rnd(M, K, seed=61).bfloat16(); w = rnd(N, K, seed=62, scale=K ** -0.5).bfloat16()
    bias = rnd(N, seed=63)
    res = rnd(M, N, seed=64).bfloat16()
    g = rnd(M, N, seed=65, scale=0.5).bfloat16()
    # references from the shipped kernel (itself held to fp32 torch element by element in test_hip_gemm_park.py)
    refs = {"plain": ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=8),
            "gelu": ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=8),
            "res": ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_BF16, cfg=8),
            "dgelu": ops.gemm(a, w, None, out=torch.empty_like(res), res=g, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=8)}
    d8 = torch.empty_like(res)
    refs["dsave"] = ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=8, out2=d8)
    for rep in range(4):
        d4 = torch.full_like(res, float("nan"))
        outs = {"plain": ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=PK4),
                "gelu": ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU, cfg=PK4),
                "res": ops.gemm(a, w, bias, res=res, epi=ops.EPI_RES_BF16, cfg=PK4),
                "dgelu": ops.gemm(a, w, None, out=torch.empty_like(res), res=g, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=PK4),
                "dsave": ops.gemm(a, w, bias, epi=ops.EPI_BF16, act=ops.ACT_GELU_DSAVE, cfg=PK4, out2=d4)}
        for k, o in outs.items():
            nbad = int((o.view(torch.int16) != refs[k].view(torch.int16)).sum())
            assert nbad == 0, (k, rep, nbad)
        assert torch.equal(d4, d8), rep
