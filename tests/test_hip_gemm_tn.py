"""GPU: the weight-gradient GEMM on token-major operands (`vl_gemm_tn_splitk_accum_f32`, csrc/vl_gemm_tn.hip): g += dy^T x with
dy [tokens, N_out], x [tokens, K_in] as the backward holds them - fragments through the LDS transpose read instead of
transposed copies.  Bit-exact on small-integer operands (every product and partial sum is exact in fp32: any row / column /
token mix-up of the DMA swizzle or of the transpose read shows), fp32-torch on random data, strided operands, accumulation
into a strided gradient view, the fall-back conditions, and the transposing path it replaces as a second reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from vitlens_hip import ops
    return ops


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("R,M,N", [(4096, 1024, 4096), (4096, 4096, 1024), (8192, 3072, 1024), (257 * 256, 1024, 4096),
                                   (64 * 70, 4096, 1024),        # 70 steps in 4 slices of 18, 18, 18, 16
                                   (257 * 256, 1024, 1024)])     # out_proj at the C3 micro-batch: 16 tiles x 16 slices of 65 ... 53 steps
def test_small_integers_are_bit_exact(R, M, N):
    ops = _ops()
    g = torch.Generator().manual_seed(R + M)
    dy = torch.randint(-2, 3, (R, M), generator=g).float().bfloat16().cuda()
    x = torch.randint(-2, 3, (R, N), generator=g).float().bfloat16().cuda()
    out = torch.zeros(M, N, device="cuda")
    assert ops.gemm_dw_tn(dy, x, out)
    ref = dy.float().t() @ x.float()                    # |sum| <= 4 R < 2^24: exact in fp32 whatever the order
    assert torch.equal(out, ref)
    # position-coded operands: dy[t, m] = [m == t mod M], x[t, n] = (t + 3 n) mod 7 - a wrong token or column cannot cancel
    t = torch.arange(R, device="cuda")
    dy2 = torch.zeros(R, M, device="cuda"); dy2[t, t % M] = 1.0
    x2 = ((t[:, None] + 3 * torch.arange(N, device="cuda")[None, :]) % 7).float()
    out2 = torch.zeros(M, N, device="cuda")
    assert ops.gemm_dw_tn(dy2.bfloat16(), x2.bfloat16(), out2)
    assert torch.equal(out2, dy2.t() @ x2)


def test_random_strided_accumulate_and_alpha():
    ops = _ops()
    R, M, N = 64 * 64, 1024, 1024 * 3
    g = torch.Generator().manual_seed(5)
    wide_dy = torch.randn(R, M + 512, generator=g).bfloat16().cuda()      # dy = a column window of a wider tensor (dqkv-style views)
    wide_x = torch.randn(R, N + 256, generator=g).bfloat16().cuda()
    dy, x = wide_dy[:, 256:256 + M], wide_x[:, :N]
    assert dy.data_ptr() % 16 == 0
    gbuf = torch.randn(M, N + 64, generator=g).cuda()                     # gradient = a strided view that already holds a value
    gv = gbuf[:, :N]
    before = gv.clone()
    assert ops.gemm_dw_tn(dy, x, gv, alpha=0.5)
    ref = before.double() + 0.5 * (dy.double().t() @ x.double())
    assert relerr(gv, ref) < 2e-6
    assert torch.equal(gbuf[:, N:], gbuf[:, N:])                          # (untouched tail columns stay finite)
    # the transposing path computes the same sums from the same bf16 operands: fp32 accumulation-order noise only
    old = before.clone()
    rp = R
    ops.gemm_dw(ops.transpose_to_bf16(dy, ldo=rp), ops.transpose_to_bf16(x, ldo=rp), old, alpha=0.5)
    assert relerr(gv, old) < 2e-6


def test_shapes_that_do_not_fit_are_refused_without_a_launch():
    ops = _ops()
    out = torch.zeros(1024, 1024, device="cuda")
    dy = torch.ones(4096, 1024, device="cuda").bfloat16(); x = torch.ones(4096, 1024, device="cuda").bfloat16()
    assert not ops.gemm_dw_tn(dy, x, out)                                  # 16 tiles x 4 slices < 192 work items
    assert not ops.gemm_dw_tn(dy[:4000], x[:4000], out)                    # tokens not a multiple of 64
    assert not ops.gemm_dw_tn(dy.float(), x, out)                          # fp32 operand
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("R,C,stride", [(65792, 4096, 4096), (1000, 1024, 1536), (777, 72, 72), (300, 100, 100)])
def test_bias_gradient_column_sum(R, C, stride):
    """`colsum` (what produces the bias gradient next to the token-major dW): the 16-byte bf16 kernel (C % 8 == 0, aligned rows)
    and the one-column-per-thread kernel for everything else, accumulating into `out` with a scale."""
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    a = torch.randn(R, stride, generator=g).bfloat16().cuda()[:, :C]
    out = torch.full((C,), 3.0, device="cuda")
    ops.colsum(a, out, 0.25)
    ref = 3.0 + 0.25 * a.double().sum(0)
    assert relerr(out, ref) < 1e-5
    # the same columns as fp32 (the Perceiver's residual-gradient stream): the 16-byte kernel for C % 4 == 0 on aligned rows,
    # the one-column-per-thread kernel for the view shifted by one column
    af = torch.randn(R, stride + 4, generator=g).cuda()
    for view in (af[:, :C], af[:, 1:1 + C]):
        outf = torch.full((C,), -2.0, device="cuda")
        ops.colsum(view, outf, 0.5)
        assert relerr(outf, -2.0 + 0.5 * view.double().sum(0)) < 1e-5
        again = torch.full((C,), -2.0, device="cuda")
        ops.colsum(view, again, 0.5)
        assert torch.equal(outf, again)                       # deterministic two-stage reduction


@pytest.mark.parametrize("R,M,N", [(256 * 64, 64, 1024),      # Perceiver to_q / to_k gradient: 64 output columns
                                   (512 * 64, 1024, 128),     # to_out: a 128-column operand on the x side
                                   (196 * 64, 1024, 384),     # a 384-wide context (1.5 tiles)
                                   (64 * 64, 768, 608)])      # patch-convolution columns: 588 = 3 * 14 * 14 padded to 608 by im2col
def test_narrow_operands_through_the_padded_path(R, M, N):
    """`gemm_dw_tn_any`: operands whose column counts are not whole 256-column tiles go through a zero-padded copy of the
    narrow one; the result accumulates into `g` with `alpha` exactly like the whole-tile path.  Small integers: bit-exact."""
    ops = _ops()
    g = torch.Generator().manual_seed(R + M + N)
    dy = torch.randint(-2, 3, (R, M), generator=g).float().bfloat16().cuda()
    x = torch.randint(-2, 3, (R, N), generator=g).float().bfloat16().cuda()
    out = torch.full((M, N), 3.0, device="cuda")
    assert ops.gemm_dw_tn_any(dy, x, out, alpha=0.5)
    ref = 3.0 + 0.5 * (dy.float().t() @ x.float())
    assert torch.equal(out, ref)
    # random data, a second call on the same shapes (the padded staging buffers are reused: stale pad columns would show)
    dy2 = torch.randn(R, M, generator=g).bfloat16().cuda(); x2 = torch.randn(R, N, generator=g).bfloat16().cuda()
    out2 = torch.zeros(M, N, device="cuda")
    assert ops.gemm_dw_tn_any(dy2, x2, out2)
    assert relerr(out2, dy2.double().t() @ x2.double()) < 2e-6
    assert not ops.gemm_dw_tn_any(dy2[:100], x2[:100], out2)               # rows not a multiple of 64: refused, no launch
