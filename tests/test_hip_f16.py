"""GPU parity of the IEEE-half entries that carry the frozen text tower (round 5; include/vitlens_hip.h: vl_gemm_f16,
vl_attn_fwd_f16, vl_layernorm_fwd with VL_F16 output) against fp32 torch on the same fp16 operands.  Replaces
F.linear / F.multi_head_attention_forward / LayerNorm of TriCLIP.encode_text (open_clip/model.py:528-540,
transformer.py:226-272) for that tower only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from vitlens_hip import ops
    return ops


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("M,N,K", [(256 * 77, 2304, 768), (256 * 3, 768, 3072), (256 * 308, 3072, 768), (256, 512, 512)])
def test_gemm_f16_epilogues(M, N, K):
    """Text-tower shapes (77 x 256 captions and the whole 1 024-caption batch: uneven last round of the persistent kernel),
    plain / GELU fp16 output and the fp32 residual epilogue; fp16 rounding is 2^-11: bounds 8x tighter than the bf16 tests."""
    ops = _ops()
    a = rnd(M, K, seed=1).half(); w = rnd(N, K, seed=2, scale=K ** -0.5).half()
    bias = rnd(N, seed=3)
    acc = a.float() @ w.float().t() + bias
    out = ops.gemm_f16(a, w, bias, out=torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16))
    assert out.dtype == torch.float16 and bool(torch.isfinite(out).all())
    assert relerr(out, acc) < 6e-4, relerr(out, acc)
    assert bool(((out.float() - acc).abs() <= acc.abs() * 2.0 ** -10 + 1e-3).all())
    out = ops.gemm_f16(a, w, bias, act=ops.ACT_GELU)
    assert relerr(out, torch.nn.functional.gelu(acc)) < 6e-4
    out = ops.gemm_f16(a, w, None, alpha=0.5)
    assert relerr(out, 0.5 * (acc - bias)) < 6e-4
    res = rnd(M, N, seed=4)
    o32 = ops.gemm_f16(a, w, bias, res=res, epi=ops.EPI_RES_F32)
    assert o32.dtype == torch.float32 and relerr(o32, acc + res) < 2e-6
    x = res.clone()
    ops.gemm_f16(a, w, bias, out=x, res=x, epi=ops.EPI_RES_F32)
    assert torch.equal(x, o32)
    # bit-identical from launch to launch
    assert torch.equal(ops.gemm_f16(a, w, bias), ops.gemm_f16(a, w, bias))


def test_gemm_f16_identity_saturation_and_refusals():
    ops = _ops()
    K, N = 512, 768
    eye = torch.eye(K)
    a = torch.cat([eye * (2.0 ** r) for r in range(4)], 0).half().cuda()                      # [2048, 512]
    w = ((torch.arange(N * K).reshape(N, K) * 7 % 251) - 125).float().half().cuda()
    out = ops.gemm_f16(a, w)
    ref = torch.cat([w.float().t() * (2.0 ** r) for r in range(4)], 0)
    assert torch.equal(out.float(), ref)                                                       # exact in fp16: any row / column mix-up shows
    # fp16 has five exponent bits: the 16-bit store saturates instead of producing infinities
    big = torch.full((256, 512), 60000.0, device="cuda").half()
    one = torch.zeros(256, 512, device="cuda").half(); one[:, :4] = 1.0
    out = ops.gemm_f16(big, one)                                                               # 4 x 60000 = 240000 > 65504
    assert bool(torch.isfinite(out).all()) and float(out.max()) == 65504.0
    out = ops.gemm_f16(-big, one)
    assert float(out.min()) == -65504.0
    with pytest.raises(RuntimeError):
        ops.gemm_f16(a[:300], w)                     # ragged rows: the caller pads
    with pytest.raises(RuntimeError):
        ops.gemm_f16(a[:, :256].contiguous(), w[:, :256].contiguous())      # K < 512
    with pytest.raises(TypeError):
        ops.gemm_f16(a.bfloat16(), w)


@pytest.mark.parametrize("B,H,L,causal", [(8, 12, 77, True), (5, 8, 77, True), (3, 12, 77, False), (2, 4, 288, True), (4, 12, 20, True)])
def test_attn_fwd_f16(B, H, L, causal):
    ops = _ops()
    dh = 64
    D = H * dh
    qkv = rnd(B * L, 3 * D, seed=11).half()
    q, k, v = (ops.heads_view(qkv, B, L, H, dh, i * D) for i in range(3))
    out = torch.full((B * L, D), float("nan"), device="cuda", dtype=torch.float16)
    lse = torch.empty(B, H, L, device="cuda")
    ops.attn_fwd(q, k, v, out, lse=lse, causal=causal, qscale=dh ** -0.5 * ops.LOG2E)
    s = (q.float() @ k.float().transpose(-1, -2)) * dh ** -0.5
    if causal:
        s = s + torch.full((L, L), float("-inf"), device="cuda").triu_(1)
    ref = (torch.softmax(s, -1) @ v.float()).permute(0, 2, 1, 3).reshape(B * L, D)
    assert bool(torch.isfinite(out).all())
    assert relerr(out, ref) < 2e-3, relerr(out, ref)          # (bf16 kernel: 1e-2)
    assert relerr(lse, torch.logsumexp(s, -1)) < 1e-3
    with pytest.raises(RuntimeError):
        q32 = ops.heads_view(rnd(B * L, 3 * 32 * H, seed=1).half(), B, L, H, 32, 0)
        ops.attn_fwd(q32, q32, q32, torch.empty(B * L, 32 * H, device="cuda", dtype=torch.float16))


def test_layernorm_f16_output():
    ops = _ops()
    rows, D = 616, 768
    x = rnd(rows, D, seed=21, scale=3.0) + 0.5
    w, b = 1 + 0.2 * rnd(D, seed=22), 0.1 * rnd(D, seed=23)
    out = torch.zeros(768, D, device="cuda", dtype=torch.float16)
    ops.layernorm(x, w, b, out, rows, D)
    ref = torch.nn.functional.layer_norm(x, (D,), w, b)
    assert relerr(out[:rows], ref) < 4e-4
    assert float(out[rows:].abs().max()) == 0.0             # rows beyond `rows` are not touched
    idx = torch.tensor([3, 0, 5], device="cuda")
    pooled = torch.zeros(4, D, device="cuda", dtype=torch.float16)
    ops.layernorm(x, w, b, pooled, 3, D, x_row_stride=D, row_index=idx, row_mul=7)
    assert relerr(pooled[:3], ref[[3, 7, 19]]) < 4e-4
