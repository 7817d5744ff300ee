"""CPU: host-side arithmetic of the product that needs no GPU (the library only has to load)."""


def test_layernorm_fold_operands_are_the_algebra_of_ln_then_linear():
    """ops.fold_ln_linear (host side of vl_gemm_lnfold_bf16): with Wg = bf16(W * gamma), c = row sums of the ROUNDED Wg and
    d = b + W beta,   rstd * (x Wg^T - mean * c) + d   is exactly   ((x - mean) * rstd) Wg^T + d   (the cancellation of the mean
    term needs c from the rounded weights), and equals LN(x) W^T + b up to the bf16 rounding of Wg."""
    import torch
    from vitlens_hip import ops
    g = torch.Generator().manual_seed(0)
    M, K, N = 64, 256, 96
    x = torch.randn(M, K, generator=g, dtype=torch.float64) * 3 + 1.5
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    gamma = 1 + 0.3 * torch.randn(K, generator=g)
    beta = 0.2 * torch.randn(K, generator=g)
    wg, d, c = ops.fold_ln_linear(w, b, gamma, beta)
    assert wg.dtype == torch.bfloat16 and d.dtype == torch.float32 and c.dtype == torch.float32
    mu = x.mean(1, keepdim=True); rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    folded = rstd * (x @ wg.double().t() - mu * c.double()) + d.double()
    direct = ((x - mu) * rstd) @ wg.double().t() + d.double()
    assert float((folded - direct).abs().max()) < 1e-5                       # c is an fp32 sum of 256 bf16 values
    ref = torch.nn.functional.layer_norm(x, (K,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + b.double()
    assert float((folded - ref).norm() / ref.norm()) < 4e-3                  # bf16 rounding of W * gamma only
    assert float((d.double() - (b.double() + w.double() @ beta.double())).abs().max()) < 1e-5
