"""CPU study for a planned kernel change: LayerNorm folded into the consuming GEMM.
reference path:  h = bf16(LN(x)); y = bf16(h @ W^T + b)         (the autocast rounding points of the reference)
folded path:     Wg = bf16(W * gamma); y = bf16(rstd * (x @ Wg^T - mean * c) + d),  c = rowsum(Wg), d = W @ beta + b
both against fp64, for plain rows, rows with a large common offset, and rows with a few huge channels."""
import torch

torch.manual_seed(0)
M, K, N = 2048, 1024, 1024
W = torch.randn(N, K) * K ** -0.5
gamma = 1 + 0.2 * torch.randn(K); beta = 0.1 * torch.randn(K); b = 0.1 * torch.randn(N)
bf = lambda t: t.bfloat16().float()


def run(x, tag):
    xb = bf(x)                                            # the residual stream is bf16 in the bench configuration
    xd = xb.double()
    mean = xd.mean(1, keepdim=True); var = xd.var(1, unbiased=False, keepdim=True); rstd = (var + 1e-5).rsqrt()
    truth = ((xd - mean) * rstd * gamma.double() + beta.double()) @ bf(W).double().t() + b.double()
    h = bf(((xb - mean.float()) * rstd.float()) * gamma + beta)
    ref = bf(h @ bf(W).t() + b)
    Wg = bf(W * gamma)
    c = Wg.sum(1); d = bf(W) @ beta + b
    acc = xb @ Wg.t()
    fold = bf(rstd.float() * (acc - mean.float() * c) + d)
    # the same with the weight kept exact in gamma (error of rounding W*gamma instead of W separated out)
    rel = lambda a: float((a.double() - truth).norm() / truth.norm())
    print(f"{tag:34s} reference path {rel(ref):.2e}   folded {rel(fold):.2e}   (|mean|/std of rows: {float((mean.abs() * rstd).mean()):.2f})")


x = torch.randn(M, K)
run(x, "plain rows")
run(x + 10.0, "rows offset by 10 sigma")
run(x + 100.0, "rows offset by 100 sigma")
xo = x.clone(); xo[:, :4] *= 100.0
run(xo, "4 channels x 100")
xo2 = x.clone(); xo2[:, 7] += 300.0
run(xo2, "one channel + 300 (massive act.)")
