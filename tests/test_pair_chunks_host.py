"""CPU: the row-blocked InfoNCE bookkeeping of `vitlens_hip.step.pair_forward / pair_backward` (block offsets, ragged last
block, merged column statistics, per-block re-weighting, accumulation into dy and d/dscale) on a torch stand-in for the
HIP ops: every block size must give the unblocked result and torch autograd's, for square, local-loss-shaped (label
offset; rows only, as `pair_loss_and_grads` calls it: both directions of a local loss are row losses of two
rectangular matrices) and one-directional (w_col = 0) cases.  The kernels' side of the same paths: tests/test_hip_loss.py."""
import sys
import types

import pytest
import torch

import test_step_gloo as G


def _ops():
    o = G._fake_ops()
    o.EPI_F32, o.EPI_RES_F32 = 1, 2

    def gemm(a, w, bias=None, out=None, res=None, epi=1, alpha=1.0, **kw):
        y = alpha * a @ w.t()
        if epi == o.EPI_RES_F32:
            y = y + res
        if out is not None:
            out.copy_(y)
            return out
        return y
    o.gemm = gemm
    return o


@pytest.mark.parametrize("R,C,off,w_row,w_col", [(37, 37, 0, 0.5, 0.5), (24, 96, 48, 1.0, 0.0), (40, 40, 0, 1.0, 0.0), (9, 30, 21, 0.5, 0.0), (64, 64, 0, 0.25, 0.75)])
def test_row_blocks_equal_whole_matrix_and_autograd(R, C, off, w_row, w_col, monkeypatch):
    from vitlens_hip import step as ST
    monkeypatch.setattr(ST, "ops", _ops())
    g = torch.Generator().manual_seed(R * 100 + C)
    x = torch.nn.functional.normalize(torch.randn(R, 16, generator=g), dim=-1)
    y = torch.nn.functional.normalize(torch.randn(C, 16, generator=g), dim=-1)
    scale = 14.3
    xr, yr, sr = x.clone().requires_grad_(True), y.clone().requires_grad_(True), torch.tensor(scale, requires_grad=True)
    logits = sr * xr @ yr.t()
    lab = torch.arange(R) + off
    ref = w_row * torch.nn.functional.cross_entropy(logits, lab)
    if w_col:
        ref = ref + w_col * (torch.logsumexp(logits, 0)[lab] - logits[torch.arange(R), lab]).mean()
    ref.backward()
    outs = {}
    for rb in (0, 1, 4, 7, R - 1, R, 1000):
        loss, ctx = ST.pair_forward(x, y, scale, label_off=off, w_row=w_row, w_col=w_col, chunk_rows=rb)
        dx, dy, ds = ST.pair_backward(ctx)
        outs[rb] = (loss, dx, dy, ds)
        assert abs(float(loss) - float(ref.detach())) < 1e-5, (rb, float(loss), float(ref.detach()))
        assert float((dx - xr.grad).abs().max()) < 1e-5 and float((dy - yr.grad).abs().max()) < 1e-5, rb
        assert abs(float(ds) - float(sr.grad)) < 1e-4 * max(1.0, abs(float(sr.grad))), (rb, float(ds), float(sr.grad))
    assert ST._chunk_rows(R, C, R) == 0 and ST._chunk_rows(R, C, 0) == 0 and ST._chunk_rows(R, C, 4) == 4
    assert ST._chunk_rows(8192, 8192, None) == ST.LOGITS_CHUNK_ROWS and ST._chunk_rows(1024, 1024, None) == 0
