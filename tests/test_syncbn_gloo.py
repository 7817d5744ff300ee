"""CPU, 2 and 8 processes on gloo: the SyncBatchNorm host logic of the point-cloud tokenizer (`PointTokenizerTrainer._bn` /
`_bn_bwd`, --use-bn-sync) through the real `TorchComm`, with the HIP `ops` replaced by a torch-CPU stand-in of the four
split passes.  Asserted per rank: the collective sequence and payloads (ONE all-gather of 2C+1 floats per BatchNorm
forward, ONE all-reduce of 2C floats per backward), and the numbers - outputs, input gradients and running statistics
equal torch's F.batch_norm autograd on the CONCATENATION of the ranks' (unequal) batches, which is what
torch.nn.SyncBatchNorm computes; dgamma / dbeta stay local sums, as SyncBatchNorm leaves them for DDP."""
import os
import sys
import types

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = 8
# unequal per-rank row counts: the merge must weight by the gathered counts
ROWS_BY_WORLD = {2: (5, 11), 8: (5, 11, 3, 7, 2, 9, 4, 6)}


def _fake_ops():
    o = types.SimpleNamespace()

    def bn_stats_local(x):
        x = x.double()
        m = x.mean(0)
        cnt = torch.tensor([x.shape[0]], dtype=torch.int32).view(torch.float32)
        return torch.cat([m.float(), ((x - m) ** 2).sum(0).float(), cnt])

    def bn_stats_merge(gathered, rm=None, rv=None, momentum=0.1):
        Cc = (gathered.shape[1] - 1) // 2
        n, mu, m2 = 0.0, torch.zeros(Cc, dtype=torch.float64), torch.zeros(Cc, dtype=torch.float64)
        for g in gathered:
            nk = float(g[2 * Cc:].view(torch.int32)[0])
            d = g[:Cc].double() - mu
            mu = mu + d * (nk / (n + nk)); m2 = m2 + g[Cc:2 * Cc].double() + d * d * (n * nk / (n + nk)); n += nk
        if rm is not None:
            rm.mul_(1 - momentum).add_(momentum * mu.float()); rv.mul_(1 - momentum).add_(momentum * (m2 / (n - 1)).float())
        return mu.float(), (m2 / n).float(), torch.tensor([int(n)], dtype=torch.int32)

    def bn_apply(x, mean, var, gamma, beta, eps=1e-5, relu=False):
        y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
        return torch.relu(y) if relu else y

    def _dyp(dy, x, mean, var, gamma, beta, eps, relu):
        xh = (x - mean) * torch.rsqrt(var + eps)
        return (dy * ((xh * gamma + beta) > 0) if relu else dy), xh

    def bn_bwd_reduce(dy, x, mean, var, gamma, beta, dgamma, dbeta, eps=1e-5, relu=False):
        d, xh = _dyp(dy, x, mean, var, gamma, beta, eps, relu)
        dbeta += d.sum(0); dgamma += (d * xh).sum(0)
        return torch.cat([d.sum(0), (d * xh).sum(0)])

    def bn_bwd_apply(dy, x, mean, var, gamma, beta, sums, total, eps=1e-5, relu=False):
        d, xh = _dyp(dy, x, mean, var, gamma, beta, eps, relu)
        n = float(total)
        return gamma * torch.rsqrt(var + eps) * (d - sums[:x.shape[1]] / n - xh * sums[x.shape[1]:] / n)
    o.bn_stats_local, o.bn_stats_merge, o.bn_apply, o.bn_bwd_reduce, o.bn_bwd_apply = (
        bn_stats_local, bn_stats_merge, bn_apply, bn_bwd_reduce, bn_bwd_apply)
    return o


def _data(world):
    ROWS = ROWS_BY_WORLD[world]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(sum(ROWS), C, generator=g) * 0.7 + torch.randn(C, generator=g) * 3
    dy = torch.randn(sum(ROWS), C, generator=g)
    gamma = 1 + 0.1 * torch.randn(C, generator=g); beta = 0.1 * torch.randn(C, generator=g)
    return x, dy, gamma, beta


def _worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_step_gloo import _Rec
    from vitlens_hip import points as PT, step as ST
    PT.ops = _fake_ops()
    x, dy, gamma, beta = _data(world)
    ROWS = ROWS_BY_WORLD[world]
    lo = sum(ROWS[:rank]); s = slice(lo, lo + ROWS[rank])
    k, a = "encoder.first_conv.1", "t."
    # the PRODUCT object through its own __init__ (pure tensor work: masters, running statistics, bf16 operands) on a tiny
    # PointBERT-shaped state_dict, C channels in the first BatchNorm
    gs = torch.Generator().manual_seed(7)
    rn = lambda *sh: torch.randn(*sh, generator=gs) * 0.1
    E, T = 12, 10
    sd = {a + "encoder.first_conv.0.weight": rn(C, 3, 1), a + "encoder.first_conv.0.bias": rn(C),
          a + "encoder.first_conv.3.weight": rn(2 * C, C, 1), a + "encoder.first_conv.3.bias": rn(2 * C),
          a + "encoder.second_conv.0.weight": rn(4 * C, 4 * C, 1), a + "encoder.second_conv.0.bias": rn(4 * C),
          a + "encoder.second_conv.3.weight": rn(E, 4 * C, 1), a + "encoder.second_conv.3.bias": rn(E),
          a + "reduce_dim.weight": rn(T, E), a + "reduce_dim.bias": rn(T),
          a + "pos_embed.0.weight": rn(C, 3), a + "pos_embed.0.bias": rn(C),
          a + "pos_embed.2.weight": rn(T, C), a + "pos_embed.2.bias": rn(T),
          a + "encoder.first_conv.1.weight": gamma.clone(), a + "encoder.first_conv.1.bias": beta.clone(),
          a + "encoder.first_conv.1.running_mean": torch.zeros(C), a + "encoder.first_conv.1.running_var": torch.ones(C),
          a + "encoder.second_conv.1.weight": torch.ones(4 * C), a + "encoder.second_conv.1.bias": torch.zeros(4 * C),
          a + "encoder.second_conv.1.running_mean": torch.zeros(4 * C), a + "encoder.second_conv.1.running_var": torch.ones(4 * C)}
    tr = PT.PointTokenizerTrainer(sd, a, None, "cpu", bn_training=True, bn_sync=_Rec(ST.TorchComm()), world_size=world)
    h, stats = tr._bn(x[s], k)
    dx = tr._bn_bwd(dy[s], x[s], stats, k)
    ret[rank] = dict(log=tr.bn_sync.log, h=h, dx=dx, rm=tr.running[k][0], rv=tr.running[k][1], total=int(stats[2]),
                     dg=tr.grads[a + k + ".weight"], db=tr.grads[a + k + ".bias"])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_syncbn_host_logic_gloo(world):
    ROWS = ROWS_BY_WORLD[world]
    port = 30200 + (os.getpid() + 97 * world) % 190
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    x, dy, gamma, beta = _data(world)
    xr = x.clone().requires_grad_(True); g = gamma.clone().requires_grad_(True); b = beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = torch.relu(torch.nn.functional.batch_norm(xr, rm, rv, g, b, training=True, momentum=0.1, eps=1e-5))
    y.backward(dy)
    for r in range(world):
        o = ret[r]
        assert o["log"] == [("all_gather", (2 * C + 1,)), ("all_reduce", 2 * C)], o["log"]
        lo = sum(ROWS[:r]); s = slice(lo, lo + ROWS[r])
        assert o["total"] == sum(ROWS)
        assert torch.allclose(o["h"], y.detach()[s], atol=1e-5) and torch.allclose(o["dx"], xr.grad[s], atol=1e-5)
        assert torch.allclose(o["rm"], rm, atol=1e-6) and torch.allclose(o["rv"], rv, atol=1e-6)
    assert torch.allclose(sum(ret[r]["dg"] for r in range(world)), g.grad, atol=1e-4)
    assert torch.allclose(sum(ret[r]["db"] for r in range(world)), b.grad, atol=1e-4)
