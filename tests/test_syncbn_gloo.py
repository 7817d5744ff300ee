"""CPU, 2 processes on gloo: the SyncBatchNorm host logic of the point-cloud tokenizer (`PointTokenizerTrainer._bn` /
`_bn_bwd`, --use-bn-sync) through the real `TorchComm`, with the HIP `ops` replaced by a torch-CPU stand-in of the four
split passes.  Asserted per rank: the collective sequence and payloads (ONE all-gather of 2C+1 floats per BatchNorm
forward, ONE all-reduce of 2C floats per backward), and the numbers - outputs, input gradients and running statistics
equal torch's F.batch_norm autograd on the CONCATENATION of the ranks' (unequal) batches, which is what
torch.nn.SyncBatchNorm computes; dgamma / dbeta stay local sums, as SyncBatchNorm leaves them for DDP."""
import os
import sys
import types

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = 8
ROWS = (5, 11)            # unequal per-rank row counts: the merge must weight by the gathered counts


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


def _data():
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
    x, dy, gamma, beta = _data()
    lo = sum(ROWS[:rank]); s = slice(lo, lo + ROWS[rank])
    k, a = "encoder.first_conv.1", "t."
    tr = PT.PointTokenizerTrainer.__new__(PT.PointTokenizerTrainer)
    tr.a, tr.device, tr.bn_training, tr.world = a, torch.device("cpu"), True, world
    tr.bn_sync = _Rec(ST.TorchComm())
    tr.masters = {a + k + ".weight": gamma.clone(), a + k + ".bias": beta.clone()}
    tr.running = {k: (torch.zeros(C), torch.ones(C))}
    tr.grads = {}
    h, stats = tr._bn(x[s], k)
    dx = tr._bn_bwd(dy[s], x[s], stats, k)
    ret[rank] = dict(log=tr.bn_sync.log, h=h, dx=dx, rm=tr.running[k][0], rv=tr.running[k][1], total=int(stats[2]),
                     dg=tr.grads[a + k + ".weight"], db=tr.grads[a + k + ".bias"])
    dist.destroy_process_group()


def test_syncbn_host_logic_world2_gloo():
    world = 2
    port = 30200 + os.getpid() % 90
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    x, dy, gamma, beta = _data()
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
    assert torch.allclose(ret[0]["dg"] + ret[1]["dg"], g.grad, atol=1e-4)
    assert torch.allclose(ret[0]["db"] + ret[1]["db"], b.grad, atol=1e-4)
