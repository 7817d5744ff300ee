"""GPU: SyncBatchNorm for the point-cloud tokenizer (--use-bn-sync, training/point_cloud/pc_tri_main.py:372-373 ->
torch.nn.SyncBatchNorm): the statistics / backward passes split at the rank exchange (vl_bn_stats_local ->
all-gather -> vl_bn_stats_merge; vl_bn_bwd_reduce -> all-reduce -> vl_bn_bwd_apply).  SyncBatchNorm is, by definition,
BatchNorm over the concatenation of the ranks' batches - so the checks are against torch's fp32 F.batch_norm autograd on
the concatenated rows, and against the one-rank kernels / trainer on the global batch."""
import pytest
import torch

from test_hip_train import _pc_cfgs, _rnd, _run_ranks, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,C,cuts", [(4100, 512, (0, 1000, 1003, 4100)), (512, 128, (0, 256, 512)), (333, 64, (0, 1, 333))])
def test_split_kernels_equal_full_batch_and_torch(R, C, cuts):
    from vitlens_hip import ops
    F = torch.nn.functional
    x = (_rnd(R, C, seed=1) * 0.7 + _rnd(C, seed=2) * 3.0).bfloat16()          # columns with |mean| >> std
    gamma = 1 + 0.1 * _rnd(C, seed=3); beta = 0.1 * _rnd(C, seed=4)
    rm0 = 0.1 * _rnd(C, seed=5); rv0 = 1 + 0.2 * torch.rand(C, generator=torch.Generator().manual_seed(6))
    dy = _rnd(R, C, seed=7).bfloat16()
    xr = x.float().requires_grad_(True); g = gamma.clone().requires_grad_(True); b = beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    torch.relu(F.batch_norm(xr, rm, rv, g, b, training=True, momentum=0.1, eps=1e-5)).backward(dy.float())
    xd, dyd, gd, bd = x.cuda(), dy.cuda(), gamma.cuda(), beta.cuda()
    parts = [slice(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
    # forward: per-rank (mean, M2, count) -> merge
    gathered = torch.stack([ops.bn_stats_local(xd[p]) for p in parts])
    rmd, rvd = rm0.clone().cuda(), rv0.clone().cuda()
    mean, var, total = ops.bn_stats_merge(gathered, rmd, rvd, 0.1)
    assert int(total) == R
    assert relerr(mean, xr.detach().mean(0)) < 1e-5 and relerr(var, xr.detach().var(0, unbiased=False)) < 1e-4
    assert relerr(rmd, rm) < 1e-5 and relerr(rvd, rv) < 1e-4
    fm, fv = ops.bn_stats(xd)                                                    # the one-rank kernel on all rows
    assert relerr(mean, fm) < 1e-6 and relerr(var, fv) < 1e-5
    # backward: local dgamma/dbeta + sums -> sum over ranks -> elementwise pass with the global count
    dgs = [torch.zeros(C, device="cuda") for _ in parts]; dbs = [torch.zeros(C, device="cuda") for _ in parts]
    sums = [ops.bn_bwd_reduce(dyd[p], xd[p], mean, var, gd, bd, dgs[i], dbs[i], relu=True) for i, p in enumerate(parts)]
    tot = torch.stack(sums).sum(0)
    dx = torch.cat([ops.bn_bwd_apply(dyd[p], xd[p], mean, var, gd, bd, tot, total, relu=True) for p in parts])
    assert relerr(sum(dgs), g.grad) < 2e-3 and relerr(sum(dbs), b.grad) < 2e-3
    assert relerr(dx, xr.grad) < 6e-3, relerr(dx, xr.grad)
    dg1 = torch.zeros(C, device="cuda"); db1 = torch.zeros(C, device="cuda")
    dx1 = ops.bn_bwd(dyd, xd, fm, fv, gd, bd, dg1, db1, relu=True, train=True)
    assert relerr(dx, dx1) < 4e-3 and relerr(sum(dgs), dg1) < 1e-4 and relerr(sum(dbs), db1) < 1e-4


def test_point_tokenizer_two_ranks_with_syncbn_equal_one_rank_on_the_global_batch():
    """Two ranks (threads on one GPU, in-process communicator), 2 clouds each, SyncBN on, against ONE trainer with plain
    train-mode BatchNorm on the 4-cloud batch: same tokens, same running statistics on every rank, and the ranks'
    parameter gradients add up to the global-batch gradients."""
    from vitlens_hip.points import PointTokenizerTrainer
    sd, ins, outs, grads, tc, xc, lc = _pc_cfgs()
    a = "visual.visual_adapter."
    pts, start = ins["visual_x"].cuda(), ins["fps_start"].cuda()
    one = PointTokenizerTrainer(sd, a, lc, "cuda", bn_training=True)
    ref = one.forward(pts, start)
    dctx = _rnd(*ref.shape, seed=21).cuda()
    one.backward(dctx)
    rows = ref.shape[0] // 2

    def fn(r, comm):
        tr = PointTokenizerTrainer(sd, a, lc, "cuda", bn_training=True, bn_sync=comm, world_size=2)
        s = slice(2 * r, 2 * r + 2)
        out = tr.forward(pts[s], start[s])
        tr.backward(dctx[r * rows:(r + 1) * rows].contiguous())
        torch.cuda.synchronize()
        return out, tr.grads, tr.running
    res = _run_ranks(2, fn)
    got = torch.cat([res[0][0], res[1][0]])
    assert relerr(got, ref) < 4e-3, relerr(got, ref)
    for k in one.running:
        for r in range(2):
            for i in range(2):
                assert relerr(res[r][2][k][i], one.running[k][i]) < 1e-4, (k, r, i)
        assert torch.equal(res[0][2][k][0], res[1][2][k][0]) and torch.equal(res[0][2][k][1], res[1][2][k][1])
    bad = {}
    for name, gref in one.grads.items():
        if float(gref.norm()) < 1e-6 * max(1.0, float(one.masters[name].norm())):
            continue                                                             # identically-zero gradients (bias in front of a BN)
        e = relerr(res[0][1][name] + res[1][1][name], gref)
        if e > 2e-2:
            bad[name[len(a):]] = round(e, 4)
    assert not bad, bad
