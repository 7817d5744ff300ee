"""CPU, multi-process (gloo): the feature all-gather of the contrastive exchange step.  `gather_features`
is device-agnostic host logic; its outputs (ordering) and autograd semantics (peers detached / local slice
differentiable / gather_with_grad = reduce-scatter backward) are checked on world_size 2 and 4 against the
loss values and feature gradients the REFERENCE produced on the same inputs (tests/golden/gather_w*.npz).
The loss formula applied to the gathered features here is the oracle's (test-side checker)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, npz_path, ret):
    for p in (os.path.join(ROOT, "vit-lens_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from open_clip.loss import gather_features, gather_packed
    import vitlens_oracle as O
    z = {k: torch.from_numpy(v) for k, v in np.load(npz_path).items()}
    errs = []
    ls0 = torch.tensor(14.285714)
    for ll in (0, 1):
        for gg in (0, 1):
            x = z[f"in/x{rank}"].clone().requires_grad_(True); y = z[f"in/y{rank}"].clone().requires_grad_(True)
            ls = ls0.clone().requires_grad_(True)
            ax, ay = gather_features(x, y, bool(ll), bool(gg), rank, world)
            b = x.shape[0]
            for r in range(world):   # rank-major order
                if not torch.equal(ax[r * b:(r + 1) * b].detach(), z[f"in/x{r}"]):
                    errs.append(f"order ll{ll} gg{gg} r{r}")
            if ll:
                lab = torch.arange(b) + b * rank
                loss = (O.cross_entropy_rows(ls * x @ ay.t(), lab) + O.cross_entropy_rows(ls * y @ ax.t(), lab)) / 2
            else:
                lab = torch.arange(world * b)
                lg = ls * ax @ ay.t()
                loss = (O.cross_entropy_rows(lg, lab) + O.cross_entropy_rows(lg.t(), lab)) / 2
            loss.backward()
            tag = f"rank{rank}/dual_ll{ll}_gg{gg}"
            for got, name in ((loss.detach(), "_loss"), (x.grad, "_gx"), (y.grad, "_gy"), (ls.grad, "_gls")):
                ref = z[tag + name]
                if not torch.allclose(got, ref, rtol=2e-4, atol=2e-6):
                    errs.append(f"{tag}{name}: {float((got - ref).abs().max()):.3e}")
    # one packed collective for the tri-modal step
    i = z[f"in/x{rank}"].clone().requires_grad_(True); t = z[f"in/y{rank}"].clone().requires_grad_(True)
    v = z[f"in/z{rank}"].clone().requires_grad_(True)
    ai, at, av = gather_packed([i, t, v], False, False, rank, world)
    loss = O.tri_clip_loss(ai, at, av, ls0)
    loss.backward()
    for got, name in ((loss.detach(), "tri_loss"), (i.grad, "tri_gi"), (t.grad, "tri_gt"), (v.grad, "tri_gv")):
        ref = z[f"rank{rank}/{name}"]
        if not torch.allclose(got, ref, rtol=2e-4, atol=2e-6):
            errs.append(f"{name}: {float((got - ref).abs().max()):.3e}")
    ret[rank] = errs
    dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29721), (4, 29723)])
def test_gather_features_semantics(world, port):
    npz = os.path.join(ROOT, "tests", "golden", f"gather_w{world}.npz")
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, npz, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret[r] == [], (r, ret[r])


def _comm_worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vitlens_hip.step import TorchComm
    comm = TorchComm()
    errs = []
    b, E = 3, 8
    local = torch.full((b, E), float(rank + 1))
    out = torch.empty(world * b, E)
    comm.all_gather(out, local)                                   # rank-major concat (loss.py:75-76)
    for r in range(world):
        if not torch.equal(out[r * b:(r + 1) * b], torch.full((b, E), float(r + 1))):
            errs.append(f"all_gather slot {r}")
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    comm.all_reduce_sum(flat)                                     # DDP gradient sum
    if not torch.equal(flat, torch.arange(10, dtype=torch.float32) * sum(range(1, world + 1))):
        errs.append("all_reduce")
    d_all = torch.arange(world * b * E, dtype=torch.float32).reshape(world * b, E) * (rank + 1)
    mine = torch.empty(b, E)
    try:
        comm.reduce_scatter_sum(mine, d_all)                       # backward of a gather WITH grad
        ref = torch.arange(world * b * E, dtype=torch.float32).reshape(world * b, E)[rank * b:(rank + 1) * b] * sum(range(1, world + 1))
        if not torch.equal(mine, ref):
            errs.append("reduce_scatter")
    except RuntimeError as e:                                      # gloo has no reduce_scatter: RCCL ("nccl") does
        if "reduce_scatter" not in str(e).lower() and "not supported" not in str(e).lower() and "unsupported" not in str(e).lower():
            errs.append(f"reduce_scatter raised {e}")
    ret[rank] = errs
    dist.destroy_process_group()


def test_step_communicator_on_gloo():
    """The communicator the fused training steps use (step.TorchComm): its two forward/backward collectives on a
    real 2-process group (gloo stands in for RCCL; reduce_scatter is checked where the backend provides it)."""
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_comm_worker, args=(2, 29727, ret), nprocs=2, join=True)
    for r in range(2):
        assert ret[r] == [], (r, ret[r])


def _utils_worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from open_clip import utils as U
    errs = []
    a = torch.full((3,), float(rank + 1)); b = torch.full((2, 2), 10.0 * (rank + 1))
    U.scaled_all_reduce([a, b])
    tot = sum(range(1, world + 1)) / world
    if not (torch.allclose(a, torch.full((3,), tot)) and torch.allclose(b, torch.full((2, 2), 10.0 * tot))):
        errs.append("scaled_all_reduce")
    c = torch.full((2,), float(rank + 1)); U.scaled_all_reduce([c], is_scale=False)
    if not torch.allclose(c, torch.full((2,), float(sum(range(1, world + 1))))):
        errs.append("all_reduce unscaled")
    g = U.concat_all_gather(torch.full((2, 3), float(rank)))
    if g.shape != (2 * world, 3) or not all(bool((g[2 * r:2 * r + 2] == r).all()) for r in range(world)):
        errs.append("concat_all_gather")
    q = torch.full((rank + 1, 2), float(rank))                     # ragged first dimension
    allq = U.all_gather(q)
    exp = torch.cat([torch.full((r + 1, 2), float(r)) for r in range(world)])
    if not torch.equal(allq, exp):
        errs.append("all_gather ragged")
    others = U.all_gather(q, exclude_self=True)
    exp = torch.cat([torch.full((r + 1, 2), float(r)) for r in range(world) if r != rank])
    if not torch.equal(others, exp):
        errs.append("all_gather exclude_self")
    if U.get_world_size() != world or U.get_rank() != rank or U.is_main_process() != (rank == 0):
        errs.append("rank helpers")
    ret[rank] = errs
    dist.destroy_process_group()


def test_eval_collectives_on_gloo():
    """scaled_all_reduce / concat_all_gather / ragged all_gather of the evaluation path (utils.py:134-175, 295-330)."""
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_utils_worker, args=(3, 29729, ret), nprocs=3, join=True)
    for r in range(3):
        assert ret[r] == [], (r, ret[r])
