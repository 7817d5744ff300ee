"""GPU: the RCCL code path of the training steps on ONE GPU (world size 1, backend "nccl").

`SCALE_r0x.json` has been a skipped record in every round (no multi-GPU node was reachable from the build), so until this
test no line of the NCCL branch of `vitlens_hip.step.TorchComm` had executed on hardware.  The worker
(tests/rccl_w1_worker.py, own process) initialises a one-rank RCCL process group and drives the depth / audio / point-cloud
steps with `force_comm=True` through the real communicator: `all_gather_into_tensor`, `reduce_scatter_tensor` under
gather_with_grad, the async per-block all-reduce buckets and their `wait()` ordering, the flat all-reduce, SyncBatchNorm's
exchange - compared with the same steps built without a communicator (bit-equal where the arithmetic is the same).
Reference: training/distributed.py:95-108, open_clip/loss.py:55-76."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_steps_on_a_one_rank_rccl_communicator():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_w1_worker.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-6000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ok"] and out["backend"] == "nccl"
    assert len(out["cases"]) == 8
    print(json.dumps(out["cases"], indent=1))
    for c in out["cases"]:
        if not c["local_loss"] and not c["bn_sync"]:
            assert c["worst_rel_diff"] == 0.0, c
    assert any("reduce_scatter" in c["collectives"] for c in out["cases"])
    assert any("all_reduce_async" in c["collectives"] for c in out["cases"])


def test_bench_force_dist_reports_the_collectives():
    """`bench.py --gpus 1 --force-dist`: the driver's script on a one-rank RCCL group; the JSON line carries the exchange's
    time per step."""
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1",
                        "--batch", "64", "--micro-batch", "32", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    # the result line is the LAST line of stdout: RCCL's version banner (C stdio, flushed at exit) used to follow it
    assert [l for l in r.stdout.splitlines() if l.strip()][-1].startswith("{"), r.stdout[-600:]
    assert out["n_gpus"] == 1 and out["config"].get("force_dist") is True
    coll = out["collective_ms_per_step"]
    assert "all_gather" in coll and ("all_reduce_wait" in coll or "all_reduce" in coll), coll
    assert out["value"] > 0


def test_abi_collectives_on_a_one_rank_communicator():
    """The C ABI's own exchange entries (include/vitlens_hip.h: vl_comm_create, vl_allgather_embed, vl_reducescatter_grad,
    vl_allreduce_grad; csrc/vl_comm.cpp - RCCL resolved at run time) on ONE GPU through `vitlens_hip.step.AbiComm`: at world size
    1 every collective is the identity; then the depth step with `force_comm=True` on that communicator - packed all-gather,
    reduce-scatter under gather_with_grad, async per-block buckets on the communicator's stream, the flat remainder - must be
    BIT-equal to the step without a communicator.  Own process: the RCCL communicator must not leak into the pytest session."""
    code = r'''
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1])))
for p in (ROOT, os.path.join(ROOT, "vit-lens_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from rccl_w1_worker import cfgs
from vitlens_hip import step as ST
comm = ST.AbiComm(0, 1)
x = torch.randn(6, 5, device="cuda"); g = torch.empty(6, 5, device="cuda")
comm.all_gather(g, x); torch.cuda.synchronize(); assert torch.equal(g, x)
r = torch.empty(6, 5, device="cuda"); comm.reduce_scatter_sum(r, x); torch.cuda.synchronize(); assert torch.equal(r, x)
y = x.clone(); comm.all_reduce_sum(y); torch.cuda.synchronize(); assert torch.equal(y, x)
z = x.clone(); h = comm.all_reduce_sum_async(z); h.wait(); torch.cuda.synchronize(); assert torch.equal(z, x)
sd, ins, tc, xc, lc = cfgs("tiny_depth.npz")
args = (ins["image"].cuda(), ins["text"].cuda(), ins["visual_x"].cuda())
res = {}
for gwg in (False, True):
    runs = []
    for c in (None, comm):
        st = ST.TriModalDepthStep(sd, tc, xc, "cuda", micro_batch=2, unlock_first_n=1, lr=1e-3, comm=c, force_comm=c is not None,
                                  gather_with_grad=gwg)
        losses = [float(st.step(*args)) for _ in range(3)]
        torch.cuda.synchronize()
        runs.append((losses, {k: v.clone() for k, v in st.masters.items()}))
    same = runs[0][0] == runs[1][0] and all(torch.equal(runs[0][1][k], runs[1][1][k]) for k in runs[0][1])
    res[str(gwg)] = same
comm.close()
print(json.dumps(res))
'''
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-5000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out == {"False": True, "True": True}, out
