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
    assert out["n_gpus"] == 1 and out["config"].get("force_dist") is True
    coll = out["collective_ms_per_step"]
    assert "all_gather" in coll and ("all_reduce_wait" in coll or "all_reduce" in coll), coll
    assert out["value"] > 0
