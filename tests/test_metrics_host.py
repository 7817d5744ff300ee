"""CPU: the evaluation metric accumulators (open_clip/metrics: Accuracy, MAP, Recall) against the
imported reference classes (build container only; the reference hard-codes `.cuda()` in `initialize`, which the
subprocess maps to the identity) and, on gloo world 2, the cross-rank merge against one process on the concatenated data."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data(seed=0):
    g = torch.Generator().manual_seed(seed)
    n, c, e, nt = 57, 12, 16, 23
    logits = torch.randn(n, c, generator=g)
    labels = torch.randint(0, c, (n,), generator=g)
    multi = (torch.rand(n, c, generator=g) < 0.25).float()
    multi[torch.arange(n), labels] = 1.0
    multi[:c] = torch.maximum(multi[:c], torch.eye(c))                            # every class has a positive
    img = torch.nn.functional.normalize(torch.randn(n, e, generator=g), dim=-1)
    txt = torch.nn.functional.normalize(torch.randn(nt, e, generator=g), dim=-1)
    img_ids = torch.randint(0, nt, (n,), generator=g)                            # id of the caption that matches each image
    return dict(logits=logits, labels=labels, multi=multi, img=img, txt=txt, img_ids=img_ids, txt_ids=torch.arange(nt), ids=torch.arange(n))


_REF = r'''
import json, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
torch.Tensor.cuda = lambda self, *a, **k: self
import ref_loader
ref_loader.load()
from open_clip.metrics import Accuracy, MAP, Recall
from test_metrics_host import _data
d = _data()
cuts = [0, 20, 21, 57]
out = {}
for name, tg in (("acc", d["labels"]), ("acc_multi", d["multi"])):
    a = Accuracy(); a.initialize()
    for lo, hi in zip(cuts, cuts[1:]):
        a.compute(d["ids"][lo:hi], d["logits"][lo:hi], tg[lo:hi])
    out[name] = a.merge_results(output_predict=True)
m = MAP(); m.initialize()
for lo, hi in zip(cuts, cuts[1:]):
    m.compute(d["ids"][lo:hi], d["logits"][lo:hi], d["multi"][lo:hi])
r = m.merge_results(output_predict=False)
out["map"] = {"map": float(r["map"]), "map_cnt": int(r["map_cnt"])}
rc = Recall(); rc.initialize(d["txt_ids"], d["txt"])
for lo, hi in zip(cuts, cuts[1:]):
    rc.compute(d["img_ids"][lo:hi], d["img"][lo:hi])
out["recall"] = rc.merge_results(output_predict=True)
print("JSON" + json.dumps(out))
'''


def _mine(d, cuts=(0, 20, 21, 57)):
    from open_clip.metrics import MAP, Accuracy, Recall
    out = {}
    for name, tg in (("acc", d["labels"]), ("acc_multi", d["multi"])):
        a = Accuracy(); a.initialize()
        for lo, hi in zip(cuts, cuts[1:]):
            a.compute(d["ids"][lo:hi], d["logits"][lo:hi], tg[lo:hi])
        out[name] = a.merge_results(output_predict=True)
    m = MAP(); m.initialize()
    for lo, hi in zip(cuts, cuts[1:]):
        m.compute(d["ids"][lo:hi], d["logits"][lo:hi], d["multi"][lo:hi])
    r = m.merge_results()
    out["map"] = {"map": float(r["map"]), "map_cnt": int(r["map_cnt"])}
    rc = Recall(); rc.initialize(d["txt_ids"], d["txt"])
    for lo, hi in zip(cuts, cuts[1:]):
        rc.compute(d["img_ids"][lo:hi], d["img"][lo:hi])
    rc.image_ids, rc.image_logits = rc.gathered()
    sim = rc.image_logits @ d["txt"].t()                                         # the GEMM runs on the HIP kernel in merge_results (GPU test)
    out["recall"] = rc.retrieval_eval(sim, sim.t(), output_predict=True)
    return out


def _norm(o):
    return json.loads(json.dumps(o))                                             # int keys -> str, tuples -> lists, as the reference's dump


@pytest.mark.needs_reference
def test_metrics_equal_the_reference():
    r = subprocess.run([sys.executable, "-c", _REF, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    mine = _norm(_mine(_data()))
    assert mine["acc"] == ref["acc"] and mine["acc_multi"] == ref["acc_multi"]
    assert mine["map"]["map_cnt"] == ref["map"]["map_cnt"] and abs(mine["map"]["map"] - ref["map"]["map"]) < 1e-9
    assert mine["recall"] == ref["recall"]
    assert 0 < ref["recall"]["txt_r10"] < 100 and 0 < ref["acc"]["accuracy"] < 1    # the case is not degenerate


def _worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = _data()
    if world == 2:
        lo, hi = (0, 20) if rank == 0 else (20, 57)                               # ragged split: the gathers must handle unequal shares
        cuts = (0, 7, hi - lo)
    else:                                                                         # world 3: the last rank has NO batch at all
        lo, hi = ((0, 20), (20, 57), (57, 57))[rank]
        cuts = (0, 7, hi - lo) if hi > lo else (0,)
    ret[rank] = _norm(_mine({k: (v[lo:hi] if v.shape[0] == 57 else v) for k, v in d.items()}, cuts=cuts))
    dist.destroy_process_group()


def test_metrics_merge_across_ranks_on_gloo():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, 30070 + os.getpid() % 100, ret), nprocs=world, join=True)
    one = _norm(_mine(_data()))
    for r in range(world):
        assert ret[r] == one, r


def test_metrics_merge_with_a_rank_that_saw_no_batch():
    """Fewer batches than ranks (the tail of an evaluation set): the empty rank learns dtype / trailing shape of every gathered
    tensor from its peers (BaseMetric._collected) instead of offering a 1-d int64 placeholder to a 2-d float gather."""
    world = 3
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, 30170 + os.getpid() % 100, ret), nprocs=world, join=True)
    one = _norm(_mine(_data()))
    for r in range(world):
        assert ret[r] == one, r


def test_multi_hot_accuracy_counts_the_target_value():
    """Soft / weighted multi-hot targets: the reference adds up `targets.gather(1, pred)` (metrics/accuracy.py:22-23), not a 0/1 hit."""
    from open_clip.metrics import Accuracy
    logits = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5]])
    targets = torch.tensor([[0.0, 0.5, 1.0], [2.0, 0.0, 0.0], [1.0, 0.0, 0.0]])
    a = Accuracy(); a.initialize()
    a.compute(torch.arange(3), logits, targets)
    r = a.merge_results()
    assert r["score_sum"] == 2.5 and r["score_cnt"] == 3 and abs(r["accuracy"] - 2.5 / 3) < 1e-12
