"""CPU: host logic of the zero-shot classifier (template averaging, batching, top-k counting) with a stub text encoder,
against a direct restatement; build container only: against the imported reference's builder on the same stub."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _StubModel:
    """encode_text = fixed random projection of a bag-of-token-ids histogram (deterministic, CPU)."""

    def __init__(self, vocab=49408, dim=32):
        g = torch.Generator().manual_seed(0)
        self.table = torch.randn(vocab, dim, generator=g)

    def encode_text(self, ids):
        return self.table[ids].sum(dim=1)


def _oc():
    import importlib
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        del sys.modules[k]
    return importlib.import_module("open_clip")


CLASSES = ["dog", "cat", "guitar", "airplane", "chair", "tree", "piano"]
TEMPLATES = ["a photo of a {}.", "a depth map of a {}.", lambda c: f"a point cloud of a {c}."]


def test_classifier_is_unit_mean_of_unit_template_embeddings():
    oc = _oc()
    m = _StubModel()
    for tmpl in (TEMPLATES[:2], TEMPLATES[2:]):
        w = oc.build_zero_shot_classifier(m, oc.tokenize, CLASSES, tmpl, num_classes_per_batch=3, device="cpu")
        assert w.shape == (32, len(CLASSES))
        for j, c in enumerate(CLASSES):
            e = torch.stack([m.encode_text(oc.tokenize([t.format(c) if isinstance(t, str) else t(c)]))[0] for t in tmpl])
            e = e / e.norm(dim=-1, keepdim=True)
            ref = e.mean(0); ref = ref / ref.norm()
            assert torch.allclose(w[:, j], ref, atol=1e-6)
        w_all = oc.build_zero_shot_classifier(m, oc.tokenize, CLASSES, tmpl, num_classes_per_batch=None, device="cpu")
        assert torch.allclose(w, w_all, atol=1e-6)


def test_accuracy_counts_topk_hits():
    oc = _oc()
    out = torch.tensor([[0.1, 0.9, 0.0, 0.2], [0.8, 0.1, 0.05, 0.7], [0.3, 0.2, 0.1, 0.4]])
    tgt = torch.tensor([1, 3, 2])
    assert oc.accuracy(out, tgt, topk=(1, 2, 4)) == [1.0, 2.0, 3.0]


_REF = r'''
import json, sys, torch
sys.path.insert(0, sys.argv[1])
import ref_loader
oc = ref_loader.load()
from open_clip.zero_shot_classifier import build_zero_shot_classifier
g = torch.Generator().manual_seed(0)
table = torch.randn(49408, 32, generator=g)
class M:
    def encode_text(self, ids): return table[ids].sum(dim=1)
w = build_zero_shot_classifier(M(), oc.tokenize, ["dog", "cat", "guitar", "airplane", "chair", "tree", "piano"],
                               ["a photo of a {}.", "a depth map of a {}."], num_classes_per_batch=3, device="cpu")
from open_clip.zero_shot_classifier import build_zero_shot_classifier_legacy
w2 = build_zero_shot_classifier_legacy(M(), oc.tokenize, ["dog", "cat", "guitar", "airplane", "chair", "tree", "piano"],
                                       ["a photo of a {}.", "a depth map of a {}."], device="cpu")
print("JSON" + json.dumps([w.tolist(), w2.tolist()]))
'''


@pytest.mark.needs_reference
def test_classifier_equals_reference_builder():
    import json
    r = subprocess.run([sys.executable, "-c", _REF, os.path.join(ROOT, "oracle")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref, ref_legacy = (torch.tensor(t) for t in json.loads(r.stdout[r.stdout.index("JSON") + 4:]))
    oc = _oc()
    w = oc.build_zero_shot_classifier(_StubModel(), oc.tokenize, CLASSES, TEMPLATES[:2], num_classes_per_batch=3, device="cpu")
    assert torch.allclose(w, ref, atol=1e-6)
    w2 = oc.build_zero_shot_classifier_legacy(_StubModel(), oc.tokenize, CLASSES, TEMPLATES[:2], device="cpu")
    assert torch.allclose(w2, ref_legacy, atol=1e-6) and torch.allclose(w2, w, atol=1e-6)


_REF_ACC = r'''
import json, sys, torch
sys.path.insert(0, sys.argv[1])
import ref_loader
ref_loader.load()
import training.zero_shot as Z
g = torch.Generator().manual_seed(3)
out = torch.randn(64, 12, generator=g)
tgt = torch.randint(0, 12, (64,), generator=g)
res = {"accuracy": Z.accuracy(out, tgt, topk=(1, 5))}
r, c = Z.acc(out, tgt, topk=(1, 3, 5))
res["acc"] = [float(x) for x in r]; res["acc_correct"] = c.int().tolist()
t2 = tgt.clone()
r, c = Z.cond_acc(out, t2, idx_mapping=[2, 7, 9], merge_idx=100, topk=(1, 3, 5))
res["cond"] = [float(x) for x in r]; res["cond_correct"] = c.int().tolist(); res["cond_target"] = t2.tolist()
print("JSON" + json.dumps(res))
'''


@pytest.mark.needs_reference
def test_accuracy_helpers_equal_reference():
    """accuracy / acc / cond_acc of training/zero_shot.py:36-81 (imported, build container only)."""
    import json
    r = subprocess.run([sys.executable, "-c", _REF_ACC, os.path.join(ROOT, "oracle")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    oc = _oc()
    g = torch.Generator().manual_seed(3)
    out = torch.randn(64, 12, generator=g)
    tgt = torch.randint(0, 12, (64,), generator=g)
    assert oc.accuracy(out, tgt, topk=(1, 5)) == ref["accuracy"]
    res, c = oc.acc(out, tgt, topk=(1, 3, 5))
    assert [float(x) for x in res] == ref["acc"] and c.int().tolist() == ref["acc_correct"]
    t2 = tgt.clone()
    res, c = oc.cond_acc(out, t2, idx_mapping=[2, 7, 9], merge_idx=100, topk=(1, 3, 5))
    assert [float(x) for x in res] == ref["cond"] and c.int().tolist() == ref["cond_correct"] and t2.tolist() == ref["cond_target"]


def test_zero_shot_run_loop_counts(monkeypatch):
    """training.zero_shot.run (reference training/zero_shot.py:84-110) on a stub model: batches of unequal size, top-1 /
    top-5 fractions over all samples; the scoring GEMM is replaced by torch here (it is the HIP kernel on the GPU)."""
    import importlib
    _oc()
    Z = importlib.import_module("training.zero_shot")
    monkeypatch.setattr(Z, "zero_shot_logits", lambda f, w, logit_scale=100.0: logit_scale * f @ w)
    g = torch.Generator().manual_seed(0)
    w = torch.nn.functional.normalize(torch.randn(16, 10, generator=g), dim=0)
    feats = torch.nn.functional.normalize(torch.randn(50, 16, generator=g), dim=-1)
    target = (feats @ w).argmax(1)
    target[:10] = (target[:10] + 1) % 10                                         # 10 wrong labels

    class Model:
        def __call__(self, image=None):
            return {"image_features": image}
    batches = [(feats[:7], target[:7]), (feats[7:30], target[7:30]), (feats[30:], target[30:])]
    from types import SimpleNamespace
    top1, top5 = Z.run(Model(), w, batches, SimpleNamespace(device="cpu"))
    logits = 100.0 * feats @ w
    assert top1 == int((logits.argmax(1) == target).sum()) / 50 == 0.8
    assert top1 <= top5 <= 1.0
