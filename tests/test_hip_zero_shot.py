"""GPU, end to end: zero-shot classification on the HIP path (reference: training/zero_shot.py:155-258
`test_zeroshot_3d_core`, open_clip/zero_shot_classifier.py:27-90).  The classifier is built from prompt templates THROUGH
the HIP text tower (`build_zero_shot_classifier` -> tokenize -> `model.encode_text`), the samples go through
`encode_visual` (depth Lens -> ViT), the scores through `zero_shot_logits`, the metric through `accuracy` / the generic
loop `training.zero_shot.run` - and every stage is compared with the oracle's pipeline on the same seeded weights."""
import importlib
import sys
from types import SimpleNamespace

import pytest
import torch

import vitlens_oracle as O

pytestmark = pytest.mark.gpu

CLASSES = ["airplane", "bathtub", "chair", "guitar", "lamp", "piano", "toilet"]           # ModelNet40-style names
TEMPLATES = ["a point cloud model of {}.", "There is a {} in the scene.", "a depth photo of a {}.", "{}"]


def _oc():
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        f = getattr(sys.modules[k], "__file__", "") or ""
        if "vit-lens_amd" not in f:
            del sys.modules[k]
    oc = importlib.import_module("open_clip")
    assert "vit-lens_amd" in oc.__file__
    return oc


def _unit(x):
    return x / x.norm(dim=-1, keepdim=True)


def test_zero_shot_pipeline_vs_oracle():
    oc = _oc()
    from mm_vit_lens.model_cfg import fetch_model_cfg
    from training.zero_shot import run
    cfg = fetch_model_cfg(modality="depth")
    cfg.perceiver_num_latents = 49                     # ViT-B/32: 7 x 7 depth patches through the identity Perceiver
    torch.manual_seed(0)
    model = oc.tri_create_model("ViT-B-32", None, precision="fp32", device="cuda", output_dict=True, args=cfg)
    model.eval()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    tokenizer = oc.get_tokenizer("ViT-B-32")
    # ---- classifier through the HIP text tower vs the oracle's
    clf = oc.build_zero_shot_classifier(model, tokenizer, CLASSES, TEMPLATES, num_classes_per_batch=3, device="cuda")
    assert clf.shape == (512, len(CLASSES))
    tspec = O.TextSpec(width=512, heads=8, layers=12, embed_dim=512)
    cols = []
    for c in CLASSES:
        emb = O.encode_text(sd, tokenizer([t.format(c) for t in TEMPLATES]), tspec)
        cols.append(_unit(_unit(emb).mean(0)))
    ref_clf = torch.stack(cols, 1)
    assert float((clf.float().cpu() - ref_clf).abs().max()) < 2e-3
    assert float((1 - torch.nn.functional.cosine_similarity(clf.float().cpu().t(), ref_clf.t(), dim=-1)).max()) < 1e-4
    legacy = oc.build_zero_shot_classifier_legacy(model, tokenizer, CLASSES, TEMPLATES, device="cuda")
    assert float((legacy - clf).abs().max()) < 2e-3
    # ---- samples through encode_visual, logits, top-k
    g = torch.Generator().manual_seed(3)
    N = 24
    depth = torch.randn(N, 1, 224, 224, generator=g)
    tower = O.TowerSpec(width=768, layers=12, heads=12, patch=32, image_size=224, embed_dim=512)
    lens = O.LensSpec(modality="depth", perceiver_identity=True)
    ref_f = O.encode_visual(sd, depth, tower, lens, normalize=True)
    with torch.no_grad():
        f = model.encode_visual(depth.cuda(), normalize=True)
    assert float((f.float().cpu() @ f.float().cpu().t() - ref_f @ ref_f.t()).abs().max()) < 1e-3
    logits = oc.zero_shot_logits(f, clf, logit_scale=100.0).float().cpu()
    ref_logits = 100.0 * ref_f @ ref_clf
    err = float((logits - ref_logits).abs().max())
    # visual and text features of a random-init model are nearly orthogonal (|logit| <= ~3 at scale 100): the bound is the
    # absolute one of the cosine matrices (1e-3 x scale), plus 3e-2 relative L2 (measured 1.8e-2, max |err| 0.04)
    assert err < 0.1 and float((logits - ref_logits).norm() / ref_logits.norm()) < 3e-2, err
    # identical top-1 wherever the oracle's own margin is above the numerical noise of the comparison
    top2 = ref_logits.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 4 * err
    assert int(clear.sum()) >= N // 3, (int(clear.sum()), err)
    assert torch.equal(logits.argmax(1)[clear], ref_logits.argmax(1)[clear])
    # the metric and the evaluation loop: targets = the oracle's predictions on the clear samples, a wrong class elsewhere
    target = ref_logits.argmax(1).clone()
    target[~clear] = (ref_logits.argsort(1)[:, 0])[~clear]               # the LOWEST-scoring class: never in anybody's top-1
    n_clear = float(clear.sum())
    a1, a5 = oc.accuracy(logits.cuda(), target.cuda(), topk=(1, 5))
    r1, r5 = oc.accuracy(ref_logits, target, topk=(1, 5))
    assert a1 == r1 == n_clear and a5 == r5
    loader = [(depth[i:i + 8], target[i:i + 8]) for i in range(0, N, 8)]
    top1, top5 = run(model, clf, loader, SimpleNamespace(device="cuda"), input_key="visual_x", feature_key="visual_features")
    assert abs(top1 - n_clear / N) < 1e-9 and abs(top5 - r5 / N) < 1e-9
