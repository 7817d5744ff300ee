"""CPU: the per-epoch training drivers (training/train.py: `tri_train_one_epoch`, `train_dual_one_epoch`) against the
imported reference functions (build container only) on a small pure-torch stand-in model and loss: after one epoch with
a cosine schedule, gradient clipping and - in the second configuration - feature-cache gradient accumulation, every
parameter must be what the reference's loop leaves behind.  The loops are model-agnostic host logic; what they drive on
the GPU is covered by tests/test_hip_api.py (the same call sequence on the HIP towers)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_COMMON = r'''
import math, sys, types, torch
import torch.nn as nn
import torch.nn.functional as F

class Stub(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.img, self.txt, self.vis = nn.Linear(12, 8), nn.Embedding(50, 8), nn.Linear(6, 8)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
    def encode_image(self, x, normalize=False):
        f = self.img(x); return F.normalize(f, dim=-1) if normalize else f
    def encode_text(self, t, normalize=False):
        f = self.txt(t).mean(1); return F.normalize(f, dim=-1) if normalize else f
    def encode_visual(self, v, normalize=False):
        f = self.vis(v); return F.normalize(f, dim=-1) if normalize else f
    def forward(self, image=None, text=None, visual_x=None):
        return {"image_features": self.encode_image(image, True), "text_features": self.encode_text(text, True),
                "visual_features": self.encode_visual(visual_x, True), "logit_scale": self.logit_scale.exp()}

def ce(a, b, s):
    l = s * a @ b.t(); y = torch.arange(l.shape[0])
    return (F.cross_entropy(l, y) + F.cross_entropy(l.t(), y)) / 2

def tri_loss(image_features, text_features, visual_features, logit_scale, output_dict=False):
    return {"contrastive_loss": ce(image_features, visual_features, logit_scale) + ce(text_features, visual_features, logit_scale)}

def dual_loss(A_features, B_features, logit_scale, output_dict=False, key="x"):
    return {key: ce(A_features, B_features, logit_scale)}

class Loader(list):
    pass

def make_data(n_batches=6, bs=5):
    g = torch.Generator().manual_seed(1)
    dl = Loader({"image": torch.randn(bs, 12, generator=g), "caption": torch.randint(0, 50, (bs, 7), generator=g),
                 "depth": torch.randn(bs, 6, generator=g)} for _ in range(n_batches))
    dl.num_batches, dl.num_samples = n_batches, n_batches * bs
    return {"train": types.SimpleNamespace(dataloader=dl, set_epoch=lambda e: None)}

def make_args(accum):
    return types.SimpleNamespace(device="cpu", precision="fp32", distill=False, accum_freq=accum, skip_scheduler=False, v_key="depth",
                                 contra_loss_type="general", horovod=False, grad_clip_norm=0.5, log_every_n_steps=2, world_size=1,
                                 batch_size=5, wandb=False, rank=0, local_rank=0, align_to="text")

def drive(T, S, which, accum):
    model = Stub()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)
    args = make_args(accum)
    sched = S.cosine_lr(opt, 1e-2, 2, 2 * (6 // accum))
    fn, loss = (T.tri_train_one_epoch, tri_loss) if which == "tri" else (T.train_dual_one_epoch, dual_loss)
    for epoch in range(2):
        fn(model, make_data(), loss, epoch, opt, None, sched, None, args)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}
'''

_REF = _COMMON + r'''
sys.path.insert(0, sys.argv[1])
import ref_loader
ref_loader.load()
import training.train as T, training.scheduler as S
out = {f"{w}{a}": drive(T, S, w, a) for w in ("tri", "dual") for a in (1, 3)}
torch.save(out, sys.argv[2])
'''


@pytest.mark.needs_reference
def test_epoch_drivers_equal_the_reference(tmp_path):
    r = subprocess.run([sys.executable, "-c", _REF, os.path.join(ROOT, "oracle"), str(tmp_path / "ref.pt")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = torch.load(tmp_path / "ref.pt")
    ns = {}
    exec(_COMMON, ns)
    import training.scheduler as S
    import training.train as T
    assert "vit-lens_amd" in T.__file__
    for w in ("tri", "dual"):
        for a in (1, 3):
            got = ns["drive"](T, S, w, a)
            for k, v in ref[f"{w}{a}"].items():
                assert torch.allclose(got[k], v, rtol=0, atol=1e-7), (w, a, k, float((got[k] - v).abs().max()))
    # the run did something: parameters moved, logit_scale stayed inside its clamp
    assert float((ref["tri1"]["img.weight"] - ns["Stub"]().img.weight).abs().max()) > 1e-3
    assert 0 <= float(ref["tri3"]["logit_scale"]) <= 4.6052


def test_refusals_and_meter():
    from types import SimpleNamespace
    import training.train as T
    m = T.AverageMeter(); m.update(2.0, 3); m.update(4.0, 1)
    assert m.val == 4.0 and m.sum == 10.0 and m.count == 4 and m.avg == 2.5
    for bad in (dict(distill=True), dict(contra_loss_type="label_mask"), dict(horovod=True)):
        with pytest.raises(NotImplementedError):
            T._refuse(SimpleNamespace(**bad))
