"""GPU: the OpenShape flavour (reference VitLens-OpenShape/src): `CLIPBindWrap` (models/clip_bind.py:9-101)
around a pnsa point-cloud Lens + ViT tower, trained against PRECOMPUTED unit-norm image / text features with the
tri-modal loss and its retrieval accuracies (loss.py:80-185) by the step body of train.py:784-843 - against the oracle's
autograd of the same pipeline.  Both projection cases of the wrapper: the tower's own `proj` (out_channel == embed_dim,
proj_layer = Identity) and a dropped `proj` + trainable Linear(width, out_channel)."""
import importlib
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import vitlens_oracle as O
from golden_util import GOLDEN, load_npz, split, specs_from_meta

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _oc():
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        f = getattr(sys.modules[k], "__file__", "") or ""
        if "vit-lens_amd" not in f:
            del sys.modules[k]
    oc = importlib.import_module("open_clip")
    assert "vit-lens_amd" in oc.__file__
    return oc


@pytest.mark.parametrize("out_channel", [32, 48])
def test_clip_bind_step_vs_oracle(out_channel):
    oc = _oc()
    import openshape
    z = np.load(os.path.join(GOLDEN, "tiny_pnsa.npz"))
    cfg = json.loads(str(z["meta"]))["cfg"]
    _, _, _, _, meta = split(load_npz("tiny_pc.npz"))
    a = dict(meta["args"])
    a.update(pc_tokenizer="pnsa", pc_in_channel=cfg["in_dim"], pc_num_group=cfg["num_group"], pc_group_size=cfg["group_size"],
             pc_radius=cfg["radius"], pc_encoder_dims=cfg["encoder_dims"], pc_trans_dim=cfg["trans_dim"],
             perceiver_input_chan=cfg["trans_dim"], skip_trans_first_n_layers=None, unlock_cls=True)
    a.pop("model", None)
    meta = {"args": a, "model_cfg": meta["model_cfg"]}
    E = meta["model_cfg"]["embed_dim"]
    args = SimpleNamespace(**a, model=SimpleNamespace(out_channel=out_channel))
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "tiny-lens.json"), "w") as f:
            json.dump(meta["model_cfg"], f)
        oc.add_model_config(td)
        torch.manual_seed(4)
        tri = oc.tri_create_model("tiny-lens", None, precision="fp32", device="cuda", output_dict=True, args=args)
    tri.load_state_dict({"visual.visual_adapter." + k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd/")}, strict=False)
    wrap = openshape.CLIPBindWrap(args, model=tri).cuda()
    assert isinstance(wrap.proj_layer, torch.nn.Identity) == (out_channel == E)
    if out_channel != E:
        assert wrap.backbone.proj is None and "proj" not in dict(wrap.backbone.named_parameters())
    wrap.lock(unlocked_groups=0, freeze_bn_stats=False, unlock_cls=True)
    wrap.train()
    trainable = {n for n, p in wrap.named_parameters() if p.requires_grad}
    assert "backbone.class_embedding" in trainable and any("visual_adapter.sa." in n for n in trainable)
    assert not any(n.startswith("backbone.transformer.") for n in trainable)
    scale_net = openshape.LogitScaleNetwork().cuda()
    loss_fn = openshape.TriClipLoss()
    params = [p for p in wrap.parameters() if p.requires_grad] + list(scale_net.parameters())
    opt = torch.optim.SGD(params, lr=0.0)                                   # the step runs; parameters and .grad stay for the comparison
    g = torch.Generator().manual_seed(12)
    xyz, feats, start = torch.tensor(z["in/xyz"]), torch.tensor(z["in/features"]), torch.tensor(z["in/fps_start"])
    B = xyz.shape[0]
    text_feat = torch.nn.functional.normalize(torch.randn(B, out_channel, generator=g), dim=-1)
    img_feat = torch.nn.functional.normalize(torch.randn(B, out_channel, generator=g), dim=-1)
    data = {"xyz_dense": xyz, "features_dense": feats, "text_feat": [t[None] for t in text_feat], "img_feat": [t[None] for t in img_feat]}
    out = openshape.openclip_step(wrap, scale_net, loss_fn, opt, data, device="cuda", fps_start=start.cuda())
    assert set(out) == {"contrastive_loss", "i_contra_loss", "t_contra_loss", "i2v_acc", "v2i_acc", "t2v_acc", "v2t_acc"}
    # ---- the oracle's pipeline on the same parameters
    sd = {"visual." + k: v.detach().float().cpu().clone() for k, v in wrap.backbone.state_dict().items()}
    tower, _, lens = specs_from_meta(meta)
    names = ["visual." + n[len("backbone."):] for n in trainable if n.startswith("backbone.")]
    sdr = dict(sd)
    for n in names:
        sdr[n] = sd[n].clone().requires_grad_(True)
    lin = None
    if out_channel != E:
        sdr["visual.proj"] = torch.eye(tower.width)                         # no projection inside the tower
        lin = (wrap.proj_layer.weight.detach().float().cpu().clone().requires_grad_(True),
               wrap.proj_layer.bias.detach().float().cpu().clone().requires_grad_(True))
    ls = scale_net.logit_scale.detach().float().cpu().clone().requires_grad_(True)
    t, _, _ = O.pnsa_tokens(sdr, "visual.visual_adapter.", feats, xyz, cfg["num_group"], cfg["radius"], cfg["group_size"], start, training=True)
    pred = O.vit_trunk(sdr, "visual.", O.perceiver(sdr, "visual.perceiver.", t, lens), tower, lens.use_orig_pos)
    if lin is not None:
        pred = pred @ lin[0].t() + lin[1]
    pred = O.l2_normalize(pred)
    li, lt = O.clip_loss(img_feat, pred, ls.exp()), O.clip_loss(text_feat, pred, ls.exp())
    (li + lt).backward()
    assert abs(float(out["contrastive_loss"]) - float(li + lt)) < 3e-2
    assert abs(float(out["i_contra_loss"]) - float(li)) < 2e-2 and abs(float(out["t_contra_loss"]) - float(lt)) < 2e-2
    lab = torch.arange(B)
    sim_i, sim_t = img_feat @ pred.detach().t(), text_feat @ pred.detach().t()
    acc = lambda s: float((s.argmax(1) == lab).float().mean())
    # accuracies: compared where the oracle's arg-max is not a near tie
    def clear(s):
        top2 = s.topk(2, dim=1).values
        return bool(((top2[:, 0] - top2[:, 1]) > 2e-3).all())
    if clear(sim_i) and clear(sim_i.t()) and clear(sim_t):
        assert abs(float(out["i2v_acc"]) - acc(sim_i)) < 1e-6 and abs(float(out["v2i_acc"]) - acc(sim_i.t())) < 1e-6
        assert abs(float(out["t2v_acc"]) - acc(sim_t)) < 1e-6
    assert float(out["v2t_acc"]) == float(out["t2v_acc"])                  # loss.py:161
    # ---- gradients
    got = {n: p.grad for n, p in wrap.named_parameters() if p.requires_grad}
    checked = 0
    for n in names:
        gk = got["backbone." + n[len("visual."):]]
        assert gk is not None, n
        if "mlp_convs" in n and n.endswith(".bias"):
            continue
        tol = 0.30 if "visual_adapter.sa" in n else 8e-2
        assert relerr(gk, sdr[n].grad) < tol, (n, relerr(gk, sdr[n].grad))
        checked += 1
    assert checked >= 20
    gs, rs = float(scale_net.logit_scale.grad), float(ls.grad)
    assert abs(gs - rs) < 5e-2 * abs(rs) + 2e-3, (gs, rs)
    if lin is not None:
        assert relerr(wrap.proj_layer.weight.grad, lin[0].grad) < 6e-2 and relerr(wrap.proj_layer.bias.grad, lin[1].grad) < 6e-2
