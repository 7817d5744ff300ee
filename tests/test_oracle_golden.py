"""CPU: the oracle (oracle/vitlens_oracle.py) reproduces the golden vectors that
oracle/gen_golden.py obtained from the imported reference (fp32, <=1e-5 relative)."""
import numpy as np
import pytest
import torch

import vitlens_oracle as O
from golden_util import load_npz, split, specs_from_meta

TOL = dict(rtol=2e-5, atol=2e-6)


def close(a, b, **kw):
    t = dict(TOL); t.update(kw)
    torch.testing.assert_close(a, b, **t)


@pytest.mark.parametrize("modality", ["depth", "audio", "pc", "eeg", "tactile", "audio_tied"])
def test_tiny_towers(modality):
    sd, ins, outs, grads, meta = split(load_npz(f"tiny_{modality}.npz"))
    tower, text, lens = specs_from_meta(meta)
    img = O.encode_image(sd, ins["image"], tower)
    close(img, outs["image_raw"])
    close(O.l2_normalize(img), outs["image_features"])
    txt = O.encode_text(sd, ins["text"], text)
    close(txt, outs["text_raw"])
    close(O.l2_normalize(txt), outs["text_features"])
    vis = O.encode_visual(sd, ins["visual_x"], tower, lens, fps_start=ins.get("fps_start"))
    close(vis, outs["visual_raw"], rtol=1e-4, atol=1e-5)
    close(O.l2_normalize(vis), outs["visual_features"], rtol=1e-4, atol=1e-5)
    close(sd["logit_scale"].exp(), outs["logit_scale"])


@pytest.mark.parametrize("modality", ["depth", "audio", "pc", "eeg", "tactile"])
def test_losses_and_feature_grads(modality):
    sd, ins, outs, grads, meta = split(load_npz(f"tiny_{modality}.npz"))
    f = {k: outs[k + "_features"].clone().requires_grad_(True) for k in ("image", "text", "visual")}
    ls = outs["logit_scale"].clone().requires_grad_(True)
    tri = O.tri_clip_loss(f["image"], f["text"], f["visual"], ls)
    close(tri, outs["tri_loss"])
    tri.backward()
    for k in f:
        close(f[k].grad, outs[f"tri_grad_{k}"], atol=1e-6)
    close(ls.grad, outs["tri_grad_logit_scale"], atol=1e-6)
    x = outs["visual_features"].clone().requires_grad_(True)
    y = outs["text_features"].clone().requires_grad_(True)
    ls = outs["logit_scale"].clone().requires_grad_(True)
    dual = O.clip_loss(x, y, ls)
    close(dual, outs["dual_loss"])
    dual.backward()
    close(x.grad, outs["dual_grad_x"], atol=1e-6)
    close(y.grad, outs["dual_grad_y"], atol=1e-6)
    close(ls.grad, outs["dual_grad_logit_scale"], atol=1e-6)


@pytest.mark.parametrize("modality", ["depth", "audio", "eeg", "tactile", "audio_tied"])
def test_step_param_grads(modality):
    """Autograd through the oracle = the reference's backward for the trainable tower."""
    sd, ins, outs, grads, meta = split(load_npz(f"tiny_{modality}.npz"))
    tower, text, lens = specs_from_meta(meta)
    sd = {k: (v.clone().requires_grad_(True) if (k.startswith("visual.") or k == "logit_scale") else v)
          for k, v in sd.items()}
    if meta["args"].get("perceiver_weight_tie_layers"):        # shared modules: one tensor object under every tied layer index
        O.tie_perceiver_layers(sd, lens.depth)
    i = O.encode_image(sd, ins["image"], tower, normalize=True)
    t = O.encode_text(sd, ins["text"], text, normalize=True)
    v = O.encode_visual(sd, ins["visual_x"], tower, lens, normalize=True)
    loss = O.tri_clip_loss(i, t, v, sd["logit_scale"].exp())
    close(loss, outs["step_loss"])
    loss.backward()
    assert len(grads) > 10
    for k, g in grads.items():
        close(sd[k].grad, g, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("bn_train", [False, True])
def test_step_param_grads_point_cloud(bn_train):
    """PC recipe: tokenizer (incl. BatchNorm affine) + Perceiver + ViT gradients; eval-mode BN against tiny_pc.npz,
    train-mode BN (batch statistics + running-stat update) against tiny_pc_bntrain.npz."""
    sd, ins, outs, grads, meta = split(load_npz("tiny_pc.npz"))
    tower, text, lens = specs_from_meta(meta)
    if bn_train:
        _, _, outs, grads, _ = split(load_npz("tiny_pc_bntrain.npz"))
        after = {k[9:]: torch.from_numpy(v) for k, v in load_npz("tiny_pc_bntrain.npz").items() if k.startswith("sd_after/")}
    sd = {k: (v.clone().requires_grad_(True) if ((k.startswith("visual.") and "running" not in k) or k == "logit_scale") else v)
          for k, v in sd.items()}
    running = {}
    i = O.encode_image(sd, ins["image"], tower, normalize=True)
    t = O.encode_text(sd, ins["text"], text, normalize=True)
    v = O.encode_visual(sd, ins["visual_x"], tower, lens, normalize=True, fps_start=ins["fps_start"], training=bn_train,
                        running_out=running)
    close(v, outs["visual_features"], rtol=1e-4, atol=1e-5)
    loss = O.tri_clip_loss(i, t, v, sd["logit_scale"].exp())
    close(loss, outs["step_loss"])
    loss.backward()
    assert len(grads) > 100
    for k, g in grads.items():
        # conv biases in front of a train-mode BatchNorm have a mathematically zero gradient (round-off only)
        atol = 1e-6 if (bn_train and k.endswith(("first_conv.0.bias", "second_conv.0.bias"))) else 2e-6
        close(sd[k].grad, g, rtol=5e-4, atol=atol)
    if bn_train:
        assert len(running) == 4
        for k, r in running.items():
            close(r, after[k], rtol=1e-5, atol=1e-6)


def test_point_grouping_indices():
    sd, ins, outs, grads, meta = split(load_npz("tiny_pc.npz"))
    tower, text, lens = specs_from_meta(meta)
    tok, pos, cidx, nidx = O.point_tokens(sd, "visual.", ins["visual_x"], lens, ins["fps_start"])
    pts = ins["visual_x"]
    B, G = cidx.shape
    center = torch.gather(pts, 1, cidx[:, :, None].expand(B, G, 3))
    assert torch.equal(center, outs["pc_center"])          # FPS indices bit-exact
    nb = torch.gather(pts[:, None].expand(B, G, pts.shape[1], 3), 2,
                      nidx[..., None].expand(B, G, nidx.shape[-1], 3)) - center[:, :, None]
    got = np.sort(nb.numpy().reshape(B, G, -1), axis=-1)   # neighbour order unspecified: compare as sets
    assert np.array_equal(got, outs["pc_neighborhood_sorted"].numpy())
    close(tok, outs["pc_tokens"], rtol=1e-4, atol=1e-5)
    close(pos, outs["pc_pos"])


def test_per_op():
    z = {k: torch.from_numpy(v) for k, v in load_npz("per_op.npz").items()}
    sd = {k[len("blk/sd/"):]: v for k, v in z.items() if k.startswith("blk/sd/")}
    close(O.resblock(sd, "", z["blk/in"], 3), z["blk/out"])
    close(O.resblock(sd, "", z["blk/in"], 3, O.causal_mask(5)), z["blk/out_causal"])
    close(O.layer_norm(z["ln/in"], z["ln/w"], z["ln/b"]), z["ln/out"])
    sd = {k[len("ff/sd/"):]: v for k, v in z.items() if k.startswith("ff/sd/")}
    close(O.lens_ff(sd, "", z["ff/in"]), z["ff/out"])
    sd = {k[len("xattn/sd/"):]: v for k, v in z.items() if k.startswith("xattn/sd/")}
    close(O.lens_attention(sd, "", z["xattn/in"], z["xattn/ctx"], 1, 16), z["xattn/out"])
    close(O.conv_patchify(z["ast/in"].unsqueeze(1).transpose(2, 3), z["ast/w"], (5, 5)), z["ast/out"])


@pytest.mark.parametrize("world", [2, 4])
def test_gathered_loss_values(world):
    z = {k: torch.from_numpy(v) for k, v in load_npz(f"gather_w{world}.npz").items()}
    xs = [z[f"in/x{r}"] for r in range(world)]
    ys = [z[f"in/y{r}"] for r in range(world)]
    zs = [z[f"in/z{r}"] for r in range(world)]
    ls = torch.tensor(14.285714)
    for r in range(world):
        for ll in (0, 1):
            xr = [x.clone().requires_grad_(True) for x in xs]
            yr = [y.clone().requires_grad_(True) for y in ys]
            lsr = ls.clone().requires_grad_(True)
            loss = O.gathered_clip_loss(xr, yr, lsr, r, bool(ll))
            close(loss, z[f"rank{r}/dual_ll{ll}_gg0_loss"])
            loss.backward()
            close(xr[r].grad, z[f"rank{r}/dual_ll{ll}_gg0_gx"], atol=1e-6)
            close(yr[r].grad, z[f"rank{r}/dual_ll{ll}_gg0_gy"], atol=1e-6)
            close(lsr.grad, z[f"rank{r}/dual_ll{ll}_gg0_gls"], atol=1e-5)
        # tri loss value: every rank computes the same global loss
        allx, ally, allz = torch.cat(xs), torch.cat(ys), torch.cat(zs)
        close(O.tri_clip_loss(allx, ally, allz, ls), z[f"rank{r}/tri_loss"])


def test_oracle_on_c1_example_photographs():
    """The oracle against what the IMPORTED reference computed for BASELINE config C1's actual inputs (four example
    photographs + the example.py texts, tests/golden/c1_examples.npz): ViT-B/32 features from the reference pipeline's
    preprocessed tensors and token ids, on the seeded weights the fixture names.  Also: the oracle's numpy restatement of the
    preprocessing reproduces the reference's tensors from the JPEG bytes, byte for byte."""
    import io
    import sys
    import warnings
    from PIL import Image
    import preproc_oracle as po
    z = load_npz("c1_examples.npz")
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        if "vit-lens_amd" not in (getattr(sys.modules[k], "__file__", "") or ""):
            del sys.modules[k]
    import open_clip as oc
    from open_clip.constants import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
    from mm_vit_lens.model_cfg import fetch_model_cfg
    torch.manual_seed(int(z["meta/seed"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = oc.tri_create_model("ViT-B-32", None, precision="fp32", device="cpu", output_dict=True, args=fetch_model_cfg(modality="image"))
    sd = {k: v.detach().float() for k, v in m.state_dict().items()}
    tower = O.TowerSpec(width=768, layers=12, heads=12, patch=32, image_size=224, embed_dim=512)
    tspec = O.TextSpec(width=512, heads=8, layers=12, embed_dim=512)
    fi = O.encode_image(sd, torch.from_numpy(z["pre"]), tower, normalize=True)
    ft = O.encode_text(sd, torch.from_numpy(z["text_ids"]), tspec, normalize=True)
    assert float((fi - torch.from_numpy(z["image_features"])).abs().max()) < 2e-5
    assert float((ft - torch.from_numpy(z["text_features"])).abs().max()) < 2e-5
    p = torch.softmax(100.0 * fi @ ft.t(), dim=-1)
    assert float((p - torch.from_numpy(z["probs"])).abs().max()) < 1e-4
    for i, n in enumerate(["bird", "fire", "dog", "beach"]):
        im = np.array(Image.open(io.BytesIO(z[f"jpeg/{n}"].tobytes())).convert("RGB"))
        got = po.image_eval_transform(im, 224, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD)
        assert np.array_equal(got, z["pre"][i]), n
