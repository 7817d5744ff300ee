"""CPU: the drop-in boundary.  state_dict names/shapes of the product's TriCLIP equal the reference's for
every hot-path modality (build container only: imports the reference), and the public names exist."""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_REF_KEYS = r'''
import json, sys, tempfile, os, torch
sys.path.insert(0, sys.argv[1])
import ref_loader
oc = ref_loader.load()
sys.path.insert(0, os.path.join(sys.argv[1]))
import gen_golden as G
out = {}
with tempfile.TemporaryDirectory() as td:
    json.dump(G.TINY, open(os.path.join(td, "tiny-lens.json"), "w"))
    oc.add_model_config(td)
    for m in ("depth", "audio", "pc", "eeg"):
        model = oc.tri_create_model("tiny-lens", None, precision="fp32", device="cpu", output_dict=True, args=G.tiny_args(m))
        model.lock_image_tower(); model.lock_text_tower()
        model.lock_visual_tower(unlock_trans_first_n_layers=1, unlock_cls=(m == "audio"))
        out[m] = {"keys": {k: list(v.shape) for k, v in model.state_dict().items()},
                  "trainable": sorted(n for n, p in model.named_parameters() if p.requires_grad),
                  "trainable_groups": {},
                  "args": {k: v for k, v in G.tiny_args(m).items() if isinstance(v, (int, float, str, bool, type(None)))}}
        for k in (1, 2, 3):       # grouped (LiT) unlock from the tail, + positional embedding
            model.lock_visual_tower(unlocked_groups=k, unlock_pos_emb=True)
            out[m]["trainable_groups"][str(k)] = sorted(n for n, p in model.named_parameters() if p.requires_grad and n.startswith("visual."))
print("JSON" + json.dumps(out))
'''


@pytest.mark.needs_reference
def test_state_dict_and_lock_recipes_match_reference():
    r = subprocess.run([sys.executable, "-c", _REF_KEYS, os.path.join(ROOT, "oracle")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    import importlib
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        del sys.modules[k]
    oc = importlib.import_module("open_clip")
    assert "vit-lens_amd" in oc.__file__
    from types import SimpleNamespace
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden as G
    with tempfile.TemporaryDirectory() as td:
        json.dump(G.TINY, open(os.path.join(td, "tiny-lens.json"), "w"))
        oc.add_model_config(td)
        for m in ("depth", "audio", "pc", "eeg"):
            args = SimpleNamespace(**ref[m]["args"])
            model = oc.tri_create_model("tiny-lens", None, device="cpu", output_dict=True, args=args)
            mine = {k: list(v.shape) for k, v in model.state_dict().items()}
            assert mine == ref[m]["keys"], (m, set(mine) ^ set(ref[m]["keys"]),
                                           [k for k in mine if k in ref[m]["keys"] and mine[k] != ref[m]["keys"][k]])
            model.lock_image_tower(); model.lock_text_tower()
            model.lock_visual_tower(unlock_trans_first_n_layers=1, unlock_cls=(m == "audio"))
            tr = sorted(n for n, p in model.named_parameters() if p.requires_grad)
            assert tr == ref[m]["trainable"], (m, set(tr) ^ set(ref[m]["trainable"]))
            for k in (1, 2, 3):
                model.lock_visual_tower(unlocked_groups=k, unlock_pos_emb=True)
                tr = sorted(n for n, p in model.named_parameters() if p.requires_grad and n.startswith("visual."))
                assert tr == ref[m]["trainable_groups"][str(k)], (m, k, set(tr) ^ set(ref[m]["trainable_groups"][str(k)]))


def test_public_surface():
    import importlib
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        del sys.modules[k]
    oc = importlib.import_module("open_clip")
    for name in ("ModalityType", "tokenize", "get_tokenizer", "tri_create_model", "tri_create_model_and_transforms",
                 "create_loss", "ClipLoss", "ClipLossGeneral", "TriClipLoss", "gather_features", "list_models",
                 "add_model_config", "get_model_config", "TriCLIP", "image_transform", "AugmentationCfg"):
        assert hasattr(oc, name), name
    assert oc.ModalityType.PC == "pc" and oc.ModalityType.DEPTH == "depth"
    assert "ViT-L-14" in oc.list_models()
    import mm_vit_lens
    assert hasattr(mm_vit_lens.ViTLens, "encode")


def test_cpu_model_forward_fails_loudly():
    import importlib
    oc = importlib.import_module("open_clip")
    from mm_vit_lens.model_cfg import fetch_model_cfg
    m = oc.tri_create_model("ViT-B-32", device="cpu", args=fetch_model_cfg(modality="image"))
    with pytest.raises(RuntimeError):
        m.encode_image(torch.zeros(1, 3, 224, 224))


# ------------------------------------------------------------------------------------------------ checkpoints / registry
def _tiny_vitlens(monkeypatch, modalities):
    """ViTLens on the tiny golden configuration (CPU construction only; the forward needs a GPU)."""
    import importlib
    from types import SimpleNamespace
    from golden_util import load_npz, split
    oc = importlib.import_module("open_clip")
    import mm_vit_lens.vitlens as V
    metas = {m: split(load_npz(f"tiny_{m}.npz"))[4] for m in ("depth", "audio")}
    td = tempfile.mkdtemp()
    json.dump(metas["depth"]["model_cfg"], open(os.path.join(td, "tiny-lens.json"), "w"))
    V.tri_create_model.__globals__["add_model_config"](td)      # the registry ViTLens resolves names in (test_public_surface re-imports open_clip)

    def fake_cfg(modality, model_option="vitlensL"):
        a = dict(metas["depth" if modality in ("image", "text") else modality]["args"])
        if modality in ("image", "text"):
            a.update(visual_modality_type="image", use_perceiver=False, use_visual_adapter=False)
        a["model"] = "tiny-lens"
        return SimpleNamespace(**a)
    monkeypatch.setattr(V, "fetch_model_cfg", fake_cfg)
    return V.ViTLens(modality_loaded=list(modalities), device="cpu")


def test_vitlens_checkpoint_roundtrip_and_reference_format(monkeypatch, tmp_path):
    """export_checkpoint writes the reference's release layout (`vitlens.<modality>.<submodule keys>`: the image tower,
    the text tower, the modality's `visual` tower - mm_vit_lens/vitlens.py:63-118,153-159), load_checkpoint reads it back
    (round trip is exact), and a hand-made reference-format file lands in the right parameters."""
    from golden_util import load_npz, split
    mods = ["image", "text", "depth", "audio"]
    torch.manual_seed(0)
    a = _tiny_vitlens(monkeypatch, mods)
    path = str(tmp_path / "vitlens.pt")
    a.export_checkpoint(path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert ck["model_var"] == "vitlensL" and ck["modality_loaded"] == mods
    keys = set(ck["state_dict"].keys())
    assert "vitlens.image.conv1.weight" in keys and "vitlens.image.transformer.resblocks.0.attn.in_proj_weight" in keys
    assert "vitlens.text.token_embedding.weight" in keys and "vitlens.text.text_projection" in keys
    assert "vitlens.text.transformer.resblocks.0.mlp.c_fc.weight" in keys
    assert "vitlens.depth.visual_adapter.conv1.weight" in keys and "vitlens.audio.perceiver.latents" in keys
    assert not [k for k in keys if ".visual." in k or k.startswith("vitlens.depth.image.") or k.startswith("vitlens.image.image.")]
    assert not [k for k in keys if k.startswith("vitlens.depth.transformer.resblocks.0.") and ck["state_dict"][k].shape[-1] ==
                a.vitlens["text"].positional_embedding.shape[1] and "visual" in k]
    torch.manual_seed(1)
    b = _tiny_vitlens(monkeypatch, mods)
    assert not torch.equal(a.vitlens["depth"].class_embedding, b.vitlens["depth"].class_embedding)
    missing, unexpected = b.load_checkpoint(path)
    assert not missing and not unexpected
    sa, sb = a.state_dict(), b.state_dict()
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k], sb[k]) for k in sa)
    # the reference's `load_from_ckpt` is the DIRECTORY holding <model_var>.pt (vitlens.py:24,118-133); no download here
    rel = tmp_path / "model_release"
    rel.mkdir()
    with pytest.raises(FileNotFoundError):
        b.load_checkpoint(str(rel))
    a.export_checkpoint(str(rel / "vitlensL.pt"))
    torch.manual_seed(5)
    e = _tiny_vitlens(monkeypatch, mods)
    assert e.load_checkpoint(str(rel)) == ([], [])
    assert all(torch.equal(sa[k], v) for k, v in e.state_dict().items())
    # a release-format file built from the reference-generated golden weights
    sd = split(load_npz("tiny_depth.npz"))[0]
    ref_fmt = {"vitlens.depth." + k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
    ref_fmt.update({"vitlens.image." + k[len("image."):]: v for k, v in sd.items() if k.startswith("image.")})
    ref_fmt.update({"vitlens.text." + k: v for k, v in sd.items()
                    if k.startswith(("transformer.", "token_embedding.", "ln_final.")) or k in ("positional_embedding", "text_projection")})
    torch.save({"state_dict": {"module." + k: v for k, v in ref_fmt.items()}}, path)
    c = _tiny_vitlens(monkeypatch, ["image", "text", "depth"])
    missing, unexpected = c.load_checkpoint(path)
    assert not missing and not unexpected
    assert torch.equal(c.vitlens["depth"].visual_adapter.conv1.weight, sd["visual.visual_adapter.conv1.weight"])
    assert torch.equal(c.vitlens["image"].image.conv1.weight, sd["image.conv1.weight"])
    assert torch.equal(c.vitlens["text"].token_embedding.weight, sd["token_embedding.weight"])
    # training checkpoint -> one modality (load_modality_from_pt_ckpt)
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, path)
    torch.manual_seed(2)
    d = _tiny_vitlens(monkeypatch, ["depth"])
    d.load_modality_from_pt_ckpt("depth", path)
    assert torch.equal(d.vitlens["depth"].positional_embedding, sd["visual.positional_embedding"])


def test_list_models_natural_order_and_pos_embed_resize():
    import importlib
    oc = importlib.import_module("open_clip")
    from open_clip import factory as F
    with tempfile.TemporaryDirectory() as td:
        base = oc.get_model_config("ViT-B-32")
        for n in ("zz-net-2", "zz-net-10", "zz-net-1"):
            json.dump(base, open(os.path.join(td, n + ".json"), "w"))
        oc.add_model_config(td)
        names = [n for n in oc.list_models() if n.startswith("zz-net")]
        assert names == ["zz-net-1", "zz-net-2", "zz-net-10"], names
    # checkpoint with a 7x7 (+cls) grid into a tower that resamples to 16 latents: bicubic to 4x4; into 20 latents: + nearest
    from types import SimpleNamespace

    def model_with(n_lat):
        vis = SimpleNamespace(cfg=SimpleNamespace(image_size=224, patch_size=32, exp_args=SimpleNamespace(perceiver_num_latents=n_lat)),
                              use_perceiver=True)
        return SimpleNamespace(visual=vis)
    g = torch.Generator().manual_seed(0)
    pe = torch.randn(50, 8, generator=g)
    sd = {"visual.positional_embedding": pe.clone()}
    F.resize_pos_embed(sd, model_with(16))
    out = sd["visual.positional_embedding"]
    assert out.shape == (17, 8) and torch.equal(out[0], pe[0])
    ref = torch.nn.functional.interpolate(pe[1:].reshape(1, 7, 7, 8).permute(0, 3, 1, 2), size=(4, 4), mode="bicubic",
                                          antialias=True, align_corners=False).permute(0, 2, 3, 1).reshape(16, 8)
    assert torch.allclose(out[1:], ref)
    sd = {"visual.positional_embedding": pe.clone()}
    F.resize_pos_embed(sd, model_with(20))
    assert sd["visual.positional_embedding"].shape == (21, 8)
    sd = {"visual.positional_embedding": pe.clone()}
    F.resize_pos_embed(sd, model_with(49))
    assert torch.equal(sd["visual.positional_embedding"], pe)


def test_set_bn_sync_is_a_noop_without_ranks():
    import open_clip
    from types import SimpleNamespace
    m = open_clip.model.VisionTransformer.__new__(open_clip.model.VisionTransformer)
    m.modality, m._bn_sync = "pc", None
    assert m.set_bn_sync(True)._bn_sync is None                                   # torch.distributed not initialised
    comm = SimpleNamespace(all_gather=None, all_reduce_sum=None)
    assert m.set_bn_sync(True, comm=comm, world_size=2)._bn_sync == (comm, 2)
    assert m.set_bn_sync(False)._bn_sync is None
    with pytest.raises(ValueError):
        m.set_bn_sync(True, comm=comm)
    m.modality = "depth"
    assert m.set_bn_sync(True, comm=comm, world_size=2)._bn_sync is None


_REF_RESIZE = r'''
import json, sys, torch
from types import SimpleNamespace
sys.path.insert(0, sys.argv[1])
import ref_loader
oc = ref_loader.load()
from open_clip.model import resize_pos_embed
g = torch.Generator().manual_seed(0)
pe = torch.randn(50, 8, generator=g)
out = {}
for name, n_lat, grid in (("p16", 16, 7), ("p20", 20, 7), ("p49", 49, 7), ("p100", 100, 7), ("g14", None, 14), ("g7", None, 7)):
    vis = SimpleNamespace(grid_size=(grid, grid), use_perceiver=n_lat is not None,
                          vision_cfg=SimpleNamespace(exp_args=SimpleNamespace(perceiver_num_latents=n_lat)))
    sd = {"visual.positional_embedding": pe.clone()}
    resize_pos_embed(sd, SimpleNamespace(visual=vis))
    out[name] = sd["visual.positional_embedding"].tolist()
print("JSON" + json.dumps(out))
'''


@pytest.mark.needs_reference
def test_resize_pos_embed_equals_the_reference():
    """open_clip/model.py:1079-1150 (imported, build container only) against the product's factory.resize_pos_embed: grid
    -> latent counts that are / are not perfect squares (bicubic, then the nearest resample), equal lengths, plain grids."""
    from types import SimpleNamespace
    r = subprocess.run([sys.executable, "-c", _REF_RESIZE, os.path.join(ROOT, "oracle")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    import importlib
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        del sys.modules[k]
    F = importlib.import_module("open_clip.factory")
    g = torch.Generator().manual_seed(0)
    pe = torch.randn(50, 8, generator=g)
    for name, n_lat, grid in (("p16", 16, 7), ("p20", 20, 7), ("p49", 49, 7), ("p100", 100, 7), ("g14", None, 14), ("g7", None, 7)):
        vis = SimpleNamespace(cfg=SimpleNamespace(image_size=32 * grid, patch_size=32, exp_args=SimpleNamespace(perceiver_num_latents=n_lat)),
                              use_perceiver=n_lat is not None)
        sd = {"visual.positional_embedding": pe.clone()}
        F.resize_pos_embed(sd, SimpleNamespace(visual=vis))
        want = torch.tensor(ref[name])
        assert sd["visual.positional_embedding"].shape == want.shape, name
        assert torch.allclose(sd["visual.positional_embedding"], want, atol=1e-6), name


_REF_LOAD = r'''
import json, os, sys, tempfile, torch
sys.path.insert(0, sys.argv[1])
import ref_loader
oc = ref_loader.load()
import gen_golden as G
from open_clip.factory import load_checkpoint
work = sys.argv[2]
json.dump(G.TINY, open(os.path.join(work, "tiny-lens.json"), "w"))
oc.add_model_config(work)
args = G.tiny_args("depth")
torch.manual_seed(0)
a = oc.tri_create_model("tiny-lens", None, precision="fp32", device="cpu", output_dict=True, args=args)
# an open_clip-style training checkpoint: the image encoder is called `visual`, DDP prefix, wrapped in {"state_dict": ...}
sd = {k: v for k, v in a.state_dict().items() if not k.startswith(("image.", "visual."))}
sd.update({"visual." + k[len("image."):]: v for k, v in a.state_dict().items() if k.startswith("image.")})
torch.save({"epoch": 3, "state_dict": {"module." + k: v.clone() for k, v in sd.items()}}, os.path.join(work, "clip.pt"))
torch.manual_seed(1)
b = oc.tri_create_model("tiny-lens", None, precision="fp32", device="cpu", output_dict=True, args=args)
inc = load_checkpoint(b, os.path.join(work, "clip.pt"), strict=False, args=args)
torch.save({k: v.clone() for k, v in b.state_dict().items()}, os.path.join(work, "loaded.pt"))
print("JSON" + json.dumps({"missing": sorted(inc.missing_keys), "unexpected": sorted(inc.unexpected_keys),
                           "args": {k: v for k, v in args.items() if isinstance(v, (int, float, str, bool, type(None)))}}))
'''


@pytest.mark.needs_reference
def test_load_checkpoint_equals_the_reference(tmp_path):
    """factory.load_checkpoint (factory.py:119-160, imported, build container only) on an open_clip-style checkpoint
    (`visual.*` = the image encoder, `module.` prefix, {"state_dict": ...}): the same tensors end up in every parameter
    of the tri-modal model (`visual.* -> image.*` duplication AND the modality tower initialised from the ViT), the same
    keys are reported missing / unexpected."""
    from types import SimpleNamespace
    r = subprocess.run([sys.executable, "-c", _REF_LOAD, os.path.join(ROOT, "oracle"), str(tmp_path)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    want = torch.load(tmp_path / "loaded.pt")
    import importlib
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        del sys.modules[k]
    oc = importlib.import_module("open_clip")
    oc.add_model_config(str(tmp_path))
    args = SimpleNamespace(**ref["args"])
    torch.manual_seed(2)
    model = oc.tri_create_model("tiny-lens", None, device="cpu", output_dict=True, args=args)
    inc = oc.load_checkpoint(model, str(tmp_path / "clip.pt"), strict=False, args=args)
    assert sorted(inc.missing_keys) == ref["missing"], set(inc.missing_keys) ^ set(ref["missing"])
    assert sorted(inc.unexpected_keys) == ref["unexpected"], set(inc.unexpected_keys) ^ set(ref["unexpected"])
    got = model.state_dict()
    assert set(got) == set(want)
    loaded = [k for k in want if k not in ref["missing"]]
    assert len(loaded) > 50
    for k in loaded:
        assert torch.equal(got[k].cpu(), want[k]), k
    # and through tri_create_model(pretrained=path), the way the training mains call it
    torch.manual_seed(3)
    m2 = oc.tri_create_model("tiny-lens", str(tmp_path / "clip.pt"), device="cpu", output_dict=True, args=args)
    for k in loaded:
        assert torch.equal(m2.state_dict()[k].cpu(), want[k]), k


_REF_CFG = r'''
import json, sys
sys.path.insert(0, sys.argv[1])
import ref_loader
ref_loader.load()
from mm_vit_lens.model_cfg import fetch_model_cfg
out = {}
for m in ("image", "pc", "depth", "audio", "tactile", "eeg"):
    try:
        c = fetch_model_cfg(modality=m, model_option="vitlensL")
    except Exception as e:
        out[m] = {"__error__": repr(e)}
        continue
    out[m] = {k: v for k, v in dict(c).items() if isinstance(v, (int, float, str, bool, type(None)))}
print("JSON" + json.dumps(out))
'''


@pytest.mark.needs_reference
def test_model_cfg_values_equal_the_reference():
    """mm_vit_lens/model_cfg.py (imported, build container only): for every modality the configuration has exactly the
    reference's keys and values, plus the four attributes TriCLIP.forward / lock read and the reference's defaults omit."""
    r = subprocess.run([sys.executable, "-c", _REF_CFG, os.path.join(ROOT, "oracle")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    from mm_vit_lens.model_cfg import fetch_model_cfg
    for m, want in ref.items():
        assert "__error__" not in want, (m, want)
        mine = vars(fetch_model_cfg(modality=m))
        assert {k: mine.get(k, "<missing>") for k in want} == want, m
        assert set(mine) - set(want) == {"unlock_from_head", "vid_use_fpos", "vid_use_ltpos", "vid_distill_tokens"}, m


_SIG_TARGETS = [
    ("open_clip", "tokenize"), ("open_clip", "get_tokenizer"), ("open_clip", "tri_create_model"),
    ("open_clip", "tri_create_model_and_transforms"), ("open_clip", "tri_create_model_from_pretrained"), ("open_clip", "get_cast_dtype"),
    ("open_clip", "get_input_dtype"), ("open_clip", "create_loss"), ("open_clip", "add_model_config"),
    ("open_clip", "get_model_config"), ("open_clip", "image_transform"),
    ("open_clip.model", "TriCLIP.encode_image"), ("open_clip.model", "TriCLIP.encode_text"), ("open_clip.model", "TriCLIP.encode_visual"),
    ("open_clip.model", "TriCLIP.forward"), ("open_clip.model", "TriCLIP.lock_image_tower"), ("open_clip.model", "TriCLIP.lock_visual_tower"),
    ("open_clip.model", "TriCLIP.lock_text_tower"), ("open_clip.model", "TriCLIP.set_grad_checkpointing"),
    ("open_clip.loss", "ClipLoss.__init__"), ("open_clip.loss", "ClipLoss.forward"), ("open_clip.loss", "ClipLossGeneral.__init__"),
    ("open_clip.loss", "ClipLossGeneral.forward"), ("open_clip.loss", "TriClipLoss.__init__"), ("open_clip.loss", "TriClipLoss.forward"),
    ("open_clip.loss", "gather_features"), ("open_clip.factory", "load_checkpoint"), ("open_clip.factory", "resize_pos_embed"), ("open_clip.model", "resize_pos_embed"),
    ("open_clip.zero_shot_classifier", "build_zero_shot_classifier"), ("open_clip.zero_shot_classifier", "build_zero_shot_classifier_legacy"),
    ("mm_vit_lens.model_cfg", "fetch_model_cfg"),
    ("training.train", "tri_train_one_epoch"), ("training.train", "train_dual_one_epoch"), ("training.train", "backward"),
    ("training.scheduler", "cosine_lr"), ("training.scheduler", "const_lr"), ("training.scheduler", "const_lr_cooldown"),
    ("training.zero_shot", "run"),
]

_REF_SIG = r'''
import importlib, inspect, json, sys
sys.path.insert(0, sys.argv[1])
import ref_loader
ref_loader.load()
out = {}
for mod, name in json.loads(sys.argv[2]):
    obj = importlib.import_module(mod)
    for part in name.split("."):
        obj = getattr(obj, part)
    out[mod + ":" + name] = [[p.name, None if p.default is inspect.Parameter.empty else repr(p.default), str(p.kind)]
                             for p in inspect.signature(obj).parameters.values()]
print("JSON" + json.dumps(out))
'''


@pytest.mark.needs_reference
def test_public_signatures_equal_the_reference():
    """SURVEY 8(b) "callables to keep": parameter names, order and defaults of the drop-in surface against the imported
    reference (build container only).  The product may only APPEND optional parameters (e.g. `device`, `chunk_rows`)."""
    import importlib
    import inspect
    r = subprocess.run([sys.executable, "-c", _REF_SIG, os.path.join(ROOT, "oracle"), json.dumps(_SIG_TARGETS)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    for k in [k for k in sys.modules if k.split(".")[0] in ("open_clip", "mm_vit_lens", "training")]:
        del sys.modules[k]
    bad = []
    for mod, name in _SIG_TARGETS:
        obj = importlib.import_module(mod)
        for part in name.split("."):
            obj = getattr(obj, part)
        mine = [[p.name, None if p.default is inspect.Parameter.empty else repr(p.default), str(p.kind)]
                for p in inspect.signature(obj).parameters.values()]
        want = ref[mod + ":" + name]
        want_names = [w[0] for w in want if w[2] not in ("VAR_KEYWORD", "VAR_POSITIONAL")]
        mine_names = [m[0] for m in mine if m[2] not in ("VAR_KEYWORD", "VAR_POSITIONAL")]
        if mine_names[:len(want_names)] != want_names:
            bad.append((name, "names", mine_names, want_names))
            continue
        for m, w in zip(mine, want):
            if w[2] in ("VAR_KEYWORD", "VAR_POSITIONAL"):
                continue
            site = w[0] == "cache_dir" or (name.startswith("build_zero_shot_classifier") and w[0] == "device")
            if m[1] != w[1] and not site:          # cache_dir: a site path; the classifier's device defaults to the GPU (there is no CPU path)
                bad.append((name, m[0], m[1], w[1]))
        extra = [m for m in mine[len(want_names):] if m[2] not in ("VAR_KEYWORD", "VAR_POSITIONAL")]
        if any(m[1] is None for m in extra):
            bad.append((name, "appended parameter without default", extra))
    assert not bad, bad


_REF_CONST = r'''
import json, sys, torch
sys.path.insert(0, sys.argv[1])
import ref_loader
oc = ref_loader.load()
import open_clip.constants as K
out = {"const": {k: getattr(K, k) for k in dir(K) if k.isupper() and isinstance(getattr(K, k), (str, tuple, list))},
       "modality": vars(K.ModalityType),
       "cast": {p: str(oc.get_cast_dtype(p)) for p in ("fp32", "amp", "bf16", "fp16", "pure_bf16", "pure_fp16", "amp_bf16")},
       "input": {p: str(oc.get_input_dtype(p)) for p in ("fp32", "amp", "bf16", "fp16", "pure_bf16", "pure_fp16", "amp_bf16")}}
print("JSON" + json.dumps(out))
'''


@pytest.mark.needs_reference
def test_constants_and_dtype_helpers_equal_the_reference():
    r = subprocess.run([sys.executable, "-c", _REF_CONST, os.path.join(ROOT, "oracle")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    import importlib
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        del sys.modules[k]
    oc = importlib.import_module("open_clip")
    K = importlib.import_module("open_clip.constants")
    for k, v in ref["const"].items():
        got = getattr(K, k)
        assert (list(got) if isinstance(got, tuple) else got) == v, k
    assert vars(K.ModalityType) == ref["modality"]
    for p, v in ref["cast"].items():
        assert str(oc.get_cast_dtype(p)) == v, p
    for p, v in ref["input"].items():
        assert str(oc.get_input_dtype(p)) == v, p


def test_tri_create_model_from_pretrained(tmp_path):
    """factory.py:425-464: weights are required; the checkpoint's tensors arrive in the model; the evaluation transform comes
    along unless `return_transform=False`."""
    import importlib
    oc = importlib.import_module("open_clip")
    cfg = oc.get_model_config("ViT-B-32")
    cfg["vision_cfg"].update(width=64, layers=2, patch_size=32)
    cfg["text_cfg"].update(width=64, heads=2, layers=2)
    cfg["embed_dim"] = 32
    json.dump(cfg, open(tmp_path / "tiny-b32.json", "w"))
    oc.add_model_config(str(tmp_path))
    with pytest.raises(RuntimeError):
        oc.tri_create_model_from_pretrained("tiny-b32", None)
    torch.manual_seed(0)
    src = oc.tri_create_model("tiny-b32", None, device="cpu")
    torch.save({"state_dict": src.state_dict()}, tmp_path / "w.pt")
    torch.manual_seed(1)
    model, preprocess = oc.tri_create_model_from_pretrained("tiny-b32", str(tmp_path / "w.pt"))
    sd = src.state_dict()
    for k, v in sd.items():
        # load_checkpoint's rule (factory.py:141-151): whatever the file calls `visual.*` is ALSO the image encoder
        want = sd.get("visual." + k[len("image."):], v) if k.startswith("image.") else v
        assert torch.equal(model.state_dict()[k], want), k
    assert preprocess.image_size == 224 and not preprocess.is_train
    assert isinstance(oc.tri_create_model_from_pretrained("tiny-b32", str(tmp_path / "w.pt"), return_transform=False), type(src))


def test_precision_modes_say_what_they_run():
    """`precision` is accepted with the reference's spellings and every mode SAYS what it runs: "fp32" = true fp32 arithmetic
    for eval-mode inference (round 5) and bf16 operands on fp32 streams for training, with a warning that names the split;
    "amp" runs as amp_bf16 with a warning instead of silently computing something else (reference:
    open_clip/factory.py:260-295, training/precision.py:5-12)."""
    import warnings
    import open_clip as oc
    from mm_vit_lens.model_cfg import fetch_model_cfg
    args = fetch_model_cfg(modality="image")
    with pytest.warns(UserWarning, match="no fp32 backward"):
        m = oc.tri_create_model("ViT-B-32", precision="fp32", device="cpu", args=args)
    assert m.precision_requested == "fp32" and "residual stream fp32" in m.precision_effective and "bf16 x bf16" in m.precision_effective
    assert "true fp32 arithmetic" in m.precision_effective and m.image.arith_f32 and m.visual.arith_f32
    assert "text tower: fp16 x fp16" in m.precision_effective
    with pytest.warns(UserWarning, match="runs as amp_bf16"):
        m = oc.tri_create_model("ViT-B-32", precision="amp", device="cpu", args=args)
    assert "residual stream bf16" in m.precision_effective and not m.image.arith_f32
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m = oc.tri_create_model("ViT-B-32", precision="amp_bf16", device="cpu", args=args)       # what it runs: no warning
    assert m.precision_requested == "amp_bf16"
    with pytest.raises(NotImplementedError):
        oc.tri_create_model("ViT-B-32", precision="fp16", device="cpu", args=args)
