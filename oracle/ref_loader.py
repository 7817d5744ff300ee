"""Import harness for the upstream reference (TEST INFRASTRUCTURE — build container only).

Puts /root/reference/vitlens/src on sys.path and registers in-memory shells for
third-party modules the image lacks (easydict, ftfy, torchvision, timm, termcolor).
None of the shells supplies hot-path arithmetic: `DropPath` is the
identity because the reference itself uses nn.Identity at drop_path == 0
(modal_3d/models/pointbert/point_encoder.py:104), `ftfy.fix_text` is the identity
(exact for ASCII captions only).

Nothing here is copied from the reference; nothing here travels to the GPU box as a
dependency of the product path (only `oracle/gen_golden.py` and tests that are skipped
when /root/reference is absent import this file).
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("VITLENS_REFERENCE", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "vitlens", "src")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "open_clip"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _AttrDict(dict):
    """Minimal attribute dictionary (behavioural stand-in for easydict.EasyDict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            return _AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(_AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, other=None, **kw):
        other = dict(other or {})
        other.update(kw)
        for k, v in other.items():
            self[k] = v


def install_shells():
    import torch.nn as nn

    try:
        import easydict  # noqa: F401
    except ImportError:
        _mod("easydict", EasyDict=_AttrDict)
    try:
        import ftfy  # noqa: F401
    except ImportError:
        _mod("ftfy", fix_text=lambda s: s)
    try:
        import termcolor  # noqa: F401
    except ImportError:
        _mod("termcolor", colored=lambda s, *a, **k: s)
    try:
        import torchvision  # noqa: F401
    except ImportError:

        class _Shell:
            def __init__(self, *a, **k):
                pass

        class _InterpolationMode:
            BICUBIC = "bicubic"
            BILINEAR = "bilinear"
            NEAREST = "nearest"

        names = ["Normalize", "Compose", "RandomResizedCrop", "Resize", "CenterCrop",
                 "ToTensor", "Lambda", "RandomCrop", "ColorJitter", "RandomHorizontalFlip"]
        tr = _mod("torchvision.transforms", InterpolationMode=_InterpolationMode,
                  **{n: type(n, (_Shell,), {}) for n in names})
        fn = _mod("torchvision.transforms.functional")
        tr.functional = fn
        misc = _mod("torchvision.ops.misc", FrozenBatchNorm2d=type("FrozenBatchNorm2d", (nn.Module,), {}))
        ops = _mod("torchvision.ops", misc=misc)
        _mod("torchvision", transforms=tr, ops=ops)
    try:
        import timm  # noqa: F401
    except ImportError:
        hub = _mod("timm.models.hub")
        layers = _mod("timm.models.layers", DropPath=nn.Identity,
                      trunc_normal_=nn.init.trunc_normal_)
        models = _mod("timm.models", hub=hub, layers=layers)
        _mod("timm", models=models)


def load():
    """Return the reference's `open_clip` package (imports it on first use)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_SRC)
    install_shells()
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    # a product-side package of the same name may already be imported: evict it
    for k in [k for k in sys.modules if k == "open_clip" or k.startswith("open_clip.")]:
        f = getattr(sys.modules[k], "__file__", "") or ""
        if not f.startswith(REF_SRC):
            del sys.modules[k]
    import open_clip  # noqa: E402

    assert open_clip.__file__.startswith(REF_SRC), open_clip.__file__
    return open_clip


def lens_args(modality: str, **overrides):
    """exp_args for a modality from the reference's own mm_vit_lens.model_cfg
    (+ the 4 attributes SURVEY §8c found missing from its default cfg)."""
    load()
    from mm_vit_lens.model_cfg import fetch_model_cfg

    cfg = fetch_model_cfg(modality=modality)
    for k, v in dict(unlock_from_head=False, vid_use_fpos=False, vid_use_ltpos=False,
                     vid_distill_tokens=False).items():
        if k not in cfg:
            cfg[k] = v
    cfg.update(overrides)
    return cfg


def load_pointnet_util():
    """The reference's modal_3d/models/pointnet/pointnet_util.py as a stand-alone module (the `pnsa` tokenizer).  It
    imports `dgl.geometry` and `torch_redstone` at module level; neither is installed.  They are provided as shells: with
    `dgl.geometry.farthest_point_sampler` missing, the reference's own `farthest_point_sample` takes its fallback
    (pointnet_util.py:83-98, a torch FPS whose start index comes from `torch.randint`), and `rst.Lambda` is the one-line
    module wrapper it is in torch_redstone.  Used by tests / gen_golden only."""
    import importlib.util
    import types
    import torch.nn as nn
    path = os.path.join(REF_SRC, "open_clip", "modal_3d", "models", "pointnet", "pointnet_util.py")
    if "dgl" not in sys.modules:
        dgl = types.ModuleType("dgl"); dgl.geometry = types.ModuleType("dgl.geometry")
        sys.modules["dgl"], sys.modules["dgl.geometry"] = dgl, dgl.geometry
    if "torch_redstone" not in sys.modules:
        class Lambda(nn.Module):
            def __init__(self, fn):
                super().__init__(); self.fn = fn

            def forward(self, *a, **k):
                return self.fn(*a, **k)
        sys.modules["torch_redstone"] = types.SimpleNamespace(Lambda=Lambda)
    sample = types.ModuleType("open_clip.util.Sample")
    sample.Sample = dict
    names = ("open_clip", "open_clip.util", "open_clip.util.Sample")
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules.update({"open_clip": types.ModuleType("open_clip"), "open_clip.util": types.ModuleType("open_clip.util"),
                        "open_clip.util.Sample": sample})
    try:
        spec = importlib.util.spec_from_file_location("ref_pointnet_util", path)
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod
