"""CPU, build container only: the oracle against the IMPORTED reference at FULL size (ViT-L/14 towers, the audio
Lens with its 2 x (cross + 3 self) Perceiver).  The 1.7 GB state_dict cannot be a fixture, so the comparison runs
in-process: the reference model is built with a fixed seed, its own state_dict is handed to the oracle, and the
features of both must agree to fp32 round-off."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import json, sys, torch
sys.path.insert(0, sys.argv[1])
import ref_loader
oc = ref_loader.load()
import vitlens_oracle as O
torch.manual_seed(3)
args = ref_loader.lens_args("audio")
model = oc.tri_create_model("ViT-L-14", None, precision="fp32", device="cpu", output_dict=True, args=args).eval()
sd = {k: v.detach() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(4)
image = torch.randn(2, 3, 224, 224, generator=g)
audio = torch.randn(2, 512, 128, generator=g) * 0.5
text = oc.tokenize(["a dog barking", "rain on a tin roof"])
with torch.no_grad():
    ri = model.encode_image(image); rt = model.encode_text(text); rv = model.encode_visual(audio)
    oi = O.encode_image(sd, image, O.TowerSpec())
    ot = O.encode_text(sd, text, O.TextSpec())
    lens = O.LensSpec(modality="audio", perceiver_identity=False, depth=2, self_per_cross=3)
    ov = O.encode_visual(sd, audio, O.TowerSpec(), lens)
rel = lambda a, b: float((a - b).norm() / b.norm())
out = {"image": rel(oi, ri), "text": rel(ot, rt), "audio": rel(ov, rv), "n_params": sum(v.numel() for v in sd.values())}
del model, sd
# point-cloud Lens (PointBERT tokenizer: FPS 512 centres of 8192 points, kNN 32, mini-PointNet; Perceiver depth 4)
torch.manual_seed(5)
model = oc.tri_create_model("ViT-L-14", None, precision="fp32", device="cpu", output_dict=True, args=ref_loader.lens_args("pc")).eval()
sd = {k: v.detach() for k, v in model.state_dict().items()}
pts = torch.rand(2, 8192, 3, generator=g) * 2 - 1
torch.manual_seed(11); start = torch.randint(0, 8192, (2,), dtype=torch.long)     # what misc.fps draws (misc.py:60)
torch.manual_seed(11)
with torch.no_grad():
    rv = model.encode_visual(pts)
    lens = O.LensSpec(modality="pc", perceiver_identity=False, depth=4, self_per_cross=1, input_chan=384)
    ov = O.encode_visual(sd, pts, O.TowerSpec(), lens, fps_start=start)
out["pc"] = rel(ov, rv)
print("JSON" + json.dumps(out))
'''


@pytest.mark.needs_reference
def test_oracle_equals_reference_at_vitl_size():
    r = subprocess.run([sys.executable, "-c", _SCRIPT, os.path.join(ROOT, "oracle")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    assert res["n_params"] > 8e8
    for k in ("image", "text", "audio", "pc"):
        assert res[k] < 2e-5, res


_FLOOR = r'''
import json, sys, torch
sys.path.insert(0, sys.argv[1])
import ref_loader
oc = ref_loader.load()
import vitlens_oracle as O
spec = O.TextSpec()
g = torch.Generator().manual_seed(77)            # the weights / texts of tests/test_hip_towers.py::test_vitl_text_tower_vs_oracle
sd = O.init_text(spec, g)
text = O.synth_text(4, g)
torch.manual_seed(0)
model = oc.tri_create_model("ViT-L-14", None, precision="fp32", device="cpu", output_dict=True, args=ref_loader.lens_args("depth")).eval()
model.load_state_dict(sd, strict=False)
cosm = lambda a: (lambda n: n @ n.t())(torch.nn.functional.normalize(a.float(), dim=-1))
with torch.no_grad():
    f32 = model.encode_text(text)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        fbf = model.encode_text(text)
print("JSON" + json.dumps({"cos_matrix": float((cosm(fbf) - cosm(f32)).abs().max()),
                           "feature_cos": float(torch.nn.functional.cosine_similarity(fbf.float(), f32, dim=-1).min()),
                           "mutual": float(cosm(f32)[0, 1])}))
'''


@pytest.mark.needs_reference
def test_reference_amp_bf16_floor_of_the_text_tower():
    """north_star asks for cosine matrices within 1e-3 of the fp32 CPU path.  On the random-init ViT-L text tower (features
    with a mutual cosine of ~0.6) the REFERENCE ITSELF under its amp_bf16 autocast is 1.4e-3 away from its own fp32 path;
    the HIP path (bf16 operands, fp32 residual stream) is held to 2e-3 on the same case (tests/test_hip_towers.py) and to
    1e-3 on the per-feature cosine.  This test pins that floor so the relaxed bound is a measured property of bf16
    arithmetic on this input, not a loosened criterion."""
    r = subprocess.run([sys.executable, "-c", _FLOOR, os.path.join(ROOT, "oracle")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout[r.stdout.index("JSON") + 4:])
    assert 1.0e-3 < res["cos_matrix"] < 2.0e-3, res
    assert res["feature_cos"] > 0.9999 and 0.5 < res["mutual"] < 0.7, res
