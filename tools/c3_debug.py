import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import bench
from vitlens_hip import engine, ops, step as vstep
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
B = int(os.environ.get('DBG_B', 256))
images = torch.randn(B, 3, 224, 224, generator=g).to(dev)
depths = torch.randn(B, 1, 224, 224, generator=g).to(dev)
texts = bench.synth_text(B, g).to(dev)
sd = bench.seeded_tri_weights()
combos = (("f32 old-epi", torch.float32, 1), ("f32 lds-epi", torch.float32, 0), ("bf16 old-epi", torch.bfloat16, 1), ("bf16 lds-epi", torch.bfloat16, 0))
if os.environ.get("DBG_STEPS"):
    combos = (("bf16 lds-epi", torch.bfloat16, 0), ("bf16 old-epi", torch.bfloat16, 1))
for name, rd, wide in combos:
    ops.set_wide_stores(wide)
    tr = vstep.TriModalDepthStep(sd, engine.TowerCfg(), engine.TextCfg(), dev, micro_batch=256, unlock_first_n=4, frozen_res_dtype=rd)
    for it in range(int(os.environ.get("DBG_STEPS", 0))):
        l = tr.step(images, texts, depths)
        print("   step", it, float(l), "logit_scale", float(tr.logit_scale), "finite masters",
              all(bool(torch.isfinite(v).all()) for v in tr.masters.values()), flush=True)
    loss = tr.forward_backward(images, texts, depths)
    gn = {k: float(v.float().norm()) for k, v in tr.grads.items()}
    bad = [k for k, v in gn.items() if v != v or v == float("inf")]
    print(name, "loss", float(loss), "nonfinite grads", bad[:4], "gnorm logit_scale", gn["logit_scale"], "fc0", gn.get("visual.transformer.resblocks.0.mlp.c_fc.weight"), flush=True)
    fi = tr.image.encode_image(images[:8]); ft = tr.text.encode_text(texts[:8])
    print("   image feat finite", bool(torch.isfinite(fi).all()), "text feat finite", bool(torch.isfinite(ft).all()), flush=True)
    del tr
    torch.cuda.empty_cache()
ops.set_wide_stores(0)
