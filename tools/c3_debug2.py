import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import bench
from vitlens_hip import engine, ops, step as vstep
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
B, mb = 1024, 256
images = torch.randn(B, 3, 224, 224, generator=g).to(dev)
depths = torch.randn(B, 1, 224, 224, generator=g).to(dev)
texts = bench.synth_text(B, g).to(dev)
sd = bench.seeded_tri_weights()
fin = lambda t: bool(torch.isfinite(t.float()).all())
for rd in (torch.float32, torch.bfloat16):
    tr = vstep.TriModalDepthStep(sd, engine.TowerCfg(), engine.TextCfg(), dev, micro_batch=mb, unlock_first_n=4, frozen_res_dtype=rd)
    for i in range(B // mb):
        s = slice(i * mb, (i + 1) * mb)
        fi = tr.image.encode_image(images[s]); ft = tr.text.encode_text(texts[s]); fv = tr._trainer(i).forward(depths[s])
        print(rd, "mb", i, "image", fin(fi), "text", fin(ft), "visual", fin(fv), "mem GB", round(torch.cuda.memory_allocated() / 2**30, 1), flush=True)
        if not fin(fv):
            S = tr._trainer(i).tower.saved(mb, 257)
            first_bad = next((j for j, x in enumerate(S.X) if not fin(x)), None)
            print("   first non-finite residual snapshot", first_bad, "xpre", fin(S.xpre), "q0", fin(S.q[0]), "a0", fin(S.a[0]), "u0", fin(S.u[0]), flush=True)
    loss = tr.forward_backward(images, texts, depths)
    print(rd, "loss", float(loss), flush=True)
    del tr; torch.cuda.empty_cache()
