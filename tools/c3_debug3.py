import os, sys, math, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import bench
from vitlens_hip import engine, ops, step as vstep
from vitlens_hip.step import pair_loss_and_grads
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
B, mb = 1024, 256
images = torch.randn(B, 3, 224, 224, generator=g).to(dev)
depths = torch.randn(B, 1, 224, 224, generator=g).to(dev)
texts = bench.synth_text(B, g).to(dev)
sd = bench.seeded_tri_weights()
fin = lambda t: bool(torch.isfinite(t.float()).all())
self = vstep.TriModalDepthStep(sd, engine.TowerCfg(), engine.TextCfg(), dev, micro_batch=mb, unlock_first_n=4, frozen_res_dtype=torch.bfloat16)
nmb = B // mb
for i in range(nmb):
    self._trainer(i)
self._alloc_flat_grads()
self.flat_grad.zero_()
E = 768
fi = torch.empty(B, E, device=dev); ft = torch.empty(B, E, device=dev)
fv = torch.empty(B, E, device=dev); vraw = torch.empty(B, E, device=dev); vnorm = torch.empty(B, device=dev)
for i in range(nmb):
    s = slice(i * mb, (i + 1) * mb)
    ops.l2_normalize(self.image.encode_image(images[s]), out=fi[s])
    ops.l2_normalize(self.text.encode_text(texts[s]), out=ft[s])
    vraw[s] = self._trainer(i).forward(depths[s])
    print("mb", i, "fi", fin(fi[s]), "ft", fin(ft[s]), "vraw", fin(vraw[s]), flush=True)
ops.l2_normalize(vraw, out=fv, norms=vnorm)
print("fv", fin(fv), "vnorm", fin(vnorm), flush=True)
scale = float(self.logit_scale.exp())
kw = dict(local_loss=False, gather_with_grad=False, need_x=False)
l1, _, dv1, ds1 = pair_loss_and_grads(self.comm, 0, 1, fi, fv, fi, fv, scale, **kw)
l2, _, dv2, ds2 = pair_loss_and_grads(self.comm, 0, 1, ft, fv, ft, fv, scale, **kw)
print("l1", float(l1), "l2", float(l2), "dv1", fin(dv1), "dv2", fin(dv2), "ds", float(ds1), float(ds2), flush=True)
dvraw = ops.l2_normalize_bwd(fv, dv1 + dv2, vnorm)
print("dvraw", fin(dvraw), flush=True)
for i in range(nmb):
    self._trainer(i).backward(dvraw[i * mb:(i + 1) * mb].contiguous())
    bad = [k for k, v in self.grads.items() if not fin(v)]
    print("after backward mb", i, "non-finite grads:", bad[:6], flush=True)
    if bad and i == 0:
        t = self._trainer(0).tower; S = t.saved(mb, 257)
        print("   dx", fin(S.dx), "dxb", fin(S.dxb), "du", fin(S.du), "dh", fin(S.dh), "dO", fin(S.dO), "dqkv", fin(S.dqkv), "delta", fin(S.delta), flush=True)
