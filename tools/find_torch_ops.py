"""Which Python lines launch torch-native (aten) GPU kernels inside a fused C3 step - the step is meant to be the library's own
kernels; whatever aten launches is plumbing that may be worth folding away.  torch.profiler with stacks, one step at b = 512 as
2 x 256.  Not imported by the product."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
import bench
from vitlens_hip import engine, step as vstep

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 512
sd = bench.seeded_tri_weights()
images = torch.randn(B, 3, 224, 224, generator=g).to(dev)
depths = torch.randn(B, 1, 224, 224, generator=g).to(dev)
texts = bench.synth_text(B, g).to(dev)
tr = vstep.TriModalDepthStep(sd, engine.TowerCfg(), engine.TextCfg(), dev, micro_batch=256, unlock_first_n=4,
                             frozen_res_dtype=torch.bfloat16, train_res_dtype=torch.bfloat16)
for _ in range(2):
    tr.step(images, texts, depths)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(images, texts, depths)
    torch.cuda.synchronize()
sites = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    site = next((f for f in ev.stack if "vit-lens_amd" in f or "bench.py" in f), "?")
    key = (ev.name, site.strip()[-110:], str(ev.input_shapes)[:60])
    sites[key] += 1; dur[key] += ev.device_time_total
for key, n in sorted(sites.items(), key=lambda kv: -dur[kv[0]])[:30]:
    print(f"{dur[key]:9.0f} us  x{n:4d}  {key[0]:28s} {key[2]:60s} {key[1]}")
