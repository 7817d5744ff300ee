#!/usr/bin/env python
"""ViT-L text tower forward per 1 024 captions (the C3 step's text batch) for the three operand modes of TextEngine."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import vitlens_oracle as O
from vitlens_hip import engine as E

g = torch.Generator().manual_seed(0)
sd = O.init_text(O.TextSpec(), g)
text = O.synth_text(1024, g).cuda()
for arith in ("f16", "bf16x2", "bf16"):
    eng = E.TextEngine(sd, E.TextCfg(), "cuda", res_dtype=torch.bfloat16, arith=arith)
    for _ in range(2):
        eng.encode_text(text)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.encode_text(text)
    e1.record(); torch.cuda.synchronize()
    print(f"arith={arith:7s}: {e0.elapsed_time(e1) / 5:.3f} ms per 1024 captions")
