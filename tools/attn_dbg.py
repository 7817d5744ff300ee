import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
import torch
from vitlens_hip import ops
torch.manual_seed(0)
B, H, L, dh = 1, 1, 64, 64
q = torch.randn(B, H, L, dh).bfloat16().cuda(); k = torch.randn(B, H, L, dh).bfloat16().cuda()
v = torch.arange(L * dh).float().reshape(1, 1, L, dh) % 97
v = v.bfloat16().cuda()
out = torch.zeros(B * L, H * dh, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, L, device="cuda")
ops.attn_fwd(q, k, v, out, lse=lse, qscale=1.0)
s = (q.float().cpu() / ops.LOG2E) @ k.float().cpu().transpose(-1, -2)
ref = (torch.softmax(s, -1) @ v.float().cpu())[0, 0]
o = out.float().cpu()
err = (o - ref).abs()
print("lse err", float((lse.cpu()[0, 0] - torch.logsumexp(s, -1)[0, 0]).abs().max()))
print("err by d-block of 4 (rows 0..3):")
for r in range(3):
    print([round(float(err[r, d:d + 4].max()), 2) for d in range(0, 64, 4)])
print("row0 got ", [round(float(x), 1) for x in o[0, :16]])
print("row0 ref ", [round(float(x), 1) for x in ref[0, :16]])
print("row33 got", [round(float(x), 1) for x in o[33, :16]])
print("row33 ref", [round(float(x), 1) for x in ref[33, :16]])
