import sys; sys.path.insert(0,'/root/repo/vit-lens_amd')
import torch
from vitlens_hip import ops
def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
M,N,K=65536,4096,1024
a = rnd(M, K, seed=1).bfloat16().cuda(); w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().cuda(); bias = rnd(N, seed=3).cuda()
acc = a.float() @ w.float().t() + bias
bad=[]
for rep in range(8):
    out = ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=8).float()
    d=(out-acc).abs(); bad.append(int((d > acc.abs()*2.0**-6 + 2e-2).sum()))
print("plain cfg8 violations per run", bad)
