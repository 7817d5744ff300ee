import sys; sys.path.insert(0,'/root/repo/vit-lens_amd')
import torch, collections
from vitlens_hip import ops
def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
M,N,K=65536,4096,1024
a = rnd(M, K, seed=1).bfloat16().cuda(); w = rnd(N, K, seed=2, scale=K ** -0.5).bfloat16().cuda(); bias = rnd(N, seed=3).cuda()
acc = a.float() @ w.float().t() + bias
for rep in range(3):
    out = ops.gemm(a, w, bias, epi=ops.EPI_BF16, cfg=8).float()
    d=(out-acc).abs(); viol = d > acc.abs()*2.0**-6 + 2e-2
    idx = viol.nonzero()
    print("violations", len(idx))
    if len(idx)==0: continue
    r,c = idx[:,0], idx[:,1]
    isbias = (out[r,c] - bias[c].bfloat16().float()).abs() < 1e-6
    print("  equal to bias alone:", int(isbias.sum()), "of", len(idx))
    tiles = collections.Counter(zip((r//256).tolist(), (c//256).tolist()))
    print("  tiles", list(tiles.items())[:12])
    inrow = collections.Counter((r%256).tolist()); incol = collections.Counter((c%256).tolist())
    print("  row%256 values", sorted(inrow.items())[:40])
    print("  col%256 values", sorted(incol.items())[:40])
    # k-range check: is the wrong value a partial sum (some k-steps missing)?
    rr, cc = int(r[0]), int(c[0])
    part = (a[rr].float()[None,:] * w[cc].float()[None,:]).reshape(16,64).sum(1).cumsum(0)
    print("  first bad", rr, cc, "out", float(out[rr,cc]), "want", float(acc[rr,cc]), "bias", float(bias[cc]), "partial sums + bias", [round(float(x+bias[cc]),4) for x in part])
