"""Text tower (12 x 768, 77 tokens, 256 captions) by GEMM shape: plain bf16 weights (bf16 / fp32 residual stream) against the
two-term weights of TextEngine(wsplit=True)."""
import sys, torch, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in ("vit-lens_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, q))
import vitlens_oracle as O
from vitlens_hip import engine as E, ops
g = torch.Generator().manual_seed(1)
sd = O.init_text(O.TextSpec(), g)
text = O.synth_text(256, g).cuda()
import bench
for ws, rd in ((False, torch.bfloat16), (False, torch.float32), (True, torch.float32)):
    eng = E.TextEngine(sd, E.TextCfg(), "cuda", res_dtype=rd, wsplit=ws)
    timer = bench.GemmTimer(ops); timer.install(E)
    for _ in range(3): eng.encode_text(text)
    torch.cuda.synchronize(); timer.on = True; t0 = time.perf_counter()
    for _ in range(10): eng.encode_text(text)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3; timer.on = False
    print(f"wsplit={ws} res={rd}: {dt:.3f} ms per 256 texts")
    for sh in sorted(timer.summary(), key=lambda s: -s["avg_ms"] * s["launches"]):
        print(f"    {sh['kind']:8s} M={sh['M']} N={sh['N']} K={sh['K']} epi={sh['epi']} act={sh['act']}: {sh['avg_ms']*1e3:7.1f} us x {sh['launches']//10}  {sh['tflops']:7.1f} TF/s")
    ops.gemm = timer._gemm
