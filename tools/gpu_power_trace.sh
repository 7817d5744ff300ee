#!/bin/bash
# Round 4: board power and shader clock sampled (rocm-smi, 4 Hz) while the C3 bench runs, then while the attention probe and the
# LayerNorm kernel run alone: what "power-limited" means in watts and MHz on this box.
mkdir -p gpurun_out; L=gpurun_out/power_trace.log; : > $L
sample() {   # $1 = label, $2 = pid to follow
  while kill -0 $2 2>/dev/null; do
    p=$(/opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | python -c "
import sys, json
try:
    d = json.load(sys.stdin); c = d[sorted(d)[0]]
    pw = [v for k, v in c.items() if 'ower' in k and 'W' in k]
    pick = lambda pre: ([v for k, v in c.items() if k.startswith(pre) and 'speed' in k] or [v for k, v in c.items() if k.startswith(pre)])
    sc = pick('sclk')
    mc = pick('mclk')
    print(pw[0] if pw else '?', sc[0] if sc else '?', mc[0] if mc else '?')
except Exception as e:
    print('?', '?', '?')
")
    echo "$1 $(date +%s.%N | cut -c1-14) $p" >> $L
    sleep 0.2
  done
}
python bench.py --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/power_bench.json 2>/dev/null &
sample c3_step $!
N=400 python tools/attn_probe.py > /dev/null 2>&1 &
sample attention $!
python - <<'PY' &
import sys, os
sys.path.insert(0, "vit-lens_amd")
import torch
from vitlens_hip import ops
M, D = 65792, 1024
x = torch.randn(M, D, device="cuda").bfloat16(); h = torch.empty_like(x)
g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
a = torch.randn(65536, 1024, device="cuda").bfloat16(); w = (torch.randn(4096, 1024, device="cuda") / 32).bfloat16(); o = torch.empty(65536, 4096, device="cuda", dtype=torch.bfloat16)
import time
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(200): ops.layernorm(x, g, b, h, M, D)
    torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(50): ops.gemm(a, w, None, out=o)
    torch.cuda.synchronize()
PY
sample layernorm_then_gemm $!
tail -1 gpurun_out/power_bench.json | cut -c1-200
python - <<'PY'
import collections, re
rows = collections.defaultdict(list)
for l in open("gpurun_out/power_trace.log"):
    f = l.split()
    if len(f) < 4 or f[2] == '?': continue
    num = lambda s: float(re.sub(r"[^0-9.]", "", s) or 0)
    rows[f[0]].append((float(f[1]), num(f[2]), num(f[3])))
for k, v in rows.items():
    pw = [p for _, p, _ in v]; sc = [s for _, _, s in v]
    print(f"{k:24s} samples {len(v):3d}  power W min/mean/max {min(pw):6.0f} {sum(pw)/len(pw):6.0f} {max(pw):6.0f}   sclk MHz min/mean/max {min(sc):5.0f} {sum(sc)/len(sc):5.0f} {max(sc):5.0f}")
PY
