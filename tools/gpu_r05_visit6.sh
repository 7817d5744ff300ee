#!/bin/bash
# Round 5, last checks (no product change since the evidence): the ViT-L fp32-arithmetic test added afterwards, the drop-in module
# API against the fused step at b = 256 on the final tree, and what the fp32 inference path costs (ViT-L/14 image tower, 64 images).
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_f32.py -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep -E "fp32 arithmetic|passed|failed|^E  " | tee gpurun_out/r05v6_f32_tests.log
for v in "--via step" "--via api"; do
  echo "b = 256 $v: $(timeout 300 python bench.py --batch 256 --steps 8 --warmup 2 --no-cpu-baseline $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step', d['value'], 'triplets/s')")" | tee -a gpurun_out/r05v6_api_vs_step.log
done
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05v6_fp32_mode_rate.log
import sys, time
sys.path.insert(0, "vit-lens_amd"); sys.path.insert(0, "oracle")
import torch
import vitlens_oracle as O
from vitlens_hip import engine as E, f32 as F
g = torch.Generator().manual_seed(0)
sd = O.init_tower(O.TowerSpec(), g, "image.")
img = torch.randn(64, 3, 224, 224, generator=g).cuda()
for name, eng in (("fp32 arithmetic (vl_gemm_f32 / vl_attn_fwd_f32)", F.VitEngineF32(sd, "image.", E.TowerCfg(), "cuda")),
                  ("bf16 operands (the training / amp path)", E.VitEngine(sd, "image.", E.TowerCfg(), "cuda", res_dtype=torch.bfloat16))):
    fn = eng.encode if hasattr(eng, "encode") else eng.encode_image
    fn(img); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"ViT-L/14 image tower, 64 images, {name}: {dt * 1e3:.1f} ms = {64 / dt:.0f} img/s = {64 * 162.03e9 / dt / 1e12:.1f} TFLOP/s")
PY
