"""Throughput of the on-GPU preprocessing next to the reference's CPU pipeline on the same inputs.

GPU side: bytes already resident in HBM (the image transform's two kernels per image), and, separately, with the
host->device copy of each decoded image inside the timed region.  CPU side ("port" of what the reference's data-loader
workers run): Pillow bicubic resize + crop + torch ToTensor/Normalize for images, F.interpolate for disparity maps,
one thread.  Prints one JSON object; `python tools/preproc_bench.py > gpurun_out/preproc_bench.json`."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))

from open_clip.constants import OPENAI_DATASET_MEAN as MEAN, OPENAI_DATASET_STD as STD  # noqa: E402
from open_clip.modal_depth.processors.vt_processor import DepthProcessorEval  # noqa: E402
from open_clip.transform import image_transform  # noqa: E402
from vitlens_hip import preproc  # noqa: E402


def gpu_time(fn, n_warm=1):
    for _ in range(n_warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


def main():
    N, H, W = 256, 375, 500                                                       # ImageNet-typical decoded size
    rng = np.random.default_rng(0)
    host = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(N)]
    dev = [torch.from_numpy(h).cuda() for h in host]
    t = image_transform(224, is_train=False)
    out = torch.empty(N, 3, 224, 224, device="cuda")
    res = {"image": {"n": N, "in": [H, W, 3], "out": [3, 224, 224]}}
    s = gpu_time(lambda: [preproc.image_to_tensor(d, 224, MEAN, STD, out=out[i]) for i, d in enumerate(dev)])
    res["image"]["gpu_resident_img_per_s"] = round(N / s, 1)
    res["image"]["gpu_resident_us_per_img"] = round(s / N * 1e6, 2)
    res["image"]["bytes_per_img"] = H * W * 3 + 3 * 224 * 224 * 4                  # decoded bytes read once + float32 written once
    t0 = time.perf_counter()
    for i, h in enumerate(host):
        t(h, out=out[i])
    torch.cuda.synchronize(); s2 = time.perf_counter() - t0
    res["image"]["gpu_incl_h2d_img_per_s"] = round(N / s2, 1)
    # batched: a list of images in two launches.  "kernels" = the two launches alone (descriptors prepared), the roofline
    # figure; "resident" adds the host-side planning; "incl_h2d" adds the copies of the decoded bytes
    bpi = res["image"]["bytes_per_img"]
    plan = preproc.plan_images(dev, 224, MEAN, STD, out=out)
    s = gpu_time(plan.launch, n_warm=2)
    res["image"]["batched_kernels_img_per_s"] = round(N / s, 1)
    res["image"]["batched_kernels_gb_per_s"] = round(N * bpi / s / 1e9, 1)
    res["image"]["batched_kernels_us"] = round(s * 1e6, 1)
    s = gpu_time(lambda: preproc.images_to_tensor(dev, 224, MEAN, STD, out=out))
    res["image"]["batched_resident_img_per_s"] = round(N / s, 1)
    hosts_t = [torch.from_numpy(h) for h in host]
    preproc.images_to_tensor(hosts_t, 224, MEAN, STD, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter(); preproc.images_to_tensor(hosts_t, 224, MEAN, STD, out=out); torch.cuda.synchronize()
    res["image"]["batched_incl_h2d_img_per_s"] = round(N / (time.perf_counter() - t0), 1)
    sizes = [(375, 500), (500, 375), (480, 640), (333, 500), (600, 800), (256, 256), (427, 640), (768, 1024)]
    var = [torch.from_numpy(rng.integers(0, 256, (sizes[i % 8][0], sizes[i % 8][1], 3), dtype=np.uint8)).cuda() for i in range(N)]
    plan = preproc.plan_images(var, 224, MEAN, STD, out=out)
    s = gpu_time(plan.launch, n_warm=2)
    res["image"]["batched_mixed_sizes_kernels_img_per_s"] = round(N / s, 1)
    res["image"]["batched_mixed_sizes_kernels_gb_per_s"] = round((sum(v.numel() for v in var) + N * 3 * 224 * 224 * 4) / s / 1e9, 1)
    try:
        from PIL import Image
        torch.set_num_threads(1)
        n_cpu = 64
        pil = [Image.fromarray(h) for h in host[:n_cpu]]
        m, sd = torch.as_tensor(MEAN)[:, None, None], torch.as_tensor(STD)[:, None, None]
        t0 = time.perf_counter()
        for im in pil:
            nh, nw = preproc.resized_output_size(H, W, 224)
            top, left = preproc.center_crop_origin(nh, nw, 224)
            r = im.resize((nw, nh), Image.BICUBIC).crop((left, top, left + 224, top + 224)).convert("RGB")
            torch.from_numpy(np.array(r)).permute(2, 0, 1).contiguous().to(torch.float32).div(255).sub_(m).div_(sd)
        res["image"]["cpu_1thread_img_per_s"] = round(n_cpu / (time.perf_counter() - t0), 1)
    except ImportError:
        pass
    # disparity maps (SUN RGB-D typical 530 x 730)
    Nd, Hd, Wd = 128, 530, 730
    d_host = [torch.rand(Hd, Wd) * 80 for _ in range(Nd)]
    d_dev = [d.cuda() for d in d_host]
    proc = DepthProcessorEval()
    outd = torch.empty(Nd, 1, 224, 224, device="cuda")
    s = gpu_time(lambda: [proc(d, out=outd[i]) for i, d in enumerate(d_dev)])
    res["depth"] = {"n": Nd, "in": [Hd, Wd], "gpu_resident_maps_per_s": round(Nd / s, 1), "gpu_resident_us_per_map": round(s / Nd * 1e6, 2)}
    n_cpu = 32
    t0 = time.perf_counter()
    for d in d_host[:n_cpu]:
        x = d.clamp(min=0.01).clamp(max=75.0) / 75
        nh, nw = preproc.resized_output_size(Hd, Wd, 224)
        r = torch.nn.functional.interpolate(x[None, None], (nh, nw), mode="bicubic", align_corners=False, antialias=True)[0, 0]
        top, left = preproc.center_crop_origin(nh, nw, 224)
        (r[top:top + 224, left:left + 224] - 0.0418) / 0.0295
    res["depth"]["cpu_1thread_maps_per_s"] = round(n_cpu / (time.perf_counter() - t0), 1)
    # point clouds: 10000 -> 8192 FPS + unit sphere (the 3D recipe's loader)
    from open_clip.modal_3d.processors.pc_processor import PCProcessorEval
    B = 32
    pcs = torch.randn(B, 10000, 3).cuda()
    pp = PCProcessorEval(npoint=8192, uniform=True)
    start = np.zeros(B, dtype=np.int64)
    s = gpu_time(lambda: pp.process_batch(pcs, start=start))
    res["pc"] = {"n": B, "in": [10000, 3], "npoint": 8192, "gpu_resident_clouds_per_s": round(B / s, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
