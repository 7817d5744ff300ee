"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy) of the evaluation-time modality preprocessing.

Only tests/ may import this file; the product path (vit-lens_amd/open_clip/transform.py,
modal_depth/processors) never does.

Image (vitlens/src/open_clip/transform.py:138-155): Resize(size, BICUBIC) -> CenterCrop(size) -> convert("RGB")
-> ToTensor -> Normalize.  On a PIL image torchvision's Resize is `Image.resize((w, h), BICUBIC)`, i.e. Pillow's
two-pass 8-bit resampler (third-party, absent from /root/reference; Pillow 12.2.0 is installed in this image):
`src/libImaging/Resample.c` precompute_coeffs / normalize_coeffs_8bpc / ImagingResampleHorizontal_8bpc /
ImagingResampleVertical_8bpc.  The restatement below is PINNED against Pillow itself (tests/test_preproc_oracle.py
runs `Image.resize` on random images and requires byte equality), and the ToTensor / Normalize arithmetic against
the torch ops torchvision calls (`.to(float32).div(255)`, `.sub_(mean).div_(std)`).

Depth (vitlens/src/open_clip/modal_depth/processors/vt_processor.py:292-337, transforms_rgbd.py:366-411):
DepthNorm(clamp to [min_depth, max_depth], / max_depth) -> Resize(224, bicubic) -> CenterCrop(224) -> Normalize.
torchvision is not installed here, so `depth_eval_reference` restates its tensor path with the torch function it
calls (torchvision/transforms/_functional_tensor.py:resize -> torch.nn.functional.interpolate(mode="bicubic",
align_corners=False, antialias=...)); `resize_bicubic_f32` is the numpy restatement of that ATen kernel
(aten/src/ATen/native/cpu/UpSampleKernel.cpp: separable weights, a = -0.5 with antialias, a = -0.75 without) and is
pinned against torch.nn.functional.interpolate in the same test file.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2                      # Resample.c: coefficients are 22-bit fixed point


def _bicubic_filter(x, a=-0.5):
    """Resample.c:bicubic_filter (Keys kernel, a = -0.5), double precision."""
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size, out_size, in0=0.0, in1=None, support=2.0):
    """Resample.c:precompute_coeffs + normalize_coeffs_8bpc.  Returns (bounds [out,2] int32, kk [out,ksize] int32, ksize)."""
    in1 = float(in_size) if in1 is None else in1
    scale = (in1 - in0) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - sup + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_bicubic_u8(img, out_w, out_h, box=None):
    """Image.resize((out_w, out_h), BICUBIC[, box]) for an [H, W, C] uint8 array: horizontal pass into an 8-bit
    intermediate (rounded + clipped), then the vertical pass (Resample.c:ImagingResampleInner)."""
    H, W, C = img.shape
    x0, y0, x1, y1 = (0.0, 0.0, float(W), float(H)) if box is None else [float(v) for v in box]
    need_h = out_w != W or x0 != 0 or x1 != W
    need_v = out_h != H or y0 != 0 or y1 != H
    cur = img
    if need_h:
        b, kk, _ = pil_coeffs(W, out_w, x0, x1)
        tmp = np.zeros((H, out_w, C), np.uint8)
        src = cur.astype(np.int64)
        for xx in range(out_w):
            xmin, n = int(b[xx, 0]), int(b[xx, 1])
            acc = (src[:, xmin:xmin + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    if need_v:
        b, kk, _ = pil_coeffs(H, out_h, y0, y1)
        out = np.zeros((out_h, cur.shape[1], C), np.uint8)
        src = cur.astype(np.int64)
        for yy in range(out_h):
            ymin, n = int(b[yy, 0]), int(b[yy, 1])
            acc = (src[ymin:ymin + n] * kk[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur


def resized_output_size(h, w, size):
    """torchvision.transforms.functional._compute_resized_output_size for an int size (shorter edge -> size)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)          # (new_h, new_w)


def center_crop_origin(h, w, size):
    """torchvision.transforms.functional.center_crop: top/left of the crop window (Python round = half to even)."""
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def to_tensor_normalize(u8_hwc, mean, std):
    """ToTensor (.to(float32).div(255), CHW) then Normalize (.sub_(mean).div_(std)), float32 arithmetic."""
    x = u8_hwc.astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32)[None, None, :]
    s = np.asarray(std, np.float32)[None, None, :]
    return np.ascontiguousarray(((x - m) / s).transpose(2, 0, 1))


def image_eval_transform(u8_hwc, size, mean, std):
    """transform.py:138-155 (is_train=False, resize_longest_max=False) on an RGB uint8 [H, W, 3] array."""
    H, W, _ = u8_hwc.shape
    nh, nw = resized_output_size(H, W, size)
    r = u8_hwc if (nh, nw) == (H, W) else pil_resize_bicubic_u8(u8_hwc, nw, nh)
    top, left = center_crop_origin(nh, nw, size)
    return to_tensor_normalize(r[top:top + size, left:left + size], mean, std)


# ----------------------------------------------------------------------------------------------- depth (float path)
def _cubic_aa(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def aten_aa_weights(in_size, out_size):
    """UpSampleKernel.cpp:HelperInterpBase::_compute_indices_min_size_weights_aa (bicubic, align_corners=False):
    float32 arithmetic as ATen does for a float input.  Returns (xmin [out], xsize [out], w [out, max] float32)."""
    f = np.float32
    scale = f(in_size) / f(out_size)
    support = f(2.0) * scale if scale >= 1.0 else f(2.0)
    invscale = f(1.0) / scale if scale >= 1.0 else f(1.0)
    maxs = int(math.ceil(float(support))) * 2 + 1
    xmin = np.zeros(out_size, np.int64)
    xsize = np.zeros(out_size, np.int64)
    w = np.zeros((out_size, maxs), np.float32)
    for i in range(out_size):
        center = scale * (f(i) + f(0.5))
        lo = max(int(np.int64(center - support + f(0.5))), 0)
        hi = min(int(np.int64(center + support + f(0.5))), in_size)
        n = hi - lo
        tot = f(0.0)
        for j in range(n):
            v = f(_cubic_aa(float(f(f(j + lo) - center + f(0.5)) * invscale)))
            w[i, j] = v
            tot = f(tot + v)
        if tot != 0:
            w[i, :n] = w[i, :n] / tot
        xmin[i], xsize[i] = lo, n
    return xmin, xsize, w


def resize_bicubic_f32(x, out_h, out_w, antialias=True):
    """torch.nn.functional.interpolate(x[None, None], (out_h, out_w), mode="bicubic", align_corners=False,
    antialias=antialias) for a float32 [H, W] array (horizontal pass, then vertical, as the separable ATen kernel)."""
    H, W = x.shape
    x = x.astype(np.float32)
    if antialias:
        xm, xs, wx = aten_aa_weights(W, out_w)
        tmp = np.zeros((H, out_w), np.float32)
        for i in range(out_w):
            tmp[:, i] = (x[:, xm[i]:xm[i] + xs[i]] * wx[i, :xs[i]][None, :]).sum(1, dtype=np.float32)
        ym, ys, wy = aten_aa_weights(H, out_h)
        out = np.zeros((out_h, out_w), np.float32)
        for i in range(out_h):
            out[i] = (tmp[ym[i]:ym[i] + ys[i]] * wy[i, :ys[i]][:, None]).sum(0, dtype=np.float32)
        return out
    A = -0.75                                                                    # UpSample.h:get_cubic_upsample_coefficients

    def coeffs(t):
        c1 = lambda v: ((A + 2) * v - (A + 3)) * v * v + 1
        c2 = lambda v: ((A * v - 5 * A) * v + 8 * A) * v - 4 * A
        return [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]

    def axis(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        idx, wts = [], []
        for i in range(n_out):
            # one rounding: the AVX2 build of ATen contracts scale * (i + 0.5) - 0.5 into a fused multiply-subtract
            src = float(np.float32(float(scale) * (i + 0.5) - 0.5))
            i0 = math.floor(src)
            idx.append([min(max(i0 + k, 0), n_in - 1) for k in (-1, 0, 1, 2)])
            wts.append(coeffs(src - i0))
        return np.asarray(idx), np.asarray(wts, np.float32)
    iy, wy = axis(H, out_h)
    ix, wx = axis(W, out_w)
    out = np.zeros((out_h, out_w), np.float32)
    for a in range(4):
        for b in range(4):
            out += (wy[:, a][:, None] * wx[:, b][None, :]) * x[iy[:, a]][:, ix[:, b]]
    return out


def depth_eval_transform(depth, size=224, depth_mean=0.0418, depth_std=0.0295, max_depth=75.0, min_depth=0.01,
                         clamp_max_before_scale=True, antialias=True):
    """DepthProcessorEval.__call__ (vt_processor.py:292-337) on a float [H, W] disparity map -> [1, size, size]."""
    d = np.maximum(depth.astype(np.float32), np.float32(min_depth))
    if clamp_max_before_scale:
        d = np.minimum(d, np.float32(max_depth))
    d = d / np.float32(max_depth)
    H, W = d.shape
    nh, nw = resized_output_size(H, W, size)
    r = resize_bicubic_f32(d, nh, nw, antialias)
    top, left = center_crop_origin(nh, nw, size)
    r = r[top:top + size, left:left + size]
    return ((r - np.float32(depth_mean)) / np.float32(depth_std))[None]
