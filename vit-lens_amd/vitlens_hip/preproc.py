"""Host side of the on-GPU image / depth preprocessing (kernels in csrc/vl_preproc.hip).

The resamplers are table driven: per output index of a resized axis a first tap, a tap count and the coefficients.
The tables depend only on (input size, output size), so they are built once on the host - exactly where Pillow and
ATen build theirs - cached, and kept on the device.

  * `pil_bicubic_tables`  = Pillow `Resample.c:precompute_coeffs` + `normalize_coeffs_8bpc` (double precision, Keys
    kernel a = -0.5, support scaled when shrinking, 22-bit fixed point rounded half away from zero);
  * `aten_bicubic_tables` = ATen `UpSampleKernel.cpp` weights: antialiased (`_compute_indices_min_size_weights_aa`,
    float32, a = -0.5, normalised) or plain (`get_cubic_upsample_coefficients`, a = -0.75, 4 taps from floor(src) - 1,
    border-clamped in the kernel).

`image_to_tensor` / `depth_to_tensor` run Resize(shorter edge) -> CenterCrop -> Normalize restricted to the crop
window (torchvision `_compute_resized_output_size`, `center_crop`); see open_clip/transform.py and
open_clip/modal_depth/processors/vt_processor.py for the reference-facing classes."""
import collections
import ctypes
import functools
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


def resized_output_size(h, w, size):
    """Resize(int): shorter edge -> size, longer = int(size * long / short) (torchvision functional.resize)."""
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    return (new_long, size) if w <= h else (size, new_long)


def center_crop_origin(h, w, size):
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def _keys(x, a):
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@functools.lru_cache(maxsize=256)
def pil_bicubic_tables(in_size, out_size, in0=0.0, in1=None):
    """-> (bounds [out,2] int32, kk [out,ksize] int32, ksize); operation order as in Resample.c so that every double
    is the one Pillow computes (the row sum is a left-to-right running sum)."""
    in1 = float(in_size) if in1 is None else float(in1)
    scale = (in1 - in0) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = in0 + (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)               # C cast: truncation (values >= -1.5 -> ok)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    j = np.arange(ksize, dtype=np.int64)[None, :]
    w = _keys(((j + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss, -0.5)
    w = np.where(j < xmax[:, None], w, 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]                                             # sequential, as the C loop
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    fx = w * float(1 << PRECISION_BITS)
    kk = np.where(w < 0, (-0.5 + fx).astype(np.int64), (0.5 + fx).astype(np.int64)).astype(np.int32)   # (int) truncates
    kk = np.where(j < xmax[:, None], kk, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, kk, ksize


@functools.lru_cache(maxsize=256)
def aten_bicubic_tables(in_size, out_size, antialias=True):
    """-> (bounds [out,2] int32, weights [out,ksize] float32, ksize) in float32 arithmetic, as ATen for a float input."""
    f = np.float32
    scale = f(in_size) / f(out_size)
    i = np.arange(out_size, dtype=np.float32)
    if antialias:
        support = f(2.0) * scale if scale >= 1.0 else f(2.0)
        invscale = f(1.0) / scale if scale >= 1.0 else f(1.0)
        ksize = int(math.ceil(float(support))) * 2 + 1
        center = scale * (i + f(0.5))
        xmin = np.maximum((center - support + f(0.5)).astype(np.int64), 0)
        xsize = np.minimum((center + support + f(0.5)).astype(np.int64), in_size) - xmin
        j = np.arange(ksize, dtype=np.int64)[None, :]
        arg = (((j + xmin[:, None]).astype(np.float32) - center[:, None] + f(0.5)) * invscale).astype(np.float32)
        w = _keys(arg.astype(np.float64), -0.5).astype(np.float32)
        w = np.where(j < xsize[:, None], w, f(0.0)).astype(np.float32)
        tot = np.cumsum(w, axis=1, dtype=np.float32)[:, -1:]
        w = np.where(tot != 0, w / np.where(tot != 0, tot, f(1.0)), w).astype(np.float32)
        return np.stack([xmin, xsize], 1).astype(np.int32), np.ascontiguousarray(w), ksize
    A = -0.75
    # ATen's AVX2 build contracts scale * (i + 0.5) - 0.5 into one fused multiply-subtract: a single rounding to float32
    src = (np.float64(scale) * (np.arange(out_size, dtype=np.float64) + 0.5) - 0.5).astype(np.float32).astype(np.float64)
    i0 = np.floor(src)
    t = src - i0
    c1 = lambda v: ((A + 2) * v - (A + 3)) * v * v + 1
    c2 = lambda v: ((A * v - 5 * A) * v + 8 * A) * v - 4 * A
    w = np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], 1).astype(np.float32)
    bounds = np.stack([i0.astype(np.int64) - 1, np.full(out_size, 4, np.int64)], 1).astype(np.int32)
    return bounds, np.ascontiguousarray(w), 4


_DEVICE_TABLES = collections.OrderedDict()      # (kind, in, out[, antialias], device) -> tables; LRU, a data set has few distinct sizes
_DEVICE_TABLES_MAX = 512                        # <= ~10 KB each on the device


def _on_device(key, builder, device):
    k = (key, str(device))
    hit = _DEVICE_TABLES.get(k)
    if hit is not None:
        _DEVICE_TABLES.move_to_end(k)
        return hit
    bounds, coef, ksize = builder()
    hit = (torch.from_numpy(bounds).to(device), torch.from_numpy(coef).to(device), ksize, bounds)
    _DEVICE_TABLES[k] = hit
    if len(_DEVICE_TABLES) > _DEVICE_TABLES_MAX:
        _DEVICE_TABLES.popitem(last=False)      # tensors still referenced by kernels in flight stay alive through the allocator's stream ordering
    return hit


def _window(bounds_host, first, count, limit):
    """Source rows/cols touched by outputs [first, first+count): (lo, n) clamped to [0, limit)."""
    b = bounds_host[first:first + count]
    lo = max(int(b[:, 0].min()), 0)
    hi = min(int((b[:, 0] + b[:, 1]).max()), limit)
    return lo, hi - lo


def _floats(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def image_to_tensor(img_u8, size, mean, std, out=None, box=None, want_u8=False):
    """img_u8: uint8 [H, W, C] CUDA tensor (C = 1..4).  Resize(size, BICUBIC) -> CenterCrop(size) -> ToTensor ->
    Normalize(mean, std) -> float32 [C, size, size] (written into `out` if given).
    box = (top, left, height, width) with size = (out_h, out_w): RandomResizedCrop's crop-then-resize to exactly that
    size (the caller draws the box).  want_u8: also return the resized uint8 crop [size, size, C]."""
    from . import ops
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.is_cuda
    dev = img_u8.device
    if box is not None:
        top, left, bh, bw = box
        img_u8 = img_u8[top:top + bh, left:left + bw]
        oh, ow = size if isinstance(size, (tuple, list)) else (size, size)
        nh, nw, ctop, cleft, ch, cw = oh, ow, 0, 0, oh, ow
    else:
        H, W = img_u8.shape[:2]
        nh, nw = resized_output_size(H, W, size)
        ctop, cleft = center_crop_origin(nh, nw, size)
        ch = cw = size
        if ctop < 0 or cleft < 0:
            raise ValueError("image smaller than the crop after Resize: padding is not part of the evaluation transform")
    H, W, C = img_u8.shape
    if img_u8.stride(2) != 1 or img_u8.stride(1) != C:
        img_u8 = img_u8.contiguous()
    hb, hk, hks, hb_host = _on_device(("pil", W, nw), lambda: pil_bicubic_tables(W, nw), dev)
    vb, vk, vks, vb_host = _on_device(("pil", H, nh), lambda: pil_bicubic_tables(H, nh), dev)
    row0, nrows = _window(vb_host, ctop, ch, H)
    tmp = torch.empty(nrows, cw, C, device=dev, dtype=torch.uint8)
    ops.check(ops._lib.vl_resample_h_u8(ops._p(img_u8), img_u8.stride(0), W, C, row0, nrows, ops._p(hb), ops._p(hk), hks, cleft, cw,
                                        ops._p(tmp), ops._stream()))
    if out is None:
        out = torch.empty(C, ch, cw, device=dev, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (C, ch, cw)
    u8 = torch.empty(ch, cw, C, device=dev, dtype=torch.uint8) if want_u8 else None
    ops.check(ops._lib.vl_resample_v_u8_norm(ops._p(tmp), cw, C, row0, nrows, ops._p(vb), ops._p(vk), vks, ctop, ch, _floats(mean), _floats(std),
                                             ops._p(out), ops._p(u8), ops._stream()))
    return (out, u8) if want_u8 else out


def plane_to_tensor(plane, size, mean, std, clamp=None, antialias=True, out=None, crop=None):
    """plane: float32 [H, W] CUDA tensor.  clamp = (lo, hi, divide_by) is applied as the source is read (DepthNorm;
    (0, 255, 255) turns the float-cast bytes of an image channel into ToTensor's [0, 1] range).  Resize(size, bicubic)
    -> CenterCrop(crop or size) -> (x - mean) / std -> float32 [crop, crop] written to `out` (any [.., crop, crop] view)."""
    from . import ops
    assert plane.dtype == torch.float32 and plane.dim() == 2 and plane.is_cuda
    dev = plane.device
    if plane.stride(1) != 1:
        plane = plane.contiguous()
    crop = size if crop is None else crop
    H, W = plane.shape
    nh, nw = resized_output_size(H, W, size)
    ctop, cleft = center_crop_origin(nh, nw, crop)
    if ctop < 0 or cleft < 0:
        raise ValueError("input smaller than the crop after Resize")
    hb, hw, hks, _ = _on_device(("aten", W, nw, antialias), lambda: aten_bicubic_tables(W, nw, antialias), dev)
    vb, vw, vks, vb_host = _on_device(("aten", H, nh, antialias), lambda: aten_bicubic_tables(H, nh, antialias), dev)
    row0, nrows = _window(vb_host, ctop, crop, H)
    tmp = torch.empty(nrows, crop, device=dev, dtype=torch.float32)
    lo, hi, div = clamp if clamp is not None else (0.0, 0.0, 1.0)
    ops.check(ops._lib.vl_resample_h_f32(ops._p(plane), plane.stride(0), W, row0, nrows, ops._p(hb), ops._p(hw), hks, cleft, crop,
                                         int(clamp is not None), float(lo), float(hi), float(div), ops._p(tmp), ops._stream()))
    if out is None:
        out = torch.empty(crop, crop, device=dev, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == crop * crop
    ops.check(ops._lib.vl_resample_v_f32_norm(ops._p(tmp), crop, H, row0, ops._p(vb), ops._p(vw), vks, ctop, crop, float(mean),
                                              float(std), ops._p(out), ops._stream()))
    return out


def depth_to_tensor(depth, size, mean, std, clamp=None, antialias=True, out=None):
    """depth: float32 [H, W] CUDA tensor.  clamp = (lo, hi, divide_by) = DepthNorm; Resize(size, bicubic) ->
    CenterCrop(size) -> (x - mean) / std -> float32 [1, size, size]."""
    if out is None:
        out = torch.empty(1, size, size, device=depth.device, dtype=torch.float32)
    plane_to_tensor(depth, size, mean, std, clamp=clamp, antialias=antialias, out=out)
    return out


def float_image_to_tensor(img_u8, size, crop, mean, std, antialias=True, out=None):
    """ToTensor FIRST, then Resize on the float tensor (the tactile recipe, modal_tactile/processors/tact_processor.py:
    286-295): img_u8 uint8 [H, W, C] CUDA -> float32 [C, crop, crop]; each channel through the float resampler with the
    /255 of ToTensor fused into the read."""
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.is_cuda
    C = img_u8.shape[2]
    planes = img_u8.permute(2, 0, 1).to(torch.float32).contiguous()               # dtype cast only; the arithmetic is in the kernels
    if out is None:
        out = torch.empty(C, crop, crop, device=img_u8.device, dtype=torch.float32)
    for c in range(C):
        plane_to_tensor(planes[c], size, mean[c], std[c], clamp=(0.0, 255.0, 255.0), antialias=antialias, out=out[c], crop=crop)
    return out


class BatchPlan:
    """Device-side description of one batched transform: descriptors, packed tables, intermediate, and the tensors that
    keep the source bytes alive.  `launch()` enqueues the two kernels; it can be repeated (same inputs, same output)."""

    def __init__(self, desc, tables, tmp, out, n, C, ch, cw, max_nrows, mean, std, keep):
        self.desc, self.tables, self.tmp, self.out, self.keep = desc, tables, tmp, out, keep
        self.n, self.C, self.ch, self.cw, self.max_nrows, self.mean, self.std = n, C, ch, cw, max_nrows, mean, std

    def launch(self):
        from . import ops
        ops.check(ops._lib.vl_resample_batch_u8_norm(ops._p(self.desc), self.n, self.C, self.ch, self.cw, self.max_nrows,
                                                     ops._p(self.tables), ops._p(self.tmp), _floats(self.mean), _floats(self.std),
                                                     ops._p(self.out), ops._stream()))
        return self.out


def plan_images(images, size, mean, std, out=None, boxes=None):
    """Host side of `images_to_tensor`: copies host images to the device (one buffer, one slice per image - small
    pageable copies are the fast path of the HIP runtime, a single 100 MB pageable copy is not), builds the packed
    per-axis tables of the distinct sizes and one int64 descriptor row per image."""
    n = len(images)
    assert n > 0
    dev = out.device if out is not None else next((im.device for im in images if im.is_cuda), torch.device("cuda"))
    ch, cw = (size, size) if isinstance(size, int) else tuple(size)
    C = images[0].shape[2]
    host_idx = [i for i, im in enumerate(images) if not im.is_cuda]
    host_off, total = {}, 0
    for i in host_idx:
        host_off[i] = total
        total += images[i].numel()
    packed = torch.empty(total, device=dev, dtype=torch.uint8) if host_idx else None
    for i in host_idx:
        packed[host_off[i]:host_off[i] + images[i].numel()].copy_(images[i].reshape(-1), non_blocking=True)
    parts, offsets, cursor = [], {}, 0

    def table(in_size, out_size):
        nonlocal cursor
        key = (in_size, out_size)
        if key not in offsets:
            b, kk, ks = pil_bicubic_tables(in_size, out_size)
            offsets[key] = (cursor, cursor + b.size, ks, b)
            parts.extend((b.reshape(-1), kk.reshape(-1)))
            cursor += b.size + kk.size
        return offsets[key]
    desc = np.zeros((n, 16), np.int64)
    tmp_bytes, max_nrows, keep = 0, 0, [packed]
    for i, im in enumerate(images):
        assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == C
        H, W = im.shape[:2]
        if im.is_cuda:
            if im.stride(2) != 1 or im.stride(1) != C:
                im = im.contiguous()
            keep.append(im)
            base, stride = im.data_ptr(), im.stride(0)
        else:
            base, stride = packed.data_ptr() + host_off[i], W * C
        if boxes is not None:
            top, left, H, W = boxes[i]
            base += top * stride + left * C
            nh, nw, ctop, cleft = ch, cw, 0, 0
        else:
            nh, nw = resized_output_size(H, W, ch)
            ctop, cleft = center_crop_origin(nh, nw, ch)
            if ctop < 0 or cleft < 0:
                raise ValueError("image smaller than the crop after Resize")
        hb_off, hk_off, hks, _ = table(W, nw)
        vb_off, vk_off, vks, vb_host = table(H, nh)
        row0, nrows = _window(vb_host, ctop, ch, H)
        desc[i, :14] = (base, stride, W, row0, nrows, cleft, ctop, hb_off, hk_off, hks, vb_off, vk_off, vks, tmp_bytes)
        tmp_bytes += nrows * cw * C
        max_nrows = max(max_nrows, nrows)
    tables = torch.from_numpy(np.concatenate(parts)).to(dev, non_blocking=True)
    desc_d = torch.from_numpy(desc).to(dev, non_blocking=True)
    tmp = torch.empty(tmp_bytes, device=dev, dtype=torch.uint8)
    if out is None:
        out = torch.empty(n, C, ch, cw, device=dev, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (n, C, ch, cw)
    return BatchPlan(desc_d, tables, tmp, out, n, C, ch, cw, max_nrows, mean, std, keep)


def images_to_tensor(images, size, mean, std, out=None, boxes=None):
    """A LIST of uint8 [H, W, C] images of different sizes (CPU or CUDA tensors, same C) -> float32 [n, C, size, size]
    in two kernel launches (vl_resample_batch_u8_norm): same arithmetic and bits as `image_to_tensor` per image.
    boxes[i] = (top, left, h, w): crop-then-resize (training)."""
    return plan_images(images, size, mean, std, out=out, boxes=boxes).launch()
