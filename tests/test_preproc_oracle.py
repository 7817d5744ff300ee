"""CPU: pins oracle/preproc_oracle.py (N3 preprocessing restatement) against the libraries the reference's transforms
call - Pillow's 8-bit bicubic resampler (byte equality), torch's ToTensor / Normalize arithmetic (bit equality) and
torch.nn.functional.interpolate(bicubic, antialias on/off) for the depth path."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import preproc_oracle as po  # noqa: E402

Image = pytest.importorskip("PIL.Image")

SIZES = [(37, 53, 24, 31), (240, 320, 224, 298), (500, 375, 298, 224), (224, 224, 224, 224), (64, 48, 224, 168),
         (231, 517, 100, 224), (1000, 30, 400, 12), (300, 300, 299, 301)]


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_pil_bicubic_resize_is_byte_exact(h, w, oh, ow):
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if h == 37:
        img[:, ::2] = 255; img[:, 1::2] = 0                           # ringing: exercises the clip to [0, 255]
    want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
    got = po.pil_resize_bicubic_u8(img, ow, oh)
    assert got.dtype == np.uint8 and np.array_equal(got, want)


def test_pil_resize_with_box_is_byte_exact():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (180, 260, 3), dtype=np.uint8)
    box = (17, 9, 201, 150)                                            # RandomResizedCrop = crop, then resize
    want = np.asarray(Image.fromarray(img).crop(box).resize((224, 224), Image.BICUBIC))
    got = po.pil_resize_bicubic_u8(img[9:150, 17:201], 224, 224)
    assert np.array_equal(got, want)


def test_to_tensor_normalize_matches_torch_bitwise():
    from open_clip.constants import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
    u8 = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    t = torch.from_numpy(u8).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.as_tensor(OPENAI_DATASET_MEAN, dtype=torch.float32)[:, None, None]
    std = torch.as_tensor(OPENAI_DATASET_STD, dtype=torch.float32)[:, None, None]
    want = t.sub_(mean).div_(std).numpy()
    got = po.to_tensor_normalize(u8, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("h,w", [(240, 320), (530, 730), (224, 300), (100, 80)])
def test_image_eval_transform_matches_pil_pipeline(h, w):
    from open_clip.constants import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    nh, nw = po.resized_output_size(h, w, 224)
    pil = Image.fromarray(img).resize((nw, nh), Image.BICUBIC)
    top, left = po.center_crop_origin(nh, nw, 224)
    pil = pil.crop((left, top, left + 224, top + 224)).convert("RGB")
    t = torch.from_numpy(np.array(pil)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    want = t.sub_(torch.as_tensor(OPENAI_DATASET_MEAN)[:, None, None]).div_(torch.as_tensor(OPENAI_DATASET_STD)[:, None, None])
    got = po.image_eval_transform(img, 224, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD)
    assert got.shape == (3, 224, 224) and np.array_equal(got, want.numpy())


@pytest.mark.parametrize("antialias", [True, False])
@pytest.mark.parametrize("h,w,oh,ow", [(530, 730, 224, 308), (120, 90, 298, 224), (224, 224, 224, 224), (61, 47, 20, 33)])
def test_float_bicubic_matches_torch_interpolate(h, w, oh, ow, antialias):
    g = torch.Generator().manual_seed(h)
    x = torch.rand(h, w, generator=g)
    want = torch.nn.functional.interpolate(x[None, None], (oh, ow), mode="bicubic", align_corners=False, antialias=antialias)[0, 0]
    got = po.resize_bicubic_f32(x.numpy(), oh, ow, antialias)
    assert np.abs(got - want.numpy()).max() < 2e-6


def test_depth_eval_transform_matches_the_torch_pipeline():
    g = torch.Generator().manual_seed(3)
    d = torch.rand(427, 561, generator=g) * 90 - 2                     # values below min_depth and above max_depth
    x = d.clamp(min=0.01).clamp(max=75.0) / 75.0
    nh, nw = po.resized_output_size(427, 561, 224)
    r = torch.nn.functional.interpolate(x[None, None], (nh, nw), mode="bicubic", align_corners=False, antialias=True)[0, 0]
    top, left = po.center_crop_origin(nh, nw, 224)
    want = (r[top:top + 224, left:left + 224] - 0.0418) / 0.0295
    got = po.depth_eval_transform(d.numpy())
    assert got.shape == (1, 224, 224) and np.abs(got[0] - want.numpy()).max() < 1e-4


@pytest.mark.needs_reference
def test_oracle_on_the_reference_example_photographs():
    """BASELINE config C1's inputs: the four JPEGs of /root/reference/assets/example (build container only) through
    Pillow + torch (what `image_transform(224, is_train=False)` of the reference computes) against the restatement -
    real photographs with smooth gradients and saturated regions instead of noise; and the product's tables on the
    same geometries."""
    import glob
    from open_clip.constants import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
    from vitlens_hip import preproc
    files = sorted(glob.glob("/root/reference/assets/example/image_*.jpg"))
    if len(files) < 4:
        pytest.skip("reference example images not present")
    for f in files:
        im = Image.open(f).convert("RGB")
        w, h = im.size
        nh, nw = po.resized_output_size(h, w, 224)
        top, left = po.center_crop_origin(nh, nw, 224)
        r = im.resize((nw, nh), Image.BICUBIC).crop((left, top, left + 224, top + 224))
        t = torch.from_numpy(np.array(r)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        want = t.sub_(torch.as_tensor(OPENAI_DATASET_MEAN)[:, None, None]).div_(torch.as_tensor(OPENAI_DATASET_STD)[:, None, None])
        got = po.image_eval_transform(np.array(im), 224, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD)
        assert np.array_equal(got, want.numpy()), f
        for n_in, n_out in ((w, nw), (h, nh)):
            b, kk, ks = preproc.pil_bicubic_tables(n_in, n_out)
            ob, okk, oks = po.pil_coeffs(n_in, n_out)
            assert ks == oks and np.array_equal(b, ob) and np.array_equal(kk, okk)


def test_pil_resize_random_geometries_byte_exact():
    """60 random (in, out) geometries incl. 1-pixel axes, extreme aspect ratios, up- and down-scaling by up to 12x:
    the restatement AND the product's vectorised tables (through the oracle's integer passes) against Image.resize."""
    from vitlens_hip import preproc
    rng = np.random.default_rng(2024)
    for t in range(60):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        oh, ow = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        if t % 6 == 0:
            h, ow = int(rng.integers(200, 400)), int(rng.integers(1, 30))         # strong down-scale on one axis, up on the other
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(po.pil_resize_bicubic_u8(img, ow, oh), want), (h, w, oh, ow)
        for n_in, n_out in ((w, ow), (h, oh)):
            b, kk, ks = preproc.pil_bicubic_tables(n_in, n_out)
            ob, okk, oks = po.pil_coeffs(n_in, n_out)
            assert ks == oks and np.array_equal(b, ob) and np.array_equal(kk, okk), (n_in, n_out)


@pytest.mark.parametrize("antialias", [True, False])
def test_float_bicubic_random_geometries(antialias):
    """40 random geometries of the float path: tables of the product, applied with numpy, against F.interpolate."""
    from vitlens_hip import preproc
    rng = np.random.default_rng(7)
    for _ in range(40):
        h, w, oh, ow = (int(v) for v in rng.integers(2, 120, 4))
        x = torch.from_numpy(rng.random((h, w), dtype=np.float32))
        want = torch.nn.functional.interpolate(x[None, None], (oh, ow), mode="bicubic", align_corners=False, antialias=antialias)[0, 0].numpy()
        hb, hw, _ = preproc.aten_bicubic_tables(w, ow, antialias)
        vb, vw, _ = preproc.aten_bicubic_tables(h, oh, antialias)
        xs = x.numpy()
        tmp = np.stack([(xs[:, np.clip(hb[i, 0] + np.arange(hb[i, 1]), 0, w - 1)] * hw[i, :hb[i, 1]]).sum(1) for i in range(ow)], 1)
        got = np.stack([(tmp[np.clip(vb[i, 0] + np.arange(vb[i, 1]), 0, h - 1)] * vw[i, :vb[i, 1]][:, None]).sum(0) for i in range(oh)], 0)
        assert np.abs(got - want).max() < 5e-6, (h, w, oh, ow, antialias, np.abs(got - want).max())
