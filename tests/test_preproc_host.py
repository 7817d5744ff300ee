"""CPU: host logic of the on-GPU preprocessing (vitlens_hip/preproc.py, open_clip/transform.py) - the resampling tables
against the oracle's restatement of Pillow / ATen (which tests/test_preproc_oracle.py pins on the libraries themselves),
output-size / crop arithmetic, the factory's argument handling and the crop-box draw."""
import numpy as np
import pytest
import torch

import preproc_oracle as po
from vitlens_hip import preproc

PAIRS = [(730, 308), (530, 224), (320, 298), (224, 224), (48, 168), (517, 224), (1000, 400), (30, 12), (300, 301), (3, 224)]


@pytest.mark.parametrize("n_in,n_out", PAIRS)
def test_pil_tables_are_pillows_integers(n_in, n_out):
    b, kk, ks = preproc.pil_bicubic_tables(n_in, n_out)
    ob, okk, oks = po.pil_coeffs(n_in, n_out)
    assert ks == oks and b.dtype == np.int32 and kk.dtype == np.int32
    assert np.array_equal(b, ob) and np.array_equal(kk, okk)


def test_pil_tables_with_a_box():
    b, kk, ks = preproc.pil_bicubic_tables(184, 224, 0.0, 184.0)
    ob, okk, oks = po.pil_coeffs(184, 224, 0.0, 184.0)
    assert ks == oks and np.array_equal(b, ob) and np.array_equal(kk, okk)


@pytest.mark.parametrize("n_in,n_out", PAIRS)
def test_aten_antialias_tables(n_in, n_out):
    b, w, ks = preproc.aten_bicubic_tables(n_in, n_out, True)
    xm, xs, ow = po.aten_aa_weights(n_in, n_out)
    assert np.array_equal(b[:, 0], xm) and np.array_equal(b[:, 1], xs) and w.dtype == np.float32
    assert ks == ow.shape[1] and np.array_equal(w, ow)


def test_aten_plain_tables_reproduce_interpolate():
    x = torch.rand(1, 1, 1, 97, generator=torch.Generator().manual_seed(0))
    want = torch.nn.functional.interpolate(x, (1, 41), mode="bicubic", align_corners=False)[0, 0, 0].numpy()
    b, w, ks = preproc.aten_bicubic_tables(97, 41, False)
    assert ks == 4 and (b[:, 1] == 4).all()
    idx = np.clip(b[:, :1] + np.arange(4)[None, :], 0, 96)
    got = (x[0, 0, 0].numpy()[idx] * w).sum(1)
    assert np.abs(got - want).max() < 1e-6


def test_size_and_crop_arithmetic():
    for h, w in [(480, 640), (640, 480), (224, 224), (225, 1000), (531, 730), (100, 80)]:
        assert preproc.resized_output_size(h, w, 224) == po.resized_output_size(h, w, 224)
        nh, nw = preproc.resized_output_size(h, w, 224)
        assert min(nh, nw) == 224
        assert preproc.center_crop_origin(nh, nw, 224) == po.center_crop_origin(nh, nw, 224)
    assert preproc.center_crop_origin(224, 229, 224) == (0, 2)                    # round-half-even of 2.5, as Python's round


def test_factory_arguments_and_crop_box():
    from open_clip.constants import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
    from open_clip.transform import AugmentationCfg, image_transform, random_resized_crop_params
    t = image_transform((224, 224), is_train=False, device="cpu")
    assert t.image_size == 224 and t.mean == OPENAI_DATASET_MEAN and t.std == OPENAI_DATASET_STD and not t.is_train
    t = image_transform(224, is_train=True, mean=0.5, std=0.25, aug_cfg={"scale": (0.4, 1.0)}, device="cpu")
    assert t.is_train and t.mean == (0.5,) * 3 and t.std == (0.25,) * 3 and t.scale == (0.4, 1.0)
    assert image_transform(224, True, aug_cfg=AugmentationCfg(), device="cpu").scale == (0.9, 1.0)
    with pytest.raises(NotImplementedError):
        image_transform(224, False, resize_longest_max=True)
    torch.manual_seed(0)
    a = [random_resized_crop_params(300, 400, (0.08, 1.0)) for _ in range(50)]
    torch.manual_seed(0)
    assert a == [random_resized_crop_params(300, 400, (0.08, 1.0)) for _ in range(50)]
    for top, left, h, w in a:
        assert 0 <= top and 0 <= left and top + h <= 300 and left + w <= 400 and h > 0 and w > 0
    assert random_resized_crop_params(10, 1000, (0.9, 1.0)) == (0, 493, 10, 13)   # no admissible draw: central fallback


def test_device_table_cache_is_bounded():
    preproc._DEVICE_TABLES.clear()
    for i in range(preproc._DEVICE_TABLES_MAX + 40):
        preproc._on_device(("pil", 300 + i, 224), lambda i=i: preproc.pil_bicubic_tables(300 + i, 224), "cpu")
    assert len(preproc._DEVICE_TABLES) == preproc._DEVICE_TABLES_MAX
    assert (("pil", 300, 224), "cpu") not in preproc._DEVICE_TABLES               # least recently used went first
    hit = preproc._on_device(("pil", 340, 224), lambda: (_ for _ in ()).throw(AssertionError("must be cached")), "cpu")
    assert hit[2] == preproc.pil_bicubic_tables(340, 224)[2]
    preproc._DEVICE_TABLES.clear()
