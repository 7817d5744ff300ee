"""GPU: on-GPU image / depth preprocessing (csrc/vl_preproc.hip through open_clip/transform.py and
open_clip/modal_depth/processors) against the oracle - byte-exact resampling and bit-exact float32 output for the
8-bit image path (and against Pillow + torch directly where Pillow is installed), tolerance 2e-4 on the normalised
disparity for the float path (values span +-34; the tap sums are fp32 in a different association than ATen's)."""
import numpy as np
import pytest
import torch

import preproc_oracle as po

pytestmark = pytest.mark.gpu

from open_clip.constants import OPENAI_DATASET_MEAN as MEAN, OPENAI_DATASET_STD as STD  # noqa: E402


def _img(h, w, c=3, seed=0):
    rng = np.random.default_rng(seed + h * 7 + w)
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    img[: h // 3, ::2] = 255                                                      # hard edges: ringing beyond [0, 255]
    img[: h // 3, 1::2] = 0
    return img


@pytest.mark.parametrize("h,w", [(240, 320), (530, 730), (730, 530), (224, 224), (224, 300), (100, 80), (1080, 1920), (225, 224)])
def test_image_eval_transform_bit_exact(h, w):
    from open_clip.transform import image_transform
    from vitlens_hip import preproc
    img = _img(h, w)
    want = po.image_eval_transform(img, 224, MEAN, STD)
    got = image_transform(224, is_train=False)(img)
    assert got.is_cuda and got.shape == (3, 224, 224) and got.dtype == torch.float32
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    nh, nw = po.resized_output_size(h, w, 224)
    top, left = po.center_crop_origin(nh, nw, 224)
    _, u8 = preproc.image_to_tensor(torch.from_numpy(img).cuda(), 224, MEAN, STD, want_u8=True)
    assert np.array_equal(u8.cpu().numpy(), po.pil_resize_bicubic_u8(img, nw, nh)[top:top + 224, left:left + 224])


def test_image_transform_against_pillow_and_torch_directly():
    Image = pytest.importorskip("PIL.Image")
    from open_clip.transform import image_transform
    t = image_transform(224, is_train=False)
    imgs = [Image.fromarray(_img(375, 500)), Image.fromarray(_img(333, 250, 1)[..., 0]), Image.fromarray(_img(64, 48))]
    got = t.batch(imgs)
    assert got.shape == (3, 3, 224, 224)
    for i, im in enumerate(imgs):
        nh, nw = po.resized_output_size(im.size[1], im.size[0], 224)
        r = im.resize((nw, nh), Image.BICUBIC)
        top, left = po.center_crop_origin(nh, nw, 224)
        r = r.crop((left, top, left + 224, top + 224)).convert("RGB")
        x = torch.from_numpy(np.array(r)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        want = x.sub_(torch.as_tensor(MEAN)[:, None, None]).div_(torch.as_tensor(STD)[:, None, None])
        assert torch.equal(got[i].cpu(), want), i


def test_random_resized_crop_is_crop_then_resize():
    from open_clip.transform import image_transform, random_resized_crop_params
    img = _img(300, 420, seed=3)
    t = image_transform(224, is_train=True, aug_cfg={"scale": (0.3, 1.0)})
    torch.manual_seed(11)
    box = random_resized_crop_params(300, 420, (0.3, 1.0))
    torch.manual_seed(11)
    got = t(img)
    top, left, bh, bw = box
    want = po.to_tensor_normalize(po.pil_resize_bicubic_u8(img[top:top + bh, left:left + bw], 224, 224), MEAN, STD)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("antialias", [True, False])
@pytest.mark.parametrize("h,w", [(427, 561), (530, 730), (224, 224), (120, 90), (760, 1280)])
def test_depth_processor_vs_oracle(h, w, antialias):
    from open_clip.modal_depth.processors.vt_processor import DepthProcessorEval
    g = torch.Generator().manual_seed(h + w)
    d = torch.rand(h, w, generator=g) * 90 - 2                                    # below min_depth and above max_depth
    want = po.depth_eval_transform(d.numpy(), antialias=antialias)
    proc = DepthProcessorEval(antialias=antialias)
    got = proc(d.unsqueeze(0))
    assert got.is_cuda and got.shape == (1, 224, 224)
    assert np.abs(got.cpu().numpy() - want).max() < 2e-4
    x = d.clamp(min=0.01).clamp(max=75.0) / 75
    nh, nw = po.resized_output_size(h, w, 224)
    r = torch.nn.functional.interpolate(x[None, None], (nh, nw), mode="bicubic", align_corners=False, antialias=antialias)[0, 0]
    top, left = po.center_crop_origin(nh, nw, 224)
    ref = (r[top:top + 224, left:left + 224] - 0.0418) / 0.0295                    # the torch pipeline itself
    assert (got[0].cpu() - ref).abs().max() < 2e-4
    assert torch.equal(proc.batch([d, d.numpy()])[1], got)
