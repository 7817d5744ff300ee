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


def test_tactile_processor_vs_the_torch_pipeline():
    """TactileRGBProcessorEval (tact_processor.py:281-300): ToTensor -> Resize(256, bicubic) on the float tensor ->
    CenterCrop(224) -> Normalize, against the same steps in torch (F.interpolate = torchvision's tensor resize)."""
    from open_clip.modal_tactile.processors.tact_processor import TactileRGBProcessorEval
    for (h, w), antialias in (((480, 640), True), ((320, 240), True), ((480, 640), False)):
        img = _img(h, w, seed=5)
        got = TactileRGBProcessorEval(antialias=antialias)(img)
        assert got.is_cuda and got.shape == (3, 224, 224)
        x = torch.from_numpy(img).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        nh, nw = po.resized_output_size(h, w, 256)
        r = torch.nn.functional.interpolate(x[None], (nh, nw), mode="bicubic", align_corners=False, antialias=antialias)[0]
        top, left = po.center_crop_origin(nh, nw, 224)
        r = r[:, top:top + 224, left:left + 224]
        want = (r - torch.as_tensor(MEAN)[:, None, None]) / torch.as_tensor(STD)[:, None, None]
        assert (got.cpu() - want).abs().max() < 1e-4, (h, w, antialias)
        for c in range(3):                                                        # and the numpy restatement, channel by channel
            o = po.resize_bicubic_f32(x[c].numpy(), nh, nw, antialias)[top:top + 224, left:left + 224]
            assert np.abs(got[c].cpu().numpy() - (o - np.float32(MEAN[c])) / np.float32(STD[c])).max() < 1e-4


def test_data_processors_from_files(tmp_path):
    """mm_vit_lens.data_processors with the reference's inputs - lists of file paths - against the oracle on the decoded
    arrays: PNG images (bit-exact), torch-saved disparity maps, .npy point clouds (FPS start drawn by np.random as in the
    reference), a tactile PNG, torch-saved EEG."""
    Image = pytest.importorskip("PIL.Image")
    import vitlens_oracle as O
    from mm_vit_lens import data_processors as DP
    imgs = [_img(240, 320, seed=1), _img(400, 300, seed=2)]
    paths = []
    for i, im in enumerate(imgs):
        paths.append(str(tmp_path / f"im{i}.png")); Image.fromarray(im).save(paths[-1])
    got = DP.ImageProcessor()(paths, device="cuda")
    assert got.shape == (2, 3, 224, 224) and got.is_cuda
    for i, im in enumerate(imgs):
        assert np.array_equal(got[i].cpu().numpy(), po.image_eval_transform(im, 224, MEAN, STD))
    d = torch.rand(300, 410, generator=torch.Generator().manual_seed(4)) * 80
    dp = str(tmp_path / "d.pt"); torch.save(d, dp)
    gd = DP.DepthProcessor()([dp, d], device="cuda")
    assert gd.shape == (2, 1, 224, 224) and np.abs(gd[0].cpu().numpy() - po.depth_eval_transform(d.numpy())).max() < 2e-4
    assert torch.equal(gd[0], gd[1])
    pc = np.random.default_rng(0).standard_normal((9000, 6)).astype(np.float32)
    pp = str(tmp_path / "pc.npy"); np.save(pp, pc)
    np.random.seed(7)
    gp = DP.PointCloudProcessor(n_sample_points=1024)(pp, device="cuda")
    np.random.seed(7)
    start = np.random.randint(0, 9000)
    assert gp.shape == (1, 1024, 6)
    sel, _ = O.pc_farthest_point_sample(pc, 1024, start)
    assert np.abs(gp[0].cpu().numpy() - O.pc_norm(sel)).max() < 2e-5
    tac = _img(480, 640, seed=9)
    tpath = str(tmp_path / "t.png"); Image.fromarray(tac).save(tpath)
    gt = DP.TactileProcessor()([tpath, tac], device="cuda")
    assert gt.shape == (2, 3, 224, 224) and torch.equal(gt[0], gt[1])
    e = torch.randn(128, 500, generator=torch.Generator().manual_seed(1))
    ep = str(tmp_path / "e.pth"); torch.save(e, ep)
    ge = DP.EEGProcessor()(ep, device="cuda")
    assert ge.shape == (1, 128, 512) and ge.is_cuda


def test_batched_transform_equals_the_per_image_path():
    """vl_resample_batch_u8_norm (a list of images of different sizes in two launches, host images in one packed copy)
    against the per-image kernels: identical bits, for the evaluation transform and for given crop boxes; mixed host /
    device inputs, greyscale and PIL images."""
    from open_clip.transform import image_transform
    from vitlens_hip import preproc
    shapes = [(240, 320), (530, 730), (730, 530), (224, 224), (100, 80), (375, 500), (225, 224), (64, 48)]
    host = [torch.from_numpy(_img(h, w, seed=i)) for i, (h, w) in enumerate(shapes)]
    mixed = [im.cuda() if i % 2 else im for i, im in enumerate(host)]
    got = preproc.images_to_tensor(mixed, 224, MEAN, STD)
    assert got.shape == (len(shapes), 3, 224, 224)
    for i, im in enumerate(host):
        assert torch.equal(got[i], preproc.image_to_tensor(im.cuda(), 224, MEAN, STD)), shapes[i]
        assert np.array_equal(got[i].cpu().numpy(), po.image_eval_transform(im.numpy(), 224, MEAN, STD))
    boxes = [(3, 5, h - 20, w - 30) for h, w in shapes]
    gb = preproc.images_to_tensor(mixed, (224, 224), MEAN, STD, boxes=boxes)
    for i, im in enumerate(host):
        assert torch.equal(gb[i], preproc.image_to_tensor(im.cuda(), (224, 224), MEAN, STD, box=boxes[i])), shapes[i]
    t = image_transform(224, is_train=False)
    items = [host[0].numpy(), _img(333, 250, 1)[..., 0], host[2].cuda()]
    b = t.batch(items)
    for i, it in enumerate(items):
        assert torch.equal(b[i], t(it)), i
    tt = image_transform(224, is_train=True, aug_cfg={"scale": (0.3, 1.0)})
    torch.manual_seed(5)
    bt = tt.batch(items)
    torch.manual_seed(5)
    for i, it in enumerate(items):
        assert torch.equal(bt[i], tt(it)), i
