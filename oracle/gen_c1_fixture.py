"""Generate tests/golden/c1_examples.npz: BASELINE config C1 on its ACTUAL inputs (build container only).

    python oracle/gen_c1_fixture.py

TEST INFRASTRUCTURE - data only.  The four example photographs of the reference's README walk-through
(/root/reference/assets/example/image_{bird,fire,dog,beach}.jpg with the texts of /root/reference/example.py:26-31) go
through the IMPORTED reference on the CPU in fp32:

  JPEG bytes -> Pillow decode -> Resize(224, BICUBIC) + CenterCrop + ToTensor + Normalize (what the reference's
  `image_transform(224, is_train=False)` computes; torchvision is not installed, the pipeline is the Pillow + torch calls it
  makes - pinned byte for byte in tests/test_preproc_oracle.py) -> reference TriCLIP ViT-B-32 `encode_image`,
  captions -> reference `tokenize` -> `encode_text`, softmax(100 * I @ T^T).

No checkpoint can be fetched offline, so the weights are the PRODUCT model's seeded random init (torch.manual_seed(0),
CPU generator: the GPU test re-creates exactly these tensors) loaded into the reference model by name.  Written: the JPEG
files' bytes (data files of the reference, 3.3 MB), the preprocessed tensors, the token ids, the reference's features and
probabilities.  No reference source is stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "c1_examples.npz")
NAMES = ["bird", "fire", "dog", "beach"]
TEXTS = ["a bird", "crackling fire", "a dog", "sea wave"]
SEED = 0


def product_weights():
    """state_dict of the product's seeded ViT-B-32 image model (names = the reference's)."""
    sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
    import warnings
    import open_clip as oc
    from mm_vit_lens.model_cfg import fetch_model_cfg
    torch.manual_seed(SEED)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = oc.tri_create_model("ViT-B-32", None, precision="fp32", device="cpu", output_dict=True, args=fetch_model_cfg(modality="image"))
    sd = {k: v.detach().float().clone() for k, v in m.state_dict().items()}
    sys.path.remove(os.path.join(ROOT, "vit-lens_amd"))
    for k in [k for k in sys.modules if k.split(".")[0] in ("open_clip", "mm_vit_lens", "training", "vitlens_hip", "openshape")]:
        del sys.modules[k]
    return sd


def main():
    from PIL import Image
    sd = product_weights()
    sys.path.insert(0, HERE)
    import ref_loader
    oc = ref_loader.load()
    from open_clip.constants import OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
    args = ref_loader.lens_args("image")
    model = oc.tri_create_model("ViT-B-32", None, precision="fp32", device="cpu", output_dict=True, args=args)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    out = {"meta/seed": np.int64(SEED), "meta/texts": np.array(TEXTS)}
    pre = []
    for n in NAMES:
        path = f"/root/reference/assets/example/image_{n}.jpg"
        raw = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        out[f"jpeg/{n}"] = raw
        im = Image.open(path).convert("RGB")
        w, h = im.size
        # torchvision Resize(224): shorter side -> 224, the other int(224 * long / short); CenterCrop: round((n - 224) / 2)
        if w <= h:
            nw, nh = 224, int(224 * h / w)
        else:
            nh, nw = 224, int(224 * w / h)
        top, left = int(round((nh - 224) / 2.0)), int(round((nw - 224) / 2.0))
        r = im.resize((nw, nh), Image.BICUBIC).crop((left, top, left + 224, top + 224))
        t = torch.from_numpy(np.array(r)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
        t = t.sub_(torch.as_tensor(OPENAI_DATASET_MEAN)[:, None, None]).div_(torch.as_tensor(OPENAI_DATASET_STD)[:, None, None])
        pre.append(t)
        out[f"size/{n}"] = np.array([h, w], dtype=np.int64)
    image = torch.stack(pre)
    text = oc.tokenize(TEXTS)
    with torch.no_grad():
        o = model(image=image, text=text)
    fi, ft = o["image_features"].float(), o["text_features"].float()
    out["pre"] = image.numpy()
    out["text_ids"] = text.numpy()
    out["image_features"] = fi.numpy()
    out["text_features"] = ft.numpy()
    out["probs"] = torch.softmax(100.0 * fi @ ft.t(), dim=-1).numpy()
    out["logit_scale"] = o["logit_scale"].detach().float().reshape(1).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;  probs diag", np.diag(out["probs"]))


if __name__ == "__main__":
    main()
