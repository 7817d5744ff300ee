"""Zero-shot classification loop (reference: training/zero_shot.py:84-110 `run`, :36-81 accuracy helpers):
for every batch of an iterable `(inputs, target)` encode through the model, score against the template-averaged text
classifier (`open_clip.build_zero_shot_classifier`) on the HIP GEMM and count top-1 / top-5 hits.  The data loaders and
the per-dataset wrappers of the reference (`test_zeroshot_3d_core`, `test_rgbd_cls_single`, ...) are dataset plumbing and
stay out; they reduce to this loop with a different `feature_key` / input keyword."""
import torch

from open_clip import accuracy, zero_shot_logits


def run(model, classifier, dataloader, args, input_key="image", feature_key="image_features", logit_scale=100.0):
    """-> (top1, top5) as fractions of the samples seen.  `input_key` / `feature_key`: ("image", "image_features") is the
    reference's `run`; ("visual_x", "visual_features") scores the modality tower."""
    device = getattr(args, "device", None) or "cuda"
    top1 = top5 = n = 0.0
    with torch.no_grad():
        for inputs, target in dataloader:
            inputs, target = inputs.to(device), target.to(device)
            output = model(**{input_key: inputs})
            feats = output[feature_key] if isinstance(output, dict) else output[0]
            logits = zero_shot_logits(feats, classifier, logit_scale=logit_scale)
            a1, a5 = accuracy(logits, target, topk=(1, 5))
            top1 += a1; top5 += a5; n += inputs.size(0)
    return top1 / n, top5 / n
