"""Zero-shot classifier construction and scoring with the reference's call surface
(open_clip/zero_shot_classifier.py:27-90, training/zero_shot.py `accuracy`): the step that follows contrastive
training.  The text tower runs on the HIP engine through `model.encode_text`; the per-class template
average is a [classes, templates, E] reduction on a few KB and stays in torch; the scoring GEMM
`logit_scale * features @ classifier` runs on the HIP GEMM with the hi/lo bf16 split used for the InfoNCE logits."""
from typing import Callable, Optional, Sequence, Union

import torch


def _chunks(seq, n):
    for i in range(0, len(seq), n):
        yield seq[i:i + n]


def _unit(x: torch.Tensor) -> torch.Tensor:
    return x / x.norm(dim=-1, keepdim=True)


@torch.no_grad()
def build_zero_shot_classifier(model, tokenizer, classnames: Sequence[str], templates: Sequence[Union[Callable, str]],
                               num_classes_per_batch: Optional[int] = 10, device: Union[str, torch.device] = "cuda",
                               use_tqdm: bool = False) -> torch.Tensor:
    """-> classifier weights [embed_dim, num_classes]: for every class the unit-normalised mean of the unit-normalised
    text embeddings of all its prompt templates."""
    if not (isinstance(templates, Sequence) and len(templates) > 0):
        raise AssertionError("templates must be a non-empty sequence")
    if not (isinstance(classnames, Sequence) and len(classnames) > 0):
        raise AssertionError("classnames must be a non-empty sequence")
    fmt = isinstance(templates[0], str)
    groups = list(_chunks(list(classnames), num_classes_per_batch or len(classnames)))
    if use_tqdm:
        import tqdm
        groups = tqdm.tqdm(groups)
    cols = []
    for names in groups:
        prompts = [(t.format(c) if fmt else t(c)) for c in names for t in templates]
        emb = model.encode_text(tokenizer(prompts).to(device)).float()
        per_class = _unit(emb).reshape(len(names), len(templates), -1).mean(dim=1)
        cols.append(_unit(per_class).t())
    return torch.cat(cols, dim=1)


@torch.no_grad()
def build_zero_shot_classifier_legacy(model, tokenizer, classnames: Sequence[str], templates: Sequence[Union[Callable, str]],
                                      device: Union[str, torch.device] = "cuda", use_tqdm: bool = False) -> torch.Tensor:
    """The one-class-at-a-time variant (zero_shot_classifier.py:93-133): same weights as `build_zero_shot_classifier`
    with one text-tower call per class."""
    return build_zero_shot_classifier(model, tokenizer, classnames, templates, num_classes_per_batch=1, device=device, use_tqdm=use_tqdm)


def zero_shot_logits(features: torch.Tensor, classifier: torch.Tensor, logit_scale: float = 100.0) -> torch.Tensor:
    """logit_scale * features [N, E] @ classifier [E, C] on the HIP GEMM (fp32-accurate operands via the bf16 hi/lo split)."""
    from vitlens_hip import ops
    if features.device.type != "cuda":
        raise RuntimeError("zero_shot_logits runs on the MI355X kernels only")
    w = classifier.t().contiguous().float()                     # [C, E]
    C = w.shape[0]
    Cp = (C + 3) // 4 * 4                                       # the GEMM wants N % 4 == 0
    if Cp != C:
        w = torch.cat([w, torch.zeros(Cp - C, w.shape[1], device=w.device)], dim=0)
    logits = ops.gemm(ops.split_bf16x3(features.contiguous().float(), 0), ops.split_bf16x3(w, 1), None, epi=ops.EPI_F32,
                      alpha=float(logit_scale))
    return logits[:, :C]


def accuracy(output: torch.Tensor, target: torch.Tensor, topk=(1,)):
    """Number of correct predictions within the top-k, for every k (training/zero_shot.py:24-27 convention: counts, not
    fractions)."""
    k = max(topk)
    pred = output.topk(k, dim=1, largest=True, sorted=True).indices          # [N, k]
    hit = pred.eq(target.view(-1, 1))
    return [float(hit[:, :kk].any(dim=1).float().sum().item()) for kk in topk]


def acc(output: torch.Tensor, target: torch.Tensor, topk=(1,)):
    """Top-k accuracy in PERCENT of the batch plus the [maxk, N] hit matrix (training/zero_shot.py:45-59; the 3D
    evaluation accumulates `correct` per class from it)."""
    with torch.no_grad():
        maxk = max(topk)
        pred = output.topk(maxk, 1, True, True).indices.t()
        correct = pred.eq(target.reshape(1, -1).expand_as(pred))
        n = target.size(0)
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / n) for k in topk], correct


def cond_acc(output: torch.Tensor, target: torch.Tensor, idx_mapping, merge_idx: int = 100, topk=(1,)):
    """`acc` after merging the classes in `idx_mapping` into `merge_idx`, in the targets (IN PLACE, as the reference) and
    in the predictions; a sample counts once within the top-k (training/zero_shot.py:62-81)."""
    with torch.no_grad():
        maxk = max(topk)
        pred = output.topk(maxk, 1, True, True).indices
        for idx in idx_mapping:
            target[target == idx] = merge_idx
            pred[pred == idx] = merge_idx
        pred = pred.t()
        correct = pred.eq(target.reshape(1, -1).expand_as(pred))
        n = target.size(0)
        return [correct[:k].float().max(dim=0)[0].sum(0, keepdim=True).mul_(100.0 / n) for k in topk], correct
