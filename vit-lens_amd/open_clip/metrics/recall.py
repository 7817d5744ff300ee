"""Retrieval recall@{1,5,10} in both directions (reference: open_clip/metrics/recall.py:8-78).  The image (or audio /
video) features are collected per batch and gathered across ranks; the similarity matrix against the text features is
one GEMM on the HIP kernel (fp32-accurate bf16 hi/lo split, `zero_shot_logits`), the ranking is top-10 + id matching."""
import torch

from .base_metric import BaseMetric


class Recall(BaseMetric):
    def __init__(self):
        super().__init__()

    def initialize(self, text_ids, text_logits):
        self.text_ids, self.text_logits = text_ids, text_logits
        self._reset()

    def compute(self, image_ids, image_logits):
        self._push(image_ids=image_ids, image_logits=image_logits)

    def gathered(self):
        """All ranks' (ids, features), rank-major (a rank without batches contributes zero rows: BaseMetric._collected)."""
        like = self.text_logits.new_zeros((0,) + tuple(self.text_logits.shape[1:]))
        ids_like = torch.as_tensor(self.text_ids)[:0]       # (text_ids may be a Python list: the placeholder of a rank that pushed nothing)
        return self._collected("image_ids", like=ids_like), self._collected("image_logits", like=like)

    def merge_results(self, output_predict=False):
        from ..zero_shot_classifier import zero_shot_logits
        self.image_ids, self.image_logits = self.gathered()
        image_logits = self.image_logits
        sim_i2t = zero_shot_logits(image_logits, self.text_logits.t(), logit_scale=1.0)        # [N_img, N_txt]
        return self.retrieval_eval(sim_i2t, sim_i2t.t(), output_predict)

    def retrieval_eval(self, scores_i2t, scores_t2i, output_predict=False):
        def side(scores, cand_ids, query_ids):
            k = min(10, scores.size(1))
            rank = scores.topk(k=k, dim=1).indices
            predict = cand_ids[None, :].expand(rank.size(0), -1).gather(1, rank)
            hits = [predict[:, :r].eq(query_ids[:, None]).any(1).sum().item() for r in (1, 5, 10)]
            return predict, [100.0 * h / scores.size(0) for h in hits]
        predict_txt, (tr_r1, tr_r5, tr_r10) = side(scores_i2t, self.text_ids, self.image_ids)
        predict_img, (ir_r1, ir_r5, ir_r10) = side(scores_t2i, self.image_ids, self.text_ids)
        tr_mean, ir_mean = (tr_r1 + tr_r5 + tr_r10) / 3, (ir_r1 + ir_r5 + ir_r10) / 3
        predict_txt_results, predict_img_results = {}, {}
        if output_predict:
            for i, p in zip(self.image_ids.cpu().tolist(), predict_txt.cpu().tolist()):
                predict_txt_results[i] = p
            for i, p in zip(self.text_ids.cpu().tolist(), predict_img.cpu().tolist()):
                predict_img_results[i] = p
        return {"txt_r1": tr_r1, "txt_r5": tr_r5, "txt_r10": tr_r10, "txt_r_mean": tr_mean, "img_count": scores_i2t.size(0),
                "img_r1": ir_r1, "img_r5": ir_r5, "img_r10": ir_r10, "img_r_mean": ir_mean, "r_mean": (tr_mean + ir_mean) / 2,
                "txt_count": scores_t2i.size(0), "predict_txt": predict_txt_results, "predict_img": predict_img_results}
