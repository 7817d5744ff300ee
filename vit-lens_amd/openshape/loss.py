"""The OpenShape tri-modal loss: open_clip's TriClipLoss plus the four retrieval accuracies its training loop logs
(reference: VitLens-OpenShape/src/loss.py:80-185).  The loss terms are the HIP pair kernels of open_clip.loss; the
accuracies are arg-max comparisons on the same similarities (one thin GEMM per pair)."""
import torch

from open_clip.loss import TriClipLoss as _TriClipLoss, gather_packed


class TriClipLoss(_TriClipLoss):
    def _acc(self, x, all_y, off):
        from open_clip.zero_shot_classifier import zero_shot_logits
        logits = zero_shot_logits(x.detach(), all_y.detach().t(), logit_scale=1.0)
        labels = torch.arange(x.shape[0], device=x.device) + off
        return (logits.argmax(dim=1) == labels).float().mean()

    def forward(self, image_features, text_features, visual_features, logit_scale, output_dict=False):
        gi = gt = None
        ai, at, av = image_features, text_features, visual_features
        if self.world_size > 1:
            ai, at, av = gather_packed([image_features, text_features, visual_features], self.local_loss,
                                       self.gather_with_grad, self.rank, self.world_size)
            gi, gt = (ai, av), (at, av)
        i_loss = self.pair_loss(image_features, visual_features, logit_scale, gi)
        t_loss = self.pair_loss(text_features, visual_features, logit_scale, gt)
        total = i_loss + t_loss
        if not output_dict:
            return total
        local = self.world_size > 1 and self.local_loss
        off = image_features.shape[0] * self.rank if local else 0
        qi, qt, qv = (image_features, text_features, visual_features) if local or self.world_size == 1 else (ai, at, av)
        t2v = self._acc(qt, av, off)
        return {"contrastive_loss": total, "i_contra_loss": i_loss, "t_contra_loss": t_loss,
                "i2v_acc": self._acc(qi, av, off), "v2i_acc": self._acc(qv, ai, off),
                "t2v_acc": t2v, "v2t_acc": t2v}          # (loss.py:161: v2t is computed from the text rows as well)
