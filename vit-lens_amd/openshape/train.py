"""Step body of the OpenShape training loop (reference: VitLens-OpenShape/src/train.py:784-843,
`Trainer.train_one_epoch_openclip`, accum_freq == 1): precomputed text / image features in, point cloud through the
CLIPBindWrap tower, the tri-modal loss with its accuracies, backward, optimizer step."""
import torch


def openclip_step(model, logit_scale_net, loss_fn, optimizer, data, device="cuda", text_proj=None, image_proj=None,
                  fps_start=None):
    """data: dict with "xyz_dense" [B,N,3], "features_dense" [B,N,C], "text_feat" / "img_feat" lists of [1,E] (or [B,E]
    tensors).  Returns the loss dictionary of openshape.TriClipLoss (tensors)."""
    optimizer.zero_grad()
    stack = lambda v: (torch.vstack(list(v)) if not torch.is_tensor(v) else v).to(device=device, dtype=torch.float32)
    text_feat, img_feat = stack(data["text_feat"]), stack(data["img_feat"])
    kw = {} if fps_start is None else {"fps_start": fps_start}
    pred_feat = model(data["features_dense"].to(device), xyz=data["xyz_dense"].to(device), **kw)
    logit_scale = logit_scale_net(None)
    if text_proj is not None:
        text_feat = text_proj(text_feat)
    if image_proj is not None:
        img_feat = image_proj(img_feat)
    from open_clip.model import _normalize
    text_feat, img_feat, pred_feat = _normalize(text_feat), _normalize(img_feat), _normalize(pred_feat)
    loss_dict = loss_fn(img_feat, text_feat, pred_feat, logit_scale, output_dict=True)
    loss_dict["contrastive_loss"].backward()
    optimizer.step()
    return loss_dict
