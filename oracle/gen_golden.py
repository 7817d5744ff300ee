"""Generate tests/golden/*.npz by running the IMPORTED reference (build container only).

    python oracle/gen_golden.py            # writes tests/golden/

TEST INFRASTRUCTURE.  The reference is imported from /root/reference through
oracle/ref_loader.py; only *data* (seeded inputs, the reference's randomly initialised
state_dicts for tiny configs, and the reference's outputs) is written.  The fixtures pin
oracle/vitlens_oracle.py (tests/test_oracle_golden.py) and, through it, the HIP path.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

TINY = {
    "embed_dim": 32,
    "vision_cfg": {"image_size": 32, "layers": 2, "width": 64, "patch_size": 8, "head_width": 32},
    "text_cfg": {"context_length": 16, "vocab_size": 96, "width": 64, "heads": 2, "layers": 2},
}


def tiny_args(modality):
    common = dict(perceiver_num_latents=16, perceiver_latent_dim=64, perceiver_latent_heads=2,
                  perceiver_latent_dim_head=32, perceiver_cross_dim_head=64, perceiver_cross_heads=1)
    if modality == "depth":
        return ref_loader.lens_args("depth", **common)
    if modality == "audio_tied":    # perceiver_weight_tie_layers: layers 1, 2 of a depth-3 Perceiver are the same modules
        return ref_loader.lens_args("audio", audio_mel_bins=32, audio_target_length=48, audio_fstride=6, audio_tstride=6,
                                    perceiver_input_chan=64, perceiver_depth=3, perceiver_self_per_cross_attn=1,
                                    perceiver_weight_tie_layers=True, **common)
    if modality == "audio":
        return ref_loader.lens_args("audio", audio_mel_bins=32, audio_target_length=48,
                                    audio_fstride=6, audio_tstride=6, perceiver_input_chan=64,
                                    perceiver_depth=2, perceiver_self_per_cross_attn=2, **common)
    if modality == "pc":
        return ref_loader.lens_args("pc", pc_num_group=16, pc_group_size=8, pc_encoder_dims=64,
                                    pc_trans_dim=64, pc_npoints=256, perceiver_input_chan=64,
                                    perceiver_depth=2, perceiver_self_per_cross_attn=1,
                                    pc_tokenizer="pointbert", pc_in_channel=3, pc_radius=0.2, **common)
    if modality == "tactile":       # no Lens: the visual tower is a second trainable ViT with its own patch embedding
        return ref_loader.lens_args("tactile")
    if modality == "eeg":
        return ref_loader.lens_args("eeg", eeg_chans=8, eeg_time_len=40, eeg_window_size=3, eeg_stride=2,
                                    perceiver_input_chan=64, perceiver_depth=1, perceiver_self_per_cross_attn=1, **common)
    raise ValueError(modality)


def np_sd(model):
    return {"sd/" + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()
            if "num_batches_tracked" not in k}


def tiny_case(oc, modality, seed):
    torch.manual_seed(seed)
    args = tiny_args(modality)
    model = oc.tri_create_model("tiny-lens", None, precision="fp32", device="cpu",
                                output_dict=True, args=args)
    model.eval()
    if modality == "pc":  # make eval-mode BatchNorm non-trivial
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.8, 1.2)
                m.weight.data.normal_(1, 0.05)
                m.bias.data.normal_(0, 0.05)
    g = torch.Generator().manual_seed(seed + 100)
    B = 4
    image = torch.randn(B, 3, 32, 32, generator=g)
    text = torch.zeros(B, 16, dtype=torch.long)
    for i in range(B):
        k = 3 + i
        text[i, 0] = 94
        text[i, 1:1 + k] = torch.randint(1, 94, (k,), generator=g)
        text[i, 1 + k] = 95
    extra = {}
    if modality == "depth":
        vx = torch.randn(B, 1, 32, 32, generator=g)
    elif modality == "tactile":
        vx = torch.randn(B, 3, 32, 32, generator=g)
    elif modality in ("audio", "audio_tied"):
        vx = torch.randn(B, 48, 32, generator=g)
    elif modality == "eeg":
        vx = torch.randn(B, 8, 40, generator=g)
    else:
        vx = torch.rand(B, 256, 3, generator=g) * 2 - 1
        # reproduce the reference's internal torch.randint start (misc.py:60)
        torch.manual_seed(seed + 7)
        extra["in/fps_start"] = torch.randint(0, 256, (B,), dtype=torch.long).numpy()
        torch.manual_seed(seed + 7)
    vx.requires_grad_(False)
    out = model(image=image, text=text, visual_x=vx)
    res = np_sd(model)
    res.update(extra)
    res["in/image"] = image.numpy(); res["in/text"] = text.numpy(); res["in/visual_x"] = vx.numpy()
    for k in ("image_features", "text_features", "visual_features"):
        res["out/" + k] = out[k].detach().numpy()
    res["out/logit_scale"] = out["logit_scale"].detach().numpy()
    # un-normalised tower outputs
    if modality == "pc":
        torch.manual_seed(seed + 7)
    res["out/visual_raw"] = model.encode_visual(vx).detach().numpy()
    res["out/image_raw"] = model.encode_image(image).detach().numpy()
    res["out/text_raw"] = model.encode_text(text).detach().numpy()
    if modality == "pc":
        ad = model.visual.visual_adapter
        torch.manual_seed(seed + 7)
        nb, center = ad.group_divider(vx)
        res["out/pc_center"] = center.detach().numpy()
        res["out/pc_neighborhood_sorted"] = np.sort(nb.detach().numpy().reshape(B, 16, -1), axis=-1)
        torch.manual_seed(seed + 7)
        tok = ad(vx)
        res["out/pc_tokens"] = tok["x"].detach().numpy(); res["out/pc_pos"] = tok["pos"].detach().numpy()
    # losses (world_size 1) + feature grads
    from open_clip.loss import TriClipLoss, ClipLossGeneral
    feats = [torch.tensor(res["out/" + k], requires_grad=True)
             for k in ("image_features", "text_features", "visual_features")]
    ls = torch.tensor(float(out["logit_scale"]), requires_grad=True)
    tri = TriClipLoss()(feats[0], feats[1], feats[2], ls)
    tri.backward()
    res["out/tri_loss"] = tri.detach().numpy()
    for n, f in zip(("image", "text", "visual"), feats):
        res[f"out/tri_grad_{n}"] = f.grad.numpy()
    res["out/tri_grad_logit_scale"] = ls.grad.numpy()
    x = torch.tensor(res["out/visual_features"], requires_grad=True)
    y = torch.tensor(res["out/text_features"], requires_grad=True)
    ls2 = torch.tensor(float(out["logit_scale"]), requires_grad=True)
    dual = ClipLossGeneral()(x, y, ls2)
    dual.backward()
    res["out/dual_loss"] = dual.detach().numpy()
    res["out/dual_grad_x"] = x.grad.numpy(); res["out/dual_grad_y"] = y.grad.numpy()
    res["out/dual_grad_logit_scale"] = ls2.grad.numpy()
    # parameter grads of the tri-modal step for the trainable (visual) tower
    model.zero_grad()
    if modality == "pc":
        torch.manual_seed(seed + 7)
    out2 = model(image=image, text=text, visual_x=vx)
    loss = TriClipLoss()(out2["image_features"], out2["text_features"], out2["visual_features"],
                         out2["logit_scale"])
    loss.backward()
    res["out/step_loss"] = loss.detach().numpy()
    for name, prm in model.named_parameters():
        if name.startswith("visual.") or name == "logit_scale":
            if prm.grad is not None:
                res["grad/" + name] = prm.grad.numpy()
    meta = {"modality": modality, "seed": seed, "model_cfg": TINY,
            "args": {k: v for k, v in args.items() if isinstance(v, (int, float, str, bool, type(None)))}}
    res["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"tiny_{modality}.npz"), **res)
    print(f"tiny_{modality}: {len(res)} arrays, loss {float(loss):.6f}")


def per_op_case(oc):
    """Per-op vectors from reference modules at small shapes."""
    from open_clip.transformer import ResidualAttentionBlock, LayerNorm
    from open_clip.perceiver import FeedForward, Attention
    from open_clip.modal_audio.models.AST_tokenizer import AST_tokenizer
    torch.manual_seed(11)
    res = {}
    blk = ResidualAttentionBlock(48, 3).eval()
    x = torch.randn(5, 2, 48)                       # LND
    mask = torch.full((5, 5), float("-inf")).triu_(1)
    res.update({"blk/sd/" + k: v.detach().numpy() for k, v in blk.state_dict().items()})
    res["blk/in"] = x.permute(1, 0, 2).contiguous().numpy()
    res["blk/out"] = blk(x).permute(1, 0, 2).detach().contiguous().numpy()
    res["blk/out_causal"] = blk(x, attn_mask=mask).permute(1, 0, 2).detach().contiguous().numpy()
    ln = LayerNorm(48); ln.weight.data.normal_(1, .1); ln.bias.data.normal_(0, .1)
    res["ln/w"] = ln.weight.detach().numpy(); res["ln/b"] = ln.bias.detach().numpy()
    res["ln/in"] = res["blk/in"]; res["ln/out"] = ln(torch.tensor(res["blk/in"])).detach().numpy()
    ff = FeedForward(32).eval()
    xi = torch.randn(2, 7, 32)
    res.update({"ff/sd/" + k: v.detach().numpy() for k, v in ff.state_dict().items()})
    res["ff/in"] = xi.numpy(); res["ff/out"] = ff(xi).detach().numpy()
    at = Attention(32, 20, heads=1, dim_head=16).eval()
    ctx = torch.randn(2, 11, 20)
    res.update({"xattn/sd/" + k: v.detach().numpy() for k, v in at.state_dict().items()})
    res["xattn/in"] = xi.numpy(); res["xattn/ctx"] = ctx.numpy()
    res["xattn/out"] = at(xi, context=ctx).detach().numpy()
    ast = AST_tokenizer(fstride=5, tstride=5, input_fdim=24, input_tdim=33, patch_size=(8, 8), width=16)
    sp = torch.randn(2, 33, 24)
    o = ast(sp)
    res["ast/w"] = ast.conv1.weight.detach().numpy(); res["ast/pos"] = ast.pos_emb.detach().numpy()
    res["ast/in"] = sp.numpy(); res["ast/out"] = o["x"].detach().numpy()
    np.savez_compressed(os.path.join(OUT, "per_op.npz"), **res)
    print("per_op:", len(res), "arrays")


def tokenizer_case(oc):
    texts = [
        "An airplane", "A car", "A dog", "A guitar", "a bird", "a photo of a cat.",
        "A depth map of a kitchen, with chairs & a table!", "Hello, World!!  multiple   spaces\tand\ttabs",
        "it's a dog's life; they're here, I've been, we'll go, he'd say, I'm",
        "3D point cloud of an L-shaped sofa (grey) #42", "numbers 1234567890 and 3.14159",
        "UPPER lower MiXeD CaSe", "", " ", "a", "!!!", "hyphen-ated_words and under_scores",
        "email@example.com http://example.com/path?q=1", "sound of rain falling on a tin roof",
        "a " * 100, "supercalifragilisticexpialidocious antidisestablishmentarianism " * 8,
        "the quick brown fox jumps over the lazy dog", "<start_of_text> literal special <end_of_text>",
        "a spectrogram of dog barking", "a point cloud model of a chair", "brackets [x] {y} (z) <w>",
        "quotes 'single' \"double\" `back`", "math: 1+1=2, 2*3=6, 10/5=2, 2^8=256",
        "trailing punctuation...", "semi;colon:colon,comma.period", "percent 50% dollar $5 amp & at @",
        "newline\nseparated\nlines", "tabs\tand\r\ncarriage returns", "zero-shot classification of ModelNet40",
        "This is a photo of a airplane.", "a rendering of a lamp", "a cropped photo of the sink",
        "the sound of sea waves", "audio of a helicopter", "a bad photo of a bathroom",
        "a depth photo of a bedroom", "a tactile image of fabric", "eeg signal of a person viewing a panda",
        "x" * 200, "ab " * 60, "I'll've'd", "don't can't won't", "ALL CAPS SENTENCE WITH NUMBERS 123",
        "end with space ", " start with space", "mid  double  space", "a.b.c.d.e.f.g",
    ]
    ids = oc.tokenize(texts).numpy()
    ids16 = oc.tokenize(texts, context_length=16).numpy()
    with open(os.path.join(OUT, "tokenizer_kat.json"), "w") as f:
        json.dump({"texts": texts, "ids77": ids.tolist(), "ids16": ids16.tolist()}, f)
    print("tokenizer_kat:", len(texts), "strings")


def _dist_worker(rank, world, port, feats, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref_loader.load()
    from open_clip.loss import ClipLossGeneral, TriClipLoss
    out = {}
    for local_loss in (False, True):
        for gwg in (False, True):
            x = feats["x"][rank].clone().requires_grad_(True)
            y = feats["y"][rank].clone().requires_grad_(True)
            ls = torch.tensor(14.285714, requires_grad=True)
            loss = ClipLossGeneral(local_loss=local_loss, gather_with_grad=gwg, rank=rank,
                                   world_size=world)(x, y, ls)
            loss.backward()
            tag = f"dual_ll{int(local_loss)}_gg{int(gwg)}"
            out[tag + "_loss"] = loss.detach().numpy()
            out[tag + "_gx"] = x.grad.numpy(); out[tag + "_gy"] = y.grad.numpy()
            out[tag + "_gls"] = ls.grad.numpy()
    i = feats["x"][rank].clone().requires_grad_(True)
    t = feats["y"][rank].clone().requires_grad_(True)
    v = feats["z"][rank].clone().requires_grad_(True)
    ls = torch.tensor(14.285714, requires_grad=True)
    loss = TriClipLoss(rank=rank, world_size=world)(i, t, v, ls)
    loss.backward()
    out["tri_loss"] = loss.detach().numpy()
    out["tri_gi"] = i.grad.numpy(); out["tri_gt"] = t.grad.numpy(); out["tri_gv"] = v.grad.numpy()
    ret[rank] = out
    dist.destroy_process_group()


def dist_case(world=2, b=3, d=16, port=29611):
    import torch.multiprocessing as mp
    g = torch.Generator().manual_seed(5)
    def unit(n):
        t = torch.randn(n, d, generator=g)
        return t / t.norm(dim=-1, keepdim=True)
    feats = {k: [unit(b) for _ in range(world)] for k in ("x", "y", "z")}
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_dist_worker, args=(world, port, feats, ret), nprocs=world, join=True)
    res = {}
    for k in feats:
        for r in range(world):
            res[f"in/{k}{r}"] = feats[k][r].numpy()
    for r in range(world):
        for k, v in ret[r].items():
            res[f"rank{r}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, f"gather_w{world}.npz"), **res)
    print(f"gather_w{world}:", len(res), "arrays")


def pc_bn_train_case(oc, seed):
    """Same tiny point-cloud model and inputs as tiny_pc.npz, but with the PointTokenizer's BatchNorm layers in
    TRAIN mode (batch statistics, running-stat update) - what `model.train()` gives the PC recipe
    (training/train.py:90).  Only the quantities that differ from tiny_pc.npz are stored."""
    torch.manual_seed(seed)
    args = tiny_args("pc")
    model = oc.tri_create_model("tiny-lens", None, precision="fp32", device="cpu", output_dict=True, args=args)
    model.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.8, 1.2)
            m.weight.data.normal_(1, 0.05)
            m.bias.data.normal_(0, 0.05)
    ref = np.load(os.path.join(OUT, "tiny_pc.npz"))
    for k, v in model.state_dict().items():      # the two cases must share weights
        if "num_batches_tracked" not in k:
            assert np.array_equal(ref["sd/" + k], v.numpy()), k
    image, text, vx = (torch.tensor(ref["in/" + k]) for k in ("image", "text", "visual_x"))
    model.visual.visual_adapter.train()
    from open_clip.loss import TriClipLoss
    torch.manual_seed(seed + 7)
    out = model(image=image, text=text, visual_x=vx)
    loss = TriClipLoss()(out["image_features"], out["text_features"], out["visual_features"], out["logit_scale"])
    loss.backward()
    res = {"out/step_loss": loss.detach().numpy(), "out/visual_features": out["visual_features"].detach().numpy()}
    for name, prm in model.named_parameters():
        if (name.startswith("visual.") or name == "logit_scale") and prm.grad is not None:
            res["grad/" + name] = prm.grad.numpy()
    for k, v in model.state_dict().items():
        if "running_" in k:
            res["sd_after/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_pc_bntrain.npz"), **res)
    print(f"tiny_pc_bntrain: {len(res)} arrays, loss {float(loss):.6f}")


def pnsa_case(seed: int):
    """The `pnsa` point tokenizer (pointnet_util.py:345-368) alone, eval- and train-mode BatchNorm: weights, inputs, the
    FPS start the reference's fallback sampler drew, tokens, FPS / ball-query indices, and the gradients of a random
    upstream gradient w.r.t. every parameter (train mode).  10 000 x 6 inputs are the OpenShape geometry; the tiny case
    keeps the channel layout (xyz + 3 feature channels) at 600 points."""
    from types import SimpleNamespace
    ref = ref_loader.load_pointnet_util()
    cfg = SimpleNamespace(num_group=32, radius=0.3, group_size=16, in_dim=3, encoder_dims=64, trans_dim=64)
    torch.manual_seed(seed)
    tok = ref.PointNSATokenizer(cfg)
    with torch.no_grad():
        for bn in tok.sa.mlp_bns:
            bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
            bn.weight.normal_(1, 0.1); bn.bias.normal_(0, 0.1)
    g = torch.Generator().manual_seed(seed + 1)
    B, N = 4, 600
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    feats = torch.rand(B, N, 3, generator=g)
    res = {"meta": np.array(json.dumps({"cfg": vars(cfg), "B": B, "N": N}))}
    for k, v in tok.state_dict().items():
        res["sd/" + k] = v.numpy().copy()
    res["in/xyz"], res["in/features"] = xyz.numpy(), feats.numpy()
    torch.manual_seed(seed + 2)
    res["in/fps_start"] = torch.randint(0, N, (B,), dtype=torch.long).numpy()
    for mode in ("eval", "train"):
        tok.train(mode == "train")
        torch.manual_seed(seed + 2)
        out = tok(feats, xyz=xyz)["x"]
        res[f"out/{mode}/tokens"] = out.detach().numpy()
        if mode == "train":
            dctx = torch.randn(out.shape, generator=g)
            res["in/dctx"] = dctx.numpy()
            tok.zero_grad()
            out.backward(dctx)
            for name, prm in tok.named_parameters():
                res["grad/" + name] = prm.grad.numpy().copy()
            for k, v in tok.state_dict().items():
                if "running_" in k:
                    res["sd_after/" + k] = v.numpy().copy()
    torch.manual_seed(seed + 2)
    cidx = ref.farthest_point_sample(xyz, cfg.num_group)
    res["out/fps_idx"] = cidx.numpy()
    res["out/ball_idx"] = ref.query_ball_point(cfg.radius, cfg.group_size, xyz, ref.index_points(xyz, cidx)).numpy()
    np.savez_compressed(os.path.join(OUT, "tiny_pnsa.npz"), **res)
    print(f"tiny_pnsa: {len(res)} arrays")


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "pnsa":
        pnsa_case(seed=int(sys.argv[3]))
        return
    oc = ref_loader.load()
    if len(sys.argv) > 2 and sys.argv[1] == "--only":      # e.g. `--only eeg 23`: add one case without touching the others
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "tiny-lens.json"), "w") as f:
                json.dump(TINY, f)
            oc.add_model_config(td)
            tiny_case(oc, sys.argv[2], seed=int(sys.argv[3]))
        return
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "tiny-lens.json"), "w") as f:
            json.dump(TINY, f)
        oc.add_model_config(td)
        for i, m in enumerate(("depth", "audio", "pc")):
            tiny_case(oc, m, seed=20 + i)
        pc_bn_train_case(oc, seed=22)
    per_op_case(oc)
    tokenizer_case(oc)
    dist_case(2)
    dist_case(4, port=29613)


if __name__ == "__main__":
    main()
