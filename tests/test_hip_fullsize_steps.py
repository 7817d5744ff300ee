"""GPU: one TRAINING step of each recipe at FULL ViT-L geometry (C3 depth tri-modal, C4 audio dual, C5 point-cloud
tri-modal) against the oracle's autograd on the same seeded weights - loss and a sample of gradients spanning the whole
backward (first / last unlocked block, adapter, Perceiver, logit_scale) - plus properties at the BENCH geometry
(M = 257*256 token rows per micro-batch: persistent kernel + tail kernel), where the oracle is too slow: finiteness and
invariance of loss / gradients under the micro-batch split (different GEMM dispatch, same mathematics).

Round 1 had these only at width 64; the tail-row residual bug (NaN loss at b = 256) was found by bench.py, not by a test.
Tolerances (round 3): per recipe, from the measured values (gpurun_out/r03_errs, the worst tensor of each run) plus ~50 %:
  C3 f32 residual stream: rel. L2 2.7e-2 / cosine 0.99963 measured -> 4e-2 / 0.999;   C3 bf16 stream: 6.2e-2 / 0.99807 -> 9e-2 / 0.997;
  C4: 2.3e-2 / 0.99978 -> 4e-2 / 0.999;   C5: 1.07e-1 / 0.99424 on EVERY tensor (the kNN / max-pool routing of the bf16
  forward differs from the fp32 oracle's in a few groups, which shifts all downstream activations) -> 1.4e-1 / 0.992.
  d loss / d logit_scale: relative, 3 % (measured 0.3 - 1.2 %); its well-conditioned check is the correlated-features test."""
import math

import pytest
import torch

import vitlens_oracle as O

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def _weights(lens, seed=5):
    g = torch.Generator().manual_seed(seed)
    tower, text = O.TowerSpec(), O.TextSpec()
    sd = O.init_tower(tower, g, "image.")
    sd.update(O.init_tower(tower, g, "visual.", with_conv=False, tokens=lens.num_latents if not lens.perceiver_identity else None))
    sd.update(O.init_text(text, g))
    sd.update(O.init_lens(tower, lens, g))
    return sd, tower, text, g


def _oracle_step(sd, train_names, fwd):
    sdc = {k: v.clone().float() for k, v in sd.items()}
    for k in train_names:
        sdc[k].requires_grad_(True)
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    loss = fwd(sdc)
    loss.backward()
    return float(loss), {k: sdc[k].grad for k in train_names}


def _record(tag, errs):
    """Measured errors of a run, for setting the bounds: VL_RECORD_ERRS=<dir> writes them as json."""
    import json
    import os
    d = os.environ.get("VL_RECORD_ERRS")
    if d:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"fullsize_errs_{tag}.json"), "w") as f:
            json.dump(errs, f, indent=1)


def _check(step, ref_loss, ref_grads, loss, names, tol=8e-2, tag=None, cos_min=0.99):
    assert abs(float(loss) - ref_loss) < 3e-2, (float(loss), ref_loss)
    bad, errs = {}, {"loss": [float(loss), ref_loss]}
    for k in names:
        g = step.grads[k] if k in step.grads else None
        ref = ref_grads[k]
        if k.endswith("conv1.weight"):
            g = step.grads[k + "_gemm"][:, :ref[0].numel()].reshape(ref.shape)
        elif k == "logit_scale":
            ref = ref.reshape(1)
        assert g is not None, k
        if g.numel() == ref.numel():
            ref = ref.reshape(g.shape)                # (1x1 convolutions are kept as [out, in] matrices)
        e, c = relerr(g, ref), cosine(g, ref)
        errs[k] = [round(e, 5), round(c, 6)] if k != "logit_scale" else [float(g), float(ref)]
        if k == "logit_scale":          # a scalar; at random init a difference of nearly cancelling sums (|g| ~ 0.02 - 0.12)
            if abs(float(g) - float(ref)) > 3e-2 * abs(float(ref)) + 2e-4:
                bad[k] = (float(g), float(ref))
        elif e > (tol[k] if isinstance(tol, dict) else tol) or c < cos_min:
            bad[k] = (round(e, 4), round(c, 5))
    if tag:
        _record(tag, errs)
    assert not bad, bad


def test_c3_depth_step_vitl_vs_oracle_autograd():
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="depth", perceiver_identity=True)
    sd, tower, text, g = _weights(lens)
    B = 4
    img = torch.randn(B, 3, 224, 224, generator=g); dep = torch.randn(B, 1, 224, 224, generator=g); txt = O.synth_text(B, g)
    names = ["logit_scale", "visual.visual_adapter.conv1.weight", "visual.visual_adapter.pos_emb"]
    for l in (0, 3):
        p = f"visual.transformer.resblocks.{l}."
        names += [p + n for n in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "mlp.c_fc.weight",
                                  "mlp.c_fc.bias", "mlp.c_proj.weight", "ln_1.weight", "ln_2.bias")]

    def fwd(s):
        with torch.no_grad():
            fi = O.encode_image(s, img, tower, normalize=True); ft = O.encode_text(s, txt, text, normalize=True)
        fv = O.encode_visual(s, dep, tower, lens, normalize=True)
        return O.tri_clip_loss(fi, ft, fv, s["logit_scale"].exp())
    ref_loss, ref_grads = _oracle_step(sd, names, fwd)
    for res_dtype in (torch.float32, torch.bfloat16):
        st = ST.TriModalDepthStep(sd, E.TowerCfg(), E.TextCfg(), "cuda", micro_batch=2, unlock_first_n=4,
                                  train_res_dtype=res_dtype, frozen_res_dtype=res_dtype)
        loss = st.forward_backward(img.cuda(), txt.cuda(), dep.cuda())
        f32 = res_dtype == torch.float32
        _check(st, ref_loss, ref_grads, loss, names, tol=4e-2 if f32 else 9e-2, cos_min=0.999 if f32 else 0.997,
               tag="c3_" + ("f32" if f32 else "bf16"))
        del st
        torch.cuda.empty_cache()


def test_c3_logit_scale_gradient_at_a_non_degenerate_loss():
    """d loss / d logit_scale with a RELATIVE bound.  At random init the three feature sets are uncorrelated and the gradient
    is a difference of nearly cancelling sums (hence the absolute bound above).  Here the depth Lens tower is made a copy of
    the image tower (adapter = channel sum of conv1, same trunk) and is fed the grey image: matching image / depth features
    are nearly identical, the image<->depth pair loss is far below ln(B) and the gradient is a well-conditioned number."""
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="depth", perceiver_identity=True)
    sd, tower, text, g = _weights(lens, seed=11)
    for k in list(sd):
        if k.startswith("image.") and k != "image.conv1.weight":
            sd["visual." + k[len("image."):]] = sd[k].clone()
    sd["visual.visual_adapter.conv1.weight"] = sd["image.conv1.weight"].sum(1, keepdim=True).clone()
    sd["visual.visual_adapter.pos_emb"] = torch.zeros_like(sd["visual.visual_adapter.pos_emb"])
    B = 8
    dep = torch.randn(B, 1, 224, 224, generator=g)
    img = dep.expand(-1, 3, -1, -1) + 0.05 * torch.randn(B, 3, 224, 224, generator=g)
    txt = O.synth_text(B, g)
    names = ["logit_scale", "visual.visual_adapter.pos_emb"]

    def fwd(s):
        with torch.no_grad():
            fi = O.encode_image(s, img, tower, normalize=True); ft = O.encode_text(s, txt, text, normalize=True)
        fv = O.encode_visual(s, dep, tower, lens, normalize=True)
        return O.tri_clip_loss(fi, ft, fv, s["logit_scale"].exp())
    ref_loss, ref_grads = _oracle_step(sd, names, fwd)
    st = ST.TriModalDepthStep(sd, E.TowerCfg(), E.TextCfg(), "cuda", micro_batch=4, unlock_first_n=1)
    loss = st.forward_backward(img.cuda(), txt.cuda(), dep.cuda())
    gs, rs = float(st.grads["logit_scale"]), float(ref_grads["logit_scale"])
    _record("c3_logit_scale", {"loss": [float(loss), ref_loss], "logit_scale": [gs, rs]})
    assert ref_loss < 2 * math.log(B) - 0.3, ref_loss          # clearly below the 2 ln B of uncorrelated features
    assert abs(float(loss) - ref_loss) < 3e-2
    assert abs(rs) > 0.1 and abs(gs - rs) < 1e-2 * abs(rs), (gs, rs)          # measured: -0.32931 vs -0.32937


def test_c4_audio_step_vitl_vs_oracle_autograd():
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="audio", perceiver_identity=False, depth=2, self_per_cross=3, num_latents=256, latent_dim=1024,
                      input_chan=1024)
    sd, tower, text, g = _weights(lens)
    B = 4
    aud = torch.randn(B, 512, 128, generator=g) * 0.5; txt = O.synth_text(B, g)
    P = "visual.perceiver.layers."
    names = ["logit_scale", "visual.class_embedding", "visual.visual_adapter.conv1.weight", "visual.visual_adapter.pos_emb",
             "visual.perceiver.latents", P + "0.0.fn.to_q.weight", P + "0.0.fn.to_kv.weight", P + "0.0.norm_context.weight",
             P + "0.1.fn.net.0.weight", P + "0.1.fn.net.2.weight", P + "0.2.1.0.fn.to_q.weight", P + "0.2.1.0.fn.to_kv.weight",
             P + "1.2.2.0.fn.to_out.weight", P + "1.2.2.1.fn.net.0.bias", P + "1.2.2.1.norm.weight"]

    def fwd(s):
        with torch.no_grad():
            ft = O.encode_text(s, txt, text, normalize=True)
        fv = O.encode_visual(s, aud, tower, lens, normalize=True)
        return O.clip_loss(fv, ft, s["logit_scale"].exp())
    ref_loss, ref_grads = _oracle_step(sd, names, fwd)
    lc = E.LensCfg(modality="audio", perceiver_identity=False, depth=2, self_per_cross=3)
    st = ST.DualAudioStep(sd, E.TowerCfg(), E.TextCfg(), lc, "cuda", micro_batch=2)
    loss = st.forward_backward(aud.cuda(), txt.cuda())
    st.grads.update(st.reference_named_grads())          # (two micro-batches: the merged buffer, not one trainer's)
    _check(st, ref_loss, ref_grads, loss, names, tol=4e-2, cos_min=0.999, tag="c4")


def test_c5_pc_step_vitl_vs_oracle_autograd():
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="pc", perceiver_identity=False, depth=4, self_per_cross=1, num_latents=256, latent_dim=1024,
                      input_chan=384)
    sd, tower, text, g = _weights(lens)
    B = 4
    img = torch.randn(B, 3, 224, 224, generator=g); txt = O.synth_text(B, g)
    pts = torch.rand(B, 8192, 3, generator=g) * 2 - 1
    start = torch.randint(0, 8192, (B,), generator=g)
    P = "visual.perceiver.layers."
    names = ["logit_scale", "visual.perceiver.latents", P + "0.0.fn.to_kv.weight", P + "0.1.fn.net.2.weight",
             P + "3.2.0.0.fn.to_q.weight", P + "3.2.0.1.fn.net.0.weight", "visual.visual_adapter.reduce_dim.weight",
             "visual.visual_adapter.pos_embed.2.weight"]

    def fwd(s):
        with torch.no_grad():
            fi = O.encode_image(s, img, tower, normalize=True); ft = O.encode_text(s, txt, text, normalize=True)
        fv = O.encode_visual(s, pts, tower, lens, normalize=True, fps_start=start, training=False)
        return O.tri_clip_loss(fi, ft, fv, s["logit_scale"].exp())
    ref_loss, ref_grads = _oracle_step(sd, names, fwd)
    lc = E.LensCfg(modality="pc", perceiver_identity=False, depth=4, self_per_cross=1, input_chan=384)
    st = ST.TriModalPCStep(sd, E.TowerCfg(), E.TextCfg(), lc, "cuda", micro_batch=2, bn_training=False)
    loss = st.forward_backward(img.cuda(), txt.cuda(), pts.cuda(), start.cuda())
    st.grads.update(st.reference_named_grads())
    _check(st, ref_loss, ref_grads, loss, names, tol=1.4e-1, cos_min=0.992, tag="c5")


@pytest.mark.parametrize("res_dtype", [torch.bfloat16])
def test_c3_step_at_bench_geometry_is_finite_and_split_invariant(res_dtype):
    """b = 256 (M = 65792 token rows per GEMM: whole rounds on the persistent kernel + tail kernel) against the same batch
    as two micro-batches of 128 (M = 32896: different row split): same loss, same gradients up to bf16 noise, all finite.
    Recycled workspaces are poisoned with NaN first, so a tail row reading outside its micro-batch shows up."""
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="depth", perceiver_identity=True)
    sd, tower, text, g = _weights(lens, seed=7)
    B = 256
    img = torch.randn(B, 3, 224, 224, generator=g).cuda(); dep = torch.randn(B, 1, 224, 224, generator=g).cuda()
    txt = O.synth_text(B, g).cuda()
    junk = torch.full((1 << 28,), float("nan"), device="cuda"); del junk          # 1 GiB of NaN back to the allocator
    res = []
    for mb in (256, 128):
        st = ST.TriModalDepthStep(sd, E.TowerCfg(), E.TextCfg(), "cuda", micro_batch=mb, unlock_first_n=2,
                                  train_res_dtype=res_dtype, frozen_res_dtype=res_dtype)
        loss = st.forward_backward(img, txt, dep)
        assert torch.isfinite(loss), float(loss)
        assert all(bool(torch.isfinite(v).all()) for v in st.grads.values())
        res.append((float(loss), {k: v.detach().clone() for k, v in st.grads.items()}))
        del st
        torch.cuda.empty_cache()
    assert abs(res[0][0] - res[1][0]) < 2e-3, (res[0][0], res[1][0])
    assert abs(res[0][0] - 2 * math.log(B)) < 1.0        # random init: TriClipLoss = two pair losses, each near ln(256)
    # (measured 3.2-3.6e-2 on the block-0 MLP gradients: 24 blocks of bf16 streams summed in a different order)
    bad = {k: relerr(res[1][1][k], v) for k, v in res[0][1].items() if relerr(res[1][1][k], v) > 5e-2}
    assert not bad, bad


def test_c3_at_its_literal_configuration_b1024_as_4x256_with_adamw():
    """BASELINE.json configs[2] exactly as bench.py runs it: batch 1024 per GPU in FOUR micro-batches of 256, first 4 blocks +
    adapter + logit_scale trainable, bf16 streams, the frozen towers beside the trainable one on a second stream, one AdamW
    step.  Until round 6 only bench.py ran nmb = 4: a micro-batch-index bug that appears at 4 but not at 2 would have been
    found by the driver's bench, not by a test.  Checked: finite loss near 2 ln 1024 and finite gradients; loss and gradients
    equal the SAME batch as two micro-batches of 512 up to bf16 summation-order noise; every micro-batch contributes (the
    adapter gradient changes when any one quarter of the depth batch is replaced); AdamW moves every master and a second
    step runs.  142 GB of activations per configuration: the two step objects live one after the other."""
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="depth", perceiver_identity=True)
    sd, tower, text, g = _weights(lens, seed=11)
    B = 1024
    img = torch.randn(B, 3, 224, 224, generator=g).cuda(); dep = torch.randn(B, 1, 224, 224, generator=g).cuda()
    txt = O.synth_text(B, g).cuda()
    bf = torch.bfloat16
    res = {}
    for mb in (256, 512):
        st = ST.TriModalDepthStep(sd, E.TowerCfg(), E.TextCfg(), "cuda", micro_batch=mb, unlock_first_n=4,
                                  train_res_dtype=bf, frozen_res_dtype=bf)
        assert st._overlap_active                                  # the product default
        loss = st.forward_backward(img, txt, dep)
        assert len(st.trainers) == B // mb
        assert torch.isfinite(loss), float(loss)
        assert all(bool(torch.isfinite(v).all()) for v in st.grads.values())
        res[mb] = (float(loss), {k: v.detach().clone() for k, v in st.grads.items()})
        if mb == 256:
            before = {k: v.clone() for k, v in st.masters.items()}
            st.optimizer_step()
            still = [k for k, v in st.masters.items() if torch.equal(v, before[k]) and k != "logit_scale"]
            assert not still, still
            # each quarter of the batch reaches the gradients: replace one micro-batch's depth maps, the adapter gradient moves
            base = res[256][1]["visual.visual_adapter.conv1.weight_gemm"]
            st.load_state_dict(sd)
            for q in range(4):
                dep2 = dep.clone(); dep2[q * 256:(q + 1) * 256].mul_(-1.0)
                st.forward_backward(img, txt, dep2)
                assert relerr(st.grads["visual.visual_adapter.conv1.weight_gemm"], base) > 1e-2, q
            loss2 = st.step(img, txt, dep)
            assert torch.isfinite(loss2)
        del st
        torch.cuda.empty_cache()
    assert abs(res[256][0] - 2 * math.log(B)) < 1.0, res[256][0]
    assert abs(res[256][0] - res[512][0]) < 2e-3, (res[256][0], res[512][0])
    bad = {k: relerr(res[512][1][k], v) for k, v in res[256][1].items() if relerr(res[512][1][k], v) > 5e-2}
    assert not bad, bad


def test_c5_pc_backward_vitl_given_forward_routing_and_upstream_gradient():
    """C5's BACKWARD at ViT-L geometry with the two things that make the end-to-end comparison loose taken out (round 4):
    (1) the ROUTING - the max-pools' arg-max rows are read from the HIP forward's own activations and the oracle's autograd
    runs its mini-PointNet with those indices on the kernel's own grouped patches; (2) the UPSTREAM gradient - at random
    init with 4 samples dL/dfeatures is a difference of nearly equal terms, so the 1e-2 forward error of the features turns
    into 7 % on dL/dfeatures and from there, uniformly, on every parameter gradient (measured with given routing only:
    6.5-7.4e-2, cosine 0.9975 on all ten tensors).  Here both sides back-propagate the SAME seeded dL/dfeatures through
    tokenizer -> Perceiver (4 layers) -> 24 locked ViT-L blocks: what is left is the bf16 operand rounding of the backward
    kernels (measured 0.6-0.9e-2 relative, cosine >= 0.99996 on every tensor).  The end-to-end test above keeps its 1.4e-1 as the envelope of routing flips + loss conditioning."""
    from test_hip_train import _point_tokens_routed
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="pc", perceiver_identity=False, depth=4, self_per_cross=1, num_latents=256, latent_dim=1024,
                      input_chan=384)
    sd, tower, text, g = _weights(lens)
    B = 4
    img = torch.randn(B, 3, 224, 224, generator=g); txt = O.synth_text(B, g)
    pts = torch.rand(B, 8192, 3, generator=g) * 2 - 1
    start = torch.randint(0, 8192, (B,), generator=g)
    dfeat = torch.randn(B, 768, generator=g) * 0.05
    P = "visual.perceiver.layers."
    names = ["visual.perceiver.latents", P + "0.0.fn.to_kv.weight", P + "0.1.fn.net.2.weight",
             P + "3.2.0.0.fn.to_q.weight", P + "3.2.0.1.fn.net.0.weight", "visual.visual_adapter.reduce_dim.weight",
             "visual.visual_adapter.pos_embed.2.weight", "visual.visual_adapter.encoder.first_conv.0.weight",
             "visual.visual_adapter.encoder.second_conv.3.weight"]
    lc = E.LensCfg(modality="pc", perceiver_identity=False, depth=4, self_per_cross=1, input_chan=384)
    st = ST.TriModalPCStep(sd, E.TowerCfg(), E.TextCfg(), lc, "cuda", micro_batch=B, bn_training=False)
    st.forward_backward(img.cuda(), txt.cuda(), pts.cuda(), start.cuda())        # allocates the gradient views
    st.flat_grad.zero_()
    tr = st.trainers[0]
    feat = tr.forward(pts.cuda(), start.cuda())
    tr.backward(dfeat.cuda())
    st.grads.update(tr.perc.reference_named_grads())
    tk = tr.tok
    M = lens.pc_group_size
    patches, f, f2, c3 = tk.ctx[0], tk.ctx[5], tk.ctx[11], tk.ctx[13]
    BG = f.shape[0] // M
    idx1 = f.float().view(BG, M, -1).argmax(dim=1).cpu()
    idx2 = f2.float().view(BG, M, -1).argmax(dim=1).cpu()
    gp = patches[:, :3].float().cpu().view(BG, M, 3).transpose(1, 2).contiguous()
    centers = c3[:, :3].float().cpu()
    a = "visual.visual_adapter."
    ref_feat = []

    def fwd(s):
        tok = _point_tokens_routed(s, a, gp, centers, lens, idx1, idx2, False).reshape(B, lens.pc_num_group, -1)
        lat = O.perceiver(s, "visual.perceiver.", tok, lens)
        fv = O.vit_trunk(s, "visual.", lat, tower, lens.use_orig_pos)
        ref_feat.append(fv.detach())
        return (fv * dfeat).sum()
    _, ref_grads = _oracle_step(sd, names, fwd)
    assert relerr(feat, ref_feat[0]) < 2e-2, relerr(feat, ref_feat[0])
    bad, errs = {}, {}
    for k in names:
        gk, ref = st.grads[k], ref_grads[k]
        ref = ref.reshape(gk.shape)
        errs[k] = [round(relerr(gk, ref), 5), round(cosine(gk, ref), 6)]
        if errs[k][0] > 2e-2 or errs[k][1] < 0.9995:          # measured 0.6-0.9e-2 / 0.99996+
            bad[k] = errs[k]
    _record("c5_given_routing_and_upstream", errs)
    print("c5 backward given routing + upstream gradient:", errs)
    assert not bad, bad


def _poison():
    junk = torch.full((1 << 28,), float("nan"), device="cuda")          # 1 GiB of NaN back to the allocator
    del junk


def test_c4_step_at_bench_geometry_is_finite_and_split_invariant():
    """C4 at bench.py's geometry (b = 256: 65 536 latent rows and 307 200 context rows per GEMM - GEGLU / DGEGLU / fp32-residual
    epilogues on the persistent kernel, whole rounds + leftover rows in the ViT trunk) against two micro-batches of 128: same
    loss, same gradients up to bf16 noise, everything finite, recycled memory poisoned with NaN first."""
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="audio", perceiver_identity=False, depth=2, self_per_cross=3, num_latents=256, latent_dim=1024,
                      input_chan=1024)
    sd, tower, text, g = _weights(lens, seed=8)
    B = 256
    aud = (torch.randn(B, 512, 128, generator=g) * 0.5).cuda(); txt = O.synth_text(B, g).cuda()
    lc = E.LensCfg(modality="audio", perceiver_identity=False, depth=2, self_per_cross=3)
    _poison()
    res = []
    for mb in (256, 128):
        st = ST.DualAudioStep(sd, E.TowerCfg(), E.TextCfg(), lc, "cuda", micro_batch=mb, train_res_dtype=torch.bfloat16,
                              frozen_res_dtype=torch.bfloat16)
        loss = st.forward_backward(aud, txt)
        assert torch.isfinite(loss), float(loss)
        assert all(bool(torch.isfinite(v).all()) for v in st.grads.values())
        res.append((float(loss), {k: v.detach().clone() for k, v in st.grads.items()}))
        del st
        torch.cuda.empty_cache()
    assert abs(res[0][0] - res[1][0]) < 2e-3, (res[0][0], res[1][0])
    assert abs(res[0][0] - math.log(B)) < 0.5
    errs = {k: relerr(res[1][1][k], v) for k, v in res[0][1].items() if float(v.abs().max()) > 0}
    print("c4 split invariance, worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:3])
    bad = {k: e for k, e in errs.items() if e > 6e-2}
    assert not bad, bad


def test_c5_step_at_bench_geometry_is_finite_and_split_invariant():
    """C5 at bench.py's geometry (b = 128: 65 536 x 32 grouped points through the mini-PointNet, 65 536 latent rows): with
    train-mode BatchNorm (per-micro-batch statistics, what bench.py runs) finiteness and the running-statistics update; with
    frozen statistics (where the micro-batch split does not change the mathematics) b = 128 against 2 x 64."""
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="pc", perceiver_identity=False, depth=4, self_per_cross=1, num_latents=256, latent_dim=1024,
                      input_chan=384)
    sd, tower, text, g = _weights(lens, seed=9)
    B = 128
    img = torch.randn(B, 3, 224, 224, generator=g).cuda(); txt = O.synth_text(B, g).cuda()
    pts = torch.rand(B, 8192, 3, generator=g) * 2 - 1
    pts = (pts / pts.norm(dim=-1).amax(1)[:, None, None]).cuda()
    start = torch.randint(0, 8192, (B,), generator=g).cuda()
    lc = E.LensCfg(modality="pc", perceiver_identity=False, depth=4, self_per_cross=1, input_chan=384)
    kw = dict(train_res_dtype=torch.bfloat16, frozen_res_dtype=torch.bfloat16)
    _poison()
    st = ST.TriModalPCStep(sd, E.TowerCfg(), E.TextCfg(), lc, "cuda", micro_batch=128, bn_training=True, **kw)
    rm0 = {k: v[0].clone() for k, v in st.tok.running.items()}
    loss = st.forward_backward(img, txt, pts, start)
    assert torch.isfinite(loss) and abs(float(loss) - 2 * math.log(B)) < 1.0, float(loss)
    assert all(bool(torch.isfinite(v).all()) for v in st.grads.values())
    assert all(bool(torch.isfinite(v[0]).all()) and not torch.equal(v[0], rm0[k]) for k, v in st.tok.running.items())
    st.optimizer_step()
    assert all(bool(torch.isfinite(v).all()) for v in st.masters.values())
    del st
    torch.cuda.empty_cache()
    res = []
    for mb in (128, 64):
        st = ST.TriModalPCStep(sd, E.TowerCfg(), E.TextCfg(), lc, "cuda", micro_batch=mb, bn_training=False, **kw)
        loss = st.forward_backward(img, txt, pts, start)
        assert torch.isfinite(loss) and all(bool(torch.isfinite(v).all()) for v in st.grads.values())
        res.append((float(loss), {k: v.detach().clone() for k, v in st.grads.items()}))
        del st
        torch.cuda.empty_cache()
    assert abs(res[0][0] - res[1][0]) < 2e-3, (res[0][0], res[1][0])
    errs = {k: relerr(res[1][1][k], v) for k, v in res[0][1].items() if float(v.abs().max()) > 0}
    print("c5 split invariance, worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:3])
    # (measured 7.3e-2 on the first cross-attention's latent-side tensors: another GEMM dispatch for the 64-sample micro-batch
    #  moves a few bf16 activations across the max-pools' arg-max ties; everything else stays below 6e-2)
    bad = {k: e for k, e in errs.items() if e > 1e-1}
    assert not bad, bad


def _given_upstream(names, grads_of, ref_grads, tag, tol, cos_min):
    bad, errs = {}, {}
    for k in names:
        gk, ref = grads_of(k), ref_grads[k]
        ref = ref.reshape(gk.shape)
        errs[k] = [round(relerr(gk, ref), 5), round(cosine(gk, ref), 6)]
        if errs[k][0] > tol or errs[k][1] < cos_min:
            bad[k] = errs[k]
    _record(tag, errs)
    print(tag, errs)
    assert not bad, bad


@pytest.mark.parametrize("res_dtype", [torch.float32, torch.bfloat16])
def test_c3_depth_backward_vitl_given_upstream_gradient(res_dtype):
    """C3's BACKWARD at ViT-L geometry with the loss conditioning taken out (round 5; the C5 test above did this in round 4):
    at random init with 4 samples dL/dfeatures is a difference of nearly equal terms, so the end-to-end comparison
    (test_c3_depth_step_vitl_vs_oracle_autograd: 4e-2 / 9e-2) mostly measures how a 1e-2 forward error is amplified there.
    Here the HIP trainer and the oracle's autograd back-propagate the SAME seeded dL/d(raw features) through ln_post / proj,
    24 ViT-L blocks (the first 4 trainable) and the depth adapter: what is left is the bf16 operand rounding of the backward
    kernels and - with the bf16 stream - of the residual-gradient stream."""
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="depth", perceiver_identity=True)
    sd, tower, text, g = _weights(lens)
    B = 4
    img = torch.randn(B, 3, 224, 224, generator=g); dep = torch.randn(B, 1, 224, 224, generator=g); txt = O.synth_text(B, g)
    dfeat = torch.randn(B, 768, generator=g) * 0.05
    names = ["visual.visual_adapter.conv1.weight", "visual.visual_adapter.pos_emb"]
    for l in (0, 3):
        p = f"visual.transformer.resblocks.{l}."
        names += [p + n for n in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "mlp.c_fc.weight",
                                  "mlp.c_fc.bias", "mlp.c_proj.weight", "ln_1.weight", "ln_2.bias")]
    ref_feat = []

    def fwd(s):
        fv = O.encode_visual(s, dep, tower, lens, normalize=False)
        ref_feat.append(fv.detach())
        return (fv * dfeat).sum()
    _, ref_grads = _oracle_step(sd, names, fwd)
    st = ST.TriModalDepthStep(sd, E.TowerCfg(), E.TextCfg(), "cuda", micro_batch=B, unlock_first_n=4,
                              train_res_dtype=res_dtype, frozen_res_dtype=res_dtype)
    st.forward_backward(img.cuda(), txt.cuda(), dep.cuda())              # allocates the flat gradient views
    st.flat_grad.zero_()
    tr = st._trainer(0)
    feat = tr.forward(dep.cuda())
    tr.backward(dfeat.cuda().contiguous(), None)
    assert relerr(feat, ref_feat[0]) < 2e-2, relerr(feat, ref_feat[0])

    def grads_of(k):
        if k.endswith("conv1.weight"):
            ref = ref_grads[k]
            return st.grads[k + "_gemm"][:, :ref[0].numel()].reshape(ref.shape)
        return st.grads[k]
    f32 = res_dtype == torch.float32
    _given_upstream(names, grads_of, ref_grads, "c3_given_upstream_" + ("f32" if f32 else "bf16"),
                    tol=1.5e-2 if f32 else 2.5e-2, cos_min=0.9995)      # measured 0.51-0.74e-2 (f32 stream), 0.90-1.33e-2 (bf16 stream), cosine >= 0.99992


def test_c4_audio_backward_vitl_given_upstream_gradient():
    """The same for C4: AST tokenizer -> Perceiver (2 x (cross + 3 self)) -> 24 locked ViT-L blocks, the SAME seeded
    dL/d(raw features) on both sides."""
    from vitlens_hip import engine as E, step as ST
    lens = O.LensSpec(modality="audio", perceiver_identity=False, depth=2, self_per_cross=3, num_latents=256, latent_dim=1024,
                      input_chan=1024)
    sd, tower, text, g = _weights(lens)
    B = 4
    aud = torch.randn(B, 512, 128, generator=g) * 0.5; txt = O.synth_text(B, g)
    dfeat = torch.randn(B, 768, generator=g) * 0.05
    P = "visual.perceiver.layers."
    names = ["visual.class_embedding", "visual.visual_adapter.conv1.weight", "visual.visual_adapter.pos_emb",
             "visual.perceiver.latents", P + "0.0.fn.to_q.weight", P + "0.0.fn.to_kv.weight", P + "0.0.norm_context.weight",
             P + "0.1.fn.net.0.weight", P + "0.1.fn.net.2.weight", P + "0.2.1.0.fn.to_q.weight", P + "0.2.1.0.fn.to_kv.weight",
             P + "1.2.2.0.fn.to_out.weight", P + "1.2.2.1.fn.net.0.bias", P + "1.2.2.1.norm.weight"]
    ref_feat = []

    def fwd(s):
        fv = O.encode_visual(s, aud, tower, lens, normalize=False)
        ref_feat.append(fv.detach())
        return (fv * dfeat).sum()
    _, ref_grads = _oracle_step(sd, names, fwd)
    lc = E.LensCfg(modality="audio", perceiver_identity=False, depth=2, self_per_cross=3)
    st = ST.DualAudioStep(sd, E.TowerCfg(), E.TextCfg(), lc, "cuda", micro_batch=B)
    st.forward_backward(aud.cuda(), txt.cuda())
    st.flat_grad.zero_()
    tr = st.trainers[0]
    feat = tr.forward(aud.cuda())
    tr.backward(dfeat.cuda().contiguous())
    st.grads.update(tr.perc.reference_named_grads())
    assert relerr(feat, ref_feat[0]) < 2e-2, relerr(feat, ref_feat[0])

    def grads_of(k):
        if k.endswith("conv1.weight"):
            ref = ref_grads[k]
            return st.grads[k + "_gemm"][:, :ref[0].numel()].reshape(ref.shape)
        return st.grads[k]
    _given_upstream(names, grads_of, ref_grads, "c4_given_upstream", tol=1.5e-2, cos_min=0.9995)      # measured 0.44-0.80e-2, cosine >= 0.99997
