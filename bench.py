#!/usr/bin/env python
"""Benchmark of the ViT-Lens contrastive hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: when no RANK is in the environment (the plain `python bench.py --gpus N` form) this process itself re-executes
the script under `torch.distributed.run --nnodes=1 --nproc-per-node N` (one rank per GPU, rendezvous on 127.0.0.1);
under an existing torchrun launch it is a rank.  Either way WORLD_SIZE must equal --gpus (asserted), rank 0 prints the
one JSON line with `n_gpus: N`, the per-rank step times and the collectives' share of the step.

Default workload = BASELINE.json configs[2] ("c3"), the largest single-GPU configuration and the one the north_star
target is stated on: ViT-L Lens(depth)->ViT + image + text tri-modal InfoNCE TRAINING step (forward of three towers,
TriClipLoss, backward through the visual tower, AdamW), synthetic batch 1024 per GPU in micro-batches of 256 (weak
scaling; for N>1 one packed RCCL all-gather of the embeddings + one flat gradient all-reduce inside the timed region).
`--workload c2` = configs[1] (ViT-L/14 image-tower forward, batch 256), c4 / c5 = the audio / point-cloud recipes.
A "step" = one pass of that path over one batch.  Inputs are resident in HBM when timing starts.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the bf16 MFMA GEMM; achieved =
algorithmic FLOPs / HIP-event time of those launches, measured live in the timed region) and
`cpu_baseline` (the oracle = CPU restatement of the reference, timed on this node's host cores on a
bounded sample; a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
LN_FOLDED = True          # set in main() from vitlens_hip.engine.LN_FOLD / --ln-fold
for p in (os.path.join(ROOT, "vit-lens_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0       # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
GF_PER_IMAGE = 162.03           # ViT-L/14 tower forward, 257 tokens, 2*MAC


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5", "selftest"],
                    help="c2: ViT-L/14 image-tower forward b=256 (BASELINE.json configs[1]); "
                         "c3 (default): depth tri-modal InfoNCE TRAINING step b=1024 (BASELINE.json configs[2]); "
                         "c4: audio Lens <-> text dual InfoNCE training step (configs[3]); "
                         "c5: point-cloud Lens + image + text tri-modal training step (configs[4]); "
                         "selftest: launcher / collective plumbing on CPU + gloo with a stub step (tests/test_bench_launcher.py)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 256 for c2/c4, 1024 for c3, 128 for c5)")
    ap.add_argument("--micro-batch", type=int, default=0, help="default: 256 (c3, c4), 128 (c5)")
    ap.add_argument("--res-dtype", default="bf16", choices=["f32", "bf16"],
                    help="residual-stream dtype of the forward towers: bf16 = what the reference's amp_bf16 autocast keeps "
                         "(conv/linear outputs and the residual adds are bf16 there; parity: tests/test_hip_towers.py), "
                         "f32 = the more precise variant (C2 'fp32 variant' of BASELINE.json)")
    ap.add_argument("--via", default="step", choices=["step", "api"],
                    help="c3 only.  step (default): vitlens_hip.step.TriModalDepthStep, the fused step object; api: the reference's "
                         "loop body verbatim on the drop-in modules (training/train.py:133-152,212-249): tri_create_model + lock "
                         "recipe, out = model(image, text, visual_x); loss(**out).backward(); torch.optim.AdamW.step(); clamp")
    ap.add_argument("--gemm-cfg", type=int, default=-1)
    ap.add_argument("--bn-sync", action="store_true", help="c5 with --gpus N: SyncBatchNorm in the point tokenizer (--use-bn-sync)")
    ap.add_argument("--text-arith", default="f16", choices=["f16", "bf16x2", "bf16"],
                    help="operands of the frozen text tower: f16 (default: IEEE half, cosine matrices 1-2e-4 from the fp32 CPU path at "
                         "the bf16 rate), bf16x2 (round 4: two-term bf16 weights, 6-8e-4 at twice the flops), bf16 (the reference's "
                         "amp_bf16 arithmetic, 0.8-1.9e-3)")
    ap.add_argument("--ln-fold", default=None, choices=["on", "off"],
                    help="LayerNorms of the frozen ViT blocks folded into the GEMMs either side of them (on) or run as their own "
                         "passes (off); default: the library's (on since round 5)")
    ap.add_argument("--no-overlap-frozen", dest="overlap_frozen", action="store_false", default=True,
                    help="c3 / c4 / c5: run the frozen image / text towers' forwards in the launch stream instead of on a second HIP "
                         "stream beside the trainable tower's forward (the product default since round 6: -1.35 %% per C3 step, "
                         "bit-equal results).  With the overlap on, `value` / `ms_per_step` are measured on that default path and "
                         "the per-launch `roofline` figures come from a short SERIAL pass after the timed region (a launch's "
                         "HIP-event duration only measures that launch when it owns the chip); `roofline.measured_on` says which")
    ap.add_argument("--overlap-frozen", dest="overlap_frozen", action="store_true", help="(round-5 spelling; now the default)")
    ap.add_argument("--no-overlap-backward", dest="overlap_backward", action="store_false", default=True,
                    help="c3 / c4 / c5: the two halves of the micro-batches run their backward on ONE stream (default: on two, "
                         "round 6; same arithmetic either way)")
    ap.add_argument("--serial-steps", type=int, default=2, help="steps of the serial roofline pass (overlap on only)")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1 only: initialise a ONE-rank RCCL communicator and run the step's multi-rank code path on it "
                         "(packed all-gather, bucketed async all-reduce; prints collective_ms_per_step) - the API / stream "
                         "contract of the exchange exercised on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="CPU-baseline sample: images (c2, default 48) / triplets per step (c3, default 8)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="default: one thread per physical core of the host")
    ap.add_argument("--detail", default="", help="optional path for a per-GEMM-shape breakdown (json)")
    return ap.parse_args()


def seeded_vitl_weights(seed=1234):
    """Random-init ViT-L/14 image tower with the reference's init distributions (no checkpoints offline).
    Generated by the PRODUCT side (plain torch RNG), key names = reference state_dict."""
    import math
    g = torch.Generator().manual_seed(seed)
    D, Hd, E, T, P, Ly = 1024, 4096, 768, 256, 14, 24
    s = D ** -0.5
    rn = lambda *sh, std=1.0: torch.randn(*sh, generator=g) * std
    un = lambda *sh, bound=1.0: (torch.rand(*sh, generator=g) * 2 - 1) * bound
    sd = {"image.class_embedding": rn(D, std=s), "image.positional_embedding": rn(T + 1, D, std=s),
          "image.proj": rn(D, E, std=s), "image.conv1.weight": un(D, 3, P, P, bound=(3 * P * P) ** -0.5)}
    for n in ("ln_pre", "ln_post"):
        sd[f"image.{n}.weight"] = 1 + rn(D, std=0.02); sd[f"image.{n}.bias"] = rn(D, std=0.02)
    for i in range(Ly):
        p = f"image.transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            sd[p + n + ".weight"] = 1 + rn(D, std=0.02); sd[p + n + ".bias"] = rn(D, std=0.02)
        sd[p + "attn.in_proj_weight"] = un(3 * D, D, bound=math.sqrt(6.0 / (4 * D)))
        sd[p + "attn.in_proj_bias"] = rn(3 * D, std=0.02)
        sd[p + "attn.out_proj.weight"] = un(D, D, bound=s); sd[p + "attn.out_proj.bias"] = rn(D, std=0.02)
        sd[p + "mlp.c_fc.weight"] = un(Hd, D, bound=s); sd[p + "mlp.c_fc.bias"] = un(Hd, bound=s)
        sd[p + "mlp.c_proj.weight"] = un(D, Hd, bound=Hd ** -0.5); sd[p + "mlp.c_proj.bias"] = un(D, bound=Hd ** -0.5)
    return sd


class GemmTimer:
    """Wraps ops.gemm with HIP events on the launch stream (torch's current stream)."""

    def __init__(self, ops):
        self.ops, self.records, self.on = ops, [], False
        self._gemm = ops.gemm

    def install(self, engine_mod):
        def gemm(a, w, *args, **kw):
            if not self.on:
                return self._gemm(a, w, *args, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = self._gemm(a, w, *args, **kw); e1.record()
            epi = kw.get("epi", 0); act = kw.get("act", 0)
            self.records.append((("gemm", a.shape[0], w.shape[0], a.shape[1], epi, act), e0, e1))
            return out

        self.ops.gemm = gemm
        # the LayerNorm-folding entries (ops.gemm_lnfold / gemm_res_rowstats: persistent kernel on the main rows + the leftover
        # rows through ops.gemm) are timed as ONE record per call under the key of the GEMM they stand for
        inner, lnf, rrs = self._gemm, self.ops.gemm_lnfold, self.ops.gemm_res_rowstats

        def timed(key, fn, *args, **kw):
            if not self.on:
                return fn(*args, **kw)
            self.ops.gemm = inner                      # the leftover-row call inside must not add a record of its own
            try:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); out = fn(*args, **kw); e1.record()
            finally:
                self.ops.gemm = gemm
            self.records.append((key, e0, e1))
            return out

        def gemm_lnfold(x, fold, mean, rstd, out, *args, **kw):
            key = ("gemm", x.shape[0], fold[0].shape[0], x.shape[1], self.ops.EPI_BF16, kw.get("act", 0))
            return timed(key, lnf, x, fold, mean, rstd, out, *args, **kw)

        def gemm_res_rowstats(a, w, *args, **kw):
            key = ("gemm", a.shape[0], w.shape[0], a.shape[1], self.ops.EPI_RES_BF16, 0)
            return timed(key, rrs, a, w, *args, **kw)

        self.ops.gemm_lnfold, self.ops.gemm_res_rowstats = gemm_lnfold, gemm_res_rowstats
        g16 = self.ops.gemm_f16

        def gemm_f16(a, w, *args, **kw):      # the fp16 text tower's launches count in the all-GEMM figures like any other
            key = ("gemm_f16", a.shape[0], w.shape[0], a.shape[1], kw.get("epi", 0), kw.get("act", 0))
            return timed(key, g16, a, w, *args, **kw)

        self.ops.gemm_f16 = gemm_f16

    def summary(self):
        agg = {}
        for key, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            d = agg.setdefault(key, [0, 0.0])
            d[0] += 1; d[1] += ms
        out = []
        for (kind, M, N, K, epi, act), (n, ms) in agg.items():
            fl = 2.0 * M * N * K
            out.append({"kind": kind, "M": M, "N": N, "K": K, "epi": epi, "act": act, "launches": n,
                        "avg_ms": ms / n, "tflops": fl / (ms / n * 1e-3) / 1e12, "flops_per_launch": fl})
        return out


class TimedComm:
    """TorchComm with HIP events around every collective (on the stream the step launches on), so that a multi-GPU run
    reports how much of its step is the exchange: the packed embedding all-gather, the gradient all-reduce(s) and, under
    gather_with_grad, the reduce-scatters.  Async bucket all-reduces are timed at their wait()."""

    def __init__(self, inner, host_clock=False):
        self.inner, self.on, self.ev, self.host_clock = inner, False, [], host_clock

    def _timed(self, name, fn, *a):
        if not self.on:
            return fn(*a)
        if self.host_clock:           # CPU / gloo selftest: collectives are synchronous, the host clock is the measurement
            t0 = time.perf_counter(); r = fn(*a)
            self.ev.append((name, (time.perf_counter() - t0) * 1e3, None))
            return r
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a); e1.record()
        self.ev.append((name, e0, e1))
        return r

    def all_gather(self, out, inp):
        return self._timed("all_gather", self.inner.all_gather, out, inp)

    def all_reduce_sum(self, t):
        return self._timed("all_reduce", self.inner.all_reduce_sum, t)

    def reduce_scatter_sum(self, out, inp):
        return self._timed("reduce_scatter", self.inner.reduce_scatter_sum, out, inp)

    def all_reduce_sum_async(self, t):
        h = self.inner.all_reduce_sum_async(t)
        outer = self

        class _H:
            def wait(self_h):
                return outer._timed("all_reduce_wait", h.wait)
        return _H()

    def summary(self, steps):
        agg = {}
        for name, e0, e1 in self.ev:
            agg[name] = agg.get(name, 0.0) + (e0 if e1 is None else e0.elapsed_time(e1))
        return {k: round(v / steps, 3) for k, v in agg.items()}


def hbm_traffic(dom):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    collected by tools/gpu_evidence_r03.sh with this very command and corrected per the MI355X guide by
    tools/traffic_summary.py).  PMC collection cannot run inside the timed process, so the number is read from
    profiles/; None when no summary for this kernel shape has been committed."""
    for name in ("r06_hbm_traffic_c3.json", "r05_hbm_traffic_c3.json", "r04b_hbm_traffic_c3.json", "r04_hbm_traffic_c3.json", "hbm_traffic.json"):        # newest summary first
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            for e in t.get("shapes", [t]):
                if (e.get("shape") == [dom["M"], dom["N"], dom["K"]] and e.get("epi", dom["epi"]) == dom["epi"]
                        and e.get("act", dom["act"]) == dom["act"]):
                    return round(e["traffic_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            continue
    return None


def pmc_mfma_busy(dom):
    """MFMA-busy fraction IN CYCLES of the dominant kernel from the committed SQ counter passes (SQ_VALU_MFMA_BUSY_CYCLES /
    (kernel cycles x 1024 SIMDs); tools/gpu_evidence_r05.sh -> profiles/r05_gemm_pmc.json, earlier rounds' files behind it).
    The chip is power-managed: through a GEMM loop the board sits at its 1 400 W limit and the shader clock at ~1.86 GHz instead
    of 2.4 (profiles/r05_power_trace.log), so the wall-clock fraction of the 2.5 PFLOP/s peak (`frac`) is lower than the
    fraction of cycles the matrix pipes are busy.  Kernel names: `gemm_nt_pk_kernel<EPI, ACT, F16>` since round 5 (the third
    parameter was the main-loop selector, always true, in the round-3 / round-4 files)."""
    base = {(1024, 4096, 3, 0): (3, 0), (4096, 1024, 0, 1): (0, 1), (4096, 1024, 0, 4): (0, 4), (4096, 1024, 6, 4): (6, 4),
            (3072, 1024, 0, 0): (0, 0)}.get((dom["N"], dom["K"], dom["epi"], dom["act"]))
    if base is None:
        return None
    epi, act = base
    # with the LayerNorm folding most launches of these shapes are the folding instantiations of the same kernel
    fold_act = {(3, 0): 20, (0, 1): 11, (0, 4): 14, (0, 0): 10}.get(base)
    cands = []
    if LN_FOLDED and fold_act is not None:
        cands += [(f, f"gemm_nt_pk_kernel<{epi}, {fold_act}, false>") for f in ("r06_gemm_pmc.json", "r05_gemm_pmc.json")]
    cands += [(f, f"gemm_nt_pk_kernel<{epi}, {act}, false>") for f in ("r06_gemm_pmc.json", "r05_gemm_pmc.json")]
    for src in ("r04b_gemm_pmc.json", "r04_gemm_pmc.json", "r03d_gemm_pmc.json", "r03c_gemm_pmc.json"):   # r04: round-4 epilogue; r03d: 16x16x32 main loop
        cands.append((src, f"gemm_nt_pk_kernel<{epi}, {act}, true>"))
        cands.append((src, f"gemm_nt_pk_kernel<{epi}, {act}>"))
    for src, key in cands:
        try:
            k = json.load(open(os.path.join(ROOT, "profiles", src)))["kernels"][key]
            return {"mfma_busy_frac_cycles": k["mfma_busy_frac"], "wait_any_frac": k["wait_any_frac"], "kernel_cycles": k["kernel_cycles"],
                    "kernel": key, "source": "profiles/" + src}
        except (OSError, ValueError, KeyError, TypeError):
            continue
    return None


def cpu_baseline(sample, sd, threads):
    import vitlens_oracle as O
    spec = O.TowerSpec()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(99)
    img = torch.randn(sample, 3, 224, 224, generator=g)
    with torch.no_grad():
        O.encode_image(sd, img[:1], spec, normalize=True)          # warm-up
        t0 = time.time()
        O.encode_image(sd, img, spec, normalize=True)
        dt = time.time() - t0
    return {"value": sample / dt, "unit": "modality-pairs/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32 ViT-L/14 encode_image on {sample} synthetic 224x224 images, 1 timed pass "
                      f"({dt:.1f} s) after a 1-image warm-up"}


def host_threads(req=0):
    """One thread per physical core (SMT siblings only add contention to fp32 GEMMs); `cores` in the JSON = what was used."""
    if req > 0:
        return req
    n = os.cpu_count() or 1
    try:
        import psutil
        n = psutil.cpu_count(logical=False) or n
    except Exception:
        n = max(1, n // 2)
    return n


def cpu_baseline_c3(n_trip, sd, threads, steps=3):
    """The oracle's tri-modal depth-recipe TRAINING step (same lock recipe: adapter + first 4 blocks + logit_scale trainable,
    TriClipLoss, torch autograd backward, torch.optim.AdamW) on the host cores: 1 warm-up step, then `steps` timed steps
    of `n_trip` triplets each (N = 8, all cores, >= 3 timed steps)."""
    import vitlens_oracle as O
    torch.set_num_threads(threads)
    tower, text, lens = O.TowerSpec(), O.TextSpec(), O.LensSpec(modality="depth", perceiver_identity=True)
    g = torch.Generator().manual_seed(99)
    img = torch.randn(n_trip, 3, 224, 224, generator=g); dep = torch.randn(n_trip, 1, 224, 224, generator=g)
    txt = synth_text(n_trip, g)
    sdc = {k: v.clone().float() for k, v in sd.items()}
    train = ["logit_scale", "visual.visual_adapter.conv1.weight", "visual.visual_adapter.pos_emb"] + \
            [k for k in sdc if any(k.startswith(f"visual.transformer.resblocks.{l}.") for l in range(4))]
    for k in train:
        sdc[k].requires_grad_(True)
    opt = torch.optim.AdamW([sdc[k] for k in train], lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)

    def one():
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            fi = O.encode_image(sdc, img, tower, normalize=True); ft = O.encode_text(sdc, txt, text, normalize=True)
        fv = O.encode_visual(sdc, dep, tower, lens, normalize=True)
        loss = O.tri_clip_loss(fi, ft, fv, sdc["logit_scale"].exp())
        loss.backward(); opt.step()
        return float(loss)
    one()
    t0 = time.time()
    for _ in range(steps):
        one()
    dt = time.time() - t0
    return {"value": n_trip * steps / dt, "unit": "modality-pairs/sec", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 tri-modal depth TRAINING step (3 ViT-L towers fwd, TriClipLoss, autograd backward of the visual "
                      f"tower, AdamW) on {n_trip} synthetic triplets per step, {steps} timed steps ({dt:.1f} s) after 1 warm-up "
                      f"step, {threads} threads = physical cores of this host ({os.cpu_count()} logical)"}


def cpu_baseline_lens(workload, n, sd, threads, steps=2):
    """The oracle's TRAINING step of the audio dual recipe (c4: AST tokenizer + Perceiver + class token trainable, locked ViT,
    ClipLossGeneral vs text) or of the point-cloud tri-modal recipe (c5: PointBERT tokenizer with train-mode BatchNorm +
    Perceiver + class token trainable, TriClipLoss) on the host cores: torch autograd backward + torch.optim.AdamW,
    1 warm-up step, then `steps` timed steps of `n` samples each."""
    import vitlens_oracle as O
    torch.set_num_threads(threads)
    tower, text = O.TowerSpec(), O.TextSpec()
    g = torch.Generator().manual_seed(99)
    txt = synth_text(n, g)
    sdc = {k: v.clone().float() for k, v in sd.items()}
    if workload == "c4":
        lens = O.LensSpec(modality="audio", perceiver_identity=False, depth=2, self_per_cross=3, num_latents=256, latent_dim=1024,
                          input_chan=1024)
        x = torch.randn(n, 512, 128, generator=g) * 0.5
    else:
        lens = O.LensSpec(modality="pc", perceiver_identity=False, depth=4, self_per_cross=1, num_latents=256, latent_dim=1024,
                          input_chan=384)
        img = torch.randn(n, 3, 224, 224, generator=g)
        x = torch.rand(n, 8192, 3, generator=g) * 2 - 1
        start = torch.randint(0, 8192, (n,), generator=g)
    train = [k for k in sdc if k == "logit_scale" or k == "visual.class_embedding" or k.startswith("visual.perceiver.")
             or (k.startswith("visual.visual_adapter.") and "running" not in k and "num_batches" not in k)]
    for k in train:
        sdc[k].requires_grad_(True)
    opt = torch.optim.AdamW([sdc[k] for k in train], lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)

    def one():
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            ft = O.encode_text(sdc, txt, text, normalize=True)
            fi = O.encode_image(sdc, img, tower, normalize=True) if workload == "c5" else None
        if workload == "c4":
            fv = O.encode_visual(sdc, x, tower, lens, normalize=True)
            loss = O.clip_loss(fv, ft, sdc["logit_scale"].exp())
        else:
            fv = O.encode_visual(sdc, x, tower, lens, normalize=True, fps_start=start, training=True, running_out={})
            loss = O.tri_clip_loss(fi, ft, fv, sdc["logit_scale"].exp())
        loss.backward(); opt.step()
    one()
    t0 = time.time()
    for _ in range(steps):
        one()
    dt = time.time() - t0
    what = ("audio dual (Lens trainable, locked ViT, ClipLossGeneral vs text)" if workload == "c4"
            else "point-cloud tri-modal (PointBERT tokenizer w/ train-mode BatchNorm + Perceiver trainable, TriClipLoss)")
    return {"value": n * steps / dt, "unit": "modality-pairs/sec", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 {what} TRAINING step (forward, autograd backward, AdamW) on {n} synthetic samples per step, "
                      f"{steps} timed steps ({dt:.1f} s) after 1 warm-up step, {threads} threads = physical cores of this host "
                      f"({os.cpu_count()} logical)"}


def seeded_tri_weights(seed=1234):
    """image + text + depth-Lens `visual` towers + logit_scale (product-side random init, reference key names)."""
    import math
    sd = seeded_vitl_weights(seed)
    vis = seeded_vitl_weights(seed + 1)
    for k, v in vis.items():
        if k != "image.conv1.weight":
            sd["visual." + k[len("image."):]] = v
    g = torch.Generator().manual_seed(seed + 2)
    D, W, Ly, V, C, E = 1024, 768, 12, 49408, 77, 768
    rn = lambda *sh, std=1.0: torch.randn(*sh, generator=g) * std
    sd["visual.visual_adapter.conv1.weight"] = (torch.rand(D, 1, 14, 14, generator=g) * 2 - 1) / 14.0
    sd["visual.visual_adapter.pos_emb"] = rn(256, D, std=D ** -0.5)
    sd["token_embedding.weight"] = rn(V, W, std=0.02); sd["positional_embedding"] = rn(C, W, std=0.01)
    sd["text_projection"] = rn(W, E, std=W ** -0.5)
    sd["ln_final.weight"] = torch.ones(W); sd["ln_final.bias"] = torch.zeros(W)
    ps = (W ** -0.5) * ((2 * Ly) ** -0.5)
    for i in range(Ly):
        p = f"transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            sd[p + n + ".weight"] = torch.ones(W); sd[p + n + ".bias"] = torch.zeros(W)
        sd[p + "attn.in_proj_weight"] = rn(3 * W, W, std=W ** -0.5); sd[p + "attn.in_proj_bias"] = torch.zeros(3 * W)
        sd[p + "attn.out_proj.weight"] = rn(W, W, std=ps); sd[p + "attn.out_proj.bias"] = torch.zeros(W)
        sd[p + "mlp.c_fc.weight"] = rn(4 * W, W, std=(2 * W) ** -0.5); sd[p + "mlp.c_fc.bias"] = torch.zeros(4 * W)
        sd[p + "mlp.c_proj.weight"] = rn(W, 4 * W, std=ps); sd[p + "mlp.c_proj.bias"] = torch.zeros(W)
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07))
    return sd


def synth_text(n, gen, ctx=77, vocab=49408):
    t = torch.zeros(n, ctx, dtype=torch.long)
    for i in range(n):
        k = int(torch.randint(4, 21, (1,), generator=gen))
        t[i, 0] = vocab - 2; t[i, 1:1 + k] = torch.randint(1, vocab - 2, (k,), generator=gen); t[i, 1 + k] = vocab - 1
    return t


def emit_result_line(out, rank, use_dist):
    """Rank 0's ONE JSON line, as the LAST line of the job's stdout.  Native libraries print through C stdio, which is fully
    buffered on a pipe and flushed at process exit: RCCL's five-line version banner came out AFTER the JSON line of a
    `--force-dist` run (round 6, `gpurun_out/fd_stdout.log`) - a driver that reads the last line would not find the result.
    So: every rank flushes C stdio, the ranks meet, the process group is torn down, ranks other than 0 say nothing more, and
    rank 0 prints the line and then closes its stdout for whatever an exit handler might still write."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if rank == 0:
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: become the launcher.  Re-executes this script with the same
    arguments as N ranks of one node under torch.distributed.run (reference: env-based world discovery of
    training/distributed.py:45-108; the spawn is the part torchrun does for it).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    try:        # nothing this launcher process holds in C stdio may come out behind the ranks' result line
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    return subprocess.call(cmd, env=env)


def selftest_main(a, rank, world):
    """CPU + gloo stand-in for the timed loop: same launch, barrier / max-over-ranks timing, TimedComm accounting and JSON
    line as the GPU workloads, with a stub step (one matmul + the packed all-gather + one all-reduce)."""
    import torch.distributed as dist
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from vitlens_hip import step as _vstep          # loads the built library (no GPU needed to load it)
        comm = TimedComm(_vstep.TorchComm(), host_clock=True)
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(a.batch or 64, 48, generator=g); w = torch.randn(48, 48, generator=g)
    gathered = torch.empty(world * x.shape[0], 48)
    grad = torch.ones(1000)

    def step():
        f = torch.nn.functional.normalize(x @ w, dim=-1)
        if comm is not None:
            comm.all_gather(gathered, f)
            comm.all_reduce_sum(grad)
        return f.sum()
    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
        comm.on = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        f = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank_ms, coll = None, None
    if world > 1:
        comm.on = False
        coll = comm.summary(a.steps)
        allt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([dt], dtype=torch.float64))
        per_rank_ms = [round(float(t) / a.steps * 1e3, 3) for t in allt]
        dt = max(float(t) for t in allt)
    if rank == 0:
        ms = dt / a.steps * 1e3
        out = {"metric": "selftest steps/sec (launcher + collective plumbing, CPU/gloo stub step: NOT a performance number)",
               "value": round(world * x.shape[0] * a.steps / dt, 2), "unit": "rows/sec", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "config": {"workload": "selftest", "parallelism": f"dp{world}"},
               "roofline": None, "cpu_baseline": None}
        if world > 1:
            out["per_rank_ms_per_step"] = per_rank_ms
            out["collective_ms_per_step"] = coll
            out["collective_share"] = round(sum(coll.values()) / ms, 4) if coll else 0.0
    emit_result_line(out if rank == 0 else None, rank, world > 1)


def main():
    a = parse()
    have_rank = "RANK" in os.environ
    if a.gpus > 1 and not have_rank:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks"
    if a.workload == "selftest":
        return selftest_main(a, rank, world)
    if a.batch == 0:
        a.batch = {"c2": 256, "c3": 1024, "c4": 256, "c5": 128}[a.workload]
    if a.micro_batch == 0:
        a.micro_batch = 128 if a.workload == "c5" else 256
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from vitlens_hip import engine as _engine
    if a.ln_fold is not None:            # (not given: the library default - folded - decides)
        _engine.LN_FOLD = a.ln_fold == "on"
    global LN_FOLDED
    LN_FOLDED = bool(_engine.LN_FOLD)
    dist = None
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from vitlens_hip import engine, ops
    timer = GemmTimer(ops); timer.install(engine)
    comm = None
    if use_dist:
        from vitlens_hip import step as _vstep
        comm = TimedComm(_vstep.TorchComm())
    res_dtype = torch.float32 if a.res_dtype == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randn(a.batch, 3, 224, 224, generator=g).to(dev)          # resident in HBM
    if a.workload == "c2":
        sd = seeded_vitl_weights()
        eng = engine.VitEngine(sd, "image.", engine.TowerCfg(), dev, res_dtype=res_dtype, gemm_cfg=a.gemm_cfg)
        gathered = torch.empty(world * a.batch, 768, device=dev) if use_dist else None

        def step():
            f = eng.encode_image(images, normalize=True)
            if use_dist:
                dist.all_gather_into_tensor(gathered, f)        # one fused RCCL all-gather of [b,768]
            return f
    elif a.workload in ("c4", "c5"):
        # weights: the drop-in API's own seeded random init of the full tri-modal ViT-L model for this modality
        import open_clip
        from mm_vit_lens.model_cfg import fetch_model_cfg
        from vitlens_hip import step as vstep
        mod = "audio" if a.workload == "c4" else "pc"
        torch.manual_seed(1234)
        model = open_clip.tri_create_model("ViT-L-14", None, device="cpu", args=fetch_model_cfg(modality=mod))
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        tower_cfg, lens_cfg = model.visual._cfgs()
        del model
        texts = synth_text(a.batch, g).to(dev)
        if mod == "audio":
            audio = (torch.randn(a.batch, 512, 128, generator=g) * 0.5).to(dev)
            trainer = vstep.DualAudioStep(sd, tower_cfg, engine.TextCfg(), lens_cfg, dev, micro_batch=a.micro_batch,
                                          rank=rank, world_size=world, gemm_cfg=a.gemm_cfg, frozen_res_dtype=res_dtype, train_res_dtype=res_dtype, comm=comm,
                                          force_comm=a.force_dist, text_arith=a.text_arith, overlap_frozen=a.overlap_frozen, overlap_backward=a.overlap_backward)

            def step():
                return trainer.step(audio, texts)
        else:
            pts = torch.rand(a.batch, 8192, 3, generator=g) * 2 - 1
            pts = pts - pts.mean(1, keepdim=True)
            pts = (pts / pts.norm(dim=-1).amax(1)[:, None, None]).to(dev)          # unit-sphere normalise (pc_processor.py:32-38)
            start = torch.randint(0, 8192, (a.batch,), generator=g).to(dev)
            trainer = vstep.TriModalPCStep(sd, tower_cfg, engine.TextCfg(), lens_cfg, dev, micro_batch=a.micro_batch,
                                           rank=rank, world_size=world, gemm_cfg=a.gemm_cfg, bn_training=True, frozen_res_dtype=res_dtype, train_res_dtype=res_dtype, comm=comm,
                                           bn_sync=a.bn_sync, force_comm=a.force_dist, text_arith=a.text_arith, overlap_frozen=a.overlap_frozen, overlap_backward=a.overlap_backward)

            def step():
                return trainer.step(images, texts, pts, start)
    elif a.via == "api":
        import math
        from types import SimpleNamespace
        import open_clip
        from mm_vit_lens.model_cfg import fetch_model_cfg
        assert world == 1, "--via api is the single-process drop-in path (the reference wraps it in DDP)"
        torch.manual_seed(1234)
        model = open_clip.tri_create_model("ViT-L-14", None, precision="fp32" if a.res_dtype == "f32" else "amp_bf16", device=dev,
                                           output_dict=True, args=fetch_model_cfg(modality="depth"))
        model.lock_image_tower(); model.lock_text_tower(); model.lock_visual_tower(unlock_trans_first_n_layers=4)
        loss_fn = open_clip.create_loss(SimpleNamespace(local_loss=False, gather_with_grad=False, rank=0, world_size=1, horovod=False,
                                                        n_tower=3, use_dual_loss=False, cache_dir=None))
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.2)
        sd = None
        depths = torch.randn(a.batch, 1, 224, 224, generator=g).to(dev)
        texts = synth_text(a.batch, g).to(dev)
        a.micro_batch = a.batch          # the module API holds one activation set: no micro-batching

        def step():
            opt.zero_grad(set_to_none=True)
            out = model(image=images, text=texts, visual_x=depths)
            loss = loss_fn(**out)
            loss.backward()
            opt.step()
            with torch.no_grad():
                model.logit_scale.clamp_(0, math.log(100))
            return loss.detach()
    else:
        from vitlens_hip import step as vstep
        sd = seeded_tri_weights()
        depths = torch.randn(a.batch, 1, 224, 224, generator=g).to(dev)
        texts = synth_text(a.batch, g).to(dev)
        trainer = vstep.TriModalDepthStep(sd, engine.TowerCfg(), engine.TextCfg(), dev, micro_batch=a.micro_batch,
                                          unlock_first_n=4, rank=rank, world_size=world, gemm_cfg=a.gemm_cfg, frozen_res_dtype=res_dtype, train_res_dtype=res_dtype, comm=comm,
                                          force_comm=a.force_dist, text_arith=a.text_arith, overlap_frozen=a.overlap_frozen, overlap_backward=a.overlap_backward)

        def step():
            return trainer.step(images, texts, depths)

    trainer_obj = locals().get("trainer")
    overlapped = bool(trainer_obj is not None and getattr(trainer_obj, "_overlap_active", False))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    timer.on = not overlapped          # two towers sharing the chip: per-launch events are taken in the serial pass below
    if comm is not None:
        comm.on = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        f = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.on = False
    if comm is not None:
        comm.on = False
    serial_ms = None
    if overlapped:
        # per-launch roofline figures: the SAME step with the frozen towers in the launch stream, outside the timed region
        # (`value` / `ms_per_step` above are the default, overlapped path); every rank runs it - the step has collectives
        trainer_obj.overlap_frozen = False              # (one stream for everything: `_overlap_active` is False)
        step(); torch.cuda.synchronize()
        timer.on = True
        ts = time.perf_counter()
        for _ in range(max(1, a.serial_steps)):
            step()
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - ts) / max(1, a.serial_steps) * 1e3
        timer.on = False
        trainer_obj.overlap_frozen = True
    per_rank_ms, coll = None, None
    if use_dist:
        coll = comm.summary(a.steps)
        allt = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([dt], device=dev, dtype=torch.float64))
        per_rank_ms = [round(float(x) / a.steps * 1e3, 3) for x in allt]
        dt = max(float(x) for x in allt)
    assert torch.isfinite(f).all()
    if a.workload != 'c2':
        out_loss = float(f)

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * a.batch * a.steps / dt
        gf_unit = {"c2": GF_PER_IMAGE, "c3": 531.3, "c4": 543.6, "c5": 733.1}[a.workload]      # SURVEY.md 8(d) / BASELINE.md section 3
        shapes = timer.summary()
        dom = max(shapes, key=lambda s: s["avg_ms"] * s["launches"]) if shapes else None
        tot_fl = sum(s["flops_per_launch"] * s["launches"] for s in shapes)
        tot_ms = sum(s["avg_ms"] * s["launches"] for s in shapes)
        roof = None
        gemm_steps = max(1, a.serial_steps) if overlapped else a.steps       # steps the GEMM records cover
        gemm_step_ms = serial_ms if overlapped else ms
        if dom:
            ach = dom["tflops"]
            pm = pmc_mfma_busy(dom)
            roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": hbm_traffic(dom),
                    "kernel": (f"vl_gemm_bf16_ex M={dom['M']} N={dom['N']} K={dom['K']} epi={dom['epi']} act={dom['act']} "
                               "(gemm_nt_pk_kernel<EPI,ACT>, the persistent 256x256 kernel, on the whole rounds of row tiles + "
                               "gemm_tail_kernel<EPI> on the leftover 256 rows; avg_launch_ms covers both launches)"),
                    "avg_launch_ms": round(dom["avg_ms"], 4), "flops_per_launch": dom["flops_per_launch"],
                    "all_gemm_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                    "all_gemm_frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                    "gemm_share_of_step": round(tot_ms / (gemm_step_ms * gemm_steps), 4),
                    "step_tflops": round(gf_unit * a.batch / (ms * 1e-3) / 1e3, 1),
                    "step_frac": round(gf_unit * a.batch / (ms * 1e-3) / 1e3 / PEAK_BF16_TFLOPS, 4),
                    # provenance, field by field: what THIS process measured on THIS box, and what is read from files
                    # committed under profiles/ (PMC passes cannot run inside the timed process)
                    "measured_on": (f"serial pass: {gemm_steps} step(s) after the timed region with the frozen towers in the launch "
                                    f"stream ({serial_ms:.1f} ms/step there); HIP events on the launch stream, this run, this box"
                                    if overlapped else "timed region: HIP events on the launch stream, this run, this box"),
                    "measured_in_this_run": ["achieved", "frac", "avg_launch_ms", "all_gemm_tflops", "all_gemm_frac",
                                             "gemm_share_of_step", "step_tflops", "step_frac"],
                    "from_committed_profiles": {"fields": ["traffic", "pmc"],
                                                "note": "builder's rocprofv3 PMC passes over this command on an earlier box; "
                                                        "NOT measured by this run"},
                    "pmc": pm}
        out = {"metric": "modality-pairs/sec (ViT-L, 224^2 patches)", "value": round(value, 2),
               "unit": "modality-pairs/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": ("C2: ViT-L/14 image-tower forward (encode_image+normalize), 224x224, "
                                       f"batch {a.batch}/GPU" + (", + RCCL all-gather of embeddings" if world > 1 else ""))
                          if a.workload == "c2" else
                          ({"c3": "C3: ViT-L Lens(depth)->ViT + image + text tri-modal InfoNCE TRAINING step (fwd+bwd+AdamW), "
                                  f"batch {a.batch}/GPU in micro-batches of {a.micro_batch}, first 4 blocks + adapter trainable",
                            "c4": "C4: ViT-L Lens(audio: AST tokenizer + Perceiver depth 2 x (cross + 3 self)) -> locked ViT vs text, "
                                  f"dual InfoNCE TRAINING step (fwd+bwd+AdamW), batch {a.batch}/GPU in micro-batches of {a.micro_batch}",
                            "c5": "C5: ViT-L Lens(point cloud: FPS/kNN PointBERT tokenizer w/ train-mode BatchNorm + Perceiver depth 4) "
                                  f"-> locked ViT + image + text tri-modal InfoNCE TRAINING step, batch {a.batch}/GPU in "
                                  f"micro-batches of {a.micro_batch}"}[a.workload]
                           + (", packed RCCL embedding all-gather + flat gradient all-reduce" if world > 1 else "")),
                          "global_batch": world * a.batch, "residual_dtype": a.res_dtype,
                          "accumulate": "fp32", "parallelism": f"dp{world}", "gemm_cfg": a.gemm_cfg, "via": a.via,
                          "text_tower_operands": {"f16": "fp16", "bf16x2": "bf16 x 2 weight terms", "bf16": "bf16"}[a.text_arith] if a.workload != "c2" else "n/a",
                          "layernorm": "folded into the GEMMs (frozen blocks)" if LN_FOLDED else "own passes",
                          "frozen_towers": ("second HIP stream beside the trainable tower's forward" if overlapped else "launch stream"),
                          "backward": ("two halves of the micro-batches on two HIP streams" if (overlapped and a.overlap_backward and a.batch // a.micro_batch >= 2)
                                       else "one stream"),
                          **({"force_dist": True} if a.force_dist else {})},
               "roofline": roof}
        if a.workload != "c2":
            out["final_loss"] = round(out_loss, 6)      # of the last timed step: the same for every launch configuration of one tree
        if use_dist:         # diagnosis of a scaling run: per-rank step time (stragglers) and the exchange's share of the step
            out["per_rank_ms_per_step"] = per_rank_ms
            out["collective_ms_per_step"] = coll
            out["collective_share"] = round(sum(coll.values()) / ms, 4) if coll else 0.0
        if not a.no_cpu_baseline and world == 1 and a.workload == "c2":
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample or 48, sd, host_threads(a.cpu_threads))
        elif not a.no_cpu_baseline and world == 1 and a.workload == "c3" and sd is not None:
            out["cpu_baseline"] = cpu_baseline_c3(a.cpu_sample or 8, sd, host_threads(a.cpu_threads))
        elif not a.no_cpu_baseline and world == 1 and a.workload in ("c4", "c5"):
            out["cpu_baseline"] = cpu_baseline_lens(a.workload, a.cpu_sample or 4, sd, host_threads(a.cpu_threads))
        else:
            out["cpu_baseline"] = None
        if a.detail:
            os.makedirs(os.path.dirname(os.path.abspath(a.detail)), exist_ok=True)
            with open(a.detail, "w") as fh:
                json.dump({"shapes": shapes, "ms_per_step": ms}, fh, indent=1)
    emit_result_line(out if rank == 0 else None, rank, use_dist)


if __name__ == "__main__":
    main()
