"""CPU: host-side pieces of bench.py that do not need a GPU - synthetic inputs have the documented shape/semantics
(SURVEY 8d), the seeded weights carry the reference's parameter names, and the traffic summary is wired."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_synthetic_text_has_eot_at_argmax():
    import bench
    g = torch.Generator().manual_seed(0)
    t = bench.synth_text(16, g)
    assert t.shape == (16, 77) and t.dtype == torch.long
    assert bool((t[:, 0] == 49406).all())                       # SOT
    eot = t.argmax(dim=-1)                                        # encode_text pools at argmax (model.py:538)
    assert bool((t[torch.arange(16), eot] == 49407).all())
    assert bool(((eot >= 5) & (eot <= 21)).all())
    for i in range(16):
        assert bool((t[i, eot[i] + 1:] == 0).all())             # zero padding after EOT


def test_seeded_weights_have_reference_names_and_shapes():
    import bench
    sd = bench.seeded_tri_weights()
    for k, shape in (("image.conv1.weight", (1024, 3, 14, 14)), ("image.positional_embedding", (257, 1024)),
                     ("image.proj", (1024, 768)), ("visual.visual_adapter.conv1.weight", (1024, 1, 14, 14)),
                     ("visual.visual_adapter.pos_emb", (256, 1024)), ("token_embedding.weight", (49408, 768)),
                     ("text_projection", (768, 768)), ("transformer.resblocks.11.mlp.c_fc.weight", (3072, 768)),
                     ("visual.transformer.resblocks.23.attn.in_proj_weight", (3072, 1024)), ("logit_scale", ())):
        assert tuple(sd[k].shape) == shape, k
    assert "visual.conv1.weight" not in sd                       # the Lens tower has no RGB patch embedding of its own


def test_committed_hbm_traffic_summary_matches_the_dominant_gemm():
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    # headline entry = the dominant launch of the C3 step since round 3b: c_proj + bf16 residual on the persistent kernel
    assert t["shape"] == [65792, 1024, 4096] and "gemm_nt_pk_kernel<3, 0" in t["kernel"]
    got = bench.hbm_traffic({"M": 65792, "N": 1024, "K": 4096, "epi": 3, "act": 0})
    algorithmic = 2 * (65792 * 4096 + 1024 * 4096 + 2 * 65792 * 1024)                           # A + W + residual in + out
    assert got is not None and algorithmic <= got <= 1.5 * algorithmic
    # the c_fc family is in the same file, told apart by epilogue / activation
    fc = bench.hbm_traffic({"M": 65792, "N": 4096, "K": 1024, "epi": 0, "act": 1})
    fc_save = bench.hbm_traffic({"M": 65792, "N": 4096, "K": 1024, "epi": 0, "act": 4})
    alg_fc = 2 * (65792 * 1024 + 4096 * 1024 + 65792 * 4096)
    assert alg_fc <= fc <= 2.2 * alg_fc and fc_save > fc + 0.9 * 2 * 65792 * 4096               # + the gelu' tensor
    assert bench.hbm_traffic({"M": 1, "N": 2, "K": 3, "epi": 0, "act": 0}) is None


def test_traffic_summary_tells_two_gemm_shapes_of_one_kernel_apart(tmp_path):
    """tools/traffic_summary.py: one template instance (`gemm_nt_pk_kernel<3, 0, true>`) serves c_proj (K = 4096) and out_proj
    (K = 1024); "hi" / "lo" keep the launches above / below the midpoint of the kernel's FETCH_SIZE range, the corrections are
    the guide's (KiB -> bytes, gfx950 doubling of wide reads), and the first spec becomes the file's headline entry."""
    import csv
    import subprocess
    import sys
    name = "void (anonymous namespace)::gemm_nt_pk_kernel<3, 0, true>((anonymous namespace)::GemmP)"
    for c, vals in (("FETCH_SIZE", [100, 101, 400, 401, 402]), ("WRITE_SIZE", [50, 50, 60, 60, 60])):
        d = tmp_path / c
        d.mkdir()
        with open(d / "x_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for v in vals:
                w.writerow([name, c, v])
    out = tmp_path / "o.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_summary.py"), str(tmp_path / "FETCH_SIZE"),
                        str(tmp_path / "WRITE_SIZE"), str(out), "gemm_nt_pk_kernel<3, 0,|1,2,3|hi|3,0", "gemm_nt_pk_kernel<3, 0,|1,2,4|lo"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    t = json.load(open(out))
    assert t["shape"] == [1, 2, 3] and t["epi"] == 3 and t["act"] == 0 and t["launches_sampled"] == 3
    assert t["traffic_bytes_per_launch"] == 401 * 1024 * 2 + 60 * 1024
    lo = t["shapes"][1]
    assert lo["shape"] == [1, 2, 4] and lo["launches_sampled"] == 2 and lo["traffic_bytes_per_launch"] == 100.5 * 1024 * 2 + 50 * 1024


def test_committed_counter_summary_is_found_under_the_round5_kernel_names():
    """`roofline.pmc`: with the LayerNorm folding on (the default) the dominant c_proj launches are the row-statistics
    instantiation `gemm_nt_pk_kernel<3, 20, false>` (third template parameter: fp16 operands, false); the newest committed counter file
    is preferred, and a shape without a committed pass yields None.  (The derived `effective_clock_ghz` left the bench line
    in round 6: it mixed this run's event times with cycles and launch durations from committed files.)"""
    import bench
    dom = {"M": 65792, "N": 1024, "K": 4096, "epi": 3, "act": 0, "avg_ms": 0.4664}
    bench.LN_FOLDED = True
    p = bench.pmc_mfma_busy(dom)
    assert p is not None and p["source"] in ("profiles/r06_gemm_pmc.json", "profiles/r05_gemm_pmc.json")      # newest committed pass first
    assert p["kernel"] == "gemm_nt_pk_kernel<3, 20, false>"
    assert 0.5 < p["mfma_busy_frac_cycles"] < 0.8 and 6e5 < p["kernel_cycles"] < 1e6
    bench.LN_FOLDED = False
    assert bench.pmc_mfma_busy(dom)["kernel"] == "gemm_nt_pk_kernel<3, 0, false>"
    bench.LN_FOLDED = True
    assert bench.pmc_mfma_busy({"N": 5, "K": 7, "epi": 0, "act": 0}) is None
    assert not hasattr(bench, "leftover_launch_ms")
