"""Point-cloud preprocessing: the oracle's numpy restatement against the reference's own functions (CPU,
build container only), and the GPU processor against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

import vitlens_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cloud(n, c, seed):
    g = np.random.default_rng(seed)
    pts = g.normal(size=(n, c)).astype(np.float32)
    pts[:, :3] *= np.array([1.0, 0.4, 2.0], dtype=np.float32)          # anisotropic, off-centre
    pts[:, :3] += np.array([0.3, -1.0, 0.5], dtype=np.float32)
    return pts


@pytest.mark.needs_reference
def test_oracle_pc_processing_equals_reference_functions():
    ref_root = "/root/reference/vitlens/src/open_clip/modal_3d/processors"
    if not os.path.isdir(ref_root):
        pytest.skip("reference tree not present")
    import importlib.util, types
    sys.modules.setdefault("omegaconf", types.SimpleNamespace(OmegaConf=None))      # only used by the config plumbing
    spec = importlib.util.spec_from_file_location("ref_pc_processor", os.path.join(ref_root, "pc_processor.py"))
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    pts = _cloud(700, 6, 1)
    np.random.seed(5)
    want = ref.farthest_point_sample(pts, 128)
    np.random.seed(5)
    start = np.random.randint(0, 700)
    got, idx = O.pc_farthest_point_sample(pts, 128, start)
    assert np.array_equal(got, want)
    assert np.allclose(O.pc_norm(got), ref.pc_norm(want), rtol=0, atol=0)
    # the random-subset branch (pc_processor.py:41-45,81-84): the product draws np.random.permutation(N)[:n], the reference
    # shuffles arange(N) in place - the same draws of the legacy global RNG
    np.random.seed(9)
    want = ref.random_sample(pts, 100)
    np.random.seed(9)
    assert np.array_equal(pts[np.random.permutation(700)[:100]], want)
    # and the example clouds of the reference (8192 points = npoint: the processor takes the random-subset branch)
    files = sorted(__import__("glob").glob("/root/reference/assets/example/pc_*.npy"))
    for f in files[:2]:
        pc = np.load(f)
        np.random.seed(3)
        want = ref.PCProcessorEval(8192, True)(pc).numpy()
        np.random.seed(3)
        sub = np.random.permutation(pc.shape[0])[:8192]
        assert np.array_equal(O.pc_norm(pc[sub]), want), f


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,npoint", [(10000, 3, 8192), (3000, 6, 1024), (9000, 3, 8192)])
def test_gpu_pc_processor_matches_oracle(N, C, npoint):
    sys.path.insert(0, os.path.join(ROOT, "vit-lens_amd"))
    from open_clip.modal_3d.processors.pc_processor import PCProcessorEval
    B = 3
    pcs = np.stack([_cloud(N, C, 10 + b) for b in range(B)])
    start = np.array([7, N - 1, N // 2], dtype=np.int64)
    proc = PCProcessorEval(npoint=npoint, uniform=True)
    out = proc.process_batch(pcs, start=start).cpu().numpy()
    assert out.shape == (B, npoint, C)
    for b in range(B):
        sel, idx = O.pc_farthest_point_sample(pcs[b], npoint, int(start[b]))
        ref = O.pc_norm(sel)
        assert np.abs(out[b] - ref).max() < 2e-5, np.abs(out[b] - ref).max()       # same points in the same order, fp32 reductions
        assert abs(np.sqrt((out[b] ** 2).sum(1)).max() - 1.0) < 1e-5
    # random-subset mode with the indices given, and the single-cloud call convention of the reference
    sub = np.stack([np.random.default_rng(b).permutation(N)[:npoint] for b in range(B)])
    out2 = PCProcessorEval(npoint=npoint, uniform=False).process_batch(pcs, subset=sub).cpu().numpy()
    assert np.abs(out2[1] - O.pc_norm(pcs[1][sub[1]])).max() < 2e-5
    np.random.seed(3)
    one = proc(pcs[0])
    np.random.seed(3)
    sel, _ = O.pc_farthest_point_sample(pcs[0], npoint, np.random.randint(0, N))
    assert one.shape == (npoint, C) and np.abs(one.cpu().numpy() - O.pc_norm(sel)).max() < 2e-5
