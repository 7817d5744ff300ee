"""CPU, build container only: the oracle's restatement of the `pnsa` point tokenizer (PointNet set abstraction,
open_clip/modal_3d/models/pointnet/pointnet_util.py:345-368) against the imported reference class.

The reference module imports `dgl.geometry` and `torch_redstone` at module level; neither is installed.  For THIS test
the two names are provided as empty shells: with `dgl.geometry.farthest_point_sampler` missing the reference's own
`farthest_point_sample` takes its documented fallback (pointnet_util.py:83-98, a torch FPS whose start index comes from
`torch.randint` - seeded here and replayed into the oracle), and `rst.Lambda` is the one-line module wrapper it is in
torch_redstone.  Nothing of this is product code: the product has no pnsa tokenizer yet; this pins
the oracle the kernel will be built against."""
import os
from types import SimpleNamespace

import pytest
import torch

import vitlens_oracle as O

REF = "/root/reference/vitlens/src/open_clip/modal_3d/models/pointnet/pointnet_util.py"


def _reference_module():
    import ref_loader
    return ref_loader.load_pointnet_util()


@pytest.mark.needs_reference
@pytest.mark.parametrize("training", [False, True])
def test_pnsa_tokens_equal_the_reference(training):
    if not os.path.exists(REF):
        pytest.skip("reference tree not present")
    ref = _reference_module()
    cfg = SimpleNamespace(num_group=24, radius=0.35, group_size=8, in_dim=3, encoder_dims=32, trans_dim=48)
    torch.manual_seed(0)
    tok = ref.PointNSATokenizer(cfg)
    with torch.no_grad():
        for bn in tok.sa.mlp_bns:                                                 # non-trivial running statistics for eval mode
            bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
            bn.weight.normal_(1, 0.1); bn.bias.normal_(0, 0.1)
    tok.train(training)
    g = torch.Generator().manual_seed(1)
    B, N = 3, 400
    xyz = torch.rand(B, N, 3, generator=g) * 2 - 1
    feats = torch.rand(B, N, 3, generator=g)
    sd = {"a." + k: v.detach().clone() for k, v in tok.state_dict().items()}
    torch.manual_seed(7)
    start = torch.randint(0, N, (B,), dtype=torch.long)                           # what the fallback FPS draws first
    torch.manual_seed(7)
    want = tok(feats, xyz=xyz)["x"]
    got, cidx, bidx = O.pnsa_tokens(sd, "a.", feats, xyz, cfg.num_group, cfg.radius, cfg.group_size, start, training=training)
    assert got.shape == want.shape == (B, cfg.num_group, cfg.trans_dim)
    assert float((got - want).abs().max()) < 2e-5, float((got - want).abs().max())
    # the pieces, individually: FPS picks, ball query (incl. short groups filled with the first index)
    torch.manual_seed(7)
    assert torch.equal(cidx, ref.farthest_point_sample(xyz, cfg.num_group))
    new_xyz = ref.index_points(xyz, cidx)
    assert torch.equal(bidx, ref.query_ball_point(cfg.radius, cfg.group_size, xyz, new_xyz))
    tight = ref.query_ball_point(0.05, 8, xyz, new_xyz)
    assert torch.equal(O.ball_query_indices(0.05, 8, xyz, new_xyz), tight)
    assert bool((tight == tight[..., :1]).all(-1).any())                          # some groups hold only their centre


def test_pnsa_oracle_against_the_committed_golden():
    """tests/golden/tiny_pnsa.npz (generated from the reference by oracle/gen_golden.py --only pnsa 31): tokens in eval and
    train mode, FPS / ball-query indices bit-exact, and every parameter gradient of a random upstream gradient through
    the oracle's autograd.  This is the fixture the HIP pnsa tokenizer will be checked against on the GPU box."""
    import json
    import numpy as np
    from golden_util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "tiny_pnsa.npz"))
    cfg = json.loads(str(z["meta"]))["cfg"]
    sd = {"a." + k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd/")}
    xyz, feats, start = torch.tensor(z["in/xyz"]), torch.tensor(z["in/features"]), torch.tensor(z["in/fps_start"])
    args = (cfg["num_group"], cfg["radius"], cfg["group_size"], start)
    ev, cidx, bidx = O.pnsa_tokens(sd, "a.", feats, xyz, *args, training=False)
    assert torch.equal(cidx, torch.tensor(z["out/fps_idx"])) and torch.equal(bidx, torch.tensor(z["out/ball_idx"]))
    assert float((ev - torch.tensor(z["out/eval/tokens"])).abs().max()) < 2e-5
    sdr = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    tr, _, _ = O.pnsa_tokens(sdr, "a.", feats, xyz, *args, training=True)
    assert float((tr.detach() - torch.tensor(z["out/train/tokens"])).abs().max()) < 2e-5
    tr.backward(torch.tensor(z["in/dctx"]))
    n = 0
    for k in z.files:
        if k.startswith("grad/"):
            g, want = sdr["a." + k[5:]].grad, torch.tensor(z[k])
            if "mlp_convs" in k and k.endswith(".bias"):
                # a per-channel constant in front of a train-mode BatchNorm is removed by the batch mean: identically zero
                # gradient, both sides hold round-off only
                wn = float(torch.tensor(z[k[:-4] + "weight"]).norm())
                assert float(want.norm()) < 1e-3 * wn and float(g.norm()) < 1e-3 * wn, k
            else:
                assert float((g - want).norm() / (want.norm() + 1e-12)) < 1e-4, k
            n += 1
    assert n == 16
