"""PointBERT tokenizer of the 3D Lens on the HIP kernels (PointTokenizer.forward,
open_clip/modal_3d/models/pointbert/point_encoder.py:350-362): FPS -> kNN grouping -> mini-PointNet
(Encoder, dvae.py:179-212) -> reduce_dim, plus the centre MLP positional embedding; returns x + pos.

BatchNorm runs in EVAL mode (running statistics folded into the 1x1-conv weights at load time).  Train-mode
batch statistics (and SyncBN) for the PC recipe are not implemented yet and raise.
"""
import torch

from . import ops

BF = torch.bfloat16


def _fold_bn(w, b, sd, p, eps=1e-5):
    s = sd[p + "weight"].float() / torch.sqrt(sd[p + "running_var"].float() + eps)
    return w * s[:, None], (b - sd[p + "running_mean"].float()) * s + sd[p + "bias"].float()


def _padk(w, K=64):
    out = torch.zeros(w.shape[0], K)
    out[:, :w.shape[1]] = w
    return out


class PointTokenizerEngine:
    def __init__(self, sd, a: str, lens, device, gemm_cfg=-1, bn_training=False):
        if bn_training:
            raise NotImplementedError("the inference engine folds the running statistics; train-mode BatchNorm = PointTokenizerTrainer")
        self.lens, self.device, self.cfg = lens, torch.device(device), gemm_cfg
        f = lambda k: sd[a + k].detach().float().cpu()
        bn_sd = {k: v.detach().cpu() for k, v in sd.items() if k.startswith((a + "encoder.first_conv.1.", a + "encoder.second_conv.1."))}
        w1, b1 = _fold_bn(f("encoder.first_conv.0.weight")[:, :, 0], f("encoder.first_conv.0.bias"), bn_sd, a + "encoder.first_conv.1.")
        w3, b3 = _fold_bn(f("encoder.second_conv.0.weight")[:, :, 0], f("encoder.second_conv.0.bias"), bn_sd, a + "encoder.second_conv.1.")
        half = w3.shape[1] // 2
        d = lambda t, dt=BF: t.to(device=device, dtype=dt).contiguous()
        self.w1, self.b1 = d(_padk(w1)), d(b1, torch.float32)
        self.w2, self.b2 = d(f("encoder.first_conv.3.weight")[:, :, 0]), d(f("encoder.first_conv.3.bias"), torch.float32)
        self.w3g, self.w3l, self.b3 = d(w3[:, :half]), d(w3[:, half:]), d(b3, torch.float32)
        self.w4, self.b4 = d(f("encoder.second_conv.3.weight")[:, :, 0]), d(f("encoder.second_conv.3.bias"), torch.float32)
        self.wr, self.br = d(f("reduce_dim.weight")), d(f("reduce_dim.bias"), torch.float32)
        self.wp0, self.bp0 = d(_padk(f("pos_embed.0.weight"))), d(f("pos_embed.0.bias"), torch.float32)
        self.wp2, self.bp2 = d(f("pos_embed.2.weight")), d(f("pos_embed.2.bias"), torch.float32)

    def group(self, pts: torch.Tensor, fps_start=None, want_idx=False):
        L = self.lens
        pts = pts.to(self.device).contiguous().float()
        if fps_start is None:   # misc.py:60 draws the first centre at random
            fps_start = torch.randint(0, pts.shape[1], (pts.shape[0],), device=self.device, dtype=torch.long)
        cidx, centers = ops.fps(pts, fps_start.to(self.device), L.pc_num_group)
        patches, nidx = ops.knn_group(pts, cidx, L.pc_group_size, Kp=64, want_idx=want_idx)
        return cidx, centers, patches, nidx

    def forward(self, pts: torch.Tensor, fps_start=None) -> torch.Tensor:
        """pts [B,N,3] -> tokens + pos, bf16 [B*G, trans_dim]."""
        M, c = self.lens.pc_group_size, self.cfg
        cidx, centers, patches, _ = self.group(pts, fps_start)
        h1 = ops.gemm(patches, self.w1, self.b1, act=ops.ACT_RELU, cfg=c)                    # conv 3->128 + BN + ReLU
        f = ops.gemm(h1, self.w2, self.b2, cfg=c)                                            # conv 128->256
        t = ops.gemm(ops.group_max(f, M), self.w3g, self.b3, cfg=c)                          # global half of conv 512->512
        h2 = torch.empty(f.shape[0], self.w3l.shape[0], device=self.device, dtype=BF)
        ops.gemm(f, self.w3l, None, out=h2, res=t, res_div=M, epi=ops.EPI_RES_BF16, act=ops.ACT_RELU, cfg=c)
        tok = ops.gemm(ops.group_max(ops.gemm(h2, self.w4, self.b4, cfg=c), M), self.wr, self.br, cfg=c)   # [B*G, trans]
        p1 = ops.gemm(ops.pad3(centers), self.wp0, self.bp0, act=ops.ACT_GELU, cfg=c)
        out = torch.empty_like(tok)
        ops.gemm(p1, self.wp2, self.bp2, out=out, res=tok, epi=ops.EPI_RES_BF16, cfg=c)      # pos + tokens
        return out


class PointTokenizerTrainer:
    """Trainable PointTokenizer (point_encoder.py:325-362): same dataflow as PointTokenizerEngine but with the
    BatchNorm layers kept apart from the 1x1 convolutions (train-mode batch statistics + running-stat update, or
    eval-mode running statistics), activations kept for the backward pass, and all parameter gradients
    accumulated into `grads` under the reference's parameter names (conv weights as [out, in] matrices;
    the 3-channel inputs are zero-padded to 64 for the GEMM, gradients are cut back to 3 columns).

    FPS / kNN produce indices only (misc.fps, knn_point run under no_grad-equivalent integer ops), so the
    backward stops at the gathered, centre-subtracted patches."""

    KP = 64
    BN_EPS, BN_MOMENTUM = 1e-5, 0.1           # nn.BatchNorm1d defaults (dvae.py:186,191)

    def __init__(self, sd, a: str, lens, device, grads=None, gemm_cfg=-1, bn_training=True, bn_sync=None, world_size=1):
        """bn_sync: a communicator (all_gather / all_reduce_sum, e.g. step.TorchComm) turns the two BatchNorm layers into
        SyncBatchNorm over `world_size` ranks (--use-bn-sync); None = per-rank statistics, the reference's default."""
        self.a, self.lens, self.device, self.cfg, self.bn_training = a, lens, torch.device(device), gemm_cfg, bn_training
        self.bn_sync, self.world = bn_sync, world_size
        self.grads = {} if grads is None else grads
        f32 = lambda k: sd[a + k].detach().float().to(device).contiguous().clone()      # masters never alias the caller's tensors
        m = self.masters = {}
        for k in ("encoder.first_conv.0", "encoder.first_conv.3", "encoder.second_conv.0", "encoder.second_conv.3"):
            m[a + k + ".weight"] = f32(k + ".weight")[:, :, 0].contiguous(); m[a + k + ".bias"] = f32(k + ".bias")
        for k in ("encoder.first_conv.1", "encoder.second_conv.1"):
            m[a + k + ".weight"] = f32(k + ".weight"); m[a + k + ".bias"] = f32(k + ".bias")
        for k in ("reduce_dim", "pos_embed.0", "pos_embed.2"):
            m[a + k + ".weight"] = f32(k + ".weight"); m[a + k + ".bias"] = f32(k + ".bias")
        self.running = {k: (f32(k + ".running_mean"), f32(k + ".running_var")) for k in ("encoder.first_conv.1", "encoder.second_conv.1")}
        self.op = {}
        self.refresh_operands()
        self.ctx = None

    def load_params(self, sd):
        """Masters and running statistics re-read from `sd` (reference names and shapes), in place; operands re-derived."""
        for name, m in self.masters.items():
            m.copy_(sd[name].detach().reshape(m.shape))
        for k, (rm, rv) in self.running.items():
            rm.copy_(sd[self.a + k + ".running_mean"]); rv.copy_(sd[self.a + k + ".running_var"])
        self.refresh_operands()

    # bf16 GEMM operands (forward: W, backward: W^T) derived from the f32 masters
    def refresh_operands(self):
        a, m, o = self.a, self.masters, self.op
        bf = lambda t: t.to(BF).contiguous()

        def padk(w):
            out = torch.zeros(w.shape[0], self.KP, device=w.device, dtype=BF)
            out[:, :w.shape[1]] = w.to(BF)
            return out
        w3 = m[a + "encoder.second_conv.0.weight"]
        half = w3.shape[1] // 2
        o["w1"] = padk(m[a + "encoder.first_conv.0.weight"])
        o["w2"] = bf(m[a + "encoder.first_conv.3.weight"]); o["w2T"] = bf(m[a + "encoder.first_conv.3.weight"].t())
        o["w3g"], o["w3l"] = bf(w3[:, :half]), bf(w3[:, half:])
        o["w3gT"], o["w3lT"] = bf(w3[:, :half].t()), bf(w3[:, half:].t())
        o["w4"] = bf(m[a + "encoder.second_conv.3.weight"]); o["w4T"] = bf(m[a + "encoder.second_conv.3.weight"].t())
        o["wr"] = bf(m[a + "reduce_dim.weight"]); o["wrT"] = bf(m[a + "reduce_dim.weight"].t())
        o["wp0"] = padk(m[a + "pos_embed.0.weight"])
        o["wp2"] = bf(m[a + "pos_embed.2.weight"]); o["wp2T"] = bf(m[a + "pos_embed.2.weight"].t())

    def grad_buffer(self, name):
        g = self.grads.get(name)
        if g is None:
            g = torch.zeros_like(self.masters[name]); self.grads[name] = g
        return g

    def _bn(self, z, k):
        a, m = self.a, self.masters
        total = None
        if self.bn_training and self.bn_sync is not None:
            rm, rv = self.running[k]
            local = ops.bn_stats_local(z)
            gathered = torch.empty(self.world, local.numel(), device=self.device, dtype=torch.float32)
            self.bn_sync.all_gather(gathered.view(-1), local)
            mean, var, total = ops.bn_stats_merge(gathered, rm, rv, self.BN_MOMENTUM)
        elif self.bn_training:
            rm, rv = self.running[k]
            mean, var = ops.bn_stats(z, rm, rv, self.BN_MOMENTUM)
        else:
            mean, var = self.running[k]
        h = ops.bn_apply(z, mean, var, m[a + k + ".weight"], m[a + k + ".bias"], self.BN_EPS, relu=True)
        return h, (mean, var, total)

    def _bn_bwd(self, dh, z, stats, k):
        a, m = self.a, self.masters
        mean, var, total = stats
        args = (dh, z, mean, var, m[a + k + ".weight"], m[a + k + ".bias"])
        if total is None:
            return ops.bn_bwd(*args, self.grad_buffer(a + k + ".weight"), self.grad_buffer(a + k + ".bias"), self.BN_EPS, relu=True,
                              train=self.bn_training)
        sums = ops.bn_bwd_reduce(*args, self.grad_buffer(a + k + ".weight"), self.grad_buffer(a + k + ".bias"), self.BN_EPS, relu=True)
        self.bn_sync.all_reduce_sum(sums)
        return ops.bn_bwd_apply(*args, sums, total, self.BN_EPS, relu=True)

    def forward(self, pts: torch.Tensor, fps_start=None) -> torch.Tensor:
        """pts [B,N,3] -> tokens + pos, bf16 [B*G, trans_dim]."""
        L, a, m, o, c = self.lens, self.a, self.masters, self.op, self.cfg
        M = L.pc_group_size
        pts = pts.to(self.device).contiguous().float()
        if fps_start is None:
            fps_start = torch.randint(0, pts.shape[1], (pts.shape[0],), device=self.device, dtype=torch.long)
        cidx, centers = ops.fps(pts, fps_start.to(self.device), L.pc_num_group)
        patches, _ = ops.knn_group(pts, cidx, L.pc_group_size, Kp=self.KP)
        z1 = ops.gemm(patches, o["w1"], m[a + "encoder.first_conv.0.bias"], cfg=c)
        h1, s1 = self._bn(z1, "encoder.first_conv.1")
        f = ops.gemm(h1, o["w2"], m[a + "encoder.first_conv.3.bias"], cfg=c)
        g = ops.group_max(f, M)
        t = ops.gemm(g, o["w3g"], m[a + "encoder.second_conv.0.bias"], cfg=c)
        z3 = torch.empty(f.shape[0], o["w3l"].shape[0], device=self.device, dtype=BF)
        ops.gemm(f, o["w3l"], None, out=z3, res=t, res_div=M, epi=ops.EPI_RES_BF16, cfg=c)
        h2, s3 = self._bn(z3, "encoder.second_conv.1")
        f2 = ops.gemm(h2, o["w4"], m[a + "encoder.second_conv.3.bias"], cfg=c)
        g2 = ops.group_max(f2, M)
        tok = ops.gemm(g2, o["wr"], m[a + "reduce_dim.bias"], cfg=c)
        c3 = ops.pad3(centers, self.KP)
        u = torch.empty(c3.shape[0], o["wp0"].shape[0], device=self.device, dtype=BF)
        p1 = ops.gemm(c3, o["wp0"], m[a + "pos_embed.0.bias"], act=ops.ACT_GELU_DSAVE, cfg=c, out2=u)
        out = torch.empty_like(tok)
        ops.gemm(p1, o["wp2"], m[a + "pos_embed.2.bias"], out=out, res=tok, epi=ops.EPI_RES_BF16, cfg=c)
        self.ctx = (patches, z1, s1[0], s1[1], h1, f, g, z3, s3[0], s3[1], h2, f2, g2, c3, u, p1, s1[2], s3[2])
        return out

    def _dw_into(self, dy, x, g):
        """g += dy^T x (bf16 [rows, *] operands): token-major operands for the dW kernel (narrow channel counts zero-padded to
        whole tiles, ops.gemm_dw_tn_any) - the rows are B * groups * points, two transposed copies of them per weight were
        3 % of the C5 step; the transposing NT path where the row count is not a multiple of 64."""
        if ops.gemm_dw_tn_any(dy, x, g):
            return
        rp = (dy.shape[0] + 63) // 64 * 64
        ops.gemm_dw(ops.transpose_to_bf16(dy, ldo=rp), ops.transpose_to_bf16(x, ldo=rp), g, cfg=self.cfg)

    def _dw(self, name, dy, x, cols=None):
        """grads[name] += dy^T x (both bf16 [rows, *])."""
        g = self.grad_buffer(name)
        if cols is None:
            self._dw_into(dy, x, g)
        else:       # zero-padded input channels: compute the padded product, accumulate the real columns
            full = torch.zeros(dy.shape[1], x.shape[1], device=self.device, dtype=torch.float32)
            self._dw_into(dy, x, full)
            ops.axpy(g, full[:, :cols].contiguous(), 1.0)

    def _db(self, name, dy):
        ops.colsum(dy, self.grad_buffer(name))

    def backward(self, dctx: torch.Tensor):
        """dctx f32|bf16 [B*G, trans_dim] = gradient w.r.t. the returned tokens+pos."""
        L, a, m, o, c = self.lens, self.a, self.masters, self.op, self.cfg
        M = L.pc_group_size
        patches, z1, m1, v1, h1, f, g, z3, m3, v3, h2, f2, g2, c3, u, p1, t1, t3 = self.ctx
        dout = dctx if dctx.dtype == BF else ops.cast_bf16(dctx.contiguous())
        # positional MLP: pos = W2 gelu(W0 c + b0) + b2
        self._dw(a + "pos_embed.2.weight", dout, p1); self._db(a + "pos_embed.2.bias", dout)
        du = torch.empty_like(u)
        ops.gemm(dout, o["wp2T"], None, out=du, res=u, epi=ops.EPI_DGELU, act=ops.ACT_GELU_DSAVE, cfg=c)
        self._dw(a + "pos_embed.0.weight", du, c3, cols=3); self._db(a + "pos_embed.0.bias", du)
        # token branch
        self._dw(a + "reduce_dim.weight", dout, g2); self._db(a + "reduce_dim.bias", dout)
        dg2 = ops.gemm(dout, o["wrT"], None, cfg=c)
        df2 = ops.group_max_bwd(f2, dg2, M)
        self._dw(a + "encoder.second_conv.3.weight", df2, h2); self._db(a + "encoder.second_conv.3.bias", df2)
        dh2 = ops.gemm(df2, o["w4T"], None, cfg=c)
        dz3 = self._bn_bwd(dh2, z3, (m3, v3, t3), "encoder.second_conv.1")
        dt = ops.group_sum(dz3, M)
        gw = self.grad_buffer(a + "encoder.second_conv.0.weight")
        half = gw.shape[1] // 2
        self._dw_into(dz3, f, gw[:, half:])
        self._dw_into(dt, g, gw[:, :half])
        self._db(a + "encoder.second_conv.0.bias", dt)
        dg = ops.gemm(dt, o["w3gT"], None, cfg=c)
        dfl = ops.gemm(dz3, o["w3lT"], None, cfg=c)
        df = ops.group_max_bwd(f, dg, M, base=dfl)
        self._dw(a + "encoder.first_conv.3.weight", df, h1); self._db(a + "encoder.first_conv.3.bias", df)
        dh1 = ops.gemm(df, o["w2T"], None, cfg=c)
        dz1 = self._bn_bwd(dh1, z1, (m1, v1, t1), "encoder.first_conv.1")
        self._dw(a + "encoder.first_conv.0.weight", dz1, patches, cols=3); self._db(a + "encoder.first_conv.0.bias", dz1)


class PNSATokenizerTrainer(PointTokenizerTrainer):
    """The `pnsa` point tokenizer (PointNSATokenizer, open_clip/modal_3d/models/pointnet/pointnet_util.py:345-368) on the
    HIP kernels, forward and backward: PointNetSetAbstraction (:184-227: FPS centres, ball query, centre-subtracted xyz ++
    point features, three 1x1 Conv2d + BatchNorm2d + ReLU, max over the group) then `lift` = Conv1d over
    [centre xyz ++ group feature] + LayerNorm.  forward(features [B,N,in_dim], xyz [B,N,3]) -> tokens bf16 [B*S, trans_dim]
    (no positional term: the Sample holds "x" only).

    One class serves inference and training: bn_training=False applies the running statistics and updates nothing.
    FPS / ball query produce indices only, so the backward stops at the gathered rows."""

    def __init__(self, sd, a: str, lens, device, grads=None, gemm_cfg=-1, bn_training=True, bn_sync=None, world_size=1):
        self.a, self.lens, self.device, self.cfg, self.bn_training = a, lens, torch.device(device), gemm_cfg, bn_training
        self.bn_sync, self.world = bn_sync, world_size
        self.grads = {} if grads is None else grads
        f32 = lambda k: sd[a + k].detach().float().to(device).contiguous().clone()
        m = self.masters = {}
        for i in range(3):
            w = f32(f"sa.mlp_convs.{i}.weight")
            m[a + f"sa.mlp_convs.{i}.weight"] = w.reshape(w.shape[0], -1).contiguous(); m[a + f"sa.mlp_convs.{i}.bias"] = f32(f"sa.mlp_convs.{i}.bias")
            m[a + f"sa.mlp_bns.{i}.weight"] = f32(f"sa.mlp_bns.{i}.weight"); m[a + f"sa.mlp_bns.{i}.bias"] = f32(f"sa.mlp_bns.{i}.bias")
        wl = f32("lift.0.weight")
        m[a + "lift.0.weight"] = wl.reshape(wl.shape[0], -1).contiguous(); m[a + "lift.0.bias"] = f32("lift.0.bias")
        m[a + "lift.2.weight"] = f32("lift.2.weight"); m[a + "lift.2.bias"] = f32("lift.2.bias")
        self.running = {f"sa.mlp_bns.{i}": (f32(f"sa.mlp_bns.{i}.running_mean"), f32(f"sa.mlp_bns.{i}.running_var")) for i in range(3)}
        self.in_ch = m[a + "sa.mlp_convs.0.weight"].shape[1]               # 3 + in_dim
        self.op = {}
        self.refresh_operands()
        self.ctx = None

    @staticmethod
    def _pad64(n):
        return (n + 63) // 64 * 64

    def refresh_operands(self):
        a, m, o = self.a, self.masters, self.op

        def padk(w):
            out = torch.zeros(w.shape[0], self._pad64(w.shape[1]), device=w.device, dtype=BF)
            out[:, :w.shape[1]] = w.to(BF)
            return out
        for i in range(3):
            w = m[a + f"sa.mlp_convs.{i}.weight"]
            o[f"w{i}"] = padk(w)
            if i:
                o[f"w{i}T"] = w.t().to(BF).contiguous()
        wl = m[a + "lift.0.weight"]
        o["wl"] = padk(wl)
        o["wlT_feat"] = wl[:, 3:].t().to(BF).contiguous()                    # [C', trans]: dX of the group-feature columns only

    def forward(self, features: torch.Tensor, xyz: torch.Tensor = None, fps_start=None) -> torch.Tensor:
        if xyz is None:
            raise ValueError("the pnsa tokenizer needs the point coordinates: visual(features, xyz=xyz) (pointnet_util.py:362-363)")
        L, a, m, o, c = self.lens, self.a, self.masters, self.op, self.cfg
        S, ns = L.pc_num_group, L.pc_group_size
        xyz = xyz.to(self.device).contiguous().float()
        feats = features.to(self.device).contiguous().float()
        if fps_start is None:       # farthest_point_sample draws its first centre at random (pointnet_util.py:91)
            fps_start = torch.randint(0, xyz.shape[1], (xyz.shape[0],), device=self.device, dtype=torch.long)
        cidx, centers = ops.fps(xyz, fps_start.to(self.device), S)
        patches, bidx = ops.ball_group(xyz, feats, cidx, L.pc_radius, ns, Kp=o["w0"].shape[1], want_idx=True)
        x, zs, stats = patches, [], []
        for i in range(3):
            z = ops.gemm(x, o[f"w{i}"], m[a + f"sa.mlp_convs.{i}.bias"], cfg=c)
            h, st = self._bn(z, f"sa.mlp_bns.{i}")
            zs.append((x, z, h)); stats.append(st)
            x = h
        feat = ops.group_max(x, ns)                                           # [B*S, C']
        Kl = o["wl"].shape[1]
        lift_in = torch.zeros(feat.shape[0], Kl, device=self.device, dtype=BF)
        lift_in[:, :3] = centers.reshape(-1, 3)
        lift_in[:, 3:3 + feat.shape[1]] = feat
        y = ops.gemm(lift_in, o["wl"], m[a + "lift.0.bias"], cfg=c)           # Conv1d(C' + 3 -> trans)
        R, Tr = y.shape
        mean = torch.empty(R, device=self.device, dtype=torch.float32); rstd = torch.empty_like(mean)
        tok = torch.empty(R, Tr, device=self.device, dtype=BF)
        ops.layernorm(y, m[a + "lift.2.weight"], m[a + "lift.2.bias"], tok, R, Tr, mean=mean, rstd=rstd)
        self.ctx = (zs, stats, feat, lift_in, y, mean, rstd)
        self.last_idx = (cidx, bidx)
        return tok

    def backward(self, dctx: torch.Tensor):
        """dctx f32|bf16 [B*S, trans_dim] = gradient w.r.t. the returned tokens."""
        L, a, m, o, c = self.lens, self.a, self.masters, self.op, self.cfg
        ns = L.pc_group_size
        zs, stats, feat, lift_in, y, mean, rstd = self.ctx
        R, Tr = y.shape
        dtok = dctx.contiguous()
        ops.layernorm_bwd_params(dtok, y, mean, rstd, self.grad_buffer(a + "lift.2.weight"), self.grad_buffer(a + "lift.2.bias"), R, Tr)
        dy = torch.empty(R, Tr, device=self.device, dtype=BF)
        ops.layernorm_bwd(dtok, y, mean, rstd, m[a + "lift.2.weight"], R, Tr, dx=dy)
        self._dw(a + "lift.0.weight", dy, lift_in, cols=m[a + "lift.0.weight"].shape[1]); self._db(a + "lift.0.bias", dy)
        dfeat = ops.gemm(dy, o["wlT_feat"], None, cfg=c)                      # [B*S, C']
        dh = ops.group_max_bwd(zs[2][2], dfeat, ns)
        for i in (2, 1, 0):
            x, z, _ = zs[i]
            dz = self._bn_bwd(dh, z, stats[i], f"sa.mlp_bns.{i}")
            self._dw(a + f"sa.mlp_convs.{i}.weight", dz, x, cols=self.in_ch if i == 0 else None)
            self._db(a + f"sa.mlp_convs.{i}.bias", dz)
            if i:
                dh = ops.gemm(dz, o[f"w{i}T"], None, cfg=c)
