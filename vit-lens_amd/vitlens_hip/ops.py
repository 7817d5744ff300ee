"""Tensor-level wrappers over the C ABI.  Every function enqueues HIP kernels on the current
torch stream and returns immediately.  Inputs must live on a GPU: there is no CPU fallback."""
import ctypes as C
import threading

import torch

from ._lib import load_library, check

F32, BF16, F16 = 0, 1, 2
EPI_BF16, EPI_F32, EPI_RES_F32, EPI_RES_BF16, EPI_GEGLU, EPI_DGELU, EPI_DGEGLU = 0, 1, 2, 3, 5, 6, 7
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
ACT_GELU_DSAVE = 4          # forward: out = gelu(pre), out2 = gelu'(pre);  with EPI_DGELU: res is that gelu' tensor
LOG2E = 1.4426950408889634

_lib = load_library()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("vitlens_hip ops need GPU tensors (no CPU fallback)")
    return C.c_void_p(t.data_ptr())


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _chk2d(t, name, dtype=None):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a row-major 2-D tensor, got shape {tuple(t.shape)} strides {t.stride()}")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def gemm(a, w, bias=None, out=None, res=None, epi=EPI_BF16, act=ACT_NONE, alpha=1.0, cfg=-1, out2=None, res_div=1):
    """out = epilogue(a @ w.T).  a [M,K] bf16, w [N,K] bf16.  out2: optional pre-activation copy (bf16; with
    act=ACT_GELU_DSAVE: gelu'(pre) instead, the operand of the backward's EPI_DGELU launched with the same act)."""
    if act == ACT_GELU_DSAVE and epi == EPI_BF16 and out2 is None:
        raise ValueError("gemm: ACT_GELU_DSAVE needs out2")
    _chk2d(a, "a", torch.bfloat16); _chk2d(w, "w", torch.bfloat16)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm: K mismatch {a.shape} vs {w.shape}")
    if K % 64:   # kernels stream 64-wide K slabs: zero-pad odd reduction lengths (toy configs only)
        Kp = (K + 63) // 64 * 64
        a = torch.nn.functional.pad(a, (0, Kp - K)); w = torch.nn.functional.pad(w, (0, Kp - K))
        K = Kp
    if out is None:
        n_out = N // 2 if epi == EPI_GEGLU else (2 * N if epi == EPI_DGEGLU else N)
        odt = torch.float32 if epi in (EPI_F32, EPI_RES_F32) else torch.bfloat16
        out = torch.empty(M, n_out, device=a.device, dtype=odt)
    _chk2d(out, "out")
    if res is not None:
        _chk2d(res, "res")
        if res.stride(0) != out.stride(0) or res.dtype != out.dtype:
            raise ValueError("gemm: residual must share out's row stride and dtype")
        if res_div > 1 and res.shape[0] * res_div < M:
            raise ValueError("gemm: broadcast residual has too few rows")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N):
        raise ValueError("gemm: bias must be f32 [N]")
    if out2 is not None and (out2.stride(0) != out.stride(0) * (2 if epi == EPI_GEGLU else 1) or out2.dtype != torch.bfloat16):
        raise ValueError("gemm: out2 must be bf16 with out's row stride (twice that for GEGLU)")
    check(_lib.vl_gemm_bf16_ex(_p(a), _p(w), _p(bias), _p(out), _p(res), _p(out2), M, N, K, a.stride(0), w.stride(0),
                               out.stride(0), float(alpha), epi, act, res_div, cfg, _stream()))
    return out


def gemm_f16(a, w, bias=None, out=None, res=None, epi=EPI_BF16, act=ACT_NONE, alpha=1.0):
    """The persistent 256x256 GEMM on IEEE-half operands (vl_gemm_f16; the frozen text tower): a [M,K], w [N,K] fp16;
    epi = EPI_BF16 -> out fp16 [M,N] = act(a @ w.T + bias), epi = EPI_RES_F32 -> out f32 = res + a @ w.T + bias (in place
    allowed).  M, N multiples of 256, K a multiple of 64 and >= 512: the caller pads (TextEngine keeps whole row tiles)."""
    _chk2d(a, "a", torch.float16); _chk2d(w, "w", torch.float16)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm_f16: K mismatch {a.shape} vs {w.shape}")
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if epi == EPI_RES_F32 else torch.float16)
    _chk2d(out, "out", torch.float32 if epi == EPI_RES_F32 else torch.float16)
    if res is not None:
        _chk2d(res, "res", torch.float32)
        if res.stride(0) != out.stride(0):
            raise ValueError("gemm_f16: residual must share out's row stride")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N):
        raise ValueError("gemm_f16: bias must be f32 [N]")
    check(_lib.vl_gemm_f16(_p(a), _p(w), _p(bias), _p(out), _p(res), M, N, K, a.stride(0), w.stride(0), out.stride(0),
                           float(alpha), epi, act, _stream()))
    return out


# ---- true fp32 arithmetic (inference under precision="fp32": vitlens_hip/f32.py) ----
def gemm_f32(a, w, bias=None, out=None, res=None, act=ACT_NONE, alpha=1.0):
    """out f32 = act(alpha * a @ w.T + bias) (+ res), a [M,K], w [N,K] f32, any M and N, K % 4 == 0 (vl_gemm_f32: fp32-input MFMA)."""
    _chk2d(a, "a", torch.float32); _chk2d(w, "w", torch.float32)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm_f32: K mismatch {a.shape} vs {w.shape}")
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    _chk2d(out, "out", torch.float32)
    if res is not None:
        _chk2d(res, "res", torch.float32)
        if res.stride(0) != out.stride(0):
            raise ValueError("gemm_f32: residual must share out's row stride")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N):
        raise ValueError("gemm_f32: bias must be f32 [N]")
    check(_lib.vl_gemm_f32(_p(a), _p(w), _p(bias), _p(out), _p(res), M, N, K, a.stride(0), w.stride(0), out.stride(0),
                           float(alpha), act, _stream()))
    return out


def attn_fwd_f32(q, k, v, out, lse=None, causal=False, scale=1.0):
    """softmax(scale * q k^T [+ causal mask]) v on strided f32 [B,H,L,dh] views -> out f32 [B*Lq, H*dh] (dh = 32 or 64)."""
    import ctypes
    B, H, Lq, dh = q.shape
    Lk = k.shape[2]
    vals = []
    for t in (q, k, v):
        if t.dim() != 4 or t.stride(3) != 1 or t.dtype != torch.float32:
            raise ValueError("attn_fwd_f32: operands must be f32 [B,H,L,dh] views with unit last stride")
        vals += [t.stride(0), t.stride(1), t.stride(2)]
    if out.dtype != torch.float32:
        raise TypeError("attn_fwd_f32: out must be f32")
    check(_lib.vl_attn_fwd_f32(_p(q), _p(k), _p(v), (ctypes.c_long * 9)(*vals), _p(out), _p(lse), B, H, Lq, Lk, dh, float(scale),
                               1 if causal else 0, _stream()))
    return out


def im2col_f32(x, kh, kw, sh, sw, Kp, transpose_hw=False):
    """im2col with f32 patches (the conv stem under precision="fp32")."""
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("im2col_f32: need contiguous f32 input")
    N, Cc = x.shape[0], x.shape[1]
    H, W = (x.shape[3], x.shape[2]) if transpose_hw else (x.shape[2], x.shape[3])
    gh, gw = (H - kh) // sh + 1, (W - kw) // sw + 1
    out = torch.empty(N * gh * gw, Kp, device=x.device, dtype=torch.float32)
    check(_lib.vl_im2col_f32(_p(x), _p(out), N, Cc, H, W, kh, kw, sh, sw, Kp, 1 if transpose_hw else 0, _stream()))
    return out, gh, gw


# ---- LayerNorm folded into the GEMMs either side of it (frozen pre-LN blocks, bf16 residual stream; include/vitlens_hip.h) ----
def fold_ln_linear(w, b, gamma, beta):
    """Operands of  LN(x; gamma, beta) @ w.T + b  for vl_gemm_lnfold_bf16: (Wg bf16 [N,K] = bf16(w * gamma), bias_f f32 [N] =
    b + w @ beta, c f32 [N] = row sums of the ROUNDED Wg - the mean term must cancel against exactly what the MFMAs add up)."""
    w = w.detach().float()
    wg = (w * gamma.detach().float()[None, :]).to(torch.bfloat16).contiguous()
    c = wg.float().sum(1).contiguous()
    d = (b.detach().float() + (w * beta.detach().float()[None, :]).sum(1)).contiguous()       # (w @ beta without a BLAS call: build-time only)
    return wg, d, c


def _fold_rows(x, out, N, K):
    """Rows of this problem that take the folded path (0: the shape does not fit the persistent kernel)."""
    M = x.shape[0]
    if x.dtype != torch.bfloat16 or out.dtype != torch.bfloat16 or K < 512 or K % 64 or N % 256 or x.stride(0) % 8 or out.stride(0) % 8:
        return 0
    if x.data_ptr() % 16 or out.data_ptr() % 16:
        return 0
    return int(_lib.vl_gemm_main_rows(M, N))


def _leftover_cfg(rows, N, cfg):
    """Kernel of the leftover rows of a row-split GEMM, as vl_gemm.hip's run_gemm picks it (64x64 LDS-DMA tiles when they fill
    most of the chip, else the split-K tail kernel); a standalone call of that size would get 128x128 tiles and a serial K."""
    if cfg >= 0 or rows > 512:
        return cfg
    return 11 if ((rows + 63) // 64) * ((N + 63) // 64) >= 192 else 9


def gemm_lnfold(x, fold, mean, rstd, out, w, b, ln_w, ln_b, h_ws, act=ACT_NONE, out2=None, eps=1e-5, cfg=-1, h_ready=False):
    """out = act(LN(x) @ w.T + b) with the LayerNorm folded into the GEMM's epilogue for the rows the persistent kernel takes
    (x raw bf16 rows, mean / rstd their statistics, fold = fold_ln_linear(...)); the leftover rows - and every row of a shape
    that kernel refuses - go through layernorm + gemm on the plain operands (h_ws: a bf16 [>= leftover rows, K] buffer;
    h_ready: ln_row_stats(..., h_left=h_ws, h_row0=fold_rows(...)) already left their LayerNorm output there)."""
    wg, d, c = fold
    M, K = x.shape
    N = wg.shape[0]
    mm = _fold_rows(x, out, N, K)
    if mm:
        check(_lib.vl_gemm_lnfold_bf16(_p(x), _p(wg), _p(d), _p(c), _p(mean), _p(rstd), _p(out), _p(out2), mm, N, K,
                                       x.stride(0), wg.stride(0), out.stride(0), act, _stream()))
    if mm < M:
        r = M - mm
        h = h_ws[:r]
        if not h_ready:
            layernorm(x[mm:], ln_w, ln_b, h, r, K, x_row_stride=x.stride(0), mean=mean[mm:], rstd=rstd[mm:], eps=eps)
        gemm(h, w, b, out=out[mm:], epi=EPI_BF16, act=act, cfg=_leftover_cfg(r, N, cfg) if mm else cfg,
             out2=None if out2 is None else out2[mm:])
    return out


def gemm_res_rowstats(a, w, b, out, res, part, cfg=-1):
    """out = res + a @ w.T + b (bf16 residual stream, in place allowed); for the rows the persistent kernel takes it also leaves
    (sum, sum of squares) per row and 64-column slice in part f32 [rows, N/64, 2].  Returns that row count (0: none)."""
    M, K = a.shape
    N = w.shape[0]
    mm = _fold_rows(a, out, N, K) if (res.dtype == torch.bfloat16 and res.stride(0) == out.stride(0) and res.data_ptr() % 16 == 0
                                      and part.numel() >= M * (N // 64) * 2) else 0
    if mm:
        check(_lib.vl_gemm_res_rowstats_bf16(_p(a), _p(w), _p(b), _p(out), _p(res), _p(part), mm, N, K, a.stride(0), w.stride(0),
                                             out.stride(0), _stream()))
    if mm < M:
        gemm(a[mm:], w, b, out=out[mm:], res=res[mm:], epi=EPI_RES_BF16, cfg=_leftover_cfg(M - mm, N, cfg) if mm else cfg)
    return mm


def fold_rows(x, out, N):
    """Rows of `out = f(LN(x) @ W[N, K].T)` that gemm_lnfold runs through the folded epilogue (the others: layernorm + gemm)."""
    return _fold_rows(x, out, N, x.shape[1])


def ln_row_stats(part, x, m_main, mean, rstd, eps=1e-5, ln_w=None, ln_b=None, h_left=None, h_row0=None):
    """mean / rstd of the rows of x (bf16 [rows, D]): rows < m_main from the partial sums `part` of gemm_res_rowstats, the
    others from the rows themselves.  h_left (bf16 [>= rows - h_row0, D], with ln_w / ln_b): the LayerNorm output of the rows
    >= h_row0 (the consuming gemm_lnfold's leftover rows, h_row0 = fold_rows(...) >= m_main) from the same launch."""
    rows, D = x.shape
    if h_left is not None:
        if h_row0 is None or h_row0 < m_main or h_left.dtype != torch.bfloat16 or h_left.shape[0] < rows - h_row0:
            raise ValueError("ln_row_stats: h_left needs h_row0 >= m_main and a bf16 buffer of at least rows - h_row0 rows")
    check(_lib.vl_ln_row_stats(_p(part) if m_main else None, D // 64, _p(x), x.stride(0), D, m_main, rows, float(eps), _p(mean),
                               _p(rstd), _p(ln_w), _p(ln_b), _p(h_left), h_left.stride(0) if h_left is not None else 0,
                               int(h_row0 or 0), _stream()))


def logits_gemm(xb, yb, scale):
    """scale * xb @ yb^T as f32 [R, C] for ANY number of columns (a partial last batch, a batch of 6 ...): the GEMM wants
    N % 4 == 0, so the column operand is zero-padded to a multiple of 4 rows and a [R, C] view of the padded result is
    returned (the CE kernels take the row stride)."""
    Cn = yb.shape[0]
    Cp = (Cn + 3) // 4 * 4
    if Cp != Cn:
        yb = torch.cat([yb, torch.zeros(Cp - Cn, yb.shape[1], device=yb.device, dtype=yb.dtype)], 0)
    out = gemm(xb, yb, None, epi=EPI_F32, alpha=scale)
    return out if Cp == Cn else out[:, :Cn]


def gemm_dw(dyt, xt, g, cfg=-1, alpha=1.0):
    """g += dyt @ xt^T: the weight-gradient GEMM (dyt [N_out, R], xt [K_in, R] bf16 - both already transposed so
    that the token axis R is the reduction axis; g f32 [N_out, K_in], may be a strided view).  Few output tiles
    and a long reduction -> split-K over the CUs; otherwise the regular GEMM with the accumulate epilogue."""
    _chk2d(dyt, "dyt", torch.bfloat16); _chk2d(xt, "xt", torch.bfloat16); _chk2d(g, "g", torch.float32)
    M, K = dyt.shape
    N = xt.shape[0]
    nk = K // 64
    if K % 64 == 0 and M % 256 == 0 and N % 256 == 0 and nk >= 64:
        # whole 256x256 tiles: k-slices on the persistent kernel (fp32 partials + fixed-order reduce)
        t256 = (M // 256) * (N // 256)
        for splits in (1, 2, 4, 8, 16):
            if nk % splits == 0 and t256 * splits >= 192 and nk // splits >= 16:
                ws = torch.empty(splits * M * N, device=g.device, dtype=torch.float32)
                check(_lib.vl_gemm_splitk_accum_f32(_p(dyt), _p(xt), _p(g), M, N, K, dyt.stride(0), xt.stride(0), g.stride(0),
                                                    float(alpha), splits, _p(ws), _stream()))
                return g
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if K % 64 == 0 and tiles * 2 <= 256 and nk >= 32:
        splits = max(1, min(512 // tiles, nk // 16))
        if splits > 1:
            ws = torch.empty(splits * M * N, device=g.device, dtype=torch.float32)
            check(_lib.vl_gemm_splitk_accum_f32(_p(dyt), _p(xt), _p(g), M, N, K, dyt.stride(0), xt.stride(0), g.stride(0),
                                                float(alpha), splits, _p(ws), _stream()))
            return g
    return gemm(dyt, xt, None, out=g, res=g, epi=EPI_RES_F32, cfg=cfg, alpha=alpha)


def gemm_dw_tn(dy, x, g, alpha=1.0):
    """g[N_out, K_in] += dy^T x on the operands as the backward holds them (dy [R, N_out], x [R, K_in] bf16, row = token):
    no transposed copies - the kernel stages token-major k-slabs and reads its fragments with the LDS transpose read.
    Returns False (nothing launched) when the shape does not fit (whole 256x256 tiles, R % 64 == 0, enough work items)."""
    if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or g.dtype != torch.float32 or dy.dim() != 2 or x.dim() != 2:
        return False
    R, M = dy.shape
    N = x.shape[1]
    if x.shape[0] != R or R % 64 or M % 256 or N % 256 or dy.stride(1) != 1 or x.stride(1) != 1 or g.stride(1) != 1:
        return False
    if dy.stride(0) % 8 or x.stride(0) % 8 or g.stride(0) % 4 or dy.data_ptr() % 16 or x.data_ptr() % 16 or tuple(g.shape) != (M, N):
        return False
    splits = tn_splits(R // 64, (M // 256) * (N // 256))
    if not splits:
        return False
    ws = torch.empty(splits * M * N, device=g.device, dtype=torch.float32)
    check(_lib.vl_gemm_tn_splitk_accum_f32(_p(dy), _p(x), _p(g), M, N, R, dy.stride(0), x.stride(0), g.stride(0),
                                           float(alpha), splits, _p(ws), _stream()))
    return True


def tn_splits(nk: int, tiles: int, few_tiles_ok: bool = False) -> int:
    """K slices for the token-major dW kernel: the smallest of 1, 2, 4, 8, 16 that gives >= 192 work items (tiles x slices) with
    >= 16 steps of 64 tokens per slice; slices are ceil(nk / splits) steps long, the last one may be shorter but not below the
    4 steps the kernel's DMA look-ahead needs (vl_gemm_tn_splitk_accum_f32 checks the same).  0 = the shape does not fit.
    few_tiles_ok (round 6: the Perceiver's narrow projections, 1-8 output tiles over 32 k-512 k tokens): up to 64 slices, and when
    even those do not fill the chip, the largest admissible count."""
    valid = lambda splits: (nk // splits >= 16 and nk - (-(-nk // (-(-nk // splits))) - 1) * (-(-nk // splits)) >= 4)
    cands = (1, 2, 4, 8, 16, 32, 64) if few_tiles_ok else (1, 2, 4, 8, 16)
    best = 0
    for splits in cands:
        if not valid(splits):
            continue
        if tiles * splits >= 192:
            return splits
        best = splits
    return best if few_tiles_ok else 0


_pad_cache = {}


def _zero_padded(t, cols, tag):
    """t [R, c] bf16 -> a [R, cols] bf16 buffer (cached per shape and role) whose first c columns are t and the rest zero."""
    # (the source width is part of the key: the pad columns must stay zero; the stream too: two backwards run side by side;
    #  the host thread too: a buffer is filled and consumed by two separate enqueues)
    key = (tag, t.device, torch.cuda.current_stream(t.device).cuda_stream, threading.get_ident(), t.shape[0], t.shape[1], cols)
    buf = _pad_cache.get(key)
    if buf is None:
        buf = torch.zeros(t.shape[0], cols, device=t.device, dtype=torch.bfloat16)      # the pad columns are written once: zeros
        _pad_cache[key] = buf
    buf[:, :t.shape[1]].copy_(t)
    return buf


def gemm_dw_tn_any(dy, x, g, alpha=1.0):
    """g[N_out, K_in] += dy^T x like gemm_dw_tn, for operands whose column counts are NOT multiples of 256 (the Perceiver's
    cross-attention projections: 64 / 128 columns; a 384-wide context): the narrow operand is zero-padded to whole 256-column
    tiles - a copy of the SMALL matrix - instead of transposing BOTH operands for the NT kernel (a 64-134 MB round trip of the
    large one per call, `transpose64_kernel` in the C4 / C5 kernel statistics of round 5).  Returns False when the shape does
    not fit (rows not a multiple of 64, non-bf16 operands)."""
    if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or g.dtype != torch.float32 or dy.dim() != 2 or x.dim() != 2:
        return False
    R, M = dy.shape
    N = x.shape[1]
    if x.shape[0] != R or R % 64 or tuple(g.shape) != (M, N) or dy.stride(1) != 1 or x.stride(1) != 1:
        return False
    Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
    if Mp == M and Np == N and gemm_dw_tn(dy, x, g, alpha):
        return True
    if dy.stride(0) % 8 or x.stride(0) % 8 or dy.data_ptr() % 16 or x.data_ptr() % 16:
        return False
    splits = tn_splits(R // 64, (Mp // 256) * (Np // 256), few_tiles_ok=True)
    if not splits:
        return False
    dyp = dy if Mp == M else _zero_padded(dy, Mp, "dy")
    xp = x if Np == N else _zero_padded(x, Np, "x")
    direct = Mp == M and Np == N and g.stride(1) == 1 and g.stride(0) % 4 == 0
    gp = g if direct else torch.zeros(Mp, Np, device=g.device, dtype=torch.float32)
    ws = torch.empty(splits * Mp * Np, device=g.device, dtype=torch.float32)
    check(_lib.vl_gemm_tn_splitk_accum_f32(_p(dyp), _p(xp), _p(gp), Mp, Np, R, dyp.stride(0), xp.stride(0), gp.stride(0),
                                           float(alpha), splits, _p(ws), _stream()))
    if not direct:
        g += gp[:M, :N]
    return True


def _bhld_strides(*views):
    """(batch, head, row) element strides of [B,H,L,dh] views (last stride 1) as a ctypes long array."""
    import ctypes
    vals = []
    for t in views:
        if t.dim() != 4 or t.stride(3) != 1 or t.dtype != views[0].dtype or t.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("attention operands must be bf16 (or, forward only, fp16) [B,H,L,dh] views of one dtype with unit last stride")
        vals += [t.stride(0), t.stride(1), t.stride(2)]
    return (ctypes.c_long * len(vals))(*vals)


def heads_view(x2d, B, L, H, dh, col0=0):
    """[B*L, W] token-major matrix -> the [B,H,L,dh] view of its column block [col0, col0 + H*dh) (no copy)."""
    W = x2d.stride(0)
    return x2d.as_strided((B, H, L, dh), (L * W, dh, W, 1), x2d.storage_offset() + col0)


def attn_fwd(q, k, v, out, lse=None, causal=False, qscale=1.0):
    """q [B,H,Lq,dh], k, v [B,H,Lk,dh] strided bf16 views (see heads_view) -> out [B*Lq, H*dh] bf16.
    q is multiplied by qscale (softmax_scale*log2e) inside the kernel; pass 1 for a pre-scaled q."""
    B, H, Lq, dh = q.shape
    Lk = k.shape[2]
    st = _bhld_strides(q, k, v)
    if q.dtype == torch.float16:          # the frozen text tower's operands (head dim 64, <= 288 keys)
        if k.dtype != q.dtype or v.dtype != q.dtype or out.dtype != q.dtype:
            raise TypeError("attn_fwd: fp16 q needs fp16 k, v and out")
        check(_lib.vl_attn_fwd_f16(_p(q), _p(k), _p(v), st, _p(out), _p(lse), B, H, Lq, Lk, dh, float(qscale),
                                   1 if causal else 0, _stream()))
        return out
    check(_lib.vl_attn_fwd_bf16(_p(q), _p(k), _p(v), st, _p(out), _p(lse), B, H, Lq, Lk, dh, float(qscale),
                                1 if causal else 0, _stream()))
    return out


def layernorm(x, w, b, out, rows, D, x_row_stride=None, row_index=None, row_mul=0, mean=None, rstd=None,
              eps=1e-5):
    xs = D if x_row_stride is None else x_row_stride
    check(_lib.vl_layernorm_fwd(_p(x), _dt(x), xs, _p(row_index), row_mul, _p(w), _p(b), _p(out), _dt(out),
                                out.stride(-2) if out.dim() >= 2 else D, _p(mean), _p(rstd), rows, D,
                                float(eps), _stream()))
    return out


def assemble_ln_pre(tokens, cls, pos, pos2, w, b, out, B, T, D, eps=1e-5, xpre=None, mean=None, rstd=None):
    check(_lib.vl_assemble_ln_pre(_p(tokens), _dt(tokens), _p(cls), _p(pos), _p(pos2), _p(w), _p(b), _p(out),
                                  _dt(out), _p(xpre), _p(mean), _p(rstd), B, T, D, float(eps), _stream()))
    return out


def l2_normalize(x, out=None, out_bf16=None, norms=None, eps=1e-12):
    _chk2d(x, "x", torch.float32)
    if out is None:
        out = torch.empty_like(x)
    check(_lib.vl_l2_normalize(_p(x), _p(out), _p(out_bf16), _p(norms), x.shape[0], x.shape[1], float(eps),
                               _stream()))
    return out


def l2_normalize_bwd(f, df, norms, eps=1e-12):
    dx = torch.empty_like(f)
    check(_lib.vl_l2_normalize_bwd(_p(f), _p(df), _p(norms), _p(dx), f.shape[0], f.shape[1], float(eps),
                                   _stream()))
    return dx


def im2col(x, kh, kw, sh, sw, Kp, transpose_hw=False, out=None):
    """x [N,C,H,W] f32 (or [N,C,W,H] stored when transpose_hw) -> bf16 [N*gh*gw, Kp]."""
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("im2col: need contiguous f32 input")
    N, Cc = x.shape[0], x.shape[1]
    H, W = (x.shape[3], x.shape[2]) if transpose_hw else (x.shape[2], x.shape[3])
    gh, gw = (H - kh) // sh + 1, (W - kw) // sw + 1
    if out is None:
        out = torch.empty(N * gh * gw, Kp, device=x.device, dtype=torch.bfloat16)
    check(_lib.vl_im2col_bf16(_p(x), _p(out), N, Cc, H, W, kh, kw, sh, sw, Kp, 1 if transpose_hw else 0,
                              _stream()))
    return out, gh, gw


def text_embed(ids, tok_emb, pos, out):
    B, L = ids.shape
    check(_lib.vl_text_embed(_p(ids), _p(tok_emb), _p(pos), _p(out), _dt(out), B, L, tok_emb.shape[1],
                             tok_emb.shape[0], _stream()))
    return out


def cast_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(_lib.vl_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()))
    return out


def add_rows(x, table, out, rows, T, D):
    check(_lib.vl_add_rows(_p(x), _dt(x), _p(table), _p(out), _dt(out), rows, T, D, _stream()))
    return out


def transpose_to_bf16(x, ldo=None, out=None):
    """x [R,C] (f32|bf16, row-major) -> bf16 [C, ldo] with zero-filled tail columns."""
    _chk2d(x, "x")
    R, Cc = x.shape
    ldo = R if ldo is None else ldo
    if out is None:
        out = torch.empty(Cc, ldo, device=x.device, dtype=torch.bfloat16)
    check(_lib.vl_transpose_to_bf16(_p(x), _dt(x), x.stride(0), R, Cc, _p(out), ldo, _stream()))
    return out


def transpose_colsum(x, ldo, colsum_out=None, scale=1.0, out=None):
    """x [R,C] -> bf16 [C, ldo] (as transpose_to_bf16) and, fused, colsum_out[c] += scale * sum_r x[r, c] (bias gradient).
    Falls back to the separate kernels when the shape does not fit the fused one."""
    _chk2d(x, "x")
    R, Cc = x.shape
    ok = Cc % 64 == 0 and ldo % 8 == 0 and x.stride(0) % 8 == 0 and R >= 256 and x.data_ptr() % 16 == 0
    if not ok:
        out = transpose_to_bf16(x, ldo=ldo, out=out)
        if colsum_out is not None:
            colsum(x, colsum_out, scale)
        return out
    if out is None:
        out = torch.empty(Cc, ldo, device=x.device, dtype=torch.bfloat16)
    ws = _ws_for(x.device, ldo, Cc, 1) if colsum_out is not None else None
    check(_lib.vl_transpose_colsum_bf16(_p(x), _dt(x), x.stride(0), R, Cc, _p(out), ldo, _p(colsum_out), float(scale), _p(ws),
                                        _stream()))
    return out


def split_bf16x3(x, pattern):
    _chk2d(x, "x", torch.float32)
    out = torch.empty(x.shape[0], 3 * x.shape[1], device=x.device, dtype=torch.bfloat16)
    check(_lib.vl_split_bf16x3(_p(x.contiguous()), _p(out), x.shape[0], x.shape[1], pattern, _stream()))
    return out


def ce_stats(logits, label_off=0, want_cols=True):
    _chk2d(logits, "logits", torch.float32)
    R, Cc = logits.shape
    dev = logits.device
    row_lse = torch.empty(R, device=dev, dtype=torch.float32)
    diag = torch.empty(R, device=dev, dtype=torch.float32)
    col_lse = torch.empty(Cc, device=dev, dtype=torch.float32) if want_cols else None
    ws = torch.empty(2 * ((R + 63) // 64) * Cc, device=dev, dtype=torch.float32) if want_cols else None
    check(_lib.vl_ce_stats(_p(logits), logits.stride(0), R, Cc, label_off, _p(row_lse), _p(col_lse), _p(diag),
                           _p(ws), _stream()))
    return row_lse, col_lse, diag


def ce_loss_accum(loss, row_lse, col_lse, diag, R, Cc, label_off, w_row, w_col):
    check(_lib.vl_ce_loss_accum(_p(row_lse), _p(col_lse), _p(diag), R, Cc, label_off, float(w_row), float(w_col),
                                _p(loss), _stream()))


def ce_grad(logits, row_lse, col_lse, label_off, w_row, w_col, logit_scale, dscale, need_g=True, need_gt=True):
    R, Cc = logits.shape
    dev = logits.device
    ldg = (Cc + 63) // 64 * 64
    ldgt = (R + 63) // 64 * 64
    G = torch.empty(R, ldg, device=dev, dtype=torch.bfloat16) if need_g else None
    GT = torch.empty(Cc, ldgt, device=dev, dtype=torch.bfloat16) if need_gt else None
    ws = None
    if dscale is not None:
        ws = torch.empty(int(_lib.vl_ce_grad_ws_floats(R, Cc, ldg if need_g else 0, ldgt if need_gt else 0)), device=dev,
                         dtype=torch.float32)
    check(_lib.vl_ce_grad(_p(logits), logits.stride(0), R, Cc, label_off, _p(row_lse), _p(col_lse), float(w_row),
                          float(w_col), _p(G), ldg, _p(GT), ldgt, float(logit_scale), _p(dscale), _p(ws), _stream()))
    return G, GT


def device_info(device=0):
    arch = C.create_string_buffer(64)
    cus, clk, mem = C.c_int(), C.c_int(), C.c_long()
    check(_lib.vl_device_info(device, arch, 64, C.byref(cus), C.byref(clk), C.byref(mem)))
    return {"arch": arch.value.decode(), "cus": cus.value, "clock_khz": clk.value, "hbm_bytes": mem.value}


# ------------------------------------------------------------------------------------------------ backward
def layernorm_bwd(dy, x, mean, rstd, w, rows, D, dres=None, dx=None, dx_bf16=None, x_row_stride=None,
                  dy_row_stride=None, dx_row_stride=None):
    """dx = dLN(dy) (+ dres).  dres / dx form the residual-gradient stream: f32 or bf16 (both the same dtype); dx_bf16 is
    an optional extra bf16 copy (the next GEMM operand when the stream is f32)."""
    g = dx if dx is not None else dres
    gdt = F32 if g is None else _dt(g)
    if dres is not None and dx is not None and dres.dtype != dx.dtype:
        raise TypeError("layernorm_bwd: dres and dx must share a dtype")
    check(_lib.vl_layernorm_bwd_g(_p(dy), _dt(dy), D if dy_row_stride is None else dy_row_stride, _p(x), _dt(x),
                                  D if x_row_stride is None else x_row_stride, _p(mean), _p(rstd), _p(w), _p(dres),
                                  _p(dx), gdt, _p(dx_bf16), D if dx_row_stride is None else dx_row_stride, rows, D, _stream()))


_colreduce_ws = {}


def _ws_for(device, rows, cols, planes):
    """Workspace of the deterministic two-stage column reductions (cached per device, grown on demand)."""
    need = int(_lib.vl_colreduce_ws_floats(rows, cols, planes))             # the kernels' own slab geometry, not a copy of it
    # one workspace per device AND stream: two micro-batches' backwards run on two HIP streams (step.py, round 6)
    # (and per host thread: threads that enqueue on one stream interleave their launches - tests emulate ranks that way)
    key = (device, torch.cuda.current_stream(device).cuda_stream, threading.get_ident())
    t = _colreduce_ws.get(key)
    if t is None or t.numel() < need:
        t = torch.empty(need, device=device, dtype=torch.float32)
        _colreduce_ws[key] = t
    return t


def layernorm_bwd_params(dy, x, mean, rstd, dw, db, rows, D, x_row_stride=None, dy_row_stride=None):
    ws = _ws_for(dy.device, rows, D, 2)
    check(_lib.vl_layernorm_bwd_params(_p(dy), _dt(dy), D if dy_row_stride is None else dy_row_stride, _p(x), _dt(x),
                                       D if x_row_stride is None else x_row_stride, _p(mean), _p(rstd), _p(dw), _p(db),
                                       rows, D, _p(ws), _stream()))


def colsum(a, out, scale=1.0):
    _chk2d(a, "a")
    ws = _ws_for(a.device, a.shape[0], a.shape[1], 1)
    check(_lib.vl_colsum(_p(a), _dt(a), a.stride(0), _p(out), a.shape[0], a.shape[1], float(scale), _p(ws), _stream()))


def geglu_bf16(h, out):
    check(_lib.vl_geglu_bf16(_p(h), _p(out), h.shape[0], h.shape[1] // 2, _stream()))
    return out


def gelu_bf16(u, out):
    check(_lib.vl_gelu_bf16(_p(u), _p(out), u.numel(), _stream()))
    return out


def attn_bwd(q, k, v, dO, o, lse, delta, dq, dk, dv, ld_dq, ld_dkv, causal=False, softmax_scale=None, qscale=None, fused=None):
    """q, k, v, dO, o: strided [B,H,L,dh] bf16 views read in place (o = the forward's token-major output viewed by
    heads_view); delta [B,H,Lq] f32 workspace (filled by the two-kernel path); dq/dk/dv token-major destinations.
    fused: None = the one-kernel backward whenever the library takes the shape (vl_attn_bwd_fused_supported: head dim 64,
    self-attention, no mask, L <= 257), True = insist on it (error otherwise), False = the two-kernel path."""
    B, H, Lq, dh = q.shape
    Lk = k.shape[2]
    scale = dh ** -0.5 if softmax_scale is None else softmax_scale
    qs = scale * LOG2E if qscale is None else qscale
    st = _bhld_strides(q, k, v, dO, o)
    if fused is None:
        fused = bool(_lib.vl_attn_bwd_fused_supported(Lq, Lk, dh, 1 if causal else 0))
    if fused:
        if Lq != Lk or causal:
            raise ValueError("attn_bwd(fused=True): self-attention without a causal mask only")
        check(_lib.vl_attn_bwd_fused_bf16(_p(q), _p(k), _p(v), _p(dO), _p(o), st, _p(lse), _p(dq), _p(dk), _p(dv),
                                          ld_dq, ld_dkv, B, H, Lq, dh, float(qs), float(scale), _stream()))
        return
    check(_lib.vl_attn_bwd_bf16(_p(q), _p(k), _p(v), _p(dO), _p(o), st, _p(lse), _p(delta), _p(dq), _p(dk), _p(dv),
                                ld_dq, ld_dkv, B, H, Lq, Lk, dh, float(qs), 1 if causal else 0, float(scale), _stream()))


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("adamw_step: contiguous f32 tensors required")
    check(_lib.vl_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
                             float(weight_decay), int(step), float(grad_scale), _stream()))


def clamp_scalar(p, lo, hi):
    check(_lib.vl_clamp_scalar(_p(p), float(lo), float(hi), _stream()))


def axpy(y, x, alpha=1.0):
    check(_lib.vl_axpy_f32(_p(y), _p(x), float(alpha), y.numel(), _stream()))


def scale_exp(x, log_scale, out=None, mul=1.0):
    """out = x * exp(log_scale) * mul with log_scale a 1-element f32 tensor ON THE DEVICE (no host read)."""
    if x.dtype != torch.float32 or not x.is_contiguous() or log_scale.dtype != torch.float32:
        raise ValueError("scale_exp: contiguous f32 tensors required")
    out = torch.empty_like(x) if out is None else out
    check(_lib.vl_scale_exp_f32(_p(x), _p(out), x.numel(), _p(log_scale), float(mul), _stream()))
    return out


def batch_rowsum(x, out, B, T, D, batch_stride_rows, row_offset):
    check(_lib.vl_batch_rowsum(_p(x), _p(out), B, T, D, batch_stride_rows, row_offset, _stream()))


# ------------------------------------------------------------------------------------------------ point clouds
def fps(xyz, start, G, want_centers=True):
    B, N, _ = xyz.shape
    idx = torch.empty(B, G, device=xyz.device, dtype=torch.int64)
    centers = torch.empty(B, G, 3, device=xyz.device, dtype=torch.float32) if want_centers else None
    check(_lib.vl_fps(_p(xyz.contiguous()), _p(start.contiguous()), _p(idx), _p(centers), B, N, G, _stream()))
    return idx, centers


def pc_gather_normalize(pts, idx=None):
    """pts [B,N,C] f32, idx [B,G] int64 or None -> [B,G,C] f32: the selected points centred and scaled into the unit sphere."""
    B, N, Cc = pts.shape
    G = N if idx is None else idx.shape[1]
    out = torch.empty(B, G, Cc, device=pts.device, dtype=torch.float32)
    check(_lib.vl_pc_gather_normalize(_p(pts.contiguous()), _p(idx.contiguous() if idx is not None else None), _p(out), B, N, G, Cc,
                                      _stream()))
    return out


def knn_group(xyz, center_idx, k, Kp=64, want_idx=False):
    B, N, _ = xyz.shape
    G = center_idx.shape[1]
    nidx = torch.empty(B, G, k, device=xyz.device, dtype=torch.int32) if want_idx else None
    patches = torch.empty(B * G * k, Kp, device=xyz.device, dtype=torch.bfloat16)
    check(_lib.vl_knn_group(_p(xyz.contiguous()), _p(center_idx.contiguous()), _p(nidx), _p(patches), B, N, G, k, Kp, _stream()))
    return patches, nidx


def ball_group(xyz, feats, center_idx, radius, nsample, Kp=64, want_idx=False):
    """Ball query + grouping of the `pnsa` tokenizer (pointnet_util.py:101-161).  xyz [B,N,3] f32, feats [B,N,D] f32 or
    None, center_idx [B,S] int64 -> (patches bf16 [B*S*nsample, Kp] = (xyz_j - centre ++ feats_j, zero padded),
    idx int32 [B,S,nsample] or None).  The radius is squared in double and rounded to f32, as `d > radius ** 2` does."""
    import numpy as np
    B, N, _ = xyz.shape
    S = center_idx.shape[1]
    D = 0 if feats is None else feats.shape[2]
    idx = torch.empty(B, S, nsample, device=xyz.device, dtype=torch.int32) if want_idx else None
    patches = torch.empty(B * S * nsample, Kp, device=xyz.device, dtype=torch.bfloat16)
    check(_lib.vl_ball_group(_p(xyz.contiguous()), _p(feats.contiguous() if feats is not None else None), _p(center_idx.contiguous()),
                             _p(idx), _p(patches), B, N, S, D, float(np.float32(float(radius) ** 2)), nsample, Kp, _stream()))
    return patches, idx


def group_max(x, M, out_dtype=torch.bfloat16):
    _chk2d(x, "x", torch.bfloat16)
    groups = x.shape[0] // M
    out = torch.empty(groups, x.shape[1], device=x.device, dtype=out_dtype)
    check(_lib.vl_group_max(_p(x), x.stride(0), _p(out), _dt(out), out.stride(0), groups, M, x.shape[1], _stream()))
    return out


def pad3(c, Kp=64):
    c = c.reshape(-1, 3).contiguous()
    out = torch.empty(c.shape[0], Kp, device=c.device, dtype=torch.bfloat16)
    check(_lib.vl_pad3_bf16(_p(c), _p(out), c.shape[0], Kp, _stream()))
    return out


def _bn_chunks(R):
    return max(1, min(1024, R // 64))


def bn_stats(x, running_mean=None, running_var=None, momentum=0.1):
    """Batch statistics of nn.BatchNorm1d over the rows of x [R,C] bf16 -> (mean, biased var) f32 [C]."""
    _chk2d(x, "x", torch.bfloat16)
    R, C = x.shape
    n = _bn_chunks(R)
    ws = torch.empty((n + 1) * 2 * C, device=x.device, dtype=torch.float32)
    mean = torch.empty(C, device=x.device, dtype=torch.float32); var = torch.empty_like(mean)
    check(_lib.vl_bn_stats(_p(x), x.stride(0), R, C, _p(ws), n, _p(mean), _p(var), _p(running_mean), _p(running_var),
                           momentum, _stream()))
    return mean, var


def bn_apply(x, mean, var, gamma, beta, eps=1e-5, relu=False, out=None):
    _chk2d(x, "x", torch.bfloat16)
    out = torch.empty_like(x) if out is None else out
    check(_lib.vl_bn_apply(_p(x), x.stride(0), _p(mean), _p(var), _p(gamma), _p(beta), eps, int(relu), _p(out), out.stride(0),
                           x.shape[0], x.shape[1], _stream()))
    return out


def bn_bwd(dy, x, mean, var, gamma, beta, dgamma, dbeta, eps=1e-5, relu=False, train=True, need_dx=True):
    """dgamma/dbeta (f32 [C]) are accumulated; returns dx bf16 [R,C] (None when need_dx is False)."""
    _chk2d(dy, "dy", torch.bfloat16); _chk2d(x, "x", torch.bfloat16)
    R, C = x.shape
    n = _bn_chunks(R)
    ws = torch.empty((n + 1) * 2 * C, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x) if need_dx else None
    check(_lib.vl_bn_bwd(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(mean), _p(var), _p(gamma), _p(beta), eps, int(relu),
                         int(train), _p(ws), n, _p(dgamma), _p(dbeta), _p(dx), dx.stride(0) if need_dx else 0, R, C, _stream()))
    return dx


# ---- SyncBatchNorm: the statistics / backward passes split where the ranks exchange data (csrc/vl_bn.hip) ----
def bn_stats_local(x):
    """This rank's share of the batch statistics of x [R,C] bf16 -> f32 [2C+1]: mean, M2, row count (int bits)."""
    _chk2d(x, "x", torch.bfloat16)
    R, C = x.shape
    n = _bn_chunks(R)
    ws = torch.empty((n + 1) * 2 * C, device=x.device, dtype=torch.float32)
    local = torch.empty(2 * C + 1, device=x.device, dtype=torch.float32)
    check(_lib.vl_bn_stats_local(_p(x), x.stride(0), R, C, _p(ws), n, _p(local), _stream()))
    return local


def bn_stats_merge(gathered, running_mean=None, running_var=None, momentum=0.1):
    """gathered f32 [W, 2C+1] (all ranks' bn_stats_local) -> (mean, biased var, total) with total an int32 [1] tensor
    holding the global row count; running statistics updated with the global unbiased variance."""
    W, n = gathered.shape
    Cc = (n - 1) // 2
    mean = torch.empty(Cc, device=gathered.device, dtype=torch.float32); var = torch.empty_like(mean)
    total = torch.empty(1, device=gathered.device, dtype=torch.int32)
    check(_lib.vl_bn_stats_merge(_p(gathered.contiguous()), W, Cc, _p(mean), _p(var), _p(running_mean), _p(running_var), momentum,
                                 _p(total), _stream()))
    return mean, var, total


def bn_bwd_reduce(dy, x, mean, var, gamma, beta, dgamma, dbeta, eps=1e-5, relu=False):
    """dgamma/dbeta accumulated (local sums, as SyncBatchNorm); returns sums f32 [2C] = (sum dy', sum dy'*xhat) to all-reduce."""
    _chk2d(dy, "dy", torch.bfloat16); _chk2d(x, "x", torch.bfloat16)
    R, C = x.shape
    n = _bn_chunks(R)
    ws = torch.empty((n + 1) * 2 * C, device=x.device, dtype=torch.float32)
    sums = torch.empty(2 * C, device=x.device, dtype=torch.float32)
    check(_lib.vl_bn_bwd_reduce(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(mean), _p(var), _p(gamma), _p(beta), eps, int(relu),
                                _p(ws), n, _p(dgamma), _p(dbeta), _p(sums), R, C, _stream()))
    return sums


def bn_bwd_apply(dy, x, mean, var, gamma, beta, sums, total, eps=1e-5, relu=False):
    """dx bf16 [R,C] from the globally summed `sums` and the global row count `total` (int32 [1] on the device)."""
    _chk2d(dy, "dy", torch.bfloat16); _chk2d(x, "x", torch.bfloat16)
    R, C = x.shape
    dx = torch.empty_like(x)
    check(_lib.vl_bn_bwd_apply(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(mean), _p(var), _p(gamma), _p(beta), eps, int(relu),
                               _p(sums), _p(total), _p(dx), dx.stride(0), R, C, _stream()))
    return dx


def group_max_bwd(f, dg, M, base=None):
    _chk2d(f, "f", torch.bfloat16); _chk2d(dg, "dg", torch.bfloat16)
    out = torch.empty_like(f)
    check(_lib.vl_group_max_bwd(_p(f), f.stride(0), _p(dg), dg.stride(0), _p(base), base.stride(0) if base is not None else 0,
                                _p(out), out.stride(0), f.shape[0] // M, M, f.shape[1], _stream()))
    return out


def group_sum(x, M):
    _chk2d(x, "x", torch.bfloat16)
    out = torch.empty(x.shape[0] // M, x.shape[1], device=x.device, dtype=torch.bfloat16)
    check(_lib.vl_group_sum(_p(x), x.stride(0), _p(out), out.stride(0), x.shape[0] // M, M, x.shape[1], _stream()))
    return out


