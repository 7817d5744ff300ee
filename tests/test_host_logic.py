"""CPU: host-side arithmetic of the product that needs no GPU (the library only has to load)."""


def test_layernorm_fold_operands_are_the_algebra_of_ln_then_linear():
    """ops.fold_ln_linear (host side of vl_gemm_lnfold_bf16): with Wg = bf16(W * gamma), c = row sums of the ROUNDED Wg and
    d = b + W beta,   rstd * (x Wg^T - mean * c) + d   is exactly   ((x - mean) * rstd) Wg^T + d   (the cancellation of the mean
    term needs c from the rounded weights), and equals LN(x) W^T + b up to the bf16 rounding of Wg."""
    import torch
    from vitlens_hip import ops
    g = torch.Generator().manual_seed(0)
    M, K, N = 64, 256, 96
    x = torch.randn(M, K, generator=g, dtype=torch.float64) * 3 + 1.5
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    gamma = 1 + 0.3 * torch.randn(K, generator=g)
    beta = 0.2 * torch.randn(K, generator=g)
    wg, d, c = ops.fold_ln_linear(w, b, gamma, beta)
    assert wg.dtype == torch.bfloat16 and d.dtype == torch.float32 and c.dtype == torch.float32
    mu = x.mean(1, keepdim=True); rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    folded = rstd * (x @ wg.double().t() - mu * c.double()) + d.double()
    direct = ((x - mu) * rstd) @ wg.double().t() + d.double()
    assert float((folded - direct).abs().max()) < 1e-5                       # c is an fp32 sum of 256 bf16 values
    ref = torch.nn.functional.layer_norm(x, (K,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + b.double()
    assert float((folded - ref).norm() / ref.norm()) < 4e-3                  # bf16 rounding of W * gamma only
    assert float((d.double() - (b.double() + w.double() @ beta.double())).abs().max()) < 1e-5


def test_row_split_of_the_folded_gemms_follows_the_dispatcher():
    """vl_gemm_main_rows = the rows vl_gemm_bf16 (cfg -1) hands to the persistent 256x256 kernel (no GPU needed: 256 CUs are
    assumed without a device): whole rounds of the 256 workgroups where they exist, every full row tile otherwise, nothing for
    problems below 3/4 of one round or widths that are not whole tiles; ops._leftover_cfg picks the leftover rows' kernel the
    way the dispatcher does (64x64 LDS-DMA tiles when they fill most of the chip, else the split-K tail kernel)."""
    from vitlens_hip import _lib, ops
    lib = _lib.load_library()
    T = 257 * 256                                   # 256 images x 257 tokens
    assert lib.vl_gemm_main_rows(T, 3072) == 65536 and lib.vl_gemm_main_rows(T, 1024) == 65536 and lib.vl_gemm_main_rows(T, 4096) == 65536
    assert lib.vl_gemm_main_rows(4 * T, 1024) == 4 * 65536                      # 1028 row tiles: 1024 in whole rounds
    assert lib.vl_gemm_main_rows(65 * 256, 1024) == 64 * 256                    # 260 tiles: one round of 64 row tiles
    assert lib.vl_gemm_main_rows(49 * 256, 1024) == 49 * 256                    # 196 tiles: no whole round, every full row tile
    assert lib.vl_gemm_main_rows(17 * 256, 1024) == 0                           # 68 tiles < 192: small-tile kernels
    assert lib.vl_gemm_main_rows(T, 1664 * 3) == 0 and lib.vl_gemm_main_rows(0, 1024) == 0 and lib.vl_gemm_main_rows(T, 0) == 0
    assert lib.vl_gemm_main_rows(65536 + 40, 1024) == 65536
    assert ops._leftover_cfg(256, 3072, -1) == 11 and ops._leftover_cfg(256, 4096, -1) == 11      # 4 x 48 / 4 x 64 tiles of 64 x 64
    assert ops._leftover_cfg(256, 1024, -1) == 9                                                  # 64 tiles: split-K tail kernel
    assert ops._leftover_cfg(256, 1024, 5) == 5 and ops._leftover_cfg(4096, 1024, -1) == -1       # explicit cfg / many rows: untouched


def test_folding_entries_refuse_bad_arguments_without_touching_the_gpu():
    """Argument checks of the round-4 entries run before any HIP call: status 1 + a message from vl_last_error()."""
    from vitlens_hip import _lib
    lib = _lib.load_library()
    assert lib.vl_ln_row_stats(None, 0, None, 0, 0, 0, 0, 1e-5, None, None, None, None, None, 0, 0, None) == 1
    assert b"bad shape" in lib.vl_last_error()
    assert lib.vl_ln_row_stats(None, 16, None, 1024, 1024, 256, 512, 1e-5, None, None, None, None, None, 0, 0, None) == 1
    assert b"partial statistics missing" in lib.vl_last_error()
    assert lib.vl_gemm_lnfold_bf16(None, None, None, None, None, None, None, None, 256, 256, 512, 512, 512, 256, 0, None) == 1
    assert b"null operand" in lib.vl_last_error()
    assert lib.vl_gemm_res_rowstats_bf16(None, None, None, None, None, None, 256, 256, 512, 512, 512, 256, None) == 1
    assert b"null operand" in lib.vl_last_error()


def test_layernorm_fold_on_rows_with_massive_channels():
    """The arithmetic of the folded path, emulated on the CPU (bf16-rounded operands, fp32 accumulation as the MFMAs do), on
    rows shaped like the residual stream of a TRAINED ViT: two channels hundreds of standard deviations out, per-row offsets,
    gamma spread over 0.05 .. 3.  Against an fp64 LayerNorm + Linear the folded evaluation rstd * (x Wg^T - mean * c) + d is no
    worse than the path it replaces (bf16(LN(x)) bf16(W)^T + b): the subtraction of the mean term happens in fp32 on exact
    products, and the LayerNorm output is never rounded to bf16."""
    import torch
    from vitlens_hip import ops
    g = torch.Generator().manual_seed(3)
    M, K, N = 512, 1024, 256
    x = torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 2) + torch.randn(M, 1, generator=g) * 0.7
    x[:, 7] += 300.0; x[:, 500] -= 180.0
    x = x.bfloat16().float()                                  # the residual stream IS bf16
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    gamma = 0.05 + 2.95 * torch.rand(K, generator=g)
    beta = torch.randn(K, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + b.double()
    mu = x.mean(1, keepdim=True); rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    # the path it replaces: LayerNorm in fp32, output rounded to bf16, bf16 weight, fp32 accumulation
    h = torch.nn.functional.layer_norm(x, (K,), gamma, beta, 1e-5).bfloat16().float()
    plain = h @ w.bfloat16().float().t() + b
    # the folded path: raw bf16 rows against bf16(W gamma), row statistics and column vectors applied in fp32
    wg, d, c = ops.fold_ln_linear(w, b, gamma, beta)
    folded = rstd * (x @ wg.float().t()) + ((-mu * rstd) * c[None, :] + d[None, :])
    e_plain = float((plain.double() - ref).norm() / ref.norm())
    e_fold = float((folded.double() - ref).norm() / ref.norm())
    assert e_fold < 4e-3 and e_fold < 1.2 * e_plain, (e_fold, e_plain)
