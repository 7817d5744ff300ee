"""ctypes binding of libvitlens_hip.so (the C ABI declared in include/vitlens_hip.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LibraryNotBuilt(ImportError):
    pass


def lib_path() -> str:
    return os.path.join(_HERE, "libvitlens_hip.so")


P, I, L, F = C.c_void_p, C.c_int, C.c_long, C.c_float

# include/vitlens_hip.h: VL_ABI_VERSION (tests/test_abi.py compares the two)
ABI_VERSION = 601

# name -> argtypes (all functions return int status unless listed in _RET)
SIGNATURES = {
    "vl_version": [],
    "vl_device_info": [I, C.c_char_p, I, C.POINTER(I), C.POINTER(I), C.POINTER(L)],
    "vl_gemm_bf16": [P, P, P, P, P, I, I, I, I, I, I, F, I, I, I, P],
    "vl_gemm_splitk_accum_f32": [P, P, P, I, I, I, I, I, L, F, I, P, P],
    "vl_gemm_tn_splitk_accum_f32": [P, P, P, I, I, I, I, I, L, F, I, P, P],
    "vl_attn_fwd_bf16": [P, P, P, P, P, P, I, I, I, I, I, F, I, P],
    "vl_gemm_f16": [P, P, P, P, P, I, I, I, I, I, I, F, I, I, P],
    "vl_gemm_f32": [P, P, P, P, P, I, I, I, I, I, I, F, I, P],
    "vl_attn_fwd_f32": [P, P, P, P, P, P, I, I, I, I, I, F, I, P],
    "vl_im2col_f32": [P, P, I, I, I, I, I, I, I, I, I, I, P],
    "vl_attn_fwd_f16": [P, P, P, P, P, P, I, I, I, I, I, F, I, P],
    "vl_layernorm_fwd": [P, I, L, P, L, P, P, P, I, L, P, P, I, I, F, P],
    "vl_gemm_main_rows": [I, I],
    "vl_gemm_lnfold_bf16": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "vl_gemm_res_rowstats_bf16": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "vl_ln_row_stats": [P, I, P, L, I, I, I, F, P, P, P, P, P, L, I, P],
    "vl_assemble_ln_pre": [P, I, P, P, P, P, P, P, I, P, P, P, I, I, I, F, P],
    "vl_l2_normalize": [P, P, P, P, I, I, F, P],
    "vl_l2_normalize_bwd": [P, P, P, P, I, I, F, P],
    "vl_im2col_bf16": [P, P, I, I, I, I, I, I, I, I, I, I, P],
    "vl_text_embed": [P, P, P, P, I, I, I, I, I, P],
    "vl_cast_f32_bf16": [P, P, L, P],
    "vl_add_rows": [P, I, P, P, I, L, I, I, P],
    "vl_transpose_to_bf16": [P, I, L, I, I, P, L, P],
    "vl_transpose_colsum_bf16": [P, I, L, I, I, P, L, P, F, P, P],
    "vl_split_bf16x3": [P, P, L, I, I, P],
    "vl_ce_stats": [P, L, I, I, I, P, P, P, P, P],
    "vl_ce_loss_accum": [P, P, P, I, I, I, F, F, P, P],
    "vl_ce_grad": [P, L, I, I, I, P, P, F, F, P, L, P, L, F, P, P, P],
    "vl_ce_grad_ws_floats": [I, I, L, L],
    "vl_gemm_bf16_ex": [P, P, P, P, P, P, I, I, I, I, I, I, F, I, I, I, I, P],
    "vl_kaldi_fbank": [P, L, I, L, P, P, P, I, I, I, I, I, F, F, F, P],
    "vl_fps": [P, P, P, P, I, I, I, P],
    "vl_pc_gather_normalize": [P, P, P, I, I, I, I, P],
    "vl_resample_h_u8": [P, L, I, I, I, I, P, P, I, I, I, P, P],
    "vl_resample_v_u8_norm": [P, I, I, I, I, P, P, I, I, I, P, P, P, P, P],
    "vl_resample_batch_u8_norm": [P, I, I, I, I, I, P, P, P, P, P, P],
    "vl_resample_h_f32": [P, L, I, I, I, P, P, I, I, I, I, F, F, F, P, P],
    "vl_resample_v_f32_norm": [P, I, I, I, P, P, I, I, I, F, F, P, P],
    "vl_knn_group": [P, P, P, P, I, I, I, I, I, P],
    "vl_ball_group": [P, P, P, P, P, I, I, I, I, F, I, I, P],
    "vl_group_max": [P, L, P, I, L, L, I, I, P],
    "vl_pad3_bf16": [P, P, L, I, P],
    "vl_bn_stats": [P, L, I, I, P, I, P, P, P, P, F, P],
    "vl_bn_stats_local": [P, L, I, I, P, I, P, P],
    "vl_bn_stats_merge": [P, I, I, P, P, P, P, F, P, P],
    "vl_bn_bwd_reduce": [P, L, P, L, P, P, P, P, F, I, P, I, P, P, P, I, I, P],
    "vl_bn_bwd_apply": [P, L, P, L, P, P, P, P, F, I, P, P, P, L, I, I, P],
    "vl_bn_apply": [P, L, P, P, P, P, F, I, P, L, L, I, P],
    "vl_bn_bwd": [P, L, P, L, P, P, P, P, F, I, I, P, I, P, P, P, L, I, I, P],
    "vl_group_max_bwd": [P, L, P, L, P, L, P, L, L, I, I, P],
    "vl_group_sum": [P, L, P, L, L, I, I, P],
    "vl_layernorm_bwd": [P, I, L, P, I, L, P, P, P, P, P, P, L, I, I, P],
    "vl_layernorm_bwd_g": [P, I, L, P, I, L, P, P, P, P, P, I, P, L, I, I, P],
    "vl_colreduce_ws_floats": [I, I, I],
    "vl_layernorm_bwd_params": [P, I, L, P, I, L, P, P, P, P, I, I, P, P],
    "vl_colsum": [P, I, L, P, I, I, F, P, P],
    "vl_gelu_bf16": [P, P, L, P],
    "vl_geglu_bf16": [P, P, L, I, P],
    "vl_attn_bwd_bf16": [P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, I, I, F, I, F, P],
    "vl_attn_bwd_fused_supported": [I, I, I, I],
    "vl_attn_bwd_fused_bf16": [P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, I, F, F, P],
    "vl_adamw_step": [P, P, P, P, L, F, F, F, F, F, I, F, P],
    "vl_clamp_scalar": [P, F, F, P],
    "vl_axpy_f32": [P, P, F, L, P],
    "vl_scale_exp_f32": [P, P, L, P, F, P],
    "vl_batch_rowsum": [P, P, I, I, I, L, L, P],
    # RCCL exchange (csrc/vl_comm.cpp): comm handles are opaque pointers
    "vl_comm_unique_id": [P],
    "vl_comm_create": [C.POINTER(P), P, I, I],
    "vl_comm_destroy": [P],
    "vl_allgather_embed": [P, P, P, L, P],
    "vl_reducescatter_grad": [P, P, P, L, P],
    "vl_allreduce_grad": [P, P, L, P],
}


# functions that do not return a status code
_RET = {"vl_colreduce_ws_floats": L, "vl_ce_grad_ws_floats": L}


def load_library():
    """Load the shared library or fail loudly (there is no fallback path)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise LibraryNotBuilt(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vit-lens_amd/csrc` (hipcc --offload-arch=gfx950). No CPU/eager fallback exists.")
    lib = C.CDLL(path)
    lib.vl_last_error.restype = C.c_char_p
    lib.vl_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = _RET.get(name, I)
    got = int(lib.vl_version())
    if got != ABI_VERSION:
        raise RuntimeError(f"{path} reports ABI version {got}, this binding was written for {ABI_VERSION} "
                           "(include/vitlens_hip.h: VL_ABI_VERSION): rebuild the library (`make -C vit-lens_amd/csrc`)")
    _LIB = lib
    return lib


def check(status: int):
    if status != 0:
        raise RuntimeError("libvitlens_hip: " + load_library().vl_last_error().decode())
