"""ctypes binding of libklab_mm.so (the C ABI declared in include/klab_mm.h).

The product path fails loudly when the HIP library is missing: there is no CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KLAB_LIB", os.path.join(HERE, "libklab_mm.so"))  # KLAB_LIB: A/B-test another build

F32, BF16, FP8 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
AUX_NONE, AUX_NONZERO, AUX_DGELU = 0, 1, 2
ERR_UNSUPPORTED, ERR_BADARG = -2, -3

vp, i32, i64, f32, u32 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint32


class GemmArgs(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("dtype", i32),
                ("A", vp), ("lda", i64), ("a_kmajor", i32),
                ("B", vp), ("ldb", i64), ("b_kmajor", i32),
                ("C", vp), ("ldc", i64), ("c_dtype", i32), ("accumulate", i32),
                ("alpha", f32), ("alpha_dev", vp), ("bias", vp), ("act", i32),
                ("aux", vp), ("ldaux", i64), ("aux_mode", i32), ("aux_scale", f32),
                ("residual", vp), ("ldr", i64), ("r_dtype", i32),
                ("drop_p", f32), ("seed_dev", vp), ("drop_tag", u32), ("name_tag", i32), ("atomic_ok", i32)]


class AttnArgs(C.Structure):
    _fields_ = [("dtype", i32), ("q", vp), ("ldq", i64), ("k", vp), ("ldk", i64), ("v", vp), ("ldv", i64),
                ("bias", vp), ("causal", i32), ("ctx", vp), ("ldo", i64), ("lse", vp),
                ("B", i32), ("H", i32), ("Lq", i32), ("Lk", i32), ("dk", i32),
                ("drop_p", f32), ("seed_dev", vp), ("drop_tag", u32),
                ("dctx", vp), ("lddo", i64), ("dq", vp), ("lddq", i64), ("dk_out", vp), ("lddk", i64),
                ("dv", vp), ("lddv", i64), ("dbias", vp), ("ds_ws", vp), ("ds_defer", i32),
                ("score_scale", vp), ("bias_mod", i32)]


class AttnFusedArgs(C.Structure):  # klab_attn_fused_args
    _fields_ = [("x", vp), ("gamma", vp), ("eps", f32), ("d_model", i32), ("w", vp), ("xn", vp), ("rstd", vp), ("proj", vp), ("ldproj", i64),
                ("cross", i32), ("attn", AttnArgs)]


class SwinAttnArgs(C.Structure):
    _fields_ = [("dtype", i32), ("qkv", vp), ("ctx", vp), ("bias", vp), ("logit_scale", vp), ("lse", vp),
                ("B", i32), ("R", i32), ("w", i32), ("shift", i32), ("H", i32), ("C", i32),
                ("dctx", vp), ("dqkv", vp), ("dbias", vp), ("dlogit_scale", vp),
                ("bwd_ws", vp), ("bwd_ws_bytes", C.c_size_t), ("bias_table", vp), ("dbias_table", vp),
                ("v_bias", vp), ("dv_bias", vp)]


# every exported entry point of include/klab_mm.h: name -> argtypes (restype is always int)
SIGNATURES = {
    "klab_version": [],
    "klab_gemm": [C.POINTER(GemmArgs), vp],
    "klab_gemm_probe_enable": [i32],
    "klab_gemm_probe_read": [vp, vp, vp],
    "klab_rmsnorm_fwd": [vp, vp, vp, i32, vp, vp, i32, i32, f32, i32, i32, i32, f32, vp, u32, vp],
    "klab_rmsnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, f32, u32, f32, u32, vp, vp],
    "klab_rmsnorm_bwd_part": [vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, f32, u32, f32, u32, vp, vp],
    "klab_rmsnorm_part_rows": [i32],
    "klab_colpart_reduce": [vp, i64, i32, i32, vp, i32, vp],
    "klab_adam_step": [vp, i32, i64, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, f32, f32, vp],
    "klab_adam_step_range": [vp, i32, i64, i64, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, f32, f32, vp],
    "klab_layernorm_fwd": [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, f32, i32, i32, i32, f32, vp, u32, vp],
    "klab_layernorm_bwd": [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, u32, vp],
    "klab_rmsnorm_fwd_q8": [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, u32, vp],
    "klab_layernorm_fwd_q8": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "klab_gelu_fwd": [vp, vp, i32, i64, vp],
    "klab_gelu_fwd_q8": [vp, vp, vp, vp, i32, i32, vp],
    "klab_layernorm_bwd_bias": [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, u32, vp],
    "klab_t5_attn_fwd": [C.POINTER(AttnArgs), vp],
    "klab_t5_attn_bwd": [C.POINTER(AttnArgs), vp],
    "klab_t5_attn_fused_fwd": [C.POINTER(AttnFusedArgs), vp],
    "klab_t5_decode_attn": [i32, vp, i64, vp, vp, i64, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp],
    "klab_dbias_reduce": [vp, i32, vp, i32, i32, i32, i32, vp],
    "klab_swin_mlp_fused": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "klab_gemm_grouped": [vp, i32, vp],
    "klab_gemm_fp8": [C.POINTER(GemmArgs), vp, vp, i64, vp],
    "klab_quant_fp8_rows": [vp, i64, i32, i32, vp, i64, vp, vp],
    "klab_quant_fp8_arena": [vp, i32, i64, vp, vp, vp, vp],
    "klab_swin_qkv_attn_fused": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "klab_swin_linear_ln_fused": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "klab_swin_patch_embed_fused": [vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp],
    "klab_swin_proj_ln_fused": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "klab_swin_attn_fwd": [C.POINTER(SwinAttnArgs), vp],
    "klab_swin_attn_bwd": [C.POINTER(SwinAttnArgs), vp],
    "klab_swin_attn_bwd_ws_bytes": [i32, i32, i32, i32, i32, i32],
    "klab_swin_cpb_bias": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "klab_swin_cpb_table": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "klab_swin_cpb_table_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "klab_cast_pack": [vp, i32, i64, vp, i32, vp],
    "klab_embed_fwd": [vp, i32, i32, i32, i32, vp, i32, vp, i32, i32, f32, vp, u32, vp, vp],
    "klab_embed_bwd": [vp, i32, i32, i32, i32, vp, vp, i32, i32, i32, f32, vp, u32, vp],
    "klab_relbias_fwd": [vp, vp, vp, i32, i32, i32, vp],
    "klab_relbias_bwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "klab_ce_fwd": [vp, i64, i32, vp, i32, i32, vp, vp, vp, i32, vp],
    "klab_ce_count": [vp, i32, vp, vp],
    "klab_im2col_patch": [vp, vp, i32, i32, i32, i32, i32, vp],
    "klab_im2col_patch_ld": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "klab_merge_gather": [vp, vp, i32, i32, i32, i32, vp],
    "klab_merge_scatter": [vp, vp, i32, i32, i32, vp],
    "klab_colsum": [vp, i64, i32, i32, i32, vp, vp],
    "klab_convert": [vp, vp, i32, i64, f32, vp],
    "klab_add_f32": [vp, vp, i64, vp],
    "klab_image_preprocess_ws_bytes": [i32, i32, i32],
    "klab_segv_trace_install": [i32],
    "klab_segv_set_context": [C.c_char_p],
    "klab_image_preprocess": [vp, vp, i32, i32, i32, i32, i32, i32, i32, C.c_double, vp, vp, vp, vp, C.c_size_t, vp],
    "klab_jpeg_read_info": [vp, C.c_size_t, vp],
    "klab_jpeg_entropy_decode": [vp, C.c_size_t, vp, vp, vp],
    "klab_jpeg_entropy_decode_batch": [vp, vp, i32, vp, vp, vp, vp, i32],
    "klab_jpeg_decode_ws_bytes": [vp, i32],
    "klab_jpeg_decode_device": [vp, vp, vp, vp, i32, vp, vp, C.c_size_t, vp],
}


class JpegInfo(C.Structure):  # klab_jpeg_info
    _fields_ = [("width", i32), ("height", i32), ("ncomp", i32), ("precision", i32), ("progressive", i32), ("supported", i32),
                ("colour", i32), ("hmax", i32), ("vmax", i32), ("mcus_x", i32), ("mcus_y", i32), ("hs", i32 * 3), ("vs", i32 * 3),
                ("bw", i32 * 3), ("bh", i32 * 3), ("tq", i32 * 3), ("coef_blocks", C.c_longlong)]


class JpegItem(C.Structure):  # klab_jpeg_item
    _fields_ = [("info", JpegInfo), ("coef_block0", C.c_longlong), ("rgb_off", C.c_longlong)]

_lib = None


class KlabError(RuntimeError):
    pass


def load():
    """dlopen the HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KlabError(f"{LIB_PATH} is missing: run `python -m klab_multimodalmodel_amd.build` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: its wheel bundles its own libamdhip64 -- loaded before ours, our library's HIP symbols bind to that same
    # runtime instance (loaded after, the process would hold two HIP runtimes and every torch pointer / stream handed to the
    # C ABI would be foreign to ours: hipErrorNoDevice at klab_engine_bind)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argt in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argt
        fn.restype = C.c_size_t if name.endswith("_ws_bytes") else i32
    _lib = lib
    return lib


def check(rc, what=""):
    if rc == 0:
        return
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(f"klab: unsupported shape/alignment in {what}")
    if rc == ERR_BADARG:
        raise ValueError(f"klab: bad argument to {what}")
    raise KlabError(f"klab: {what} failed with hipError {rc}")


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.bfloat16:
        return BF16
    if torch_dtype == "fp8":
        return FP8
    raise ValueError(f"unsupported dtype {torch_dtype}")


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
