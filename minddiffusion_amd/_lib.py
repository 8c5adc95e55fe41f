"""ctypes binding of libmdx.so (include/mdx.h).  The product path has NO fallback: if the HIP
library is missing or a kernel call fails, we raise."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDX_LIBRARY selects another build of the SAME C-ABI (the diagnostics build libmdx_trace.so, tools/gemm_trace.py)
LIB_PATH = os.environ.get("MDX_LIBRARY") or os.path.join(_HERE, "libmdx.so")

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_long = ctypes.c_long
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t


class MdxError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    """struct mdx_gemm_desc (include/mdx.h)."""
    _fields_ = [
        ("a", c_void_p), ("a2", c_void_p), ("c1", c_int), ("c2", c_int),
        ("w", c_void_p), ("bias", c_void_p), ("rowbias", c_void_p), ("rowbias_ld", c_int),
        ("residual", c_void_p), ("residual_ld", c_int),
        ("out", c_void_p), ("out_ld", c_int),
        ("B", c_int), ("H", c_int), ("W", c_int), ("N", c_int),
        ("ksize", c_int), ("stride", c_int), ("upsample", c_int),
        ("epilogue", c_int), ("out_mode", c_int), ("splitk", c_int),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("out_bs", c_long),
        ("out2", c_void_p), ("out2_ld", c_int), ("n_split", c_int), ("asym_pad", c_int),
        ("stats_out", c_void_p), ("ln_stats", c_void_p), ("ln_s", c_void_p), ("ln_nt", c_int), ("ln_eps", c_float),
        ("colstats_out", c_void_p), ("colstats_cap", c_int), ("defer_reduce", c_int),
        ("tile_m", c_int), ("tile_n", c_int),
        ("gn_colstats", c_void_p), ("gn_gamma", c_void_p), ("gn_beta", c_void_p), ("gn_nrb", c_int), ("gn_silu", c_int),
        ("gn_eps", c_float),
        ("skip_a", c_void_p), ("skip_a2", c_void_p), ("skip_c1", c_int), ("skip_c2", c_int), ("skip_w", c_void_p),
        ("w_frag", c_int), ("stages", c_int), ("w_sub", c_void_p),
        ("xattn_k", c_void_p), ("xattn_vt", c_void_p), ("xattn_len", c_int), ("xattn_cap", c_int), ("xattn_scale", c_float),
    ]


class StTailDesc(ctypes.Structure):
    """struct mdx_st_tail_desc (include/mdx.h)."""
    _fields_ = [
        ("attn_out", c_void_p), ("tok", c_void_p), ("x_in", c_void_p), ("out", c_void_p),
        ("ctx_k", c_void_p), ("ctx_vt", c_void_p), ("wstream", c_void_p), ("vec", c_void_p),
        ("colstats_out", c_void_p), ("debug_out", c_void_p), ("debug_stage", c_int),
        ("B", c_int), ("tokens", c_int), ("C", c_int), ("heads", c_int), ("dim_head", c_int),
        ("ctx_len", c_int), ("ctx_cap", c_int), ("scale", c_float), ("ln_eps", c_float), ("tile_rows", c_int), ("warm", c_int),
    ]


class StHeadDesc(ctypes.Structure):
    """struct mdx_st_head_desc (include/mdx.h)."""
    _fields_ = [
        ("x", c_void_p), ("colstats", c_void_p), ("nrb", c_int), ("wstream", c_void_p), ("vec", c_void_p),
        ("tok", c_void_p), ("qk", c_void_p), ("vt", c_void_p), ("vt_ld", c_int), ("debug_out", c_void_p), ("debug_stage", c_int),
        ("B", c_int), ("tokens", c_int), ("C", c_int), ("gn_eps", c_float), ("ln_eps", c_float), ("tile_rows", c_int),
        ("warm", c_int),
    ]


EPI_NONE, EPI_GEGLU, EPI_GELU, EPI_QUICKGELU = 0, 1, 2, 3
OUT_ROWMAJOR, OUT_TRANSPOSED = 0, 1

# name -> (restype, argtypes); also the list the CPU test checks against include/mdx.h
SIGNATURES = {
    "mdx_version": (c_int, []),
    "mdx_last_error": (ctypes.c_char_p, []),
    "mdx_set_option": (c_int, [ctypes.c_char_p, c_int]),
    "mdx_get_option": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int)]),
    "mdx_nchw_to_nhwc_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdx_nhwc_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdx_groupnorm_ws_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mdx_groupnorm_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_int, c_float, c_int, c_void_p, c_void_p]),
    "mdx_groupnorm_colstats_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float,
                                           c_int, c_void_p]),
    "mdx_colstats_fold_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mdx_groupnorm_from_splitk_f16": (c_int, [ctypes.POINTER(GemmDesc), c_void_p, c_void_p, c_void_p, c_int, c_float, c_int,
                                              c_void_p]),
    "mdx_groupnorm_from_splitk_ok": (c_int, [ctypes.POINTER(GemmDesc), c_int]),
    "mdx_groupnorm_scaleshift_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "mdx_layernorm_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "mdx_gemm_f16": (c_int, [ctypes.POINTER(GemmDesc), c_void_p]),
    "mdx_gemm_workspace_bytes": (c_size_t, [ctypes.POINTER(GemmDesc)]),
    "mdx_gemm_release_workspace": (c_int, [c_void_p]),
    "mdx_gemm_bind_counters": (c_int, [c_void_p, c_void_p]),
    "mdx_st_tail_sched_barriers": (c_int, [c_int, c_int]),
    "mdx_st_head_sched_barriers": (c_int, [c_int, c_int]),
    "mdx_gemm_release_counters": (c_int, []),
    "mdx_gemm_tune": (c_int, [ctypes.POINTER(GemmDesc), c_void_p, c_void_p, c_size_t, c_int, ctypes.POINTER(c_int),
                              ctypes.POINTER(ctypes.c_float)]),
    "mdx_gemm_check": (c_int, [ctypes.POINTER(GemmDesc)]),
    "mdx_gemm_query": (c_int, [ctypes.POINTER(GemmDesc), ctypes.POINTER(c_int)]),
    "mdx_attention_f16": (c_int, [c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_void_p, c_long, c_int,
                                  c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "mdx_attention_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "mdx_attention_splitkv_f16": (c_int, [c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_void_p, c_long, c_int,
                                          c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p,
                                          c_size_t, c_void_p]),
    "mdx_attention_causal_f16": (c_int, [c_void_p, c_long, c_int, c_void_p, c_long, c_int, c_void_p, c_long, c_int,
                                  c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "mdx_st_head_f16": (c_int, [ctypes.POINTER(StHeadDesc), c_void_p]),
    "mdx_st_head_supported": (c_int, [c_int, c_int, c_int]),
    "mdx_st_head_stream_bytes": (c_size_t, [c_int]),
    "mdx_st_tail_f16": (c_int, [ctypes.POINTER(StTailDesc), c_void_p]),
    "mdx_st_tail_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mdx_st_tail_stream_bytes": (c_size_t, [c_int]),
    "mdx_timestep_embedding_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "mdx_dense_small_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p]),
    "mdx_sampler_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mdx_avgpool2x2_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mdx_upsample_nearest2x_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mdx_glide_text_embed_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                         c_int, c_int, c_void_p]),
    "mdx_glide_superres_input_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mdx_glide_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_int, c_float, c_void_p,
                                   c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mdx_glide_kv_select_f16": (c_int, [c_void_p, c_int, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "mdx_pack_b_operand_f16": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_void_p]),
    "mdx_vae_gaussian_sample_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mdx_softmax_rows_f16": (c_int, [c_void_p, c_long, c_int, c_int, c_float, c_void_p]),
    "mdx_probe_mfma_32x32x16_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdx_probe_mfma_16x16x32_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdx_probe_gemm_trace": (c_int, [c_void_p, c_size_t]),
    "mdx_probe_valu_rate": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    "mdx_probe_mix_rate": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    "mdx_probe_l2_stream": (c_int, [c_void_p, c_size_t, ctypes.c_uint, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mdx_probe_dma_stream": (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
}

_lib = None
_ENV_OPTIONS = {
    "MDX_GEMM_TUNED": ("gemm_tuned", int), "MDX_GEMM_BM": ("gemm_bm", int), "MDX_GEMM_BN": ("gemm_bn", int),
    "MDX_GEMM_CFG": ("gemm_ring", lambda v: int(v.split(",")[-1])), "MDX_GEMM_HALO": ("gemm_halo", int),
    "MDX_GEMM_HALO8": ("gemm_halo8", int), "MDX_GEMM_SPLITK_FIXUP_MAX": ("gemm_splitk_fixup_max", int),
    "MDX_GEMM_SPREAD": ("gemm_spread", int), "MDX_HALO_NSB": ("halo_nsb", int), "MDX_GN_MIN_BLOCKS": ("gn_min_blocks", int),
    "MDX_GN_FUSED": ("gn_fused", int), "MDX_GN_COL_CHUNKS": ("gn_col_chunks", int),
    "MDX_GEMM_CONV8P": ("gemm_conv8p", int), "MDX_GEMM_CONV8P_MIN_M": ("gemm_conv8p_min_m", int),
    "MDX_GEMM_SUBPIXEL_MIN_TILES": ("gemm_subpixel_min_tiles", int), "MDX_GEMM_CONV8P_VAR": ("gemm_conv8p_var", int),
    "MDX_ATTN8": ("attn8", int), "MDX_ATTN8_MIN_BLOCKS": ("attn8_min_blocks", int),
    "MDX_GN_WIDE_ROWS": ("gn_wide_rows", int), "MDX_GN_BOOST_MB": ("gn_boost_mb", int), "MDX_ATTN_OCC3": ("attn_occ3", int), "MDX_ATTN_KV_SPLIT": ("attn_kv_split", int),
    "MDX_ATTN_FAST_STAGE": ("attn_fast_stage", int), "MDX_GN_PREFETCH": ("gn_prefetch", int), "MDX_GEMM_LN_PREFETCH": ("gemm_ln_prefetch", int),
    "MDX_GEMM_DENSE_ISSUE": ("gemm_dense_issue", int), "MDX_GEMM_LEAN_DENSE": ("gemm_lean_dense", int),
    "MDX_ATTN_PIPE": ("attn_pipe", int),
}


def load():
    """Load libmdx.so (built by ``__graft_entry__.build()`` / ``make -C minddiffusion_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MdxError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C minddiffusion_amd/csrc`.")
    # PyTorch owns the device memory and the streams we are handed, so this library must run on the SAME HIP runtime
    # instance: torch bundles its own libamdhip64 -- import it first so that the dynamic loader resolves libmdx.so's
    # dependency to the copy already in the process (loading /opt/rocm's copy first gives two runtimes, and the second
    # one reports "no ROCm-capable device").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    # The library reads no environment variables (include/mdx.h: mdx_set_option).  The MDX_* variables the tools and the
    # experiment scripts under tools/exp/ pass are translated HERE, once, at load time -- BEFORE the handle is cached, so that a
    # bad value does not leave a half-configured library behind for later calls.
    for env, (name, conv) in _ENV_OPTIONS.items():
        if env in os.environ:
            rc = lib.mdx_set_option(name.encode(), conv(os.environ[env]))
            if rc != 0:
                raise MdxError(f"{env}: {lib.mdx_last_error().decode()}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mdx_last_error()
        raise MdxError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
