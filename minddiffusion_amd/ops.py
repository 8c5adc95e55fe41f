"""Thin torch-tensor wrappers over the C-ABI (include/mdx.h).  torch supplies device memory and
the current HIP stream; all arithmetic happens in libmdx.so.  No CPU fallbacks."""
import ctypes
import os

import torch

from . import _lib
from ._lib import GemmDesc, EPI_NONE, EPI_GEGLU, EPI_GELU, EPI_QUICKGELU, OUT_ROWMAJOR, OUT_TRANSPOSED  # noqa: F401

f16 = torch.float16
f32 = torch.float32


# Library-level options (experiment / tuning switches that used to be environment variables read inside the library):
# one explicit table instead of getenv() calls scattered through the code.
# -1 auto | 0 never | 32 | 64 rows per block of the fused SpatialTransformer tail (MDX_UNET_ST_TAIL presets it for A/B runs)
_OPTIONS = {"unet_st_tail": int(os.environ.get("MDX_UNET_ST_TAIL", "-1")),
            "unet_st_head": int(os.environ.get("MDX_UNET_ST_HEAD", "-1")),     # 0 = keep GroupNorm / proj_in / qkv unfused
            # 3x3 convs with at most this many output rows stream fragment-major weights to registers (0 = never)
            "unet_conv_stream": int(os.environ.get("MDX_UNET_CONV_STREAM", "128")),
            # GroupNorm inputs with > 64 row blocks per sample: 1 = pre-fold their column partials (mdx_colstats_fold_f32), 0 = the
            # two-launch statistics pass.  Measured round 3: GLIDE 256x256 7.57 -> 7.44 images/s with the fold, SDv2 96x96 equal
            "gn_colstats_fold": int(os.environ.get("MDX_GN_COLSTATS_FOLD", "0")),
            # GroupNorms of at most this many pixels per sample take over the split-K reduce of the conv in front of them
            # (mdx_groupnorm_from_splitk_f16).  Round 3, same box: 0 -> 4.437 ms per UNet step / 285 launches, 64 -> 4.422 / 274,
            # 256 -> 4.431 / 263 (round 2 had measured +0.7 %: the reduce kernels it replaces were cheaper then)
            "unet_gn_splitk_fuse": int(os.environ.get("MDX_UNET_GN_SPLITK_FUSE", "256")),
            # streamed convs: > 0 = 128-column tiles with the four waves side by side and this many K splits at most.  Measured
            # SLOWER (M = 128 convs 18.5 -> 20.4-23.6 us): off
            "unet_conv_stream_w4": int(os.environ.get("MDX_UNET_CONV_STREAM_W4", "0")),
            # 1 = a ResBlock's 1x1 skip_connection rides on its second 3x3 conv as extra K tiles (mdx_gemm_desc.skip_w)
            "unet_skip_fuse": int(os.environ.get("MDX_UNET_SKIP_FUSE", "1")),
            # GroupNorm + SiLU inside the consuming 3x3 conv at levels with at least this many output rows (0 = never)
            # (mdx_gemm_desc.gn_colstats).  Measured round 3 at UNet batch 2: 7 GroupNorm launches fewer (-0.10 ms) but the convs'
            # in-LDS normalisation pass costs as much (+0.10 ms): 4.603 -> 4.641 ms per step.  Off.
            "unet_gn_conv_fuse": int(os.environ.get("MDX_UNET_GN_CONV_FUSE", "0")),
            # 1 = when a plan is built, launches whose shape the measured tile table (csrc/gemm_tuned.inc) does not list are timed
            # once on the device (mdx_gemm_tune, a few ms per distinct shape) and keep the fastest form; answers live in tune_cache
            "unet_tune_first_use": int(os.environ.get("MDX_UNET_TUNE_FIRST_USE", "0")),
            # 1 = Upsample convs carry the sub-pixel weights (mdx_gemm_desc.w_sub: 4 Cin instead of 9 Cin products per output on the
            # un-upsampled tensor); the library uses them wherever the eight-wave conv core applies
            "unet_subpixel_upsample": int(os.environ.get("MDX_UNET_SUBPIXEL_UPSAMPLE", "1")),
            # SpatialTransformer.norm (GroupNorm without an activation) applied inside proj_in (mdx_gemm_desc.gn_colstats on a dense
            # launch: one packed fma per A fragment) at levels with at least this many tokens per sample (0 = never): no GroupNorm
            # launch, no normalised copy of the tensor.  Measured round 5 (tools/eval_ab.py, profiles/r05_gn_proj_fuse_ab.txt): at
            # 1024, SDv2 batch 2 233 ops instead of 238 at EQUAL time (4.2824 vs 4.2825 ms: every consumer block folds the statistics
            # before its first MFMA, which costs what the 7 us launch did), Wukong batch 16 +0.4 %, 96 x 96 batch 8 +0.5 %; at 256
            # (the levels whose GroupNorm launch doubles as a split-K reduce) +0.5 % / +1.0 % / +1.6 %.  Off.
            "unet_gn_proj_fuse": int(os.environ.get("MDX_UNET_GN_PROJ_FUSE", "0")),
            # (round 6) SDv2 (head dim 64): BasicTransformerBlock.attn2 -- to_q + the attention over the cached 77 context keys -- as ONE
            # launch, the attention as the epilogue of the query projection (mdx_gemm_desc.xattn_k); 0 = projection + attention launches
            "unet_xattn_fuse": int(os.environ.get("MDX_UNET_XATTN_FUSE", "1")),
            # (round 6) classifier-free guidance evaluates cat([x] * 2) with [uncond ; cond] contexts (plms.py:192-195): up to the
            # first cross-attention both halves are the same numbers.  From this batch on (0 = never) a sampler's guidance call runs
            # conv_in .. the first self-attention (and its output projection where that is a launch of its own) on ONE half and write the result to both (UNetModel._dup_body)
            "unet_cfg_dup": int(os.environ.get("MDX_UNET_CFG_DUP", "4")),
            # 1 = every cfg_dup call first compares the two halves of x (and of per-sample timesteps / temb rows) on the device and raises if
            # they differ: a host synchronisation per call, for bringing up a new sampler -- the shipped samplers build both halves
            # from one tensor
            "unet_cfg_dup_check": int(os.environ.get("MDX_UNET_CFG_DUP_CHECK", "0")),
            # Taichu-GLIDE AttentionBlock (unet.py:267-297): 1 = q | k | v of the image tokens in ONE launch (q | k row-major into a
            # [B, text + image, 2 C] buffer, V transposed: mdx_gemm_desc.n_split with out_bs) instead of three -- 44 launches fewer
            # per base evaluation.  0 = three launches (A/B)
            "glide_qkv_merge": int(os.environ.get("MDX_GLIDE_QKV_MERGE", "1")),
            # Taichu-GLIDE AttentionBlock.norm (GroupNorm without an activation) applied inside the merged q | k | v projection
            # (mdx_gemm_desc.gn_colstats on a dense launch): 22 GroupNorm launches fewer per base evaluation
            "glide_gn_qkv_fuse": int(os.environ.get("MDX_GLIDE_GN_QKV_FUSE", "0"))}


def set_option(name, value):
    """Python-level planner options (unet_st_tail, unet_st_head) or, for any other name, a library option
    (include/mdx.h: mdx_set_option -- gemm_tuned, gemm_bm, gemm_splitk_fixup_max, ...)."""
    if name in _OPTIONS:
        _OPTIONS[name] = int(value)
        return
    _lib.check(_lib.load().mdx_set_option(name.encode(), int(value)), "mdx_set_option")


def get_option(name):
    if name in _OPTIONS:
        return _OPTIONS[name]
    out = ctypes.c_int(0)
    _lib.check(_lib.load().mdx_get_option(name.encode(), ctypes.byref(out)), "mdx_get_option")
    return int(out.value)


def capture_graph(ops_list):
    """Capture the calls of ops_list into one hipGraph and return it.  Python's cyclic garbage collector is held off for the
    duration: a collection in the middle of a capture can finalise an OLDER plan (its graph, its buffers) -- hipGraphExecDestroy /
    hipFree, calls the runtime refuses while a stream is capturing -- and an error inside a C++ destructor aborts the process
    (seen once in round 3: a full test run died in a text-encoder capture while 40 earlier plans were waiting for the collector)."""
    import gc
    g = torch.cuda.CUDAGraph()
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(g):
            for op in ops_list:
                op()
    finally:
        if was_enabled:
            gc.enable()
    return g


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.MdxError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.MdxError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.MdxError(f"{name}: tensor must be contiguous")


def nchw_to_nhwc(x, cpad, out=None):
    """[B,C,H,W] fp32 -> [B,H*W,cpad] fp16 (zero-padded channels)."""
    _chk(x, f32, "x")
    B, C, H, W = x.shape
    if out is None:
        out = torch.empty((B, H * W, cpad), dtype=f16, device=x.device)
    _lib.check(_lib.load().mdx_nchw_to_nhwc_f16(_ptr(x), _ptr(out), B, C, H, W, cpad, _stream()), "mdx_nchw_to_nhwc_f16")
    return out


def nhwc_to_nchw(x, C, H, W, out=None):
    """[B,H*W,Cs] fp16 -> [B,C,H,W] fp32."""
    _chk(x, f16, "x")
    B, HW, Cs = x.shape
    assert HW == H * W
    if out is None:
        out = torch.empty((B, C, H, W), dtype=f32, device=x.device)
    _lib.check(_lib.load().mdx_nhwc_to_nchw_f32(_ptr(x), _ptr(out), B, C, H, W, Cs, _stream()), "mdx_nhwc_to_nchw_f32")
    return out


def groupnorm_ws_floats(B, HW, C, groups=32):
    return int(_lib.load().mdx_groupnorm_ws_floats(B, HW, C, groups))


def groupnorm(x1, x2, gamma, beta, eps, silu, ws=None, out=None, groups=32):
    """GroupNorm(groups)(cat(x1,x2)) [+SiLU]; x* [B,HW,C*] fp16, gamma/beta fp32."""
    _chk(x1, f16, "x1"); _chk(x2, f16, "x2"); _chk(gamma, f32, "gamma"); _chk(beta, f32, "beta")
    B, HW, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[2]
    C = C1 + C2
    if ws is None:
        ws = torch.empty(groupnorm_ws_floats(B, HW, C, groups), dtype=f32, device=x1.device)
    if out is None:
        out = torch.empty((B, HW, C), dtype=f16, device=x1.device)
    _lib.check(_lib.load().mdx_groupnorm_f16(_ptr(x1), C1, _ptr(x2), C2, _ptr(gamma), _ptr(beta), _ptr(out), B, HW,
                                             groups, float(eps), int(bool(silu)), _ptr(ws), _stream()),
               "mdx_groupnorm_f16")
    return out


class FoldedColStats:
    """Column partials with more than 64 row blocks per sample (include/mdx.h: mdx_colstats_fold_f32): `raw` [B * nrb, C, 2] is
    what the producer writes, `folded` [B * nrb2, C, 2] what the GroupNorm reads; groupnorm_colstats launches the fold."""

    def __init__(self, raw, nrb, batch):
        f = (nrb + 63) // 64
        self.raw, self.nrb, self.batch = raw, nrb, batch
        self.nrb2 = (nrb + f - 1) // f
        self.folded = torch.zeros((batch * self.nrb2, raw.shape[1], 2), dtype=f32, device=raw.device)

    def data_ptr(self):         # (the producer's descriptor points at the raw partials)
        return self.raw.data_ptr()

    def fold(self):
        _lib.check(_lib.load().mdx_colstats_fold_f32(_ptr(self.raw), self.nrb, _ptr(self.folded), self.nrb2, self.batch,
                                                     self.raw.shape[1], _stream()), "mdx_colstats_fold_f32")
        return self.folded, self.nrb2


def groupnorm_colstats(x1, cs1, nrb1, x2, cs2, nrb2, gamma, beta, eps, silu, out=None, groups=32, scale=None, shift=None,
                       mod_ld=0):
    """GroupNorm(groups)(cat(x1, x2)) [* (1 + scale) + shift] [+SiLU] with the statistics folded from the producers' column
    partials (mdx_gemm_desc.colstats_out): one launch, one read of x (plus one small fold launch per source with more than
    64 row blocks per sample)."""
    if isinstance(cs1, FoldedColStats):
        cs1, nrb1 = cs1.fold()
    if isinstance(cs2, FoldedColStats):
        cs2, nrb2 = cs2.fold()
    B, HW, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[2]
    if out is None:
        out = torch.empty((B, HW, C1 + C2), dtype=f16, device=x1.device)
    _lib.check(_lib.load().mdx_groupnorm_colstats_f16(_ptr(x1), C1, _ptr(cs1), int(nrb1), _ptr(x2), C2, _ptr(cs2), int(nrb2),
                                                      _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift), int(mod_ld),
                                                      _ptr(out), B, HW, groups, float(eps), int(bool(silu)), _stream()),
               "mdx_groupnorm_colstats_f16")
    return out


def groupnorm_from_splitk_ok(desc, groups=32):
    return bool(_lib.load().mdx_groupnorm_from_splitk_ok(ctypes.byref(desc), groups))


def groupnorm_from_splitk(desc, gamma, beta, eps, silu, out, groups=32):
    """GroupNorm fused with the split-K reduce of the conv / Dense `desc` (launched with defer_reduce = 1)."""
    _lib.check(_lib.load().mdx_groupnorm_from_splitk_f16(ctypes.byref(desc), _ptr(gamma), _ptr(beta), _ptr(out), groups,
                                                         float(eps), int(bool(silu)), _stream()),
               "mdx_groupnorm_from_splitk_f16")
    return out


def groupnorm_scaleshift(x1, x2, gamma, beta, scale, shift, mod_ld, eps, silu, ws=None, out=None, groups=32):
    """GLIDE ResBlock FiLM norm: silu?(GN(cat(x1,x2)) * (1 + scale[b]) + shift[b]); scale/shift fp32 views [B, C]."""
    _chk(x1, f16, "x1"); _chk(x2, f16, "x2")
    B, HW, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[2]
    C = C1 + C2
    if ws is None:
        ws = torch.empty(groupnorm_ws_floats(B, HW, C, groups), dtype=f32, device=x1.device)
    if out is None:
        out = torch.empty((B, HW, C), dtype=f16, device=x1.device)
    _lib.check(_lib.load().mdx_groupnorm_scaleshift_f16(
        _ptr(x1), C1, _ptr(x2), C2, _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift), int(mod_ld), _ptr(out), B, HW,
        groups, float(eps), int(bool(silu)), _ptr(ws), _stream()), "mdx_groupnorm_scaleshift_f16")
    return out


def avgpool2x2(x, B, H, W, C, out=None):
    if out is None:
        out = torch.empty((B, (H // 2) * (W // 2), C), dtype=f16, device=x.device)
    _lib.check(_lib.load().mdx_avgpool2x2_f16(_ptr(x), _ptr(out), B, H, W, C, _stream()), "mdx_avgpool2x2_f16")
    return out


def upsample_nearest2x(x, B, H, W, C, out=None):
    if out is None:
        out = torch.empty((B, 4 * H * W, C), dtype=f16, device=x.device)
    _lib.check(_lib.load().mdx_upsample_nearest2x_f16(_ptr(x), _ptr(out), B, H, W, C, _stream()),
               "mdx_upsample_nearest2x_f16")
    return out


def glide_text_embed(tokens, mask, tok_emb, pos, pad, out=None):
    """tokens/mask int32 [B,T]; tables fp16; -> [B,T,width] fp16."""
    B, T = tokens.shape
    width = pos.shape[1]
    if out is None:
        out = torch.empty((B, T, width), dtype=f16, device=tokens.device)
    _lib.check(_lib.load().mdx_glide_text_embed_f16(_ptr(tokens), _ptr(mask), _ptr(tok_emb), _ptr(pos), _ptr(pad),
                                                    _ptr(out), B, T, width, tok_emb.shape[0], _stream()),
               "mdx_glide_text_embed_f16")
    return out


def glide_superres_input(x, low, out=None):
    """x [B,3,S,S] fp32, low [B,3,s,s] fp32 -> NHWC fp16 [B, S*S, 8] = [x | bilinear(quantised low) | 0 0]."""
    B, _, S, _ = x.shape
    if out is None:
        out = torch.empty((B, S * S, 8), dtype=f16, device=x.device)
    _lib.check(_lib.load().mdx_glide_superres_input_f16(_ptr(x), _ptr(low), _ptr(out), B, S, low.shape[2], _stream()),
               "mdx_glide_superres_input_f16")
    return out


def glide_step(x, out_c, out_u, ld, scale, coef8, mode, noise_scale, noise, x_next, pred_x0):
    B, _, H, W = x.shape
    c8 = (ctypes.c_float * 8)(*[float(v) for v in coef8])
    _lib.check(_lib.load().mdx_glide_step_f32(_ptr(x), _ptr(out_c), _ptr(out_u), int(ld), float(scale),
                                              ctypes.cast(c8, ctypes.c_void_p), int(mode), float(noise_scale),
                                              _ptr(noise), _ptr(x_next), _ptr(pred_x0), B, H, W, _stream()),
               "mdx_glide_step_f32")


def glide_kv_slots(entries, device):
    """Device array of struct mdx_glide_kv_slot (include/mdx.h) from (src, dst, src_entry_bytes, dst_batch_bytes, src_pitch,
    dst_pitch, rows, row_bytes) tuples with tensor src / dst; returns (int64 tensor [n, 7] that holds the structs, n)."""
    import numpy as np
    rows = []
    for src, dst, seb, dbb, sp, dp, r, rb in entries:
        for v in (src.data_ptr(), dst.data_ptr(), seb, dbb, sp, dp, rb):
            if v % 16:
                raise _lib.MdxError("glide_kv_slots: pointers and byte counts must be multiples of 16")
        rows.append([src.data_ptr(), dst.data_ptr(), seb, dbb, sp, dp, (int(r) & 0xffffffff) | (int(rb) << 32)])
    return torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(device), len(rows)


def glide_kv_select(slots, nslots, entry0, entry_per_b, b0, nb, blocks_per_copy=8):
    _lib.check(_lib.load().mdx_glide_kv_select_f16(_ptr(slots), int(nslots), int(entry0), int(entry_per_b), int(b0), int(nb),
                                                   int(blocks_per_copy), _stream()), "mdx_glide_kv_select_f16")


def layernorm(x, gamma, beta, eps, out=None):
    _chk(x, f16, "x"); _chk(gamma, f32, "gamma"); _chk(beta, f32, "beta")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.load().mdx_layernorm_f16(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out), rows, C, float(eps), _stream()),
               "mdx_layernorm_f16")
    return out


def pack_gemm_weight(w2d):
    """[N, K] weight (K already in the kernel's K order) -> the packed fp16 storage mdx_gemm_f16 reads
    (include/mdx.h): N, K zero-padded to multiples of 64, tile-major [N/64][K/64][64][8 chunks][8], chunks
    pre-swizzled (position q of row r holds logical chunk q ^ ((r >> 1) & 7)).  Pure data movement, done once
    at weight-load time."""
    N, K = w2d.shape
    dev = w2d.device
    Np, Kp = (N + 63) // 64 * 64, (K + 63) // 64 * 64
    wp = torch.zeros((Np, Kp), dtype=f16, device=dev)
    wp[:N, :K] = w2d
    t = wp.view(Np // 64, 64, Kp // 64, 8, 8).permute(0, 2, 1, 3, 4)          # [panel, ktile, row, chunk, 8]
    r = torch.arange(64, device=dev)
    src = torch.arange(8, device=dev)[None, :] ^ ((r >> 1) & 7)[:, None]       # [row, position] -> logical chunk
    idx = src[None, None, :, :, None].expand(Np // 64, Kp // 64, 64, 8, 8)
    return torch.gather(t, 3, idx).contiguous().view(-1)


def conv_weight_k_order(w4d, cin_pad=None, cout_pad=None):
    """[Cout, Cin, kh, kw] -> [Cout_pad, K] in the kernel's K order: [cin/64][kh*kw][64] when Cin % 64 == 0
    (channel-chunk major, tap minor), else tap-major [kh*kw][Cin_pad]."""
    w4d = w4d.to(torch.float32)
    co, ci, kh, kw = w4d.shape
    cip = cin_pad or ci
    cop = cout_pad or co
    if cip % 64 == 0:
        assert cip == ci
        p = torch.zeros((cop, ci // 64, kh * kw, 64), dtype=torch.float32, device=w4d.device)
        p[:co] = w4d.reshape(co, ci // 64, 64, kh * kw).permute(0, 1, 3, 2)
        return p.reshape(cop, kh * kw * ci)
    p = torch.zeros((cop, kh * kw, cip), dtype=torch.float32, device=w4d.device)
    p[:co, :, :ci] = w4d.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
    return p.reshape(cop, kh * kw * cip)


def pack_conv_weight(w4d, cin_pad=None, cout_pad=None):
    return pack_gemm_weight(conv_weight_k_order(w4d, cin_pad, cout_pad))


def subpixel_conv_weight(w4d):
    """[Cout, Cin, 3, 3] of a conv that follows a nearest-2x upsample (Upsample.construct, openaimodel.py:57-60) -> the four
    2 x 2 convs of the LOW-resolution tensor it is equal to, [4 Cout, Cin, 2, 2] (parity (dy, dx) major): output pixel
    (2y + dy, 2x + dx) reads upsampled rows 2y + dy + ky - 1, i.e. source rows {y - 1, y, y} (dy = 0) or {y, y, y + 1} (dy = 1), so the
    taps that fall on one source row are summed (fp32, rounded to fp16 once by the packer): rows {w0, w1 + w2} / {w0 + w1, w2},
    columns likewise.  4 Cin instead of 9 Cin products per output."""
    w = w4d.to(torch.float32)
    rows = {0: (w[:, :, 0], w[:, :, 1] + w[:, :, 2]), 1: (w[:, :, 0] + w[:, :, 1], w[:, :, 2])}       # [Cout, Cin, 3 (kx)] each
    out = []
    for dy in (0, 1):
        for dx in (0, 1):
            taps = []
            for a in (0, 1):
                r = rows[dy][a]
                cols = (r[:, :, 0], r[:, :, 1] + r[:, :, 2]) if dx == 0 else (r[:, :, 0] + r[:, :, 1], r[:, :, 2])
                taps.append(torch.stack(cols, -1))                                                     # [Cout, Cin, 2 (b)]
            out.append(torch.stack(taps, 2))                                                           # [Cout, Cin, 2 (a), 2 (b)]
    return torch.cat(out, 0)


def pack_subpixel_conv_weight(w4d):
    """mdx_gemm_desc.w_sub (include/mdx.h): the packed per-parity 2 x 2 weights, logical [4 Cout][4 Cin] in the conv K order."""
    assert w4d.shape[0] % 64 == 0 and w4d.shape[1] % 64 == 0
    return pack_conv_weight(subpixel_conv_weight(w4d))


def pack_conv_weight_frag(w4d):
    """[Cout, Cin, 3, 3] -> the MFMA-fragment-major packing of mdx_gemm_desc.w_frag (Cout % 64 == 0, Cin % 64 == 0)."""
    co, ci = w4d.shape[0], w4d.shape[1]
    assert co % 64 == 0 and ci % 64 == 0
    return pack_frag_weight(conv_weight_k_order(w4d).to(f16)).reshape(-1)


def unpack_gemm_weight(packed, N, K):
    """Inverse of pack_gemm_weight: the tile-major, pre-swizzled storage -> [N, K] (plan-time re-packing of a conv that turns
    out to want the fragment-major form once its M is known)."""
    Np, Kp = (N + 63) // 64 * 64, (K + 63) // 64 * 64
    t = packed.view(Np // 64, Kp // 64, 64, 8, 8)
    r = torch.arange(64, device=packed.device)
    src = torch.arange(8, device=packed.device)[None, :] ^ ((r >> 1) & 7)[:, None]      # position -> logical chunk
    inv = torch.argsort(src, 1)                                                          # logical chunk -> position
    idx = inv[None, None, :, :, None].expand(Np // 64, Kp // 64, 64, 8, 8)
    return torch.gather(t, 3, idx).permute(0, 2, 1, 3, 4).reshape(Np, Kp)[:N, :K].contiguous()


def make_gemm_desc(a, w, N, B, H, W, c1, out, out_ld, a2=None, c2=0, bias=None, rowbias=None, rowbias_ld=0,
                   residual=None, residual_ld=0, ksize=1, stride=1, upsample=0, epilogue=EPI_NONE,
                   out_mode=OUT_ROWMAJOR, splitk=0, workspace=None, out_bs=0, out2=None, out2_ld=0, n_split=0, asym_pad=0,
                   stats_out=None, ln_stats=None, ln_s=None, ln_eps=1e-5, tile_m=0, tile_n=0, colstats_out=None, stages=0,
                   w_frag=0, skip_a=None, skip_a2=None, skip_c1=0, skip_c2=0, skip_w=None, gn_colstats=None, gn_nrb=0,
                   gn_gamma=None, gn_beta=None, gn_eps=1e-5, gn_silu=1, w_sub=None, xattn_k=None, xattn_vt=None, xattn_len=0,
                   xattn_cap=0, xattn_scale=0.0):
    d = GemmDesc()
    d.a = a.data_ptr()
    d.a2 = 0 if a2 is None else a2.data_ptr()
    d.c1, d.c2 = int(c1), int(c2)
    d.w = w.data_ptr()
    d.bias = 0 if bias is None else bias.data_ptr()
    d.rowbias = 0 if rowbias is None else rowbias.data_ptr()
    d.rowbias_ld = int(rowbias_ld)
    d.residual = 0 if residual is None else residual.data_ptr()
    d.residual_ld = int(residual_ld)
    d.out = out.data_ptr()
    d.out_ld = int(out_ld)
    d.B, d.H, d.W, d.N = int(B), int(H), int(W), int(N)
    d.ksize, d.stride, d.upsample = int(ksize), int(stride), int(upsample)
    d.epilogue, d.out_mode, d.splitk = int(epilogue), int(out_mode), int(splitk)
    d.workspace = 0 if workspace is None else workspace.data_ptr()
    d.workspace_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    d.out_bs = int(out_bs)
    d.out2 = 0 if out2 is None else out2.data_ptr()
    d.out2_ld, d.n_split, d.asym_pad = int(out2_ld), int(n_split), int(asym_pad)
    # LayerNorm fold (include/mdx.h): producer statistics / consumer correction
    d.stats_out = 0 if stats_out is None else stats_out.data_ptr()
    d.ln_stats = 0 if ln_stats is None else ln_stats.data_ptr()
    d.ln_s = 0 if ln_s is None else ln_s.data_ptr()
    d.ln_nt = 0 if ln_stats is None else (int(c1) + int(c2)) // 64
    d.ln_eps = float(ln_eps)
    d.tile_m, d.tile_n, d.stages, d.w_frag = int(tile_m), int(tile_n), int(stages), int(w_frag)
    d.gn_colstats = 0 if gn_colstats is None else gn_colstats.data_ptr()
    d.gn_gamma = 0 if gn_gamma is None else gn_gamma.data_ptr()
    d.gn_beta = 0 if gn_beta is None else gn_beta.data_ptr()
    d.gn_nrb, d.gn_silu, d.gn_eps = int(gn_nrb), int(gn_silu), float(gn_eps)
    d.skip_a = 0 if skip_a is None else skip_a.data_ptr()
    d.skip_a2 = 0 if skip_a2 is None else skip_a2.data_ptr()
    d.skip_c1, d.skip_c2 = int(skip_c1), int(skip_c2)
    d.skip_w = 0 if skip_w is None else skip_w.data_ptr()
    d.colstats_out = 0 if colstats_out is None else colstats_out.data_ptr()
    d.colstats_cap = 0 if colstats_out is None else int(colstats_out.shape[0])
    d.w_sub = 0 if w_sub is None else w_sub.data_ptr()
    if xattn_k is not None:      # cross-attention as the epilogue of the query projection (include/mdx.h)
        d.xattn_k, d.xattn_vt = xattn_k.data_ptr(), xattn_vt.data_ptr()
        d.xattn_len, d.xattn_cap, d.xattn_scale = int(xattn_len), int(xattn_cap), float(xattn_scale)
    return d


def fold_layernorm(wt, gamma, beta, bias=None):
    """Host side of the LayerNorm fold (include/mdx.h, mdx_gemm_desc.ln_stats): for y = LN(x; gamma, beta) W^T + b returns
    (fp16 gamma (.) W [N, K], S fp32 [N], W beta + b fp32 [N]).  S is summed from the fp16-ROUNDED product, i.e. from
    exactly the numbers the MFMA multiplies, so the mean term cancels to accumulation noise."""
    w32 = wt.to(torch.float64)
    wg = (w32 * gamma.to(torch.float64)[None, :]).to(f16)
    s = wg.to(torch.float64).sum(1).to(f32).contiguous()
    cb = w32 @ beta.to(torch.float64)
    if bias is not None:
        cb = cb + bias.to(torch.float64)
    return wg.contiguous(), s, cb.to(f32).contiguous()


def gemm_workspace_bytes(desc):
    return int(_lib.load().mdx_gemm_workspace_bytes(ctypes.byref(desc)))


_WS_HEAD = 16384      # MDX_GEMM_WS_HEAD (include/mdx.h): bytes of arrival counters per workspace


def new_gemm_workspace(nbytes, device):
    """A split-K workspace for mdx_gemm_f16 (fp32) with its arrival counters: ONE zeroed torch allocation [16 KiB of counters |
    workspace], the counters bound to the workspace address (mdx_gemm_bind_counters) -- so libmdx.so allocates no device memory
    for launches on it (include/mdx.h, ownership rule) -- and unbound when the tensor dies (before the caching allocator can hand
    the address to anything else).  The workspace itself needs no initialisation."""
    import weakref
    base = torch.zeros((_WS_HEAD + max(int(nbytes), 16)) // 4 + 1, dtype=f32, device=device)
    ws = base[_WS_HEAD // 4:]
    lib = _lib.load()
    _lib.check(lib.mdx_gemm_bind_counters(ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(base.data_ptr())),
               "mdx_gemm_bind_counters")
    # unbind when the workspace tensor object dies (the plans hold exactly this object; its storage cannot be freed earlier)
    weakref.finalize(ws, lib.mdx_gemm_release_workspace, ctypes.c_void_p(ws.data_ptr()))
    return ws


def gemm_query(desc):
    """(tile_m, tile_n, splitk, halo, from_tuned_table, colstats rows per block, in-kernel split-K reduce) mdx_gemm_f16
    would use for this descriptor (no launch)."""
    out = (ctypes.c_int * 7)()
    _lib.check(_lib.load().mdx_gemm_query(ctypes.byref(desc), out), "mdx_gemm_query")
    return tuple(int(v) for v in out)


def account_gemm_launches(meta):
    """Plan post-pass (after the shared workspace is patched into the descriptors): launches per GEMM op = the kernel
    plus a split-K reduce launch unless the split is reduced in the kernel; the op's info string gets the real split."""
    for m in meta:
        d = m.get("desc")
        if d is None:
            continue
        q = gemm_query(d)
        m["launches"] = 2 if (q[2] > 1 and not q[6] and not d.defer_reduce) else 1
        m["info"] = m["info"].split(" split=")[0] + f" split={q[2] if q[2] > 1 else 0}" + ("i" if q[6] else "")


# First-use tuning (include/mdx.h mdx_gemm_tune): the user-side cache.  key = gemm_shape_key(desc) -> (tile_m, tile_n, splitk,
# stages, us of the library's choice, us of the best form).  save_tune_cache / load_tune_cache keep it across processes.
tune_cache = {}
_tune_flush = {}


def gemm_shape_key(d):
    """What a launch form depends on: problem shape + launch variant (the tile table's key, gemm.hip tuned_variant())."""
    var = ((1 if d.c2 > 0 else 0) | (d.epilogue << 1) | (8 if d.n_split else 0) | (16 if d.ln_stats else 0)
           | (32 if d.stats_out else 0) | (64 if d.out_mode == 1 else 0) | (128 if d.colstats_out else 0)
           | (256 if d.residual else 0) | (512 if d.rowbias else 0) | (1024 if d.skip_w else 0) | (2048 if d.gn_gamma else 0)
           | (4096 if d.w_sub else 0))      # (w_sub: the sub-pixel form of an Upsample conv resolves to another kernel)
    # B, H, W are part of the key: HALO eligibility, patch geometry (16-wide patches vs the 8 x 8 two-sample form) and the
    # eight-wave cores' tile counts depend on them, not only on M = B H W
    return (d.B * d.H * d.W, d.N, d.ksize * d.ksize * (d.c1 + d.c2), d.ksize, d.stride, d.upsample, var, d.B, d.H, d.W)


def gemm_tune(desc, reps=5, cold=True):
    """mdx_gemm_tune on the current stream: (tile_m, tile_n, splitk, stages, us_auto, us_best); zeros = keep the library's
    choice.  cold: a 512 MiB fill evicts L2 / Infinity Cache before every timed launch (weights arrive cold, as in a UNet
    evaluation).  SYNCHRONISES; overwrites desc.out."""
    flush, nbytes = None, 0
    if cold:
        dev = torch.cuda.current_device()
        if dev not in _tune_flush:
            _tune_flush[dev] = torch.empty(512 << 20, dtype=torch.uint8, device=f"cuda:{dev}")
        flush, nbytes = ctypes.c_void_p(_tune_flush[dev].data_ptr()), _tune_flush[dev].numel()
    best, us = (ctypes.c_int * 4)(), (ctypes.c_float * 2)()
    _lib.check(_lib.load().mdx_gemm_tune(ctypes.byref(desc), _stream(), flush, nbytes, int(reps), best, us), "mdx_gemm_tune")
    return tuple(int(v) for v in best) + (float(us[0]), float(us[1]))


def tune_untuned(descs, reps=5):
    """First-use pass of a planner: every descriptor whose shape neither the tile table nor tune_cache knows is measured once
    (descriptors tied to their launch form -- colstats_out, defer_reduce, w_frag -- and forced ones are left alone); the
    cached form is applied to every descriptor of that shape.  Returns the number of shapes measured."""
    measured = 0
    for d in descs:
        if (d.colstats_out or d.defer_reduce or d.w_frag or d.skip_w or d.gn_gamma or d.tile_m or d.tile_n or d.splitk
                or d.stages):
            continue
        key = gemm_shape_key(d)
        if key not in tune_cache:
            if gemm_query(d)[4]:        # the measured table has this shape
                continue
            tune_cache[key] = gemm_tune(d, reps)
            measured += 1
        tm, tn, sk, stg = tune_cache[key][:4]
        old = (d.tile_m, d.tile_n, d.splitk, d.stages)
        d.tile_m, d.tile_n, d.splitk, d.stages = tm, tn, sk, stg
        if _lib.load().mdx_gemm_check(ctypes.byref(d)) != 0:      # a cached form this descriptor cannot take (e.g. a loaded cache)
            d.tile_m, d.tile_n, d.splitk, d.stages = old
    return measured


def release_tune_scratch():
    _tune_flush.clear()


TUNE_CACHE_VERSION = 2      # bump when gemm_shape_key changes: entries keyed the old way can never match and are dropped on load


def save_tune_cache(path):
    import json
    with open(path, "w") as f:
        json.dump({"version": TUNE_CACHE_VERSION, "key_fields": 10,
                   "entries": [[list(k), list(v)] for k, v in sorted(tune_cache.items())]}, f)


def load_tune_cache(path):
    """Returns the number of entries taken.  A file written under another key layout (no / other version, other key length) is
    ignored with a message: its keys would never match and every shape would silently be measured again."""
    import json
    import warnings
    with open(path) as f:
        data = json.load(f)
    if not isinstance(data, dict) or data.get("version") != TUNE_CACHE_VERSION:
        warnings.warn(f"load_tune_cache: {path} was written under another key layout (version "
                      f"{data.get('version') if isinstance(data, dict) else 'none'}, this build: {TUNE_CACHE_VERSION}); ignored")
        return 0
    n = 0
    for k, v in data["entries"]:
        if len(k) != 10:
            continue
        tune_cache[tuple(k)] = tuple(v)
        n += 1
    return n


def gemm_run(desc):
    _lib.check(_lib.load().mdx_gemm_f16(ctypes.byref(desc), _stream()), "mdx_gemm_f16")


def gemm(a, w, N, B, H, W, c1, out=None, **kw):
    """Convenience one-shot (tests): allocates output (row-major [M, out_cols]) and split-K workspace."""
    _chk(a, f16, "a"); _chk(w, f16, "w")
    ksize, stride, upsample = kw.get("ksize", 1), kw.get("stride", 1), kw.get("upsample", 0)
    Hs, Ws = (2 * H, 2 * W) if upsample else (H, W)
    pad = 1 if ksize == 3 else 0
    Ho = (Hs + 2 * pad - ksize) // stride + 1
    Wo = (Ws + 2 * pad - ksize) // stride + 1
    M = B * Ho * Wo
    epi = kw.get("epilogue", EPI_NONE)
    out_mode = kw.get("out_mode", OUT_ROWMAJOR)
    if out is None:
        if out_mode == OUT_TRANSPOSED:
            ld = kw.pop("out_ld", Ho * Wo)
            out = torch.zeros((B, N, ld), dtype=f16, device=a.device)
        else:
            cols = N // 2 if epi == EPI_GEGLU else N
            ld = kw.pop("out_ld", cols)
            out = torch.empty((M, ld), dtype=f16, device=a.device)
    else:
        ld = kw.pop("out_ld")
    d = make_gemm_desc(a, w, N, B, H, W, c1, out, ld, **kw)
    if not d.workspace:
        need = gemm_workspace_bytes(d)
        if need:
            ws = new_gemm_workspace(need, a.device)
            d.workspace = ws.data_ptr()
            d.workspace_bytes = need
    gemm_run(d)
    return out


def attention(q_ptr, k_ptr, vt_ptr, o_ptr, B, heads, D, Nq, Nk, scale, q_bs, q_ld, k_bs, k_ld, vt_bs, vt_ld, o_bs, o_ld,
              causal=False, ws=None, kv_splits=0):
    """Raw-pointer form (q/k may be column slices of one fused projection buffer); see include/mdx.h.
    `ws`: a zero-initialised uint8 workspace tensor (attention_workspace) -> the split-KV form, `kv_splits` 0 = auto."""
    lib = _lib.load()
    if ws is not None and not causal:
        _lib.check(lib.mdx_attention_splitkv_f16(
            ctypes.c_void_p(q_ptr), q_bs, q_ld, ctypes.c_void_p(k_ptr), k_bs, k_ld, ctypes.c_void_p(vt_ptr), vt_bs, vt_ld,
            ctypes.c_void_p(o_ptr), o_bs, o_ld, B, heads, D, Nq, Nk, float(scale), int(kv_splits), ctypes.c_void_p(ws.data_ptr()),
            ws.numel() * ws.element_size(), _stream()), "mdx_attention_splitkv_f16")
        return
    fn = lib.mdx_attention_causal_f16 if causal else lib.mdx_attention_f16
    _lib.check(fn(ctypes.c_void_p(q_ptr), q_bs, q_ld, ctypes.c_void_p(k_ptr), k_bs, k_ld, ctypes.c_void_p(vt_ptr), vt_bs,
                  vt_ld, ctypes.c_void_p(o_ptr), o_bs, o_ld, B, heads, D, Nq, Nk, float(scale), _stream()),
               "mdx_attention_causal_f16" if causal else "mdx_attention_f16")


def attention_ws_bytes(B, heads, D, Nq, Nk):
    """Bytes of workspace the library's own split choice needs for this shape (0: it would not split)."""
    return int(_lib.load().mdx_attention_ws_bytes(B, heads, D, Nq, Nk))


def attention_workspace(nbytes, device):
    """Zeroed workspace for the split-KV attention: the arrival counters at its head must be zero on the first use and every
    launch leaves them zero, so ONE workspace serves all attention launches of a plan (one stream) in turn."""
    return torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def timestep_embedding(t, dim, max_period=10000.0, out=None):
    _chk(t, f32, "t")
    M = t.shape[0]
    if out is None:
        out = torch.empty((M, dim), dtype=f32, device=t.device)
    _lib.check(_lib.load().mdx_timestep_embedding_f32(_ptr(t), _ptr(out), M, dim, float(max_period), _stream()),
               "mdx_timestep_embedding_f32")
    return out


def dense_small(x, w, b, act_in=False, act_out=False, out=None):
    """out[M,N] = act_out(act_in(x) @ w^T + b); x/out fp32, w fp16 [N,K]."""
    _chk(x, f32, "x"); _chk(w, f16, "w"); _chk(b, f32, "b")
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=f32, device=x.device)
    _lib.check(_lib.load().mdx_dense_small_f32(_ptr(x), x.stride(0), _ptr(w), _ptr(b), _ptr(out), out.stride(0), M, N, K,
                                               int(act_in), int(act_out), _stream()), "mdx_dense_small_f32")
    return out


def sampler_step(x, eps_u, eps_c, eps_ld, cfg_scale, olds, coef, sqrt_at, sqrt_one_minus_at, sqrt_a_prev, dir_coef,
                 sigma, noise, e_t_out, x_prev, pred_x0):
    """Fused CFG + multistep mix + DDIM update (see include/mdx.h)."""
    B, C, H, W = x.shape
    c4 = (ctypes.c_float * 4)(*[float(v) for v in coef])
    o = list(olds) + [None] * (3 - len(olds))
    _lib.check(_lib.load().mdx_sampler_step_f32(
        _ptr(x), _ptr(eps_u), _ptr(eps_c), int(eps_ld), float(cfg_scale), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]),
        ctypes.cast(c4, ctypes.c_void_p), float(sqrt_at), float(sqrt_one_minus_at), float(sqrt_a_prev), float(dir_coef),
        float(sigma), _ptr(noise), _ptr(e_t_out), _ptr(x_prev), _ptr(pred_x0), B, C, H, W, _stream()),
        "mdx_sampler_step_f32")


def probe_mfma(a, b):
    c = torch.empty((64, 16), dtype=f32, device=a.device)
    _lib.check(_lib.load().mdx_probe_mfma_32x32x16_f16(_ptr(a), _ptr(b), _ptr(c), _stream()), "mdx_probe_mfma")
    return c


def probe_mfma16(a, b):
    c = torch.empty((64, 4), dtype=f32, device=a.device)
    _lib.check(_lib.load().mdx_probe_mfma_16x16x32_f16(_ptr(a), _ptr(b), _ptr(c), _stream()), "mdx_probe_mfma16")
    return c


def pack_b_operand(src2d, out=None):
    """Row-major fp16 activation matrix [rows, K] (any row stride) -> packed B operand of mdx_gemm_f16."""
    rows, K = src2d.shape
    n = ((rows + 63) // 64) * ((K + 63) // 64) * 4096
    if out is None:
        out = torch.empty(n, dtype=f16, device=src2d.device)
    assert out.numel() >= n and src2d.stride(1) == 1
    _lib.check(_lib.load().mdx_pack_b_operand_f16(_ptr(src2d), src2d.stride(0), rows, K, _ptr(out), _stream()),
               "mdx_pack_b_operand_f16")
    return out


def softmax_rows(x2d, scale):
    """In place: x[r] = softmax(scale * x[r]) on fp16 scores [rows, cols]."""
    rows, cols = x2d.shape
    assert x2d.stride(1) == 1
    _lib.check(_lib.load().mdx_softmax_rows_f16(_ptr(x2d), x2d.stride(0), rows, cols, float(scale), _stream()),
               "mdx_softmax_rows_f16")
    return x2d


def vae_gaussian_sample(moments, zc, noise, out):
    """moments NHWC fp16 [B, HW, ld]; noise / out NCHW fp32 [B, zc, h, w] (noise None -> the mode)."""
    B, HW, ld = moments.shape
    _lib.check(_lib.load().mdx_vae_gaussian_sample_f32(_ptr(moments), ld, _ptr(noise), _ptr(out), B, zc, HW, _stream()),
               "mdx_vae_gaussian_sample_f32")
    return out


def wire_groupnorm_colstats(gn_calls, meta, batch, device, table):
    """Plan-time post-pass shared by the UNet planners (run AFTER the split-K workspace is patched into the descriptors, so
    that mdx_gemm_query sees the real split factors): every GroupNorm whose inputs are GEMM outputs and that would take the
    two-launch path (>= ~1k pixels per sample) gets its statistics from its producers' epilogues
    (mdx_gemm_desc.colstats_out) -- gn_stats and its read pass disappear.  gn_calls: dicts with x1, x2, prod = (desc of x1's
    producer, desc of x2's producer), meta = index into `meta`; sets call["cs"] = (cs1, nrb1, cs2, nrb2).  `table` keeps the
    statistics buffers alive (descriptor address -> tensor)."""
    import math
    fold_many = get_option("gn_colstats_fold") != 0
    for c in gn_calls:
        _, HW, C1 = c["x1"].shape
        # fused SpatialTransformer head / GroupNorm inside the consuming conv: the statistics feed that launch (any block count)
        is_head = c.get("head") is not None or c.get("conv") is not None or c.get("proj") is not None
        C2 = 0 if c["x2"] is None else c["x2"].shape[2]
        cpg = (C1 + C2) // 32
        L = cpg // math.gcd(cpg, 8)           # chunk columns of the minimal whole-group column block
        if not is_head and L <= 64 and HW * L * 16 <= (64 << 10):
            meta[c["meta"]]["launches"] = 1   # the one-launch fused kernel (norm.hip groupnorm_impl)
            continue

        fresh = []      # producers wired by THIS GroupNorm (undone if its other source cannot supply statistics)

        def stats_of(d, cx):
            if isinstance(d, _lib.StTailDesc):      # fused SpatialTransformer tail: per-row-block column sums of its output
                key = ctypes.addressof(d)
                if key not in table:
                    rows = d.tile_rows
                    if d.C != cx or HW % rows or HW // rows > 128:
                        return None
                    buf = torch.zeros((batch * (HW // rows), cx, 2), dtype=f32, device=device)
                    d.colstats_out = buf.data_ptr()
                    table[key] = (buf, HW // rows)
                    fresh.append((key, d))
                return table[key]
            if d is None or d.N != cx or d.out_ld != cx or d.defer_reduce:
                return None
            key = ctypes.addressof(d)
            if key in table:
                return table[key]
            # The statistics epilogue is part of the launch VARIANT the tile table is keyed by: ask with the field already set
            # (any non-null value), or the row blocks the query reports are those of a different tile / split choice than the
            # launch will make (a 64-row split-K reduce writing into a buffer sized for 128-row tiles).
            d.colstats_out = 8
            # ... and with an AMPLE workspace: the planners size the shared workspace again after this pass (to the largest ideal need
            # of any descriptor), so the launch will take the variant's ideal form -- with the workspace of the first sizing the query
            # can report a fallback (e.g. an un-split 128-row tile where the tuned row splits five ways and its reduce kernel writes
            # 64-row blocks: found by tools/shape_sweep.py --model glide at a 32-pixel base, round 5 -- the launch then refused
            # the statistics buffer as too small)
            keep_ws = d.workspace_bytes
            d.workspace_bytes = 1 << 40
            rows = gemm_query(d)[5]
            d.workspace_bytes = keep_ws
            if rows <= 0 or HW % rows or HW // rows > 4096:
                d.colstats_out = 0
                return None
            buf = torch.zeros((batch * (HW // rows), cx, 2), dtype=f32, device=device)
            d.colstats_out, d.colstats_cap = buf.data_ptr(), batch * (HW // rows)
            d._cs_rows = rows       # (python-side) what the buffer was sized for: check_colstats_wiring() re-asks after the final sizing
            if HW // rows > 64 and not is_head and fold_many:
                # > 64 row blocks per sample (GLIDE's 128 x 128 / 256 x 256 levels: 512 HALO patches): folding them in EVERY
                # gn_apply block cost more than the statistics pass it saved (profiles/r02_e_ab.txt); they are folded ONCE by
                # a small launch in front of the GroupNorm instead (mdx_colstats_fold_f32)
                table[key] = (FoldedColStats(buf, HW // rows, batch), HW // rows)
            elif HW // rows > 64 and not is_head:
                d.colstats_out = 0
                return None
            else:
                table[key] = (buf, HW // rows)
            fresh.append((key, d))
            return table[key]
        s1 = stats_of(c["prod"][0], C1)
        if c.get("proj") is not None:
            # GroupNorm (no activation) in front of a Dense / 1x1 conv: the consumer applies it to its A fragments when the
            # producer can supply column statistics in <= 64 row blocks per sample and an M tile stays inside one sample;
            # otherwise this call falls back to the GroupNorm launch it was planned with (handled below like any other)
            pj = c.pop("proj")
            dd = pj["desc"]
            ok = False
            if s1 is not None and not isinstance(s1[0], FoldedColStats) and int(s1[1]) <= 64:
                keep = (dd.a, dd.gn_colstats, dd.gn_nrb, dd.gn_gamma, dd.gn_beta, dd.gn_eps, dd.gn_silu)
                dd.a, dd.gn_colstats, dd.gn_nrb = c["x1"].data_ptr(), s1[0].data_ptr(), int(s1[1])
                dd.gn_gamma, dd.gn_beta, dd.gn_eps, dd.gn_silu = c["g"].data_ptr(), c["b"].data_ptr(), float(c["eps"]), 0
                ok = _lib.load().mdx_gemm_check(ctypes.byref(dd)) == 0 and HW % gemm_query(dd)[0] == 0
                if not ok:
                    dd.a, dd.gn_colstats, dd.gn_nrb, dd.gn_gamma, dd.gn_beta, dd.gn_eps, dd.gn_silu = keep
            if ok:
                dd._gn_src = (c["x1"], s1[0])       # (python-side: keeps the raw input and the statistics buffer alive)
                meta[c["meta"]]["dead"] = True      # the planner drops the GroupNorm op
                meta[pj["meta"]]["info"] += " +groupnorm(in)"
                continue
            for key, d in fresh:
                table.pop(key, None)
                d.colstats_out = 0
                if hasattr(d, "colstats_cap"):
                    d.colstats_cap = 0
            fresh.clear()
            is_head = False
            L_ = cpg // math.gcd(cpg, 8)
            if L_ <= 64 and HW * L_ * 16 <= (64 << 10):
                meta[c["meta"]]["launches"] = 1
                continue
            s1 = stats_of(c["prod"][0], C1)
        if is_head:
            if s1 is not None and c.get("head") is not None:
                c["head"].colstats, c["head"].nrb = s1[0].data_ptr(), int(s1[1])
            elif s1 is not None:
                c["conv"].gn_colstats, c["conv"].gn_nrb = s1[0].data_ptr(), int(s1[1])
            continue
        s2 = stats_of(c["prod"][1], C2) if C2 else (None, 0)
        if s1 is None or s2 is None:
            # one source cannot supply statistics: the GroupNorm takes the two-launch path, and a producer that was wired only
            # for it must not keep paying for the statistics epilogue (nor resolve to that launch variant's tile-table row)
            for key, d in fresh:
                table.pop(key, None)
                d.colstats_out = 0
                if hasattr(d, "colstats_cap"):
                    d.colstats_cap = 0
            continue
        c["cs"] = (s1[0], s1[1], s2[0], s2[1])
        meta[c["meta"]]["launches"] = 1 + isinstance(s1[0], FoldedColStats) + isinstance(s2[0], FoldedColStats)


# ---------------------------------------------------------------------------------------------------------------
# Row-local fused SpatialTransformer tail (include/mdx.h: mdx_st_tail_f16, csrc/stchain.hip)
def check_colstats_wiring(descs):
    """Called by the planners AFTER the shared split-K workspace has its final size: wire_groupnorm_colstats sized every statistics
    buffer for the row blocks the launch reports under an AMPLE workspace (its ideal form); the launch takes that form only if
    the final workspace really holds it.  A planner that kept the first-sized workspace would fail at its first launch with a
    colstats_cap mismatch -- fail here instead, with the descriptor named."""
    for d in descs:
        rows = getattr(d, "_cs_rows", None)
        if rows is None or not d.colstats_out:
            continue
        now = gemm_query(d)[5]
        if now != rows:
            raise _lib.MdxError(f"GroupNorm statistics wiring: the launch M={d.B * d.H * d.W} N={d.N} k{d.ksize} writes {now}-row "
                                f"blocks under the final workspace ({d.workspace_bytes} bytes) but its statistics buffer was sized "
                                f"for {rows}-row blocks: size the workspace from gemm_workspace_bytes() AFTER the wiring pass")


def pack_frag_weight(w2d):
    """[N, K] nn.Dense weight -> MFMA-fragment-major pieces [N/32 column tiles][K/16 k-steps][64 lanes * 8 halves]:
    piece (ct, s)[lane] = W[32 ct + lane % 32][16 s + 8 (lane // 32) + 0..7] -- the first operand of
    v_mfma_f32_32x32x16_f16 exactly as a lane holds it, so a wave loads one piece with ONE coalesced 1 KiB instruction."""
    N, K = w2d.shape
    assert N % 32 == 0 and K % 16 == 0
    t = w2d.to(f16).reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4)     # [ct, s, hi, l31, 8]
    return t.reshape(N // 32, K // 16, 512).contiguous()


def pack_st_tail(wo1, wq2, wo2, w1, w2, wpo, bo1, g2, be2, bo2, g3, be3, b1, b2, bpo):
    """Weights / vectors of one SpatialTransformer block (depth 1) in the layout mdx_st_tail_f16 streams:
    wave w of the C/32 waves owns output columns [32w, 32w+32) of every GEMM of the chain; its pieces are stored in
    consumption order -- to_out1, to_q2, to_out2, 4 x (ff1 chunk: 'a' and 'gate' pieces interleaved per k-step, then the ff2
    K-chunk), proj_out -- as ONE contiguous stream.  w1: [8C, C] (reference order: a rows then gate rows), w2: [C, 4C].
    Returns (stream fp16 [C/32, 16 C/16, 512], vec fp32 [16 C])."""
    C = wo1.shape[0]
    NW, KS = C // 32, C // 16
    f = pack_frag_weight
    fo1, fq2, fo2, fpo = f(wo1), f(wq2), f(wo2), f(wpo)        # [NW, KS, 512]
    f1 = f(w1)                                                  # [8C/32, KS, 512]: tiles [0, 4C/32) = a, then gate
    f2 = f(w2)                                                  # [NW, 4C/16, 512]
    parts = [fo1, fq2, fo2]
    for c in range(4):
        a = f1[c * NW:(c + 1) * NW]
        g = f1[4 * C // 32 + c * NW:4 * C // 32 + (c + 1) * NW]
        parts.append(torch.stack([a, g], 2).reshape(NW, 2 * KS, 512))
        parts.append(f2[:, c * KS:(c + 1) * KS])
    parts.append(fpo)
    stream = torch.cat(parts, 1).contiguous()
    assert stream.shape == (NW, 16 * KS, 512)
    vec = torch.cat([v.to(f32).reshape(-1) for v in (bo1, g2, be2, bo2, g3, be3, b1, b2, bpo)]).contiguous()
    assert vec.numel() == 16 * C
    return stream, vec


def st_tail_supported(C, heads, dim_head, tokens, tile_rows):
    return bool(_lib.load().mdx_st_tail_supported(int(C), int(heads), int(dim_head), int(tokens), int(tile_rows)))


def make_st_tail_desc(attn_out, tok, x_in, out, ctx_k, ctx_vt, stream, vec, B, tokens, C, heads, dim_head, ctx_len, ctx_cap,
                      tile_rows=64, ln_eps=1e-5, colstats_out=None, debug_out=None, debug_stage=0):
    d = _lib.StTailDesc()
    d.attn_out, d.tok, d.x_in, d.out = attn_out.data_ptr(), tok.data_ptr(), x_in.data_ptr(), out.data_ptr()
    d.ctx_k, d.ctx_vt, d.wstream, d.vec = ctx_k.data_ptr(), ctx_vt.data_ptr(), stream.data_ptr(), vec.data_ptr()
    d.colstats_out = 0 if colstats_out is None else colstats_out.data_ptr()
    d.debug_out = 0 if debug_out is None else debug_out.data_ptr()
    d.debug_stage = int(debug_stage)
    d.B, d.tokens, d.C, d.heads, d.dim_head = int(B), int(tokens), int(C), int(heads), int(dim_head)
    d.ctx_len, d.ctx_cap = int(ctx_len), int(ctx_cap)
    d.scale, d.ln_eps, d.tile_rows = float(dim_head) ** -0.5, float(ln_eps), int(tile_rows)
    d.warm = int(os.environ.get("MDX_ST_TAIL_WARM", "1"))
    return d


def st_tail_run(desc):
    _lib.check(_lib.load().mdx_st_tail_f16(ctypes.byref(desc), _stream()), "mdx_st_tail_f16")


def pack_st_head(wpi, wq, wk, wv, gn_g, gn_b, bpi, g1, be1):
    """Weights / vectors of the fused SpatialTransformer head (mdx_st_head_f16): per wave its proj_in, to_q, to_k, to_v column
    tiles as one contiguous stream of MFMA-fragment pieces.  Returns (stream fp16 [C/32, 4 C/16, 512], vec fp32 [5 C])."""
    f = pack_frag_weight
    stream = torch.cat([f(wpi), f(wq), f(wk), f(wv)], 1).contiguous()
    vec = torch.cat([t.to(f32).reshape(-1) for t in (gn_g, gn_b, bpi, g1, be1)]).contiguous()
    return stream, vec


def st_head_supported(C, tokens, tile_rows):
    return bool(_lib.load().mdx_st_head_supported(int(C), int(tokens), int(tile_rows)))


def make_st_head_desc(x, colstats, nrb, stream, vec, tok, qk, vt, vt_ld, B, tokens, C, tile_rows=32, gn_eps=1e-6, ln_eps=1e-5,
                      debug_out=None, debug_stage=0):
    d = _lib.StHeadDesc()
    d.x, d.colstats, d.nrb = x.data_ptr(), colstats.data_ptr(), int(nrb)
    d.wstream, d.vec = stream.data_ptr(), vec.data_ptr()
    d.tok, d.qk, d.vt, d.vt_ld = tok.data_ptr(), qk.data_ptr(), vt.data_ptr(), int(vt_ld)
    d.debug_out = 0 if debug_out is None else debug_out.data_ptr()
    d.debug_stage = int(debug_stage)
    d.B, d.tokens, d.C = int(B), int(tokens), int(C)
    d.gn_eps, d.ln_eps, d.tile_rows = float(gn_eps), float(ln_eps), int(tile_rows)
    d.warm = int(os.environ.get("MDX_ST_TAIL_WARM", "1"))
    return d


def st_head_run(desc):
    _lib.check(_lib.load().mdx_st_head_f16(ctypes.byref(desc), _stream()), "mdx_st_head_f16")
