"""UNetModel -- MI355X-native mirror of the reference's
vision/stablediffusionv2/ldm/modules/diffusionmodules/openaimodel.py:245-577 (and the Wukong copy).

Same constructor keywords (the ``unet_config.params`` keys of the reference YAMLs) and the same
``construct(x, timesteps, context)`` call; parameters are loaded by the reference's names
 via ``load_state_dict``.  Execution is completely different from the
reference's per-primitive MindSpore graph:

  * activations live in HBM as NHWC fp16 ([B, H*W, C] == token-major), so SpatialTransformer's
    NCHW<->NLC transposes (attention.py:243-253) vanish and every conv is an implicit GEMM;
  * the forward pass is PLANNED once per (B, H, W): a flat list of C-ABI kernel calls on
    pre-allocated, liveness-reused buffers (static addresses => capturable as ONE hipGraph);
  * Concat of skip connections (openaimodel.py:568) is never materialised (two-source kernels),
    nearest-2x Upsample is folded into the following conv's gather, the ResBlock time-embedding
    add / bias / residual adds / GEGLU are GEMM epilogues;
  * cross-attention K / V^T of the text context (constant over the sampling loop) are cached.

There is no CPU path: every op is a HIP kernel from libmdx.so.
"""
import ctypes
import math

import os
import weakref

import numpy as np
import torch

from .... import ops
from ...._lib import MdxError
from ....weights import check_state_dict

f16, f32 = torch.float16, torch.float32


def _round_up(x, m):
    return (x + m - 1) // m * m


class _Arena:
    """Liveness-based buffer reuse at PLAN time (exact-size buckets).  Execution never allocates."""

    def __init__(self, device):
        self.device = device
        self.free = {}
        self.bases = []   # every allocation, kept alive for the plan's lifetime: GEMM descriptors hold raw pointers
        self.total = 0

    def get(self, shape, dtype=f16):
        n = int(np.prod(shape))
        nbytes = _round_up(n * torch.empty((), dtype=dtype).element_size(), 256)
        lst = self.free.get(nbytes)
        if lst:
            base = lst.pop()
        else:
            base = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.bases.append(base)
            self.total += nbytes
        t = base[: n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(*shape)
        t._mdx_base = base
        return t

    def release(self, t):
        base = t._mdx_base
        self.free.setdefault(base.numel(), []).append(base)


class UNetModel:
    def __init__(self, image_size=32, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None,
                 n_embed=None, legacy=True, use_linear_in_transformer=False, device="cuda:0"):
        # openaimodel.py:305-321 argument checks
        if use_spatial_transformer:
            assert context_dim is not None, "context_dim is required with use_spatial_transformer"
        if context_dim is not None:
            assert use_spatial_transformer, "context_dim requires use_spatial_transformer"
        if num_heads == -1:
            assert num_head_channels != -1, "Either num_heads or num_head_channels has to be set"
        if num_head_channels == -1:
            assert num_heads != -1, "Either num_heads or num_head_channels has to be set"
        if not use_spatial_transformer:
            raise NotImplementedError("AttentionBlock is an empty stub in the reference (openaimodel.py:208-242)")
        if dims != 2:
            raise NotImplementedError("dims != 2: the image UNet is the only instantiation on the denoising path")
        self.conv_resample = bool(conv_resample)
        self.num_classes = num_classes
        self.use_scale_shift_norm = bool(use_scale_shift_norm)
        self.resblock_updown = bool(resblock_updown)
        self.transformer_depth = int(transformer_depth)
        self.n_embed = n_embed                       # predict_codebook_ids (openaimodel.py:344, 527-531)
        assert self.transformer_depth >= 1
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = list(attention_resolutions)
        self.channel_mult = list(channel_mult)
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.context_dim = int(context_dim)
        self.legacy = legacy
        self.use_linear = use_linear_in_transformer
        self.use_fp16 = use_fp16
        self.device = torch.device(device)
        self.time_embed_dim = model_channels * 4
        self.input_blocks, self.middle_block, self.output_blocks = self._structure()
        self.cin_pad = _round_up(in_channels, 8)
        self.final_channels = out_channels if n_embed is None else int(n_embed)
        self.cout_pad = _round_up(self.final_channels, 8)
        self.w = None          # packed device weights
        self._plans = {}
        self._ctx_key = None
        self._ctx_ref = None
        self.use_graph = True
        self.max_context_len = 80  # 77 CLIP tokens rounded up to a multiple of 8 (V^T rows are 16-B chunked)
        self.last_launch_count = 0

    # ------------------------------------------------------------------ structure (openaimodel.py:351-526)
    def _heads(self, ch, num_heads):
        if self.num_head_channels == -1:
            dim_head = ch // num_heads
        else:
            num_heads = ch // self.num_head_channels
            dim_head = self.num_head_channels
        if self.legacy:
            dim_head = ch // num_heads  # use_spatial_transformer branch of openaimodel.py:375-376
        return num_heads, dim_head

    def _structure(self):
        mc = self.model_channels
        nh = self.num_heads
        inb = [[("conv", self.in_channels, mc)]]
        chans = [mc]
        ch, ds = mc, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(self.num_res_blocks):
                layers = [("res", ch, mult * mc)]
                ch = mult * mc
                if ds in self.attention_resolutions:
                    nh, dh = self._heads(ch, nh)
                    layers.append(("st", ch, nh, dh))
                inb.append(layers)
                chans.append(ch)
            if level != len(self.channel_mult) - 1:
                inb.append([("resdown", ch, ch) if self.resblock_updown else ("down", ch)])
                chans.append(ch)
                ds *= 2
        nh, dh = self._heads(ch, nh)
        mid = [("res", ch, ch), ("st", ch, nh, dh), ("res", ch, ch)]
        outb = []
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(self.num_res_blocks + 1):
                ich = chans.pop()
                layers = [("res", ch + ich, mc * mult)]
                ch = mc * mult
                if ds in self.attention_resolutions:
                    nh, dh = self._heads(ch, nh)
                    layers.append(("st", ch, nh, dh))
                if level and i == self.num_res_blocks:
                    layers.append(("resup", ch, ch) if self.resblock_updown else ("up", ch))
                    ds //= 2
                outb.append(layers)
        return inb, mid, outb

    def _named_layers(self):
        for i, blk in enumerate(self.input_blocks):
            for j, layer in enumerate(blk):
                yield f"input_blocks.{i}.{j}.", layer
        for j, layer in enumerate(self.middle_block):
            yield f"middle_block.{j}.", layer
        for i, blk in enumerate(self.output_blocks):
            for j, layer in enumerate(blk):
                yield f"output_blocks.{i}.{j}.", layer

    def parameter_shapes(self):
        """name -> shape, in the reference's naming."""
        mc, ted, ctx = self.model_channels, self.time_embed_dim, self.context_dim
        s = {"time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
             "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,)}
        if self.num_classes is not None:
            s["label_emb.embedding_table"] = (self.num_classes, ted)
        ssn = 2 if self.use_scale_shift_norm else 1
        for pre, layer in self._named_layers():
            kind = layer[0]
            if kind == "conv":
                s[pre + "conv.weight"] = (layer[2], layer[1], 3, 3)
                s[pre + "conv.bias"] = (layer[2],)
            elif kind in ("res", "resdown", "resup"):
                cin, cout = layer[1], layer[2]
                s[pre + "in_layers_norm.gamma"] = (cin,)
                s[pre + "in_layers_norm.beta"] = (cin,)
                s[pre + "in_layers_conv.conv.weight"] = (cout, cin, 3, 3)
                s[pre + "in_layers_conv.conv.bias"] = (cout,)
                s[pre + "emb_layers.1.weight"] = (ssn * cout, ted)
                s[pre + "emb_layers.1.bias"] = (ssn * cout,)
                s[pre + "out_layers_norm.gamma"] = (cout,)
                s[pre + "out_layers_norm.beta"] = (cout,)
                s[pre + "out_layers_conv.conv.weight"] = (cout, cout, 3, 3)
                s[pre + "out_layers_conv.conv.bias"] = (cout,)
                if cin != cout:
                    s[pre + "skip_connection.conv.weight"] = (cout, cin, 1, 1)
                    s[pre + "skip_connection.conv.bias"] = (cout,)
            elif kind == "st":
                ch, inner = layer[1], layer[2] * layer[3]
                s[pre + "norm.gamma"] = (ch,)
                s[pre + "norm.beta"] = (ch,)
                s[pre + "proj_in.weight"] = (inner, ch) if self.use_linear else (inner, ch, 1, 1)
                s[pre + "proj_in.bias"] = (inner,)
                s[pre + "proj_out.weight"] = (ch, inner) if self.use_linear else (ch, inner, 1, 1)
                s[pre + "proj_out.bias"] = (ch,)
                for k in range(self.transformer_depth):
                    t = pre + f"transformer_blocks.{k}."
                    for a, cd in (("attn1.", inner), ("attn2.", ctx)):
                        s[t + a + "to_q.weight"] = (inner, inner)
                        s[t + a + "to_k.weight"] = (inner, cd)
                        s[t + a + "to_v.weight"] = (inner, cd)
                        s[t + a + "to_out.0.weight"] = (inner, inner)
                        s[t + a + "to_out.0.bias"] = (inner,)
                    s[t + "ff.net.0.proj.weight"] = (inner * 8, inner)
                    s[t + "ff.net.0.proj.bias"] = (inner * 8,)
                    s[t + "ff.net.2.weight"] = (inner, inner * 4)
                    s[t + "ff.net.2.bias"] = (inner,)
                    for n in ("norm1", "norm2", "norm3"):
                        s[t + n + ".gamma"] = (inner,)
                        s[t + n + ".beta"] = (inner,)
            elif kind == "down" and self.conv_resample:     # Downsample(use_conv=False) is a parameter-free average pool
                s[pre + "op.conv.weight"] = (layer[1], layer[1], 3, 3)
                s[pre + "op.conv.bias"] = (layer[1],)
            elif kind == "up" and self.conv_resample:
                s[pre + "conv.conv.weight"] = (layer[1], layer[1], 3, 3)
                s[pre + "conv.conv.bias"] = (layer[1],)
        s["out.0.gamma"] = (mc,)
        s["out.0.beta"] = (mc,)
        s["out.2.conv.weight"] = (self.out_channels, mc, 3, 3)
        s["out.2.conv.bias"] = (self.out_channels,)
        if self.n_embed is not None:        # the reference builds `out` as well (openaimodel.py:520-531); only id_predictor runs
            s["id_predictor.0.gamma"] = (mc,)
            s["id_predictor.0.beta"] = (mc,)
            s["id_predictor.1.conv.weight"] = (self.n_embed, mc, 1, 1)
            s["id_predictor.1.conv.bias"] = (self.n_embed,)
        return s

    # ------------------------------------------------------------------ weights
    def _dev(self, a, dtype):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _pack_conv(self, wt, cin_pad=None, cout_pad=None):
        """[Cout,Cin,kh,kw] -> the packed GEMM weight storage (ops.pack_conv_weight, include/mdx.h)."""
        wt = wt if isinstance(wt, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(wt))
        return ops.pack_conv_weight(wt.to(self.device), cin_pad, cout_pad)

    def _pack_dense(self, wt):
        """nn.Dense weight [out,in] -> packed GEMM weight storage."""
        return ops.pack_gemm_weight(self._dev(wt, f16))

    def _pad_vec(self, v, n):
        v = self._dev(v, f32)
        if v.numel() == n:
            return v
        out = torch.zeros(n, dtype=f32, device=self.device)
        out[: v.numel()] = v
        return out

    def load_state_dict(self, params, strict=True):
        """params: name -> array/tensor keyed by the reference's parameter names.  Packs everything into
        the kernels' layouts on the device (fp16 weights, fp32 biases / norm affine)."""
        shapes = self.parameter_shapes()
        # every owned parameter is needed to run: missing keys always raise; strict additionally rejects unexpected ones
        check_state_dict(shapes, {k: v for k, v in params.items() if strict or k in shapes}, True, "UNetModel.load_state_dict")
        P = params
        w = {}
        w["te0.w"] = self._dev(P["time_embed.0.weight"], f16)
        w["te0.b"] = self._dev(P["time_embed.0.bias"], f32)
        w["te2.w"] = self._dev(P["time_embed.2.weight"], f16)
        w["te2.b"] = self._dev(P["time_embed.2.bias"], f32)
        if self.num_classes is not None:
            w["label_emb"] = self._dev(P["label_emb.embedding_table"], f32)
        emb_w, emb_b, self._emb_off = [], [], {}
        off = 0
        for pre, layer in self._named_layers():
            kind = layer[0]
            if kind == "conv":
                w[pre + "w"] = self._pack_conv(P[pre + "conv.weight"], cin_pad=self.cin_pad)
                w[pre + "b"] = self._dev(P[pre + "conv.bias"], f32)
            elif kind in ("res", "resdown", "resup"):
                cin, cout = layer[1], layer[2]
                for n in ("in_layers_norm", "out_layers_norm"):
                    w[pre + n + ".g"] = self._dev(P[pre + n + ".gamma"], f32)
                    w[pre + n + ".b"] = self._dev(P[pre + n + ".beta"], f32)
                w[pre + "conv1.w"] = self._pack_conv(P[pre + "in_layers_conv.conv.weight"])
                w[pre + "conv1.b"] = self._dev(P[pre + "in_layers_conv.conv.bias"], f32)
                w[pre + "conv2.w"] = self._pack_conv(P[pre + "out_layers_conv.conv.weight"])
                w[pre + "conv2.b"] = self._dev(P[pre + "out_layers_conv.conv.bias"], f32)
                if cin != cout:
                    w[pre + "skip.w"] = self._pack_conv(P[pre + "skip_connection.conv.weight"])
                    w[pre + "skip.b"] = self._dev(P[pre + "skip_connection.conv.bias"], f32)
                emb_w.append(self._dev(P[pre + "emb_layers.1.weight"], f16))
                emb_b.append(self._dev(P[pre + "emb_layers.1.bias"], f32))
                self._emb_off[pre] = off
                off += cout * (2 if self.use_scale_shift_norm else 1)
            elif kind == "st":
                inner = layer[2] * layer[3]
                w[pre + "norm.g"] = self._dev(P[pre + "norm.gamma"], f32)
                w[pre + "norm.b"] = self._dev(P[pre + "norm.beta"], f32)
                for n in ("proj_in", "proj_out"):
                    wt = self._dev(P[pre + n + ".weight"], f16)
                    w[pre + n + ".w"] = self._pack_dense(wt.reshape(wt.shape[0], wt.shape[1]))  # 1x1 conv == Dense in NHWC
                    w[pre + n + ".b"] = self._dev(P[pre + n + ".bias"], f32)
                for k in range(self.transformer_depth):
                    t = pre + f"transformer_blocks.{k}."
                    # self-attention: ONE [q | k | v] projection launch; the q|k columns are stored row-major and the v columns
                    # transposed (mdx_gemm_desc.n_split), which needs 2 * inner to be a multiple of 128
                    wq, wk, wv = (self._dev(P[t + f"attn1.to_{n}.weight"], f16) for n in "qkv")
                    for n in ("norm1", "norm2", "norm3"):
                        w[t + n + ".g"] = self._dev(P[t + n + ".gamma"], f32)
                        w[t + n + ".b"] = self._dev(P[t + n + ".beta"], f32)
                    # LayerNorm fold (mdx_gemm_desc.ln_stats): norm1/2/3 disappear into the GEMMs around them -- the consumer
                    # weights become gamma (.) W, with S = row sums and W beta (+ b) as the bias (ops.fold_layernorm)
                    fold = inner % 64 == 0 and os.environ.get("MDX_UNET_LN_FOLD", "1") != "0"

                    def put(name, wt, norm, bias=None):
                        if fold:
                            wt, w[name + ".s"], w[name + ".cb"] = ops.fold_layernorm(wt, w[t + norm + ".g"], w[t + norm + ".b"], bias)
                        w[name + ".w"] = self._pack_dense(wt)
                    if (2 * wq.shape[0]) % 128 == 0 and os.environ.get("MDX_UNET_QKV_MERGE", "1") != "0":
                        put(t + "attn1.qkv", torch.cat([wq, wk, wv], 0), "norm1")
                    else:   # fall back to a [q | k] launch and a transposed-store v launch
                        w[t + "attn1.qk.w"] = self._pack_dense(torch.cat([wq, wk], 0))
                        w[t + "attn1.v.w"] = self._pack_dense(wv)
                    put(t + "attn2.q", self._dev(P[t + "attn2.to_q.weight"], f16), "norm2")
                    w[t + "attn2.k.w"] = self._pack_dense(P[t + "attn2.to_k.weight"])
                    w[t + "attn2.v.w"] = self._pack_dense(P[t + "attn2.to_v.weight"])
                    for a in ("attn1", "attn2"):
                        w[t + a + ".o.w"] = self._pack_dense(P[t + a + ".to_out.0.weight"])
                        w[t + a + ".o.b"] = self._dev(P[t + a + ".to_out.0.bias"], f32)
                    # GEGLU (attention.py:41-51): interleave 64 'x' rows with their 64 'gate' rows per 128-wide tile
                    gw = self._dev(P[t + "ff.net.0.proj.weight"], f16)
                    gb = self._dev(P[t + "ff.net.0.proj.bias"], f32)
                    half = 4 * inner
                    assert half % 64 == 0
                    nt = half // 64
                    w[t + "ff1.b"] = torch.stack([gb[:half].reshape(nt, 64), gb[half:].reshape(nt, 64)], 1).reshape(-1).contiguous()
                    put(t + "ff1", torch.stack([gw[:half].reshape(nt, 64, inner), gw[half:].reshape(nt, 64, inner)], 1)
                        .reshape(2 * half, inner), "norm3", w[t + "ff1.b"])
                    w[t + "ff2.w"] = self._pack_dense(P[t + "ff.net.2.weight"])
                    w[t + "ff2.b"] = self._dev(P[t + "ff.net.2.bias"], f32)
                    # row-local fused tail (mdx_st_tail_f16): to_out1 .. proj_out as one launch where a level has enough
                    # token rows to fill the chip; its own packing (MFMA-fragment-major per-wave streams, unfolded LayerNorms)
                    if (self.transformer_depth == 1 and inner == layer[1]
                            and ops.st_tail_supported(inner, layer[2], layer[3], 64, 64)):
                        po = self._dev(P[pre + "proj_out.weight"], f16)
                        w[t + "tail.stream"], w[t + "tail.vec"] = ops.pack_st_tail(
                            self._dev(P[t + "attn1.to_out.0.weight"], f16), self._dev(P[t + "attn2.to_q.weight"], f16),
                            self._dev(P[t + "attn2.to_out.0.weight"], f16), gw, self._dev(P[t + "ff.net.2.weight"], f16),
                            po.reshape(po.shape[0], po.shape[1]),
                            w[t + "attn1.o.b"], w[t + "norm2.g"], w[t + "norm2.b"], w[t + "attn2.o.b"], w[t + "norm3.g"],
                            w[t + "norm3.b"], gb, w[t + "ff2.b"], w[pre + "proj_out.b"])
                        if ops.st_head_supported(inner, 64, 64):
                            pi = self._dev(P[pre + "proj_in.weight"], f16)
                            w[t + "head.stream"], w[t + "head.vec"] = ops.pack_st_head(
                                pi.reshape(pi.shape[0], pi.shape[1]), wq, wk, wv, w[pre + "norm.g"], w[pre + "norm.b"],
                                w[pre + "proj_in.b"], w[t + "norm1.g"], w[t + "norm1.b"])
            elif kind == "down" and self.conv_resample:
                w[pre + "w"] = self._pack_conv(P[pre + "op.conv.weight"])
                w[pre + "b"] = self._dev(P[pre + "op.conv.bias"], f32)
            elif kind == "up" and self.conv_resample:
                w[pre + "w"] = self._pack_conv(P[pre + "conv.conv.weight"])
                w[pre + "b"] = self._dev(P[pre + "conv.conv.bias"], f32)
                if layer[1] % 64 == 0 and ops.get_option("unet_subpixel_upsample"):
                    # sub-pixel form of nearest-2x + conv (mdx_gemm_desc.w_sub): 4 Cin instead of 9 Cin products per output; the
                    # library uses it where the eight-wave conv core applies and falls back to `w` + the upsampling gather elsewhere
                    wt = P[pre + "conv.conv.weight"]
                    wt = wt if isinstance(wt, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(wt))
                    w[pre + "wsub"] = ops.pack_subpixel_conv_weight(wt.to(self.device))
        w["emb.w"] = torch.cat(emb_w, 0).contiguous()
        w["emb.b"] = torch.cat(emb_b, 0).contiguous()
        self._emb_total = off
        if self.n_embed is None:
            w["out.g"] = self._dev(P["out.0.gamma"], f32)
            w["out.b"] = self._dev(P["out.0.beta"], f32)
            w["out.w"] = self._pack_conv(P["out.2.conv.weight"], cout_pad=self.cout_pad)
            w["out.cb"] = self._pad_vec(P["out.2.conv.bias"], self.cout_pad)
        else:
            w["out.g"] = self._dev(P["id_predictor.0.gamma"], f32)
            w["out.b"] = self._dev(P["id_predictor.0.beta"], f32)
            w["out.w"] = self._pack_conv(P["id_predictor.1.conv.weight"], cout_pad=self.cout_pad)
            w["out.cb"] = self._pad_vec(P["id_predictor.1.conv.bias"], self.cout_pad)
        self.w = w
        self._frag_w = {}
        self._plans = {}
        self._ctx_key = None
        self._ctx_ref = None
        return self

    def weight_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.w.values())

    # ------------------------------------------------------------------ planning
    class _Plan:
        pass

    def _plan(self, B, H, W, _fuse_head=True, _selfctx=False):
        # _selfctx: the `context=None` form of construct() (DiffusionWrapper keys None / 'concat' / 'adm', WK ddpm.py:361-374):
        # attn2 attends to its own normalised input (attention.py:133 `context = default(context, x)`), so its to_k / to_v run per
        # step on LayerNorm(tokens) instead of once on the text context; plans of the two forms are kept side by side
        key = (B, H, W, "selfctx") if _selfctx else (B, H, W)
        if key in self._plans:
            return self._plans[key]
        if self.w is None:
            raise MdxError("UNetModel: load_state_dict() must be called before the first forward")
        for lvl in range(len(self.channel_mult) - 1):
            if (H >> lvl) % 2 or (W >> lvl) % 2:
                raise MdxError(f"UNetModel: latent {H}x{W} is not divisible by 2^{len(self.channel_mult) - 1}")
        dev = self.device
        w = self.w
        P = UNetModel._Plan()
        A = _Arena(dev)
        main, ctxops, descs = [], [], []
        meta = []      # parallel to `main`: {"kind", "flops", "launches"} for profiling / roofline accounting
        gn_need = [0]

        def emit(fn, kind, flops=0, launches=1, info=""):
            main.append(fn)
            meta.append({"kind": kind, "flops": int(flops), "launches": launches, "info": info})
        P.x_static = torch.zeros((B, self.in_channels, H, W), dtype=f32, device=dev)
        P.t_static = torch.zeros((B,), dtype=f32, device=dev)
        TC = self.max_context_len
        P.B, P.H, P.W = B, H, W

        producer = {}     # device address of a tensor -> the GEMM descriptor that wrote it last (planning order == run order)
        _arena_get = A.get

        def _get(shape, dtype=f16):     # a buffer handed out again is no longer "the output of that GEMM"
            t = _arena_get(shape, dtype)
            producer.pop(t.data_ptr(), None)
            return t
        A.get = _get
        op_index = {}     # descriptor -> index of its op in `main`

        def add_gemm(oplist, **kw):
            d = ops.make_gemm_desc(**kw)
            d._w_tensor = kw["w"]       # (python-side attribute: the packed weight tensor this descriptor points at)
            descs.append(d)
            if oplist is main:
                producer[kw["out"].data_ptr()] = d
                op_index[ctypes.addressof(d)] = len(main)
            fn = (lambda d=d: ops.gemm_run(d))
            if oplist is main:
                ks, st, up = kw.get("ksize", 1), kw.get("stride", 1), kw.get("upsample", 0)
                hs_, ws2 = (2 * kw["H"], 2 * kw["W"]) if up else (kw["H"], kw["W"])
                pad = 1 if ks == 3 else 0
                m_rows = kw["B"] * ((hs_ + 2 * pad - ks) // st + 1) * ((ws2 + 2 * pad - ks) // st + 1)
                kdim = ks * ks * (kw["c1"] + kw.get("c2", 0))
                emit(fn, "gemm", 2 * m_rows * kw["N"] * kdim, 1, f"M={m_rows} N={kw['N']} K={kdim} k{ks}s{st}u{up}")
                meta[-1]["desc"] = d        # launches / split are filled in by ops.account_gemm_launches below
            else:
                oplist.append(fn)

        gn_calls = []

        def add_gn(x1, x2, g, b, eps, silu, out, scale=None, shift=None):
            Bq, HW, C1 = x1.shape
            C2 = 0 if x2 is None else x2.shape[2]
            gn_need[0] = max(gn_need[0], ops.groupnorm_ws_floats(Bq, HW, C1 + C2))
            call = dict(x1=x1, x2=x2, g=g, b=b, eps=eps, silu=silu, out=out, cs=None, meta=len(meta), film=scale is not None,
                        prod=(producer.get(x1.data_ptr()), None if x2 is None else producer.get(x2.data_ptr())))
            producer.pop(out.data_ptr(), None)      # the GroupNorm output is not a GEMM output
            gn_calls.append(call)

            def run(c=call):
                if c.get("fs") is not None:     # the producer deferred its split-K reduce to this GroupNorm (one launch for both)
                    return ops.groupnorm_from_splitk(c["fs"], c["g"], c["b"], c["eps"], c["silu"], c["out"])
                if c["cs"] is not None:     # statistics from the producers' epilogues: one launch, one read of x
                    cs1, n1, cs2, n2 = c["cs"]
                    return ops.groupnorm_colstats(c["x1"], cs1, n1, c["x2"], cs2, n2, c["g"], c["b"], c["eps"], c["silu"],
                                                  out=c["out"], scale=scale, shift=shift,
                                                  mod_ld=self._emb_total if scale is not None else 0)
                if scale is not None:       # use_scale_shift_norm: GN(h) * (1 + scale) + shift (openaimodel.py:193-198)
                    return ops.groupnorm_scaleshift(c["x1"], c["x2"], c["g"], c["b"], scale, shift, self._emb_total, c["eps"],
                                                    c["silu"], ws=P.gn_ws, out=c["out"])
                return ops.groupnorm(c["x1"], c["x2"], c["g"], c["b"], c["eps"], c["silu"], ws=P.gn_ws, out=c["out"])
            emit(run, "groupnorm", 0, 2, f"B={Bq} HW={HW} C={C1 + C2}")

        # ---- time embedding (openaimodel.py:550-551, 150-157): 4 tiny launches
        mc, ted = self.model_channels, self.time_embed_dim
        t_emb = torch.empty((B, mc), dtype=f32, device=dev)
        e1 = torch.empty((B, ted), dtype=f32, device=dev)
        emb = torch.empty((B, ted), dtype=f32, device=dev)
        P.emb_all = torch.empty((B, self._emb_total), dtype=f32, device=dev)
        emit(lambda: ops.timestep_embedding(P.t_static, mc, out=t_emb), "small")
        emit(lambda: ops.dense_small(t_emb, w["te0.w"], w["te0.b"], act_out=True, out=e1), "small", 2 * B * mc * ted)
        emit(lambda: ops.dense_small(e1, w["te2.w"], w["te2.b"], out=emb), "small", 2 * B * ted * ted)
        if self.num_classes is not None:    # emb + label_emb(y) (openaimodel.py:552-554): a [B, ted] row gather, outside the graph
            P.y_static = torch.zeros((B,), dtype=torch.long, device=dev)
            emit(lambda: emb.add_(w["label_emb"].index_select(0, P.y_static)), "small")
        emit(lambda: ops.dense_small(emb, w["emb.w"], w["emb.b"], act_in=True, out=P.emb_all), "small",
             2 * B * ted * self._emb_total)
        P.temb_ops = len(main)   # main[:temb_ops] only fill P.emb_all: skipped when the caller hands the rows in

        xin = A.get((B, H * W, self.cin_pad))
        emit(lambda: ops.nchw_to_nhwc(P.x_static, self.cin_pad, out=xin), "small")
        # checkpoint of the guidance-duplicate prefix (_dup_body): the ops of the first conv and of the first self-attention, and the
        # tensors that are live behind the latter
        ck = {"xin": xin}

        P.ctx_pad = None
        P.attn_ws = None         # split-KV attention workspace (ops.attention_workspace), sized after the walk
        attn_ws_need = [0]
        ctx_kv = {}
        xattn_descs = []

        def conv3(src, cin, cout, wt, bias, h, wd, stride=1, upsample=0, rowbias=None, residual=None, src2=None, c2=0,
                  skip=None, gn=None, wsub=None):
            """skip = (x, x2, c1, c2, packed 1x1 weights): the ResBlock's skip_connection rides on this launch as extra K
            tiles (mdx_gemm_desc.skip_w); `bias` then holds the sum of both convs' biases."""
            hs, ws_ = (2 * h, 2 * wd) if upsample else (h, wd)
            ho, wo = (hs + 2 - 3) // stride + 1, (ws_ + 2 - 3) // stride + 1
            out = A.get((B, ho * wo, cout))
            kw = {}
            if skip is not None:
                kw = dict(skip_a=skip[0], skip_a2=skip[1], skip_c1=skip[2], skip_c2=skip[3], skip_w=skip[4])
            if gn is not None:      # GroupNorm + SiLU of `src` inside the conv; the statistics pointer is wired after planning
                kw.update(gn_gamma=gn[0], gn_beta=gn[1], gn_eps=gn[2], gn_silu=1)
            if wsub is not None:
                kw["w_sub"] = wsub
            add_gemm(main, a=src, w=wt, N=cout, B=B, H=h, W=wd, c1=cin - c2, out=out, out_ld=cout, a2=src2, c2=c2,
                     bias=bias, rowbias=rowbias, rowbias_ld=self._emb_total if rowbias is not None else 0,
                     residual=residual, residual_ld=cout if residual is not None else 0, ksize=3, stride=stride,
                     upsample=upsample, **kw)
            if skip is not None:
                meta[-1]["flops"] += 2 * B * ho * wo * cout * (skip[2] + skip[3])
                meta[-1]["info"] += f" +skip1x1 K={skip[2] + skip[3]}"
            if gn is not None:
                dd = descs[-1]
                gn_convs.append(dd)
                meta[-1]["info"] += " +groupnorm(in)"
                gn_calls.append(dict(x1=src, x2=None, conv=dd, meta=len(meta) - 1, film=False,
                                     prod=(producer.get(src.data_ptr()), None)))
            return out, ho, wo

        def dense(oplist, src, rows_b, tokens, cin, nout, wt, bias=None, residual=None, epilogue=ops.EPI_NONE,
                  out=None, out_ld=None, out_mode=ops.OUT_ROWMAJOR, src2=None, c2=0, arena=True, **fold):
            cols = nout // 2 if epilogue == ops.EPI_GEGLU else nout
            if out is None:
                out = A.get((rows_b, tokens, cols))
                out_ld = cols
            add_gemm(oplist, a=src, w=wt, N=nout, B=rows_b, H=tokens, W=1, c1=cin - c2, out=out, out_ld=out_ld,
                     a2=src2, c2=c2, bias=bias, residual=residual, residual_ld=cols if residual is not None else 0,
                     epilogue=epilogue, out_mode=out_mode, **fold)
            return out

        ln_stats = {}
        tails = []

        def tail_rows(t, n, heads, dh):
            """Rows per block of the fused SpatialTransformer tail for this block, or 0 = unfused launches.  The fused launch
            needs enough row blocks to fill the chip: 32-row blocks from 192 blocks up (UNet batch 2 at a 64 x 64 latent = 256),
            64-row blocks (half the weight traffic through L2) once those alone give >= 2 blocks per CU.
            ops.set_option("unet_st_tail", 0 | 32 | 64) forces a choice (0 = never)."""
            if (t + "tail.stream") not in w or self.transformer_depth != 1 or _selfctx:
                return 0
            if TC > 96 or TC % 8:       # mdx_st_tail_f16 holds the context keys in registers: capacity <= 96, multiple of 8
                return 0                # (a UNet built with a longer max_context_len keeps the unfused launches, any length)
            forced = ops.get_option("unet_st_tail")
            cands = [forced] if forced in (32, 64) else ([] if forced == 0 else [64, 32])
            for r in cands:
                if not ops.st_tail_supported(heads * dh, heads, dh, n, r):
                    continue
                blocks = B * n // r
                if forced in (32, 64) or (r == 64 and blocks >= 512) or (r == 32 and blocks >= 192):
                    return r
            return 0

        heads_fused = []

        def head_rows(t, x, n, heads, dh):
            """Rows per block of the fused SpatialTransformer head (GroupNorm .. q|k|v^T), or 0.  Only together with the fused
            tail, and only when x's producer is a GEMM / conv launch (its epilogue supplies the GroupNorm column partials)."""
            if not _fuse_head or (t + "head.stream") not in w or ops.get_option("unet_st_head") == 0:
                return 0
            r = tail_rows(t, n, heads, dh)
            if not r or not ops.st_head_supported(heads * dh, n, r) or producer.get(x.data_ptr()) is None:
                return 0
            return r

        def fused_tail(t, rows_t, o, tok, x, ch, inner, heads, dh, n):
            """Everything after the self-attention core of block `t` + proj_out + the residual: ONE row-local launch."""
            kc = torch.zeros((B, TC, inner), dtype=f16, device=dev)
            vtc = torch.zeros((B, inner, TC), dtype=f16, device=dev)
            ctx_kv[t] = (kc, vtc)
            out = A.get((B, n, ch))
            td = ops.make_st_tail_desc(o, tok, x, out, kc, vtc, w[t + "tail.stream"], w[t + "tail.vec"], B, n, ch, heads,
                                       dh, 1, TC, tile_rows=rows_t)
            tails.append(td)
            td._bufs = (o, tok, x, out, kc, vtc)      # keeps the views alive; tools / tests read them
            producer[out.data_ptr()] = td

            def run_tail(td=td):
                td.ctx_len = P.ctx_len      # read at call time like the attention ops; a captured graph bakes it in
                ops.st_tail_run(td)
            emit(run_tail, "gemm", 2 * B * n * 16 * inner * inner + 4 * B * heads * n * 77 * dh, 1,
                 f"st_tail M={B * n} C={inner} rows={rows_t} (to_out1..proj_out fused)")
            return out

        def stats_buf(rows, width):
            """{sum, sumsq} per 64-column slice of a token row: written by the GEMM that produces the rows, read by the
            GEMM that consumes LayerNorm(rows) (mdx_gemm_desc.stats_out / ln_stats).  One buffer per shape: the stream is
            in order and every consumer runs before the next producer."""
            if (rows, width) not in ln_stats:
                ln_stats[(rows, width)] = torch.zeros((rows, width // 64, 2), dtype=f32, device=dev)
            return ln_stats[(rows, width)]

        gn_convs = []

        def gn_conv_ok(src, c_in, c_out, hh, ww, wt):
            """Can GroupNorm + SiLU of `src` run inside the 3x3 conv that consumes it (mdx_gemm_desc.gn_colstats)?  At the levels
            with at least `unet_gn_conv_fuse` output rows (below that the GroupNorm launch doubles as the split-K reduce of the
            conv in front of it), for inputs some GEMM launch produced (its epilogue supplies the statistics), convs that
            resolve to the HALO kernel with 64-column tiles."""
            mrows = ops.get_option("unet_gn_conv_fuse")
            if not _fuse_head or not mrows or B * hh * ww < mrows or c_in % 64 or c_in > 640 or (hh == 8 and ww == 8):
                return False
            if producer.get(src.data_ptr()) is None:
                return False
            probe = ops.make_gemm_desc(a=src, w=wt, N=c_out, B=B, H=hh, W=ww, c1=c_in, out=src, out_ld=c_out, ksize=3)
            q = ops.gemm_query(probe)
            return q[3] == 1 and q[1] == 64

        def skip_fusable(a2, c1, c2, cout, ho, wo, wt):
            """Can the ResBlock's 1x1 skip_connection ride on its second conv (mdx_gemm_desc.skip_w)?  Channel counts in whole
            64-channel K tiles, and the conv must resolve to the HALO 3x3 kernel."""
            if not ops.get_option("unet_skip_fuse") or c1 % 64 or c2 % 64 or cout % 64:
                return False
            probe = ops.make_gemm_desc(a=a2, w=wt, N=cout, B=B, H=ho, W=wo, c1=cout, out=a2, out_ld=cout, ksize=3)
            return ops.gemm_query(probe)[3] == 1

        def resblock(pre, x, x2, cin, cout, h, wd, mode=None):
            """ResBlock.construct openaimodel.py:176-205; x2 = skip tensor of the (virtual) concat.  mode 'up' / 'down' is the
            resblock_updown form: nearest-2x / 2x2 average pooling of BOTH the normalised branch and the skip input
            (the nearest-2x of the branch is folded into conv1's gather)."""
            c2 = 0 if x2 is None else x2.shape[2]
            hw = h * wd
            eoff = self._emb_off[pre]
            film = self.use_scale_shift_norm
            rowbias = None if film else P.emb_all[:, eoff:eoff + cout]  # view: pointer = base + eoff, ld = emb_total
            gn1 = x2 is None and mode is None and gn_conv_ok(x, cin, cout, h, wd, w[pre + "conv1.w"])
            a = None
            if not gn1:
                a = A.get((B, hw, cin))
                add_gn(x, x2, w[pre + "in_layers_norm.g"], w[pre + "in_layers_norm.b"], 1e-5, True, a)
            if mode == "up":
                assert x2 is None and cin == cout
                hbuf, ho, wo = conv3(a, cin, cout, w[pre + "conv1.w"], w[pre + "conv1.b"], h, wd, upsample=1, rowbias=rowbias)
                xs = A.get((B, ho * wo, cin))
                emit(lambda x=x, xs=xs: ops.upsample_nearest2x(x, B, h, wd, cin, out=xs), "small")
            elif mode == "down":
                assert x2 is None and cin == cout
                ap = A.get((B, hw // 4, cin))
                emit(lambda a=a, ap=ap: ops.avgpool2x2(a, B, h, wd, cin, out=ap), "small")
                hbuf, ho, wo = conv3(ap, cin, cout, w[pre + "conv1.w"], w[pre + "conv1.b"], h // 2, wd // 2, rowbias=rowbias)
                A.release(ap)
                xs = A.get((B, hw // 4, cin))
                emit(lambda x=x, xs=xs: ops.avgpool2x2(x, B, h, wd, cin, out=xs), "small")
            elif gn1:     # GroupNorm + SiLU of x inside conv1 (mdx_gemm_desc.gn_colstats): no GroupNorm launch, no normalised copy
                hbuf, ho, wo = conv3(x, cin, cout, w[pre + "conv1.w"], w[pre + "conv1.b"], h, wd, rowbias=rowbias,
                                     gn=(w[pre + "in_layers_norm.g"], w[pre + "in_layers_norm.b"], 1e-5))
                xs = x
            else:
                hbuf, ho, wo = conv3(a, cin, cout, w[pre + "conv1.w"], w[pre + "conv1.b"], h, wd, rowbias=rowbias)
                xs = x
            if a is not None:
                A.release(a)
            gn2 = (not film and mode is None) and gn_conv_ok(hbuf, cout, cout, ho, wo, w[pre + "conv2.w"])
            if gn2:
                # out_layers: GroupNorm + SiLU of conv1's output inside conv2 (statistics from conv1's epilogue)
                gnp = (w[pre + "out_layers_norm.g"], w[pre + "out_layers_norm.b"], 1e-5)
                if cin != cout and skip_fusable(hbuf, cin - c2, c2, cout, ho, wo, w[pre + "conv2.w"]):
                    if (pre + "conv2skip.b") not in w:
                        w[pre + "conv2skip.b"] = (w[pre + "conv2.b"] + w[pre + "skip.b"]).contiguous()
                    out, _, _ = conv3(hbuf, cout, cout, w[pre + "conv2.w"], w[pre + "conv2skip.b"], ho, wo, gn=gnp,
                                      skip=(x, x2, cin - c2, c2, w[pre + "skip.w"]))
                    A.release(hbuf)
                    return out, ho, wo
                if cin != cout:
                    skip = dense(main, x, B, hw, cin, cout, w[pre + "skip.w"], bias=w[pre + "skip.b"], src2=x2, c2=c2)
                else:
                    assert x2 is None
                    skip = xs
                out, _, _ = conv3(hbuf, cout, cout, w[pre + "conv2.w"], w[pre + "conv2.b"], ho, wo, residual=skip, gn=gnp)
                A.release(hbuf)
                if skip is not x:
                    A.release(skip)
                return out, ho, wo
            a2 = A.get((B, ho * wo, cout))
            if film:
                add_gn(hbuf, None, w[pre + "out_layers_norm.g"], w[pre + "out_layers_norm.b"], 1e-5, True, a2,
                       scale=P.emb_all[:, eoff:eoff + cout], shift=P.emb_all[:, eoff + cout:eoff + 2 * cout])
            else:
                add_gn(hbuf, None, w[pre + "out_layers_norm.g"], w[pre + "out_layers_norm.b"], 1e-5, True, a2)
            A.release(hbuf)
            if cin != cout and mode is None and skip_fusable(a2, cin - c2, c2, cout, ho, wo, w[pre + "conv2.w"]):
                # skip_connection (1x1 over the raw input, openaimodel.py:174) as extra K tiles of conv2: one launch less per
                # ResBlock whose channel count changes, and the skip tensor never exists
                if (pre + "conv2skip.b") not in w:
                    w[pre + "conv2skip.b"] = (w[pre + "conv2.b"] + w[pre + "skip.b"]).contiguous()
                out, _, _ = conv3(a2, cout, cout, w[pre + "conv2.w"], w[pre + "conv2skip.b"], ho, wo,
                                  skip=(x, x2, cin - c2, c2, w[pre + "skip.w"]))
                A.release(a2)
                return out, ho, wo
            if cin != cout:
                skip = dense(main, x, B, hw, cin, cout, w[pre + "skip.w"], bias=w[pre + "skip.b"], src2=x2, c2=c2)
            else:
                assert x2 is None
                skip = xs
            out, _, _ = conv3(a2, cout, cout, w[pre + "conv2.w"], w[pre + "conv2.b"], ho, wo, residual=skip)
            A.release(a2)
            if skip is not x:
                A.release(skip)
            return out, ho, wo

        ragged_vt = []

        def vt_buffer(n, inner):
            """V^T [B, inner, row length] for attention.  Token counts that are not a multiple of 8 (5 x 5 ... 7 x 7 images at the
            deepest level of 320 / 384 / 448-pixel runs; the reference takes any multiple of 64 pixels) get rows padded to 8 in a
            DEDICATED zeroed buffer: the transposed store never writes the pad, the attention kernel's 16-byte V^T loads read it
            (times a zero probability), so it must stay finite -- an arena buffer would hand it another tensor's bytes."""
            if n % 8 == 0:
                return A.get((B, inner, n))
            t_ = torch.zeros((B, inner, _round_up(n, 8)), dtype=f16, device=dev)
            ragged_vt.append(t_)
            return t_

        def vt_release(t_):
            if not any(t_ is r for r in ragged_vt):
                A.release(t_)

        def transformer(pre, x, ch, heads, dh, h, wd):
            """SpatialTransformer.construct attention.py:237-256 + `transformer_depth` BasicTransformerBlocks :181-185
            (NHWC == tokens)."""
            n = h * wd
            inner = heads * dh
            scale = dh ** -0.5
            t0 = pre + "transformer_blocks.0."
            rows_h = head_rows(t0, x, n, heads, dh)
            if rows_h:
                # fused head + fused tail: GroupNorm .. q|k|v^T in one launch, the attention core, to_out1 .. proj_out in one
                tok = A.get((B, n, inner))
                qk = A.get((B, n, 2 * inner))
                vt = A.get((B, inner, n))
                hd = ops.make_st_head_desc(x, x, 1, w[t0 + "head.stream"], w[t0 + "head.vec"], tok, qk, vt, n, B, n, ch,
                                           tile_rows=rows_h)
                hd.colstats = 0      # wired by ops.wire_groupnorm_colstats from x's producer (or the plan is rebuilt without it)
                heads_fused.append(hd)
                hd._bufs = (x, tok, qk, vt)
                gn_calls.append(dict(x1=x, x2=None, head=hd, meta=len(meta), film=False,
                                     prod=(producer.get(x.data_ptr()), None)))
                emit(lambda hd=hd: ops.st_head_run(hd), "gemm", 2 * B * n * 4 * inner * inner, 1,
                     f"st_head M={B * n} C={inner} rows={rows_h} (GroupNorm..q|k|v fused)")
                o = A.get((B, n, inner))
                attn_ws_need[0] = max(attn_ws_need[0], ops.attention_ws_bytes(B, heads, dh, n, n))
                emit(lambda qk=qk, vt=vt, o=o: ops.attention(
                    qk.data_ptr(), qk.data_ptr() + inner * 2, vt.data_ptr(), o.data_ptr(), B, heads, dh, n, n, scale,
                    n * 2 * inner, 2 * inner, n * 2 * inner, 2 * inner, inner * n, n, n * inner, inner, ws=P.attn_ws),
                    "attention", 4 * B * heads * n * n * dh, 1, f"self B={B} h={heads} N={n} d={dh}")
                if "op" not in ck and "conv_in" in ck and len(hs) == 1:     # (only the first conv's output is held as a skip)
                    ck.update(op=main[-1], live=(x, tok, o))
                out = fused_tail(t0, tail_rows(t0, n, heads, dh), o, tok, x, ch, inner, heads, dh, n)
                A.release(qk); A.release(vt); A.release(tok); A.release(o)
                return out
            a = A.get((B, n, ch))
            add_gn(x, None, w[pre + "norm.g"], w[pre + "norm.b"], 1e-6, False, a)
            # LayerNorm fold: `st` receives the row statistics from each producer of the token stream
            st = stats_buf(B * n, inner) if (t0 + "attn2.q.s") in w else None
            fold1 = st is not None and (t0 + "attn1.qkv.s") in w

            def consumer(name):   # kwargs of a GEMM that consumes LN(rows) with folded weights
                return dict(bias=w[name + ".cb"], ln_stats=st, ln_s=w[name + ".s"], ln_eps=1e-5)
            tok = dense(main, a, B, n, ch, inner, w[pre + "proj_in.w"], bias=w[pre + "proj_in.b"],
                        stats_out=st if fold1 else None)
            # SpatialTransformer.norm has no activation (attention.py:243-247): proj_in can apply it to its A fragments from the
            # producer's column statistics (mdx_gemm_desc.gn_colstats on a dense launch).  Decided when the statistics are wired
            # (ops.wire_groupnorm_colstats): on success the GroupNorm op above is dropped and proj_in reads the raw x
            pf = ops.get_option("unet_gn_proj_fuse")
            if (_fuse_head and pf and n >= pf and n % 64 == 0 and ch % 64 == 0 and ch <= 2560
                    and producer.get(x.data_ptr()) is not None):
                gn_calls[-1]["proj"] = dict(desc=descs[-1], meta=len(meta) - 1)
            A.release(a)
            for k in range(self.transformer_depth):
                t = pre + f"transformer_blocks.{k}."
                last = k == self.transformer_depth - 1
                # --- attn1 (self)
                ln = A.get((B, n, inner))
                if not fold1:
                    emit(lambda ln=ln, tok=tok, t=t: ops.layernorm(tok, w[t + "norm1.g"], w[t + "norm1.b"], 1e-5, out=ln),
                         "layernorm")
                vt = vt_buffer(n, inner)
                nv = vt.shape[2]        # row length of V^T: n, or n rounded up to 8 (ragged_vt)
                if (t + "attn1.qkv.w") in w:
                    qk = A.get((B, n, 2 * inner))
                    add_gemm(main, a=tok if fold1 else ln, w=w[t + "attn1.qkv.w"], N=3 * inner, B=B, H=n, W=1, c1=inner, out=qk,
                             out_ld=2 * inner, out2=vt, out2_ld=nv, n_split=2 * inner,
                             **(consumer(t + "attn1.qkv") if fold1 else {}))
                else:
                    qk = dense(main, ln, B, n, inner, 2 * inner, w[t + "attn1.qk.w"])
                    dense(main, ln, B, n, inner, inner, w[t + "attn1.v.w"], out=vt, out_ld=nv, out_mode=ops.OUT_TRANSPOSED)
                o = ln  # reuse: ln is dead after the projections
                attn_ws_need[0] = max(attn_ws_need[0], ops.attention_ws_bytes(B, heads, dh, n, n))
                emit(lambda qk=qk, vt=vt, o=o, nv=nv: ops.attention(
                    qk.data_ptr(), qk.data_ptr() + inner * 2, vt.data_ptr(), o.data_ptr(), B, heads, dh, n, n, scale,
                    n * 2 * inner, 2 * inner, n * 2 * inner, 2 * inner, inner * nv, nv, n * inner, inner, ws=P.attn_ws),
                    "attention", 4 * B * heads * n * n * dh, 1, f"self B={B} h={heads} N={n} d={dh}")
                first = "op" not in ck and "conv_in" in ck and len(hs) == 1     # (only the first conv's output is held as a skip)
                rows_t = tail_rows(t, n, heads, dh)
                if first:       # splice point "behind the self-attention": every plan has it
                    ck.update(op=main[-1], live=(x, tok, o))
                if rows_t:
                    out = fused_tail(t, rows_t, o, tok, x, ch, inner, heads, dh, n)
                    A.release(qk); vt_release(vt); A.release(tok); A.release(ln)
                    return out
                tok2 = dense(main, o, B, n, inner, inner, w[t + "attn1.o.w"], bias=w[t + "attn1.o.b"], residual=tok,
                             stats_out=st)
                if first:       # attn1's output projection (+ the row statistics attn2.q's LayerNorm fold reads) is still context-free:
                    # a later splice point, used when BOTH plans of a guidance-duplicate pair have it (the fused tail starts at to_out)
                    ck.update(op2=main[-1], live2=(x, tok2) + (() if st is None else (st,)))
                A.release(qk); vt_release(vt); A.release(tok)
                # --- attn2 (cross): K / V^T of the context are produced by the context plan
                # (round 6) head dim 64 (SDv2): the 77-key attention rides on the query projection as its EPILOGUE -- one 64-column
                # tile is one head (mdx_gemm_desc.xattn_k): no attention launch, no fp16 round trip of q; bit-identical to the two launches
                xfuse = (ops.get_option("unet_xattn_fuse") and not _selfctx and dh == 64 and TC <= 128 and TC % 8 == 0
                         and n % 64 == 0)
                if xfuse:
                    kc = torch.zeros((B, TC, inner), dtype=f16, device=dev)
                    vtc = torch.zeros((B, inner, TC), dtype=f16, device=dev)
                    ctx_kv[t] = (kc, vtc)
                    xkw = dict(tile_n=64, splitk=1, tile_m=0 if n % 128 == 0 else 64, xattn_k=kc, xattn_vt=vtc, xattn_len=TC, xattn_cap=TC, xattn_scale=scale, out=o,
                               out_ld=inner)
                if st is None:
                    emit(lambda ln=ln, tok2=tok2, t=t: ops.layernorm(tok2, w[t + "norm2.g"], w[t + "norm2.b"], 1e-5, out=ln),
                         "layernorm")
                    q2 = dense(main, ln, B, n, inner, inner, w[t + "attn2.q.w"], **(xkw if xfuse else {}))
                else:
                    q2 = dense(main, tok2, B, n, inner, inner, w[t + "attn2.q.w"], **consumer(t + "attn2.q"), **(xkw if xfuse else {}))
                    if _selfctx:    # k / v below read LayerNorm(tok2) itself: one explicit launch (this form is not a hot path)
                        emit(lambda ln=ln, tok2=tok2, t=t: ops.layernorm(tok2, w[t + "norm2.g"], w[t + "norm2.b"], 1e-5, out=ln),
                             "layernorm")
                if xfuse:
                    xattn_descs.append(descs[-1])      # (their xattn_len follows the context: _ensure_context)
                    meta[-1]["flops"] += 4 * B * heads * n * 77 * dh
                    meta[-1]["info"] += f" +cross-attention h={heads} d={dh}"
                    q2 = None
                if _selfctx:
                    # context = default(context, x) (attention.py:133): keys / values are projections of attn2's own input
                    k2 = dense(main, ln, B, n, inner, inner, w[t + "attn2.k.w"])
                    v2t = vt_buffer(n, inner)
                    nv2 = v2t.shape[2]
                    dense(main, ln, B, n, inner, inner, w[t + "attn2.v.w"], out=v2t, out_ld=nv2, out_mode=ops.OUT_TRANSPOSED)
                    attn_ws_need[0] = max(attn_ws_need[0], ops.attention_ws_bytes(B, heads, dh, n, n))
                    emit(lambda q2=q2, k2=k2, v2t=v2t, o=o, nv2=nv2: ops.attention(
                        q2.data_ptr(), k2.data_ptr(), v2t.data_ptr(), o.data_ptr(), B, heads, dh, n, n, scale,
                        n * inner, inner, n * inner, inner, inner * nv2, nv2, n * inner, inner, ws=P.attn_ws),
                        "attention", 4 * B * heads * n * n * dh, 1, f"attn2-self B={B} h={heads} N={n} d={dh}")
                    A.release(k2); vt_release(v2t)
                elif not xfuse:
                    kc = torch.zeros((B, TC, inner), dtype=f16, device=dev)
                    vtc = torch.zeros((B, inner, TC), dtype=f16, device=dev)
                    ctx_kv[t] = (kc, vtc)
                    emit(lambda q2=q2, kc=kc, vtc=vtc, o=o: ops.attention(
                        q2.data_ptr(), kc.data_ptr(), vtc.data_ptr(), o.data_ptr(), B, heads, dh, n, P.ctx_len, scale,
                        n * inner, inner, TC * inner, inner, inner * TC, TC, n * inner, inner),
                        "attention", 4 * B * heads * n * 77 * dh, 1, f"cross B={B} h={heads} N={n} d={dh}")
                tok3 = dense(main, o, B, n, inner, inner, w[t + "attn2.o.w"], bias=w[t + "attn2.o.b"], residual=tok2,
                             stats_out=st)
                if q2 is not None:
                    A.release(q2)
                A.release(tok2)
                # --- feed-forward (GEGLU fused in the first GEMM's epilogue)
                if st is None:
                    emit(lambda ln=ln, tok3=tok3, t=t: ops.layernorm(tok3, w[t + "norm3.g"], w[t + "norm3.b"], 1e-5, out=ln),
                         "layernorm")
                    g = dense(main, ln, B, n, inner, 8 * inner, w[t + "ff1.w"], bias=w[t + "ff1.b"], epilogue=ops.EPI_GEGLU)
                else:
                    g = dense(main, tok3, B, n, inner, 8 * inner, w[t + "ff1.w"], epilogue=ops.EPI_GEGLU, **consumer(t + "ff1"))
                # the next block's norm1 reads its row statistics from this block's last producer
                tok = dense(main, g, B, n, 4 * inner, inner, w[t + "ff2.w"], bias=w[t + "ff2.b"], residual=tok3,
                            stats_out=st if (fold1 and not last) else None)
                A.release(g); A.release(tok3); A.release(ln)
            out = dense(main, tok, B, n, inner, ch, w[pre + "proj_out.w"], bias=w[pre + "proj_out.b"], residual=x)
            A.release(tok)
            return out

        # ---- context plan: to_k / to_v of attn2 for every SpatialTransformer (attention.py:119-121)
        P.ctx_pad = torch.zeros((B, TC, self.context_dim), dtype=f16, device=dev)
        P.ctx_len = 0

        # ---- walk the UNet (openaimodel.py:556-576)
        h, wd = H, W
        hs = []
        cur = None

        def held(t):        # still needed as a skip connection?
            return any(t is s_[0] for s_ in hs)

        def layer_op(pre, layer, cur, skip, h, wd):
            kind = layer[0]
            if kind == "res":
                return resblock(pre, cur, skip, layer[1], layer[2], h, wd)
            if kind in ("resdown", "resup"):
                return resblock(pre, cur, None, layer[1], layer[2], h, wd, mode=kind[3:])
            if kind == "st":
                return transformer(pre, cur, layer[1], layer[2], layer[3], h, wd), h, wd
            if kind == "down":          # Downsample openaimodel.py:63-88
                if self.conv_resample:
                    return conv3(cur, layer[1], layer[1], w[pre + "w"], w[pre + "b"], h, wd, stride=2)
                new = A.get((B, (h // 2) * (wd // 2), layer[1]))
                emit(lambda cur=cur, new=new: ops.avgpool2x2(cur, B, h, wd, layer[1], out=new), "small")
                return new, h // 2, wd // 2
            if kind == "up":            # Upsample openaimodel.py:33-60 (the nearest-2x is folded into the conv's gather)
                if self.conv_resample:
                    return conv3(cur, layer[1], layer[1], w[pre + "w"], w[pre + "b"], h, wd, upsample=1, wsub=w.get(pre + "wsub"))
                new = A.get((B, 4 * h * wd, layer[1]))
                emit(lambda cur=cur, new=new: ops.upsample_nearest2x(cur, B, h, wd, layer[1], out=new), "small")
                return new, 2 * h, 2 * wd
            raise ValueError(kind)

        for i, blk in enumerate(self.input_blocks):
            for j, layer in enumerate(blk):
                pre = f"input_blocks.{i}.{j}."
                if layer[0] == "conv":
                    cur, h, wd = conv3(xin, self.cin_pad, layer[2], w[pre + "w"], w[pre + "b"], h, wd)
                    ck["conv_in"] = main[-1]
                    A.release(xin)
                    continue
                new, h2, w2 = layer_op(pre, layer, cur, None, h, wd)
                if not held(cur):
                    A.release(cur)
                cur, h, wd = new, h2, w2
            hs.append((cur, h, wd))
        ck.setdefault("op", None)      # (no self-attention in input block 1: no guidance-duplicate prefix)
        for j, layer in enumerate(self.middle_block):
            new, h, wd = layer_op(f"middle_block.{j}.", layer, cur, None, h, wd)
            if not held(cur):
                A.release(cur)
            cur = new
        for i, blk in enumerate(self.output_blocks):
            skip, sh, sw = hs.pop()
            assert (sh, sw) == (h, wd)
            for j, layer in enumerate(blk):
                new, h, wd = layer_op(f"output_blocks.{i}.{j}.", layer, cur, skip if j == 0 else None, h, wd)
                A.release(cur)
                if j == 0:
                    A.release(skip)
                cur = new
        a = A.get((B, h * wd, mc))
        # `out` = GroupNorm -> SiLU -> 3x3 conv; predict_codebook_ids: `id_predictor` = GroupNorm -> 1x1 conv (:520-531, 573-576)
        add_gn(cur, None, w["out.g"], w["out.b"], 1e-5, self.n_embed is None, a)
        P.eps_nhwc = torch.empty((B, h * wd, self.cout_pad), dtype=f16, device=dev)
        add_gemm(main, a=a, w=w["out.w"], N=self.cout_pad, B=B, H=h, W=wd, c1=mc, out=P.eps_nhwc, out_ld=self.cout_pad,
                 bias=w["out.cb"], ksize=3 if self.n_embed is None else 1)

        for t, (kc, vtc) in ctx_kv.items():
            inner = kc.shape[2]
            add_gemm(ctxops, a=P.ctx_pad, w=w[t + "attn2.k.w"], N=inner, B=B, H=TC, W=1, c1=self.context_dim, out=kc,
                     out_ld=inner)
            add_gemm(ctxops, a=P.ctx_pad, w=w[t + "attn2.v.w"], N=inner, B=B, H=TC, W=1, c1=self.context_dim, out=vtc,
                     out_ld=TC, out_mode=ops.OUT_TRANSPOSED)

        # shared workspaces (sized for the hungriest op), patched into every descriptor
        need = max([ops.gemm_workspace_bytes(d) for d in descs] + [0])
        P.gemm_ws = ops.new_gemm_workspace(need, dev)
        for d in descs:
            d.workspace = P.gemm_ws.data_ptr()
            d.workspace_bytes = P.gemm_ws.numel() * 4
        P.gn_ws = torch.empty(max(gn_need[0], 4), dtype=f32, device=dev)
        if attn_ws_need[0]:
            P.attn_ws = ops.attention_workspace(attn_ws_need[0], dev)
        # ---- weight-streaming form of the small-M 3x3 convs (mdx_gemm_desc.w_frag): at M = 128 (the 8 x 8 level at UNet batch
        # 2) a conv is a 30 MB weight stream with almost no arithmetic; the launches that resolve to 128 x 64 HALO tiles read a
        # fragment-major copy of their weights straight into registers, twelve 1 KiB pieces in flight per wave.  Measured
        # (round 3, op profile): M = 128 convs 20.3 -> 18.8 us, M = 512 convs 30.0 -> 31.0 us (each piece is fetched by the two
        # waves that share its columns, which halves the unique bytes in flight): default threshold 128.
        if ops.get_option("unet_conv_stream"):
            for d in descs:
                M = d.B * d.H * d.W      # (stride 1: output rows)
                if not (d.ksize == 3 and d.stride == 1 and not d.upsample and d.c2 == 0 and d.c1 % 64 == 0 and d.N % 64 == 0
                        and M <= ops.get_option("unet_conv_stream") and d.out_mode == ops.OUT_ROWMAJOR and not d.skip_w
                        and not d.gn_gamma):
                    continue
                q = ops.gemm_query(d)
                w4 = ops.get_option("unet_conv_stream_w4")
                if w4 and q[3] == 1 and d.N % 128 == 0 and d.c1 // 64 >= 5 and not d.colstats_out and not d.defer_reduce:
                    # side-by-side waves on 128-column tiles (every weight piece fetched once per block), slab split-K
                    keep = (d.tile_m, d.tile_n, d.splitk)
                    d.tile_m, d.tile_n, d.splitk = 128, 128, min(d.c1 // 64, w4)
                    q = ops.gemm_query(d)
                    if not (q[0] == 128 and q[1] == 128 and q[3] == 1 and q[2] > 4 and not q[6]):
                        d.tile_m, d.tile_n, d.splitk = keep
                        q = ops.gemm_query(d)
                if not (q[0] == 128 and q[1] in (64, 128) and q[3] == 1):
                    continue
                if q[1] == 128 and not (q[2] > 1 and not q[6]):
                    continue
                wkey = d._w_tensor.data_ptr()
                if wkey not in self._frag_w:    # (one fragment-major copy per weight, shared by the plans of every shape)
                    self._frag_w[wkey] = ops.pack_frag_weight(ops.unpack_gemm_weight(d._w_tensor, d.N, 9 * d.c1)).reshape(-1)
                d.w, d.w_frag = self._frag_w[wkey].data_ptr(), 1
        # ---- GroupNorm statistics from the producers (mdx_gemm_desc.colstats_out): every GroupNorm input of the UNet is a conv
        # / Dense output (openaimodel.py:136,159,521; attention.py:83), so the launch that stores it can also emit per-column
        # {sum, sumsq} of each of its row blocks; the GroupNorm then folds those instead of re-reading the tensor (gn_stats
        # disappears: one launch and one read instead of two).  Done for the tensors that take the two-launch path today
        # (>= 1024 pixels per sample); the small deep-level tensors already use the one-launch fused kernel.
        P.colstats = {}
        # The one-launch GroupNorm of the deep levels can also BE the split-K reduce of the conv right in front of it
        # (mdx_gemm_desc.defer_reduce): -22 launches per evaluation at equal time (ops._OPTIONS["unet_gn_splitk_fuse"]).
        fuse_hw = ops.get_option("unet_gn_splitk_fuse")     # fuse for tensors of at most this many pixels per sample (0 = never)
        if fuse_hw:
            for c in gn_calls:
                if c.get("head") is not None or c.get("conv") is not None or c.get("proj") is not None:
                    continue
                _, HW, C1 = c["x1"].shape
                if HW > fuse_hw:
                    continue
                cpg = C1 // 32
                L = cpg // math.gcd(cpg, 8)
                d = c["prod"][0]
                if (c["x2"] is None and not c["film"] and L <= 64 and HW * L * 16 <= (64 << 10) and isinstance(d, ops.GemmDesc)
                        and d.N == C1
                        and d.out_ld == C1 and op_index.get(ctypes.addressof(d)) == c["meta"] - 1
                        and ops.gemm_query(d)[2] > 1 and ops.groupnorm_from_splitk_ok(d)):
                    d.defer_reduce = 1
                    c["fs"] = d
                    meta[c["meta"] - 1]["launches"] = 1
        if os.environ.get("MDX_UNET_GN_COLSTATS", "1") != "0":
            ops.wire_groupnorm_colstats(gn_calls, meta, B, dev, P.colstats)
        else:
            ops.wire_groupnorm_colstats([], meta, B, dev, P.colstats)
            for c in gn_calls:      # launch accounting of the one-launch fused kernel
                _, HW, C1 = c["x1"].shape
                cpg = (C1 + (0 if c["x2"] is None else c["x2"].shape[2])) // 32
                L = cpg // math.gcd(cpg, 8)
                if L <= 64 and HW * L * 16 <= (64 << 10):
                    meta[c["meta"]]["launches"] = 1
        if any(not hd.colstats for hd in heads_fused) or any(not dd.gn_colstats for dd in gn_convs):
            # a fused head whose input tensor's producer cannot emit column statistics in its final launch form: plan again
            # with the unfused GroupNorm / proj_in / qkv launches (plans are built once per shape)
            return self._plan(B, H, W, _fuse_head=False, _selfctx=_selfctx)
        if any(m.get("dead") for m in meta):      # GroupNorm launches that moved into the GEMM behind them
            keep = [i for i, m in enumerate(meta) if not m.get("dead")]
            main[:] = [main[i] for i in keep]
            meta[:] = [meta[i] for i in keep]
        # ---- first-use tuning (off by default): a resolution / batch the tile table was not measured at runs the cost model's
        # tiles, 10-20 % off on some shapes; with the option on, every such launch form is timed once per shape (ops.tune_cache)
        if ops.get_option("unet_tune_first_use"):
            tws = ops.new_gemm_workspace(256 << 20, dev)
            for d in descs:
                d.workspace, d.workspace_bytes = tws.data_ptr(), tws.numel() * 4
            P.tuned_shapes = ops.tune_untuned(descs)
            torch.cuda.synchronize()
            del tws
            ops.release_tune_scratch()
            for d in descs:
                d.workspace, d.workspace_bytes = P.gemm_ws.data_ptr(), P.gemm_ws.numel() * 4
        # the statistics epilogue is part of the tile table's launch-variant key: a wired producer may resolve to another row
        # (another split) than the one the shared workspace was sized for -- size it again and grow it if needed
        need2 = max([ops.gemm_workspace_bytes(d) for d in descs] + [0])
        if need2 > P.gemm_ws.numel() * 4:
            P.gemm_ws = ops.new_gemm_workspace(need2, dev)
            for d in descs:
                d.workspace = P.gemm_ws.data_ptr()
                d.workspace_bytes = P.gemm_ws.numel() * 4
        ops.check_colstats_wiring(descs)
        ops.account_gemm_launches(meta)     # last: the column-statistics wiring above can change a launch's table row
        P.main, P.ctxops, P.descs, P.meta = main, ctxops, descs, meta
        assert len(main) == len(meta)
        P.arena_bytes = A.total
        P.ln_stats = ln_stats
        P.arena = A   # owns the activation buffers (descriptors only hold raw device pointers)
        P.tails = tails
        P.ragged_vt = ragged_vt     # (owned by the plan: descriptors hold raw pointers)
        P.heads_fused = heads_fused
        P.ctx_kv = ctx_kv     # the cached context K / V^T buffers: descriptors hold raw pointers only
        P.xattn_descs = xattn_descs
        P.ck = ck if (ck.get("op") is not None and not _selfctx) else None
        P.graph = None
        P.dup_graph = None
        P.graph_failed = False
        self._plans[key] = P
        return P

    # ------------------------------------------------------------------ execution
    def _ensure_context(self, P, context):
        """Project the text context through every attn2.to_k / to_v once per context tensor: it is constant
        across the sampling loop (the reference recomputes it at 16 sites x 51 calls)."""
        # The cache is tied to the tensor OBJECT (weak reference) and its version counter, not to its address: a later
        # conditioning tensor can be allocated at the same address with the same version, and must not hit the cache.
        key = (context.data_ptr(), context._version, tuple(context.shape), context.dtype, id(P))
        alive = self._ctx_ref() if self._ctx_ref is not None else None
        if key == self._ctx_key and alive is context:
            return
        Bc, T, Dc = context.shape
        if Bc != P.B or Dc != self.context_dim:
            raise MdxError(f"context shape {tuple(context.shape)} does not match batch {P.B} / context_dim {self.context_dim}")
        if T > self.max_context_len:
            raise MdxError(f"context length {T} > max_context_len={self.max_context_len}")
        P.ctx_pad.zero_()
        P.ctx_pad[:, :T].copy_(context)
        P.ctx_len = T
        for d in getattr(P, "xattn_descs", ()):      # the query projections that carry their cross-attention (mdx_gemm_desc.xattn_len)
            d.xattn_len = T
        for op in P.ctxops:
            op()
        self._ctx_key = key
        self._ctx_ref = weakref.ref(context)
        # the attention ops read P.ctx_len at call time; a captured graph bakes it in
        if getattr(P, "graph_ctx_len", None) != T:
            P.graph = None
            P.dup_graph = None

    def time_embedding_table(self, t):
        """Everything the UNet derives from the timestep alone -- sinusoid, time_embed MLP and the 22 ResBlock
        emb_layers (openaimodel.py:550-551, 150-157, 188) -- for ALL the steps of a sampling run in one batched pass:
        t [S] -> [S, emb_total] fp32.  A sampler computes it once per sample() (the 51 MB of emb weights are then read
        once per run instead of once per step) and passes row i to forward_nhwc(..., temb=row)."""
        if self.w is None:
            raise MdxError("UNetModel: load_state_dict() must be called before time_embedding_table()")
        if self.num_classes is not None:
            raise MdxError("UNetModel: a class-conditional UNet adds label_emb(y) before the ResBlock projections; "
                           "there is no timestep-only table")
        t = torch.as_tensor(t, dtype=f32).to(self.device).reshape(-1).contiguous()
        w = self.w
        e0 = ops.timestep_embedding(t, self.model_channels)
        e1 = ops.dense_small(e0, w["te0.w"], w["te0.b"], act_out=True)
        e2 = ops.dense_small(e1, w["te2.w"], w["te2.b"])
        return ops.dense_small(e2, w["emb.w"], w["emb.b"], act_in=True)

    def forward_nhwc(self, x, timesteps, context=None, temb=None, y=None, cfg_dup=False):
        """Run the UNet; returns the plan's static NHWC fp16 eps buffer [B, H*W, 8] (first 4 channels valid).
        The buffer is overwritten by the next call.  temb: optional row(s) of time_embedding_table() for `timesteps`
        ([emb_total] or [B, emb_total]); the time-embedding launches are then skipped.  y: [B] class labels of a
        class-conditional UNet (num_classes).  cfg_dup: the caller states that the two halves of the batch carry the SAME x and
        timesteps and differ only in `context` (classifier-free guidance, plms.py:192-195 `x_in = cat([x] * 2)`): the launches in
        front of the first cross-attention may then run on one half (_dup_body)."""
        if (y is not None) != (self.num_classes is not None):      # openaimodel.py:545-547
            raise MdxError("UNetModel: must specify y if and only if the model is class-conditional")
        if y is not None and temb is not None:
            raise MdxError("UNetModel: time_embedding_table() rows do not carry the label embedding; pass timesteps with y")
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise MdxError("UNetModel: x must be a CUDA(HIP) tensor (no CPU fallback)")
        B, C, H, W = x.shape
        if C != self.in_channels:
            raise MdxError(f"UNetModel: expected {self.in_channels} input channels, got {C}")
        if context is None:
            # construct(x, t) / construct(x, t, y=) of DiffusionWrapper keys None / 'concat' / 'adm' (WK ddpm.py:361-374): attn2's
            # to_k / to_v (Dense(context_dim, inner)) then see the block's own tokens, which only type-checks -- in the reference
            # as here -- when context_dim equals the transformer width at every attention level
            if getattr(self, "_selfctx_bad", None) is None:      # (structure and context_dim are fixed at construction: checked once,
                self._selfctx_bad = sorted({l[2] * l[3] for _, l in self._named_layers()      # not on every sampler step)
                                            if l[0] == "st" and l[2] * l[3] != self.context_dim})
            bad = self._selfctx_bad
            if bad:
                raise MdxError(f"UNetModel: context=None makes attn2 attend to its own input (attention.py:133), which needs "
                               f"context_dim == transformer width; context_dim={self.context_dim}, widths {bad}")
            P = self._plan(B, H, W, _selfctx=True)
        else:
            P = self._plan(B, H, W)
            self._ensure_context(P, context)
        P.x_static.copy_(x)
        if temb is not None:
            if temb.shape[-1] != self._emb_total or temb.dtype != f32:
                raise MdxError(f"UNetModel: temb must be fp32 [.., {self._emb_total}] rows of time_embedding_table()")
            P.emb_all.copy_(temb)          # [emb_total] broadcasts over the batch
        else:
            P.t_static.copy_(timesteps.to(device=self.device, dtype=f32) if isinstance(timesteps, torch.Tensor)
                             else torch.as_tensor(timesteps, dtype=f32, device=self.device))
            if y is not None:
                yy = torch.as_tensor(y).to(device=self.device, dtype=torch.long).reshape(-1)
                if yy.shape[0] != B:
                    raise MdxError(f"UNetModel: y has {yy.shape[0]} labels for a batch of {B}")
                if int(yy.min()) < 0 or int(yy.max()) >= self.num_classes:
                    raise MdxError(f"UNetModel: class label outside [0, {self.num_classes})")
                P.y_static.copy_(yy)
            for op in P.main[:P.temb_ops]:
                op()
        dup = self._dup_body(P) if (cfg_dup and context is not None and y is None) else None
        if dup is not None and ops.get_option("unet_cfg_dup_check"):
            hb = B // 2
            same = torch.equal(P.x_static[:hb], P.x_static[hb:]) and torch.equal(P.emb_all[:hb], P.emb_all[hb:])
            if not same:
                raise MdxError("UNetModel: cfg_dup=True, but the two halves of the batch do not carry the same x / timesteps")
        if self.use_graph and not P.graph_failed:
            if dup is not None:
                if P.dup_graph is None:
                    self._capture(P, dup)
                if P.dup_graph is not None:
                    P.dup_graph.replay()
                    return P.eps_nhwc
            if P.graph is None:
                self._capture(P)
            if P.graph is not None:
                P.graph.replay()
                return P.eps_nhwc
        body = dup if dup is not None else P.main[P.temb_ops:]
        for op in body:
            op()
        self.last_launch_count = P.temb_ops + len(body)
        return P.eps_nhwc

    def _dup_body(self, P):
        """The op list of one evaluation whose batch is [uncond ; cond] of the SAME latents (classifier-free guidance, plms.py:192-195):
        until the first cross-attention both halves compute the same numbers -- conv_in, the first ResBlock, the first
        SpatialTransformer's GroupNorm / proj_in / qkv, its self-attention (at 64^2 .. 96^2 tokens the largest attention of the
        network) and, where it is a launch of its own in both plans, attn1's output projection.  Those launches run from the plan of
        HALF the batch; its live tensors (the ResBlock output, the token stream and the attention output -- or the token stream behind
        to_out and its LayerNorm row statistics) are then written to both halves of this plan's buffers and this plan continues
        with what follows.  conv_in itself runs at the full batch (its output is the outermost skip connection, with column
        statistics for the last GroupNorm: writing it costs what copying it would).  None when the option is off
        (ops option unet_cfg_dup = smallest batch, 0 = never), the batch is odd or the network has no attention at its first level."""
        mn = ops.get_option("unet_cfg_dup")
        if not mn or P.B < mn or P.B % 2 or P.ck is None:
            return None
        if hasattr(P, "dup_body"):
            return P.dup_body
        P.dup_body = None
        h = P.B // 2
        PA = self._plan(h, P.H, P.W)
        if PA.ck is None:
            return None
        # the splice point: behind attn1's output projection when both plans launch it (a plan whose fused tail starts at to_out
        # does not), else behind the self-attention
        late = "op2" in P.ck and "op2" in PA.ck and len(P.ck["live2"]) == len(PA.ck["live2"])
        ko, kl = ("op2", "live2") if late else ("op", "live")
        ib, ic = P.main.index(P.ck[ko]), P.main.index(P.ck["conv_in"])
        ja, jc = PA.main.index(PA.ck[ko]), PA.main.index(PA.ck["conv_in"])
        xin_a, xin_b = PA.ck["xin"], P.ck["xin"]
        copies = [lambda: xin_a.copy_(xin_b[:h]), lambda: PA.emb_all.copy_(P.emb_all[:h])]
        spread = []
        for src, dst in zip(PA.ck[kl], P.ck[kl]):
            if not (dst.shape[0] == 2 * src.shape[0] and dst.shape[1:] == src.shape[1:] and dst.dtype == src.dtype):
                raise MdxError(f"UNetModel: guidance-duplicate prefix: live tensors of the two plans do not pair up "
                               f"({tuple(src.shape)} vs {tuple(dst.shape)})")
            spread.append(lambda src=src, dst=dst, k=src.shape[0]: (dst[:k].copy_(src), dst[k:].copy_(src)))
        P.dup_half = PA
        P.dup_body = P.main[P.temb_ops:ic + 1] + copies + PA.main[jc:ja + 1] + spread + P.main[ib + 1:]
        small = {"kind": "small", "flops": 0, "launches": 1, "info": "guidance-duplicate prefix: copy"}
        P.dup_meta = (P.meta[P.temb_ops:ic + 1] + [dict(small) for _ in copies] + PA.meta[jc:ja + 1]
                      + [dict(small, launches=2) for _ in spread] + P.meta[ib + 1:])     # (parallel to dup_body, for the profilers)
        return P.dup_body

    def _capture(self, P, dup=None):
        """Capture the whole forward as one hipGraph (kills ~450 launch gaps per call)."""
        try:
            body = dup if dup is not None else P.main[P.temb_ops:]   # the graph starts from P.emb_all (filled eagerly or from the sampler's table)
            for op in body:  # warm-up outside capture
                op()
            torch.cuda.synchronize()
            if dup is not None:
                P.dup_graph = ops.capture_graph(body)
            else:
                P.graph = ops.capture_graph(body)
            P.graph_ctx_len = P.ctx_len
        except Exception as e:  # pragma: no cover - depends on the runtime
            P.graph = None
            P.graph_failed = True
            import warnings
            warnings.warn(f"hipGraph capture failed, running eagerly: {e}")

    def construct(self, x, timesteps=None, context=None, y=None):
        """openaimodel.py:536-576.  x [N,C,H,W], timesteps [N], context [N,T,context_dim] -> eps [N,C,H,W] fp32."""
        eps = self.forward_nhwc(x, timesteps, context, y=y)
        B, _, H, W = x.shape
        return ops.nhwc_to_nchw(eps, self.final_channels, H, W)

    __call__ = construct
    forward = construct
