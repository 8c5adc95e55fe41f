"""Text2ImUNet / SuperResText2ImUNet -- MI355X-native mirrors of the reference's
Taichu-GLIDE/model/glide_text2im/model/text2im_model.py:25-238 (with unet.py:89-573, xf.py:36-154,
simple_nn.py:38-169 underneath).

Same constructor keywords as the reference's `create_model` / `create_upsample_model` call sites
(model_creator.py:51-75, 110-135) and the same call: ``net(x, timesteps, tokens, mask)`` /
``net(x, timesteps, low_res, tokens, mask)`` -> [N, 6, H, W].  Parameters load by the reference's Cell attribute
names (`load_state_dict`).  Execution is planned once per (N, H, W) into a flat list of C-ABI kernel calls
(NHWC fp16 activations, one hipGraph) exactly like the LDM UNet (ldm/modules/diffusionmodules/openaimodel.py):

  * ResBlock (unet.py:178-218): GN+SiLU -> [nearest-2x folded into the conv gather | AvgPool kernel] -> conv3x3;
    FiLM `GN(h)*(1+scale)+shift` + SiLU is ONE GroupNorm launch (mdx_groupnorm_scaleshift_f16); all ResBlocks'
    emb_layers run as one small-M GEMV; conv2 fuses bias + skip add.
  * AttentionBlock (unet.py:254-310): the legacy per-head [q|k|v] rows of `qkv` / [k|v] rows of `encoder_kv` are
    re-ordered at load time into plain q / k / v projections; text keys and image keys are written by their GEMMs
    straight into one [ctx+T] key buffer (row-major K, transposed V) so the flash-attention kernel sees a single
    key range; scale ch^-1/4 on q and k == ch^-1/2 on the logits.
  * the 16-layer text transformer runs INSIDE every step, as in the reference (the unconditional prompt is redrawn
    each step, main_funcs.py:37); its MLP uses the tanh-GELU GEMM epilogue.
"""
import math

import numpy as np
import torch

from ... import ops
from ..._lib import MdxError
from ...weights import check_state_dict
from ...ldm.modules.diffusionmodules.openaimodel import _Arena, _round_up

f16, f32 = torch.float16, torch.float32
XF_LN_EPS = 1e-7   # MindSpore nn.LayerNorm default epsilon (xf.py:26-33 passes none)


class Text2ImUNet:
    super_res = False

    def __init__(self, text_ctx, xf_width, xf_layers, xf_heads, xf_final_ln, n_vocab, in_channels=3,
                 model_channels=192, out_channels=6, num_res_blocks=3, attention_resolutions=(2, 4, 8), dropout=0.0,
                 channel_mult=(1, 2, 3, 4), use_fp16=True, num_heads=1, num_head_channels=64, num_heads_upsample=-1,
                 use_scale_shift_norm=True, resblock_updown=True, cache_text_emb=False, xf_padding=True, dtype=None,
                 image_size=None, device="cuda:0", **unused):
        if not (use_scale_shift_norm and resblock_updown and xf_final_ln and xf_padding):
            raise NotImplementedError("only the reference's shipped GLIDE options are supported "
                                      "(use_scale_shift_norm, resblock_updown, xf_final_ln, xf_padding = True)")
        if num_head_channels != 64 or xf_width // xf_heads != 64:
            raise NotImplementedError("attention head width must be 64 (default_options.py:24,33)")
        self.text_ctx, self.xf_width, self.xf_layers, self.xf_heads = text_ctx, xf_width, xf_layers, xf_heads
        self.n_vocab = n_vocab
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = tuple(attention_resolutions)
        self.channel_mult = tuple(channel_mult)
        self.num_head_channels = num_head_channels
        self.image_size = image_size
        self.device = torch.device(device)
        self.time_embed_dim = 4 * model_channels
        self.cin_pad = _round_up(in_channels, 8)
        self.cout_pad = _round_up(out_channels, 8)
        self.input_blocks, self.middle_block, self.output_blocks = self._structure()
        self.w = None
        self._plans = {}
        self.use_graph = True

    # ------------------------------------------------------------------ structure (unet.py:398-534)
    def _structure(self):
        mc, cm = self.model_channels, self.channel_mult
        ch = int(cm[0] * mc)
        inb = [[("conv", self.in_channels, ch)]]
        chans = [ch]
        ds = 1
        for level, mult in enumerate(cm):
            for _ in range(self.num_res_blocks):
                layers = [("res", ch, int(mult * mc), "")]
                ch = int(mult * mc)
                if ds in self.attention_resolutions:
                    layers.append(("attn", ch, ch // self.num_head_channels))
                inb.append(layers)
                chans.append(ch)
            if level != len(cm) - 1:
                inb.append([("res", ch, ch, "down")])
                chans.append(ch)
                ds *= 2
        mid = [("res", ch, ch, ""), ("attn", ch, ch // self.num_head_channels), ("res", ch, ch, "")]
        outb = []
        for level, mult in list(enumerate(cm))[::-1]:
            for i in range(self.num_res_blocks + 1):
                ich = chans.pop()
                layers = [("res", ch + ich, int(mc * mult), "")]
                ch = int(mc * mult)
                if ds in self.attention_resolutions:
                    layers.append(("attn", ch, ch // self.num_head_channels))
                if level and i == self.num_res_blocks:
                    layers.append(("res", ch, ch, "up"))
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
        mc, ted, xw = self.model_channels, self.time_embed_dim, self.xf_width
        s = {"time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
             "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,)}
        for pre, layer in self._named_layers():
            if layer[0] == "conv":
                s[pre + "conv.weight"] = (layer[2], layer[1], 3, 3)
                s[pre + "conv.bias"] = (layer[2],)
            elif layer[0] == "res":
                cin, cout = layer[1], layer[2]
                s[pre + "in_layers_0.gamma"] = (cin,)
                s[pre + "in_layers_0.beta"] = (cin,)
                s[pre + "in_layers_2.conv.weight"] = (cout, cin, 3, 3)
                s[pre + "in_layers_2.conv.bias"] = (cout,)
                s[pre + "emb_layers.1.weight"] = (2 * cout, ted)
                s[pre + "emb_layers.1.bias"] = (2 * cout,)
                s[pre + "out_layers_0.gamma"] = (cout,)
                s[pre + "out_layers_0.beta"] = (cout,)
                s[pre + "out_layers_3.conv.weight"] = (cout, cout, 3, 3)
                s[pre + "out_layers_3.conv.bias"] = (cout,)
                if cin != cout:
                    s[pre + "skip_connection.conv.weight"] = (cout, cin, 1, 1)
                    s[pre + "skip_connection.conv.bias"] = (cout,)
            else:
                c = layer[1]
                s[pre + "norm.gamma"] = (c,)
                s[pre + "norm.beta"] = (c,)
                s[pre + "qkv.conv.weight"] = (3 * c, c, 1)
                s[pre + "qkv.conv.bias"] = (3 * c,)
                s[pre + "encoder_kv.conv.weight"] = (2 * c, xw, 1)
                s[pre + "encoder_kv.conv.bias"] = (2 * c,)
                s[pre + "proj_out.conv.weight"] = (c, c, 1)
                s[pre + "proj_out.conv.bias"] = (c,)
        ch0 = int(self.channel_mult[0] * mc)
        s["out.0.gamma"] = (ch0,)
        s["out.0.beta"] = (ch0,)
        s["out2.conv.weight"] = (self.out_channels, ch0, 3, 3)
        s["out2.conv.bias"] = (self.out_channels,)
        for l in range(self.xf_layers):
            t = f"transformer.resblocks.{l}."
            s[t + "ln_1.gamma"] = (xw,); s[t + "ln_1.beta"] = (xw,)
            s[t + "attn.c_qkv.weight"] = (3 * xw, xw); s[t + "attn.c_qkv.bias"] = (3 * xw,)
            s[t + "attn.c_proj.weight"] = (xw, xw); s[t + "attn.c_proj.bias"] = (xw,)
            s[t + "ln_2.gamma"] = (xw,); s[t + "ln_2.beta"] = (xw,)
            s[t + "mlp.c_fc.weight"] = (4 * xw, xw); s[t + "mlp.c_fc.bias"] = (4 * xw,)
            s[t + "mlp.c_proj.weight"] = (xw, 4 * xw); s[t + "mlp.c_proj.bias"] = (xw,)
        s["final_ln.gamma"] = (xw,); s["final_ln.beta"] = (xw,)
        s["token_embedding.embedding_table"] = (self.n_vocab, xw)
        s["positional_embedding"] = (self.text_ctx, xw)
        s["padding_embedding"] = (self.text_ctx, xw)
        s["transformer_proj.weight"] = (ted, xw); s["transformer_proj.bias"] = (ted,)
        return s

    # ------------------------------------------------------------------ weights
    def _dev(self, a, dtype):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _conv(self, wt, cin_pad=None, cout_pad=None):
        return ops.pack_conv_weight(self._dev(wt, f32), cin_pad, cout_pad)

    def _dense(self, wt):
        return ops.pack_gemm_weight(self._dev(wt, f16))

    def load_state_dict(self, params, strict=True):
        shapes = self.parameter_shapes()
        check_state_dict(shapes, {k: v for k, v in params.items() if strict or k in shapes}, True,
                         f"{type(self).__name__}.load_state_dict")
        P, w = params, {}
        ted = self.time_embed_dim
        w["te0.w"] = self._dev(P["time_embed.0.weight"], f16)
        w["te0.b"] = self._dev(P["time_embed.0.bias"], f32)
        # emb = time_embed.2(e1) + transformer_proj(xf_out[:, -1])  (text2im_model.py:102-105) as ONE small GEMV
        # over the concatenated input [e1 | last token]
        w["te2proj.w"] = torch.cat([self._dev(P["time_embed.2.weight"], f16),
                                    self._dev(P["transformer_proj.weight"], f16)], 1).contiguous()
        w["te2proj.b"] = (self._dev(P["time_embed.2.bias"], f32) + self._dev(P["transformer_proj.bias"], f32)).contiguous()
        emb_w, emb_b, self._emb_off, off = [], [], {}, 0
        for pre, layer in self._named_layers():
            if layer[0] == "conv":
                w[pre + "w"] = self._conv(P[pre + "conv.weight"], cin_pad=self.cin_pad)
                w[pre + "b"] = self._dev(P[pre + "conv.bias"], f32)
            elif layer[0] == "res":
                cin, cout = layer[1], layer[2]
                w[pre + "n1.g"] = self._dev(P[pre + "in_layers_0.gamma"], f32)
                w[pre + "n1.b"] = self._dev(P[pre + "in_layers_0.beta"], f32)
                w[pre + "n2.g"] = self._dev(P[pre + "out_layers_0.gamma"], f32)
                w[pre + "n2.b"] = self._dev(P[pre + "out_layers_0.beta"], f32)
                w[pre + "conv1.w"] = self._conv(P[pre + "in_layers_2.conv.weight"])
                w[pre + "conv1.b"] = self._dev(P[pre + "in_layers_2.conv.bias"], f32)
                if len(layer) > 3 and layer[3] == "up" and cin % 64 == 0 and cout % 64 == 0 and ops.get_option("unet_subpixel_upsample"):
                    # the up-sampling ResBlock's conv1 follows a nearest-2x (unet.py:178-218): sub-pixel weights (mdx_gemm_desc.w_sub)
                    w[pre + "conv1.wsub"] = ops.pack_subpixel_conv_weight(self._dev(P[pre + "in_layers_2.conv.weight"], f32))
                w[pre + "conv2.w"] = self._conv(P[pre + "out_layers_3.conv.weight"])
                w[pre + "conv2.b"] = self._dev(P[pre + "out_layers_3.conv.bias"], f32)
                if cin != cout:
                    w[pre + "skip.w"] = self._conv(P[pre + "skip_connection.conv.weight"])
                    w[pre + "skip.b"] = self._dev(P[pre + "skip_connection.conv.bias"], f32)
                emb_w.append(self._dev(P[pre + "emb_layers.1.weight"], f16))
                emb_b.append(self._dev(P[pre + "emb_layers.1.bias"], f32))
                self._emb_off[pre] = off
                off += 2 * cout
            else:
                c, heads = layer[1], layer[2]
                w[pre + "norm.g"] = self._dev(P[pre + "norm.gamma"], f32)
                w[pre + "norm.b"] = self._dev(P[pre + "norm.beta"], f32)
                # legacy order (unet.py:293-295): rows of head h are [q(64) | k(64) | v(64)]
                qkv = self._dev(P[pre + "qkv.conv.weight"], f16).reshape(heads, 3, 64, c)
                qb = self._dev(P[pre + "qkv.conv.bias"], f32).reshape(heads, 3, 64)
                for idx, nm in enumerate("qkv"):
                    w[pre + nm + ".w"] = self._dense(qkv[:, idx].reshape(c, c))
                    w[pre + nm + ".b"] = qb[:, idx].reshape(c).contiguous()
                # (round 6) one launch for the three image-token projections: rows [q | k | v] (mdx_gemm_desc.n_split = 2 c: q | k
                # row-major, V transposed)
                w[pre + "qkv.w"] = self._dense(torch.cat([qkv[:, 0].reshape(c, c), qkv[:, 1].reshape(c, c), qkv[:, 2].reshape(c, c)], 0))
                w[pre + "qkv.b"] = torch.cat([qb[:, 0].reshape(c), qb[:, 1].reshape(c), qb[:, 2].reshape(c)], 0).contiguous()
                ekv = self._dev(P[pre + "encoder_kv.conv.weight"], f16).reshape(heads, 2, 64, self.xf_width)
                eb = self._dev(P[pre + "encoder_kv.conv.bias"], f32).reshape(heads, 2, 64)
                for idx, nm in enumerate(("ek", "ev")):
                    w[pre + nm + ".w"] = self._dense(ekv[:, idx].reshape(c, self.xf_width))
                    w[pre + nm + ".b"] = eb[:, idx].reshape(c).contiguous()
                w[pre + "proj.w"] = self._dense(self._dev(P[pre + "proj_out.conv.weight"], f16).reshape(c, c))
                w[pre + "proj.b"] = self._dev(P[pre + "proj_out.conv.bias"], f32)
        w["emb.w"] = torch.cat(emb_w, 0).contiguous()
        w["emb.b"] = torch.cat(emb_b, 0).contiguous()
        self._emb_total = off
        w["out.g"] = self._dev(P["out.0.gamma"], f32)
        w["out.b"] = self._dev(P["out.0.beta"], f32)
        w["out2.w"] = self._conv(P["out2.conv.weight"], cout_pad=self.cout_pad)
        ob = torch.zeros(self.cout_pad, dtype=f32, device=self.device)
        ob[: self.out_channels] = self._dev(P["out2.conv.bias"], f32)
        w["out2.b"] = ob
        xw, xh = self.xf_width, self.xf_heads
        for l in range(self.xf_layers):
            t = f"transformer.resblocks.{l}."
            for n in ("ln_1", "ln_2"):
                w[t + n + ".g"] = self._dev(P[t + n + ".gamma"], f32)
                w[t + n + ".b"] = self._dev(P[t + n + ".beta"], f32)
            # xf.py:79-81: qkv.view(b, ctx, heads, 3*64) then split -> rows of head h are [q | k | v]
            cq = self._dev(P[t + "attn.c_qkv.weight"], f16).reshape(xh, 3, 64, xw)
            cb = self._dev(P[t + "attn.c_qkv.bias"], f32).reshape(xh, 3, 64)
            w[t + "qk.w"] = self._dense(torch.cat([cq[:, 0].reshape(xw, xw), cq[:, 1].reshape(xw, xw)], 0))
            w[t + "qk.b"] = torch.cat([cb[:, 0].reshape(xw), cb[:, 1].reshape(xw)], 0).contiguous()
            w[t + "v.w"] = self._dense(cq[:, 2].reshape(xw, xw))
            w[t + "v.b"] = cb[:, 2].reshape(xw).contiguous()
            w[t + "proj.w"] = self._dense(P[t + "attn.c_proj.weight"])
            w[t + "proj.b"] = self._dev(P[t + "attn.c_proj.bias"], f32)
            w[t + "fc.w"] = self._dense(P[t + "mlp.c_fc.weight"])
            w[t + "fc.b"] = self._dev(P[t + "mlp.c_fc.bias"], f32)
            w[t + "fc2.w"] = self._dense(P[t + "mlp.c_proj.weight"])
            w[t + "fc2.b"] = self._dev(P[t + "mlp.c_proj.bias"], f32)
        w["final_ln.g"] = self._dev(P["final_ln.gamma"], f32)
        w["final_ln.b"] = self._dev(P["final_ln.beta"], f32)
        w["tok"] = self._dev(P["token_embedding.embedding_table"], f16)
        w["pos"] = self._dev(P["positional_embedding"], f16)
        w["pad"] = self._dev(P["padding_embedding"], f16)
        self.w = w
        self._plans = {}
        return self

    # ------------------------------------------------------------------ planning
    class _Plan:
        pass

    def _plan(self, B, H, W):
        key = (B, H, W)
        if key in self._plans:
            return self._plans[key]
        if self.w is None:
            raise MdxError("load_state_dict() must be called before the first forward")
        n_down = len(self.channel_mult) - 1
        if H % (1 << n_down) or W % (1 << n_down):
            raise MdxError(f"image {H}x{W} is not divisible by 2^{n_down}")
        dev, w = self.device, self.w
        P = Text2ImUNet._Plan()
        A = _Arena(dev)
        main, meta, descs = [], [], []
        gn_need = [4]
        ctx, xw, xh = self.text_ctx, self.xf_width, self.xf_heads
        mc, ted = self.model_channels, self.time_embed_dim
        P.B, P.H, P.W = B, H, W
        P.x_static = torch.zeros((B, 3, H, W), dtype=f32, device=dev)
        P.low_static = None
        P.t_static = torch.zeros((B,), dtype=f32, device=dev)
        P.tok_static = torch.zeros((B, ctx), dtype=torch.int32, device=dev)
        P.mask_static = torch.ones((B, ctx), dtype=torch.int32, device=dev)

        text_mode = [False]      # True while the text-only prefix of the plan is being emitted (see forward_nhwc: text_epoch)

        late_text = [False]      # True while an AttentionBlock emits its encoder_kv projections: text-only too, spliced into the prefix
        late_main, late_meta = [], []

        def emit(fn, kind, flops=0, launches=1, info=""):
            if late_text[0]:
                late_main.append(fn)
                late_meta.append({"kind": kind, "flops": int(flops), "launches": launches, "info": info, "text": True})
                return late_meta[-1]
            main.append(fn)
            meta.append({"kind": kind, "flops": int(flops), "launches": launches, "info": info, "text": text_mode[0]})
            return meta[-1]

        producer, gn_calls = {}, []     # tensor address -> the GEMM descriptor that wrote it last; GroupNorm calls
        _arena_get = A.get

        def _get(shape, dtype=f16):     # a buffer handed out again is no longer "the output of that GEMM" (pools, attention, ...)
            t = _arena_get(shape, dtype)
            producer.pop(t.data_ptr(), None)
            return t
        A.get = _get

        def gemm(**kw):
            d = ops.make_gemm_desc(**kw)
            descs.append(d)
            producer[kw["out"].data_ptr()] = d
            ks, st, up = kw.get("ksize", 1), kw.get("stride", 1), kw.get("upsample", 0)
            hs_, ws2 = (2 * kw["H"], 2 * kw["W"]) if up else (kw["H"], kw["W"])
            pad = 1 if ks == 3 else 0
            m_rows = kw["B"] * ((hs_ + 2 * pad - ks) // st + 1) * ((ws2 + 2 * pad - ks) // st + 1)
            kdim = ks * ks * (kw["c1"] + kw.get("c2", 0))
            mrec = emit(lambda d=d: ops.gemm_run(d), "gemm", 2 * m_rows * kw["N"] * kdim, 1, f"M={m_rows} N={kw['N']} K={kdim} k{ks}")
            mrec["desc"] = d                # launches / split are filled in by ops.account_gemm_launches below

        def gn(x1, x2, g, b, silu, out, scale=None, shift=None):
            Bq, HW, C1 = x1.shape
            C = C1 + (0 if x2 is None else x2.shape[2])
            gn_need[0] = max(gn_need[0], ops.groupnorm_ws_floats(Bq, HW, C))
            call = dict(x1=x1, x2=x2, cs=None, meta=len(meta), g=g, b=b, eps=1e-5,
                        prod=(producer.get(x1.data_ptr()), None if x2 is None else producer.get(x2.data_ptr())))
            producer.pop(out.data_ptr(), None)
            gn_calls.append(call)

            def run(c=call):
                if c["cs"] is not None:      # statistics from the producers' epilogues (ops.wire_groupnorm_colstats)
                    cs1, n1, cs2, n2 = c["cs"]
                    return ops.groupnorm_colstats(x1, cs1, n1, x2, cs2, n2, g, b, 1e-5, silu, out=out, scale=scale,
                                                  shift=shift, mod_ld=self._emb_total if scale is not None else 0)
                if scale is None:
                    return ops.groupnorm(x1, x2, g, b, 1e-5, silu, ws=P.gn_ws, out=out)
                return ops.groupnorm_scaleshift(x1, x2, g, b, scale, shift, self._emb_total, 1e-5, silu, ws=P.gn_ws, out=out)
            emit(run, "groupnorm", 0, 2, f"B={Bq} HW={HW} C={C}")

        def dense(src, rows_b, tokens, cin, nout, wt, bias=None, residual=None, epilogue=ops.EPI_NONE, out=None,
                  out_ld=None, out_mode=ops.OUT_ROWMAJOR, out_bs=0, src2=None, c2=0):
            if out is None:
                out = A.get((rows_b, tokens, nout))
                out_ld = nout
            # text-only launches never split K: their sums must not depend on how many prompt rows a pass carries (the whole-loop
            # tables of begin_loop run the same GEMMs on other row counts and promise the same bits)
            gemm(a=src, w=wt, N=nout, B=rows_b, H=tokens, W=1, c1=cin - c2, out=out, out_ld=out_ld, a2=src2, c2=c2,
                 bias=bias, residual=residual, residual_ld=nout if residual is not None else 0, epilogue=epilogue,
                 out_mode=out_mode, out_bs=out_bs, splitk=1 if (text_mode[0] or late_text[0]) else 0)
            return out

        def conv3(src, cin, cout, wt, bias, h, wd, upsample=0, residual=None, skip=None, wsub=None):
            """skip = (x, x2, c1, c2, packed 1x1 weights): the ResBlock's skip_connection rides on this launch as extra K tiles
            (mdx_gemm_desc.skip_w, as in the latent-diffusion planner); `bias` then holds the sum of both convs' biases."""
            ho, wo = (2 * h, 2 * wd) if upsample else (h, wd)
            out = A.get((B, ho * wo, cout))
            kw = {}
            if skip is not None:
                kw = dict(skip_a=skip[0], skip_a2=skip[1], skip_c1=skip[2], skip_c2=skip[3], skip_w=skip[4])
            if wsub is not None:
                kw["w_sub"] = wsub
            gemm(a=src, w=wt, N=cout, B=B, H=h, W=wd, c1=cin, out=out, out_ld=cout, bias=bias, residual=residual,
                 residual_ld=cout if residual is not None else 0, ksize=3, upsample=upsample, **kw)
            if skip is not None:
                meta[-1]["flops"] += 2 * B * ho * wo * cout * (skip[2] + skip[3])
                meta[-1]["info"] += f" +skip1x1 K={skip[2] + skip[3]}"
            return out, ho, wo

        def skip_fusable(a2, c1, c2, cout, ho, wo, wt):
            """unet.py:214-218 `skip_connection(x) + h`: can the 1x1 conv ride on conv2 (whole 64-channel K tiles, HALO kernel)?"""
            if not ops.get_option("unet_skip_fuse") or c1 % 64 or c2 % 64 or cout % 64:
                return False
            probe = ops.make_gemm_desc(a=a2, w=wt, N=cout, B=B, H=ho, W=wo, c1=cout, out=a2, out_ld=cout, ksize=3)
            return ops.gemm_query(probe)[3] == 1

        # ---- text transformer (text2im_model.py:88-99, xf.py:36-154) on all B rows.  It depends on the tokens alone, so it is the
        # plan's PREFIX: a caller whose tokens do not change between calls (the super-resolution loop: 27 steps on one prompt;
        # the base model's unconditional half is re-drawn every step, main_funcs.py:37) skips it (forward_nhwc text_epoch)
        text_mode[0] = True
        cat_in = torch.empty((B, ted + xw), dtype=f32, device=dev)      # [silu-free e1 | last text token]
        x_tok = A.get((B, ctx, xw))
        emit(lambda: ops.glide_text_embed(P.tok_static, P.mask_static, w["tok"], w["pos"], w["pad"], out=x_tok), "small")
        ln = A.get((B, ctx, xw))
        for l in range(self.xf_layers):
            t = f"transformer.resblocks.{l}."
            emit(lambda t=t, x_tok=x_tok: ops.layernorm(x_tok, w[t + "ln_1.g"], w[t + "ln_1.b"], XF_LN_EPS, out=ln), "layernorm")
            qk = dense(ln, B, ctx, xw, 2 * xw, w[t + "qk.w"], bias=w[t + "qk.b"])
            vt = A.get((B, xw, ctx))
            dense(ln, B, ctx, xw, xw, w[t + "v.w"], bias=w[t + "v.b"], out=vt, out_ld=ctx, out_mode=ops.OUT_TRANSPOSED)
            ao = A.get((B, ctx, xw))
            emit(lambda qk=qk, vt=vt, ao=ao: ops.attention(
                qk.data_ptr(), qk.data_ptr() + xw * 2, vt.data_ptr(), ao.data_ptr(), B, xh, 64, ctx, ctx, 64 ** -0.5,
                ctx * 2 * xw, 2 * xw, ctx * 2 * xw, 2 * xw, xw * ctx, ctx, ctx * xw, xw),
                "attention", 4 * B * xh * ctx * ctx * 64)
            x2 = dense(ao, B, ctx, xw, xw, w[t + "proj.w"], bias=w[t + "proj.b"], residual=x_tok)
            A.release(qk); A.release(vt); A.release(ao); A.release(x_tok)
            emit(lambda t=t, x2=x2: ops.layernorm(x2, w[t + "ln_2.g"], w[t + "ln_2.b"], XF_LN_EPS, out=ln), "layernorm")
            hfc = dense(ln, B, ctx, xw, 4 * xw, w[t + "fc.w"], bias=w[t + "fc.b"], epilogue=ops.EPI_GELU)
            x_tok = dense(hfc, B, ctx, 4 * xw, xw, w[t + "fc2.w"], bias=w[t + "fc2.b"], residual=x2)
            A.release(hfc); A.release(x2)
        xf_out = A.get((B, ctx, xw))     # kept for every AttentionBlock's encoder_kv
        emit(lambda x_tok=x_tok: ops.layernorm(x_tok, w["final_ln.g"], w["final_ln.b"], XF_LN_EPS, out=xf_out), "layernorm")
        A.release(ln)
        emit(lambda: cat_in[:, ted:].copy_(xf_out[:, -1]), "small")     # dtype-converting copy (plumbing); text-only too
        text_mode[0] = False
        P.n_text = len(main)      # ops [0, n_text) write xf_out and cat_in[:, ted:] -- both dedicated / never released

        # ---- time embedding + xf_proj (text2im_model.py:102-105), then all emb_layers (unet.py:163-170) at once
        t_emb = torch.empty((B, mc), dtype=f32, device=dev)
        emb = torch.empty((B, ted), dtype=f32, device=dev)
        P.emb_all = torch.empty((B, self._emb_total), dtype=f32, device=dev)
        emit(lambda: ops.timestep_embedding(P.t_static, mc, out=t_emb), "small")
        emit(lambda: ops.dense_small(t_emb, w["te0.w"], w["te0.b"], act_out=True, out=cat_in[:, :ted]), "small")
        # (round 5) the SiLU in front of every emb_layer (unet.py:163-170) is applied ONCE, on the way out of the GEMV that produces
        # emb -- `emb` feeds nothing else here -- instead of on the way into the emb_layers GEMV, where every one of its ~25 000
        # output columns re-evaluated it on all B x 768 inputs (the 159 us "small" op of profiles/r05_glide_op_profile.txt, every
        # step).  Same value either way: silu of the same fp32 number.
        emit(lambda: ops.dense_small(cat_in, w["te2proj.w"], w["te2proj.b"], act_out=True, out=emb), "small")
        emit(lambda: ops.dense_small(emb, w["emb.w"], w["emb.b"], out=P.emb_all), "small")
        n_emb_mark = len(main)    # ops [n_text, n_emb) turn (timestep, last text token) into P.emb_all: a loop runs them once (begin_loop)

        def resblock(pre, x, x2, cin, cout, mode, h, wd):
            """unet.py:178-218 (scale-shift norm; up/down act on BOTH h and x)."""
            hw = h * wd
            a = A.get((B, hw, cin))
            gn(x, x2, w[pre + "n1.g"], w[pre + "n1.b"], True, a)
            if mode == "up":
                assert x2 is None and cin == cout
                hbuf, ho, wo = conv3(a, cin, cout, w[pre + "conv1.w"], w[pre + "conv1.b"], h, wd, upsample=1,
                                     wsub=w.get(pre + "conv1.wsub"))
                xs = A.get((B, ho * wo, cin))
                emit(lambda x=x, xs=xs: ops.upsample_nearest2x(x, B, h, wd, cin, out=xs), "small")
            elif mode == "down":
                assert x2 is None and cin == cout
                ap = A.get((B, hw // 4, cin))
                emit(lambda a=a, ap=ap: ops.avgpool2x2(a, B, h, wd, cin, out=ap), "small")
                hbuf, ho, wo = conv3(ap, cin, cout, w[pre + "conv1.w"], w[pre + "conv1.b"], h // 2, wd // 2)
                A.release(ap)
                xs = A.get((B, hw // 4, cin))
                emit(lambda x=x, xs=xs: ops.avgpool2x2(x, B, h, wd, cin, out=xs), "small")
            else:
                hbuf, ho, wo = conv3(a, cin, cout, w[pre + "conv1.w"], w[pre + "conv1.b"], h, wd)
                xs = x
            A.release(a)
            eoff = self._emb_off[pre]
            a2 = A.get((B, ho * wo, cout))
            gn(hbuf, None, w[pre + "n2.g"], w[pre + "n2.b"], True, a2,
               scale=P.emb_all[:, eoff:eoff + cout], shift=P.emb_all[:, eoff + cout:eoff + 2 * cout])
            A.release(hbuf)
            if cin != cout:
                c2 = 0 if x2 is None else x2.shape[2]
                if mode not in ("up", "down") and skip_fusable(a2, cin - c2, c2, cout, ho, wo, w[pre + "conv2.w"]):
                    if (pre + "conv2skip.b") not in w:
                        w[pre + "conv2skip.b"] = (w[pre + "conv2.b"] + w[pre + "skip.b"]).contiguous()
                    out, _, _ = conv3(a2, cout, cout, w[pre + "conv2.w"], w[pre + "conv2skip.b"], ho, wo,
                                      skip=(x, x2, cin - c2, c2, w[pre + "skip.w"]))
                    A.release(a2)
                    return out, ho, wo
                skip = dense(x, B, hw, cin, cout, w[pre + "skip.w"], bias=w[pre + "skip.b"], src2=x2, c2=c2)
            else:
                assert x2 is None
                skip = xs
            out, _, _ = conv3(a2, cout, cout, w[pre + "conv2.w"], w[pre + "conv2.b"], ho, wo, residual=skip)
            A.release(a2)
            if skip is not x:
                A.release(skip)
            return out, ho, wo

        kv_keep = []

        def attnblock(pre, x, c, heads, h, wd):
            """unet.py:254-310: q from the image, keys/values = [text (ctx) | image (T)]."""
            T = h * wd
            nk = ctx + T
            a = A.get((B, T, c))
            gn(x, None, w[pre + "norm.g"], w[pre + "norm.b"], False, a)
            # keys / values = [text | image]: the text rows (encoder_kv of xf_out, unet.py:289-297) depend on the tokens alone, so
            # they live in buffers of their own (not the arena: they must survive the step) and their two projections join the
            # plan's text prefix; only the image rows are written per step
            # Memory: these buffers are per (B, H, W) PLAN and are never arena-reused -- B * (text + image) * 3 c halves per
            # AttentionBlock (base model at 16 rows: 22 blocks, 0.14 GB; up-sampler at 8 rows: 0.02 GB), plus a second captured
            # graph (graph_main) per plan.  Whether the text rows are current is the CALLER's statement (text_epoch): nothing on the
            # device re-checks tok_static, and forward_nhwc() without an epoch recomputes the prefix
            vtb = torch.zeros((B, c, nk), dtype=f16, device=dev)
            if ops.get_option("glide_qkv_merge") and T % 8 == 0:
                # (round 6) q | k | v of the image tokens in ONE launch: q and k side by side in a [B, text + image, 2 c] buffer --
                # rows [ctx, nk) written here (q | k), the k half of rows [0, ctx) by the text projection (its q half is never read)
                # -- and V transposed into vtb; the attention kernel takes q and k as strided views of that buffer
                qkb = torch.zeros((B, nk, 2 * c), dtype=f16, device=dev)
                kbuf = qkb[:, :, c:]
                kv_keep.append((kbuf, vtb, qkb))
                gemm(a=a, w=w[pre + "qkv.w"], N=3 * c, B=B, H=T, W=1, c1=c, out=qkb[:, ctx:], out_ld=2 * c, bias=w[pre + "qkv.b"],
                     out_bs=nk * 2 * c, out2=vtb[:, :, ctx:], out2_ld=nk, n_split=2 * c)
                # AttentionBlock.norm has no activation (unet.py:267-272): the merged projection can apply it to its A fragments from
                # the producer's column statistics (mdx_gemm_desc.gn_colstats on a dense launch, as the LDM planner's
                # unet_gn_proj_fuse).  Decided when the statistics are wired; on success the GroupNorm op is dropped
                if (ops.get_option("glide_gn_qkv_fuse") and T % 64 == 0 and c % 64 == 0 and c <= 2560
                        and producer.get(x.data_ptr()) is not None):
                    gn_calls[-1]["proj"] = dict(desc=descs[-1], meta=len(meta) - 1)
                q_ptr, q_bs, q_ld = qkb[:, ctx:].data_ptr(), nk * 2 * c, 2 * c
                k_ptr, k_bs, k_ld = kbuf.data_ptr(), nk * 2 * c, 2 * c
                q = None
            else:
                q = dense(a, B, T, c, c, w[pre + "q.w"], bias=w[pre + "q.b"])
                kbuf = torch.zeros((B, nk, c), dtype=f16, device=dev)
                kv_keep.append((kbuf, vtb))
                dense(a, B, T, c, c, w[pre + "k.w"], bias=w[pre + "k.b"], out=kbuf[:, ctx:], out_ld=c, out_bs=nk * c)
                dense(a, B, T, c, c, w[pre + "v.w"], bias=w[pre + "v.b"], out=vtb[:, :, ctx:], out_ld=nk,
                      out_mode=ops.OUT_TRANSPOSED)
                q_ptr, q_bs, q_ld = q.data_ptr(), T * c, c
                k_ptr, k_bs, k_ld = kbuf.data_ptr(), nk * c, c
            late_text[0] = True
            dense(xf_out, B, ctx, xw, c, w[pre + "ek.w"], bias=w[pre + "ek.b"], out=kbuf, out_ld=k_ld, out_bs=k_bs)
            dense(xf_out, B, ctx, xw, c, w[pre + "ev.w"], bias=w[pre + "ev.b"], out=vtb, out_ld=nk,
                  out_mode=ops.OUT_TRANSPOSED)
            late_text[0] = False
            o = a   # the normed input is dead after the projections
            emit(lambda vtb=vtb, o=o, q_ptr=q_ptr, q_bs=q_bs, q_ld=q_ld, k_ptr=k_ptr, k_bs=k_bs, k_ld=k_ld: ops.attention(
                q_ptr, k_ptr, vtb.data_ptr(), o.data_ptr(), B, heads, 64, T, nk, 64 ** -0.5,
                q_bs, q_ld, k_bs, k_ld, c * nk, nk, T * c, c), "attention", 4 * B * heads * T * nk * 64)
            out = dense(o, B, T, c, c, w[pre + "proj.w"], bias=w[pre + "proj.b"], residual=x)
            if q is not None:
                A.release(q)
            A.release(a)
            return out

        # ---- UNet walk (text2im_model.py:106-123)
        xin = A.get((B, H * W, self.cin_pad))
        if self.super_res:
            s_low = self.low_size
            P.low_static = torch.zeros((B, 3, s_low, s_low), dtype=f32, device=dev)
            emit(lambda: ops.glide_superres_input(P.x_static, P.low_static, out=xin), "small")
        else:
            emit(lambda: ops.nchw_to_nhwc(P.x_static, self.cin_pad, out=xin), "small")
        h, wd = H, W
        hs, cur = [], None
        for i, blk in enumerate(self.input_blocks):
            for j, layer in enumerate(blk):
                pre = f"input_blocks.{i}.{j}."
                if layer[0] == "conv":
                    cur, h, wd = conv3(xin, self.cin_pad, layer[2], w[pre + "w"], w[pre + "b"], h, wd)
                    A.release(xin)
                elif layer[0] == "res":
                    new, h2, w2 = resblock(pre, cur, None, layer[1], layer[2], layer[3], h, wd)
                    if not any(cur is s_[0] for s_ in hs):
                        A.release(cur)
                    cur, h, wd = new, h2, w2
                else:
                    new = attnblock(pre, cur, layer[1], layer[2], h, wd)
                    A.release(cur)
                    cur = new
            hs.append((cur, h, wd))
        for j, layer in enumerate(self.middle_block):
            pre = f"middle_block.{j}."
            if layer[0] == "res":
                new, _, _ = resblock(pre, cur, None, layer[1], layer[2], layer[3], h, wd)
            else:
                new = attnblock(pre, cur, layer[1], layer[2], h, wd)
            if not any(cur is s_[0] for s_ in hs):
                A.release(cur)
            cur = new
        for i, blk in enumerate(self.output_blocks):
            skip, sh, sw = hs.pop()
            assert (sh, sw) == (h, wd)
            for j, layer in enumerate(blk):
                pre = f"output_blocks.{i}.{j}."
                if layer[0] == "res" and j == 0:
                    new, _, _ = resblock(pre, cur, skip, layer[1], layer[2], layer[3], h, wd)
                    A.release(cur); A.release(skip)
                    cur = new
                elif layer[0] == "res":
                    new, h, wd = resblock(pre, cur, None, layer[1], layer[2], layer[3], h, wd)
                    A.release(cur)
                    cur = new
                else:
                    new = attnblock(pre, cur, layer[1], layer[2], h, wd)
                    A.release(cur)
                    cur = new
        ch0 = int(self.channel_mult[0] * mc)
        a = A.get((B, h * wd, ch0))
        gn(cur, None, w["out.g"], w["out.b"], True, a)
        P.out_nhwc = torch.empty((B, h * wd, self.cout_pad), dtype=f16, device=dev)
        gemm(a=a, w=w["out2.w"], N=self.cout_pad, B=B, H=h, W=wd, c1=ch0, out=P.out_nhwc, out_ld=self.cout_pad,
             bias=w["out2.b"], ksize=3)

        need = max([ops.gemm_workspace_bytes(d) for d in descs] + [16])
        P.gemm_ws = ops.new_gemm_workspace(need, dev)
        for d in descs:
            d.workspace = P.gemm_ws.data_ptr()
            d.workspace_bytes = P.gemm_ws.numel() * 4
        if late_main:       # the AttentionBlocks' encoder_kv projections join the text prefix (they read xf_out only)
            nt = P.n_text
            main[nt:nt] = late_main
            meta[nt:nt] = late_meta
            for c_ in gn_calls:
                if c_["meta"] >= nt:
                    c_["meta"] += len(late_main)
            P.n_text = nt + len(late_main)
            n_emb_mark += len(late_main)
        P.n_emb = n_emb_mark
        P.kv_keep = kv_keep
        P.gn_ws = torch.empty(gn_need[0], dtype=f32, device=dev)
        P.colstats = {}
        import os
        ops.wire_groupnorm_colstats(gn_calls if os.environ.get("MDX_UNET_GN_COLSTATS", "1") != "0" else [], meta, B, dev,
                                    P.colstats)
        # a wired producer may resolve to another tile-table row than the one the workspace was sized for: size it again
        need2 = max([ops.gemm_workspace_bytes(d) for d in descs] + [16])
        if need2 > P.gemm_ws.numel() * 4:
            P.gemm_ws = ops.new_gemm_workspace(need2, dev)
            for d in descs:
                d.workspace = P.gemm_ws.data_ptr()
                d.workspace_bytes = P.gemm_ws.numel() * 4
        if any(m.get("dead") for m in meta):      # GroupNorm launches that moved into the GEMM behind them (all behind the emb chain)
            keep = [i for i, m in enumerate(meta) if not m.get("dead")]
            assert all(i >= P.n_emb for i, m in enumerate(meta) if m.get("dead"))
            main[:] = [main[i] for i in keep]
            meta[:] = [meta[i] for i in keep]
        ops.check_colstats_wiring(descs)
        ops.account_gemm_launches(meta)
        P.main, P.meta, P.descs, P.arena = main, meta, descs, A
        P.keep = (t_emb, cat_in, emb, xf_out)
        P.graph, P.graph_main, P.graph_body, P.graph_failed = None, None, None, False
        P.text_epoch = None       # what the text prefix was last run for (forward_nhwc)
        P.loop_epoch = None       # the begin_loop() context whose text rows the key / value buffers hold
        self._plans[key] = P
        return P

    # ------------------------------------------------------------------ execution
    def forward_nhwc(self, x, timesteps, tokens, mask, low_res=None, text_epoch=None):
        """Returns the plan's static NHWC fp16 output [N, H*W, 8] (6 channels valid; overwritten by the next call).
        text_epoch: an opaque value the caller changes whenever (tokens, mask) change (None = they may have changed).  The text
        transformer -- 16 layers, ~130 launches, a tenth of a step -- depends on the tokens alone; when the epoch equals the one
        of this plan's previous call the plan's text prefix is skipped and its outputs (xf_out, the last-token embedding) are
        reused.  The reference recomputes it every step (text2im_model.py:88-99); the values are the same."""
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise MdxError("x must be a CUDA(HIP) tensor (no CPU fallback)")
        B, _, H, W = x.shape
        P = self._plan(B, H, W)
        cached = text_epoch is not None and P.text_epoch == text_epoch
        P.x_static.copy_(x)
        P.t_static.copy_(torch.as_tensor(timesteps).to(device=self.device, dtype=f32).expand(B))
        if not cached:
            P.tok_static.copy_(torch.as_tensor(tokens).to(self.device))
            P.mask_static.copy_(torch.as_tensor(mask).to(self.device))
        if self.super_res:
            if low_res is None:
                raise MdxError("SuperResText2ImUNet needs low_res")
            P.low_static.copy_(low_res)
        P.text_epoch = None       # (an exception below must not leave a half-written prefix marked valid)
        P.loop_epoch = None       # (this call rewrites the text rows a begin_loop() context may have placed)
        body = P.main[P.n_text:] if cached else P.main
        if self.use_graph and not P.graph_failed:
            self._ensure_graphs(P)
            if P.graph is not None:
                (P.graph_main if cached else P.graph).replay()
                P.text_epoch = text_epoch
                return P.out_nhwc
        for op in body:
            op()
        P.text_epoch = text_epoch
        return P.out_nhwc

    def _ensure_graphs(self, P):
        """First use of a plan: one eager pass (whatever the static inputs hold), then three captures of the same op list -- the
        whole step, the step without its text prefix (text_epoch), the step without text prefix AND time embedding (begin_loop)."""
        if P.graph is not None or P.graph_failed or not self.use_graph:
            return
        try:
            for op in P.main:
                op()
            torch.cuda.synchronize()
            P.graph = ops.capture_graph(P.main)
            P.graph_main = ops.capture_graph(P.main[P.n_text:])
            P.graph_body = ops.capture_graph(P.main[P.n_emb:])
        except Exception as e:  # pragma: no cover
            P.graph, P.graph_main, P.graph_body, P.graph_failed = None, None, None, True
            import warnings
            warnings.warn(f"hipGraph capture failed, running eagerly: {e}")

    # ------------------------------------------------------------------ whole-loop tables (round 6)
    # The sampling loops know every prompt and every timestep before their first step: main_funcs.py:21-44 draws the unconditional
    # prompt of step k inside the loop, but from numpy's stream, independent of the model; the conditional prompts are constants.
    # So everything of a step that depends on (prompt, timestep) alone is computed ONCE per loop --
    #   * the text transformer (text2im_model.py:88-99, xf.py:126-154) on the conditional prompts and, in one pass of S x ctx rows,
    #     on all S unconditional prompts;
    #   * every AttentionBlock's encoder_kv projection of those rows (unet.py:289-297), as tables [rows, ctx, C] / [rows, C, ctx];
    #   * emb = time_embed(t) + transformer_proj(xf_out[:, -1]) and all ResBlocks' emb_layers (text2im_model.py:102-105,
    #     unet.py:163-170) for all S x N (step, row) pairs: the table the LDM path calls time_embedding_table --
    # and a step is: copy emb row-set k, ONE launch that drops prompt k's key / value rows into the blocks' text slots
    # (mdx_glide_kv_select_f16), replay of the plan's body.  Same kernels on the same values: results are bit-identical to the
    # per-step recomputation (tests/test_glide_gpu.py) as long as the text GEMMs do not split K differently, which the table pass
    # rules out (splitk = 1; the per-step launches of the benchmarked shapes resolve to unsplit table rows).
    class _TextPlan:
        pass

    def _attn_layers(self):
        return [(pre, layer[1]) for pre, layer in self._named_layers() if layer[0] == "attn"]

    def _text_plan(self, R):
        """Text transformer + every AttentionBlock's encoder_kv projection on R prompt rows: static token / mask inputs, an op
        list, and the outputs `last` [R, xw] fp32 (xf_out[:, -1]), k[j] [R, ctx, C_j], vt[j] [R, C_j, ctx] (walk order)."""
        if not hasattr(self, "_text_plans"):
            self._text_plans = {}
        if R in self._text_plans:
            return self._text_plans[R]
        if self.w is None:
            raise MdxError("load_state_dict() must be called before the first forward")
        dev, w = self.device, self.w
        ctx, xw, xh = self.text_ctx, self.xf_width, self.xf_heads
        T = Text2ImUNet._TextPlan()
        A = _Arena(dev)
        main, descs = [], []
        T.tok_static = torch.zeros((R, ctx), dtype=torch.int32, device=dev)
        T.mask_static = torch.ones((R, ctx), dtype=torch.int32, device=dev)

        def dense(src, tokens, cin, nout, wt, bias=None, residual=None, epilogue=ops.EPI_NONE, out=None, out_ld=None,
                  out_mode=ops.OUT_ROWMAJOR):
            if out is None:
                out = A.get((R, tokens, nout))
                out_ld = nout
            d = ops.make_gemm_desc(a=src, w=wt, N=nout, B=R, H=tokens, W=1, c1=cin, out=out, out_ld=out_ld, bias=bias,
                                   residual=residual, residual_ld=nout if residual is not None else 0, epilogue=epilogue,
                                   out_mode=out_mode, splitk=1)
            descs.append(d)
            main.append(lambda d=d: ops.gemm_run(d))
            return out

        x_tok = A.get((R, ctx, xw))
        main.append(lambda: ops.glide_text_embed(T.tok_static, T.mask_static, w["tok"], w["pos"], w["pad"], out=x_tok))
        ln = A.get((R, ctx, xw))
        for l in range(self.xf_layers):
            t = f"transformer.resblocks.{l}."
            main.append(lambda t=t, x_tok=x_tok: ops.layernorm(x_tok, w[t + "ln_1.g"], w[t + "ln_1.b"], XF_LN_EPS, out=ln))
            qk = dense(ln, ctx, xw, 2 * xw, w[t + "qk.w"], bias=w[t + "qk.b"])
            vt = A.get((R, xw, ctx))
            dense(ln, ctx, xw, xw, w[t + "v.w"], bias=w[t + "v.b"], out=vt, out_ld=ctx, out_mode=ops.OUT_TRANSPOSED)
            ao = A.get((R, ctx, xw))
            main.append(lambda qk=qk, vt=vt, ao=ao: ops.attention(
                qk.data_ptr(), qk.data_ptr() + xw * 2, vt.data_ptr(), ao.data_ptr(), R, xh, 64, ctx, ctx, 64 ** -0.5,
                ctx * 2 * xw, 2 * xw, ctx * 2 * xw, 2 * xw, xw * ctx, ctx, ctx * xw, xw))
            x2 = dense(ao, ctx, xw, xw, w[t + "proj.w"], bias=w[t + "proj.b"], residual=x_tok)
            A.release(qk); A.release(vt); A.release(ao); A.release(x_tok)
            main.append(lambda t=t, x2=x2: ops.layernorm(x2, w[t + "ln_2.g"], w[t + "ln_2.b"], XF_LN_EPS, out=ln))
            hfc = dense(ln, ctx, xw, 4 * xw, w[t + "fc.w"], bias=w[t + "fc.b"], epilogue=ops.EPI_GELU)
            x_tok = dense(hfc, ctx, 4 * xw, xw, w[t + "fc2.w"], bias=w[t + "fc2.b"], residual=x2)
            A.release(hfc); A.release(x2)
        xf_out = torch.empty((R, ctx, xw), dtype=f16, device=dev)
        main.append(lambda x_tok=x_tok: ops.layernorm(x_tok, w["final_ln.g"], w["final_ln.b"], XF_LN_EPS, out=xf_out))
        T.last = torch.empty((R, xw), dtype=f32, device=dev)
        main.append(lambda: T.last.copy_(xf_out[:, -1]))
        T.k, T.vt = [], []
        for pre, c in self._attn_layers():
            kb = torch.empty((R, ctx, c), dtype=f16, device=dev)
            vb = torch.empty((R, c, ctx), dtype=f16, device=dev)
            dense(xf_out, ctx, xw, c, w[pre + "ek.w"], bias=w[pre + "ek.b"], out=kb, out_ld=c)
            dense(xf_out, ctx, xw, c, w[pre + "ev.w"], bias=w[pre + "ev.b"], out=vb, out_ld=ctx, out_mode=ops.OUT_TRANSPOSED)
            T.k.append(kb)
            T.vt.append(vb)
        need = max([ops.gemm_workspace_bytes(d) for d in descs] + [16])
        T.gemm_ws = ops.new_gemm_workspace(need, dev)
        for d in descs:
            d.workspace = T.gemm_ws.data_ptr()
            d.workspace_bytes = T.gemm_ws.numel() * 4
        T.main, T.descs, T.arena, T.xf_out, T.R = main, descs, A, xf_out, R
        self._text_plans[R] = T
        return T

    class _Loop:
        pass

    def begin_loop(self, N, H, W, t_values, tokens, mask, step_tokens=None, step_mask=None):
        """Prepare a sampling loop of S = len(t_values) steps on N rows (see the block comment above).
        tokens / mask [Pc, ctx]: the prompts of rows [0, Pc), the same at every step.  step_tokens / step_mask [S, ctx] (optional):
        step k's prompt for ALL rows [Pc, N) (main_funcs.py:37-41: one random unconditional prompt per step, repeated over the
        batch, guider.py:46-47); without them Pc must equal N.  t_values[k]: the timestep the UNet sees at step k.
        Returns the context loop_step() takes; a later begin_loop() or forward_nhwc() on this plan invalidates it."""
        P = self._plan(N, H, W)
        dev = self.device
        S = len(t_values)
        tokens = torch.as_tensor(tokens).to(dev, torch.int32)
        Pc = int(tokens.shape[0])
        if step_tokens is None and Pc != N:
            raise MdxError(f"begin_loop: {Pc} prompts for {N} rows and no per-step prompts for the rest")
        if step_tokens is not None and not 0 < Pc < N:
            raise MdxError(f"begin_loop: per-step prompts need 0 < constant prompts ({Pc}) < rows ({N})")
        self._ensure_graphs(P)
        if not hasattr(self, "_loop_epochs"):
            self._loop_epochs = 0
        self._loop_epochs += 1
        L = Text2ImUNet._Loop()
        L.epoch, L.N, L.Pc, L.S, L.plan = self._loop_epochs, N, Pc, S, P
        ctx, xw, ted = self.text_ctx, self.xf_width, self.time_embed_dim
        # ---- text tables
        Tc = self._text_plan(Pc)
        Tc.tok_static.copy_(tokens)
        Tc.mask_static.copy_(torch.as_tensor(mask).to(dev, torch.int32))
        for op in Tc.main:
            op()
        Tu = None
        if step_tokens is not None:
            Tu = self._text_plan(S)
            Tu.tok_static.copy_(torch.as_tensor(step_tokens).to(dev, torch.int32))
            if step_mask is None:
                Tu.mask_static.fill_(1)
            else:
                Tu.mask_static.copy_(torch.as_tensor(step_mask).to(dev, torch.int32))
            for op in Tu.main:
                op()
        # ---- emb table [S, N, emb_total]: the plan's three small ops on all S x N (timestep, last text token) rows
        w = self.w
        cat = torch.empty((S, N, ted + xw), dtype=f32, device=dev)
        cat[:, :Pc, ted:] = Tc.last[None]
        if Tu is not None:
            cat[:, Pc:, ted:] = Tu.last[:, None]
        cat2 = cat.view(S * N, ted + xw)
        t_rows = torch.as_tensor([float(v) for v in t_values], dtype=f32, device=dev)[:, None].expand(S, N).contiguous().view(-1)
        t_emb = ops.timestep_embedding(t_rows, self.model_channels)
        ops.dense_small(t_emb, w["te0.w"], w["te0.b"], act_out=True, out=cat2[:, :ted])
        emb = ops.dense_small(cat2, w["te2proj.w"], w["te2proj.b"], act_out=True)
        L.emb_tab = ops.dense_small(emb, w["emb.w"], w["emb.b"]).view(S, N, self._emb_total)
        # ---- key / value slots: the constant rows now, the per-step rows in loop_step
        def slots(Tt):
            ent = []
            for j, kv in enumerate(P.kv_keep):
                kbuf, vtb = kv[0], kv[1]      # (kbuf may be the k half of a merged [B, nk, 2 c] q | k buffer: strides, not shapes)
                c, nk = kbuf.shape[2], kbuf.shape[1]
                ent.append((Tt.k[j], kbuf, ctx * c * 2, kbuf.stride(0) * 2, c * 2, kbuf.stride(1) * 2, ctx, c * 2))
                ent.append((Tt.vt[j], vtb, c * ctx * 2, c * nk * 2, ctx * 2, nk * 2, c, ctx * 2))
            return ops.glide_kv_slots(ent, dev)
        L.tabs = (Tc, Tu)
        if P.kv_keep:
            sc, n = slots(Tc)
            ops.glide_kv_select(sc, n, 0, 1, 0, Pc)
            L.slots_c = sc
            L.slots_u, L.nslots = slots(Tu) if Tu is not None else (None, n)
        else:
            L.slots_u, L.nslots = None, 0
        P.text_epoch = None       # (the plan's own text prefix no longer describes what the key / value buffers hold)
        P.loop_epoch = L.epoch
        return L

    def loop_step(self, L, k, x, low_res=None):
        """Step k of a begin_loop() context: returns the plan's static NHWC fp16 output, like forward_nhwc."""
        P = L.plan
        if P.loop_epoch != L.epoch:
            raise MdxError("loop_step: this loop context is stale (a later begin_loop / forward_nhwc used the plan)")
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise MdxError("x must be a CUDA(HIP) tensor (no CPU fallback)")
        P.x_static.copy_(x)
        if self.super_res:
            if low_res is None:
                raise MdxError("SuperResText2ImUNet needs low_res")
            P.low_static.copy_(low_res)
        P.emb_all.copy_(L.emb_tab[k])
        if L.slots_u is not None:
            ops.glide_kv_select(L.slots_u, L.nslots, k, 0, L.Pc, L.N - L.Pc)
        if P.graph_body is not None:
            P.graph_body.replay()
        else:
            for op in P.main[P.n_emb:]:
                op()
        return P.out_nhwc

    def construct(self, x, timesteps, tokens=None, mask=None):
        """text2im_model.py:101-123 -> [N, 6, H, W] fp32."""
        out = self.forward_nhwc(x, timesteps, tokens, mask)
        return ops.nhwc_to_nchw(out, self.out_channels, x.shape[2], x.shape[3])

    __call__ = construct


class SuperResText2ImUNet(Text2ImUNet):
    """text2im_model.py:126-238: the same UNet on [x | bilinear(low_res)] (6 input channels)."""
    super_res = True

    def __init__(self, image_size, text_ctx, xf_width, xf_layers, xf_heads, xf_final_ln, n_vocab, in_channels=6,
                 low_size=64, **kw):
        super().__init__(text_ctx, xf_width, xf_layers, xf_heads, xf_final_ln, n_vocab, in_channels=in_channels,
                         image_size=image_size, **kw)
        self.low_size = low_size

    def construct(self, x, timesteps, low_res=None, tokens=None, mask=None):
        out = self.forward_nhwc(x, timesteps, tokens, mask, low_res=low_res)
        return ops.nhwc_to_nchw(out, self.out_channels, x.shape[2], x.shape[3])

    __call__ = construct
