"""VAE Decoder on the MI355X kernels -- the inference half of the reference's
ldm/modules/diffusionmodules/model.py (Decoder :321-440, ResnetBlock :80-148, AttnBlock :151-206, Upsample :31-52,
Normalize :26-28, nonlinearity :21-23).  SURVEY.md 8(f) item 1: the step right after the denoising loop.

Same execution model as UNetModel: the constructor keywords are the reference's ``ddconfig``; ``load_state_dict``
packs the reference-named parameters once; ``_plan(B, h, w)`` walks the structure once and emits a flat list of
C-ABI calls on arena buffers, captured as one hipGraph.  Every conv runs through mdx_gemm_f16 (3x3 stride 1 -> the
HALO kernel, the nearest-2x of Upsample folded into the next conv's gather), GroupNorm(eps 1e-6)+swish through
mdx_groupnorm_f16.  AttnBlock has ONE head of d = C = 512: like the reference it materialises the [hw, hw] scores,
as two plain GEMMs (K and V^T re-laid as packed B operands, mdx_pack_b_operand_f16) around mdx_softmax_rows_f16.
"""
import numpy as np
import torch

from ...._lib import MdxError
from .... import ops
from .openaimodel import _Arena

f16, f32 = torch.float16, torch.float32


def _run_plan(net, P):
    """Replay the plan as one hipGraph (captured on first use), or eagerly if capture is unavailable."""
    if net.use_graph and not P.graph_failed:
        if P.graph is None:
            try:
                for op in P.main:
                    op()
                torch.cuda.synchronize()
                P.graph = ops.capture_graph(P.main)
            except Exception as e:  # pragma: no cover - depends on the runtime
                P.graph, P.graph_failed = None, True
                import warnings
                warnings.warn(f"hipGraph capture failed, running eagerly: {e}")
        if P.graph is not None:
            P.graph.replay()
            return
    for op in P.main:
        op()


class _PlanBuilder:
    """Emits the C-ABI call list of a VAE net on arena buffers (shared by Decoder and Encoder)."""

    def __init__(self, net, P, B):
        self.net, self.P, self.B = net, P, B
        self.dev, self.w = net.device, net.w
        self.A = _Arena(self.dev)
        self.main, self.meta, self.descs = [], [], []
        self.gn_need = 4

    def emit(self, fn, kind, flops=0, info=""):
        self.main.append(fn)
        self.meta.append({"kind": kind, "flops": int(flops), "launches": 1, "info": info})

    def gemm(self, **kw):
        d = ops.make_gemm_desc(**kw)
        self.descs.append(d)
        ks, up, st = kw.get("ksize", 1), kw.get("upsample", 0), kw.get("stride", 1)
        m_rows = kw["B"] * kw["H"] * kw["W"] * (4 if up else 1) // (st * st)
        kdim = ks * ks * kw["c1"]
        self.emit(lambda d=d: ops.gemm_run(d), "gemm", 2 * m_rows * kw["N"] * kdim,
                  f"M={m_rows} N={kw['N']} K={kdim} k{ks}s{st}u{up}")

    def gn(self, x, g, b, silu, out):
        Bq, HW, C = x.shape
        self.gn_need = max(self.gn_need, ops.groupnorm_ws_floats(Bq, HW, C))
        P = self.P
        self.emit(lambda: ops.groupnorm(x, None, g, b, 1e-6, silu, ws=P.gn_ws, out=out), "groupnorm", 0, f"B={Bq} HW={HW} C={C}")

    def conv3(self, src, cin, cout, wt, bias, h, wd, upsample=0, residual=None, n_store=None, stride=1, asym_pad=0):
        ho, wo = (2 * h, 2 * wd) if upsample else (h // stride, wd // stride)
        n_store = n_store or cout
        out = self.A.get((self.B, ho * wo, n_store))
        self.gemm(a=src, w=wt, N=n_store, B=self.B, H=h, W=wd, c1=cin, out=out, out_ld=n_store, bias=bias, residual=residual,
                  residual_ld=n_store if residual is not None else 0, ksize=3, upsample=upsample, stride=stride,
                  asym_pad=asym_pad)
        return out

    def conv1(self, src, tokens, cin, cout, wt, bias, residual=None, out=None, out_ld=None, out_mode=ops.OUT_ROWMAJOR):
        if out is None:
            out, out_ld = self.A.get((self.B, tokens, cout)), cout
        self.gemm(a=src, w=wt, N=cout, B=self.B, H=tokens, W=1, c1=cin, out=out, out_ld=out_ld, bias=bias, residual=residual,
                  residual_ld=cout if residual is not None else 0, out_mode=out_mode)
        return out

    def resblock(self, pre, x, cin, cout, h, wd):          # ResnetBlock.construct model.py:128-148, temb = None
        A, w, B = self.A, self.w, self.B
        hw = h * wd
        a = A.get((B, hw, cin))
        self.gn(x, w[pre + "norm1.g"], w[pre + "norm1.b"], True, a)
        h1 = self.conv3(a, cin, cout, w[pre + "conv1.w"], w[pre + "conv1.b"], h, wd)
        A.release(a)
        a2 = A.get((B, hw, cout))
        self.gn(h1, w[pre + "norm2.g"], w[pre + "norm2.b"], True, a2)
        A.release(h1)
        skip = x if cin == cout else self.conv1(x, hw, cin, cout, w[pre + "nin.w"], w[pre + "nin.b"])
        out = self.conv3(a2, cout, cout, w[pre + "conv2.w"], w[pre + "conv2.b"], h, wd, residual=skip)
        A.release(a2)
        if skip is not x:
            A.release(skip)
        return out

    def attnblock(self, pre, x, c, h, wd):                 # AttnBlock.construct model.py:182-206
        A, w, B = self.A, self.w, self.B
        hw = h * wd
        if hw % 8:
            raise MdxError("VAE attention needs h*w % 8 == 0")
        hn = A.get((B, hw, c))
        self.gn(x, w[pre + "norm.g"], w[pre + "norm.b"], False, hn)
        q = self.conv1(hn, hw, c, c, w[pre + "q.w"], w[pre + "q.b"])
        k = self.conv1(hn, hw, c, c, w[pre + "k.w"], w[pre + "k.b"])
        vt = A.get((B, c, hw))                         # V^T [b][c][hw]: the GEMM stores it transposed
        self.conv1(hn, hw, c, c, w[pre + "v.w"], w[pre + "v.b"], out=vt, out_ld=hw, out_mode=ops.OUT_TRANSPOSED)
        A.release(hn)
        o = A.get((B, hw, c))
        kp = A.get((((hw + 63) // 64) * ((c + 63) // 64) * 4096,))     # packed K   (rows = keys, K = c)
        vp = A.get((((c + 63) // 64) * ((hw + 63) // 64) * 4096,))     # packed V^T (rows = c,    K = keys)
        s = A.get((hw, hw))                            # scores of ONE image (the reference holds all B at once)
        scale = float(int(c) ** (-0.5))
        for b in range(B):
            self.emit(lambda b=b: ops.pack_b_operand(k[b], out=kp), "small")
            self.gemm(a=q[b], w=kp, N=hw, B=1, H=hw, W=1, c1=c, out=s, out_ld=hw)                  # w_ = bmm(q, k)
            self.emit(lambda: ops.softmax_rows(s, scale), "small")                                   # * c^-0.5, Softmax
            self.emit(lambda b=b: ops.pack_b_operand(vt[b], out=vp), "small")
            self.gemm(a=s, w=vp, N=c, B=1, H=hw, W=1, c1=hw, out=o[b], out_ld=c)                    # h_ = bmm(v, w_^T)
        for t in (q, k, vt, kp, vp, s):
            A.release(t)
        out = self.conv1(o, hw, c, c, w[pre + "proj_out.w"], w[pre + "proj_out.b"], residual=x)
        A.release(o)
        return out

    def finish(self):
        P = self.P
        need = max([ops.gemm_workspace_bytes(d) for d in self.descs] + [16])
        P.gemm_ws = ops.new_gemm_workspace(need, self.dev)
        for d in self.descs:
            d.workspace, d.workspace_bytes = P.gemm_ws.data_ptr(), P.gemm_ws.numel() * 4
        P.gn_ws = torch.empty(self.gn_need, dtype=f32, device=self.dev)
        P.main, P.meta, P.descs, P.arena = self.main, self.meta, self.descs, self.A
        P.activation_bytes = self.A.total


class Decoder:
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", device=None, use_graph=True, **ignorekwargs):
        if attn_type != "vanilla" or use_linear_attn:
            raise NotImplementedError("only attn_type='vanilla' exists in the reference (model.py:209-212)")
        if give_pre_end or tanh_out or not resamp_with_conv:
            raise NotImplementedError("give_pre_end / tanh_out / resamp_with_conv=False are not used by any shipped config")
        self.ch, self.out_ch, self.ch_mult = ch, out_ch, tuple(ch_mult)
        self.num_res_blocks, self.attn_resolutions = num_res_blocks, tuple(attn_resolutions)
        self.resolution, self.z_channels = resolution, z_channels
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()
                                                                                   if torch.cuda.is_available() else 0)
        self.use_graph = use_graph
        self.zc_pad = 8
        self.out_pad = 8
        self.w = None
        self._plans = {}

    # ------------------------------------------------------------------ structure (model.py:336-406)
    def _structure(self):
        nres = len(self.ch_mult)
        block_in = self.ch * self.ch_mult[-1]
        curr_res = self.resolution // 2 ** (nres - 1)
        seq = [("mid.block_1.", "res", block_in, block_in), ("mid.attn_1.", "attn", block_in, block_in),
               ("mid.block_2.", "res", block_in, block_in)]
        for lvl in reversed(range(nres)):
            block_out = self.ch * self.ch_mult[lvl]
            for i in range(self.num_res_blocks + 1):
                seq.append((f"up.{lvl}.block.{i}.", "res", block_in, block_out))
                block_in = block_out
                if curr_res in self.attn_resolutions:
                    seq.append((f"up.{lvl}.attn.{i}.", "attn", block_in, block_in))
            if lvl != 0:                      # model.py:423-424: the i_level == 0 Upsample exists but is never run
                seq.append((f"up.{lvl}.upsample.", "up", block_in, block_in))
                curr_res *= 2
        return seq, self.ch * self.ch_mult[-1], block_in

    def parameter_shapes(self, prefix=""):
        seq, first, last = self._structure()
        s = {prefix + "conv_in.weight": (first, self.z_channels, 3, 3), prefix + "conv_in.bias": (first,)}
        for pre, kind, cin, cout in seq:
            p = prefix + pre
            if kind == "res":
                s[p + "norm1.gamma"] = (cin,); s[p + "norm1.beta"] = (cin,)
                s[p + "conv1.weight"] = (cout, cin, 3, 3); s[p + "conv1.bias"] = (cout,)
                s[p + "norm2.gamma"] = (cout,); s[p + "norm2.beta"] = (cout,)
                s[p + "conv2.weight"] = (cout, cout, 3, 3); s[p + "conv2.bias"] = (cout,)
                if cin != cout:
                    s[p + "nin_shortcut.weight"] = (cout, cin, 1, 1); s[p + "nin_shortcut.bias"] = (cout,)
            elif kind == "attn":
                s[p + "norm.gamma"] = (cin,); s[p + "norm.beta"] = (cin,)
                for n in ("q", "k", "v", "proj_out"):
                    s[p + n + ".weight"] = (cin, cin, 1, 1); s[p + n + ".bias"] = (cin,)
            else:
                s[p + "conv.weight"] = (cin, cin, 3, 3); s[p + "conv.bias"] = (cin,)
        s[prefix + "norm_out.gamma"] = (last,); s[prefix + "norm_out.beta"] = (last,)
        s[prefix + "conv_out.weight"] = (self.out_ch, last, 3, 3); s[prefix + "conv_out.bias"] = (self.out_ch,)
        return s

    # ------------------------------------------------------------------ weights
    def _dev(self, a, dtype):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _vec(self, v, n=None):
        v = self._dev(v, f32)
        if n is None or v.numel() == n:
            return v
        out = torch.zeros(n, dtype=f32, device=self.device)
        out[: v.numel()] = v
        return out

    def _conv_w(self, wt, cin_pad=None, cout_pad=None):
        wt = wt if isinstance(wt, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(wt))
        return ops.pack_conv_weight(wt.to(self.device), cin_pad, cout_pad)

    def load_state_dict(self, params, prefix="", post_quant=None, strict=True):
        """params: reference parameter name (after `prefix`, e.g. 'decoder.') -> array.  `post_quant` = (weight
        [zc, embed, 1, 1], bias) of AutoencoderKL.post_quant_conv, which runs in front of conv_in (autoencoder.py:66)."""
        shapes = self.parameter_shapes(prefix)
        missing = [k for k in shapes if k not in params]
        if missing and strict:
            raise MdxError(f"Decoder.load_state_dict: missing {len(missing)} parameters, e.g. {missing[:3]}")
        for k, shp in shapes.items():
            if k in params and tuple(np.shape(params[k])) != tuple(shp):
                raise MdxError(f"Decoder.load_state_dict: {k} has shape {tuple(np.shape(params[k]))}, expected {shp}")
        g = lambda k: params[prefix + k]
        w = {}
        seq, first, last = self._structure()
        if post_quant is not None:
            pw, pb = post_quant
            pw = pw if isinstance(pw, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pw))
            w["pq.w"] = ops.pack_conv_weight(pw.to(self.device), self.zc_pad, self.zc_pad)
            w["pq.b"] = self._vec(pb, self.zc_pad)
        w["conv_in.w"] = self._conv_w(g("conv_in.weight"), cin_pad=self.zc_pad)
        w["conv_in.b"] = self._vec(g("conv_in.bias"))
        for pre, kind, cin, cout in seq:
            if kind == "res":
                for n in ("norm1", "norm2"):
                    w[pre + n + ".g"], w[pre + n + ".b"] = self._vec(g(pre + n + ".gamma")), self._vec(g(pre + n + ".beta"))
                for n in ("conv1", "conv2"):
                    w[pre + n + ".w"], w[pre + n + ".b"] = self._conv_w(g(pre + n + ".weight")), self._vec(g(pre + n + ".bias"))
                if cin != cout:
                    w[pre + "nin.w"] = self._conv_w(g(pre + "nin_shortcut.weight"))
                    w[pre + "nin.b"] = self._vec(g(pre + "nin_shortcut.bias"))
            elif kind == "attn":
                w[pre + "norm.g"], w[pre + "norm.b"] = self._vec(g(pre + "norm.gamma")), self._vec(g(pre + "norm.beta"))
                for n in ("q", "k", "v", "proj_out"):
                    w[pre + n + ".w"], w[pre + n + ".b"] = self._conv_w(g(pre + n + ".weight")), self._vec(g(pre + n + ".bias"))
            else:
                w[pre + "conv.w"], w[pre + "conv.b"] = self._conv_w(g(pre + "conv.weight")), self._vec(g(pre + "conv.bias"))
        w["norm_out.g"], w["norm_out.b"] = self._vec(g("norm_out.gamma")), self._vec(g("norm_out.beta"))
        w["conv_out.w"] = self._conv_w(g("conv_out.weight"), cout_pad=self.out_pad)
        w["conv_out.b"] = self._vec(g("conv_out.bias"), self.out_pad)
        self.w = w
        self._plans.clear()

    # ------------------------------------------------------------------ plan
    class _Plan:
        graph = None
        graph_failed = False

    def _plan(self, B, H, W):
        key = (B, H, W)
        if key in self._plans:
            return self._plans[key]
        if self.w is None:
            raise MdxError("Decoder: load_state_dict() must be called before decode")
        dev, w = self.device, self.w
        P = Decoder._Plan()
        pb = _PlanBuilder(self, P, B)
        A = pb.A
        P.z_static = torch.zeros((B, self.z_channels, H, W), dtype=f32, device=dev)
        zin = A.get((B, H * W, self.zc_pad))
        pb.emit(lambda: ops.nchw_to_nhwc(P.z_static, self.zc_pad, out=zin), "small")
        hcur = zin
        if "pq.w" in w:                                  # AutoencoderKL.post_quant_conv (1x1)
            hcur = pb.conv1(zin, H * W, self.zc_pad, self.zc_pad, w["pq.w"], w["pq.b"])
            A.release(zin)
        seq, first, last = self._structure()
        h, wd = H, W
        nxt = pb.conv3(hcur, self.zc_pad, first, w["conv_in.w"], w["conv_in.b"], h, wd)
        A.release(hcur)
        hcur = nxt
        for pre, kind, cin, cout in seq:
            if kind == "res":
                nxt = pb.resblock(pre, hcur, cin, cout, h, wd)
            elif kind == "attn":
                nxt = pb.attnblock(pre, hcur, cin, h, wd)
            else:                                        # Upsample model.py:45-52: nearest x2 folded into the gather
                nxt = pb.conv3(hcur, cin, cin, w[pre + "conv.w"], w[pre + "conv.b"], h, wd, upsample=1)
                h, wd = 2 * h, 2 * wd
            A.release(hcur)
            hcur = nxt
        a = A.get((B, h * wd, last))
        pb.gn(hcur, w["norm_out.g"], w["norm_out.b"], True, a)
        A.release(hcur)
        y = pb.conv3(a, last, self.out_ch, w["conv_out.w"], w["conv_out.b"], h, wd, n_store=self.out_pad)
        A.release(a)
        P.out_nchw = torch.empty((B, self.out_ch, h, wd), dtype=f32, device=dev)
        pb.emit(lambda: ops.nhwc_to_nchw(y, self.out_ch, h, wd, out=P.out_nchw), "small")
        pb.finish()
        P.out_hw = (h, wd)
        self._plans[key] = P
        return P

    # ------------------------------------------------------------------ run
    def _run(self, P):
        _run_plan(self, P)

    def construct(self, z):
        """model.py:408-440.  z [B, z_channels, h, w] fp32 on the GPU -> image [B, out_ch, 8h, 8w] fp32 (a buffer
        owned by the plan, overwritten by the next call)."""
        if not (isinstance(z, torch.Tensor) and z.is_cuda):
            raise MdxError("Decoder: z must be a CUDA(HIP) tensor (no CPU fallback)")
        B, C, H, W = z.shape
        if C != self.z_channels:
            raise MdxError(f"Decoder: expected {self.z_channels} latent channels, got {C}")
        P = self._plan(B, H, W)
        P.z_static.copy_(z)
        _run_plan(self, P)
        return P.out_nchw

    __call__ = construct


class Encoder:
    """model.py:216-318: conv_in, per level `num_res_blocks` ResnetBlocks (+ Downsample except at the last level: zero
    pad bottom/right + valid 3x3 stride 2 -> mdx_gemm_desc.asym_pad), the mid block with its single-head attention,
    GroupNorm + swish, conv_out to 2 * z_channels moments.  Same planned execution as Decoder."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", device=None, use_graph=True, **ignore_kwargs):
        if attn_type != "vanilla" or use_linear_attn or not resamp_with_conv or not double_z:
            raise NotImplementedError("only the shipped configuration (vanilla attention, conv resampling, double_z) is built")
        self.ch, self.ch_mult, self.num_res_blocks = ch, tuple(ch_mult), num_res_blocks
        self.attn_resolutions, self.resolution = tuple(attn_resolutions), resolution
        self.in_channels, self.z_channels = in_channels, z_channels
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()
                                                                                   if torch.cuda.is_available() else 0)
        self.use_graph = use_graph
        self.cin_pad, self.mom_pad = 8, (2 * z_channels + 7) // 8 * 8
        self.w = None
        self._plans = {}

    _dev, _vec, _conv_w = Decoder._dev, Decoder._vec, Decoder._conv_w

    def _structure(self):
        nres = len(self.ch_mult)
        in_mult = (1,) + self.ch_mult
        curr_res = self.resolution
        seq = []
        block_in = self.ch
        for lvl in range(nres):
            block_in, block_out = self.ch * in_mult[lvl], self.ch * self.ch_mult[lvl]
            for i in range(self.num_res_blocks):
                seq.append((f"down.{lvl}.block.{i}.", "res", block_in, block_out))
                block_in = block_out
                if curr_res in self.attn_resolutions:
                    seq.append((f"down.{lvl}.attn.{i}.", "attn", block_in, block_in))
            if lvl != nres - 1:                  # model.py:301-302: the last level's Downsample exists but never runs
                seq.append((f"down.{lvl}.downsample.", "down", block_in, block_in))
                curr_res //= 2
        seq += [("mid.block_1.", "res", block_in, block_in), ("mid.attn_1.", "attn", block_in, block_in),
                ("mid.block_2.", "res", block_in, block_in)]
        return seq, self.ch, block_in

    def parameter_shapes(self, prefix=""):
        seq, first, last = self._structure()
        s = {prefix + "conv_in.weight": (first, self.in_channels, 3, 3), prefix + "conv_in.bias": (first,)}
        for pre, kind, cin, cout in seq:
            p = prefix + pre
            if kind == "res":
                s[p + "norm1.gamma"] = (cin,); s[p + "norm1.beta"] = (cin,)
                s[p + "conv1.weight"] = (cout, cin, 3, 3); s[p + "conv1.bias"] = (cout,)
                s[p + "norm2.gamma"] = (cout,); s[p + "norm2.beta"] = (cout,)
                s[p + "conv2.weight"] = (cout, cout, 3, 3); s[p + "conv2.bias"] = (cout,)
                if cin != cout:
                    s[p + "nin_shortcut.weight"] = (cout, cin, 1, 1); s[p + "nin_shortcut.bias"] = (cout,)
            elif kind == "attn":
                s[p + "norm.gamma"] = (cin,); s[p + "norm.beta"] = (cin,)
                for n in ("q", "k", "v", "proj_out"):
                    s[p + n + ".weight"] = (cin, cin, 1, 1); s[p + n + ".bias"] = (cin,)
            else:
                s[p + "conv.weight"] = (cin, cin, 3, 3); s[p + "conv.bias"] = (cin,)
        s[prefix + "norm_out.gamma"] = (last,); s[prefix + "norm_out.beta"] = (last,)
        s[prefix + "conv_out.weight"] = (2 * self.z_channels, last, 3, 3); s[prefix + "conv_out.bias"] = (2 * self.z_channels,)
        return s

    def load_state_dict(self, params, prefix="", quant=None, strict=True):
        """`quant` = (weight [2*embed, 2*zc, 1, 1], bias) of AutoencoderKL.quant_conv, applied after conv_out (autoencoder.py:72)."""
        shapes = self.parameter_shapes(prefix)
        missing = [k for k in shapes if k not in params]
        if missing and strict:
            raise MdxError(f"Encoder.load_state_dict: missing {len(missing)} parameters, e.g. {missing[:3]}")
        for k, shp in shapes.items():
            if k in params and tuple(np.shape(params[k])) != tuple(shp):
                raise MdxError(f"Encoder.load_state_dict: {k} has shape {tuple(np.shape(params[k]))}, expected {shp}")
        g = lambda k: params[prefix + k]
        w = {}
        seq, first, last = self._structure()
        w["conv_in.w"] = self._conv_w(g("conv_in.weight"), cin_pad=self.cin_pad)
        w["conv_in.b"] = self._vec(g("conv_in.bias"))
        for pre, kind, cin, cout in seq:
            if kind == "res":
                for n in ("norm1", "norm2"):
                    w[pre + n + ".g"], w[pre + n + ".b"] = self._vec(g(pre + n + ".gamma")), self._vec(g(pre + n + ".beta"))
                for n in ("conv1", "conv2"):
                    w[pre + n + ".w"], w[pre + n + ".b"] = self._conv_w(g(pre + n + ".weight")), self._vec(g(pre + n + ".bias"))
                if cin != cout:
                    w[pre + "nin.w"] = self._conv_w(g(pre + "nin_shortcut.weight"))
                    w[pre + "nin.b"] = self._vec(g(pre + "nin_shortcut.bias"))
            elif kind == "attn":
                w[pre + "norm.g"], w[pre + "norm.b"] = self._vec(g(pre + "norm.gamma")), self._vec(g(pre + "norm.beta"))
                for n in ("q", "k", "v", "proj_out"):
                    w[pre + n + ".w"], w[pre + n + ".b"] = self._conv_w(g(pre + n + ".weight")), self._vec(g(pre + n + ".bias"))
            else:
                w[pre + "conv.w"], w[pre + "conv.b"] = self._conv_w(g(pre + "conv.weight")), self._vec(g(pre + "conv.bias"))
        w["norm_out.g"], w["norm_out.b"] = self._vec(g("norm_out.gamma")), self._vec(g("norm_out.beta"))
        w["conv_out.w"] = self._conv_w(g("conv_out.weight"), cout_pad=self.mom_pad)
        w["conv_out.b"] = self._vec(g("conv_out.bias"), self.mom_pad)
        if quant is not None:
            qw, qb = quant
            qw = qw if isinstance(qw, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(qw))
            w["q.w"] = ops.pack_conv_weight(qw.to(self.device), self.mom_pad, self.mom_pad)
            w["q.b"] = self._vec(qb, self.mom_pad)
        self.w = w
        self._plans.clear()

    _Plan = Decoder._Plan

    def _plan(self, B, H, W):
        key = (B, H, W)
        if key in self._plans:
            return self._plans[key]
        if self.w is None:
            raise MdxError("Encoder: load_state_dict() must be called before encode")
        nres = len(self.ch_mult)
        if H % (1 << (nres - 1)) or W % (1 << (nres - 1)):
            raise MdxError(f"Encoder: image {H}x{W} is not divisible by 2^{nres - 1}")
        dev, w = self.device, self.w
        P = Decoder._Plan()
        pb = _PlanBuilder(self, P, B)
        A = pb.A
        P.x_static = torch.zeros((B, self.in_channels, H, W), dtype=f32, device=dev)
        xin = A.get((B, H * W, self.cin_pad))
        pb.emit(lambda: ops.nchw_to_nhwc(P.x_static, self.cin_pad, out=xin), "small")
        seq, first, last = self._structure()
        h, wd = H, W
        hcur = pb.conv3(xin, self.cin_pad, first, w["conv_in.w"], w["conv_in.b"], h, wd)
        A.release(xin)
        for pre, kind, cin, cout in seq:
            if kind == "res":
                nxt = pb.resblock(pre, hcur, cin, cout, h, wd)
            elif kind == "attn":
                nxt = pb.attnblock(pre, hcur, cin, h, wd)
            else:                                        # Downsample model.py:70-75
                nxt = pb.conv3(hcur, cin, cin, w[pre + "conv.w"], w[pre + "conv.b"], h, wd, stride=2, asym_pad=1)
                h, wd = h // 2, wd // 2
            A.release(hcur)
            hcur = nxt
        a = A.get((B, h * wd, last))
        pb.gn(hcur, w["norm_out.g"], w["norm_out.b"], True, a)
        A.release(hcur)
        mom = pb.conv3(a, last, 2 * self.z_channels, w["conv_out.w"], w["conv_out.b"], h, wd, n_store=self.mom_pad)
        A.release(a)
        if "q.w" in w:                                   # AutoencoderKL.quant_conv (1x1) on the moments
            mom = pb.conv1(mom, h * wd, self.mom_pad, self.mom_pad, w["q.w"], w["q.b"])
        P.moments = mom                                  # NHWC fp16 [B, h*w, mom_pad] = [mean | logvar | pad]
        pb.finish()
        P.out_hw = (h, wd)
        self._plans[key] = P
        return P

    def construct(self, x):
        """model.py:293-318 (+ quant_conv when loaded).  x [B, in_channels, H, W] fp32 on the GPU -> the plan's moments
        buffer, NHWC fp16 [B, (H/8)*(W/8), mom_pad] (mean | logvar), overwritten by the next call."""
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            raise MdxError("Encoder: x must be a CUDA(HIP) tensor (no CPU fallback)")
        B, C, H, W = x.shape
        if C != self.in_channels:
            raise MdxError(f"Encoder: expected {self.in_channels} image channels, got {C}")
        P = self._plan(B, H, W)
        P.x_static.copy_(x)
        _run_plan(self, P)
        return P.moments

    __call__ = construct
