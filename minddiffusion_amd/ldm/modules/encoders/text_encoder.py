"""TextEncoder on the MI355X kernels -- mirror of the reference's ldm/modules/encoders/text_encoder.py
(TextEncoder :114-153, Transformer :100-111, ResidualAttentionBlock :81-97, MultiheadAttention :25-66; the Wukong copy
differs only in its real QuickGELU, WK text_encoder.py:67-74).  SURVEY.md 8(f) item 2: the step right before the
denoising loop -- it turns token ids into the [B, 77, width] conditioning that the UNet's cross-attention reads.

Planned executor like UNetModel: token + positional embedding (one gather kernel), then per layer
    LayerNorm -> in_proj as two GEMMs (q|k row-major, v stored transposed) -> causal flash attention -> out_proj (+x)
    LayerNorm -> c_fc with the GELU fused in the epilogue -> c_proj (+x)
and the final LayerNorm; 8 launches per layer, captured as one hipGraph.  The reference's [T, B, C] transposes
(:148-150) are layout only and disappear (token-major [B, T, C] throughout).  The sequence is padded from 77 to 80 tokens
(the transposed V store wants a multiple of 8): under the causal mask the 3 trailing pad positions cannot influence the
77 real ones, and they are dropped from the result.
"""
import numpy as np
import torch

from ...._lib import MdxError
from .... import ops
from ..diffusionmodules.openaimodel import _Arena

f16, f32 = torch.float16, torch.float32


class TextEncoder:
    def __init__(self, context_length, vocab_size, output_dim, width, layers, heads, dtype=None, act="gelu_tanh",
                 device=None, use_graph=True, ln_eps=1e-5):
        if width % heads or (width // heads) not in (40, 64, 80, 160):
            raise MdxError(f"TextEncoder: head dim {width / heads} is not supported by mdx_attention_f16 (40/64/80/160)")
        if act not in ("gelu_tanh", "quick_gelu"):
            raise ValueError("act must be 'gelu_tanh' (SDv2: nn.GELU) or 'quick_gelu' (Wukong: x * sigmoid(1.702 x))")
        self.context_length, self.vocab_size, self.output_dim = context_length, vocab_size, output_dim
        self.width, self.layers, self.heads, self.act = width, layers, heads, act
        self.t_pad = (context_length + 7) // 8 * 8
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()
                                                                                   if torch.cuda.is_available() else 0)
        self.use_graph = use_graph
        # ln_1 / ln_2 epsilon: SDv2 passes epsilon=1e-5 (text_encoder.py:84,93); Wukong builds nn.LayerNorm([d_model]) with
        # MindSpore's default 1e-7 (WK text_encoder.py:91,100).  ln_final is the default 1e-7 in both.
        self.ln_eps = float(ln_eps)
        self.w = None
        self._plans = {}

    def parameter_shapes(self, prefix=""):
        w = self.width
        s = {prefix + "embedding_table": (self.vocab_size, w), prefix + "positional_embedding": (self.context_length, w),
             prefix + "ln_final.gamma": (w,), prefix + "ln_final.beta": (w,)}
        for i in range(self.layers):
            b = f"{prefix}transformer_layer.resblocks.{i}."
            s[b + "attn.attn.in_proj.weight"] = (3 * w, w); s[b + "attn.attn.in_proj.bias"] = (3 * w,)
            s[b + "attn.attn.out_proj.weight"] = (w, w); s[b + "attn.attn.out_proj.bias"] = (w,)
            s[b + "ln_1.gamma"] = (w,); s[b + "ln_1.beta"] = (w,)
            s[b + "c_fc.weight"] = (4 * w, w); s[b + "c_fc.bias"] = (4 * w,)
            s[b + "c_proj.weight"] = (w, 4 * w); s[b + "c_proj.bias"] = (w,)
            s[b + "ln_2.gamma"] = (w,); s[b + "ln_2.beta"] = (w,)
        return s

    def _dev(self, a, dtype):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def load_state_dict(self, params, prefix="", strict=True):
        shapes = self.parameter_shapes(prefix)
        from ....weights import check_state_dict
        check_state_dict(shapes, {k: v for k, v in params.items() if strict or k in shapes}, True,
                         "TextEncoder.load_state_dict")
        wd = self.width
        g = lambda k: params[prefix + k]
        w = {"emb": self._dev(g("embedding_table"), f16)}
        pos = torch.zeros((self.t_pad, wd), dtype=f16, device=self.device)      # pad rows: zeros (never read back)
        pos[: self.context_length] = self._dev(g("positional_embedding"), f16)
        w["pos"] = pos
        w["lnf.g"], w["lnf.b"] = self._dev(g("ln_final.gamma"), f32), self._dev(g("ln_final.beta"), f32)
        for i in range(self.layers):
            b, o = f"transformer_layer.resblocks.{i}.", f"l{i}."
            ipw, ipb = self._dev(g(b + "attn.attn.in_proj.weight"), f16), self._dev(g(b + "attn.attn.in_proj.bias"), f32)
            w[o + "qk.w"], w[o + "qk.b"] = ops.pack_gemm_weight(ipw[: 2 * wd].contiguous()), ipb[: 2 * wd].contiguous()
            w[o + "v.w"], w[o + "v.b"] = ops.pack_gemm_weight(ipw[2 * wd:].contiguous()), ipb[2 * wd:].contiguous()
            w[o + "out.w"] = ops.pack_gemm_weight(self._dev(g(b + "attn.attn.out_proj.weight"), f16))
            w[o + "out.b"] = self._dev(g(b + "attn.attn.out_proj.bias"), f32)
            w[o + "fc.w"] = ops.pack_gemm_weight(self._dev(g(b + "c_fc.weight"), f16))
            w[o + "fc.b"] = self._dev(g(b + "c_fc.bias"), f32)
            w[o + "proj.w"] = ops.pack_gemm_weight(self._dev(g(b + "c_proj.weight"), f16))
            w[o + "proj.b"] = self._dev(g(b + "c_proj.bias"), f32)
            for n in ("ln_1", "ln_2"):
                w[o + n + ".g"], w[o + n + ".b"] = self._dev(g(b + n + ".gamma"), f32), self._dev(g(b + n + ".beta"), f32)
        self.w = w
        self._plans.clear()

    class _Plan:
        graph = None
        graph_failed = False

    def _plan(self, B):
        if B in self._plans:
            return self._plans[B]
        if self.w is None:
            raise MdxError("TextEncoder: load_state_dict() must be called before the first forward")
        dev, w, wd, T, H = self.device, self.w, self.width, self.t_pad, self.heads
        dh = wd // H
        P = TextEncoder._Plan()
        A = _Arena(dev)
        main, descs = [], []
        P.tokens = torch.zeros((B, T), dtype=torch.int32, device=dev)
        P.ones = torch.ones((B, T), dtype=torch.int32, device=dev)
        epi = ops.EPI_GELU if self.act == "gelu_tanh" else ops.EPI_QUICKGELU

        def gemm(**kw):
            d = ops.make_gemm_desc(**kw)
            descs.append(d)
            main.append(lambda d=d: ops.gemm_run(d))

        x = A.get((B, T, wd))
        # gather(embedding_table, ids) + positional_embedding (:144-147); mask all ones, so `pad` is never read
        main.append(lambda: ops.glide_text_embed(P.tokens, P.ones, w["emb"], w["pos"], w["pos"], out=x))
        a = A.get((B, T, wd))
        qk = A.get((B, T, 2 * wd))
        vt = A.get((B, wd, T))
        o = A.get((B, T, wd))
        h = A.get((B, T, 4 * wd))
        x2 = A.get((B, T, wd))
        scale = float(dh) ** -0.5
        for i in range(self.layers):
            L = f"l{i}."
            main.append(lambda x=x, L=L: ops.layernorm(x.view(B * T, wd), w[L + "ln_1.g"], w[L + "ln_1.b"], self.ln_eps,
                                                       out=a.view(B * T, wd)))
            gemm(a=a, w=w[L + "qk.w"], N=2 * wd, B=B, H=T, W=1, c1=wd, out=qk, out_ld=2 * wd, bias=w[L + "qk.b"])
            gemm(a=a, w=w[L + "v.w"], N=wd, B=B, H=T, W=1, c1=wd, out=vt, out_ld=T, bias=w[L + "v.b"],
                 out_mode=ops.OUT_TRANSPOSED)
            main.append(lambda: ops.attention(qk.data_ptr(), qk.data_ptr() + wd * 2, vt.data_ptr(), o.data_ptr(), B, H, dh,
                                              T, T, scale, T * 2 * wd, 2 * wd, T * 2 * wd, 2 * wd, wd * T, T, T * wd, wd,
                                              causal=True))                                     # mask :136-139, :57-60
            gemm(a=o, w=w[L + "out.w"], N=wd, B=B, H=T, W=1, c1=wd, out=x2, out_ld=wd, bias=w[L + "out.b"],
                 residual=x, residual_ld=wd)                                                     # x + attn(ln_1(x)) :94
            main.append(lambda x2=x2, L=L: ops.layernorm(x2.view(B * T, wd), w[L + "ln_2.g"], w[L + "ln_2.b"], self.ln_eps,
                                                         out=a.view(B * T, wd)))
            gemm(a=a, w=w[L + "fc.w"], N=4 * wd, B=B, H=T, W=1, c1=wd, out=h, out_ld=4 * wd, bias=w[L + "fc.b"],
                 epilogue=epi)
            gemm(a=h, w=w[L + "proj.w"], N=wd, B=B, H=T, W=1, c1=4 * wd, out=x, out_ld=wd, bias=w[L + "proj.b"],
                 residual=x2, residual_ld=wd)                                                    # x + mlp(ln_2(x)) :95
        P.out = torch.empty((B, T, wd), dtype=f16, device=dev)
        # ln_final = nn.LayerNorm([width]): MindSpore's default epsilon is 1e-7 (:132)
        main.append(lambda: ops.layernorm(x.view(B * T, wd), w["lnf.g"], w["lnf.b"], 1e-7, out=P.out.view(B * T, wd)))
        need = max([ops.gemm_workspace_bytes(d) for d in descs] + [16])
        P.gemm_ws = ops.new_gemm_workspace(need, dev)
        for d in descs:
            d.workspace, d.workspace_bytes = P.gemm_ws.data_ptr(), P.gemm_ws.numel() * 4
        P.main, P.descs, P.arena = main, descs, A
        self._plans[B] = P
        return P

    def construct(self, text):
        """text_encoder.py:141-153.  text: int token ids [B, context_length] (any integer tensor / array) ->
        [B, context_length, width] fp16 on the GPU."""
        tok = torch.as_tensor(np.asarray(text) if not isinstance(text, torch.Tensor) else text)
        if tok.dim() != 2 or tok.shape[1] != self.context_length:
            raise MdxError(f"TextEncoder: expected token ids [B, {self.context_length}], got {tuple(tok.shape)}")
        if not torch.cuda.is_available() or self.device.type != "cuda":
            raise MdxError("TextEncoder: the HIP device is required (no CPU fallback)")
        B = tok.shape[0]
        P = self._plan(B)
        P.tokens[:, : self.context_length].copy_(tok.to(device=self.device, dtype=torch.int32))
        if self.use_graph and not P.graph_failed:
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
                return P.out[:, : self.context_length].clone()   # a fresh tensor: callers keep c and uc side by side
        for op in P.main:
            op()
        return P.out[:, : self.context_length].clone()

    __call__ = construct
