"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PARITY UNPINNED by the reference (no tests / fixtures; MindSpore
cannot run here).

fp32 PyTorch-CPU restatement of the text-conditioning transformer, SURVEY.md 8(f) item 2:
    TextEncoder / Transformer / ResidualAttentionBlock / MultiheadAttention
        /root/reference/vision/stablediffusionv2/ldm/modules/encoders/text_encoder.py:25-153
    FrozenCLIPEmbedder_ZH (SDv2: 77 tokens, vocab 49408, width 1024, 23 layers, 16 heads)
        .../ldm/modules/encoders/modules.py:23-41
    Wukong-Huahua variant: real QuickGELU x * sigmoid(1.702 x)
        /root/reference/vision/wukong-huahua/ldm/modules/encoders/text_encoder.py:67-74
The tokenizers (BPE / WordPiece) are host-side preprocessing outside SURVEY 8: inputs here are token ids.
Parameter names = MindSpore Cell attribute paths (LayerNorm: gamma / beta; Dense: weight [out, in] / bias).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SD2_TEXT = dict(context_length=77, vocab_size=49408, width=1024, layers=23, heads=16, act="gelu_tanh", ln_eps=1e-5)
# wukong-huahua/ldm/modules/encoders/modules.py:30: width 768, 12 layers, 12 heads; real QuickGELU (text_encoder.py:67-74)
# ln_1 / ln_2 = nn.LayerNorm([d_model]) without an epsilon argument: MindSpore's default 1e-7 (WK text_encoder.py:91,100)
WK_TEXT = dict(context_length=77, vocab_size=49408, width=768, layers=12, heads=12, act="quick_gelu", ln_eps=1e-7)


def param_shapes(cfg=SD2_TEXT, prefix="transformer."):
    w, L = cfg["width"], cfg["layers"]
    s = {prefix + "embedding_table": (cfg["vocab_size"], w), prefix + "positional_embedding": (cfg["context_length"], w),
         prefix + "ln_final.gamma": (w,), prefix + "ln_final.beta": (w,)}
    for i in range(L):
        b = f"{prefix}transformer_layer.resblocks.{i}."
        s[b + "attn.attn.in_proj.weight"] = (3 * w, w); s[b + "attn.attn.in_proj.bias"] = (3 * w,)
        s[b + "attn.attn.out_proj.weight"] = (w, w); s[b + "attn.attn.out_proj.bias"] = (w,)
        s[b + "ln_1.gamma"] = (w,); s[b + "ln_1.beta"] = (w,)
        s[b + "c_fc.weight"] = (4 * w, w); s[b + "c_fc.bias"] = (4 * w,)
        s[b + "c_proj.weight"] = (w, 4 * w); s[b + "c_proj.bias"] = (w,)
        s[b + "ln_2.gamma"] = (w,); s[b + "ln_2.beta"] = (w,)
    return s


def init_params(cfg=SD2_TEXT, seed=0, prefix="transformer."):
    """Seeded synthetic weights: tables as the reference initialises them (TruncatedNormal 0.02 / 0.01,
    text_encoder.py:126,131), dense weights fan-in scaled so that 23 residual layers stay O(1)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in param_shapes(cfg, prefix).items():
        if name.endswith("embedding_table"):
            out[name] = np.clip(rng.standard_normal(shp), -2, 2).astype(np.float32) * 0.5
        elif name.endswith("positional_embedding"):
            out[name] = np.clip(rng.standard_normal(shp), -2, 2).astype(np.float32) * 0.25
        elif name.endswith("gamma"):
            out[name] = (1.0 + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif name.endswith("beta") or name.endswith("bias"):
            out[name] = (0.05 * rng.standard_normal(shp)).astype(np.float32)
        else:
            out[name] = (rng.standard_normal(shp) / math.sqrt(shp[1])).astype(np.float32)
    return out


def _t(p, k):
    return torch.as_tensor(p[k], dtype=torch.float32)


def _act(x, kind):
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    return F.gelu(x, approximate="tanh")        # MindSpore nn.GELU() default approximate=True


def encode_tokens(p, tokens, cfg=SD2_TEXT, prefix="transformer."):
    """TextEncoder.construct text_encoder.py:141-153.  tokens [B, context_length] int -> [B, context_length, width]."""
    tokens = torch.as_tensor(np.asarray(tokens), dtype=torch.long)
    B, T = tokens.shape
    w, H = cfg["width"], cfg["heads"]
    d = w // H
    x = _t(p, prefix + "embedding_table")[tokens] + _t(p, prefix + "positional_embedding")        # :144-147
    mask = torch.triu(torch.full((T, T), float("-inf")), 1)                                          # :136-139
    for i in range(cfg["layers"]):
        b = f"{prefix}transformer_layer.resblocks.{i}."
        a = F.layer_norm(x, (w,), _t(p, b + "ln_1.gamma"), _t(p, b + "ln_1.beta"), eps=cfg.get("ln_eps", 1e-5))   # SD2 :84,93 (1e-5); WK :91,100 (default 1e-7)
        qkv = a @ _t(p, b + "attn.attn.in_proj.weight").T + _t(p, b + "attn.attn.in_proj.bias")      # :47
        q, k, v = qkv.split(w, dim=-1)
        q = (q * d ** -0.5).reshape(B, T, H, d).permute(0, 2, 1, 3)                                  # :54-55
        k = k.reshape(B, T, H, d).permute(0, 2, 1, 3)
        v = v.reshape(B, T, H, d).permute(0, 2, 1, 3)
        wts = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1)                                  # :58-60
        o = (wts @ v).permute(0, 2, 1, 3).reshape(B, T, w)
        x = x + (o @ _t(p, b + "attn.attn.out_proj.weight").T + _t(p, b + "attn.attn.out_proj.bias"))
        a = F.layer_norm(x, (w,), _t(p, b + "ln_2.gamma"), _t(p, b + "ln_2.beta"), eps=cfg.get("ln_eps", 1e-5))
        h = _act(a @ _t(p, b + "c_fc.weight").T + _t(p, b + "c_fc.bias"), cfg["act"])
        x = x + (h @ _t(p, b + "c_proj.weight").T + _t(p, b + "c_proj.bias"))
    # ln_final = nn.LayerNorm([width]) with MindSpore's default epsilon 1e-7 (text_encoder.py:132)
    return F.layer_norm(x, (w,), _t(p, prefix + "ln_final.gamma"), _t(p, prefix + "ln_final.beta"), eps=1e-7)
