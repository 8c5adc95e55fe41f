"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PARITY UNPINNED by the reference (no tests / fixtures; MindSpore
cannot run here).

fp32 PyTorch-CPU restatement of the VAE, SURVEY.md 8(f) items 1 and 4:
    AutoencoderKL.decode / .encode    /root/reference/vision/stablediffusionv2/ldm/models/autoencoder.py:65-78
    Decoder / Encoder / ResnetBlock / AttnBlock / Upsample / Downsample / Normalize / nonlinearity
                                      .../ldm/modules/diffusionmodules/model.py:21-78, 80-206, 216-440
    LatentDiffusion.decode_first_stage .../ldm/models/diffusion/ddpm.py:286-288  (z / scale_factor)
Parameter names are the MindSpore Cell attribute paths (GroupNorm: gamma / beta; Conv2d: weight / bias).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SD_VAE = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 4, 4),
              num_res_blocks=2, attn_resolutions=(), dropout=0.0)    # configs/v2-inference.yaml:46-60


def _levels(dd):
    """(name prefix, kind, cin, cout) in execution order; kinds: res | attn | up."""
    ch, mult, nrb = dd["ch"], tuple(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(mult)
    block_in = ch * mult[-1]
    curr_res = dd["resolution"] // 2 ** (nres - 1)
    seq = [("decoder.mid.block_1.", "res", block_in, block_in), ("decoder.mid.attn_1.", "attn", block_in, block_in),
           ("decoder.mid.block_2.", "res", block_in, block_in)]
    for lvl in reversed(range(nres)):
        block_out = ch * mult[lvl]
        for i in range(nrb + 1):
            seq.append((f"decoder.up.{lvl}.block.{i}.", "res", block_in, block_out))
            block_in = block_out
            if curr_res in tuple(dd["attn_resolutions"]):
                seq.append((f"decoder.up.{lvl}.attn.{i}.", "attn", block_in, block_in))
        if lvl != 0:
            seq.append((f"decoder.up.{lvl}.upsample.", "up", block_in, block_in))   # model.py:423-424
            curr_res *= 2
    return seq, ch * mult[-1], block_in


def _enc_levels(dd):
    """Encoder execution order (model.py:293-312): (prefix, kind, cin, cout); kinds: res | attn | down."""
    ch, mult, nrb = dd["ch"], tuple(dd["ch_mult"]), dd["num_res_blocks"]
    in_mult = (1,) + mult
    curr_res = dd["resolution"]
    seq, block_in = [], ch
    for lvl in range(len(mult)):
        block_in, block_out = ch * in_mult[lvl], ch * mult[lvl]
        for i in range(nrb):
            seq.append((f"encoder.down.{lvl}.block.{i}.", "res", block_in, block_out))
            block_in = block_out
            if curr_res in tuple(dd["attn_resolutions"]):
                seq.append((f"encoder.down.{lvl}.attn.{i}.", "attn", block_in, block_in))
        if lvl != len(mult) - 1:
            seq.append((f"encoder.down.{lvl}.downsample.", "down", block_in, block_in))
            curr_res //= 2
    seq += [("encoder.mid.block_1.", "res", block_in, block_in), ("encoder.mid.attn_1.", "attn", block_in, block_in),
            ("encoder.mid.block_2.", "res", block_in, block_in)]
    return seq, ch, block_in


def param_shapes(dd=SD_VAE, embed_dim=4):
    seq, first, last = _levels(dd)
    zc = dd["z_channels"]
    s = {"post_quant_conv.weight": (zc, embed_dim, 1, 1), "post_quant_conv.bias": (zc,),
         "decoder.conv_in.weight": (first, zc, 3, 3), "decoder.conv_in.bias": (first,)}
    eseq, efirst, elast = _enc_levels(dd)
    s["quant_conv.weight"] = (2 * embed_dim, 2 * zc, 1, 1); s["quant_conv.bias"] = (2 * embed_dim,)
    s["encoder.conv_in.weight"] = (efirst, dd["in_channels"], 3, 3); s["encoder.conv_in.bias"] = (efirst,)
    s["encoder.norm_out.gamma"] = (elast,); s["encoder.norm_out.beta"] = (elast,)
    s["encoder.conv_out.weight"] = (2 * zc, elast, 3, 3); s["encoder.conv_out.bias"] = (2 * zc,)
    for pre, kind, cin, cout in seq + eseq:
        if kind == "res":
            s[pre + "norm1.gamma"] = (cin,); s[pre + "norm1.beta"] = (cin,)
            s[pre + "conv1.weight"] = (cout, cin, 3, 3); s[pre + "conv1.bias"] = (cout,)
            s[pre + "norm2.gamma"] = (cout,); s[pre + "norm2.beta"] = (cout,)
            s[pre + "conv2.weight"] = (cout, cout, 3, 3); s[pre + "conv2.bias"] = (cout,)
            if cin != cout:
                s[pre + "nin_shortcut.weight"] = (cout, cin, 1, 1); s[pre + "nin_shortcut.bias"] = (cout,)
        elif kind == "attn":
            s[pre + "norm.gamma"] = (cin,); s[pre + "norm.beta"] = (cin,)
            for n in ("q", "k", "v", "proj_out"):
                s[pre + n + ".weight"] = (cin, cin, 1, 1); s[pre + n + ".bias"] = (cin,)
        else:
            s[pre + "conv.weight"] = (cin, cin, 3, 3); s[pre + "conv.bias"] = (cin,)
    s["decoder.norm_out.gamma"] = (last,); s["decoder.norm_out.beta"] = (last,)
    s["decoder.conv_out.weight"] = (dd["out_ch"], last, 3, 3); s["decoder.conv_out.bias"] = (dd["out_ch"],)
    return s


def init_params(dd=SD_VAE, embed_dim=4, seed=0):
    """Seeded synthetic weights (fan-in scaled so that activations stay O(1) through 30 layers)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shp in param_shapes(dd, embed_dim).items():
        if name.endswith("gamma"):
            out[name] = (1.0 + 0.1 * rng.standard_normal(shp)).astype(np.float32)
        elif name.endswith("beta") or name.endswith("bias"):
            out[name] = (0.05 * rng.standard_normal(shp)).astype(np.float32)
        else:
            fan_in = int(np.prod(shp[1:]))
            out[name] = (rng.standard_normal(shp) / math.sqrt(fan_in)).astype(np.float32)
    return out


def _t(p, k):
    return torch.as_tensor(p[k], dtype=torch.float32)


def _norm(p, pre, x):   # Normalize: GroupNorm(32, C, eps=1e-6) model.py:26-28
    return F.group_norm(x, 32, _t(p, pre + "gamma"), _t(p, pre + "beta"), eps=1e-6)


def _swish(x):          # nonlinearity model.py:21-23
    return x * torch.sigmoid(x)


def _conv(p, pre, x, pad):
    return F.conv2d(x, _t(p, pre + "weight"), _t(p, pre + "bias"), padding=pad)


def resnet_block(p, pre, x, cin, cout):   # model.py:128-148 (temb is None in the VAE)
    h = _conv(p, pre + "conv1.", _swish(_norm(p, pre + "norm1.", x)), 1)
    h = _conv(p, pre + "conv2.", _swish(_norm(p, pre + "norm2.", h)), 1)
    if cin != cout:
        x = _conv(p, pre + "nin_shortcut.", x, 0)
    return x + h


def attn_block(p, pre, x):                # model.py:182-206
    h_ = _norm(p, pre + "norm.", x)
    q, k, v = (_conv(p, pre + n + ".", h_, 0) for n in ("q", "k", "v"))
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = torch.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(p, pre + "proj_out.", h_, 0)


def decode(p, z, dd=SD_VAE):
    """AutoencoderKL.decode autoencoder.py:65-68 + Decoder.construct model.py:408-440.  z [B, zc, h, w] fp32."""
    z = torch.as_tensor(z, dtype=torch.float32)
    seq, _, _ = _levels(dd)
    h = _conv(p, "post_quant_conv.", z, 0)
    h = _conv(p, "decoder.conv_in.", h, 1)
    for pre, kind, cin, cout in seq:
        if kind == "res":
            h = resnet_block(p, pre, h, cin, cout)
        elif kind == "attn":
            h = attn_block(p, pre, h)
        else:                                   # Upsample model.py:45-52: nearest x2 then conv
            h = _conv(p, pre + "conv.", F.interpolate(h, scale_factor=2, mode="nearest"), 1)
    h = _swish(_norm(p, "decoder.norm_out.", h))
    return _conv(p, "decoder.conv_out.", h, 1)


def encode_moments(p, x, dd=SD_VAE):
    """Encoder.construct model.py:293-318 followed by quant_conv (autoencoder.py:71-72).  x [B, 3, H, W] fp32."""
    x = torch.as_tensor(x, dtype=torch.float32)
    seq, _, _ = _enc_levels(dd)
    h = _conv(p, "encoder.conv_in.", x, 1)
    for pre, kind, cin, cout in seq:
        if kind == "res":
            h = resnet_block(p, pre, h, cin, cout)
        elif kind == "attn":
            h = attn_block(p, pre, h)
        else:                                   # Downsample model.py:70-75: pad bottom / right by one, valid 3x3 stride 2
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), _t(p, pre + "conv.weight"), _t(p, pre + "conv.bias"), stride=2)
    h = _swish(_norm(p, "encoder.norm_out.", h))
    h = _conv(p, "encoder.conv_out.", h, 1)
    return _conv(p, "quant_conv.", h, 0)


def encode(p, x, noise, dd=SD_VAE):
    """AutoencoderKL.encode autoencoder.py:70-78 with the N(0,1) draw injected (noise None: the mode)."""
    mean, logvar = encode_moments(p, x, dd).chunk(2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    if noise is None:
        return mean
    return mean + torch.exp(0.5 * logvar) * torch.as_tensor(noise, dtype=torch.float32)
