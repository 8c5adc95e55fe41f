"""fp32 PyTorch-CPU restatement of the reference's latent-diffusion hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED by the reference.

Every function cites the reference file:line it follows (paths relative to
/root/reference/vision/stablediffusionv2 unless prefixed WK/ = wukong-huahua).
Layout is the reference's: NCHW activations, ``nn.Dense`` weights [out,in],
``nn.Conv2d`` weights [out,in,kh,kw]; parameters are looked up in a flat dict
keyed by the reference's parameter names.

MindSpore operator semantics restated here:
  * nn.GroupNorm(32, C, eps): biased variance over (C/32, H, W), affine gamma/beta.
  * nn.LayerNorm([C], epsilon=1e-5): last axis.
  * ops.GeLU: tanh approximation.
  * ops.ResizeNearestNeighbor (align_corners=False): exact 2x pixel duplication.
  * nn.Dropout(keep_prob=1.0): identity.
"""
import contextlib
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- precision mode
# Default: all-fp32 (the reference's `use_fp16: False` path).  `with emulate_fp16():` switches every primitive below to an
# emulation of the reference's SHIPPED mode `use_fp16: True` (configs/v2-inference.yaml:19,38): each
# MindSpore op that runs `.to_float(float16)` sees fp16-rounded inputs and parameters and returns an fp16-rounded result;
# the arithmetic INSIDE an op is kept in fp32 (what Ascend's cube / vector units do internally is not in the reference
# tree, so op-boundary rounding is a LOWER bound on the reference's fp16 noise).  Where the reference stays in fp32 on
# purpose -- GroupNorm (util.py:87-108 `.to_float(ms.float32)`), the sinusoid (util.py:122-126), the sampler's x -- so does
# the emulation.  Used to measure how far the fp16 reference itself is from this fp32 oracle.
class _Mode:
    fp16 = False


@contextlib.contextmanager
def emulate_fp16(flag=True):
    old = _Mode.fp16
    _Mode.fp16 = bool(flag)
    try:
        yield
    finally:
        _Mode.fp16 = old


def r16(t):
    """Round to fp16 and back in emulation mode (identity otherwise).  Tensors, numpy arrays and Python floats."""
    if not _Mode.fp16:
        return t
    if isinstance(t, torch.Tensor):
        return t.to(torch.float16).to(torch.float32)
    if isinstance(t, np.ndarray):
        return t.astype(np.float16).astype(np.float32)
    return float(np.float32(np.float16(t)))



# ----------------------------------------------------------------------------- schedule
def make_beta_schedule(schedule="linear", n_timestep=1000, linear_start=1e-4, linear_end=2e-2):
    """ldm/modules/diffusionmodules/util.py:172-185 -- fp32 linspace of sqrt, squared."""
    if schedule != "linear":
        raise ValueError(f"schedule '{schedule}' unknown.")
    start = np.float32(linear_start ** 0.5)
    stop = np.float32(linear_end ** 0.5)
    lin = np.linspace(start, stop, n_timestep, dtype=np.float32)
    return (lin ** 2).astype(np.float32)


def register_schedule(linear_start=0.00085, linear_end=0.0120, timesteps=1000):
    """ldm/models/diffusion/ddpm.py:111-139 (inference part), dtype fp32 ('use_fp16: False')."""
    betas = make_beta_schedule("linear", timesteps, linear_start, linear_end)
    alphas = 1.0 - betas
    alphas_cumprod = np.cumprod(alphas, axis=0)
    alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    return {
        "num_timesteps": int(timesteps),
        "betas": f32(betas),
        "alphas_cumprod": f32(alphas_cumprod),
        "alphas_cumprod_prev": f32(alphas_cumprod_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(alphas_cumprod)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - alphas_cumprod)),
    }


def make_ddim_timesteps(num_ddim_timesteps, num_ddpm_timesteps=1000, method="uniform"):
    """util.py:134-148."""
    if method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ts = np.asarray(list(range(0, num_ddpm_timesteps, c)), dtype=np.int64)
    elif method == "quad":
        ts = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * 0.8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(method)
    return ts + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta=0.0):
    """util.py:151-162."""
    alphacums = np.asarray(alphacums, dtype=np.float32)
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.concatenate([alphacums[:1], alphacums[ddim_timesteps[:-1]]]).astype(np.float32)
    sigmas = np.float32(eta) * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas.astype(np.float32), alphas, alphas_prev


def timestep_embedding(timesteps, dim, max_period=10000):
    """util.py:111-131 (fp32; timesteps may be fractional)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps.to(torch.float32)[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ----------------------------------------------------------------------------- primitives
def gelu_tanh(x):
    """ops.GeLU == tanh approximation."""
    x = r16(x)
    return r16(0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3))))


def silu(x):
    """nn.SiLU().to_float(dtype) (openaimodel.py:137): casts the fp32 GroupNorm output to fp16 first in fp16 mode."""
    x = r16(x)
    return r16(x * torch.sigmoid(x))


def group_norm(x, gamma, beta, eps, groups=32):
    """nn.GroupNorm(32, C, eps) on NCHW: biased variance over each (C/32,H,W) slab."""
    n, c = x.shape[:2]
    xr = x.reshape(n, groups, -1)
    mean = xr.mean(dim=2, keepdim=True)
    var = ((xr - mean) ** 2).mean(dim=2, keepdim=True)
    y = ((xr - mean) / torch.sqrt(var + eps)).reshape(x.shape)
    shape = (1, c) + (1,) * (x.dim() - 2)
    return y * gamma.reshape(shape) + beta.reshape(shape)


def layer_norm(x, gamma, beta, eps):
    """nn.LayerNorm([C], epsilon).to_float(dtype): an fp16 op in the reference's fp16 mode (attention.py:176-178)."""
    x = r16(x)
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return r16((x - mean) / torch.sqrt(var + eps) * r16(gamma) + r16(beta))


def dense(x, w, b=None):
    """nn.Dense: y = x W^T + b, W [out,in]."""
    y = r16(x) @ r16(w).t()
    return r16(y if b is None else y + r16(b))


def conv2d(x, w, b, stride=1, padding=1):
    """nn.Conv2d(pad_mode='pad'): NCHW cross-correlation with symmetric zero pad."""
    return r16(F.conv2d(r16(x), r16(w), None if b is None else r16(b), stride=stride, padding=padding))


def upsample_nearest2x(x):
    """ops.ResizeNearestNeighbor((2H,2W)), align_corners=False -> src = floor(dst/2)."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


# ----------------------------------------------------------------------------- UNet
SD2_UNET = dict(  # configs/v2-inference.yaml:21-38
    in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_head_channels=64, num_heads=-1,
    use_spatial_transformer=True, use_linear_in_transformer=True, transformer_depth=1,
    context_dim=1024, legacy=False)

WUKONG_UNET = dict(  # WK/configs/v1-inference-chinese.yaml:24-37
    in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, num_head_channels=-1,
    use_spatial_transformer=True, use_linear_in_transformer=False, transformer_depth=1,
    context_dim=768, legacy=False)


def unet_structure(cfg):
    """Walk UNetModel.__init__ (openaimodel.py:351-526) and return the block lists.

    Each block is a list of layer descriptors:
      ('conv', cin, cout) | ('res', cin, cout) | ('st', ch, heads, dim_head)
      | ('down', ch) | ('up', ch) | ('resdown', ch, ch) | ('resup', ch, ch)   (the last two: resblock_updown=True)
    The cfg-level switches (use_scale_shift_norm, conv_resample, transformer_depth, num_classes, n_embed) do not change the
    block lists, only what a layer of a kind does.
    """
    mc = cfg["model_channels"]
    nh_cfg = cfg.get("num_heads", -1)
    nhc = cfg.get("num_head_channels", -1)
    legacy = cfg.get("legacy", True)
    ust = cfg.get("use_spatial_transformer", False)
    assert ust, "only the spatial-transformer UNet is on the hot path (AttentionBlock is an empty stub)"

    def heads_for(ch, num_heads):
        if nhc == -1:
            dim_head = ch // num_heads
        else:
            num_heads = ch // nhc
            dim_head = nhc
        if legacy:
            dim_head = ch // num_heads
        return num_heads, dim_head

    num_heads = nh_cfg
    input_blocks = [[("conv", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    cm = cfg["channel_mult"]
    for level, mult in enumerate(cm):
        for _ in range(cfg["num_res_blocks"]):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg["attention_resolutions"]:
                num_heads, dim_head = heads_for(ch, num_heads)
                layers.append(("st", ch, num_heads, dim_head))
            input_blocks.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            input_blocks.append([("resdown", ch, ch) if cfg.get("resblock_updown", False) else ("down", ch)])
            chans.append(ch)
            ds *= 2
    num_heads, dim_head = heads_for(ch, num_heads)
    middle = [("res", ch, ch), ("st", ch, num_heads, dim_head), ("res", ch, ch)]
    output_blocks = []
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(cfg["num_res_blocks"] + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg["attention_resolutions"]:
                num_heads, dim_head = heads_for(ch, num_heads)
                layers.append(("st", ch, num_heads, dim_head))
            if level and i == cfg["num_res_blocks"]:
                layers.append(("resup", ch, ch) if cfg.get("resblock_updown", False) else ("up", ch))
                ds //= 2
            output_blocks.append(layers)
    return input_blocks, middle, output_blocks


class UNetOracle:
    """UNetModel.construct (openaimodel.py:536-576) in fp32 on the CPU."""

    def __init__(self, cfg, params):
        self.cfg = dict(cfg)
        self.p = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in params.items()}
        self.input_blocks, self.middle, self.output_blocks = unet_structure(cfg)

    # -- layers ---------------------------------------------------------------
    def _res(self, pre, x, emb, mode=None):
        """ResBlock.construct openaimodel.py:176-205.  mode 'up' / 'down': the resblock_updown form, where nearest-2x /
        2x2 average pooling (Upsample / Downsample with use_conv=False, :33-88) act on BOTH the normalised branch and the
        skip input; use_scale_shift_norm: GroupNorm(h) * (1 + scale) + shift from a 2x-wide emb projection (:193-198)."""
        p = self.p
        h = group_norm(x, p[pre + "in_layers_norm.gamma"], p[pre + "in_layers_norm.beta"], 1e-5)
        h = silu(h)
        if mode == "up":
            h, x = upsample_nearest2x(h), upsample_nearest2x(x)
        elif mode == "down":
            h, x = r16(F.avg_pool2d(h, 2)), r16(F.avg_pool2d(x, 2))
        h = conv2d(h, p[pre + "in_layers_conv.conv.weight"], p[pre + "in_layers_conv.conv.bias"])
        emb_out = dense(silu(emb), p[pre + "emb_layers.1.weight"], p[pre + "emb_layers.1.bias"])
        if self.cfg.get("use_scale_shift_norm", False):
            scale, shift = emb_out[:, :, None, None].chunk(2, dim=1)
            h = group_norm(h, p[pre + "out_layers_norm.gamma"], p[pre + "out_layers_norm.beta"], 1e-5)
            h = r16(r16(h * r16(1 + scale)) + shift)
        else:
            h = r16(h + emb_out[:, :, None, None])
            h = group_norm(h, p[pre + "out_layers_norm.gamma"], p[pre + "out_layers_norm.beta"], 1e-5)
        h = silu(h)
        h = conv2d(h, p[pre + "out_layers_conv.conv.weight"], p[pre + "out_layers_conv.conv.bias"])
        if (pre + "skip_connection.conv.weight") in p:
            x = conv2d(x, p[pre + "skip_connection.conv.weight"], p[pre + "skip_connection.conv.bias"], padding=0)
        return r16(x + h)

    def _attn(self, pre, x, context, heads):
        """CrossAttention.construct attention.py:117-166 (mask branch has no effect)."""
        p = self.p
        q = dense(x, p[pre + "to_q.weight"])
        ctx = x if context is None else context
        k = dense(ctx, p[pre + "to_k.weight"])
        v = dense(ctx, p[pre + "to_v.weight"])
        b, n, c = q.shape
        d = c // heads

        def rin(t):
            bb, nn_, _ = t.shape
            return t.reshape(bb, nn_, heads, d).permute(0, 2, 1, 3).reshape(bb * heads, nn_, d)

        q, k, v = rin(q), rin(k), rin(v)
        # fp16 mode: the [b*h, N, N] scores, their scaling, the softmax and P.V are four fp16 ops (attention.py:138-152)
        sim = r16(r16(torch.matmul(q, k.transpose(1, 2))) * r16(d ** -0.5))
        attn = r16(torch.softmax(sim, dim=-1))
        out = r16(torch.matmul(attn, v))
        out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, heads * d)
        return dense(out, p[pre + "to_out.0.weight"], p[pre + "to_out.0.bias"])

    def _st(self, pre, x, context, heads):
        """SpatialTransformer.construct attention.py:237-256 + BasicTransformerBlock :181-185."""
        p = self.p
        use_linear = self.cfg.get("use_linear_in_transformer", False)
        b, c, h, w = x.shape
        x_in = x
        x = group_norm(x, p[pre + "norm.gamma"], p[pre + "norm.beta"], 1e-6)
        if not use_linear:
            x = conv2d(x, p[pre + "proj_in.weight"], p[pre + "proj_in.bias"], padding=0)
        x = x.reshape(b, c, h * w).permute(0, 2, 1)
        if use_linear:
            x = dense(x, p[pre + "proj_in.weight"], p[pre + "proj_in.bias"])
        for k in range(self.cfg.get("transformer_depth", 1)):      # attention.py:221-224, 248-249
            t = pre + f"transformer_blocks.{k}."
            x = r16(self._attn(t + "attn1.", layer_norm(x, p[t + "norm1.gamma"], p[t + "norm1.beta"], 1e-5), None, heads) + x)
            x = r16(self._attn(t + "attn2.", layer_norm(x, p[t + "norm2.gamma"], p[t + "norm2.beta"], 1e-5), context, heads) + x)
            y = layer_norm(x, p[t + "norm3.gamma"], p[t + "norm3.beta"], 1e-5)
            y = dense(y, p[t + "ff.net.0.proj.weight"], p[t + "ff.net.0.proj.bias"])  # GEGLU attention.py:41-51
            a, gate = y.chunk(2, dim=-1)
            y = r16(a * gelu_tanh(gate))
            y = dense(y, p[t + "ff.net.2.weight"], p[t + "ff.net.2.bias"])
            x = r16(y + x)
        if use_linear:
            x = dense(x, p[pre + "proj_out.weight"], p[pre + "proj_out.bias"])
        x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
        if not use_linear:
            x = conv2d(x, p[pre + "proj_out.weight"], p[pre + "proj_out.bias"], padding=0)
        return r16(x + x_in)

    def _layer(self, pre, layer, h, emb, context):
        p = self.p
        kind = layer[0]
        if kind == "conv":
            return conv2d(h, p[pre + "conv.weight"], p[pre + "conv.bias"])
        if kind == "res":
            return self._res(pre, h, emb)
        if kind == "st":
            return self._st(pre, h, context, layer[2])
        if kind in ("resdown", "resup"):
            return self._res(pre, h, emb, mode=kind[3:])
        conv_resample = self.cfg.get("conv_resample", True)
        if kind == "down":  # Downsample openaimodel.py:63-88
            if not conv_resample:
                return r16(F.avg_pool2d(h, 2))
            return conv2d(h, p[pre + "op.conv.weight"], p[pre + "op.conv.bias"], stride=2, padding=1)
        if kind == "up":  # Upsample openaimodel.py:33-60
            if not conv_resample:
                return upsample_nearest2x(h)
            return conv2d(upsample_nearest2x(h), p[pre + "conv.conv.weight"], p[pre + "conv.conv.bias"])
        raise ValueError(kind)

    # -- forward --------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, timesteps, context=None, y=None):
        p = self.p
        assert (y is not None) == (self.cfg.get("num_classes") is not None), \
            "must specify y if and only if the model is class-conditional"      # openaimodel.py:545-547
        x = r16(torch.as_tensor(x, dtype=torch.float32))              # apply_model casts x_noisy / cond (ddpm.py:291-292)
        if context is not None:   # None: attn2 attends to its own input (attention.py:133 `context = default(context, x)`)
            context = r16(torch.as_tensor(context, dtype=torch.float32))
        timesteps = torch.as_tensor(timesteps)
        t_emb = timestep_embedding(timesteps, self.cfg["model_channels"])
        emb = dense(t_emb, p["time_embed.0.weight"], p["time_embed.0.bias"])
        emb = dense(silu(emb), p["time_embed.2.weight"], p["time_embed.2.bias"])
        if y is not None:       # openaimodel.py:552-554
            emb = r16(emb + r16(p["label_emb.embedding_table"])[torch.as_tensor(np.asarray(y), dtype=torch.long)])
        hs = []
        h = x
        for i, block in enumerate(self.input_blocks):
            for j, layer in enumerate(block):
                h = self._layer(f"input_blocks.{i}.{j}.", layer, h, emb, context)
            hs.append(h)
        for j, layer in enumerate(self.middle):
            h = self._layer(f"middle_block.{j}.", layer, h, emb, context)
        for i, block in enumerate(self.output_blocks):
            h = torch.cat([h, hs.pop()], dim=1)
            for j, layer in enumerate(block):
                h = self._layer(f"output_blocks.{i}.{j}.", layer, h, emb, context)
        if self.cfg.get("n_embed") is not None:      # predict_codebook_ids: GroupNorm -> 1x1 conv, no SiLU (:527-531, 573-574)
            h = group_norm(h, p["id_predictor.0.gamma"], p["id_predictor.0.beta"], 1e-5)
            return conv2d(h, p["id_predictor.1.conv.weight"], p["id_predictor.1.conv.bias"], padding=0)
        h = silu(group_norm(h, p["out.0.gamma"], p["out.0.beta"], 1e-5))
        return conv2d(h, p["out.2.conv.weight"], p["out.2.conv.bias"])

    __call__ = forward


def unet_param_shapes(cfg):
    """Parameter name -> shape for the reference's UNetModel."""
    inb, mid, outb = unet_structure(cfg)
    mc = cfg["model_channels"]
    ted = 4 * mc
    ctx = cfg["context_dim"]
    use_linear = cfg.get("use_linear_in_transformer", False)
    ssn = 2 if cfg.get("use_scale_shift_norm", False) else 1
    conv_resample = cfg.get("conv_resample", True)
    shapes = {
        "time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
        "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,),
    }
    if cfg.get("num_classes") is not None:
        shapes["label_emb.embedding_table"] = (cfg["num_classes"], ted)

    def add_layer(pre, layer):
        kind = layer[0]
        if kind == "conv":
            shapes[pre + "conv.weight"] = (layer[2], layer[1], 3, 3)
            shapes[pre + "conv.bias"] = (layer[2],)
        elif kind in ("res", "resdown", "resup"):
            cin, cout = layer[1], layer[2]
            shapes[pre + "in_layers_norm.gamma"] = (cin,)
            shapes[pre + "in_layers_norm.beta"] = (cin,)
            shapes[pre + "in_layers_conv.conv.weight"] = (cout, cin, 3, 3)
            shapes[pre + "in_layers_conv.conv.bias"] = (cout,)
            shapes[pre + "emb_layers.1.weight"] = (ssn * cout, ted)
            shapes[pre + "emb_layers.1.bias"] = (ssn * cout,)
            shapes[pre + "out_layers_norm.gamma"] = (cout,)
            shapes[pre + "out_layers_norm.beta"] = (cout,)
            shapes[pre + "out_layers_conv.conv.weight"] = (cout, cout, 3, 3)
            shapes[pre + "out_layers_conv.conv.bias"] = (cout,)
            if cin != cout:
                shapes[pre + "skip_connection.conv.weight"] = (cout, cin, 1, 1)
                shapes[pre + "skip_connection.conv.bias"] = (cout,)
        elif kind == "st":
            ch, heads, dh = layer[1], layer[2], layer[3]
            inner = heads * dh
            shapes[pre + "norm.gamma"] = (ch,)
            shapes[pre + "norm.beta"] = (ch,)
            if use_linear:
                shapes[pre + "proj_in.weight"] = (inner, ch)
                shapes[pre + "proj_out.weight"] = (ch, inner)
            else:
                shapes[pre + "proj_in.weight"] = (inner, ch, 1, 1)
                shapes[pre + "proj_out.weight"] = (ch, inner, 1, 1)
            shapes[pre + "proj_in.bias"] = (inner,)
            shapes[pre + "proj_out.bias"] = (ch,)
            for k in range(cfg.get("transformer_depth", 1)):
                t = pre + f"transformer_blocks.{k}."
                for a, cd in (("attn1.", inner), ("attn2.", ctx)):
                    shapes[t + a + "to_q.weight"] = (inner, inner)
                    shapes[t + a + "to_k.weight"] = (inner, cd)
                    shapes[t + a + "to_v.weight"] = (inner, cd)
                    shapes[t + a + "to_out.0.weight"] = (inner, inner)
                    shapes[t + a + "to_out.0.bias"] = (inner,)
                shapes[t + "ff.net.0.proj.weight"] = (inner * 8, inner)
                shapes[t + "ff.net.0.proj.bias"] = (inner * 8,)
                shapes[t + "ff.net.2.weight"] = (inner, inner * 4)
                shapes[t + "ff.net.2.bias"] = (inner,)
                for n in ("norm1", "norm2", "norm3"):
                    shapes[t + n + ".gamma"] = (inner,)
                    shapes[t + n + ".beta"] = (inner,)
        elif kind == "down" and conv_resample:        # Downsample(use_conv=False) is a parameter-free 2x2 average pool
            shapes[pre + "op.conv.weight"] = (layer[1], layer[1], 3, 3)
            shapes[pre + "op.conv.bias"] = (layer[1],)
        elif kind == "up" and conv_resample:
            shapes[pre + "conv.conv.weight"] = (layer[1], layer[1], 3, 3)
            shapes[pre + "conv.conv.bias"] = (layer[1],)

    for i, block in enumerate(inb):
        for j, layer in enumerate(block):
            add_layer(f"input_blocks.{i}.{j}.", layer)
    for j, layer in enumerate(mid):
        add_layer(f"middle_block.{j}.", layer)
    for i, block in enumerate(outb):
        for j, layer in enumerate(block):
            add_layer(f"output_blocks.{i}.{j}.", layer)
    shapes["out.0.gamma"] = (mc,)
    shapes["out.0.beta"] = (mc,)
    shapes["out.2.conv.weight"] = (cfg["out_channels"], mc, 3, 3)
    shapes["out.2.conv.bias"] = (cfg["out_channels"],)
    if cfg.get("n_embed") is not None:
        shapes["id_predictor.0.gamma"] = (mc,)
        shapes["id_predictor.0.beta"] = (mc,)
        shapes["id_predictor.1.conv.weight"] = (cfg["n_embed"], mc, 1, 1)
        shapes["id_predictor.1.conv.bias"] = (cfg["n_embed"],)
    return shapes


def init_params(cfg, seed=0, zero_init=False, dtype=np.float32):
    """Seeded synthetic weights: fan-in scaled normals so activations stay O(1).

    zero_init=True reproduces the reference constructor's ``zero_module`` layers
    (openaimodel.py:162-165,524; attention.py:223-231) for the structural KAT.
    """
    rng = np.random.RandomState(seed)
    params = {}
    for name, shape in unet_param_shapes(cfg).items():
        if name.endswith(".gamma"):
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif name.endswith(".beta"):
            v = 0.1 * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            v = 0.05 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape) / math.sqrt(fan_in)
        if zero_init and (name.endswith("out_layers_conv.conv.weight") or name.endswith("out_layers_conv.conv.bias")
                          or name.startswith("out.2.") or ".proj_out." in name):
            v = np.zeros(shape)
        params[name] = v.astype(dtype)
    return params


# ----------------------------------------------------------------------------- samplers
class ModelOracle:
    """The attributes/methods PLMSSampler needs from LatentDiffusion (plms.py:31,40-46; ddpm.py:290-306)."""

    def __init__(self, unet, linear_start=0.00085, linear_end=0.0120, timesteps=1000, conditioning_key="auto"):
        """conditioning_key: one of DiffusionWrapper's (WK ddpm.py:358), or "auto" = what the two shipped model classes use --
        'hybrid' when the conditioning is a dict with c_concat (LatentInpaintDiffusion), 'crossattn' otherwise."""
        self.unet = unet
        assert conditioning_key in [None, "concat", "crossattn", "hybrid", "adm", "auto"]
        self.conditioning_key = conditioning_key
        s = register_schedule(linear_start, linear_end, timesteps)
        self.num_timesteps = s["num_timesteps"]
        self.betas = s["betas"]
        self.alphas_cumprod = s["alphas_cumprod"]
        self.alphas_cumprod_prev = s["alphas_cumprod_prev"]
        self.sqrt_alphas_cumprod = s["sqrt_alphas_cumprod"]
        self.sqrt_one_minus_alphas_cumprod = s["sqrt_one_minus_alphas_cumprod"]
        self.parameterization = "eps"
        self.calls = 0

    def apply_model(self, x, t, cond=None):
        """ddpm.py:290-306: a bare conditioning becomes {'c_concat': cond} for key 'concat' and {'c_crossattn': cond} otherwise
        (:299-300), a dict (hybrid, inpaint.py:84-88) passes through; then DiffusionWrapper.construct, WK ddpm.py:360-377, by key:
        None -> unet(x, t); 'concat' -> unet(cat(x, c_concat), t); 'crossattn' -> unet(x, t, context); 'hybrid' -> both;
        'adm' -> unet(x, t, y=c_crossattn)."""
        self.calls += 1
        key = self.conditioning_key
        if key == "auto":
            key = "hybrid" if isinstance(cond, dict) and cond.get("c_concat") is not None else "crossattn"
        if not isinstance(cond, dict):
            cond = {"c_concat" if key == "concat" else "c_crossattn": cond}
        c_concat, c_crossattn = cond.get("c_concat"), cond.get("c_crossattn")
        if key is None:
            return self.unet(x, t)
        if key == "concat":
            return self.unet(torch.cat([x, torch.as_tensor(c_concat, dtype=torch.float32)], 1), t)
        if key == "crossattn":
            return self.unet(x, t, c_crossattn)
        if key == "hybrid":
            return self.unet(torch.cat([x, torch.as_tensor(c_concat, dtype=torch.float32)], 1), t, c_crossattn)
        return self.unet(x, t, y=c_crossattn)      # 'adm'

    def q_sample(self, x0, t, noise):
        """ddpm.py:197-200."""
        a = r16(torch.as_tensor(self.sqrt_alphas_cumprod))[t].reshape(-1, 1, 1, 1)
        b = r16(torch.as_tensor(self.sqrt_one_minus_alphas_cumprod))[t].reshape(-1, 1, 1, 1)
        return a * x0 + b * noise


def sample(model, S, batch_size, shape, conditioning, x_T, sampler="plms", eta=0.0,
           unconditional_guidance_scale=1.0, unconditional_conditioning=None, noise_fn=None,
           temperature=1.0, log_every_t=100, mask=None, x0=None, blend_noises=None, timesteps=None,
           ddim_use_original_steps=False, noise_dropout=0.0, dropout_masks=None, score_corrector=None,
           corrector_kwargs=None, quantize_x0=False):
    """PLMSSampler.sample/plms_sampling/p_sample_plms (plms.py:69-247).

    sampler='ddim' applies get_x_prev_and_pred_x0 (plms.py:210-228) with e'=e_t each step
    (SURVEY 0.4): S model calls instead of S+1, eta may be != 0 (noise from noise_fn(shape)).
    timesteps / ddim_use_original_steps: plms.py:134-142, 205-208 (a prefix of the DDIM grid, or every DDPM step with the
    model's own alphas_cumprod tables).  noise_dropout: plms.py:224-225 `ops.dropout(noise, p)` = zero with probability p,
    survivors scaled by 1/(1-p); dropout_masks[k] (1 = keep) is the k-th draw.
    score_corrector / corrector_kwargs: plms.py:199-201 (every model output goes through
    `score_corrector.modify_score(model, e_t, x, t, c, **corrector_kwargs)`); quantize_x0: plms.py:218-219
    (`pred_x0, _, *_ = model.first_stage_model.quantize(pred_x0)`).
    Returns (samples, intermediates).
    """
    if sampler == "plms" and eta != 0:
        raise ValueError("ddim_eta must be 0 for PLMS")  # plms.py:35-36
    ts = make_ddim_timesteps(S, model.num_timesteps)
    # fp16 mode: the schedule tables are tensors of the MODEL dtype (ddpm.py:129-139 `to_mindspore = partial(ms.Tensor,
    # dtype=self.dtype)`), so everything the sampler derives from them starts from fp16-rounded alphas_cumprod and is itself
    # an fp16 op result (plms.py:47-67); `ms.numpy.full` then widens the selected scalars to fp32 (plms.py:212-215)
    ac, acp = r16(np.asarray(model.alphas_cumprod, np.float32)), r16(np.asarray(model.alphas_cumprod_prev, np.float32))
    sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(ac, ts, eta)
    sigmas = r16(sigmas)
    sqrt_one_minus_alphas = r16(np.sqrt(r16(1.0 - alphas)).astype(np.float32))
    if ddim_use_original_steps:   # plms.py:134-135, 141-142, 205-208; sigmas: plms.py:64-67
        n_orig = model.num_timesteps if timesteps is None else int(timesteps)
        ts = np.arange(n_orig, dtype=np.int64)
        alphas, alphas_prev = ac, acp
        sqrt_one_minus_alphas = r16(np.sqrt(r16(1.0 - ac)).astype(np.float32))
        sigmas = r16((np.float32(eta) * np.sqrt((1 - acp) / (1 - ac) * (1 - ac / acp))).astype(np.float32))
    elif timesteps is not None:   # plms.py:137-139
        subset_end = int(min(timesteps / ts.shape[0], 1) * ts.shape[0]) - 1
        ts = ts[:subset_end]
    b = batch_size
    img = torch.as_tensor(x_T, dtype=torch.float32)
    assert tuple(img.shape) == (b,) + tuple(shape)
    # dict conditioning = hybrid (WK inpaint.py:84-88): {"c_concat": [B,5,h,w], "c_crossattn": [B,T,D]}; the uncond dict
    # carries the SAME c_concat (inpaint.py:87-88), WK plms.py:188-205 concatenates key by key
    c_cat = uc_cat = None
    if isinstance(conditioning, dict):
        c_cat = torch.as_tensor(conditioning["c_concat"], dtype=torch.float32)
        conditioning = conditioning["c_crossattn"]
        if isinstance(unconditional_conditioning, dict):
            # WK plms.py:191-201: [uncond[k]; cond[k]] for EVERY key -- the unconditional c_concat may differ from the conditional one
            if unconditional_conditioning.get("c_concat") is not None:
                uc_cat = torch.as_tensor(unconditional_conditioning["c_concat"], dtype=torch.float32)
            unconditional_conditioning = unconditional_conditioning["c_crossattn"]
    def as_cond(c):     # text context / extra input channels: fp32; class labels ('adm'): integers stay integers
        if c is None:
            return None
        c = torch.as_tensor(c)
        return c.to(torch.float32) if c.is_floating_point() else c
    cond, uc = as_cond(conditioning), as_cond(unconditional_conditioning)
    if mask is not None:
        mask = torch.as_tensor(mask, dtype=torch.float32)
        x0 = torch.as_tensor(x0, dtype=torch.float32)

    def wrap(c, n):
        if c_cat is None:
            return c
        parts = [c_cat] * n if (n == 1 or uc_cat is None) else [uc_cat, c_cat]      # batch = [uncond ; cond]
        return {"c_concat": torch.cat(parts, 0), "c_crossattn": c}
    time_range = np.flip(ts)
    total = len(ts)
    intermediates = {"x_inter": [img], "pred_x0": [img]}
    old_eps = []
    scale = float(unconditional_guidance_scale)

    def get_model_output(x, t):  # plms.py:188-203
        if uc is None or scale == 1.0:
            e_t = model.apply_model(x, t, wrap(cond, 1))
        else:
            x_in = torch.cat([x, x], 0)
            t_in = torch.cat([t, t], 0)
            c_in = wrap(torch.cat([uc, cond], 0), 2)
            e_u, e_c = model.apply_model(x_in, t_in, c_in).chunk(2, dim=0)
            e_t = r16(e_u + r16(scale * r16(e_c - e_u)))     # fp16 mode: eps is an fp16 tensor, each op rounds (plms.py:197)
        if score_corrector is not None:
            assert getattr(model, "parameterization", "eps") == "eps"
            e_t = score_corrector.modify_score(model, e_t, x, t, cond, **(corrector_kwargs or {}))
        return e_t

    drops = [0]

    def x_prev_and_pred_x0(x, e_t, index):  # plms.py:210-228
        a_t = torch.tensor(alphas[index])
        a_prev = torch.tensor(alphas_prev[index])
        sigma_t = torch.tensor(sigmas[index])
        s1m = torch.tensor(sqrt_one_minus_alphas[index])
        pred_x0 = (x - s1m * e_t) / a_t.sqrt()
        if quantize_x0:
            pred_x0, _, *_ = model.first_stage_model.quantize(pred_x0)
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
        if float(sigma_t) != 0.0:
            noise = sigma_t * torch.as_tensor(noise_fn(tuple(x.shape)), dtype=torch.float32) * temperature
            if noise_dropout > 0.0:   # plms.py:224-225
                keep = torch.as_tensor(dropout_masks[drops[0]], dtype=torch.float32)
                drops[0] += 1
                noise = noise * keep / (1.0 - noise_dropout)
        else:
            noise = 0.0
        return a_prev.sqrt() * pred_x0 + dir_xt + noise, pred_x0

    for i, step in enumerate(time_range):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.int64)
        t_next = torch.full((b,), int(time_range[min(i + 1, total - 1)]), dtype=torch.int64)
        if mask is not None:   # plms.py:153-157 (WK: q_sample(x0, ts, randn)); blend_noises[i] injects the draw
            img_orig = model.q_sample(x0, t, torch.as_tensor(blend_noises[i], dtype=torch.float32))
            img = img_orig * mask + (1.0 - mask) * img
        e_t = get_model_output(img, t)
        if sampler == "ddim":
            e_prime = e_t
        elif len(old_eps) == 0:
            x_prev, _ = x_prev_and_pred_x0(img, e_t, index)
            e_next = get_model_output(x_prev, t_next)
            e_prime = r16(r16(e_t + e_next) / 2)
        elif len(old_eps) == 1:
            e_prime = r16(r16(r16(3 * e_t) - old_eps[-1]) / 2)
        elif len(old_eps) == 2:
            e_prime = r16(r16(r16(r16(23 * e_t) - r16(16 * old_eps[-1])) + r16(5 * old_eps[-2])) / 12)
        else:
            e_prime = r16(r16(r16(r16(r16(55 * e_t) - r16(59 * old_eps[-1])) + r16(37 * old_eps[-2]))
                              - r16(9 * old_eps[-3])) / 24)
        img, pred_x0 = x_prev_and_pred_x0(img, e_prime, index)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        if index % log_every_t == 0 or index == total - 1:
            intermediates["x_inter"].append(img)
            intermediates["pred_x0"].append(pred_x0)
    return img, intermediates
