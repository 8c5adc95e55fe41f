"""AutoencoderKL -- mirror of the reference's ldm/models/autoencoder.py (:22-78): ``decode(z)`` =
``decoder(post_quant_conv(z))`` and ``encode(x)`` = a sample of the diagonal Gaussian given by
``quant_conv(encoder(x))`` (the inpainting / img2img pre-processing, wukong-huahua/inpaint.py:79-81).

Attach to ``LatentDiffusion.first_stage_model`` and ``decode_first_stage`` / ``DiffusionPipeline`` produce images:

    vae = AutoencoderKL(ddconfig=SD_VAE_DDCONFIG, embed_dim=4)
    vae.load_state_dict(params)            # reference names: post_quant_conv.*, decoder.*
    model.first_stage_model = vae
"""
import torch

from ... import ops
from ..modules.diffusionmodules.model import Decoder, Encoder


class AutoencoderKL:
    def __init__(self, ddconfig, embed_dim, ckpt_path=None, ignore_keys=(), image_key="image", colorize_nlabels=None,
                 monitor=None, use_fp16=False, device=None, use_graph=True):
        assert ddconfig["double_z"]
        self.embed_dim = embed_dim
        self.ddconfig = dict(ddconfig)
        self.decoder = Decoder(device=device, use_graph=use_graph, **ddconfig)
        self.encoder = Encoder(device=device, use_graph=use_graph, **ddconfig)
        self.generator = None          # torch.Generator for encode()'s noise (None: the default generator)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=()):
        """autoencoder.py:44-54: read a MindSpore .ckpt (minddiffusion_amd/ms_checkpoint.py), drop keys that start with
        one of `ignore_keys`, load.  Accepts a bare VAE checkpoint or a whole LatentDiffusion one (first_stage_model.*)."""
        from ...ms_checkpoint import VAE_PREFIX, load_checkpoint
        sd = load_checkpoint(path)
        if any(k.startswith(VAE_PREFIX) for k in sd):
            sd = {k[len(VAE_PREFIX):]: v for k, v in sd.items() if k.startswith(VAE_PREFIX)}
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
        # Keys this model does not own (loss.*, discriminator.* of a training checkpoint ...) are reported, not fatal -- the
        # reference's load_param_into_net ignores them and the CLI prints the list (autoencoder.py:44-54); MISSING keys of a
        # checkpoint that carries an encoder still raise.
        own = self.parameter_shapes()
        self.unexpected_keys = sorted(k for k in sd if k not in own)
        if self.unexpected_keys:
            import warnings
            warnings.warn(f"AutoencoderKL.init_from_ckpt: {len(self.unexpected_keys)} checkpoint keys are not parameters of this "
                          f"model and were ignored: {self.unexpected_keys[:8]}{' ...' if len(self.unexpected_keys) > 8 else ''}")
        sd = {k: v for k, v in sd.items() if k in own}
        self.load_state_dict(sd, strict="encoder.conv_in.weight" in sd)
        return self

    def parameter_shapes(self):
        zc = self.ddconfig["z_channels"]
        s = {"post_quant_conv.weight": (zc, self.embed_dim, 1, 1), "post_quant_conv.bias": (zc,)}
        s.update(self.decoder.parameter_shapes("decoder."))
        s["quant_conv.weight"] = (2 * self.embed_dim, 2 * zc, 1, 1)
        s["quant_conv.bias"] = (2 * self.embed_dim,)
        s.update(self.encoder.parameter_shapes("encoder."))
        return s

    def load_state_dict(self, params, strict=True):
        """Reference names: post_quant_conv.*, decoder.*, quant_conv.*, encoder.*.  A decode-only checkpoint (no encoder.*)
        loads with strict=False; encode() then raises.  strict also rejects keys this model does not own."""
        from ...weights import check_state_dict
        if strict:
            check_state_dict(self.parameter_shapes(), params, True, "AutoencoderKL.load_state_dict")
        self.decoder.load_state_dict(params, prefix="decoder.",
                                     post_quant=(params["post_quant_conv.weight"], params["post_quant_conv.bias"]),
                                     strict=strict)
        if "encoder.conv_in.weight" in params:
            self.encoder.load_state_dict(params, prefix="encoder.",
                                         quant=(params["quant_conv.weight"], params["quant_conv.bias"]), strict=strict)
        elif strict:
            from ..._lib import MdxError
            raise MdxError("AutoencoderKL.load_state_dict: encoder.* parameters missing (pass strict=False for decode only)")

    def decode(self, z):
        """autoencoder.py:65-68.  Returns a fresh tensor (the decoder's own output buffer is reused by the next call)."""
        return self.decoder(z).clone()

    def encode(self, x, noise=None, sample=True):
        """autoencoder.py:70-78: mean, logvar = split(quant_conv(encoder(x))); logvar clipped to [-30, 20];
        returns mean + exp(0.5 logvar) * N(0, 1)  ([B, embed_dim, H/8, W/8] fp32).  `noise` injects the draw (tests);
        sample=False returns the mode."""
        mom = self.encoder(x)
        B = x.shape[0]
        h, w = self.encoder._plans[(B, x.shape[2], x.shape[3])].out_hw
        zc = self.embed_dim
        out = torch.empty((B, zc, h, w), dtype=torch.float32, device=mom.device)
        if sample and noise is None:
            noise = torch.randn(out.shape, device=mom.device, dtype=torch.float32, generator=self.generator)
        if noise is not None:
            noise = noise.to(device=mom.device, dtype=torch.float32).contiguous()
        return ops.vae_gaussian_sample(mom, zc, noise if sample else None, out)
