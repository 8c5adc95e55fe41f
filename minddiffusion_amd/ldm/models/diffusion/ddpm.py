"""LatentDiffusion -- the inference half of the reference's ldm/models/diffusion/ddpm.py
(DDPM.register_schedule :111-161, LatentDiffusion.apply_model :290-306, q_sample :197-200,
DiffusionWrapper :346-357; Wukong keyword form WK/.../ddpm.py:276-278, 360-377).

Only what the samplers and txt2img.py touch is mirrored: the schedule attributes, ``apply_model``,
``q_sample``, ``get_learned_conditioning`` / ``decode_first_stage`` (delegating to injected
text-encoder / VAE objects -- both are outside the hot path).
Training (`construct` / `p_losses`) is out of scope.
"""
import numpy as np
import torch

from ...._lib import MdxError
from .... import ops  # noqa: F401  (loads the HIP library binding early)
from ...util import instantiate_from_config
from ...modules.diffusionmodules.util import make_beta_schedule


class DiffusionWrapper:
    """ddpm.py:346-357 (SD2) / WK ddpm.py:354-377: all five conditioning keys.  None / 'concat' / 'adm' call the UNet without a
    context -- attn2 then attends to its own input (attention.py:133) -- which the reference's UNet, like this one, only accepts
    when context_dim equals the transformer width at every attention level."""

    def __init__(self, diff_model_config, conditioning_key):
        self.diffusion_model = (instantiate_from_config(diff_model_config)
                                if isinstance(diff_model_config, dict) else diff_model_config)
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, "concat", "crossattn", "hybrid", "adm"]

    def construct(self, x, t, c_concat=None, c_crossattn=None):
        key = self.conditioning_key
        if key in ("concat", "hybrid"):                             # WK ddpm.py:363-365, 368-371
            if c_concat is None:
                raise MdxError(f"{key!r} conditioning needs c_concat")
            x = torch.cat([x, _first(c_concat).to(device=x.device, dtype=x.dtype)], 1)
        if key is None or key == "concat":                          # WK ddpm.py:361-365
            return self.diffusion_model(x, t)
        if key == "adm":                                            # WK ddpm.py:372-374: class labels ride in c_crossattn
            if c_crossattn is None:
                raise MdxError("'adm' conditioning needs the class labels in c_crossattn")
            return self.diffusion_model(x, t, y=_first(c_crossattn))
        if c_crossattn is None:
            raise MdxError(f"{key!r} conditioning needs c_crossattn")
        return self.diffusion_model(x, t, context=_first(c_crossattn))     # WK ddpm.py:366-371

    __call__ = construct


def _first(c):
    """The reference wraps conditionings in lists (ddpm.py:299-304); unwrap a single tensor."""
    while isinstance(c, (list, tuple)):
        c = c[0]
    return c


class LatentDiffusion:
    def __init__(self, unet_config, linear_start=1e-4, linear_end=2e-2, timesteps=1000, beta_schedule="linear",
                 conditioning_key="crossattn", scale_factor=1.0, parameterization="eps", use_fp16=False,
                 first_stage_config=None, cond_stage_config=None, image_size=64, channels=4, v_posterior=0.0,
                 **unused):
        assert parameterization in ["eps", "x0"], 'currently only supporting "eps" and "x0"'
        self.parameterization = parameterization
        self.image_size = image_size
        self.channels = channels
        self.scale_factor = scale_factor
        self.v_posterior = v_posterior
        self.use_fp16 = use_fp16
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.first_stage_model = None   # inject a VAE object with .decode(z) to use decode_first_stage
        self.cond_stage_model = None    # inject a text encoder with .encode(list[str]) for get_learned_conditioning
        self.register_schedule(beta_schedule=beta_schedule, timesteps=timesteps, linear_start=linear_start,
                               linear_end=linear_end)

    # ---- ddpm.py:111-161 (host numpy; the reference stores these as tensors of the model dtype, we keep fp32)
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        f = lambda a: np.asarray(a, dtype=np.float32)
        self.betas = f(betas)
        self.alphas_cumprod = f(alphas_cumprod)
        self.alphas_cumprod_prev = f(alphas_cumprod_prev)
        self.sqrt_alphas_cumprod = f(np.sqrt(alphas_cumprod))
        self.sqrt_one_minus_alphas_cumprod = f(np.sqrt(1. - alphas_cumprod))
        self.log_one_minus_alphas_cumprod = f(np.log(1. - alphas_cumprod))
        self.sqrt_recip_alphas_cumprod = f(np.sqrt(1. / alphas_cumprod))
        self.sqrt_recipm1_alphas_cumprod = f(np.sqrt(1. / alphas_cumprod - 1))

    @property
    def unet(self):
        return self.model.diffusion_model

    # ---- ddpm.py:290-306 (positional cond) and WK ddpm.py:276-278 (keywords); dict = hybrid conditioning (inpaint.py:84)
    def _split_cond(self, cond, c_concat=None, c_crossattn=None):
        key = self.model.conditioning_key
        if isinstance(cond, dict):
            c_concat = cond.get("c_concat", c_concat)
            c_crossattn = cond.get("c_crossattn", c_crossattn)
        elif cond is not None:
            # ddpm.py:299-300: a bare conditioning goes to the keyword the wrapper's key reads
            if key == "concat":
                c_concat = cond
            else:
                c_crossattn = cond
        c_crossattn, c_concat = _first(c_crossattn), _first(c_concat)
        if key in ("crossattn", "hybrid", "adm") and c_crossattn is None:
            raise MdxError(f"apply_model: conditioning_key={key!r} needs "
                           f"{'the class labels' if key == 'adm' else 'a cross-attention conditioning tensor'}")
        if (c_concat is not None) != (key in ("hybrid", "concat")):
            raise MdxError(f"apply_model: conditioning_key={key!r} "
                           f"{'needs' if c_concat is None else 'does not take'} c_concat")
        return c_concat, c_crossattn

    def apply_model(self, x_noisy, t, cond=None, return_ids=False, c_concat=None, c_crossattn=None):
        c_concat, c_crossattn = self._split_cond(cond, c_concat, c_crossattn)
        return self.model(x_noisy, t, c_concat=c_concat, c_crossattn=c_crossattn)

    def apply_model_nhwc(self, x_noisy, t, cond, temb=None, cfg_dup=False):
        """Fast path used by the samplers: returns the UNet's static NHWC fp16 eps buffer [B, H*W, 8]
        (valid until the next call) so the fused sampler-step kernel can consume it without a layout pass.
        `x_noisy` already carries the c_concat channels for hybrid / concat conditioning (the sampler writes them once);
        `cond` is the text context ('crossattn' / 'hybrid'), the class labels ('adm'), or ignored (None / 'concat').
        temb: the row of time_embedding_table() that belongs to `t` (optional).  cfg_dup: the batch is [uncond ; cond] of the same
        x_noisy and t (UNetModel.forward_nhwc)."""
        key = self.model.conditioning_key
        if key == "adm":
            return self.unet.forward_nhwc(x_noisy, t, None, y=_first(cond))
        if key is None or key == "concat":
            return self.unet.forward_nhwc(x_noisy, t, None, temb=temb)
        return self.unet.forward_nhwc(x_noisy, t, cond, temb=temb, cfg_dup=cfg_dup)

    def time_embedding_table(self, t):
        """UNetModel.time_embedding_table: the timestep-only part of the UNet for all steps of a run, batched."""
        return self.unet.time_embedding_table(t)

    # ---- ddpm.py:197-200
    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            raise MdxError("q_sample: pass the noise explicitly (the reference's SD2 call site omits it and fails)")
        a = torch.as_tensor(self.sqrt_alphas_cumprod, device=x_start.device)[t].reshape(-1, 1, 1, 1)
        b = torch.as_tensor(self.sqrt_one_minus_alphas_cumprod, device=x_start.device)[t].reshape(-1, 1, 1, 1)
        return a * x_start + b * noise

    # ---- ddpm.py:274-288: outside the hot path, delegated to injected objects
    def get_learned_conditioning(self, c):
        if self.cond_stage_model is None:
            raise MdxError("no text encoder attached (cond_stage_model is out of the hot-path scope): "
                           "pass precomputed conditioning tensors [B,77,context_dim] instead")
        return self.cond_stage_model.encode(c)

    def encode_first_stage(self, x):                      # WK ddpm.py:270-271
        if self.first_stage_model is None:
            raise MdxError("no VAE attached (first_stage_model)")
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, z):                # WK ddpm.py:273-274
        return self.scale_factor * z

    def decode_first_stage(self, z, predict_cids=False):
        if self.first_stage_model is None:
            raise MdxError("no VAE attached (first_stage_model is out of the hot-path scope)")
        return self.first_stage_model.decode(1. / self.scale_factor * z)


class LatentInpaintDiffusion(LatentDiffusion):
    """WK ldm/models/diffusion/ddpm.py:339-352: hybrid conditioning -- the UNet sees cat(x, mask, masked-image latent)
    (9 channels, configs/wukong-huahua_inpaint_inference.yaml) plus the text cross-attention context."""

    def __init__(self, concat_keys=("mask", "masked_image"), masked_image_key="masked_image", finetune_keys=None,
                 *args, **kwargs):
        kwargs.setdefault("conditioning_key", "hybrid")
        super().__init__(*args, **kwargs)
        self.masked_image_key = masked_image_key
        assert self.masked_image_key in concat_keys
        self.concat_keys = concat_keys
