"""DiffusionPipeline -- what the reference's txt2img.py main loop does around the sampler
(vision/stablediffusionv2/txt2img.py:242-268; Wukong: wukong-huahua/txt2img.py:255-281):

    uc = model.get_learned_conditioning(B * [""]); c = model.get_learned_conditioning(prompts)
    shape = [4, H // 8, W // 8]
    samples, _ = sampler.sample(S=steps, conditioning=c, batch_size=B, shape=shape, verbose=False,
                                unconditional_guidance_scale=scale, unconditional_conditioning=uc, eta=eta, x_T=x_T)
    x = model.decode_first_stage(samples)

Text encoding and VAE decode are outside the hot path: pass precomputed
conditioning tensors (c, uc), or attach ``model.cond_stage_model`` / ``model.first_stage_model``.
With torch.distributed initialised the global batch is sharded across ranks (distributed.py).
"""
import numpy as np
import torch

from . import distributed as D
from ._lib import MdxError
from .ldm.models.diffusion.ddim import DDIMSampler
from .ldm.models.diffusion.plms import PLMSSampler


class DiffusionPipeline:
    def __init__(self, model, sampler="ddim", device=None):
        self.model = model
        self.device = torch.device(device) if device is not None else model.unet.device
        if isinstance(sampler, str):
            if sampler not in ("ddim", "plms", "dpm_solver"):
                raise ValueError("sampler must be 'ddim', 'plms' or 'dpm_solver'")
            if sampler == "dpm_solver":
                from .ldm.models.diffusion.dpm_solver import DPMSolverSampler
                sampler = DPMSolverSampler(model)                      # txt2img.py --dpm_solver
            else:
                sampler = (DDIMSampler if sampler == "ddim" else PLMSSampler)(model)
        self.sampler = sampler

    def start_noise(self, batch, shape, seed):
        """numpy RandomState(seed).randn -- the reference's own practice for reproducible x_T
        (wukong-huahua/inpaint.py:68-70); MindSpore's StandardNormal stream is not reproducible."""
        return torch.from_numpy(np.random.RandomState(seed).randn(batch, *shape).astype(np.float32))

    def __call__(self, prompts=None, c=None, uc=None, H=512, W=512, steps=50, scale=9.0, eta=0.0, x_T=None, seed=42,
                 decode=False, gather=False, callback=None, img_callback=None, batch_size=None, per_sample_uc=False):
        """Multi-rank runs: every rank calls with the same `prompts` list (or the same `batch_size` when rank 0 passes
        precomputed (c, uc) tensors); only rank 0's c / uc / x_T are used, the other ranks may pass None."""
        rank, n = D.world()
        shape = [4, H // 8, W // 8]                                   # txt2img.py:253
        if prompts is not None and rank == 0:
            uc = self.model.get_learned_conditioning(len(prompts) * [""])   # txt2img.py:246-248
            c = self.model.get_learned_conditioning(list(prompts))            # txt2img.py:251
        if n > 1:
            B = len(prompts) if prompts is not None else (int(c.shape[0]) if c is not None else batch_size)
            if B is None:
                raise MdxError("DiffusionPipeline: ranks without the conditioning tensor need prompts= or batch_size= "
                               "(the global batch) to size the broadcast")
            unet = self.model.unet
            if rank == 0 and x_T is None:
                x_T = self.start_noise(B, shape, seed)
            to_dev = lambda t: None if t is None else t.to(self.device)
            # exactly ONE collective: T, the uc form and the noise flag ride in the payload header (distributed.py)
            c, uc, x_T = D.broadcast_conditioning(
                to_dev(c) if rank == 0 else None, to_dev(uc) if rank == 0 else None, to_dev(x_T) if rank == 0 else None,
                B, (int(getattr(unet, "max_context_len", 80)), int(unet.context_dim)), shape, self.device,
                per_sample_uc=per_sample_uc)
        else:
            if c is None:
                raise MdxError("DiffusionPipeline: pass prompts (with a text encoder attached) or (c, uc) tensors")
            B = int(c.shape[0])
            if x_T is None:
                x_T = self.start_noise(B, shape, seed)
            c = c.to(self.device, torch.float16)
            uc = None if uc is None else uc.to(self.device, torch.float16)
            if uc is not None and uc.shape[0] == 1 and B > 1:
                uc = uc.expand(B, -1, -1).contiguous()
            x_T = x_T.to(self.device)
        local_b = int(c.shape[0])
        samples, inter = self.sampler.sample(S=steps, conditioning=c, batch_size=local_b, shape=shape, verbose=False,
                                             unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                             eta=eta, x_T=x_T, callback=callback, img_callback=img_callback)
        if decode:
            x = self.model.decode_first_stage(samples)                # txt2img.py:265-266
            samples = torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)
        if gather and n > 1:
            samples = D.gather_latents(samples)
        return samples
