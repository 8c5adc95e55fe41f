"""DPMSolverSampler -- MI355X-native mirror of the reference's ldm/models/diffusion/dpm_solver/sampler.py:20-92
(SURVEY 8(f) item 3).  Same constructor and ``sample(...)`` keywords; as wired by the reference:
``NoiseScheduleVP('discrete', alphas_cumprod)``, classifier-free guidance on a noise-prediction model,
``DPM_Solver(predict_x0=True, thresholding=False).sample(steps=S, skip_type='time_uniform', method='multistep',
order=2, lower_order_final=True)``: S UNet evaluations at FRACTIONAL timesteps (t - 1/1000) * 1000.

Execution: per step one UNet forward (replayed hipGraph, CFG batch [uncond; cond] built once) and ONE launch of the
fused ``mdx_sampler_step_f32`` kernel.  With e = CFG-combined eps, x0 = (x - sigma_s e) / alpha_s the 2M update
    x_next = A x + c0 x0 + c1 x0_prev,   x = alpha_s x0 + sigma_s e
is exactly that kernel's  sqrt_a_prev * pred_x0 + dir_coef * e' + sigma * noise  with
    sqrt_at = alpha_s, sqrt_one_minus_at = sigma_s, sqrt_a_prev = A alpha_s + c0, dir_coef = A sigma_s,
    sigma = c1, noise = x0_prev  -- so no extra elementwise passes and no extra kernel.
Differences from the reference, both on the fp32 side of its fp16 arithmetic: the time grid and the schedule scalars
are float64 on the host (the reference casts the grid to fp16, dpm_solver.py:415), and x stays fp32 (sampler.py:88
casts the start noise to fp16).
"""
import os

import numpy as np
import torch

from ..... import ops
from ....._lib import MdxError
from ..plms import _first_tensor
from .dpm_solver import NoiseScheduleVP, multistep_2m_plan


class DPMSolverSampler:
    def __init__(self, model, **kwargs):
        self.model = model
        self.alphas_cumprod = np.asarray(model.alphas_cumprod, dtype=np.float64)
        self.generator = kwargs.get("generator", None)

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def _eps_nhwc(self, x, t, cond, temb=None, cfg_dup=False):
        if hasattr(self.model, "apply_model_nhwc"):
            kw = {} if temb is None else {"temb": temb}
            if cfg_dup:     # both halves of the batch are the same x and t: only the contexts differ
                kw["cfg_dup"] = True
            return self.model.apply_model_nhwc(x, t, cond, **kw)
        e = self.model.apply_model(x, t, cond).to(torch.float32).contiguous()
        return ops.nchw_to_nhwc(e, (e.shape[1] + 7) // 8 * 8)

    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        if conditioning is None:
            raise MdxError("DPMSolverSampler: conditioning is required (classifier-free guidance on a text-conditional UNet)")
        cond = _first_tensor(conditioning)
        if cond.shape[0] != batch_size:
            print(f"Warning: Got {cond.shape[0]} conditionings but batch-size is {batch_size}")
        if mask is not None or x0 is not None:
            raise NotImplementedError("mask/x0 blending is implemented by PLMSSampler / DDIMSampler (WK inpaint.py uses PLMS); "
                                      "the reference never passes it to DPMSolverSampler")
        if not (isinstance(cond, torch.Tensor) and cond.is_cuda):
            raise MdxError("conditioning must be a CUDA(HIP) tensor [B, T, context_dim]")
        dev = cond.device
        C, H, W = shape
        size = (batch_size, C, H, W)
        if x_T is None:
            img = torch.randn(size, device=dev, dtype=torch.float32, generator=self.generator)
        else:
            img = torch.as_tensor(x_T).to(device=dev, dtype=torch.float32).contiguous().clone()
        uc = _first_tensor(unconditional_conditioning) if unconditional_conditioning is not None else None
        scale = float(unconditional_guidance_scale)
        use_cfg = not (uc is None or scale == 1.)                   # model_wrapper :316-317
        b = batch_size
        if use_cfg:
            c_in = torch.cat([uc.to(cond.dtype), cond], 0).contiguous()   # [uncond; cond] (:320-322), built once
            x_in = torch.empty((2 * b, C, H, W), device=dev, dtype=torch.float32)
        else:
            c_in, x_in = cond.contiguous(), None
        nb = 2 * b if use_cfg else b

        ns = NoiseScheduleVP("discrete", alphas_cumprod=self.alphas_cumprod)
        plan = multistep_2m_plan(ns, S, order=2, lower_order_final=True)
        t_all = torch.tensor([p["t_input"] for p in plan], dtype=torch.float32, device=dev)
        # timestep-only part of the UNet for all S (fractional) timesteps in one batched pass (see PLMSSampler)
        temb_all = None
        if hasattr(self.model, "time_embedding_table") and os.environ.get("MDX_SAMPLER_TEMB_TABLE", "1") != "0":
            temb_all = self.model.time_embedding_table(t_all)
        t_all = t_all[:, None].expand(S, nb).contiguous()
        x0_bufs = [torch.empty_like(img), torch.empty_like(img)]    # data predictions at the last two grid points
        x_next = torch.empty_like(img)
        for k, p in enumerate(plan):
            if use_cfg:
                x_in[:b].copy_(img)
                x_in[b:].copy_(img)
                eps = self._eps_nhwc(x_in, t_all[k], c_in, None if temb_all is None else temb_all[k], cfg_dup=True)
                eps_u, eps_c = eps[:b], eps[b:]
            else:
                eps_u, eps_c = None, self._eps_nhwc(img, t_all[k], c_in, None if temb_all is None else temb_all[k])
            cur, prev = x0_bufs[k & 1], x0_bufs[(k & 1) ^ 1]
            f = np.float32
            ops.sampler_step(img, eps_u, eps_c, eps_c.shape[-1], scale, [], (1., 0., 0., 0.),
                             f(p["alpha"]), f(p["sigma"]), f(p["A"] * p["alpha"] + p["c0"]), f(p["A"] * p["sigma"]),
                             f(p["c1"]), prev if p["c1"] != 0.0 else None, None, x_next, cur)
            img, x_next = x_next, img
            if callback:
                callback(k)
            if img_callback:
                img_callback(cur, k)
        return img, None
