"""PLMSSampler -- MI355X-native mirror of the reference's ldm/models/diffusion/plms.py
(make_schedule :34-67, sample :69-122, plms_sampling :124-179, p_sample_plms :182-247; Wukong copy
WK/ldm/models/diffusion/plms.py:185-215 for the dict-conditioning form).

Same class name, constructor and ``sample(...)`` keyword surface.  What changes is the execution:
  * the host loop only enqueues work: one UNet forward (a replayed hipGraph) and ONE fused
    elementwise kernel per step (mdx_sampler_step_f32: CFG combine + Adams-Bashforth mix + x0/dir/x_prev),
    instead of ~15 separately dispatched MindSpore ops (plms.py:192-197, 218-226, 235-244);
  * the constant [2B,77,D] CFG context concat (plms.py:194) is built once, not every step, and its
    cross-attention K/V projections are cached inside the UNet;
  * per-step scalars are host floats (no `ms.numpy.full` tensors, plms.py:212-215).
Results are identical in structure to the reference (S+1 UNet calls for PLMS, intermediates dict,
callbacks once per step).
"""
import os

import numpy as np
import torch

from .... import ops
from ...._lib import MdxError
from ...modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps


def _first_tensor(c):
    while isinstance(c, (dict, list, tuple)):
        c = c[list(c.keys())[0]] if isinstance(c, dict) else c[0]
    return c


class _SamplerBase:
    """Shared loop for PLMSSampler (multistep) and DDIMSampler (single step)."""

    multistep = True

    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.verbose_print = kwargs.get("verbose_print", False)
        self.generator = kwargs.get("generator", None)

    # ---- plms.py:34-67
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if self.multistep and ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize,
                                                  num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps,
                                                  verbose=verbose and self.verbose_print)
        alphas_cumprod = np.asarray(self.model.alphas_cumprod, dtype=np.float32)
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        self.betas = self.model.betas
        self.alphas_cumprod = alphas_cumprod
        self.alphas_cumprod_prev = np.asarray(self.model.alphas_cumprod_prev, dtype=np.float32)
        self.sqrt_alphas_cumprod = np.sqrt(alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1. - alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1. - alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1. / alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1. / alphas_cumprod - 1)
        ddim_sigmas, ddim_alphas, ddim_alphas_prev = make_ddim_sampling_parameters(
            alphacums=alphas_cumprod, ddim_timesteps=self.ddim_timesteps, eta=ddim_eta,
            verbose=verbose and self.verbose_print)
        self.ddim_sigmas = ddim_sigmas
        self.ddim_alphas = ddim_alphas
        self.ddim_alphas_prev = ddim_alphas_prev
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - ddim_alphas).astype(np.float32)
        self.ddim_sigmas_for_original_num_steps = np.float32(ddim_eta) * np.sqrt(
            (1 - self.alphas_cumprod_prev) / (1 - self.alphas_cumprod) *
            (1 - self.alphas_cumprod / self.alphas_cumprod_prev))

    # ---- plms.py:69-122
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0.,
               score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100,
               unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        if conditioning is not None:
            cbs = _first_tensor(conditioning).shape[0]
            if cbs != batch_size:
                print(f"Warning: Got {cbs} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f'Data shape for {type(self).__name__} sampling is {size}')
        # the reference's sample() always runs the S-step DDIM grid (plms.py:107-121); `timesteps` /
        # `ddim_use_original_steps` are plms_sampling() options (plms.py:134-142) and are forwarded when given
        return self.plms_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0,
                                  ddim_use_original_steps=bool(kwargs.get("ddim_use_original_steps", False)),
                                  timesteps=kwargs.get("timesteps"), dropout_masks=kwargs.get("dropout_masks"),
                                  step_noises=kwargs.get("step_noises"),
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose,
                                  blend_noises=kwargs.get("blend_noises"))

    # ---- model call: prefer the NHWC fast path of our LatentDiffusion; any object with the reference's
    #      apply_model(x, t, cond) -> NCHW eps still works (its output is re-laid-out by a HIP kernel).
    def _eps_nhwc(self, x, t, cond, temb=None, cfg_dup=False):
        if hasattr(self.model, "apply_model_nhwc"):
            kw = {} if temb is None else {"temb": temb}
            if cfg_dup:     # both halves of the batch are the same x and t: only the contexts differ
                kw["cfg_dup"] = True
            return self.model.apply_model_nhwc(x, t, cond, **kw), None
        e = self.model.apply_model(x, t, cond)
        e = e.to(torch.float32).contiguous()
        buf = ops.nchw_to_nhwc(e, (e.shape[1] + 7) // 8 * 8)
        return buf, buf

    # ---- plms.py:124-179 + 182-247
    def plms_sampling(self, cond, shape, x_T=None, ddim_use_original_steps=False, callback=None, timesteps=None,
                      quantize_denoised=False, mask=None, x0=None, img_callback=None, log_every_t=100,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, verbose=True, blend_noises=None,
                      dropout_masks=None, step_noises=None):
        if mask is not None and x0 is None:
            raise ValueError("mask blending needs x0 (plms.py:154)")
        # quantize_x0 / score_corrector (plms.py:199-201, 218-219) call into caller-supplied objects (a first stage with
        # .quantize, a corrector with .modify_score) between the pieces of the fused step; see step() below
        if score_corrector is not None and self.model.parameterization != "eps":
            raise AssertionError('score_corrector needs parameterization == "eps" (plms.py:200)')
        if quantize_denoised and not hasattr(getattr(self.model, "first_stage_model", None), "quantize"):
            raise MdxError("quantize_x0 needs model.first_stage_model.quantize (plms.py:219)")
        corrector_kwargs = corrector_kwargs or {}
        if not 0. <= float(noise_dropout) < 1.:
            raise ValueError("noise_dropout must be in [0, 1)")
        # hybrid (inpainting) conditioning: {"c_concat": mask + masked-image latent, "c_crossattn": text} (inpaint.py:84-88,
        # WK plms.py:188-205); the concat part is the same for the cond and uncond halves
        # The other DiffusionWrapper keys (WK ddpm.py:361-374) arrive the way plms.py:188-195 hands them over: a bare conditioning
        # goes to the keyword the key reads -- 'concat': extra input channels (cond and uncond halves may differ), 'adm': class
        # labels [B] (concatenated [uncond; cond] like a text context), None: no conditioning at all.
        key = getattr(getattr(self.model, "model", None), "conditioning_key", "crossattn")
        c_cat = uc_cat = None
        if key == "concat":
            c_cat = _first_tensor(cond["c_concat"] if isinstance(cond, dict) else cond)
            if unconditional_conditioning is not None:
                uc_cat = _first_tensor(unconditional_conditioning["c_concat"] if isinstance(unconditional_conditioning, dict)
                                       else unconditional_conditioning)
            cond = unconditional_conditioning = None
        elif isinstance(cond, dict) and "c_concat" in cond:
            c_cat = _first_tensor(cond["c_concat"])
            cond = cond["c_crossattn"]
            if isinstance(unconditional_conditioning, dict):
                # WK plms.py:191-201 concatenates [uncond[k]; cond[k]] for EVERY dict key: an unconditional c_concat that differs from
                # the conditional one goes into the unconditional half (inpaint.py passes the same tensor in both)
                if unconditional_conditioning.get("c_concat") is not None:
                    uc_cat = _first_tensor(unconditional_conditioning["c_concat"])
                unconditional_conditioning = unconditional_conditioning["c_crossattn"]
        if key is None:
            cond = unconditional_conditioning = None
        cond = _first_tensor(cond) if cond is not None else None
        uc = _first_tensor(unconditional_conditioning) if unconditional_conditioning is not None else None
        if key in ("crossattn", "hybrid") and not (isinstance(cond, torch.Tensor) and cond.is_cuda):
            raise MdxError("conditioning must be a CUDA(HIP) tensor [B, T, context_dim]")
        if key == "adm" and cond is None:
            raise MdxError("'adm' conditioning needs the class labels [B] as `conditioning`")
        if key == "concat" and c_cat is None:
            raise MdxError("'concat' conditioning needs the extra input channels [B, C', H, W] as `conditioning`")
        if isinstance(cond, torch.Tensor) and cond.is_cuda:
            dev = cond.device
        elif isinstance(c_cat, torch.Tensor) and c_cat.is_cuda:
            dev = c_cat.device
        elif isinstance(x_T, torch.Tensor) and x_T.is_cuda:
            dev = x_T.device
        else:
            dev = self.model.unet.device
        if cond is not None:
            cond = torch.as_tensor(cond).to(dev)
            uc = None if uc is None else torch.as_tensor(uc).to(dev)
        b = shape[0]
        if x_T is None:
            img = torch.randn(shape, device=dev, dtype=torch.float32, generator=self.generator)
        else:
            img = torch.as_tensor(x_T).to(device=dev, dtype=torch.float32).contiguous().clone()
        # ---- which timesteps, and which tables `index` selects from (plms.py:134-142, 205-208)
        if ddim_use_original_steps:
            # every DDPM step (or the first `timesteps` of them): tables are the model's alphas_cumprod themselves.  The
            # reference reads `self.model.ddim_sigmas_for_original_num_steps` (plms.py:208), an attribute make_schedule
            # creates on the SAMPLER (plms.py:64-67): we use the sampler's.
            n_orig = int(self.ddpm_num_timesteps if timesteps is None else timesteps)
            if not 0 < n_orig <= self.ddpm_num_timesteps:
                raise ValueError(f"timesteps must be in (0, {self.ddpm_num_timesteps}] with ddim_use_original_steps")
            time_range = np.arange(n_orig - 1, -1, -1, dtype=np.int64)
            alphas, alphas_prev = self.alphas_cumprod, self.alphas_cumprod_prev
            sqrt_one_minus_alphas = self.sqrt_one_minus_alphas_cumprod
            sigmas = np.asarray(self.ddim_sigmas_for_original_num_steps, dtype=np.float32)
        else:
            ts = self.ddim_timesteps
            if timesteps is not None:                                # plms.py:137-139: a prefix of the DDIM grid
                n_ddim = ts.shape[0]
                subset_end = int(min(timesteps / n_ddim, 1) * n_ddim) - 1
                ts = ts[:subset_end]
                if ts.shape[0] == 0:
                    raise ValueError(f"timesteps={timesteps} selects an empty subset of the {n_ddim}-step DDIM grid")
            time_range = np.flip(ts)
            alphas, alphas_prev = self.ddim_alphas, self.ddim_alphas_prev
            sqrt_one_minus_alphas, sigmas = self.ddim_sqrt_one_minus_alphas, self.ddim_sigmas
        total_steps = int(time_range.shape[0])
        if verbose:
            print(f"Running {type(self).__name__} Sampling with {total_steps} timesteps")
        scale = float(unconditional_guidance_scale)
        use_cfg = not ((uc is None and uc_cat is None) or scale == 1.)
        nb = 2 * b if use_cfg else b
        if cond is None:
            c_in = None
        else:
            c_in = torch.cat([uc.to(cond.dtype), cond], 0).contiguous() if use_cfg else cond.contiguous()   # built ONCE
        x_in = None                                                  # (plms.py:194 rebuilds the concat every step)
        cx = shape[1]
        if use_cfg or c_cat is not None:
            ccat = 0 if c_cat is None else int(c_cat.shape[1])
            x_in = torch.empty((nb, cx + ccat) + tuple(shape[2:]), device=dev, dtype=torch.float32)
            if c_cat is not None:                                    # DiffusionWrapper 'hybrid': cat(x, c_concat), written once
                cc = c_cat.to(device=dev, dtype=torch.float32)
                x_in[nb - b:, cx:] = cc                              # batch = [uncond ; cond]
                if use_cfg:
                    x_in[:b, cx:] = cc if uc_cat is None else uc_cat.to(device=dev, dtype=torch.float32)
        # the two halves of a guidance batch are the same UNet input unless the unconditional branch has its own c_concat
        same_halves = use_cfg and c_in is not None and (c_cat is None or uc_cat is None)
        if mask is not None:
            mask = torch.as_tensor(mask).to(device=dev, dtype=torch.float32)
            x0 = torch.as_tensor(x0).to(device=dev, dtype=torch.float32)
        # per-step timestep vectors, fp32 on the device (the UNet's sinusoid takes float timesteps, util.py:111-131)
        t_all = torch.as_tensor(np.ascontiguousarray(time_range), dtype=torch.float32, device=dev)
        # the timestep-only part of the UNet (time_embed MLP + the ResBlock emb_layers, openaimodel.py:550-551,188) for
        # ALL steps in one batched pass; step i then hands row i to the UNet instead of recomputing it
        temb_all = None
        if (hasattr(self.model, "time_embedding_table") and os.environ.get("MDX_SAMPLER_TEMB_TABLE", "1") != "0"
                and getattr(getattr(self.model, "unet", None), "num_classes", None) is None):     # (label_emb(y) joins emb per call)
            temb_all = self.model.time_embedding_table(t_all)
        t_all = t_all[:, None].expand(total_steps, nb).contiguous()

        intermediates = {'x_inter': [img.clone()], 'pred_x0': [img.clone()]}
        pool = [torch.empty_like(img) for _ in range(4)]  # eps history buffers (3 live + 1 being written)
        hist = []                                         # newest first; at most 3 (plms.py:169-171)
        pred_x0 = torch.empty_like(img)
        x_next = torch.empty_like(img)

        def model_eps(x, i):
            t_row, temb = t_all[i], (None if temb_all is None else temb_all[i])
            if use_cfg:
                x_in[:b, :cx].copy_(x)
                x_in[b:, :cx].copy_(x)
                eps, keep = self._eps_nhwc(x_in, t_row, c_in, temb, cfg_dup=same_halves)
                return eps[:b], eps[b:], keep           # batch = [uncond ; cond] (plms.py:192-195)
            if x_in is not None:
                x_in[:, :cx].copy_(x)
                eps, keep = self._eps_nhwc(x_in, t_row, c_in, temb)
                return None, eps, keep
            eps, keep = self._eps_nhwc(x, t_row, c_in, temb)
            return None, eps, keep

        drop_count, noise_count = [0], [0]

        hook_e = torch.empty_like(img) if score_corrector is not None else None
        hook_x = torch.empty_like(img) if score_corrector is not None else None
        hook_p = torch.empty_like(img) if quantize_denoised else None

        def step(x, eps_u, eps_c, index, coef, olds, e_out, x_out, p_out, xm=None, tm=None):
            """One get_x_prev_and_pred_x0 (plms.py:210-228) on e' = coef[0] * e_t + sum coef[k] * olds[k-1], e_t = the
            CFG-combined model output (optionally written to e_out).  xm / tm: the image and timestep the model output
            was evaluated at (what a score corrector is handed, plms.py:201)."""
            a_t, a_prev = np.float32(alphas[index]), np.float32(alphas_prev[index])
            sigma_t = np.float32(sigmas[index])
            if score_corrector is not None:
                # e_t leaves the fused kernel (pass 1: CFG combine only), goes through the caller's modify_score, and
                # re-enters as a single NHWC fp16 model output (pass 2: multistep mix + update)
                ops.sampler_step(x, eps_u, eps_c, eps_c.shape[-1], scale, [], (1., 0., 0., 0.), 1., 0., 1., 0., 0., None,
                                 hook_e, hook_x, None)
                tvec = torch.full((b,), int(tm), device=dev, dtype=torch.long)
                e_mod = score_corrector.modify_score(self.model, hook_e, x if xm is None else xm, tvec, cond,
                                                     **corrector_kwargs)
                e_mod = torch.as_tensor(e_mod).to(device=dev, dtype=torch.float32).contiguous()
                if tuple(e_mod.shape) != tuple(x.shape):
                    raise MdxError(f"score_corrector returned shape {tuple(e_mod.shape)}, expected {tuple(x.shape)}")
                eps_c = ops.nchw_to_nhwc(e_mod, (x.shape[1] + 7) // 8 * 8)
                eps_u = None
            noise = None
            if float(sigma_t) != 0.0:
                if step_noises is not None:   # tests inject the k-th N(0,1) draw to compare with the oracle (eta != 0)
                    noise = torch.as_tensor(step_noises[noise_count[0]]).to(device=dev, dtype=torch.float32) * temperature
                else:
                    noise = torch.randn(x.shape, device=dev, dtype=torch.float32, generator=self.generator) * temperature
                noise_count[0] += 1
                if noise_dropout > 0.:
                    # plms.py:224-225 `ops.dropout(noise, p=noise_dropout)`: zero with probability p, scale the rest by
                    # 1 / (1 - p).  dropout_masks[k] (1 = keep) injects the k-th draw so that tests can feed the oracle
                    # the same mask; otherwise it comes from the sampler's generator.
                    if dropout_masks is not None:
                        keep = torch.as_tensor(dropout_masks[drop_count[0]]).to(device=dev, dtype=torch.float32)
                    else:
                        keep = (torch.rand(x.shape, device=dev, generator=self.generator) >= noise_dropout).to(torch.float32)
                    drop_count[0] += 1
                    noise = noise * keep / (1. - float(noise_dropout))
            if quantize_denoised and p_out is None:
                p_out = hook_p
            ops.sampler_step(x, eps_u, eps_c, eps_c.shape[-1], scale, olds, coef,
                             np.sqrt(a_t), np.float32(sqrt_one_minus_alphas[index]), np.sqrt(a_prev),
                             np.sqrt(np.float32(1.) - a_prev - sigma_t ** 2), sigma_t, noise, e_out, x_out, p_out)
            if quantize_denoised:
                # pred_x0, _, *_ = first_stage_model.quantize(pred_x0) (plms.py:218-219); x_prev is linear in pred_x0
                q = self.model.first_stage_model.quantize(p_out)[0]
                q = torch.as_tensor(q).to(device=dev, dtype=torch.float32)
                x_out.add_((q - p_out) * float(np.sqrt(a_prev)))
                p_out.copy_(q)

        for i, step_t in enumerate(time_range):
            index = total_steps - i - 1
            if mask is not None:                                     # plms.py:153-157 (WK: q_sample gets explicit noise)
                ts = torch.full((b,), int(step_t), device=dev, dtype=torch.long)
                if blend_noises is not None:                          # tests inject the draws to compare with the oracle
                    noise = torch.as_tensor(blend_noises[i]).to(device=dev, dtype=torch.float32)
                else:
                    noise = torch.randn(x0.shape, device=dev, dtype=torch.float32, generator=self.generator)
                img_orig = self.model.q_sample(x0, ts, noise)
                img = img_orig * mask + (1. - mask) * img
            eps_u, eps_c, _keep = model_eps(img, i)
            if not self.multistep:
                step(img, eps_u, eps_c, index, (1., 0., 0., 0.), [], None, x_next, pred_x0, tm=step_t)
            else:
                e_buf = pool.pop()
                n_old = len(hist)
                if n_old == 0:
                    # Pseudo Improved Euler (2nd order), plms.py:231-235: S+1 UNet calls in total
                    step(img, eps_u, eps_c, index, (1., 0., 0., 0.), [], e_buf, x_next, None, tm=step_t)
                    i_next = min(i + 1, total_steps - 1)
                    eps_u2, eps_c2, _keep2 = model_eps(x_next, i_next)
                    step(img, eps_u2, eps_c2, index, (.5, .5, 0., 0.), [e_buf], None, x_next, pred_x0,
                         xm=x_next, tm=time_range[i_next])
                elif n_old == 1:  # Adams-Bashforth 2, plms.py:236-238
                    step(img, eps_u, eps_c, index, (3. / 2, -1. / 2, 0., 0.), hist[:1], e_buf, x_next, pred_x0, tm=step_t)
                elif n_old == 2:  # AB-3, plms.py:239-241
                    step(img, eps_u, eps_c, index, (23. / 12, -16. / 12, 5. / 12, 0.), hist[:2], e_buf, x_next,
                         pred_x0, tm=step_t)
                else:             # AB-4, plms.py:242-244
                    step(img, eps_u, eps_c, index, (55. / 24, -59. / 24, 37. / 24, -9. / 24), hist[:3], e_buf,
                         x_next, pred_x0, tm=step_t)
                hist.insert(0, e_buf)
                if len(hist) > 3:
                    pool.append(hist.pop())
            img, x_next = x_next, img
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img.clone())
                intermediates['pred_x0'].append(pred_x0.clone())
        return img, intermediates


class PLMSSampler(_SamplerBase):
    """Reference: ldm/models/diffusion/plms.py:27 ``class PLMSSampler``."""
    multistep = True
