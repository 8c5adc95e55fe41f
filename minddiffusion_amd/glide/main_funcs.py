"""gaussian_p_sample_loop / ddim_sample_loop -- mirrors of the reference's main_funcs.py:21-69 (same signatures).
The host loop only enqueues one UNet graph replay + one fused sampler kernel per step."""
import numpy as np
import torch


def gaussian_p_sample_loop(diffusion_model, token, mask, shape, num_timesteps, tokenizer=None, text_ctx=128,
                           noise=None, progress=False, dtype=None, vocab_len=50001, rng=None, uncond_tokens=None,
                           step_noises=None):
    """main_funcs.py:21-44.  `rng` (numpy RandomState) draws the per-step random unconditional prompt
    (main_funcs.py:37: randint(1, vocab_len-1, (text_ctx,))); `uncond_tokens` / `step_noises` inject them for tests."""
    dev = diffusion_model.model.device
    img = noise.to(dev, torch.float32) if noise is not None else torch.randn(tuple(shape), device=dev)
    rng = rng or np.random.RandomState()
    ones = np.ones((text_ctx,), np.int32)
    # every step's unconditional prompt, drawn in the loop's order from the same stream the per-step draws of main_funcs.py:37 use
    # (they depend on nothing the model computes): knowing them all, the model runs its text transformer once per loop
    # (GenerativePSampleDiffusionModel.begin_loop) instead of once per step
    draws = [np.asarray(uncond_tokens[k]).astype(np.int32) if uncond_tokens is not None
             else rng.randint(1, vocab_len - 1, (text_ctx,)).astype(np.int32) for k in range(num_timesteps)]
    announce = hasattr(diffusion_model, "begin_loop") and num_timesteps == getattr(diffusion_model, "num_timesteps", -1)
    if announce:
        diffusion_model.begin_loop(token, mask, np.stack(draws))
    try:
        for k, i in enumerate(list(range(num_timesteps))[::-1]):
            sample, _ = diffusion_model(x=img, timesteps=torch.tensor([i], dtype=torch.int32), token=token, mask=mask,
                                        random_token=draws[k], random_mask=ones,
                                        noise=None if step_noises is None else step_noises[k])
            img = sample
    finally:
        if announce:
            diffusion_model.end_loop()
    return img


def ddim_sample_loop(super_res_model, up_shape, samples, token, mask, num_timesteps, noise=None, progress=False,
                     dtype=None):
    """main_funcs.py:47-69 (start noise x 0.997, 'fast27' steps)."""
    dev = super_res_model.model.device
    if noise is not None:
        img = noise.to(dev, torch.float32)
    else:
        img = torch.randn(tuple(up_shape), device=dev) * 0.997
    announce = hasattr(super_res_model, "begin_loop") and num_timesteps == getattr(super_res_model, "num_timesteps", -1)
    if announce:
        super_res_model.begin_loop(token, mask)
    try:
        for i in list(range(num_timesteps))[::-1]:
            sample, _ = super_res_model(x=img, timesteps=torch.tensor([i], dtype=torch.int32), token=token, mask=mask,
                                        samples=samples)
            img = sample
    finally:
        if announce:
            super_res_model.end_loop()
    return img
