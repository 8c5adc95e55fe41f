"""GlidePipeline -- the body of the reference's Taichu-GLIDE/src/txt2img.py:113-126 (base model 64x64 with
classifier-free guidance, then the 256x256 up-sampler), batch-sharded over the ranks of one node (BASELINE
config 4: batch 16 over 2 GPUs):

    samples = gaussian_p_sample_loop(diffusion_model, token, mask, shape=input_shape, num_timesteps=60, ...)[:pics]
    samples = ddim_sample_loop(super_res_model, samples=samples, token=token_up, mask=mask_up, up_shape=..., num_timesteps=27)

Sharding: every rank holds both models (built for its LOCAL batch: init_diffusion_model(shape=(2 * P_local, 3, 64, 64))),
rank 0 holds the prompts of the GLOBAL batch.  One packed RCCL broadcast (distributed.broadcast_glide_inputs) hands every
rank the prompts, the per-step random unconditional token ids -- drawn ONCE on rank 0, because main_funcs.py:37-38 redraws
them every step and all ranks must condition on the same unconditional prompts -- and the noise seed.  No collective inside
the loops.  Noise is drawn per IMAGE from `seed + global image index`, so an image's trajectory does not depend on how the
batch is sharded.
"""
import numpy as np
import torch

from .. import distributed as D
from .main_funcs import ddim_sample_loop, gaussian_p_sample_loop


class GlidePipeline:
    def __init__(self, diffusion_model, super_res_model, text_ctx=128, vocab_len=50001):
        self.dm, self.sr = diffusion_model, super_res_model
        self.text_ctx, self.vocab_len = int(text_ctx), int(vocab_len)
        self.device = diffusion_model.model.device

    def draw_uncond_tokens(self, seed, steps):
        """main_funcs.py:37: randint(1, vocab_len - 1, (text_ctx,)) per step -- all steps drawn up front from one stream."""
        return np.random.RandomState(seed).randint(1, self.vocab_len - 1, (steps, self.text_ctx)).astype(np.int32)

    def image_noise(self, seed, lo, hi, steps):
        """Per-image generators: image g draws its start noise [3,64,64], its `steps` ancestral noises and its up-sampler
        start noise [3,S,S] (x 0.997, main_funcs.py:57) from torch.Generator(seed + g), whatever rank owns it."""
        dev = self.device
        S = int(self.sr.shape[-1])
        h = int(self.dm.shape[-1])
        x0, per_step, up0 = [], [], []
        for g in range(lo, hi):
            gen = torch.Generator(device=dev)
            gen.manual_seed(int(seed) + g)
            x0.append(torch.randn((3, h, h), device=dev, generator=gen))
            per_step.append(torch.randn((steps, 3, h, h), device=dev, generator=gen))
            up0.append(torch.randn((3, S, S), device=dev, generator=gen) * 0.997)
        return torch.stack(x0), torch.stack(per_step, 1).contiguous(), torch.stack(up0)

    def __call__(self, tokens=None, mask=None, seed=0, tokens_up=None, mask_up=None, uncond_tokens=None, gather=False,
                 base_only=False):
        """tokens / mask [P_global, text_ctx] int (rank 0; None elsewhere).  Returns this rank's images
        [P_local, 3, 256, 256] (or, with gather=True, all of them on rank 0 and None elsewhere)."""
        rank, n = D.world()
        P_local = int(self.dm.pics_generated)
        P = P_local * n
        steps = int(self.dm.num_timesteps)
        if rank == 0 and uncond_tokens is None:
            uncond_tokens = self.draw_uncond_tokens(seed, steps)
        tok, msk, tok_up, msk_up, unc, seed, (lo, hi) = D.broadcast_glide_inputs(
            tokens, mask, uncond_tokens, seed, P, self.text_ctx, steps, self.device, tokens_up=tokens_up, mask_up=mask_up)
        assert hi - lo == P_local
        x0, noises, up0 = self.image_noise(seed, lo, hi, steps)
        rep = lambda t: torch.cat([t, t], 0)                       # the reference's 2P-row convention
        base = gaussian_p_sample_loop(self.dm, rep(tok), rep(msk), (2 * P_local,) + tuple(x0.shape[1:]), steps,
                                      text_ctx=self.text_ctx, noise=rep(x0), vocab_len=self.vocab_len,
                                      uncond_tokens=unc, step_noises=noises)[:P_local]
        out = base if base_only else ddim_sample_loop(self.sr, tuple(up0.shape), base, tok_up, msk_up,
                                                      int(self.sr.num_timesteps), noise=up0)
        self.last_uncond_tokens = unc
        return D.gather_latents(out) if (gather and n > 1) else out
