"""How far is the reference's SHIPPED arithmetic (`use_fp16: True`: fp16 schedule tables ddpm.py:129-139, fp16 LayerNorm
attention.py:176-178, fp16 scores / softmax / PV attention.py:138-152; Taichu-GLIDE: fp16 timestep embedding
simple_nn.py:150-169, fp16 diffusion arithmetic src/txt2img.py:92-93) from the all-fp32 oracle the parity tests use -- and
where does the GPU path (fp16 storage, fp32 accumulation / statistics / softmax) sit between the two?

MindSpore cannot run here, so the oracle is unpinned; `oracle.ldm.emulate_fp16()` is the only further information
available: op-boundary fp16 rounding of the same restatement (a LOWER bound on the reference's own fp16 noise).
For every case three distances are logged (gpurun_out/parity_log.jsonl -> profiles/parity_r02.jsonl):
    d(fp32 oracle, fp16-emulated oracle)   d(GPU, fp32 oracle)   d(GPU, fp16-emulated oracle)
and the test asserts the GPU is no farther from the fp16-mode reference than the fp32 oracle is, up to the GPU's own
stated tolerance against the fp32 oracle (triangle inequality with slack 0: d(G,E) <= d(O,E) + tol(G,O))."""
import json
import os

import numpy as np
import pytest
import torch

from _util import LOG, metrics
from oracle import glide as OG
from oracle import ldm as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _three_way(name, gpu, o32, o16, tol_gpu_o32):
    d_oe = metrics(o16, o32)
    d_go = metrics(gpu, o32)
    d_ge = metrics(gpu, o16)
    finite16 = bool(np.isfinite(o16.detach().cpu().numpy() if isinstance(o16, torch.Tensor) else o16).all())
    rec = dict(name=f"fp16ref_{name}", d_oracle32_vs_fp16emu=d_oe["rel_l2"], d_gpu_vs_oracle32=d_go["rel_l2"],
               d_gpu_vs_fp16emu=d_ge["rel_l2"], fp16emu_finite=finite16, tol_gpu_vs_oracle32=tol_gpu_o32)
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print("PARITY", json.dumps(rec))
    assert d_go["finite"] and d_go["rel_l2"] <= tol_gpu_o32, rec
    if finite16:
        assert d_ge["rel_l2"] <= d_oe["rel_l2"] + tol_gpu_o32, rec
    return rec


def _tiny_ldm():
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    cfg = dict(TINY_UNET)
    ocfg = dict(cfg)
    ocfg.setdefault("num_heads", -1)
    ocfg.setdefault("num_head_channels", -1)
    params = O.init_params(ocfg, seed=3)
    net = UNetModel(**cfg)
    net.load_state_dict(params)
    return cfg, net, O.UNetOracle(ocfg, params), LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)


def test_fp16_reference_distance_ldm_tiny_single_call_and_50_steps():
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    cfg, net, oracle, model = _tiny_ldm()
    rng = np.random.RandomState(0)
    B, H, W, T = 2, 8, 8, 6
    x = rng.randn(B, 4, H, W).astype(np.float32)
    ctx = rng.randn(B, T, cfg["context_dim"]).astype(np.float32)
    uc = np.repeat(rng.randn(1, T, cfg["context_dim"]).astype(np.float32), B, 0)
    ts = torch.full((B,), 981.0)
    dev = lambda a: torch.tensor(a, device=DEV)
    o32 = oracle(x, ts, ctx)
    with O.emulate_fp16():
        o16 = oracle(x, ts, ctx)
    _three_way("tiny_unet_single_call", net(dev(x), ts.to(DEV), dev(ctx)).cpu(), o32, o16, 5e-3)
    omodel = O.ModelOracle(oracle)
    for sampler, cls in (("ddim", DDIMSampler), ("plms", PLMSSampler)):
        kw = dict(unconditional_guidance_scale=7.5, unconditional_conditioning=uc)
        s32, _ = O.sample(omodel, 50, B, (4, H, W), ctx, x, sampler, **kw)
        with O.emulate_fp16():
            s16, _ = O.sample(omodel, 50, B, (4, H, W), ctx, x, sampler, **kw)
        got, _ = cls(model).sample(50, B, (4, H, W), conditioning=dev(ctx), x_T=dev(x), verbose=False,
                                   unconditional_guidance_scale=7.5, unconditional_conditioning=dev(uc))
        _three_way(f"tiny_{sampler}_50_steps_cfg7.5", got.cpu(), s32, s16, 1e-2)


def test_fp16_reference_distance_sd2_full_size_single_step():
    """BASELINE config 0 / 1 building block at full size."""
    from minddiffusion_amd.configs import SD2_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    torch.set_num_threads(min(96, os.cpu_count() or 8))
    params = O.init_params(O.SD2_UNET, seed=0)
    net = UNetModel(**dict(SD2_UNET))
    net.load_state_dict(params)
    oracle = O.UNetOracle(O.SD2_UNET, params)
    x = np.random.RandomState(42).randn(1, 4, 64, 64).astype(np.float32)
    ctx = np.random.RandomState(1).randn(1, 77, 1024).astype(np.float32)
    o32 = oracle(x, torch.tensor([981.0]), ctx)
    with O.emulate_fp16():
        o16 = oracle(x, torch.tensor([981.0]), ctx)
    got = net(torch.tensor(x, device=DEV), torch.tensor([981.0], device=DEV), torch.tensor(ctx, device=DEV)).cpu()
    _three_way("sd2_full_single_step_B1_64x64", got, o32, o16, 5e-3)


def test_fp16_reference_distance_glide_tiny_unet_and_loops():
    """Taichu-GLIDE: the fp16 timestep embedding (t * freq at t ~ 1000 has an fp16 spacing of 0.5) and the fp16 diffusion
    arithmetic are the visible deviations of the reference's fp16 mode."""
    from minddiffusion_amd.glide.diffusion_creator import init_diffusion_model
    from minddiffusion_amd.glide.main_funcs import gaussian_p_sample_loop
    from test_glide_gpu import OTINY, TINY
    params = OG.init_params(OTINY, seed=2)
    P, steps = 2, 10
    dm = init_diffusion_model(options=TINY, guidance_scale=3.0, shape=(2 * P, 3, 16, 16), params=params)
    oracle = OG.GlideUNetOracle(OTINY, params)
    rng = np.random.RandomState(11)
    x_T = rng.randn(P, 3, 16, 16).astype(np.float32)
    tok = rng.randint(1, 99, (P, 16)).astype(np.int32)
    mask = np.ones((P, 16), np.int32)
    unc = rng.randint(1, 99, (steps, 16)).astype(np.int32)
    noises = rng.randn(steps, P, 3, 16, 16).astype(np.float32)
    # single guided evaluation at the highest respaced timestep
    x2 = np.concatenate([x_T, x_T], 0)
    t = torch.full((2 * P,), 999.0)
    tk = np.concatenate([tok, np.repeat(unc[:1], P, 0)], 0)
    mk = np.concatenate([mask, mask], 0)
    o32 = oracle(x2, t, tk, mk)
    with OG.emulate_fp16():
        o16 = oracle(x2, t, tk, mk)
    got = dm.model(torch.tensor(x2, device=DEV), t.to(DEV), torch.tensor(tk, device=DEV), torch.tensor(mk, device=DEV)).cpu()
    _three_way("glide_tiny_unet_t999", got, o32, o16, 5e-3)
    sch = OG.respaced_schedule("squaredcos_cap_v2", 1000, "10")
    s32 = OG.p_sample_loop(oracle, sch, x_T, tok, mask, 3.0, unc, noises)
    with OG.emulate_fp16():
        s16 = OG.p_sample_loop(oracle, sch, x_T, tok, mask, 3.0, unc, noises)
    tok2, mask2 = np.concatenate([tok, tok], 0), np.concatenate([mask, mask], 0)
    g = gaussian_p_sample_loop(dm, torch.tensor(tok2), torch.tensor(mask2), (2 * P, 3, 16, 16), steps, text_ctx=16,
                               noise=torch.tensor(x2), vocab_len=100, uncond_tokens=list(unc),
                               step_noises=[torch.tensor(n, device=DEV) for n in noises])[:P].cpu()
    _three_way("glide_tiny_p_sample_loop_10_steps", g, s32, s16, 3e-2)
