"""CPU tests pinning the GLIDE oracle: schedules against goldens generated from the reference's own
gaussian_computation.py (tests/golden/make_glide_schedule_golden.py), structural identities of the network."""
import os

import numpy as np
import torch

from oracle import glide as G

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "glide_schedule.npz"))

TINY_GLIDE = dict(image_size=16, in_channels=3, out_channels=6, model_channels=64, num_res_blocks=1,
                  channel_mult=(1, 2), num_head_channels=64, attention_resolutions=(1, 2), text_ctx=16, xf_width=64,
                  xf_layers=2, xf_heads=1, n_vocab=100, noise_schedule="squaredcos_cap_v2", timestep_respacing="10",
                  diffusion_steps=1000)


def test_schedules_match_reference_goldens():
    np.testing.assert_allclose(G.named_beta_schedule("squaredcos_cap_v2", 1000), GOLD["cosine_betas_1000"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(G.named_beta_schedule("linear", 1000), GOLD["linear_betas_1000"], rtol=0, atol=1e-15)
    assert sorted(G.space_timesteps(1000, "60")) == list(GOLD["space_60"])
    assert sorted(G.space_timesteps(1000, "fast27")) == list(GOLD["space_fast27"])
    assert sorted(G.space_timesteps(1000, "100,50")) == list(GOLD["space_100_50"])


def test_respaced_tables():
    s = G.respaced_schedule("squaredcos_cap_v2", 1000, "60")
    assert len(s["betas"]) == 60 and list(s["timestep_map"][:3]) == [0, 17, 34] and s["timestep_map"][-1] == 999
    # respaced betas reproduce the base alpha-bar at the kept steps (diffusion_creator.py:96-106)
    ac = np.cumprod(1 - GOLD["cosine_betas_1000"])
    np.testing.assert_allclose(s["alphas_cumprod"], ac[s["timestep_map"]], rtol=0, atol=2e-6)   # betas are cast to fp32 (:200)
    assert s["post_logvar_clipped"][0] == s["post_logvar_clipped"][1]
    u = G.respaced_schedule("linear", 1000, "fast27")
    assert len(u["betas"]) == 27 and u["timestep_map"][-1] == 997


def test_structure_and_param_counts():
    inb, mid, outb = G.unet_structure(G.BASE_OPTIONS)
    layers = [l for b in inb + [mid] + outb for l in b]
    assert sum(l[0] == "res" for l in layers) == 12 + 3 + 2 + 16 + 3   # 3/level in, 3 down, mid 2, 4/level out, 3 up
    assert sorted({l[2] for l in layers if l[0] == "attn"}) == [6, 9, 12]
    n = sum(int(np.prod(s)) for s in G.param_shapes(G.BASE_OPTIONS).values())
    assert abs(n / 1e6 - 384.9) < 1.0, n       # 384.9 M incl. the text transformer
    nu = sum(int(np.prod(s)) for s in G.param_shapes(G.UPSAMPLE_OPTIONS).values())
    assert abs(nu / 1e6 - 398.2) < 1.0, nu


def test_tiny_forward_and_zero_init():
    o = TINY_GLIDE
    rng = np.random.RandomState(0)
    x = rng.randn(2, 3, 16, 16).astype(np.float32)
    tok = rng.randint(1, 99, (2, 16))
    mask = np.ones((2, 16), np.int64)
    mask[1, 10:] = 0
    net = G.GlideUNetOracle(o, G.init_params(o, seed=0))
    out = net(x, torch.tensor([400.0, 400.0]), tok, mask)
    assert out.shape == (2, 6, 16, 16) and torch.isfinite(out).all()
    # padded positions are replaced by the padding embedding: changing masked-out token ids changes nothing
    tok2 = tok.copy()
    tok2[1, 10:] = 7
    assert torch.equal(net(x, torch.tensor([400.0, 400.0]), tok2, mask), out)
    z = G.GlideUNetOracle(o, G.init_params(o, seed=0, zero_init=True))(x, torch.tensor([400.0, 400.0]), tok, mask)
    assert float(z.abs().max()) == 0.0      # out2 is a zero_module conv (unet.py:540-546)


def test_sampling_loops_run_and_last_step_is_noise_free():
    o = TINY_GLIDE
    net = G.GlideUNetOracle(o, G.init_params(o, seed=1))
    sch = G.respaced_schedule(o["noise_schedule"], 1000, "10")   # (fewer steps drive the fp32 respaced alpha-bar to 0)
    rng = np.random.RandomState(3)
    x_T = rng.randn(1, 3, 16, 16).astype(np.float32)
    tok = rng.randint(1, 99, (1, 16)); mask = np.ones((1, 16), np.int64)
    unc = rng.randint(1, 99, (10, 16))
    noises = rng.randn(10, 1, 3, 16, 16).astype(np.float32)
    a = G.p_sample_loop(net, sch, x_T, tok, mask, 3.0, unc, noises)
    noises2 = noises.copy(); noises2[-1] += 5.0          # the t == 0 step must ignore its noise (PSample :86-89)
    b = G.p_sample_loop(net, sch, x_T, tok, mask, 3.0, unc, noises2)
    assert torch.equal(a, b) and torch.isfinite(a).all() and float(a.abs().max()) <= 1.0 + 1e-5
    lb = G.legacy_bilinear(torch.arange(16.0).reshape(1, 1, 4, 4), 8)
    assert lb.shape == (1, 1, 8, 8) and float(lb[0, 0, 0, 1]) == 0.5 and float(lb[0, 0, 7, 7]) == 15.0
