"""CPU tests pinning the oracle (oracle/ldm.py): SURVEY App. C known-answer values,
naive-numpy second opinions per primitive, and structural identities of the reference graph."""
import numpy as np
import pytest
import torch

from oracle import ldm, naive

TINY = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[2, 1],
            num_res_blocks=1, channel_mult=[1, 2], num_head_channels=32, num_heads=-1,
            use_spatial_transformer=True, use_linear_in_transformer=True, transformer_depth=1,
            context_dim=64, legacy=False)


def test_schedule_kat():
    s = ldm.register_schedule()
    assert abs(s["betas"][0] - 8.4999995e-4) < 1e-9
    assert abs(s["betas"][999] - 1.1999999e-2) < 2e-8
    ac = s["alphas_cumprod"]
    for idx, val in ((0, 0.99915), (1, 0.998296), (499, 0.27766952), (981, 0.005775489), (999, 0.0046600895)):
        assert abs(ac[idx] - val) < 2e-6 * max(1.0, val / 1e-3) or abs(ac[idx] - val) / val < 2e-5, (idx, ac[idx])
    assert s["alphas_cumprod_prev"][0] == 1.0


def test_ddim_params_kat():
    s = ldm.register_schedule()
    ts = ldm.make_ddim_timesteps(50)
    assert len(ts) == 50 and ts[0] == 1 and ts[1] == 21 and ts[-1] == 981
    sig, a, ap = ldm.make_ddim_sampling_parameters(s["alphas_cumprod"], ts, 0.0)
    assert np.all(sig == 0)
    assert abs(a[0] - 0.998296) < 1e-6 and abs(a[49] - 0.005775489) < 1e-7
    assert abs(ap[0] - 0.99915) < 1e-6 and abs(ap[1] - 0.998296) < 1e-6 and abs(ap[49] - 0.007281713) < 1e-7
    assert abs(np.sqrt(1 - a)[49] - 0.99710804) < 1e-6


def test_timestep_embedding_kat():
    e = ldm.timestep_embedding(torch.tensor([981]), 320).numpy()[0]
    np.testing.assert_allclose(e[:3], [0.6799572, -0.7984292, 0.578114], atol=1e-4)
    np.testing.assert_allclose(e[160:163], [0.7332518, 0.6020887, 0.815956], atol=1e-4)
    np.testing.assert_allclose(e, naive.timestep_embedding([981], 320)[0], atol=2e-4)


def test_glide_schedule_golden_loads():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "glide_schedule.npz"))
    assert abs(g["cosine_betas_1000"][0] - 4.12842248e-05) < 1e-12
    assert g["cosine_betas_1000"][-1] == 0.999
    assert list(g["space_fast27"][:5]) == [0, 22, 44, 66, 88] and len(g["space_60"]) == 60
    assert g["space_60"][1] == 17 and g["space_60"][-1] == 999


def test_primitives_vs_naive():
    rng = np.random.RandomState(0)
    x = rng.standard_normal((2, 64, 5, 6)).astype(np.float32)
    g = rng.standard_normal(64).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    got = ldm.group_norm(torch.tensor(x), torch.tensor(g), torch.tensor(b), 1e-5).numpy()
    np.testing.assert_allclose(got, naive.group_norm(x, g, b, 1e-5), atol=2e-5)
    w = rng.standard_normal((7, 64, 3, 3)).astype(np.float32) / 24
    bias = rng.standard_normal(7).astype(np.float32)
    for stride in (1, 2):
        got = ldm.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(bias), stride=stride).numpy()
        np.testing.assert_allclose(got, naive.conv2d(x, w, bias, stride=stride), atol=2e-5)
    t = rng.standard_normal((2, 9, 64)).astype(np.float32)
    got = ldm.layer_norm(torch.tensor(t), torch.tensor(g), torch.tensor(b), 1e-5).numpy()
    np.testing.assert_allclose(got, naive.layer_norm(t, g, b, 1e-5), atol=2e-5)
    np.testing.assert_allclose(ldm.gelu_tanh(torch.tensor(t)).numpy(), naive.gelu_tanh(t.astype(np.float64)), atol=1e-6)
    np.testing.assert_allclose(ldm.silu(torch.tensor(t)).numpy(), naive.silu(t.astype(np.float64)), atol=1e-6)
    up = ldm.upsample_nearest2x(torch.tensor(x)).numpy()
    assert up.shape == (2, 64, 10, 12) and np.all(up[:, :, 3, 5] == x[:, :, 1, 2])


def test_attention_vs_naive():
    rng = np.random.RandomState(1)
    cfg = dict(TINY)
    params = ldm.init_params(cfg, seed=3)
    net = ldm.UNetOracle(cfg, params)
    pre = "input_blocks.1.1.transformer_blocks.0.attn2."
    x = rng.standard_normal((2, 10, 64)).astype(np.float32)
    ctx = rng.standard_normal((2, 7, 64)).astype(np.float32)
    got = net._attn(pre, torch.tensor(x), torch.tensor(ctx), heads=2).numpy()
    p = params
    q = x @ p[pre + "to_q.weight"].T
    k = ctx @ p[pre + "to_k.weight"].T
    v = ctx @ p[pre + "to_v.weight"].T
    ref = naive.attention(q, k, v, 2) @ p[pre + "to_out.0.weight"].T + p[pre + "to_out.0.bias"]
    np.testing.assert_allclose(got, ref, atol=2e-5)


def test_structure_counts_sd2():
    """22 ResBlocks, 16 SpatialTransformers, 865.9 M parameters, heads 5/10/20."""
    inb, mid, outb = ldm.unet_structure(ldm.SD2_UNET)
    layers = [l for blk in inb + [mid] + outb for l in blk]
    assert sum(l[0] == "res" for l in layers) == 22
    assert sum(l[0] == "st" for l in layers) == 16
    assert sorted({l[2] for l in layers if l[0] == "st"}) == [5, 10, 20]
    assert all(l[3] == 64 for l in layers if l[0] == "st")
    assert len(inb) == 12 and len(outb) == 12
    n = sum(int(np.prod(s)) for s in ldm.unet_param_shapes(ldm.SD2_UNET).values())
    assert abs(n / 1e6 - 865.9) < 0.1, n
    nw = sum(int(np.prod(s)) for s in ldm.unet_param_shapes(ldm.WUKONG_UNET).values())
    assert abs(nw / 1e6 - 859.5) < 0.1, nw
    _, midw, _ = ldm.unet_structure(ldm.WUKONG_UNET)
    assert midw[1][2] == 8 and midw[1][3] == 160


def test_zero_init_unet_outputs_zero():
    """zero_module'd out conv (openaimodel.py:524) => freshly constructed UNet outputs exactly 0."""
    params = ldm.init_params(TINY, seed=0, zero_init=True)
    net = ldm.UNetOracle(TINY, params)
    x = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    out = net(x, torch.tensor([981, 981]), torch.randn(2, 5, 64, generator=torch.Generator().manual_seed(1)))
    assert out.shape == (2, 4, 8, 8) and float(out.abs().max()) == 0.0


def _tiny_model(seed=0):
    net = ldm.UNetOracle(TINY, ldm.init_params(TINY, seed=seed))
    return ldm.ModelOracle(net)


def test_sampler_identities():
    model = _tiny_model()
    rng = np.random.RandomState(42)
    x_T = rng.randn(1, 4, 8, 8).astype(np.float32)
    c = np.random.RandomState(1).randn(1, 5, 64).astype(np.float32)
    uc = np.random.RandomState(2).randn(1, 5, 64).astype(np.float32)
    # PLMS: S+1 UNet calls (plms.py:231-235); DDIM: S calls.
    model.calls = 0
    s_plms, inter = ldm.sample(model, 5, 1, (4, 8, 8), c, x_T, "plms", unconditional_guidance_scale=3.0,
                               unconditional_conditioning=uc)
    assert model.calls == 6 and s_plms.shape == (1, 4, 8, 8)
    model.calls = 0
    s_ddim, _ = ldm.sample(model, 5, 1, (4, 8, 8), c, x_T, "ddim", unconditional_guidance_scale=3.0,
                           unconditional_conditioning=uc)
    assert model.calls == 5
    assert torch.isfinite(s_plms).all() and torch.isfinite(s_ddim).all()
    assert float((s_plms - s_ddim).abs().max()) > 0
    # CFG with scale 1 == conditional only (plms.py:189-190)
    a, _ = ldm.sample(model, 4, 1, (4, 8, 8), c, x_T, "ddim", unconditional_guidance_scale=1.0,
                      unconditional_conditioning=uc)
    b, _ = ldm.sample(model, 4, 1, (4, 8, 8), c, x_T, "ddim")
    assert torch.equal(a, b)
    # eta != 0 is rejected by PLMS (plms.py:35-36) but allowed by DDIM
    with pytest.raises(ValueError):
        ldm.sample(model, 4, 1, (4, 8, 8), c, x_T, "plms", eta=0.5)
    n = ldm.sample(model, 4, 1, (4, 8, 8), c, x_T, "ddim", eta=0.5,
                   noise_fn=lambda shp: np.random.RandomState(7).randn(*shp).astype(np.float32))[0]
    assert float((n - b).abs().max()) > 0


def test_ddim_single_step_formula():
    """One DDIM step == closed form of plms.py:218-226 with e'=e_t."""
    model = _tiny_model(1)
    x_T = np.random.RandomState(0).randn(2, 4, 8, 8).astype(np.float32)
    c = np.random.RandomState(1).randn(2, 5, 64).astype(np.float32)
    out, _ = ldm.sample(model, 1, 2, (4, 8, 8), c, x_T, "ddim")
    ts = ldm.make_ddim_timesteps(1)
    assert list(ts) == [1]
    e = model.apply_model(torch.tensor(x_T), torch.tensor([1, 1]), torch.tensor(c))
    a_t = model.alphas_cumprod[1]
    a_prev = model.alphas_cumprod[0]
    pred = (torch.tensor(x_T) - np.sqrt(1 - a_t) * e) / np.sqrt(a_t)
    ref = np.sqrt(a_prev) * pred + np.sqrt(1 - a_prev) * e
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=1e-5)


class _RecordingModel:
    """apply_model = 0.1 * x, recording the timesteps it is called with (sampler-option tests)."""
    parameterization = "eps"

    def __init__(self):
        s = ldm.register_schedule()
        self.num_timesteps = s["num_timesteps"]
        self.alphas_cumprod = s["alphas_cumprod"]
        self.alphas_cumprod_prev = s["alphas_cumprod_prev"]
        self.ts = []

    def apply_model(self, x, t, cond):
        self.ts.append(int(t[0]))
        return 0.1 * x


def test_sampler_options_timesteps_subset_and_original_steps():
    """plms.py:134-142, 205-208: `timesteps` keeps a prefix of the DDIM grid; `ddim_use_original_steps` walks every DDPM
    step with the model-level alpha tables; one step of the latter has the closed form of plms.py:218-226."""
    x_T = np.random.RandomState(0).randn(1, 4, 4, 4).astype(np.float32)
    c = np.zeros((1, 2, 8), np.float32)
    m = _RecordingModel()
    ldm.sample(m, 10, 1, (4, 4, 4), c, x_T, "ddim", timesteps=6)
    assert m.ts == [401, 301, 201, 101, 1]              # int(min(6/10, 1) * 10) - 1 = 5 entries of [1, 101, ..., 901]
    m = _RecordingModel()
    ldm.sample(m, 10, 1, (4, 4, 4), c, x_T, "ddim", timesteps=25)   # ratio clamps to 1 -> 9 of the 10
    assert m.ts == [801, 701, 601, 501, 401, 301, 201, 101, 1]
    m = _RecordingModel()
    ldm.sample(m, 10, 1, (4, 4, 4), c, x_T, "plms", timesteps=3, ddim_use_original_steps=True)
    assert m.ts == [2, 1, 1, 0]                         # PLMS: the extra call of the first step uses t_next
    m = _RecordingModel()
    out, _ = ldm.sample(m, 10, 1, (4, 4, 4), c, x_T, "ddim", timesteps=1, ddim_use_original_steps=True)
    a_t, a_prev = m.alphas_cumprod[0], m.alphas_cumprod_prev[0]
    assert a_prev == 1.0
    x = torch.tensor(x_T)
    pred = (x - np.sqrt(1 - a_t) * 0.1 * x) / np.sqrt(a_t)
    np.testing.assert_allclose(out.numpy(), pred.numpy(), atol=1e-6)   # dir term vanishes at a_prev = 1


def test_sampler_option_noise_dropout():
    """plms.py:224-225: dropped entries receive no noise, kept entries noise / (1 - p)."""
    x_T = np.random.RandomState(0).randn(1, 4, 4, 4).astype(np.float32)
    c = np.zeros((1, 2, 8), np.float32)
    noise = np.random.RandomState(1).randn(1, 4, 4, 4).astype(np.float32)
    keep = (np.random.RandomState(2).rand(1, 4, 4, 4) >= 0.5).astype(np.float32)
    run = lambda **kw: ldm.sample(_RecordingModel(), 1, 1, (4, 4, 4), c, x_T, "ddim", eta=1.0,
                                  noise_fn=lambda shp: noise, **kw)[0].numpy()
    base = ldm.sample(_RecordingModel(), 1, 1, (4, 4, 4), c, x_T, "ddim", eta=0.0)[0].numpy()
    full = run()
    drop = run(noise_dropout=0.5, dropout_masks=[keep])
    sig = ldm.make_ddim_sampling_parameters(ldm.register_schedule()["alphas_cumprod"], ldm.make_ddim_timesteps(1), 1.0)[0][0]
    assert sig > 0
    # eta changes the dir coefficient too, so compare the two eta = 1 runs with each other
    np.testing.assert_allclose(drop - (full - sig * noise), sig * noise * keep / 0.5, atol=1e-5)
    assert np.abs(full - base).max() > 0


def test_sampler_hooks_change_the_trajectory_as_specified():
    """score_corrector / quantize_x0 (plms.py:199-201, 218-219) in the oracle: an identity corrector and an identity quantizer
    leave the trajectory unchanged, a grid quantizer puts every pred_x0 on the grid, the corrector sees S + 1 PLMS calls."""
    cfg = dict(in_channels=4, out_channels=4, model_channels=32, attention_resolutions=[1], num_res_blocks=1, channel_mult=[1],
               num_head_channels=32, num_heads=-1, use_spatial_transformer=True, use_linear_in_transformer=True,
               transformer_depth=1, context_dim=32, legacy=False)
    model = ldm.ModelOracle(ldm.UNetOracle(cfg, ldm.init_params(cfg, seed=4)))
    rng = np.random.RandomState(0)
    x_T, c = rng.randn(1, 4, 8, 8).astype(np.float32), rng.randn(1, 3, 32).astype(np.float32)
    base, _ = ldm.sample(model, 4, 1, (4, 8, 8), c, x_T, "plms")

    class Ident:
        n = 0

        def modify_score(self, model, e_t, x, t, c, **kw):
            Ident.n += 1
            return e_t * kw.get("gain", 1.0)

        def quantize(self, z):
            return z, None, None
    model.first_stage_model = Ident()
    same, _ = ldm.sample(model, 4, 1, (4, 8, 8), c, x_T, "plms", score_corrector=Ident(), quantize_x0=True)
    assert torch.equal(same, base) and Ident.n == 5
    diff, _ = ldm.sample(model, 4, 1, (4, 8, 8), c, x_T, "plms", score_corrector=Ident(), corrector_kwargs=dict(gain=0.5))
    assert not torch.equal(diff, base)

    class Grid:
        def quantize(self, z):
            return (z * 2).round() / 2, None, None
    model.first_stage_model = Grid()
    _, inter = ldm.sample(model, 4, 1, (4, 8, 8), c, x_T, "ddim", quantize_x0=True, log_every_t=1)
    for p0 in inter["pred_x0"][1:]:
        assert float((p0 * 2 - (p0 * 2).round()).abs().max()) == 0.0
