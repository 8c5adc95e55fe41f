"""GPU parity for the Taichu-GLIDE path: new kernels one by one, then the planned
Text2ImUNet / SuperResText2ImUNet and the two sampling loops against oracle/glide.py.
Tolerances as in test_unet_gpu.py (fp16 storage / fp32 accumulate vs fp32 oracle)."""
import math

import numpy as np
import pytest
import torch

from _util import check, h16
from oracle import glide as OG
from oracle import ldm as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

TINY = dict(image_size=16, num_channels=64, num_res_blocks=1, channel_mult=(1, 2), num_heads=1, num_head_channels=64,
            num_heads_upsample=-1, attention_resolutions=(1, 2), dropout=0.0, text_ctx=16, xf_width=64, xf_layers=2,
            xf_heads=1, xf_final_ln=True, n_vocab=100, xf_padding=True, diffusion_steps=1000,
            noise_schedule="squaredcos_cap_v2", timestep_respacing="10", use_scale_shift_norm=True,
            resblock_updown=True, use_fp16=True, cache_text_emb=False)
OTINY = dict(OG.BASE_OPTIONS, image_size=16, model_channels=64, num_res_blocks=1, channel_mult=(1, 2),
             attention_resolutions=(1, 2), text_ctx=16, xf_width=64, xf_layers=2, xf_heads=1, n_vocab=100,
             timestep_respacing="10")


@pytest.fixture(scope="module")
def ops():
    from minddiffusion_amd import ops as _ops
    return _ops


def dev16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.float16)


def dev32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, torch.float32)


def nhwc(x):
    b, c, h, w = x.shape
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(b, h * w, c))


def from_nhwc(y, b, h, w):
    return y.reshape(b, h, w, -1).transpose(0, 3, 1, 2)


def test_groupnorm_scaleshift(ops):
    """unet.py:203-208: GN(h) * (1 + scale) + shift, then SiLU."""
    rng = np.random.RandomState(0)
    B, C, H, W = 3, 192, 8, 8
    x = h16(rng.standard_normal((B, C, H, W)) + 0.2)
    g, b = rng.standard_normal(C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    emb = rng.standard_normal((B, 1000)).astype(np.float32)
    sc, sh = emb[:, 100:100 + C], emb[:, 100 + C:100 + 2 * C]
    ref = O.silu(O.group_norm(torch.tensor(x), torch.tensor(g), torch.tensor(b), 1e-5)
                 * (1 + torch.tensor(sc)[:, :, None, None]) + torch.tensor(sh)[:, :, None, None])
    e = dev32(emb)
    y = ops.groupnorm_scaleshift(dev16(nhwc(x)), None, dev32(g), dev32(b), e[:, 100:100 + C],
                                 e[:, 100 + C:100 + 2 * C], 1000, 1e-5, True)
    check("groupnorm_scaleshift", from_nhwc(y.float().cpu().numpy(), B, H, W), ref, rel_l2=1e-3, max_abs=3e-2)


def test_avgpool_and_upsample(ops):
    rng = np.random.RandomState(1)
    B, C, H, W = 2, 64, 6, 8
    x = h16(rng.standard_normal((B, C, H, W)))
    y = ops.avgpool2x2(dev16(nhwc(x)), B, H, W, C)
    ref = torch.nn.functional.avg_pool2d(torch.tensor(x), 2, 2)
    check("avgpool2x2", from_nhwc(y.float().cpu().numpy(), B, H // 2, W // 2), ref, rel_l2=1e-3)
    u = ops.upsample_nearest2x(dev16(nhwc(x)), B, H, W, C)
    np.testing.assert_array_equal(from_nhwc(u.float().cpu().numpy(), B, 2 * H, 2 * W),
                                  O.upsample_nearest2x(torch.tensor(x)).numpy())


def test_text_embed(ops):
    rng = np.random.RandomState(2)
    B, T, Wd, V = 3, 16, 64, 50
    tok = rng.randint(0, V, (B, T)).astype(np.int32)
    mask = (rng.rand(B, T) > 0.3).astype(np.int32)
    te, pe, pad = [h16(rng.standard_normal(s)) for s in ((V, Wd), (T, Wd), (T, Wd))]
    ref = np.where(mask[..., None] != 0, te[tok] + pe[None], np.broadcast_to(pad[None], (B, T, Wd)))
    out = ops.glide_text_embed(torch.tensor(tok, device=DEV), torch.tensor(mask, device=DEV), dev16(te), dev16(pe), dev16(pad))
    check("glide_text_embed", out, ref, rel_l2=1e-3)


def test_superres_input(ops):
    """[x | legacy-bilinear(round((low+1)*127.5)/127.5-1)] (text2im_model.py:214-216, gaussian_diffusion.py:307-313)."""
    rng = np.random.RandomState(3)
    B, S, sl = 2, 32, 8
    x = rng.standard_normal((B, 3, S, S)).astype(np.float32)
    low = np.clip(rng.standard_normal((B, 3, sl, sl)) * 0.5, -1, 1).astype(np.float32)
    q = torch.round((torch.tensor(low) + 1) * 127.5) / 127.5 - 1
    ref = torch.cat([torch.tensor(x), OG.legacy_bilinear(q, S)], 1)
    out = ops.glide_superres_input(dev32(x), dev32(low))
    got = from_nhwc(out.float().cpu().numpy(), B, S, S)
    check("glide_superres_input", got[:, :6], ref, rel_l2=1e-3)
    assert np.all(got[:, 6:] == 0)


@pytest.mark.parametrize("mode,i", [(0, 5), (0, 0), (1, 5), (1, 0)])
def test_glide_step(ops, mode, i):
    """Guider + PMeanVariance + PSample / DDimSample (guider.py:73-86, gaussian_diffusion.py:79-142, 229-254)."""
    from minddiffusion_amd.glide.diffusion_creator import _Schedule
    rng = np.random.RandomState(mode + i)
    B, H, W = 2, 8, 8
    sch = OG.respaced_schedule("squaredcos_cap_v2", 1000, "10")
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    oc = h16(rng.standard_normal((B, 6, H, W)))
    ou = h16(rng.standard_normal((B, 6, H, W)))
    noise = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    scale = 5.0
    if mode == 0:
        eps = torch.tensor(ou[:, :3]) + scale * (torch.tensor(oc[:, :3]) - torch.tensor(ou[:, :3]))
        mean, logvar, x0, _ = OG.p_mean_variance(sch, torch.tensor(x), eps, torch.tensor(oc[:, 3:]), i)
        ref = mean + (0.0 if i == 0 else 1.0) * torch.exp(0.5 * logvar) * torch.tensor(noise)
    else:
        _, _, x0, e2 = OG.p_mean_variance(sch, torch.tensor(x), torch.tensor(oc[:, :3]), torch.tensor(oc[:, 3:]), i)
        ab = float(sch["alphas_cumprod_prev"][i])
        ref = math.sqrt(ab) * x0 + math.sqrt(1 - ab) * e2

    def buf(o):
        t = np.zeros((B, H * W, 8), np.float32)
        t[:, :, :6] = nhwc(o)
        return dev16(t)

    s = _Schedule("squaredcos_cap_v2", 1000, "10")
    xd = dev32(x)
    xn, px = torch.empty_like(xd), torch.empty_like(xd)
    ops.glide_step(xd, buf(oc), buf(ou) if mode == 0 else None, 8, scale, s.coef8(i), mode,
                   1.0 if (mode == 0 and i != 0) else 0.0, dev32(noise) if (mode == 0 and i != 0) else None, xn, px)
    check(f"glide_step_mode{mode}_i{i}", xn, ref, rel_l2=2e-5)
    check(f"glide_step_mode{mode}_i{i}_x0", px, x0, rel_l2=2e-5)


def test_gemm_gelu_and_batch_strided_store(ops):
    """MDX_EPI_GELU (xf.py:52-59) and writing a projection into a token sub-range of [B][ctx+T][C] (unet.py:296-300)."""
    rng = np.random.RandomState(5)
    B, T, ctx, K, N = 2, 64, 16, 64, 128
    a = h16(rng.standard_normal((B * T, K)))
    w = h16(rng.standard_normal((N, K)) / 8)
    bv = rng.standard_normal(N).astype(np.float32)
    wp = ops.pack_gemm_weight(dev16(w))
    out = ops.gemm(dev16(a), wp, N, 1, B * T, 1, K, bias=dev32(bv), epilogue=ops.EPI_GELU)
    check("gemm_gelu", out, O.gelu_tanh(torch.tensor(a) @ torch.tensor(w).T + torch.tensor(bv)), rel_l2=1e-3)
    kbuf = torch.full((B, ctx + T, N), 7.0, dtype=torch.float16, device=DEV)
    ad, bd = dev16(a), dev32(bv)     # descriptors hold raw pointers: keep the tensors alive
    d = ops.make_gemm_desc(ad, wp, N, B, T, 1, K, kbuf[:, ctx:], N, bias=bd, out_bs=(ctx + T) * N)
    ops.gemm_run(d)
    torch.cuda.synchronize()
    ref = (a @ w.T + bv).reshape(B, T, N)
    check("gemm_batch_strided_rows", kbuf[:, ctx:], ref, rel_l2=1e-3)
    assert float((kbuf[:, :ctx] - 7.0).abs().max()) == 0.0     # the text-key rows were not touched


def _build(graph=True):
    from minddiffusion_amd.glide.diffusion_creator import create_model
    params = OG.init_params(OTINY, seed=0)
    net = create_model(**TINY)
    net.use_graph = graph
    net.load_state_dict(params)
    return net, OG.GlideUNetOracle(OTINY, params)


@pytest.mark.parametrize("graph,opts", [(False, {}), (True, {}), (True, {"glide_gn_qkv_fuse": 1}), (True, {"glide_qkv_merge": 0})])
def test_tiny_text2im_unet(graph, opts, ops):
    """opts: the planner options that change the AttentionBlock's launch list (round 6) -- AttentionBlock.norm applied inside the
    merged q | k | v projection (opt-in: measured slower, profiles/r06_glide_gn_qkv_fuse_ab.txt) and the three-launch projection."""
    old = {k: ops.get_option(k) for k in opts}
    for k, v in opts.items():
        ops.set_option(k, v)
    try:
        _tiny_text2im_unet(graph, "".join(f"_{k}{v}" for k, v in opts.items()))
    finally:
        for k, v in old.items():
            ops.set_option(k, v)


def _tiny_text2im_unet(graph, tag):
    net, oracle = _build(graph)
    rng = np.random.RandomState(7)
    B = 4
    x = rng.randn(B, 3, 16, 16).astype(np.float32)
    tok = rng.randint(1, 99, (B, 16)).astype(np.int32)
    mask = np.ones((B, 16), np.int32)
    mask[1, 9:] = 0
    ref = oracle(x, torch.full((B,), 333.0), tok, mask)
    got = net(torch.tensor(x, device=DEV), torch.full((B,), 333.0, device=DEV), torch.tensor(tok, device=DEV),
              torch.tensor(mask, device=DEV))
    check(f"glide_tiny_unet_graph{int(graph)}{tag}", got, ref, rel_l2=5e-3, max_abs=5e-2)


def test_tiny_p_sample_loop():
    """gaussian_p_sample_loop with CFG (main_funcs.py:21-44) on injected random prompts and noises."""
    from minddiffusion_amd.glide.diffusion_creator import init_diffusion_model
    from minddiffusion_amd.glide.main_funcs import gaussian_p_sample_loop
    params = OG.init_params(OTINY, seed=2)
    P, steps = 2, 10
    dm = init_diffusion_model(options=TINY, guidance_scale=3.0, shape=(2 * P, 3, 16, 16), params=params)
    assert dm.num_timesteps == steps
    oracle = OG.GlideUNetOracle(OTINY, params)
    sch = OG.respaced_schedule("squaredcos_cap_v2", 1000, "10")
    rng = np.random.RandomState(11)
    x_T = rng.randn(P, 3, 16, 16).astype(np.float32)
    tok = rng.randint(1, 99, (P, 16)).astype(np.int32)
    mask = np.ones((P, 16), np.int32)
    unc = rng.randint(1, 99, (steps, 16)).astype(np.int32)
    noises = rng.randn(steps, P, 3, 16, 16).astype(np.float32)
    traj = []
    ref = OG.p_sample_loop(oracle, sch, x_T, tok, mask, 3.0, unc, noises, trajectory=traj)
    tok2, mask2 = np.concatenate([tok, tok], 0), np.concatenate([mask, mask], 0)   # second half is overwritten (guider.py:46)
    # (1) teacher-forced along the ORACLE's trajectory (not amplified by the chaotic 10-step recursion):
    #   (a) the guided network output at every x_t carries the tight, well-conditioned tolerance;
    #   (b) x_{t-1}: the first respaced step has alpha-bar ~ 2e-9, so x0 = 2.4e4 * (...) is clipped to +-1 for EVERY
    #       element and an eps difference of 1e-2 flips the sign of a few of them (|d| = 2 * coef1 = 0.34): bound the
    #       bulk (99 % within 2e-2) and the energy, not the flipped elements.  The step kernel itself is exact on
    #       identical network outputs (test_glide_step, 1e-6).
    ones = np.ones((16,), np.int32)
    for k, i in enumerate(range(steps - 1, -1, -1)):
        xk = torch.cat([traj[k], traj[k]], 0)
        t = torch.full((2 * P,), float(sch["timestep_map"][i]))
        tk = torch.cat([torch.tensor(tok), torch.tensor(unc[k])[None].expand(P, -1)], 0)
        mk = torch.cat([torch.tensor(mask), torch.ones(P, 16, dtype=torch.int32)], 0)
        ref_out = oracle(xk, t, tk, mk)
        dout = dm.model.forward_nhwc(xk.to(DEV), t.to(DEV), tk.to(DEV), mk.to(DEV))
        got_out = dout.reshape(2 * P, 16, 16, -1)[..., :6].permute(0, 3, 1, 2).float()
        check(f"glide_tiny_guided_net_out_step{i}", got_out, ref_out, rel_l2=5e-3, max_abs=5e-2)
        got_k, _ = dm(x=xk.to(DEV), timesteps=torch.tensor([i], dtype=torch.int32), token=torch.tensor(tok2),
                      mask=torch.tensor(mask2), random_token=unc[k], random_mask=ones,
                      noise=torch.tensor(noises[k], device=DEV))
        check(f"glide_tiny_p_sample_step{i}", got_k[:P], traj[k + 1], rel_l2=2e-2, abs_q=(0.99, 2e-2))
    # (2) free-running loop: 10 ancestral steps with CFG 3, x0 clipping and exp(logvar/2) noise scaling on random
    # weights compound the per-call fp16 error (2e-3) chaotically -- which elements flip at the clip changes with any
    # change of fp32 summation order -- so bound the energy of the difference and the bulk of the distribution
    x2 = np.concatenate([x_T, x_T], 0)
    got = gaussian_p_sample_loop(dm, torch.tensor(tok2), torch.tensor(mask2), (2 * P, 3, 16, 16), steps, text_ctx=16,
                                 noise=torch.tensor(x2), vocab_len=100, uncond_tokens=list(unc),
                                 step_noises=[torch.tensor(n, device=DEV) for n in noises])[:P]
    check("glide_tiny_p_sample_loop", got, ref, rel_l2=1e-2)      # measured 4.2e-3 (round 2)


def test_loop_tables_are_bit_identical_to_per_step_recomputation():
    """Round 6 (Text2ImUNet.begin_loop): text transformer, encoder_kv projections and the time-embedding chain of ALL steps run
    once per loop; a step copies its rows (mdx_glide_kv_select_f16) and replays the plan's body.  Same kernels on the same values:
    the loop must end on the SAME BITS as the loop that recomputes everything every step -- base model (per-step random
    unconditional prompts, a padded conditional prompt) and up-sampler -- and a step whose prompt is not the announced one must
    fall back to the full evaluation, not serve the table."""
    from minddiffusion_amd.glide import diffusion_creator as DC
    from minddiffusion_amd.glide.main_funcs import ddim_sample_loop, gaussian_p_sample_loop
    params = OG.init_params(OTINY, seed=2)
    P, steps = 2, 10
    dm = DC.init_diffusion_model(options=TINY, guidance_scale=3.0, shape=(2 * P, 3, 16, 16), params=params)
    rng = np.random.RandomState(41)
    x_T = rng.randn(P, 3, 16, 16).astype(np.float32)
    tok = rng.randint(1, 99, (P, 16)).astype(np.int32)
    mask = np.ones((P, 16), np.int32)
    mask[1, 9:] = 0                                                     # a padded prompt: the padding embedding rows
    unc = rng.randint(1, 99, (steps, 16)).astype(np.int32)
    noises = [torch.tensor(n, device=DEV) for n in rng.randn(steps, P, 3, 16, 16).astype(np.float32)]
    tok2, mask2 = torch.tensor(np.concatenate([tok, tok], 0)), torch.tensor(np.concatenate([mask, mask], 0))
    x2 = torch.tensor(np.concatenate([x_T, x_T], 0))
    run = lambda: gaussian_p_sample_loop(dm, tok2, mask2, (2 * P, 3, 16, 16), steps, text_ctx=16, noise=x2, vocab_len=100,
                                         uncond_tokens=list(unc), step_noises=noises)[:P].clone()
    keep = DC._LOOP_TABLES
    try:
        DC._LOOP_TABLES = True
        calls = []
        orig = dm.model.loop_step
        dm.model.loop_step = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        fast = run()
        dm.model.loop_step = orig
        assert len(calls) == steps, f"the loop took the table path on {len(calls)} of {steps} steps"
        DC._LOOP_TABLES = False
        slow = run()
        assert torch.equal(fast, slow), float((fast - slow).abs().max())
        # a step that is handed ANOTHER unconditional prompt than the announced one is evaluated in full
        DC._LOOP_TABLES = True
        dm.begin_loop(tok2, mask2, unc)
        other = (unc[0] + 3) % 90 + 1
        a, _ = dm(x=x2.to(DEV), timesteps=torch.tensor([steps - 1], dtype=torch.int32), token=tok2, mask=mask2,
                  random_token=other, random_mask=np.ones((16,), np.int32), noise=noises[0])
        dm.end_loop()
        b, _ = dm(x=x2.to(DEV), timesteps=torch.tensor([steps - 1], dtype=torch.int32), token=tok2, mask=mask2,
                  random_token=other, random_mask=np.ones((16,), np.int32), noise=noises[0])
        assert torch.equal(a, b)
        # ---- up-sampler (constant prompts; S = 27 timesteps)
        opts = dict(TINY, image_size=32, channel_mult=(1, 1, 2), noise_schedule="linear", timestep_respacing="fast27", low_size=8)
        oopts = dict(OTINY, in_channels=6, image_size=32, channel_mult=(1, 1, 2), noise_schedule="linear",
                     timestep_respacing="fast27")
        sr = DC.init_super_res_model(options=opts, shape=(P, 3, 32, 32), params=OG.init_params(oopts, seed=4))
        x = torch.tensor(rng.randn(P, 3, 32, 32).astype(np.float32) * 0.997)
        low = torch.tensor(np.clip(rng.randn(P, 3, 8, 8) * 0.5, -1, 1).astype(np.float32), device=DEV)
        tk, mk = torch.tensor(tok), torch.tensor(mask)
        up = lambda: ddim_sample_loop(sr, (P, 3, 32, 32), low, tk, mk, 27, noise=x).clone()
        DC._LOOP_TABLES = True
        f2 = up()
        DC._LOOP_TABLES = False
        s2 = up()
        assert torch.equal(f2, s2), float((f2 - s2).abs().max())
    finally:
        DC._LOOP_TABLES = keep


def test_glide_kv_select_kernel():
    """mdx_glide_kv_select_f16: table entries -> the text slots of key / transposed-value buffers, broadcast (entry_per_b = 0) and
    per-row (entry_per_b = 1) forms, several slots of different widths in one launch; everything else in the buffers untouched."""
    from minddiffusion_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    S, B, ctx = 5, 6, 16
    ent, bufs = [], []
    for c, T in ((64, 40), (96, 8), (128, 24)):
        nk = ctx + T
        kt = torch.randn((S, ctx, c), device=DEV, generator=g).half()
        vt = torch.randn((S, c, ctx), device=DEV, generator=g).half()
        kb = torch.randn((B, nk, c), device=DEV, generator=g).half()
        vb = torch.randn((B, c, nk), device=DEV, generator=g).half()
        ent.append((kt, kb, ctx * c * 2, nk * c * 2, c * 2, c * 2, ctx, c * 2))
        ent.append((vt, vb, c * ctx * 2, c * nk * 2, ctx * 2, nk * 2, c, ctx * 2))
        bufs.append((kt, vt, kb, vb, kb.clone(), vb.clone()))
    slots, n = ops.glide_kv_slots(ent, DEV)
    ops.glide_kv_select(slots, n, 3, 0, 2, 3)           # rows 2..4 <- entry 3
    ops.glide_kv_select(slots, n, 1, 1, 0, 2, blocks_per_copy=3)     # rows 0, 1 <- entries 1, 2
    torch.cuda.synchronize()
    for kt, vt, kb, vb, kb0, vb0 in bufs:
        ek, ev = kb0.clone(), vb0.clone()
        ek[2:5, :ctx] = kt[3]
        ev[2:5, :, :ctx] = vt[3]
        ek[0, :ctx], ek[1, :ctx] = kt[1], kt[2]
        ev[0, :, :ctx], ev[1, :, :ctx] = vt[1], vt[2]
        assert torch.equal(kb, ek) and torch.equal(vb, ev)


def test_tiny_p_sample_loop_60_steps():
    """The benchmarked LENGTH of the base loop (main_funcs.py:21-44 with timestep_respacing "60"): sixty guided ancestral
    steps with a fresh random unconditional prompt and fresh noise per step, on the tiny model (the oracle runs 120 rows)."""
    from minddiffusion_amd.glide.diffusion_creator import init_diffusion_model
    from minddiffusion_amd.glide.main_funcs import gaussian_p_sample_loop
    params = OG.init_params(OTINY, seed=6)
    P, steps = 2, 60
    dm = init_diffusion_model(options=dict(TINY, timestep_respacing="60"), guidance_scale=3.0, shape=(2 * P, 3, 16, 16),
                              params=params)
    assert dm.num_timesteps == steps
    oracle = OG.GlideUNetOracle(dict(OTINY, timestep_respacing="60"), params)
    sch = OG.respaced_schedule("squaredcos_cap_v2", 1000, "60")
    rng = np.random.RandomState(31)
    x_T = rng.randn(P, 3, 16, 16).astype(np.float32)
    tok = rng.randint(1, 99, (P, 16)).astype(np.int32)
    mask = np.ones((P, 16), np.int32)
    unc = rng.randint(1, 99, (steps, 16)).astype(np.int32)
    noises = rng.randn(steps, P, 3, 16, 16).astype(np.float32)
    ref = OG.p_sample_loop(oracle, sch, x_T, tok, mask, 3.0, unc, noises)
    tok2, mask2 = np.concatenate([tok, tok], 0), np.concatenate([mask, mask], 0)
    got = gaussian_p_sample_loop(dm, torch.tensor(tok2), torch.tensor(mask2), (2 * P, 3, 16, 16), steps, text_ctx=16,
                                 noise=torch.tensor(np.concatenate([x_T, x_T], 0)), vocab_len=100, uncond_tokens=list(unc),
                                 step_noises=[torch.tensor(n, device=DEV) for n in noises])[:P]
    check("glide_tiny_p_sample_loop_60", got, ref, rel_l2=1e-2)


def test_tiny_superres_unet_and_ddim_loop():
    from minddiffusion_amd.glide.diffusion_creator import init_super_res_model
    from minddiffusion_amd.glide.main_funcs import ddim_sample_loop
    opts = dict(TINY, image_size=32, channel_mult=(1, 1, 2), noise_schedule="linear", timestep_respacing="fast27",
                low_size=8)
    oopts = dict(OTINY, in_channels=6, image_size=32, channel_mult=(1, 1, 2), noise_schedule="linear",
                 timestep_respacing="fast27")
    params = OG.init_params(oopts, seed=4)
    P = 2
    sr = init_super_res_model(options=opts, shape=(P, 3, 32, 32), params=params)
    assert sr.num_timesteps == 27
    oracle = OG.GlideUNetOracle(oopts, params)
    rng = np.random.RandomState(13)
    x = rng.randn(P, 3, 32, 32).astype(np.float32)
    low = np.clip(rng.randn(P, 3, 8, 8) * 0.5, -1, 1).astype(np.float32)
    tok = rng.randint(1, 99, (P, 16)).astype(np.int32)
    mask = np.ones((P, 16), np.int32)
    lowq = torch.round((torch.tensor(low) + 1) * 127.5) / 127.5 - 1
    ref1 = oracle(x, torch.full((P,), 500.0), tok, mask, low_res=lowq)
    got1 = sr.model(torch.tensor(x, device=DEV), torch.full((P,), 500.0, device=DEV), torch.tensor(low, device=DEV),
                    torch.tensor(tok, device=DEV), torch.tensor(mask, device=DEV))
    check("glide_tiny_superres_unet", got1, ref1, rel_l2=5e-3, max_abs=5e-2)
    sch = OG.respaced_schedule("linear", 1000, "fast27")
    ref = OG.ddim_sample_loop(oracle, sch, x * 0.997, low, tok, mask)
    got = ddim_sample_loop(sr, (P, 3, 32, 32), torch.tensor(low, device=DEV), torch.tensor(tok), torch.tensor(mask), 27,
                           noise=torch.tensor(x * 0.997))
    check("glide_tiny_ddim_superres_loop", got, ref, rel_l2=1e-2)  # the benchmarked 27 steps; measured 3.4e-3 (round 2)
    # round 5: the loop hands the same token objects to every step, so the text transformer (the plan's prefix) ran once, not 27
    # times.  The cached form must be BIT-identical to recomputing it, and a changed prompt must not be served the old prefix.
    net = sr.model
    Pl = net._plans[(P, 32, 32)]
    assert 0 < Pl.n_text < len(Pl.main) and all(m["text"] for m in Pl.meta[:Pl.n_text]) and not any(m["text"] for m in Pl.meta[Pl.n_text:])
    dx, dl = torch.tensor(x, device=DEV), torch.tensor(low, device=DEV)
    dt, dm_ = torch.tensor(tok, device=DEV), torch.tensor(mask, device=DEV)
    t5 = torch.full((P,), 300.0, device=DEV)
    full = net.forward_nhwc(dx, t5, dt, dm_, low_res=dl).clone()                        # no epoch: everything recomputed
    a = net.forward_nhwc(dx, t5, dt, dm_, low_res=dl, text_epoch=("t", 1)).clone()      # new epoch: recomputed, then remembered
    b = net.forward_nhwc(dx, t5, dt, dm_, low_res=dl, text_epoch=("t", 1)).clone()      # same epoch: prefix skipped
    assert torch.equal(full, a) and torch.equal(a, b)
    tok2 = tok.copy()
    tok2[:, 3] = (tok2[:, 3] + 7) % 90 + 1
    c = net.forward_nhwc(dx, t5, torch.tensor(tok2, device=DEV), dm_, low_res=dl, text_epoch=("t", 2)).clone()
    assert not torch.equal(c, a)
    ref2 = net.forward_nhwc(dx, t5, torch.tensor(tok2, device=DEV), dm_, low_res=dl).clone()
    assert torch.equal(c, ref2)
    # the sampler object notices an in-place change of the caller's token tensor (version counter) and a new numpy array (digest)
    tk = torch.tensor(tok)
    s1, _ = sr(x=dx, timesteps=torch.tensor([5]), token=tk, mask=torch.tensor(mask), samples=dl)
    e1 = sr._epoch
    s1b, _ = sr(x=dx, timesteps=torch.tensor([5]), token=tk, mask=sr._text[1], samples=dl)
    assert sr._epoch == e1 and torch.equal(s1, s1b)
    tk[:, 3] = torch.tensor(tok2[:, 3])
    s2, _ = sr(x=dx, timesteps=torch.tensor([5]), token=tk, mask=sr._text[1], samples=dl)
    assert sr._epoch == e1 + 1 and not torch.equal(s2, s1)


def test_full_size_glide_base_single_step():
    """BASELINE config 4 building block: the full Taichu-GLIDE base model (192 ch x (1,2,3,4), 16-layer text
    transformer, 385 M parameters) for ONE guided evaluation (P = 1 -> UNet batch 2) vs the fp32 CPU oracle."""
    import os
    from minddiffusion_amd.glide.default_options import model_and_diffusion_defaults
    from minddiffusion_amd.glide.diffusion_creator import create_model
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    params = OG.init_params(OG.BASE_OPTIONS, seed=0)
    net = create_model(**model_and_diffusion_defaults())
    net.load_state_dict(params)
    oracle = OG.GlideUNetOracle(OG.BASE_OPTIONS, params)
    rng = np.random.RandomState(5)
    x = np.repeat(rng.randn(1, 3, 64, 64).astype(np.float32), 2, 0)
    tok = rng.randint(1, 50000, (2, 128)).astype(np.int32)
    mask = np.ones((2, 128), np.int32)
    mask[0, 40:] = 0
    ref = oracle(x, torch.full((2,), 982.0), tok, mask)
    got = net(torch.tensor(x, device=DEV), torch.full((2,), 982.0, device=DEV), torch.tensor(tok, device=DEV),
              torch.tensor(mask, device=DEV))
    check("glide_full_base_single_step", got, ref, rel_l2=5e-3, max_abs=5e-2)
