"""GPU end-to-end parity: UNetModel / PLMSSampler / DDIMSampler (HIP path through the C-ABI) against the
fp32 CPU oracle on identical seeded weights and inputs.

Tolerances (fp16 storage + fp32 accumulation vs an all-fp32 oracle):
  single UNet call : rel-L2 <= 5e-3 at the latent level
  5/10-step trajectory: rel-L2 <= 1e-2, max|d| <= 1e-2 * max|ref| (synthetic weights + CFG 7.5 give |x| ~ 50)
"""
import numpy as np
import pytest
import torch

from _util import check
from oracle import ldm as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tiny_cfg():
    from minddiffusion_amd.configs import TINY_UNET
    return dict(TINY_UNET)


def _oracle_cfg(cfg):
    c = dict(cfg)
    c.setdefault("num_heads", -1)
    c.setdefault("num_head_channels", -1)
    return c


def _build(cfg, params, graph):
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    net = UNetModel(**cfg)
    net.use_graph = graph
    net.load_state_dict(params)
    return net


def _inputs(B, H, W, T, D, seed=0):
    rng = np.random.RandomState(seed)
    x = rng.randn(B, 4, H, W).astype(np.float32)
    ctx = rng.randn(B, T, D).astype(np.float32)
    return x, ctx


@pytest.mark.parametrize("graph", [False, True])
def test_tiny_unet_forward(graph):
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=0)
    net = _build(cfg, params, graph)
    oracle = O.UNetOracle(_oracle_cfg(cfg), params)
    # (2, 12, 12) and (1, 10, 14): 36 / 35-token levels -- token counts that are not multiples of 8 (the deepest level of a
    # 384- or 320 x 448-pixel run): V^T rows are padded in a dedicated zeroed buffer and stored element-wise (round 5)
    for (B, H, W, T, t) in ((2, 8, 8, 5, 981.0), (1, 16, 16, 77, 21.0), (3, 8, 12, 9, 500.0), (2, 12, 12, 7, 333.0),
                            (1, 10, 14, 11, 77.0)):
        x, ctx = _inputs(B, H, W, T, cfg["context_dim"], seed=B + H)
        ts = np.full((B,), t, np.float32)
        ref = oracle(x, torch.tensor(ts), ctx)
        got = net(torch.tensor(x, device=DEV), torch.tensor(ts, device=DEV), torch.tensor(ctx, device=DEV))
        check(f"tiny_unet_forward_graph{int(graph)}_B{B}_{H}x{W}_T{T}", got, ref, rel_l2=5e-3, max_abs=5e-2)
        # replay determinism: same inputs -> bit-identical output
        got2 = net(torch.tensor(x, device=DEV), torch.tensor(ts, device=DEV), torch.tensor(ctx, device=DEV))
        assert torch.equal(got, got2)


VARIANTS = {
    # the UNetModel constructor switches outside the shipped YAMLs (openaimodel.py:283-531), each against the oracle
    "depth2": dict(transformer_depth=2),
    "scale_shift": dict(use_scale_shift_norm=True),
    "updown": dict(resblock_updown=True),
    "updown_scale_shift_depth2": dict(resblock_updown=True, use_scale_shift_norm=True, transformer_depth=2),
    "pool_resample": dict(conv_resample=False),
    "class_cond": dict(num_classes=10),
    "codebook_ids": dict(n_embed=24),
    "wukong_style_depth2": None,
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("graph", [False, True])
def test_unet_constructor_variants(name, graph):
    from minddiffusion_amd.configs import SMALL_WUKONG_UNET
    from minddiffusion_amd._lib import MdxError
    if name == "wukong_style_depth2":      # conv proj_in / proj_out, 8 heads (head dims 40 / 80: no LayerNorm fold), depth 2
        cfg = dict(SMALL_WUKONG_UNET, transformer_depth=2)
    else:
        cfg = dict(_tiny_cfg(), **VARIANTS[name])
    ocfg = _oracle_cfg(cfg)
    params = O.init_params(ocfg, seed=11)
    net = _build(cfg, params, graph)
    oracle = O.UNetOracle(ocfg, params)
    for (B, H, W, T) in ((2, 8, 8, 5), (1, 16, 16, 77)):
        x, ctx = _inputs(B, H, W, T, cfg["context_dim"], seed=B + H)
        ts = np.linspace(7.0, 940.0, B).astype(np.float32)
        kw = {}
        if cfg.get("num_classes"):
            kw["y"] = [(3 * b + 1) % cfg["num_classes"] for b in range(B)]
        ref = oracle(x, torch.tensor(ts), ctx, **kw)
        dkw = {k: torch.tensor(v, device=DEV) for k, v in kw.items()}
        got = net(torch.tensor(x, device=DEV), torch.tensor(ts, device=DEV), torch.tensor(ctx, device=DEV), **dkw)
        assert tuple(got.shape) == tuple(ref.shape)
        check(f"tiny_unet_variant_{name}_graph{int(graph)}_B{B}_{H}x{W}", got, ref, rel_l2=5e-3, max_abs=5e-2)
    if cfg.get("num_classes"):
        x, ctx = _inputs(2, 8, 8, 5, cfg["context_dim"])
        args = (torch.tensor(x, device=DEV), torch.tensor([1.0, 2.0], device=DEV), torch.tensor(ctx, device=DEV))
        with pytest.raises(MdxError, match="class-conditional"):
            net(*args)
        with pytest.raises(MdxError, match="outside"):
            net(*args, y=torch.tensor([0, 10]))
        with pytest.raises(MdxError, match="timestep-only"):
            net.time_embedding_table([1.0])


def test_zero_init_unet_is_exactly_zero():
    """Structural KAT: the reference constructor zero-inits out convs / proj_out (zero_module) => output == 0."""
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=1, zero_init=True)
    net = _build(cfg, params, False)
    x, ctx = _inputs(2, 8, 8, 5, cfg["context_dim"])
    got = net(torch.tensor(x, device=DEV), torch.full((2,), 981.0, device=DEV), torch.tensor(ctx, device=DEV))
    assert float(got.abs().max()) == 0.0


def test_context_cache_invalidation():
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=2)
    net = _build(cfg, params, True)
    oracle = O.UNetOracle(_oracle_cfg(cfg), params)
    x, ctx = _inputs(2, 8, 8, 7, cfg["context_dim"], seed=5)
    xd, td = torch.tensor(x, device=DEV), torch.full((2,), 401.0, device=DEV)
    c1 = torch.tensor(ctx, device=DEV)
    a = net(xd, td, c1)
    # different tensor, different content (a pure permutation of the tokens would not do: cross-attention is invariant to it,
    # and the two outputs then differ by fp16 rounding noise at best)
    ctx2 = (0.8 * ctx[:, ::-1] + 0.1).astype(np.float32)
    c2 = torch.tensor(ctx2, device=DEV)
    b = net(xd, td, c2)
    ref_b = oracle(x, torch.full((2,), 401.0), ctx2)
    check("context_cache_second_context", b, ref_b, rel_l2=5e-3)
    c1.mul_(0.5)                                          # in-place update of the first tensor
    c = net(xd, td, c1)
    ref_c = oracle(x, torch.full((2,), 401.0), ctx * 0.5)
    check("context_cache_inplace_update", c, ref_c, rel_l2=5e-3)
    assert float((a - b).abs().max()) > 1e-3


@pytest.mark.parametrize("graph", [False, True])
def test_context_length_may_change_between_calls(graph):
    """The cross-attention that rides on its query projection (mdx_gemm_desc.xattn_len, round 6) and the attention launches read the
    context length of the CURRENT call: 7, then 12, then 3 text tokens through one plan (the captured graph is rebuilt when the length
    changes), each against the oracle; and the planner option off gives the same bits."""
    from minddiffusion_amd import ops
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=4)
    net = _build(cfg, params, graph)
    oracle = O.UNetOracle(_oracle_cfg(cfg), params)
    assert any("+cross-attention" in m["info"] for m in net._plan(2, 8, 8).meta), "the tiny UNet (head dim 64, 64 tokens) should fuse"
    outs = []
    for T in (7, 12, 3, 12):
        x, ctx = _inputs(2, 8, 8, T, cfg["context_dim"], seed=10 + T)
        got = net(torch.tensor(x, device=DEV), torch.full((2,), 250.0, device=DEV), torch.tensor(ctx, device=DEV))
        check(f"context_length_{T}_graph{int(graph)}", got, oracle(x, torch.full((2,), 250.0), ctx), rel_l2=5e-3)
        outs.append(got.clone())
    old = ops.get_option("unet_xattn_fuse")
    ops.set_option("unet_xattn_fuse", 0)
    try:
        net0 = _build(cfg, params, graph)
        x, ctx = _inputs(2, 8, 8, 12, cfg["context_dim"], seed=22)
        two = net0(torch.tensor(x, device=DEV), torch.full((2,), 250.0, device=DEV), torch.tensor(ctx, device=DEV))
    finally:
        ops.set_option("unet_xattn_fuse", old)
    assert torch.equal(two, outs[-1]) and torch.equal(outs[1], outs[-1]), "fused cross-attention changes the UNet's bits"


@pytest.mark.parametrize("name", ["tiny", "wukong_style", "wukong_style_64x64", "depth2_updown", "no_attention_at_level_0"])
@pytest.mark.parametrize("graph", [False, True])
def test_guidance_duplicate_prefix(name, graph):
    """A guidance batch is cat([x] * 2) with [uncond ; cond] contexts (plms.py:192-195): forward_nhwc(cfg_dup=True) runs conv_in .. the
    first self-attention from the HALF-batch plan and writes its live tensors to both halves (UNetModel._dup_body, round 6).  Checked
    against the oracle at even and odd half batches, with the sampler's time-embedding rows, across a context-length change, and
    next to the plain evaluation of the same inputs (other tiles at half the batch: equal to rounding, not to the bit)."""
    from minddiffusion_amd import ops
    from minddiffusion_amd.configs import SMALL_WUKONG_UNET
    cfg = {"tiny": _tiny_cfg(), "wukong_style": dict(SMALL_WUKONG_UNET), "wukong_style_64x64": dict(SMALL_WUKONG_UNET),
           "depth2_updown": dict(_tiny_cfg(), transformer_depth=2, resblock_updown=True),
           "no_attention_at_level_0": dict(_tiny_cfg(), attention_resolutions=[2])}[name]
    ocfg = _oracle_cfg(cfg)
    params = O.init_params(ocfg, seed=21)
    net = _build(cfg, params, graph)
    oracle = O.UNetOracle(ocfg, params)
    old = ops.get_option("unet_cfg_dup")
    ops.set_option("unet_cfg_dup", 2)
    try:
        shapes = ((4, 8, 8, 7, False), (6, 8, 8, 12, True), (2, 16, 16, 77, True), (4, 8, 8, 5, False))
        if name == "wukong_style_64x64":
            # 4096 tokens at C = 320: the batch-2 plan runs the fused SpatialTransformer tail (256 row blocks), the batch-1 plan the
            # unfused launches (128 < 192, tail_rows): the two plans splice behind the self-attention, the one point both have
            shapes = ((2, 64, 64, 77, True),)
            assert net._plan(2, 64, 64).tails and not net._plan(1, 64, 64).tails
        for (B, H, W, T, use_table) in shapes:
            x, ctx = _inputs(B, H, W, T, cfg["context_dim"], seed=3 * B + T)
            x[B // 2:] = x[:B // 2]
            ts = np.full((B,), 437.0, np.float32)
            xd, td, cd = torch.tensor(x, device=DEV), torch.tensor(ts, device=DEV), torch.tensor(ctx, device=DEV)
            kw = {"temb": net.time_embedding_table(td[:1])[0]} if use_table else {}
            got = ops.nhwc_to_nchw(net.forward_nhwc(xd, td, cd, cfg_dup=True, **kw), net.final_channels, H, W).clone()
            P = net._plan(B, H, W)
            if name == "no_attention_at_level_0":
                assert P.ck is None and net._dup_body(P) is None
            else:
                assert P.dup_body is not None and (P.dup_graph is not None) == graph
                assert sum(m["flops"] for m in P.dup_meta) < 0.99 * sum(m["flops"] for m in P.meta)
            check(f"guidance_dup_{name}_graph{int(graph)}_B{B}_{H}x{W}_T{T}", got, oracle(x, torch.tensor(ts), ctx), rel_l2=5e-3,
                  max_abs=5e-2)
            plain = ops.nhwc_to_nchw(net.forward_nhwc(xd, td, cd, **kw), net.final_channels, H, W)
            rel = float((got - plain).norm() / plain.norm())
            assert rel < 2e-3, rel
            again = ops.nhwc_to_nchw(net.forward_nhwc(xd, td, cd, cfg_dup=True, **kw), net.final_channels, H, W)
            assert torch.equal(got, again), "replay of the guidance-duplicate body is not deterministic"
        if name == "tiny":      # the bring-up check (option unet_cfg_dup_check): halves that differ are refused
            from minddiffusion_amd._lib import MdxError
            ops.set_option("unet_cfg_dup_check", 1)
            try:
                net.forward_nhwc(xd, td, cd, cfg_dup=True)
                bad = xd.clone()
                bad[-1, 0, 0, 0] += 1.0
                with pytest.raises(MdxError, match="do not carry the same"):
                    net.forward_nhwc(bad, td, cd, cfg_dup=True)
                tb = td.clone()
                tb[-1] += 1.0
                with pytest.raises(MdxError, match="do not carry the same"):
                    net.forward_nhwc(xd, tb, cd, cfg_dup=True)
            finally:
                ops.set_option("unet_cfg_dup_check", 0)
        # below the option's batch, and with the option off, the call is the plain evaluation
        ops.set_option("unet_cfg_dup", 8)
        assert net._dup_body(net._plan(shapes[0][0], shapes[0][1], shapes[0][2])) is None
    finally:
        ops.set_option("unet_cfg_dup", old)


@pytest.mark.parametrize("sampler,S,scale", [("plms", 5, 3.0), ("ddim", 5, 3.0), ("ddim", 4, 1.0), ("plms", 10, 7.5)])
def test_tiny_sampler_trajectory(sampler, S, scale):
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=3)
    net = _build(cfg, params, True)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    omodel = O.ModelOracle(O.UNetOracle(_oracle_cfg(cfg), params))
    B, H, W, T = 2, 8, 8, 6
    x_T = np.random.RandomState(42).randn(B, 4, H, W).astype(np.float32)
    c = np.random.RandomState(1).randn(B, T, cfg["context_dim"]).astype(np.float32)
    uc = np.repeat(np.random.RandomState(2).randn(1, T, cfg["context_dim"]).astype(np.float32), B, 0)
    ref, ref_inter = O.sample(omodel, S, B, (4, H, W), c, x_T, sampler, unconditional_guidance_scale=scale,
                              unconditional_conditioning=uc)
    cls = PLMSSampler if sampler == "plms" else DDIMSampler
    calls = []
    got, inter = cls(model).sample(S, B, (4, H, W), conditioning=torch.tensor(c, device=DEV),
                                   x_T=torch.tensor(x_T, device=DEV), unconditional_guidance_scale=scale,
                                   unconditional_conditioning=torch.tensor(uc, device=DEV), verbose=False,
                                   callback=lambda i: calls.append(i))
    assert calls == list(range(S))
    assert len(inter["x_inter"]) == len(ref_inter["x_inter"])
    check(f"tiny_{sampler}_S{S}_scale{scale}", got, ref, rel_l2=1e-2, max_rel=1e-2)
    check(f"tiny_{sampler}_S{S}_scale{scale}_pred_x0", inter["pred_x0"][-1], ref_inter["pred_x0"][-1], rel_l2=2e-2)


@pytest.mark.parametrize("S,scale", [(10, 7.5), (20, 7.5), (15, 1.0)])
def test_tiny_dpm_solver_trajectory(S, scale):
    """DPMSolverSampler (DPM-Solver++ 2M, time_uniform, fractional UNet timesteps, S model calls)
    against oracle/dpm_solver.py on the same tiny UNet; S < 15 takes the lower_order_final branch."""
    from oracle import dpm_solver as OD
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.dpm_solver import DPMSolverSampler
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=3)
    net = _build(cfg, params, True)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    omodel = O.ModelOracle(O.UNetOracle(_oracle_cfg(cfg), params))
    B, H, W, T = 2, 8, 8, 6
    x_T = np.random.RandomState(42).randn(B, 4, H, W).astype(np.float32)
    c = np.random.RandomState(1).randn(B, T, cfg["context_dim"]).astype(np.float32)
    uc = np.repeat(np.random.RandomState(2).randn(1, T, cfg["context_dim"]).astype(np.float32), B, 0)
    ref, solver = OD.sample(omodel, S, B, (4, H, W), c, x_T, unconditional_guidance_scale=scale,
                            unconditional_conditioning=uc)
    assert solver.nfe == S
    calls, x0s = [], []
    got, inter = DPMSolverSampler(model).sample(S, B, (4, H, W), conditioning=torch.tensor(c, device=DEV),
                                                x_T=torch.tensor(x_T, device=DEV), unconditional_guidance_scale=scale,
                                                unconditional_conditioning=torch.tensor(uc, device=DEV), verbose=False,
                                                callback=lambda i: calls.append(i),
                                                img_callback=lambda x0, i: x0s.append(float(x0.abs().max())))
    assert inter is None and calls == list(range(S)) and len(x0s) == S and all(np.isfinite(x0s))
    check(f"tiny_dpm_solver_S{S}_scale{scale}", got, ref, rel_l2=1e-2, max_rel=1e-2)


@pytest.mark.parametrize("blend", [False, True])
def test_tiny_inpaint_hybrid_trajectory(blend):
    """Wukong inpainting path: LatentInpaintDiffusion with 'hybrid' conditioning -- the UNet sees
    cat(x, mask, masked-image latent) = 9 channels (WK ddpm.py:339-371, inpaint.py:65-106) -- driven by PLMS with dict
    conditioning; `blend` additionally exercises the mask / x0 blend of plms.py:153-157 with injected noise."""
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentInpaintDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    cfg = dict(_tiny_cfg(), in_channels=9)
    params = O.init_params(_oracle_cfg(cfg), seed=6)
    net = _build(cfg, params, True)
    model = LatentInpaintDiffusion(unet_config=net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    assert model.model.conditioning_key == "hybrid" and model.masked_image_key in model.concat_keys
    omodel = O.ModelOracle(O.UNetOracle(_oracle_cfg(cfg), params))
    B, H, W, T, S = 2, 8, 8, 6, 5
    rng = np.random.RandomState(8)
    x_T = rng.randn(B, 4, H, W).astype(np.float32)
    c = rng.randn(B, T, cfg["context_dim"]).astype(np.float32)
    uc = np.repeat(rng.randn(1, T, cfg["context_dim"]).astype(np.float32), B, 0)
    m = (rng.rand(B, 1, H, W) > 0.5).astype(np.float32)                      # inpaint.py:76-78: resized mask
    masked_latent = rng.randn(B, 4, H, W).astype(np.float32)                 # encode_first_stage(masked image)
    c_cat = np.concatenate([m, masked_latent], 1)
    noises = [rng.randn(B, 4, H, W).astype(np.float32) for _ in range(S)]
    kw = dict(mask=m, x0=masked_latent, blend_noises=noises) if blend else dict(x0=masked_latent)
    ref, _ = O.sample(omodel, S, B, (4, H, W), {"c_concat": c_cat, "c_crossattn": c}, x_T, "plms",
                      unconditional_guidance_scale=7.5,
                      unconditional_conditioning={"c_concat": c_cat, "c_crossattn": uc},
                      **({k: v for k, v in kw.items() if k != "x0"} if not blend else kw))
    dev = lambda a: torch.tensor(a, device=DEV)
    got, _ = PLMSSampler(model).sample(S, B, (4, H, W), conditioning={"c_concat": dev(c_cat), "c_crossattn": dev(c)},
                                       x_T=dev(x_T), unconditional_guidance_scale=7.5,
                                       unconditional_conditioning={"c_concat": dev(c_cat), "c_crossattn": dev(uc)},
                                       verbose=False, **{k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
    check(f"tiny_inpaint_hybrid_blend{int(blend)}", got, ref, rel_l2=1e-2, max_rel=1e-2)
    # the single-call surface too: keyword form of WK ddpm.py:276-278
    t = torch.full((B,), 500.0, device=DEV)
    e = model.apply_model(dev(x_T), t, c_concat=[dev(c_cat)], c_crossattn=[dev(c)])
    eo = omodel.apply_model(torch.tensor(x_T), torch.full((B,), 500.0), {"c_concat": torch.tensor(c_cat), "c_crossattn": torch.tensor(c)})
    check("tiny_inpaint_apply_model", e, eo, rel_l2=5e-3, max_abs=5e-2)


def test_tiny_inpaint_hybrid_unconditional_c_concat_may_differ():
    """WK plms.py:191-201 concatenates [uncond[k]; cond[k]] for every dict key: an unconditional c_concat that differs from the
    conditional one (inpaint.py passes the same tensor in both, so nothing else exercises it) must reach the unconditional half of
    the CFG batch.  Oracle = the same key-by-key concatenation; also checked: the result is NOT what the shared-c_concat run gives."""
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentInpaintDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    cfg = dict(_tiny_cfg(), in_channels=9)
    params = O.init_params(_oracle_cfg(cfg), seed=6)
    net = _build(cfg, params, True)
    model = LatentInpaintDiffusion(unet_config=net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    omodel = O.ModelOracle(O.UNetOracle(_oracle_cfg(cfg), params))
    B, H, W, T, S = 2, 8, 8, 6, 5
    rng = np.random.RandomState(18)
    x_T = rng.randn(B, 4, H, W).astype(np.float32)
    c = rng.randn(B, T, cfg["context_dim"]).astype(np.float32)
    uc = np.repeat(rng.randn(1, T, cfg["context_dim"]).astype(np.float32), B, 0)
    c_cat = np.concatenate([(rng.rand(B, 1, H, W) > 0.5).astype(np.float32), rng.randn(B, 4, H, W).astype(np.float32)], 1)
    uc_cat = np.concatenate([np.ones((B, 1, H, W), np.float32), np.zeros((B, 4, H, W), np.float32)], 1)
    ref, _ = O.sample(omodel, S, B, (4, H, W), {"c_concat": c_cat, "c_crossattn": c}, x_T, "plms", unconditional_guidance_scale=7.5,
                      unconditional_conditioning={"c_concat": uc_cat, "c_crossattn": uc})
    dev = lambda a: torch.tensor(a, device=DEV)
    run = lambda ucc: PLMSSampler(model).sample(S, B, (4, H, W), conditioning={"c_concat": dev(c_cat), "c_crossattn": dev(c)},
                                                x_T=dev(x_T), unconditional_guidance_scale=7.5,
                                                unconditional_conditioning={"c_concat": dev(ucc), "c_crossattn": dev(uc)},
                                                verbose=False)[0]
    got, same = run(uc_cat), run(c_cat)
    check("tiny_inpaint_hybrid_uncond_c_concat_differs", got, ref, rel_l2=1e-2, max_rel=1e-2)
    assert float((got - same).norm() / same.norm()) > 5e-2, "the unconditional c_concat did not reach the unconditional half"


def test_groupnorm_folds_into_proj_in_where_the_plan_says_so():
    """Planner option unet_gn_proj_fuse: SpatialTransformer.norm moves into proj_in (one launch less per transformer block, no
    normalised copy).  The fused plan has fewer ops than the unfused one, at least one launch carries "+groupnorm(in)", and both
    agree with the oracle (the fold adds fp16 scale / shift rounding: same bounds as every UNet test)."""
    from minddiffusion_amd import ops
    cfg = dict(_tiny_cfg(), model_channels=128, attention_resolutions=[1, 2])     # 128 / 256-wide transformers: K tiles of 64
    params = O.init_params(_oracle_cfg(cfg), seed=13)
    oracle = O.UNetOracle(_oracle_cfg(cfg), params)
    B, H, W, T = 2, 32, 32, 9
    x, ctx = _inputs(B, H, W, T, cfg["context_dim"], seed=5)
    ts = np.full((B,), 321.0, np.float32)
    ref = oracle(x, torch.tensor(ts), ctx)
    keep = {k: ops.get_option(k) for k in ("unet_gn_proj_fuse", "unet_st_head", "unet_st_tail")}
    counts = {}
    try:
        ops.set_option("unet_st_head", 0)        # the transformers take the launch-per-op path the fold applies to
        ops.set_option("unet_st_tail", 0)
        for mode in (0, 64):
            ops.set_option("unet_gn_proj_fuse", mode)
            net = _build(cfg, params, True)
            got = net(torch.tensor(x, device=DEV), torch.tensor(ts, device=DEV), torch.tensor(ctx, device=DEV))
            check(f"unet_gn_proj_fuse{mode}", got, ref, rel_l2=5e-3, max_abs=5e-2)
            P = net._plans[(B, H, W)]
            counts[mode] = (len(P.main), sum("+groupnorm(in)" in m["info"] for m in P.meta))
    finally:
        for k, v in keep.items():
            ops.set_option(k, v)
    assert counts[0][1] == 0 and counts[64][1] >= 1, counts
    assert counts[64][0] == counts[0][0] - counts[64][1], counts


def _selfctx_cfg(**kw):
    """A UNet the reference's `context=None` call type-checks on: attention only where the transformer width equals context_dim
    (WK attention.py:133 `context = default(context, x)` feeds the block's own tokens to to_k / to_v = Dense(context_dim, inner))."""
    return dict(dict(image_size=8, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[2], num_res_blocks=1,
                     channel_mult=[1, 2], num_head_channels=64, use_spatial_transformer=True, use_linear_in_transformer=True,
                     transformer_depth=1, context_dim=128, legacy=False), **kw)


@pytest.mark.parametrize("key", [None, "concat", "adm"])
def test_diffusion_wrapper_keys_without_context(key):
    """DiffusionWrapper conditioning keys None / 'concat' / 'adm' (WK ddpm.py:360-377): one apply_model call and a PLMS-5
    trajectory with classifier-free guidance where the key has something to guide, against the oracle's restatement."""
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    from minddiffusion_amd._lib import MdxError
    cfg = _selfctx_cfg(**({"in_channels": 7} if key == "concat" else {}), **({"num_classes": 10} if key == "adm" else {}))
    params = O.init_params(_oracle_cfg(cfg), seed=21)
    net = _build(cfg, params, True)
    model = LatentDiffusion(unet_config=net, linear_start=0.00085, linear_end=0.0120, timesteps=1000, conditioning_key=key)
    omodel = O.ModelOracle(O.UNetOracle(_oracle_cfg(cfg), params), conditioning_key=key)
    B, H, W, S = 2, 8, 8, 5
    rng = np.random.RandomState(22)
    x_T = rng.randn(B, 4, H, W).astype(np.float32)
    dev = lambda a: torch.tensor(a, device=DEV)
    if key == "concat":
        c, uc, scale = rng.randn(B, 3, H, W).astype(np.float32), np.zeros((B, 3, H, W), np.float32), 3.0
    elif key == "adm":
        c, uc, scale = np.array([3, 7]), np.array([0, 0]), 3.0
    else:
        c, uc, scale = None, None, 1.0
    t = np.full((B,), 500.0, np.float32)
    e = model.apply_model(dev(x_T), dev(t), None if c is None else dev(c))
    eo = omodel.apply_model(torch.tensor(x_T), torch.tensor(t), None if c is None else torch.as_tensor(c))
    check(f"wrapper_key_{key}_apply_model", e, eo, rel_l2=5e-3, max_abs=5e-2)
    ref, _ = O.sample(omodel, S, B, (4, H, W), c, x_T, "plms", unconditional_guidance_scale=scale, unconditional_conditioning=uc)
    got, _ = PLMSSampler(model).sample(S, B, (4, H, W), conditioning=None if c is None else dev(c), x_T=dev(x_T),
                                       unconditional_guidance_scale=scale,
                                       unconditional_conditioning=None if uc is None else dev(uc), verbose=False)
    check(f"wrapper_key_{key}_plms{S}", got, ref, rel_l2=1e-2, max_rel=1e-2)
    # a UNet whose transformer widths differ from context_dim cannot run without a context -- in the reference the Dense
    # shapes do not match; here the call is refused before anything is launched
    bad = _build(_tiny_cfg(), O.init_params(_oracle_cfg(_tiny_cfg()), seed=0), False)
    with pytest.raises(MdxError, match="context_dim"):
        bad(dev(x_T), dev(t))


def test_ddim_eta_runs_and_plms_rejects_eta():
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    cfg = _tiny_cfg()
    net = _build(cfg, O.init_params(_oracle_cfg(cfg), seed=4), True)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120)
    c = torch.randn(1, 5, cfg["context_dim"], device=DEV)
    x_T = torch.randn(1, 4, 8, 8, device=DEV)
    with pytest.raises(ValueError):
        PLMSSampler(model).sample(4, 1, (4, 8, 8), conditioning=c, x_T=x_T, eta=0.5, verbose=False)
    a, _ = DDIMSampler(model).sample(4, 1, (4, 8, 8), conditioning=c, x_T=x_T, eta=0.0, verbose=False)
    b, _ = DDIMSampler(model).sample(4, 1, (4, 8, 8), conditioning=c, x_T=x_T, eta=0.7, verbose=False)
    assert torch.isfinite(b).all() and float((a - b).abs().max()) > 0


def test_sd2_full_size_single_step():
    """BASELINE config 0: SDv2 UNet single denoise step, 1x4x64x64 latent, random text embedding, vs the fp32
    CPU oracle on the same seeded weights (0.8 TFLOP on the CPU: about a minute)."""
    from minddiffusion_amd.configs import SD2_UNET
    import os
    torch.set_num_threads(min(32, os.cpu_count() or 8))   # 0.8 TFLOP fp32 on the host
    params = O.init_params(O.SD2_UNET, seed=0)
    net = _build(dict(SD2_UNET), params, True)
    oracle = O.UNetOracle(O.SD2_UNET, params)
    x = np.random.RandomState(42).randn(1, 4, 64, 64).astype(np.float32)
    ctx = np.random.RandomState(1).randn(1, 77, 1024).astype(np.float32)
    ref = oracle(x, torch.tensor([981.0]), ctx)
    got = net(torch.tensor(x, device=DEV), torch.tensor([981.0], device=DEV), torch.tensor(ctx, device=DEV))
    check("sd2_full_single_step_B1_64x64", got, ref, rel_l2=5e-3, max_abs=5e-2)


def test_sd2_full_size_ddim_cfg_trajectory():
    """BASELINE config 1 at full size, shortened to 4 DDIM steps (the step count must divide 1000, util.py:134-148): SDv2 UNet, 64x64 latent, CFG 9.0 (UNet batch 2), the
    sampler's per-run time-embedding table and the captured hipGraph -- the exact code path bench.py times -- against the
    fp32 CPU oracle's sampler on the same seeded weights (8 UNet row evaluations on the host, about a minute).
    Tolerance: fp16 storage over four chained evaluations with guidance scale 9 amplifying eps differences."""
    from minddiffusion_amd.configs import SD2_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    import os
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    params = O.init_params(O.SD2_UNET, seed=4)
    net = _build(dict(SD2_UNET), params, True)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    omodel = O.ModelOracle(O.UNetOracle(O.SD2_UNET, params))
    S, scale = 4, 9.0
    x_T = np.random.RandomState(42).randn(1, 4, 64, 64).astype(np.float32)
    c = np.random.RandomState(1).randn(1, 77, 1024).astype(np.float32)
    uc = np.random.RandomState(2).randn(1, 77, 1024).astype(np.float32)
    ref, ref_inter = O.sample(omodel, S, 1, (4, 64, 64), c, x_T, "ddim", unconditional_guidance_scale=scale,
                              unconditional_conditioning=uc)
    got, inter = DDIMSampler(model).sample(S, 1, (4, 64, 64), conditioning=torch.tensor(c, device=DEV),
                                           x_T=torch.tensor(x_T, device=DEV), unconditional_guidance_scale=scale,
                                           unconditional_conditioning=torch.tensor(uc, device=DEV), verbose=False)
    check("sd2_full_ddim4_cfg9_latent", got, ref, rel_l2=1e-2, max_rel=1e-2)
    check("sd2_full_ddim4_cfg9_pred_x0", inter["pred_x0"][-1], ref_inter["pred_x0"][-1], rel_l2=2e-2)


def test_sd2_768_single_step():
    """BASELINE config 3 building block: the SDv2 UNet on a 96x96 latent (768x768 px): levels 96/48/24/12, so the
    24x24 and 12x12 convs fall off the 8x16-patch HALO kernel onto the generic implicit-GEMM path, and the
    self-attention runs over N = 9216 tokens (2.15 TFLOP on the CPU oracle)."""
    from minddiffusion_amd.configs import SD2_UNET
    import os
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    params = O.init_params(O.SD2_UNET, seed=1)
    net = _build(dict(SD2_UNET), params, True)
    oracle = O.UNetOracle(O.SD2_UNET, params)
    x = np.random.RandomState(43).randn(1, 4, 96, 96).astype(np.float32)
    ctx = np.random.RandomState(2).randn(1, 77, 1024).astype(np.float32)
    ref = oracle(x, torch.tensor([661.0]), ctx)
    got = net(torch.tensor(x, device=DEV), torch.tensor([661.0], device=DEV), torch.tensor(ctx, device=DEV))
    check("sd2_full_single_step_B1_96x96", got, ref, rel_l2=5e-3, max_abs=5e-2)


def test_wukong_full_size_single_step():
    """BASELINE config 2 building block: the Wukong-Huahua UNet (num_heads = 8 => head dims 40 / 80 / 160, 1x1-conv
    proj_in / proj_out, context_dim 768) on a 64x64 latent, one evaluation vs the fp32 CPU oracle."""
    from minddiffusion_amd.configs import WUKONG_UNET
    import os
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    params = O.init_params(O.WUKONG_UNET, seed=2)
    net = _build(dict(WUKONG_UNET), params, True)
    oracle = O.UNetOracle(O.WUKONG_UNET, params)
    x = np.random.RandomState(44).randn(1, 4, 64, 64).astype(np.float32)
    ctx = np.random.RandomState(3).randn(1, 77, 768).astype(np.float32)
    ref = oracle(x, torch.tensor([301.0]), ctx)
    got = net(torch.tensor(x, device=DEV), torch.tensor([301.0], device=DEV), torch.tensor(ctx, device=DEV))
    check("wukong_full_single_step_B1_64x64", got, ref, rel_l2=5e-3, max_abs=5e-2)


def test_context_cache_is_not_fooled_by_address_reuse():
    """The cross-attention K/V cache is keyed on the conditioning tensor OBJECT: a new prompt whose tensor lands at the
    same address (the caching allocator hands freed blocks back) with the same version counter must be re-projected."""
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=11)
    net = _build(cfg, params, True)
    oracle = O.UNetOracle(_oracle_cfg(cfg), params)
    x, _ = _inputs(2, 8, 8, 5, cfg["context_dim"], seed=3)
    ts = np.full((2,), 400.0, np.float32)
    xd, td = torch.tensor(x, device=DEV), torch.tensor(ts, device=DEV)
    rng = np.random.RandomState(5)
    ca = rng.randn(2, 5, cfg["context_dim"]).astype(np.float32)
    cb = rng.randn(2, 5, cfg["context_dim"]).astype(np.float32)
    ctx = torch.tensor(ca, device=DEV)
    ptr = ctx.data_ptr()
    out_a = net(xd, td, ctx).clone()
    del ctx
    ctx2 = torch.tensor(cb, device=DEV)            # same size: the allocator normally returns the same block
    same_block = ctx2.data_ptr() == ptr
    out_b = net(xd, td, ctx2).clone()
    check(f"context_cache_reuse_sameblock{int(same_block)}", out_b, oracle(x, torch.tensor(ts), cb), rel_l2=5e-3, max_abs=5e-2)
    assert not torch.equal(out_a, out_b)


@pytest.mark.parametrize("graph", [True, False])
def test_time_embedding_table_rows_equal_per_call_embedding(graph):
    """UNetModel.time_embedding_table batches the timestep-only part of the UNet (sinusoid, time_embed MLP, the ResBlock
    emb_layers: openaimodel.py:550-551, 150-157, 188) over all steps of a run; a call that is handed row i must give
    bit-identical eps to the call that recomputes it from t[i] (integer and DPM-Solver's fractional timesteps)."""
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=8)
    net = _build(cfg, params, graph)
    x, ctx = _inputs(2, 8, 8, 7, cfg["context_dim"], seed=9)
    xd, cd = torch.tensor(x, device=DEV), torch.tensor(ctx, device=DEV)
    ts = torch.tensor([981.0, 501.0, 500.25, 1.0, 0.0] + [float(v) for v in range(3, 40, 4)], device=DEV)
    table = net.time_embedding_table(ts)
    assert table.shape == (ts.numel(), net._emb_total) and table.dtype == torch.float32
    for i in (0, 2, 4, 9):
        ref = net.forward_nhwc(xd, ts[i].expand(2), cd).clone()
        got = net.forward_nhwc(xd, None, cd, temb=table[i]).clone()
        assert torch.equal(ref, got), f"row {i}"
        got2 = net.forward_nhwc(xd, None, cd, temb=table[i][None].expand(2, -1)).clone()
        assert torch.equal(ref, got2)
    # different rows per batch element == different timesteps per batch element
    mixed = net.forward_nhwc(xd, None, cd, temb=table[[1, 3]]).clone()
    assert torch.equal(mixed, net.forward_nhwc(xd, ts[[1, 3]], cd))
    with pytest.raises(Exception):
        net.forward_nhwc(xd, None, cd, temb=table[0][:-1])


def test_plan_buffers_survive_allocator_churn():
    """Regression: GEMM descriptors hold raw device pointers, so the plan must own its activation buffers
    (they used to be freed with the planning arena and recycled by the caching allocator)."""
    import gc
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=6)
    net = _build(cfg, params, True)
    oracle = O.UNetOracle(_oracle_cfg(cfg), params)
    x, ctx = _inputs(2, 16, 16, 9, cfg["context_dim"], seed=11)
    xd, td, cd = torch.tensor(x, device=DEV), torch.full((2,), 301.0, device=DEV), torch.tensor(ctx, device=DEV)
    ref = oracle(x, torch.full((2,), 301.0), ctx)
    first = net(xd, td, cd)
    gc.collect()
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), float(i), device=DEV) for i in range(64)]   # recycle whatever was freed
    del junk
    torch.cuda.empty_cache()
    again = net(xd, td, cd)
    assert torch.equal(first, again)
    check("tiny_unet_after_allocator_churn", again, ref, rel_l2=5e-3, max_abs=5e-2)


def test_wukong_style_unet_and_plms():
    """Wukong-Huahua deltas: num_heads=8 (head dims 40/80), 1x1-conv proj_in/out, ctx dim != 1024,
    PLMS sampler with guidance 7.5 (its txt2img default)."""
    from minddiffusion_amd.configs import SMALL_WUKONG_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    cfg = dict(SMALL_WUKONG_UNET)
    ocfg = _oracle_cfg(cfg)
    params = O.init_params(ocfg, seed=7)
    net = _build(cfg, params, True)
    oracle = O.UNetOracle(ocfg, params)
    x, ctx = _inputs(2, 8, 8, 77, cfg["context_dim"], seed=21)
    ts = np.full((2,), 621.0, np.float32)
    ref = oracle(x, torch.tensor(ts), ctx)
    got = net(torch.tensor(x, device=DEV), torch.tensor(ts, device=DEV), torch.tensor(ctx, device=DEV))
    check("wukong_style_unet_forward", got, ref, rel_l2=5e-3, max_abs=5e-2)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    uc = np.repeat(np.random.RandomState(2).randn(1, 77, cfg["context_dim"]).astype(np.float32), 2, 0)
    x_T = np.random.RandomState(42).randn(2, 4, 8, 8).astype(np.float32)
    ref_s, _ = O.sample(O.ModelOracle(oracle), 5, 2, (4, 8, 8), ctx, x_T, "plms", unconditional_guidance_scale=7.5,
                        unconditional_conditioning=uc)
    # Wukong's apply_model takes keywords and dict conditioning (WK plms.py:185-205): both forms must work
    got_s, _ = PLMSSampler(model).sample(5, 2, (4, 8, 8), conditioning={"c_crossattn": [torch.tensor(ctx, device=DEV)]},
                                         x_T=torch.tensor(x_T, device=DEV), unconditional_guidance_scale=7.5,
                                         unconditional_conditioning={"c_crossattn": [torch.tensor(uc, device=DEV)]},
                                         verbose=False)
    check("wukong_style_plms_S5", got_s, ref_s, rel_l2=1e-2, max_rel=1e-2)
    e = model.apply_model(torch.tensor(x, device=DEV), torch.tensor(ts, device=DEV),
                          c_crossattn=torch.tensor(ctx, device=DEV))
    assert torch.equal(e, got)


@pytest.mark.parametrize("mode", ["subset", "original", "dropout"])
def test_sampler_options_vs_oracle(mode):
    """plms_sampling options of the reference that its CLIs never set (plms.py:134-142, 205-208, 224-225):
    `timesteps` prefix of the DDIM grid, `ddim_use_original_steps` (every DDPM step, model-level alpha tables) and
    `noise_dropout` (eta != 0; the N(0,1) draws and the keep masks are injected on both sides)."""
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=12)
    net = _build(cfg, params, True)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    omodel = O.ModelOracle(O.UNetOracle(_oracle_cfg(cfg), params))
    B, H, W, T = 2, 8, 8, 6
    rng = np.random.RandomState(13)
    x_T = rng.randn(B, 4, H, W).astype(np.float32)
    c = rng.randn(B, T, cfg["context_dim"]).astype(np.float32)
    uc = np.repeat(rng.randn(1, T, cfg["context_dim"]).astype(np.float32), B, 0)
    dev = lambda a: torch.tensor(a, device=DEV)
    common = dict(unconditional_guidance_scale=5.0)
    if mode == "subset":       # S = 10 grid, timesteps = 6 -> the first int(0.6 * 10) - 1 = 5 DDIM timesteps, PLMS (6 UNet calls)
        ref, ri = O.sample(omodel, 10, B, (4, H, W), c, x_T, "plms", unconditional_conditioning=uc, timesteps=6, **common)
        got, gi = PLMSSampler(model).sample(10, B, (4, H, W), conditioning=dev(c), x_T=dev(x_T), verbose=False,
                                            unconditional_conditioning=dev(uc), timesteps=6, **common)
        assert len(gi["x_inter"]) == len(ri["x_inter"])
    elif mode == "original":   # the 7 lowest DDPM steps t = 6..0 with alphas_cumprod[t] / alphas_cumprod_prev[t]
        ref, _ = O.sample(omodel, 10, B, (4, H, W), c, x_T, "ddim", unconditional_conditioning=uc, timesteps=7,
                          ddim_use_original_steps=True, **common)
        got, _ = DDIMSampler(model).sample(10, B, (4, H, W), conditioning=dev(c), x_T=dev(x_T), verbose=False,
                                           unconditional_conditioning=dev(uc), timesteps=7, ddim_use_original_steps=True,
                                           **common)
    else:                      # DDIM eta = 0.6 with 30 % noise dropout
        S = 5
        noises = [rng.randn(B, 4, H, W).astype(np.float32) for _ in range(S)]
        masks = [(rng.rand(B, 4, H, W) >= 0.3).astype(np.float32) for _ in range(S)]
        it = iter(noises)
        ref, _ = O.sample(omodel, S, B, (4, H, W), c, x_T, "ddim", eta=0.6, unconditional_conditioning=uc,
                          noise_fn=lambda shp: next(it), noise_dropout=0.3, dropout_masks=masks, **common)
        got, _ = DDIMSampler(model).sample(S, B, (4, H, W), conditioning=dev(c), x_T=dev(x_T), verbose=False, eta=0.6,
                                           unconditional_conditioning=dev(uc), noise_dropout=0.3, dropout_masks=masks,
                                           step_noises=noises, **common)
        # without injected masks the draw comes from the generator: finite, and different from the no-dropout run
        free, _ = DDIMSampler(model).sample(S, B, (4, H, W), conditioning=dev(c), x_T=dev(x_T), verbose=False, eta=0.6,
                                            unconditional_conditioning=dev(uc), noise_dropout=0.3, step_noises=noises, **common)
        assert torch.isfinite(free).all() and not torch.equal(free, got)
    check(f"tiny_sampler_option_{mode}", got, ref, rel_l2=1e-2, max_rel=1e-2)


class _Corrector:
    """A score corrector in the reference's calling convention (plms.py:201): works on whatever tensor library it is handed."""

    def __init__(self):
        self.calls = []

    def modify_score(self, model, e_t, x, t, c, gain=1.0):
        self.calls.append((tuple(e_t.shape), [int(v) for v in t]))
        return gain * e_t + 0.05 * x


class _SoftQuantizer:
    """Stands in for a VQ first stage in the (z_q, loss, info) calling convention.  A real codebook lookup (or any hard
    clipping) is not a parity-testable map: on this random-weight UNet the oracle's own fp32 and fp16-emulated runs end 3-12 %
    apart with `clamp` as the quantizer, because pred_x0 = (x - s e) / sqrt(a_t) divides fp16-level differences in e by
    sqrt(a_t) ~ 0.07 at the high-noise steps and the clipped trajectory leaves the region where the sampler contracts.  A smooth
    contraction (z - 0.3 tanh z: moves the samples by ~19 % rel-L2, oracle fp32-vs-fp16 distance 2e-3) exercises the same code."""

    def quantize(self, z):
        return z - 0.3 * torch.tanh(z), None, (None, None, None)


@pytest.mark.parametrize("which", ["plms_corrector", "ddim_corrector_quantize", "plms_quantize"])
def test_sampler_hooks_vs_oracle(which):
    """score_corrector / corrector_kwargs and quantize_x0 (plms.py:199-201, 218-219): caller-supplied objects inside the step."""
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    from minddiffusion_amd._lib import MdxError
    cfg = _tiny_cfg()
    params = O.init_params(_oracle_cfg(cfg), seed=21)
    net = _build(cfg, params, True)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    omodel = O.ModelOracle(O.UNetOracle(_oracle_cfg(cfg), params))
    B, H, W, T, S = 2, 8, 8, 6, 5
    rng = np.random.RandomState(5)
    x_T = rng.randn(B, 4, H, W).astype(np.float32)
    c = rng.randn(B, T, cfg["context_dim"]).astype(np.float32)
    uc = np.repeat(rng.randn(1, T, cfg["context_dim"]).astype(np.float32), B, 0)
    dev = lambda a: torch.tensor(a, device=DEV)
    kind = "plms" if which.startswith("plms") else "ddim"
    Sampler = PLMSSampler if kind == "plms" else DDIMSampler
    kw, okw = dict(unconditional_guidance_scale=4.0), {}
    cor_ref, cor_got = _Corrector(), _Corrector()
    if "corrector" in which:
        kw.update(corrector_kwargs=dict(gain=0.9))
        okw.update(score_corrector=cor_ref)
    quant = "quantize" in which
    if quant:
        with pytest.raises(MdxError, match="quantize"):        # no first stage with .quantize attached
            Sampler(model).sample(S, B, (4, H, W), conditioning=dev(c), x_T=dev(x_T), verbose=False, quantize_x0=True)
        model.first_stage_model = omodel.first_stage_model = _SoftQuantizer()
    ref, ri = O.sample(omodel, S, B, (4, H, W), c, x_T, kind, unconditional_conditioning=uc, quantize_x0=quant,
                       log_every_t=1, **okw, **kw)
    got, gi = Sampler(model).sample(S, B, (4, H, W), conditioning=dev(c), x_T=dev(x_T), verbose=False, log_every_t=1,
                                    unconditional_conditioning=dev(uc), quantize_x0=quant,
                                    score_corrector=cor_got if "corrector" in which else None, **kw)
    if "corrector" in which:       # same number of calls (PLMS: S + 1), on the same timesteps, on the un-doubled batch
        assert cor_got.calls == cor_ref.calls and len(cor_got.calls) == S + (kind == "plms")
    if quant:
        assert len(gi["pred_x0"]) == len(ri["pred_x0"]) == S + 1
        plain, _ = Sampler(model).sample(S, B, (4, H, W), conditioning=dev(c), x_T=dev(x_T), verbose=False,
                                         unconditional_conditioning=dev(uc),
                                         score_corrector=_Corrector() if "corrector" in which else None, **kw)
        assert float((plain - got).norm() / plain.norm()) > 0.05       # the quantizer was in the loop
        check(f"tiny_sampler_hook_{which}_pred_x0", gi["pred_x0"][-1], ri["pred_x0"][-1], rel_l2=1e-2, max_rel=1e-2)
    check(f"tiny_sampler_hook_{which}", got, ref, rel_l2=1e-2, max_rel=1e-2)
