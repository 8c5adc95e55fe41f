"""CPU-side tests of the host logic and the C-ABI surface (no kernel is launched here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import ldm as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    """include/mdx.h is the contract: every function it declares must be exported by libmdx.so and bound."""
    from minddiffusion_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mdx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(mdx_[a-z0-9_]+)\s*\(", header))
    declared.discard("mdx_gemm_desc")
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in mdx.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == declared
    assert lib.mdx_version() == 1


def test_gemm_desc_struct_matches_header_field_order():
    from minddiffusion_amd import _lib
    header = open(os.path.join(ROOT, "include", "mdx.h")).read()
    body = header[header.index("typedef struct mdx_gemm_desc {"):header.index("} mdx_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
    assert names == [f[0] for f in _lib.GemmDesc._fields_]


def test_argument_validation_without_gpu():
    """Bad descriptors are rejected on the host before any launch (error code + message, no exception across the ABI)."""
    from minddiffusion_amd import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.mdx_gemm_f16(ctypes.byref(d), None) == -1
    assert b"null pointer" in lib.mdx_last_error()
    d.a, d.w, d.out = 16, 16, 16
    d.c1, d.B, d.H, d.W, d.N, d.ksize, d.stride, d.out_ld = 12, 1, 4, 4, 64, 3, 1, 64
    assert lib.mdx_gemm_f16(ctypes.byref(d), None) == -1
    assert b"multiples of 8" in lib.mdx_last_error()
    assert lib.mdx_attention_f16(16, 0, 64, 16, 0, 64, 16, 0, 64, 16, 0, 64, 1, 1, 48, 8, 8, 1.0, None) == -1
    assert b"head dim" in lib.mdx_last_error()
    assert lib.mdx_groupnorm_ws_floats(2, 4096, 320, 32) == 2 * 64 * 32 * 2


def test_parameter_names_and_structure_match_oracle():
    from minddiffusion_amd.configs import SD2_UNET, WUKONG_UNET, TINY_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    tiny_o = dict(TINY_UNET, num_heads=-1)
    for cfg, ocfg in ((SD2_UNET, O.SD2_UNET), (WUKONG_UNET, O.WUKONG_UNET), (TINY_UNET, tiny_o)):
        net = UNetModel(**cfg)
        assert net.parameter_shapes() == O.unet_param_shapes(ocfg)
        inb, mid, outb = O.unet_structure(ocfg)
        assert (net.input_blocks, net.middle_block, net.output_blocks) == (inb, mid, outb)


UNET_VARIANTS = {
    # every UNetModel constructor switch the reference implements (openaimodel.py:283-531) beyond the shipped YAMLs
    "depth2": dict(transformer_depth=2),
    "scale_shift": dict(use_scale_shift_norm=True),
    "updown": dict(resblock_updown=True),
    "updown_scale_shift": dict(resblock_updown=True, use_scale_shift_norm=True, transformer_depth=2),
    "pool_resample": dict(conv_resample=False),
    "class_cond": dict(num_classes=10),
    "codebook_ids": dict(n_embed=24),
}


@pytest.mark.parametrize("name", sorted(UNET_VARIANTS))
def test_unet_constructor_variants_match_oracle_structure(name):
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    cfg = dict(TINY_UNET, **UNET_VARIANTS[name])
    ocfg = dict(cfg, num_heads=-1)
    net = UNetModel(**cfg)
    assert net.parameter_shapes() == O.unet_param_shapes(ocfg)
    assert (net.input_blocks, net.middle_block, net.output_blocks) == O.unet_structure(ocfg)
    # the oracle itself runs the variant and is sensitive to the switch
    p = O.init_params(ocfg, seed=3)
    rng = np.random.RandomState(0)
    x, ctx = rng.randn(2, 4, 8, 8).astype(np.float32), rng.randn(2, 5, 64).astype(np.float32)
    kw = dict(y=[1, 7]) if "num_classes" in cfg else {}
    out = O.UNetOracle(ocfg, p)(x, [3, 900], ctx, **kw)
    assert tuple(out.shape) == (2, cfg.get("n_embed") or 4, 8, 8) and bool(torch.isfinite(out).all())
    if "num_classes" in cfg:
        assert float((O.UNetOracle(ocfg, p)(x, [3, 900], ctx, y=[2, 7]) - out)[0].abs().max()) > 1e-4
        with pytest.raises(AssertionError):
            O.UNetOracle(ocfg, p)(x, [3, 900], ctx)


@pytest.mark.parametrize("name", sorted(UNET_VARIANTS))
def test_unet_constructor_variants_plan_on_the_host(name):
    """The planner is host code: for every constructor variant the op list is built (no launch) from the reference's block
    structure, every GEMM op carries its descriptor and a resolved launch count, and the variant's own ops are present."""
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_numpy
    cfg = dict(TINY_UNET, **UNET_VARIANTS[name])
    net = UNetModel(device="cpu", **cfg)
    net.load_state_dict(synthetic_unet_params_numpy(net.parameter_shapes(), 0))
    base = UNetModel(device="cpu", **TINY_UNET)
    base.load_state_dict(synthetic_unet_params_numpy(base.parameter_shapes(), 0))
    P, P0 = net._plan(2, 8, 8), base._plan(2, 8, 8)
    assert len(P.main) == len(P.meta) and all(m["launches"] in (1, 2) for m in P.meta)
    gemms = [m for m in P.meta if m["kind"] == "gemm"]
    assert gemms and all("desc" in m and " split=" in m["info"] for m in gemms)
    blocks = sum(1 for k in base.w if k.endswith("attn2.q.w"))
    n = lambda Pl, kind: sum(m["kind"] == kind for m in Pl.meta)
    xf = lambda Pl: sum("+cross-attention" in m["info"] for m in Pl.meta)      # attn2 riding on its query projection (head dim 64, round 6)
    if name == "depth2":                # a second BasicTransformerBlock per SpatialTransformer: 2 attentions and 6 GEMMs each
        assert n(P, "attention") + xf(P) == n(P0, "attention") + xf(P0) + 2 * blocks and n(P, "gemm") == n(P0, "gemm") + 6 * blocks   # (+ 2 context GEMMs outside the op list)
        assert len(P.ctxops) == len(P0.ctxops) + 2 * blocks
    elif name == "pool_resample":       # the resampling convs become parameter-free pooling / nearest ops
        assert n(P, "gemm") == n(P0, "gemm") - 2 and n(P, "small") == n(P0, "small") + 2
    elif name.startswith("updown"):     # ResBlocks replace the Downsample / Upsample convs: + 2 GroupNorms each, pooled skip inputs
        assert n(P, "groupnorm") == n(P0, "groupnorm") + 4
    elif name == "class_cond":          # emb + label_emb(y): one more op in front of the emb_layers projection
        assert P.temb_ops == P0.temb_ops + 1 and hasattr(P, "y_static")
    elif name == "codebook_ids":        # the head is a 1x1 conv to n_embed channels
        assert tuple(P.eps_nhwc.shape) == (2, 64, 24) and P.descs[-1 - 2 * blocks].ksize == 1
    if name == "scale_shift":           # emb_layers project to (scale, shift): twice the rows
        assert net._emb_total == 2 * base._emb_total


def test_unet_unsupported_constructor_arguments_raise():
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    with pytest.raises(NotImplementedError):
        UNetModel(**dict(TINY_UNET, dims=3))
    with pytest.raises(NotImplementedError):      # AttentionBlock is an empty stub in the reference
        UNetModel(**dict(TINY_UNET, use_spatial_transformer=False, context_dim=None))


def test_product_schedule_equals_oracle():
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler

    class _Unet:  # the schedule does not touch the UNet
        pass
    m = LatentDiffusion(_Unet(), linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    s = O.register_schedule()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_one_minus_alphas_cumprod"):
        np.testing.assert_array_equal(getattr(m, k), s[k])
    assert m.num_timesteps == 1000
    smp = PLMSSampler(m)
    smp.make_schedule(50, ddim_eta=0.0, verbose=False)
    sig, a, ap = O.make_ddim_sampling_parameters(s["alphas_cumprod"], O.make_ddim_timesteps(50), 0.0)
    np.testing.assert_array_equal(smp.ddim_timesteps, O.make_ddim_timesteps(50))
    np.testing.assert_array_equal(smp.ddim_alphas, a)
    np.testing.assert_array_equal(smp.ddim_alphas_prev, ap)
    np.testing.assert_array_equal(smp.ddim_sigmas, sig)
    with pytest.raises(ValueError):
        smp.make_schedule(50, ddim_eta=0.1, verbose=False)     # plms.py:35-36
    d = DDIMSampler(m)
    d.make_schedule(50, ddim_eta=0.3, verbose=False)            # allowed for DDIM
    sig2, _, _ = O.make_ddim_sampling_parameters(s["alphas_cumprod"], O.make_ddim_timesteps(50), 0.3)
    np.testing.assert_array_equal(d.ddim_sigmas, sig2)


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU instead of computing on the CPU."""
    from minddiffusion_amd._lib import MdxError
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    net = UNetModel(**TINY_UNET)
    with pytest.raises(MdxError):
        net(torch.zeros(1, 4, 8, 8), torch.zeros(1), torch.zeros(1, 5, 64))


def test_diffusion_wrapper_routes_every_conditioning_key():
    """DiffusionWrapper.construct, WK ddpm.py:360-377, and the bare-conditioning rule of apply_model (ddpm.py:299-300): which
    keyword reaches the UNet for each of the five keys (host logic only: the UNet is a recorder)."""
    from minddiffusion_amd._lib import MdxError
    from minddiffusion_amd.ldm.models.diffusion.ddpm import DiffusionWrapper, LatentDiffusion

    class Rec:
        def __call__(self, x, t, context=None, y=None):
            self.seen = dict(x=x, context=context, y=y)
            return x
    x, t = torch.zeros(2, 4, 8, 8), torch.zeros(2)
    cc, ctx, lab = torch.ones(2, 3, 8, 8), torch.ones(2, 5, 16), torch.tensor([1, 2])
    for key, cond, want_c, want_ctx, want_y in ((None, None, 4, None, None), ("concat", cc, 7, None, None),
                                                 ("crossattn", ctx, 4, ctx, None), ("adm", lab, 4, None, lab),
                                                 ("hybrid", {"c_concat": [cc], "c_crossattn": [ctx]}, 7, ctx, None)):
        rec = Rec()
        m = LatentDiffusion(unet_config=rec, conditioning_key=key)
        m.apply_model(x, t, cond)
        assert rec.seen["x"].shape[1] == want_c, key
        assert rec.seen["context"] is want_ctx and rec.seen["y"] is want_y, key
    with pytest.raises(AssertionError):
        DiffusionWrapper(Rec(), "text")
    with pytest.raises(MdxError):
        LatentDiffusion(unet_config=Rec(), conditioning_key="concat").apply_model(x, t, None)
    with pytest.raises(MdxError):
        LatentDiffusion(unet_config=Rec(), conditioning_key="adm").apply_model(x, t, None)
    with pytest.raises(MdxError):
        LatentDiffusion(unet_config=Rec(), conditioning_key="crossattn").apply_model(x, t, {"c_concat": cc, "c_crossattn": ctx})


def test_distributed_selects_rccl_when_a_gpu_is_present(monkeypatch):
    """The only untested thing about N > 1 should be RCCL itself: with MDX_DIST_BACKEND unset and a GPU visible, init_from_env
    picks "nccl" (= RCCL on ROCm), pins the device BEFORE creating the group and passes device_id; _broadcast then hands the DEVICE
    payload to dist.broadcast directly (no host staging) and neither times nor synchronises unless time_collectives is set."""
    from minddiffusion_amd import distributed as D
    calls = []
    monkeypatch.delenv("MDX_DIST_BACKEND", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setattr(D.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(D.torch.cuda, "set_device", lambda i: calls.append(("set_device", i)))
    monkeypatch.setattr(D.dist, "is_initialized", lambda: False)
    monkeypatch.setattr(D.dist, "init_process_group", lambda **kw: calls.append(("init", kw)))
    assert D.init_from_env() == (1, 2, 1)
    assert calls[0] == ("set_device", 1)
    assert calls[1][0] == "init" and calls[1][1]["backend"] == "nccl" and calls[1][1]["device_id"] == torch.device("cuda", 1)
    # gloo is only taken on request (several ranks on one GPU) or without a GPU
    calls.clear()
    monkeypatch.setenv("MDX_DIST_BACKEND", "gloo")
    D.init_from_env()
    assert calls == [("init", {"backend": "gloo"})]
    calls.clear()
    monkeypatch.delenv("MDX_DIST_BACKEND")
    monkeypatch.setattr(D.torch.cuda, "is_available", lambda: False)
    D.init_from_env()
    assert calls == [("init", {"backend": "gloo"})]

    class DevPayload:       # what _broadcast touches of a device tensor
        is_cuda = True

        def numel(self):
            return 10

        def element_size(self):
            return 2

        def cpu(self):
            raise AssertionError("the RCCL branch must not stage the payload through the host")
    sent = []
    monkeypatch.setattr(D.dist, "get_backend", lambda: "nccl")
    monkeypatch.setattr(D.dist, "broadcast", lambda t, src: sent.append((t, src)))
    monkeypatch.setattr(D.torch.cuda, "Event", lambda **kw: (_ for _ in ()).throw(AssertionError("timed without opt-in")))
    D.reset_collective_stats()
    assert D.time_collectives is False
    pl = DevPayload()
    D._broadcast(pl, 0)
    assert sent == [(pl, 0)] and D.collective_stats["broadcasts"] == 1 and D.collective_stats["bytes"] == 20
    assert D.collective_stats["ms"] == 0.0


def test_tune_cache_file_is_versioned(tmp_path):
    """ops.save_tune_cache / load_tune_cache: a file written under another key layout is ignored with a warning instead of
    loading entries that can never match (every shape would silently be measured again)."""
    import json
    import warnings
    from minddiffusion_amd import ops
    saved = dict(ops.tune_cache)
    try:
        ops.tune_cache.clear()
        key = (2048, 640, 640, 1, 1, 0, 4096 | 256, 2, 1024, 1)
        ops.tune_cache[key] = (64, 64, 1, 3, 12.5, 11.0)
        f = tmp_path / "tc.json"
        ops.save_tune_cache(str(f))
        ops.tune_cache.clear()
        assert ops.load_tune_cache(str(f)) == 1 and ops.tune_cache[key] == (64, 64, 1, 3, 12.5, 11.0)
        ops.tune_cache.clear()
        old = tmp_path / "old.json"
        old.write_text(json.dumps([[[2048, 640, 640, 1, 1, 0, 256], [64, 64, 1, 3, 12.5, 11.0]]]))      # rounds 1-3: 7-field keys
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert ops.load_tune_cache(str(old)) == 0
        assert w and "another key layout" in str(w[0].message) and not ops.tune_cache
    finally:
        ops.tune_cache.clear()
        ops.tune_cache.update(saved)


def test_instantiate_from_config_reference_targets():
    from minddiffusion_amd.ldm.util import instantiate_from_config
    from minddiffusion_amd.configs import TINY_UNET
    net = instantiate_from_config({"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": TINY_UNET})
    assert type(net).__name__ == "UNetModel" and net.model_channels == 64


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under minddiffusion_amd/ may reference it."""
    pkg = os.path.join(ROOT, "minddiffusion_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, fn)


def test_unet_plan_builds_and_every_descriptor_validates():
    """Plan the tiny UNet with host tensors (nothing is launched): exercises the arena, the layer walk and
    the C-side validation of every GEMM/conv descriptor the forward pass would issue."""
    import ctypes
    from minddiffusion_amd import _lib
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_numpy
    lib = _lib.load()
    net = UNetModel(device="cpu", **TINY_UNET)
    net.load_state_dict(synthetic_unet_params_numpy(net.parameter_shapes(), 0))
    for shape in ((2, 8, 8), (3, 8, 12)):
        P = net._plan(*shape)
        assert len(P.main) > 100 and len(P.ctxops) == 2 * 7
        for d in P.descs:
            rc = lib.mdx_gemm_check(ctypes.byref(d))
            assert rc == 0, lib.mdx_last_error()
    with pytest.raises(_lib.MdxError):
        net._plan(1, 7, 8)   # not divisible by the downsampling factor


def test_guidance_duplicate_op_list_on_host():
    """UNetModel._dup_body composed from host-planned plans (nothing is launched): the full-batch plan up to conv_in, two copies into
    the half-batch plan, that plan's ops up to the splice point, the copies back, the rest of the full-batch plan -- with the metadata
    list in step; no prefix for an odd batch, below the option's batch, or for a UNet without attention at its first level."""
    from minddiffusion_amd import ops
    from minddiffusion_amd.configs import TINY_UNET, SMALL_WUKONG_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_numpy
    old = ops.get_option("unet_cfg_dup")
    try:
        for cfg in (TINY_UNET, SMALL_WUKONG_UNET):
            net = UNetModel(device="cpu", **cfg)
            net.load_state_dict(synthetic_unet_params_numpy(net.parameter_shapes(), 0))
            ops.set_option("unet_cfg_dup", 4)
            P, PA = net._plan(4, 8, 8), net._plan(2, 8, 8)
            body = net._dup_body(P)
            assert body is not None and len(body) == len(P.dup_meta) and P.dup_half is PA
            ic = P.main.index(P.ck["conv_in"])
            assert body[:ic + 1 - P.temb_ops] == P.main[P.temb_ops:ic + 1]                 # layout pass + conv_in at the full batch
            key = "op2" if ("op2" in P.ck and "op2" in PA.ck) else "op"
            ja, ib = PA.main.index(PA.ck[key]), P.main.index(P.ck[key])
            assert body[-(len(P.main) - ib - 1):] == P.main[ib + 1:]                        # the full-batch plan behind the splice
            assert all(op in body for op in PA.main[PA.main.index(PA.ck["conv_in"]):ja + 1])
            assert not any(op in body for op in P.main[ic + 1:ib + 1])                      # the duplicated launches are gone
            assert sum(m["flops"] for m in P.dup_meta) < sum(m["flops"] for m in P.meta[P.temb_ops:])
            assert net._dup_body(net._plan(3, 8, 8)) is None                                # odd batch
            assert net._dup_body(net._plan(2, 8, 8)) is None                                # below the option's batch
            ops.set_option("unet_cfg_dup", 0)
            assert net._dup_body(P) is None
        ops.set_option("unet_cfg_dup", 2)
        net = UNetModel(device="cpu", **dict(TINY_UNET, attention_resolutions=[2]))
        net.load_state_dict(synthetic_unet_params_numpy(net.parameter_shapes(), 0))
        assert net._plan(4, 8, 8).ck is None and net._dup_body(net._plan(4, 8, 8)) is None
    finally:
        ops.set_option("unet_cfg_dup", old)


TINY_GLIDE = dict(image_size=16, num_channels=64, num_res_blocks=1, channel_mult=(1, 2), num_heads=1,
                  num_head_channels=64, num_heads_upsample=-1, attention_resolutions=(1, 2), dropout=0.0, text_ctx=16,
                  xf_width=64, xf_layers=2, xf_heads=1, xf_final_ln=True, n_vocab=100, xf_padding=True,
                  diffusion_steps=1000, noise_schedule="squaredcos_cap_v2", timestep_respacing="10",
                  use_scale_shift_norm=True, resblock_updown=True, use_fp16=True, cache_text_emb=False)


def test_glide_names_structure_and_plan_on_host():
    """GLIDE mirror: parameter names/shapes equal the oracle's, the product's schedules equal the reference goldens,
    and the planned op list validates descriptor by descriptor (nothing is launched)."""
    import ctypes
    from oracle import glide as OG
    from minddiffusion_amd import _lib
    from minddiffusion_amd.glide import gaussian_computation as gc
    from minddiffusion_amd.glide.default_options import model_and_diffusion_defaults, model_and_diffusion_upsample
    from minddiffusion_amd.glide.diffusion_creator import create_model, create_upsample_model, _Schedule
    gold = np.load(os.path.join(ROOT, "tests", "golden", "glide_schedule.npz"))
    np.testing.assert_array_equal(gc.get_named_beta_schedule("squaredcos_cap_v2", 1000), gold["cosine_betas_1000"])
    np.testing.assert_array_equal(gc.get_named_beta_schedule("linear", 1000), gold["linear_betas_1000"])
    assert sorted(gc.space_timesteps(1000, "60")) == list(gold["space_60"])
    assert sorted(gc.space_timesteps(1000, "fast27")) == list(gold["space_fast27"])
    s, so = _Schedule("squaredcos_cap_v2", 1000, "60"), OG.respaced_schedule("squaredcos_cap_v2", 1000, "60")
    np.testing.assert_array_equal(s.timestep_map, so["timestep_map"])
    np.testing.assert_array_equal(s.coef1, so["coef1"])
    np.testing.assert_array_equal(s.post_logvar, so["post_logvar_clipped"])
    # full-size models: names and shapes only
    base = create_model(device="cpu", **model_and_diffusion_defaults())
    assert base.parameter_shapes() == OG.param_shapes(OG.BASE_OPTIONS)
    up = create_upsample_model(device="cpu", **model_and_diffusion_upsample())
    assert up.parameter_shapes() == OG.param_shapes(OG.UPSAMPLE_OPTIONS)
    # tiny model: plan on the host and validate every descriptor
    otiny = dict(OG.BASE_OPTIONS, image_size=16, model_channels=64, num_res_blocks=1, channel_mult=(1, 2),
                 attention_resolutions=(1, 2), text_ctx=16, xf_width=64, xf_layers=2, xf_heads=1, n_vocab=100)
    net = create_model(device="cpu", **TINY_GLIDE)
    assert net.parameter_shapes() == OG.param_shapes(otiny)
    net.load_state_dict(OG.init_params(otiny, seed=0))
    lib = _lib.load()
    P = net._plan(4, 16, 16)
    assert len(P.main) > 80
    for d in P.descs:
        assert lib.mdx_gemm_check(ctypes.byref(d)) == 0, lib.mdx_last_error()
    upt = create_upsample_model(device="cpu", low_size=8, **dict(TINY_GLIDE, image_size=32, channel_mult=(1, 1, 2)))
    oup = dict(otiny, in_channels=6, image_size=32, channel_mult=(1, 1, 2))
    upt.load_state_dict(OG.init_params(oup, seed=1))
    Pu = upt._plan(2, 32, 32)
    for d in Pu.descs:
        assert lib.mdx_gemm_check(ctypes.byref(d)) == 0, lib.mdx_last_error()


def test_vae_decoder_names_structure_and_plan_on_host():
    """VAE decode mirror: parameter names/shapes equal the oracle's for the shipped ddconfig, and
    the planned op list of a tiny decoder validates descriptor by descriptor (nothing is launched)."""
    import ctypes
    from oracle import vae as OV
    from minddiffusion_amd import _lib
    from minddiffusion_amd.configs import SD_VAE_DDCONFIG, TINY_VAE_DDCONFIG
    from minddiffusion_amd.ldm.models.autoencoder import AutoencoderKL
    full = AutoencoderKL(ddconfig=SD_VAE_DDCONFIG, embed_dim=4, device="cpu")
    assert full.parameter_shapes() == OV.param_shapes(OV.SD_VAE, 4)
    assert sum(int(np.prod(s)) for s in full.parameter_shapes().values()) == 83_653_863   # the SD VAE, encoder + decoder
    dd = dict(TINY_VAE_DDCONFIG)
    vae = AutoencoderKL(ddconfig=dd, embed_dim=4, device="cpu")
    assert vae.parameter_shapes() == OV.param_shapes(dd, 4)
    vae.load_state_dict(OV.init_params(dd, seed=0))
    lib = _lib.load()
    P = vae.decoder._plan(2, 16, 16)
    assert P.out_hw == (32, 32) and tuple(P.out_nchw.shape) == (2, 3, 32, 32)
    kinds = [m["kind"] for m in P.meta]
    # conv_in + post_quant + (3 mid/2 + 2 up-level) resblocks ... : every GEMM descriptor must validate on the host
    assert kinds.count("gemm") == len(P.descs) and len(P.descs) >= 20
    for d in P.descs:
        assert lib.mdx_gemm_check(ctypes.byref(d)) == 0, lib.mdx_last_error()
    Pe = vae.encoder._plan(2, 32, 32)
    assert Pe.out_hw == (16, 16) and tuple(Pe.moments.shape) == (2, 256, 8)
    for d in Pe.descs:
        assert lib.mdx_gemm_check(ctypes.byref(d)) == 0, lib.mdx_last_error()
    assert any(d.asym_pad == 1 and d.stride == 2 for d in Pe.descs)


def test_mindspore_checkpoint_reader_roundtrip(tmp_path):
    """MindSpore .ckpt wire format: writer -> reader round trip incl. fp16 / int payloads, a
    parameter split over several same-tag slices, concatenated messages, prefix stripping and the LatentDiffusion split.
    A hand-assembled message pins the field numbers independently of our own writer."""
    from minddiffusion_amd import ms_checkpoint as C
    rng = np.random.RandomState(0)
    params = {
        "model.diffusion_model.time_embed.0.weight": rng.standard_normal((8, 4)).astype(np.float32),
        "model.diffusion_model.input_blocks.0.0.conv.bias": rng.standard_normal(7).astype(np.float16),
        "first_stage_model.decoder.conv_in.weight": rng.standard_normal((3, 2, 3, 3)).astype(np.float32),
        "cond_stage_model.transformer.embedding_table": rng.standard_normal((5, 6)).astype(np.float32),
        "global_step": np.array([123456789012], dtype=np.int64),
        "scalar": np.array(2.5, dtype=np.float32),
    }
    path = str(tmp_path / "m.ckpt")
    C.save_checkpoint(params, path, slice_bytes=40)          # forces multi-slice parameters
    got = C.load_checkpoint(path)
    assert list(got) == list(params)
    for k, v in params.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        np.testing.assert_array_equal(got[k], v)
    unet = C.load_checkpoint(path, strip_prefix=C.UNET_PREFIX)
    assert sorted(unet) == ["input_blocks.0.0.conv.bias", "time_embed.0.weight"]
    u, v, t = C.load_latent_diffusion(path)
    assert "time_embed.0.weight" in u and "decoder.conv_in.weight" in v and "transformer.embedding_table" in t
    # hand-assembled: Checkpoint{ value{ tag="w", tensor{ dims=[2,2] (packed), tensor_type="Float32", content } } }
    content = np.arange(4, dtype=np.float32).tobytes()
    tensor = bytes([0x0A, 0x02, 0x02, 0x02]) + bytes([0x12, 7]) + b"Float32" + bytes([0x1A, 16]) + content
    value = bytes([0x0A, 1]) + b"w" + bytes([0x12, len(tensor)]) + tensor
    raw = bytes([0x0A, len(value)]) + value
    p2 = str(tmp_path / "h.ckpt")
    open(p2, "wb").write(raw)
    np.testing.assert_array_equal(C.load_checkpoint(p2)["w"], np.arange(4, dtype=np.float32).reshape(2, 2))
    # bfloat16 payloads widen to float32
    bf = (np.array([1.0, -2.5, 3.0], np.float32).view(np.uint32) >> 16).astype(np.uint16).tobytes()
    tensor = bytes([0x08, 3]) + bytes([0x12, 8]) + b"BFloat16" + bytes([0x1A, len(bf)]) + bf
    value = bytes([0x0A, 1]) + b"b" + bytes([0x12, len(tensor)]) + tensor
    open(p2, "wb").write(bytes([0x0A, len(value)]) + value)
    np.testing.assert_array_equal(C.load_checkpoint(p2)["b"], np.array([1.0, -2.5, 3.0], np.float32))


def test_layernorm_fold_host_algebra():
    """ops.fold_layernorm (host side of mdx_gemm_desc.ln_stats): LN(x; gamma, beta) W^T + b must equal
    rstd * (x (gamma (.) W)^T - mean * S) + (W beta + b) -- checked in float64 on rows whose mean is as large as
    their spread, with the fp16 rounding of gamma (.) W taken into account on both sides (BasicTransformerBlock
    attention.py:176-185)."""
    from minddiffusion_amd import ops
    rng = np.random.RandomState(0)
    M, K, N = 16, 128, 64
    x = torch.tensor(rng.standard_normal((M, K)) * 1.5 + 1.0).half().double()
    w = torch.tensor(rng.standard_normal((N, K)) / np.sqrt(K)).half()
    g = torch.tensor(1 + 0.3 * rng.standard_normal(K), dtype=torch.float32)
    be = torch.tensor(0.3 * rng.standard_normal(K), dtype=torch.float32)
    b = torch.tensor(rng.standard_normal(N), dtype=torch.float32)
    wg, s, cb = ops.fold_layernorm(w, g, be, b)
    assert wg.dtype == torch.float16 and s.shape == (N,) and cb.shape == (N,)
    mean = x.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((x - mean) ** 2).mean(1, keepdim=True) + 1e-5)
    folded = rstd * (x @ wg.double().T - mean * s.double()) + cb.double()
    # reference with the SAME fp16-rounded gamma (.) W, i.e. the only approximation the fold introduces is that rounding
    exact = ((x - mean) * rstd) @ wg.double().T + (w.double() @ be.double() + b.double())
    assert float((folded - exact).abs().max()) < 1e-5      # S is summed in float64 and stored in fp32: ~1e-7 * |mean * S|
    plain = ((x - mean) * rstd * g.double() + be.double()) @ w.double().T + b.double()
    assert float((folded - plain).abs().max()) < 5e-3      # fp16 rounding of gamma (.) W (2^-11 relative per weight)


def test_tuned_tile_table_is_well_formed():
    """csrc/gemm_tuned.inc (tools/tune_gemm.py): one {M, N, K, ksize, tile_m, tile_n, splitk[, variant + 1[, stages]]} row per
    shape and launch variant (rows without the 8th field predate the variant key and match any variant; stages = LDS ring depth
    2 .. 6, 10 | 11 = depth 2 | 3 in the eight-wave form of the generic kernel, 0 / absent = the library's rule).  256-row tiles exist for 3x3 convs
    only (HALO kernel), the eight-wave form for 128-row tiles only."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "minddiffusion_amd", "csrc",
                        "gemm_tuned.inc")
    seen = set()
    for ln in open(path):
        if ln.lstrip().startswith("//") or not ln.strip():
            continue
        m = re.match(r"\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, (\d+))?\},", ln)
        assert m, ln
        M, N, K, ks, bm, bn, ns = map(int, m.groups()[:7])
        var1, st = int(m.group(8) or 0), int(m.group(9) or 0)
        assert M > 0 and N % 8 == 0 and K > 0 and ks in (1, 3) and bm in (64, 128, 256) and bn in (0, 64, 128) and 1 <= ns <= 32
        assert st in (0, 2, 3, 4, 5, 6, 10, 11) and (bm != 256 or ks == 3) and (st < 10 or bm == 128) and (st < 5 or st >= 10 or bm < 256)
        assert 0 <= var1 <= 1024
        assert (M, N, K, ks, var1) not in seen, f"duplicate shape {ln}"
        seen.add((M, N, K, ks, var1))


def test_unet_plan_folds_layernorm_into_its_gemms(monkeypatch):
    """With the LayerNorm fold (default) a transformer block issues no layernorm launch: proj_in / to_out carry
    stats_out, the q|k|v, cross-attention q and GEGLU GEMMs carry ln_stats + ln_s.  MDX_UNET_LN_FOLD=0 restores the three
    explicit launches per block (attention.py:176-185)."""
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_numpy

    def plan(fold):
        monkeypatch.setenv("MDX_UNET_LN_FOLD", "1" if fold else "0")
        net = UNetModel(device="cpu", **TINY_UNET)
        net.load_state_dict(synthetic_unet_params_numpy(net.parameter_shapes(), 0))
        return net, net._plan(2, 8, 8)
    net, P = plan(True)
    blocks = sum(1 for k in net.w if k.endswith("attn2.q.w"))
    assert blocks >= 3
    assert sum(m["kind"] == "layernorm" for m in P.meta) == 0
    assert sum(1 for d in P.descs if d.ln_stats) == 3 * blocks and sum(1 for d in P.descs if d.stats_out) == 3 * blocks
    for d in P.descs:
        if d.ln_stats:
            assert d.ln_s and d.ln_nt * 64 == d.c1 and abs(d.ln_eps - 1e-5) < 1e-12 and d.bias
    _, P0 = plan(False)
    assert sum(m["kind"] == "layernorm" for m in P0.meta) == 3 * blocks
    assert not any(d.ln_stats or d.stats_out for d in P0.descs)
    assert len(P0.main) == len(P.main) + 3 * blocks


def test_tuned_table_drives_the_split_choice():
    """mdx_gemm_workspace_bytes (host only, no launch) must follow csrc/gemm_tuned.inc for the shapes it lists: 0 when the entry
    says one split, else the MDX_GEMM_WS_HEAD reserved bytes + splitk partials of the tile-padded output (which cover the
    [M][N] slabs of the reduce-kernel form as well)."""
    from minddiffusion_amd import _lib, ops
    lib = _lib.load()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "minddiffusion_amd", "csrc",
                        "gemm_tuned.inc")
    rows = [tuple(int(g or 0) for g in m.groups()) for m in
            (re.match(r"\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, \d+)?\},", ln) for ln in open(path)) if m]
    plain = {r[:4] for r in rows if r[7] in (0, 1)}
    checked = 0
    for M, N, K, ks, bm, bn, ns, var1 in rows:
        # dense rows of the plain variant (or wildcard rows of shapes without a plain-variant row) pin the hook-up
        if ks != 1 or M % 64 or K % 64 or var1 > 1 or (var1 == 0 and sum(1 for r in rows if r[:4] == (M, N, K, ks)) > 1):
            continue
        a = torch.zeros((1,), dtype=torch.float16)      # pointers are never dereferenced on this path
        d = ops.make_gemm_desc(a, a, N, 1, M, 1, K, a, N)
        need = lib.mdx_gemm_workspace_bytes(ctypes.byref(d))
        tm, tn = ops.gemm_query(d)[:2]
        assert (bm in (0, tm)) and (bn in (0, tn))
        padded = -(-M // tm) * tm * -(-N // tn) * tn
        assert need == (16384 + ns * padded * 4 if ns > 1 else 0), (M, N, K, bm, bn, ns, need)
        checked += 1
        if checked >= 12:
            break
    assert checked >= 6


def test_split_k_form_is_decided_on_the_host():
    """mdx_gemm_query (host only): a split launch of at most 4 splits with a row-major output reduces in the kernel (7th output),
    deeper splits, transposed outputs and deferred reduces keep the [split][M][N] slabs + reduce kernel; the workspace the
    library asks for covers the reserved head (MDX_GEMM_WS_HEAD) and the tile-padded partials; a workspace that is too small for a
    forced split is an error, not a silent fallback."""
    from minddiffusion_amd import _lib, ops
    lib = _lib.load()
    a = torch.zeros((1,), dtype=torch.float16)      # pointers are never dereferenced on this path
    M, N, K = 304, 200, 4096

    def desc(splitk, **kw):
        d = ops.make_gemm_desc(a, a, N, 1, M, 1, K, a, kw.pop("out_ld", N), splitk=splitk, tile_m=128, tile_n=128, **kw)
        d.workspace, d.workspace_bytes = 4096, 1 << 30
        return d
    for splitk, inkernel in ((1, 0), (2, 1), (4, 1), (5, 0), (16, 0)):
        q = ops.gemm_query(desc(splitk))
        assert q[:3] == (128, 128, splitk) and q[6] == inkernel, (splitk, q)
        need = lib.mdx_gemm_workspace_bytes(ctypes.byref(desc(splitk)))
        per = 3 * 128 * 2 * 128       # tile-padded partials (304 x 200 in 128 x 128 tiles): they cover the [M][N] slabs too
        assert need == (0 if splitk == 1 else 16384 + splitk * per * 4)
    assert ops.gemm_query(desc(3, out_mode=ops.OUT_TRANSPOSED, out_ld=M))[6] == 0
    d = desc(3)
    d.defer_reduce = 1
    assert ops.gemm_query(d)[6] == 0
    small = desc(4)
    small.workspace_bytes = 16384 + 2 * 3 * 128 * 2 * 128 * 4     # room for two partials only
    with pytest.raises(_lib.MdxError, match="workspace too small"):
        ops.gemm_query(small)
    auto = desc(0)
    auto.workspace_bytes = 16384 + 2 * 3 * 128 * 2 * 128 * 4     # the auto choice is clamped to what fits instead
    assert 1 <= ops.gemm_query(auto)[2] <= 2


def test_checkpoints_with_reference_prefixes_load_all_models(tmp_path):
    """MindSpore .ckpt files keyed the way the reference's CLIs find them load into the mirrors --
    `model.diffusion_model.` / `first_stage_model.` / `cond_stage_model.` for LatentDiffusion (ddpm.py:75,350), and the
    Taichu-GLIDE training-wrapper names that src/txt2img.py:34-57 rewrites (drop `diffusion_with_p_sample`, insert
    `model` after `guider_net` for the base model).  Renamed / surplus keys must be reported, not dropped."""
    from oracle import glide as OG
    from oracle import ldm as OL
    from minddiffusion_amd import ms_checkpoint as C
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.glide import diffusion_creator as DC
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    # ---- LatentDiffusion checkpoint -> UNetModel
    ocfg = dict(TINY_UNET, num_heads=TINY_UNET.get("num_heads", -1), num_head_channels=TINY_UNET.get("num_head_channels", -1))
    up = OL.init_params(ocfg, seed=0)
    ck = {C.UNET_PREFIX + k: v for k, v in up.items()}
    ck["first_stage_model.decoder.conv_in.bias"] = np.zeros(3, np.float32)
    path = str(tmp_path / "ldm.ckpt")
    C.save_checkpoint(ck, path, slice_bytes=1 << 16)
    unet_sd, vae_sd, _ = C.load_latent_diffusion(path)
    net = UNetModel(device="cpu", **TINY_UNET)
    net.load_state_dict(unet_sd)
    assert sorted(unet_sd) == sorted(net.parameter_shapes()) and list(vae_sd) == ["decoder.conv_in.bias"]
    renamed = dict(unet_sd)
    renamed["out.2.conv.biass"] = renamed.pop("out.2.conv.bias")
    with pytest.raises(KeyError, match="missing"):
        net.load_state_dict(renamed)
    with pytest.raises(KeyError, match="unexpected"):
        net.load_state_dict(dict(unet_sd, extra_key=np.zeros(1, np.float32)))
    net.load_state_dict(dict(unet_sd, extra_key=np.zeros(1, np.float32)), strict=False)   # surplus keys tolerated on request
    # ---- GLIDE: key rewrite (pure function), then both models through ckpt_path
    assert DC.rewrite_checkpoint_keys({"diffusion_with_p_sample.p_mean_variance.guider_net.out2.conv.weight": 1}, "base") \
        == {"p_mean_variance.guider_net.model.out2.conv.weight": 1}
    assert DC.rewrite_checkpoint_keys({"diffusion_with_p_sample.p_mean_variance.guider_net.out2.conv.weight": 1}, "supres") \
        == {"p_mean_variance.guider_net.out2.conv.weight": 1}
    otiny = dict(OG.BASE_OPTIONS, image_size=16, model_channels=64, num_res_blocks=1, channel_mult=(1, 2),
                 attention_resolutions=(1, 2), text_ctx=16, xf_width=64, xf_layers=2, xf_heads=1, n_vocab=100)
    bp = OG.init_params(otiny, seed=0)
    wrapped = {"diffusion_with_p_sample.p_mean_variance.guider_net." + k: v for k, v in bp.items()}
    wrapped["global_step"] = np.array([7], np.int64)                     # optimizer / bookkeeping entries are ignored
    bpath = str(tmp_path / "glide_base.ckpt")
    C.save_checkpoint(wrapped, bpath)
    opts = dict(TINY_GLIDE, device="cpu", noise_schedule="squaredcos_cap_v2", diffusion_steps=1000, timestep_respacing="10")
    m = DC.init_diffusion_model(opts, 3.0, (4, 3, 16, 16), ckpt_path=bpath)
    ref = DC.create_model(**opts)
    ref.load_state_dict(bp)
    for k in ref.w:
        assert torch.equal(m.model.w[k], ref.w[k]), k
    oup = dict(otiny, in_channels=6, image_size=32, channel_mult=(1, 1, 2))
    sp = OG.init_params(oup, seed=1)
    uopts = dict(opts, image_size=32, channel_mult=(1, 1, 2), low_size=8, noise_schedule="linear", timestep_respacing="fast27")
    for name, sd in (("wrapped", {"diffusion_with_p_sample.p_mean_variance.guider_net." + k: v for k, v in sp.items()}),
                     ("bare", sp)):                                     # src/txt2img.py:104 and diffusion_creator.py:49-50
        spath = str(tmp_path / f"glide_up_{name}.ckpt")
        C.save_checkpoint(sd, spath)
        s = DC.init_super_res_model(uopts, (2, 3, 32, 32), ckpt_path=spath)
        refu = DC.create_upsample_model(**uopts)
        refu.load_state_dict(sp)
        for k in refu.w:
            assert torch.equal(s.model.w[k], refu.w[k]), (name, k)
    # a base checkpoint offered as the up-sampler (or a truncated file) must raise, never load silently
    with pytest.raises((KeyError, ValueError)):
        DC.init_super_res_model(uopts, (2, 3, 32, 32), ckpt_path=bpath)
    short = dict(list(wrapped.items())[:-5])
    C.save_checkpoint(short, bpath)
    with pytest.raises(KeyError):
        DC.init_diffusion_model(opts, 3.0, (4, 3, 16, 16), ckpt_path=bpath)
    # strip_prefix filters BEFORE validation: a malformed entry outside the prefix does not abort the load
    good = {C.UNET_PREFIX + "a": np.ones(2, np.float32)}
    p3 = str(tmp_path / "mixed.ckpt")
    C.save_checkpoint(good, p3)
    bad_tensor = bytes([0x08, 5]) + bytes([0x12, 7]) + b"Float32" + bytes([0x1A, 4]) + bytes(4)   # dims [5], 1 element
    value = bytes([0x0A, 5]) + b"other" + bytes([0x12, len(bad_tensor)]) + bad_tensor
    with open(p3, "ab") as f:
        f.write(bytes([0x0A, len(value)]) + value)
    assert list(C.load_checkpoint(p3, strip_prefix=C.UNET_PREFIX)) == ["a"]
    with pytest.raises(ValueError):
        C.load_checkpoint(p3)


def _checkpoint_proto_classes(packed_dims):
    """The published MindSpore schema (mindspore/ccsrc/utils/checkpoint.proto) as a protobuf descriptor built at test time:
    an INDEPENDENT serializer (google.protobuf's own encoder) for the files ms_checkpoint.load_checkpoint must read."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = f"mdx_test_checkpoint_{int(packed_dims)}.proto"
    fd.package = f"mdx_test_ckpt{int(packed_dims)}"
    fd.syntax = "proto2"
    ck = fd.message_type.add()
    ck.name = "Checkpoint"
    val = ck.nested_type.add()
    val.name = "Value"
    tp = fd.message_type.add()
    tp.name = "TensorProto"
    for name, num, label, typ in (("dims", 1, F.LABEL_REPEATED, F.TYPE_INT64), ("tensor_type", 2, F.LABEL_REQUIRED, F.TYPE_STRING),
                                  ("tensor_content", 3, F.LABEL_REQUIRED, F.TYPE_BYTES)):
        f = tp.field.add()
        f.name, f.number, f.label, f.type = name, num, label, typ
        if name == "dims" and packed_dims:
            f.options.packed = True
    f = val.field.add()
    f.name, f.number, f.label, f.type = "tag", 1, F.LABEL_REQUIRED, F.TYPE_STRING
    f = val.field.add()
    f.name, f.number, f.label, f.type, f.type_name = "tensor", 2, F.LABEL_REQUIRED, F.TYPE_MESSAGE, f".{fd.package}.TensorProto"
    f = ck.field.add()
    f.name, f.number, f.label, f.type, f.type_name = "value", 1, F.LABEL_REPEATED, F.TYPE_MESSAGE, f".{fd.package}.Checkpoint.Value"
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName(f"{fd.package}.Checkpoint"))


def _write_like_mindspore(Checkpoint, params, path, slice_elems=None, types=None):
    """What mindspore.train.serialization._exec_save does (restated from its published source): one serialized Checkpoint
    message per parameter -- per SLICE of a parameter larger than the slice size, every slice under the same tag with the full
    dims -- written back to back.  `types` overrides tensor_type per name (BFloat16 payloads are raw uint16)."""
    ms_name = {"float32": "Float32", "float16": "Float16", "float64": "Float64", "int64": "Int64", "int32": "Int32",
               "uint8": "UInt8", "bool": "Bool"}
    with open(path, "wb") as f:
        for name, arr in params.items():
            arr = np.asarray(arr)
            ttype = (types or {}).get(name, ms_name.get(arr.dtype.name))
            flat = arr.reshape(-1)
            step = slice_elems or max(flat.size, 1)
            for off in range(0, max(flat.size, 1), step):
                msg = Checkpoint()
                v = msg.value.add()
                v.tag = name
                v.tensor.dims.extend(int(d) for d in arr.shape)
                v.tensor.tensor_type = ttype
                v.tensor.tensor_content = flat[off:off + step].tobytes()
                f.write(msg.SerializeToString())


@pytest.mark.parametrize("packed_dims", [False, True])
def test_checkpoint_reader_against_independent_protobuf_writer(tmp_path, packed_dims):
    """a checkpoint whose bytes never passed through ms_checkpoint.save_checkpoint.  The file is produced
    by google.protobuf from the published checkpoint.proto schema the way MindSpore's save_checkpoint lays messages out
    (stablediffusionv2/txt2img.py:52-63 and Taichu-GLIDE/src/txt2img.py:34-57 read such files with ms.load_checkpoint): sliced
    parameters, Float16 / BFloat16 / Int64 payloads, scalar (dims-less) entries, proto2 unpacked AND packed `dims`.  It is
    loaded through load_checkpoint, load_latent_diffusion -> UNetModel.load_state_dict, and glide.load_ckpt."""
    from oracle import glide as OG
    from oracle import ldm as OL
    from minddiffusion_amd import ms_checkpoint as C
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.glide import diffusion_creator as DC
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    Checkpoint = _checkpoint_proto_classes(packed_dims)
    rng = np.random.RandomState(5)
    # ---- raw reader: every payload type, slicing, scalars, negative ints, a unicode tag
    bf = rng.standard_normal((3, 5)).astype(np.float32)
    bf_bits = (bf.view(np.uint32) >> 16).astype(np.uint16)                   # truncated bfloat16 payload
    raw = {"a.f32": rng.standard_normal((4, 3, 2)).astype(np.float32), "b.f16": rng.standard_normal((7,)).astype(np.float16),
           "c.bf16": bf_bits, "d.i64": np.array([[-3, 2 ** 40], [7, -(2 ** 50)]], np.int64), "e.scalar": np.float32(2.5),
           "f.big": rng.standard_normal((33, 17)).astype(np.float32), "g.参数": np.arange(6, dtype=np.int32).reshape(2, 3),
           "h.bool": np.array([True, False, True])}
    p1 = str(tmp_path / "raw.ckpt")
    _write_like_mindspore(Checkpoint, raw, p1, slice_elems=50, types={"c.bf16": "BFloat16"})
    got = C.load_checkpoint(p1)
    assert list(got) == list(raw)
    for k, v in raw.items():
        if k == "c.bf16":
            np.testing.assert_array_equal(got[k], (bf_bits.astype(np.uint32) << 16).view(np.float32))
            assert got[k].dtype == np.float32 and got[k].shape == (3, 5)
        else:
            np.testing.assert_array_equal(got[k], np.asarray(v))
            assert got[k].dtype == np.asarray(v).dtype and got[k].shape == np.asarray(v).shape, k
    # the protobuf runtime parses our own writer's bytes too (the two serializers agree in both directions)
    p1b = str(tmp_path / "ours.ckpt")
    C.save_checkpoint({k: v for k, v in raw.items() if k != "c.bf16"}, p1b, slice_bytes=64)
    msg = Checkpoint()
    msg.MergeFromString(open(p1b, "rb").read())          # concatenated messages merge into one repeated field
    assert [v.tag for v in msg.value][:2] == ["a.f32", "a.f32"] and msg.value[0].tensor.tensor_type == "Float32"
    assert list(msg.value[0].tensor.dims) == [4, 3, 2]
    # ---- LatentDiffusion checkpoint (fp16 weights, as the released SD / Wukong checkpoints are stored) -> UNetModel
    ocfg = dict(TINY_UNET, num_heads=TINY_UNET.get("num_heads", -1), num_head_channels=TINY_UNET.get("num_head_channels", -1))
    up = {k: v.astype(np.float16) for k, v in OL.init_params(ocfg, seed=3).items()}
    ck = {C.UNET_PREFIX + k: v for k, v in up.items()}
    ck[C.VAE_PREFIX + "decoder.conv_in.bias"] = np.zeros(3, np.float32)
    ck[C.TEXT_PREFIX + "transformer.positional_embedding"] = rng.standard_normal((77, 8)).astype(np.float32)
    ck["global_step"] = np.array([12345], np.int64)
    p2 = str(tmp_path / "ldm.ckpt")
    _write_like_mindspore(Checkpoint, ck, p2, slice_elems=1 << 12)
    unet_sd, vae_sd, text_sd = C.load_latent_diffusion(p2)
    assert sorted(unet_sd) == sorted(up) and list(vae_sd) == ["decoder.conv_in.bias"]
    assert list(text_sd) == ["transformer.positional_embedding"]
    for k in up:
        np.testing.assert_array_equal(unet_sd[k], up[k])
    net = UNetModel(device="cpu", **TINY_UNET).load_state_dict(unet_sd)
    ref = UNetModel(device="cpu", **TINY_UNET).load_state_dict(up)
    for k in ref.w:
        assert torch.equal(net.w[k], ref.w[k]), k
    # ---- Taichu-GLIDE base checkpoint with the training-wrapper names -> glide.load_ckpt's key rewrite
    otiny = dict(OG.BASE_OPTIONS, image_size=16, model_channels=64, num_res_blocks=1, channel_mult=(1, 2),
                 attention_resolutions=(1, 2), text_ctx=16, xf_width=64, xf_layers=2, xf_heads=1, n_vocab=100)
    bp = OG.init_params(otiny, seed=4)
    wrapped = {"diffusion_with_p_sample.p_mean_variance.guider_net." + k: v for k, v in bp.items()}
    p3 = str(tmp_path / "glide.ckpt")
    _write_like_mindspore(Checkpoint, wrapped, p3, slice_elems=1 << 11)
    opts = dict(TINY_GLIDE, device="cpu", noise_schedule="squaredcos_cap_v2", diffusion_steps=1000, timestep_respacing="10")
    m = DC.init_diffusion_model(opts, 3.0, (4, 3, 16, 16), ckpt_path=p3)
    refg = DC.create_model(**opts)
    refg.load_state_dict(bp)
    for k in refg.w:
        assert torch.equal(m.model.w[k], refg.w[k]), k


def test_library_options_replace_environment_variables():
    """the library reads no environment variables; its tuning switches are mdx_set_option names."""
    import subprocess
    from minddiffusion_amd import _lib, ops
    lib = _lib.load()
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms
    assert ops.get_option("gemm_splitk_fixup_max") == 4 and ops.get_option("gemm_tuned") == 1
    ops.set_option("gemm_splitk_fixup_max", 2)
    try:
        assert ops.get_option("gemm_splitk_fixup_max") == 2
        # the option reaches the launch decision: a 3-way split no longer reduces in the kernel
        d = ops.GemmDesc()
        d.a = d.w = d.out = 4096
        d.c1, d.N, d.B, d.H, d.W, d.ksize, d.stride, d.out_ld, d.splitk = 1280, 1280, 2, 256, 1, 1, 1, 1280, 3
        d.workspace, d.workspace_bytes = 4096, 64 << 20
        assert ops.gemm_query(d)[2] == 3 and ops.gemm_query(d)[6] == 0
        ops.set_option("gemm_splitk_fixup_max", 4)
        assert ops.gemm_query(d)[6] == 1
    finally:
        ops.set_option("gemm_splitk_fixup_max", 4)
    with pytest.raises(_lib.MdxError, match="unknown option"):
        ops.set_option("no_such_option", 1)
    assert lib.mdx_set_option(b"gemm_bm", 0) == 0
    # round 5: the issue-order switches exist, default on, and the option table and its name table stay in step (the last name
    # resolves, so no name is missing in front of it)
    for name in ("gn_prefetch", "gemm_dense_issue", "gemm_ln_prefetch", "attn_fast_stage"):
        assert ops.get_option(name) == 1, name
    assert ops.get_option("halo_nsb") == 0 and ops.get_option("attn_kv_split") == 1 and ops.get_option("gn_boost_mb") == 40


def test_unet_plan_fuses_the_320_channel_transformer_blocks():
    """Round 3, host side: at the level where a SpatialTransformer has 320 channels and the plan has >= 192 row blocks of 32 token
    rows (64 x 64 latent, UNet batch 2) the planner emits THREE ops per block -- fused head (GroupNorm .. q|k|v), self-attention,
    fused tail (to_out .. proj_out) -- instead of eleven (attention.py:83-84, 96-185, 212, 231-256); 640-channel blocks and plans
    with too few rows keep the unfused launches; unet_st_head / unet_st_tail = 0 switch the fusion off.  Nothing is launched."""
    from minddiffusion_amd import _lib, ops
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_numpy
    cfg = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[1, 2], num_res_blocks=1,
               channel_mult=[1, 2], num_head_channels=64, use_spatial_transformer=True, use_linear_in_transformer=True,
               transformer_depth=1, context_dim=1024, legacy=False)
    lib = _lib.load()

    def build():
        net = UNetModel(device="cpu", **cfg)
        return net.load_state_dict(synthetic_unet_params_numpy(net.parameter_shapes(), 0))
    launches = lambda P: sum(m["launches"] for m in P.meta)
    kinds = lambda P: {k: sum(m["kind"] == k for m in P.meta) for k in ("gemm", "groupnorm", "attention")}
    P = build()._plan(2, 64, 64)
    blocks320 = 3                                   # one on the way down, two on the way up (num_res_blocks + 1)
    assert len(P.tails) == blocks320 and len(P.heads_fused) == blocks320
    for td in P.tails:
        assert (td.C, td.heads, td.dim_head, td.tile_rows) == (320, 5, 64, 32) and td.B * td.tokens == 8192
        assert lib.mdx_st_tail_supported(td.C, td.heads, td.dim_head, td.tokens, td.tile_rows) == 1
    assert lib.mdx_st_tail_supported(640, 10, 64, 1024, 32) == 0 and lib.mdx_st_head_supported(640, 1024, 32) == 0
    infos = [m["info"] for m in P.meta]
    assert sum(i.startswith("st_head") for i in infos) == blocks320 and sum(i.startswith("st_tail") for i in infos) == blocks320
    pf = ops.get_option("unet_gn_proj_fuse")
    try:
        ops.set_option("unet_st_tail", 0)
        ops.set_option("unet_st_head", 0)
        ops.set_option("unet_gn_proj_fuse", 1024)
        P1 = build()._plan(2, 64, 64)               # launch-per-op transformers, SpatialTransformer.norm folded into proj_in (round 5)
        ops.set_option("unet_gn_proj_fuse", 0)
        P0x = build()._plan(2, 64, 64)              # ... with the GroupNorm launches of rounds 1-4 (cross-attention on its projection: round 6)
        ops.set_option("unet_xattn_fuse", 0)
        P0 = build()._plan(2, 64, 64)               # ... and with one launch per op, as in rounds 1-2
    finally:
        ops.set_option("unet_st_tail", -1)
        ops.set_option("unet_st_head", -1)
        ops.set_option("unet_gn_proj_fuse", pf)
        ops.set_option("unet_xattn_fuse", 1)
    assert not P0.tails and not P0.heads_fused
    # round 6: with head dim 64 every unfused transformer block's cross-attention rides on its query projection (one launch fewer)
    nx = sum("+cross-attention" in m["info"] for m in P0x.meta)
    assert nx == sum(m["info"].startswith("cross ") for m in P0.meta) > blocks320 and len(P0.main) - len(P0x.main) == nx
    assert all(m["desc"].xattn_k and m["desc"].tile_n == 64 and m["desc"].splitk == 1 for m in P0x.meta if "+cross-attention" in m["info"])
    # round 5: every transformer whose level has >= unet_gn_proj_fuse tokens per sample loses its GroupNorm launch to proj_in
    # (all of them here: 4096 and 1024 tokens), the GEMM count is unchanged, and the fused-head plan P already had none to lose at
    # the 320-channel level but folds the 640-channel ones
    folded1 = sum("+groupnorm(in)" in m["info"] for m in P1.meta)
    assert folded1 == sum(m["info"].startswith("self ") for m in P1.meta) and folded1 > blocks320
    assert kinds(P0)["groupnorm"] - kinds(P1)["groupnorm"] == folded1 and kinds(P0)["gemm"] == kinds(P1)["gemm"]
    assert len(P0x.main) - len(P1.main) == folded1 and not any("+groupnorm(in)" in m["info"] for m in P0.meta)
    assert not any("+groupnorm(in)" in m["info"] for m in P.meta)      # the option is off by default (measured equal-or-slower)
    try:
        ops.set_option("unet_gn_proj_fuse", 1024)
        Pf = build()._plan(2, 64, 64)
    finally:
        ops.set_option("unet_gn_proj_fuse", pf)
    assert sum("+groupnorm(in)" in m["info"] for m in Pf.meta) == folded1 - blocks320      # the fused heads already hold theirs
    for m in P1.meta:
        if "+groupnorm(in)" in m["info"]:
            d = m["desc"]
            assert d.gn_colstats and d.gn_nrb > 0 and d.gn_silu == 0 and d.ksize == 1 and (d.H * d.W) % ops.gemm_query(d)[0] == 0
    try:
        ops.set_option("unet_gn_proj_fuse", 0)
        ops.set_option("unet_xattn_fuse", 0)
        P = build()._plan(2, 64, 64)                # (the comparison below is about the fused head / tail alone)
    finally:
        ops.set_option("unet_gn_proj_fuse", pf)
        ops.set_option("unet_xattn_fuse", 1)
    k, k0 = kinds(P), kinds(P0)
    # per fused block: 8 GEMMs (proj_in, q|k|v, to_out, q, to_out, ff1, ff2, proj_out) become 2, the GroupNorm and the
    # cross-attention launch disappear
    assert k0["gemm"] - k["gemm"] == 6 * blocks320 and k0["groupnorm"] - k["groupnorm"] == blocks320
    assert k0["attention"] - k["attention"] == blocks320
    assert launches(P0) - launches(P) == 8 * blocks320
    # a 16 x 16 latent has 16 row blocks of 32 rows: far too few to fill the chip -> unfused
    Ps = build()._plan(2, 16, 16)
    assert not Ps.tails and not Ps.heads_fused


def test_fused_transformer_weight_streams_follow_the_documented_layout():
    """ops.pack_frag_weight / pack_st_tail / pack_st_head against an index-level restatement of DESIGN.md section 2: piece (column
    tile ct, k-step s) holds, for lane l, W[32 ct + l % 32][16 s + 8 (l // 32) + 0..7] (the operand v_mfma_f32_32x32x16_f16 takes
    from that lane); a wave's stream is its pieces of to_out1, to_q2, to_out2, then per hidden chunk the ff1 'a' and 'gate' pieces
    of a k-step side by side followed by the chunk's ff2 pieces, then proj_out -- and the library's stream-size entry points agree
    with the packers (the kernels bound their buffer loads by these sizes).  Host only."""
    from minddiffusion_amd import _lib, ops
    lib = _lib.load()
    rng = np.random.RandomState(12)
    C = 64                                           # layout rules do not depend on C; the kernels themselves need 320
    NW, KS = C // 32, C // 16
    mk = lambda n, k: torch.tensor(rng.standard_normal((n, k)).astype(np.float16))
    w = mk(96, 48)
    f = ops.pack_frag_weight(w)
    assert f.shape == (3, 3, 512)
    for ct, s, lane, e in [(0, 0, 0, 0), (2, 1, 37, 5), (1, 2, 63, 7), (2, 0, 31, 3), (0, 2, 32, 0)]:
        assert f[ct, s, lane * 8 + e] == w[32 * ct + lane % 32, 16 * s + 8 * (lane // 32) + e]
    wo1, wq2, wo2, wpo = (mk(C, C) for _ in range(4))
    w1, w2 = mk(8 * C, C), mk(C, 4 * C)
    vecs = [torch.tensor(rng.standard_normal(n).astype(np.float32)) for n in (C, C, C, C, C, C, 8 * C, C, C)]
    stream, vec = ops.pack_st_tail(wo1, wq2, wo2, w1, w2, wpo, *vecs)
    assert stream.shape == (NW, 16 * KS, 512) and vec.numel() == 16 * C
    assert torch.equal(vec, torch.cat(vecs))
    piece = lambda W, wave, s: ops.pack_frag_weight(W)[wave, s]
    for wave in range(NW):
        pos = 0
        for W in (wo1, wq2, wo2):                    # three C x C units, KS pieces each
            for s in range(KS):
                assert torch.equal(stream[wave, pos], piece(W, wave, s))
                pos += 1
        for c in range(4):                           # hidden chunk c: columns [cC, cC + C) of 'a' and of 'gate'
            a_rows, g_rows = w1[c * C:(c + 1) * C], w1[4 * C + c * C:4 * C + (c + 1) * C]
            for s in range(KS):
                assert torch.equal(stream[wave, pos], piece(a_rows, wave, s))
                assert torch.equal(stream[wave, pos + 1], piece(g_rows, wave, s))
                pos += 2
            for s in range(KS):                      # ff2 over this chunk's K range
                assert torch.equal(stream[wave, pos], piece(w2[:, c * C:(c + 1) * C], wave, s))
                pos += 1
        for s in range(KS):
            assert torch.equal(stream[wave, pos], piece(wpo, wave, s))
            pos += 1
        assert pos == 16 * KS
    hs, hv = ops.pack_st_head(wo1, wq2, wo2, wpo, *vecs[:5])
    assert hs.shape == (NW, 4 * KS, 512) and hv.numel() == 5 * C
    for wave in range(NW):
        for u, W in enumerate((wo1, wq2, wo2, wpo)):
            for s in range(KS):
                assert torch.equal(hs[wave, u * KS + s], piece(W, wave, s))
    for c in (320, 640):                             # 16 C^2 (tail) / 4 C^2 (head) fp16 elements
        assert lib.mdx_st_tail_stream_bytes(c) == 16 * c * c * 2 and lib.mdx_st_head_stream_bytes(c) == 4 * c * c * 2
    assert stream.numel() * 2 == 16 * C * C * 2 and hs.numel() * 2 == 4 * C * C * 2


def test_bench_uses_the_committed_pmc_passes_only_for_the_launch_mix_they_were_taken_on():
    """bench.py attaches `roofline.traffic` and per-family `mfma_busy` from the rocprofv3 PMC passes committed under profiles/
    (counters cannot be read inside the timed process).  The files carry their own GEMM-family launch count per evaluation; a
    build whose plan issues another count (a different launch mix) must get nulls and the reason, not stale numbers."""
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(os.path.join(root, "profiles", bench.PMC_FILE)))
    n = doc["families"]["gemm"]["launches_per_eval"] + doc["families"]["splitk_reduce"]["launches_per_eval"]
    per_launch, note = bench.pmc_traffic(round(n))
    assert per_launch and 1e6 < per_launch < 1e9 and bench.PMC_FILE in note
    total = sum(doc["families"][k][f] for k in ("gemm", "splitk_reduce") for f in ("read_MB_per_eval", "write_MB_per_eval"))
    assert abs(per_launch * n - total * 1e6) <= n
    stale, why = bench.pmc_traffic(round(n) + 40)
    assert stale is None and "stale" in why
    roof = {"families": {"gemm": {}, "attention": {}, "groupnorm": {}}}
    bench.pmc_mfma_busy(roof, round(n))
    assert 0.0 < roof["mfma_busy"] < 1.0 and 0.0 < roof["families"]["attention"]["mfma_busy"] < 1.0
    roof2 = {"families": {"gemm": {}}}
    bench.pmc_mfma_busy(roof2, round(n) + 40)
    assert "mfma_busy" not in roof2 and "stale" in roof2["mfma_busy_note"]


def test_subpixel_conv_weight_is_the_upsampled_conv():
    """ops.subpixel_conv_weight (mdx_gemm_desc.w_sub): nearest-2x followed by a 3 x 3 conv, pad 1 (Upsample.construct,
    openaimodel.py:57-60) equals, for each output parity (dy, dx), a 2 x 2 conv of the low-resolution tensor at rows
    {y - 1 + dy, y + dy} / columns {x - 1 + dx, x + dx} with the taps that fall on one source pixel summed -- checked in float64
    against torch's own upsample + conv, borders included."""
    import torch.nn.functional as F
    from minddiffusion_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 6, 7, generator=g, dtype=torch.float64)
    w = torch.randn(3, 5, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    ws = ops.subpixel_conv_weight(w.to(torch.float32)).to(torch.float64).reshape(2, 2, 3, 5, 2, 2)
    out = torch.zeros_like(ref)
    xp = F.pad(x, (1, 1, 1, 1))
    for dy in range(2):
        for dx in range(2):
            # taps (a, b) read padded rows y + dy + a, columns x + dx + b  (padded index = source index + 1)
            o = F.conv2d(xp[:, :, dy:dy + 6 + 1, dx:dx + 7 + 1], ws[dy, dx])
            out[:, :, dy::2, dx::2] = o
    assert torch.allclose(out, ref, atol=1e-5), float((out - ref).abs().max())


def test_lean_dense_kernel_wait_counts_equal_the_loads_it_issues(tmp_path):
    """csrc/dense.hip counts its epilogue prefetches in the FIRST wait of the K loop (s_waitcnt vmcnt(tiles in flight + E)): if the
    compiler removed or added one of those loads, the wait would let the first K tile be read before it has landed.  Compile the file to
    gfx950 assembly and check, for every instantiation, that the non-LDS buffer loads in front of the first barrier are exactly E, that
    nothing spills, and that the first wait names E."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "dense.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include",
                    f"-I{ROOT}/minddiffusion_amd/csrc", "-Wno-unused-function", "-S", "--cuda-device-only", "-o", str(out),
                    f"{ROOT}/minddiffusion_amd/csrc/dense.hip"], check=True, capture_output=True)
    txt = out.read_text()
    n = 0
    for m in re.finditer(r"^_ZN12_GLOBAL__N_112dense_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d)ELb[01]EEEvN7mdx_int10GemmParamsE:\s*;", txt, re.M):
        bm, bn, ns, pf = (int(m.group(i)) for i in (1, 2, 3, 4))
        body = txt[m.end():txt.index(".end_amdhsa_kernel", m.end())]
        lines = body.split("\n")
        first_bar = next(i for i, l in enumerate(lines) if "s_barrier" in l)
        pre = lines[:first_bar]
        issued = sum(1 for l in pre if "buffer_load" in l and " lds" not in l)
        assert not any("global_load" in l or "scratch_" in l for l in pre), (bm, bn, ns, pf)
        passes = bm // (256 // (bn // 8))
        nx = min(passes, 4) if pf >= 2 else 0
        E = (4 if bn == 128 else 2) + nx + (24 if pf >= 3 else 0) + (1 if pf >= 2 else 0) + (2 if pf in (1, 2) else 0)
        assert issued == E, f"dense_kernel<{bm},{bn},{ns},{pf}>: {issued} prefetch loads issued, the first wait counts {E}"
        waits = [int(x) for l in pre for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", l)]
        lpt = (bm + bn) // 8 // 4
        assert waits and all((w - E) % lpt == 0 and 0 <= (w - E) // lpt <= ns - 2 for w in waits), (bm, bn, ns, pf, waits, E)
        assert re.search(r"ScratchSize: (\d+)", txt[m.end():]).group(1) == "0", f"dense_kernel<{bm},{bn},{ns},{pf}> spills"
        n += 1
    assert n >= 50, n

