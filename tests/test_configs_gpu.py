"""One parity test per BASELINE.json config AT ITS BENCHMARKED SHAPE (bench.py CONFIGS): the UNet batch, latent size and
sampler that `bench.py --config ...` times, through the same plan / hipGraph / tile-table dispatch -- so every row of
csrc/gemm_tuned.inc that a benchmark exercises is also exercised (and asserted to be hit) under a parity check.

How a full batch is checked without running the fp32 CPU oracle on every row (0.8-2.1 TFLOP per row on the host):
  * ORACLE rows: one or two rows of the full-batch HIP output are compared with oracle rows computed on the CPU
    (same tolerance as the B = 1 tests: rel-L2 <= 5e-3);
  * CONSISTENCY rows: further rows of the full-batch output are compared with B = 1 HIP evaluations of the same inputs
    (which test_unet_gpu.py pins against the oracle): equal up to fp16 rounding of the different tile / split-K choices,
    rel-L2 <= 4e-3 (two fp16 paths, each ~2e-3 from the oracle, measured 1.7-2.1e-3);
  * trajectories run at the benchmarked batch, the oracle follows image 0 (trajectories of different images are
    independent).
Measured values go to gpurun_out/parity_log.jsonl (copied to profiles/parity_r02.jsonl).
"""
import os
import re

import numpy as np
import pytest
import torch

from _util import ROOT, check
from oracle import glide as OG
from oracle import ldm as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def metrics_rel(got, ref):
    from _util import metrics
    return metrics(got, ref)["rel_l2"]


def _short_fixture(key, S, compute):
    """Oracle end point (final latent, last pred_x0) of one of the SHORT full-size trajectories: from tests/golden/short_traj.npz
    (written by tests/golden/make_trajectory_goldens.py short -- same UNet seeds and inputs; the oracle's fp32 outputs) when the
    fixture holds it with the expected step count, else computed here (compute() -> (final, pred_x0))."""
    import json as _json
    path = os.path.join(ROOT, "tests", "golden", "short_traj.npz")
    if os.path.exists(path):
        z = np.load(path)
        meta = _json.loads(str(z["meta"]))["cases"].get(key)
        if meta is not None and meta["S"] == S and key + "_final" in z.files:
            return torch.tensor(z[key + "_final"]), torch.tensor(z[key + "_pred_x0"])
    return compute()


def _threads():
    torch.set_num_threads(min(96, os.cpu_count() or 8))


def _tuned_table():
    rows = {}
    for ln in open(os.path.join(ROOT, "minddiffusion_amd", "csrc", "gemm_tuned.inc")):
        m = re.match(r"\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, \d+)?\}", ln)
        if m:
            v = [int(g or 0) for g in m.groups()]
            rows.setdefault(tuple(v[:4]), []).append(tuple(v[4:]))
    return rows


def _assert_tuned_rows_hit(plan, name, min_hits):
    """Every descriptor of the plan whose (M, N, K, ksize) is in the measured tile table must resolve to that row's
    tile_m / tile_n / split (mdx_gemm_query = the decision path of mdx_gemm_f16 without the launch)."""
    from minddiffusion_amd import ops
    table = _tuned_table()
    hits, kinds = 0, set()
    for d in plan.descs:
        M = d.B * d.H * d.W
        key = (M, d.N, d.ksize * d.ksize * (d.c1 + d.c2), d.ksize)
        tm, tn, ns, halo, tuned = ops.gemm_query(d)[:5]
        kinds.add((tm, tn, halo))
        if key in table and d.stride == 1 and not d.upsample and tuned:
            # one of the shape's rows (exact launch variant, or the variant-less row) must be what the library resolved to
            ok = [(bm, bn, tns) for bm, bn, tns, _ in table[key]
                  if tm == bm and (bn == 0 or tn == bn) and 1 <= ns <= max(tns, 1) and (ns > 1) == (tns > 1)]
            assert ok, (name, key, (tm, tn, ns), table[key])
            hits += 1
        else:
            assert not tuned or key in table
    print("PARITY", {"name": f"{name}_tuned_rows_hit", "hits": hits, "descs": len(plan.descs), "tile_kinds": sorted(kinds)})
    assert hits >= min_hits, f"{name}: only {hits} descriptors hit the tuned table (expected >= {min_hits})"
    return hits


def _unet(cfg, ocfg, seed):
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    params = O.init_params(ocfg, seed=seed)
    net = UNetModel(**dict(cfg))
    net.use_graph = True
    net.load_state_dict(params)
    return net, O.UNetOracle(ocfg, params)


def _ldm_full_batch_case(name, cfg, ocfg, B, hw, ctx_dim, t, oracle_rows, consistency_rows, seed, min_hits):
    _threads()
    net, oracle = _unet(cfg, ocfg, seed)
    rng = np.random.RandomState(100 + seed)
    x = rng.randn(B, 4, hw, hw).astype(np.float32)
    ctx = rng.randn(B, 77, ctx_dim).astype(np.float32)
    ts = np.full((B,), t, np.float32)
    dev = lambda a: torch.tensor(a, device=DEV)
    got = net(dev(x), dev(ts), dev(ctx)).cpu()
    assert net._plans[(B, hw, hw)].graph is not None, "the benchmarked path replays a hipGraph"
    _assert_tuned_rows_hit(net._plans[(B, hw, hw)], name, min_hits)
    for r in oracle_rows:
        ref = oracle(x[r:r + 1], torch.tensor(ts[r:r + 1]), ctx[r:r + 1])
        check(f"{name}_B{B}_row{r}_vs_oracle", got[r:r + 1], ref, rel_l2=5e-3, max_abs=5e-2)
    for r in consistency_rows:
        one = net(dev(x[r:r + 1]), dev(ts[r:r + 1]), dev(ctx[r:r + 1])).cpu()
        check(f"{name}_B{B}_row{r}_vs_B1_hip", got[r:r + 1], one, rel_l2=4e-3, max_abs=2e-2)
    return net, oracle


# --------------------------------------------------------------------------------------------- config 1: SDv2 512, DDIM-50, B=1
def test_config1_sd2_512_ddim10_cfg9_full_size_trajectory():
    """BASELINE configs[1] (the headline): SDv2 UNet, 64x64 latent, CFG 9.0, batch 1 (UNet batch 2), per-run time-embedding
    table + hipGraph = the path bench.py times -- ten DDIM steps against the oracle's sampler (20 oracle row evaluations),
    plus the tile-table assertion at UNet batch 2.  (Round 4: the oracle's end point comes from tests/golden/short_traj.npz when present.)"""
    from minddiffusion_amd.configs import SD2_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    _threads()
    net, oracle = _unet(SD2_UNET, O.SD2_UNET, 4)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    S, scale = 10, 9.0
    x_T = np.random.RandomState(42).randn(1, 4, 64, 64).astype(np.float32)
    c = np.random.RandomState(1).randn(1, 77, 1024).astype(np.float32)
    uc = np.random.RandomState(2).randn(1, 77, 1024).astype(np.float32)
    def compute():
        ref, ref_inter = O.sample(O.ModelOracle(oracle), S, 1, (4, 64, 64), c, x_T, "ddim", unconditional_guidance_scale=scale,
                                  unconditional_conditioning=uc)
        return ref, ref_inter["pred_x0"][-1]
    ref, ref_px0 = _short_fixture("config1_ddim10", S, compute)
    got, inter = DDIMSampler(model).sample(S, 1, (4, 64, 64), conditioning=torch.tensor(c, device=DEV),
                                           x_T=torch.tensor(x_T, device=DEV), unconditional_guidance_scale=scale,
                                           unconditional_conditioning=torch.tensor(uc, device=DEV), verbose=False)
    P = net._plans[(2, 64, 64)]
    assert P.graph is not None
    _assert_tuned_rows_hit(P, "config1_sd2_512_unet_b2", 8)
    check("config1_sd2_512_ddim10_cfg9_latent", got, ref, rel_l2=1e-2, max_rel=1e-2)   # measured 3.2e-3 / 3.7e-3
    check("config1_sd2_512_ddim10_cfg9_pred_x0", inter["pred_x0"][-1], ref_px0, rel_l2=2e-2)


def test_config1_plan_replay_stress_split_k_tickets():
    """2 000 replays of the hipGraph bench.py times (SDv2, UNet batch 2, 64 x 64: ~30 in-kernel split-K launches per replay
    share one workspace with every other GEMM of the plan) with the partial area of that workspace NaN-poisoned again and
    again between replays: the output must stay bit-identical to the first replay -- a partial read too early, a stale cache
    line or a ticket seen before its writer's stores would show as a NaN or a changed bit -- and agree with the same plan
    built with the separate reduce kernel (gemm_splitk_fixup_max = 0) to fp16 rounding."""
    from minddiffusion_amd import ops
    from minddiffusion_amd.configs import SD2_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    params = O.init_params(O.SD2_UNET, seed=4)
    rng = np.random.RandomState(3)
    x = torch.tensor(rng.randn(2, 4, 64, 64).astype(np.float32), device=DEV)
    ctx = torch.tensor(rng.randn(2, 77, 1024).astype(np.float32), device=DEV)
    t = torch.full((2,), 501.0, device=DEV)
    ops.set_option("gemm_splitk_fixup_max", 0)
    try:
        ref_net = UNetModel(**dict(SD2_UNET)).load_state_dict(params)
        ref = ref_net.forward_nhwc(x, t, ctx).clone()
        assert not any(ops.gemm_query(d)[6] for d in ref_net._plans[(2, 64, 64)].descs)
        del ref_net
    finally:
        ops.set_option("gemm_splitk_fixup_max", 4)
    net = UNetModel(**dict(SD2_UNET)).load_state_dict(params)
    first = net.forward_nhwc(x, t, ctx).clone()
    P = net._plans[(2, 64, 64)]
    assert P.graph is not None
    n_fix = sum(1 for d in P.descs if ops.gemm_query(d)[6])
    assert n_fix >= 20, f"expected the in-kernel split-K form on >= 20 launches of the plan, got {n_fix}"
    check("config1_plan_ticket_form_vs_reduce_kernel_form", first, ref, rel_l2=4e-3)    # two fp16 paths (measured 1.8e-3)
    for rep in range(2000):
        if rep % 7 == 0:
            P.gemm_ws.fill_(float("nan"))       # head included: the arrival counters are library-owned
        P.graph.replay()
        if rep % 250 == 249:
            torch.cuda.synchronize()
            assert torch.equal(P.eps_nhwc, first), f"replay {rep}: output changed"
    torch.cuda.synchronize()
    assert torch.equal(P.eps_nhwc, first)
    print("PARITY", {"name": "config1_plan_replay_stress", "replays": 2000, "in_kernel_splitk_launches_per_replay": n_fix})


@pytest.mark.parametrize("sampler", ["ddim", "plms"])
def test_config1_length_50_step_trajectory_tiny_unet(sampler):
    """The benchmarked LENGTH: 50 sampler steps (DDIM: 50 UNet calls, PLMS: 51) with CFG on the tiny UNet, at the stated
    trajectory bar (rel-L2 <= 1e-2 'after 50 steps')."""
    from minddiffusion_amd.configs import TINY_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    cfg = dict(TINY_UNET)
    ocfg = dict(cfg)
    ocfg.setdefault("num_heads", -1)
    ocfg.setdefault("num_head_channels", -1)
    net, oracle = _unet(cfg, ocfg, 3)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    omodel = O.ModelOracle(oracle)
    B, H, W, T, S, scale = 2, 8, 8, 6, 50, 7.5
    x_T = np.random.RandomState(42).randn(B, 4, H, W).astype(np.float32)
    c = np.random.RandomState(1).randn(B, T, cfg["context_dim"]).astype(np.float32)
    uc = np.repeat(np.random.RandomState(2).randn(1, T, cfg["context_dim"]).astype(np.float32), B, 0)
    omodel.calls = 0
    ref, _ = O.sample(omodel, S, B, (4, H, W), c, x_T, sampler, unconditional_guidance_scale=scale,
                      unconditional_conditioning=uc)
    assert omodel.calls == (S + 1 if sampler == "plms" else S)
    cls = PLMSSampler if sampler == "plms" else DDIMSampler
    got, _ = cls(model).sample(S, B, (4, H, W), conditioning=torch.tensor(c, device=DEV), x_T=torch.tensor(x_T, device=DEV),
                               unconditional_guidance_scale=scale, unconditional_conditioning=torch.tensor(uc, device=DEV),
                               verbose=False)
    check(f"config1_length_tiny_{sampler}_S50_cfg7.5", got, ref, rel_l2=1e-2, max_rel=1e-2)


# --------------------------------------------------------------------------------------------- config 2: Wukong 512, PLMS, B=8
def test_config2_wukong_512_unet_batch16_and_plms():
    """BASELINE configs[2]: Wukong-Huahua UNet (8 heads -> d = 40 / 80 / 160, 1x1-conv proj, ctx 768) at UNet batch 16
    (8 images x CFG), 64x64 latent; then FIVE PLMS steps at batch 8 -- pseudo improved Euler, AB-2, AB-3 and two AB-4 steps
    (wukong-huahua/ldm/models/diffusion/plms.py:231-244): 6 UNet calls at UNet batch 16 -- with the oracle following image 0
    (12 oracle rows)."""
    from minddiffusion_amd.configs import WUKONG_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    net, oracle = _ldm_full_batch_case("config2_wukong_512", WUKONG_UNET, O.WUKONG_UNET, 16, 64, 768, 301.0,
                                       oracle_rows=[3], consistency_rows=[0, 15], seed=2, min_hits=5)
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    S, scale, Bi = 5, 7.5, 8
    rng = np.random.RandomState(7)
    x_T = rng.randn(Bi, 4, 64, 64).astype(np.float32)
    c = rng.randn(Bi, 77, 768).astype(np.float32)
    uc = np.repeat(rng.randn(1, 77, 768).astype(np.float32), Bi, 0)
    got, _ = PLMSSampler(model).sample(S, Bi, (4, 64, 64), conditioning={"c_crossattn": [torch.tensor(c, device=DEV)]},
                                       x_T=torch.tensor(x_T, device=DEV), unconditional_guidance_scale=scale,
                                       unconditional_conditioning={"c_crossattn": [torch.tensor(uc, device=DEV)]}, verbose=False)
    ref, _ = _short_fixture("config2_plms5", S, lambda: (O.sample(O.ModelOracle(oracle), S, 1, (4, 64, 64), c[:1], x_T[:1], "plms",
                                                                  unconditional_guidance_scale=scale,
                                                                  unconditional_conditioning=uc[:1])[0], None))
    check("config2_wukong_512_plms5_B8_image0", got[:1], ref, rel_l2=1e-2, max_rel=2e-2)


# --------------------------------------------------------------------------------------------- every UNet batch 1..16 at 64 x 64
@pytest.mark.parametrize("model", ["sd2", "wukong"])
def test_full_size_unet_every_row_of_batches_that_are_not_benchmarked(model):
    """Round-5 regression (found by the full-size inpainting test, whose CLI default batch is 4 = UNet batch 8): the measured tile
    table is keyed by (M, N, K, ksize), and at UNet batch 8 the 8 x 8 level shares its key with 256-row rows measured at batch 2 on
    the 16 x 16 level; the rows 4-7 of every batch-8 evaluation were wrong (rel 0.5) while batches 2 / 4 / 6 / 12 / 16 -- the ones
    the benchmarks and the other tests run -- were right.  Here EVERY row of UNet batches 3, 8 and 10 is compared with its own
    batch-1 evaluation (which test_unet_gpu / the trajectory fixtures pin against the oracle): <= 4e-3, two fp16 paths."""
    from minddiffusion_amd.configs import SD2_UNET, WUKONG_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_device
    cfg, cd = (SD2_UNET, 1024) if model == "sd2" else (WUKONG_UNET, 768)
    net = UNetModel(**dict(cfg))
    net.use_graph = False
    net.load_state_dict(synthetic_unet_params_device(net.parameter_shapes(), seed=0, device=DEV))
    rng = np.random.RandomState(11)
    for B in (3, 8, 10):
        x = torch.tensor(rng.randn(B, 4, 64, 64).astype(np.float32), device=DEV)
        ctx = torch.tensor(rng.randn(B, 77, cd).astype(np.float32), device=DEV)
        t = torch.full((B,), 500.0, device=DEV)
        full = net(x, t, ctx).clone()
        worst = 0.0
        for r in range(B):
            one = net(x[r:r + 1].clone(), t[:1], ctx[r:r + 1].clone())
            worst = max(worst, metrics_rel(full[r:r + 1], one))
        print("PARITY", {"name": f"{model}_unet_batch{B}_worst_row_vs_batch1", "rel_l2": worst})
        assert worst <= 4e-3, (model, B, worst)


_SWEEP_PAIRS = [("sd2", 32, 3), ("sd2", 32, 8), ("sd2", 32, 13), ("sd2", 40, 5), ("sd2", 48, 4), ("sd2", 48, 10), ("sd2", 64, 5),
                ("sd2", 64, 8), ("sd2", 72, 3), ("wukong", 32, 4), ("wukong", 32, 16), ("wukong", 40, 6), ("wukong", 48, 7),
                ("wukong", 64, 4), ("wukong", 64, 12)]


@pytest.mark.parametrize("model", ["sd2", "wukong"])
def test_shape_sweep_subset_with_an_oracle_row_per_pair(model):
    """Round 6 (round-5 review item 6): a subset of tools/shape_sweep.py inside the suite, and not only batch-B against batch-1 of
    the SAME library -- per (latent, batch) pair that neither the benchmarks nor the other parity tests run, ONE row of the batch-B
    evaluation is compared with the fp32 oracle (rows rotate: B - 1, so that the rows a wrong tile-table row would leave unwritten --
    the second half of a tile pair, round 5 -- are the ones looked at) and every row with its own batch-1 evaluation.  The tile table
    is keyed by (M, N, K, ksize, launch variant): this is the check that a row measured at one (B, H, W) is right at another
    factorisation of M."""
    from minddiffusion_amd.configs import SD2_UNET, WUKONG_UNET
    _threads()
    cfg, ocfg, cd = (SD2_UNET, O.SD2_UNET, 1024) if model == "sd2" else (WUKONG_UNET, O.WUKONG_UNET, 768)
    net, oracle = _unet(cfg, ocfg, 9)
    net.use_graph = False
    dev = lambda a: torch.tensor(a, device=DEV)
    for (m, hw, B) in [q for q in _SWEEP_PAIRS if q[0] == model]:
        rng = np.random.RandomState(1000 * hw + B)
        x = rng.randn(B, 4, hw, hw).astype(np.float32)
        ctx = rng.randn(B, 77, cd).astype(np.float32)
        ts = np.full((B,), 400.0 + B, np.float32)
        full = net(dev(x), dev(ts), dev(ctx)).cpu()
        r = B - 1
        ref = oracle(x[r:r + 1], torch.tensor(ts[r:r + 1]), ctx[r:r + 1])
        check(f"sweep_{model}_latent{hw}_B{B}_row{r}_vs_oracle", full[r:r + 1], ref, rel_l2=5e-3, max_abs=5e-2)
        worst = 0.0
        for q in range(B):
            one = net(dev(x[q:q + 1]), dev(ts[q:q + 1]), dev(ctx[q:q + 1])).cpu()
            worst = max(worst, metrics_rel(full[q:q + 1], one))
        print("PARITY", {"name": f"sweep_{model}_latent{hw}_B{B}_worst_row_vs_batch1", "rel_l2": worst})
        assert worst <= 4e-3, (model, hw, B, worst)
        net._plans.clear()
        torch.cuda.empty_cache()


def test_full_size_unet_320_pixel_latent_with_ragged_token_counts():
    """320 x 320 pixels = a 40 x 40 latent: the reference takes any multiple of 64 pixels (txt2img.py --H / --W), and the 10 x 10 and
    5 x 5 levels then have 100 / 25 tokens per sample -- not multiples of 8 (round 5: planning used to refuse them).  Full SDv2
    UNet at batch 2, row 1 against the oracle, row 0 against its batch-1 evaluation."""
    from minddiffusion_amd.configs import SD2_UNET
    _threads()
    net, oracle = _unet(SD2_UNET, O.SD2_UNET, 3)
    rng = np.random.RandomState(40)
    x = rng.randn(2, 4, 40, 40).astype(np.float32)
    ctx = rng.randn(2, 77, 1024).astype(np.float32)
    ts = np.full((2,), 700.0, np.float32)
    dev = lambda a: torch.tensor(a, device=DEV)
    got = net(dev(x), dev(ts), dev(ctx)).cpu()
    assert len(net._plans[(2, 40, 40)].ragged_vt) > 0
    ref = oracle(x[1:2], torch.tensor(ts[1:2]), ctx[1:2])
    check("sd2_latent40_B2_row1_vs_oracle", got[1:2], ref, rel_l2=5e-3, max_abs=5e-2)
    one = net(dev(x[:1]), dev(ts[:1]), dev(ctx[:1])).cpu()
    check("sd2_latent40_B2_row0_vs_B1_hip", got[:1], one, rel_l2=4e-3, max_abs=2e-2)


# --------------------------------------------------------------------------------------------- Wukong inpainting, full size
def test_inpaint_wukong_full_size():
    """SURVEY 8(f) item 4 at full size: configs/wukong-huahua_inpaint_inference.yaml (the Wukong UNet on 9 input channels,
    LatentInpaintDiffusion / 'hybrid') driven as wukong-huahua/inpaint.py:65-106 drives it with its CLI defaults -- batch 4,
    PLMS with S = 30 (a 31-point grid: 32 evaluations at UNet batch 8), scale 7.5, dict conditioning with the SAME c_concat on both CFG halves.
    Oracle: committed fixture tests/golden/traj_inpaint_wukong_plms30.npz (one apply_model row + the trajectory of image 0)."""
    import json as _json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_trajectory_goldens import inputs_inpaint
    from minddiffusion_amd.configs import WUKONG_INPAINT_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentInpaintDiffusion
    from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
    z = np.load(os.path.join(ROOT, "tests", "golden", "traj_inpaint_wukong_plms30.npz"))
    meta = _json.loads(str(z["meta"]))
    inp = inputs_inpaint()
    # (S = 30 on the reference's uniform grid is range(0, 1000, 1000 // 30): 31 timesteps, so PLMS makes 32 model calls)
    assert meta["S"] == inp["S"] and meta["unet_seed"] == inp["seed"] and meta["unet_calls"] == 32
    ocfg = dict(O.WUKONG_UNET, in_channels=9)
    net, _ = _unet(WUKONG_INPAINT_UNET, ocfg, inp["seed"])
    model = LatentInpaintDiffusion(unet_config=net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    dev = lambda a: torch.tensor(a, device=DEV)
    B = 4
    # one call through the reference's keyword surface (WK ddpm.py:276-278), all four images; row 0 against the oracle row
    e = model.apply_model(dev(inp["x_T"]), torch.full((B,), 500.0, device=DEV), c_concat=[dev(inp["c_cat"])],
                          c_crossattn=[dev(inp["c"])])
    check("inpaint_wukong_full_apply_model_row0", e[:1], torch.tensor(z["apply_model_t500"]), rel_l2=5e-3, max_abs=5e-2)
    got, _ = PLMSSampler(model).sample(inp["S"], B, (4, 64, 64),
                                       conditioning={"c_concat": dev(inp["c_cat"]), "c_crossattn": dev(inp["c"])},
                                       x_T=dev(inp["x_T"]), unconditional_guidance_scale=inp["scale"],
                                       unconditional_conditioning={"c_concat": dev(inp["c_cat"]), "c_crossattn": dev(inp["uc"])},
                                       x0=dev(inp["c_cat"][:, 1:]), verbose=False)     # inpaint.py:104 passes x0 and no mask
    assert net._plans[(2 * B, 64, 64)].graph is not None, "the sampler replays a hipGraph of the 9-channel UNet at batch 8"
    check("inpaint_wukong_full_plms30_B4_image0", got[:1], torch.tensor(z["final"].astype(np.float32)), rel_l2=5e-3, max_rel=2e-2)


# --------------------------------------------------------------------------------------------- config 3: SDv2 768, 4 images / GPU
def test_config3_sd2_768_unet_batch8_latent96():
    """BASELINE configs[3] per-GPU share: SDv2 UNet on a 96x96 latent at UNet batch 8 (4 images x CFG): M = 73 728-row
    launches, N = 9216-token self-attention, the 24x24 / 12x12 convs on the generic kernel."""
    from minddiffusion_amd.configs import SD2_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddim import DDIMSampler
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    net, oracle = _ldm_full_batch_case("config3_sd2_768", SD2_UNET, O.SD2_UNET, 8, 96, 1024, 661.0, oracle_rows=[5],
                                       consistency_rows=[0], seed=1, min_hits=4)
    # four DDIM steps at the benchmarked batch (4 images x CFG 7.5 on the 96 x 96 latent), the oracle follows image 0
    model = LatentDiffusion(net, linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    S, scale, Bi = 4, 7.5, 4
    rng = np.random.RandomState(17)
    x_T = rng.randn(Bi, 4, 96, 96).astype(np.float32)
    c = rng.randn(Bi, 77, 1024).astype(np.float32)
    uc = np.repeat(rng.randn(1, 77, 1024).astype(np.float32), Bi, 0)
    got, _ = DDIMSampler(model).sample(S, Bi, (4, 96, 96), conditioning=torch.tensor(c, device=DEV),
                                       x_T=torch.tensor(x_T, device=DEV), unconditional_guidance_scale=scale,
                                       unconditional_conditioning=torch.tensor(uc, device=DEV), verbose=False)
    ref, _ = _short_fixture("config3_ddim4", S, lambda: (O.sample(O.ModelOracle(oracle), S, 1, (4, 96, 96), c[:1], x_T[:1], "ddim",
                                                                  unconditional_guidance_scale=scale,
                                                                  unconditional_conditioning=uc[:1])[0], None))
    check("config3_sd2_768_ddim4_B4_image0", got[:1], ref, rel_l2=1e-2, max_rel=2e-2)


# --------------------------------------------------------------------------------------------- config 4: Taichu-GLIDE, 8 images / GPU
def test_config4_glide_base_batch16_and_superres_batch8_full_size():
    """BASELINE configs[4] per-GPU share: the base model at 2P = 16 rows (64x64, 16-layer text transformer in the step) and
    the FULL-SIZE super-resolution UNet (6 levels, 192..768 channels) at P = 8 on 256x256: the only place the HALO conv sees
    W = 256 and the VAE-scale GroupNorm slabs inside a UNet."""
    from minddiffusion_amd.glide.default_options import model_and_diffusion_defaults, model_and_diffusion_upsample
    from minddiffusion_amd.glide.diffusion_creator import create_model, create_upsample_model
    _threads()
    rng = np.random.RandomState(5)
    dev = lambda a: torch.tensor(a, device=DEV)
    # ---- base, 2P = 16: rows 0..7 prompts, rows 8..15 the (shared) random unconditional prompt
    P = 8
    bp = OG.init_params(OG.BASE_OPTIONS, seed=0)
    net = create_model(**model_and_diffusion_defaults())
    net.load_state_dict(bp)
    xs = rng.randn(P, 3, 64, 64).astype(np.float32)
    x = np.concatenate([xs, xs], 0)
    tok = rng.randint(1, 50000, (2 * P, 128)).astype(np.int32)
    tok[P:] = tok[P]
    mask = np.ones((2 * P, 128), np.int32)
    mask[2, 40:] = 0
    t = np.full((2 * P,), 982.0, np.float32)
    got = net(dev(x), dev(t), dev(tok), dev(mask)).cpu()
    assert net._plans[(2 * P, 64, 64)].graph is not None
    _assert_tuned_rows_hit(net._plans[(2 * P, 64, 64)], "config4_glide_base_b16", 3)
    rows = [2, P + 2]                                      # a padded prompt row and its unconditional partner
    oracle = OG.GlideUNetOracle(OG.BASE_OPTIONS, bp)
    ref = oracle(x[rows], torch.tensor(t[rows]), tok[rows], mask[rows])
    check("config4_glide_base_B16_rows_vs_oracle", got[rows], ref, rel_l2=5e-3, max_abs=5e-2)
    two = net(dev(x[rows]), dev(t[rows]), dev(tok[rows]), dev(mask[rows])).cpu()
    check("config4_glide_base_B16_rows_vs_B2_hip", got[rows], two, rel_l2=4e-3, max_abs=2e-2)
    del net, oracle, bp
    torch.cuda.empty_cache()
    # ---- super-resolution UNet, P = 8 at 256x256 (1.28 TFLOP per row on the oracle: one row)
    up = OG.init_params(OG.UPSAMPLE_OPTIONS, seed=1)
    sr = create_upsample_model(**model_and_diffusion_upsample())
    sr.load_state_dict(up)
    xu = rng.randn(P, 3, 256, 256).astype(np.float32)
    low = np.clip(rng.randn(P, 3, 64, 64) * 0.5, -1, 1).astype(np.float32)
    tu = np.full((P,), 500.0, np.float32)
    gotu = sr(dev(xu), dev(tu), dev(low), dev(tok[:P]), dev(mask[:P])).cpu()
    assert sr._plans[(P, 256, 256)].graph is not None
    _assert_tuned_rows_hit(sr._plans[(P, 256, 256)], "config4_glide_superres_b8", 3)
    oracle = OG.GlideUNetOracle(OG.UPSAMPLE_OPTIONS, up)
    lowq = torch.round((torch.tensor(low[2:3]) + 1) * 127.5) / 127.5 - 1
    refu = oracle(xu[2:3], torch.tensor(tu[2:3]), tok[2:3], mask[2:3], low_res=lowq)
    check("config4_glide_superres_B8_row2_vs_oracle", gotu[2:3], refu, rel_l2=5e-3, max_abs=5e-2)
    oneu = sr(dev(xu[2:3]), dev(tu[2:3]), dev(low[2:3]), dev(tok[2:3]), dev(mask[2:3])).cpu()
    check("config4_glide_superres_B8_row2_vs_B1_hip", gotu[2:3], oneu, rel_l2=4e-3, max_abs=2e-2)


def test_glide_base_at_a_32_pixel_image_plans_and_matches_its_batch1_rows():
    """Round-5 regression (found by tools/shape_sweep.py --model glide): at a 32 x 32 image and batch 2 the full-size base UNet has a
    conv whose column-statistics launch variant resolves to a tile-table row that splits K five ways (reduce kernel, 64-row statistics
    blocks).  The GroupNorm wiring asked for the row-block count while the shared workspace still had its first size -- too small for
    that split, so the query reported the un-split 128-row form -- and the launch then refused the statistics buffer
    ('colstats_out holds 16 row blocks, this launch writes 32').  The wiring now asks with an ample workspace (ops.py).  Checked:
    the plan builds, runs, and every row equals its own batch-1 evaluation to fp16 rounding."""
    from minddiffusion_amd.glide.default_options import model_and_diffusion_defaults
    from minddiffusion_amd.glide.diffusion_creator import create_model
    from minddiffusion_amd.weights import synthetic_unet_params_device
    net = create_model(**model_and_diffusion_defaults())
    net.load_state_dict(synthetic_unet_params_device(net.parameter_shapes(), seed=0, device=DEV))
    net.use_graph = False
    rng = np.random.RandomState(32)
    B = 3
    x = torch.tensor(rng.randn(B, 3, 32, 32).astype(np.float32), device=DEV)
    tok = torch.tensor(rng.randint(1, 50000, (B, 128)).astype(np.int32), device=DEV)
    msk = torch.ones((B, 128), dtype=torch.int32, device=DEV)
    t = torch.full((B,), 500.0, device=DEV)
    ones = [net.forward_nhwc(x[r:r + 1].clone(), t[:1], tok[r:r + 1], msk[r:r + 1]).clone() for r in range(B)]
    for b in (2, 3):
        full = net.forward_nhwc(x[:b].clone(), t[:b], tok[:b], msk[:b]).clone()
        for r in range(b):
            check(f"glide_base_32px_B{b}_row{r}_vs_B1", full[r:r + 1], ones[r], rel_l2=4e-3, max_abs=5e-2)


def test_config4_glide_full_size_loops():
    """BASELINE configs[4], the LOOPS on the full-size models (Taichu-GLIDE/model/glide_text2im/main_funcs.py:21-69): ten
    guided ancestral steps of the 385 M-parameter base model (learned variance, x0 clipping, per-step random unconditional
    prompt: 20 oracle rows) and three DDIM steps of the full-size up-sampler on 256 x 256 (3 oracle rows of 1.28 TFLOP), with
    the same tokens / noises injected on both sides.  (Fewer than ~10 ancestral steps make the respaced cosine schedule
    degenerate in fp32: beta' rounds to 1.)"""
    from minddiffusion_amd.glide.default_options import model_and_diffusion_defaults, model_and_diffusion_upsample
    from minddiffusion_amd.glide.diffusion_creator import init_diffusion_model, init_super_res_model
    from minddiffusion_amd.glide.main_funcs import ddim_sample_loop, gaussian_p_sample_loop
    _threads()
    rng = np.random.RandomState(23)
    P, steps = 1, 10
    bp = OG.init_params(OG.BASE_OPTIONS, seed=0)
    opts = dict(model_and_diffusion_defaults(), timestep_respacing=str(steps))
    dm = init_diffusion_model(options=opts, guidance_scale=5.0, shape=(2 * P, 3, 64, 64), params=bp)
    assert dm.num_timesteps == steps
    oracle = OG.GlideUNetOracle(OG.BASE_OPTIONS, bp)
    sch = OG.respaced_schedule("squaredcos_cap_v2", 1000, str(steps))
    x_T = rng.randn(P, 3, 64, 64).astype(np.float32)
    tok = rng.randint(1, 50000, (P, 128)).astype(np.int32)
    mask = np.ones((P, 128), np.int32)
    mask[0, 50:] = 0
    unc = rng.randint(1, 50000, (steps, 128)).astype(np.int32)
    noises = rng.randn(steps, P, 3, 64, 64).astype(np.float32)
    traj = []
    ref = OG.p_sample_loop(oracle, sch, x_T, tok, mask, 5.0, unc, noises, trajectory=traj)
    tok2, mask2 = np.concatenate([tok, tok], 0), np.concatenate([mask, mask], 0)
    # (1) every step on its own, from the ORACLE's x_t (teacher-forced: what one guided evaluation + PSample gets wrong, not
    # what ten of them compound to).  The first respaced step has alpha-bar ~ 2e-9: x0 is clipped to +-1 everywhere and a
    # 1e-2 difference in eps flips a few signs, so that step is bounded in energy and in its bulk (tests/test_glide_gpu.py).
    ones = np.ones((128,), np.int32)
    for k, i in enumerate(range(steps - 1, -1, -1)):
        xk = torch.cat([traj[k], traj[k]], 0)
        got_k, _ = dm(x=xk.to(DEV), timesteps=torch.tensor([i], dtype=torch.int32), token=torch.tensor(tok2),
                      mask=torch.tensor(mask2), random_token=unc[k], random_mask=ones,
                      noise=torch.tensor(noises[k], device=DEV))
        if k == 0:      # measured 1.3e-2 (a handful of flipped signs, |d| = 0.34), every later step 1.0-2.2e-3
            check(f"config4_glide_full_base_p_sample_step{i}", got_k[:P], traj[k + 1], rel_l2=2e-2, abs_q=(0.99, 2e-2))
        else:
            check(f"config4_glide_full_base_p_sample_step{i}", got_k[:P], traj[k + 1], rel_l2=5e-3, max_rel=5e-2)
    # (2) the free-running loop.  Ten coarse ancestral steps at guidance 5 with x0 clipping on RANDOM weights amplify the
    # per-evaluation fp16 distance (1.9e-3, test above) about tenfold -- measured 1.9e-2 on the first run of this test (round 3);
    # the trained model's loop contracts towards an image, this one does not.  The bound is the energy of the difference.
    got = gaussian_p_sample_loop(dm, torch.tensor(tok2), torch.tensor(mask2), (2 * P, 3, 64, 64), steps, text_ctx=128,
                                 noise=torch.tensor(np.concatenate([x_T, x_T], 0)), vocab_len=50001, uncond_tokens=list(unc),
                                 step_noises=[torch.tensor(n, device=DEV) for n in noises])[:P]
    check("config4_glide_full_base_p_sample_loop10", got, ref, rel_l2=3e-2)
    # What the reference's OWN fp16 mode does on these two loops (tests/golden/glide_threeway.json, written by
    # make_trajectory_goldens.py with `oracle.emulate_fp16`): on this 10-step loop its end point is NOT FINITE (sqrt_recip ~ 2e4 at
    # the first respaced step overflows fp16 and inf - inf follows), on the 3-step up-sampler loop below it ends 7.3e-2 (max 2.0) from
    # the fp32 oracle -- the GPU path (fp32 sampler arithmetic, fp16 storage) measures 1.9e-2 / 8.6e-3: inside that envelope.
    import json as _json
    tw = _json.load(open(os.path.join(ROOT, "tests", "golden", "glide_threeway.json")))
    print("PARITY", _json.dumps(dict(name="config4_glide_full_loops_fp16emu_reference", **tw)))
    # The fp16-emulated oracle does not survive this loop (the fixture records a non-finite end point), so there is no finite
    # envelope to tie the base loop to: its bound is the rel-L2 above alone.  The fixture is asserted to still say so -- if a
    # regenerated one ever reports a finite fp16 loop, this test must get the envelope assertion the up-sampler loop has below.
    assert tw["base_loop10_fp16emu_finite"] is False
    del dm, oracle, bp
    torch.cuda.empty_cache()
    up = OG.init_params(OG.UPSAMPLE_OPTIONS, seed=1)
    uopts = dict(model_and_diffusion_upsample(), timestep_respacing="3")
    sr = init_super_res_model(options=uopts, shape=(P, 3, 256, 256), params=up)
    assert sr.num_timesteps == 3
    oracle = OG.GlideUNetOracle(OG.UPSAMPLE_OPTIONS, up)
    schu = OG.respaced_schedule("linear", 1000, "3")
    xs = rng.randn(P, 3, 256, 256).astype(np.float32) * 0.997
    low = np.clip(rng.randn(P, 3, 64, 64) * 0.5, -1, 1).astype(np.float32)
    refu = OG.ddim_sample_loop(oracle, schu, xs, low, tok, mask)
    gotu = ddim_sample_loop(sr, (P, 3, 256, 256), torch.tensor(low, device=DEV), torch.tensor(tok), torch.tensor(mask), 3,
                            noise=torch.tensor(xs))
    # three coarse DDIM steps on random weights end SATURATED at the clip (|x| = 1 almost everywhere): an element whose x0 sits at
    # the edge differs by up to 0.5 (measured: rel-L2 8.6e-3, max 0.51) -- bound the energy and the bulk
    m = check("config4_glide_full_superres_ddim_loop3", gotu, refu, rel_l2=1e-2, abs_q=(0.99, 2e-2))
    assert m["rel_l2"] <= 2 * tw["up_loop3_d_oracle32_vs_fp16emu"]


def test_first_use_tuner_at_a_resolution_the_tile_table_does_not_list():
    """include/mdx.h mdx_gemm_tune + the planner's unet_tune_first_use option (no reference counterpart: the reference's graph
    compiler picks kernels per shape).  A 768 x 512 image (96 x 64 latent, UNet batch 2) is not one of the benchmarked shapes, so
    most of its launches resolve through the cost model; with the option on the plan measures every such shape once (user-side
    cache ops.tune_cache) and must then be at least as fast per evaluation as the cost-model plan (5 % slack for timer and box
    noise), with the same result to the distance of two fp16 paths that tile / split differently."""
    from minddiffusion_amd import ops
    from minddiffusion_amd.configs import SD2_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    params = O.init_params(O.SD2_UNET, seed=6)
    rng = np.random.RandomState(8)
    x = torch.tensor(rng.randn(2, 4, 64, 96).astype(np.float32), device=DEV)
    ctx = torch.tensor(rng.randn(2, 77, 1024).astype(np.float32), device=DEV)
    t = torch.full((2,), 301.0, device=DEV)

    def timed(net):
        out = net.forward_nhwc(x, t, ctx).clone()
        P = net._plans[(2, 64, 96)]
        assert P.graph is not None
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                P.graph.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        return out, P, best
    net0 = UNetModel(**dict(SD2_UNET)).load_state_dict(params)
    out0, P0, ms0 = timed(net0)
    untuned = sum(1 for d in P0.descs if not ops.gemm_query(d)[4])
    assert untuned >= 40, f"expected most launches of this shape to miss the tile table, {untuned} of {len(P0.descs)} do"
    del net0, P0
    ops.tune_cache.clear()
    ops.set_option("unet_tune_first_use", 1)
    try:
        net1 = UNetModel(**dict(SD2_UNET)).load_state_dict(params)
        out1, P1, ms1 = timed(net1)
    finally:
        ops.set_option("unet_tune_first_use", 0)
    changed = sum(1 for v in ops.tune_cache.values() if any(v[:4]))
    print("PARITY", {"name": "first_use_tuner_96x64_latent", "shapes_measured": P1.tuned_shapes, "shapes_changed": changed,
                     "ms_per_eval_cost_model": round(ms0, 4), "ms_per_eval_tuned": round(ms1, 4)})
    assert P1.tuned_shapes >= 10 and len(ops.tune_cache) == P1.tuned_shapes
    check("first_use_tuner_96x64_latent_tuned_vs_cost_model", out1, out0, rel_l2=4e-3)
    assert ms1 <= ms0 * 1.05, f"tuned plan slower than the cost-model plan: {ms1:.3f} vs {ms0:.3f} ms"
    # the cache is the caller's: a second network of the same shape measures nothing
    ops.set_option("unet_tune_first_use", 1)
    try:
        net2 = UNetModel(**dict(SD2_UNET)).load_state_dict(params)
        net2.forward_nhwc(x, t, ctx)
        assert net2._plans[(2, 64, 96)].tuned_shapes == 0
    finally:
        ops.set_option("unet_tune_first_use", 0)
        ops.tune_cache.clear()
