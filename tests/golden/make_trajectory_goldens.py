"""Benchmarked-LENGTH trajectories of the CPU oracle, committed as fixtures so that the GPU suite can check the full-size models
over the full sampler length without spending oracle time on the GPU box (one oracle evaluation costs 4-20 s of host time):

    traj_config1_sd2_512_ddim50.npz     SDv2 UNet, 64x64 latent, DDIM-50, CFG 9.0, batch 1        (BASELINE configs[1], 100 rows)
    traj_config2_wukong_plms50.npz      Wukong UNet, 64x64, PLMS-50 (51 evaluations), CFG 7.5, image 0 of the batch of 8 (102 rows)
    traj_config3_sd2_768_ddim50.npz     SDv2 UNet, 96x96 latent, DDIM-50, CFG 7.5, image 0 of the 4 per GPU  (100 rows)
    traj_config4_glide_60_27.npz        Taichu-GLIDE base 60 guided ancestral steps (120 rows) + up-sampler 27 DDIM steps at 256x256
    traj_inpaint_wukong_plms30.npz      Wukong inpainting UNet (9 input channels), 64x64, PLMS-30 (31 evaluations), CFG 7.5, hybrid
                                        conditioning, image 0 of the CLI's batch of 4 (62 rows) + one apply_model row
    glide_threeway.json                 d(fp32 oracle, fp16-emulated oracle) on the full-size GLIDE 10-step / 3-step loops of
                                        tests/test_configs_gpu.py::test_config4_glide_full_size_loops (sets their bounds)

Every file holds the oracle's outputs as fp16 (final latent, last pred_x0, a few intermediate latents), the fp16-EMULATED oracle's
final latent where it was run (`*_emu16`: the distance a reference running `use_fp16: True` would itself sit from the fp32 oracle),
the seeds / shapes needed to rebuild the inputs, and the commit the oracle was at.  Inputs are NOT stored: the tests rebuild them from
the seeds exactly as this script does (`inputs_*` below are imported by the tests).

Run in the build container (CPU only; ~1-2 h on 8 cores):   python tests/golden/make_trajectory_goldens.py [case ...]
The oracle is test infrastructure (oracle/__init__.py); nothing under minddiffusion_amd/ is imported here.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        return "unknown"


# ------------------------------------------------------------------------------------------------ inputs (shared with the tests)
def inputs_config1():
    """= tests/test_configs_gpu.py::test_config1_* (UNet seed 4)."""
    return dict(seed=4, S=50, scale=9.0, sampler="ddim", hw=64, ctx_dim=1024,
                x_T=np.random.RandomState(42).randn(1, 4, 64, 64).astype(np.float32),
                c=np.random.RandomState(1).randn(1, 77, 1024).astype(np.float32),
                uc=np.random.RandomState(2).randn(1, 77, 1024).astype(np.float32))


def inputs_config2():
    """= test_config2_* (Wukong UNet seed 2, batch of 8 images, the oracle follows image 0)."""
    rng = np.random.RandomState(7)
    x_T = rng.randn(8, 4, 64, 64).astype(np.float32)
    c = rng.randn(8, 77, 768).astype(np.float32)
    uc = np.repeat(rng.randn(1, 77, 768).astype(np.float32), 8, 0)
    return dict(seed=2, S=50, scale=7.5, sampler="plms", hw=64, ctx_dim=768, x_T=x_T, c=c, uc=uc)


def inputs_config3():
    """= test_config3_* (SDv2 UNet seed 1, 96 x 96 latent, 4 images, the oracle follows image 0)."""
    rng = np.random.RandomState(17)
    x_T = rng.randn(4, 4, 96, 96).astype(np.float32)
    c = rng.randn(4, 77, 1024).astype(np.float32)
    uc = np.repeat(rng.randn(1, 77, 1024).astype(np.float32), 4, 0)
    return dict(seed=1, S=50, scale=7.5, sampler="ddim", hw=96, ctx_dim=1024, x_T=x_T, c=c, uc=uc)


def inputs_config4(base_steps=60, up_steps=27):
    """Taichu-GLIDE at the lengths of src/txt2img.py:141-144 / main_funcs.py:21-69: one prompt (P = 1)."""
    rng = np.random.RandomState(29)
    P = 1
    tok = rng.randint(1, 50000, (P, 128)).astype(np.int32)
    mask = np.ones((P, 128), np.int32)
    mask[0, 37:] = 0
    return dict(P=P, guidance=5.0, base_steps=base_steps, up_steps=up_steps, tok=tok, mask=mask,
                x_T=rng.randn(P, 3, 64, 64).astype(np.float32),
                unc=rng.randint(1, 50000, (base_steps, 128)).astype(np.int32),
                noises=rng.randn(base_steps, P, 3, 64, 64).astype(np.float32),
                up_x_T=(rng.randn(P, 3, 256, 256) * 0.997).astype(np.float32))


def inputs_inpaint():
    """= tests/test_configs_gpu.py::test_inpaint_wukong_full_size (Wukong 9-channel inpainting UNet seed 5, the CLI's batch of 4
    images at its 30 PLMS steps and scale 7.5, wukong-huahua/inpaint.py:65-106 + its argparse defaults; the oracle follows image 0).
    c_concat = cat(resized mask, masked-image latent) as inpaint.py:84-92 builds it."""
    rng = np.random.RandomState(31)
    x_T = rng.randn(4, 4, 64, 64).astype(np.float32)
    c = rng.randn(4, 77, 768).astype(np.float32)
    uc = np.repeat(rng.randn(1, 77, 768).astype(np.float32), 4, 0)
    m = (rng.rand(4, 1, 64, 64) > 0.5).astype(np.float32)
    masked_latent = rng.randn(4, 4, 64, 64).astype(np.float32)
    return dict(seed=5, S=30, scale=7.5, sampler="plms", hw=64, ctx_dim=768, x_T=x_T, c=c, uc=uc,
                c_cat=np.concatenate([m, masked_latent], 1))


def inputs_glide_loops_test():
    """The draws of test_config4_glide_full_size_loops, in its order."""
    rng = np.random.RandomState(23)
    P, steps = 1, 10
    d = dict(P=P, steps=steps)
    d["x_T"] = rng.randn(P, 3, 64, 64).astype(np.float32)
    d["tok"] = rng.randint(1, 50000, (P, 128)).astype(np.int32)
    mask = np.ones((P, 128), np.int32)
    mask[0, 50:] = 0
    d["mask"] = mask
    d["unc"] = rng.randint(1, 50000, (steps, 128)).astype(np.int32)
    d["noises"] = rng.randn(steps, P, 3, 64, 64).astype(np.float32)
    d["xs"] = rng.randn(P, 3, 256, 256).astype(np.float32) * 0.997
    d["low"] = np.clip(rng.randn(P, 3, 64, 64) * 0.5, -1, 1).astype(np.float32)
    return d


# ------------------------------------------------------------------------------------------------ cases
def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _ldm_case(name, ocfg_name, inp, with_emu=True):
    from oracle import ldm as O
    ocfg = getattr(O, ocfg_name)
    t0 = time.time()
    params = O.init_params(ocfg, seed=inp["seed"])
    model = O.ModelOracle(O.UNetOracle(ocfg, params))
    hw = inp["hw"]
    kw = dict(unconditional_guidance_scale=inp["scale"], unconditional_conditioning=inp["uc"][:1])
    model.calls = 0
    ref, inter = O.sample(model, inp["S"], 1, (4, hw, hw), inp["c"][:1], inp["x_T"][:1], inp["sampler"], log_every_t=10, **kw)
    calls = model.calls
    out = dict(final=ref.numpy().astype(np.float16), pred_x0=inter["pred_x0"][-1].numpy().astype(np.float16),
               x_inter=np.stack([x.numpy() for x in (inter["x_inter"][0], inter["x_inter"][len(inter["x_inter"]) // 2],
                                                     inter["x_inter"][-1])]).astype(np.float16),
               final_f32_norm=np.float64(ref.double().norm()))
    meta = dict(name=name, oracle_cfg=ocfg_name, unet_seed=inp["seed"], S=inp["S"], sampler=inp["sampler"], scale=inp["scale"],
                latent=hw, unet_calls=calls, commit=_commit(), oracle_seconds=round(time.time() - t0, 1))
    print(name, "fp32 trajectory done in", meta["oracle_seconds"], "s,", calls, "model calls", flush=True)
    if with_emu:
        t1 = time.time()
        with O.emulate_fp16():
            emu, _ = O.sample(model, inp["S"], 1, (4, hw, hw), inp["c"][:1], inp["x_T"][:1], inp["sampler"], **kw)
        out["final_emu16"] = emu.numpy().astype(np.float16)
        meta["d_oracle32_vs_fp16emu"] = _rel(emu, ref)
        meta["emu_seconds"] = round(time.time() - t1, 1)
        print(name, "fp16-emulated trajectory:", meta["d_oracle32_vs_fp16emu"], flush=True)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, f"traj_{name}.npz"), **out)
    print("wrote", f"traj_{name}.npz", meta, flush=True)


def case_config1():
    _ldm_case("config1_sd2_512_ddim50", "SD2_UNET", inputs_config1())


def case_config2():
    _ldm_case("config2_wukong_plms50", "WUKONG_UNET", inputs_config2())


def case_config3():
    _ldm_case("config3_sd2_768_ddim50", "SD2_UNET", inputs_config3())


def case_config4():
    from oracle import glide as OG
    inp = inputs_config4()
    t0 = time.time()
    bp = OG.init_params(OG.BASE_OPTIONS, seed=0)
    net = OG.GlideUNetOracle(OG.BASE_OPTIONS, bp)
    sch = OG.respaced_schedule("squaredcos_cap_v2", 1000, str(inp["base_steps"]))
    traj = []
    base = OG.p_sample_loop(net, sch, inp["x_T"], inp["tok"], inp["mask"], inp["guidance"], inp["unc"], inp["noises"],
                            trajectory=traj)
    print("glide base loop done", round(time.time() - t0, 1), "s", flush=True)
    out = dict(base_final=base.numpy().astype(np.float16),
               base_traj=np.stack([t.numpy() for t in traj[::10]]).astype(np.float16))
    del net, bp
    up = OG.init_params(OG.UPSAMPLE_OPTIONS, seed=1)
    sr = OG.GlideUNetOracle(OG.UPSAMPLE_OPTIONS, up)
    schu = OG.respaced_schedule("linear", 1000, "fast27")
    # the up-sampler starts from the fp16-rounded base result on both sides (the fixture carries it): stage 2 is teacher-forced
    low = torch.tensor(out["base_final"].astype(np.float32))
    t1 = time.time()
    fin = OG.ddim_sample_loop(sr, schu, inp["up_x_T"], low, inp["tok"], inp["mask"])
    print("glide up-sampler loop done", round(time.time() - t1, 1), "s", flush=True)
    out["up_final"] = fin.numpy().astype(np.float16)
    meta = dict(name="config4_glide_60_27", base_steps=inp["base_steps"], up_steps=len(schu["betas"]), guidance=inp["guidance"],
                base_seed=0, up_seed=1, input_seed=29, commit=_commit(), oracle_seconds=round(time.time() - t0, 1))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "traj_config4_glide_60_27.npz"), **out)
    print("wrote traj_config4_glide_60_27.npz", meta, flush=True)


def case_glide_threeway():
    """d(O32, E16) for the two loops of test_config4_glide_full_size_loops (same inputs): what the reference's own fp16 mode
    does to these trajectories.  The test bounds the GPU's distance by twice this."""
    from oracle import glide as OG
    from oracle import ldm as O
    d = inputs_glide_loops_test()
    res = dict(commit=_commit())
    bp = OG.init_params(OG.BASE_OPTIONS, seed=0)
    net = OG.GlideUNetOracle(OG.BASE_OPTIONS, bp)
    sch = OG.respaced_schedule("squaredcos_cap_v2", 1000, str(d["steps"]))
    o32 = OG.p_sample_loop(net, sch, d["x_T"], d["tok"], d["mask"], 5.0, d["unc"], d["noises"])
    with O.emulate_fp16():
        o16 = OG.p_sample_loop(net, sch, d["x_T"], d["tok"], d["mask"], 5.0, d["unc"], d["noises"])
    res["base_loop10_d_oracle32_vs_fp16emu"] = _rel(o16, o32)
    res["base_loop10_fp16emu_finite"] = bool(torch.isfinite(o16).all())
    res["base_loop10_max_abs"] = float((o16 - o32).abs().max())
    res["base_loop10_q99_abs"] = float(torch.quantile((o16 - o32).abs().flatten(), 0.99))
    print(res, flush=True)
    del net, bp
    up = OG.init_params(OG.UPSAMPLE_OPTIONS, seed=1)
    sr = OG.GlideUNetOracle(OG.UPSAMPLE_OPTIONS, up)
    schu = OG.respaced_schedule("linear", 1000, "3")
    u32 = OG.ddim_sample_loop(sr, schu, d["xs"], d["low"], d["tok"], d["mask"])
    with O.emulate_fp16():
        u16 = OG.ddim_sample_loop(sr, schu, d["xs"], d["low"], d["tok"], d["mask"])
    res["up_loop3_d_oracle32_vs_fp16emu"] = _rel(u16, u32)
    res["up_loop3_fp16emu_finite"] = bool(torch.isfinite(u16).all())
    res["up_loop3_max_abs"] = float((u16 - u32).abs().max())
    res["up_loop3_q99_abs"] = float(torch.quantile((u16 - u32).abs().flatten()[:4000000], 0.99))
    with open(os.path.join(HERE, "glide_threeway.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("wrote glide_threeway.json", res, flush=True)


def case_short():
    """The SHORT full-size trajectories of tests/test_configs_gpu.py (config 1 DDIM-10, config 2 PLMS-5, config 3 DDIM-4 -- same UNets
    and inputs as the benchmarked-length cases above, fewer steps), kept in fp32: 40 oracle row evaluations the GPU box then does
    not spend (240 s of its 1200 s test budget in round 3)."""
    from oracle import ldm as O
    out, meta = {}, dict(commit=_commit(), cases={})
    for key, ocfg_name, inp, S in (("config1_ddim10", "SD2_UNET", inputs_config1(), 10), ("config2_plms5", "WUKONG_UNET", inputs_config2(), 5),
                                   ("config3_ddim4", "SD2_UNET", inputs_config3(), 4)):
        t0 = time.time()
        ocfg = getattr(O, ocfg_name)
        model = O.ModelOracle(O.UNetOracle(ocfg, O.init_params(ocfg, seed=inp["seed"])))
        hw = inp["hw"]
        ref, inter = O.sample(model, S, 1, (4, hw, hw), inp["c"][:1], inp["x_T"][:1], inp["sampler"],
                              unconditional_guidance_scale=inp["scale"], unconditional_conditioning=inp["uc"][:1])
        out[key + "_final"] = ref.numpy().astype(np.float32)
        out[key + "_pred_x0"] = inter["pred_x0"][-1].numpy().astype(np.float32)
        meta["cases"][key] = dict(oracle_cfg=ocfg_name, unet_seed=inp["seed"], S=S, sampler=inp["sampler"], scale=inp["scale"], latent=hw,
                                  unet_calls=model.calls, oracle_seconds=round(time.time() - t0, 1))
        print(key, meta["cases"][key], flush=True)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "short_traj.npz"), **out)
    print("wrote short_traj.npz", flush=True)


def case_inpaint():
    """Full-size inpainting model (configs/wukong-huahua_inpaint_inference.yaml) at the CLI's length: one apply_model row at
    t = 500 (fp32) and the PLMS-30 trajectory (31 evaluations, CFG 7.5, hybrid dict conditioning) of image 0."""
    from oracle import ldm as O
    inp = inputs_inpaint()
    t0 = time.time()
    ocfg = dict(O.WUKONG_UNET, in_channels=9)
    model = O.ModelOracle(O.UNetOracle(ocfg, O.init_params(ocfg, seed=inp["seed"])), conditioning_key="hybrid")
    one = model.apply_model(torch.tensor(inp["x_T"][:1]), torch.full((1,), 500.0),
                            {"c_concat": torch.tensor(inp["c_cat"][:1]), "c_crossattn": torch.tensor(inp["c"][:1])})
    out = dict(apply_model_t500=one.numpy().astype(np.float32))
    print("inpaint apply_model row done", round(time.time() - t0, 1), "s", flush=True)
    model.calls = 0
    ref, inter = O.sample(model, inp["S"], 1, (4, 64, 64), {"c_concat": inp["c_cat"][:1], "c_crossattn": inp["c"][:1]},
                          inp["x_T"][:1], "plms", unconditional_guidance_scale=inp["scale"],
                          unconditional_conditioning={"c_concat": inp["c_cat"][:1], "c_crossattn": inp["uc"][:1]})
    out["final"] = ref.numpy().astype(np.float16)
    out["pred_x0"] = inter["pred_x0"][-1].numpy().astype(np.float16)
    meta = dict(name="inpaint_wukong_plms30", oracle_cfg="WUKONG_UNET + in_channels=9", unet_seed=inp["seed"], S=inp["S"],
                sampler="plms", scale=inp["scale"], latent=64, unet_calls=model.calls, commit=_commit(),
                oracle_seconds=round(time.time() - t0, 1))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "traj_inpaint_wukong_plms30.npz"), **out)
    print("wrote traj_inpaint_wukong_plms30.npz", meta, flush=True)


CASES = dict(short=case_short, inpaint=case_inpaint, config1=case_config1, config2=case_config2, config3=case_config3, config4=case_config4,
             glide_threeway=case_glide_threeway)

if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count() or 8)))
    for n in (sys.argv[1:] or list(CASES)):
        CASES[n]()
