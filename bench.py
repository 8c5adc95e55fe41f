#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native minddiffusion hot path.

Metric (BASELINE.json): 512x512 txt2img latents/sec (50-step DDIM) at 1/2/4/8 MI355X; per-UNet-step ms.
Workload at every N: BASELINE.json configs[1] per GPU -- SDv2 UNet (865.9 M params, synthetic seeded
weights), 64x64 latent, 50-step DDIM with classifier-free guidance 9.0 (UNet batch 2), batch 1 per GPU,
fp16 storage / fp32 accumulate.  A "step" is ONE complete 50-step trajectory of the per-GPU batch through
DiffusionPipeline (text embeddings are synthetic; rank 0 owns them and they reach the other ranks by one
RCCL broadcast inside the timed region).  Weak scaling: per-GPU work is fixed.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for how roofline / cpu_baseline are obtained).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TFLOP_PER_UNET_ROW_64 = 0.804      # SDv2 UNet, one eval, one batch row, 64x64 latent
MFMA_PEAK_TFLOPS = 2500.0          # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)


# BASELINE.json configs -> per-GPU workloads.  `sd2_512` (configs[1]) is the headline / default.
CONFIGS = {
    "sd2_512": dict(family="ldm", unet="sd2", latent=64, sampler="ddim", steps=50, scale=9.0, batch=1, ctx_dim=1024,
                    tflop_per_row=0.804, unit="latents/s", metric="512x512 txt2img latents/sec (50-step DDIM)",
                    workload="SDv2 txt2img 512x512 (64x64 latent), 50-step DDIM, CFG 9.0, batch 1 per GPU "
                             "(BASELINE.json configs[1])"),
    "wukong_512_plms": dict(family="ldm", unet="wukong", latent=64, sampler="plms", steps=50, scale=7.5, batch=8,
                            ctx_dim=768, tflop_per_row=0.803, unit="latents/s",
                            metric="512x512 txt2img latents/sec (50-step PLMS, Wukong-Huahua)",
                            workload="Wukong-Huahua txt2img 512x512, PLMS 50 steps (51 UNet calls), CFG 7.5, batch 8 "
                                     "per GPU (BASELINE.json configs[2])"),
    "sd2_768": dict(family="ldm", unet="sd2", latent=96, sampler="ddim", steps=50, scale=7.5, batch=4, ctx_dim=1024,
                    tflop_per_row=2.149, unit="latents/s", metric="768x768 txt2img latents/sec (50-step DDIM)",
                    workload="SDv2 txt2img 768x768 (96x96 latent), 50-step DDIM, CFG 7.5, 4 images per GPU "
                             "(BASELINE.json configs[3]: batch 32 over 8 GPUs)"),
    # the same run followed by AutoencoderKL.decode (1.24 TFLOP per 512x512 image) -> images/s
    "sd2_512_images": dict(family="ldm", unet="sd2", latent=64, sampler="ddim", steps=50, scale=9.0, batch=1,
                           ctx_dim=1024, tflop_per_row=0.804, vae=True, vae_tflop=1.24, unit="images/s",
                           metric="512x512 txt2img images/sec (50-step DDIM + VAE decode)",
                           workload="SDv2 txt2img 512x512: 50-step DDIM, CFG 9.0, batch 1 per GPU, then "
                                    "AutoencoderKL.decode to a 512x512 image (configs[1] + SURVEY 8(f) item 1)"),
    # SURVEY 8(f) items 1 + 2: the whole txt2img.py body -- text encoder (synthetic token ids), DDIM-50, VAE decode
    "sd2_512_e2e": dict(family="ldm", unet="sd2", latent=64, sampler="ddim", steps=50, scale=9.0, batch=1, ctx_dim=1024,
                        tflop_per_row=0.804, vae=True, vae_tflop=1.24, text=True, unit="images/s",
                        metric="512x512 txt2img images/sec (text encoder + 50-step DDIM + VAE decode)",
                        workload="SDv2 txt2img 512x512 end to end: FrozenCLIPEmbedder_ZH on [prompt; empty prompt] token "
                                 "ids, 50-step DDIM with CFG 9.0, AutoencoderKL.decode; batch 1 per GPU (txt2img.py:242-268)"),
    # txt2img.py --dpm_solver (DPM-Solver++ 2M; S UNet evaluations at fractional timesteps)
    "sd2_512_dpm_solver": dict(family="ldm", unet="sd2", latent=64, sampler="dpm_solver", steps=50, scale=9.0, batch=1,
                               ctx_dim=1024, tflop_per_row=0.804, unit="latents/s",
                               metric="512x512 txt2img latents/sec (50-step DPM-Solver++ 2M)",
                               workload="SDv2 txt2img 512x512, DPM-Solver++ (multistep order 2, time_uniform) with the "
                                        "CLI's default 50 steps, CFG 9.0, batch 1 per GPU (configs[1] with --dpm_solver)"),
    # wukong-huahua/inpaint.py with its CLI defaults (batch 4, PLMS 30 steps, scale 7.5) on the 9-channel UNet
    "wukong_512_inpaint": dict(family="ldm", unet="wukong_inpaint", latent=64, sampler="plms", steps=30, scale=7.5, batch=4,
                               ctx_dim=768, tflop_per_row=0.803, inpaint=True, unit="latents/s",
                               metric="512x512 inpainting latents/sec (30-step PLMS, Wukong-Huahua inpaint)",
                               workload="Wukong-Huahua inpainting 512x512 (LatentInpaintDiffusion, hybrid conditioning: UNet input "
                                        "= latent + resized mask + masked-image latent = 9 channels), PLMS S = 30 (31-point grid, 32 UNet "
                                        "calls), CFG 7.5, batch 4 per GPU (inpaint.py CLI defaults; every rank its own images)"),
    "glide_256": dict(family="glide", batch=8, scale=5.0, tflop_per_image=63.2, unit="images/s",
                      metric="Taichu-GLIDE 256x256 images/sec (60-step guided base + 27-step DDIM super-res)",
                      workload="Taichu-GLIDE 64x64 base (60 ancestral steps, CFG 5, UNet batch 2P) + 256x256 super-res "
                               "(27 DDIM steps), 8 images per GPU (BASELINE.json configs[4]: batch 16 over 2 GPUs)"),
}


def build_glide(device):
    from minddiffusion_amd.glide.default_options import model_and_diffusion_defaults, model_and_diffusion_upsample
    from minddiffusion_amd.glide.diffusion_creator import init_diffusion_model, init_super_res_model
    from minddiffusion_amd.weights import synthetic_unet_params_device
    P = CONFIGS["glide_256"]["batch"]
    ob = dict(model_and_diffusion_defaults(), device=str(device))
    ou = dict(model_and_diffusion_upsample(), device=str(device))
    dm = init_diffusion_model(ob, CONFIGS["glide_256"]["scale"], (2 * P, 3, 64, 64))
    dm.model.load_state_dict(synthetic_unet_params_device(dm.model.parameter_shapes(), seed=0, device=device))
    sr = init_super_res_model(ou, (P, 3, 256, 256))
    sr.model.load_state_dict(synthetic_unet_params_device(sr.model.parameter_shapes(), seed=1, device=device))
    torch.cuda.synchronize()
    return dm, sr


def build_vae(device):
    from minddiffusion_amd.configs import SD_VAE_DDCONFIG
    from minddiffusion_amd.ldm.models.autoencoder import AutoencoderKL
    from minddiffusion_amd.weights import synthetic_unet_params_device
    vae = AutoencoderKL(ddconfig=SD_VAE_DDCONFIG, embed_dim=4, device=device)
    vae.load_state_dict(synthetic_unet_params_device(vae.parameter_shapes(), seed=3, device=device))
    torch.cuda.synchronize()
    return vae


def build_text_encoder(device):
    from minddiffusion_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder_ZH
    from minddiffusion_amd.weights import synthetic_unet_params_device
    rng = np.random.RandomState(11)
    # the BPE vocabulary is not shipped: a deterministic stand-in tokenizer (random ids, end-token padded) feeds the
    # real encoder; "" -> all end tokens, like the reference's empty prompt
    def tokenizer(texts):
        out = np.full((len(texts), 77), 49407, np.int64)
        for i, t in enumerate(texts):
            n = min(75, len(t.split()) * 2)
            out[i, 0] = 49406
            out[i, 1:1 + n] = rng.randint(0, 49406, n)
        return out
    enc = FrozenCLIPEmbedder_ZH(tokenizer=tokenizer, device=device)
    shapes = enc.parameter_shapes()
    params = synthetic_unet_params_device(shapes, seed=5, device=device)
    for k in params:   # tables: O(0.5) entries so that the first LayerNorm sees a realistic signal
        if k.endswith("embedding_table") or k.endswith("positional_embedding"):
            params[k] = torch.randn(shapes[k], device=device) * 0.5
    enc.load_state_dict(params)
    torch.cuda.synchronize()
    return enc


def build_model(device, cfg_name="sd2"):
    from minddiffusion_amd.configs import SD2_LDM, SD2_UNET, WUKONG_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_device
    from minddiffusion_amd.configs import WUKONG_INPAINT_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentInpaintDiffusion
    ucfg = dict({"sd2": SD2_UNET, "wukong": WUKONG_UNET, "wukong_inpaint": WUKONG_INPAINT_UNET}[cfg_name])
    net = UNetModel(device=device, **ucfg)
    net.load_state_dict(synthetic_unet_params_device(net.parameter_shapes(), seed=0, device=device))
    torch.cuda.synchronize()
    cls = LatentInpaintDiffusion if cfg_name == "wukong_inpaint" else LatentDiffusion
    model = cls(unet_config=net, **{k: SD2_LDM[k] for k in ("linear_start", "linear_end", "timesteps", "scale_factor")})
    return model


HBM_PEAK_GBS = 8000.0              # MI355X HBM3E peak (MI355X_MICROARCH.md)


def family_profile(P, ops_list=None, passes=3, meta=None):
    """Per-op HIP-event timing of one evaluation of a planned network (UNetModel / Text2ImUNet plan `P`): the ops run in
    their real sequence on the launch stream (so weights stream from HBM, activations sit where the previous op left them),
    each bracketed by an event pair recorded on that stream.  Returns {kind: {ms, ops, launches, flops, bytes}} with the
    minimum over `passes` of each family's summed time.  kinds: gemm (implicit-GEMM conv / dense incl. their split-K
    reduce), attention, groupnorm, layernorm, small."""
    ops_list = P.main if ops_list is None else ops_list
    meta = P.meta[len(P.meta) - len(ops_list):] if meta is None else meta
    assert len(meta) == len(ops_list)
    best = None
    for _ in range(passes):
        evs = []
        for op in ops_list:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()          # torch's current stream == the stream libmdx launches on (ops._stream)
            op()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        fam = {}
        for (e0, e1), m in zip(evs, meta):
            f = fam.setdefault(m["kind"], {"ms": 0.0, "ops": 0, "launches": 0, "flops": 0, "bytes": 0})
            f["ms"] += e0.elapsed_time(e1)
            f["ops"] += 1
            f["launches"] += m["launches"]
            f["flops"] += m["flops"]
            if m["kind"] in ("groupnorm", "layernorm"):
                mm = dict(kv.split("=") for kv in m["info"].split() if "=" in kv)
                if {"B", "HW", "C"} <= set(mm):      # algorithmic traffic: read the fp16 tensor once, write it once
                    f["bytes"] += 2 * 2 * int(mm["B"]) * int(mm["HW"]) * int(mm["C"])
        if best is None:
            best = fam
        else:
            for k in fam:
                if fam[k]["ms"] < best[k]["ms"]:
                    best[k]["ms"] = fam[k]["ms"]
    return best


def family_fractions(fams, weight=1.0, acc=None):
    """Accumulate (weighted by evaluations per unit) family totals across plans."""
    acc = {} if acc is None else acc
    for k, f in fams.items():
        a = acc.setdefault(k, {"ms": 0.0, "launches": 0.0, "flops": 0.0, "bytes": 0.0})
        for key in a:
            a[key] += weight * f[key]
    return acc


def roofline_from_families(acc, evals_label):
    """The `roofline` object: dominant kernel family = gemm (MFMA-bound), plus per-family fractions: attention vs the
    MFMA peak, GroupNorm / LayerNorm vs the HBM peak (algorithmic bytes = one read + one write of the fp16 tensor)."""
    g = acc["gemm"]
    achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
    fam_out = {}
    for k, f in sorted(acc.items()):
        ent = {"ms": round(f["ms"], 4), "launches": round(f["launches"], 1)}
        if f["flops"] and k in ("gemm", "attention"):
            tf = f["flops"] / (f["ms"] * 1e-3) / 1e12
            ent.update(bound="mfma", tflops=round(tf, 1), frac=round(tf / MFMA_PEAK_TFLOPS, 4))
        elif f["bytes"]:
            gbs = f["bytes"] / (f["ms"] * 1e-3) / 1e9
            ent.update(bound="hbm", gbs=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
        fam_out[k] = ent
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None,
        "kernel": "gemm_kernel / conv3x3_halo_kernel (+ splitk_reduce) / st_head_kernel / st_tail_kernel: every implicit-GEMM conv3x3 / "
                  "conv1x1 / dense launch and every fused SpatialTransformer head / tail launch (chains of dense GEMMs) of "
                  + evals_label + ", HIP events per op on the launch stream, ops in their real sequence",
        "launches_per_unit_of_profile": round(g["launches"], 1),
        "avg_launch_us": round(g["ms"] * 1e3 / max(g["launches"], 1), 2),
        "algorithmic_gflop_per_launch": round(g["flops"] / max(g["launches"], 1) / 1e9, 3),
        "algorithmic_tflop_gemm_only": round(g["flops"] / 1e12, 4),
        # `achieved` is ALGORITHMIC: every launch is credited with the FLOPs of the op the reference performs.  The Upsample convs
        # (nearest-2x + 3x3, openaimodel.py:57-60) run as sub-pixel convs that execute 2.25x fewer multiply-adds than the gather
        # form they are credited with (four pre-summed 2x2 taps per output parity), as a Winograd conv would: two / three of the
        # ~160 launches, 10 % of the credited FLOPs at UNet batch 2 (0.136 of 1.353 TFLOP; executed: 0.060), i.e. the family figure is
        # 5.6 % above what the matrix pipe did -- stated here so that it is not read as matrix-pipe work.
        "flop_accounting": "algorithmic (reference op); sub-pixel Upsample convs credited with the gather form's FLOPs (2.25x what they execute)",
        "families": fam_out,
    }


def cpu_baseline(n_evals=3):
    """The oracle (fp32 PyTorch-CPU restatement -- the MindSpore reference cannot run here) timed on this host:
    one SDv2 UNet evaluation, B=1, 64x64 (BASELINE config 0 = 0.804 TFLOP).  One DDIM-50 + CFG latent = 100 such
    evaluations, so latents/s = 1 / (100 * t_eval)."""
    from oracle import ldm as O
    threads = torch.get_num_threads()
    params = O.init_params(O.SD2_UNET, seed=0)
    net = O.UNetOracle(O.SD2_UNET, params)
    x = np.random.RandomState(42).randn(1, 4, 64, 64).astype(np.float32)
    ctx = np.random.RandomState(1).randn(1, 77, 1024).astype(np.float32)
    ts = []
    for _ in range(n_evals + 1):
        t0 = time.time()
        net(x, torch.tensor([981.0]), ctx)
        ts.append(time.time() - t0)
    t_eval = float(np.median(ts[1:])) if len(ts) > 1 else ts[0]
    return {
        "value": round(1.0 / (100.0 * t_eval), 6), "unit": "latents/s", "cores": threads, "kind": "port",
        "sample": f"median of {n_evals} timed SDv2 UNet evals (B=1, 64x64 latent, fp32, oracle/ldm.py) after 1 warm-up: "
                  f"{t_eval:.2f} s/eval (all: {', '.join(f'{t:.2f}' for t in ts[1:])}); one 50-step DDIM+CFG latent = "
                  f"100 evals (extrapolated)",
        "host_cpu_count": os.cpu_count(),
    }


PMC_FILE = "r06_pmc_traffic.json"
PMC_MFMA_FILE = "r06_pmc_mfma.json"      # the headline config; the other LDM configs: r06_pmc_mfma_<config>.json


def pmc_traffic(gemm_launches_now):
    """L2<->fabric bytes per GEMM-family launch (implicit-GEMM kernels + their split-K reduces) from the rocprofv3 PMC passes
    committed under profiles/ (tools/pmc_traffic.py: FETCH_SIZE x2-corrected + WRITE_SIZE; PMC counters cannot be read from
    inside this process, and `--pmc` may not be combined with the timing run).  The passes are taken on THIS bench command;
    the file is only used when its GEMM-family launch count per evaluation equals the plan's launch count in this process --
    otherwise it describes another launch mix and `traffic` is null.  Returns (bytes per launch | None, note)."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            doc = json.load(f)
        fam = doc["families"]
        g, r = fam["gemm"], fam.get("splitk_reduce", {"launches_per_eval": 0, "read_MB_per_eval": 0, "write_MB_per_eval": 0})
        n = g["launches_per_eval"] + r["launches_per_eval"]
        if abs(n - gemm_launches_now) > 0.01 * gemm_launches_now:    # (the trace averages over evaluations: not an integer)
            return None, (f"profiles/{PMC_FILE} holds {n:.0f} GEMM-family launches per evaluation, this build issues "
                          f"{gemm_launches_now}: stale PMC passes, not reported")
        total = (g["read_MB_per_eval"] + g["write_MB_per_eval"] + r["read_MB_per_eval"] + r["write_MB_per_eval"]) * 1e6
        note = (f"L2<->fabric bytes per GEMM-family launch (MALL hits included): rocprofv3 --pmc FETCH_SIZE (x2, gfx950 "
                f"correction) and WRITE_SIZE, separate passes of `{doc.get('command', '?')}` ({n:.0f} launches per evaluation "
                f"= this run's); whole evaluation {doc.get('total_read_MB_per_eval')} MB read + "
                f"{doc.get('total_write_MB_per_eval')} MB written vs 4.45 GB algorithmic (1.73 GB weights + 2 x 1.36 GB "
                f"activations); profiles/{PMC_FILE}")
        return int(total / n), note
    except Exception as e:      # no committed passes for this build
        return None, f"no PMC passes for this build ({type(e).__name__})"


def pmc_mfma_busy(roof, gemm_launches_now, PMC_MFMA_FILE=PMC_MFMA_FILE):
    """Matrix-pipe occupancy per kernel family of THIS bench command, from the committed rocprofv3 PMC pass
    (tools/pmc_mfma.py: SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32 SIMDs per shader engine)).
    PMC counters cannot be read from inside the timed process; the file is used only when its GEMM-family launch count per
    evaluation equals this run's (same launch mix), else `mfma_busy` stays absent and the reason is given."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_MFMA_FILE)) as f:
            doc = json.load(f)
        n = doc["families"]["gemm"]["launches_per_eval"] + doc["families"].get("splitk_reduce", {}).get("launches_per_eval", 0)
        if abs(n - gemm_launches_now) > 0.01 * gemm_launches_now:
            roof["mfma_busy_note"] = (f"profiles/{PMC_MFMA_FILE} holds {n:.0f} GEMM-family launches per evaluation, this build "
                                      f"issues {gemm_launches_now}: stale PMC pass, not reported")
            return
        for fam, ent in roof["families"].items():
            src = doc["families"].get(fam)
            if src is not None and src.get("mfma_busy") is not None:
                ent["mfma_busy"] = src["mfma_busy"]
        g = doc["families"]["gemm"]
        roof["mfma_busy"] = g.get("mfma_busy")
        roof["mfma_busy_note"] = (f"matrix-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32 SIMDs per SE) per "
                                  f"family, rocprofv3 --pmc pass of `{doc.get('command', '?')}`; profiles/{PMC_MFMA_FILE}")
    except Exception as e:
        roof["mfma_busy_note"] = f"no MFMA-busy PMC pass for this build ({type(e).__name__})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="sd2_512", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="headline line only (the default N = 1 run also times BASELINE configs[2..4], see OTHER_CONFIGS)")
    args = ap.parse_args()

    from minddiffusion_amd import distributed as D

    rank, world, local_rank = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if os.environ.get("MDX_BENCH_SHARE_GPU") == "1":     # tests only: every rank on device 0 (with MDX_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    result = run_config(args.config, args, rank, world, device, args.steps, args.warmup,
                        cpu=(world == 1 and not args.no_cpu_baseline and args.config == "sd2_512"))
    if rank == 0 and world == 1 and args.config == "sd2_512" and not args.no_other_configs:
        # The other BASELINE configs' per-GPU shares, timed by the SAME command the driver runs: >= 3 timed
        # units each after one warm-up; the headline fields above are untouched.
        result["other_configs"] = {}
        for name in OTHER_CONFIGS:
            torch.cuda.empty_cache()
            r = run_config(name, args, rank, world, device, 3, 1, cpu=False)
            result["other_configs"][name] = {
                "metric": r["metric"], "value": r["value"], "unit": r["unit"], "steps": r["steps"], "warmup": r["warmup"],
                "ms_per_step": r["ms_per_step"], "per_unet_step_ms": r.get("per_unet_step_ms"),
                "workload": r["config"]["workload"], "global_batch": r["config"]["global_batch"],
                "hip_graph": r["config"]["hip_graph"], "cfg_dup_prefix": r["config"].get("cfg_dup_prefix"),
                "families": r["roofline"]["families"],
                "gemm_family_tflops": r["roofline"]["achieved"], "whole_path": r["roofline"]["whole_path"]}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


OTHER_CONFIGS = ("wukong_512_plms", "sd2_768", "glide_256")


def run_config(config, args, rank, world, device, steps, warmup, cpu):
    """Build the models of one CONFIGS entry, time `steps` units after `warmup`, profile the plan's kernel families.  Returns
    the result dict on rank 0 (None elsewhere)."""
    from minddiffusion_amd import distributed as D
    from minddiffusion_amd.pipeline import DiffusionPipeline
    cfg = CONFIGS[config]
    batch = cfg["batch"]
    Bg = batch * world

    if cfg["family"] == "ldm":
        model = build_model(device, cfg["unet"])
        if args.no_graph:
            model.unet.use_graph = False
        if cfg.get("vae"):
            model.first_stage_model = build_vae(device)
        if cfg.get("text"):
            model.cond_stage_model = build_text_encoder(device)
        pipe = DiffusionPipeline(model, sampler=cfg["sampler"], device=device)
        h = w = cfg["latent"]
        # synthetic prompts: N(0,1) text embeddings [B,77,ctx] (seed 1), one unconditional row (seed 2), x_T seed 42
        c = uc = x_T = None
        if rank == 0:
            rs = np.random.RandomState
            c = torch.from_numpy(rs(1).randn(Bg, 77, cfg["ctx_dim"]).astype(np.float32)).to(device, torch.float16)
            uc = torch.from_numpy(rs(2).randn(1, 77, cfg["ctx_dim"]).astype(np.float32)).to(device, torch.float16)
            x_T = torch.from_numpy(rs(42).randn(Bg, 4, h, w).astype(np.float32)).to(device)

        prompts = ["a photograph of an astronaut riding a horse"] * Bg

        if cfg.get("inpaint"):
            # inpaint.py:65-106: dict conditioning {c_concat: cat(resized mask, masked-image latent), c_crossattn: text}, the
            # same c_concat on the unconditional half, x0 = the masked-image latent, no mask blend; every rank runs its own
            # `batch` images (synthetic, seeded per rank: the hybrid conditioning is not part of the txt2img broadcast)
            from minddiffusion_amd.ldm.models.diffusion.plms import PLMSSampler
            rs = np.random.RandomState(100 + rank)
            td = lambda a, dt=torch.float32: torch.from_numpy(a.astype(np.float32)).to(device, dt)
            ic = td(rs.randn(batch, 77, cfg["ctx_dim"]), torch.float16)
            iuc = td(np.repeat(rs.randn(1, 77, cfg["ctx_dim"]), batch, 0), torch.float16)
            ix = td(rs.randn(batch, 4, h, w))
            icat = td(np.concatenate([(rs.rand(batch, 1, h, w) > 0.5), rs.randn(batch, 4, h, w)], 1))
            isampler = PLMSSampler(model)

        def one_step():
            if cfg.get("inpaint"):
                return isampler.sample(cfg["steps"], batch, (4, h, w), conditioning={"c_concat": icat, "c_crossattn": ic},
                                       x_T=ix, unconditional_guidance_scale=cfg["scale"],
                                       unconditional_conditioning={"c_concat": icat, "c_crossattn": iuc}, x0=icat[:, 1:],
                                       verbose=False)[0]
            if cfg.get("text"):     # rank 0 encodes [prompts; empty prompts]; the pipeline broadcasts the embeddings
                return pipe(prompts=prompts, x_T=x_T, H=8 * h, W=8 * w, steps=cfg["steps"], scale=cfg["scale"], eta=0.0,
                            decode=True)
            return pipe(c=c, uc=uc, x_T=x_T, H=8 * h, W=8 * w, steps=cfg["steps"], scale=cfg["scale"], eta=0.0,
                        decode=bool(cfg.get("vae")), batch_size=Bg)
    else:
        # Taichu-GLIDE/src/txt2img.py:113-126 through the library's sharded pipeline: rank 0 holds the prompts of the
        # global batch; ONE packed broadcast carries them, the per-step unconditional token ids (identical on every rank,
        # SURVEY 8(e)) and the noise seed; each rank then runs base (60 steps, CFG) + up-sampler (27 steps) on its shard
        from minddiffusion_amd.glide.pipeline import GlidePipeline
        dm, sr = build_glide(device)
        if args.no_graph:
            dm.model.use_graph = sr.model.use_graph = False
        gp = GlidePipeline(dm, sr, text_ctx=128, vocab_len=50001)
        tok = msk = None
        if rank == 0:
            tok = np.random.RandomState(1).randint(1, 50000, (Bg, 128)).astype(np.int32)
            msk = np.ones((Bg, 128), np.int32)

        def one_step():
            return gp(tokens=tok, mask=msk, seed=42)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        out = one_step()
    barrier()
    D.reset_collective_stats()
    D.time_collectives = True       # (opt-in: the library does not time or synchronise its broadcast on its own)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_step()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0          # this rank's own clock (before the closing barrier): per-rank rate below
    barrier()
    elapsed = time.perf_counter() - t0
    dist_info = {"dist_backend": None, "rccl_world_size": 1, "broadcasts_per_step": 0, "broadcast_bytes": 0,
                 "broadcast_ms": 0.0, "per_rank_units_per_s": None}
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
        # self-proving N > 1 record: the backend torch.distributed actually runs ("nccl" IS RCCL on ROCm), its world size, what
        # the ONE collective per step moved and how long it took (HIP events around it, outside the trajectory), and the
        # slowest / fastest rank's own rate
        rates = torch.tensor([batch * steps / own], device=device, dtype=torch.float64)
        lo, hi = rates.clone(), rates.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        cs = D.collective_stats
        dist_info = {"dist_backend": torch.distributed.get_backend(), "rccl_world_size": torch.distributed.get_world_size(),
                     "broadcasts_per_step": cs["broadcasts"] / steps, "broadcast_bytes": cs["bytes"] // max(cs["broadcasts"], 1),
                     "broadcast_ms": round(cs["ms"] / max(cs["broadcasts"], 1), 4),
                     "per_rank_units_per_s": {"min": round(float(lo.item()), 4), "max": round(float(hi.item()), 4)}}
    assert torch.isfinite(out).all()

    ms_per_step = elapsed / steps * 1e3
    units_per_s = Bg * steps / elapsed
    if rank == 0:
        result = {
            "metric": cfg["metric"], "value": round(units_per_s, 4), "unit": cfg["unit"], "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": cfg["workload"] + "; synthetic seeded weights + synthetic text conditioning",
                       "name": config, "global_batch": Bg, "parallelism": f"batch-shard x{world}",
                       "hip_graph": None, **dist_info},
        }
        if cfg["family"] == "ldm":
            # per-UNet-step ms: HIP events around apply_model (CFG batch = 2 x per-GPU batch), median of 20 warm calls
            nb = 2 * batch
            ctx = torch.randn(nb, 77, cfg["ctx_dim"], device=device, dtype=torch.float16)
            xs = torch.randn(nb, model.unet.in_channels, h, w, device=device)     # (9 channels for the inpainting UNet)
            tsv = torch.full((nb,), 501.0, device=device)
            # (round 6) what the sampler's guidance call ran: the guidance-duplicate prefix from ops option unet_cfg_dup rows on
            dupkw = {"cfg_dup": True} if getattr(model.unet._plan(nb, h, w), "dup_graph", None) is not None else {}
            if dupkw:
                xs[batch:] = xs[:batch]
            for _ in range(3):
                model.apply_model_nhwc(xs, tsv, ctx, **dupkw)
            evs = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                model.apply_model_nhwc(xs, tsv, ctx, **dupkw)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            result["per_unet_step_ms"] = round(float(np.median([a.elapsed_time(b) for a, b in evs])), 3)
            result["config"].update(ddim_steps=cfg["steps"], cfg_scale=cfg["scale"], unet_batch_per_gpu=nb,
                                    sampler=cfg["sampler"])
            from minddiffusion_amd.ldm.modules.diffusionmodules.util import make_ddim_timesteps
            # (the reference's uniform grid has range(0, 1000, 1000 // S) points: 50 for S = 50, 31 for S = 30)
            n_grid = len(make_ddim_timesteps("uniform", cfg["steps"], 1000, verbose=False)) if cfg["sampler"] != "dpm_solver" else cfg["steps"]
            n_evals = n_grid + (1 if cfg["sampler"] == "plms" else 0)
            tflop_per_unit = cfg["tflop_per_row"] * 2 * n_evals + cfg.get("vae_tflop", 0.0)   # CFG doubles the rows
            Pl = model.unet._plan(nb, h, w)
            # what actually ran in the timed region: a captured hipGraph replay, or (capture failed / --no-graph) eager launches
            result["config"]["hip_graph"] = Pl.graph is not None or getattr(Pl, "dup_graph", None) is not None
            # (round 6) a guidance batch of >= ops option unet_cfg_dup rows runs conv_in .. the first self-attention on one half
            # (UNetModel._dup_body): profile the op list the sampler's graph was captured from
            dup = getattr(Pl, "dup_graph", None) is not None
            result["config"]["cfg_dup_prefix"] = dup
            fams = family_profile(Pl, Pl.dup_body, meta=Pl.dup_meta) if dup else family_profile(Pl, Pl.main[Pl.temb_ops:])
            roof = roofline_from_families(family_fractions(fams), f"one UNet evaluation at batch {nb}")
            gemm_launches = fams["gemm"]["launches"]
            if cfg.get("vae"):   # VAE decode alone: HIP events around AutoencoderKL.decode, median of 10 warm calls
                zz = torch.randn(batch, 4, h, w, device=device)
                for _ in range(2):
                    model.first_stage_model.decode(zz)
                evs = []
                for _ in range(10):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    model.first_stage_model.decode(zz)
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                vms = float(np.median([a.elapsed_time(b) for a, b in evs]))
                if cfg.get("text"):
                    evs = []
                    for _ in range(12):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        model.get_learned_conditioning(prompts)
                        e1.record()
                        evs.append((e0, e1))
                    torch.cuda.synchronize()
                    result["text_encode_ms"] = round(float(np.median([a.elapsed_time(b) for a, b in evs[2:]])), 3)
                result["vae_decode_ms"] = round(vms, 3)
                result["vae_decode_tflops"] = round(cfg["vae_tflop"] * batch / vms * 1e3, 1)
            if config == "sd2_512":
                roof["traffic"], roof["traffic_note"] = pmc_traffic(gemm_launches)
                pmc_mfma_busy(roof, gemm_launches)
            elif os.path.exists(os.path.join(ROOT, "profiles", f"r06_pmc_mfma_{config}.json")):
                pmc_mfma_busy(roof, gemm_launches, f"r06_pmc_mfma_{config}.json")
        else:
            # Taichu-GLIDE: one image = 60 guided base evaluations (UNet batch 2P) + 27 super-resolution evaluations (batch P);
            # profile both plans and weight them by their evaluation counts
            tflop_per_unit = cfg["tflop_per_image"]
            Pb, Ps = dm.model._plan(2 * batch, 64, 64), sr.model._plan(batch, 256, 256)
            result["config"]["hip_graph"] = Pb.graph is not None and Ps.graph is not None
            # (round 6) a step runs the plan's BODY only: the text transformer, the encoder_kv projections and the time-embedding
            # chain of all steps run once per loop (Text2ImUNet.begin_loop) and are timed below as `loop_prefix_ms`, not folded
            # into the per-family figures
            from minddiffusion_amd.glide import diffusion_creator as _DC
            tables = _DC._LOOP_TABLES
            acc = family_fractions(family_profile(Pb, Pb.main[Pb.n_emb:] if tables else None), float(dm.num_timesteps))
            acc = family_fractions(family_profile(Ps, Ps.main[Ps.n_emb:] if tables else None), float(sr.num_timesteps), acc)
            roof = roofline_from_families(acc, f"{dm.num_timesteps} base evaluations at batch {2 * batch} + "
                                               f"{sr.num_timesteps} super-resolution evaluations at batch {batch}")
            if tables:
                tk = np.random.RandomState(1).randint(1, 50000, (2 * batch, 128)).astype(np.int32)
                mk = np.ones((2 * batch, 128), np.int32)
                un = np.random.RandomState(2).randint(1, 50000, (dm.num_timesteps, 128)).astype(np.int32)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                for _ in range(2):      # second pass timed (the first builds the table plans)
                    ev[0].record()
                    dm.begin_loop(tk, mk, un)
                    ev[1].record()
                    sr.begin_loop(tk[:batch], mk[:batch])
                    ev[2].record()
                torch.cuda.synchronize()
                dm.end_loop(); sr.end_loop()
                result["config"]["loop_prefix_ms"] = {"base": round(ev[0].elapsed_time(ev[1]), 3), "super_res": round(ev[1].elapsed_time(ev[2]), 3),
                                                      "note": "once per loop (text transformer on all prompts incl. the 60 unconditional ones, "
                                                              "encoder_kv tables, emb table of all steps); inside the timed region, outside `families`"}
        whole = units_per_s / world * tflop_per_unit
        roof["whole_path"] = {"achieved": round(whole, 2), "frac": round(whole / MFMA_PEAK_TFLOPS, 4),
                              "algorithmic_tflop_per_unit": tflop_per_unit,
                              "note": "units/s/GPU x SURVEY 8(d) TFLOP per unit (all kernels, launch gaps included)"}
        if roof.get("achieved") is None:
            roof["achieved"], roof["frac"] = roof["whole_path"]["achieved"], roof["whole_path"]["frac"]
        result["roofline"] = roof
        result["cpu_baseline"] = cpu_baseline() if cpu else None
        return result
    return None


if __name__ == "__main__":
    main()
