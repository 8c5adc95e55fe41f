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


def build_model(device, cfg_name="sd2"):
    from minddiffusion_amd.configs import SD2_LDM, SD2_UNET, WUKONG_UNET
    from minddiffusion_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_device
    ucfg = dict(SD2_UNET if cfg_name == "sd2" else WUKONG_UNET)
    net = UNetModel(device=device, **ucfg)
    net.load_state_dict(synthetic_unet_params_device(net.parameter_shapes(), seed=0, device=device))
    torch.cuda.synchronize()
    model = LatentDiffusion(net, **{k: SD2_LDM[k] for k in ("linear_start", "linear_end", "timesteps", "scale_factor")})
    return model


def dominant_kernel_roofline(model, B, h, w, ctx, passes=3):
    """Per-launch HIP-event timing of the dominant kernel class (gemm_kernel: every implicit-GEMM conv / dense
    launch of one UNet evaluation, in their real sequence so weights stream from HBM, not from a warm cache)."""
    net = model.unet
    P = net._plan(B, h, w)
    x = torch.randn(B, 4, h, w, device=net.device)
    t = torch.full((B,), 981.0, device=net.device)
    net._ensure_context(P, ctx)
    P.x_static.copy_(x)
    P.t_static.copy_(t)
    idx = [i for i, m in enumerate(P.meta) if m["kind"] == "gemm"]
    flops = sum(P.meta[i]["flops"] for i in idx)
    times = []
    for _ in range(passes):
        evs = {}
        for i, op in enumerate(P.main):
            if i in evs or P.meta[i]["kind"] != "gemm":
                op()
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()          # recorded on torch's current stream == the stream the kernels are launched on
            op()
            e1.record()
            evs[i] = (e0, e1)
        torch.cuda.synchronize()
        times.append(sum(a.elapsed_time(b) for a, b in evs.values()))  # ms
    t_ms = min(times)
    launches = sum(P.meta[i]["launches"] for i in idx)
    achieved = flops / (t_ms * 1e-3) / 1e12
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None,
        "kernel": "gemm_kernel (implicit-GEMM conv3x3/conv1x1/dense; all launches of one UNet eval, in sequence)",
        "launches_per_unet_eval": launches,
        "avg_launch_us": round(t_ms * 1e3 / max(launches, 1), 2),
        "algorithmic_gflop_per_launch": round(flops / max(launches, 1) / 1e9, 3),
        "algorithmic_tflop_per_unet_eval_gemm_only": round(flops / 1e12, 4),
    }


def cpu_baseline(n_evals=1):
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
    t_eval = min(ts[1:]) if len(ts) > 1 else ts[0]
    return {
        "value": round(1.0 / (100.0 * t_eval), 6), "unit": "latents/s", "cores": threads, "kind": "port",
        "sample": f"{n_evals} timed SDv2 UNet eval(s) (B=1, 64x64 latent, fp32, oracle/ldm.py) after 1 warm-up: "
                  f"{t_eval:.2f} s/eval; one 50-step DDIM+CFG latent = 100 evals (extrapolated)",
        "host_cpu_count": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--scale", type=float, default=9.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    from minddiffusion_amd import distributed as D
    from minddiffusion_amd.pipeline import DiffusionPipeline

    rank, world, local_rank = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    model = build_model(device)
    if args.no_graph:
        model.unet.use_graph = False
    pipe = DiffusionPipeline(model, sampler="ddim", device=device)
    Bg = args.batch * world
    h = w = 64
    # synthetic prompts: N(0,1) text embeddings [B,77,1024] (seed 1), one unconditional row (seed 2), x_T seed 42
    c = uc = x_T = None
    if rank == 0:
        c = torch.from_numpy(np.random.RandomState(1).randn(Bg, 77, 1024).astype(np.float32)).to(device, torch.float16)
        uc = torch.from_numpy(np.random.RandomState(2).randn(1, 77, 1024).astype(np.float32)).to(device, torch.float16)
        x_T = torch.from_numpy(np.random.RandomState(42).randn(Bg, 4, h, w).astype(np.float32)).to(device)

    def one_step():
        return pipe(c=c, uc=uc, x_T=x_T, H=8 * h, W=8 * w, steps=args.ddim_steps, scale=args.scale, eta=0.0)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all()

    ms_per_step = elapsed / args.steps * 1e3
    latents_per_s = Bg * args.steps / elapsed
    result = None
    if rank == 0:
        # per-UNet-step ms: HIP events around apply_model (CFG batch = 2 x per-GPU batch), median of 20 warm calls
        net = model.unet
        nb = 2 * args.batch
        ctx = torch.randn(nb, 77, 1024, device=device, dtype=torch.float16)
        xs = torch.randn(nb, 4, h, w, device=device)
        tsv = torch.full((nb,), 501.0, device=device)
        for _ in range(3):
            model.apply_model_nhwc(xs, tsv, ctx)
        evs = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.apply_model_nhwc(xs, tsv, ctx)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        unet_ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
        tflop_per_latent = TFLOP_PER_UNET_ROW_64 * 2 * args.ddim_steps   # CFG doubles the rows
        whole = latents_per_s / world * tflop_per_latent
        roof = dominant_kernel_roofline(model, nb, h, w, ctx)
        roof["whole_path"] = {"achieved": round(whole, 2), "frac": round(whole / MFMA_PEAK_TFLOPS, 4),
                              "algorithmic_tflop_per_latent": tflop_per_latent,
                              "note": "latents/s/GPU x SURVEY 8(d) TFLOP-per-latent (all kernels, launch gaps included)"}
        result = {
            "metric": "512x512 txt2img latents/sec (50-step DDIM)", "value": round(latents_per_s, 4),
            "unit": "latents/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "SDv2 txt2img 512x512 (64x64 latent), 50-step DDIM, CFG 9.0, batch 1 per GPU "
                                   "(BASELINE.json configs[1]); synthetic seeded weights + N(0,1) text embeddings",
                       "global_batch": Bg, "ddim_steps": args.ddim_steps, "cfg_scale": args.scale,
                       "unet_batch_per_gpu": nb, "parallelism": f"batch-shard x{world}",
                       "hip_graph": bool(net.use_graph)},
            "per_unet_step_ms": round(unet_ms, 3),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
