#!/usr/bin/env python
"""Screening sweep (round 5): every row of a UNet evaluation at batch B must equal its own batch-1 evaluation up to fp16 rounding
(<= 4e-3), for batches and latent sizes the benchmarks and the parity tests do NOT run.  The tile table is keyed by
(M, N, K, ksize); this is the check that a row measured at one (B, H, W) is not wrong at another factorization of M.

    python tools/shape_sweep.py --model sd2 --latents 32,64,96 --batches 1-16
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sweep_glide(args):
    """The two Taichu-GLIDE UNets (base at --latents pixels, default 64; up-sampler at 4x that with a low-res input of the base size):
    every row of batch B against its own batch-1 evaluation."""
    from bench import build_glide
    dev = "cuda:0"
    dm, sr = build_glide(dev)
    lo, _, hi = args.batches.partition("-")
    batches = list(range(int(lo), int(hi or lo) + 1))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    bad = 0
    for hw in [int(v) for v in args.latents.split(",")]:
        for name, net, size, low in (("glide-base", dm.model, hw, None), ("glide-up", sr.model, 4 * hw, sr.model.low_size)):
            net.use_graph = False
            rng = np.random.RandomState(size)
            Bm = max(batches)
            x = torch.tensor(rng.randn(Bm, 3, size, size).astype(np.float32), device=dev)
            tok = torch.tensor(rng.randint(1, 50000, (Bm, 128)).astype(np.int32), device=dev)
            msk = torch.ones((Bm, 128), dtype=torch.int32, device=dev)
            lr = torch.tensor(rng.randn(Bm, 3, low, low).astype(np.float32), device=dev) if low else None
            t = torch.full((Bm,), 500.0, device=dev)
            kw = lambda a, b: dict(low_res=lr[a:b].clone()) if low else {}
            ones = [net.forward_nhwc(x[r:r + 1].clone(), t[:1], tok[r:r + 1], msk[r:r + 1], **kw(r, r + 1)).clone() for r in range(Bm)]
            for B in batches:
                if B == 1:
                    continue
                full = net.forward_nhwc(x[:B].clone(), t[:B], tok[:B], msk[:B], **kw(0, B))
                errs = [rel(full[r:r + 1], ones[r]) for r in range(B)]
                flag = "" if max(errs) <= args.tol else "   <-- MISMATCH rows " + str([r for r, e in enumerate(errs) if e > args.tol])
                bad += bool(flag)
                print(f"{name} {size:3d} px B {B:2d}  worst {max(errs):.2e}{flag}", flush=True)
                net._plans.pop((B, size, size), None)
            net._plans.clear()
            torch.cuda.empty_cache()
    print("mismatching (size, batch) pairs:", bad)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="sd2")
    ap.add_argument("--latents", default="32,48,64,96")
    ap.add_argument("--batches", default="2-16")
    ap.add_argument("--tol", type=float, default=4e-3)
    args = ap.parse_args()
    if args.model == "glide":
        return sweep_glide(args)
    from minddiffusion_amd.configs import SD2_UNET, WUKONG_UNET
    from minddiffusion_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel
    from minddiffusion_amd.weights import synthetic_unet_params_device
    dev = "cuda:0"
    cfg, cd = (SD2_UNET, 1024) if args.model == "sd2" else (WUKONG_UNET, 768)
    net = UNetModel(**dict(cfg))
    net.use_graph = False
    net.load_state_dict(synthetic_unet_params_device(net.parameter_shapes(), seed=0, device=dev))
    lo, _, hi = args.batches.partition("-")
    batches = list(range(int(lo), int(hi or lo) + 1))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    bad = 0
    for hw in [int(v) for v in args.latents.split(",")]:
        rng = np.random.RandomState(hw)
        Bm = max(batches)
        x = torch.tensor(rng.randn(Bm, 4, hw, hw).astype(np.float32), device=dev)
        ctx = torch.tensor(rng.randn(Bm, 77, cd).astype(np.float32), device=dev)
        t = torch.full((Bm,), 500.0, device=dev)
        ones = [net(x[r:r + 1].clone(), t[:1], ctx[r:r + 1].clone()).clone() for r in range(Bm)]
        for B in batches:
            if B == 1:
                continue
            full = net(x[:B].clone(), t[:B], ctx[:B].clone())
            errs = [rel(full[r:r + 1], ones[r]) for r in range(B)]
            flag = "" if max(errs) <= args.tol else "   <-- MISMATCH rows " + str([r for r, e in enumerate(errs) if e > args.tol])
            bad += bool(flag)
            print(f"{args.model} latent {hw:3d} B {B:2d}  worst {max(errs):.2e}{flag}", flush=True)
            net._plans.pop((B, hw, hw), None)      # plans hold their activations: drop them as we go
        net._plans.clear()
        torch.cuda.empty_cache()
    print("mismatching (latent, batch) pairs:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
