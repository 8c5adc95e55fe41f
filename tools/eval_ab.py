#!/usr/bin/env python
"""Same-box A/B of one UNet evaluation (hipGraph replay) under two option sets, interleaved in one process (cdna guide 5.4 rule 24):
the boxes of the pool differ by +-3-5 % for one build, so whole-evaluation claims need both arms on one box.

    python tools/eval_ab.py --model wukong --batch 16 --latent 64 --arms "base:gemm_conv8p=0,unet_subpixel_upsample=0" "new:"

An arm is  name:opt=value,opt=value  (library options of include/mdx.h or planner options of ops._OPTIONS; options an arm does not
name keep their defaults).  Every arm builds its own network + plan under its options (the library options are set again before each
timed burst, because launch forms are resolved per call when a graph is captured -- the captured graph then holds them).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="sd2")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--arms", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--guidance", action="store_true", help="the batch is [uncond ; cond] of the same latents and the call says so "
                    "(forward_nhwc(cfg_dup=True)): arms can then differ in unet_cfg_dup")
    args = ap.parse_args()
    from bench import build_model
    from minddiffusion_amd import ops
    dev = torch.device("cuda:0")
    arms = []
    for spec in args.arms:
        name, _, rest = spec.partition(":")
        opts = {}
        for kv in [x for x in rest.split(",") if x]:
            k, v = kv.split("=")
            opts[k] = int(v)
        arms.append((name, opts))
    allopts = sorted({k for _, o in arms for k in o})
    defaults = {k: ops.get_option(k) for k in allopts}
    B, h = args.batch, args.latent
    plans = {}
    for name, opts in arms:
        for k in allopts:
            ops.set_option(k, opts.get(k, defaults[k]))
        model = build_model(dev, args.model)
        net = model.unet
        net.use_graph = True
        x = torch.randn(B, 4, h, h, device=dev)
        ctx = torch.randn(B, 77, net.context_dim, device=dev, dtype=torch.float16)
        t = torch.full((B,), 501.0, device=dev)
        if args.guidance:
            x[B // 2:] = x[:B // 2]
        net.forward_nhwc(x, t, ctx, cfg_dup=args.guidance)
        P = net._plans[(B, h, h)]
        g = P.dup_graph if (args.guidance and P.dup_graph is not None) else P.graph
        assert g is not None
        plans[name] = (net, P, g, len(P.dup_body) if g is P.dup_graph else len(P.main))
        torch.cuda.synchronize()
    for k in allopts:
        ops.set_option(k, defaults[k])
    times = {name: [] for name, _ in arms}
    for r in range(args.rounds):
        for name, _ in arms:
            g = plans[name][2]
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / args.iters)
    res = dict(model=args.model, batch=B, latent=h, arms={})
    base = None
    for name, opts in arms:
        ms = sorted(times[name])
        med = ms[len(ms) // 2]
        base = base or med
        res["arms"][name] = dict(options=opts, ms_per_eval_median=round(med, 4), ms_per_eval_min=round(ms[0], 4),
                                 all=[round(x, 4) for x in times[name]], launches=plans[name][3])
        print(f"{args.model} B={B} latent={h}  {name:12s} {med:8.4f} ms / evaluation (min {ms[0]:.4f})  {100 * (med / base - 1):+.2f} %  "
              f"ops {plans[name][3]}  {opts}", flush=True)
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
